#!/usr/bin/env python
"""Phase breakdown of the round-5 kNN body (knn16_body): wave 0 of every workgroup accumulates the 100 MHz wall clock per
phase.  Needs scratch/libvcr_probe.so (python profiles/experiments/probe_build.py)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "scratch", "libvcr_probe.so")
WAVES = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def main():
    import torch
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import native
    native.LIB_PATH = LIB
    L = native.lib()
    L.vcr_dbg_probe_knn.argtypes = [C.c_void_p, C.c_int]
    L.vcr_dbg_probe_knn.restype = C.c_int
    g = torch.Generator().manual_seed(0)
    full = np.zeros((4096, 32), np.uint64)
    names = ["chains+selection", "slow path", "final feat", "row wait (!xt)", "final xyz", "#slow feat x100", "#slow xyz x100", "7"]
    for B, N, k in ((32, 1024, 20), (64, 4096, 40)):
        f = torch.randn(B, N, 64, generator=g).cuda()
        sq = (f ** 2).sum(-1).contiguous()
        ft = f.view(B, N, 4, 4, 4).transpose(3, 4).reshape(B, N, 64).contiguous()
        xyz = torch.rand(B, N, 3, generator=g) - 0.5
        xyz4 = torch.cat((xyz, (xyz ** 2).sum(-1, keepdim=True)), -1).cuda().contiguous()
        def pair():
            import ctypes
            args, keep = [], []
            for x, s_, Cc in ((f, sq, 64), (xyz4, None, 4)):
                idx = torch.empty(B, N, k, dtype=torch.int32, device="cuda")
                t_ = torch.zeros(1 + B * N, dtype=torch.int32, device="cuda")
                args.append(native.KnnArgs(native.ptr(x), x.stride(1), native.ptr(s_), B, N, Cc, k, native.ptr(idx), native.ptr(t_), B * N, WAVES))
                keep.append((idx, t_))
            args[0].xt = native.ptr(ft)
            L.vcr_knn_pair_f32.argtypes = [ctypes.POINTER(native.KnnArgs), ctypes.POINTER(native.KnnArgs), ctypes.c_void_p]
            L.vcr_knn_pair_f32.restype = ctypes.c_int
            native.check(L.vcr_knn_pair_f32(ctypes.byref(args[0]), ctypes.byref(args[1]), ctypes.c_void_p(native.stream_ptr())), "pair")
            return keep
        for name, x, s, xt in (("feat64", f, sq, ft), ("xyz", xyz4, None, None), ("pair", None, None, None)):
            run = pair if name == "pair" else (lambda: native.knn(x, s, k, exact_ties=True, waves=WAVES, xt=xt))
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            L.vcr_dbg_probe_knn(None, 1)
            run()
            torch.cuda.synchronize()
            L.vcr_dbg_probe_knn(full.ctypes.data, 0)
            t = full[:, :8].astype(np.float64) * 0.01
            used = t.sum(1) > 0
            med = np.median(t[used], 0)
            tot = t[used][:, :5].sum(1)
            cnt = np.median(full[:, 5:7][used].astype(np.float64), 0)
            print(f"B={B} N={N} k={k} {name} waves={WAVES}: {used.sum()} workgroups; median us per wave: " +
                  ", ".join(f"{n} {v:.1f}" for n, v in zip(names[:5], med[:5])) + f"  | total {med[:5].sum():.1f} "
                  f"(min {tot.min():.1f} max {tot.max():.1f}) slow calls feat {cnt[0]:.0f} xyz {cnt[1]:.0f}", flush=True)


if __name__ == "__main__":
    main()
