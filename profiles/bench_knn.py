#!/usr/bin/env python
"""Timing of vcr_knn_f32 (feature-space C=64 and Cartesian C=4) at BASELINE configs[1] (32 clouds of 1024, k=20) and
configs[4] (64 clouds of 4096, k=40), for the candidate-split choices, the 16-query-wave kernel (waves = 8), with / without
the tie replay, and of the one-launch pair (vcr_knn_pair_f32) with either feature-space kernel.
Run on the GPU box:  python profiles/bench_knn.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vcrnet_amd  # noqa: E402,F401
from vcrnet_amd import native  # noqa: E402


def bench(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    for B, N, k in ((32, 1024, 20), (64, 4096, 40), (48, 768, 20), (32, 2048, 20)):
        f = torch.randn(B, N, 64, generator=g).cuda()
        sq = (f ** 2).sum(-1).contiguous()
        xyz = torch.rand(B, N, 3, generator=g) - 0.5
        xyz4 = torch.cat((xyz, (xyz ** 2).sum(-1, keepdim=True)), -1).cuda().contiguous()
        for name, x, s in (("feat64", f, sq), ("xyz", xyz4, None)):
            row = []
            for waves in (0, 1, 2, 4) + ((8,) if name == "feat64" else ()):
                row.append(bench(lambda: native.knn(x, s, k, exact_ties=False, waves=waves)))
            ties = bench(lambda: native.knn(x, s, k, exact_ties=True))
            print(f"B={B:3d} N={N:5d} k={k:2d} {name:7s}: auto {row[0]:8.1f} us | S=1 {row[1]:8.1f}  S=2 {row[2]:8.1f}  "
                  f"S=4 {row[3]:8.1f}" + (f"  16-query waves {row[4]:8.1f}" if len(row) > 4 else "") + f" | auto + tie replay {ties:8.1f} us")
        # LPDNet's two searches as one launch (vcr_knn_pair_f32), both feature-space kernels
        for w8 in (1, 8):
            L = native.lib()
            import ctypes as C
            args = []
            keep = []
            for x, s_, Cc in ((f, sq, 64), (xyz4, None, 4)):
                idx = torch.empty(B, N, k, dtype=torch.int32, device="cuda")
                ties_ = torch.zeros(1 + B * N, dtype=torch.int32, device="cuda")
                a = native.KnnArgs(native.ptr(x), x.stride(1), native.ptr(s_), B, N, Cc, k, native.ptr(idx), native.ptr(ties_), B * N,
                                   w8 if Cc == 64 else 0)
                a.tie_defer = 1
                args.append(a); keep.append((idx, ties_))
            L.vcr_knn_pair_f32.argtypes = [C.POINTER(native.KnnArgs), C.POINTER(native.KnnArgs), C.c_void_p]
            L.vcr_knn_pair_f32.restype = C.c_int
            fn = lambda: native.check(L.vcr_knn_pair_f32(C.byref(args[0]), C.byref(args[1]), C.c_void_p(native.stream_ptr())), "pair")
            print(f"B={B:3d} N={N:5d} k={k:2d} pair launch, feature-space kernel = {'16' if w8 == 8 else '32'}-query waves: {bench(fn):8.1f} us")


if __name__ == "__main__":
    main()
