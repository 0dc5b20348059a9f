#!/usr/bin/env python
"""Experiment: phase breakdown of the kNN kernels (wave 0 of every workgroup of cloud 0 accumulates the 100 MHz wall
clock per phase).  Needs the probe library scratch/libvcr_probe.so (python profiles/experiments/probe_build.py)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "scratch", "libvcr_probe.so")


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
    names = {"feat64": ["prefetch issue", "MFMA wait+filter", "make_room", "pick", "log push", "drain", "-", "-"],
             "xyz": ["dist+filter", "log+drain", "-", "-", "-", "-", "-", "-"]}
    for B, N, k in ((32, 1024, 20), (64, 4096, 40)):
        f = torch.randn(B, N, 64, generator=g).cuda()
        sq = (f ** 2).sum(-1).contiguous()
        xyz = torch.rand(B, N, 3, generator=g) - 0.5
        xyz4 = torch.cat((xyz, (xyz ** 2).sum(-1, keepdim=True)), -1).cuda().contiguous()
        for name, x, s in (("feat64", f, sq), ("xyz", xyz4, None)):
            for waves in (0, 1, 2):          # 0 = automatic: the 16-query kernel on these grids (feat64)
                for _ in range(2):
                    native.knn(x, s, k, exact_ties=False, waves=waves)
                torch.cuda.synchronize()
                L.vcr_dbg_probe_knn(None, 1)
                native.knn(x, s, k, exact_ties=False, waves=waves)
                torch.cuda.synchronize()
                L.vcr_dbg_probe_knn(full.ctypes.data, 0)
                buf = full[:, :8]
                t = buf.astype(np.float64) * 0.01
                used = t.sum(1) > 0
                med = np.median(t[used], 0)
                print(f"B={B} N={N} k={k} {name} S={waves}: {used.sum()} workgroups; median us per wave: " +
                      ", ".join(f"{n} {v:.1f}" for n, v in zip(names[name], med)) + f"  | total {med.sum():.1f}")


if __name__ == "__main__":
    main()
