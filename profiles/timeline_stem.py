#!/usr/bin/env python
"""Experiment: phase breakdown of the fused stem launch (waves 0 and 7 of every workgroup accumulate the 100 MHz wall
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
    L.vcr_dbg_probe_pointwise.argtypes = [C.c_void_p, C.c_int]
    L.vcr_dbg_probe_pointwise.restype = C.c_int
    g = torch.Generator().manual_seed(0)
    w1 = torch.randn(64, 3, generator=g).cuda(); b1 = torch.randn(64, generator=g).cuda()
    w2 = (torch.randn(64, 64, generator=g) / 8).cuda(); b2 = torch.randn(64, generator=g).cuda()
    wpq = (torch.randn(256, 64, generator=g) / 8).cuda(); bpq = torch.randn(256, generator=g).cuda()
    names = ["weight loads issued", "weights arrived + barrier", "x + conv1", "conv2 + tile to LDS", "feature epilogue",
             "P|Q MFMAs + stores issued", "stores acknowledged", "-"]
    full = np.zeros((4096, 32), np.uint64)
    for B, N in ((1, 16), (32, 1024), (48, 768)):
        x = (torch.rand(B, 3, N, generator=g) - 0.5).cuda()
        ft = torch.empty(B, N, 64, device="cuda")
        for _ in range(3):
            native.pointwise(x, w1, b1, w2, b2, wpq, bpq, feat_t=ft)
        torch.cuda.synchronize()
        L.vcr_dbg_probe_pointwise(None, 1)
        native.pointwise(x, w1, b1, w2, b2, wpq, bpq, feat_t=ft)
        torch.cuda.synchronize()
        L.vcr_dbg_probe_pointwise(full.ctypes.data, 0)
        buf = full[:512, :8]
        t = buf.astype(np.float64) * 0.01
        for wv in (0, 1):
            tw = t[wv::2]
            used = tw.sum(1) > 0
            med = np.median(tw[used], 0)
            print(f"B={B} N={N} wave {0 if wv == 0 else 7}: {used.sum()} workgroups; median us: " +
                  ", ".join(f"{n} {v:.2f}" for n, v in zip(names, med)) + f"  | total {med.sum():.1f}")


if __name__ == "__main__":
    main()
