#!/usr/bin/env python
"""Per-shape timing of vcr_linear_f32 for the 14 linear launches of one forward at BASELINE configs[1]
(M = 2*16*1024 rows): the opt-in persistent kernel (variant 32, deferred epilogue) vs the default
one-tile-per-workgroup kernels (variant 0) and BK 16 forced also with a residual (variant 64).  Run on the GPU box:  python profiles/bench_linear_shapes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vcrnet_amd  # noqa: E402,F401
from vcrnet_amd import native  # noqa: E402

M = 2 * 16 * 1024
SHAPES = [("dg1_pq", 256, 64, 0, 0, 0), ("sn1_pq", 512, 128, 0, 0, 0), ("conv3", 512, 512, 0, 0, 1),
          ("qkv", 1536, 512, 0, 1, 0), ("wo", 512, 512, 1, 0, 1), ("ffn1", 1024, 512, 0, 1, 0),
          ("ffn2", 1024 // 2, 1024, 1, 0, 1), ("cross.q", 512, 512, 0, 1, 0), ("cross.kv", 1024, 512, 0, 1, 0)]


def bench(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    dev = "cuda"
    print(f"{'site':10s} {'N':>5s} {'K':>5s}  {'persist us':>10s} {'TF/s':>7s}   {'default us':>10s} {'TF/s':>7s}")
    for name, N, K, res, ln, st in SHAPES:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev) if res else None
        y = torch.empty(M, N, device=dev)
        lnarg = None
        if ln:
            stats = torch.rand(M, K // 64, 2, device=dev) + 1.0
            lnarg = (stats, torch.randn(N, device=dev), 1e-6)
        out = []
        for variant in (32, 0, 64):
            fn = lambda: native.linear(x, w, b, residual=r, out=y, ln=lnarg, want_stats=bool(st), variant=variant)
            ms = bench(fn)
            out.append((ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12))
        print(f"{name:10s} {N:5d} {K:5d}  {out[0][0]:10.1f} {out[0][1]:7.1f}   {out[1][0]:10.1f} {out[1][1]:7.1f}   "
              f"bk16+res {out[2][0]:10.1f} {out[2][1]:7.1f}")


if __name__ == "__main__":
    main()
