#!/usr/bin/env python
"""Per-shape timing of vcr_linear_f32 for the 14 linear launches of one forward at BASELINE configs[1]
(M = 2*16*1024 rows, or --rows M): the 32x32x2 MFMA kernels (variant 1024) vs the 16x16x4 ones (variant 16), each with
the automatic BK and with the other BK forced.  Run on the GPU box:  python profiles/bench_linear_shapes.py [--rows M]"""
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
    global M
    if "--rows" in sys.argv:
        M = int(sys.argv[sys.argv.index("--rows") + 1])
    dev = "cuda"
    print(f"M = {M} rows; us per launch (TFLOP/s)")
    print(f"{'site':10s} {'N':>5s} {'K':>5s}  {'32x32x2 auto':>18s} {'32x32x2 other BK':>18s} {'16x16x4 auto':>18s} {'16x16x4 other BK':>18s}")
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
        other = 64 if res else 8                          # automatic: BK 32 with a residual, BK 16 without
        for variant in (1024, 1024 | other, 16, 16 | other):
            fn = lambda: native.linear(x, w, b, residual=r, out=y, ln=lnarg, want_stats=bool(st), variant=variant)
            ms = bench(fn)
            out.append((ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12))
        print(f"{name:10s} {N:5d} {K:5d}  " + " ".join(f"{u:9.1f} ({tf:6.1f})" for u, tf in out))


if __name__ == "__main__":
    main()
