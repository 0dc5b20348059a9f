#!/usr/bin/env python
"""A/B of vcr_edgeconv_f32 between two builds of the library (same process cannot load both: one run per library).
  python profiles/experiments/bench_edgeconv.py [path/to/lib.so]
Shapes: BASELINE configs[1] (32 clouds x 1024, k 20), configs[2] (48 x 768), configs[3] share (32 x 2048), configs[4]
(64 x 4096, k 40), and two ragged ones.  Prints us per launch, TFLOP/s of the convDG2 GEMM, and a checksum (the two
builds must print the same: bit-identical results)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vcrnet_amd  # noqa: E402,F401
from vcrnet_amd import native  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] != "-":
    native.LIB_PATH = os.path.abspath(sys.argv[1])
BX3 = "bf16x3" in sys.argv                                  # vcr_edgeconv_bf16x3_f32 instead (x2 agrees to fp32-GEMM rounding)
g = torch.Generator().manual_seed(0)


def bench(fn, reps=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print("library", native.LIB_PATH)
SHAPES = ((32, 1024, 20), (48, 768, 20), (32, 2048, 20), (64, 4096, 40), (3, 77, 20), (5, 333, 40))
if "sizes" in sys.argv:                                     # k and launch size separated: both k at both sizes
    SHAPES = ((32, 1024, 20), (32, 1024, 40), (64, 4096, 20), (64, 4096, 40), (128, 1024, 20), (8, 4096, 20))
for B, N, k in SHAPES:
    if BX3 and k not in (20, 40):
        continue
    M = B * N
    pq = torch.randn(M, 256, generator=g).cuda()
    w2 = (torch.randn(128, 128, generator=g) / 11).cuda()
    b2 = torch.randn(128, generator=g).cuda()
    idx = torch.randint(0, N, (M, k), generator=g, dtype=torch.int32).cuda()
    x1, x2 = native.edgeconv(pq, idx, N, w2, b2, bf16x3=BX3)
    # reference: the definition, in torch (max is exact; the GEMM agrees to fp32 rounding)
    sel = slice(0, min(M, 4096))
    base = (torch.arange(M, device="cuda") // N * N)[sel, None]
    H = torch.relu(pq[:, :128][(base + idx[sel].long())] + pq[sel, None, 128:])
    r1 = H.max(1).values
    r2 = torch.relu((H @ w2.t()).max(1).values + b2)
    ok1 = torch.equal(x1[sel], r1)
    d2 = (x2[sel] - r2).abs().max().item()
    us = bench(lambda: native.edgeconv(pq, idx, N, w2, b2, bf16x3=BX3))
    tf = 2.0 * M * k * 128 * 128 / (us * 1e-6) / 1e12
    print(f"B={B:3d} N={N:5d} k={k}: {us:8.1f} us  {tf:6.1f} TF/s   x1 exact {ok1}  max|x2 - ref| {d2:.1e}   checksum "
          f"{x1.double().sum().item():.10e} {x2.double().sum().item():.10e}", flush=True)
