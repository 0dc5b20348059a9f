#!/usr/bin/env python
"""vcr_sdpa_f32 (fp32 MFMA) against vcr_sdpa_bf16x3_f32 (exact 3-way bf16 splits on the bf16 matrix pipe) at the
attention shapes of the BASELINE configs: time per launch, fp32-equivalent TFLOP/s, error against fp64 on a sample.
Run on the GPU box:  python profiles/bench_sdpa.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vcrnet_amd  # noqa: E402,F401
from vcrnet_amd import native  # noqa: E402

if os.environ.get("VCR_LIB"):                              # a variant build (profiles/experiments/probe_build.py --out ...)
    native.LIB_PATH = os.path.abspath(os.environ["VCR_LIB"])


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
    h = 4
    print(f"{'shape':24s} {'fp32 us':>9s} {'TF/s':>7s}   {'bf16x3 us':>9s} {'TF/s eq':>7s}   err64 fp32 / bf16x3")
    for nb, N in ((32, 1024), (48, 768), (32, 2048), (64, 4096)):
        qkv = torch.randn(nb * N, 3 * h * 128, device="cuda")
        q, k, v = qkv[:, :512], qkv[:, 512:1024], qkv[:, 1024:]
        fl = 4.0 * nb * h * N * N * 128
        res = []
        for mode in (False, True):
            fn = lambda: native.sdpa(q, k, v, nb, h, N, N, 1 / math.sqrt(128), bf16x3=mode)
            ms = bench(fn, reps=10 if N >= 4096 else 30)
            out = fn()
            # fp64 reference on batch 0, head 0, first 256 queries
            qq, kk, vv = (t[:N, :128].double() for t in (q, k, v))
            ref = torch.softmax(qq[:256] @ kk.T / math.sqrt(128), -1) @ vv
            err = (out[:256, :128].double() - ref).abs().max().item()
            res.append((ms * 1e3, fl / (ms * 1e-3) / 1e12, err))
        print(f"nb={nb:3d} N={N:5d} h=4 d=128   {res[0][0]:9.1f} {res[0][1]:7.1f}   {res[1][0]:9.1f} {res[1][1]:7.1f}   "
              f"{res[0][2]:.2e} / {res[1][2]:.2e}")


if __name__ == "__main__":
    main()
