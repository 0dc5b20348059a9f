#!/usr/bin/env python
"""vcr_linear_bf16x3_f32 per shape: us per launch, fp32-equivalent TFLOP/s (peak 2500 / 6 = 417), a checksum of the output
(two builds of the library must print the same: bit-identical), and the error against float64.
  python profiles/experiments/bench_linear_bf16x3.py [path/to/lib.so]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vcrnet_amd  # noqa
from vcrnet_amd import native
if len(sys.argv) > 1:
    native.LIB_PATH = os.path.abspath(sys.argv[1])
g = torch.Generator().manual_seed(0)
def bench(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print("library", native.LIB_PATH)
M = 32768
for name, N, K, res, ln, st in (("sn1_pq", 512, 128, 0, 0, 0), ("conv3", 512, 512, 0, 0, 1), ("qkv", 1536, 512, 0, 1, 0), ("wo", 512, 512, 1, 0, 1),
                                ("ffn1", 1024, 512, 0, 1, 0), ("ffn2", 512, 1024, 1, 0, 1)):
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).cuda() if res else None
    lnarg = ((torch.rand(M, K // 64, 2, generator=g) + 1.0).cuda(), torch.randn(N, generator=g).cuda(), 1e-6) if ln else None
    planes = native.split_bf16x3(w)
    y = torch.empty(M, N, device="cuda")
    fn = lambda: native.linear_bf16x3(x, planes, N, bias=b, residual=r, out=y, ln=lnarg, want_stats=bool(st))
    out = fn()
    yy = out[0] if isinstance(out, tuple) else out
    us = bench(fn)
    err = "-"
    if not ln:
        ref = x[:2048].double() @ w.double().t() + b.double() + (r[:2048].double() if res else 0)
        err = f"{float((yy[:2048].double() - ref).abs().max()):.1e}"
    print(f"{name:7s} N={N:4d} K={K:4d}: {us:7.1f} us  {2.0 * M * N * K / (us * 1e-6) / 1e12:6.1f} TF/s-eq ({2.0 * M * N * K / (us * 1e-6) / 1e12 / 416.7:.2f})  "
          f"checksum {yy.double().sum().item():.10e}  max|err| vs fp64 {err}", flush=True)
