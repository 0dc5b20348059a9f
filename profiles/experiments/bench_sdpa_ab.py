#!/usr/bin/env python
"""vcr_sdpa_f32 per shape for one build of the library: us, TFLOP/s, a checksum (two builds must agree: bit-identical).
  python profiles/experiments/bench_sdpa_ab.py [path/to/lib.so]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vcrnet_amd  # noqa
from vcrnet_amd import native
if len(sys.argv) > 1:
    native.LIB_PATH = os.path.abspath(sys.argv[1])
def bench(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
g = torch.Generator().manual_seed(0)
h = 4
print("library", native.LIB_PATH)
for nb, N in ((32, 1024), (48, 768), (32, 2048), (64, 4096), (6, 333)):
    qkv = torch.randn(nb * N, 3 * h * 128, generator=g).cuda()
    q, k, v = qkv[:, :512], qkv[:, 512:1024], qkv[:, 1024:]
    fn = lambda: native.sdpa(q, k, v, nb, h, N, N, 1 / math.sqrt(128))
    out = fn()
    us = bench(fn, reps=10 if N >= 4096 else 30)
    # statistics form (no P V): row (max, sum)
    st = native.sdpa(q, k, v, nb, h, N, N, 1 / math.sqrt(128), want_rowstat=True, pv=False)
    stc = st[1].double().sum().item() if isinstance(st, tuple) else 0.0
    print(f"nb={nb:3d} N={N:5d}: {us:9.1f} us  {4.0 * nb * h * N * N * 128 / (us * 1e-6) / 1e12:6.1f} TF/s   checksum {out.double().sum().item():.10e}  stats {stc:.10e}", flush=True)
