#!/usr/bin/env python
"""vcr_softcorr_f32 (the whole-mode soft-correspondence head, pairscore op 0) for one or more builds of pairscore.hip compiled alone:
us per launch, fp32 TFLOP/s and a checksum.  python profiles/experiments/bench_softcorr_ab.py lib1.so [lib2.so ...]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vcrnet_amd  # noqa: F401,E402
from vcrnet_amd.native import SoftcorrArgs, ptr, stream_ptr  # noqa: E402

g = torch.Generator().manual_seed(0)
B, N, E = 16, 1024, 512
q = (torch.randn(B * N, E, generator=g) * 0.3).cuda()
k = (torch.randn(B * N, E, generator=g) * 0.3).cuda()
xyz = torch.randn(B * N, 3, generator=g)
side = lambda e: torch.cat((xyz, (e.cpu() ** 2).sum(1, keepdim=True)), 1).cuda()
qs, ks = side(q), side(k)
corr = torch.empty(B * N, 4, device="cuda")
for rnd in range(2):
    for path in sys.argv[1:]:
        L = C.CDLL(os.path.abspath(path))
        L.vcr_softcorr_f32.argtypes = [C.POINTER(SoftcorrArgs), C.c_void_p]
        a = SoftcorrArgs(ptr(q), E, ptr(k), E, ptr(qs), ptr(ks), ptr(corr), B, N, N, E, 0, 1.0, None, 0)
        fn = lambda: L.vcr_softcorr_f32(C.byref(a), C.c_void_p(stream_ptr()))
        assert fn() == 0
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 40 * 1e3
        print(f"{os.path.basename(path):24s} {us:7.1f} us  {2.0 * B * N * N * E / us / 1e6:6.1f} TFLOP/s  checksum {corr[:, :3].double().sum().item():.9e}", flush=True)
