#!/usr/bin/env python
"""Timings only of the kNN kernels of one library build (variants: profiles/experiments/knn_variant_build.sh):
  python profiles/experiments/knn16_time.py LIB [WAVES=16]
feat / xyz alone and the one-launch pair at BASELINE configs[1] / [2] / [3]-share / [4]."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vcrnet_amd  # noqa: E402,F401
from vcrnet_amd import native as nat  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] != "-":
    nat.LIB_PATH = os.path.join(ROOT, sys.argv[1])
W = int(sys.argv[2]) if len(sys.argv) > 2 else 16


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
    return a.elapsed_time(b) / reps * 1e3


def pair_fn(f, sq, x4, k, waves, xt):
    L = nat.lib()
    B, N, _ = f.shape
    args, keep = [], []
    for x, s_, Cc in ((f, sq, 64), (x4, None, 4)):
        idx = torch.empty(B, N, k, dtype=torch.int32, device="cuda")
        t_ = torch.zeros(1 + B * N, dtype=torch.int32, device="cuda")
        args.append(nat.KnnArgs(nat.ptr(x), x.stride(1), nat.ptr(s_), B, N, Cc, k, nat.ptr(idx), nat.ptr(t_), B * N, waves))
        keep.append((idx, t_))
    args[0].xt = nat.ptr(xt)
    L.vcr_knn_pair_f32.argtypes = [C.POINTER(nat.KnnArgs), C.POINTER(nat.KnnArgs), C.c_void_p]
    L.vcr_knn_pair_f32.restype = C.c_int
    return (lambda: nat.check(L.vcr_knn_pair_f32(C.byref(args[0]), C.byref(args[1]), C.c_void_p(nat.stream_ptr())), "pair")), keep


g = torch.Generator().manual_seed(0)
out = []
for B, N, k in ((32, 1024, 20), (48, 768, 20), (32, 2048, 20), (64, 4096, 40)):
    f = torch.randn(B, N, 64, generator=g).cuda()
    sq = (f ** 2).sum(-1).contiguous()
    ft = f.view(B, N, 4, 4, 4).transpose(3, 4).reshape(B, N, 64).contiguous()
    xyz = torch.rand(B, N, 3, generator=g) - 0.5
    x4 = torch.cat((xyz, (xyz ** 2).sum(-1, keepdim=True)), -1).cuda().contiguous()
    t = (bench(lambda: nat.knn(f, sq, k, waves=W, xt=ft)), bench(lambda: nat.knn(x4, None, k, waves=W if W == 16 else 0)),
         bench(pair_fn(f, sq, x4, k, W, ft)[0]))
    out.append(f"{B}x{N} k{k}: feat {t[0]:7.1f} xyz {t[1]:7.1f} pair {t[2]:7.1f}")
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'product':28s} waves={W:2d} | " + " | ".join(out), flush=True)
