#!/usr/bin/env python
"""vcr_gathermax_args.order at the shape where the forward's convSN1 gathers go through L2 (clouds beyond 2048 points): the
Cartesian kNN graph of uniform clouds, points served in index order against Morton rank order.   python profiles/bench_gathermax_order.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vcrnet_amd  # noqa: E402,F401
from vcrnet_amd import native as nat  # noqa: E402


def bench(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    for B, N, k in ((64, 4096, 40), (64, 4096, 20), (32, 8192, 20), (32, 3000, 20)):
        g = torch.Generator().manual_seed(N + k)
        xyz = torch.rand(B, N, 3, generator=g) - 0.5
        x4 = torch.cat((xyz, (xyz ** 2).sum(-1, keepdim=True)), -1).cuda().contiguous()
        idx = nat.knn(x4, None, k).view(B * N, k)
        pq = torch.randn(B * N, 512, generator=g).cuda()
        perm = nat.knn_order(x4)["perm"].view(-1)
        y0 = nat.gathermax(pq, 256, idx, N, variant=1)
        y1 = nat.gathermax(pq, 256, idx, N, variant=1, order=perm)
        t0 = min(bench(lambda: nat.gathermax(pq, 256, idx, N, variant=1)) for _ in range(3))
        t1 = min(bench(lambda: nat.gathermax(pq, 256, idx, N, variant=1, order=perm)) for _ in range(3))
        gb = B * N * k * 1024 / 1e9
        print(f"{B:3d} x {N:5d} k={k:2d}: index order {t0:8.1f} us ({gb / t0 * 1e3:6.1f} TB/s of gathers)   Morton order {t1:8.1f} us "
              f"({gb / t1 * 1e3:6.1f} TB/s)   {100 * (t0 / t1 - 1):+.0f} %   bit-identical: {torch.equal(y0, y1)}", flush=True)


if __name__ == "__main__":
    main()
