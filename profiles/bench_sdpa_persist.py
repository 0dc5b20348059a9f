#!/usr/bin/env python
"""vcr_sdpa_args.variant: the tile kernel (1) against the persistent kernel (2) at the attention-output shapes of the BASELINE
configs, alternated on the same box (three rounds), grouped (encoder + decoder self-attention, the forward's form) and plain
(the cross-attention's).  Bit-identity is asserted on every shape.   python profiles/bench_sdpa_persist.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vcrnet_amd  # noqa: E402,F401
from vcrnet_amd import native  # noqa: E402


def bench(fn, reps):
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
    h = 4
    for nb, N, grouped in ((32, 1024, True), (32, 1024, False), (48, 768, True), (32, 2048, True), (32, 2048, False),
                           (64, 4096, True), (64, 4096, False)):
        if grouped:
            qkv = torch.randn(nb * N, 6 * h * 128, device="cuda")
            kw = dict(groups=(2, 1536, 1536, 1536))
            q, k, v = qkv[:, :512], qkv[:, 512:1024], qkv[:, 1024:1536]
            fl = 2 * 4.0 * nb * h * N * N * 128
        else:
            qkv = torch.randn(nb * N, 3 * h * 128, device="cuda")
            kw = dict(kv_batch_shift=nb // 2)
            q, k, v = qkv[:, :512], qkv[:, 512:1024], qkv[:, 1024:]
            fl = 4.0 * nb * h * N * N * 128
        run = lambda var: native.sdpa(q, k, v, nb, h, N, N, 1 / math.sqrt(128), variant=var, **kw)
        same = torch.equal(run(1), run(2))
        reps = 5 if N >= 4096 else 20
        t = {1: [], 2: []}
        for _ in range(3):
            for var in (1, 2):
                t[var].append(bench(lambda: run(var), reps))
        b1, b2 = min(t[1]), min(t[2])
        print(f"nb={nb:3d} N={N:5d} {'grouped x2' if grouped else 'plain     '}  tile {b1:9.1f} us {fl / b1 / 1e6:6.1f} TF/s ({', '.join(f'{x:.0f}' for x in t[1])})"
              f"   persistent {b2:9.1f} us {fl / b2 / 1e6:6.1f} TF/s ({', '.join(f'{x:.0f}' for x in t[2])})   {100 * (b1 / b2 - 1):+.1f} %  bit-identical: {same}",
              flush=True)


if __name__ == "__main__":
    main()
