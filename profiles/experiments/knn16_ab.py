#!/usr/bin/env python
"""Round 5: the atomic-log 16-query kNN kernels (waves = NEW) against the round-4 kernels (waves = 8 / 1 / 2): neighbour
sets on random, tie-heavy, duplicate-point and index-structured clouds (single launches and the one-launch pair, with
and without the tie replay), then timings at BASELINE configs[1] / [2] / [3] / [4].
Run on the GPU box:  python profiles/experiments/knn16_ab.py [NEW=16] [quick]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vcrnet_amd  # noqa: E402,F401
from vcrnet_amd import native as nat  # noqa: E402

NEW = int(sys.argv[1]) if len(sys.argv) > 1 else 16
QUICK = "quick" in sys.argv


def bench(fn, reps=20):
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


def pair_fn(f, sq, x4, k, waves, ties=True, xt=None):
    L = nat.lib()
    B, N, _ = f.shape
    args, keep = [], []
    for x, s_, Cc in ((f, sq, 64), (x4, None, 4)):
        idx = torch.empty(B, N, k, dtype=torch.int32, device="cuda")
        t_ = torch.zeros(1 + B * N, dtype=torch.int32, device="cuda") if ties else None
        a = nat.KnnArgs(nat.ptr(x), x.stride(1), nat.ptr(s_), B, N, Cc, k, nat.ptr(idx), nat.ptr(t_), B * N if ties else 0, waves)
        args.append(a); keep.append((idx, t_))
    args[0].xt = nat.ptr(xt)
    L.vcr_knn_pair_f32.argtypes = [C.POINTER(nat.KnnArgs), C.POINTER(nat.KnnArgs), C.c_void_p]
    L.vcr_knn_pair_f32.restype = C.c_int

    def fn():
        nat.check(L.vcr_knn_pair_f32(C.byref(args[0]), C.byref(args[1]), C.c_void_p(nat.stream_ptr())), "pair")
    return fn, keep


def sets(t):
    return np.sort(t.cpu().numpy(), -1)


def clouds(kind, B, N, rs):
    if kind == "random":
        f = rs.randn(B, N, 64).astype(np.float32); xyz = rs.rand(B, N, 3).astype(np.float32)
    elif kind == "ties":
        f = rs.randint(0, 3, size=(B, N, 64)).astype(np.float32); xyz = rs.randint(0, 9, size=(B, N, 3)).astype(np.float32)
    elif kind == "grid16":      # neighbours at index distance +-16 m: all in one lane row of the 16-candidate tiles
        i = np.arange(N)
        xyz = np.stack((i % 16, i // 16, np.zeros(N)), -1)[None].repeat(B, 0).astype(np.float32)
        xyz += rs.rand(B, N, 3).astype(np.float32) * 0.01
        f = np.tile(xyz, (1, 1, 22))[:, :, :64].astype(np.float32) + rs.randn(B, N, 64).astype(np.float32) * 0.01
    elif kind == "sorted":      # ascending along x: a query's neighbours are its index neighbours; late candidates always win
        xyz = np.sort(rs.rand(B, N, 3).astype(np.float32), 1)
        f = np.sort(rs.randn(B, N, 64).astype(np.float32), 1)
    elif kind == "allsame":
        f = np.ones((B, N, 64), np.float32); xyz = np.ones((B, N, 3), np.float32)
    f, xyz = torch.from_numpy(f).cuda(), torch.from_numpy(xyz).cuda()
    sq = (f ** 2).sum(-1).contiguous()
    x4 = torch.cat((xyz, (xyz ** 2).sum(-1, keepdim=True)), -1).contiguous()
    return f, sq, x4


def main():
    rs = np.random.RandomState(5)
    bad = 0
    shapes = [(32, 1024, 20), (3, 1000, 20), (70, 250, 5), (9, 2401, 40), (5, 3000, 20), (17, 777, 40), (2, 4096, 40),
              (64, 64, 20), (40, 41, 40), (30, 700, 50), (12, 1030, 62)]
    if QUICK:
        shapes = shapes[:4]
    for kind in ("random", "ties", "grid16", "sorted", "allsame"):
        for B, N, k in shapes:
            f, sq, x4 = clouds(kind, B, N, rs)
            ft = f.view(B, N, 4, 4, 4).transpose(3, 4).reshape(B, N, 64).contiguous()
            exact = kind not in ("allsame",)
            msgs = []
            for ties in (True, False):
                # reference kernels: round 4's 16-query kernels (same rank-0 rule on duplicate points: the lowest index goes)
                r64 = sets(nat.knn(f, sq, k, exact_ties=ties, waves=8))
                r3 = sets(nat.knn(x4, None, k, exact_ties=ties, waves=0))
                n64 = sets(nat.knn(f, sq, k, exact_ties=ties, waves=NEW))
                n64t = sets(nat.knn(f, sq, k, exact_ties=ties, waves=NEW, xt=ft))
                n3 = sets(nat.knn(x4, None, k, exact_ties=ties, waves=NEW))
                if ties and exact:
                    d = [int((n64 != r64).any(-1).sum()), int((n64t != r64).any(-1).sum()), int((n3 != r3).any(-1).sum())]
                else:
                    # without the replay (or with every point the same) a tied row may keep any of the tied candidates: check
                    # the VALUES of the kept neighbours instead of their indices
                    def vals(x, sqv, idx, c64):
                        xx = x[..., :3] if not c64 else x
                        g = torch.gather(xx, 1, torch.from_numpy(idx.astype(np.int64)).cuda().reshape(B, -1, 1).expand(-1, -1, xx.shape[-1]))
                        g = g.view(B, N, k, -1)
                        dd = ((g - xx[:, :, None, :]) ** 2).sum(-1)
                        return np.sort(dd.cpu().numpy(), -1)
                    d = [int((~np.isclose(vals(f, sq, n64, True), vals(f, sq, r64, True), rtol=1e-5, atol=1e-5)).any(-1).sum()),
                         int((~np.isclose(vals(f, sq, n64t, True), vals(f, sq, r64, True), rtol=1e-5, atol=1e-5)).any(-1).sum()),
                         int((~np.isclose(vals(x4, None, n3, False), vals(x4, None, r3, False), rtol=1e-5, atol=1e-5)).any(-1).sum())]
                    # and every row must hold k distinct, valid indices
                    for nm, arr in (("f", n64), ("ft", n64t), ("x", n3)):
                        if (arr < 0).any() or (arr >= N).any() or (np.diff(arr, axis=-1) == 0).any():
                            d.append(-1)
                if k <= 40:
                    fnp, keep = pair_fn(f, sq, x4, k, NEW, ties=ties, xt=ft)
                    fnp(); torch.cuda.synchronize()
                    p64, p3 = sets(keep[0][0]), sets(keep[1][0])
                    d += [int((p64 != n64).any(-1).sum()), int((p3 != n3).any(-1).sum())]
                msgs.append(f"ties={int(ties)} {d}")
                bad += any(x != 0 for x in d)
            print(f"{kind:8s} B={B:3d} N={N:5d} k={k:2d}: rows differing [feat, feat-xt, xyz, pair-feat, pair-xyz] " + " | ".join(msgs), flush=True)
    print("MISMATCHING CASES:", bad, flush=True)

    g = torch.Generator().manual_seed(0)
    for B, N, k in ((32, 1024, 20), (48, 768, 20), (32, 2048, 20), (64, 4096, 40)):
        f = torch.randn(B, N, 64, generator=g).cuda()
        sq = (f ** 2).sum(-1).contiguous()
        ft = f.view(B, N, 4, 4, 4).transpose(3, 4).reshape(B, N, 64).contiguous()
        xyz = torch.rand(B, N, 3, generator=g) - 0.5
        x4 = torch.cat((xyz, (xyz ** 2).sum(-1, keepdim=True)), -1).cuda().contiguous()
        row = []
        for w in (8, NEW):
            row.append((bench(lambda: nat.knn(f, sq, k, waves=w, xt=ft)), bench(lambda: nat.knn(x4, None, k, waves=w if w == NEW else 0)),
                        bench(pair_fn(f, sq, x4, k, w, xt=ft)[0])))
        print(f"B={B:3d} N={N:5d} k={k:2d}  round 4: feat {row[0][0]:8.1f} xyz {row[0][1]:8.1f} pair {row[0][2]:8.1f} us   |   "
              f"new: feat {row[1][0]:8.1f} xyz {row[1][1]:8.1f} pair {row[1][2]:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
