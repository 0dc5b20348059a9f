#!/usr/bin/env python
"""Does the issue-bound kNN launch hide under an MFMA-bound launch when the two run on different streams?
  python profiles/experiments/overlap_knn_edgeconv.py
BASELINE configs[1] stage shapes (32 clouds x 1024, k 20).  Prints, per launch pair, the solo times, their sum, and the
wall time of the two enqueued on two streams (fork / join by events), plus the half-batch pipeline
  s1: knn(h1) -> edgeconv(h1)      s2: knn(h2) -> edgeconv(h2)
against knn(32) -> edgeconv(32) on one stream.  Measurement only (nothing here is product code)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vcrnet_amd  # noqa: E402,F401
from vcrnet_amd import native as nat  # noqa: E402

g = torch.Generator().manual_seed(0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
KEEP = []


def wall(fn, reps=40):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def two_streams(fa, fb):
    def run():
        cur = torch.cuda.current_stream()
        e0 = torch.cuda.Event()
        e0.record(cur)
        for s, f in ((s1, fa), (s2, fb)):
            s.wait_event(e0)
            with torch.cuda.stream(s):
                f()
            e = torch.cuda.Event()
            e.record(s)
            cur.wait_event(e)
    return run


def stage_inputs(B, N, k):
    f = torch.randn(B, N, 64, generator=g).cuda()
    sq = (f ** 2).sum(-1).contiguous()
    ft = f.view(B, N, 4, 4, 4).transpose(3, 4).reshape(B, N, 64).contiguous()
    M = B * N
    pq = torch.randn(M, 256, generator=g).cuda()
    w2 = (torch.randn(128, 128, generator=g) / 11).cuda()
    b2 = torch.randn(128, generator=g).cuda()
    idx = torch.randint(0, N, (M, k), generator=g, dtype=torch.int32).cuda()
    xyz = torch.rand(B, N, 3, generator=g) - 0.5
    x4 = torch.cat((xyz, (xyz ** 2).sum(-1, keepdim=True)), -1).cuda().contiguous()
    L = nat.lib()
    args = []
    for x, s_, Cc in ((f, sq, 64), (x4, None, 4)):                        # the stage's one-launch pair of searches
        ix = torch.empty(B, N, k, dtype=torch.int32, device="cuda")
        t_ = torch.zeros(1 + B * N, dtype=torch.int32, device="cuda")
        args.append(nat.KnnArgs(nat.ptr(x), x.stride(1), nat.ptr(s_), B, N, Cc, k, nat.ptr(ix), nat.ptr(t_), B * N, 0))
        KEEP.append((x, s_, ix, t_, ft))
    args[0].xt = nat.ptr(ft)
    L.vcr_knn_pair_f32.argtypes = [C.POINTER(nat.KnnArgs), C.POINTER(nat.KnnArgs), C.c_void_p]
    L.vcr_knn_pair_f32.restype = C.c_int
    knn = lambda: nat.check(L.vcr_knn_pair_f32(C.byref(args[0]), C.byref(args[1]), C.c_void_p(nat.stream_ptr())), "pair")  # noqa: E731
    ec = lambda: nat.edgeconv(pq, idx, N, w2, b2)                         # noqa: E731
    return knn, ec


N, k = 1024, 20
knn32, ec32 = stage_inputs(32, N, k)
knn16a, ec16a = stage_inputs(16, N, k)
knn16b, ec16b = stage_inputs(16, N, k)
X = torch.randn(32 * N, 512, generator=g).cuda()
Wq = (torch.randn(1536, 512, generator=g) / 23).cuda()
bq = torch.randn(1536, generator=g).cuda()
lin = lambda: nat.linear(X, Wq, bq)                                      # noqa: E731

t = {n: wall(f) for n, f in (("knn32", knn32), ("ec32", ec32), ("knn16", knn16a), ("ec16", ec16a), ("lin", lin))}
print("solo us:", "  ".join(f"{n} {v:7.1f}" for n, v in t.items()), flush=True)
for name, fa, fb, ta, tb in (("knn32 || ec32 ", knn32, ec32, "knn32", "ec32"), ("knn16 || ec16 ", knn16a, ec16b, "knn16", "ec16"),
                             ("knn32 || lin  ", knn32, lin, "knn32", "lin"), ("ec32  || lin  ", ec32, lin, "ec32", "lin"),
                             ("knn16 || knn16", knn16a, knn16b, "knn16", "knn16"), ("ec16  || ec16 ", ec16a, ec16b, "ec16", "ec16")):
    w = wall(two_streams(fa, fb))
    print(f"{name}: sum {t[ta] + t[tb]:7.1f}  max {max(t[ta], t[tb]):7.1f}  two streams {w:7.1f} us", flush=True)
serial = wall(lambda: (knn32(), ec32()))
pipe = wall(two_streams(lambda: (knn16a(), ec16a()), lambda: (knn16b(), ec16b())))
print(f"knn32 -> ec32 on one stream {serial:7.1f} us;  half-batch pipeline on two streams {pipe:7.1f} us", flush=True)
