"""kNN pair launch on clouds that carry copies of points: python profiles/bench_knn_copies.py.  Every copy is a row with a shared
best value (util.py:159 drops the copy Tensor.topk returns first), i.e. a row the launch replays: what that costs, by the fraction
of a cloud's points that are copies."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import vcrnet_amd  # noqa
from vcrnet_amd import native as nat
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for B, N, k in ((32, 1024, 20), (32, 2048, 20), (64, 4096, 40)):
    rs = np.random.RandomState(N)
    for frac in (0.0, 0.01, 0.1, 0.5):
        xyz = rs.rand(B, N, 3).astype(np.float32) - 0.5
        w1, w2 = rs.randn(3, 64).astype(np.float32) * 0.8, rs.randn(64, 64).astype(np.float32) * 0.2
        n2 = int(N * frac / 2)
        for b in range(B):
            p = rs.permutation(N)
            xyz[b, p[:n2]] = xyz[b, p[n2:2 * n2]]
        feat = np.maximum(np.maximum(xyz @ w1 + 0.1, 0) @ w2 + 0.05, 0)
        feat = torch.from_numpy(np.ascontiguousarray(feat)).cuda()
        x4 = torch.from_numpy(np.concatenate((xyz, (xyz ** 2).sum(-1, keepdims=True)), -1).astype(np.float32)).cuda()
        sq = (feat ** 2).sum(-1).contiguous()
        ft = feat.view(B, N, 4, 4, 4).transpose(3, 4).reshape(B, N, 64).contiguous()
        us = timed(lambda: nat.knn_pair(feat, sq, x4, k, xt=ft, tie_slots=(k > 20)))
        lazy = timed(lambda: nat.knn_pair(feat, sq, x4, k, xt=ft, exact_ties=False)) if "exact_ties" in nat.knn_pair.__code__.co_varnames else float("nan")
        print(f"B={B} N={N} k={k} copies {frac:4.0%} of the points: pair launch {us:8.1f} us  (without tie_scratch {lazy:8.1f})", flush=True)
