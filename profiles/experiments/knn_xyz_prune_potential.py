import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib
vc = importlib.import_module('vcrnet_amd')
from vcrnet_amd import synth

def morton(xyz, bits=10):
    lo, hi = xyz.min(0), xyz.max(0)
    q = np.clip(((xyz - lo) / (hi - lo + 1e-9) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(len(xyz), np.int64)
    for b in range(bits):
        for d in range(3):
            code |= ((q[:, d] >> b) & 1) << (3 * b + d)
    return code

def frac(xyz, k, order):
    p = xyz[order]
    N = len(p)
    d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1)
    kth = np.sort(d2, 1)[:, k + 1]                       # (k+2)-th smallest incl. self ~ the list's threshold
    nq, nt = N // 16, N // 16
    qlo = p[: nq * 16].reshape(nq, 16, 3).min(1); qhi = p[: nq * 16].reshape(nq, 16, 3).max(1)
    clo = p[: nt * 16].reshape(nt, 16, 3).min(1); chi = p[: nt * 16].reshape(nt, 16, 3).max(1)
    gap = np.maximum(0, np.maximum(clo[None] - qhi[:, None], qlo[:, None] - chi[None]))
    mind2 = (gap ** 2).sum(-1)                            # [nq, nt] box-to-box distance^2
    thr = kth[: nq * 16].reshape(nq, 16).max(1)
    return (mind2 <= thr[:, None] * 1.001 + 1e-6).mean()

for N, k, kind in ((1024, 20, "object"), (2048, 20, "uniform"), (4096, 40, "uniform")):
    src, tgt, _, _, _ = synth.make_batch(0, 2, N, kind=kind)
    fs = []
    for c in list(src) + list(tgt):
        xyz = c.T.astype(np.float64)
        fs.append((frac(xyz, k, np.arange(N)), frac(xyz, k, np.argsort(xyz[:, 0])), frac(xyz, k, np.argsort(morton(xyz)))))
    fs = np.array(fs).mean(0)
    print(f"N={N} k={k} {kind}: candidate tiles a 16-query wave must visit -- input order {fs[0]:.2f}, sorted by x {fs[1]:.2f}, Morton order {fs[2]:.2f}")
