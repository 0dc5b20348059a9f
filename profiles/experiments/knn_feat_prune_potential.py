import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import importlib
vc = importlib.import_module('vcrnet_amd')
from vcrnet_amd import synth
import oracle
from helpers import cfg_weights

def morton(xyz, bits=10):
    lo, hi = xyz.min(0), xyz.max(0)
    q = np.clip(((xyz - lo) / (hi - lo + 1e-9) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(len(xyz), np.int64)
    for b in range(bits):
        for d in range(3):
            code |= ((q[:, d] >> b) & 1) << (3 * b + d)
    return code

def frac_centroid(f, k, order, T=16):
    p = f[order]; N = len(p)
    d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1)
    kth = np.sort(d2, 1)[:, k + 1]
    nt = N // T
    c = p[: nt * T].reshape(nt, T, -1)
    cen = c.mean(1); rad = np.sqrt(((c - cen[:, None]) ** 2).sum(-1)).max(1)
    dq = np.sqrt(((p[:, None, :] - cen[None]) ** 2).sum(-1))             # [N, nt]
    lb = np.maximum(0, dq - rad[None]) ** 2
    need = lb <= kth[:, None] * 1.001 + 1e-9                              # per query
    nq = N // 16
    wave_need = need[: nq * 16].reshape(nq, 16, nt).any(1)               # a wave skips a tile only if all 16 queries agree
    return need.mean(), wave_need.mean()

for regime in ("default", "trained", "randemb"):
    w = cfg_weights(regime)
    for N, k, kind in ((1024, 20, "object"), (4096, 40, "uniform")):
        src, tgt, _, _, _ = synth.make_batch(0, 1, N, kind=kind)
        rec = {}
        cfg = oracle.OracleConfig(k=k, record=rec)
        s, t = torch.from_numpy(src), torch.from_numpy(tgt)
        try:
            oracle.vcrnet_forward(w, s, t, cfg)
        except Exception as e:
            print("oracle failed", e); continue
        out = []
        for side, xyz in (("emb_src", src[0].T), ("emb_tgt", tgt[0].T)):
            f = rec[side]["x64"][0].numpy().astype(np.float64)
            if f.shape[0] == 64: f = f.T
            a = frac_centroid(f, k, np.arange(N)); b = frac_centroid(f, k, np.argsort(morton(xyz.astype(np.float64))))
            out.append((a[1], b[0], b[1]))
        o = np.array(out).mean(0)
        print(f"{regime:8s} N={N} k={k}: tiles a 16-query wave must visit (centroid + radius bound, perfect threshold): input order {o[0]:.2f}; Morton order {o[2]:.2f} (a single query: {o[1]:.2f})", flush=True)

print("--- two-phase: threshold from the Morton-near tiles only (own tile +- W), then a fixed visit mask")
def frac_two_phase(f, k, order, W, T=16):
    p = f[order]; N = len(p)
    d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1)
    nt = N // T
    c = p[: nt * T].reshape(nt, T, -1)
    cen = c.mean(1); rad = np.sqrt(((c - cen[:, None]) ** 2).sum(-1)).max(1)
    dq = np.sqrt(((p[:, None, :] - cen[None]) ** 2).sum(-1))
    lb = np.maximum(0, dq - rad[None]) ** 2
    nq = N // 16
    tot = 0.0
    for g in range(nq):
        lo, hi = max(0, g - W), min(nt, g + W + 1)
        if hi - lo < 2 * W + 1:                                           # keep 2W+1 tiles at the ends
            if lo == 0: hi = min(nt, 2 * W + 1)
            else: lo = max(0, nt - 2 * W - 1)
        near = d2[g * 16:(g + 1) * 16, lo * T: hi * T]
        thr1 = np.sort(near, 1)[:, k + 1]                                 # (k+2)-th smallest of the near candidates (incl. self)
        need = (lb[g * 16:(g + 1) * 16] <= thr1[:, None] * 1.001 + 1e-9).any(0)
        need[lo:hi] = True
        tot += need.mean()
    return tot / nq
for regime in ("default", "randemb"):
    w = cfg_weights(regime)
    for N, k, kind in ((1024, 20, "object"), (4096, 40, "uniform")):
        src, tgt, _, _, _ = synth.make_batch(0, 1, N, kind=kind)
        rec = {}
        oracle.vcrnet_forward(w, torch.from_numpy(src), torch.from_numpy(tgt), oracle.OracleConfig(k=k, record=rec))
        f = rec["emb_src"]["x64"][0].numpy().astype(np.float64)
        if f.shape[0] == 64: f = f.T
        order = np.argsort(morton(src[0].T.astype(np.float64)))
        print(f"{regime:8s} N={N} k={k}: " + "  ".join(f"W={W}: {frac_two_phase(f, k, order, W):.2f}" for W in (2, 4, 8)), flush=True)
