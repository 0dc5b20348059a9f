#!/usr/bin/env python
"""How far the REFERENCE is from ITSELF on the paths whose tests cannot be held to the plain BASELINE tolerance
(SURVEY F5): the same reference code, weights and inputs run with 8 / 2 / 1 CPU threads (oneDNN / MKL pick different
kernels and summation orders) and in float64.  Written to tests/golden/selfdiv.npz; the GPU tests bound the HIP path's
distance from the recorded 8-thread run by the reference's own spread recorded next to it.

Runs only in the build container (imports /root/reference unmodified; stub pynvml).  Usage:
    python tests/golden/gen_selfdiv_golden.py            # ~10 min on 8 cores

Cases
  c3        BASELINE configs[2] itself: partial overlap, clouds cropped 1024 -> 768, B = 24, vcrnetIter(iter=3).
            Per run: per-iteration (R, t), the five discrete selections of every pass, the composed pose.
            Per pair of runs: flips per iteration and kind, pairs of the 24 whose final pose agrees within 1e-4 / 1e-5.
  it2_n256  whole mode, vcrnetIter(iter=2), N = 256, B = 2 (items 400..): composed pose per run.
  it2_eval  the same for the four items of eval_harness.npz's whole_it2 case (810..813): one of them sits on a near-tie
            that the float64 twin resolves the other way (5.6e-3 on R) -- refinement passes after the first see inputs
            that differ at 1e-7, so even whole mode inherits the kNN's discreteness there.
  n77, n21  whole mode, tiny clouds (items 200..): pose per run.
  pn_n256   emb_nn = pointnet, whole mode, the two pairs of pointnet_n256_b2.npz (items 160, 161): per-point features
            without any neighbourhood make weak correspondences (singular values of H down to 0.03), and the
            reference's float64 twin moves the pose by 3.9e-4 -- four times the BASELINE tolerance.
Usage: gen_selfdiv_golden.py [case ...]  (no arguments = every case; named cases are merged into the existing file)
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
stub = types.ModuleType("pynvml")
stub.nvmlInit = lambda: None
stub.nvmlDeviceGetHandleByIndex = lambda i: i
stub.nvmlDeviceGetMemoryInfo = lambda h: SimpleNamespace(used=0)
sys.modules["pynvml"] = stub
sys.path.insert(0, REF)

import model.vcrnet_model as ref_vcr            # noqa: E402
import vcrnet_amd                               # noqa: E402,F401
from vcrnet_amd import synth, weights           # noqa: E402

RUNS = [("t8", 8, torch.float32), ("t2", 2, torch.float32), ("t1", 1, torch.float32), ("f64", 8, torch.float64)]
R_TOL, T_TOL = 1e-4, 1e-5


def build(partial, dtype, emb_nn="lpdnet"):
    args = SimpleNamespace(emb_dims=512, cycle=False, emb_nn=emb_nn, pointer="transformer", vcp_nn="topK",
                           partial=partial, overlap2=synth.OVERLAP2_0575 if partial else 0.75, t3d=False, tfea=False,
                           n_blocks=1, dropout=0.0, ff_dims=1024, n_heads=4)
    net = ref_vcr.VCRNet(args)
    res = net.load_state_dict(weights.generate_weights(1234, lpd=weights.load_lpd_fixture(), emb_nn=emb_nn), strict=False)
    assert not res.missing_keys and not res.unexpected_keys
    return net.eval().to(dtype)


def run_iter(net, src, tgt, iters, partial):
    """The reference's vcrnetIter, with every forward's outputs and Tensor.topk results recorded."""
    calls, passes = [], []
    orig = torch.Tensor.topk

    def rec(t, *a, **kw):
        o = orig(t, *a, **kw)
        calls.append((o[0].detach().clone(), o[1].detach().clone()))
        return o

    def hook(mod, inp, out):
        B = inp[0].shape[0]
        c = list(calls)
        del calls[:]
        d = {"R": out[2].double().numpy().copy(), "t": out[3].double().numpy().copy()}
        if partial:
            c = c[4:]                                                   # four kNN calls (feature + Cartesian, two clouds)
            i16 = lambda i: i.reshape(B, -1).numpy().astype(np.int16)
            d["keep_dir_src"], d["keep_dir_tgt"] = i16(c[0][1]), i16(c[1][1])        # transformer.py:42, both directions
            d["sel_tgt"], d["sel_src"] = i16(c[2][1]), i16(c[3][1])                  # vcrnet_model.py:223,245
            d["argmax_tgt"], d["pair_src"] = i16(c[4][1]), i16(c[6][1])              # :297, :312
        passes.append(d)

    torch.Tensor.topk = rec
    h = net.register_forward_hook(hook)
    try:
        with torch.no_grad():
            out = ref_vcr.vcrnetIter(net, src, tgt, iter=iters)
    finally:
        h.remove()
        torch.Tensor.topk = orig
    assert len(passes) == iters
    return out[2].double().numpy(), out[3].double().numpy(), passes


def set_diff(a, b):
    return sum(len(set(x) ^ set(y)) // 2 for x, y in zip(a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])))


def flips(a, b):
    """Same accounting as tests/test_hip_forced.py:count_flips."""
    B = a["sel_src"].shape[0]
    pairs = lambda d: {(s, int(d["sel_src"][s, i]), int(d["sel_tgt"][s, d["argmax_tgt"][s, i]]))
                       for s in range(B) for i in d["pair_src"][s]}
    return (set_diff(a["keep_dir_src"], b["keep_dir_src"]) + set_diff(a["keep_dir_tgt"], b["keep_dir_tgt"]),
            set_diff(a["sel_src"], b["sel_src"]) + set_diff(a["sel_tgt"], b["sel_tgt"]), len(pairs(a) ^ pairs(b)) // 2)


def all_runs(first, B, N, iters, partial, kind="object", emb_nn="lpdnet"):
    src, tgt, R_gt, t_gt, eul = synth.make_batch(first, B, N, partial=partial, kind=kind)
    res = {}
    for name, threads, dtype in RUNS:
        torch.set_num_threads(threads)
        net = build(partial, dtype, emb_nn)
        res[name] = run_iter(net, torch.from_numpy(src).to(dtype), torch.from_numpy(tgt).to(dtype), iters, partial)
        print(f"  {name}: done", flush=True)
    torch.set_num_threads(8)
    return res, (R_gt, t_gt, eul)


def euler_mse(R, eul):
    from scipy.spatial.transform import Rotation
    e = np.asarray([Rotation.from_matrix(m).as_euler("zyx", degrees=True) for m in R])
    return float(np.mean((e - np.degrees(eul)) ** 2))


def case_c3(out, names):
    print("c3: partial, B=24, N=768, iter=3")
    B, iters = 24, 3
    res, (R_gt, t_gt, eul) = all_runs(3000, B, 1024, iters, True)
    out["c3/first"], out["c3/B"], out["c3/iters"] = np.int32(3000), np.int32(B), np.int32(iters)
    out["c3/R_final"] = np.stack([res[n][0] for n in names]).astype(np.float32)
    out["c3/t_final"] = np.stack([res[n][1] for n in names]).astype(np.float32)
    out["c3/R_iter"] = np.stack([[p["R"] for p in res[n][2]] for n in names]).astype(np.float32)     # [run, it, B, 3, 3]
    out["c3/t_iter"] = np.stack([[p["t"] for p in res[n][2]] for n in names]).astype(np.float32)
    for k in ("keep_dir_src", "keep_dir_tgt", "sel_src", "sel_tgt", "argmax_tgt", "pair_src"):        # the 8-thread run's
        out["c3/t8/" + k] = np.stack([p[k] for p in res["t8"][2]])
    # the reference against itself: every pair of runs
    pn, fl, within, med, mx = [], [], [], [], []
    for i, a in enumerate(names):
        for b in names[i + 1:]:
            f = [flips(pa, pb) for pa, pb in zip(res[a][2], res[b][2])]
            dR = np.abs(res[a][0] - res[b][0]).reshape(B, -1).max(1)
            dt = np.abs(res[a][1] - res[b][1]).reshape(B, -1).max(1)
            pn.append(f"{a}-{b}"); fl.append(f); within.append(int(((dR <= R_TOL) & (dt <= T_TOL)).sum()))
            med.append(float(np.median(dR))); mx.append(float(dR.max()))
            print(f"  {a} vs {b}: flips per iteration (keys, overlap, pairs) {f}; {within[-1]}/{B} final poses within tolerance, "
                  f"median|dR| {med[-1]:.2e} max|dR| {mx[-1]:.2e}")
    out["c3/pairings"], out["c3/flips"] = np.array(pn), np.array(fl, dtype=np.int32)          # [pairing, it, kind]
    out["c3/within_tol"], out["c3/median_dR"], out["c3/max_dR"] = np.array(within), np.array(med), np.array(mx)
    out["c3/rot_mse"] = np.array([euler_mse(res[n][0], eul) for n in names])                  # testVCRNet's rot_MSE (deg^2)
    out["c3/trans_mse"] = np.array([float(np.mean((res[n][1] - t_gt) ** 2)) for n in names])
    print("  rot_MSE per run", out["c3/rot_mse"], "trans_MSE per run", out["c3/trans_mse"])


SMALL = {"it2_n256": (400, 2, 256, 2), "it2_eval": (810, 4, 256, 2), "n77": (200, 2, 77, 1), "n21": (200, 3, 21, 1),
         "pn_n256": (160, 2, 256, 1, "pointnet")}
# round 4: the partial-overlap case of eval_harness.npz (items 830..835 as three batches of two, clouds cropped 256 -> 192,
# one pass): which of its six pairs the reference itself does not reproduce
PARTIAL_EVAL = ("partial_eval", 830, 2, 3, 256, 1)


def case_partial_eval(out, names):
    tag, first, B, nb, N, iters = PARTIAL_EVAL
    print(tag)
    Rs, ts = {n: [] for n in names}, {n: [] for n in names}
    for b in range(nb):
        res, _ = all_runs(first + b * B, B, N, iters, True)
        for n in names:
            Rs[n].append(res[n][0]); ts[n].append(res[n][1])
    R = np.stack([np.concatenate(Rs[n]) for n in names]); t = np.stack([np.concatenate(ts[n]) for n in names])
    P = B * nb
    out[f"{tag}/first"], out[f"{tag}/B"], out[f"{tag}/nbatches"], out[f"{tag}/N"], out[f"{tag}/iters"] = map(np.int32, (first, B, nb, N, iters))
    out[f"{tag}/R"], out[f"{tag}/t"] = R, t
    pr = np.max([np.abs(R[i] - R[j]).reshape(P, -1).max(1) for i in range(4) for j in range(i)], 0)
    pt = np.max([np.abs(t[i] - t[j]).reshape(P, -1).max(1) for i in range(4) for j in range(i)], 0)
    out[f"{tag}/spread_R_pair"], out[f"{tag}/spread_t_pair"] = pr, pt
    print(f"  spread over runs per pair: R {pr}  t {pt}")


def case_small(out, names, tag):
    first, B, N, iters = SMALL[tag][:4]
    print(tag)
    res, _ = all_runs(first, B, N, iters, False, emb_nn=(SMALL[tag] + ("lpdnet",))[4])
    R = np.stack([res[n][0] for n in names]); t = np.stack([res[n][1] for n in names])
    out[f"{tag}/first"], out[f"{tag}/B"], out[f"{tag}/N"], out[f"{tag}/iters"] = map(np.int32, (first, B, N, iters))
    out[f"{tag}/R"], out[f"{tag}/t"] = R, t                                               # float64 (f64 run kept exact)
    pr = np.max([np.abs(R[i] - R[j]).reshape(B, -1).max(1) for i in range(4) for j in range(i)], 0)
    pt = np.max([np.abs(t[i] - t[j]).reshape(B, -1).max(1) for i in range(4) for j in range(i)], 0)
    out[f"{tag}/spread_R_pair"], out[f"{tag}/spread_t_pair"] = pr, pt                     # per pair, max over the run pairings
    out[f"{tag}/spread_R"], out[f"{tag}/spread_t"] = np.float64(pr.max()), np.float64(pt.max())
    print(f"  spread over runs per pair: R {pr}  t {pt}")


if __name__ == "__main__":
    path = os.path.join(HERE, "selfdiv.npz")
    todo = sys.argv[1:] or ["c3"] + list(SMALL) + ["partial_eval"]
    out = {}
    if sys.argv[1:] and os.path.exists(path):
        z = np.load(path)
        out = {k: z[k] for k in z.files}
    names = [r[0] for r in RUNS]
    out["runs"] = np.array(names)
    for tag in todo:
        if tag == "c3":
            case_c3(out, names)
        elif tag == "partial_eval":
            case_partial_eval(out, names)
        else:
            case_small(out, names, tag)
    np.savez_compressed(path, **out)
    print("selfdiv.npz:", os.path.getsize(path) / 1e6, "MB")
