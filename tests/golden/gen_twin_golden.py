#!/usr/bin/env python
"""The reference's float64 twin, recorded for the accuracy ledger (VERDICT r4 task 3).

For every regime fixture written by gen_golden.py (it holds the fp32 reference's recordings and the twin's poses) this
script runs the REFERENCE model in float64 -- same code, same weights, same recorded inputs (it*_in) -- once more and
stores what the ledger compares against: the twin's final embeddings (the head's inputs, strided like the fp32 ones), its
pose, and in partial mode its discrete selections (kept keys, overlap sets, arg-max targets and values, hard pairs).
Output: tests/golden/<fixture>_twin.npz.  Build container only (imports /root/reference); the GPU box reads the files.
Usage:  python tests/golden/gen_twin_golden.py [fixture ...]"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G                                   # noqa: E402  (stub pynvml, reference imports, helpers)
from vcrnet_amd import synth, weights                    # noqa: E402

G.LPD = weights.load_lpd_fixture()

REGIMES = ("seed4321", "trained", "randemb")
SHAPES = {"whole_n1024_b2": dict(partial=False, k=None), "whole_k40_n512_b1": dict(partial=False, k=40),
          "partial_n768_b2_it3": dict(partial=True, k=None)}


def run(name):
    regime, shape = name.split("_", 1)
    cfg = SHAPES[shape]
    g = np.load(os.path.join(HERE, name + ".npz"))
    overlap2 = synth.OVERLAP2_0575 if cfg["partial"] else 0.75
    args = G.ref_args(partial=cfg["partial"], overlap2=overlap2)
    net = G.ref_vcr.VCRNet(args)
    G.load_into(net, weights.regime_weights(regime, lpd=G.LPD, emb_nn="lpdnet", vcp_nn="topK", pointer="transformer", n_blocks=1))
    if cfg["k"] is not None:
        net.emb_nn.k = cfg["k"]
    twin = copy.deepcopy(net).double()
    cs = int(g["cstride"])
    tgt = torch.from_numpy(g["tgt"]).double()
    out = {}
    with torch.no_grad():
        for it in range(int(g["iters"])):
            p = f"it{it}_"
            cur = torch.from_numpy(g[p + "in"]).double()
            store, handles = G.hook_outputs({"head": twin.head})
            with G.TopkLog() as tl:
                o = twin(cur, tgt)
            for h in handles:
                h.remove()
            assert np.array_equal(o[2].numpy(), g[p + "R_f64"]) and np.array_equal(o[3].numpy(), g[p + "t_f64"]), name
            h_in = store["head"][0][0]
            out[p + "femb_src"] = h_in[0][:, ::cs].contiguous().numpy()          # float64, strided like it*_femb_src
            out[p + "femb_tgt"] = h_in[1][:, ::cs].contiguous().numpy()
            if cfg["partial"]:
                B = cur.shape[0]
                c = [(v.reshape(B, -1).numpy(), i.reshape(B, -1).numpy()) for v, i in tl.calls[4:]]     # after the 4 kNN calls
                for nm, (v, i) in zip(("keep_dir_src", "keep_dir_tgt", "sel_tgt", "sel_src", "argmax_tgt", "argmax_val", "pair_src"), c):
                    out[p + nm] = v if nm == "argmax_val" else i.astype(np.int16)
                # the sanity check of the recorded flip counts (gen_golden.py computed them from this same twin)
                sd = lambda a, b: sum(len(set(x) ^ set(y)) // 2 for x, y in zip(a, b))
                assert sd(out[p + "sel_src"], g[p + "sel_src"]) + sd(out[p + "sel_tgt"], g[p + "sel_tgt"]) == int(g[p + "twin_flips"][1]), name
    np.savez_compressed(os.path.join(HERE, name + "_twin.npz"), **out)
    print("wrote", name + "_twin.npz", {k: v.shape for k, v in list(out.items())[:4]}, flush=True)


if __name__ == "__main__":
    names = sys.argv[1:] or [f"{r}_{s}" for r in REGIMES for s in SHAPES]
    for n in names:
        run(n)
