"""The weights axis of the fixtures, CPU side: the oracle against the reference's recordings under the three
non-default weight regimes of vcrnet_amd.weights.regime_weights (second seed; trained-like = peaky soft-maxes and
large LayerNorm offsets; random feature extractor) -- whole N = 1024, k = 40, and the partial-overlap path with every
discrete selection of three teacher-forced passes (tests/golden/gen_golden.py, round-4 cases).  The reference loads
arbitrary checkpoints (util/initPara.py:248-254): one point in weight space pins nothing about the others."""
import numpy as np
import pytest
import torch

import oracle
from helpers import REGIMES, cfg_weights, golden, set_mismatch
from test_oracle_golden import check_common

torch.set_num_threads(8)


def regime_case(name, **cfgkw):
    g = golden(name)
    w = cfg_weights(str(g["regime"]))
    rec = {}
    return g, w, oracle.OracleConfig(k=int(g["k"]), overlap2=float(g["overlap2"]), record=rec, **cfgkw), rec


@pytest.mark.parametrize("regime", REGIMES)
@pytest.mark.parametrize("shape", ["whole_n1024_b2", "whole_k40_n512_b1"])
def test_whole_under_regime(regime, shape):
    g, w, cfg, rec = regime_case(f"{regime}_{shape}")
    assert str(g["regime"]) == regime
    out = oracle.vcrnet_forward(w, torch.from_numpy(g["src"]), torch.from_numpy(g["tgt"]), cfg)
    check_common(g, rec, out)


@pytest.mark.parametrize("regime", REGIMES)
def test_partial_teacher_forced_under_regime(regime):
    g, w, cfg, rec = regime_case(f"{regime}_partial_n768_b2_it3", partial=True)
    tgt = torch.from_numpy(g["tgt"])
    for it in range(int(g["iters"])):
        p = f"it{it}_"
        rec.clear()
        out = oracle.vcrnet_forward(w, torch.from_numpy(g[p + "in"]), tgt, cfg)
        for mine, theirs in (("sel_tgt", "sel_tgt"), ("sel_src", "sel_src"), ("key_keep_src", "keep_dir_src"),
                             ("key_keep_tgt", "keep_dir_tgt")):
            assert set_mismatch(rec[mine].numpy(), g[p + theirs]) == 0, (it, mine)
        assert np.array_equal(rec["pair_src"].numpy(), g[p + "pair_src"])
        assert np.array_equal(rec["argmax_tgt"].numpy(), g[p + "argmax_tgt"])
        check_common(g, rec, out, p=p)


def test_the_trained_regime_is_peaky_and_offset():
    """What the regime is for: cross-attention rows two orders of magnitude more peaked than the default recipe's
    near-uniform ones (recorded by the reference run), LayerNorm gains / offsets far from (1, 0)."""
    peak_trained = float(golden("trained_whole_n1024_b2")["it0_peak_cross_attn"])
    peak_seed = float(golden("seed4321_whole_n1024_b2")["it0_peak_cross_attn"])
    assert peak_trained > 50 * peak_seed and peak_seed < 2.0 / 1024
    w = cfg_weights("trained")
    b2 = torch.cat([v.flatten() for k, v in w.items() if k.endswith(".b_2")])
    a2 = torch.cat([v.flatten() for k, v in w.items() if k.endswith(".a_2")])
    assert b2.abs().max() > 1.5 and a2.min() < 0.55 and a2.max() > 1.95
    d = cfg_weights("default")
    assert all(torch.equal(d[k], w[k]) for k in d if k.startswith("emb_nn.") and "conv3" not in k)   # LPD fixture kept
    r = cfg_weights("randemb")
    assert not torch.equal(r["emb_nn.convDG1.0.weight"], d["emb_nn.convDG1.0.weight"])


@pytest.mark.parametrize("name,kw", [("trained_dgcnn_n256_b2", dict(emb_nn="dgcnn")), ("trained_att_n256_b2", dict(vcp_nn="att")),
                                     ("trained_cycle_n256_b2", dict(cycle=True))])
def test_other_branches_under_the_trained_regime(name, kw):
    """The trained-like regime through the constructor's other branches: DGCNN embedding (BatchNorm folded), the VcpAtt head,
    cycle (second head with the roles swapped)."""
    g = golden(name)
    wkw = {k: kw[k] for k in ("emb_nn", "vcp_nn") if k in kw}
    w = cfg_weights("trained", **wkw)
    rec = {}
    cfg = oracle.OracleConfig(k=int(g["k"]), overlap2=float(g["overlap2"]), record=rec, **kw)
    out = oracle.vcrnet_forward(w, torch.from_numpy(g["src"]), torch.from_numpy(g["tgt"]), cfg)
    check_common(g, rec, out, lpd=kw.get("emb_nn", "lpdnet") == "lpdnet")
