"""Pin the CPU oracle against vectors recorded from the reference itself
(tests/golden/gen_golden.py).  Tolerances: the reference's own fp32 noise floor is
<= 2e-6 on R and <= 1e-6 on t; the oracle uses the same primitives so
we require 1e-5 on activations and the BASELINE tolerances (1e-4 / 1e-5) on (R, t)."""
import numpy as np
import pytest
import torch

import oracle
from helpers import cfg_weights, golden, set_mismatch, sl

torch.set_num_threads(8)


def run_case(name, **cfgkw):
    g = golden(name)
    wkw = {k: cfgkw[k] for k in ("emb_nn", "vcp_nn", "pointer", "n_blocks") if k in cfgkw}
    w = cfg_weights(**wkw)
    rec = {}
    cfg = oracle.OracleConfig(k=int(g["k"]), overlap2=float(g["overlap2"]), record=rec, **cfgkw)
    return g, w, cfg, rec


def check_common(g, rec, out, p="it0_", cs=None, lpd=True, pointer=True):
    cs = int(g["cstride"]) if cs is None else cs
    srcK, corrK, R, t, R_ba, t_ba = out
    if lpd:
        for cloud in ("src", "tgt"):
            r = rec["emb_" + cloud]
            assert set_mismatch(r["idx_feat"].numpy(), g[p + f"idx_feat_{cloud}"]) == 0
            assert set_mismatch(r["idx_xyz"].numpy(), g[p + f"idx_xyz_{cloud}"]) == 0
            np.testing.assert_allclose(sl(r["x64"], max(1, cs // 2)), g[p + f"x64_{cloud}"], atol=1e-5)
            for nm in ("x1", "x2", "x3"):
                np.testing.assert_allclose(sl(r[nm], cs), g[p + f"{nm}_{cloud}"], atol=1e-5)
    np.testing.assert_allclose(sl(rec["src_emb0"], cs), g[p + "emb0_src"], atol=1e-5)
    np.testing.assert_allclose(sl(rec["tgt_emb0"], cs), g[p + "emb0_tgt"], atol=1e-5)
    if pointer:
        np.testing.assert_allclose(sl(rec["memory_src"].transpose(2, 1), cs), g[p + "mem_src"], atol=2e-5)
        np.testing.assert_allclose(sl(rec["memory_tgt"].transpose(2, 1), cs), g[p + "mem_tgt"], atol=2e-5)
    np.testing.assert_allclose(sl(rec["src_emb"], cs), g[p + "femb_src"], atol=2e-5)
    np.testing.assert_allclose(sl(rec["tgt_emb"], cs), g[p + "femb_tgt"], atol=2e-5)
    np.testing.assert_allclose(rec["H"].numpy(), g[p + "H"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(corrK.numpy(), g[p + "corrK"], atol=1e-5)
    np.testing.assert_allclose(srcK.numpy(), g[p + "srcK"], atol=0)
    np.testing.assert_allclose(R.numpy(), g[p + "R"], atol=1e-4)
    np.testing.assert_allclose(t.numpy(), g[p + "t"], atol=1e-5)
    np.testing.assert_allclose(R_ba.numpy(), g[p + "R_ba"], atol=1e-4)
    np.testing.assert_allclose(t_ba.numpy(), g[p + "t_ba"], atol=1e-5)


@pytest.mark.parametrize("name", ["whole_n256_b2", "whole_n1024_b2", "whole_k40_n512_b1", "whole_k40_n4096_b2"])
def test_whole(name):
    g, w, cfg, rec = run_case(name)
    out = oracle.vcrnet_forward(w, torch.from_numpy(g["src"]), torch.from_numpy(g["tgt"]), cfg)
    check_common(g, rec, out)


@pytest.mark.parametrize("name,kw", [("cycle_n256_b2", {}), ("attcycle_n256_b2", dict(vcp_nn="att")),
                                     ("distcycle_n256_b2", dict(vcp_nn="dist"))])
def test_cycle(name, kw):
    """args.cycle (vcrnet_model.py:511-513): the second head runs with the ROLES swapped -- for VcpAtt that means
    linears_emb[0] on the target embeddings -- and its solve is (R_ba, t_ba)."""
    g, w, cfg, rec = run_case(name, cycle=True, **kw)
    out = oracle.vcrnet_forward(w, torch.from_numpy(g["src"]), torch.from_numpy(g["tgt"]), cfg)
    check_common(g, rec, out)
    # vcrnetIter discards the cycle head's pose and returns the inverse of the composed one (vcrnet_model.py:40-41)
    it = oracle.vcrnet_iter(w, torch.from_numpy(g["src"]), torch.from_numpy(g["tgt"]), cfg, iters=1)
    np.testing.assert_allclose(it[4].numpy(), np.transpose(g["it0_R"], (0, 2, 1)), atol=1e-4)
    assert np.abs(it[4].numpy() - g["it0_R_ba"]).max() > 1e-3      # ... which is NOT the cycle head's


@pytest.mark.parametrize("name,kw", [("att_n256_b2", dict(vcp_nn="att")), ("dist_n256_b2", dict(vcp_nn="dist")),
                                     ("identity_n256_b2", dict(pointer="identity"))])
def test_alt_heads(name, kw):
    g, w, cfg, rec = run_case(name, **kw)
    out = oracle.vcrnet_forward(w, torch.from_numpy(g["src"]), torch.from_numpy(g["tgt"]), cfg)
    check_common(g, rec, out, pointer=kw.get("pointer", "transformer") == "transformer")


def test_dgcnn():
    g, w, cfg, rec = run_case("dgcnn_n256_b2", emb_nn="dgcnn")
    out = oracle.vcrnet_forward(w, torch.from_numpy(g["src"]), torch.from_numpy(g["tgt"]), cfg)
    for cloud in ("src", "tgt"):
        assert set_mismatch(rec["emb_" + cloud]["idx_xyz"].numpy(), g[f"it0_idx_xyz_{cloud}"]) == 0
    check_common(g, rec, out, lpd=False)


@pytest.mark.parametrize("name", ["partial_n192_b2_it2", "partial_n768_b2_it3"])
def test_partial_teacher_forced(name):
    g, w, cfg, rec = run_case(name, partial=True)
    tgt = torch.from_numpy(g["tgt"])
    iters = int(g["iters"])
    for it in range(iters):
        p = f"it{it}_"
        rec.clear()
        out = oracle.vcrnet_forward(w, torch.from_numpy(g[p + "in"]), tgt, cfg)
        assert set_mismatch(rec["sel_tgt"].numpy(), g[p + "sel_tgt"]) == 0
        assert set_mismatch(rec["sel_src"].numpy(), g[p + "sel_src"]) == 0
        assert np.array_equal(rec["pair_src"].numpy(), g[p + "pair_src"])
        # decoder key pruning, named by the cloud whose points are the keys (model(src,tgt): memory = enc(src))
        assert set_mismatch(rec["key_keep_src"].numpy(), g[p + "keep_dir_src"]) == 0
        assert set_mismatch(rec["key_keep_tgt"].numpy(), g[p + "keep_dir_tgt"]) == 0
        assert np.array_equal(rec["argmax_tgt"].numpy(), g[p + "argmax_tgt"])
        check_common(g, rec, out, p=p)
    # free-running iteration reproduces the composed pose too (same machine, same primitives)
    fr = oracle.vcrnet_iter(w, torch.from_numpy(g["src"]), tgt, cfg, iters=iters)
    np.testing.assert_allclose(fr[2].numpy(), g["R_final"], atol=1e-4)
    np.testing.assert_allclose(fr[3].numpy(), g["t_final"], atol=1e-5)


def test_dgcnn_partial():
    g, w, cfg, rec = run_case("dgcnn_partial_n192_b2", emb_nn="dgcnn", partial=True)
    out = oracle.vcrnet_forward(w, torch.from_numpy(g["src"]), torch.from_numpy(g["tgt"]), cfg)
    assert set_mismatch(rec["sel_tgt"].numpy(), g["it0_sel_tgt"]) == 0
    assert set_mismatch(rec["sel_src"].numpy(), g["it0_sel_src"]) == 0
    assert np.array_equal(rec["pair_src"].numpy(), g["it0_pair_src"])
    check_common(g, rec, out, lpd=False)


def test_pointnet():
    """emb_nn = pointnet (model/vcrnet_model.py:65-87, :468-469): whole mode, and the partial-overlap path with every
    selection of both iterations equal to the reference's."""
    g, w, cfg, rec = run_case("pointnet_n256_b2", emb_nn="pointnet")
    out = oracle.vcrnet_forward(w, torch.from_numpy(g["src"]), torch.from_numpy(g["tgt"]), cfg)
    check_common(g, rec, out, lpd=False)
    g, w, cfg, rec = run_case("pointnet_partial_n192_b2_it2", emb_nn="pointnet", partial=True)
    tgt = torch.from_numpy(g["tgt"])
    for it in range(int(g["iters"])):
        p = f"it{it}_"
        rec.clear()
        out = oracle.vcrnet_forward(w, torch.from_numpy(g[p + "in"]), tgt, cfg)
        for nm in ("sel_tgt", "sel_src"):
            assert set_mismatch(rec[nm].numpy(), g[p + nm]) == 0
        assert set_mismatch(rec["key_keep_src"].numpy(), g[p + "keep_dir_src"]) == 0
        assert set_mismatch(rec["key_keep_tgt"].numpy(), g[p + "keep_dir_tgt"]) == 0
        assert np.array_equal(rec["pair_src"].numpy(), g[p + "pair_src"])
        check_common(g, rec, out, p=p, lpd=False)


def test_two_transformer_blocks():
    """args.n_blocks = 2 (model/transformer.py:245,257-259): two encoder and two decoder layers, whole mode and
    partial-overlap mode with every decoder layer's kept keys equal to the reference's."""
    g, w, cfg, rec = run_case("nblocks2_n256_b2", n_blocks=2)
    out = oracle.vcrnet_forward(w, torch.from_numpy(g["src"]), torch.from_numpy(g["tgt"]), cfg)
    check_common(g, rec, out)
    g, w, cfg, rec = run_case("nblocks2_partial_n192_b2", n_blocks=2, partial=True)
    out = oracle.vcrnet_forward(w, torch.from_numpy(g["src"]), torch.from_numpy(g["tgt"]), cfg)
    for layer in (0, 1):
        assert set_mismatch(rec[f"key_keep_src_l{layer}"].numpy(), g[f"it0_keep_dir_src_l{layer}"]) == 0
        assert set_mismatch(rec[f"key_keep_tgt_l{layer}"].numpy(), g[f"it0_keep_dir_tgt_l{layer}"]) == 0
    for nm in ("sel_tgt", "sel_src"):
        assert set_mismatch(rec[nm].numpy(), g["it0_" + nm]) == 0
    assert np.array_equal(rec["pair_src"].numpy(), g["it0_pair_src"])
    check_common(g, rec, out)


def test_dcp():
    g = golden("dcp_n256_b2")
    w = cfg_weights()
    cfg = oracle.OracleConfig()
    R, t, R_ba, t_ba, s, corr = oracle.dcp_forward(w, torch.from_numpy(g["src"]), torch.from_numpy(g["tgt"]), cfg)
    np.testing.assert_allclose(corr.numpy(), g["corr"], atol=1e-5)
    np.testing.assert_allclose(R.numpy(), g["R"], atol=1e-4)
    np.testing.assert_allclose(t.numpy(), g["t"], atol=1e-5)
    np.testing.assert_allclose(t_ba.numpy(), g["t_ba"], atol=1e-5)


def test_svd_known_answer_and_reflection():
    """Analytic pins: exact rigid pair -> generating (R,t); det R = +1 even when
    the unconstrained optimum is a reflection (exercises vcrnet_model.py:382-386)."""
    rs = np.random.RandomState(0)
    src = torch.from_numpy(rs.uniform(-1, 1, (3, 3, 200)).astype(np.float32))
    th = 0.7
    Rz = torch.tensor([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], dtype=torch.float32)
    t0 = torch.tensor([0.1, -0.2, 0.3])
    corr = torch.matmul(Rz, src) + t0.view(1, 3, 1)
    R, t = oracle.rigid_svd(src, corr)
    np.testing.assert_allclose(R.numpy(), np.broadcast_to(Rz.numpy(), (3, 3, 3)), atol=1e-5)
    np.testing.assert_allclose(t.numpy(), np.broadcast_to(t0.numpy(), (3, 3)), atol=1e-5)
    mirror = torch.diag(torch.tensor([1.0, 1.0, -1.0]))
    R2, _ = oracle.rigid_svd(src, torch.matmul(mirror, src))
    assert torch.allclose(torch.det(R2), torch.ones(3), atol=1e-5)


def test_edgeconv_split_identity():
    """max_j relu(W[x_j;x_i]+b) == relu(max_j Wn x_j + Wc x_i + b)."""
    w = cfg_weights()
    rs = np.random.RandomState(1)
    x = torch.from_numpy(rs.normal(size=(1, 64, 96)).astype(np.float32))
    idx = oracle.knn_indices(x, 20)
    W, b = w["emb_nn.convDG1.0.weight"].view(128, 128), w["emb_nn.convDG1.0.bias"]
    g = oracle.graph_feature(x, idx)
    full = torch.relu(torch.nn.functional.conv2d(g, W.view(128, 128, 1, 1), b)).max(dim=-1)[0]
    P = torch.matmul(W[:, :64], x)[0]          # [128, N]
    Q = torch.matmul(W[:, 64:], x)[0] + b.view(-1, 1)
    split = torch.relu(P[:, idx[0]].max(dim=-1)[0] + Q)
    np.testing.assert_allclose(split.numpy(), full[0].numpy(), atol=2e-6)
