"""The weights axis of the fixtures, GPU side (VERDICT r3 task 1): the HIP path through the C-ABI against the
reference's recordings under three more weight regimes (vcrnet_amd.weights.regime_weights: second seed; trained-like =
peaky soft-maxes, large LayerNorm offsets; random feature extractor), in fp32 AND both split-arithmetic modes:
whole N = 1024, k = 40 at N = 512, and the partial-overlap path at N = 768 with the reference's discrete selections of
all three passes forced.

Tolerance = BASELINE (1e-4 on R, 1e-5 on t), flat.  Round 4 added the distance the reference shows from its OWN
float64 twin (it*_R_f64 / it*_t_f64, printed below); round 5's accuracy ledger (tests/test_hip_ledger.py,
profiles/accuracy_ledger.txt) measured both implementations against that twin and, with the head's scores accumulated in
blocks like ATen's sgemm, the HIP path stays within 1.5e-5 / 2e-6 of the fp32 reference in every regime: the allowance
is gone.  Free-running selection flips against the fp32 reference are bounded by 1.5x (+2) what the reference's own twin
flips against it (round 4: 3x)."""
import numpy as np
import pytest
import torch

from helpers import REGIMES, golden
from test_hip_forward import R_TOL, T_TOL, assert_mostly_close, build_net
from test_hip_forced import count_flips, golden_selections

pytestmark = pytest.mark.gpu

MODES = ["fp32", "bf16x3", "bf16x3+sdpa"]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("regime", REGIMES)
@pytest.mark.parametrize("shape", ["whole_n1024_b2", "whole_k40_n512_b1"])
def test_whole_under_regime(regime, shape, mode):
    g = golden(f"{regime}_{shape}")
    net, _ = build_net(regime=regime)
    net.emb_nn.k = int(g["k"])
    net.linear_mode = mode
    src, tgt = torch.from_numpy(g["src"]).cuda(), torch.from_numpy(g["tgt"]).cuda()
    with torch.no_grad():
        srcK, corrK, R, t, R_ba, t_ba, emb = net._forward_fused(src, tgt, want_emb=True)
    cs = int(g["cstride"])
    B, N = src.shape[0], src.shape[2]
    e = emb.cpu().view(2, B, N, 512)
    scale = max(1.0, float(np.abs(g["it0_femb_src"]).max()))       # trained regime: embeddings an order of magnitude larger
    assert_mostly_close(e[0].transpose(1, 2)[:, ::cs].numpy(), g["it0_femb_src"], atol=5e-4 * scale, hard=2e-2 * scale)
    assert_mostly_close(e[1].transpose(1, 2)[:, ::cs].numpy(), g["it0_femb_tgt"], atol=5e-4 * scale, hard=2e-2 * scale)
    assert_mostly_close(corrK.cpu().numpy(), g["it0_corrK"], atol=5e-4)
    sR = float(np.abs(g["it0_R"] - g["it0_R_f64"]).max())
    st = float(np.abs(g["it0_t"] - g["it0_t_f64"]).max())
    dR = float(np.abs(R.cpu().numpy() - g["it0_R"]).max())
    dt = float(np.abs(t.cpu().numpy() - g["it0_t"]).max())
    print(f"{regime}/{shape}/{mode}: max|dR| {dR:.2e} max|dt| {dt:.2e}  (reference vs its float64 twin {sR:.2e} / {st:.2e}; "
          f"peak cross-attention probability {float(g['it0_peak_cross_attn']):.4f})")
    assert dR <= R_TOL and dt <= T_TOL, (dR, dt, sR, st)
    np.testing.assert_allclose(R_ba.cpu().numpy(), g["it0_R_ba"], atol=R_TOL)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("regime", REGIMES)
def test_partial_forced_under_regime(regime, mode):
    g = golden(f"{regime}_partial_n768_b2_it3")
    net, _ = build_net(regime=regime, partial=True, overlap2=float(g["overlap2"]))
    net.linear_mode = mode
    assert net.fused_supported()
    tgt = torch.from_numpy(g["tgt"]).cuda()
    for it in range(int(g["iters"])):
        p = f"it{it}_"
        cur, force = torch.from_numpy(g[p + "in"]).cuda(), golden_selections(g, p)
        with torch.no_grad():
            out = net._forward_fused(cur, tgt, force=force, want_selections=True)
            free = net._forward_fused(cur, tgt, want_selections=True)
        for k, v in force.items():
            assert torch.equal(out[6][k][0].cpu(), v), k
        assert np.array_equal(out[0].cpu().numpy(), g[p + "srcK"]) and np.array_equal(out[1].cpu().numpy(), g[p + "corrK"])
        dR, dt = np.abs(out[2].cpu().numpy() - g[p + "R"]).max(), np.abs(out[3].cpu().numpy() - g[p + "t"]).max()
        fl = count_flips(free[6], force)
        fR, ft = np.abs(free[2].cpu().numpy() - g[p + "R"]).max(), np.abs(free[3].cpu().numpy() - g[p + "t"]).max()
        print(f"{regime}/partial/{mode} it{it}: forced max|dR| {dR:.2e} max|dt| {dt:.2e}; free-running flips {fl}, "
              f"max|dR| {fR:.2e} (reference vs its float64 twin, free-running: flips {g[p + 'twin_flips'].tolist()}, "
              f"max|dR| {np.abs(g[p + 'R'] - g[p + 'R_f64']).max():.2e})")
        assert dR <= R_TOL and dt <= T_TOL, (it, dR, dt)
        B, N = cur.shape[0], cur.shape[2]
        # free-running flips: the bounds of tests/test_hip_forced.py, or 1.5 times (+2) what the reference's own float64 twin
        # flips against it on this very input (it*_twin_flips; a random feature extractor leaves the hard-pair scores
        # nearly tied: the twin moves 22-37 of the 392 pairs there, 0 under the second seed).  The fp32 reference's own
        # rounding is in that count whatever the other side does; the HIP path adds less than half of it again (the
        # ledger: 7-23 flips against the twin where the reference has 22-37) -- round 4 needed 3x here (54-56 observed).
        tw = np.max([g[f"it{j}_twin_flips"] for j in range(int(g["iters"]))], axis=0)
        assert fl["keys"] <= max(2, 2 * B * N // 100, (3 * int(tw[0]) + 1) // 2 + 2), (fl, tw)
        assert fl["overlap"] <= max(4, B * N // 50, (3 * int(tw[1]) + 1) // 2 + 2), (fl, tw)
        assert fl["pairs"] <= max(2, fl["n_pairs"] // 20, (3 * int(tw[2]) + 1) // 2 + 2), (fl, tw)
        if fl["keys"] == fl["overlap"] == fl["pairs"] == 0:
            assert fR <= R_TOL and ft <= T_TOL, (it, fR, ft)


@pytest.mark.parametrize("regime", REGIMES)
def test_iter_loop_forced_under_regime(regime):
    """ONE vcr_vcrnet_iter_f32 call (device-side loop) with the reference's selections of every pass."""
    g = golden(f"{regime}_partial_n768_b2_it3")
    iters = int(g["iters"])
    net, _ = build_net(regime=regime, partial=True, overlap2=float(g["overlap2"]))
    src, tgt = torch.from_numpy(g["src"]).cuda(), torch.from_numpy(g["tgt"]).cuda()
    per = [golden_selections(g, f"it{it}_") for it in range(iters)]
    force = {k: torch.stack([d[k] for d in per]) for k in per[0]}
    with torch.no_grad():
        out = net._forward_fused(src, tgt, iters=iters, force=force)
    dR, dt = np.abs(out[2].cpu().numpy() - g["R_final"]).max(), np.abs(out[3].cpu().numpy() - g["t_final"]).max()
    print(f"{regime}: forced {iters}-pass loop max|dR| {dR:.2e} max|dt| {dt:.2e}")
    assert dR <= R_TOL and dt <= T_TOL, (dR, dt)


@pytest.mark.parametrize("name,kw", [("trained_dgcnn_n256_b2", dict(emb_nn="dgcnn")), ("trained_att_n256_b2", dict(vcp_nn="att")),
                                     ("trained_cycle_n256_b2", dict(cycle=True))])
def test_other_branches_under_the_trained_regime(name, kw):
    """DGCNN / VcpAtt / cycle under the trained-like weights, against the reference's recording (+ its float64 twin's distance)."""
    g = golden(name)
    net, _ = build_net(regime="trained", **kw)
    src, tgt = torch.from_numpy(g["src"]).cuda(), torch.from_numpy(g["tgt"]).cuda()
    with torch.no_grad():
        out = net(src, tgt)
    sR = float(np.abs(g["it0_R"] - g["it0_R_f64"]).max())
    st = float(np.abs(g["it0_t"] - g["it0_t_f64"]).max())
    dR = float(np.abs(out[2].cpu().numpy() - g["it0_R"]).max())
    dt = float(np.abs(out[3].cpu().numpy() - g["it0_t"]).max())
    print(f"{name}: max|dR| {dR:.2e} max|dt| {dt:.2e} (reference vs its float64 twin {sR:.2e} / {st:.2e})")
    assert dR <= R_TOL and dt <= T_TOL, (dR, dt, sR, st)
    if kw.get("cycle"):                                  # the second head's own solve (vcrnet_model.py:511-513)
        np.testing.assert_allclose(out[4].cpu().numpy(), g["it0_R_ba"], atol=R_TOL)
        np.testing.assert_allclose(out[5].cpu().numpy(), g["it0_t_ba"], atol=T_TOL + 1e-5)
