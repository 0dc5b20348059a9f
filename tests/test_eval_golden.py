"""Eval-harness arithmetic against the REFERENCE's own test_one_epoch / testVCRNet
(model/vcrnet_model.py:521-649, 768-815), recorded by tests/golden/gen_eval_golden.py into eval_harness.npz.

CPU: EvalAccumulator fed with the six tensors the reference's vcrnetIter / vcrnetIcpNet returned per batch must
reproduce every one of test_one_epoch's 17 return values, testVCRNet's A->B / B->A figures and its printed line.
GPU: evaluate.main (the counterpart of main.py --eval) on the same items through the HIP path, against the same record.
"""
import numpy as np
import pytest
import torch

import vcrnet_amd  # noqa: F401
from helpers import golden
from vcrnet_amd import evalmetrics, synth

CASES = ["whole_pose", "whole_point", "whole_mix", "whole_it2", "cycle_pose", "attcycle_pose", "partial_it1", "icp_it0"]
RETURNS = ("loss_pose", "cycle_loss", "mse_ab", "mae_ab", "mse_ba", "mae_ba", "rotations_ab", "translations_ab",
           "rotations_ab_pred", "translations_ab_pred", "rotations_ba", "translations_ba", "rotations_ba_pred",
           "translations_ba_pred", "eulers_ab", "eulers_ba", "loss_vcrnet")
# testVCRNet's local -> key of EvalAccumulator.final() / final_ba()
AB = {"test_loss_VCRNet": "loss", "test_loss_Pose": "loss_pose", "test_cycle_loss_Pose": "cycle_loss", "test_mse_ab": "mse",
      "test_rmse_ab": "rmse", "test_mae_ab": "mae", "test_r_mse_ab": "rot_mse", "test_r_rmse_ab": "rot_rmse",
      "test_r_mae_ab": "rot_mae", "test_t_mse_ab": "trans_mse", "test_t_rmse_ab": "trans_rmse", "test_t_mae_ab": "trans_mae"}
BA = {"test_mse_ba": "mse", "test_rmse_ba": "rmse", "test_mae_ba": "mae", "test_r_mse_ba": "rot_mse",
      "test_r_rmse_ba": "rot_rmse", "test_r_mae_ba": "rot_mae", "test_t_mse_ba": "trans_mse", "test_t_rmse_ba": "trans_rmse",
      "test_t_mae_ba": "trans_mae"}


def case(g, name):
    c = {k[len(name) + 1:]: g[k] for k in g if k.startswith(name + "/")}
    c.update(first=int(c["first"]), batch=int(c["batch"]), nbatches=int(c["nbatches"]), N=int(c["N"]), iters=int(c["iters"]),
             cycle=bool(c["cycle"]), partial=bool(c["partial"]), loss=str(c["loss"]), vcp_nn=str(c["vcp_nn"]))
    return c


@pytest.mark.parametrize("name", CASES)
def test_accumulator_reproduces_the_reference_harness(name):
    c = case(golden("eval_harness"), name)
    acc = evalmetrics.EvalAccumulator(cycle=c["cycle"], loss=c["loss"])
    T = torch.from_numpy
    for b in range(c["nbatches"]):
        src, tgt, R, t, eul = synth.make_batch(c["first"] + b * c["batch"], c["batch"], c["N"], partial=c["partial"])
        out = tuple(T(c[f"b{b}/{nm}"]) for nm in ("srcK", "corrK", "R", "t", "R_ba", "t_ba"))
        acc.add_batch(T(src), T(tgt), T(R), T(t), T(eul), out)
    ret = acc.returns()
    assert len(ret) == len(RETURNS)
    for nm, mine in zip(RETURNS, ret):
        np.testing.assert_allclose(np.asarray(mine, dtype=np.float64), c["ret/" + nm].astype(np.float64), rtol=1e-6, atol=1e-7,
                                   err_msg=nm)
    m, mb = acc.final(), acc.final_ba()
    for loc, key in AB.items():
        assert abs(m[key] - float(c["final/" + loc])) <= 1e-6 * max(1.0, abs(float(c["final/" + loc]))), (loc, m[key])
    for loc, key in BA.items():
        assert abs(mb[key] - float(c["final/" + loc])) <= 1e-6 * max(1.0, abs(float(c["final/" + loc]))), (loc, mb[key])
    lines = [str(x) for x in c["lines"]]
    assert lines[:2] == ["==FINAL TEST==", "A--------->B"]
    assert evalmetrics.EvalAccumulator.format_final(m) == lines[2]            # the reference's printed line, verbatim
    if c["cycle"]:      # the reference prints the header, then raises on its own format string (:801-806)
        assert lines[3] == "B--------->A" and str(c["raised"]).startswith("TypeError")
    else:
        assert len(lines) == 3


def _floats(line):
    import re
    return np.array([float(x) for x in re.findall(r": (-?\d+\.\d+)", line)])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_evaluate_main_vs_the_reference_harness(name):
    """evaluate.main = main.py --eval -> testVCRNet with the HIP path underneath, on the recorded items.  Whole-mode
    cases: every predicted pose within the BASELINE tolerance of the reference's, the aggregates to 1e-3 relative.
    The partial-overlap and ICP cases contain discrete decisions (hard pairs; nearest neighbours + an iteration count)
    that the reference does not reproduce against itself across thread counts (tests/golden/partial_selfdiv.npz):
    poses are compared per pair, and pairs without a flip are inside the tolerance."""
    import evaluate
    c = case(golden("eval_harness"), name)
    argv = ["--items", str(c["batch"] * c["nbatches"]), "--batch", str(c["batch"]), "--points", str(c["N"]), "--iters",
            str(c["iters"]), "--first-item", str(c["first"]), "--loss", c["loss"], "--vcp-nn", c["vcp_nn"]]
    argv += ["--partial"] if c["partial"] else []
    argv += ["--cycle"] if c["cycle"] else []
    res = evaluate.main(argv)
    ret = dict(zip(RETURNS, res["returns"]))
    for nm in ("rotations_ab", "translations_ab", "rotations_ba", "translations_ba", "eulers_ab", "eulers_ba"):   # labels
        np.testing.assert_allclose(ret[nm], c["ret/" + nm], atol=1e-6, err_msg=nm)
    dR = np.abs(ret["rotations_ab_pred"] - c["ret/rotations_ab_pred"]).reshape(len(ret["rotations_ab_pred"]), -1).max(1)
    dt = np.abs(ret["translations_ab_pred"] - c["ret/translations_ab_pred"]).max(1)
    print(f"{name}: max|dR| {dR.max():.2e} max|dt| {dt.max():.2e} vs the reference's test_one_epoch")
    mine = _floats(evalmetrics.EvalAccumulator.format_final(res["ab"]))
    ref = _floats(str(c["lines"][2]))
    assert mine.shape == ref.shape == (12,)
    if c["iters"] == 0:
        # ICP refinement (--iter 0): nearest neighbours + an iteration count are discrete, and agree on every recorded pair
        assert (dR <= 1e-4).all() and (dt <= 1e-5).all(), (dR, dt)
        np.testing.assert_allclose(mine, ref, rtol=1e-3, atol=2e-6)
    elif c["partial"]:
        # The reference reproduces these six pairs against itself to 4e-7 (tests/golden/selfdiv.npz, partial_eval: 8 / 2 / 1
        # threads and its float64 twin) -- but one of them hangs on a near-tie: the hard pairs are the K' sources with the
        # largest soft-max peak (vcrnet_model.py:312), a 512-d negative-distance soft-max turns 1e-6 of embedding rounding
        # into ~1e-4 of peak value (test_hip_forward.assert_mostly_close), and pair 831 ranks its last-in / first-out
        # candidates 8.7e-5 apart.  The oracle (pinned to the reference) gives every pair's margin at that boundary:
        # pairs decided by more than 5e-4 must match the reference's pose, the others may flip one hard pair.
        import oracle
        from vcrnet_amd import synth
        from helpers import cfg_weights
        w = cfg_weights()
        margin = []
        for b in range(c["nbatches"]):
            src, tgt, _, _, _ = synth.make_batch(c["first"] + b * c["batch"], c["batch"], c["N"], partial=True)
            rec = {}
            oracle.vcrnet_forward(w, torch.from_numpy(src), torch.from_numpy(tgt),
                                  oracle.OracleConfig(partial=True, overlap2=synth.OVERLAP2_0575, record=rec))
            kp = rec["pair_src"].shape[1]
            sv = torch.sort(rec["pair_val"], dim=1, descending=True)[0]
            margin += ((sv[:, kp - 1] - sv[:, kp]) / sv[:, kp]).tolist()
        margin = np.array(margin)
        sd = golden("selfdiv")
        assert int(sd["partial_eval/first"]) == c["first"] and len(sd["partial_eval/spread_R_pair"]) == len(dR)
        ok = (dR <= 1e-4 + sd["partial_eval/spread_R_pair"]) & (dt <= 1e-5 + sd["partial_eval/spread_t_pair"])
        print(f"{name}: hard-pair boundary margins {np.array2string(margin, precision=2)}; pairs off the reference: {np.flatnonzero(~ok)}")
        assert (ok | (margin < 5e-4)).all(), (dR, dt, margin)
        np.testing.assert_allclose(mine, ref, rtol=1e-3 if ok.all() else 2e-2, atol=2e-6)
    else:
        tol_R, tol_t = np.full(len(dR), 1e-4), np.full(len(dt), 1e-5 * max(1, c["iters"]))   # per pass; composed passes add up
        rtol = 1e-3
        if name == "whole_it2":
            # passes after the first start from inputs that differ at 1e-7, so a kNN near-tie can resolve differently: the
            # reference's own float64 twin moves pair 811 by 5.6e-3 (tests/golden/selfdiv.npz, it2_eval).  Tolerance per
            # pair = BASELINE + what the reference's own runs differ by on that pair
            sd = golden("selfdiv")
            assert int(sd["it2_eval/first"]) == c["first"] and int(sd["it2_eval/B"]) == len(dR)
            tol_R, tol_t = tol_R + sd["it2_eval/spread_R_pair"], tol_t + sd["it2_eval/spread_t_pair"]
            rtol = 2e-2
        assert (dR <= tol_R).all() and (dt <= tol_t).all(), (dR, dt, tol_R, tol_t)
        np.testing.assert_allclose(mine, ref, rtol=rtol, atol=2e-6)
