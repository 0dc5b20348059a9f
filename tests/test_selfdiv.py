"""Tests whose tolerance is NOT the plain BASELINE one carry the reference's own spread next to it: tests/golden/selfdiv.npz
(written by tests/golden/gen_selfdiv_golden.py) holds the unmodified reference run on the same inputs with 8 / 2 / 1 CPU
threads and in float64.  The HIP path is compared with the recorded 8-thread fp32 run and must stay inside the
BASELINE tolerance widened by what the reference shows against itself.

Finding recorded there: at B = 24 (BASELINE configs[2]) the reference is bit-identical across thread counts, so the
only self-distance it offers is fp32 vs its own float64 twin -- the same code with rounding errors 2^29 times smaller.
That twin flips (keys, overlap, hard pairs) = (3, 2, 3) selections in the first pass, (8, 26, 36) in the second,
(27, 84, 139) in the third, and leaves 15 of the 24 final poses within 1e-4 / 1e-5: the partial path amplifies ANY
fp32-level perturbation that much.  An fp32 implementation with a different (fixed) summation order is one such
perturbation."""
import numpy as np
import pytest
import torch

from helpers import golden
from test_hip_forward import build_net, R_TOL, T_TOL
from test_hip_forced import count_flips

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["it2_n256", "n77", "n21"])
def test_whole_mode_small_cases_vs_the_recorded_reference(tag):
    """vcrnetIter(iter=2) at N = 256 and single passes on tiny clouds (N = 77, N = 21 = k+1): HIP vs the recorded
    8-thread reference run; tolerance = BASELINE + the spread the reference shows over its own four runs."""
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import synth
    from vcrnet_amd.module import vcrnetIter
    g = golden("selfdiv")
    first, B, N, iters = (int(g[f"{tag}/{k}"]) for k in ("first", "B", "N", "iters"))
    net, _ = build_net()
    src, tgt, _, _, _ = synth.make_batch(first, B, N)
    with torch.no_grad():
        out = vcrnetIter(net, torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda(), iter=iters)
    R_ref, t_ref = g[f"{tag}/R"][0], g[f"{tag}/t"][0]                  # run 0 = 8 threads, fp32
    dR = np.abs(out[2].cpu().numpy() - R_ref).max()
    dt = np.abs(out[3].cpu().numpy() - t_ref).max()
    sR, st = float(g[f"{tag}/spread_R"]), float(g[f"{tag}/spread_t"])
    print(f"{tag}: HIP vs reference max|dR| {dR:.2e} max|dt| {dt:.2e}; reference vs itself {sR:.2e} / {st:.2e}")
    assert dR <= R_TOL + sR and dt <= T_TOL + st, (dR, dt, sR, st)


def test_config3_free_running_inside_the_references_own_envelope():
    """BASELINE configs[2] (B = 24, clouds cropped 1024 -> 768, iter = 3), ONE free-running vcr_vcrnet_iter_f32 call
    against the recorded reference run: selection flips in the first pass (the only pass whose inputs are identical),
    poses within tolerance and the aggregate rot / trans MSE, each bounded by what the reference's float64
    twin shows against the same run."""
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import evalmetrics, synth
    from vcrnet_amd.module import vcrnetIter
    g = golden("selfdiv")
    B, iters, first = int(g["c3/B"]), int(g["c3/iters"]), int(g["c3/first"])
    runs = [str(x) for x in g["runs"]]
    pairings = [str(x) for x in g["c3/pairings"]]
    i8, i64, pj = runs.index("t8"), runs.index("f64"), pairings.index("t8-f64")
    net, _ = build_net(partial=True, overlap2=synth.OVERLAP2_0575)
    src, tgt, R_gt, t_gt, eul = synth.make_batch(first, B, 1024, partial=True)
    s, t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    with torch.no_grad():
        out = net._forward_fused(s, t, iters=iters, want_selections=True, iter_api=True)
        pub = vcrnetIter(net, s, t, iter=iters)
    assert torch.equal(pub[2], out[2]) and torch.equal(pub[3], out[3])          # the public wrapper is that same call
    i32 = lambda a: torch.from_numpy(a.astype(np.int32))
    env = g["c3/flips"][pj]                                                     # [iteration, (keys, overlap, pairs)]
    for it in range(iters):
        p = lambda k: g[f"c3/t8/{k}"][it]
        ref = {"keys": torch.cat((i32(p("keep_dir_src")), i32(p("keep_dir_tgt"))), 0), "sel_src": i32(p("sel_src")),
               "sel_tgt": i32(p("sel_tgt")), "argmax": i32(p("argmax_tgt")), "pairs": i32(p("pair_src"))}
        fl = count_flips(out[6], ref, it)
        print(f"config 3 pass {it}: HIP-vs-reference flips keys {fl['keys']} overlap {fl['overlap']} pairs {fl['pairs']} "
              f"(reference fp32-vs-fp64: {tuple(int(x) for x in env[it])})")
        if it == 0:       # identical inputs: in total no more than the reference's own twin flips (+1), and no kind of
            tot = fl["keys"] + fl["overlap"] + fl["pairs"]             # selection more than twice what the twin shows (+2)
            assert tot <= int(env[0].sum()) + 1, (fl, env[0])
            assert fl["keys"] <= 2 * env[0][0] + 2 and fl["overlap"] <= 2 * env[0][1] + 2 and fl["pairs"] <= 2 * env[0][2] + 2, fl
    R, tt = out[2].cpu().numpy(), out[3].cpu().numpy()
    dR = np.abs(R - g["c3/R_final"][i8]).reshape(B, -1).max(1)
    dt = np.abs(tt - g["c3/t_final"][i8]).reshape(B, -1).max(1)
    ok, ok_ref = int(((dR <= R_TOL) & (dt <= T_TOL)).sum()), int(g["c3/within_tol"][pj])
    print(f"config 3: {ok}/{B} final poses within 1e-4 / 1e-5 of the reference (its fp64 twin: {ok_ref}/{B}); "
          f"median|dR| {np.median(dR):.2e} (twin {float(g['c3/median_dR'][pj]):.2e}), max|dR| {dR.max():.2e} "
          f"(twin {float(g['c3/max_dR'][pj]):.2e})")
    assert ok >= ok_ref - 1
    assert np.median(dR) <= max(R_TOL, 2 * float(g["c3/median_dR"][pj]))
    # the tail: pairs thrown off by a flipped selection land anywhere within a few degrees (max|dR| is one chaotic outlier,
    # 4e-2 for the twin, 8e-2..9e-2 here depending on the build): bound their NUMBER by the twin's, and the worst by 0.25
    twin = np.abs(g["c3/R_final"][i64] - g["c3/R_final"][i8]).reshape(B, -1).max(1)
    assert int((dR > 1e-2).sum()) <= int((twin > 1e-2).sum()) + 1 and dR.max() < 0.25, (dR, twin)
    # aggregate figures of testVCRNet over the 24 pairs
    acc = evalmetrics.EvalAccumulator()
    T = torch.from_numpy
    acc.add_batch(T(src), T(tgt), T(R_gt), T(t_gt), T(eul), tuple(x.cpu() for x in out[:6]))
    m = acc.final()
    r8, r64 = float(g["c3/rot_mse"][i8]), float(g["c3/rot_mse"][i64])
    t8, t64 = float(g["c3/trans_mse"][i8]), float(g["c3/trans_mse"][i64])
    print(f"config 3 rot_MSE {m['rot_mse']:.4f} (reference {r8:.4f}, its twin {r64:.4f}); trans_MSE {m['trans_mse']:.6f} "
          f"(reference {t8:.6f}, twin {t64:.6f})")
    assert abs(m["rot_mse"] - r8) <= max(0.01 * r8, 1.5 * abs(r64 - r8))       # 1 %, or the reference's own spread
    assert abs(m["trans_mse"] - t8) <= max(0.01 * t8, 1.5 * abs(t64 - t8))
