"""End-to-end GPU parity of the drop-in VCRNet module (HIP path through the C-ABI) against
(a) golden vectors recorded from the reference and (b) the CPU oracle on fresh seeded inputs.
Tolerance = BASELINE.json north_star: 1e-4 on R, 1e-5 on t."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import oracle
from helpers import cfg_weights, golden

pytestmark = pytest.mark.gpu

R_TOL, T_TOL = 1e-4, 1e-5


def assert_mostly_close(got, ref, atol, frac=0.995, hard=2e-2):
    """Activations / soft correspondences: a single fp32 near-tie in a neighbour set (which the
    reference itself resolves differently between fp32 and fp64, or 1 vs 8 threads -- SURVEY F5)
    moves a handful of points, and the 512-d negative-distance soft-max amplifies 1e-6 embedding
    noise to ~1e-4 in src_corr.  Require `frac` of the entries within `atol` and all within `hard`;
    the binding tolerance is the one on (R, t)."""
    d = np.abs(np.asarray(got, dtype=np.float64) - np.asarray(ref, dtype=np.float64))
    assert d.max() <= hard, d.max()
    assert (d <= atol).mean() >= frac, ((d <= atol).mean(), d.max())


def make_args(**kw):
    a = dict(emb_dims=512, cycle=False, emb_nn="lpdnet", pointer="transformer", vcp_nn="topK", partial=False,
             overlap2=0.75, t3d=False, tfea=False, n_blocks=1, dropout=0.0, ff_dims=1024, n_heads=4)
    a.update(kw)
    return SimpleNamespace(**a)


def build_net(**kw):
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd.module import VCRNet
    regime = kw.pop("regime", "default")
    over = {k: kw.pop(k) for k in ("seed", "scale") if k in kw}          # weight-regime overrides (profiles/fuzz_*.py)
    net = VCRNet(make_args(**kw))
    wkw = {k: kw[k] for k in ("emb_nn", "vcp_nn", "pointer", "n_blocks") if k in kw}
    wkw.update(over)
    w = cfg_weights(regime, **wkw)
    missing = net.load_state_dict(w, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return net.cuda().eval(), w


@pytest.mark.parametrize("name", ["whole_n256_b2", "whole_n1024_b2", "whole_k40_n512_b1", "whole_k40_n4096_b2"])
def test_whole_vs_reference_golden(name):
    g = golden(name)
    net, _ = build_net()
    net.emb_nn.k = int(g["k"])
    src, tgt = torch.from_numpy(g["src"]).cuda(), torch.from_numpy(g["tgt"]).cuda()
    with torch.no_grad():
        srcK, corrK, R, t, R_ba, t_ba, emb = net._forward_fused(src, tgt, want_emb=True)
        out = net(src, tgt)
    assert out[0] is src
    cs = int(g["cstride"])
    B, N = src.shape[0], src.shape[2]
    e = emb.cpu().view(2, B, N, 512)
    assert_mostly_close(e[0].transpose(1, 2)[:, ::cs].numpy(), g["it0_femb_src"], atol=5e-4)
    assert_mostly_close(e[1].transpose(1, 2)[:, ::cs].numpy(), g["it0_femb_tgt"], atol=5e-4)
    assert_mostly_close(corrK.cpu().numpy(), g["it0_corrK"], atol=5e-4)
    np.testing.assert_allclose(R.cpu().numpy(), g["it0_R"], atol=R_TOL)
    np.testing.assert_allclose(t.cpu().numpy(), g["it0_t"], atol=T_TOL)
    np.testing.assert_allclose(R_ba.cpu().numpy(), g["it0_R_ba"], atol=R_TOL)
    # t_ba = -R^T t is derived (vcrnet_model.py:516) and inherits |dR|*|t| + |dt|; held to the plain t tolerance
    print(f"{name}: max|dt_ba|={np.abs(t_ba.cpu().numpy() - g['it0_t_ba']).max():.2e}")
    np.testing.assert_allclose(t_ba.cpu().numpy(), g["it0_t_ba"], atol=T_TOL)
    np.testing.assert_allclose(out[2].cpu().numpy(), R.cpu().numpy(), atol=0)
    print(f"{name}: max|dR|={np.abs(R.cpu().numpy() - g['it0_R']).max():.2e} max|dt|={np.abs(t.cpu().numpy() - g['it0_t']).max():.2e}")


@pytest.mark.parametrize("B,N,kind", [(4, 1024, "object"), (3, 320, "object"), (2, 2048, "uniform"), (4, 2048, "uniform"),
                                      (5, 1000, "object"), (2, 77, "object"), (3, 21, "object")])   # ragged / minimal N; (4, 2048): the ordered kNN search
def test_whole_vs_oracle(B, N, kind):
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import synth
    net, w = build_net()
    src, tgt, _, _, _ = synth.make_batch(200, B, N, kind=kind)
    src_t, tgt_t = torch.from_numpy(src), torch.from_numpy(tgt)
    ref = oracle.vcrnet_forward(w, src_t, tgt_t, oracle.OracleConfig())
    with torch.no_grad():
        out = net(src_t.cuda(), tgt_t.cuda())
    # No sample is excluded: rows whose k-th and (k+1)-th neighbour distances are EXACTLY equal (about 1 in 10^4; the
    # reference's pick there is an artefact of libstdc++'s nth_element) are replayed by the kNN kernels, so even
    # those samples match.
    # Tiny clouds (down to N = k+1 = 21, the smallest legal one): the covariance averages over few correspondences,
    # the fp32 oracle itself sits 3e-6 from its fp64 twin there and its multi-threaded rounding varies run to run on
    # the 256-core box: these two shapes are the n77 / n21 cases of tests/golden/selfdiv.npz (same items), and t gets the
    # BASELINE tolerance plus the spread the reference showed over its own four runs there (3.8e-6 / 5.3e-6).
    r_tol, t_tol = R_TOL, T_TOL
    if N <= 128:
        sd = golden("selfdiv")
        tag = f"n{N}"
        assert (int(sd[tag + "/first"]), int(sd[tag + "/B"]), int(sd[tag + "/N"])) == (200, B, N)
        r_tol, t_tol = R_TOL + float(sd[tag + "/spread_R"]), T_TOL + float(sd[tag + "/spread_t"])
    assert_mostly_close(out[1].cpu().numpy(), ref[1].numpy(), atol=5e-4)
    dR, dt = np.abs(out[2].cpu().numpy() - ref[2].numpy()).max(), np.abs(out[3].cpu().numpy() - ref[3].numpy()).max()
    print(f"whole vs oracle B={B} N={N}: max|dR| {dR:.2e} max|dt| {dt:.2e} (tolerance {r_tol:.2e} / {t_tol:.2e})")
    assert dR <= r_tol and dt <= t_tol, (dR, dt)


@pytest.mark.parametrize("B,N,k", [(4, 2048, 20), (2, 4096, 40), (5, 2500, 20)])
def test_ordered_knn_search_changes_no_bit_of_the_forward(B, N, k):
    """Clouds of 2048+ points take the ordered kNN search (Morton ranking, tiles skipped by their balls; forward.hip); a kNN
    tuning value (knn_waves = 8: the same 16-query-wave kernels, asked for explicitly) keeps the plain scan.  The neighbour SETS
    are equal and everything downstream is a max over a point's neighbours, so every output is bit-identical."""
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import synth
    net, _ = build_net()
    net.emb_nn.k = k
    src, tgt, _, _, _ = synth.make_batch(900, B, N, kind="uniform")
    s, t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    with torch.no_grad():
        tr = vcrnet_amd.native.LaunchTrace()
        ordered = net._forward_fused(s, t, want_emb=True, trace=tr.trace)
        torch.cuda.synchronize()
        assert any(n == "knn:rank" for n, _ in tr.launches()), "the ordered search did not run"
        net.knn_waves = 8
        plain = net._forward_fused(s, t, want_emb=True)
    for a, b in zip(ordered, plain):
        if torch.is_tensor(a):
            assert torch.equal(a, b)


def test_config5_shape_vs_oracle():
    """BASELINE configs[4] end to end: uniform clouds, N = 4096, k = 40 (LPDNet.k override, lpdnet_model.py:81), one
    fused call, against the CPU oracle on fresh inputs (the recorded reference run of the same shape is
    whole_k40_n4096_b2 above) at the BASELINE tolerance."""
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import synth
    net, w = build_net()
    net.emb_nn.k = 40
    src, tgt, _, _, _ = synth.make_batch(5000, 2, 4096, kind="uniform")
    src_t, tgt_t = torch.from_numpy(src), torch.from_numpy(tgt)
    ref = oracle.vcrnet_forward(w, src_t, tgt_t, oracle.OracleConfig(k=40))
    with torch.no_grad():
        out = net(src_t.cuda(), tgt_t.cuda())
    dR, dt = np.abs(out[2].cpu().numpy() - ref[2].numpy()).max(), np.abs(out[3].cpu().numpy() - ref[3].numpy()).max()
    print(f"config 5 shape (N=4096, k=40): max|dR|={dR:.2e} max|dt|={dt:.2e}")
    assert_mostly_close(out[1].cpu().numpy(), ref[1].numpy(), atol=5e-4)
    assert dR <= R_TOL and dt <= T_TOL


@pytest.mark.parametrize("k,N", [(50, 512), (62, 300), (33, 200)])
def test_other_neighbourhood_sizes_vs_oracle(k, N):
    """LPDNet.k is an attribute the reference's callers may set to anything (lpdnet_model.py:81, util.py:143-160 has no
    limit): neighbourhoods other than the path's 20 / 40 -- up to the library's 62 -- through the generic EdgeConv / gather
    kernels and the 64-entry kNN lists, against the oracle at the BASELINE tolerance; 63 is refused loudly."""
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import native, synth
    net, w = build_net()
    net.emb_nn.k = k
    src, tgt, _, _, _ = synth.make_batch(5100 + k, 2, N)
    src_t, tgt_t = torch.from_numpy(src), torch.from_numpy(tgt)
    ref = oracle.vcrnet_forward(w, src_t, tgt_t, oracle.OracleConfig(k=k))
    with torch.no_grad():
        out = net(src_t.cuda(), tgt_t.cuda())
    dR, dt = np.abs(out[2].cpu().numpy() - ref[2].numpy()).max(), np.abs(out[3].cpu().numpy() - ref[3].numpy()).max()
    print(f"k = {k}, N = {N}: max|dR| {dR:.2e} max|dt| {dt:.2e}")
    assert dR <= R_TOL and dt <= T_TOL
    net.emb_nn.k = 63
    with torch.no_grad(), pytest.raises(native.VcrHipError, match="unsupported"):
        net(src_t.cuda(), tgt_t.cuda())


@pytest.mark.parametrize("name,kw", [("dist_n256_b2", dict(vcp_nn="dist")), ("identity_n256_b2", dict(pointer="identity"))])
def test_fused_variants_vs_golden(name, kw):
    g = golden(name)
    net, _ = build_net(**kw)
    with torch.no_grad():
        out = net(torch.from_numpy(g["src"]).cuda(), torch.from_numpy(g["tgt"]).cuda())
    assert_mostly_close(out[1].cpu().numpy(), g["it0_corrK"], atol=5e-4)
    np.testing.assert_allclose(out[2].cpu().numpy(), g["it0_R"], atol=R_TOL)
    np.testing.assert_allclose(out[3].cpu().numpy(), g["it0_t"], atol=T_TOL)


def test_permutation_invariance_and_inverse_pose():
    """SURVEY section 4 properties: (R,t) invariant to the point order of tgt; R_ba = R^T, t_ba = -R^T t."""
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import synth
    net, _ = build_net()
    src, tgt, _, _, _ = synth.make_batch(300, 2, 512)
    s, t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    perm = torch.randperm(512, generator=torch.Generator().manual_seed(0)).cuda()
    with torch.no_grad():
        a = net(s, t)
        b = net(s, t[:, :, perm])
    np.testing.assert_allclose(a[2].cpu().numpy(), b[2].cpu().numpy(), atol=R_TOL)
    np.testing.assert_allclose(a[3].cpu().numpy(), b[3].cpu().numpy(), atol=T_TOL)
    R, tt = a[2].cpu(), a[3].cpu()
    np.testing.assert_allclose(a[4].cpu().numpy(), R.transpose(1, 2).numpy(), atol=1e-6)
    np.testing.assert_allclose(a[5].cpu().numpy(), -torch.matmul(R.transpose(1, 2), tt.unsqueeze(2)).squeeze(2).numpy(), atol=1e-6)
    assert torch.allclose(torch.det(R), torch.ones(2), atol=1e-5)


def test_cpu_input_fails_loudly():
    net, _ = build_net()
    with torch.no_grad(), pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 64), torch.zeros(1, 3, 64))


def check_iter_against_oracle(net, w, cfg, src, tgt, iters, out):
    """vcrnetIter(iter >= 2) against the CPU oracle WITHOUT comparing two free-running refinement loops: a pass after the
    first starts from a source moved by a pose that differs at 1e-7, where a kNN near-tie can resolve differently (the
    reference's own float64 twin moves one of four poses by 5.6e-3 that way, tests/golden/selfdiv.npz it2_eval), and the
    oracle's fp32 rounding itself changes with the host's thread count -- such a comparison is a coin toss per box.
    Instead (a) every pass is teacher-forced: the HIP forward on the ORACLE's moved source of that pass against the
    oracle's pass, at the BASELINE tolerance; (b) the device-side loop equals the composition of HIP forwards on its own
    moved sources (the plumbing of vcr_vcrnet_iter_f32: pose_step, composition, inverse)."""
    per = []
    oracle.vcrnet_iter(w, src, tgt, cfg, iters=iters, per_iter=per)
    t_d = tgt.cuda()
    with torch.no_grad():
        for it, (R_ref, t_ref, cur_ref, _) in enumerate(per):                 # (a)
            o = net(cur_ref.cuda(), t_d)
            np.testing.assert_allclose(o[2].cpu().numpy(), R_ref.numpy(), atol=R_TOL, err_msg=f"pass {it}")
            np.testing.assert_allclose(o[3].cpu().numpy(), t_ref.numpy(), atol=T_TOL, err_msg=f"pass {it}")
        cur, R_f, t_f = src.cuda(), None, None                                # (b)
        for _ in range(iters):
            o = net(cur, t_d)
            R, t = o[2], o[3]
            cur = torch.matmul(R, cur) + t.unsqueeze(2)
            R_f, t_f = (R, t) if R_f is None else (torch.matmul(R, R_f), torch.matmul(R, t_f.unsqueeze(2)).squeeze(2) + t)
    np.testing.assert_allclose(out[2].cpu().numpy(), R_f.cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(out[3].cpu().numpy(), t_f.cpu().numpy(), atol=2e-6)


def test_iter_wrapper_whole():
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import synth
    from vcrnet_amd.module import vcrnetIter
    net, w = build_net()
    src, tgt, _, _, _ = synth.make_batch(400, 2, 256)
    s, t = torch.from_numpy(src), torch.from_numpy(tgt)
    with torch.no_grad():
        out = vcrnetIter(net, s.cuda(), t.cuda(), iter=2)
    check_iter_against_oracle(net, w, oracle.OracleConfig(), s, t, 2, out)


@pytest.mark.parametrize("iters", [1, 2])
@pytest.mark.parametrize("vcp", ["topK", "att"])
def test_iter_wrapper_with_cycle_returns_the_inverse_pose(vcp, iters):
    """vcrnetIter recomputes (R_ba, t_ba) as the inverse of the composed pose (vcrnet_model.py:40-41) even when
    args.cycle makes VCRNet.forward return the second head's solve -- also for iter = 1, where the device loop is a
    single forward."""
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import synth
    from vcrnet_amd.module import vcrnetIter
    net, w = build_net(cycle=True, vcp_nn=vcp)
    src, tgt, _, _, _ = synth.make_batch(430, 2, 256)
    s, t = torch.from_numpy(src), torch.from_numpy(tgt)
    ref = oracle.vcrnet_iter(w, s, t, oracle.OracleConfig(cycle=True, vcp_nn=vcp), iters=iters)
    with torch.no_grad():
        out = vcrnetIter(net, s.cuda(), t.cuda(), iter=iters)
        fwd = net(s.cuda(), t.cuda())
    R, tt = out[2].cpu(), out[3].cpu()
    np.testing.assert_allclose(out[4].cpu().numpy(), R.transpose(1, 2).numpy(), atol=1e-6)
    np.testing.assert_allclose(out[5].cpu().numpy(), -torch.matmul(R.transpose(1, 2), tt.unsqueeze(2)).squeeze(2).numpy(),
                               atol=1e-6)
    if iters == 1:
        np.testing.assert_allclose(R.numpy(), ref[2].numpy(), atol=R_TOL)
        np.testing.assert_allclose(tt.numpy(), ref[3].numpy(), atol=T_TOL)
        np.testing.assert_allclose(out[4].cpu().numpy(), ref[4].numpy(), atol=R_TOL)
    else:             # two free-running loops are not comparable pass by pass (see check_iter_against_oracle)
        check_iter_against_oracle(net, w, oracle.OracleConfig(cycle=True, vcp_nn=vcp), s, t, iters, out)
    # the plain forward DOES return the cycle head's pose, which is not the inverse
    assert np.abs(fwd[4].cpu().numpy() - fwd[2].cpu().transpose(1, 2).numpy()).max() > 1e-4


def test_data_parallel_wrapped_net_takes_the_device_loop():
    """The reference's caller always wraps the net in nn.DataParallel (util/initPara.py:260 -> vcrnet_model.py:26).
    On one device the wrapper is a pass-through, and vcrnetIter still runs the single-call device loop."""
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import synth
    from vcrnet_amd.module import VCRNet, vcrnetIter
    net, _ = build_net()
    dp = torch.nn.DataParallel(net, device_ids=[0])
    src, tgt, _, _, _ = synth.make_batch(440, 3, 256)
    s, t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    calls = []
    orig = VCRNet.forward_iter
    VCRNet.forward_iter = lambda self, *a: (calls.append(1), orig(self, *a))[1]
    try:
        with torch.no_grad():
            a = vcrnetIter(net, s, t, iter=2)
            b = vcrnetIter(dp, s, t, iter=2)
            c = dp(s, t)
            d = net(s, t)
    finally:
        VCRNet.forward_iter = orig
    assert len(calls) == 2                                  # both took forward_iter (one vcr_vcrnet_iter_f32 call each)
    for x, y in zip(a[1:], b[1:]):
        assert torch.equal(x, y)
    for x, y in zip(c[1:], d[1:]):
        assert torch.equal(x, y)
    # a checkpoint saved through the wrapper carries the "module." prefix and loads back (SURVEY section 5 gotcha)
    net2, _ = build_net()
    net2.load_state_dict(dp.state_dict(), strict=True)


def test_module_on_its_own_device_without_set_device():
    """A module and inputs on cuda:0 run on cuda:0's stream whatever the current device is (the guard that matters on
    multi-GPU boxes: net.to('cuda:1')(x) without torch.cuda.set_device(1)); a device mismatch raises."""
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import native, synth
    net, _ = build_net()
    src, tgt, _, _, _ = synth.make_batch(450, 2, 128)
    s, t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    with torch.no_grad():
        out = net(s, t)
        with pytest.raises(native.VcrHipError):
            net(s, t.cpu())
    assert out[2].device == s.device
    assert native.stream_ptr(s.device) == torch.cuda.current_stream(s.device).cuda_stream


@pytest.mark.parametrize("mode", ["bf16x3", "bf16x3+sdpa"])
@pytest.mark.parametrize("name", ["whole_n256_b2", "whole_n1024_b2", "whole_k40_n512_b1", "whole_k40_n4096_b2"])
def test_bf16x3_linear_mode_vs_reference_golden(name, mode):
    """Opt-in linear_mode='bf16x3' (exact 3-way bf16 splits on the bf16 matrix pipe for the linears and EdgeConv's
    convDG2; '+sdpa': the attention products too) keeps the BASELINE tolerances, at both k of the path and up to
    BASELINE configs[4]'s N = 4096."""
    g = golden(name)
    net, _ = build_net()
    net.emb_nn.k = int(g["k"])
    net.linear_mode = mode
    src, tgt = torch.from_numpy(g["src"]).cuda(), torch.from_numpy(g["tgt"]).cuda()
    with torch.no_grad():
        out = net(src, tgt)
    dR = np.abs(out[2].cpu().numpy() - g["it0_R"]).max()
    dt = np.abs(out[3].cpu().numpy() - g["it0_t"]).max()
    print(f"{name} {mode}: max|dR|={dR:.2e} max|dt|={dt:.2e}")
    assert dR <= R_TOL and dt <= T_TOL


def test_c_abi_error_codes():
    """Errors come back as codes (vcr_hip.h): negative for bad arguments / workspace / unsupported shapes, never an
    exception or a crash across the boundary; vcr_strerror names them."""
    import ctypes as C
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import native, synth
    L = native.lib()
    net, _ = build_net()
    net._pack()
    src, tgt, _, _, _ = synth.make_batch(0, 2, 128)
    s, t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    B, N = 2, 128
    f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device="cuda")
    corr4, src4, R, tt, Rb, tb = f(B, N, 4), f(B, N, 4), f(B, 3, 3), f(B, 3), f(B, 3, 3), f(B, 3)
    io = native.VcrnetIo(native.ptr(s), native.ptr(t), B, N, native.ptr(corr4), native.ptr(src4), native.ptr(R),
                         native.ptr(tt), native.ptr(Rb), native.ptr(tb), None)
    need = L.vcr_vcrnet_workspace_bytes(C.byref(net._cw), B, N)
    ws = torch.empty(need + 256, dtype=torch.uint8, device="cuda")
    base = ws.data_ptr() + (-ws.data_ptr()) % 256
    stream = C.c_void_p(native.stream_ptr())
    call = lambda io_, p, n: L.vcr_vcrnet_forward_f32(C.byref(net._cw), C.byref(io_), C.c_void_p(p), n, stream)
    assert call(io, base, need) == 0
    assert call(io, base, need // 2) < 0 and b"workspace" in L.vcr_strerror(call(io, base, need // 2))
    assert call(io, base + 4, need) < 0                                   # misaligned workspace
    bad = native.VcrnetIo(native.ptr(s), None, B, N, native.ptr(corr4), native.ptr(src4), native.ptr(R),
                          native.ptr(tt), native.ptr(Rb), native.ptr(tb), None)
    assert call(bad, base, need) < 0                                      # NULL input
    tiny = native.VcrnetIo(native.ptr(s), native.ptr(t), B, 8, native.ptr(corr4), native.ptr(src4), native.ptr(R),
                           native.ptr(tt), native.ptr(Rb), native.ptr(tb), None)
    assert call(tiny, base, need) < 0                                     # k + 1 > N
    # kernel entry points: bad pitch, unsupported width
    x = f(64, 48)
    with pytest.raises(native.VcrHipError):
        native.linear(x, f(32, 48), None)                                 # K % 32 != 0
    with pytest.raises(native.VcrHipError):
        native.layernorm(f(16, 256), f(256), f(256))                      # LayerNorm width != 512
    torch.cuda.synchronize()


@pytest.mark.parametrize("kw,mode", [({}, "fp32"), ({}, "bf16x3+sdpa"), (dict(partial=True), "fp32"),
                                     (dict(emb_nn="dgcnn"), "fp32"), (dict(vcp_nn="dist", cycle=True), "fp32"),
                                     (dict(emb_nn="pointnet"), "fp32"), (dict(vcp_nn="att"), "bf16x3"),
                                     (dict(pointer="identity"), "fp32"), (dict(partial=True, _iters=3), "fp32")])
def test_workspace_contents_do_not_matter(kw, mode):
    """The workspace is laid out by buffer liveness (forward.hip: Plan): a buffer's bytes are whatever an earlier, dead buffer of
    the same call -- or the previous call -- left there.  Nothing may be read before it is written: the pooled workspace is
    filled with zeros / NaN bit patterns / a ramp between calls and every output must come out bit-identical."""
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import synth
    kw = dict(kw)
    iters = kw.pop("_iters", 1)                            # > 1: the device-side vcrnetIter loop (vcr_vcrnet_iter_f32)
    net, _ = build_net(**kw)
    net.linear_mode = mode
    partial = bool(kw.get("partial"))
    src, tgt, _, _, _ = synth.make_batch(77, 3, 384, partial=partial)
    s, t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()

    def run(fill):
        pooled = [b["ws"] for idle in net._shared.pool.values() for b in idle]
        assert pooled, "the first call leaves its workspace in the pool"
        for ws in pooled:
            if fill == "ramp":
                ws.copy_((torch.arange(ws.numel(), device=ws.device) * 37 % 251).to(torch.uint8))
            else:
                ws.fill_(fill)
        with torch.no_grad():
            out = net._forward_fused(s, t, want_emb=iters == 1, iters=iters)
        torch.cuda.synchronize()
        return [o.clone() for o in out if torch.is_tensor(o)]

    with torch.no_grad():
        net._forward_fused(s, t, want_emb=iters == 1, iters=iters)
    torch.cuda.synchronize()
    ref = run(0)
    for fill in (0xFF, 0x7F, "ramp"):                      # 0xFFFFFFFF / 0x7F7F7F7F: NaN and 3.4e38 as floats, huge indices as ints
        got = run(fill)
        assert len(got) == len(ref)
        for i, (a, b) in enumerate(zip(got, ref)):
            assert torch.equal(a, b), (fill, i, (a.float() - b.float()).abs().max().item())


@pytest.mark.parametrize("kw,mode,merge", [({}, "fp32", True), ({}, "fp32", False), ({}, "bf16x3", True), ({}, "bf16x3+sdpa", True),
                                           ({}, "bf16x3+sdpa", False), (dict(partial=True), "fp32", True),
                                           (dict(partial=True), "bf16x3+sdpa", True), (dict(emb_nn="dgcnn"), "fp32", True),
                                           (dict(emb_nn="pointnet"), "fp32", True), (dict(vcp_nn="dist", cycle=True), "fp32", True),
                                           (dict(vcp_nn="att", cycle=True), "bf16x3", True), (dict(pointer="identity"), "fp32", True),
                                           (dict(partial=True, _iters=3), "fp32", True), (dict(_iters=2, _n=2048, _b=4), "fp32", True)])
def test_planned_and_flat_workspace_layouts_agree(kw, mode, merge):
    """forward.hip overlays workspace buffers by hand-registered (first, last) launch numbers.  A lifetime one launch too short
    lets a launch overwrite rows a later one still reads -- and test_workspace_contents_do_not_matter cannot see that (both of
    its runs share the layout).  vcr_vcrnet_weights.workspace_flat gives every buffer its own memory: every output of every
    path must be bit-identical between the two layouts (the 2048-point case takes the ordered kNN search, whose arrays live
    in the embedding buffer until conv3)."""
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import synth
    kw = dict(kw)
    iters, n, b = kw.pop("_iters", 1), kw.pop("_n", 384), kw.pop("_b", 2)     # (4 x 2048: enough query groups for the ordered search)
    partial = bool(kw.get("partial"))
    src, tgt, _, _, _ = synth.make_batch(78, b, n, partial=partial, kind="object" if n < 2048 else "uniform")
    s, t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    outs, sizes = [], []
    for flat in (False, True):
        net, _ = build_net(**kw)
        net.linear_mode, net.merge_encdec, net.workspace_flat = mode, merge, flat
        with torch.no_grad():
            out = net._forward_fused(s, t, want_emb=iters == 1, iters=iters, want_selections=partial)
        torch.cuda.synchronize()
        sel = out[-1] if partial else {}
        outs.append([o.clone() for o in out if torch.is_tensor(o)] + [sel[k_].clone() for k_ in sorted(sel)])
        sizes.append(sum(b["ws"].numel() for idle in net._shared.pool.values() for b in idle))
    assert sizes[1] > 1.3 * sizes[0], sizes                 # the flat layout really is another layout
    assert len(outs[0]) == len(outs[1])
    for i, (a, b) in enumerate(zip(*outs)):
        assert torch.equal(a, b), (i, (a.float() - b.float()).abs().max().item())


def test_runs_on_the_callers_stream():
    """Every launch goes to the stream the caller is on (torch.cuda.current_stream()): the same results on a side
    stream, and nothing leaks onto the default stream (the output is only valid after the SIDE stream is waited on)."""
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import synth
    net, _ = build_net()
    src, tgt, _, _, _ = synth.make_batch(321, 2, 256)
    s, t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    with torch.no_grad():
        ref = net(s, t)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            out = net(s, t)
        side.synchronize()
    for a, b in zip(out[1:], ref[1:]):
        assert torch.equal(a, b)


def test_forward_captures_into_a_hip_graph():
    """vcr_vcrnet_forward_f32 neither allocates nor synchronises, so a whole forward records into ONE HIP graph
    (torch.cuda.CUDAGraph) and its replays are bit-identical to the eager call -- repeatedly, on the capture stream and on
    the default stream, with unrelated work in between.  (The tie counters are zeroed by a kernel, not hipMemsetAsync: with
    a memset node in the graph a replay on the default stream after unrelated default-stream work hung or faulted on this
    ROCm / torch build.)  Measured on MI355X: replay and eager run the same 5.17 ms per step at configs[1] -- the path is
    GPU-bound."""
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import synth
    net, _ = build_net()
    src, tgt, _, _, _ = synth.make_batch(77, 2, 256)
    s, t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    with torch.no_grad():
        ref = net(s, t)
        torch.cuda.synchronize()
        g, st = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        with torch.cuda.stream(st):
            net(s, t)                                        # warm the workspace / packed weights on this stream
            st.synchronize()
            with torch.cuda.graph(g, stream=st):
                out = net(s, t)
        torch.cuda.synchronize()
        for rep in range(6):
            ctx = torch.cuda.stream(st) if rep % 2 else torch.cuda.stream(torch.cuda.current_stream())
            with ctx:
                for x in out[1:]:
                    x.zero_()
                g.replay()
                g.replay()
                torch.cuda.synchronize()
                for a, b in zip(out[1:], ref[1:]):
                    assert torch.equal(a, b), rep
                _ = torch.randn(1 << 16, device="cuda").sum().item()       # unrelated allocation + kernels + sync


@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "bf16x3+sdpa"])
@pytest.mark.parametrize("partial", [False, True])
def test_merged_encoder_decoder_launches_change_nothing(partial, mode):
    """enc.qkv + dec.qkv as ONE stacked GEMM and the encoder's / decoder's self-attention as ONE grouped launch
    (vcr_vcrnet_weights.fold_encdec_qkv / split.encdec_qkv, vcr_sdpa_args.ngroups) against the four separate launches:
    the same tiles with the same arithmetic, so every output is bit-identical -- in fp32 and (round 5) in both
    exact-split modes, whose attention kernel takes groups since."""
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import synth
    kw = dict(partial=True, overlap2=synth.OVERLAP2_0575) if partial else {}
    net, _ = build_net(**kw)
    net.linear_mode = mode
    src, tgt, _, _, _ = synth.make_batch(7100, 3, 320, partial=partial)
    s, t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    outs = []
    for merge in (True, False):
        net.merge_encdec = merge
        with torch.no_grad():
            outs.append(net._forward_fused(s, t, want_emb=True))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("nb,N,grouped", [(2, 1024, False), (2, 1024, True), (3, 300, False), (16, 1024, False)])
def test_sdpa_output_with_the_keys_split(nb, N, grouped):
    """vcr_sdpa_args.split_work on attention-OUTPUT launches: grids of less than one round of workgroups (small batches;
    the grouped encoder + decoder self-attention included, ragged key counts too) deal the keys to several workgroups per
    query block, each writing an unnormalised partial output and its (max, sum), merged by one more kernel -- the plain
    launch's output to rounding.  A grid of a full round or more (last shape) is launched as without the scratch."""
    import math
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import native
    g = torch.Generator().manual_seed(nb + N)
    h, sc = 4, 1 / math.sqrt(128)
    if grouped:
        qkv2 = (torch.randn(nb * N, 6 * 512, generator=g) * 0.7).cuda()
        args = (qkv2[:, :512], qkv2[:, 512:1024], qkv2[:, 1024:1536], nb, h, N, N, sc)
        kw = dict(kv_batch_shift=1, groups=(2, 1536, 1536, 1536))
    else:
        q, k, v = ((torch.randn(nb * N, 512, generator=g) * 0.7).cuda() for _ in range(3))
        args, kw = (q, k, v, nb, h, N, N, sc), dict(kv_batch_shift=nb // 2)
    plain = native.sdpa(*args, **kw)
    split = native.sdpa(*args, split=True, **kw)
    torch.testing.assert_close(split, plain, atol=3e-6, rtol=1e-5)
    # whether the launcher splits follows from the grid and THIS device's CU count (attention.hip: query blocks of 128,
    # two resident workgroups per CU, at least two splits of >= 4 key tiles each must fit one round)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    blocks = (N + 127) // 128 * h * nb * (2 if grouped else 1)
    splits = blocks * 2 <= 2 * cus and ((N + 31) // 32) // 2 >= 4
    assert torch.equal(split, plain) == (not splits), (blocks, cus)
    assert splits == (nb != 16) or cus != 256


def test_grouped_sdpa_equals_separate_launches():
    import math
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import native
    g = torch.Generator().manual_seed(3)
    nb, h, N = 4, 4, 200
    qkv2 = torch.randn(nb * N, 6 * 512, generator=g).cuda()
    grouped = native.sdpa(qkv2[:, :512], qkv2[:, 512:1024], qkv2[:, 1024:1536], nb, h, N, N, 1 / math.sqrt(128),
                          kv_batch_shift=1, groups=(2, 1536, 1536, 1536))
    for gi in range(2):
        o = 1536 * gi
        one = native.sdpa(qkv2[:, o:o + 512], qkv2[:, o + 512:o + 1024], qkv2[:, o + 1024:o + 1536], nb, h, N, N,
                          1 / math.sqrt(128), kv_batch_shift=1)
        assert torch.equal(grouped[gi], one)
