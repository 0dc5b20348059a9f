"""GPU parity of every C-ABI kernel against the CPU oracle on the same seeded inputs (-m gpu).
Integer outputs (neighbour sets) must match exactly up to fp32 near-ties adjudicated in fp64;
floating-point outputs within the tolerance written next to each assert."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from helpers import cfg_weights, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import native
    native.lib()
    return native


@pytest.fixture(scope="module")
def W():
    return cfg_weights()


def dev(t):
    return t.cuda().contiguous()


def knn_sets_ok(x_cf, idx_gpu, k, tol_rel=2e-6):
    """x_cf [B,C,N] cpu; idx_gpu [B,N,k].  Exact set match vs the oracle, except rows where fp32
    near-ties at the k-th place make the choice ambiguous (checked against fp64 distances)."""
    ref = oracle.knn_indices(x_cf, k).numpy()
    got = idx_gpu.cpu().numpy().astype(np.int64)
    a, b = np.sort(got, -1), np.sort(ref, -1)
    bad = np.argwhere((a != b).any(-1))
    d64 = oracle.neg_sqdist_knn(x_cf.double()).numpy()
    scale = np.abs(d64).max()
    for bi, i in bad:
        row = d64[bi, i]
        order = np.sort(row)[::-1]
        kth = order[k]                       # value of the last kept rank (rank 0 dropped -> ranks 1..k)
        top = order[0]
        for j in set(got[bi, i]) ^ set(ref[bi, i]):
            # every disputed index must sit within rounding of the k-th value, or of rank 0 (the dropped one)
            assert abs(row[j] - kth) <= tol_rel * scale or abs(row[j] - top) <= tol_rel * scale, (bi, i, j)
    assert all(len(set(r)) == k for r in got.reshape(-1, k)), "duplicate neighbour"
    return len(bad)


def test_pointwise(nat, W):
    g = torch.Generator().manual_seed(0)
    x = torch.rand(3, 3, 200, generator=g) * 2 - 1
    xyz4, f64, sq = nat.pointwise(dev(x), dev(W["emb_nn.conv1_lpd.weight"].view(64, 3)), dev(W["emb_nn.conv1_lpd.bias"]),
                                  dev(W["emb_nn.conv2_lpd.weight"].view(64, 64)), dev(W["emb_nn.conv2_lpd.bias"]))
    h = F.relu(F.conv1d(x, W["emb_nn.conv1_lpd.weight"], W["emb_nn.conv1_lpd.bias"]))
    h = F.relu(F.conv1d(h, W["emb_nn.conv2_lpd.weight"], W["emb_nn.conv2_lpd.bias"]))
    torch.testing.assert_close(f64.cpu(), h.transpose(1, 2), atol=2e-6, rtol=1e-5)
    torch.testing.assert_close(sq.cpu(), (h ** 2).sum(1), atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(xyz4.cpu()[..., :3], x.transpose(1, 2), atol=0, rtol=0)
    torch.testing.assert_close(xyz4.cpu()[..., 3], (x ** 2).sum(1), atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("N", [1024, 1000, 333, 77, 21])
def test_pointwise_matches_cpu_rounding_bit_for_bit(nat, W, N):
    """conv1/conv2 (bias-first fma chains) and |feat|^2 in ATen's association -- cascade sum for the points its
    32-wide vector loop covers, four interleaved accumulators for the last N % 32 -- are bit-identical to the CPU
    ops the reference runs (multi-threaded, batch >= 2), so the feature-space distance matrix cannot differ."""
    g = torch.Generator().manual_seed(N)
    x = torch.rand(4, 3, N, generator=g) * 2 - 1
    _, f64, sq = nat.pointwise(dev(x), dev(W["emb_nn.conv1_lpd.weight"].view(64, 3)), dev(W["emb_nn.conv1_lpd.bias"]),
                               dev(W["emb_nn.conv2_lpd.weight"].view(64, 64)), dev(W["emb_nn.conv2_lpd.bias"]))
    h = F.relu(F.conv1d(x, W["emb_nn.conv1_lpd.weight"], W["emb_nn.conv1_lpd.bias"]))
    h = F.relu(F.conv1d(h, W["emb_nn.conv2_lpd.weight"], W["emb_nn.conv2_lpd.bias"]))
    assert torch.equal(f64.cpu(), h.transpose(1, 2))
    assert torch.equal(sq.cpu(), (h ** 2).sum(1))
    # the same launch with the first EdgeConv's P | Q projection fused in: conv2 and the projection on the matrix pipe (an
    # MFMA is a k-ascending fma chain that starts at its accumulator = the bias) -- the three stem outputs do not move by a
    # bit, and P | Q is what vcr_linear_f32 makes of feat64 to fp32 rounding
    w1 = W["emb_nn.convDG1.0.weight"].view(128, 128)
    wpq = dev(torch.cat((w1[:, :64], w1[:, 64:]), 0).contiguous())
    bpq = dev(torch.cat((torch.zeros(128), W["emb_nn.convDG1.0.bias"])))
    xyz4b, f64b, sqb, pq = nat.pointwise(dev(x), dev(W["emb_nn.conv1_lpd.weight"].view(64, 3)), dev(W["emb_nn.conv1_lpd.bias"]),
                                         dev(W["emb_nn.conv2_lpd.weight"].view(64, 64)), dev(W["emb_nn.conv2_lpd.bias"]), wpq, bpq)
    assert torch.equal(f64b, f64) and torch.equal(sqb, sq) and torch.equal(xyz4b, _)
    ref_pq = (f64.cpu().double().view(-1, 64) @ wpq.cpu().double().t() + bpq.cpu().double())
    assert (pq.cpu().double() - ref_pq).abs().max().item() <= 4e-6 * 8 + 1e-6
    torch.testing.assert_close(pq, nat.linear(f64.view(-1, 64), wpq, bpq), atol=2e-6, rtol=1e-6)
    # the optional transposed copy (every group of 16 channels as its 4 x 4 transpose: the operand layout of the 16-query
    # kNN waves) from both launches, and the kNN reading it: same bits, same neighbours
    ft_a, ft_b = torch.full_like(f64, float("nan")), torch.full_like(f64, float("nan"))
    nat.pointwise(dev(x), dev(W["emb_nn.conv1_lpd.weight"].view(64, 3)), dev(W["emb_nn.conv1_lpd.bias"]),
                  dev(W["emb_nn.conv2_lpd.weight"].view(64, 64)), dev(W["emb_nn.conv2_lpd.bias"]), feat_t=ft_a)
    nat.pointwise(dev(x), dev(W["emb_nn.conv1_lpd.weight"].view(64, 3)), dev(W["emb_nn.conv1_lpd.bias"]),
                  dev(W["emb_nn.conv2_lpd.weight"].view(64, 64)), dev(W["emb_nn.conv2_lpd.bias"]), wpq, bpq, feat_t=ft_b)
    want_t = f64.view(4, N, 4, 4, 4).transpose(3, 4).reshape(4, N, 64)
    assert torch.equal(ft_a, want_t) and torch.equal(ft_b, want_t)
    if N >= 64:
        k = 20
        for waves in (0, 8):
            plain = nat.knn(f64, sq, k, waves=waves)
            assert torch.equal(nat.knn(f64, sq, k, waves=waves, xt=ft_a), plain)
        xyz4c = _.contiguous()
        i64, i3 = nat.knn_pair(f64, sq, xyz4c, k)
        j64, j3 = nat.knn_pair(f64, sq, xyz4c, k, xt=ft_a)
        assert torch.equal(i64, j64) and torch.equal(i3, j3)


@pytest.mark.parametrize("N,k", [(256, 20), (1024, 20), (192, 20), (100, 20), (512, 40), (64, 5), (512, 50), (300, 62), (64, 62)])
def test_knn_feature_space(nat, W, N, k):
    g = golden("whole_n1024_b2")
    rs = np.random.RandomState(N + k)
    x3 = torch.from_numpy(g["src"][:, :, rs.permutation(1024)[:N]])
    h = F.relu(F.conv1d(x3, W["emb_nn.conv1_lpd.weight"], W["emb_nn.conv1_lpd.bias"]))
    h = F.relu(F.conv1d(h, W["emb_nn.conv2_lpd.weight"], W["emb_nn.conv2_lpd.bias"]))
    feat = dev(h.transpose(1, 2))
    sq = dev((h ** 2).sum(1))
    idx = nat.knn(feat, sq, k)
    nbad = knn_sets_ok(h, idx, k)
    assert nbad <= max(2, h.shape[0] * N // 100)


@pytest.mark.parametrize("N,k", [(256, 20), (1024, 20), (768, 20), (100, 20), (512, 40), (2048, 20), (4096, 40), (21, 20),
                                 (45, 40), (1000, 5), (1024, 50), (4096, 62), (63, 62)])
def test_knn_cartesian(nat, N, k):
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import synth
    src = torch.from_numpy(synth.make_batch(3, 2, N, kind="object" if N <= 2048 else "uniform")[0])
    xyz4 = torch.cat((src.transpose(1, 2), (src ** 2).sum(1).unsqueeze(-1)), -1)
    idx = nat.knn(dev(xyz4), None, k)
    nbad = knn_sets_ok(src, idx, k)
    assert nbad <= max(2, 2 * N // 100)
    # the candidate split (1 / 2 / 4 waves per 16 queries, chosen from the grid size) must not change any set
    ref = np.sort(idx.cpu().numpy(), -1)
    for waves in (1, 2, 4):
        assert (np.sort(nat.knn(dev(xyz4), None, k, waves=waves).cpu().numpy(), -1) == ref).all(), waves


def test_knn_duplicates_drop_rank0(nat):
    """Duplicate points: 'drop rank 0' keeps the OTHER copy as a neighbour (util.py:159 semantics)."""
    rs = np.random.RandomState(5)
    p = rs.uniform(-1, 1, (1, 3, 64)).astype(np.float32)
    p[0, :, 1] = p[0, :, 0]
    src = torch.from_numpy(p)
    xyz4 = torch.cat((src.transpose(1, 2), (src ** 2).sum(1).unsqueeze(-1)), -1)
    idx = nat.knn(dev(xyz4), None, 4).cpu().numpy()
    assert (1 in idx[0, 0]) != (0 in idx[0, 0])   # exactly one of the twin copies survives for point 0
    assert (1 in idx[0, 1]) != (0 in idx[0, 1])
    assert (np.sort(idx, -1) == np.sort(oracle.knn_indices(src, 4).numpy(), -1)).all()     # ... the one Tensor.topk keeps


@pytest.mark.parametrize("N,k", [(64, 4), (300, 20), (747, 20), (1024, 20), (1343, 20), (1344, 20), (2048, 20), (700, 40), (2700, 40),
                                 (333, 62), (4100, 62)])     # (k + 1) * 64 <= N switches Tensor.topk's algorithm
def test_knn_shared_best_value_follows_torch_topk(nat, N, k):
    """Copies of a point (pairs, triples, one point more often than there are neighbours; in real clouds also a neighbour whose
    distance ROUNDS to the point's own): the best value of such a row is shared, and which of the tied entries Tensor.topk returns first -- the
    one util.py:159 drops -- comes out of ATen's sort of the selected entries (std::sort after nth_element, or partial_sort's
    heap sort), not out of the point index.  The kernels list such rows like boundary ties and the replay sorts the way ATen
    does: every row's neighbour set equals the reference's, in every launch form (candidate splits, pair launch, 16-query waves)."""
    rs = np.random.RandomState(N * 7 + k)
    B = 3
    def with_copies(x):                                    # x [B, C, N]
        for b in range(B):
            perm, pos = rs.permutation(N), 0
            for copies, groups in ((2, N // 8), (3, N // 16), (k + 3, 1)):
                for _ in range(groups):
                    if pos + copies <= N:
                        x[b][:, perm[pos + 1: pos + copies]] = x[b][:, perm[pos]: perm[pos] + 1]
                        pos += copies
        return x
    # (coordinates on a 1/64 grid: every product and sum of the distance is exact in fp32, whatever order a BLAS adds them in)
    p = with_copies(rs.randint(-64, 65, (B, 3, N)).astype(np.float32) / 64)
    src = torch.from_numpy(p)
    xyz4 = dev(torch.cat((src.transpose(1, 2), (src ** 2).sum(1).unsqueeze(-1)), -1))
    D = oracle.neg_sqdist_knn(src)
    top2 = torch.topk(D, 2, dim=-1).values
    assert int((top2[..., 0] == top2[..., 1]).sum()) >= N // 4, "the input was meant to share the best value on many rows"
    ref3 = np.sort(oracle.knn_indices(src, k).numpy(), -1)
    for waves in (0, 1, 2, 4):
        got = np.sort(nat.knn(xyz4, None, k, waves=waves).cpu().numpy(), -1)
        assert (got == ref3).all(), f"waves {waves}: {int((got != ref3).any(-1).sum())} rows differ (xyz)"
    f = torch.from_numpy(with_copies(rs.randint(0, 3, size=(B, 64, N)).astype(np.float32) + (rs.rand(B, 64, N) < 0.1).astype(np.float32) * 0.5))
    feat, sq = dev(f.transpose(1, 2)), dev((f ** 2).sum(1))
    ref64 = np.sort(oracle.knn_indices(f, k).numpy(), -1)
    for waves in (0, 1, 8) + ((2, 4) if k <= 20 else ()):
        got = np.sort(nat.knn(feat, sq, k, waves=waves).cpu().numpy(), -1)
        assert (got == ref64).all(), f"waves {waves}: {int((got != ref64).any(-1).sum())} rows differ (features)"
    a, b = nat.knn_pair(feat, sq, xyz4, k)
    assert (np.sort(a.cpu().numpy(), -1) == ref64).all() and (np.sort(b.cpu().numpy(), -1) == ref3).all()
    ft = feat.view(B, N, 4, 4, 4).transpose(3, 4).reshape(B, N, 64).contiguous()
    a, b = nat.knn_pair(feat, sq, xyz4, k, xt=ft)
    assert (np.sort(a.cpu().numpy(), -1) == ref64).all() and (np.sort(b.cpu().numpy(), -1) == ref3).all()


@pytest.mark.parametrize("M,N,K,relu,res", [(300, 200, 64, True, False), (1024, 512, 512, False, True),
                                            (257, 1536, 128, False, False), (128, 64, 1024, True, True)])
def test_linear(nat, M, N, K, relu, res):
    g = torch.Generator().manual_seed(M + N + K)
    xw = torch.randn(M, K + 32, generator=g)
    x = xw[:, :K]                                            # strided rows (ldx > K)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g) if res else None
    xd = dev(xw)[:, :K]
    y = nat.linear(xd, dev(w), dev(b), relu=relu, residual=dev(r) if res else None)
    ref = x.double() @ w.double().t() + b.double()
    if relu:
        ref = ref.clamp_min(0)
    if res:
        ref = ref + r.double()
    err = (y.cpu().double() - ref).abs().max().item()
    assert err <= 4e-6 * math.sqrt(K) + 1e-6, err            # fp32 fma-chain error ~ eps * sqrt(K) * |a.b|


@pytest.mark.parametrize("M,N,K,relu,res,ln", [(33000, 512, 512, False, True, False),   # residual: BK 32 kernels
                                                 (20000, 1536, 512, True, False, True),    # LayerNorm-in, BK 16 kernels
                                                 (6600, 256, 64, False, False, False),     # K = 64
                                                 (7001, 128, 32, True, True, False),       # K = 32: one k-step; ragged M
                                                 (513, 200, 96, False, True, False)])      # ragged N
def test_linear_mfma_shapes_and_staging_variants(nat, M, N, K, relu, res, ln):
    """vcr_linear_args.variant: the 16x16x4 MFMA kernels (bit 4) against the 32x32x2 ones (bit 10) -- two fixed k orders,
    so fp32 rounding apart, both held to fp64 -- and the selectors that must NOT change results: BK 16 / 32 (bits 6 / 3)
    and the register-staged alignment-free kernel (bit 2) are bit-identical to the 32x32x2 default."""
    g = torch.Generator().manual_seed(M + N + K)
    x = dev(torch.randn(M, K, generator=g))
    w = dev(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = dev(torch.randn(N, generator=g))
    r = dev(torch.randn(M, N, generator=g)) if res else None
    lnarg = None
    xr = x
    if ln:
        a_, b_ = dev(torch.rand(K, generator=g) + 0.5), dev(torch.randn(K, generator=g))
        _, st = nat.linear(x, dev(torch.eye(K)), None, want_stats=True)   # row statistics of x itself
        std = x.double().std(-1, keepdim=True)
        xr = (a_.double() * (x.double() - x.double().mean(-1, keepdim=True)) / (std + 1e-6) + b_.double())
        w0 = w
        w, cs, b0 = nat.fold_layernorm(w, b, a_, b_)
        lnarg = (st, cs, 1e-6)
        ref = xr @ w0.double().t() + b.double()
        b = b0
    else:
        ref = x.double() @ w.double().t() + b.double()
    if relu:
        ref = ref.clamp_min(0)
    if res:
        ref = ref + r.double()
    stats_ok = N % 64 == 0
    run = lambda v: nat.linear(x, w, b, relu=relu, residual=r, ln=lnarg, want_stats=stats_ok, variant=v)
    y32, y16 = run(1024), run(16)
    if stats_ok:
        (y32, s32), (y16, s16) = y32, y16
        torch.testing.assert_close(s16, s32, rtol=2e-5, atol=2e-4)
        # the statistics are (sum, sum of squared deviations from the 64-column segment's mean) of the stored outputs
        seg = y32.double().view(M, N // 64, 64)
        torch.testing.assert_close(s32[..., 0].double(), seg.sum(-1), rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(s32[..., 1].double(), ((seg - seg.mean(-1, keepdim=True)) ** 2).sum(-1), rtol=1e-4, atol=1e-4)
    tol = (4e-6 * math.sqrt(K) + 1e-6) * (6 if ln else 1)
    e32, e16 = (y32.double() - ref).abs().max().item(), (y16.double() - ref).abs().max().item()
    assert e32 <= tol and e16 <= tol, (e32, e16)
    assert not torch.equal(y32, y16) or K <= 32            # different k orders (K = 32 on one 16-wide group pair can coincide)
    unpack = lambda o: o[0] if stats_ok else o
    for v in (8 | 1024, 64 | 1024):                        # the BK choice does not change a shape's result
        assert torch.equal(unpack(run(v)), y32), v
    assert torch.equal(unpack(run(8 | 16)), y16) and torch.equal(unpack(run(64 | 16)), y16)
    # 96-row tiles (bit 11; 3 x 4 tiles of 16x16 per wave): every element is the same k-ordered chain as in the
    # 128-row 16x16x4 kernel, so outputs AND statistics are bit-identical, at both k-slabs; bit 12 pins 128 rows
    for v in (2048, 2048 | 8, 2048 | 64, 4096 | 16, 8192, 16384, 8192 | 8):      # (bits 13 / 14: 64- / 32-row tiles)
        o = run(v)
        assert torch.equal(unpack(o), y16), v
        if stats_ok:
            assert torch.equal(o[1], s16), v
    with pytest.raises(nat.VcrHipError):                   # 96-row tiles do not exist for the 32x32x2 shape
        run(2048 | 1024)
    for bad in (8192 | 16384, 8192 | 64, 16384 | 1024):   # nor the lower ones, which are BK 32 kernels
        with pytest.raises(nat.VcrHipError):
            run(bad)
    if not ln and not stats_ok:
        assert torch.equal(nat.linear(x, w, b, relu=relu, residual=r, variant=4), y32)      # register-staged fallback
    for bad in (1, 32, 128, 512):                          # retired selectors are refused
        with pytest.raises(nat.VcrHipError):
            nat.linear(x, w, b, variant=bad)


def test_folded_layernorm_survives_a_large_row_offset(nat):
    """Rows with |mean| >> std (mean ~ 1e3, std ~ 1): the per-segment moments are taken about the segment mean and
    combined exactly (Chan et al.), so the folded LayerNorm still matches torch's x.std() -- the one-pass
    sum(x^2) - sum(x)^2 / n form loses every digit of the variance here."""
    g = torch.Generator().manual_seed(77)
    M, K, N = 1000, 512, 256
    x = torch.randn(M, K, generator=g) + 1000.0 * (torch.rand(M, 1, generator=g) + 0.5)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    a_, b_ = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g)
    xd = dev(x)
    _, st = nat.linear(xd, dev(torch.eye(K)), None, want_stats=True)
    var = st[..., 1].double().sum(-1).cpu()
    mean_seg = st[..., 0].double().cpu() / 64
    mean = mean_seg.mean(-1)
    m2 = var + 64 * ((mean_seg - mean[:, None]) ** 2).sum(-1)
    torch.testing.assert_close((m2 / (K - 1)).sqrt(), x.double().std(-1), rtol=1e-4, atol=0)
    wf, cs, bf = nat.fold_layernorm(dev(w), dev(b), dev(a_), dev(b_))
    y = nat.linear(xd, wf, bf, ln=(st, cs, 1e-6)).cpu().double()
    xn = a_.double() * (x.double() - x.double().mean(-1, keepdim=True)) / (x.double().std(-1, keepdim=True) + 1e-6) + b_.double()
    ref = xn @ w.double().t() + b.double()
    # the GEMM itself runs on un-centred fp32 rows of magnitude 1e3: its rounding (eps * |x| * |w| * sqrt(K)) remains
    assert (y - ref).abs().max().item() < 5e-2, (y - ref).abs().max().item()
    assert (y - ref).abs().mean().item() < 5e-3


@pytest.mark.parametrize("M,k,K,N,store", [(700, 20, 64, 128, True), (300, 40, 128, 256, False), (129, 7, 64, 64, True)])
def test_linear_with_fused_edge_max(nat, M, k, K, N, store):
    """DGCNN's x.max(dim=-1) (vcrnet_model.py:112-118) folded into the producing GEMM's epilogue: the max over each
    point's k consecutive edge rows, by integer atomic max on the post-ReLU values -- bit-equal to the separate
    segmax pass over the stored rows, with and without storing the per-edge rows themselves."""
    g = torch.Generator().manual_seed(M + k)
    x = dev(torch.randn(M * k, K, generator=g))
    w = dev(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = dev(torch.randn(N, generator=g))
    y_ref = nat.linear(x, w, b, relu=True)
    ref = nat.segmax(y_ref, M, k)
    cat = torch.full((M, 512), float("nan"), device="cuda")
    out = cat[:, 192:192 + N]
    out.zero_()
    y = nat.linear(x, w, b, relu=True, segmax=(out, k), store=store)
    assert torch.equal(out, ref)
    assert torch.isnan(cat[:, :192]).all() and torch.isnan(cat[:, 192 + N:]).all()
    if store:
        assert torch.equal(y, y_ref)
    for v in (16, 2048, 8192, 16384):                      # the fused max on every tile height of the 16x16x4 kernels
        out.zero_()
        nat.linear(x, w, b, relu=True, segmax=(out, k), store=False, variant=v)
        torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
        if v == 16:
            out16 = out.clone()
        assert torch.equal(out, out16), v


def test_edgerows_with_max_and_zero_base(nat):
    g = torch.Generator().manual_seed(3)
    B, N, k = 2, 300, 20
    pq = dev(torch.randn(B * N, 128, generator=g))
    idx = dev(torch.randint(0, N, (B * N, k), generator=g).int())
    h0 = nat.edgerows(pq, 64, idx, N)
    cat = torch.full((B * N, 512), float("nan"), device="cuda")
    h1 = nat.edgerows(pq, 64, idx, N, ymax=cat, zero_to=512)
    assert torch.equal(h0, h1)
    assert torch.equal(cat[:, :64], nat.segmax(h0, B * N, k)) and (cat[:, 64:] == 0).all()
    P, Q = pq[:, :64].cpu(), pq[:, 64:].cpu()
    nbr = (idx.cpu().long() + (torch.arange(B * N) // N * N).view(-1, 1))
    ref = torch.relu(P[nbr] + Q[:, None, :]).reshape(B * N * k, 64)
    assert torch.equal(h0.cpu(), ref)


@pytest.mark.parametrize("B,N,k", [(2, 300, 20), (3, 101, 20), (2, 130, 40), (16, 1024, 20)])
def test_edgechain_equals_the_separate_kernels(nat, B, N, k):
    """DGCNN's EdgeConv chain as ONE kernel (no per-edge tensor written) against the same chain run as separate
    launches (edge rows, three GEMMs with the fused max): same fp32 MFMA products in the same order -> the same
    bits; ragged point counts (M not a multiple of the 8- or 4-point group), both k of the path; and against fp64."""
    g = torch.Generator().manual_seed(B * N + k)
    M = B * N
    pq = dev(torch.randn(M, 128, generator=g))
    idx = dev(torch.randint(0, N, (M, k), generator=g).int())
    W = [dev(torch.randn(o, i, generator=g) / math.sqrt(i)) for o, i in ((64, 64), (128, 64), (256, 128))]
    bs = [dev(torch.randn(o, generator=g) * 0.3) for o in (64, 128, 256)]
    cat = torch.full((M, 512), float("nan"), device="cuda")
    h = nat.edgerows(pq, 64, idx, N, ymax=cat, zero_to=512)
    col = 64
    for wt, b in zip(W, bs):
        last = wt.shape[0] == 256
        h = nat.linear(h, wt, b, relu=True, segmax=(cat[:, col:col + wt.shape[0]], k), store=not last)
        col += wt.shape[0]
    out = torch.full((M, 512), float("nan"), device="cuda")
    nat.edgechain(pq, idx, N, W[0], bs[0], W[1], bs[1], W[2], bs[2], out=out)
    assert not torch.isnan(out).any()
    diff = (out - cat).abs().max().item()
    print(f"edgechain B={B} N={N} k={k}: max|chain - separate| = {diff:.2e}")
    assert diff <= 2e-6
    if M <= 1000:
        P, Q = pq[:, :64].cpu().double(), pq[:, 64:].cpu().double()
        nbr = idx.cpu().long() + (torch.arange(M) // N * N).view(-1, 1)
        x = torch.relu(P[nbr] + Q[:, None, :])
        ref = [x.max(1)[0]]
        for wt, b in zip(W, bs):
            x = torch.relu(x @ wt.cpu().double().t() + b.cpu().double())
            ref.append(x.max(1)[0])
        torch.testing.assert_close(out.cpu().double(), torch.cat(ref, 1), atol=2e-5, rtol=1e-5)
    with pytest.raises(nat.VcrHipError):                   # other k: the caller falls back to the separate kernels
        nat.edgechain(pq, idx[:, :10].contiguous(), N, W[0], bs[0], W[1], bs[1], W[2], bs[2])


def test_rows4_with_first_edgeconv_projection(nat):
    """DGCNN's conv1 per point (K = 3) in the pass that lays the points out as rows: equal to the K-padded MFMA GEMM it
    replaced (an fp32 fma chain in k order either way)."""
    g = torch.Generator().manual_seed(5)
    B, N = 3, 333
    x = torch.randn(B, 3, N, generator=g)
    w = torch.zeros(128, 32)
    w[:, :3] = torch.randn(128, 3, generator=g)
    b = torch.randn(128, generator=g)
    rows, pq = nat.rows4_pq(dev(x), dev(w), dev(b))
    assert torch.equal(rows, nat.to_rows4(dev(x)))
    xin = torch.zeros(B * N, 32)
    xin[:, :3] = x.transpose(1, 2).reshape(B * N, 3)
    ref = nat.linear(dev(xin), dev(w), dev(b))
    d = (pq - ref).abs().max().item()
    print(f"rows4_pq vs padded GEMM: max|diff| = {d:.2e}")
    assert d <= 1e-6
    torch.testing.assert_close(pq.cpu().double(), xin.double() @ w.double().t() + b.double(), atol=2e-6, rtol=1e-6)


def test_layernorm(nat):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1000, 512, generator=g) * 3 + 0.5
    a, b = torch.rand(512, generator=g) + 0.5, torch.randn(512, generator=g)
    r = torch.randn(1000, 512, generator=g)
    xyz4 = torch.randn(1000, 4, generator=g)
    y = nat.layernorm(dev(x), dev(a), dev(b))
    ref = oracle.layer_norm(x, a, b)
    torch.testing.assert_close(y.cpu(), ref, atol=5e-6, rtol=1e-5)
    y2, side = nat.layernorm(dev(x), dev(a), dev(b), residual=dev(r), xyz4=dev(xyz4))
    torch.testing.assert_close(y2.cpu(), ref + r, atol=5e-6, rtol=1e-5)
    torch.testing.assert_close(side.cpu()[:, :3], xyz4[:, :3], atol=0, rtol=0)
    torch.testing.assert_close(side.cpu()[:, 3], ((ref + r) ** 2).sum(-1), atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("N,k", [(256, 20), (512, 40), (96, 7)])
def test_edgeconv_and_gathermax(nat, W, N, k):
    g = torch.Generator().manual_seed(N)
    B = 2
    x = torch.randn(B, 64, N, generator=g).abs()
    idx = oracle.knn_indices(x, k)
    act = lambda t: F.leaky_relu(t, 0.0)
    gf = oracle.graph_feature(x, idx)
    h1 = act(F.conv2d(gf, W["emb_nn.convDG1.0.weight"], W["emb_nn.convDG1.0.bias"]))
    x1 = h1.max(-1)[0]
    x2 = act(F.conv2d(h1, W["emb_nn.convDG2.0.weight"], W["emb_nn.convDG2.0.bias"])).max(-1)[0]
    w1 = W["emb_nn.convDG1.0.weight"].view(128, 128)
    wpq = torch.cat((w1[:, :64], w1[:, 64:]), 0)
    bpq = torch.cat((torch.zeros(128), W["emb_nn.convDG1.0.bias"]))
    rows = dev(x.transpose(1, 2).reshape(B * N, 64))
    pq = nat.linear(rows, dev(wpq), dev(bpq))
    gx1, gx2 = nat.edgeconv(pq, dev(idx.int().reshape(B * N, k)), N, dev(W["emb_nn.convDG2.0.weight"].view(128, 128)),
                            dev(W["emb_nn.convDG2.0.bias"]))
    torch.testing.assert_close(gx1.cpu().view(B, N, 128), x1.transpose(1, 2), atol=3e-6, rtol=1e-5)
    torch.testing.assert_close(gx2.cpu().view(B, N, 128), x2.transpose(1, 2), atol=3e-6, rtol=1e-5)
    # SN1-style single conv: gather + max
    gf3 = oracle.graph_feature(x2, idx)
    x3 = act(F.conv2d(gf3, W["emb_nn.convSN1.0.weight"], W["emb_nn.convSN1.0.bias"])).max(-1)[0]
    w3 = W["emb_nn.convSN1.0.weight"].view(256, 256)
    pq3 = nat.linear(dev(x2.transpose(1, 2).reshape(B * N, 128)), dev(torch.cat((w3[:, :128], w3[:, 128:]), 0)),
                     dev(torch.cat((torch.zeros(256), W["emb_nn.convSN1.0.bias"]))))
    gx3 = nat.gathermax(pq3, 256, dev(idx.int().reshape(B * N, k)), N)
    torch.testing.assert_close(gx3.cpu().view(B, N, 256), x3.transpose(1, 2), atol=3e-6, rtol=1e-5)


@pytest.mark.parametrize("B,N,k,C", [(32, 1024, 20, 256), (48, 768, 20, 256), (24, 1000, 40, 256), (64, 333, 20, 96),
                                     (200, 64, 20, 32), (32, 1066, 20, 256), (32, 1067, 20, 256), (4, 1024, 20, 256)])
def test_gathermax_lds_and_l2_paths_are_exact(nat, B, N, k, C):
    """vcr_gathermax_f32 picks the LDS-staged kernel on large grids of clouds of <= 1066 points (k = 20 / 40) and the L2
    gathers otherwise (the last two shapes): y = relu(max_j P[nbr_j] + Q) either way, to the bit (a max is exact, the add
    and the ReLU are the same two fp32 operations as in torch)."""
    g = torch.Generator().manual_seed(B * N + k)
    pq = dev(torch.randn(B * N, 2 * C, generator=g))
    idx = dev(torch.randint(0, N, (B * N, k), generator=g, dtype=torch.int32))
    y = nat.gathermax(pq, C, idx, N)
    P, Q = pq[:, :C].view(B, N, C), pq[:, C:].view(B, N, C)
    gathered = torch.gather(P.unsqueeze(1).expand(B, N, N, C), 2,
                            idx.view(B, N, k, 1).long().expand(B, N, k, C)) if B * N * N * C < 2 ** 27 else None
    if gathered is None:
        gathered = torch.stack([P[b][idx.view(B, N, k)[b].long()] for b in range(B)])
    ref = torch.relu(gathered.max(2).values + Q).view(B * N, C)
    assert torch.equal(y, ref)
    # every form the args can force (vcr_gathermax_args.variant: 1 = L2, 32 / 16 / 8 = LDS slices) gives the same bits, or
    # refuses when the slice of one cloud does not fit a workgroup's LDS
    for variant in (1, 32, 16, 8):
        fits = N * (variant + 4) * 4 <= 160 * 1024
        if variant == 1 or fits:
            assert torch.equal(nat.gathermax(pq, C, idx, N, variant=variant), ref), variant
        else:
            with pytest.raises(nat.VcrHipError):
                nat.gathermax(pq, C, idx, N, variant=variant)
    with pytest.raises(nat.VcrHipError):
        nat.gathermax(pq, C, idx, N, variant=5)
    # vcr_gathermax_args.order: the L2 form's waves take the points of every cloud in a caller-given order (the forward hands
    # it the clouds' Morton ranking): every point is still served once, by the same arithmetic
    order = dev(torch.stack([torch.randperm(N, generator=g) for _ in range(B)]).to(torch.int32).view(-1))
    assert torch.equal(nat.gathermax(pq, C, idx, N, variant=1, order=order), ref)
    assert torch.equal(nat.gathermax(pq, C, idx, N, order=order), ref)          # (the LDS forms ignore it)


@pytest.mark.parametrize("B,N,k", [(2, 300, 20), (3, 101, 20), (2, 130, 40), (16, 1024, 20), (1, 203, 7)])
def test_edgeconv_bf16x3(nat, B, N, k):
    """vcr_edgeconv_bf16x3_f32 (convDG2 as exact 3-way bf16 splits on the bf16 matrix pipe): x1 -- a max over the same
    fp32 rows -- is bit-identical to the fp32 kernel's, x2 is as close to fp64 as the fp32-MFMA kernel's; ragged point
    counts (M not a multiple of the 8- / 4-point group), both k of the path; any other k is refused."""
    g = torch.Generator().manual_seed(B * N + k)
    M = B * N
    pq = dev(torch.randn(M, 256, generator=g))
    idx = dev(torch.randint(0, N, (M, k), generator=g).int())
    w2 = dev(torch.randn(128, 128, generator=g) / math.sqrt(128))
    b2 = dev(torch.randn(128, generator=g) * 0.3)
    if k not in (20, 40):
        with pytest.raises(nat.VcrHipError):
            nat.edgeconv(pq, idx, N, w2, b2, bf16x3=True)
        return
    x1, x2 = nat.edgeconv(pq, idx, N, w2, b2)
    y1, y2 = nat.edgeconv(pq, idx, N, w2, b2, bf16x3=True)
    assert torch.equal(x1, y1)
    P, Q = pq[:, :128].cpu().double(), pq[:, 128:].cpu().double()
    nbr = idx.cpu().long() + (torch.arange(M) // N * N).view(-1, 1)
    h = torch.relu(P[nbr] + Q[:, None, :]).float().double()      # the kernels round P + Q to fp32 before the GEMM
    ref = torch.relu((h @ w2.cpu().double().t()).max(1)[0] + b2.cpu().double())
    e32, e3 = (x2.cpu().double() - ref).abs().max().item(), (y2.cpu().double() - ref).abs().max().item()
    print(f"edgeconv B={B} N={N} k={k}: max|x2 - fp64|  fp32-MFMA {e32:.2e}  bf16x3 {e3:.2e}")
    assert e3 <= 4e-6 * math.sqrt(128) + 1e-6 and e32 <= 4e-6 * math.sqrt(128) + 1e-6


@pytest.mark.parametrize("bf16x3", [False, True])
@pytest.mark.parametrize("N,shift", [(256, 0), (192, 0), (1024, 2), (100, 1), (300, 0)])
def test_sdpa(nat, N, shift, bf16x3):
    """bf16x3 = the opt-in exact-split kernel on the bf16 matrix pipe (vcr_sdpa_bf16x3_f32): same tolerance."""
    g = torch.Generator().manual_seed(N)
    nb, h = 4, 4
    q, k, v = (torch.randn(nb, N, h * 128, generator=g) for _ in range(3))
    kk, vv = torch.roll(k, -shift, 0), torch.roll(v, -shift, 0)          # batch b uses kv of (b + shift) % nb
    split = lambda t: t.view(nb, N, h, 128).transpose(1, 2)
    ref = oracle.attention(split(q) * 2, split(kk) * 2, split(vv)).transpose(1, 2).reshape(nb * N, h * 128)
    qkv = dev(torch.cat((q * 2, k * 2, v), -1).view(nb * N, 3 * h * 128))   # fused-QKV row layout, pitch 1536
    run = lambda mode: nat.sdpa(qkv[:, :512], qkv[:, 512:1024], qkv[:, 1024:], nb, h, N, N, 1 / math.sqrt(128),
                                kv_batch_shift=shift, bf16x3=mode).cpu()
    if not bf16x3:
        torch.testing.assert_close(run(False), ref, atol=5e-6, rtol=1e-5)
        return
    # the split kernel is a different, equally accurate summation: hold it to the same formula in fp64, where its
    # error must not exceed the fp32-MFMA kernel's (which the fp32 oracle pins above)
    ref64 = oracle.attention(split(q).double() * 2, split(kk).double() * 2, split(vv).double())
    ref64 = ref64.transpose(1, 2).reshape(nb * N, h * 128)
    e32, e3 = (run(False).double() - ref64).abs().max().item(), (run(True).double() - ref64).abs().max().item()
    print(f"sdpa N={N}: max|err| vs fp64  fp32-MFMA {e32:.2e}  bf16x3 {e3:.2e}")
    assert e3 <= 1.25 * e32 + 5e-7 and e3 < 2e-5


@pytest.mark.parametrize("nb,N,nk,shift,groups", [(32, 1024, 1024, 0, False), (32, 1000, 1000, 16, False), (48, 700, 650, 3, False),
                                                   (24, 1024, 1024, 1, True), (64, 300, 300, 0, False)])
def test_sdpa_persistent_kernel_is_bit_identical(nat, nb, N, nk, shift, groups):
    """vcr_sdpa_args.variant 2: 2 x CUs workgroups walk the tile kernel's work items (next item's Q / K / V requested before the
    current item's epilogue).  Same arithmetic in the same order: every bit of the output equals the tile kernel's -- full and
    ragged query blocks and key tiles, a key-batch shift, the grouped (encoder + decoder) form, item counts that do and do not
    divide over the workgroups.  A call it does not cover (here: fewer than two items per workgroup) runs the tile kernel."""
    g = torch.Generator().manual_seed(nb * N + nk)
    h = 4
    if groups:
        qkv = dev(torch.randn(nb * N, 6 * h * 128, generator=g))          # [enc Q|K|V | dec Q|K|V]
        kw = dict(kv_batch_shift=shift, groups=(2, 1536, 1536, 1536))
        run = lambda var: nat.sdpa(qkv[:, :512], qkv[:, 512:1024], qkv[:, 1024:1536], nb, h, N, nk, 1 / math.sqrt(128), variant=var, **kw)
    else:
        q = dev(torch.randn(nb * N, h * 128, generator=g))
        kv = dev(torch.randn(nb * nk, 2 * h * 128, generator=g))
        run = lambda var: nat.sdpa(q, kv[:, :512], kv[:, 512:], nb, h, N, nk, 1 / math.sqrt(128), kv_batch_shift=shift, variant=var)
    a, b = run(1), run(2)
    assert torch.equal(a, b), (a - b).abs().max().item()
    assert torch.equal(run(0), a)
    small = dev(torch.randn(2 * 256, 3 * h * 128, generator=g))             # 16 items: the tile kernel whatever the selector says
    f = lambda var: nat.sdpa(small[:, :512], small[:, 512:1024], small[:, 1024:], 2, h, 256, 256, 1 / math.sqrt(128), variant=var)
    assert torch.equal(f(1), f(2))


def test_sdpa_bf16x3_masked_ragged_and_error_vs_fp64(nat):
    """Key mask + ragged nq != nk through the exact-split kernel; its error against an fp64 reference is no larger than
    the fp32-MFMA kernel's (the six-product split carries fp32-GEMM accuracy); statistics forms are refused."""
    g = torch.Generator().manual_seed(31)
    nb, h, NQ, NK = 2, 4, 300, 170
    q = torch.randn(nb, NQ, h * 128, generator=g) * 1.5
    k, v = (torch.randn(nb, NK, h * 128, generator=g) for _ in range(2))
    keep = torch.rand(nb, NK, generator=g) < 0.7
    sp = lambda t, n: t.view(nb, n, h, 128).transpose(1, 2).double()
    s = torch.matmul(sp(q, NQ), sp(k, NK).transpose(-2, -1)) / math.sqrt(128)
    ref = torch.matmul(torch.softmax(s.masked_fill(~keep.view(nb, 1, 1, NK), float("-inf")), -1), sp(v, NK))
    ref = ref.transpose(1, 2).reshape(nb * NQ, h * 128)
    args = (dev(q.view(nb * NQ, -1)), dev(k.view(nb * NK, -1)), dev(v.view(nb * NK, -1)), nb, h, NQ, NK, 1 / math.sqrt(128))
    e32 = (nat.sdpa(*args, key_keep=dev(keep.to(torch.uint8))).cpu().double() - ref).abs().max().item()
    e3 = (nat.sdpa(*args, key_keep=dev(keep.to(torch.uint8)), bf16x3=True).cpu().double() - ref).abs().max().item()
    print(f"sdpa masked ragged: max|err| vs fp64  fp32-MFMA {e32:.2e}  bf16x3 {e3:.2e}")
    assert e3 < 3e-6 and e3 <= 1.5 * e32 + 2e-7
    with pytest.raises(nat.VcrHipError):
        nat.sdpa(*args, want_rowstat=True, bf16x3=True)


def test_sdpa_masked_and_rowstat(nat):
    g = torch.Generator().manual_seed(9)
    nb, h, N = 2, 4, 160
    q, k, v = (torch.randn(nb, N, h * 128, generator=g) for _ in range(3))
    keep = torch.rand(nb, N, generator=g) < 0.7
    split = lambda t: t.view(nb, N, h, 128).transpose(1, 2)
    s = torch.matmul(split(q), split(k).transpose(-2, -1)) / math.sqrt(128)
    p = torch.softmax(s.masked_fill(~keep.view(nb, 1, 1, N), -1e9), -1)
    ref = torch.matmul(p, split(v)).transpose(1, 2).reshape(nb * N, h * 128)
    sm = s.masked_fill(~keep.view(nb, 1, 1, N), float("-inf"))
    for _ in range(1):
        out, rs = nat.sdpa(dev(q.view(nb * N, -1)), dev(k.view(nb * N, -1)), dev(v.view(nb * N, -1)), nb, h, N, N,
                           1 / math.sqrt(128), key_keep=dev(keep.to(torch.uint8)), want_rowstat=True)
        torch.testing.assert_close(out.cpu(), ref, atol=5e-6, rtol=1e-5)
        torch.testing.assert_close(rs.cpu()[..., 0], sm.max(-1)[0], atol=1e-5, rtol=1e-5)
        torch.testing.assert_close(rs.cpu()[..., 1], torch.exp(sm - sm.max(-1, keepdim=True)[0]).sum(-1), atol=1e-4, rtol=1e-5)
        # statistics-only launch (no V, no output) that also keeps the scaled scores
        ld = (N + 31) // 32 * 32
        sc = torch.full((nb, h, N, ld), float("nan"), device="cuda")
        _, rs2 = nat.sdpa(dev(q.view(nb * N, -1)), dev(k.view(nb * N, -1)), None, nb, h, N, N, 1 / math.sqrt(128),
                          want_rowstat=True, pv=False, score_out=sc)
        torch.testing.assert_close(rs2.cpu()[..., 0], s.max(-1)[0], atol=1e-5, rtol=1e-5)
        torch.testing.assert_close(sc.cpu()[..., :N], s, atol=1e-5, rtol=1e-5)
        assert torch.isinf(sc.cpu()[..., N:]).all()


@pytest.mark.parametrize("N,mode", [(256, 0), (200, 0), (1024, 0), (256, 1)])
def test_softcorr(nat, N, mode):
    g = torch.Generator().manual_seed(N + mode)
    B, E = 2, 512
    se = torch.randn(B, E, N, generator=g) * 0.3
    te = se[:, :, torch.randperm(N, generator=g)] + 0.05 * torch.randn(B, E, N, generator=g)
    src, tgt = torch.rand(B, 3, N, generator=g), torch.rand(B, 3, N, generator=g)
    ref = (oracle.head_topk_whole(se, te, src, tgt) if mode == 0 else oracle.head_by_dis(se, te, src, tgt))[1]
    side = lambda e, p: dev(torch.cat((p.transpose(1, 2), (e ** 2).sum(1).unsqueeze(-1)), -1).reshape(B * N, 4))
    corr4 = nat.softcorr(dev(se.transpose(1, 2).reshape(B * N, E)), dev(te.transpose(1, 2).reshape(B * N, E)),
                         side(se, src), side(te, tgt), B, N, N, mode=mode, scale=1 / math.sqrt(E))
    torch.testing.assert_close(corr4.cpu().view(B, N, 4)[..., :3], ref.transpose(1, 2), atol=3e-6, rtol=1e-5)
    # with scratch the launch may deal the streamed rows to several workgroups per owner block (here: a grid far below
    # one round of the chip) and merge the partial (max, sum, weighted xyz) records: the same soft correspondences
    args = (dev(se.transpose(1, 2).reshape(B * N, E)), dev(te.transpose(1, 2).reshape(B * N, E)), side(se, src), side(te, tgt),
            B, N, N)
    c_split = nat.softcorr(*args, mode=mode, scale=1 / math.sqrt(E), split=True)
    torch.testing.assert_close(c_split, corr4, atol=2e-6, rtol=1e-5)
    torch.testing.assert_close(c_split.cpu().view(B, N, 4)[..., :3], ref.transpose(1, 2), atol=3e-6, rtol=1e-5)


@pytest.mark.parametrize("B,N", [(20, 1024), (3, 1500)])
def test_softcorr_split_grids(nat, B, N):
    """Grids that do not fill the chip evenly (20 clouds of 1024: 320 two-tile workgroups on 256 CUs; 3 of 1500: 72, ragged):
    the split launch equals the plain one to rounding and is really taken."""
    g = torch.Generator().manual_seed(B)
    E = 512
    q, k = dev(torch.randn(B * N, E, generator=g) * 0.3), dev(torch.randn(B * N, E, generator=g) * 0.3)
    sd = lambda e: torch.cat((torch.rand(len(e), 3, device=e.device), (e.double() ** 2).sum(1, keepdim=True).float()), 1)
    qs, ks = sd(q), sd(k)
    c0 = nat.softcorr(q, k, qs, ks, B, N, N, mode=0)
    c1 = nat.softcorr(q, k, qs, ks, B, N, N, mode=0, split=True)
    torch.testing.assert_close(c1, c0, atol=2e-6, rtol=1e-5)
    assert not torch.equal(c1, c0)


def test_rigid_svd(nat):
    rs = np.random.RandomState(0)
    B, K = 6, 300
    src = torch.from_numpy(rs.uniform(-1, 1, (B, 3, K)).astype(np.float32))
    corr = torch.from_numpy(rs.uniform(-1, 1, (B, 3, K)).astype(np.float32)) * 0.3
    ang = rs.uniform(0, 1, B)
    for i in range(B):   # mostly rigid pairs with noise; sample 4 is a mirrored cloud (reflection branch)
        c, s = np.cos(ang[i]), np.sin(ang[i])
        Rz = torch.tensor([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=torch.float32)
        corr[i] = 0.05 * corr[i] + Rz @ src[i] + torch.tensor([0.1, 0.2, -0.3]).view(3, 1)
    corr[4] = torch.diag(torch.tensor([1.0, 1.0, -1.0])) @ src[4]
    cfg = oracle.OracleConfig(record={})
    Rr, tr = oracle.rigid_svd(src, corr, cfg)
    R, t, Rb, tb, H = nat.rigid_svd(dev(src.transpose(1, 2)), dev(corr.transpose(1, 2)), want_h=True)
    torch.testing.assert_close(H.cpu(), cfg.record["H"], atol=2e-4, rtol=1e-5)
    torch.testing.assert_close(R.cpu(), Rr, atol=1e-5, rtol=0)          # BASELINE tolerance is 1e-4
    torch.testing.assert_close(t.cpu(), tr, atol=1e-5, rtol=0)
    assert torch.allclose(torch.det(R.cpu()), torch.ones(B), atol=1e-5)
    torch.testing.assert_close(Rb.cpu(), Rr.transpose(1, 2), atol=1e-5, rtol=0)
    torch.testing.assert_close(tb.cpu(), -torch.matmul(Rr.transpose(1, 2), tr.unsqueeze(2)).squeeze(2), atol=1e-5, rtol=0)


def test_rigid_svd_degenerate_pairs_still_give_a_rotation(nat):
    """Rank-deficient covariances (coplanar, collinear, single-point correspondences) leave R under-determined -- in
    the reference LAPACK's completion is arbitrary -- but the result must be a proper rotation that still maps the
    centred source onto the centred correspondences wherever that is defined."""
    rs = np.random.RandomState(3)
    B, K = 4, 40
    src = torch.from_numpy(rs.uniform(-1, 1, (B, K, 3)).astype(np.float32))
    corr = src.clone()
    corr[0, :, 2] = 0.25                                             # rank 2: correspondences in a plane
    line = torch.tensor([0.3, -0.5, 0.8])
    corr[1] = torch.from_numpy(rs.uniform(-1, 1, (K, 1)).astype(np.float32)) * line + 0.1    # rank 1: on a line
    corr[2] = torch.tensor([0.2, 0.1, -0.4])                         # rank 0: every source matched to ONE target
    R, t, Rb, tb = nat.rigid_svd(dev(src), dev(corr))                # sample 3: full rank, R = I
    R, t = R.cpu(), t.cpu()
    assert torch.isfinite(R).all() and torch.isfinite(t).all()
    assert torch.allclose(torch.det(R), torch.ones(B), atol=1e-5)
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(B, 3, 3), atol=1e-5)
    assert torch.allclose(R[2], torch.eye(3), atol=1e-6) and torch.allclose(R[3], torch.eye(3), atol=1e-5)
    moved = src @ R.transpose(1, 2) + t.unsqueeze(1)                 # the mean is always matched
    assert torch.allclose(moved.mean(1), corr.mean(1), atol=1e-5)
    # rank 2: Kabsch optimum is unique -- compare with the oracle
    Ro, to = oracle.rigid_svd(src[:1].transpose(1, 2), corr[:1].transpose(1, 2))
    assert torch.allclose(R[0], Ro[0], atol=1e-4)


@pytest.mark.parametrize("name", ["whole_n1024_b2", "whole_n256_b2"])
def test_feature_space_knn_is_bit_exact_vs_reference(nat, W, name):
    """The discrete step that decides parity: conv1/conv2 features, |x|^2 and the feature-space distance
    matrix are computed with the reference's own rounding order (bias-first fma chains, ATen's cascade sum,
    k-ascending sgemm chain), so features match the recorded reference BITWISE and every neighbour set is
    identical -- no near-tie flips.  (The reference's own rounding depends on oneDNN's kernel choice: with one
    thread, or batch 1, its conv adds the bias last instead of first; the goldens used here were recorded with
    8 threads and batch 2, the multi-threaded path that the bias-first order reproduces.)"""
    g = golden(name)
    k = int(g["k"])
    cs = max(1, int(g["cstride"]) // 2)
    for cloud in ("src", "tgt"):
        x = torch.from_numpy(g[cloud])
        xyz4, f64, sq = nat.pointwise(dev(x), dev(W["emb_nn.conv1_lpd.weight"].view(64, 3)), dev(W["emb_nn.conv1_lpd.bias"]),
                                      dev(W["emb_nn.conv2_lpd.weight"].view(64, 64)), dev(W["emb_nn.conv2_lpd.bias"]))
        got = f64.cpu().transpose(1, 2)[:, ::cs].numpy()
        assert np.array_equal(got, g[f"it0_x64_{cloud}"]), "conv features differ from the reference bitwise"
        idx = nat.knn(f64, sq, k).cpu().numpy()
        a, b = np.sort(idx, -1), np.sort(g[f"it0_idx_feat_{cloud}"].astype(np.int64), -1)
        assert np.array_equal(a, b), f"{int((a != b).any(-1).sum())} neighbour sets differ"
        idx3 = nat.knn(xyz4, None, k).cpu().numpy()
        a, b = np.sort(idx3, -1), np.sort(g[f"it0_idx_xyz_{cloud}"].astype(np.int64), -1)
        assert np.array_equal(a, b)


@pytest.mark.parametrize("M,N,K,relu,res", [(300, 200, 64, True, False), (1024, 512, 512, False, True),
                                            (257, 1536, 128, False, False), (128, 64, 1024, True, True),
                                            (130, 36, 32, False, True), (300, 200, 96, True, True), (1, 4, 32, False, True)])
def test_linear_bf16x3(nat, M, N, K, relu, res):
    """The bf16-pipe linear splits every operand exactly into three bf16 pieces and keeps the six leading
    partial products: same fp32-GEMM error bound as the fp32-MFMA kernel.  (Ragged M / N with a residual: its tile is
    requested with clamped rows / columns during the last slab; K = 32, 64, 96: one, two, three slabs = the three
    instantiations of the slab body.)"""
    g = torch.Generator().manual_seed(M + N + K + 7)
    xw = torch.randn(M, K + 32, generator=g)
    x = xw[:, :K]
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g) if res else None
    planes = nat.split_bf16x3(dev(w))
    assert torch.equal(planes.view(torch.bfloat16).float().sum(0).cpu().view(N, K), w)      # the split is exact
    y = nat.linear_bf16x3(dev(xw)[:, :K], planes, N, dev(b), relu=relu, residual=dev(r) if res else None)
    ref = x.double() @ w.double().t() + b.double()
    if relu:
        ref = ref.clamp_min(0)
    if res:
        ref = ref + r.double()
    err = (y.cpu().double() - ref).abs().max().item()
    assert err <= 4e-6 * math.sqrt(K) + 1e-6, err


def test_linear_with_fused_layernorm(nat):
    """the producer's epilogue emits per-row, per-64-column (sum, sum of squared deviations from the
    segment mean) partials; the consumer runs the GEMM
    on the weight folded with the LayerNorm affine and applies (mean, 1/(std_unbiased+eps)) in its epilogue --
    equal to LayerNorm followed by Linear."""
    g = torch.Generator().manual_seed(11)
    M, K, N = 1000, 512, 1536
    x0 = torch.randn(M, 128, generator=g)
    w0 = torch.randn(K, 128, generator=g) / math.sqrt(128)
    b0 = torch.randn(K, generator=g) * 0.5 + 0.3            # non-zero mean rows: exercises the variance formula
    r0 = torch.randn(M, K, generator=g) * 2
    a, b = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g)
    w1 = torch.randn(N, K, generator=g) / math.sqrt(K)
    b1 = torch.randn(N, generator=g)
    x, stats = nat.linear(dev(x0), dev(w0), dev(b0), residual=dev(r0), want_stats=True)     # producer
    xs = x.cpu()
    seg_m2 = lambda t: ((t.view(M, K // 64, 64) - t.view(M, K // 64, 64).mean(-1, keepdim=True)) ** 2).sum(-1)
    torch.testing.assert_close(stats.cpu()[..., 0].sum(1), xs.sum(1), atol=2e-4, rtol=1e-5)
    torch.testing.assert_close(stats.cpu()[..., 1], seg_m2(xs), atol=2e-4, rtol=1e-5)
    wf, cs, bf = nat.fold_layernorm(dev(w1), dev(b1), dev(a), dev(b))
    torch.testing.assert_close(wf.cpu(), w1 * a, atol=0, rtol=0)
    torch.testing.assert_close(cs.cpu().double(), (w1 * a).double().sum(1), atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(bf.cpu().double(), b1.double() + w1.double() @ b.double(), atol=1e-6, rtol=1e-6)
    y = nat.linear(x, wf, bf, relu=True, ln=(stats, cs, 1e-6))                               # consumer
    ref = torch.relu(oracle.layer_norm(xs.double(), a.double(), b.double()) @ w1.double().t() + b1.double())
    err = (y.cpu().double() - ref).abs().max().item()
    assert err <= 5e-5, err
    y_unfused = nat.linear(nat.layernorm(x, dev(a), dev(b)), dev(w1), dev(b1), relu=True)
    assert (y.cpu() - y_unfused.cpu()).abs().max().item() <= 5e-5
    # the same producer / consumer pair through the exact-split kernel (vcr_linear_bf16x3_f32)
    x3, stats3 = nat.linear_bf16x3(dev(x0), nat.split_bf16x3(dev(w0)), K, dev(b0), residual=dev(r0), want_stats=True)
    torch.testing.assert_close(x3.cpu(), xs, atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(stats3.cpu()[..., 0].sum(1), x3.cpu().sum(1), atol=2e-4, rtol=1e-5)
    torch.testing.assert_close(stats3.cpu()[..., 1], seg_m2(x3.cpu()), atol=2e-4, rtol=1e-5)
    y3 = nat.linear_bf16x3(x3, nat.split_bf16x3(wf), N, bf, relu=True, ln=(stats3, cs, 1e-6))
    ref3 = torch.relu(oracle.layer_norm(x3.cpu().double(), a.double(), b.double()) @ w1.double().t() + b1.double())
    err3 = (y3.cpu().double() - ref3).abs().max().item()
    print(f"fused LayerNorm + linear: max|err| vs fp64  fp32-MFMA {err:.2e}  bf16x3 {err3:.2e}")
    assert err3 <= 5e-5, err3


@pytest.mark.parametrize("B,N,k", [(32, 1024, 20), (2, 1024, 20), (16, 2048, 20), (4, 777, 9), (4, 1024, 40), (32, 1024, 40), (18, 1500, 33)])
def test_knn_pair_equals_the_two_launches(nat, B, N, k):
    """vcr_knn_pair_f32 (LPDNet's two searches in one launch; falls back to two launches outside the path's regime):
    the same indices as the self-contained calls, ties included."""
    rs = np.random.RandomState(B * N + k)
    feat = dev(torch.from_numpy(rs.randint(0, 4, (B, N, 64)).astype(np.float32) + rs.randn(B, N, 64).astype(np.float32) * (B % 3 == 0)))
    xyz = rs.randint(0, 12, (B, N, 3)).astype(np.float32) if B == 4 else rs.randn(B, N, 3).astype(np.float32)
    x4 = dev(torch.from_numpy(np.concatenate((xyz, (xyz ** 2).sum(-1, keepdims=True)), -1)))
    sq = (feat ** 2).sum(-1)
    a, b = nat.knn_pair(feat, sq, x4, k)
    assert torch.equal(a, nat.knn(feat, sq, k)) and torch.equal(b, nat.knn(x4, None, k))


@pytest.mark.parametrize("B,N,k,kind", [(8, 2048, 20, "smooth"), (4, 4096, 40, "smooth"), (8, 2048, 20, "random"), (16, 1024, 20, "smooth"),
                                        (8, 2040, 20, "dup"), (8, 2048, 20, "lattice"), (6, 3000, 40, "lattice"), (4, 4096, 20, "equal"),
                                        (8, 4096, 20, "outlier"), (4, 8192, 40, "smooth"), (4, 5000, 20, "smooth"), (3, 8192, 20, "lattice"),
                                        (3, 6000, 40, "random")])
def test_knn_ordered_search_keeps_the_sets(nat, B, N, k, kind):
    """The ordered search (vcr_knn_order_f32 + vcr_knn_args.perm: Morton ranking, tiles skipped by their balls) against the plain
    pair launch: the same neighbour SET on every row -- smooth features (a function of the coordinates, as the stem's are),
    features unrelated to the coordinates (nothing can be skipped), every point twice (the rank-0 rule picks by point index),
    lattices (thousands of exact ties: the replay) and a cloud of one repeated point."""
    rs = np.random.RandomState(B * N + k)
    if kind == "lattice":
        xyz = rs.randint(0, 9, (B, N, 3)).astype(np.float32) / 8
    elif kind == "equal":
        xyz = np.zeros((B, N, 3), np.float32) + 0.25
    else:
        xyz = rs.rand(B, N, 3).astype(np.float32) - 0.5
    if kind == "dup":
        xyz[:, N // 2:] = xyz[:, : N - N // 2]
    if kind == "outlier":                                  # one far return per cloud: the ranking's box is then mean +- 4 sigma
        xyz[:, 17] = np.float32(200.0)
    w1, w2 = rs.randn(3, 64).astype(np.float32) * 0.8, rs.randn(64, 64).astype(np.float32) * 0.2
    feat = np.maximum(np.maximum(xyz @ w1 + 0.1, 0) @ w2 + 0.05, 0)
    if kind == "random":
        feat = rs.randn(B, N, 64).astype(np.float32)
    feat = dev(torch.from_numpy(np.ascontiguousarray(feat)))
    x4 = dev(torch.from_numpy(np.concatenate((xyz, (xyz ** 2).sum(-1, keepdims=True)), -1).astype(np.float32)))
    sq = (feat ** 2).sum(-1).contiguous()
    ft = feat.view(B, N, 4, 4, 4).transpose(3, 4).reshape(B, N, 64).contiguous()
    order = nat.knn_order(x4, ft, sq)
    perm = order["perm"].long().cpu()
    assert all(torch.equal(torch.sort(perm[b]).values, torch.arange(N)) for b in range(B))
    assert torch.equal(torch.gather(x4, 1, order["perm"].long()[..., None].expand(-1, -1, 4)), order["xyz4_p"])
    if kind == "outlier":
        # (the clamped box is the device's own fp32 mean / sigma: not re-derived here)
        a0, b0 = nat.knn_pair(feat, sq, x4, k, xt=ft)
        a1, b1 = nat.knn_pair(feat, sq, x4, k, xt=ft, order=order)
        for plain, ordered in ((a0, a1), (b0, b1)):
            assert torch.equal(torch.sort(plain, -1).values, torch.sort(ordered, -1).values)
        # consecutive ranks stay spatial neighbours: the mean step between rank-adjacent points is what the same cloud WITHOUT its
        # outlier gives (with the true, 200-wide box the other 4095 points would share ~5 levels per axis: steps ~1.6x longer)
        def step(x4_, perm_):
            p = x4_[0, perm_[0].long(), :3].cpu().numpy()
            p = p[np.abs(p).max(1) < 10]
            return float(np.linalg.norm(np.diff(p, axis=0), axis=1).mean())
        clean = x4.clone()
        clean[:, 17, :3] = 0.0
        clean[:, 17, 3] = 0.0
        ref = step(clean, nat.knn_order(clean)["perm"])
        got = step(x4, order["perm"])
        print(f"rank-adjacent step: with the outlier {got:.4f}, clean cloud {ref:.4f}")
        assert got < 1.25 * ref, (got, ref)
        return
    # the ranking IS the sort by (30-bit Morton code of the bounding-box-normalised coordinates, point index)
    lo, hi = xyz.min(1, keepdims=True), xyz.max(1, keepdims=True)
    ext = (hi - lo).astype(np.float32)
    u = np.where(ext > 0, (xyz - lo).astype(np.float32) / np.where(ext > 0, ext, 1).astype(np.float32), 0).astype(np.float32)
    qd = np.minimum(np.maximum(u * np.float32(1024), 0), 1023).astype(np.int64)
    code = np.zeros((B, N), np.int64)
    for bit in range(10):
        for d in range(3):
            code |= ((qd[..., d] >> bit) & 1) << (3 * bit + d)
    key = (code << 32) | np.arange(N)[None]
    assert np.array_equal(perm.numpy(), np.argsort(key, 1, kind="stable"))
    a0, b0 = nat.knn_pair(feat, sq, x4, k, xt=ft)
    a1, b1 = nat.knn_pair(feat, sq, x4, k, xt=ft, order=order)
    for plain, ordered in ((a0, a1), (b0, b1)):
        assert torch.equal(torch.sort(plain, -1).values, torch.sort(ordered, -1).values)


def _smooth_and_unrelated(B, N, seed, mix=None):
    rs = np.random.RandomState(seed)
    xyz = rs.rand(B, N, 3).astype(np.float32) - 0.5
    w1, w2 = rs.randn(3, 64).astype(np.float32) * 0.8, rs.randn(64, 64).astype(np.float32) * 0.2
    feat = np.maximum(np.maximum(xyz @ w1 + 0.1, 0) @ w2 + 0.05, 0)
    rnd = np.maximum(rs.randn(B, N, 64).astype(np.float32), 0)
    if mix is not None:                                    # clouds listed in `mix` get features unrelated to their coordinates
        for b in mix:
            feat[b] = rnd[b]
    feat = dev(torch.from_numpy(np.ascontiguousarray(feat)))
    x4 = dev(torch.from_numpy(np.concatenate((xyz, (xyz ** 2).sum(-1, keepdims=True)), -1).astype(np.float32)))
    sq = (feat ** 2).sum(-1).contiguous()
    ft = feat.view(B, N, 4, 4, 4).transpose(3, 4).reshape(B, N, 64).contiguous()
    return feat, sq, x4, ft


@pytest.mark.parametrize("N,k", [(2048, 20), (4096, 40)])
def test_ordered_knn_guard_fires_on_unrelated_features_and_changes_no_bit(nat, N, k):
    """vcr_knn_order_args.ord_ok: the ranking judges, cloud by cloud, whether 16 Morton neighbours are compact in FEATURE space
    (mean squared tile radius against the spread of the tile centroids).  Smooth features (a function of the coordinates, as the
    stem's are) pass; features unrelated to the coordinates fail and those clouds take the plain scan inside the same launch
    -- a batch may mix both.  Whatever the verdict, every row's neighbour set is the plain search's."""
    B = 16
    bad = [1, 2, 7, 15]
    feat, sq, x4, ft = _smooth_and_unrelated(B, N, 5 * N + k, mix=bad)
    order = nat.knn_order(x4, ft, sq, guard=True)
    ok, stat = order["ord_ok"].cpu().numpy(), order["ord_stat"].cpu().numpy()
    print(f"N={N}: ord_stat smooth {stat[[b for b in range(B) if b not in bad]].max():.3g} (max), unrelated {stat[bad].min():.3g} (min)")
    assert [b for b in range(B) if not ok[b]] == bad, (ok, stat)
    assert stat[bad].min() > 10 * stat[[b for b in range(B) if b not in bad]].max()       # the two populations are far apart
    a0, b0 = nat.knn_pair(feat, sq, x4, k, xt=ft)
    a1, b1 = nat.knn_pair(feat, sq, x4, k, xt=ft, order=order)                             # guarded
    ung = dict(order); ung.pop("ord_ok")
    a2, b2 = nat.knn_pair(feat, sq, x4, k, xt=ft, order=ung)                               # every cloud ordered
    for plain, other in ((a0, a1), (b0, b1), (a0, a2), (b0, b2)):
        assert torch.equal(torch.sort(plain, -1).values, torch.sort(other, -1).values)
    # a forced verdict (all clouds refused / all accepted) is honoured and changes nothing either
    for v in (0, 1):
        forced = dict(order); forced["ord_ok"] = torch.full_like(order["ord_ok"], v)
        a3, b3 = nat.knn_pair(feat, sq, x4, k, xt=ft, order=forced)
        assert torch.equal(torch.sort(a0, -1).values, torch.sort(a3, -1).values) and torch.equal(torch.sort(b0, -1).values, torch.sort(b3, -1).values)


@pytest.mark.parametrize("B,N,k,ordered", [(16, 4096, 40, True), (16, 4096, 40, False), (24, 3000, 20, False), (32, 2048, 20, True),
                                           (64, 512, 40, False), (2, 11000, 40, False)])
def test_knn_in_launch_tie_replay_through_global_slots(nat, B, N, k, ordered):
    """vcr_knn_args.tie_inline 2: the k > 20 launches (no room for a row image in LDS beside their lists) replay their tied rows
    in per-workgroup slots of tie_work (vcr_knn_tie_slot_bytes) instead of leaving them to a replay launch; k <= 20 ignores the
    slots (LDS image, or the replay launch beyond ~2400 points).  Lattice clouds and few-valued features tie on thousands of rows: every index must equal the replay
    launch's (the order inside a row included -- the replay writes libstdc++'s own order), with and without the ordered search,
    and also on rows so long that the replay launch itself needs global scratch."""
    import ctypes as C
    rs = np.random.RandomState(N + k)
    side = int(np.ceil(N ** (1 / 3))) + 1
    pts = np.stack([rs.permutation(side ** 3)[:N] for _ in range(B)])
    xyz = (np.stack([pts // (side * side), pts // side % side, pts % side], -1).astype(np.float32)) / side
    feat = dev(torch.from_numpy(rs.randint(0, 3, (B, N, 64)).astype(np.float32)))
    x4 = dev(torch.from_numpy(np.concatenate((xyz, (xyz ** 2).sum(-1, keepdims=True)), -1).astype(np.float32)))
    sq = (feat ** 2).sum(-1).contiguous()
    ft = feat.view(B, N, 4, 4, 4).transpose(3, 4).reshape(B, N, 64).contiguous()
    order = nat.knn_order(x4, ft, sq) if ordered else None
    a0, b0 = nat.knn_pair(feat, sq, x4, k, xt=ft, order=order)                      # replay launch
    a1, b1 = nat.knn_pair(feat, sq, x4, k, xt=ft, order=order, tie_slots=True)      # replayed inside the launch
    assert torch.equal(a0, a1) and torch.equal(b0, b1)
    lazy = nat.knn(x4, None, k, exact_ties=False)
    assert not torch.equal(torch.sort(lazy, -1).values, torch.sort(b0, -1).values)       # the replay did matter on this input
    # the single-search entry point, and what vcr_knn_ties_inline() answers
    c1 = nat.knn(x4, None, k, tie_slots=True)
    assert torch.equal(c1, nat.knn(x4, None, k))
    L = nat.lib()
    L.vcr_knn_ties_inline.argtypes, L.vcr_knn_ties_inline.restype = [C.POINTER(nat.KnnArgs)], C.c_int
    L.vcr_knn_tie_slot_bytes.argtypes, L.vcr_knn_tie_slot_bytes.restype = [C.c_int, C.c_int], C.c_size_t
    a = nat.KnnArgs(0x1000, 4, None, B, N, 4, k, 0x2000, 0x3000, B * N)
    lds_fits = N <= 2300 and k <= 20
    assert L.vcr_knn_ties_inline(C.byref(a)) == (1 if lds_fits and B * ((N + 15) // 16) >= 1024 else 0)
    a.tie_work, a.tie_work_bytes = 0x4000, L.vcr_knn_tie_slot_bytes(B, N)
    assert L.vcr_knn_ties_inline(C.byref(a)) == (1 if (k > 20 or (lds_fits and B * ((N + 15) // 16) >= 1024)) else 0)


def test_knn_deferred_tie_replay_for_two_launches(nat):
    """vcr_knn_args.tie_defer + vcr_knn_ties_f32: the Cartesian and the feature-space launch list their tied rows, one
    replay launch serves both -- the same indices as two self-contained calls, on inputs built to tie massively."""
    rs = np.random.RandomState(77)
    B, N, k = 2, 1024, 20
    side = int(np.ceil(N ** (1 / 3))) + 1
    pts = np.stack([rs.permutation(side ** 3)[:N] for _ in range(B)])
    xyz = np.stack([pts // (side * side), pts // side % side, pts % side], -1).astype(np.float32)      # [B,N,3]
    x4 = dev(torch.from_numpy(np.concatenate((xyz, (xyz ** 2).sum(-1, keepdims=True)), -1)))
    feat = dev(torch.from_numpy(rs.randint(0, 3, (B, N, 64)).astype(np.float32)))
    sq = (feat ** 2).sum(-1)
    ref_a, ref_b = nat.knn(x4, None, k), nat.knn(feat, sq, k)
    a, b = nat.knn_pair_deferred(x4, None, feat, sq, k)
    assert torch.equal(a, ref_a) and torch.equal(b, ref_b)
    lazy = nat.knn(feat, sq, k, exact_ties=False)
    assert not torch.equal(lazy, ref_b)                    # the replay did matter on this input


@pytest.mark.parametrize("N,k", [(1024, 20), (2048, 20), (512, 40), (4096, 40), (333, 5), (1344, 20), (1343, 20), (4096, 62),
                                 (1000, 50), (4031, 62), (4032, 62)])       # (k + 1) * 64 <= N switches Tensor.topk's algorithm
def test_knn_exact_ties_follow_torch_topk(nat, N, k):
    """Exact distance ties at the k-th neighbour: Tensor.topk on the CPU is libstdc++'s nth_element (or partial_sort
    when (k+1)*64 <= N) with a value-only comparator; the kernels detect such rows and replay that algorithm, so the
    neighbour SETS equal the reference's on EVERY row -- here on inputs built to tie massively (integer-grid points
    and small-integer features, whose distances are exact in fp32)."""
    rs = np.random.RandomState(N + k)
    B = 2
    # Cartesian: distinct points of a coarse integer grid -> thousands of equal distances per row
    side = int(np.ceil(N ** (1 / 3))) + 1
    pts = np.stack([rs.permutation(side ** 3)[:N] for _ in range(B)])
    xyz = np.stack([pts // (side * side), (pts // side) % side, pts % side], 1).astype(np.float32)   # [B,3,N]
    src = torch.from_numpy(xyz)
    xyz4 = torch.cat((src.transpose(1, 2), (src ** 2).sum(1).unsqueeze(-1)), -1)
    got = np.sort(nat.knn(dev(xyz4), None, k).cpu().numpy(), -1)
    ref = np.sort(oracle.knn_indices(src, k).numpy(), -1)
    assert (got == ref).all(), f"{int((got != ref).any(-1).sum())} rows differ (xyz)"
    plain = np.sort(nat.knn(dev(xyz4), None, k, exact_ties=False).cpu().numpy(), -1)
    assert (plain != ref).any(), "the input was meant to contain boundary ties"
    # feature space: small non-negative integers in 64 channels (norms and dot products exact in fp32)
    f = torch.from_numpy(rs.randint(0, 3, size=(B, 64, N)).astype(np.float32))
    got = np.sort(nat.knn(dev(f.transpose(1, 2)), dev((f ** 2).sum(1)), k).cpu().numpy(), -1)
    ref = np.sort(oracle.knn_indices(f, k).numpy(), -1)
    # (duplicate feature vectors share the best value: which copy Tensor.topk returns first, and util.py:159 drops, is replayed too)
    assert (got == ref).all(), f"{int((got != ref).any(-1).sum())} rows differ (features)"


def test_knn_long_rows_replay_ties_through_global_scratch(nat):
    """N = 12 000 (> 10 091: a row's distances no longer fit the replay's LDS image): with tie_work the replay runs out
    of global scratch and the neighbour sets still equal Tensor.topk's on every row of a tie-heavy cloud; WITHOUT
    tie_work the call is refused (VCR_EUNSUPPORTED) instead of silently skipping the replay.  k > 62 is refused too."""
    N, k = 12000, 20
    rs = np.random.RandomState(5)
    side = int(np.ceil(N ** (1 / 3))) + 1
    pts = rs.permutation(side ** 3)[:N][None]
    xyz = np.stack([pts // (side * side), (pts // side) % side, pts % side], 1).astype(np.float32)   # [1,3,N]
    src = torch.from_numpy(xyz)
    xyz4 = dev(torch.cat((src.transpose(1, 2), (src ** 2).sum(1).unsqueeze(-1)), -1))
    got = np.sort(nat.knn(xyz4, None, k).cpu().numpy(), -1)
    ref = np.sort(oracle.knn_indices(src, k).numpy(), -1)
    assert (got == ref).all(), f"{int((got != ref).any(-1).sum())} rows differ"
    plain = np.sort(nat.knn(xyz4, None, k, exact_ties=False).cpu().numpy(), -1)
    assert (plain != ref).any()
    with pytest.raises(nat.VcrHipError, match="unsupported"):
        nat.knn(xyz4, None, k, tie_work=False)
    with pytest.raises(nat.VcrHipError, match="unsupported"):
        nat.knn(xyz4, None, 63)


def test_linear_pair_equals_two_launches(nat):
    """vcr_linear_pair_f32: two independent linears of one kernel configuration in ONE launch (and the fallback to two
    launches when the configurations differ) -- bit-identical to the separate calls, statistics included."""
    import ctypes as C
    g = torch.Generator().manual_seed(5)
    M, K = 3000, 512
    L = nat.lib()
    L.vcr_linear_pair_f32.argtypes = [C.POINTER(nat.LinearArgs), C.POINTER(nat.LinearArgs), C.c_void_p]
    L.vcr_linear_pair_f32.restype = C.c_int
    for (Na, Nb, res_b) in ((512, 512, True), (1024, 512, False), (512, 256, True)):
        xa, xb = dev(torch.randn(M, K, generator=g)), dev(torch.randn(M + 77, K, generator=g))
        wa, wb = dev(torch.randn(Na, K, generator=g) / 22), dev(torch.randn(Nb, K, generator=g) / 22)
        ba, bb = dev(torch.randn(Na, generator=g)), dev(torch.randn(Nb, generator=g))
        ra = dev(torch.randn(M, Na, generator=g))
        rb = dev(torch.randn(M + 77, Nb, generator=g)) if res_b else None
        ya, sa = nat.linear(xa, wa, ba, residual=ra, want_stats=True)
        yb, sb = nat.linear(xb, wb, bb, residual=rb, want_stats=True)
        ya2, yb2 = torch.empty_like(ya), torch.empty_like(yb)
        sa2, sb2 = torch.empty_like(sa), torch.empty_like(sb)
        A = nat.LinearArgs(nat.ptr(xa), K, nat.ptr(wa), nat.ptr(ba), nat.ptr(ra), Na, nat.ptr(ya2), Na, M, Na, K, 0)
        Bq = nat.LinearArgs(nat.ptr(xb), K, nat.ptr(wb), nat.ptr(bb), nat.ptr(rb), Nb if res_b else 0, nat.ptr(yb2), Nb, M + 77, Nb, K, 0)
        A.stats_out, Bq.stats_out = nat.ptr(sa2), nat.ptr(sb2)
        nat.check(L.vcr_linear_pair_f32(C.byref(A), C.byref(Bq), C.c_void_p(nat.stream_ptr())), "vcr_linear_pair_f32")
        assert torch.equal(ya, ya2) and torch.equal(yb, yb2) and torch.equal(sa, sa2) and torch.equal(sb, sb2), (Na, Nb, res_b)
        # tile rows forced to 96 / 128 on both halves (the automatic choice above is made for the combined grid): the
        # 16x16x4 kernels give the same bits at either height
        ya16, sa16 = nat.linear(xa, wa, ba, residual=ra, want_stats=True, variant=16)
        yb16, sb16 = nat.linear(xb, wb, bb, residual=rb, want_stats=True, variant=16)
        for v in (2048, 4096 | 16, 8192, 16384):
            A.variant = Bq.variant = v
            for t_ in (ya2, yb2, sa2, sb2):
                t_.fill_(float("nan"))
            nat.check(L.vcr_linear_pair_f32(C.byref(A), C.byref(Bq), C.c_void_p(nat.stream_ptr())), "vcr_linear_pair_f32")
            assert torch.equal(ya16, ya2) and torch.equal(yb16, yb2) and torch.equal(sa16, sa2) and torch.equal(sb16, sb2), (Na, Nb, v)


def test_sdpa_indexed_keys_equal_the_gathered_rows(nat):
    """vcr_sdpa_args.key_index: attention over the rows key_index[kb][0..nk-1] of each key batch (the decoder's kept keys of
    partial-overlap mode), read in place -- bit-identical to attention over a dense copy of those rows in that order."""
    g = torch.Generator().manual_seed(21)
    nb, h, NQ, NS, NK = 4, 4, 200, 260, 150
    q = dev(torch.randn(nb * NQ, h * 128, generator=g))
    kv = dev(torch.randn(nb * NS, 2 * h * 128, generator=g))
    idx = torch.stack([torch.randperm(NS, generator=g)[:NK] for _ in range(nb)]).to(torch.int32)
    dense = kv.view(nb, NS, -1)[torch.arange(nb).view(-1, 1), idx.long()].reshape(nb * NK, -1).contiguous()
    ref = nat.sdpa(q, dense[:, :512], dense[:, 512:], nb, h, NQ, NK, 1 / math.sqrt(128), kv_batch_shift=1)
    out = nat.sdpa(q, kv[:, :512], kv[:, 512:], nb, h, NQ, NK, 1 / math.sqrt(128), kv_batch_shift=1, key_index=dev(idx), nk_src=NS)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("N,k", [(70001, 20), (70001, 40), (131072, 20)])
def test_knn_beyond_65535_points(nat, W, N, k):
    """util.py:143-160 has no size limit; the kNN entry points take clouds of more than 65 535 points, up to the library's
    limit of 131 072 (the reference would materialise a 19.6 / 68.7 GB distance matrix).  Neighbour sets of 192 sampled queries
    -- Cartesian and feature-space -- against the reference formula evaluated for those rows on the CPU; ties replayed
    through global scratch.  One point more is refused (VCR_EUNSUPPORTED), not computed unvalidated."""
    rs = np.random.RandomState(k)
    pts = torch.from_numpy(rs.uniform(-0.5, 0.5, size=(1, 3, N)).astype(np.float32))
    h = F.relu(F.conv1d(pts, W["emb_nn.conv1_lpd.weight"], W["emb_nn.conv1_lpd.bias"]))
    h = F.relu(F.conv1d(h, W["emb_nn.conv2_lpd.weight"], W["emb_nn.conv2_lpd.bias"]))          # [1, 64, N]
    rows = torch.from_numpy(rs.choice(N, 192, replace=False))

    def ref_sets(x):                                       # x [1, C, N]: rows of -xx - inner - xx^T (util.py:153-158)
        xt = x[0].t().contiguous()                         # [N, C]
        inner = -2 * torch.matmul(xt[rows], x[0])          # [192, N]
        xx = torch.sum(x[0] ** 2, dim=0, keepdim=True)     # [1, N]
        d = -xx - inner - xx[0, rows].unsqueeze(1)
        return np.sort(d.topk(k + 1, dim=-1)[1][:, 1:].numpy(), -1), d

    xyz4 = torch.cat((pts.transpose(1, 2), (pts ** 2).sum(1).unsqueeze(-1)), -1)
    got3 = np.sort(nat.knn(dev(xyz4), None, k).cpu().numpy()[0][rows.numpy()], -1)
    ref3, d3 = ref_sets(pts)
    bad3 = int((got3 != ref3).any(-1).sum())
    got64 = np.sort(nat.knn(dev(h.transpose(1, 2)), dev((h ** 2).sum(1)), k).cpu().numpy()[0][rows.numpy()], -1)
    ref64, d64 = ref_sets(h)
    bad64 = int((got64 != ref64).any(-1).sum())
    print(f"N = {N}, k = {k}: rows differing xyz {bad3} / 192, features {bad64} / 192")
    # (the row-subset matmul of this reference may round a distance differently from the full N x N one: allow a near-tie or two)
    assert bad3 <= 2 and bad64 <= 2
    if N == 131072:
        big = torch.zeros(1, N + 1, 4, device="cuda")
        with pytest.raises(nat.VcrHipError, match="unsupported"):
            nat.knn(big, None, k)
