"""Batch-composition soak: python profiles/fuzz_batch_split.py <seed> <trials>.  Every kernel of the path computes a row from that
row's cloud alone, so a pair's results must not depend on WHICH batch it travels in -- as long as the one size-dependent choice
that legitimately changes bits, the linears' MFMA shape, is pinned (vcr_vcrnet_weights.linear_mfma / linear_bk / linear_bm) and no
attention launch splits its keys (batches of more than half a round of workgroups).  Each trial draws an embedding / pointer /
head, an arithmetic mode, a weight regime, k, N and an even batch, runs the batch whole and as its two halves, and compares the final
embeddings bit for bit (the head's outputs are reported: its own key split depends on the grid).  What differs between the two runs is exactly what the library chooses from
the grid size: the kNN kernels' forms (candidate splits, 16- / 32-query waves, pair / small-grid / separate launches, in-launch or
separate tie replay), tile vs persistent kernels, XCD renumbering.  (This is the check that would have caught the kNN's rank-0
rule of rounds 2-5: two candidate splits kept different copies of a near-duplicate point.)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import vcrnet_amd  # noqa
from vcrnet_amd import synth
from test_hip_forward import build_net
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad, t0 = 0, time.time()
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    emb = str(rs.choice(["lpdnet", "lpdnet", "lpdnet", "dgcnn", "pointnet"]))
    pointer = str(rs.choice(["transformer", "transformer", "identity"]))
    vcp = str(rs.choice(["topK", "att", "dist"]))
    mode = str(rs.choice(["fp32", "fp32", "bf16x3", "bf16x3+sdpa"]))
    merge = bool(rs.rand() < 0.7)
    regime = str(rs.choice(["default", "seed4321", "trained", "randemb"])) if emb == "lpdnet" else "default"
    k = int(rs.choice([20, 20, 40, 7])) if emb != "pointnet" else 20
    N = int(rs.choice([1024, 1024, 2048])) if rs.rand() < 0.3 else int(rs.randint(max(k + 2, 200), 1100))
    qb = (N + 127) // 128
    half_min = 32 // qb + 1                               # cross-attention of a half: qb x 2 B' x 4 heads x 2 > 512 workgroup slots
    half = int(rs.randint(half_min, half_min + 5))
    B = 2 * half
    dups = bool(rs.rand() < 0.5)
    kw = dict(emb_nn=emb, pointer=pointer, vcp_nn=vcp)
    src, tgt, _, _, _ = synth.make_batch(int(rs.randint(0, 1000)), B, N, kind="object" if N < 2048 else "uniform")
    if dups:                                               # copies of points in both clouds (shared best values in the kNN)
        for x in (src, tgt):
            for b in range(B):
                p = rs.permutation(N)
                n2 = N // 16
                x[b][:, p[:n2]] = x[b][:, p[n2:2 * n2]]
    s, t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    net, _ = build_net(regime=regime, **kw) if emb == "lpdnet" else build_net(**kw)
    net.linear_mode, net.merge_encdec = mode, merge
    net.linear_mfma, net.linear_bk, net.linear_bm = int(rs.choice([16, 32])), int(rs.choice([16, 32])), 128
    net.emb_nn.k = k
    def run(a, b_):
        with torch.no_grad():
            out = net._forward_fused(a, b_, want_emb=True)
        torch.cuda.synchronize()
        return [o.clone() for o in out if torch.is_tensor(o)]
    whole = run(s, t)
    h1, h2 = run(s[:half].contiguous(), t[:half].contiguous()), run(s[half:].contiguous(), t[half:].contiguous())
    parts = []
    for i, (a, b_) in enumerate(zip(h1, h2)):
        if i == len(h1) - 1:                               # embeddings [2 B' N, E]: source rows of both halves, then target rows
            E = a.shape[1]
            a, b_ = a.view(2, half, N, E), b_.view(2, half, N, E)
            parts.append(torch.cat((a, b_), 1).reshape(2 * B * N, E))
        else:
            parts.append(torch.cat((a, b_), 0))
    # judged on the final embeddings (everything in front of the head).  The soft-correspondence head splits its streamed tiles
    # over more workgroups when the grid is small (pairscore.hip: a cost model of the rounds) and merges (max, sum) partials:
    # a legitimate size-dependent choice that moves the correspondences by a rounding -- reported, not counted
    same = torch.equal(whole[-1], parts[-1])
    head_same = all(torch.equal(x, y) for x, y in zip(whole[1:-1], parts[1:-1]))   # (out[0] is the source cloud itself for soft heads)
    if not same:
        e0, e1 = whole[-1], parts[-1]
        rows = (e0 != e1).any(dim=1).nonzero().flatten()
        print("   embedding rows differing:", rows.numel(), "first", rows[:8].tolist())
    bad += 0 if same else 1
    print(f"{emb:8s} {pointer:11s} {vcp:4s} {mode:12s} merged={int(merge)} {regime:8s} B={B:2d} ({half}+{half}) N={N:4d} k={k:2d} mfma={net.linear_mfma} bk={net.linear_bk} "
          f"copies={int(dups)}: embeddings {'bit-identical' if same else 'DIFFERENT  <<<<<<'}; head {'bit-identical' if head_same else 'by a rounding (split differs): max |dcorr| %.1e' % float((whole[1] - parts[1]).abs().max())}", flush=True)
    del net
    torch.cuda.empty_cache()
print("trials with a difference:", bad, "elapsed", round(time.time() - t0, 1))
