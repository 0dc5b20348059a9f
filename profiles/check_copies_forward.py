"""Clouds that carry copies of points through the WHOLE forward, against the CPU oracle: python profiles/check_copies_forward.py.
Every copy is a kNN row with a shared best value (replayed: DESIGN section 2), identical embeddings give exact ties in every
later ranking.  Whole mode: poses at the BASELINE tolerance, embeddings to fp32 rounding; partial mode: the fused driver against
the kernel-by-kernel composition and the oracle's shapes (selections among identical points are exchangeable, not comparable)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import vcrnet_amd  # noqa
from vcrnet_amd import synth, composed
from test_hip_forward import build_net
import oracle
def with_copies(x, frac, rs):
    for b in range(x.shape[0]):
        N = x.shape[2]
        p, n2 = rs.permutation(N), int(N * frac / 2)
        x[b][:, p[:n2]] = x[b][:, p[n2:2 * n2]]
    return x
bad = 0
for (B, N, k, frac, regime) in ((4, 1024, 20, 0.1, "default"), (3, 747, 20, 0.5, "trained"), (2, 2048, 20, 0.1, "default"), (2, 4096, 40, 0.05, "seed4321"),
                                (4, 500, 40, 0.2, "randemb"), (6, 300, 7, 0.3, "default")):
    rs = np.random.RandomState(B * N + k)
    net, w = build_net(regime=regime); net.emb_nn.k = k
    src, tgt, _, _, _ = synth.make_batch(5, B, N, kind="object" if N <= 2048 else "uniform")
    src, tgt = with_copies(src, frac, rs), with_copies(tgt, frac, rs)
    s, t = torch.from_numpy(src), torch.from_numpy(tgt)
    rec = {}
    ref = oracle.vcrnet_forward(w, s, t, oracle.OracleConfig(k=k, record=rec))
    with torch.no_grad():
        out = net._forward_fused(s.cuda(), t.cuda(), want_emb=True)
    dR, dt = float((out[2].cpu() - ref[2]).abs().max()), float((out[3].cpu() - ref[3]).abs().max())
    ok = dR <= 1e-4 and dt <= 1e-5
    bad += not ok
    print(f"whole   {regime:8s} B={B} N={N:4d} k={k:2d} copies {frac:4.0%}: dR {dR:.2e} dt {dt:.2e}{'' if ok else '   <<<< FAIL'}", flush=True)
for (B, Nfull, frac, regime) in ((3, 1024, 0.2, "default"), (2, 1333, 0.1, "trained"), (4, 500, 0.5, "seed4321")):
    rs = np.random.RandomState(B * Nfull)
    net, w = build_net(regime=regime, partial=True, overlap2=synth.OVERLAP2_0575)
    src, tgt, _, _, _ = synth.make_batch(9, B, Nfull, partial=True)
    src, tgt = with_copies(src, frac, rs), with_copies(tgt, frac, rs)
    s, t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    with torch.no_grad():
        f = net(s, t); c = composed.forward_composed(net, s, t)
        it = net.forward_iter(s, t, 3)
    ref = oracle.vcrnet_forward(w, s.cpu(), t.cpu(), oracle.OracleConfig(partial=True, overlap2=synth.OVERLAP2_0575))
    det = torch.det(it[2]).cpu()
    dfc = float((f[2] - c[2]).abs().max())
    ok = ref[0].shape == f[0].shape and not torch.isnan(it[2]).any().item() and float((det - 1).abs().max()) < 1e-4 and dfc < 1e-3
    bad += not ok
    print(f"partial {regime:8s} B={B} N={s.shape[2]:4d} copies {frac:4.0%}: shapes {tuple(f[0].shape)} == oracle's {ref[0].shape == f[0].shape}, det {det.numpy().round(5)}, "
          f"fused vs kernel-by-kernel dR {dfc:.1e}, dR vs oracle {float((f[2].cpu() - ref[2]).abs().max()):.1e}{'' if ok else '   <<<< FAIL'}", flush=True)
print("failures:", bad)
