"""Random-shape soak of vcrnetIter's target reuse: python profiles/fuzz_iter_reuse.py <seed> <trials>.  Each trial draws an
embedding / pointer / head, whole or partial mode, an arithmetic mode, merged or separate first sublayers, a weight regime,
B, N, k and the number of passes, and runs the loop with the reuse on and off: every output (poses, correspondences, matched
sources, partial mode's selections of every pass) must be equal bit for bit."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import vcrnet_amd  # noqa
from vcrnet_amd import synth
from test_hip_forward import build_net
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad, t0 = 0, time.time()
only = {int(x) for x in sys.argv[3].split(",")} if len(sys.argv) > 3 else None     # re-run these trials of the seed only
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    partial = bool(rs.rand() < 0.4)
    emb = str(rs.choice(["lpdnet", "lpdnet", "lpdnet", "dgcnn", "pointnet"]))
    pointer = "transformer" if partial else str(rs.choice(["transformer", "transformer", "identity"]))
    vcp = str(rs.choice(["topK", "topK", "att", "dist"])) if not partial else "topK"
    mode = str(rs.choice(["fp32", "fp32", "bf16x3", "bf16x3+sdpa"]))
    merge = bool(rs.rand() < 0.7)
    regime = str(rs.choice(["default", "seed4321", "trained", "randemb"])) if emb == "lpdnet" else "default"
    k = int(rs.choice([20, 20, 40, 7])) if emb != "pointnet" else 20
    big = rs.rand() < 0.25
    B = int(rs.randint(8, 21)) if big else int(rs.randint(1, 8))
    N = int(rs.choice([1024, 1300, 2048])) if big else int(rs.randint(max(k + 2, 64), 900))
    iters = int(rs.choice([2, 2, 3, 4]))
    kw = dict(emb_nn=emb, pointer=pointer, vcp_nn=vcp, partial=partial)
    src, tgt, _, _, _ = synth.make_batch(int(rs.randint(0, 1000)), B, N, partial=partial, kind="object" if N < 2048 else "uniform")
    s, t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    if only is not None and trial not in only:
        continue
    outs = []
    for reuse in (False, True):
        net, _ = build_net(regime=regime, **kw) if emb == "lpdnet" else build_net(**kw)
        net.linear_mode, net.merge_encdec, net.iter_reuse = mode, merge, reuse
        net.emb_nn.k = k
        with torch.no_grad():
            out = net._forward_fused(s, t, iters=iters, iter_api=True, want_selections=partial)
        torch.cuda.synchronize()
        sel = out[-1] if partial else {}
        outs.append([o.clone() for o in out if torch.is_tensor(o)] + [sel[k_].clone() for k_ in sorted(sel)])
        del net
    same = all(torch.equal(a, b) for a, b in zip(*outs))
    if not same:
        print("   max |difference| per output:", [float((a.float() - b.float()).abs().max()) for a, b in zip(*outs)])
    bad += 0 if same else 1
    print(f"{emb:8s} {pointer:11s} {vcp:4s} {'partial' if partial else 'whole  '} {mode:12s} merged={int(merge)} {regime:8s} B={B:2d} N={N:4d} k={k:2d} "
          f"passes={iters}: {'bit-identical' if same else 'DIFFERENT  <<<<<<'}", flush=True)
    torch.cuda.empty_cache()
print("trials with a difference:", bad, "elapsed", round(time.time() - t0, 1))
