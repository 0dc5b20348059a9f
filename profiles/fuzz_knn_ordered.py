"""Random-shape soak of the ordered kNN search (vcr_knn_order_f32 + vcr_knn_args.perm) against the plain pair launch:
python profiles/fuzz_knn_ordered.py <seed> <trials>.  Shapes with >= 1024 groups of 16 queries (the regime the ordered bodies run
in), N up to 8192, k = 20 / 40; clouds: uniform / clustered / lattice (exact ties) / with duplicated points; features: a smooth map
of the coordinates, or unrelated to them.  Prints the rows whose neighbour SET differs (must be 0)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import vcrnet_amd  # noqa
from vcrnet_amd import native as nat
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad_total, t0 = 0, time.time()
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    k = int(rs.choice([20, 40])); N = int(rs.choice([1024, 1500, 2048, 2049, 2500, 3000, 4095, 4096, 4097, 5000, 6001, 8191, 8192]))
    B = int(np.ceil(1024 / ((N + 15) // 16))) + int(rs.randint(0, 3))
    kind = str(rs.choice(["uniform", "clustered", "lattice", "dup", "smallscale", "bigscale", "outlier"]))
    if kind == "lattice":
        xyz = rs.randint(0, 11, (B, N, 3)).astype(np.float32) / 10
    elif kind == "clustered":
        c = rs.rand(B, 8, 3).astype(np.float32)
        xyz = c[np.arange(B)[:, None], rs.randint(0, 8, (B, N))] + rs.randn(B, N, 3).astype(np.float32) * 0.03
    else:
        xyz = rs.rand(B, N, 3).astype(np.float32) - 0.5
    if kind == "dup":
        xyz[:, N // 2:] = xyz[:, : N - N // 2]
    if kind == "outlier":                                  # far returns: the ranking's box is clamped to mean +- 4 sigma
        xyz[:, rs.randint(0, N, 3)] = np.float32(rs.choice([50.0, -200.0, 1e4]))
    scale = {"smallscale": 1e-3, "bigscale": 300.0}.get(kind, 1.0)
    xyz = (xyz * scale).astype(np.float32)
    w1, w2 = rs.randn(3, 64).astype(np.float32) / scale, rs.randn(64, 64).astype(np.float32) * 0.2
    feat = np.maximum(np.maximum(xyz @ w1 + 0.1, 0) @ w2 + 0.05, 0) * np.float32(rs.choice([1.0, 1e-2, 50.0]))
    if rs.rand() < 0.25:
        feat = rs.randn(B, N, 64).astype(np.float32)
    feat = torch.from_numpy(np.ascontiguousarray(feat.astype(np.float32))).cuda()
    x4 = torch.from_numpy(np.concatenate((xyz, (xyz ** 2).sum(-1, keepdims=True)), -1).astype(np.float32)).cuda()
    sq = (feat ** 2).sum(-1).contiguous()
    ft = feat.view(B, N, 4, 4, 4).transpose(3, 4).reshape(B, N, 64).contiguous()
    guard, slots = bool(rs.rand() < 0.7), bool(rs.rand() < 0.5)     # round 6: the per-cloud guard; ties replayed in the launch (k > 20)
    order = nat.knn_order(x4, ft, sq, guard=guard)
    a0, b0 = nat.knn_pair(feat, sq, x4, k, xt=ft)
    a1, b1 = nat.knn_pair(feat, sq, x4, k, xt=ft, order=order, tie_slots=slots)
    bad = [int((torch.sort(p, -1).values != torch.sort(o, -1).values).any(-1).sum()) for p, o in ((a0, a1), (b0, b1))]
    bad_total += sum(bad)
    acc = int(order["ord_ok"].sum()) if guard else -1
    print(f"B={B:3d} N={N:5d} k={k} {kind:10s} guard {'off' if not guard else f'{acc}/{B} accepted'} slots {int(slots)}: rows differing feat {bad[0]} xyz {bad[1]}", flush=True)
print("TOTAL differing rows", bad_total, "elapsed", round(time.time() - t0, 1))
