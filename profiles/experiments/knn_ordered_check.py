"""ordered (ranked + pruned) pair search against the plain pair search: neighbour sets and timings"""
import ctypes as C, sys, torch, numpy as np
sys.path.insert(0, '.')
import vcrnet_amd
from vcrnet_amd import native as nat, synth
if len(sys.argv) > 1: nat.LIB_PATH = sys.argv[1]
L = nat.lib()
L.vcr_knn_pair_f32.argtypes = [C.POINTER(nat.KnnArgs), C.POINTER(nat.KnnArgs), C.c_void_p]; L.vcr_knn_pair_f32.restype = C.c_int
L.vcr_knn_order_f32.argtypes = [C.POINTER(nat.KnnOrderArgs), C.c_void_p]; L.vcr_knn_order_f32.restype = C.c_int

def bench(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

def case(name, xyz, f, k):
    B, N, _ = xyz.shape
    T = (N + 15) // 16
    x4 = torch.cat((xyz, (xyz ** 2).sum(-1, keepdim=True)), -1).contiguous().cuda()
    f = f.contiguous().cuda(); sq = (f ** 2).sum(-1).contiguous()
    ft = f.view(B, N, 4, 4, 4).transpose(3, 4).reshape(B, N, 64).contiguous()
    e = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device="cuda")
    perm = e(B, N, dt=torch.int32); x4p = e(B, N, 4); c4 = e(B, T, 4); c4r = e(B, T); c4m = e(B, T)
    fp = e(B, N, 64); sqp = e(B, N); c64 = e(B, T, 64); c64s = e(B, T); c64r = e(B, T); c64m = e(B, T)
    oa = nat.KnnOrderArgs(nat.ptr(x4), nat.ptr(ft), 64, nat.ptr(sq), B, N, nat.ptr(perm), nat.ptr(x4p), nat.ptr(c4), nat.ptr(c4r),
                          nat.ptr(c4m), nat.ptr(fp), nat.ptr(sqp), nat.ptr(c64), nat.ptr(c64s), nat.ptr(c64r), nat.ptr(c64m))
    order = lambda: nat.check(L.vcr_knn_order_f32(C.byref(oa), C.c_void_p(nat.stream_ptr())), "order")
    order(); torch.cuda.synchronize()
    p = perm.long().cpu()
    assert all(torch.equal(torch.sort(p[b]).values, torch.arange(N)) for b in range(B)), "perm is not a permutation"
    def args(ordered, only=None):
        out = []
        for x, s_, Cc in ((f, sq, 64), (x4, None, 4)):
            ordered_here = ordered and (only is None or only == Cc)
            idx = torch.full((B, N, k), -1, dtype=torch.int32, device="cuda")
            t_ = torch.zeros(1 + B * N, dtype=torch.int32, device="cuda")
            a = nat.KnnArgs(nat.ptr(x), x.stride(1), nat.ptr(s_), B, N, Cc, k, nat.ptr(idx), nat.ptr(t_), B * N, 0)
            if Cc == 64: a.xt = nat.ptr(ft)
            if ordered_here:
                a.perm = nat.ptr(perm)
                if Cc == 64: a.xp, a.sqp, a.cen, a.cen_sq, a.cen_rad, a.cen_sqmax = map(nat.ptr, (fp, sqp, c64, c64s, c64r, c64m))
                else: a.xp, a.cen, a.cen_rad, a.cen_sqmax = map(nat.ptr, (x4p, c4, c4r, c4m))
            out.append((a, idx, t_))
        return out
    plain, ordd = args(False), args(True)
    run = lambda A: nat.check(L.vcr_knn_pair_f32(C.byref(A[0][0]), C.byref(A[1][0]), C.c_void_p(nat.stream_ptr())), "pair")
    run(plain); run(ordd); torch.cuda.synchronize()
    bad = []
    for (a0, i0, t0), (a1, i1, t1), nm in zip(plain, ordd, ("feat", "xyz")):
        s0, s1 = torch.sort(i0, -1).values, torch.sort(i1, -1).values
        bad.append(int((s0 != s1).any(-1).sum()))
    tp, to, tr = bench(lambda: run(plain)), bench(lambda: run(ordd)), bench(order)
    of, ox = args(True, 64), args(True, 4)
    tf, tx = bench(lambda: run(of)), bench(lambda: run(ox))
    print(f"{name:34s} B={B:3d} N={N:5d} k={k}: rows differing feat {bad[0]} xyz {bad[1]} | plain {tp:7.1f} us  ordered {to:7.1f} (feat only {tf:7.1f}, xyz only {tx:7.1f}) + ranking {tr:6.1f} us", flush=True)
    return bad

g = torch.Generator().manual_seed(0)
def stem_like(xyz):                                        # a smooth 3 -> 64 map (two pointwise layers, ReLU)
    w1 = torch.randn(3, 64, generator=g) * 0.8; w2 = torch.randn(64, 64, generator=g) * 0.2
    return torch.relu(torch.relu(xyz @ w1 + 0.1) @ w2 + 0.05)
tot = 0
for B, N, k, kind in ((32, 1024, 20, "object"), (48, 768, 20, "object"), (32, 2048, 20, "uniform"), (64, 4096, 40, "uniform"), (5, 333, 20, "object"), (3, 77, 20, "object"), (7, 1000, 40, "object")):
    src, _, _, _, _ = synth.make_batch(0, B, N, kind=kind)
    xyz = torch.from_numpy(src).transpose(1, 2).contiguous()
    tot += sum(case(f"{kind} cloud, smooth features", xyz, stem_like(xyz), k))
    if N <= 1024:
        tot += sum(case("same cloud, random features", xyz, torch.randn(B, N, 64, generator=g), k))
# duplicates and exact ties
xyz = torch.rand(4, 512, 3, generator=g); xyz[:, 256:] = xyz[:, :256]                       # every point twice
tot += sum(case("every point duplicated", xyz, stem_like(xyz), 20))
xyz = (torch.randint(0, 6, (4, 600, 3), generator=g).float()) / 6                              # lattice: many exact ties
tot += sum(case("lattice points (ties)", xyz, stem_like(xyz), 20))
xyz = torch.zeros(2, 300, 3)                                                                # all equal
tot += sum(case("all points equal", xyz, stem_like(xyz), 20))
print("TOTAL differing rows", tot)
