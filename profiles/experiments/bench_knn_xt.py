import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vcrnet_amd  # noqa
from vcrnet_amd import native
if len(sys.argv) > 1: native.LIB_PATH = sys.argv[1]
g = torch.Generator().manual_seed(0)
def bench(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for B, N, k in ((32, 1024, 20), (32, 2048, 20), (48, 768, 20)):
    f = torch.randn(B, N, 64, generator=g).cuda(); sq = (f ** 2).sum(-1).contiguous()
    ft = f.view(B, N, 4, 4, 4).transpose(3, 4).reshape(B, N, 64).contiguous()
    xyz = torch.rand(B, N, 3, generator=g) - 0.5
    x4 = torch.cat((xyz, (xyz ** 2).sum(-1, keepdim=True)), -1).cuda().contiguous()
    t0 = bench(lambda: native.knn(f, sq, k, exact_ties=False))
    t1 = bench(lambda: native.knn(f, sq, k, exact_ties=False, xt=ft))
    p0 = bench(lambda: native.knn_pair(f, sq, x4, k))
    p1 = bench(lambda: native.knn_pair(f, sq, x4, k, xt=ft))
    same = torch.equal(native.knn(f, sq, k), native.knn(f, sq, k, xt=ft))
    print(os.path.basename(native.LIB_PATH), f"B={B} N={N}: knn64 {t0:.1f} -> {t1:.1f} us (xt), pair {p0:.1f} -> {p1:.1f} us, same={same}")
