import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vcrnet_amd  # noqa
from vcrnet_amd import native
native.LIB_PATH = sys.argv[1]
g = torch.Generator().manual_seed(0)
def bench(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for B, N, k in ((32, 1024, 20), (32, 2048, 20)):
    f = torch.randn(B, N, 64, generator=g).cuda(); sq = (f ** 2).sum(-1).contiguous()
    xyz = torch.rand(B, N, 3, generator=g) - 0.5
    x4 = torch.cat((xyz, (xyz ** 2).sum(-1, keepdim=True)), -1).cuda().contiguous()
    t64 = bench(lambda: native.knn(f, sq, k, exact_ties=False))
    tp = bench(lambda: native.knn_pair(f, sq, x4, k))
    ref = native.knn(f, sq, k)
    print(f"{os.path.basename(sys.argv[1])} B={B} N={N}: knn64 {t64:.1f} us, pair+ties {tp:.1f} us")
