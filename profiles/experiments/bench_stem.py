import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vcrnet_amd  # noqa
from vcrnet_amd import native
g = torch.Generator().manual_seed(0)
def bench(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
w1 = torch.randn(64, 3, generator=g).cuda(); b1 = torch.randn(64, generator=g).cuda()
w2 = (torch.randn(64, 64, generator=g) / 8).cuda(); b2 = torch.randn(64, generator=g).cuda()
wpq = (torch.randn(256, 64, generator=g) / 8).cuda(); bpq = torch.randn(256, generator=g).cuda()
for B, N in ((32, 1024), (48, 768), (32, 2048), (64, 4096), (8, 1024), (48, 768), (32, 2048)):
    x = (torch.rand(B, 3, N, generator=g) - 0.5).cuda()
    ft = torch.empty(B, N, 64, device="cuda")
    t2 = bench(lambda: native.pointwise(x, w1, b1, w2, b2, wpq, bpq, feat_t=ft))
    print(f"  B={B} N={N}: +pq+t {t2:.1f} us")
