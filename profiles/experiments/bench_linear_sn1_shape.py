import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vcrnet_amd  # noqa
from vcrnet_amd import native
g = torch.Generator().manual_seed(0)
def bench(fn, reps=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for M, N, K in ((32768, 512, 128), (36864, 512, 128)):
    x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / 8).cuda(); b = torch.randn(N, generator=g).cuda()
    y = torch.empty(M, N, device="cuda")
    for name, v in (("auto", 0), ("bm128", 4096), ("bm96", 2048), ("bm64", 8192), ("bm32", 16384), ("bk32", 8), ("bk32+bm96", 8 + 2048), ("bk32+bm64", 8 + 8192), ("ms16", 256)):
        try:
            t = bench(lambda: native.linear(x, w, b, out=y, variant=v))
            print(f"M={M} N={N} K={K} {name:10s} {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF/s")
        except Exception as e:
            print(name, "ERR", str(e)[:80])
