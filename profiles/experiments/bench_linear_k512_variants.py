import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vcrnet_amd  # noqa
from vcrnet_amd import native
g = torch.Generator().manual_seed(0)
def bench(fn, reps=60):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
M, N, K = 32768, 512, 512
x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / 16).cuda(); b = torch.randn(N, generator=g).cuda()
res = torch.randn(M, N, generator=g).cuda(); y = torch.empty(M, N, device="cuda")
for rep in range(2):
    for name, v in (("auto", 0), ("bm64", 8192), ("bm32", 16384), ("bm96", 2048), ("bk16", 64), ("bk32", 8), ("bk32+ms32", 8 + 1024), ("bk16+ms16", 64 + 16)):
        for shape, kw in (("conv3-like (relu, stats)", dict(relu=True, want_stats=True)), ("wo-like (residual, stats)", dict(residual=res, want_stats=True))):
            try:
                t = bench(lambda: native.linear(x, w, b, out=y, variant=v, **kw))
                print(f"{shape:28s} {name:10s} {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF/s", flush=True)
            except Exception as e:
                print(shape, name, "ERR", str(e)[:60])
