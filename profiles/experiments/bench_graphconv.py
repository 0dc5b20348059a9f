"""Record of the convSN1-as-one-launch experiment (graphconv_lds_kernel.hip in this directory): needs that kernel dropped
into edgeconv.hip with its vcr_graphconv_f32 entry point and a native.graphconv() ctypes wrapper; not runnable against the
shipped library."""
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
for B, N, k, C in ((32, 1024, 20, 256), (48, 768, 20, 256), (4, 1000, 40, 256), (3, 77, 20, 64)):
    M = B * N
    x = torch.randn(M, 128, generator=g).cuda().abs()
    wpq = (torch.randn(2 * C, 128, generator=g) / 8).cuda(); bpq = torch.randn(2 * C, generator=g).cuda()
    idx = torch.randint(0, N, (M, k), generator=g, dtype=torch.int32).cuda()
    pq = native.linear(x, wpq, bpq)
    ref = native.gathermax(pq, C, idx, N)
    y = native.graphconv(x, wpq, bpq, idx, N)
    d = (y - ref).abs().max().item()
    t0 = bench(lambda: native.gathermax(native.linear(x, wpq, bpq, out=pq), C, idx, N))
    t1 = bench(lambda: native.graphconv(x, wpq, bpq, idx, N, out=y))
    print(f"B={B} N={N} k={k} C={C}: equal={torch.equal(y, ref)} max|d|={d:.2e}  linear+gathermax {t0:.1f} us, graphconv {t1:.1f} us", flush=True)
