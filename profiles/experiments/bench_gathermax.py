import os, sys, subprocess
if len(sys.argv) < 2:
    for mode in ("l2", "lds"):
        env = dict(os.environ)
        env["VCR_GATHERMAX"] = mode
        print(mode, subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True).stdout)
    sys.exit(0)
import torch
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
for B, N, k, C in ((32, 1024, 20, 256), (48, 768, 20, 256), (32, 2048, 20, 256), (16, 3000, 40, 256), (4, 100, 20, 64), (2, 1024, 20, 128)):
    pq = torch.randn(B * N, 2 * C, generator=g).cuda()
    idx = torch.randint(0, N, (B * N, k), generator=g, dtype=torch.int32).cuda()
    t = bench(lambda: native.gathermax(pq, C, idx, N))
    y = native.gathermax(pq, C, idx, N)
    P = pq[:, :C].view(B, N, C); Q = pq[:, C:].view(B, N, C)
    ref = torch.relu(torch.stack([P[b][idx.view(B, N, k)[b].long()].max(1).values for b in range(B)]) + Q).view(B * N, C)
    print(f"  B={B} N={N} k={k} C={C}: {t:.1f} us  equal={torch.equal(y, ref)}")
