"""vcr_gathermax_f32 by form (vcr_gathermax_args.variant): gathers through L2, or out of LDS with 32- / 16- / 8-channel
slices; us per launch, results compared bit for bit with the definition in torch.  Run on the GPU box."""
import os, sys
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
for B, N, k, C in ((32, 1024, 20, 256), (48, 768, 20, 256), (32, 2048, 20, 256), (16, 3000, 40, 256), (64, 4096, 40, 256),
                   (4, 100, 20, 64), (2, 1024, 20, 128)):
    pq = torch.randn(B * N, 2 * C, generator=g).cuda()
    idx = torch.randint(0, N, (B * N, k), generator=g, dtype=torch.int32).cuda()
    P = pq[:, :C].view(B, N, C); Q = pq[:, C:].view(B, N, C)
    ref = torch.relu(torch.stack([P[b][idx.view(B, N, k)[b].long()].max(1).values for b in range(min(B, 4))]) + Q[:min(B, 4)]).view(-1, C)
    out = []
    for variant, nm in ((0, "auto"), (1, "L2"), (32, "LDS 32"), (16, "LDS 16"), (8, "LDS 8")):
        try:
            y = native.gathermax(pq, C, idx, N, variant=variant)
        except native.VcrHipError as e:
            out.append(f"{nm}: -")
            continue
        ok = torch.equal(y[:ref.shape[0]], ref)
        t = bench(lambda: native.gathermax(pq, C, idx, N, variant=variant))
        out.append(f"{nm}: {t:.1f} us{'' if ok else ' WRONG'}")
    print(f"B={B} N={N} k={k} C={C}:  " + "   ".join(out), flush=True)
