#!/usr/bin/env python
"""The persistent linear kernel (linear_stream.hip, variant bit 16) against the one-tile-per-workgroup kernels (bit 15):
GEMM part bit-identical to the 32x32x2 kernels, statistics-out partials to fp32 rounding; us per launch of both.
Run on the GPU box."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vcrnet_amd  # noqa
from vcrnet_amd import native
g = torch.Generator().manual_seed(0)
def bench(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
SHAPES = [("conv3", 32768, 512, 512, 0, 0, 1, 1), ("qkv", 32768, 1536, 512, 0, 1, 0, 0), ("qkv2", 32768, 3072, 512, 0, 1, 0, 0),
          ("wo", 32768, 512, 512, 1, 0, 1, 0), ("ffn1", 32768, 1024, 512, 0, 1, 0, 1), ("ffn2", 32768, 512, 1024, 1, 0, 1, 0),
          ("cross.q", 32768, 512, 512, 0, 1, 0, 0), ("ragged", 33000, 320, 256, 1, 1, 1, 1), ("c3big", 262144, 512, 512, 0, 0, 1, 1)]
for name, M, N, K, res, ln, st, relu in SHAPES:
    x = torch.randn(M, K, generator=g).cuda() + (3.0 if st else 0.0)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).cuda() if res else None
    lnarg = ((torch.rand(M, K // 64, 2, generator=g) + 1.0).cuda(), torch.randn(N, generator=g).cuda(), 1e-6) if ln else None
    outs = {}
    for nm, variant in (("tile32", 32768 | 1024), ("tile", 32768), ("stream", 65536)):
        y = torch.empty(M, N, device="cuda")
        o = native.linear(x, w, b, relu=bool(relu), residual=r, out=y, ln=lnarg, want_stats=bool(st), variant=variant)
        outs[nm] = (o if isinstance(o, tuple) else (o, None), bench(lambda: native.linear(x, w, b, relu=bool(relu), residual=r, out=y, ln=lnarg, want_stats=bool(st), variant=variant)))
    (y32, s32), t32 = outs["tile32"]; (y0, s0), t0 = outs["tile"]; (ys, ss), ts = outs["stream"]
    same = torch.equal(y32, ys)
    dst = "-"
    if st:
        # statistics: compare the row moments they imply (sum, M2 about the segment mean) against float64 of the output
        seg = ys.double().view(M, N // 64, 64)
        ref = torch.stack((seg.sum(-1), ((seg - seg.mean(-1, keepdim=True)) ** 2).sum(-1)), -1)
        e_s = ((ss.double() - ref).abs() / (ref.abs() + 1e-3)).max().item()
        e_0 = ((s0.double() - ref).abs() / (ref.abs() + 1e-3)).max().item()
        dst = f"stats rel err vs fp64: stream {e_s:.1e} tile {e_0:.1e}"
    tf = lambda us: 2.0 * M * N * K / (us * 1e-6) / 1e12
    print(f"{name:8s} M={M} N={N} K={K}: bit-identical to the 32x32x2 tile kernel {same}  max|d| vs auto tile {float((y0 - ys).abs().max()):.1e}  {dst}   "
          f"tile32 {t32:.1f} us ({tf(t32):.1f})  tile auto {t0:.1f} us ({tf(t0):.1f})  stream {ts:.1f} us ({tf(ts):.1f} TF/s)", flush=True)
