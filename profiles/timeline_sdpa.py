#!/usr/bin/env python
"""Where an sdpa workgroup's time goes (probe build: python profiles/experiments/probe_build.py; wave 0 stamps the 100 MHz
clock at start / after the prologue (Q in registers, first K | V tile staged) / after the key loop / after the output
stores are acknowledged).  Run on the GPU box."""
import ctypes as C, math, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vcrnet_amd  # noqa
from vcrnet_amd import native
native.LIB_PATH = os.path.join(ROOT, "scratch", "libvcr_probe.so")
L = native.lib()
L.vcr_dbg_probe_attention.argtypes = [C.c_void_p, C.c_int]
full = np.zeros((4096, 32), np.uint64)
for nb, N in ((32, 1024), (32, 2048)):
    qkv = torch.randn(nb * N, 1536, device="cuda")
    q, k, v = qkv[:, :512], qkv[:, 512:1024], qkv[:, 1024:]
    fn = lambda: native.sdpa(q, k, v, nb, 4, N, N, 1 / math.sqrt(128))
    for _ in range(5): fn()
    torch.cuda.synchronize()
    L.vcr_dbg_probe_attention(None, 1)
    fn(); torch.cuda.synchronize()
    L.vcr_dbg_probe_attention(full.ctypes.data, 0)
    t = full[:, :16].astype(np.float64) * 0.01
    used = t[:, 0] > 0
    tt = t[used]; t0 = tt[:, 0].min()
    print(f"--- sdpa nb={nb} N={N}: {int(used.sum())} workgroups stamped (of {nb * 4 * ((N + 127) // 128)}); kernel span {tt[tt > 0].max() - t0:.1f} us; "
          f"start skew median {np.median(tt[:, 0] - t0):.1f} max {(tt[:, 0] - t0).max():.1f}")
    for a, b, nm in ((0, 1, "prologue"), (1, 2, "key loop"), (2, 3, "epilogue + store ack")):
        d = tt[:, b] - tt[:, a]
        print(f"    {nm:22s} median {np.median(d):7.2f}  p10 {np.percentile(d, 10):7.2f}  p90 {np.percentile(d, 90):7.2f} us")
