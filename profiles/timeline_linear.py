#!/usr/bin/env python
"""Experiment: where does a linear launch spend its time?  Builds (or reuses) a scratch copy of the library with
the `//@probe` stamps switched on (profiles/experiments/probe_build.py; wave 0 of every workgroup stamps the 100 MHz wall clock at start / after the prologue / after each
tile's k loop / at the end) and prints the distribution per phase for the persistent kernel and the round-1 kernels.

  python profiles/timeline_linear.py build     # in the build container (hipcc): writes scratch/libvcr_probe.so
  python profiles/timeline_linear.py           # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.environ.get("VCR_TL_LIB", os.path.join(ROOT, "scratch", "libvcr_probe.so"))


def build():
    subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "experiments", "probe_build.py"), "--out", LIB], check=True)


def main():
    import torch
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import native
    native.LIB_PATH = LIB
    L = native.lib()
    L.vcr_dbg_probe_linear.argtypes = [C.c_void_p, C.c_int]
    L.vcr_dbg_probe_linear.restype = C.c_int
    M = 2 * 16 * 1024
    full = np.zeros((4096, 32), np.uint64)                # [workgroup][16 wall-clock stamps | 16 shader-clock stamps]
    for name, N, K, res, ln, st in [("qkv", 1536, 512, 0, 1, 0), ("wo", 512, 512, 1, 0, 1), ("ffn2", 512, 1024, 1, 0, 1),
                                    ("cross.q", 512, 512, 0, 1, 0)]:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda") if res else None
        y = torch.empty(M, N, device="cuda")
        lnarg = (torch.rand(M, K // 64, 2, device="cuda") + 1.0, torch.randn(N, device="cuda"), 1e-6) if ln else None
        for variant in (0,):
            fn = lambda: native.linear(x, w, b, residual=r, out=y, ln=lnarg, want_stats=bool(st), variant=variant)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            L.vcr_dbg_probe_linear(None, 1)
            fn()
            torch.cuda.synchronize()
            L.vcr_dbg_probe_linear(full.ctypes.data, 0)
            buf = full[:, :16]
            t = buf.astype(np.float64) * 0.01                  # us
            used = t[:, 0] > 0
            tt = t[used]
            t0 = tt[:, 0].min()
            nmarks = int((tt > 0).sum(1).max())
            print(f"--- {name} N={N} K={K} variant={variant}: {used.sum()} workgroups, {nmarks} marks; kernel span "
                  f"{tt[tt > 0].max() - t0:.1f} us")
            print(f"    start skew: median {np.median(tt[:, 0] - t0):.1f} max {(tt[:, 0] - t0).max():.1f} us")
            # stamp slots in program order: 0 start, 1 prologue done, 2 k loop done, 4 accumulators in LDS, 9 bias / column
            # sums arrived, 7 / 8 after 1 / 4 of pass 0's eight read-compute-store steps, 5 / 6 pass 0 / 1 issued, 3 stores acknowledged
            order = [m for m in (0, 1, 2, 4, 9, 7, 8, 5, 6, 3) if (tt[:, m] > 0).any()]
            for a_, b_ in zip(order[:-1], order[1:]):
                ok = (tt[:, a_] > 0) & (tt[:, b_] > 0)
                d = tt[ok, b_] - tt[ok, a_]
                print(f"    mark {a_}->{b_}: median {np.median(d):7.2f}  p10 {np.percentile(d, 10):7.2f}  p90 "
                      f"{np.percentile(d, 90):7.2f} us   (abs end: median {np.median(tt[ok, b_] - t0):7.1f})")


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else main()
