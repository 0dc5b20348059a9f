#!/usr/bin/env python
"""In-kernel clock of the fp32 `linear` kernels' k loop (MI355X_MICROARCH.md, DVFS item 6; VERDICT r3 task 6).

Needs scratch/libvcr_probe.so (python profiles/experiments/probe_build.py, in the build container).  On the GPU box:
  python profiles/experiments/clock_probe_linear.py
For each of the forward's big linear shapes and both MFMA shapes: >= 2 s of back-to-back launches on random data, then
one launch whose stamps are read back -- lane 0 of every workgroup stamped s_memrealtime (100 MHz) and s_memtime (shader
clock) after the prologue (stamp 1) and after the k loop (stamp 2).  Reported per shape:
  launch time and TFLOP/s by HIP events; the in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz over the k loop,
  median over workgroups; the k loop's shader cycles per MFMA per SIMD against the nominal issue rate (64 for 32x32x2, 32
  for 16x16x4 -- with W workgroups per CU, W waves share a SIMD: cycles x 1 / W per MFMA); the share of the launch that
  is the k loop (median over workgroups of (stamp2 - stamp1) / (stamp3 - stamp0))."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vcrnet_amd  # noqa: E402,F401
from vcrnet_amd import native  # noqa: E402

native.LIB_PATH = os.environ.get("VCR_PROBE_LIB", os.path.join(ROOT, "scratch", "libvcr_probe.so"))


def main():
    L = native.lib()
    L.vcr_dbg_probe_linear.argtypes = [C.c_void_p, C.c_int]
    L.vcr_dbg_probe_linear.restype = C.c_int
    M = 2 * 16 * 1024
    buf = np.zeros((4096, 32), np.uint64)
    print(f"device {torch.cuda.get_device_name(0)}; M = {M} rows, random operands; {time.strftime('%Y-%m-%d %H:%M UTC', time.gmtime())}")
    for name, N, K, res, ln, st in [("qkv", 1536, 512, 0, 1, 0), ("ffn1", 1024, 512, 0, 1, 0), ("wo", 512, 512, 1, 0, 1),
                                    ("ffn2", 512, 1024, 1, 0, 1)]:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda") if res else None
        y = torch.empty(M, N, device="cuda")
        lnarg = (torch.rand(M, K // 64, 2, device="cuda") + 1.0, torch.randn(N, device="cuda"), 1e-6) if ln else None
        for variant, ms_shape in ((1024, 32), (16, 16)):
            fn = lambda: native.linear(x, w, b, residual=r, out=y, ln=lnarg, want_stats=bool(st), variant=variant)
            cfg = native.linear_config(x, w, b, residual=r, out=y, ln=lnarg, want_stats=bool(st), variant=variant) \
                if hasattr(native, "linear_config") else None
            t0 = time.time()
            n = 0
            while time.time() - t0 < 2.0:
                for _ in range(50):
                    fn()
                torch.cuda.synchronize()
                n += 50
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            L.vcr_dbg_probe_linear(None, 1)
            fn()
            torch.cuda.synchronize()
            L.vcr_dbg_probe_linear(buf.ctypes.data, 0)
            used = buf[:, 0] > 0
            wall = buf[used, 0:4].astype(np.float64)
            shader = buf[used, 16:20].astype(np.float64)
            ghz = (shader[:, 2] - shader[:, 1]) / (wall[:, 2] - wall[:, 1]) * 0.1
            bk = 32 if res else 16
            wg_per_cu = 2 if res else 4
            nk = K // bk
            mfma_per_wave = nk * (bk // (2 if ms_shape == 32 else 4)) * (4 if ms_shape == 32 else 16)
            cyc = np.median(shader[:, 2] - shader[:, 1]) / mfma_per_wave / wg_per_cu
            share = np.median((wall[:, 2] - wall[:, 1]) / np.maximum(1.0, wall[:, 3] - wall[:, 0]))
            tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
            print(f"{name:5s} N={N:4d} K={K:4d} {ms_shape}x{ms_shape}: {ms * 1e3:7.1f} us {tf:6.1f} TF/s ({tf / 157.3:.3f})  "
                  f"in-kernel clock over the k loop: median {np.median(ghz):.3f} GHz (p10 {np.percentile(ghz, 10):.3f}, p90 "
                  f"{np.percentile(ghz, 90):.3f}; {int(used.sum())} workgroups)  {cyc:5.1f} shader cycles per MFMA per SIMD at "
                  f"{wg_per_cu} waves/SIMD (nominal {64 if ms_shape == 32 else 32})  k loop = {share:.2f} of a workgroup's life  "
                  f"[{n} warm-up launches, config {cfg}]")


if __name__ == "__main__":
    main()
