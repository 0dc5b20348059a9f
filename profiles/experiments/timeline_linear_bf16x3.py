#!/usr/bin/env python
"""Experiment: phases of a vcr_linear_bf16x3_f32 workgroup (prologue / k loop / epilogue) from the `//@probe` stamps.
The translation unit is compiled alone (2 s) with the probes switched on to scratch/bx3/lib_probe.so.
  python profiles/experiments/timeline_linear_bf16x3.py build   # build container
  python profiles/experiments/timeline_linear_bf16x3.py         # GPU box"""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(ROOT, "scratch", "bx3")
LIB = os.path.join(OUT, "lib_probe.so")


def build():
    os.makedirs(OUT, exist_ok=True)
    txt = open(os.path.join(ROOT, "vcr-net_amd", "csrc", "linear_bf16x3.hip")).read()
    txt = re.sub(r"^(\s*)//@probe ", r"\1", txt, flags=re.M)
    p = os.path.join(OUT, "probe.hip")
    open(p, "w").write(txt)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                    "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "vcr-net_amd", "csrc"),
                    "-include", os.path.join(HERE, "probes.h"), "-DVCR_PROBE_TU_linear_bf16x3", "-shared", "-o", LIB, p], check=True)
    print("built", LIB)


def main():
    import torch
    sys.path.insert(0, ROOT)
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import native
    from vcrnet_amd.native import LinearArgs, ptr, stream_ptr
    L = C.CDLL(LIB)
    L.vcr_linear_bf16x3_f32.argtypes = [C.POINTER(LinearArgs), C.c_void_p, C.c_void_p]
    L.vcr_dbg_probe_linear_bf16x3.argtypes = [C.c_void_p, C.c_int]
    g = torch.Generator().manual_seed(0)
    M = 32768
    full = np.zeros((4096, 32), np.uint64)
    for name, N, K, res in (("qkv", 1536, 512, 0), ("wo", 512, 512, 1), ("ffn2", 512, 1024, 1)):
        x = torch.randn(M, K, generator=g).cuda()
        planes = native.split_bf16x3((torch.randn(N, K, generator=g) / K ** 0.5).cuda())
        b = torch.randn(N, generator=g).cuda()
        r = torch.randn(M, N, generator=g).cuda() if res else None
        y = torch.empty(M, N, device="cuda")
        a = LinearArgs(ptr(x), x.stride(0), None, ptr(b), ptr(r), N if res else 0, ptr(y), N, M, N, K, 0)
        fn = lambda: L.vcr_linear_bf16x3_f32(C.byref(a), ptr(planes), C.c_void_p(stream_ptr()))
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        L.vcr_dbg_probe_linear_bf16x3(None, 1)
        fn()
        torch.cuda.synchronize()
        L.vcr_dbg_probe_linear_bf16x3(full.ctypes.data, 0)
        t = full[:, :16].astype(np.float64) * 0.01            # us (100 MHz wall clock)
        clk = full[:, 16:].astype(np.float64)
        used = t[:, 0] > 0
        tt, cc = t[used], clk[used]
        t0 = tt[:, 0].min()
        print(f"--- {name} N={N} K={K}: {used.sum()} workgroups stamped (first 4096), span {tt.max() - t0:.1f} us, "
              f"{K // 32} slabs per tile")
        for a_, b_, what in ((0, 1, "prologue"), (1, 2, "k loop"), (2, 3, "epilogue")):
            d = tt[:, b_] - tt[:, a_]
            ghz = (cc[:, b_] - cc[:, a_]) / np.maximum(d, 1e-9) / 1e3
            print(f"    {what:9s} median {np.median(d):6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f} us"
                  f"   shader clock {np.median(ghz):.2f} GHz")
        d = tt[:, 2] - tt[:, 1]
        print(f"    per slab {np.median(d) / (K // 32):.2f} us (128-row tiles: a SIMD with two workgroups' waves needs 0.64 us per slab and workgroup at 2.4 GHz)")


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else main()
