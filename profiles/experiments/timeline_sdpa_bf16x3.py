#!/usr/bin/env python
"""Experiment: prologue / tile loop of vcr_sdpa_bf16x3_f32 from the `//@probe` stamps, with the shader clock inside the loop.
  python profiles/experiments/timeline_sdpa_bf16x3.py build   # build container (the translation unit alone, probes on)
  python profiles/experiments/timeline_sdpa_bf16x3.py [randn|forward]   # GPU box; operands N(0,1) or shaped like the forward's (LayerNorm-ed rows x 1/sqrt(512) weights)"""
import ctypes as C
import math
import os
import re
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(ROOT, "scratch", "bx3")
LIB = os.environ.get("VCR_TL_LIB", os.path.join(OUT, "lib_probe_sdpa.so"))


def build():
    os.makedirs(OUT, exist_ok=True)
    txt = open(os.path.join(ROOT, "vcr-net_amd", "csrc", "attention_bf16x3.hip")).read()
    txt = re.sub(r"^(\s*)//@probe ", r"\1", txt, flags=re.M)
    p = os.path.join(OUT, "probe_sdpa.hip")
    open(p, "w").write(txt)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                    "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "vcr-net_amd", "csrc"),
                    "-include", os.path.join(HERE, "probes.h"), "-DVCR_PROBE_TU_attention_bf16x3", "-shared", "-o", LIB, p], check=True)
    print("built", LIB)


def main():
    import torch
    sys.path.insert(0, ROOT)
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd.native import SdpaArgs, ptr, stream_ptr
    L = C.CDLL(LIB)
    L.vcr_sdpa_bf16x3_f32.argtypes = [C.POINTER(SdpaArgs), C.c_void_p]
    L.vcr_dbg_probe_attention_bf16x3.argtypes = [C.c_void_p, C.c_int]
    g = torch.Generator().manual_seed(0)
    nb, h, N = 32, 4, 1024
    full = np.zeros((4096, 32), np.uint64)
    for kind in (sys.argv[1:] or ["randn", "forward"]):
        planes = kind == "planes"                             # K | V as bf16x3 planes (vcr_sdpa_bf16x3_planes_f32), N(0,1) operands
        if kind in ("randn", "planes"):
            qkv = torch.randn(nb * N, 1536, generator=g).cuda()
        else:
            x = torch.randn(nb * N, 512, generator=g)
            x = (x - x.mean(1, keepdim=True)) / x.std(1, keepdim=True)
            qkv = (x @ (torch.randn(512, 1536, generator=g) / math.sqrt(512))).cuda()
        q, k, v = qkv[:, :512], qkv[:, 512:1024], qkv[:, 1024:]
        out = torch.empty(nb * N, 512, device="cuda")
        a = SdpaArgs()
        a.q, a.ldq, a.k, a.ldk, a.v, a.ldv = ptr(q), 1536, ptr(k), 1536, ptr(v), 1536
        a.out, a.ldo, a.nbatch, a.heads, a.nq, a.nk, a.scale = ptr(out), 512, nb, h, N, N, 1 / math.sqrt(128)
        fn = lambda: L.vcr_sdpa_bf16x3_f32(C.byref(a), C.c_void_p(stream_ptr()))
        if planes:                                            # (only in a build with profiles/experiments/kv_planes_dataflow.patch applied)
            if not hasattr(L, "vcr_sdpa_bf16x3_planes_f32"):
                print("planes   : this build has no vcr_sdpa_bf16x3_planes_f32 (apply kv_planes_dataflow.patch)")
                continue
            from vcrnet_amd import native
            from vcrnet_amd.native import SdpaPlanes
            kv = qkv[:, 512:].contiguous()
            pl = native.split_bf16x3(kv).view(3, nb * N, 1024)
            pp = SdpaPlanes(ptr(pl), ptr(pl[:, :, 512:]), 1024, nb * N * 1024)
            L.vcr_sdpa_bf16x3_planes_f32.argtypes = [C.POINTER(SdpaArgs), C.POINTER(SdpaPlanes), C.c_void_p]
            fn = lambda: L.vcr_sdpa_bf16x3_planes_f32(C.byref(a), C.byref(pp), C.c_void_p(stream_ptr()))
        assert fn() == 0
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        L.vcr_dbg_probe_attention_bf16x3(None, 1)
        fn()
        torch.cuda.synchronize()
        L.vcr_dbg_probe_attention_bf16x3(full.ctypes.data, 0)
        ok = full[:, 0] > 0
        wall = (full[ok, 2] - full[ok, 1]).astype(np.float64) * 0.01
        cyc = (full[ok, 18] - full[ok, 17]).astype(np.float64)
        pro = (full[ok, 1] - full[ok, 0]).astype(np.float64) * 0.01
        print(f"{kind:8s} {us:7.1f} us/launch = {4.0 * nb * h * N * N * 128 / us / 1e6:6.1f} TF/s-eq; {ok.sum()} workgroups: prologue "
              f"{np.median(pro):.2f} us, tile loop {np.median(wall):.1f} us at {np.median(cyc / wall) / 1e3:.2f} GHz = "
              f"{np.median(cyc) / (N // 32):.0f} cycles per 32-key tile (the two waves of a SIMD need 6144 on the matrix pipe)")


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else main()
