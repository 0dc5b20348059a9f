#!/usr/bin/env python
"""Experiment: timing ABLATIONS of vcr_sdpa_f32's key loop (fp32, the headline's attention; results of the ablated builds are wrong
by construction).  NOTE: the patch anchors below are those of the LDS-DMA staging variant (profiles/experiments/sdpa_f32_lds_dma_staging.patch
applied to attention.hip); profiles/rounds4-5/r4ah_sdpa_f32_ablate.txt was taken with the anchors of the register-staged product kernel (git history).  attention.hip compiles alone; each variant is a textual patch with the `//@probe` stamps on; the runner reports
us per launch, the key loop's shader clock and cycles per 32-key tile (two workgroups per CU: 2 waves per SIMD x 128 MFMAs x 64
cycles = 16384 at the pipe rate).
  python profiles/experiments/sdpa_f32_ablate.py build | run"""
import ctypes as C
import math
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(ROOT, "scratch", "sdpa32")
SRC = os.path.join(ROOT, "vcr-net_amd", "csrc", "attention.hip")

STAGE = "    stage(cur ^ 1, min(tile + 1, ntiles - 1));           // (the last tile re-stages itself into the idle buffer: no branch)\n"
SYNC = "    __syncthreads();                                     // (next tile landed: the barrier's vmcnt(0) covers this wave's requests)\n"
EXP = "        s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -mref));\n"
KF = "      const f32x4 kf = ld4(&st[cur].k[l31][4 * ((2 * g + half) ^ (l31 & 15))]);\n"
VF = "        const f32x4 vf = ld4(&st[cur].v[acc_row(r, half)][4 * (l31 ^ (acc_row(r, half) & 15))]);\n"
VARIANTS = {
    "base": [],
    "no_stage": [(STAGE, "")],
    "no_stage_no_sync": [(STAGE, ""), (SYNC, "")],
    "stage_same_tile": [(STAGE, "    stage(cur ^ 1, t0);                                   // ABLATION: always the first tile (cache-resident source)\n")],
    "stage_k_only": [("      if (DO_PV) glds16(vbase + (size_t)key * p.ldv + c4, &st[buf].v[8 * w + 2 * i][0]);\n", "")],
    "no_exp": [(EXP, "        s[r] = fmaf(s[r], c2, -mref);\n")],
    "no_frags": [(KF, "      const f32x4 kf = qf[(g + 5) & 15];\n"), (VF, "        const f32x4 vf = qf[r];\n")],
}


def build():
    os.makedirs(OUT, exist_ok=True)
    src = re.sub(r"^(\s*)//@probe ", r"\1", open(SRC).read(), flags=re.M)
    for name, patches in VARIANTS.items():
        txt = src
        for old, new in patches:
            assert txt.count(old) == 1, (name, old)
            txt = txt.replace(old, new)
        p = os.path.join(OUT, f"{name}.hip")
        open(p, "w").write(txt)
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                            "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "vcr-net_amd", "csrc"),
                            "-include", os.path.join(HERE, "probes.h"), "-DVCR_PROBE_TU_attention", "-shared", "-o",
                            os.path.join(OUT, f"lib_{name}.so"), p], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        print("built", name)


def run():
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd.native import SdpaArgs, ptr, stream_ptr
    g = torch.Generator().manual_seed(0)
    nb, h, N = 32, 4, 1024
    qkv = torch.randn(nb * N, 1536, generator=g).cuda()
    q, k, v = qkv[:, :512], qkv[:, 512:1024], qkv[:, 1024:]
    out = torch.empty(nb * N, 512, device="cuda")
    full = np.zeros((4096, 32), np.uint64)
    for rnd in range(2):
        for name in VARIANTS:
            L = C.CDLL(os.path.join(OUT, f"lib_{name}.so"))
            L.vcr_sdpa_f32.argtypes = [C.POINTER(SdpaArgs), C.c_void_p]
            L.vcr_dbg_probe_attention.argtypes = [C.c_void_p, C.c_int]
            a = SdpaArgs()
            a.q, a.ldq, a.k, a.ldk, a.v, a.ldv = ptr(q), 1536, ptr(k), 1536, ptr(v), 1536
            a.out, a.ldo, a.nbatch, a.heads, a.nq, a.nk, a.scale = ptr(out), 512, nb, h, N, N, 1 / math.sqrt(128)
            fn = lambda: L.vcr_sdpa_f32(C.byref(a), C.c_void_p(stream_ptr()))
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
            L.vcr_dbg_probe_attention(None, 1)
            fn()
            torch.cuda.synchronize()
            L.vcr_dbg_probe_attention(full.ctypes.data, 0)
            ok = full[:, 0] > 0
            wall = (full[ok, 2] - full[ok, 1]).astype(np.float64) * 0.01
            cyc = (full[ok, 18] - full[ok, 17]).astype(np.float64)
            print(f"{name:18s} {us:7.1f} us = {4.0 * nb * h * N * N * 128 / us / 1e6:6.1f} TFLOP/s; key loop {np.median(wall):6.1f} us at "
                  f"{np.median(cyc / wall) / 1e3:.2f} GHz = {np.median(cyc) / (N // 32):6.0f} cycles per tile", flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
