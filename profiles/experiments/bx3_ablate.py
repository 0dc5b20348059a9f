#!/usr/bin/env python
"""Experiment: what holds vcr_linear_bf16x3_f32 at ~0.37 of its 417 TFLOP/s-equivalent roofline?

Timing-only ABLATIONS of vcr-net_amd/csrc/linear_bf16x3.hip (results of the ablated builds are wrong by construction and are
never compared): the translation unit compiles alone in two seconds, so every variant is a textual patch of a scratch copy,
built to scratch/bx3/lib_<name>.so and timed through its own ctypes binding.

  python profiles/experiments/bx3_ablate.py build      # build container (hipcc)
  python profiles/experiments/bx3_ablate.py            # GPU box
"""
import ctypes as C
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "vcr-net_amd", "csrc", "linear_bf16x3.hip")
OUT = os.path.join(ROOT, "scratch", "bx3")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

A_PART = """          split_half(c >> 1, c & 1);
          if (c & 1) {
            store_a(c >> 1);
            if constexpr (more2) load_a((kt + 2) * TK, c >> 1);
          }
"""
B_PART = """          store_b(c - 8);
          if constexpr (more2) load_b((kt + 2) * TK, c - 8);
"""
BAR2 = "    lds_barrier();                                       // barrier 2: every wave holds its fragments, the image is free\n"
BAR1 = "    lds_barrier();                                       // barrier 1: slab kt is in LDS\n"
SPLIT = "    for (int e = 0; e < 2; ++e) split3(ra[i][2 * e2 + e], h[e], m[e], l[e]);\n"
FRAGS = """        fa[i][pl] = *reinterpret_cast<const bf16x8*>(fap[i] + pl * TM * TK);
        fb[i][pl] = *reinterpret_cast<const bf16x8*>(fbp[i] + pl * TN * TK);
"""
# fragments that do not come from LDS (the loop keeps its barriers; registers seeded from the staged loads so that nothing folds)
FAKE = """        fa[i][pl] = rb[(i + pl) % 6];
        fb[i][pl] = rb[(i + 2 * pl + 1) % 6];
"""

VARIANTS = {
    "base": [],
    "no_a": [(A_PART, "")],       # no activation loads, split or LDS stores
    "no_split": [(SPLIT, "    for (int e = 0; e < 2; ++e) { h[e] = __float_as_uint(ra[i][2 * e2 + e]) >> 16; m[e] = __float_as_uint(ra[i][2 * e2 + e]) & 0xffff; l[e] = h[e] ^ m[e]; }\n")],
    "no_b": [(B_PART, "")],
    "no_a_no_b": [(A_PART, ""), (B_PART, "")],
    "no_bar2": [(A_PART, ""), (B_PART, ""), (BAR2, "")],
    "no_bars": [(A_PART, ""), (B_PART, ""), (BAR2, ""), (BAR1, "")],
    "no_frag": [(FRAGS, FAKE)],
    "mfma_only": [(A_PART, ""), (B_PART, ""), (BAR2, ""), (BAR1, ""), (FRAGS, FAKE)],
}


def build():
    os.makedirs(OUT, exist_ok=True)
    for f in os.listdir(OUT):
        if f.startswith("lib_") and f != "lib_probe.so":
            os.remove(os.path.join(OUT, f))
    src = open(SRC).read()
    for name, patches in VARIANTS.items():
        txt = src
        # (every variant is held at two workgroups per CU, as the product kernel's registers do: pad the dynamic LDS request)
        patches = patches + [("  const int lds = sizeof(Stage3) + TM * 2 * sizeof(float);", "  const int lds = sizeof(Stage3) + TM * 2 * sizeof(float) + 24576;")]
        for old, new in patches:
            if txt.count(old) != 1:
                sys.exit(f"{name}: patch anchor not found exactly once:\n{old}")
            txt = txt.replace(old, new)
        txt = re.sub(r"^(\s*)//@probe ", r"\1", txt, flags=re.M)     # phase stamps on: main() reports the k loop's shader clock
        p = os.path.join(OUT, f"{name}.hip")
        open(p, "w").write(txt)
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                            "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I",
                            os.path.join(ROOT, "vcr-net_amd", "csrc"), "-include", os.path.join(ROOT, "profiles", "experiments", "probes.h"),
                            "-DVCR_PROBE_TU_linear_bf16x3", "-shared", "-Rpass-analysis=kernel-resource-usage",
                            "-o", os.path.join(OUT, f"lib_{name}.so"), p], capture_output=True, text=True)
        if r.returncode:
            sys.exit(f"{name}:\n{r.stderr}")
        use = [l.split("remark: ")[-1].split(" [")[0].strip() for l in r.stderr.splitlines() if "linear_bf16x3_kernel" in l.split("remark:")[0] or True]
        k = [i for i, l in enumerate(use) if "linear_bf16x3_kernel" in l]
        print(name, "|", " ; ".join(x for x in use[k[0]:k[0] + 12] if x.startswith(("VGPRs:", "VGPRs Spill", "Occupancy"))) if k else "")


def main():
    import torch
    sys.path.insert(0, ROOT)
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import native
    from vcrnet_amd.native import LinearArgs, ptr, stream_ptr
    g = torch.Generator().manual_seed(0)
    M = 32768
    shapes = (("qkv", 1536, 512), ("ffn2", 512, 1024))
    data = {}
    for name, N, K in shapes:
        x = torch.randn(M, K, generator=g).cuda()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        data[name] = (x, native.split_bf16x3(w), torch.randn(N, generator=g).cuda(), torch.empty(M, N, device="cuda"))
    import numpy as np
    full = np.zeros((4096, 32), np.uint64)
    names = sorted(f[4:-3] for f in os.listdir(OUT) if f.startswith("lib_") and f.endswith(".so") and f != "lib_probe.so")
    for rnd in range(2):
        for vn in names:
            L = C.CDLL(os.path.join(OUT, f"lib_{vn}.so"))
            L.vcr_linear_bf16x3_f32.argtypes = [C.POINTER(LinearArgs), C.c_void_p, C.c_void_p]
            L.vcr_linear_bf16x3_f32.restype = C.c_int
            line = f"{vn:16s}"
            for name, N, K in shapes:
                x, planes, b, y = data[name]
                a = LinearArgs(ptr(x), x.stride(0), None, ptr(b), None, 0, ptr(y), y.stride(0), M, N, K, 0)

                def fn():
                    rc = L.vcr_linear_bf16x3_f32(C.byref(a), ptr(planes), C.c_void_p(stream_ptr()))
                    assert rc == 0, rc
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 30 * 1e3
                L.vcr_dbg_probe_linear_bf16x3.argtypes = [C.c_void_p, C.c_int]
                L.vcr_dbg_probe_linear_bf16x3(None, 1)
                fn()
                torch.cuda.synchronize()
                L.vcr_dbg_probe_linear_bf16x3(full.ctypes.data, 0)
                ok = full[:, 0] > 0
                wall = (full[ok, 2] - full[ok, 1]).astype(np.float64) * 0.01            # us in the k loop (100 MHz stamps)
                cyc = (full[ok, 18] - full[ok, 17]).astype(np.float64)
                line += (f"   {name} {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF/s-eq  k loop {np.median(wall):6.2f} us at "
                         f"{np.median(cyc / wall) / 1e3:.2f} GHz = {np.median(cyc) / (K // 32):5.0f} cyc/slab")
            print(line, flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else main()
