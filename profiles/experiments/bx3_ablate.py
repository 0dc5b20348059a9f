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
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "vcr-net_amd", "csrc", "linear_bf16x3.hip")
OUT = os.path.join(ROOT, "scratch", "bx3")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

STORE = "      if (more) store_a_row(cur ^ 1, c);\n"
GLDS = "      if (more && c < 3) glds16b(wb + c * plane + (kt + 1) * TK, &st[cur ^ 1].b[c][wave * 16][0]);\n"
FRAG0 = "    frag(S, 0, fa0, fb0, 0);\n    frag(S, 0, fa0, fb0, 1);\n"
FRAG1 = "      if (c < 2) frag(S, 1, fa1, fb1, c);                 // (six 16-B reads in each of the first two chunks)\n"
SYNC = "      __builtin_amdgcn_sched_barrier(0);\n    }\n    __syncthreads();\n  }\n"
LOADA = "    if (more) load_a((kt + 1) * TK);\n"
# fragments once, before the loop (they stay in registers: the loop body has no LDS reads)
HOIST = ("  bf16x8 fa0[2][3], fb0[2][3], fa1[2][3], fb1[2][3];\n  frag(st[0], 0, fa0, fb0, 0);\n  frag(st[0], 0, fa0, fb0, 1);\n"
         "  frag(st[0], 1, fa1, fb1, 0);\n  frag(st[0], 1, fa1, fb1, 1);\n  for (int kt = 0; kt < nk; ++kt) {\n")
LOOP = "  for (int kt = 0; kt < nk; ++kt) {\n"
DECL = "    bf16x8 fa0[2][3], fb0[2][3], fa1[2][3], fb1[2][3];\n"

VARIANTS = {
    "base": [],
    "no_a": [(STORE, ""), (LOADA, "")],                                  # no activation loads, split or LDS stores
    "no_a_split": [(STORE, "      if (more && c == 0) *reinterpret_cast<f32x4*>(&st[cur ^ 1].a[0][ar0][ac * 4 * 2]) = "
                           "(ra[0] + ra[1]) + (ra[2] + ra[3]);\n")],     # loads kept (one 16-B store consumes them), no split
    "no_b": [(GLDS, "")],
    "no_frag": [(LOOP, HOIST), (DECL, ""), (FRAG0, ""), (FRAG1, "")],
    "no_sync": [(STORE, ""), (LOADA, ""), (GLDS, ""), (SYNC, "      __builtin_amdgcn_sched_barrier(0);\n    }\n  }\n")],
    "mfma_only": [(STORE, ""), (LOADA, ""), (GLDS, ""), (LOOP, HOIST), (DECL, ""), (FRAG0, ""), (FRAG1, ""),
                  (SYNC, "      __builtin_amdgcn_sched_barrier(0);\n    }\n  }\n")],
    "no_a_no_b": [(STORE, ""), (LOADA, ""), (GLDS, "")],
}


# the same MFMA-only loop on v_mfma_f32_16x16x32_bf16 (24 per output tile and slab instead of 12 of the 32x32x16 form: equal
# flops and cycles; MI355X_MICROARCH.md reports the 16x16 form ~1.15x faster in bare loops, i.e. at a higher clock)
LOOP16 = """  f32x4 acc4[2][2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc4[i][j][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa0[2][3], fb0[2][3], fa1[2][3], fb1[2][3];
  frag(st[0], 0, fa0, fb0, 0);
  frag(st[0], 0, fa0, fb0, 1);
  frag(st[0], 1, fa1, fb1, 0);
  frag(st[0], 1, fa1, fb1, 1);
  for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = c >> 1, j = c & 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bf16x8* a = (q >> 1) ? fa1[i] : fa0[i];
        const bf16x8* b = (q & 1) ? fb1[j] : fb0[j];
        f32x4 cc = acc4[i][j][q];
        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], cc, 0, 0, 0);
        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], cc, 0, 0, 0);
        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], cc, 0, 0, 0);
        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], cc, 0, 0, 0);
        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], cc, 0, 0, 0);
        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], cc, 0, 0, 0);
        acc4[i][j][q] = cc;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = acc4[i][j][r >> 2][r & 3];
"""


def build():
    os.makedirs(OUT, exist_ok=True)
    src = open(SRC).read()
    lo, hi = src.index(LOOP), src.index("  //@probe VCR_PROBE_STAMP(2);")
    VARIANTS["mfma_only_16x16"] = [(src[lo:hi], LOOP16)]
    for name, patches in VARIANTS.items():
        txt = src
        for old, new in patches:
            if txt.count(old) != 1:
                sys.exit(f"{name}: patch anchor not found exactly once:\n{old}")
            txt = txt.replace(old, new)
        p = os.path.join(OUT, f"{name}.hip")
        open(p, "w").write(txt)
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                            "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I",
                            os.path.join(ROOT, "vcr-net_amd", "csrc"), "-shared", "-Rpass-analysis=kernel-resource-usage",
                            "-o", os.path.join(OUT, f"lib_{name}.so"), p], capture_output=True, text=True)
        if r.returncode:
            sys.exit(f"{name}:\n{r.stderr}")
        use = [l.split("remark: ")[-1].strip() for l in r.stderr.splitlines()
               if "linear_bf16x3_kernel" in l or "VGPRs:" in l or "Spill" in l or "Occupancy" in l]
        print(name, "|", " ; ".join(use[:8]))


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
                line += f"   {name} {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF/s-eq"
            print(line, flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else main()
