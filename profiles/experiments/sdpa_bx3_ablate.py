#!/usr/bin/env python
"""Experiment: timing ABLATIONS of vcr_sdpa_bf16x3_f32's tile loop (results of the ablated builds are wrong by construction).
The loop's cycle count per tile is deterministic (8400 in every run, profiles/rounds4-5/r4y_timeline_sdpa_bf16x3.txt), so each removed
piece is an exact attribution.  Builds scratch/bx3/lib_sdpa_<name>.so (translation unit alone, probes on); timed by
profiles/experiments/timeline_sdpa_bf16x3.py through VCR_TL_LIB.
  python profiles/experiments/sdpa_bx3_ablate.py build | run"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(ROOT, "scratch", "bx3")
SRC = os.path.join(ROOT, "vcr-net_amd", "csrc", "attention_bf16x3.hip")

STAGE = "      stage_piece(cur ^ 1, tile + 2, s8);\n"
EXP = "      s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -mref));\n"
PS0 = "    for (int i = 0; i < 4; ++i) split3x2(s[2 * i], s[2 * i + 1], PH[0][i], PM[0][i], PL[0][i]);\n"
PS1 = "        if (sp == 0) split3x2(s[8 + 2 * dt], s[9 + 2 * dt], PH[1][dt], PM[1][dt], PL[1][dt]);\n"
KF = "      for (int pl = 0; pl < 3; ++pl) kf[pl] = *reinterpret_cast<const bf16x8*>(&S.k[pl][l31][16 * s8 + 8 * half]);\n"
VF = "          vf[pl] = *reinterpret_cast<const bf16x8*>(&S.vt[pl][32 * dt + l31][16 * sp + 8 * half]);\n"
SYNC = "    __syncthreads();\n    cur ^= 1;\n"
CHEAP_P0 = "    for (int i = 0; i < 4; ++i) { PH[0][i] = __float_as_uint(s[2 * i]); PM[0][i] = __float_as_uint(s[2 * i + 1]); PL[0][i] = PH[0][i] ^ PM[0][i]; }\n"
CHEAP_P1 = "        if (sp == 0) { PH[1][dt] = __float_as_uint(s[8 + 2 * dt]); PM[1][dt] = __float_as_uint(s[9 + 2 * dt]); PL[1][dt] = PH[1][dt] ^ PM[1][dt]; }\n"
VARIANTS = {
    "base": [],
    "no_stage": [(STAGE, "")],
    "no_exp": [(EXP, "      s[r] = fmaf(s[r], c2, -mref);\n")],
    "no_psplit": [(PS0, CHEAP_P0), (PS1, CHEAP_P1)],
    "no_kfrag": [(KF, "      for (int pl = 0; pl < 3; ++pl) kf[pl] = qm[(s8 + pl) & 7];\n")],
    "no_vfrag": [(VF, "          vf[pl] = ql[(dt + pl + 4 * sp) & 7];\n")],
    "no_sync": [(STAGE, ""), (SYNC, "    cur ^= 0;\n")],
    "mfma_only": [(STAGE, ""), (EXP, "      s[r] = fmaf(s[r], c2, -mref);\n"), (PS0, CHEAP_P0), (PS1, CHEAP_P1),
                  (KF, "      for (int pl = 0; pl < 3; ++pl) kf[pl] = qm[(s8 + pl) & 7];\n"), (VF, "          vf[pl] = ql[(dt + pl + 4 * sp) & 7];\n"),
                  (SYNC, "    cur ^= 0;\n")],
}


def build():
    src = re.sub(r"^(\s*)//@probe ", r"\1", open(SRC).read(), flags=re.M)
    for name, patches in VARIANTS.items():
        txt = src
        for old, new in patches:
            assert txt.count(old) == 1, (name, old)
            txt = txt.replace(old, new)
        p = os.path.join(OUT, f"sdpa_{name}.hip")
        open(p, "w").write(txt)
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                            "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "vcr-net_amd", "csrc"),
                            "-include", os.path.join(HERE, "probes.h"), "-DVCR_PROBE_TU_attention_bf16x3", "-shared",
                            "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(OUT, f"lib_sdpa_{name}.so"), p],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        print(name, [l.split("remark:")[1].split("[")[0].strip() for l in r.stderr.splitlines() if " VGPRs:" in l][-1])


def run():
    for name in VARIANTS:
        env = dict(os.environ, VCR_TL_LIB=os.path.join(OUT, f"lib_sdpa_{name}.so"))
        r = subprocess.run([sys.executable, os.path.join(HERE, "timeline_sdpa_bf16x3.py"), "randn"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("randn")]
        print(f"{name:10s}", line[0][9:] if line else r.stderr[-300:], flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
