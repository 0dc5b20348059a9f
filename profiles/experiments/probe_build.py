#!/usr/bin/env python
"""Build scratch/libvcr_probe.so: the product kernel sources with their `//@probe ` lines switched on.

  python profiles/experiments/probe_build.py [--set NAME=VALUE ...] [--out scratch/libvcr_probe.so] [--no-probes]

Runs in the build container (hipcc cross-compiles gfx950); the .so travels to the GPU box with the snapshot (scratch/ is
git-ignored, not gpurun-ignored).  `--set` rewrites a `constexpr int NAME = ...;` tuning constant of the sources in the
scratch copy (e.g. KNN_PEND_COL16=64, LINEAR_MS_DEFAULT=16: the sweeps those constants came from); `--no-probes` leaves
the probe lines commented (a constants-only variant).  The product build (vcr-net_amd/build.py) never sees any of this.
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "vcr-net_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE")
    ap.add_argument("--out", default=os.path.join(ROOT, "scratch", "libvcr_probe.so"))
    ap.add_argument("--no-probes", action="store_true")
    a = ap.parse_args()
    work = os.path.join(ROOT, "scratch", "probed")
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(os.path.join(work, "csrc"))
    sets = dict(s.split("=", 1) for s in a.set)
    used = set()
    for f in sorted(os.listdir(CSRC)):
        txt = open(os.path.join(CSRC, f)).read()
        if not a.no_probes:
            txt = re.sub(r"^(\s*)//@probe ", r"\1", txt, flags=re.M)
        for name, val in sets.items():
            txt, n = re.subn(r"(constexpr int %s = )[^;]+;" % re.escape(name), r"\g<1>%s;" % val, txt)
            if n:
                used.add(name)
        # the sources include "../../include/vcr_hip.h" style paths relative to csrc/: keep the same depth
        open(os.path.join(work, "csrc", f), "w").write(txt)
    missing = set(sets) - used
    if missing:
        sys.exit(f"--set: no `constexpr int NAME = ...;` found for {sorted(missing)}")
    os.makedirs(os.path.join(work, "include"), exist_ok=True)
    shutil.copy(os.path.join(ROOT, "include", "vcr_hip.h"), os.path.join(work, "include", "vcr_hip.h"))
    srcs = sorted(f for f in os.listdir(os.path.join(work, "csrc")) if f.endswith(".hip"))
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function",
             "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "vcr-net_amd", "csrc")]

    def cc(f):
        stem = f[:-4]
        cmd = [HIPCC] + flags + ([] if a.no_probes else ["-include", os.path.join(HERE, "probes.h"), f"-DVCR_PROBE_TU_{stem}"]) + \
            ["-c", os.path.join(work, "csrc", f), "-o", os.path.join(work, stem + ".o")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return f, r.returncode, r.stderr

    bad = False
    with ThreadPoolExecutor(max_workers=8) as ex:
        for f, rc, err in ex.map(cc, srcs):
            if rc:
                sys.stderr.write(f"--- {f}\n{err}\n")
                bad = True
    if bad:
        sys.exit(1)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", a.out] +
                   [os.path.join(work, f[:-4] + ".o") for f in srcs], check=True)
    print("built", a.out, "sets", sets)


if __name__ == "__main__":
    main()
