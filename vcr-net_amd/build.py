"""Build libvcr_hip.so (gfx950) in-tree with hipcc.  Usage: python vcr-net_amd/build.py [--force]"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libvcr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]   # no silent a*b+c fusion: the kernels spell out every fmaf they want


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def sources_sha16() -> str:
    """sha256 (first 16 hex digits) over the kernel sources and headers, names included, in sorted order: what a profile
    summary under profiles/ records so that bench.py can say whether the counters it quotes were taken on THIS code."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files.append(os.path.join(os.path.dirname(HERE), "include", "vcr_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def kernel_resources():
    """{demangled-ish kernel name: {"vgprs", "agprs", "sgprs", "scratch", "occupancy", "lds", "file"}} from the remarks of the last
    build (build() first: objects without a remarks file are recompiled)."""
    import re
    out = {}
    for f in sorted(os.listdir(OBJ)):
        if not f.endswith(".remarks"):
            continue
        cur = None
        for ln in open(os.path.join(OBJ, f)):
            m = re.search(r"remark: (?:Function Name: (\S+)|\s+(\w[\w ]*?)(?: \[[^\]]*\])?: (\d+)) \[-Rpass", ln)
            if not m:
                continue
            if m.group(1):
                cur = out.setdefault(m.group(1), {"file": f[:-8] + ".hip"})
            elif cur is not None:
                key = {"VGPRs": "vgprs", "AGPRs": "agprs", "TotalSGPRs": "sgprs", "ScratchSize": "scratch",
                       "Occupancy": "occupancy", "LDS Size": "lds", "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill"}.get(m.group(2))
                if key:
                    cur[key] = int(m.group(3))
    return out


def _digest(paths, extra=""):
    import hashlib
    h = hashlib.sha256(extra.encode())
    for f in paths:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _recorded(path):
    try:
        with open(path) as fh:
            return fh.read().strip()
    except OSError:
        return None


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile what changed and link.  Staleness is decided by CONTENT (a digest of the source, the headers and the flags,
    recorded next to every object), not by modification times: a file transfer -- the snapshot that carries the tree to the
    GPU box -- may reorder mtimes, and must neither trigger a rebuild there nor hide one that is needed."""
    os.makedirs(OBJ, exist_ok=True)
    hdrs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "vcr_hip.h"))
    flags = " ".join(FLAGS)
    jobs, want = [], {}
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        want[obj] = _digest([src] + hdrs, flags)
        if (force or not os.path.exists(obj) or not os.path.exists(obj[:-2] + ".remarks") or _recorded(obj + ".sha") != want[obj]):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        # the compiler's per-kernel resource remarks (registers, scratch, occupancy) are kept next to the object:
        # tests/test_kernel_resources.py holds the hot kernels to "no scratch" and to their resident-workgroup counts
        cmd = [HIPCC, "-Rpass-analysis=kernel-resource-usage"] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0:
            with open(obj[:-2] + ".remarks", "w") as fh:
                fh.write(r.stderr)
            with open(obj + ".sha", "w") as fh:
                fh.write(want[obj])
        out = r.stdout + r.stderr
        if r.returncode == 0 and not verbose:                # keep warnings, drop the remarks and their source echoes
            import re
            keep, skip = [], False
            for ln in out.splitlines():
                if "-Rpass-analysis=kernel-resource-usage" in ln:
                    skip = True
                    continue
                if skip and re.match(r"\s*(\d+\s*)?\|", ln):
                    continue
                skip = False
                keep.append(ln)
            out = "\n".join(keep) if any(("warning" in ln or "error" in ln) for ln in keep) else ""
        return src, r.returncode, out

    failed = False
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for src, rc, out in ex.map(cc, jobs):
            if rc != 0 or verbose or out.strip():
                sys.stderr.write(f"--- {os.path.basename(src)} (rc={rc})\n{out}\n")
            failed |= rc != 0
    if failed:
        raise RuntimeError("hipcc failed")
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in sources()]
    lib_want = _digest([], "".join(want[o] for o in objs))
    if force or jobs or not os.path.exists(LIB) or _recorded(os.path.join(OBJ, "libvcr_hip.so.sha")) != lib_want:
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        with open(os.path.join(OBJ, "libvcr_hip.so.sha"), "w") as fh:
            fh.write(lib_want)
    return LIB


def build_host_example() -> str:
    """examples/host_cpp/forward_host: the C-ABI driven from C++ with no Python / PyTorch in the process (its own hipMalloc'ed
    memory and stream).  Host code only -- compiles in seconds, here or on the GPU box; rebuilt when its source, the header or
    the library's digest changed."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "examples", "host_cpp", "forward_host.cpp")
    exe = os.path.join(root, "examples", "host_cpp", "forward_host")
    want = _digest([src, os.path.join(root, "include", "vcr_hip.h")], _recorded(os.path.join(OBJ, "libvcr_hip.so.sha")) or "")
    if os.path.exists(exe) and _recorded(exe + ".sha") == want:
        return exe
    r = subprocess.run([HIPCC, "-O2", "-std=c++17", "-I" + os.path.join(root, "include"), src, "-L" + HERE, "-lvcr_hip",
                        "-Wl,-rpath,$ORIGIN/../../vcr-net_amd", "-o", exe], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("forward_host.cpp failed to build:\n" + r.stdout + r.stderr)
    with open(exe + ".sha", "w") as fh:
        fh.write(want)
    return exe


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
