#!/usr/bin/env python
"""Where does a step's time go?  Reads a rocprofv3 --kernel-trace CSV (*_kernel_trace.csv) of a bench.py run and splits every
steady-state step (from one stem launch, `pointwise12`, to the next) into kernel time and idle time between kernels.

  python profiles/trace_gaps.py <dir with *_kernel_trace.csv> [first-kernel-substring]

Prints one JSON object: per step the wall span, the sum of kernel durations and the idle time (median / min / max over the
steps of the trace), `accounted_frac` = kernel sum / span, and per kernel name the median duration and the median idle gap IN
FRONT of it -- a gap in front of a kernel that is not the first of a step is the host (or the queue) not keeping up."""
import collections
import csv
import glob
import json
import os
import sys

import numpy as np


def load(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    return rows


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").replace("(anonymous namespace)::", "")[:64]


def main():
    d = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else "pointwise12"
    rows = load(d)
    starts = [i for i, r in enumerate(rows) if first in r[2]]
    steps = []
    for a, b in zip(starts[:-1], starts[1:]):
        ks = rows[a:b]
        span = rows[b][0] - ks[0][0]                         # start of this step's stem to the start of the next one's
        ksum = sum(e - s for s, e, _ in ks)
        steps.append((span, ksum, ks, rows[b][0]))
    if not steps:
        print(json.dumps({"error": "no steps found", "kernels": len(rows)}))
        return
    # steady state: steps whose span is within 1.5x of the median (drops the blocks' fences and the warm-up)
    med = float(np.median([s[0] for s in steps]))
    ss = [s for s in steps if s[0] < 1.5 * med]
    n_launch = collections.Counter(len(s[2]) for s in ss).most_common(1)[0][0]
    ss = [s for s in ss if len(s[2]) == n_launch]
    dur = collections.defaultdict(list)
    gap = collections.defaultdict(list)
    for span, ksum, ks, nxt in ss:
        for i, (s, e, n) in enumerate(ks):
            key = "%02d %s" % (i, short(n))
            dur[key].append(e - s)
            if i > 0:
                gap[key].append(s - ks[i - 1][1])
        gap["00 " + short(ks[0][2])].append(0)
        gap["zz (end of step -> next stem)"].append(nxt - ks[-1][1])
    spans = np.array([s[0] for s in ss], dtype=float) / 1e6
    ksums = np.array([s[1] for s in ss], dtype=float) / 1e6
    out = {
        "steps_in_trace": len(steps), "steady_steps": len(ss), "launches_per_step": n_launch,
        "span_ms": {"median": float(np.median(spans)), "min": float(spans.min()), "max": float(spans.max())},
        "kernel_sum_ms": {"median": float(np.median(ksums)), "min": float(ksums.min()), "max": float(ksums.max())},
        "idle_ms": {"median": float(np.median(spans - ksums))},
        "accounted_frac": float(np.median(ksums / spans)),
        "first_vs_last_steady_kernel_sum_ms": [float(ksums[:5].mean()), float(ksums[-5:].mean())],
        "per_launch_us": {k: {"dur": round(float(np.median(dur[k])) / 1e3, 2),
                              "gap_before": round(float(np.median(gap[k])) / 1e3, 2)} for k in sorted(dur)},
        "end_of_step_gap_us": round(float(np.median(gap["zz (end of step -> next stem)"])) / 1e3, 2),
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
