#!/usr/bin/env python
"""Markdown results table of a final session for DESIGN.md section 5:  python profiles/results_table.py <tag>
Reads profiles/<tag>_bench.json (the default `python bench.py`: headline + other_configs + cpu_baseline), the session's
kernel stats / PMC summary and the split line's files, and prints the table + the per-launch paragraph."""
import csv
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    tag = sys.argv[1]
    j = lambda name: json.load(open(os.path.join(HERE, f"{tag}_{name}")))
    d = j("bench.json")
    r = d["roofline"]
    st = d["knn_edgeconv_stage"]
    print("| configuration | pairs/s | ms / step | dominant kernel, frac of its peak | kNN + EdgeConv stage, frac of HBM (kNN ms) | accounted_frac |")
    print("|---|---|---|---|---|---|")
    print(f"| configs[1] N = 1024, B = 16 (headline) | **{d['value']:.1f}** | {d['ms_per_step']:.3f} | `{r['kernel']}` {r['frac']:.3f} of {r['peak']:.1f} {r['unit']} "
          f"(HIP events, {r['launches_per_step']} launches of {r['avg_launch_ms']:.4f} ms) | {st['hbm_frac']:.3f} ({st['knn_ms_per_step']:.3f}) | {d['accounted_frac']:.3f} |")
    names = {"configs[2]": "configs[2] partial N = 768, B = 24, iter 3", "configs[3] (one GPU's share)": "configs[3], ONE GPU's share: N = 2048, B = 16",
             "configs[4]": "configs[4] N = 4096, k = 40, B = 32", "configs[1], --linear-mode bf16x3+sdpa": "configs[1], `bf16x3+sdpa` (labelled)"}
    for o in d["other_configs"]:
        ro, so = o["roofline"], o["knn_edgeconv_stage"]
        print(f"| {names.get(o['baseline_config'], o['baseline_config'])} | {o['value']:.1f} | {o['ms_per_step']:.3f} | `{ro['kernel']}` {ro['frac']:.3f} of {ro['peak']:.1f} "
              f"| {so['hbm_frac']:.3f} ({so['knn_ms_per_step']:.3f}) | {o['accounted_frac']:.3f} |")
    c = d["cpu_baseline"]
    print(f"\nCPU baseline (the oracle on the box's host, best of a thread sweep): {c['value']:.2f} pairs/s on {c['cores']} of {c['host_cpus']} threads "
          f"({c.get('one_thread') or 0:.2f} single-threaded) -> GPU / CPU = {d['value'] / c['value']:.0f}x.")
    print(f"roofline.traffic {r.get('traffic')} ({r.get('traffic_source')}, {r.get('traffic_source_kernel_sources')})")
    pl = d.get("parity_ledger") or {}
    if pl:
        e = pl["emb_rms_hip_over_ref32"]
        print(f"parity_ledger: embeddings' rms error vs the float64 twin, HIP / fp32 reference: worst {e['worst']}, median {e['median']:.3f}; ledger kernel sources {pl['ledger_kernel_sources']}, build {pl['this_build_kernel_sources']}")
    # rocprofv3 stats: per-kernel share and the linear family's fraction
    path = os.path.join(HERE, f"{tag}_kernel_stats.csv")
    if os.path.exists(path):
        rows = list(csv.DictReader(open(path)))
        tot = sum(float(x["total_ns"]) for x in rows)
        print("\nrocprofv3 --kernel-trace --stats of the same command (top kernels):")
        for x in rows[:12]:
            print(f"  {x['kernel'][:60]:60s} calls {x['calls']:>6s} avg {float(x['avg_ns']) / 1e3:9.1f} us  {100 * float(x['total_ns']) / tot:5.1f} %")
    stages = d["stages"]
    print("\nstages (ms per step):", ", ".join(f"{k} {v['ms_per_step']:.3f}" for k, v in stages.items()))


if __name__ == "__main__":
    main()
