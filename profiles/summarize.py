#!/usr/bin/env python
"""Condense rocprofv3 output (gpurun_out/<dir>/.../*_kernel_stats.csv and *_counter_collection.csv)
into the small per-round files committed under profiles/.

  python profiles/summarize.py r1 gpurun_out/prof_r1 gpurun_out/pmc_fetch_r1 gpurun_out/pmc_write_r1

PMC units follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced read, so the read side is doubled."""
import collections
import csv
import glob
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def short(name):
    for key in ("linear_glds_pair_kernel", "linear_kernel", "sdpa_kernel", "edgeconv_dg_kernel", "softcorr_kernel", "layernorm512_kernel",
                "knn3_kernel", "knn64_kernel", "gathermax_kernel", "pointwise12_kernel", "rigid_svd_kernel",
                "rowside_kernel", "linear_glds16_kernel", "linear_glds_kernel", "linear_persist_kernel",
                "edgeconv_dg_packed_kernel", "edgeconv_dg_pipe_kernel", "pairscore_kernel", "rankselect_kernel", "knn_tiebreak_kernel",
                "knn_pair_kernel", "knn64c_kernel", "sdpa16_kernel", "knn_tiebreak2_kernel", "edgechain_kernel",
                "vcr_copy_words_kernel", "zero_i32_kernel", "pose_step_kernel", "edgeconv_dg_packed_bf16x3_kernel",
                "keymass4_kernel", "keymass_kernel", "statmerge_kernel", "rowstat_merge_kernel", "score_colpass_kernel",
                "score_rowpass_kernel", "gather_rows_kernel", "sdpa_bf16x3_kernel", "linear_bf16x3_kernel", "sdpa_persist_kernel", "knn_morton_kernel",
                "knn_rank_rows_kernel", "knn_order_guard_kernel", "split_bf16x3_kernel", "fold_layernorm_kernel", "make_pairs_kernel",
                "gathermax_lds_kernel", "sdpa_merge_kernel", "icp_kernel", "zero_count_kernel"):
        if key in name:
            return key + (name[name.index(key) + len(key):].split("(")[0] if "<" in name else "")
    return name[:48]


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    pmc_dirs = sys.argv[3:]
    rows = []
    for f in glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]),
                         float(r["Percentage"]), float(r["MinNs"]), float(r["MaxNs"])))
    rows.sort(key=lambda r: -r[2])
    with open(os.path.join(HERE, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
        wr = csv.writer(fh)                                # (kernel names carry template commas: quoted)
        wr.writerow(["kernel", "calls", "total_ns", "avg_ns", "percent", "min_ns", "max_ns"])
        wr.writerows(rows)
    pmc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for d in pmc_dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                a = pmc[short(r["Kernel_Name"])][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    out = {}
    for k, cs in pmc.items():
        e = {c: {"avg": v[0] / v[1], "dispatches": v[1]} for c, v in cs.items()}
        fetch = e.get("FETCH_SIZE", {}).get("avg")
        write = e.get("WRITE_SIZE", {}).get("avg")
        if fetch is not None and write is not None:
            e["hbm_bytes_per_launch"] = (2.0 * fetch + write) * 1024.0     # gfx950 FETCH_SIZE correction
        out[k] = e
    if out:
        sys.path.insert(0, os.path.dirname(HERE))
        import vcrnet_amd  # noqa: F401
        from vcrnet_amd import build as vb
        import datetime
        out["_meta"] = {"kernel_sources_sha16": vb.sources_sha16(), "tag": tag,
                        "date_utc": datetime.datetime.utcnow().strftime("%Y-%m-%d %H:%M")}
        json.dump(out, open(os.path.join(HERE, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    print("wrote", tag, len(rows), "kernels,", len(out), "pmc entries")


if __name__ == "__main__":
    main()
