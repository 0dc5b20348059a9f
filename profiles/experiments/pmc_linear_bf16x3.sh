#!/bin/bash
# usage: profiles/experiments/pmc_linear_bf16x3.sh <tag> "<counters>,<counters>"   -- one rocprofv3 --pmc pass per comma-separated
# group over profiles/experiments/bench_linear_bf16x3.py; per-kernel means to gpurun_out/<tag>_pmc_bx3_<i>.txt
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
IFS=',' read -ra GROUPS_ <<< "$*"
i=0
for g in "${GROUPS_[@]}"; do
  d=$OUT/pmc_bx3_${TAG}_$i
  rocprofv3 --kernel-trace --output-format csv --pmc $g -d $d -- python3 profiles/experiments/bench_linear_bf16x3.py > /dev/null 2> $OUT/${TAG}_pmc_bx3_$i.err
  python3 - "$d" > $OUT/${TAG}_pmc_bx3_$i.txt <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "linear_bf16x3" in r["Kernel_Name"]:
            rows[(r["Kernel_Name"][:50], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(rows):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in rows[k].items()}, "launches", len(next(iter(rows[k].values()))))
PY
  rm -rf $d
  i=$((i+1))
done
