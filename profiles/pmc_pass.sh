#!/bin/bash
# usage: [BENCH_ARGS='--linear-mfma 16'] profiles/pmc_pass.sh <tag> <counter> [<counter> ...]  -- one rocprofv3 --pmc pass (kernel-trace only) per counter
# GROUP over a short bench.py run; writes gpurun_out/pmc_<tag>_<group index>/ and a per-kernel mean table to
# gpurun_out/<tag>_pmc_<group index>.txt.  Counter groups are separated by commas: "A B,C D" = two passes.
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
IFS=',' read -ra GROUPS_ <<< "$*"
i=0
for g in "${GROUPS_[@]}"; do
  d=$OUT/pmc_${TAG}_$i
  rocprofv3 --kernel-trace --output-format csv --pmc $g -d $d -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 1 --min-seconds 0.1 $BENCH_ARGS > /dev/null 2> $OUT/${TAG}_pmc_$i.err
  python3 - "$d" > $OUT/${TAG}_pmc_$i.txt <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(rows):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in rows[k].items()}, "launches", len(next(iter(rows[k].values()))))
PY
  rm -rf $d          # the raw counter CSVs are tens of MB per pass; gpurun copies back at most 64 MiB
  i=$((i+1))
done
