#!/bin/bash
# usage: profiles/split_line_session.sh <tag> [<older tree>]   (run on the GPU box through gpurun)
# The exact-split (bf16x3+sdpa) line under the profiler, and against an older tree on the SAME box:
#  1. alternated twice: `bench.py --linear-mode bf16x3+sdpa` in <older tree> (default scratch/wt_r4 = commit 290961f) and here
#  2. profiles/host_gap.py in both arithmetic modes: host enqueue time vs device time vs the traced launches' sum
#  3. rocprofv3 --kernel-trace --stats of the split-mode bench (no HIP-event records in it) -> kernel stats + trace_gaps
#  4. the same for fp32, as the control
#  5. PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy) of the split-mode bench
TAG=$1
OLD=${2:-scratch/wt_r4}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
SPLIT="--linear-mode bf16x3+sdpa --no-cpu-baseline --no-other-configs"
pick='import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);b=d["timed_blocks"]["seconds"];print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],4), "frac", round(d["roofline"]["frac"],4), "blocks first/last ms", round(b[0]/d["steps"]*1e3,4), round(b[-1]/d["steps"]*1e3,4), "stages sum", round(sum(s["ms_per_step"] for s in d["stages"].values()),4))'
if [ -d "$OLD" ]; then
  for rep in 1 2; do
    for T in $OLD .; do
      (cd $ROOT/$T && python bench.py $SPLIT --min-seconds 5 2>/dev/null | python -c "$pick" "$T")
    done
  done > $OUT/${TAG}_ab_split.txt 2>&1
  for T in $OLD .; do
    (cd $ROOT/$T && python bench.py --no-cpu-baseline --no-other-configs --min-seconds 4 2>/dev/null | python -c "$pick" "$T fp32")
  done >> $OUT/${TAG}_ab_split.txt 2>&1
fi
python profiles/host_gap.py --linear-mode bf16x3+sdpa > $OUT/${TAG}_host_gap_split.json 2> $OUT/${TAG}_host_gap_split.err
python profiles/host_gap.py > $OUT/${TAG}_host_gap_fp32.json 2> $OUT/${TAG}_host_gap_fp32.err
for M in split fp32; do
  ARGS="--no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --min-seconds 2 --trace-every 100000"
  [ $M = split ] && ARGS="--linear-mode bf16x3+sdpa $ARGS"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_$M -- python3 bench.py $ARGS > $OUT/${TAG}_${M}_bench_under_rocprof.json 2>/dev/null
  python profiles/trace_gaps.py $OUT/prof_${TAG}_$M > $OUT/${TAG}_${M}_trace_gaps.json
done
BENCH_ARGS="--linear-mode bf16x3+sdpa"
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/pmc_${TAG}_$n -- python3 bench.py $BENCH_ARGS --no-cpu-baseline --no-other-configs --steps 3 --warmup 1 --min-seconds 0.1 > /dev/null 2>&1
done
python profiles/summarize.py ${TAG}_bf16x3_sdpa $OUT/prof_${TAG}_split $OUT/pmc_${TAG}_FETCH_SIZE $OUT/pmc_${TAG}_WRITE_SIZE $OUT/pmc_${TAG}_SQ_VALU_MFMA_BUSY_CYCLES
python profiles/summarize.py ${TAG}_fp32ctl $OUT/prof_${TAG}_fp32
cp profiles/${TAG}_*_kernel_stats.csv profiles/${TAG}_*_pmc_summary.json $OUT/ 2>/dev/null
rm -rf $OUT/prof_${TAG}_* $OUT/pmc_${TAG}_*
cat $OUT/${TAG}_ab_split.txt $OUT/${TAG}_host_gap_split.json $OUT/${TAG}_host_gap_fp32.json
grep -v per_launch $OUT/${TAG}_split_trace_gaps.json | head -30
