#!/bin/bash
# usage: profiles/session_r6c.sh <tag>   (GPU box)  -- persistent sdpa A/B, in-launch tie replay through global slots, guard at 0.8
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_tests.txt 2>&1
tail -15 $OUT/${TAG}_gpu_tests.txt
python profiles/bench_sdpa_persist.py > $OUT/${TAG}_sdpa_persist.txt 2>&1
pick='import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);s=d["stages"];print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],4), "sdpa ms", round(s["sdpa"]["ms_per_step"],4), round(s["sdpa"]["tflops"],1), "knn", round(d["knn_edgeconv_stage"]["knn_ms_per_step"],4), "stage", round(d["knn_edgeconv_stage"]["hbm_frac"],4), "acc", round(d["accounted_frac"],3))'
for rep in 1 2; do
  for v in 1 2; do
    python bench.py --sdpa-variant $v --no-cpu-baseline --no-other-configs --min-seconds 5 2>/dev/null | python -c "$pick" "configs[1] sdpa-variant $v"
  done
done > $OUT/${TAG}_sdpa_variant_bench.txt 2>&1
for v in 1 2; do
  python bench.py --sdpa-variant $v --points 4096 --k 40 --batch 32 --no-cpu-baseline --no-other-configs --min-seconds 3 2>/dev/null | python -c "$pick" "configs[4] sdpa-variant $v"
  python bench.py --sdpa-variant $v --points 2048 --batch 16 --no-cpu-baseline --no-other-configs --min-seconds 3 2>/dev/null | python -c "$pick" "configs[3] sdpa-variant $v"
  python bench.py --sdpa-variant $v --partial --points 1024 --batch 24 --iters 3 --no-cpu-baseline --no-other-configs --min-seconds 3 2>/dev/null | python -c "$pick" "configs[2] sdpa-variant $v"
done >> $OUT/${TAG}_sdpa_variant_bench.txt 2>&1
python bench.py --points 4096 --k 40 --batch 32 --no-cpu-baseline --no-other-configs --min-seconds 3 --stages > $OUT/${TAG}_config5_bench.json 2> $OUT/${TAG}_config5_launch_table.txt
python profiles/bench_knn_guard.py > $OUT/${TAG}_knn_guard.txt 2>&1
cat $OUT/${TAG}_sdpa_persist.txt $OUT/${TAG}_sdpa_variant_bench.txt; head -12 $OUT/${TAG}_config5_launch_table.txt; cat $OUT/${TAG}_knn_guard.txt
