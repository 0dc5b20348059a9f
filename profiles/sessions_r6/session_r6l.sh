#!/bin/bash
# usage: profiles/session_r6l.sh <tag>  -- vcrnetIter with target reuse: bit-identity tests, then configs[2] with and without it
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_hip_variants.py -m gpu -x -q -k "reuse or forward_sized" > $OUT/${TAG}_reuse_tests.txt 2>&1; tail -25 $OUT/${TAG}_reuse_tests.txt
pick='import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);s=d["stages"];print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],4), "stage", round(d["knn_edgeconv_stage"]["hbm_frac"],4), d["knn_edgeconv_stage"].get("clouds_processed_per_pair"), "acc", round(d["accounted_frac"],3), "frac", round(d["roofline"]["frac"],3), {k: round(v["ms_per_step"],3) for k,v in s.items()})'
for rep in 1 2; do
  python bench.py --partial --points 1024 --batch 24 --iters 3 --no-iter-reuse --no-cpu-baseline --no-other-configs --min-seconds 4 2>/dev/null | python -c "$pick" "configs[2] recompute"
  python bench.py --partial --points 1024 --batch 24 --iters 3 --no-cpu-baseline --no-other-configs --min-seconds 4 2>/dev/null | python -c "$pick" "configs[2] reuse    "
done > $OUT/${TAG}_config3_reuse_ab.txt 2>&1
python bench.py --iters 3 --no-iter-reuse --no-cpu-baseline --no-other-configs --min-seconds 3 2>/dev/null | python -c "$pick" "whole N=1024 B=16 iters 3 recompute" >> $OUT/${TAG}_config3_reuse_ab.txt 2>&1
python bench.py --iters 3 --no-cpu-baseline --no-other-configs --min-seconds 3 2>/dev/null | python -c "$pick" "whole N=1024 B=16 iters 3 reuse    " >> $OUT/${TAG}_config3_reuse_ab.txt 2>&1
cat $OUT/${TAG}_config3_reuse_ab.txt
python bench.py --partial --points 1024 --batch 24 --iters 3 --no-cpu-baseline --no-other-configs --min-seconds 3 --stages > $OUT/${TAG}_config3_bench.json 2> $OUT/${TAG}_config3_launch_table.txt; head -40 $OUT/${TAG}_config3_launch_table.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -n "passed\|failed\|Error" | head
