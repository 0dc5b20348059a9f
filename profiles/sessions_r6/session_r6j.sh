#!/bin/bash
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "gathermax" 2>&1 | tail -3
python profiles/bench_gathermax_order.py > $OUT/${TAG}_gathermax_order.txt 2>&1; cat $OUT/${TAG}_gathermax_order.txt
python bench.py --points 4096 --k 40 --batch 32 --no-cpu-baseline --no-other-configs --min-seconds 3 --stages > $OUT/${TAG}_config5_bench.json 2> $OUT/${TAG}_config5_launch_table.txt
grep "knn\|gathermax" $OUT/${TAG}_config5_launch_table.txt
python -c "
import json;d=json.load(open('$OUT/${TAG}_config5_bench.json'));print(d['value'], d['ms_per_step'], d['knn_edgeconv_stage']['hbm_frac'])"
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -n "passed\|failed"
