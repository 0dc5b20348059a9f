#!/bin/bash
# usage: profiles/session_r6e.sh <tag>   (GPU box)  -- guard rewrite check, fuzz soaks on round 6's kernels
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_tests.txt 2>&1
tail -4 $OUT/${TAG}_gpu_tests.txt
python bench.py --points 4096 --k 40 --batch 32 --no-cpu-baseline --no-other-configs --min-seconds 3 --stages > $OUT/${TAG}_config5_bench.json 2> $OUT/${TAG}_config5_launch_table.txt
grep "knn" $OUT/${TAG}_config5_launch_table.txt
python bench.py --points 2048 --batch 16 --no-cpu-baseline --no-other-configs --min-seconds 3 --stages > $OUT/${TAG}_config4_1gpu_bench.json 2> $OUT/${TAG}_config4_launch_table.txt
grep "knn" $OUT/${TAG}_config4_launch_table.txt
timeout 600 python profiles/fuzz_knn_ordered.py 606 300 > $OUT/${TAG}_fuzz_knn_ordered_300.txt 2>&1; tail -1 $OUT/${TAG}_fuzz_knn_ordered_300.txt
timeout 600 python profiles/fuzz_knn.py 61 60 > $OUT/${TAG}_fuzz_knn_60.txt 2>&1; tail -2 $OUT/${TAG}_fuzz_knn_60.txt
timeout 900 python profiles/fuzz_whole.py 62 80 > $OUT/${TAG}_fuzz_whole_80.txt 2>&1; tail -3 $OUT/${TAG}_fuzz_whole_80.txt
timeout 900 python profiles/fuzz_partial.py 63 24 > $OUT/${TAG}_fuzz_partial_24.txt 2>&1; tail -3 $OUT/${TAG}_fuzz_partial_24.txt
VCRNET_LINEAR_MODE=bf16x3+sdpa timeout 900 python profiles/fuzz_whole.py 64 40 > $OUT/${TAG}_fuzz_whole_40_bf16x3_sdpa.txt 2>&1; tail -3 $OUT/${TAG}_fuzz_whole_40_bf16x3_sdpa.txt
