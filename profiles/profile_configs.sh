#!/bin/bash
# usage: profiles/profile_configs.sh <tag>   (run on the GPU box through gpurun)
# The other BASELINE configs with the same bench command (JSON line + per-launch table each), the full-protocol CPU
# baseline of SURVEY 8d, and the parity figures the tests print.
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
python bench.py --partial --points 1024 --batch 24 --iters 3 --stages --no-cpu-baseline --no-other-configs > $OUT/${TAG}_config3_bench.json 2> $OUT/${TAG}_config3_launch_table.txt
python bench.py --points 2048 --batch 16 --stages --no-cpu-baseline --no-other-configs > $OUT/${TAG}_config4_1gpu_bench.json 2> $OUT/${TAG}_config4_launch_table.txt
python bench.py --points 4096 --k 40 --batch 32 --stages --no-cpu-baseline --no-other-configs > $OUT/${TAG}_config5_bench.json 2> $OUT/${TAG}_config5_launch_table.txt
python bench.py --emb-nn dgcnn --stages --no-cpu-baseline --no-other-configs > $OUT/${TAG}_dgcnn_bench.json 2> $OUT/${TAG}_dgcnn_launch_table.txt
python bench.py --linear-mode bf16x3 --no-cpu-baseline --no-other-configs > $OUT/${TAG}_bf16x3_bench.json 2>/dev/null
python bench.py --linear-mode bf16x3+sdpa --stages --no-cpu-baseline --no-other-configs > $OUT/${TAG}_bf16x3_sdpa_bench.json 2> $OUT/${TAG}_bf16x3_sdpa_launch_table.txt
python bench.py --emb-nn pointnet --stages --no-cpu-baseline --no-other-configs > $OUT/${TAG}_pointnet_bench.json 2> $OUT/${TAG}_pointnet_launch_table.txt
for b in 1 2 4; do python bench.py --batch $b --stages --no-cpu-baseline --no-other-configs --min-seconds 3 > $OUT/${TAG}_batch${b}_bench.json 2> $OUT/${TAG}_batch${b}_launch_table.txt; done
python bench.py --gpus 2 --backend gloo --no-cpu-baseline --no-other-configs > $OUT/${TAG}_selfspawn_2ranks_1gpu_bench.json 2>/dev/null
python -m pytest tests/test_hip_forced.py tests/test_hip_forward.py tests/test_hip_variants.py tests/test_selfdiv.py tests/test_eval_golden.py tests/test_hip_partial.py tests/test_hip_regimes.py tests/test_icp_eval.py -m gpu -q -s 2>&1 | grep -E "max\||flips|config|passed|failed|vs the reference|HIP vs|kept-key|ff_dims|margins|k = |partial=" > $OUT/${TAG}_parity_printout.txt
python evaluate.py --items 64 --batch 16 > $OUT/${TAG}_evaluate_whole.txt 2>&1
python evaluate.py --partial --iters 3 --batch 24 --items 48 > $OUT/${TAG}_evaluate_partial.txt 2>&1
python profiles/bench_knn.py > $OUT/${TAG}_bench_knn.txt 2>&1
python profiles/bench_linear_shapes.py > $OUT/${TAG}_bench_linear_shapes.txt 2>&1
python profiles/bench_sdpa.py > $OUT/${TAG}_bench_sdpa.txt 2>&1
ls -la $OUT | tail -20
