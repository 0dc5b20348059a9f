#!/bin/bash
# usage: profiles/session_r6b.sh <tag>   (GPU box)  -- after the packing-cache fix: GPU tests, the split line, the guard's calibration
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
rm -f $OUT/accuracy_ledger.txt
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_tests.txt 2>&1
tail -5 $OUT/${TAG}_gpu_tests.txt
python bench.py --linear-mode bf16x3+sdpa --no-cpu-baseline --no-other-configs --min-seconds 6 --stages > $OUT/${TAG}_bf16x3_sdpa_bench.json 2> $OUT/${TAG}_bf16x3_sdpa_launch_table.txt
python profiles/host_gap.py --linear-mode bf16x3+sdpa > $OUT/${TAG}_host_gap_split.json 2>/dev/null
python profiles/bench_knn_guard.py > $OUT/${TAG}_knn_guard.txt 2>&1
for cfg in "--points 2048 --batch 16" "--points 4096 --k 40 --batch 32"; do
  for reg in default randemb; do
    python bench.py $cfg --regime $reg --no-cpu-baseline --no-other-configs --min-seconds 3 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$cfg', '$reg', round(d['value'],1), round(d['ms_per_step'],3), 'knn ms', round(d['knn_edgeconv_stage']['knn_ms_per_step'],4), 'stage', round(d['knn_edgeconv_stage']['hbm_frac'],4), 'acc', round(d['accounted_frac'],3))"
  done
done > $OUT/${TAG}_regimes.txt 2>&1
python bench.py --stages > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench_launch_table.txt
python -c "
import json;d=json.load(open('$OUT/${TAG}_bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['accounted_frac']);[print(o['baseline_config'], round(o['value'],1), o['ms_per_step'], o['accounted_frac']) for o in d['other_configs']]"
cat $OUT/${TAG}_regimes.txt $OUT/${TAG}_knn_guard.txt $OUT/${TAG}_host_gap_split.json
