#!/bin/bash
# usage: profiles/session_final.sh <tag>   (GPU box)  -- the round's final record on the committed kernel sources:
# GPU tests + smoke, the default bench + rocprofv3 stats + PMC (profile_round.sh), the exact-split line under rocprofv3
# (split_line_session.sh without an older tree), every other BASELINE config with its launch table (profile_configs.sh)
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
rm -f $OUT/accuracy_ledger.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_tests.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" >> $OUT/${TAG}_gpu_tests.txt 2>&1
grep -n "passed\|failed\|smoke" $OUT/${TAG}_gpu_tests.txt
bash profiles/profile_round.sh $TAG > $OUT/${TAG}_profile_round.log 2>&1
bash profiles/split_line_session.sh $TAG scratch/none > $OUT/${TAG}_split_session.log 2>&1
bash profiles/profile_configs.sh $TAG > $OUT/${TAG}_profile_configs.log 2>&1
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_bench.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("traffic_source_kernel_sources"), d["accounted_frac"])
print("stage", d["knn_edgeconv_stage"]["hbm_frac"], d["knn_edgeconv_stage"]["knn_ms_per_step"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
for o in d["other_configs"]: print(o["baseline_config"], round(o["value"],1), round(o["ms_per_step"],3), round(o["roofline"]["frac"],3), o["roofline"]["kernel"], round(o["knn_edgeconv_stage"]["hbm_frac"],4), round(o["accounted_frac"],3))
PY
ls $OUT | grep $TAG | wc -l
