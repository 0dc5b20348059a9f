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
# vcrnetIter with and without the reuse of the loop-invariant target cloud, alternated (BASELINE configs[2]; whole mode, 3 passes)
pick='import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);s=d["stages"];print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],4), "stage", round(d["knn_edgeconv_stage"]["hbm_frac"],4), "clouds/pair", d["knn_edgeconv_stage"].get("clouds_processed_per_pair"), "acc", round(d["accounted_frac"],3), "frac", round(d["roofline"]["frac"],3), {k: round(v["ms_per_step"],3) for k,v in s.items()})'
for rep in 1 2; do
  python bench.py --partial --points 1024 --batch 24 --iters 3 --no-iter-reuse --no-cpu-baseline --no-other-configs --min-seconds 4 2>/dev/null | python -c "$pick" "configs[2] recompute"
  python bench.py --partial --points 1024 --batch 24 --iters 3 --no-cpu-baseline --no-other-configs --min-seconds 4 2>/dev/null | python -c "$pick" "configs[2] reuse    "
done > $OUT/${TAG}_config3_reuse_ab.txt 2>&1
python bench.py --iters 3 --no-iter-reuse --no-cpu-baseline --no-other-configs --min-seconds 3 2>/dev/null | python -c "$pick" "whole N=1024 B=16 iters 3 recompute" >> $OUT/${TAG}_config3_reuse_ab.txt 2>&1
python bench.py --iters 3 --no-cpu-baseline --no-other-configs --min-seconds 3 2>/dev/null | python -c "$pick" "whole N=1024 B=16 iters 3 reuse    " >> $OUT/${TAG}_config3_reuse_ab.txt 2>&1
cat $OUT/${TAG}_config3_reuse_ab.txt
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_bench.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("traffic_source_kernel_sources"), d["accounted_frac"])
print("stage", d["knn_edgeconv_stage"]["hbm_frac"], d["knn_edgeconv_stage"]["knn_ms_per_step"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
for o in d["other_configs"]: print(o["baseline_config"], round(o["value"],1), round(o["ms_per_step"],3), round(o["roofline"]["frac"],3), o["roofline"]["kernel"], round(o["knn_edgeconv_stage"]["hbm_frac"],4), round(o["accounted_frac"],3))
PY
ls $OUT | grep $TAG | wc -l
