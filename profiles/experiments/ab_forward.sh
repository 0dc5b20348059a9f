#!/bin/bash
# usage: profiles/experiments/ab_forward.sh <tag> <libA.so> <libB.so> [bench.py args]  -- same-box A/B of two builds of the library through
# bench.py (alternating, two rounds each); prints pairs/s and writes the per-launch tables to gpurun_out/<tag>_{A,B}_launch_table.txt
TAG=$1; A=$2; B=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
cp vcr-net_amd/libvcr_hip.so /tmp/libvcr_hip.keep
for r in 1 2; do
  for v in A B; do
    [ $v = A ] && cp $A vcr-net_amd/libvcr_hip.so || cp $B vcr-net_amd/libvcr_hip.so
    python3 bench.py --no-cpu-baseline --no-other-configs --min-seconds 3 --stages "$@" 2> gpurun_out/${TAG}_${v}_launch_table.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v round $r', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],4), 'ms')"
  done
done
cp /tmp/libvcr_hip.keep vcr-net_amd/libvcr_hip.so
