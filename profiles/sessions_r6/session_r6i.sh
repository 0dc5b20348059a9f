#!/bin/bash
# usage: profiles/session_r6i.sh <tag>  -- the ordered search beyond 4096 points: tests, fuzz, timing at 16 x 8192
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_tests.txt 2>&1
grep -n "passed\|failed" $OUT/${TAG}_gpu_tests.txt
timeout 900 python profiles/fuzz_knn_ordered.py 707 300 > $OUT/${TAG}_fuzz_knn_ordered_300.txt 2>&1; tail -1 $OUT/${TAG}_fuzz_knn_ordered_300.txt
python - > $OUT/${TAG}_knn_ordered_8192.txt 2>&1 <<'PY'
import sys, math, torch, numpy as np
sys.path.insert(0, '.')
import vcrnet_amd
from vcrnet_amd import native as nat, weights
sys.path.insert(0, 'profiles')
from bench_knn_guard import bench, stem
wd = weights.generate_weights(1234, lpd=weights.load_lpd_fixture())
for B, N, k in ((16, 8192, 20), (16, 8192, 40), (24, 6000, 20), (32, 4096, 20), (32, 4097, 20)):
    g = torch.Generator().manual_seed(N)
    xyz = torch.rand(B, N, 3, generator=g) - 0.5
    x4 = torch.cat((xyz, (xyz ** 2).sum(-1, keepdim=True)), -1).cuda().contiguous()
    feat = stem(xyz.transpose(1, 2).contiguous().cuda(), wd)
    sq = (feat ** 2).sum(-1).contiguous()
    ft = feat.view(B, N, 4, 4, 4).transpose(3, 4).reshape(B, N, 64).contiguous()
    res = {}
    def plain(): res['p'] = nat.knn_pair(feat, sq, x4, k, xt=ft, tie_slots=True)
    def ordered():
        o = nat.knn_order(x4, ft, sq, guard=True)
        res['o'] = nat.knn_pair(feat, sq, x4, k, xt=ft, order=o, tie_slots=True)
    tp, to = bench(plain, 5), bench(ordered, 5)
    same = all(torch.equal(torch.sort(res['p'][i], -1).values, torch.sort(res['o'][i], -1).values) for i in (0, 1))
    print(f"{B:3d} x {N:5d} k={k:2d}: plain {tp:9.1f} us   ordered (ranking + guard included) {to:9.1f} us   ratio {to / tp:.2f}   sets equal: {same}", flush=True)
PY
cat $OUT/${TAG}_knn_ordered_8192.txt
python profiles/bench_gathermax_order.py > $OUT/${TAG}_gathermax_order.txt 2>&1; cat $OUT/${TAG}_gathermax_order.txt
python bench.py --points 4096 --k 40 --batch 32 --no-cpu-baseline --no-other-configs --min-seconds 3 --stages > $OUT/${TAG}_config5_bench.json 2> $OUT/${TAG}_config5_launch_table.txt
grep "knn\|gathermax" $OUT/${TAG}_config5_launch_table.txt
python -c "
import json;d=json.load(open('$OUT/${TAG}_config5_bench.json'));print(d['value'], d['ms_per_step'], d['knn_edgeconv_stage']['hbm_frac'])"
