#!/usr/bin/env python
"""The ordered kNN search where its premise fails: feature families from "a smooth function of the
coordinates" to "unrelated", at the two shapes the forward takes the ordered search on (32 x 2048, k = 20; 64 x 4096, k = 40).
Per family: the guard statistic (vcr_knn_order_args.ord_stat: mean squared tile radius / spread of the tile centroids), and
the pair launch's time -- plain, ordered without the guard, ordered with it (ranking launches included) --; the neighbour
sets of the three are compared row by row.

  python profiles/bench_knn_guard.py [--ratio R]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vcrnet_amd  # noqa: E402,F401
from vcrnet_amd import native, weights  # noqa: E402


def bench(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def stem(xyz_cf, w):
    """conv1_lpd + conv2_lpd of a weight dict (lpdnet_model.py:111-112) on the device -> rows [B, N, 64]"""
    g = lambda k: w[k].cuda().float()
    out = native.pointwise(xyz_cf, g("emb_nn.conv1_lpd.weight").view(64, 3).contiguous(), g("emb_nn.conv1_lpd.bias"),
                           g("emb_nn.conv2_lpd.weight").view(64, 64).contiguous(), g("emb_nn.conv2_lpd.bias"))
    return out[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ratio", type=float, default=0.0)
    a = ap.parse_args()
    wd = weights.generate_weights(1234, lpd=weights.load_lpd_fixture())
    wr = weights.regime_weights("randemb")
    for B, N, k in ((32, 1024, 20), (32, 2048, 20), (64, 4096, 40)):   # (1024: the forward keeps the plain scan there)
        g = torch.Generator().manual_seed(N)
        xyz = (torch.rand(B, N, 3, generator=g) - 0.5)
        x4 = torch.cat((xyz, (xyz ** 2).sum(-1, keepdim=True)), -1).cuda().contiguous()
        xcf = xyz.transpose(1, 2).contiguous().cuda()
        f_def, f_rand = stem(xcf, wd), stem(xcf, wr)
        noise = torch.relu(torch.randn(B, N, 64, generator=g)).cuda()
        fams = [("stem (LPD-pretrained)", f_def), ("stem (randemb regime)", f_rand)]
        sd = float(f_def.std())
        for amp in (0.03, 0.1, 0.3, 1.0):
            fams.append((f"stem + {amp:g} sigma noise", (f_def + amp * sd * noise).contiguous()))
        fams.append(("unrelated (relu of gaussian)", noise.contiguous()))
        for name, feat in fams:
            sq = (feat ** 2).sum(-1).contiguous()
            ft = feat.view(B, N, 4, 4, 4).transpose(3, 4).reshape(B, N, 64).contiguous()
            res = {}

            def plain():
                res["p"] = native.knn_pair(feat, sq, x4, k, xt=ft)

            def ordered():
                o = native.knn_order(x4, ft, sq)
                res["o"] = native.knn_pair(feat, sq, x4, k, xt=ft, order=o)

            def guarded():
                o = native.knn_order(x4, ft, sq, guard=True, guard_ratio=a.ratio)
                res["g"] = native.knn_pair(feat, sq, x4, k, xt=ft, order=o)
                res["stat"] = o["ord_stat"]; res["ok"] = o["ord_ok"]

            tp, to, tg = bench(plain), bench(ordered), bench(guarded)
            same = all(torch.equal(torch.sort(res["p"][i], -1).values, torch.sort(res[m][i], -1).values)
                       for m in ("o", "g") for i in (0, 1))
            st = res["stat"].cpu().numpy()
            print(f"{B:3d} x {N:4d} k={k:2d}  {name:30s} ord_stat {np.median(st):8.3f} (min {st.min():.3f} max {st.max():.3f})  "
                  f"accepted {int(res['ok'].sum())}/{B}  plain {tp:7.1f} us  ordered {to:7.1f}  guarded {tg:7.1f}  sets equal: {same}",
                  flush=True)


if __name__ == "__main__":
    main()
