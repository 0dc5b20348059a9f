#!/usr/bin/env python
"""Headline benchmark: point-cloud pairs/sec of the VCR-Net registration hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL.)

One "step" = one pass of the whole hot path (VCRNet.forward: LPDNet kNN-graph embedding ->
Transformer virtual-correspondence block -> soft correspondences -> SVD rigid solve) over one batch
of 16 synthetic pairs of N=1024 points PER GPU (BASELINE.json configs[1]; weak scaling), inputs already
resident in HBM, plus -- for N > 1 -- the RCCL all-gather of the per-rank (R, t).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="pairs per GPU")
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--partial", action="store_true",
                    help="partial-overlap mode (BASELINE configs[2]): clouds cropped to int(points*0.7507) points, "
                         "key pruning + selectCom/getCopair heads; combine with --points 1024 --batch 24 --iters 3")
    ap.add_argument("--iters", type=int, default=1, help="vcrnetIter refinement passes per step (one C call)")
    ap.add_argument("--emb-nn", default="lpdnet", choices=["lpdnet", "dgcnn"],
                    help="feature extractor (--emb_nn of the reference; dgcnn uses seeded weights)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stages", action="store_true", help="print the per-launch table to stderr")
    ap.add_argument("--linear-mode", default="fp32", choices=["fp32", "bf16x3"],
                    help="fp32: linears on v_mfma_f32_32x32x2_f32 (default).  bf16x3: the same products as exact 3-way "
                         "bf16 splits on the bf16 matrix pipe (fp32-equivalent accuracy)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL on ROCm); gloo only "
                                                      "for exercising the multi-rank control flow on a 1-GPU box")
    return ap.parse_args()


def model_args(partial=False, emb_nn="lpdnet"):
    from vcrnet_amd import synth
    return SimpleNamespace(emb_dims=512, cycle=False, emb_nn=emb_nn, pointer="transformer", vcp_nn="topK",
                           partial=partial, overlap2=synth.OVERLAP2_0575 if partial else 0.75, t3d=False, tfea=False,
                           n_blocks=1, dropout=0.0, ff_dims=1024, n_heads=4)


def cpu_baseline(w, B, N, k, partial=False, iters=1):
    """The CPU oracle (a port of the reference's PyTorch CPU path) timed on this box's host cores on a
    bounded sample of the same workload."""
    import oracle
    from vcrnet_amd import synth
    nthreads = torch.get_num_threads()
    sample_B = min(B, 8 if N <= 1024 else 2)      # bounded: ~10-30 s of CPU work
    src, tgt, _, _, _ = synth.make_batch(0, sample_B, N, partial=partial, kind="object" if N <= 2048 else "uniform")
    s, t = torch.from_numpy(src), torch.from_numpy(tgt)
    cfg = oracle.OracleConfig(k=k, partial=partial, overlap2=synth.OVERLAP2_0575 if partial else 0.75)
    run = lambda: oracle.vcrnet_iter(w, s, t, cfg, iters=iters)
    t0 = time.perf_counter()
    run()                                         # warm-up
    warm = time.perf_counter() - t0
    reps = 3 if warm < 8 else 1
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        run()
        ts.append(time.perf_counter() - t0)
    best = float(np.median(ts))
    return {"value": sample_B / best, "unit": "pairs/s", "cores": nthreads, "kind": "port",
            "sample": f"oracle.vcrnet_iter(iters={iters}{', partial' if partial else ''}), B={sample_B}, N={N}, k={k}, "
                      f"fp32, median of {reps} after 1 warm-up, torch.set_num_threads={nthreads}"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks (WORLD_SIZE={world})")
    ndev = torch.cuda.device_count()
    if a.backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but {ndev} GPUs: one process per GPU")
    local = local % max(1, ndev)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)

    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import native, shard, synth, weights, workmodel
    from vcrnet_amd.module import VCRNet

    w = (weights.generate_weights(1234, lpd=weights.load_lpd_fixture()) if a.emb_nn == "lpdnet"
         else weights.generate_weights(1234, emb_nn="dgcnn"))
    net = VCRNet(model_args(a.partial, a.emb_nn))
    net.load_state_dict(w)
    net.emb_nn.k = a.k
    net.linear_mode = a.linear_mode
    net = net.to(dev).eval()

    B, N = a.batch, a.points
    # each rank owns B consecutive items of the global batch (weak scaling); inputs live in HBM
    # object-like clouds have 2048 points (the ModelNet40 convention); larger N (configs 4/5) use uniform clouds
    # built on the device from base clouds + host-drawn permutations / poses (vcr_make_pairs_f32; untimed)
    src, tgt, _, _, _ = synth.make_batch_device(rank * B, B, N, partial=a.partial,
                                                kind="object" if N <= 2048 else "uniform", device=dev)
    Nfull, N = N, src.shape[2]                     # partial mode crops the clouds (1024 -> 768)
    assert src.shape == (B, 3, N) and src.is_cuda, src.shape

    def step(trace=None):
        with torch.no_grad():
            out = net._forward_fused(src, tgt, trace=trace, iters=a.iters)
        pose = torch.cat((out[2].view(B, 9), out[3]), 1)
        if world > 1:
            pose = shard.all_gather_poses(pose, world)
        return pose

    # one-time initialisation, not a warm-up step: the first call loads the code objects, packs / folds the weights,
    # sizes the workspace and (N > 1) opens the RCCL communicator
    step()
    torch.cuda.synchronize()
    for _ in range(a.warmup):
        step()
    traces = [native.LaunchTrace() for _ in range(a.steps)]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:   # sanity: the gathered poses must hold every rank's block in rank order
        g = step()
        mine = g[rank * B:(rank + 1) * B]
        own = step()[rank * B:(rank + 1) * B]
        assert g.shape == (world * B, 12) and torch.allclose(mine, own)

    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(traces[i].trace)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # per-launch durations from the HIP events recorded inside the timed region
    fam_ms, fam_flops, fam_bytes, rows = {}, {}, {}, {}
    for tr in traces:
        for name, ms in tr.launches():
            fam = name.split(":")[0]
            fl, by = workmodel.launch_work(name, B, N, a.k, overlap2=net._overlap2)
            fam_ms[fam] = fam_ms.get(fam, 0.0) + ms
            fam_flops[fam] = fam_flops.get(fam, 0.0) + fl
            fam_bytes[fam] = fam_bytes.get(fam, 0.0) + by
            r = rows.setdefault(name, [0.0, fl, by, 0])
            r[0] += ms; r[3] += 1
        tr.close()
    if rank == 0:
        dom = max(fam_ms, key=fam_ms.get)
        bound = workmodel.FAMILY_BOUND[dom]
        if bound == "mfma":
            ach = fam_flops[dom] / (fam_ms[dom] * 1e-3) / 1e12
            peak = workmodel.PEAK_MFMA_F32_TFLOPS
            if dom == "linear" and a.linear_mode == "bf16x3":
                peak = workmodel.PEAK_MFMA_BF16_TFLOPS / 6.0    # six bf16 MFMA products per fp32-equivalent MAC
            roof = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": peak,
                    "unit": "TFLOP/s", "frac": ach / peak, "traffic": None}
            if dom == "linear" and a.linear_mode == "bf16x3":
                roof["note"] = "fp32-equivalent FLOPs; peak = 2500 TFLOP/s dense bf16 / 6 MFMAs per split product"
        else:
            ach = fam_bytes[dom] / (fam_ms[dom] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": workmodel.PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": ach / workmodel.PEAK_HBM_GBS, "traffic": None}
        # HBM traffic per launch of that kernel comes from SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
        # of this same command, condensed by profiles/summarize.py; null when no summary is committed.
        import glob
        pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
        if pmcs and (B, N, a.k, a.partial, a.iters) == (16, 1024, 20, False, 1):
            kname = {"linear": "linear_glds", "sdpa": "sdpa_kernel<false, true>",
                     "edgeconv": "edgeconv_dg_packed_kernel<20>", "softcorr": "pairscore_kernel<0>"}.get(dom)
            # a family can be several template instantiations (linear: plain / statistics-out / LayerNorm-in):
            # launch-weighted mean over the entries whose name starts with the family's kernel name
            ents = [e for k_, e in json.load(open(pmcs[-1])).items()
                    if kname and k_.startswith(kname) and "hbm_bytes_per_launch" in e]
            if ents:
                nd = [e["FETCH_SIZE"]["dispatches"] for e in ents]
                roof["traffic"] = sum(e["hbm_bytes_per_launch"] * n for e, n in zip(ents, nd)) / sum(nd)
                roof["traffic_source"] = os.path.relpath(pmcs[-1], ROOT)
        total_ms = sum(fam_ms.values())
        roof["launches_per_step"] = sum(r[3] for n, r in rows.items() if n.startswith(dom + ":")) // a.steps
        roof["avg_launch_ms"] = fam_ms[dom] / max(1, roof["launches_per_step"] * a.steps)
        stages = {f: {"ms_per_step": fam_ms[f] / a.steps, "share": fam_ms[f] / total_ms,
                      "tflops": fam_flops[f] / (fam_ms[f] * 1e-3) / 1e12, "gbs": fam_bytes[f] / (fam_ms[f] * 1e-3) / 1e9}
                  for f in sorted(fam_ms, key=fam_ms.get, reverse=True)}
        # the kNN + EdgeConv (emb_nn) stage that BASELINE.json's north_star prices against the HBM roofline:
        # algorithmic bytes 2*N*(7448 + 784*k) per pair, time = its launches inside the timed region
        emb_sites = ("pointwise:", "knn:", "edgeconv:", "gathermax:", "linear:dg1_pq", "linear:sn1_pq", "linear:conv3")
        emb_ms = sum(r[0] for n, r in rows.items() if n.startswith(emb_sites)) / a.steps
        emb_bytes = 2.0 * N * (7448 + 784 * a.k) * B
        emb_gf = sum(workmodel.launch_work(n, B, N, a.k)[0] for n in
                     ("linear:dg1_pq", "edgeconv:dg1_dg2", "linear:sn1_pq", "linear:conv3")) / 1e9
        emb_stage = {"ms_per_step": emb_ms, "algorithmic_bytes_per_pair": emb_bytes / B,
                     "achieved_gbs": emb_bytes / (emb_ms * 1e-3) / 1e9,
                     "hbm_frac": emb_bytes / (emb_ms * 1e-3) / 1e9 / workmodel.PEAK_HBM_GBS,
                     "note": "fp32 1x1 convs of this stage are MFMA-bound (SURVEY section 7): %.1f GF per step alone "
                             "need %.2f ms at the fp32 matrix peak, i.e. <= %.2f of the HBM roofline"
                             % (emb_gf, emb_gf / workmodel.PEAK_MFMA_F32_TFLOPS,
                                emb_bytes / (emb_gf / workmodel.PEAK_MFMA_F32_TFLOPS * 1e-3) / 1e9 / workmodel.PEAK_HBM_GBS)}
        if a.stages:
            for n, r in sorted(rows.items(), key=lambda kv: -kv[1][0]):
                ms = r[0] / r[3]
                print(f"{n:28s} {ms:8.3f} ms  {r[1] / ms / 1e9:8.2f} TF/s  {r[2] / ms / 1e6:9.1f} GB/s", file=sys.stderr)
        pairs = B * world * a.steps
        line = {
            "metric": "point-cloud pairs/sec (N=%d, batch %d per GPU)" % (Nfull, B), "value": pairs / elapsed, "unit": "pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.linear_mode == "fp32" else "f32 (linears as exact bf16x3 splits, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]: partial-to-partial overlap 0.575 (clouds cropped %d -> %d "
                                    "points), " % (Nfull, N) if a.partial else
                                    "BASELINE configs[1]: ModelNet40-like whole-to-whole registration, ") +
                                   "N=%d, batch=%d pairs per GPU, LPDNet(k=%d)+Transformer+VcpTopK+SVD, iter=%d, fp32; "
                                   "synthetic object clouds with the reference's transform recipe; LPD-pretrained "
                                   "emb_nn + seeded Transformer weights" % (N, B, a.k, a.iters),
                       "num_points": N, "batch_per_gpu": B, "global_batch": B * world, "k": a.k,
                       "parallelism": f"dp{world} (pairs sharded per rank, RCCL all-gather of R,t)"},
            "roofline": roof,
            "stages": stages,
            "knn_edgeconv_stage": emb_stage,
            "flops_per_pair_reference": workmodel.reference_flops_per_pair(N, a.k)["total"],
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w, B, Nfull, a.k, a.partial, a.iters)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
