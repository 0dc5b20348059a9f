#!/usr/bin/env python
"""Evaluation run of the MI355X path on synthetic registration pairs: the counterpart of the reference's
``main.py --eval`` -> ``testVCRNet`` (model/vcrnet_model.py:768-815), i.e. ``test_one_epoch`` (:546-649) over a
test set, then the ``==FINAL TEST==`` / ``A--------->B`` log lines.

    python evaluate.py --items 64 --batch 16                       # whole mode, N=1024, iter 1
    python evaluate.py --partial --iters 3 --batch 24 --items 48   # BASELINE configs[2]
    python evaluate.py --model-path checkpoints/.../model.best.t7  # a checkpoint of the reference (main.py --model_path)
    python evaluate.py --emb-nn dgcnn --pointer identity --vcp-nn dist --cycle   # the other constructor options
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 evaluate.py --items 1024

One process per GPU; items are sharded contiguously over ranks (vcrnet_amd.shard), the per-rank metric sums are
merged with one all-reduce + one all-gather (EvalAccumulator.merge).  ModelNet40 and the trained VCR-Net weights
are not available offline: clouds are synthetic and the Transformer weights are seeded, so the
absolute rot/trans errors are NOT the paper's -- only their agreement with the reference arithmetic is meaningful."""
import argparse
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main(argv=None):
    """Returns {"ab": A->B metrics, "ba": B->A metrics, "pairs": n, "returns": test_one_epoch's 17-tuple} on rank 0
    (None elsewhere)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=64, help="test-set size (pairs)")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--iters", type=int, default=1, help="--iter of the reference (0 = ICP refinement path)")
    ap.add_argument("--partial", action="store_true")
    ap.add_argument("--cycle", action="store_true", help="args.cycle: second head for (R_ba, t_ba) + the B->A log line")
    ap.add_argument("--first-item", type=int, default=0)
    ap.add_argument("--loss", default="pose", help="args.loss of the reference: pose | point | anything else = pose + 0.1 point")
    ap.add_argument("--vcp-nn", default="topK", choices=("topK", "att", "dist"))
    ap.add_argument("--emb-nn", default="lpdnet", choices=("lpdnet", "dgcnn", "pointnet"))
    ap.add_argument("--pointer", default="transformer", choices=("transformer", "identity", "none"))
    ap.add_argument("--n-blocks", type=int, default=1, help="Transformer blocks (args.n_blocks; > 1 runs layer by layer)")
    ap.add_argument("--model-path", default=None,
                    help="a checkpoint of the reference (torch.save(net.state_dict()), with or without the DataParallel "
                         "'module.' prefix), loaded with strict=False like util/initPara.py:254; default: the seeded weights")
    ap.add_argument("--backend", default="nccl")
    a = ap.parse_args(argv)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(a.backend, **({"device_id": dev} if a.backend == "nccl" else {}))

    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import evalmetrics, shard, synth, weights
    from vcrnet_amd.module import VCRNet, vcrnetIcpNet, vcrnetIter

    args = SimpleNamespace(emb_dims=512, cycle=a.cycle, emb_nn=a.emb_nn, pointer=a.pointer, vcp_nn=a.vcp_nn,
                           partial=a.partial, overlap2=synth.OVERLAP2_0575 if a.partial else 0.75, t3d=False, tfea=False,
                           n_blocks=a.n_blocks, dropout=0.0, ff_dims=1024, n_heads=4, max_iterations=50)
    w = weights.generate_weights(1234, lpd=weights.load_lpd_fixture(), vcp_nn=a.vcp_nn, emb_nn=a.emb_nn, pointer=a.pointer,
                                 n_blocks=a.n_blocks)
    net = VCRNet(args)
    net.load_state_dict(w)
    if a.model_path:                                                            # util/initPara.py:248-254
        if not os.path.exists(a.model_path):
            raise FileNotFoundError(f"can't find pretrained model {a.model_path}")
        res = net.load_state_dict(torch.load(a.model_path, map_location="cpu"), strict=False)
        if rank == 0:
            print(f"load pretrained model {a.model_path}: {len(res.missing_keys)} missing, {len(res.unexpected_keys)} unexpected keys")
    if a.emb_nn != "pointnet":
        net.emb_nn.k = a.k
    net = net.to(dev).eval()

    lo, hi = shard.shard_range(a.items, rank, world)
    lo, hi = lo + a.first_item, hi + a.first_item
    acc = evalmetrics.EvalAccumulator(cycle=a.cycle, loss=a.loss)
    kind = "object" if a.points <= 2048 else "uniform"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # The reference's DataLoader prepares the next batch in worker processes while the GPU works (util/data.py, main.py);
    # here the host-side part of a batch (the synthetic base clouds, ~0.35 ms per item) is drawn right after the previous
    # batch's forward has been ENQUEUED -- the device loop never waits for the host -- and the metrics, which read results
    # back, come last.  Same batches, same order, same figures.
    def make(first):
        n = min(a.batch, hi - first)
        src, tgt, R, t, eul = synth.make_batch_device(first, n, a.points, partial=a.partial, kind=kind, device=dev)
        return src, tgt, torch.from_numpy(R).to(dev), torch.from_numpy(t).to(dev), eul
    firsts = list(range(lo, hi, a.batch))
    nxt = make(firsts[0]) if firsts else None
    for i in range(len(firsts)):
        src, tgt, Rg, tg, eul = nxt
        with torch.no_grad():                                                   # vcrnet_model.py:546-562
            out = vcrnetIcpNet(args, net, src, tgt) if a.iters == 0 else vcrnetIter(net, src, tgt, iter=a.iters)
        nxt = make(firsts[i + 1]) if i + 1 < len(firsts) else None
        acc.add_batch(src, tgt, Rg, tg, eul, out)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    merged = acc.merge(world, device=dev if a.backend == "nccl" else "cpu")
    result = None
    if rank == 0:
        m, mb = merged.final(), merged.final_ba()
        print("==FINAL TEST==")                                                 # vcrnet_model.py:792-799
        print("A--------->B")
        print(evalmetrics.EvalAccumulator.format_final(m))
        if a.cycle:                                                             # :800-806
            print("B--------->A")
            print(evalmetrics.EvalAccumulator.format_final_ba(mb))
        print(f"[{merged.num_examples} pairs on {world} GPU(s), {elapsed:.2f} s incl. pair construction and metrics]",
              flush=True)
        result = {"ab": m, "ba": mb, "pairs": merged.num_examples, "returns": merged.returns()}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
