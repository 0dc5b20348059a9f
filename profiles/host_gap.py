#!/usr/bin/env python
"""Is a step of bench.py host-bound?  For one arithmetic mode: (1) the host time one `vcrnetIter` call takes to ENQUEUE
(perf_counter around the call, no synchronize, queue drained first every 4 steps so the host never blocks on a full queue),
(2) the device time per step of a 20-step block (HIP events at both ends, the bench's own protocol otherwise), (3) the same
block with the per-launch trace on every step: the sum of the traced launches.  (1) >= (2) means the host cannot keep the
queue fed; (3) < (2) with (1) < (2) means time between launches or clocks that differ with the event records in place.

  python profiles/host_gap.py [--linear-mode bf16x3+sdpa] [--points 1024 --batch 16 --k 20] [--seconds 4]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--linear-mode", default="fp32")
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    import bench
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import native, synth, weights
    from vcrnet_amd.module import VCRNet, vcrnetIter
    dev = torch.device("cuda", 0)
    w = weights.generate_weights(1234, lpd=weights.load_lpd_fixture())
    net = VCRNet(bench.model_args())
    net.load_state_dict(w)
    net.emb_nn.k = a.k
    net.linear_mode = a.linear_mode
    net = net.to(dev).eval()
    kind = "object" if a.points < 2048 else "uniform"
    src, tgt, _, _, _ = synth.make_batch_device(0, a.batch, a.points, kind=kind, device=dev)
    B = a.batch

    def step(trace=None):
        net.launch_trace = trace
        with torch.no_grad():
            out = vcrnetIter(net, src, tgt, iter=1)
        return torch.cat((out[2].view(B, 9), out[3]), 1)

    step(); torch.cuda.synchronize()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    # (1) host enqueue time
    host = []
    for i in range(200):
        if i % 4 == 0:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    # (2) untraced blocks, (3) fully traced blocks, alternated
    res = {"untraced": [], "traced": [], "traced_sum": [], "wall_untraced": []}
    traces = [native.LaunchTrace() for _ in range(a.steps)]
    t_end = time.perf_counter() + a.seconds
    while time.perf_counter() < t_end:
        for mode in ("untraced", "traced"):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record()
            for i in range(a.steps):
                step(traces[i].trace if mode == "traced" else None)
            e1.record()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            res[mode].append(e0.elapsed_time(e1) / a.steps)
            if mode == "untraced":
                res["wall_untraced"].append(wall / a.steps * 1e3)
            else:
                per = [sum(ms for _, ms in tr.launches()) for tr in traces]
                res["traced_sum"].append(float(np.mean(per)))
    out = {"linear_mode": a.linear_mode, "points": a.points, "batch": B, "k": a.k,
           "host_enqueue_ms_per_step": {"median": float(np.median(host)) * 1e3, "p10": float(np.percentile(host, 10)) * 1e3,
                                        "p90": float(np.percentile(host, 90)) * 1e3},
           "device_ms_per_step_untraced": float(np.median(res["untraced"])),
           "wall_ms_per_step_untraced": float(np.median(res["wall_untraced"])),
           "device_ms_per_step_traced_every_step": float(np.median(res["traced"])),
           "traced_launch_sum_ms": float(np.median(res["traced_sum"])),
           "blocks": len(res["untraced"]),
           "untraced_first_last": [res["untraced"][0], res["untraced"][-1]]}
    out["accounted_frac"] = out["traced_launch_sum_ms"] / out["device_ms_per_step_untraced"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
