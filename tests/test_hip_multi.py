"""The multi-GPU path on the hardware a 1-GPU box offers (replaces nn.DataParallel,
util/initPara.py:260):
  * RCCL itself initialises next to libvcr_hip.so (two HIP runtimes in one image: torch's bundled ROCm and the system
    one hipcc links) and shard.all_gather_poses runs on cuda:0 through backend "nccl" -- world size 1 in this process;
  * `python bench.py --gpus 2` with no launcher spawns its own ranks (two processes sharing the one GPU, gloo for the
    rendezvous) and rank 0 prints the n_gpus = 2 line;
  * evaluate.main, the counterpart of main.py --eval -> testVCRNet, over a sharded test set, against the CPU oracle's
    metrics."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle
from helpers import cfg_weights

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_world1_all_gather_next_to_libvcr_hip():
    import torch.distributed as dist
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import native, shard, synth
    from test_hip_forward import build_net
    native.lib()                                             # libvcr_hip.so is mapped BEFORE the communicator opens
    net, _ = build_net()
    src, tgt, _, _, _ = synth.make_batch(10, 2, 128)
    s, t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    with torch.no_grad():
        out = net(s, t)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        pose = shard.pack_pose(out[2], out[3])
        # world == 1 short-circuits in all_gather_poses; drive the collective itself
        got = torch.empty_like(pose)
        dist.all_gather_into_tensor(got, pose.contiguous())
        assert torch.equal(got, pose)
        tt = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        assert float(tt.item()) == 1.5
        dist.barrier()
        with torch.no_grad():                                # and the HIP path still runs after RCCL came up
            again = net(s, t)
        assert torch.equal(again[2], out[2])
    finally:
        dist.destroy_process_group()


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without torch.distributed.run: the parent spawns the two rank processes before it
    touches the GPU.  On this 1-GPU box both ranks share cuda:0 and rendezvous over gloo (--backend gloo); with 2+ GPUs
    the same command line without --backend is the RCCL run the driver's scaling sweep performs."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3",
                        "--warmup", "1", "--batch", "4", "--points", "256", "--min-seconds", "0.2"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, r.stdout                          # stdout = the ONE JSON line (library chatter goes to stderr)
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 8 and j["scaling"] == "weak"
    assert j["value"] > 0 and "roofline" in j and "cpu_baseline" not in j
    assert j["timed_blocks"]["count"] >= 1 and j["timed_blocks"]["steps_per_block"] == 3


def test_bench_kills_the_run_when_a_rank_dies_after_rendezvous():
    """Rank 1 exits right after init_process_group (injected); rank 0 is by then blocked in its first collective.  The
    parent notices the dead rank, kills rank 0 by PID and reports: rc != 0 within seconds, one JSON line with `error`."""
    import tempfile
    import time
    env = dict(os.environ, VCR_BENCH_FAIL_RANK="1", VCR_BENCH_LOGDIR=tempfile.mkdtemp(prefix="vcr_bench_logs_"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3",
                        "--warmup", "1", "--batch", "4", "--points", "256", "--min-seconds", "0.2", "--deadline-s", "300"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and time.time() - t0 < 240, (r.returncode, time.time() - t0)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["value"] is None and "rank 1 exited with code 3" in j["error"]
    assert "injected failure" in j["rank_stderr_tail"]["1"]


def _bench_env(**extra):
    env = dict(os.environ, **extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_bench_eight_ranks_on_the_one_gpu_explains_itself():
    """`python bench.py --gpus 8` as the driver's scaling sweep launches it, with the only substitution a 1-GPU lease
    allows (--backend gloo: eight fresh rank processes sharing cuda:0): one JSON line whose `multi_gpu` object carries what
    a first real 8-GPU run needs to be read -- world size, backend, every rank's own step time, the collective's time,
    the devices visible -- and per-rank logs."""
    import tempfile
    from test_shard_gloo import RCCL_LOG_XGMI
    logdir = tempfile.mkdtemp(prefix="vcr_bench_logs_")
    canned = os.path.join(tempfile.mkdtemp(prefix="vcr_rccl_log_"), "rank0.rccl.log")      # what an 8-GPU box's RCCL would write
    with open(canned, "w") as f:
        f.write(RCCL_LOG_XGMI)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--batch", "1",
                        "--points", "64", "--steps", "3", "--warmup", "1", "--min-seconds", "0.2", "--deadline-s", "500"],
                       capture_output=True, text=True, timeout=900, env=_bench_env(VCR_BENCH_LOGDIR=logdir, VCR_BENCH_RCCL_LOG=canned))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    m = j["multi_gpu"]
    assert j["n_gpus"] == 8 and j["config"]["global_batch"] == 8 and m["world_size"] == 8
    assert m["backend"] == "gloo" and m["rccl_version"] is None and "NOT RCCL" in j["config"]["parallelism"]
    assert len(m["per_rank_ms_per_step"]) == 8 and len(m["all_gather_ms_per_rank"]) == 8
    assert 0 < m["per_rank_ms_per_step_min"] <= m["per_rank_ms_per_step_max"] <= j["ms_per_step"] * 1.05
    assert m["all_gather_ms"] > 0 and m["n_devices_visible"] >= 1
    assert sorted(os.listdir(logdir)) == [f"rank{i}.err" for i in range(8)]
    assert "other_configs" not in j and "cpu_baseline" not in j
    # the transport evidence (here from the canned log: gloo writes none): parsed into the line
    assert m["xgmi_only"] is True and m["transport"]["transports"] == {"P2P/IPC": 2, "P2P/direct pointer": 1}


def test_bench_eight_ranks_rank_five_dies():
    import tempfile
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--batch", "1",
                        "--points", "64", "--steps", "3", "--warmup", "1", "--min-seconds", "0.2", "--deadline-s", "500"],
                       capture_output=True, text=True, timeout=900,
                       env=_bench_env(VCR_BENCH_FAIL_RANK="5", VCR_BENCH_LOGDIR=tempfile.mkdtemp(prefix="vcr_bench_logs_")))
    assert r.returncode != 0 and time.time() - t0 < 400, (r.returncode, time.time() - t0)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["value"] is None and j["n_gpus"] == 8 and "rank 5 exited with code 3" in j["error"]
    assert "injected failure" in j["rank_stderr_tail"]["5"]


def test_rccl_version_is_reported():
    sys.path.insert(0, ROOT)
    import bench
    v = bench.rccl_version()
    assert v and "unavailable" not in v and v[0].isdigit(), v


def test_bench_strong_scaling_shards_the_global_batch():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--batch", "8", "--strong", "--points", "256", "--min-seconds", "0.1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["scaling"] == "strong" and j["config"]["global_batch"] == 8 and j["config"]["batch_per_gpu"] == 4


def test_bench_under_torchrun_still_works():
    port = _free_port()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--points", "256", "--min-seconds", "0.1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 2 and j["steps"] == 2


def test_evaluate_main_matches_oracle_metrics(capsys):
    """evaluate.main over 32 synthetic items (two batches of 16, N = 256): the ==FINAL TEST== figures equal the ones
    the same accumulator computes from the CPU oracle's poses (whole mode: every pose within 1e-4 / 1e-5, so the
    aggregates agree to 1e-3 relative)."""
    sys.path.insert(0, ROOT)
    import evaluate
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import evalmetrics, synth
    res = evaluate.main(["--items", "32", "--batch", "16", "--points", "256", "--first-item", "6000", "--cycle"])
    text = capsys.readouterr().out
    assert "==FINAL TEST==" in text and "A--------->B" in text and "B--------->A" in text
    assert res["pairs"] == 32
    w = cfg_weights()
    ref = evalmetrics.EvalAccumulator(cycle=True)
    for first in (6000, 6016):
        src, tgt, R, t, eul = synth.make_batch(first, 16, 256)
        s, tt, Rg, tg = (torch.from_numpy(x) for x in (src, tgt, R, t))
        ref.add_batch(s, tt, Rg, tg, eul, oracle.vcrnet_iter(w, s, tt, oracle.OracleConfig(cycle=True), iters=1))
    ma, mb = ref.final(), ref.final_ba()
    for key in ("loss", "mse", "mae", "rot_mse", "rot_mae", "trans_mse", "trans_mae"):
        assert abs(res["ab"][key] - ma[key]) <= 1e-3 * abs(ma[key]) + 1e-9, ("ab", key, res["ab"][key], ma[key])
    for key in ("mse", "rot_mse", "rot_mae", "trans_mse", "trans_mae"):
        assert abs(res["ba"][key] - mb[key]) <= 1e-3 * abs(mb[key]) + 1e-9, ("ba", key, res["ba"][key], mb[key])
    line = [ln for ln in text.splitlines() if ln.startswith("EPOCH:: -1")][0]
    assert ("rot_MSE: %f" % res["ab"]["rot_mse"]) in line


def test_evaluate_loads_a_reference_checkpoint(tmp_path, capsys):
    """evaluate.py --model-path: a checkpoint as the reference writes it (torch.save of a DataParallel state dict, keys
    prefixed 'module.') is loaded with strict=False (util/initPara.py:248-254) and is what the run evaluates -- the
    figures equal those of a module loaded with the same tensors directly; and the other constructor options of the
    reference's command line (--emb-nn / --pointer / --n-blocks) reach the module."""
    sys.path.insert(0, ROOT)
    import evaluate
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import evalmetrics, synth, weights
    from vcrnet_amd.module import vcrnetIter
    from test_hip_forward import make_args
    from vcrnet_amd.module import VCRNet
    w2 = weights.generate_weights(4321, lpd=weights.load_lpd_fixture())
    path = str(tmp_path / "model.best.t7")
    torch.save({"module." + k: v for k, v in w2.items()}, path)
    argv = ["--items", "8", "--batch", "8", "--points", "256", "--first-item", "7000"]
    r0 = evaluate.main(argv)
    r1 = evaluate.main(argv + ["--model-path", path])
    assert "load pretrained model" in capsys.readouterr().out
    assert r0["ab"]["rot_mse"] != r1["ab"]["rot_mse"]
    net = VCRNet(make_args())
    net.load_state_dict(w2)
    net = net.cuda().eval()
    acc = evalmetrics.EvalAccumulator()
    src, tgt, R, t, eul = synth.make_batch_device(7000, 8, 256, device=torch.device("cuda"))
    with torch.no_grad():
        acc.add_batch(src, tgt, torch.from_numpy(R).cuda(), torch.from_numpy(t).cuda(), eul, vcrnetIter(net, src, tgt, iter=1))
    assert acc.final() == r1["ab"]
    with pytest.raises(FileNotFoundError):
        evaluate.main(argv + ["--model-path", path + ".missing"])
    for extra in (["--emb-nn", "pointnet"], ["--emb-nn", "dgcnn", "--pointer", "identity"], ["--n-blocks", "2"],
                  ["--pointer", "none", "--vcp-nn", "dist"]):
        r = evaluate.main(argv + extra)
        assert r["pairs"] == 8 and np.isfinite(r["ab"]["rot_mse"]) and r["ab"]["rot_mse"] != r0["ab"]["rot_mse"], extra


def test_evaluate_sharded_two_ranks_equals_one():
    """Two evaluate.py ranks (sharing the GPU, gloo) print the same FINAL line as one process: contiguous shards +
    EvalAccumulator.merge (one all-reduce, one all-gather) lose nothing."""
    cmd = [os.path.join(ROOT, "evaluate.py"), "--items", "12", "--batch", "4", "--points", "256", "--first-item", "6100"]
    one = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + cmd + ["--backend", "gloo"],
                         capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    pick = lambda out: [ln for ln in out.splitlines() if ln.startswith("EPOCH:: -1")][0]
    a = np.array([float(x.split(":")[-1]) for x in pick(one.stdout).split(",")[1:]])
    b = np.array([float(x.split(":")[-1]) for x in pick(two.stdout).split(",")[1:]])
    np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-6)
