"""The drop-in boundary from a host that is NOT Python: examples/host_cpp/forward_host.cpp links libvcr_hip.so, allocates
with hipMalloc, creates its own stream and calls vcr_vcrnet_forward_f32 / vcr_vcrnet_iter_f32 on a flat export of a module's
packed weights (vcrnet_amd.export_blob) -- no PyTorch, no Python in that process.  Its poses and correspondences must equal the
module's own, bit for bit (same library, same launches, other allocator and stream)."""
import os
import subprocess

import numpy as np
import pytest
import torch

from test_hip_forward import build_net

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw,mode,iters", [({}, "fp32", 1), ({}, "bf16x3+sdpa", 1), (dict(partial=True), "fp32", 3),
                                           (dict(emb_nn="dgcnn"), "fp32", 1), (dict(vcp_nn="att", cycle=True), "fp32", 1)])
def test_cpp_host_without_python_gets_the_modules_bits(tmp_path, kw, mode, iters):
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import build, export_blob, synth
    exe = build.build_host_example()
    net, _ = build_net(**kw)
    net.linear_mode = mode
    partial = bool(kw.get("partial"))
    src, tgt, _, _, _ = synth.make_batch(91, 2, 320, partial=partial)
    s, t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    with torch.no_grad():
        ref = net._forward_fused(s, t, iters=iters, iter_api=iters > 1)
    torch.cuda.synchronize()
    blob, out = str(tmp_path / "model.blob"), str(tmp_path / "out.bin")
    export_blob.write_blob(net, s, t, blob)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, blob, out, str(iters)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    print(r.stdout.strip())
    got = np.fromfile(out, dtype=np.float32)
    B = s.shape[0]
    R, tt, Rb, tb = (x.cpu().numpy() for x in ref[2:6])
    pose = np.concatenate((R.reshape(-1), tt.reshape(-1), Rb.reshape(-1), tb.reshape(-1)))
    assert np.array_equal(got[:B * 24], pose), np.abs(got[:B * 24] - pose).max()
    K = ref[1].shape[2]
    kk = min(K, 8)
    corr = ref[1].cpu().numpy()                              # [B, 3, K]
    got_c = got[B * 24:].reshape(B, kk, 4)[:, :, :3]
    assert np.array_equal(got_c, corr[:, :, :kk].transpose(0, 2, 1))
