#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference); the GPU box never sees the
reference, only the .npz files this script writes.  Usage:  python tests/golden/gen_golden.py

What it does
  * injects a stub ``pynvml`` (util/util.py:9-16 calls NVML at import; absent on AMD),
  * imports ``model.vcrnet_model`` / ``model.dcp_model`` from /root/reference unmodified,
  * re-saves pretrained/lpd-pretrained.t7 as ``vcr-net_amd/data/lpd_pretrained.npz`` (12 fp32 tensors: data),
  * builds each variant, loads ``vcrnet_amd.weights.generate_weights(seed=1234, lpd=...)`` into it,
  * feeds ``vcrnet_amd.synth.make_batch`` inputs and records outputs + intermediates:
    every ``Tensor.topk`` result in call order (kNN sets, key-pruning sets, overlap / pair
    selections), sub-module outputs via forward hooks, H, R, t, composed poses per iteration.
Large activations are stored as strided slices (the stride is stored next to them).
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

stub = types.ModuleType("pynvml")
stub.nvmlInit = lambda: None
stub.nvmlDeviceGetHandleByIndex = lambda i: i
stub.nvmlDeviceGetMemoryInfo = lambda h: SimpleNamespace(used=0)
sys.modules["pynvml"] = stub
sys.path.insert(0, REF)

import model.vcrnet_model as ref_vcr            # noqa: E402
import model.dcp_model as ref_dcp               # noqa: E402
import vcrnet_amd                               # noqa: E402,F401
from vcrnet_amd import synth, weights           # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def ref_args(**kw):
    a = dict(emb_dims=512, cycle=False, emb_nn="lpdnet", pointer="transformer", vcp_nn="topK",
             partial=False, overlap2=0.75, t3d=False, tfea=False, n_blocks=1, dropout=0.0,
             ff_dims=1024, n_heads=4, head="svd", use_mFea=False)
    a.update(kw)
    return SimpleNamespace(**a)


class TopkLog:
    """Record every Tensor.topk the reference performs, in call order."""

    def __init__(self):
        self.calls = []

    def __enter__(self):
        self._orig = torch.Tensor.topk
        log = self

        def rec(t, *a, **kw):
            out = log._orig(t, *a, **kw)
            log.calls.append((out[0].detach().clone(), out[1].detach().clone()))
            return out

        torch.Tensor.topk = rec
        return self

    def __exit__(self, *exc):
        torch.Tensor.topk = self._orig


def hook_outputs(mods):
    store = {name: [] for name in mods}
    handles = []
    for name, m in mods.items():
        handles.append(m.register_forward_hook(
            lambda mod, inp, out, name=name: store[name].append((inp, out))))
    return store, handles


def sl(t, cs=1, ps=1):
    """channels-first [B,C,N] strided slice as float32 numpy."""
    return t[:, ::cs, ::ps].contiguous().numpy().astype(np.float32)


def load_into(net, w):
    res = net.load_state_dict(w, strict=False)
    bad = [k for k in res.missing_keys if "num_batches_tracked" not in k]
    assert not bad and not res.unexpected_keys, (bad, res.unexpected_keys)
    net.eval()


def run_vcrnet(name, B, N, first_item, cstride, emb_nn="lpdnet", vcp_nn="topK", partial=False,
               cycle=False, k=None, iters=1, pointer="transformer", kind="object", n_blocks=1, regime="default"):
    overlap2 = synth.OVERLAP2_0575 if partial else 0.75
    args = ref_args(emb_nn=emb_nn, vcp_nn=vcp_nn, partial=partial, overlap2=overlap2, cycle=cycle,
                    pointer=pointer, n_blocks=n_blocks)
    net = ref_vcr.VCRNet(args)
    # regime "default" = generate_weights(1234, lpd=LPD, ...): rounds 1-3 fixtures are unchanged by the weights axis
    w = weights.regime_weights(regime, lpd=LPD, emb_nn=emb_nn, vcp_nn=vcp_nn, pointer=pointer, n_blocks=n_blocks)
    load_into(net, w)
    if k is not None:
        net.emb_nn.k = k
    twin = None
    if regime != "default":      # the reference's own float64 twin (same code, weights, inputs): its distance is the
        import copy              # spread the fp32 reference shows against itself under this weight regime
        twin = copy.deepcopy(net).double()
    src, tgt, R_gt, t_gt, eul = synth.make_batch(first_item, B, N, partial=partial, kind=kind)
    src_t, tgt_t = torch.from_numpy(src), torch.from_numpy(tgt)
    out = dict(src=src, tgt=tgt, R_gt=R_gt, t_gt=t_gt, euler_gt=eul, cstride=np.int32(cstride),
               overlap2=np.float64(overlap2), k=np.int32(k or 20), iters=np.int32(iters), regime=np.str_(regime))
    mods = {"emb": net.emb_nn, "head": net.head, "svd": net.svd}
    if pointer == "transformer":
        mods["pointer"] = net.pointer
        mods["encoder"] = net.pointer.model.encoder
    if emb_nn == "lpdnet":
        mods.update(conv2=net.emb_nn.conv2_lpd, dg1=net.emb_nn.convDG1, dg2=net.emb_nn.convDG2,
                    sn1=net.emb_nn.convSN1)
    cur = src_t
    R_f = t_f = None
    with torch.no_grad():
        for it in range(iters):
            store, handles = hook_outputs(mods)
            with TopkLog() as tl:
                srcK, corrK, R, t, R_ba, t_ba = net(cur, tgt_t)
            for h in handles:
                h.remove()
            p = f"it{it}_"
            out[p + "in"] = cur.numpy().copy()
            if twin is not None:
                with TopkLog() as tl64:
                    o64 = twin(cur.double(), tgt_t.double())
                out[p + "R_f64"], out[p + "t_f64"] = o64[2].numpy(), o64[3].numpy()
                if pointer == "transformer":   # how peaked the soft-maxes are under this regime: mean over queries of the
                    pa = net.pointer.model.decoder.layers[0].src_attn.attn / 4     # largest head-averaged cross-attention
                    out[p + "peak_cross_attn"] = np.float32(pa.max(-1)[0].mean())  # probability (uniform = 1 / keys)
            out[p + "R"], out[p + "t"] = R.numpy(), t.numpy()
            out[p + "R_ba"], out[p + "t_ba"] = R_ba.numpy(), t_ba.numpy()
            out[p + "srcK"], out[p + "corrK"] = srcK.numpy(), corrK.numpy()
            # topk log -> named index sets
            calls = list(tl.calls)
            nk = {"lpdnet": 2, "dgcnn": 1, "pointnet": 0}[emb_nn]            # kNN searches per cloud
            names = []
            for cloud in ("src", "tgt"):
                names += [f"idx_feat_{cloud}", f"idx_xyz_{cloud}"] if nk == 2 else [f"idx_xyz_{cloud}"] if nk == 1 else []
            for nm in names:
                v, i = calls.pop(0)
                out[p + nm] = i[:, :, 1:].numpy().astype(np.int16)     # rank 0 dropped (util.py:159)
            if partial and pointer == "transformer":
                for nm in ("keep_dir_src", "keep_dir_tgt"):             # model(src,tgt) then model(tgt,src)
                    for layer in range(n_blocks):                       # every decoder layer prunes its own keys
                        v, i = calls.pop(0)
                        out[p + nm + (f"_l{layer}" if n_blocks > 1 else "")] = i.reshape(B, -1).numpy().astype(np.int16)
            if partial and vcp_nn == "topK":
                for nm in ("sel_tgt", "sel_src"):
                    v, i = calls.pop(0)
                    out[p + nm] = i.reshape(B, -1).numpy().astype(np.int16)
                v, i = calls.pop(0); out[p + "argmax_tgt"] = i.reshape(B, -1).numpy().astype(np.int16)
                v, i = calls.pop(0); out[p + "argmax_val"] = v.reshape(B, -1).numpy()
                v, i = calls.pop(0); out[p + "pair_src"] = i.reshape(B, -1).numpy().astype(np.int16)
                if twin is not None and pointer == "transformer" and n_blocks == 1:
                    # the float64 twin's selections on the same input vs the fp32 run's: (kept keys, overlap sets, hard pairs)
                    # flipped, counted as tests/test_hip_forced.py:count_flips does
                    c64 = [i.reshape(B, -1).numpy() for _, i in tl64.calls[2 * nk:]]
                    t64 = dict(keep_dir_src=c64[0], keep_dir_tgt=c64[1], sel_tgt=c64[2], sel_src=c64[3], argmax_tgt=c64[4],
                               pair_src=c64[6])
                    sd = lambda a, b: sum(len(set(x) ^ set(y)) // 2 for x, y in zip(a, b))
                    prs = lambda d: {(b, int(d["sel_src"][b, i]), int(d["sel_tgt"][b, d["argmax_tgt"][b, i]]))
                                     for b in range(B) for i in d["pair_src"][b]}
                    f32 = {k_: out[p + k_] for k_ in t64}
                    out[p + "twin_flips"] = np.array(
                        [sd(f32["keep_dir_src"], t64["keep_dir_src"]) + sd(f32["keep_dir_tgt"], t64["keep_dir_tgt"]),
                         sd(f32["sel_src"], t64["sel_src"]) + sd(f32["sel_tgt"], t64["sel_tgt"]),
                         len(prs(f32) ^ prs(t64)) // 2], dtype=np.int32)
            assert cycle or not calls, len(calls)
            # sub-module outputs
            e_src, e_tgt = store["emb"][0][1], store["emb"][1][1]
            out[p + "emb0_src"], out[p + "emb0_tgt"] = sl(e_src, cstride), sl(e_tgt, cstride)
            if emb_nn == "lpdnet":
                act = lambda t_: torch.nn.functional.leaky_relu(t_, 0.0)
                for ci, cloud in enumerate(("src", "tgt")):
                    out[p + f"x64_{cloud}"] = sl(act(store["conv2"][ci][1]), max(1, cstride // 2))
                    out[p + f"x1_{cloud}"] = sl(store["dg1"][ci][1].max(dim=-1)[0], cstride)
                    out[p + f"x2_{cloud}"] = sl(store["dg2"][ci][1].max(dim=-1)[0], cstride)
                    out[p + f"x3_{cloud}"] = sl(store["sn1"][ci][1].max(dim=-1)[0], cstride)
            if pointer == "transformer":
                sp, tp = store["pointer"][0][1]
                out[p + "ptr_src"], out[p + "ptr_tgt"] = sl(sp, cstride), sl(tp, cstride)
                # encoder outputs are [B,N,E]: first call encodes src, second encodes tgt
                out[p + "mem_src"] = sl(store["encoder"][0][1].transpose(2, 1), cstride)
                out[p + "mem_tgt"] = sl(store["encoder"][1][1].transpose(2, 1), cstride)
            h_in = store["head"][0][0]
            out[p + "femb_src"], out[p + "femb_tgt"] = sl(h_in[0], cstride), sl(h_in[1], cstride)
            s_in = store["svd"][0][0]
            sc = s_in[0] - s_in[0].mean(dim=2, keepdim=True)
            cc = s_in[1] - s_in[1].mean(dim=2, keepdim=True)
            out[p + "H"] = torch.matmul(sc, cc.transpose(2, 1).contiguous()).numpy()
            # vcrnetIter composition (model/vcrnet_model.py:28-38)
            cur = ref_vcr.transform_point_cloud(cur, R, t)
            if R_f is None:
                R_f, t_f = R, t
            else:
                R_f, t_f = torch.matmul(R, R_f), torch.matmul(R, t_f.unsqueeze(2)).squeeze(2) + t
        out["R_final"], out["t_final"] = R_f.numpy(), t_f.numpy()
        if iters > 1:   # free-running reference vcrnetIter for the aggregate check
            fr = ref_vcr.vcrnetIter(net, src_t, tgt_t, iter=iters)
            out["R_iter_free"], out["t_iter_free"] = fr[2].numpy(), fr[3].numpy()
            assert np.array_equal(out["R_iter_free"], out["R_final"])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: {sum(v.nbytes for v in out.values()) / 1e6:.2f} MB raw, R[0]=\n{out['it0_R'][0]}")


def run_dcp(name, B, N, first_item, cstride, emb_nn="lpdnet"):
    args = ref_args(emb_nn=emb_nn)
    net = ref_dcp.DCP(args)
    w = weights.generate_weights(1234, lpd=LPD, emb_nn=emb_nn)
    w = {k: v for k, v in w.items() if not k.startswith("svd.")}
    w["head.reflect"] = torch.diag(torch.tensor([1.0, 1.0, -1.0]))
    load_into(net, w)
    src, tgt, R_gt, t_gt, eul = synth.make_batch(first_item, B, N)
    with torch.no_grad():
        R, t, R_ba, t_ba, s, corr = net(torch.from_numpy(src), torch.from_numpy(tgt))
    out = dict(src=src, tgt=tgt, R=R.numpy(), t=t.numpy(), R_ba=R_ba.numpy(), t_ba=t_ba.numpy(),
               corr=corr.numpy(), cstride=np.int32(cstride))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: done")


def run_icp(name, B, N, first_item):
    """--iter 0 path: VCRNet pass, then the reference ICP (model/icp_model.py) on the moved source,
    composed as vcrnetIcpNet does (model/vcrnet_model.py:46-62; its .cuda() calls are bypassed)."""
    import model.icp_model as ref_icp
    args = ref_args()
    net = ref_vcr.VCRNet(args)
    load_into(net, weights.generate_weights(1234, lpd=LPD))
    src, tgt, R_gt, t_gt, eul = synth.make_batch(first_item, B, N)
    s, t = torch.from_numpy(src), torch.from_numpy(tgt)
    icp = ref_icp.ICP(max_iterations=50)
    calls = []
    orig = icp.nearest_neighbor
    icp.nearest_neighbor = lambda a, b: (calls.append(1), orig(a, b))[1]
    with torch.no_grad():
        _, _, R, tt, _, _ = net(s, t)
        moved = ref_vcr.transform_point_cloud(s, R, tt)
        _, final, Ri, ti, Rib, tib = icp(moved, t)
        R2 = torch.matmul(Ri, R)
        t2 = torch.matmul(Ri, tt.unsqueeze(2)).squeeze(2) + ti
    out = dict(src=src, tgt=tgt, moved=moved.numpy(), icp_final=final.numpy(), R_icp=Ri.numpy(), t_icp=ti.numpy(),
               R=R2.numpy(), t=t2.numpy(), iterations=np.int32(len(calls)), R_net=R.numpy(), t_net=tt.numpy())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: {len(calls)} ICP iterations")


CASES = {
    "whole_n256_b2": lambda n: run_vcrnet(n, B=2, N=256, first_item=0, cstride=4),
    "whole_n1024_b2": lambda n: run_vcrnet(n, B=2, N=1024, first_item=10, cstride=16),
    "whole_k40_n512_b1": lambda n: run_vcrnet(n, B=1, N=512, first_item=20, cstride=16, k=40),
    "cycle_n256_b2": lambda n: run_vcrnet(n, B=2, N=256, first_item=30, cstride=16, cycle=True),
    "partial_n192_b2_it2": lambda n: run_vcrnet(n, B=2, N=256, first_item=40, cstride=8, partial=True, iters=2),
    "partial_n768_b2_it3": lambda n: run_vcrnet(n, B=2, N=1024, first_item=50, cstride=32, partial=True, iters=3),
    "dgcnn_n256_b2": lambda n: run_vcrnet(n, B=2, N=256, first_item=60, cstride=8, emb_nn="dgcnn"),
    "att_n256_b2": lambda n: run_vcrnet(n, B=2, N=256, first_item=70, cstride=16, vcp_nn="att"),
    "dist_n256_b2": lambda n: run_vcrnet(n, B=2, N=256, first_item=80, cstride=16, vcp_nn="dist"),
    "identity_n256_b2": lambda n: run_vcrnet(n, B=2, N=256, first_item=90, cstride=16, pointer="identity"),
    "dcp_n256_b2": lambda n: run_dcp(n, B=2, N=256, first_item=100, cstride=16),
    "icp_n256_b2": lambda n: run_icp(n, B=2, N=256, first_item=110),
    # round 2
    "attcycle_n256_b2": lambda n: run_vcrnet(n, B=2, N=256, first_item=120, cstride=16, vcp_nn="att", cycle=True),
    "distcycle_n256_b2": lambda n: run_vcrnet(n, B=2, N=256, first_item=130, cstride=16, vcp_nn="dist", cycle=True),
    # BASELINE configs[4] shape: uniform clouds, N=4096, k=40 (LPDNet.k override, lpdnet_model.py:81)
    "whole_k40_n4096_b2": lambda n: run_vcrnet(n, B=2, N=4096, first_item=140, cstride=128, k=40, kind="uniform"),
    "dgcnn_partial_n192_b2": lambda n: run_vcrnet(n, B=2, N=256, first_item=150, cstride=16, emb_nn="dgcnn",
                                                   partial=True, iters=1),
    # round 3: the third emb_nn of VCRNet.__init__ (model/vcrnet_model.py:468-469)
    "pointnet_n256_b2": lambda n: run_vcrnet(n, B=2, N=256, first_item=160, cstride=16, emb_nn="pointnet"),
    "pointnet_partial_n192_b2_it2": lambda n: run_vcrnet(n, B=2, N=256, first_item=170, cstride=16, emb_nn="pointnet",
                                                          partial=True, iters=2),
    # --n_blocks 2: two encoder and two decoder layers (model/transformer.py:245,257-259)
    "nblocks2_n256_b2": lambda n: run_vcrnet(n, B=2, N=256, first_item=180, cstride=16, n_blocks=2),
    "nblocks2_partial_n192_b2": lambda n: run_vcrnet(n, B=2, N=256, first_item=190, cstride=16, partial=True, n_blocks=2),
}


# round 4: the weights axis (vcrnet_amd.weights.regime_weights) -- the three shapes the HIP path is held to
# (whole N = 1024, k = 40, partial N = 768 with three teacher-forced passes and every discrete selection) under a
# second seed, a trained-like regime (peaky soft-maxes, large LayerNorm offsets) and a random feature extractor
for _r, _base in (("seed4321", 600), ("trained", 700), ("randemb", 800)):
    CASES[f"{_r}_whole_n1024_b2"] = lambda n, r=_r, f=_base: run_vcrnet(n, B=2, N=1024, first_item=f, cstride=32, regime=r)
    CASES[f"{_r}_whole_k40_n512_b1"] = lambda n, r=_r, f=_base: run_vcrnet(n, B=1, N=512, first_item=f + 10, cstride=32,
                                                                           k=40, regime=r)
    CASES[f"{_r}_partial_n768_b2_it3"] = lambda n, r=_r, f=_base: run_vcrnet(n, B=2, N=1024, first_item=f + 20, cstride=64,
                                                                             partial=True, iters=3, regime=r)


# ... and the other constructor branches under the trained-like regime (small clouds): DGCNN, the VcpAtt head, cycle
CASES["trained_dgcnn_n256_b2"] = lambda n: run_vcrnet(n, B=2, N=256, first_item=900, cstride=32, emb_nn="dgcnn", regime="trained")
CASES["trained_att_n256_b2"] = lambda n: run_vcrnet(n, B=2, N=256, first_item=910, cstride=32, vcp_nn="att", regime="trained")
CASES["trained_cycle_n256_b2"] = lambda n: run_vcrnet(n, B=2, N=256, first_item=920, cstride=32, cycle=True, regime="trained")


if __name__ == "__main__":
    # usage: gen_golden.py [case ...]   (no arguments = every case)
    sd = torch.load(os.path.join(REF, "pretrained", "lpd-pretrained.t7"), map_location="cpu")
    if len(sys.argv) == 1:
        np.savez(weights.LPD_FIXTURE, **{k: v.numpy() for k, v in sd.items()})
    LPD = weights.load_lpd_fixture()
    for name in (sys.argv[1:] or list(CASES)):
        CASES[name](name)
