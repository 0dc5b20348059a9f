"""Host-side mirror of the reference's ``VCRNet`` nn.Module contract (model/vcrnet_model.py:463-518).

Same constructor (``VCRNet(args)``), same ``forward(src, tgt)`` 6-tuple, same ``state_dict`` key names
and the attributes its callers touch (util/initPara.py:38-65,254-260) -- so ``main.py``'s eval loop can
call it unchanged -- but ``forward`` runs the hand-written HIP path of ``libvcr_hip.so`` through the
C-ABI of ``include/vcr_hip.h``.  PyTorch here is plumbing only: device memory, the current stream and
the parameter containers.  There is no CPU / eager fallback: CPU tensors or a missing library raise.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import native
from .weights import strip_module_prefix

LINEAR_MODES = {"fp32": 0, "bf16x3": 1, "bf16x3+sdpa": 2}   # vcr_vcrnet_weights.linear_mode


_CALL = threading.local()


class _call_scope:
    """One parameter walk per call (VCRNet._tensors): nested scopes share the outermost one."""

    def __enter__(self):
        self.owner = getattr(_CALL, "tensors", None) is None
        if self.owner:
            _CALL.tensors = {}
        return self

    def __exit__(self, *exc):
        if self.owner:
            _CALL.tensors = None


class _Shared:
    """State that a module and its nn.DataParallel replicas share (replicas are shallow copies: ``__dict__.copy()``, so
    they all point at THIS object): the packed weights per device and the pool of forward workspaces.  ``nn.DataParallel``
    re-creates its replicas on every forward with freshly broadcast parameters (util/initPara.py:260 wraps the net in one),
    and calls them from one host thread per replica: the packed form is therefore keyed by (device, the MASTER's parameter
    versions), owns its memory, and every access goes through the lock."""

    def __init__(self, master):
        import threading
        import weakref
        self.lock = threading.RLock()
        self.master = weakref.ref(master)
        self.packed: Dict[torch.device, Tuple] = {}         # device -> (key, P, cw): the latest packing per device
        self.pool: Dict[Tuple, list] = {}                   # workspace key -> idle workspaces (dicts: "ws", "stream")
        self.packs = 0                                      # how many times weights were packed (tests)


# ---- parameter containers with the reference's module tree (names are API) -------------------------------

class _LPDNetParams(nn.Module):
    """Parameter tree of LPDNet (model/lpdnet_model.py:78-94).  Holds weights; compute is native."""

    def __init__(self, args, negative_slope: float = 0.0):
        super().__init__()
        self.negative_slope = negative_slope
        self.k = 20                                                        # lpdnet_model.py:81
        self.emb_dims = args.emb_dims
        if getattr(args, "t3d", False) or getattr(args, "tfea", False):
            raise Exception("Not implemented")                             # T-Nets are out of scope
        act = lambda: nn.LeakyReLU(negative_slope=negative_slope)
        self.convDG1 = nn.Sequential(nn.Conv2d(128, 128, kernel_size=1, bias=True), act())
        self.convDG2 = nn.Sequential(nn.Conv2d(128, 128, kernel_size=1, bias=True), act())
        self.convSN1 = nn.Sequential(nn.Conv2d(256, 256, kernel_size=1, bias=True), act())
        self.conv1_lpd = nn.Conv1d(3, 64, kernel_size=1, bias=True)
        self.conv2_lpd = nn.Conv1d(64, 64, kernel_size=1, bias=True)
        self.conv3_lpd = nn.Conv1d(512, self.emb_dims, kernel_size=1, bias=True)


class _DGCNNParams(nn.Module):
    """Parameter tree of DGCNN (model/vcrnet_model.py:91-102)."""

    def __init__(self, emb_dims: int = 512):
        super().__init__()
        self.k = 20
        chans = [(6, 64), (64, 64), (64, 128), (128, 256), (512, emb_dims)]
        for i, (ci, co) in enumerate(chans, 1):
            setattr(self, f"conv{i}", nn.Conv2d(ci, co, kernel_size=1, bias=False))
        for i, (_, co) in enumerate(chans, 1):
            setattr(self, f"bn{i}", nn.BatchNorm2d(co))


class _PointNetParams(nn.Module):
    """Parameter tree of PointNet (model/vcrnet_model.py:65-79)."""

    def __init__(self, emb_dims: int = 512):
        super().__init__()
        self.k = 1                                                          # (no graph: the driver ignores it)
        chans = [(3, 64), (64, 64), (64, 64), (64, 128), (128, emb_dims)]
        for i, (ci, co) in enumerate(chans, 1):
            setattr(self, f"conv{i}", nn.Conv1d(ci, co, kernel_size=1, bias=False))
        for i, (_, co) in enumerate(chans, 1):
            setattr(self, f"bn{i}", nn.BatchNorm1d(co))


class _Norm(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.a_2 = nn.Parameter(torch.ones(n))
        self.b_2 = nn.Parameter(torch.zeros(n))
        self.eps = 1e-6


class _Sublayer(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.norm = _Norm(n)


class _MHA(nn.Module):
    def __init__(self, h, d):
        super().__init__()
        self.h, self.d_k = h, d // h
        self.linears = nn.ModuleList([nn.Linear(d, d) for _ in range(4)])


class _FFN(nn.Module):
    def __init__(self, d, f):
        super().__init__()
        self.w_1 = nn.Linear(d, f)
        self.w_2 = nn.Linear(f, d)


class _EncLayer(nn.Module):
    def __init__(self, d, f, h):
        super().__init__()
        self.self_attn = _MHA(h, d)
        self.feed_forward = _FFN(d, f)
        self.sublayer = nn.ModuleList([_Sublayer(d) for _ in range(2)])
        self.size = d


class _DecLayer(nn.Module):
    def __init__(self, d, f, h):
        super().__init__()
        self.size = d
        self.self_attn = _MHA(h, d)
        self.src_attn = _MHA(h, d)
        self.feed_forward = _FFN(d, f)
        self.sublayer = nn.ModuleList([_Sublayer(d) for _ in range(3)])


class _Stack(nn.Module):
    def __init__(self, layers, d):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        self.norm = _Norm(d)


class _EncDec(nn.Module):
    def __init__(self, d, f, h, n):
        super().__init__()
        self.encoder = _Stack([_EncLayer(d, f, h) for _ in range(n)], d)
        self.decoder = _Stack([_DecLayer(d, f, h) for _ in range(n)], d)


class _TransformerParams(nn.Module):
    """Parameter tree of Transformer (model/transformer.py:241-262)."""

    def __init__(self, args):
        super().__init__()
        self.emb_dims, self.N, self.ff_dims = args.emb_dims, args.n_blocks, args.ff_dims
        self.n_heads, self.overlap2 = args.n_heads, args.overlap2
        self.model = _EncDec(args.emb_dims, args.ff_dims, args.n_heads, args.n_blocks)


class _Identity(nn.Module):
    pass


class _VcpTopK(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.emb_nn, self.partial, self.overlap2 = args.emb_nn, args.partial, args.overlap2


class _VcpByDis(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.emb_nn = args.emb_nn


class _VcpAtt(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.emb_dims = args.emb_dims
        self.linears_emb = nn.ModuleList([nn.Linear(args.emb_dims, args.emb_dims) for _ in range(2)])
        self.linears_3d = nn.ModuleList([nn.Linear(3, 3) for _ in range(2)])


class _SVDHead(nn.Module):
    def __init__(self):
        super().__init__()
        r = torch.eye(3)
        r[2, 2] = -1
        self.reflect = nn.Parameter(r, requires_grad=False)                # vcrnet_model.py:353-354


# ---- the module ----------------------------------------------------------------------------------------------

class VCRNet(nn.Module):
    """Drop-in for the reference's VCRNet (model/vcrnet_model.py:463-518); HIP compute."""

    def __init__(self, args):
        super().__init__()
        self.emb_dims = args.emb_dims
        self.cycle = args.cycle
        if args.emb_nn == "pointnet":
            self.emb_nn = _PointNetParams(emb_dims=self.emb_dims)
        elif args.emb_nn == "dgcnn":
            self.emb_nn = _DGCNNParams(emb_dims=self.emb_dims)
        elif args.emb_nn == "lpdnet":
            self.emb_nn = _LPDNetParams(args)
        else:
            raise Exception("Not implemented")                             # vcrnet_model.py:475
        if args.pointer == "identity":
            self.pointer = _Identity()
        elif args.pointer == "transformer":
            self.pointer = _TransformerParams(args)
        else:
            self.pointer = None
        if args.vcp_nn == "topK":
            self.head = _VcpTopK(args)
        elif args.vcp_nn == "att":
            self.head = _VcpAtt(args)
        elif args.vcp_nn == "dist":
            self.head = _VcpByDis(args)
        else:
            raise Exception("Not implemented")                             # vcrnet_model.py:491
        self.svd = _SVDHead()
        self._emb_kind, self._vcp = args.emb_nn, args.vcp_nn
        self._partial = bool(getattr(args, "partial", False))
        self._overlap2 = float(getattr(args, "overlap2", 0.75))            # sympy Float in the reference
        self._n_heads, self._ff = args.n_heads, args.ff_dims
        # "fp32": every linear on v_mfma_f32_32x32x2_f32 (default).  "bf16x3": the same products as exact 3-way bf16
        # splits on the bf16 matrix pipe (fp32-equivalent accuracy, ~1.5x faster linears); fused whole-forward only.
        # "bf16x3+sdpa": that, and the attention products (Q K^T, P V) the same way (vcr_sdpa_bf16x3_f32).
        self.linear_mode = os.environ.get("VCRNET_LINEAR_MODE", "fp32")
        # MFMA shape / k-slab of the fp32 linears, feature-space kNN kernel: 0 = the library's choice (benchmarks)
        self.linear_mfma, self.linear_bk, self.knn_waves = 0, 0, 0
        self.linear_bm = 0                  # tile height of the fp32 linears: 0 = the library's choice, 96 / 128 (benchmarks)
        # enc.qkv + dec.qkv as one GEMM and the two self-attentions as one grouped launch (fp32 mode; same arithmetic)
        self.merge_encdec = os.environ.get("VCRNET_MERGE_ENCDEC", "1") == "1"
        self.xscore_limit_mb = 0            # partial mode: keep the cross-attention scores up to this many MiB (0 = 4096)
        self.sdpa_variant = 0               # fp32 attention-output kernel: 0 = the library's choice, 1 = tile, 2 = persistent (benchmarks)
        self.iter_reuse = True              # vcrnetIter, iter > 1: later passes reuse what the first computed from the target alone
        self.workspace_flat = False         # tests: no two workspace buffers share memory (vcr_vcrnet_weights.workspace_flat)
        self._packed: Optional[Dict[str, torch.Tensor]] = None
        self._packed_key = None
        self._cw: Optional[native.VcrnetWeights] = None
        self._shared = _Shared(self)
        # profiling hook: a native.Trace that the next forward() / vcrnetIter() call records its per-launch HIP events
        # into (bench.py sets it on the steps it traces; None = no events)
        self.launch_trace: Optional[native.Trace] = None

    # -- checkpoints saved through nn.DataParallel carry a "module." prefix --
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        return super().load_state_dict(strip_module_prefix(state_dict), strict=strict, **kw)

    def __getstate__(self):
        # pickling / copy.deepcopy: the packed weights (raw device pointers) and the shared cache stay behind; the copy
        # packs for itself on its first call
        d = self.__dict__.copy()
        d.pop("_shared", None)
        d["_packed"], d["_packed_key"], d["_cw"] = None, None, None
        return d

    # -- weight packing: once per (device, parameter versions) --------------------------------------------------
    def _tensors(self) -> Dict[str, torch.Tensor]:
        """{state_dict key: tensor} of this module -- or of this nn.DataParallel REPLICA, whose parameters() is empty by
        design (torch/nn/parallel/replicate.py: the broadcast copies are plain attributes, listed in _former_parameters).
        Walked ONCE PER CALL: forward() / _forward_fused() open a thread-local scope (_call_scope) in which the walk is
        reused by the device check, the fingerprint and the launch; nothing outlives the call, so a Parameter object
        replaced between calls is seen."""
        scope = getattr(_CALL, "tensors", None)
        if scope is not None and id(self) in scope:
            return scope[id(self)]
        out: Dict[str, torch.Tensor] = {}
        for prefix, mod in self.named_modules():
            former = mod.__dict__.get("_former_parameters") if getattr(mod, "_is_replica", False) else None
            src = former if former is not None else mod._parameters
            for k, v in list(src.items()) + list(mod._buffers.items()):
                if v is not None:
                    out[(prefix + "." if prefix else "") + k] = v
        if scope is not None:
            scope[id(self)] = out
        return out

    def _device(self) -> torch.device:
        return next(iter(self._tensors().values())).device

    def _master(self):
        """The module whose parameters define the weights: self, or -- for an nn.DataParallel replica -- the wrapped module."""
        sh = self.__dict__.get("_shared")
        m = sh.master() if sh is not None else None
        if m is None or (m is not self and not getattr(self, "_is_replica", False)):
            # a deep copy (copy.deepcopy keeps weak references as they are) or a module unpickled without its state
            self._shared = sh = _Shared(self)
            m = self
        return m, sh

    def _fingerprint(self):
        m, _ = self._master()
        ps = list(m._tensors().values())                   # the MASTER's parameters + buffers (this call's walk, see _tensors)
        dev = self._device()
        return (dev, ps[0].device, tuple(p._version for p in ps), tuple(p.data_ptr() for p in ps[:4]), self.emb_nn.k,
                self.linear_mode, self.linear_mfma, self.linear_bk, self.linear_bm, self.knn_waves, self.xscore_limit_mb,
                self.merge_encdec, bool(self.workspace_flat), int(self.sdpa_variant), bool(self.iter_reuse))

    def _pack(self):
        """Packed weights for this device and these parameter versions, shared with every replica / thread.  The packing
        kernels (clones, fold_layernorm, split_bf16x3) are enqueued on the PACKING thread's current stream; the entry
        carries an event recorded behind them, and a caller on any other stream waits for it (once per stream) before its
        forward reads the packed pointers.  Replacing an entry (the weights changed) first drains the device: forwards on
        other streams may still be reading the old packing, whose memory the caching allocator would otherwise hand to
        the new one."""
        m, sh = self._master()
        key = self._fingerprint()                          # (outside the lock: a walk over ~60 tensors per call)
        dev = key[0]
        cur = torch.cuda.current_stream(dev)
        with sh.lock:
            hit = sh.packed.get(dev)
            if hit is not None and hit[0] == key:
                self._packed_key, self._packed, self._cw, ev, waited = hit
                if cur.cuda_stream not in waited:
                    cur.wait_event(ev)
                    waited.add(cur.cuda_stream)
                return
            if hit is not None:
                torch.cuda.synchronize(dev)
            self._pack_locked(key, own=m is not self)
            ev = torch.cuda.Event()
            ev.record(cur)
            sh.packed[dev] = (self._packed_key, self._packed, self._cw, ev, {cur.cuda_stream})
            sh.packs += 1

    def _pack_locked(self, key, own):
        sd = {k: v.detach().float() for k, v in self._tensors().items()}
        P: Dict[str, torch.Tensor] = {}
        # a replica's parameters are freed when its forward returns: what the cache keeps must own its memory
        g = (lambda k: sd[k].contiguous().clone()) if own else (lambda k: sd[k].contiguous())
        cw = native.VcrnetWeights()
        if self._emb_kind == "lpdnet":
            P["c1_w"] = g("emb_nn.conv1_lpd.weight").view(64, 3).contiguous(); P["c1_b"] = g("emb_nn.conv1_lpd.bias")
            P["c2_w"] = g("emb_nn.conv2_lpd.weight").view(64, 64).contiguous(); P["c2_b"] = g("emb_nn.conv2_lpd.bias")
            w = g("emb_nn.convDG1.0.weight").view(128, 128)                # cat((neighbour, centre)): util.py:197
            P["dg1_wpq"] = torch.cat((w[:, :64], w[:, 64:]), 0).contiguous()
            P["dg1_bpq"] = torch.cat((torch.zeros_like(sd["emb_nn.convDG1.0.bias"]), sd["emb_nn.convDG1.0.bias"]))
            P["dg2_w"] = g("emb_nn.convDG2.0.weight").view(128, 128).contiguous()
            P["dg2_b"] = g("emb_nn.convDG2.0.bias")
            w = g("emb_nn.convSN1.0.weight").view(256, 256)
            P["sn1_wpq"] = torch.cat((w[:, :128], w[:, 128:]), 0).contiguous()
            P["sn1_bpq"] = torch.cat((torch.zeros_like(sd["emb_nn.convSN1.0.bias"]), sd["emb_nn.convSN1.0.bias"]))
            P["c3_w"] = g("emb_nn.conv3_lpd.weight").view(self.emb_dims, 512).contiguous()
            P["c3_b"] = g("emb_nn.conv3_lpd.bias")
            for k in ("c1_w", "c1_b", "c2_w", "c2_b", "dg1_wpq", "dg1_bpq", "dg2_w", "dg2_b", "sn1_wpq", "sn1_bpq",
                      "c3_w", "c3_b"):
                setattr(cw, k, native.ptr(P[k]))
        if self._emb_kind == "dgcnn":
            # eval-mode BatchNorm folded into the bias-free 1x1 convs (vcrnet_model.py:108-121); conv1 split into its
            # neighbour / centre halves (get_graph_feature concatenates (x_j, x_i), util.py:197), K padded 3 -> 32
            def fold(i, eps=1e-5):
                sc = sd[f"emb_nn.bn{i}.weight"] / torch.sqrt(sd[f"emb_nn.bn{i}.running_var"] + eps)
                wq = sd[f"emb_nn.conv{i}.weight"].reshape(sd[f"emb_nn.conv{i}.weight"].shape[0], -1)
                return ((wq * sc.view(-1, 1)).contiguous(),
                        (sd[f"emb_nn.bn{i}.bias"] - sd[f"emb_nn.bn{i}.running_mean"] * sc).contiguous())
            w1, b1 = fold(1)
            wpq = torch.zeros(128, 32, dtype=torch.float32, device=w1.device)
            wpq[:64, :3], wpq[64:, :3] = w1[:, :3], w1[:, 3:]
            P["dg.c1_wpq"], P["dg.c1_bpq"] = wpq, torch.cat((torch.zeros_like(b1), b1)).contiguous()
            for i in (2, 3, 4, 5):
                P[f"dg.c{i}_w"], P[f"dg.c{i}_b"] = fold(i)
            cw.emb_kind = 1
            for f in ("c1_wpq", "c1_bpq", "c2_w", "c2_b", "c3_w", "c3_b", "c4_w", "c4_b", "c5_w", "c5_b"):
                setattr(cw.dgcnn, f, native.ptr(P["dg." + f]))
        if self._emb_kind == "pointnet":
            # eval-mode BatchNorm1d folded into the bias-free pointwise convs (vcrnet_model.py:81-87)
            from .composed import _fold_bn
            for i in (1, 2, 3, 4, 5):
                P[f"pn.c{i}_w"], P[f"pn.c{i}_b"] = _fold_bn(sd, f"emb_nn.conv{i}", f"emb_nn.bn{i}")
            cw.emb_kind = 2
            for i in (1, 2):                                               # same shapes as LPDNet's stem
                setattr(cw, f"c{i}_w", native.ptr(P[f"pn.c{i}_w"])); setattr(cw, f"c{i}_b", native.ptr(P[f"pn.c{i}_b"]))
            for f in ("c3_w", "c3_b", "c4_w", "c4_b", "c5_w", "c5_b"):
                setattr(cw.pointnet, f, native.ptr(P["pn." + f]))
        if isinstance(self.pointer, _TransformerParams):
            pre = "pointer.model."

            def norm(field, name):
                P[field + ".a"], P[field + ".b"] = g(name + ".a_2"), g(name + ".b_2")
                setattr(cw, field, native.NormW(native.ptr(P[field + ".a"]), native.ptr(P[field + ".b"])))

            def mha(field, name, cross):
                W = [g(f"{name}.linears.{i}.weight") for i in range(4)]
                Bv = [g(f"{name}.linears.{i}.bias") for i in range(4)]
                m = native.MhaW()
                if cross:
                    P[field + ".wq"], P[field + ".bq"] = W[0], Bv[0]
                    P[field + ".wkv"] = torch.cat((W[1], W[2]), 0).contiguous()
                    P[field + ".bkv"] = torch.cat((Bv[1], Bv[2])).contiguous()
                    m.wq, m.bq = native.ptr(W[0]), native.ptr(Bv[0])
                    m.wkv, m.bkv = native.ptr(P[field + ".wkv"]), native.ptr(P[field + ".bkv"])
                else:
                    P[field + ".wqkv"] = torch.cat(W[:3], 0).contiguous()
                    P[field + ".bqkv"] = torch.cat(Bv[:3]).contiguous()
                    m.wqkv, m.bqkv = native.ptr(P[field + ".wqkv"]), native.ptr(P[field + ".bqkv"])
                P[field + ".wo"], P[field + ".bo"] = W[3], Bv[3]
                m.wo, m.bo = native.ptr(W[3]), native.ptr(Bv[3])
                setattr(cw, field, m)

            def ffn(field, name):
                for leaf in ("w_1.weight", "w_1.bias", "w_2.weight", "w_2.bias"):
                    P[field + "." + leaf] = g(name + "." + leaf)
                setattr(cw, field, native.FfnW(*(native.ptr(P[field + "." + leaf]) for leaf in
                                                 ("w_1.weight", "w_1.bias", "w_2.weight", "w_2.bias"))))

            e, d = pre + "encoder.layers.0", pre + "decoder.layers.0"
            norm("enc_ln0", e + ".sublayer.0.norm"); norm("enc_ln1", e + ".sublayer.1.norm")
            norm("enc_norm", pre + "encoder.norm")
            norm("dec_ln0", d + ".sublayer.0.norm"); norm("dec_ln1", d + ".sublayer.1.norm")
            norm("dec_ln2", d + ".sublayer.2.norm"); norm("dec_norm", pre + "decoder.norm")
            mha("enc_self", e + ".self_attn", False); mha("dec_self", d + ".self_attn", False)
            mha("dec_cross", d + ".src_attn", True)
            ffn("enc_ffn", e + ".feed_forward"); ffn("dec_ffn", d + ".feed_forward")
            cw.has_pointer = 1
            # LayerNorm folded into the six linears that consume one (linear.hip LN_IN; the bf16x3 kernel takes the same
            # folded weight, pre-split below)
            for site, wk, bk, nk in (("enc_qkv", "enc_self.wqkv", "enc_self.bqkv", "enc_ln0"),
                                     ("enc_ffn1", "enc_ffn.w_1.weight", "enc_ffn.w_1.bias", "enc_ln1"),
                                     ("dec_qkv", "dec_self.wqkv", "dec_self.bqkv", "dec_ln0"),
                                     ("dec_cross_q", "dec_cross.wq", "dec_cross.bq", "dec_ln1"),
                                     ("dec_cross_kv", "dec_cross.wkv", "dec_cross.bkv", "enc_norm"),
                                     ("dec_ffn1", "dec_ffn.w_1.weight", "dec_ffn.w_1.bias", "dec_ln2")):
                f = native.fold_layernorm(P[wk], P[bk], P[nk + ".a"], P[nk + ".b"])
                P["fold." + site] = f
                P["fold." + site + ".w"] = f[0]
                setattr(cw, "fold_" + site, native.FoldedW(*(native.ptr(t) for t in f)))
            if self.merge_encdec:
                # the encoder's and the decoder's first sublayers both read the embedding rows: one stacked projection
                f = tuple(torch.cat((a_, b_), 0).contiguous() for a_, b_ in zip(P["fold.enc_qkv"], P["fold.dec_qkv"]))
                P["fold.encdec_qkv"] = f
                P["fold.encdec_qkv.w"] = f[0]
                cw.fold_encdec_qkv = native.FoldedW(*(native.ptr(t) for t in f))
        else:
            cw.has_pointer = 2 if isinstance(self.pointer, _Identity) else 0
        if self.linear_mode not in LINEAR_MODES:
            raise ValueError(f"linear_mode {self.linear_mode!r}: one of {sorted(LINEAR_MODES)}")
        cw.linear_mode = LINEAR_MODES[self.linear_mode]
        if cw.linear_mode != 0:                                           # (dgcnn / pointnet: the Transformer's sites only)
            # weights pre-split into exact bf16 triplets for vcr_linear_bf16x3_f32 (fp32-equivalent products)
            # (the six LayerNorm consumers: the FOLDED weight is what their main loop multiplies)
            src = {"dg1_pq": "dg1_wpq", "sn1_pq": "sn1_wpq", "c3": "c3_w", "enc_qkv": "fold.enc_qkv.w",
                   "enc_wo": "enc_self.wo", "enc_ffn1": "fold.enc_ffn1.w", "enc_ffn2": "enc_ffn.w_2.weight",
                   "dec_qkv": "fold.dec_qkv.w", "dec_self_wo": "dec_self.wo", "dec_cross_q": "fold.dec_cross_q.w",
                   "dec_cross_kv": "fold.dec_cross_kv.w", "dec_cross_wo": "dec_cross.wo", "dec_ffn1": "fold.dec_ffn1.w",
                   "dec_ffn2": "dec_ffn.w_2.weight", "encdec_qkv": "fold.encdec_qkv.w"}
            # (`name`, not `key`: until round 6 this loop's variable shadowed the fingerprint argument, so the split modes stored
            #  a weight NAME as their cache key, never hit the cache and re-packed -- 37 small launches, and since round 5 a
            #  device drain -- on every forward: 0.63 ms of a 4.0 ms step, profiles/r6a_split_trace_gaps.json)
            for site, name in src.items():
                if name in P:
                    P["split." + site] = native.split_bf16x3(P[name])
                    setattr(cw.split, site, native.ptr(P["split." + site]))
        cw.E, cw.F, cw.heads, cw.k = self.emb_dims, self._ff, self._n_heads, int(self.emb_nn.k)
        cw.head_mode = {"topK": 0, "dist": 1, "att": 2}[self._vcp]
        if self._vcp == "att":
            for i in (0, 1):
                P[f"att.w{i}"], P[f"att.b{i}"] = g(f"head.linears_emb.{i}.weight"), g(f"head.linears_emb.{i}.bias")
                setattr(cw, f"att_w{i}", native.ptr(P[f"att.w{i}"])); setattr(cw, f"att_b{i}", native.ptr(P[f"att.b{i}"]))
        cw.cycle = int(bool(self.cycle))
        cw.linear_mfma, cw.linear_bk, cw.linear_bm = int(self.linear_mfma), int(self.linear_bk), int(self.linear_bm)
        cw.knn_waves = int(self.knn_waves)
        cw.xscore_limit_mb = int(self.xscore_limit_mb)
        cw.workspace_flat = int(bool(self.workspace_flat))
        cw.sdpa_variant = int(self.sdpa_variant)
        cw.iter_reuse = 0 if self.iter_reuse else 1
        cw.partial, cw.overlap2 = int(self._partial), self._overlap2
        self._packed, self._packed_key = P, key
        self._cw = cw

    def _take_buffers(self, B: int, N: int, device, iters: int = 1) -> Tuple[Tuple, Dict[str, torch.Tensor]]:
        """A forward workspace for this shape from the shared pool (an idle one, or a new one): concurrent calls -- two host
        threads, DataParallel replicas -- never share one; _give_buffers returns it with the stream it was used on."""
        sh = self._shared
        key = (B, N, device, int(self.emb_nn.k), int(self.xscore_limit_mb), bool(self.merge_encdec), self.linear_mode,
               self._emb_kind, self._vcp, self._partial, self._overlap2, bool(self.workspace_flat),
               bool(self.iter_reuse) and iters > 1)        # (a loop with target reuse keeps its cache behind the workspace)
        with sh.lock:
            idle = sh.pool.get(key)
            if idle:
                return key, idle.pop()
            for k_ in [k_ for k_ in sh.pool if k_[2] == device and k_ != key]:     # keep one shape resident per device
                for b_ in sh.pool.pop(k_):
                    self._retire(b_)
        L = native.lib()
        L.vcr_vcrnet_iter_workspace_bytes.restype = C.c_size_t
        nbytes = L.vcr_vcrnet_iter_workspace_bytes(C.byref(self._cw), B, N, int(iters))
        return key, {"ws": torch.empty(nbytes + 256, dtype=torch.uint8, device=device)}

    MAX_IDLE_WORKSPACES = 4                                # per shape: more concurrent calls than this allocate and free

    @staticmethod
    def _retire(bufs):
        """Drop a pooled workspace whose last user may still be running on ANOTHER stream than the one it was allocated on:
        tell the caching allocator, or it could hand the memory to a new allocation while those kernels run."""
        st = bufs.get("stream")
        if st is not None:
            bufs["ws"].record_stream(st)

    def _give_buffers(self, key, bufs):
        with self._shared.lock:
            idle = self._shared.pool.setdefault(key, [])
            if len(idle) < self.MAX_IDLE_WORKSPACES:
                idle.append(bufs)
            else:
                self._retire(bufs)

    def fused_supported(self) -> bool:
        """True when one vcr_vcrnet_forward_f32 / vcr_vcrnet_iter_f32 call covers this configuration: every
        embedding / pointer / head / cycle / partial combination except the corner cases below, which run kernel by
        kernel from composed.py (more than one Transformer block, partial mode without the Transformer, cycle with the
        partial topK head -- the last one is not defined by the reference either)."""
        if isinstance(self.pointer, _TransformerParams) and self.pointer.N != 1:
            return False                                   # args.n_blocks > 1: the layers run kernel by kernel (composed.py)
        if self._partial:
            if not isinstance(self.pointer, _TransformerParams):
                return False
            return not (self.cycle and self._vcp == "topK")
        return True

    def _check_call(self, src, tgt):
        if not (src.is_cuda and tgt.is_cuda):
            raise native.VcrHipError("vcrnet_amd.VCRNet runs on the MI355X HIP path only; move inputs to cuda "
                                     "(there is no CPU fallback by design)")
        pdev = self._device()
        if not (src.device == tgt.device == pdev):
            raise native.VcrHipError(f"src ({src.device}), tgt ({tgt.device}) and the parameters ({pdev}) must "
                                     "live on one device")
        if self.training or torch.is_grad_enabled():
            raise native.VcrHipError("inference only: call .eval() and wrap in torch.no_grad() "
                                     "(model/vcrnet_model.py:546); backward kernels are out of scope")

    # -- forward ------------------------------------------------------------------------------------------------
    def forward(self, *input):
        with _call_scope():
            return self._forward(*input)

    def _forward(self, *input):
        src, tgt = input[0], input[1]
        self._check_call(src, tgt)
        # HIP launches go to the CURRENT device's stream: make the tensors' device current for the whole call (a module
        # moved with .to('cuda:1') but called without torch.cuda.set_device(1), or an nn.DataParallel replica)
        with torch.cuda.device(src.device):
            self._pack()
            if not self.fused_supported():
                from .composed import forward_composed
                return forward_composed(self, src, tgt)
            return self._forward_fused(src, tgt, trace=self.launch_trace)

    def selection_sizes(self, N: int) -> Dict[str, int]:
        """Per-sample lengths of the partial-mode selections (transformer.py:41, vcrnet_model.py:208,284)."""
        k1 = int(N * 0.84 * self._overlap2)
        return {"keys": int(N * self._overlap2), "sel_src": k1, "sel_tgt": k1, "argmax": k1,
                "pairs": int(k1 * 0.52 * self._overlap2)}

    def _forward_fused(self, src, tgt, trace: Optional[native.Trace] = None, want_emb: bool = False, iters: int = 1,
                       force: Optional[Dict[str, torch.Tensor]] = None, want_selections: bool = False,
                       iter_api: bool = False):
        """One C-ABI call: VCRNet.forward (iters == 1) or the whole vcrnetIter loop (iters > 1).

        Partial mode only: ``force`` = {"keys": [iters, 2B, nkeep], "sel_src" / "sel_tgt": [iters, B, K1],
        "argmax": [iters, B, K1], "pairs": [iters, B, K2]} int32 (any subset; the leading dimension may be omitted for
        iters == 1) replaces the device's rankings by the caller's -- teacher forcing for the parity tests;
        ``want_selections`` appends a dict of the selections that were used, one block per iteration.
        ``iter_api``: use vcr_vcrnet_iter_f32 also for iters == 1 (vcrnetIter's contract: (R_ba, t_ba) is ALWAYS the
        inverse of the composed pose, vcrnet_model.py:40-41, even when args.cycle gives forward() a second head)."""
        with _call_scope(), torch.cuda.device(src.device):
            return self._forward_fused_on(src, tgt, trace, want_emb, iters, force, want_selections, iter_api)

    def _forward_fused_on(self, src, tgt, trace, want_emb, iters, force, want_selections, iter_api):
        self._pack()
        B, _, N = src.shape
        dev = native.same_device(src, tgt, next(iter(self._tensors().values())))
        srcc, tgtc = src.contiguous().float(), tgt.contiguous().float()
        bkey, bufs = self._take_buffers(B, N, dev, iters)
        try:
            return self._forward_with(bufs, srcc, tgtc, src, B, N, dev, trace, want_emb, iters, force, want_selections,
                                      iter_api)
        finally:
            self._give_buffers(bkey, bufs)          # (enqueued: the next user on another stream waits for this one)

    def _forward_with(self, bufs, srcc, tgtc, src, B, N, dev, trace, want_emb, iters, force, want_selections, iter_api):
        ws = bufs["ws"]
        # a workspace is reused by later calls: one on ANOTHER stream first waits for the previous user's stream
        cur = torch.cuda.current_stream(dev)
        prev = bufs.get("stream")
        if prev is not None and prev != cur:
            cur.wait_stream(prev)
        bufs["stream"] = cur
        off = (-ws.data_ptr()) % 256
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        L = native.lib()
        K = L.vcr_vcrnet_pairs(C.byref(self._cw), N)                      # N, or the hard pairs of partial mode
        corr4, src4 = f(B, K, 4), f(B, K, 4)
        R_ab, t_ab, R_ba, t_ba = f(B, 3, 3), f(B, 3), f(B, 3, 3), f(B, 3)
        emb = f(2 * B * N, self.emb_dims) if want_emb else None
        io = native.VcrnetIo(native.ptr(srcc), native.ptr(tgtc), B, N, native.ptr(corr4), native.ptr(src4),
                             native.ptr(R_ab), native.ptr(t_ab), native.ptr(R_ba), native.ptr(t_ba), native.ptr(emb))
        keepalive, sel = [], {}
        if force or want_selections:
            if not self._partial:
                raise native.VcrHipError("forced / reported selections exist in partial mode only")
            sizes = self.selection_sizes(N)
            for name in native.SELECTION_FIELDS:
                rows = 2 * B if name == "keys" else B
                if force and name in force:
                    t = force[name].to(device=dev, dtype=torch.int32).reshape(iters, rows, sizes[name]).contiguous()
                    keepalive.append(t)
                    setattr(io, "force_" + name, native.ptr(t))
                if want_selections and (self._vcp == "topK" or name == "keys"):
                    sel[name] = torch.empty(iters, rows, sizes[name], dtype=torch.int32, device=dev)
                    setattr(io, "out_" + name, native.ptr(sel[name]))
        stream = C.c_void_p(native.stream_ptr(dev))
        wsp = C.c_void_p(ws.data_ptr() + off)
        if iters != 1 or iter_api:
            rc = L.vcr_vcrnet_iter_f32(C.byref(self._cw), C.byref(io), iters, wsp, ws.numel() - off, stream,
                                       C.byref(trace) if trace is not None else None)
        elif trace is None:
            rc = L.vcr_vcrnet_forward_f32(C.byref(self._cw), C.byref(io), wsp, ws.numel() - off, stream)
        else:
            rc = L.vcr_vcrnet_forward_traced_f32(C.byref(self._cw), C.byref(io), wsp, ws.numel() - off, stream,
                                                 C.byref(trace))
        native.check(rc, "vcr_vcrnet_iter_f32" if (iters != 1 or iter_api) else "vcr_vcrnet_forward_f32")
        rows = lambda x: x[:, :, :3].transpose(1, 2).contiguous()
        hard = self._partial and self._vcp == "topK"
        srcK = rows(src4) if (hard or iters != 1) else src               # soft heads return src itself (:347)
        out = (srcK, rows(corr4), R_ab, t_ab, R_ba, t_ba)
        if want_emb:
            out = out + (emb,)
        return out + (sel,) if want_selections else out

    def forward_iter(self, src, tgt, iters: int):
        """vcrnetIter (model/vcrnet_model.py:21-43) as ONE device-side loop when the fused driver covers this
        configuration; None otherwise (the caller then loops over forward())."""
        self._check_call(src, tgt)
        if not self.fused_supported():
            return None
        return self._forward_fused(src, tgt, trace=self.launch_trace, iters=int(iters), iter_api=True)


class DCP(VCRNet):
    """Drop-in for the reference's DCP with head='svd', use_mFea=False (model/dcp_model.py:177-223): the same
    embedding + pointer, the scaled-dot-product soft correspondences fused with the SVD solve
    (dcp_model.py:126-174).  Output order (R_ab, t_ab, R_ba, t_ba, src, src_corr) differs from VCRNet's."""

    def __init__(self, args):
        if getattr(args, "head", "svd") != "svd" or getattr(args, "use_mFea", False):
            raise Exception("Not implemented")                             # dcp_model.py:198-203 (mlp head: out of scope)
        if args.pointer not in ("identity", "transformer"):
            raise Exception("Not implemented")                             # dcp_model.py:191-196
        a = dict(vars(args))
        a.update(vcp_nn="dist", partial=False)
        super().__init__(type("Args", (), a)())
        del self.svd
        self.head = _SVDHead()                                             # key 'head.reflect' (dcp_model.py:121-122)

    def forward(self, *input):
        src, corr, R_ab, t_ab, R_ba, t_ba = super().forward(*input)
        return R_ab, t_ab, R_ba, t_ba, src, corr


class ICP(nn.Module):
    """Drop-in for the reference's torch ICP (model/icp_model.py:16-108): same constructor and the same
    6-tuple ``(srcInit, src_final, R_ab, t_ab, R_ba, t_ba)``, but the whole loop -- nearest neighbour,
    best-fit transform, apply, convergence test -- runs on the device from one C-ABI call, with no
    host sync per iteration."""

    def __init__(self, max_iterations=10, tolerance=0.001):
        super().__init__()
        self.max_iterations, self.tolerance = max_iterations, tolerance
        r = torch.eye(3)
        r[2, 2] = -1
        self.reflect = nn.Parameter(r, requires_grad=False)                # icp_model.py:22-23
        self.last_iterations = None

    def forward(self, srcInit, dst):
        if not (srcInit.is_cuda and dst.is_cuda):
            raise native.VcrHipError("vcrnet_amd.ICP runs on the HIP path only (no CPU fallback)")
        final, R, t, Rb, tb, iters = native.icp(srcInit, dst, self.max_iterations, self.tolerance)
        self.last_iterations = iters                                       # device int32[1]; read lazily
        return srcInit, final, R, t, Rb, tb


def vcrnetIcpNet(args, net, src, tgt):
    """model/vcrnet_model.py:46-62 (--iter 0): one network pass, ICP on the moved source, poses composed."""
    icp = ICP(max_iterations=args.max_iterations)
    _, _, R, t, _, _ = net(src, tgt)
    moved = native.pose_step(R, t, src)[0]                                 # transform_point_cloud, util/util.py:91-96
    _, _, Ri, ti, _, _ = icp(moved, tgt)
    _, R2, t2, R_ba, t_ba = native.pose_step(Ri, ti, None, R, t)           # :55-59: R_i R, R_i t + t_i and its inverse
    return moved, tgt, R2, t2, R_ba, t_ba


def vcrnetIter(net, src, tgt, iter=1):
    """model/vcrnet_model.py:21-43: run ``iter`` passes, composing the poses on the device."""
    # The reference's caller always wraps the net in nn.DataParallel (util/initPara.py:260).  With one visible device
    # (or inputs already on the wrapped module's device and a single device id) DataParallel.forward just calls
    # module(*inputs), so the device-side loop of the wrapped module is the same computation without the wrapper's
    # scatter; with several device ids the wrapper must do its scatter / replicate, and the Python loop below drives it.
    inner = net.module if isinstance(net, nn.DataParallel) else net
    single = inner is net or len(getattr(net, "device_ids", ())) <= 1
    if inner is not net and single and getattr(net, "device_ids", None):
        # what DataParallel.forward's scatter does with one device id: inputs are moved to that device
        dev = torch.device("cuda", net.device_ids[0]) if isinstance(net.device_ids[0], int) else torch.device(net.device_ids[0])
        src, tgt = src.to(dev), tgt.to(dev)
    if isinstance(inner, VCRNet) and single and iter >= 1:
        out = inner.forward_iter(src, tgt, iter)
        if out is not None:
            return out
    cur = src
    R_f = t_f = None
    for _ in range(iter):
        srcK, corrK, R, t, _, _ = net(cur, tgt)
        # util/util.py:91-96 and vcrnet_model.py:35-41 as one vcr_pose_step_f32 launch (the device loop's own kernel)
        cur, R_f, t_f, R_ba, t_ba = native.pose_step(R.detach(), t.detach(), cur, R_f, t_f)
    return srcK, corrK, R_f, t_f, R_ba, t_ba
