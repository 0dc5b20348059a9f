"""CPU restatement of the VCR-Net per-batch registration path (plain PyTorch fp32).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  This file restates, as
stateless functions over a flat weight dictionary, the algorithm of the
reference's ``VCRNet.forward`` and the modules below it.  Every function cites
the reference lines (paths relative to /root/reference) it follows.  The tensor
primitives and their order are kept identical to the reference's so that the
discrete steps (top-k neighbour sets, arg-max correspondences) resolve the same
way on the same CPU; the structure (functional, dictionary-of-weights, one
config object) is this repository's own.

Parity status: PINNED.  ``tests/golden/gen_golden.py`` imports the reference in
the build container, loads the same weights into it and records its outputs
and intermediates; ``tests/test_oracle_golden.py`` checks every function here
against those recordings.

Weight keys are the reference's ``state_dict`` names.
Layout is the reference's channels-first ``[B, C, N]``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Weights = Dict[str, Tensor]

__all__ = [
    "OracleConfig", "neg_sqdist_knn", "neg_sqdist_head", "knn_indices", "graph_feature",
    "lpdnet_embed", "dgcnn_embed", "pointnet_embed", "layer_norm", "attention", "multi_head_attention",
    "feed_forward", "encoder_decoder", "transformer_pointer", "head_topk_whole",
    "head_select_overlap", "head_hard_pairs", "head_topk", "head_by_dis", "head_att",
    "rigid_svd", "vcrnet_forward", "vcrnet_iter", "dcp_forward", "transform_point_cloud",
    "fold_batchnorm", "icp_nearest", "icp_forward", "vcrnet_icp",
]


@dataclass
class OracleConfig:
    """The subset of the reference's ``args`` the forward path reads
    (model/vcrnet_model.py:464-493, model/lpdnet_model.py:81-84,
    model/transformer.py:244-253)."""
    emb_nn: str = "lpdnet"          # lpdnet | dgcnn | pointnet
    pointer: str = "transformer"    # transformer | identity | none
    vcp_nn: str = "topK"            # topK | att | dist
    partial: bool = False
    overlap2: float = 0.75
    cycle: bool = False
    k: int = 20                     # LPDNet.k (lpdnet_model.py:81) / get_graph_feature default (util.py:176)
    n_heads: int = 4
    n_blocks: int = 1
    emb_dims: int = 512
    record: Optional[dict] = field(default=None, repr=False)  # optional sink for intermediates

    def rec(self, name: str, value) -> None:
        if self.record is not None:
            self.record[name] = value


# ----------------------------------------------------------------------------
# graph ops
# ----------------------------------------------------------------------------

def neg_sqdist_knn(x: Tensor) -> Tensor:
    """Negative squared distance matrix in the exact expanded form of
    util/util.py:153-158: ``(-xx_j - (-2 x_i.x_j)) - xx_i`` -> [B, N, N]."""
    inner = -2 * torch.matmul(x.transpose(2, 1).contiguous(), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    d = -xx - inner
    return d - xx.transpose(2, 1).contiguous()


def knn_indices(x: Tensor, k: int) -> Tensor:
    """util/util.py:143-160: top-(k+1) of the negative distance, rank 0 dropped."""
    return neg_sqdist_knn(x).topk(k=k + 1, dim=-1)[1][:, :, 1:]


def graph_feature(x: Tensor, idx: Tensor) -> Tensor:
    """util/util.py:176-199: gather k neighbour rows and concatenate
    (neighbour, centre) -> [B, 2C, N, k].  Note: neighbour FIRST."""
    b, c, n = x.shape
    k = idx.shape[-1]
    xt = x.transpose(2, 1).contiguous()                                  # [B,N,C]
    flat = (idx + torch.arange(b).view(-1, 1, 1) * n).reshape(-1)
    nbr = xt.reshape(b * n, c)[flat].view(b, n, k, c)
    ctr = xt.view(b, n, 1, c).repeat(1, 1, k, 1)
    return torch.cat((nbr, ctr), dim=3).permute(0, 3, 1, 2)


# ----------------------------------------------------------------------------
# feature extractors
# ----------------------------------------------------------------------------

def lpdnet_embed(w: Weights, x: Tensor, cfg: OracleConfig, prefix: str = "emb_nn.") -> Tensor:
    """model/lpdnet_model.py:103-137 with t3d = tfea = False.
    LeakyReLU(negative_slope=0.0) == ReLU (lpdnet_model.py:78 via vcrnet_model.py:473)."""
    act = lambda t: F.leaky_relu(t, negative_slope=0.0)
    p = lambda name: w[prefix + name]
    b, _, n = x.shape
    x3d = x
    h = act(F.conv1d(x, p("conv1_lpd.weight"), p("conv1_lpd.bias")))      # :111
    h = act(F.conv1d(h, p("conv2_lpd.weight"), p("conv2_lpd.bias")))      # :112
    cfg.rec("x64", h)
    idx_f = knn_indices(h, cfg.k)                                          # :122 (feature space)
    cfg.rec("idx_feat", idx_f)
    g = graph_feature(h, idx_f)
    g = act(F.conv2d(g, p("convDG1.0.weight"), p("convDG1.0.bias")))      # :123
    x1 = g.max(dim=-1, keepdim=True)[0]                                    # :124
    g = act(F.conv2d(g, p("convDG2.0.weight"), p("convDG2.0.bias")))      # :125
    x2 = g.max(dim=-1, keepdim=True)[0]                                    # :126
    idx_c = knn_indices(x3d, cfg.k)                                        # :129 (Cartesian space)
    cfg.rec("idx_xyz", idx_c)
    g = graph_feature(x2.squeeze(-1), idx_c)                               # :130
    g = act(F.conv2d(g, p("convSN1.0.weight"), p("convSN1.0.bias")))      # :131
    x3 = g.max(dim=-1, keepdim=True)[0]                                    # :132
    cfg.rec("x1", x1.squeeze(-1)); cfg.rec("x2", x2.squeeze(-1)); cfg.rec("x3", x3.squeeze(-1))
    cat = torch.cat((x1, x2, x3), dim=1).squeeze(-1)                       # :134
    return act(F.conv1d(cat, p("conv3_lpd.weight"), p("conv3_lpd.bias"))).view(b, -1, n)  # :135


def fold_batchnorm(w: Weights, conv: str, bn: str, eps: float = 1e-5) -> Tuple[Tensor, Tensor]:
    """Eval-mode BatchNorm as a per-channel affine folded into the preceding
    bias-free 1x1 convolution: returns (W', b') with bn(conv(x)) == W'x + b'."""
    g, beta = w[bn + ".weight"], w[bn + ".bias"]
    mu, var = w[bn + ".running_mean"], w[bn + ".running_var"]
    s = g / torch.sqrt(var + eps)
    wt = w[conv + ".weight"]
    return wt * s.view(-1, *([1] * (wt.dim() - 1))), beta - mu * s


def _bn_eval(w: Weights, name: str, x: Tensor) -> Tensor:
    return F.batch_norm(x, w[name + ".running_mean"], w[name + ".running_var"],
                        w[name + ".weight"], w[name + ".bias"], training=False, eps=1e-5)


def dgcnn_embed(w: Weights, x: Tensor, cfg: OracleConfig, prefix: str = "emb_nn.") -> Tensor:
    """model/vcrnet_model.py:104-123 (== model/dcp_model.py:60-79), eval-mode BN."""
    b, _, n = x.shape
    idx = knn_indices(x, cfg.k)                                            # :106 via util.py:179
    cfg.rec("idx_xyz", idx)
    g = graph_feature(x, idx)
    outs = []
    for i in (1, 2, 3, 4):                                                 # :108-118
        g = F.relu(_bn_eval(w, f"{prefix}bn{i}", F.conv2d(g, w[f"{prefix}conv{i}.weight"])))
        outs.append(g.max(dim=-1, keepdim=True)[0])
    for i, o in enumerate(outs):
        cfg.rec(f"dg_x{i + 1}", o.squeeze(-1))
    cat = torch.cat(outs, dim=1)                                           # :120
    y = F.relu(_bn_eval(w, prefix + "bn5", F.conv2d(cat, w[prefix + "conv5.weight"])))
    return y.view(b, -1, n)                                                # :122


def pointnet_embed(w: Weights, x: Tensor, cfg: OracleConfig, prefix: str = "emb_nn.") -> Tensor:
    """model/vcrnet_model.py:81-87: five pointwise Conv1d(bias=False) + BatchNorm1d (eval mode) + ReLU, no graph."""
    for i in (1, 2, 3, 4, 5):
        x = F.relu(_bn_eval(w, f"{prefix}bn{i}", F.conv1d(x, w[f"{prefix}conv{i}.weight"])))
        if i == 2:
            cfg.rec("pn_x64", x)
    return x


# ----------------------------------------------------------------------------
# Transformer pointer
# ----------------------------------------------------------------------------

def layer_norm(x: Tensor, a: Tensor, b: Tensor, eps: float = 1e-6) -> Tensor:
    """model/transformer.py:141-144 -- unbiased std, eps added to the std."""
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return a * (x - mean) / (std + eps) + b


def attention(q: Tensor, k: Tensor, v: Tensor, is_src: bool = False, overlap2: float = 0.75,
              cfg: Optional[OracleConfig] = None, tag: str = "") -> Tensor:
    """model/transformer.py:13-55 with mask=None, dropout=None.
    q,k,v: [B,h,N,d].  ``is_src`` adds the key-pruning second softmax (:35-53)."""
    d_k = q.size(-1)
    scores = torch.matmul(q, k.transpose(-2, -1).contiguous()) / math.sqrt(d_k)
    p = F.softmax(scores, dim=-1)
    if is_src:
        bsz, n_head, nk, _ = k.shape
        nq = q.shape[2]
        col = torch.sum(p, dim=[1, 2], keepdim=True)                       # :40
        keep = int(nk * overlap2)                                          # :41
        idx = col.topk(k=keep, dim=-1)[1]                                  # :42  [B,1,1,keep]
        if cfg is not None:
            cfg.rec("key_keep", idx.view(bsz, keep))
            cfg.rec("key_keep" + tag, idx.view(bsz, keep))                 # tag names the cloud whose points are the keys
        mask = torch.zeros(bsz, nk, dtype=torch.bool)
        mask.view(-1)[(idx.view(bsz, keep) + torch.arange(bsz).view(-1, 1) * nk).view(-1)] = True
        scores = scores.masked_fill(~mask.view(bsz, 1, 1, nk), -1e9)       # :46-52
        p = F.softmax(scores, dim=-1)                                      # :53
    return torch.matmul(p, v)


def multi_head_attention(w: Weights, pre: str, q_in: Tensor, k_in: Tensor, v_in: Tensor, h: int,
                         is_src: bool = False, overlap2: float = 0.75,
                         cfg: Optional[OracleConfig] = None, tag: str = "") -> Tensor:
    """model/transformer.py:202-224 (the stored ``self.attn`` side effect, :216-219, is
    plotting-only and not reproduced)."""
    nb = q_in.size(0)
    d_model = q_in.size(-1)
    d_k = d_model // h
    proj = []
    for i, t in enumerate((q_in, k_in, v_in)):
        y = F.linear(t, w[f"{pre}linears.{i}.weight"], w[f"{pre}linears.{i}.bias"])
        proj.append(y.view(nb, -1, h, d_k).transpose(1, 2).contiguous())
    x = attention(proj[0], proj[1], proj[2], is_src=is_src, overlap2=overlap2, cfg=cfg, tag=tag)
    x = x.transpose(1, 2).contiguous().view(nb, -1, h * d_k)
    return F.linear(x, w[f"{pre}linears.3.weight"], w[f"{pre}linears.3.bias"])


def feed_forward(w: Weights, pre: str, x: Tensor) -> Tensor:
    """model/transformer.py:237-238 (the wrapped nn.Sequential is empty)."""
    hdn = F.relu(F.linear(x, w[pre + "w_1.weight"], w[pre + "w_1.bias"]))
    return F.linear(hdn, w[pre + "w_2.weight"], w[pre + "w_2.bias"])


def encoder_decoder(w: Weights, pre: str, src: Tensor, tgt: Tensor, cfg: OracleConfig,
                    tag: str = "") -> Tensor:
    """model/transformer.py:58-82,108-131,147-185: decode(tgt | encode(src)),
    pre-norm residual sublayers.  src,tgt: [B,N,E]."""
    h = cfg.n_heads
    ln = lambda t, name: layer_norm(t, w[name + ".a_2"], w[name + ".b_2"])
    x = src
    for i in range(cfg.n_blocks):
        lp = f"{pre}encoder.layers.{i}."
        y = ln(x, lp + "sublayer.0.norm")
        x = x + multi_head_attention(w, lp + "self_attn.", y, y, y, h)                       # :164
        x = x + feed_forward(w, lp + "feed_forward.", ln(x, lp + "sublayer.1.norm"))        # :165
    mem = ln(x, pre + "encoder.norm")                                                         # :117
    cfg.rec("memory" + tag, mem)
    x = tgt
    for i in range(cfg.n_blocks):
        lp = f"{pre}decoder.layers.{i}."
        y = ln(x, lp + "sublayer.0.norm")
        x = x + multi_head_attention(w, lp + "self_attn.", y, y, y, h)                       # :183
        y = ln(x, lp + "sublayer.1.norm")
        x = x + multi_head_attention(w, lp + "src_attn.", y, mem, mem, h,                    # :184
                                     is_src=cfg.partial, overlap2=float(cfg.overlap2), cfg=cfg,
                                     tag=tag + (f"_l{i}" if cfg.n_blocks > 1 else ""))       # (every layer prunes its own keys)
        x = x + feed_forward(w, lp + "feed_forward.", ln(x, lp + "sublayer.2.norm"))        # :185
    return ln(x, pre + "decoder.norm")                                                        # :131


def transformer_pointer(w: Weights, src_emb: Tensor, tgt_emb: Tensor, cfg: OracleConfig,
                        prefix: str = "pointer.") -> Tuple[Tensor, Tensor]:
    """model/transformer.py:264-272: same weights both directions."""
    s = src_emb.transpose(2, 1).contiguous()
    t = tgt_emb.transpose(2, 1).contiguous()
    tgt_p = encoder_decoder(w, prefix + "model.", s, t, cfg, tag="_src").transpose(2, 1).contiguous()
    src_p = encoder_decoder(w, prefix + "model.", t, s, cfg, tag="_tgt").transpose(2, 1).contiguous()
    return src_p, tgt_p


# ----------------------------------------------------------------------------
# virtual-correspondence heads
# ----------------------------------------------------------------------------

def neg_sqdist_head(a: Tensor, b: Tensor) -> Tensor:
    """Score matrix of model/vcrnet_model.py:211-216 / :287-292 / :337-342:
    ``(-xx_i - (-2 a_i.b_j)) - yy_j`` -> [B, Na, Nb] (row term first, unlike knn)."""
    inner = -2 * torch.matmul(a.transpose(2, 1).contiguous(), b)
    xx = torch.sum(a ** 2, dim=1, keepdim=True).transpose(2, 1).contiguous()
    yy = torch.sum(b ** 2, dim=1, keepdim=True)
    d = -xx - inner
    return d - yy


def head_topk_whole(src_emb: Tensor, tgt_emb: Tensor, src: Tensor, tgt: Tensor) -> Tuple[Tensor, Tensor]:
    """VcpTopK.getCopairALL, model/vcrnet_model.py:334-347."""
    scores = torch.softmax(neg_sqdist_head(src_emb, tgt_emb), dim=2)
    return src, torch.matmul(tgt, scores.transpose(2, 1).contiguous())


def _rows(x: Tensor, idx: Tensor) -> Tensor:
    """Gather points of a channels-first [B,C,N] tensor by per-sample indices [B,K] -> [B,C,K].
    Done the reference's way -- transpose to point-major, flat row gather, permute back
    (vcrnet_model.py:230-238,251-260,328-330) -- so the result has the SAME strides (a permuted
    view of [B,K,C]); downstream matmul/sum kernels then round identically, which matters for
    the near-tie arg-max decisions of the partial path."""
    b, c, n = x.shape
    k = idx.shape[1]
    flat = (idx + torch.arange(b).view(-1, 1) * n).reshape(-1)
    return x.transpose(2, 1).contiguous().view(b * n, c)[flat, :].view(b, k, c).permute(0, 2, 1)


def head_select_overlap(src: Tensor, src_emb: Tensor, tgt: Tensor, tgt_emb: Tensor, overlap2: float,
                        cfg: Optional[OracleConfig] = None):
    """VcpTopK.selectCom, model/vcrnet_model.py:190-262.  The ``*_remain`` outputs
    (np.setdiff1d, :228,249) are discarded by the caller (:181-184) and are not produced."""
    ns, nt = src.shape[2], tgt.shape[2]
    src_k = int(ns * 0.84 * overlap2)                                       # :208
    tgt_k = int(nt * 0.84 * overlap2)                                       # :209
    scores = neg_sqdist_head(src_emb, tgt_emb)
    col = torch.sum(torch.softmax(scores, dim=2), dim=1, keepdim=True)      # :221-222
    idx_t = col.topk(k=tgt_k, dim=-1)[1].view(-1, tgt_k)                    # :223
    row = torch.sum(torch.softmax(scores, dim=1), dim=2, keepdim=True)      # :243-244
    idx_s = row.topk(k=src_k, dim=-2)[1].view(-1, src_k)                    # :245
    if cfg is not None:
        cfg.rec("sel_tgt", idx_t); cfg.rec("sel_src", idx_s)
    return _rows(src, idx_s), _rows(src_emb, idx_s), _rows(tgt, idx_t), _rows(tgt_emb, idx_t)


def head_hard_pairs(src: Tensor, src_emb: Tensor, tgt: Tensor, tgt_emb: Tensor, overlap2: float,
                    cfg: Optional[OracleConfig] = None) -> Tuple[Tensor, Tensor]:
    """VcpTopK.getCopair, model/vcrnet_model.py:264-332 with tgtK = 1 (:283):
    arg-max target per source, keep the int(n*0.52*overlap2) sources with the largest
    soft-max peak; the normalised weight is exactly 1 (:320-321)."""
    ns = src.shape[2]
    src_k = int(ns * 0.52 * overlap2)                                       # :284
    p = torch.softmax(neg_sqdist_head(src_emb, tgt_emb), dim=2)             # :295
    val, idx = p.topk(k=1, dim=-1)                                          # :297-298
    cand = _rows(tgt, idx.squeeze(-1))                                      # :305-306  [B,3,ns]
    pick = torch.sum(val, dim=-1, keepdim=True).topk(k=src_k, dim=-2)[1].view(-1, src_k)   # :312
    w = torch.div(val, torch.sum(val, dim=-1, keepdim=True))                # :320-321 (== 1)
    if cfg is not None:
        cfg.rec("pair_src", pick); cfg.rec("pair_tgt", torch.gather(idx.squeeze(-1), 1, pick))
        cfg.rec("argmax_tgt", idx.squeeze(-1))
        cfg.rec("pair_val", val.squeeze(-1))                                 # the soft-max peaks that :312 ranks
    src_corr = _rows(cand * w.transpose(2, 1), pick)                        # :325 (weights are exactly 1)
    return _rows(src, pick), src_corr                                       # :328-330


def head_topk(src_emb, tgt_emb, src, tgt, cfg: OracleConfig):
    """VcpTopK.forward, model/vcrnet_model.py:173-188."""
    if cfg.partial:
        o2 = float(cfg.overlap2)
        s, se, t, te = head_select_overlap(src, src_emb, tgt, tgt_emb, o2, cfg)
        return head_hard_pairs(s, se, t, te, o2, cfg)
    return head_topk_whole(src_emb, tgt_emb, src, tgt)


def head_by_dis(src_emb, tgt_emb, src, tgt):
    """VcpByDis.forward, model/vcrnet_model.py:407-421."""
    d_k = src_emb.size(1)
    scores = torch.matmul(src_emb.transpose(2, 1).contiguous(), tgt_emb) / math.sqrt(d_k)
    scores = torch.softmax(scores, dim=2)
    return src, torch.matmul(tgt, scores.transpose(2, 1).contiguous())


def head_att(w: Weights, src_emb, tgt_emb, src, tgt, prefix: str = "head."):
    """VcpAtt.forward, model/vcrnet_model.py:434-460 (``linears_3d`` is unused there)."""
    q = F.linear(src_emb.transpose(2, 1).contiguous(), w[prefix + "linears_emb.0.weight"],
                 w[prefix + "linears_emb.0.bias"]).transpose(2, 1).contiguous()
    k = F.linear(tgt_emb.transpose(2, 1).contiguous(), w[prefix + "linears_emb.1.weight"],
                 w[prefix + "linears_emb.1.bias"]).transpose(2, 1).contiguous()
    scores = torch.softmax(neg_sqdist_head(q, k), dim=2)
    return src, torch.matmul(tgt, scores.transpose(2, 1).contiguous())


# ----------------------------------------------------------------------------
# rigid solve
# ----------------------------------------------------------------------------

def rigid_svd(src: Tensor, corr: Tensor, cfg: Optional[OracleConfig] = None) -> Tuple[Tensor, Tensor]:
    """SVDHead.forward, model/vcrnet_model.py:356-399: centre, H = S C^T, per-sample
    SVD, R = V U^T, flip V's last column when det R < 0 (:382-386), t = -R s_mean + c_mean."""
    sc = src - src.mean(dim=2, keepdim=True)
    cc = corr - corr.mean(dim=2, keepdim=True)
    H = torch.matmul(sc, cc.transpose(2, 1).contiguous())
    if cfg is not None:
        cfg.rec("H", H)
    reflect = torch.eye(3, dtype=H.dtype)                  # (dtype of the inputs: the tests run a float64 twin of the oracle too)
    reflect[2, 2] = -1
    rs = []
    for i in range(src.size(0)):
        u, _, v = torch.svd(H[i])
        r = torch.matmul(v, u.transpose(1, 0).contiguous())
        if torch.det(r) < 0:
            r = torch.matmul(torch.matmul(v, reflect), u.transpose(1, 0).contiguous())
        rs.append(r)
    R = torch.stack(rs, dim=0)
    t = torch.matmul(-R, src.mean(dim=2, keepdim=True)) + corr.mean(dim=2, keepdim=True)
    return R, t.view(src.size(0), 3)


def transform_point_cloud(p: Tensor, R: Tensor, t: Tensor) -> Tensor:
    """util/util.py:91-96 for matrix rotations."""
    return torch.matmul(R, p) + t.unsqueeze(2)


# ----------------------------------------------------------------------------
# whole models
# ----------------------------------------------------------------------------

def _embed(w: Weights, x: Tensor, cfg: OracleConfig) -> Tensor:
    if cfg.emb_nn == "lpdnet":
        return lpdnet_embed(w, x, cfg)
    if cfg.emb_nn == "dgcnn":
        return dgcnn_embed(w, x, cfg)
    if cfg.emb_nn == "pointnet":
        return pointnet_embed(w, x, cfg)
    raise Exception("Not implemented")                                      # vcrnet_model.py:475


def vcrnet_forward(w: Weights, src: Tensor, tgt: Tensor, cfg: OracleConfig):
    """VCRNet.forward, model/vcrnet_model.py:495-518 -> (srcK, src_corrK, R_ab, t_ab, R_ba, t_ba)."""
    with torch.no_grad():
        sub = cfg.record
        if sub is not None:
            cfg.record = sub.setdefault("emb_src", {})
        se = _embed(w, src, cfg)
        if sub is not None:
            cfg.record = sub.setdefault("emb_tgt", {})
        te = _embed(w, tgt, cfg)
        cfg.record = sub
        cfg.rec("src_emb0", se); cfg.rec("tgt_emb0", te)
        if cfg.pointer == "transformer":
            sp, tp = transformer_pointer(w, se, te, cfg)
            se, te = se + sp, te + tp                                       # :504-505
        elif cfg.pointer == "identity":                                     # Identity returns its inputs (:154-159)
            se, te = se + se, te + te
        cfg.rec("src_emb", se); cfg.rec("tgt_emb", te)

        def head(a_emb, b_emb, a, b):
            if cfg.vcp_nn == "topK":
                return head_topk(a_emb, b_emb, a, b, cfg)
            if cfg.vcp_nn == "att":
                return head_att(w, a_emb, b_emb, a, b)
            if cfg.vcp_nn == "dist":
                return head_by_dis(a_emb, b_emb, a, b)
            raise Exception("Not implemented")                              # :491

        srcK, corrK = head(se, te, src, tgt)                                # :507
        R_ab, t_ab = rigid_svd(srcK, corrK, cfg)                            # :509
        if cfg.cycle:
            sK2, cK2 = head(te, se, tgt, src)                               # :512
            R_ba, t_ba = rigid_svd(sK2, cK2)
        else:
            R_ba = R_ab.transpose(2, 1).contiguous()                        # :515
            t_ba = -torch.matmul(R_ba, t_ab.unsqueeze(2)).squeeze(2)        # :516
        return srcK, corrK, R_ab, t_ab, R_ba, t_ba


def vcrnet_iter(w: Weights, src: Tensor, tgt: Tensor, cfg: OracleConfig, iters: int = 1,
                forced_inputs=None, per_iter=None):
    """vcrnetIter, model/vcrnet_model.py:21-43.  ``forced_inputs`` (list of
    transformed_src, one per iteration) teacher-forces each pass;
    ``per_iter`` collects each pass's (R, t, transformed_src_in, selections) -- selections = the partial-mode
    index sets of that pass (kept keys per key cloud, overlap sets, arg-max targets, hard pairs) when
    ``cfg.record`` is a dict, else {}."""
    cur = src
    R_f = t_f = None
    for i in range(iters):
        if forced_inputs is not None:
            cur = forced_inputs[i]
        srcK, corrK, R, t, _, _ = vcrnet_forward(w, cur, tgt, cfg)
        if per_iter is not None:
            sel = {} if cfg.record is None else {
                n: cfg.record[n].clone() for n in ("key_keep_src", "key_keep_tgt", "sel_src", "sel_tgt", "argmax_tgt",
                                                   "pair_src") if n in cfg.record}
            per_iter.append((R, t, cur, sel))
        cur = transform_point_cloud(cur, R, t)                               # :28
        if R_f is None:
            R_f, t_f = R, t
        else:
            R_f = torch.matmul(R, R_f)                                       # :35
            t_f = torch.matmul(R, t_f.unsqueeze(2)).squeeze(2) + t           # :36-38
    R_ba = R_f.transpose(2, 1).contiguous()
    t_ba = -torch.matmul(R_ba, t_f.unsqueeze(2)).squeeze(2)
    return srcK, corrK, R_f, t_f, R_ba, t_ba


def icp_nearest(src: Tensor, dst: Tensor) -> Tuple[Tensor, Tensor]:
    """ICP.nearest_neighbor, model/icp_model.py:51-73: arg-max of the negative squared distance in the
    head's expanded form; returns (mean of the best values over the WHOLE batch, nearest dst points)."""
    d = neg_sqdist_head(src, dst)
    val, idx = d.topk(k=1, dim=-1)
    cand = torch.gather(dst, 2, idx.squeeze(-1).unsqueeze(1).expand(-1, 3, -1))
    return val.mean(), cand.contiguous()


def icp_forward(src_init: Tensor, dst: Tensor, max_iterations: int = 10, tolerance: float = 0.001,
                trace: Optional[list] = None):
    """ICP.forward, model/icp_model.py:26-48.  The stop test (:37) compares BATCH-mean errors."""
    src = src_init
    prev = 0
    for _ in range(max_iterations):
        err, corr = icp_nearest(src, dst)
        R, t = rigid_svd(src, corr)                                          # best_fit_transform, :75-108
        src = transform_point_cloud(src, R, t)
        if trace is not None:
            trace.append(float(err))
        if torch.abs(prev - err) < tolerance:
            break
        prev = err
    R, t = rigid_svd(src_init, src)                                          # :42
    R_ba = R.transpose(2, 1).contiguous()
    t_ba = -torch.matmul(R_ba, t.unsqueeze(2)).squeeze(2)
    return src_init, src, R, t, R_ba, t_ba


def vcrnet_icp(w: Weights, src: Tensor, tgt: Tensor, cfg: OracleConfig, max_iterations: int = 50):
    """vcrnetIcpNet, model/vcrnet_model.py:46-62 (the --iter 0 path): one network pass, ICP on the
    transformed source, poses composed."""
    _, _, R, t, _, _ = vcrnet_forward(w, src, tgt, cfg)
    moved = transform_point_cloud(src, R, t)
    _, _, Ri, ti, _, _ = icp_forward(moved, tgt, max_iterations=max_iterations)
    R2 = torch.matmul(Ri, R)                                                 # :55
    t2 = torch.matmul(Ri, t.unsqueeze(2)).squeeze(2) + ti                    # :56-57
    R_ba = R2.transpose(2, 1).contiguous()
    t_ba = -torch.matmul(R_ba, t2.unsqueeze(2)).squeeze(2)
    return moved, tgt, R2, t2, R_ba, t_ba


def dcp_forward(w: Weights, src: Tensor, tgt: Tensor, cfg: OracleConfig):
    """DCP.forward with head='svd', use_mFea=False: model/dcp_model.py:205-223 and the fused
    scoring + SVD head :126-174.  Output order (R_ab, t_ab, R_ba, t_ba, src, src_corr)."""
    with torch.no_grad():
        se, te = _embed(w, src, cfg), _embed(w, tgt, cfg)
        if cfg.pointer == "transformer":
            sp, tp = transformer_pointer(w, se, te, cfg)
            se, te = se + sp, te + tp
        elif cfg.pointer == "identity":
            se, te = se + se, te + te
        else:
            raise Exception("Not implemented")                              # dcp_model.py:195
        _, corr = head_by_dis(se, te, src, tgt)                             # :138-142
        R_ab, t_ab = rigid_svd(src, corr, cfg)                              # :144-173
        if cfg.cycle:
            _, corr2 = head_by_dis(te, se, tgt, src)
            R_ba, t_ba = rigid_svd(tgt, corr2)
        else:
            R_ba = R_ab.transpose(2, 1).contiguous()
            t_ba = -torch.matmul(R_ba, t_ab.unsqueeze(2)).squeeze(2)
        return R_ab, t_ab, R_ba, t_ba, src, corr
