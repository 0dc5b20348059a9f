"""Kernel-by-kernel forward for the module variants the single fused C entry point does not cover
(DGCNN embedding, partial-overlap mode, VcpAtt head, cycle consistency) and for stage-level parity
debugging.  Every arithmetic step is a libvcr_hip.so kernel; torch only allocates and slices."""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import native


def _sd(net) -> Dict[str, torch.Tensor]:
    return {k: v.detach().float().contiguous() for k, v in net.state_dict().items()}


def lpdnet_embed(net, x_cf: torch.Tensor, rec: Optional[dict] = None):
    """LPDNet.forward (model/lpdnet_model.py:103-137) on [Bc,3,N] -> rows [Bc*N, E], xyz4 [Bc*N,4]."""
    P = net._packed
    Bc, _, N = x_cf.shape
    k = int(net.emb_nn.k)
    xyz4, f64, sq = native.pointwise(x_cf, P["c1_w"], P["c1_b"], P["c2_w"], P["c2_b"])
    idx1 = native.knn(f64, sq, k)
    M = Bc * N
    pq1 = native.linear(f64.view(M, 64), P["dg1_wpq"], P["dg1_bpq"])
    cat = torch.empty(M, 512, dtype=torch.float32, device=x_cf.device)
    x1, x2 = native.edgeconv(pq1, idx1.view(M, k), N, P["dg2_w"], P["dg2_b"])
    idx3 = native.knn(xyz4, None, k)
    pq3 = native.linear(x2, P["sn1_wpq"], P["sn1_bpq"])
    x3 = native.gathermax(pq3, 256, idx3.view(M, k), N)
    cat[:, :128], cat[:, 128:256], cat[:, 256:] = x1, x2, x3
    emb = native.linear(cat, P["c3_w"], P["c3_b"], relu=True)
    if rec is not None:
        rec.update(x64=f64, idx_feat=idx1, idx_xyz=idx3, x1=x1, x2=x2, x3=x3, emb0=emb, xyz4=xyz4)
    return emb, xyz4.view(M, 4)


def _fold_bn(sd, conv: str, bn: str, eps: float = 1e-5):
    """Eval-mode BatchNorm folded into the preceding bias-free 1x1 conv: bn(conv(x)) = W'x + b'."""
    s = sd[bn + ".weight"] / torch.sqrt(sd[bn + ".running_var"] + eps)
    w = sd[conv + ".weight"].reshape(sd[conv + ".weight"].shape[0], -1)
    return (w * s.view(-1, 1)).contiguous(), (sd[bn + ".bias"] - sd[bn + ".running_mean"] * s).contiguous()


def dgcnn_embed(net, x_cf: torch.Tensor, rec: Optional[dict] = None):
    """DGCNN.forward (model/vcrnet_model.py:104-123, eval-mode BN folded): one Cartesian kNN, conv1 via the F7
    split (per-point P/Q + neighbour gather), conv2..conv4 as N*k GEMMs over per-edge rows, max over k after
    each, concat 512, conv5."""
    sd = _sd(net)
    Bc, _, N = x_cf.shape
    k = int(net.emb_nn.k)
    dev = x_cf.device
    M = Bc * N
    xyz4 = native.to_rows4(x_cf)
    idx = native.knn(xyz4, None, k)
    w1, b1 = _fold_bn(sd, "emb_nn.conv1", "emb_nn.bn1")                    # [64,6]: cols 0..2 neighbour, 3..5 centre
    wpq = torch.zeros(128, 32, dtype=torch.float32, device=dev)            # K padded 3 -> 32 for the MFMA tile
    wpq[:64, :3], wpq[64:, :3] = w1[:, :3], w1[:, 3:]
    xin = torch.zeros(M, 32, dtype=torch.float32, device=dev)
    xin[:, :3] = xyz4.view(M, 4)[:, :3]
    pq = native.linear(xin, wpq, torch.cat((torch.zeros_like(b1), b1)))
    cat = torch.empty(M, 512, dtype=torch.float32, device=dev)
    h = native.edgerows(pq, 64, idx.view(M, k), N)                         # relu(bn1(conv1(.)))  [M*k,64]
    native.segmax(h, M, k, out=cat[:, 0:64])
    col = 64
    for i, co in ((2, 64), (3, 128), (4, 256)):
        w, b = _fold_bn(sd, f"emb_nn.conv{i}", f"emb_nn.bn{i}")
        h = native.linear(h, w, b, relu=True)
        native.segmax(h, M, k, out=cat[:, col:col + co])
        col += co
    w5, b5 = _fold_bn(sd, "emb_nn.conv5", "emb_nn.bn5")
    emb = native.linear(cat, w5, b5, relu=True)
    if rec is not None:
        rec.update(idx_xyz=idx, cat=cat, emb0=emb, xyz4=xyz4)
    return emb, xyz4.view(M, 4)


def pointnet_embed(net, x_cf: torch.Tensor, rec: Optional[dict] = None):
    """PointNet.forward (model/vcrnet_model.py:81-87, eval-mode BN folded): five pointwise convs + ReLU, no graph."""
    sd = _sd(net)
    Bc, _, N = x_cf.shape
    M = Bc * N
    w = [_fold_bn(sd, f"emb_nn.conv{i}", f"emb_nn.bn{i}") for i in (1, 2, 3, 4, 5)]
    xyz4, h, _ = native.pointwise(x_cf, w[0][0], w[0][1], w[1][0], w[1][1])
    h = h.view(M, 64)
    for wi, bi in w[2:]:
        h = native.linear(h, wi, bi, relu=True)
    if rec is not None:
        rec.update(emb0=h, xyz4=xyz4)
    return h, xyz4.view(M, 4)


def transformer(net, emb: torch.Tensor, B: int, N: int, rec: Optional[dict] = None,
                key_keep_fn=None) -> torch.Tensor:
    """Both directions of model/transformer.py:264-272 on the 2B-batched rows [2B*N, E] (src then tgt).
    Returns the decoder output rows (pointer embeddings) in the same order."""
    P = net._packed
    E, H = net.emb_dims, net._n_heads
    sc = 1.0 / math.sqrt(E // H)
    nb = 2 * B
    ln = lambda x, f: native.layernorm(x, P[f + ".a"], P[f + ".b"])
    y = ln(emb, "enc_ln0")
    qkv = native.linear(y, P["enc_self.wqkv"], P["enc_self.bqkv"])
    att = native.sdpa(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], nb, H, N, N, sc)
    e1 = native.linear(att, P["enc_self.wo"], P["enc_self.bo"], residual=emb)
    hid = native.linear(ln(e1, "enc_ln1"), P["enc_ffn.w_1.weight"], P["enc_ffn.w_1.bias"], relu=True)
    e2 = native.linear(hid, P["enc_ffn.w_2.weight"], P["enc_ffn.w_2.bias"], residual=e1)
    mem = ln(e2, "enc_norm")
    y = ln(emb, "dec_ln0")
    qkv = native.linear(y, P["dec_self.wqkv"], P["dec_self.bqkv"])
    att = native.sdpa(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], nb, H, N, N, sc)
    d1 = native.linear(att, P["dec_self.wo"], P["dec_self.bo"], residual=emb)
    qc = native.linear(ln(d1, "dec_ln1"), P["dec_cross.wq"], P["dec_cross.bq"])
    kvc = native.linear(mem, P["dec_cross.wkv"], P["dec_cross.bkv"])
    keep = None
    if net._partial:
        # transformer.py:35-53: soft-max once, total probability mass each KEY receives over heads and queries,
        # keep the int(nk*overlap2) heaviest keys, soft-max again over those only.
        xs = torch.empty(nb, H, N, (N + 31) // 32 * 32, dtype=torch.float32, device=emb.device)
        _, rs = native.sdpa(qc, kvc[:, :E], None, nb, H, N, N, sc, kv_batch_shift=B, want_rowstat=True, pv=False,
                            score_out=xs)
        mass = native.keymass(xs, rs, N, B)               # keys of batch kb are attended by queries of batch (kb+B)%2B
        _, keep = native.rankselect(mass, int(N * net._overlap2), want_order=False, want_mask=True)
        if rec is not None:
            rec.update(key_mass=mass, key_keep=keep)
    att = native.sdpa(qc, kvc[:, :E], kvc[:, E:], nb, H, N, N, sc, kv_batch_shift=B, key_keep=keep)
    d2 = native.linear(att, P["dec_cross.wo"], P["dec_cross.bo"], residual=d1)
    hid = native.linear(ln(d2, "dec_ln2"), P["dec_ffn.w_1.weight"], P["dec_ffn.w_1.bias"], relu=True)
    d3 = native.linear(hid, P["dec_ffn.w_2.weight"], P["dec_ffn.w_2.bias"], residual=d2)
    if rec is not None:
        rec.update(mem=mem, e1=e1, e2=e2, d1=d1, d2=d2, d3=d3)
    return d3


def transformer_layers(net, emb: torch.Tensor, B: int, N: int, rec: Optional[dict] = None) -> torch.Tensor:
    """The same for ANY number of encoder / decoder layers (args.n_blocks, model/transformer.py:108-131,245,257-259): every
    encoder layer feeds the next, the decoder layers all attend to the encoder's final-norm output, and -- partial mode --
    every decoder layer prunes its own keys (clones() copies src_attn.is_src, transformer.py:9-10,252-255).  Weights are
    read per layer from the state dict; used for n_blocks != 1, which the one-call driver does not cover."""
    sd = _sd(net)
    E, H = net.emb_dims, net._n_heads
    sc = 1.0 / math.sqrt(E // H)
    nb = 2 * B
    pre = "pointer.model."
    ln = lambda x, name: native.layernorm(x, sd[name + ".a_2"], sd[name + ".b_2"])
    lin = lambda x, name, **kw: native.linear(x, sd[name + ".weight"], sd[name + ".bias"], **kw)

    def ffn(x, lp, sub):
        hid = lin(ln(x, f"{lp}sublayer.{sub}.norm"), lp + "feed_forward.w_1", relu=True)
        return lin(hid, lp + "feed_forward.w_2", residual=x)

    def self_attn(x, lp):
        y = ln(x, lp + "sublayer.0.norm")
        q, k, v = (lin(y, f"{lp}self_attn.linears.{i}") for i in range(3))
        return lin(native.sdpa(q, k, v, nb, H, N, N, sc), lp + "self_attn.linears.3", residual=x)

    x = emb
    for i in range(net.pointer.N):
        lp = f"{pre}encoder.layers.{i}."
        x = ffn(self_attn(x, lp), lp, 1)
    mem = ln(x, pre + "encoder.norm")
    x = emb
    for i in range(net.pointer.N):
        lp = f"{pre}decoder.layers.{i}."
        d1 = self_attn(x, lp)
        qc = lin(ln(d1, lp + "sublayer.1.norm"), lp + "src_attn.linears.0")
        kc, vc = lin(mem, lp + "src_attn.linears.1"), lin(mem, lp + "src_attn.linears.2")
        keep = None
        if net._partial:                                   # transformer.py:35-53, per layer
            xs = torch.empty(nb, H, N, (N + 31) // 32 * 32, dtype=torch.float32, device=emb.device)
            _, rs = native.sdpa(qc, kc, None, nb, H, N, N, sc, kv_batch_shift=B, want_rowstat=True, pv=False, score_out=xs)
            mass = native.keymass(xs, rs, N, B)
            _, keep = native.rankselect(mass, int(N * net._overlap2), want_order=False, want_mask=True)
            if rec is not None:
                rec[f"key_keep.{i}"] = keep
        att = native.sdpa(qc, kc, vc, nb, H, N, N, sc, kv_batch_shift=B, key_keep=keep)
        x = ffn(lin(att, lp + "src_attn.linears.3", residual=d1), lp, 2)
    if rec is not None:
        rec.update(mem=mem, d3=x)
    return x


def _partial_head(net, src, embf, side4, B: int, N: int, rec: Optional[dict]):
    """VcpTopK partial mode: selectCom (model/vcrnet_model.py:190-262) + getCopair (:264-332) + SVD.
    The score matrix is computed and written ONCE (with its row soft-max statistics); the two soft-maxes are
    never written: a column pass and a row pass over S give the column / row probability sums, rankselect the
    overlap sets, one more STATS pass (with arg-max) on the reduced sets the hard correspondences."""
    M1 = B * N
    o2 = net._overlap2
    se, te, ss, ts = embf[:M1], embf[M1:], side4[:M1], side4[M1:]
    src_k = int(N * 0.84 * o2)                                             # :208
    tgt_k = int(N * 0.84 * o2)                                             # :209
    # score_ij = (-|s_i|^2 + 2 s_i.t_j) - |t_j|^2 for BOTH soft-maxes (one matrix in the reference, :211-216):
    # owner = src -> score form 0; owner = tgt -> form 2 (streamed-side norm first), the same association.
    # one score GEMM pass keeps S and the row soft-max statistics; the column statistics and both probability
    # masses then come from two HBM-bound passes over S
    S = torch.empty(B, N, (N + 31) // 32 * 32, dtype=torch.float32, device=embf.device)
    rstat, _ = native.pairscore(se, te, B, N, N, op=1, score=0, own_side4=ss, str_side4=ts, score_out=S)   # softmax(dim=2)
    _, colsum, rowsum = native.scoremass(S, N, rstat)                      # :222, :244
    idx_t, _ = native.rankselect(colsum, tgt_k)                            # :223
    idx_s, _ = native.rankselect(rowsum, src_k)                            # :245
    so_e, to_e = native.gather_rows(se, idx_s, B, N), native.gather_rows(te, idx_t, B, N)        # :251-260,:235-238
    so_s, to_s = native.gather_rows(ss, idx_s, B, N), native.gather_rows(ts, idx_t, B, N)
    # getCopair on the overlap sets: peak soft-max probability = 1/l and its arg-max target (:295-298)
    st, amax = native.pairscore(so_e, to_e, B, src_k, tgt_k, op=1, score=0, own_side4=so_s, str_side4=to_s,
                                want_argmax=True)
    k2 = int(src_k * 0.52 * o2)                                            # :284
    lsum = st.view(B, src_k, 2)[:, :, 1].contiguous()
    pick, _ = native.rankselect(lsum, k2, largest=False)                   # largest peak prob == smallest l (:312)
    pair_t = torch.gather(amax.view(B, src_k), 1, pick.long()).int()       # index plumbing only
    srcK4 = native.gather_rows(so_s, pick, B, src_k)                       # :328-330
    corr4 = native.gather_rows(to_s, pair_t, B, tgt_k)                     # :325 (weights are exactly 1)
    R, t, Rb, tb = native.rigid_svd(srcK4.view(B, k2, 4), corr4.view(B, k2, 4))
    if rec is not None:
        rec.update(sel_src=idx_s, sel_tgt=idx_t, pair_src=pick, pair_tgt=pair_t, colsum=colsum, rowsum=rowsum,
                   peak_l=lsum, argmax_tgt=amax.view(B, src_k))
    tr = lambda x: x.view(B, k2, 4)[:, :, :3].transpose(1, 2).contiguous()
    return tr(srcK4), tr(corr4), R, t, Rb, tb


def forward_composed(net, src: torch.Tensor, tgt: torch.Tensor, rec: Optional[dict] = None):
    """VCRNet.forward (model/vcrnet_model.py:495-518), one kernel per step."""
    P = net._packed
    B, _, N = src.shape
    x = torch.cat((src, tgt), 0).contiguous().float()
    if net.cycle and (net._partial or net._vcp == "att"):
        raise native.VcrHipError("cycle=True is built for the whole-mode topK / dist heads only")
    emb, xyz4 = {"lpdnet": lpdnet_embed, "dgcnn": dgcnn_embed, "pointnet": pointnet_embed}[net._emb_kind](net, x, rec)
    M1 = B * N
    if P and "dec_norm.a" in P:
        d3 = (transformer if net.pointer.N == 1 else transformer_layers)(net, emb, B, N, rec)
        embf, side4 = native.layernorm(d3, P["dec_norm.a"], P["dec_norm.b"], residual=emb, xyz4=xyz4)
    else:
        from .module import _Identity
        embf, side4 = native.rowside(emb, xyz4, 2.0 if isinstance(net.pointer, _Identity) else 1.0)
    if rec is not None:
        rec.update(embf=embf, side4=side4)
    if net._vcp == "att":                                                  # VcpAtt: vcrnet_model.py:444-449
        sd = _sd(net)
        q = native.linear(embf[:M1], sd["head.linears_emb.0.weight"], sd["head.linears_emb.0.bias"])
        kx = native.linear(embf[M1:], sd["head.linears_emb.1.weight"], sd["head.linears_emb.1.bias"])
        embf = torch.cat((q, kx), 0)
        _, side4 = native.rowside(embf, xyz4, 1.0)
    if net._partial and net._vcp == "topK":
        return _partial_head(net, src, embf, side4, B, N, rec)
    mode = 1 if net._vcp == "dist" else 0
    scale = 1.0 / math.sqrt(net.emb_dims)

    def head_and_solve(a0, b0):
        corr4 = native.softcorr(embf[a0:a0 + M1], embf[b0:b0 + M1], side4[a0:a0 + M1], side4[b0:b0 + M1], B, N, N,
                                mode=mode, scale=scale)
        R, t, Rb, tb = native.rigid_svd(xyz4[a0:a0 + M1].view(B, N, 4), corr4.view(B, N, 4))
        return corr4, R, t, Rb, tb

    corr4, R_ab, t_ab, R_ba, t_ba = head_and_solve(0, M1)
    if net.cycle:                                                          # vcrnet_model.py:511-513
        _, R_ba, t_ba, _, _ = head_and_solve(M1, 0)
    src_corr = corr4.view(B, N, 4)[:, :, :3].transpose(1, 2).contiguous()
    return src, src_corr, R_ab, t_ab, R_ba, t_ba
