"""Algorithmic work per kernel launch of the whole-path forward (SURVEY.md section 8d figures):
FLOPs and kernel-boundary bytes as functions of (B pairs, N points, k, E, F).  Used by bench.py to
turn measured launch durations into roofline fractions.  Peaks from MI355X_MICROARCH.md:
fp32 matrix (v_mfma_f32_32x32x2_f32) 157.3 TFLOP/s, HBM3E 8 TB/s.

Bytes come in two kinds.  ``launch_work`` returns the HBM-COMPULSORY bytes of a launch: every distinct input row
read once, every output written once.  The neighbour gathers of the EdgeConv kernels re-read each row ~k times;
those re-reads are served by L2 / MALL, not HBM, and are returned separately by ``gather_bytes`` (SURVEY 8d counts
them as kernel-boundary traffic, which is why its per-pair figure -- used for the kNN+EdgeConv stage line -- is
larger than the sum of the compulsory bytes).  Pricing the gathers against the HBM peak gave >8 TB/s in round 1."""
from __future__ import annotations

from typing import Dict, Tuple

PEAK_MFMA_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
PEAK_MFMA_BF16_TFLOPS = 2500.0

# which roofline bounds each kernel family
FAMILY_BOUND = {"pointwise": "hbm", "knn": "mfma", "linear": "mfma", "edgeconv": "mfma", "gathermax": "hbm",
                "layernorm": "hbm", "sdpa": "mfma", "softcorr": "mfma", "rigid_svd": "hbm",
                "pairscore": "mfma", "scoremass": "hbm", "select": "hbm", "pose": "hbm"}

_LINEAR_SHAPES = {  # site -> (N_out, K) as multiples resolved below
    "dg1_pq": (256, 64), "sn1_pq": (512, 128), "conv3": ("E", 512),
    "pn_c3": (64, 64), "pn_c4": (128, 64), "pn_c5": ("E", 128),             # emb_nn = pointnet
    "enc.qkv": ("3E", "E"), "enc.wo": ("E", "E"), "enc.ffn1": ("F", "E"), "enc.ffn2": ("E", "F"),
    "dec.qkv": ("3E", "E"), "encdec.qkv": ("6E", "E"), "dec.self.wo": ("E", "E"), "dec.cross.q": ("E", "E"), "dec.cross.kv": ("2E", "E"),
    "dec.cross.wo": ("E", "E"), "dec.ffn1": ("F", "E"), "dec.ffn2": ("E", "F"),
}


def overlap_sizes(N: int, overlap2: float) -> Tuple[int, int, int]:
    """(kept keys, overlap-set size K1, hard pairs K2) of partial mode: transformer.py:41, vcrnet_model.py:208,284."""
    k1 = int(N * 0.84 * overlap2)
    return int(N * overlap2), k1, int(k1 * 0.52 * overlap2)


def launch_work(name: str, B: int, N: int, k: int = 20, E: int = 512, F: int = 1024,
                overlap2: float = 0.765970880926229) -> Tuple[float, float]:
    """(flops, bytes) of one launch called `name` ("family:site") for B pairs of N points.  A name ending in "@src" is the
    launch of a later vcrnetIter pass that runs on the source clouds' rows only (target reuse): half the work."""
    if name.endswith("@src"):
        fl, by = launch_work(name[:-4], B, N, k, E, F, overlap2)
        return 0.5 * fl, 0.5 * by
    fam, site = name.split(":", 1)
    M1, M2 = B * N, 2 * B * N
    sym = {"E": E, "2E": 2 * E, "3E": 3 * E, "6E": 6 * E, "F": F}
    r = lambda v: sym[v] if isinstance(v, str) else v
    if fam == "pointwise":
        if site == "pad":
            return 0.0, 4.0 * M2 * (4 + 32)
        g = 2 if site.startswith("src+tgt") else 1         # both clouds in one launch
        if site.endswith("+dg1_pq"):                       # ... with the first EdgeConv's P | Q projection (K = 64, N = 256)
            return g * 2.0 * M1 * (3 * 64 + 64 * 64 + 64 * 256), g * 4.0 * M1 * (3 + 4 + 64 + 64 + 1 + 256)        # (+ 64: feat64's transposed copy for the kNN)
        return g * 2.0 * M1 * (3 * 64 + 64 * 64), g * 4.0 * M1 * (3 + 4 + 64 + 1)
    if fam == "knn":
        if site == "feat64+xyz":                           # both searches in one launch
            return 2.0 * (64 + 3) * N * N * 2 * B, 4.0 * M2 * (64 + 1 + 4 + 2 * k)
        if site == "ties":                                 # the replay of the few rows with a boundary tie: latency only
            return 0.0, 4.0 * M2
        if site == "rank":                                 # Morton ranking + the rows in rank order + the tiles' balls (ordered search)
            return 0.0, 4.0 * M2 * (4 + 2 * (64 + 1 + 4) + 1 + 6)
        C = 64 if site == "feat64" else 3
        return 2.0 * C * N * N * 2 * B, 4.0 * M2 * ((C if C == 64 else 4) + (1 if C == 64 else 0) + k)
    if fam == "linear":
        if site.startswith("dg_c"):                        # DGCNN: conv1 split per point, conv2..4 on the N*k edge rows
            n, kk, rows = {"dg_c1_pq": (128, 32, M2), "dg_c2": (64, 64, M2 * k), "dg_c3": (128, 64, M2 * k),
                           "dg_c4": (256, 128, M2 * k)}[site]
            out_rows = M2 if site == "dg_c4" else rows     # conv4's per-edge rows are never stored: only the fused max
            return 2.0 * rows * n * kk, 4.0 * (rows * kk + n * kk + out_rows * n)
        if site.startswith("head.att"):                    # VcpAtt's two Linear(E,E), one cloud each
            return 2.0 * M1 * E * E, 4.0 * (2 * M1 * E + E * E)
        if "+" in site:                                    # two independent linears as one launch (vcr_linear_pair_f32)
            parts = [launch_work("linear:" + p, B, N, k, E, F, overlap2) for p in site.split("+")]
            return sum(p[0] for p in parts), sum(p[1] for p in parts)
        n, kk = (r(v) for v in _LINEAR_SHAPES[site])
        return 2.0 * M2 * n * kk, 4.0 * (M2 * kk + n * kk + M2 * n)
    if fam == "edgeconv" and site == "dg_chain":
        # DGCNN conv1-gather .. conv4 in one kernel: P|Q rows [M2,128] once, idx, the 512-wide maxima out
        return 2.0 * M2 * k * (64 * 64 + 128 * 64 + 256 * 128), 4.0 * M2 * (128 + k + 512)
    if fam == "edgeconv":      # convDG2 on the per-edge features (the per-point half of convDG1 is linear:dg1_pq)
        # compulsory: P|Q rows [M2,256] once, idx, x1 and x2 out
        return 2.0 * M2 * k * 128 * 128, 4.0 * M2 * (256 + k + 256)
    if fam == "gathermax":
        if site == "dg_c1":                                # edge rows materialised: P|Q once, idx, [M2*k,64] out, x1 + zero base
            return 1.0 * M2 * k * 64, 4.0 * M2 * (128 + k + k * 64 + 512)
        if site.startswith("dg_"):                         # segmented max over stored edge rows: streamed once
            C = {"dg_max1": 64, "dg_max2": 64, "dg_max3": 128, "dg_max4": 256}[site]
            return 1.0 * M2 * k * C, 4.0 * M2 * (k * C + C)
        return 1.0 * M2 * k * 256, 4.0 * M2 * (512 + k + 256)   # P|Q rows [M2,512] once, idx, x3 out
    if fam == "layernorm":
        if site.startswith("rowside"):
            return 2.0 * M2 * E, 4.0 * M2 * E * 2
        extra = 1 if site.endswith("+res") else 0
        return 8.0 * M2 * E, 4.0 * M2 * E * (2 + extra)
    keep, K1, K2 = overlap_sizes(N, overlap2)
    if fam == "sdpa":
        if site == "dec.cross.stats":                      # QK^T + row statistics only
            return 2.0 * 2 * B * N * N * E, 4.0 * M2 * E * 2
        g = 2 if site == "encdec.self" else 1               # encoder's and decoder's self-attention as one grouped launch
        return g * 4.0 * 2 * B * N * N * E, g * 4.0 * M2 * E * 4
    if fam == "pairscore":
        if site == "dec.cross.keymass":                    # one head: 128-d scores of every (key, query) pair
            return 2.0 * 2 * B * N * N * (E // 4), 4.0 * M2 * (E // 4) * 2
        if site == "head.scores":                          # 512-d scores, stored once
            return 2.0 * B * N * N * E + 6.0 * B * N * N, 4.0 * (M2 * E + B * N * N)
        if site == "head.copair":
            return 2.0 * B * K1 * K1 * E, 4.0 * 2 * B * K1 * E
        return 2.0 * B * N * N * E, 4.0 * M2 * E           # row/col statistics or mass passes that recompute the scores
    if fam == "scoremass":
        if site == "dec.cross.keymass":                    # one read of the [2B,H,N,N] scores
            return 4.0 * 2 * B * 4 * N * N, 4.0 * 2 * B * 4 * N * N
        return 8.0 * B * N * N, 4.0 * 2 * B * N * N
    if fam == "select":
        if site.endswith(".forced") or site.endswith(".out"):   # device-to-device copy of an index block
            return 0.0, 8.0 * 2 * B * N
        if site.startswith("gather"):
            width = E if site.endswith("emb") else 4
            if site in ("gather.emb", "gather.xyz"):        # both clouds' overlap sets in one launch
                return 0.0, 4.0 * 2 * 2 * B * K1 * width
            rows = K1 if ("src_" in site or "tgt_" in site) else K2
            return 0.0, 4.0 * 2 * B * rows * width
        n = K1 if site == "head.pairs" else N
        nb = 2 * B if site in ("dec.cross.keys", "head.src+tgt") else B
        return 1.0 * nb * n * n, 4.0 * nb * n * 2
    if fam == "pose":
        return 18.0 * M1, 4.0 * M1 * 6
    if fam == "softcorr":
        return 2.0 * B * N * N * E + 6.0 * B * N * N, 4.0 * (M2 * E + M2 * 4)
    if fam == "rigid_svd":
        return 18.0 * M1, 4.0 * M1 * 8                      # (partial mode solves on K2 pairs: even less)
    raise KeyError(name)


def gather_bytes(name: str, B: int, N: int, k: int = 20) -> float:
    """Bytes a launch moves through L2 for its neighbour gathers (k rows per point, each row re-read ~k times across
    the cloud): NOT HBM traffic.  edgeconv gathers 128-float P rows, gathermax:sn1 256-float rows, DGCNN's
    edge-row builder 64-float rows."""
    if name.endswith("@src"):
        return 0.5 * gather_bytes(name[:-4], B, N, k)
    fam, site = name.split(":", 1)
    M2 = 2 * B * N
    if fam == "edgeconv":
        return 4.0 * M2 * k * (64 if site == "dg_chain" else 128)
    if fam == "gathermax" and site == "sn1":
        return 4.0 * M2 * k * 256
    if fam == "gathermax" and site == "dg_c1":
        return 4.0 * M2 * k * 64
    return 0.0


def reference_flops_per_pair(N: int, k: int = 20, E: int = 512, F: int = 1024) -> Dict[str, float]:
    """SURVEY section 8d per-pair FLOP count of the REFERENCE formulation (un-split EdgeConv)."""
    emb = 2 * N * (3 * 64 + 64 * 64) + 2 * 64 * N * N + 2 * (2 * 128 * 128 * N * k) + 6 * N * N \
        + 2 * 256 * 256 * N * k + 2 * 512 * 512 * N
    tr = 2 * (3 * (8 * E * E * N + 4 * E * N * N) + 2 * (4 * E * F * N))
    head = 2 * E * N * N + 6 * N * N
    return {"emb": 2.0 * emb, "transformer": float(tr), "head": float(head), "svd": 18.0 * N,
            "total": 2.0 * emb + tr + head + 18.0 * N}
