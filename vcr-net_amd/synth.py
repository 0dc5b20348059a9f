"""Synthetic registration pairs.

ModelNet40 is not available offline, so base clouds are synthesised;
the rigid transform / permutation / partial-crop recipe applied to them restates
``ModelNet40.__getitem__`` of the reference (util/data.py:247-314) for the
``partition != 'train'``, ``model != 'lpd'`` branch, with the item index as the
legacy-NumPy seed exactly as the reference does (util/data.py:255-256).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import numpy as np

# Constants util/initPara.py:115-124 derives with sympy for --overlap 0.575.
RESERVE_0575 = 0.750681278255248
OVERLAP2_0575 = 0.765970880926229


def base_cloud(item: int, n: int = 2048) -> np.ndarray:
    """A ModelNet-like object: points on a seeded union of boxes / cylinders / spheres,
    zero-mean and scaled into the unit sphere (the modelnet40_ply_hdf5_2048 convention).
    Returns float32 [n, 3].  Uses its own random stream so the transform stream of
    :func:`make_pair` matches the reference's call order."""
    rs = np.random.RandomState(1_000_003 + 7919 * item)
    parts = rs.randint(2, 5)
    share = rs.dirichlet(np.ones(parts) * 2.0)
    counts = np.floor(share * n).astype(int)
    counts[0] += n - counts.sum()
    pts = []
    for c in counts:
        kind = rs.randint(3)
        centre = rs.uniform(-0.5, 0.5, 3)
        scale = rs.uniform(0.2, 0.7, 3)
        if kind == 0:      # box surface
            p = rs.uniform(-1, 1, (c, 3))
            face = rs.randint(3, size=c)
            p[np.arange(c), face] = np.sign(rs.uniform(-1, 1, c))
        elif kind == 1:    # cylinder side
            th = rs.uniform(0, 2 * np.pi, c)
            p = np.stack([np.cos(th), np.sin(th), rs.uniform(-1, 1, c)], 1)
        else:              # sphere
            v = rs.normal(size=(c, 3))
            p = v / np.linalg.norm(v, axis=1, keepdims=True)
        pts.append(p * scale + centre)
    cloud = np.concatenate(pts, 0)
    cloud = cloud - cloud.mean(0, keepdims=True)
    cloud = cloud / np.max(np.linalg.norm(cloud, axis=1))
    return cloud.astype(np.float32)


def uniform_cloud(item: int, n: int) -> np.ndarray:
    """BASELINE configs 4/5: U(-0.5, 0.5)^3 points."""
    rs = np.random.RandomState(2_000_003 + 7919 * item)
    return rs.uniform(-0.5, 0.5, (n, 3)).astype(np.float32)


def _nearest_crop(cloud3n: np.ndarray, reserve: float) -> np.ndarray:
    """util/data.py:320-329: keep the int(num*reserve) points nearest to the LAST point
    (brute force instead of sklearn; ordered by distance like kneighbors)."""
    pts = cloud3n.T
    num = int(max(pts.shape) * reserve)
    d = np.sum((pts - pts[-1:]) ** 2, axis=1)
    order = np.argsort(d, kind="stable")[:num]
    return pts[order].T


@dataclass
class Pair:
    src: np.ndarray        # [3, N] float32
    tgt: np.ndarray        # [3, N]
    R_ab: np.ndarray       # [3, 3]
    t_ab: np.ndarray       # [3]
    euler_ab: np.ndarray   # [3] (z, y, x) radians


@dataclass
class Draws:
    """The random numbers of one item, in the reference's call order (util/data.py:255-301)."""
    cloud: np.ndarray      # [P, 3] float32 base cloud
    R_ab: np.ndarray       # [3, 3] float64
    t_ab: np.ndarray       # [3] float64
    euler_ab: np.ndarray   # [3] (z, y, x) radians
    pick: np.ndarray       # [N] int32: first N rows of permutation(cloud)      (:289)
    perm_src: np.ndarray   # [N] int32                                         (:298)
    perm_tgt: np.ndarray   # [N] int32                                         (:301)


def draw_pair(item: int, num_points: int = 1024, factor: float = 4.0, kind: str = "object") -> Draws:
    """Everything random about one evaluation item.  RandomState.permutation(array) and permutation(len)
    consume the stream identically, so index permutations stand in for the reference's row shuffles."""
    cloud = base_cloud(item) if kind == "object" else uniform_cloud(item, max(num_points, 2048))
    if num_points > cloud.shape[0]:
        raise ValueError(f"object clouds hold {cloud.shape[0]} points; use kind='uniform' for num_points={num_points}")
    rs = np.random.RandomState(item)                                   # :255-256
    ax, ay, az = (rs.uniform() * np.pi / factor for _ in range(3))     # :258-260
    cx, cy, cz, sx, sy, sz = np.cos(ax), np.cos(ay), np.cos(az), np.sin(ax), np.sin(ay), np.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    R_ab = Rx.dot(Ry).dot(Rz)                                          # :277
    t_ab = np.array([rs.uniform(-0.5, 0.5) for _ in range(3)])         # :284-285
    pick = rs.permutation(cloud.shape[0])[:num_points]                 # :289
    perm_src = rs.permutation(num_points)                              # :298
    perm_tgt = rs.permutation(num_points)                              # :301
    return Draws(cloud, R_ab, t_ab, np.asarray([az, ay, ax], dtype=np.float32), pick.astype(np.int32),
                 perm_src.astype(np.int32), perm_tgt.astype(np.int32))


def inverse_labels(R_ab: np.ndarray, t_ab: np.ndarray, euler_ab: np.ndarray):
    """The B -> A ground truth the reference's loader returns next to (R_ab, t_ab, euler_ab): R_ba = R_ab^T,
    t_ba = -R_ba t_ab, euler_ba = -euler_ab[::-1] (util/data.py:278,286,295).  Batched [B,...] or single."""
    R_ba = np.swapaxes(R_ab, -1, -2)
    t_ba = -np.einsum("...ij,...j->...i", R_ba, t_ab)
    return R_ba, t_ba, -euler_ab[..., ::-1]


def make_pair(item: int, num_points: int = 1024, partial: bool = False, reserve: float = RESERVE_0575,
              factor: float = 4.0, kind: str = "object") -> Pair:
    """One evaluation item on the host (util/data.py:258-303)."""
    d = draw_pair(item, num_points, factor, kind)
    p1 = d.cloud[d.pick].T                                             # :289
    p2 = d.R_ab.dot(p1.astype(np.float64)) + d.t_ab[:, None]           # :290-291
    p1 = p1[:, d.perm_src]                                             # :298
    if partial:
        p1 = _nearest_crop(p1, reserve)                                # :299-300
    p2 = p2[:, d.perm_tgt]                                             # :301
    if partial:
        p2 = _nearest_crop(p2, reserve)                                # :302-303
    return Pair(p1.astype(np.float32), p2.astype(np.float32), d.R_ab.astype(np.float32),
                d.t_ab.astype(np.float32), d.euler_ab)


def make_batch(first_item: int, batch: int, num_points: int = 1024, partial: bool = False,
               kind: str = "object") -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Stack ``batch`` consecutive items -> (src [B,3,N], tgt [B,3,N], R_ab, t_ab, euler_ab)."""
    ps = [make_pair(first_item + i, num_points, partial=partial, kind=kind) for i in range(batch)]
    return (np.stack([p.src for p in ps]), np.stack([p.tgt for p in ps]),
            np.stack([p.R_ab for p in ps]), np.stack([p.t_ab for p in ps]),
            np.stack([p.euler_ab for p in ps]))


def make_batch_device(first_item: int, batch: int, num_points: int = 1024, partial: bool = False,
                      reserve: float = RESERVE_0575, kind: str = "object", device="cuda"):
    """The same batch as :func:`make_batch`, with the point arithmetic (gather, rigid transform, partial crop)
    done by vcr_make_pairs_f32 on base clouds resident in HBM.  Returns device tensors
    (src [B,3,K], tgt [B,3,K]) and host (R_ab, t_ab, euler_ab)."""
    import torch

    from . import native
    ds = [draw_pair(first_item + i, num_points, kind=kind) for i in range(batch)]
    dev = torch.device(device)
    up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(np.stack(a))).to(dev, dt)
    cloud = up([d.cloud for d in ds], torch.float32)
    keep = int(num_points * reserve) if partial else num_points        # util/data.py:321
    src, tgt = native.make_pairs(cloud, up([d.R_ab for d in ds], torch.float64), up([d.t_ab for d in ds], torch.float64),
                                 up([d.pick for d in ds], torch.int32), up([d.perm_src for d in ds], torch.int32),
                                 up([d.perm_tgt for d in ds], torch.int32), keep)
    return (src, tgt, np.stack([d.R_ab for d in ds]).astype(np.float32),
            np.stack([d.t_ab for d in ds]).astype(np.float32), np.stack([d.euler_ab for d in ds]))
