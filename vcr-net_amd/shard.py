"""Process-per-GPU batch sharding: the replacement for the reference's nn.DataParallel
(util/initPara.py:260).  Every op of VCRNet.forward is per-sample in eval mode, so
pairs are sharded contiguously over ranks with NO data-path collective; the only exchange is one
all-gather of the per-rank poses (12 floats per pair: R row-major + t) -- RCCL over xGMI on the GPU
box (backend "nccl"), gloo in the CPU tests."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of `total` items owned by `rank`; the first total % world ranks get one extra."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_pose(R: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    return torch.cat((R.reshape(R.shape[0], 9), t.reshape(t.shape[0], 3)), 1).contiguous()


def unpack_pose(p: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    return p[:, :9].reshape(-1, 3, 3), p[:, 9:12]


def all_gather_poses(pose: torch.Tensor, world: int) -> torch.Tensor:
    """[b,12] per rank (equal b on every rank) -> [world*b,12] on every rank, in rank order."""
    if world == 1:
        return pose
    if pose.is_cuda and dist.get_backend() != "nccl":
        # CPU-only backends (gloo in tests): stage the few hundred bytes through the host
        host = pose.detach().cpu().contiguous()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host)
        return torch.cat(parts, 0).to(pose.device)
    out = torch.empty((world * pose.shape[0], pose.shape[1]), dtype=pose.dtype, device=pose.device)
    dist.all_gather_into_tensor(out, pose.contiguous())
    return out


def all_gather_ragged(pose: torch.Tensor, total: int, world: int) -> torch.Tensor:
    """Uneven shards (total % world != 0): pad to the largest shard, gather, drop the padding."""
    if world == 1:
        return pose
    per = -(-total // world)
    pad = torch.zeros((per, pose.shape[1]), dtype=pose.dtype, device=pose.device)
    pad[: pose.shape[0]] = pose
    g = all_gather_poses(pad, world).view(world, per, -1)
    parts = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        parts.append(g[r, : hi - lo])
    return torch.cat(parts, 0)
