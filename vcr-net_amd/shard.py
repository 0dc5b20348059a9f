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


# ---- which transport did RCCL choose?  (BASELINE north_star: "RCCL all-gather ... over xGMI only") ----------------------
# rank 0 runs with NCCL_DEBUG=INFO, NCCL_DEBUG_SUBSYS=INIT,GRAPH and NCCL_DEBUG_FILE=<log> (bench.py sets them before the
# process group exists); after the first collective the log holds one line per channel and peer,
#     "... NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC"      (or P2P/direct pointer, P2P/CUMEM, SHM/direct/direct,
#                                                                    NET/Socket/0, NET/IB/0/GDRDMA ...)
# and the topology RCCL detected ("... + XGMI[48.0] - GPU/..." per link on an xGMI box, "PCI[..]" otherwise).

def rccl_debug_env(log_path: str) -> dict:
    """Environment of the rank whose RCCL log is parsed (set BEFORE torch.distributed.init_process_group)."""
    return {"NCCL_DEBUG": "INFO", "NCCL_DEBUG_SUBSYS": "INIT,GRAPH", "NCCL_DEBUG_FILE": log_path}


def parse_rccl_log(text: str) -> dict:
    """{"channels": n, "transports": {"P2P/IPC": n, ...}, "xgmi_links_in_topology": n, "net_or_shm": [...], "xgmi_only": bool|None}.
    xgmi_only: every channel of every peer is a P2P transport (no NET/..., no SHM/...) AND the detected topology lists
    XGMI links; None when the log holds no channel line at all (nothing can be said)."""
    import re
    transports: dict = {}
    for m in re.finditer(r"Channel\s+\d+(?:/\d+)?\s*:\s*\d+\[[0-9a-fA-F]+\]\s*->\s*\d+\[[0-9a-fA-F]+\]\s*(?:\[(?:send|receive)\]\s*)?via\s+([A-Za-z0-9_]+(?:/[A-Za-z0-9_ ]+?)*)\s*(?:$|\n|comm|,)",
                         text):
        key = m.group(1).strip()
        transports[key] = transports.get(key, 0) + 1
    n = sum(transports.values())
    bad = sorted(k for k in transports if not k.upper().startswith("P2P"))
    xgmi = len(re.findall(r"XGMI\[", text))
    return {"channels": n, "transports": transports, "xgmi_links_in_topology": xgmi, "net_or_shm": bad,
            "xgmi_only": None if n == 0 else (not bad and xgmi > 0)}
