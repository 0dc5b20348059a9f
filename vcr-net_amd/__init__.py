"""vcr-net_amd: MI355X-native implementation of the VCR-Net registration hot path.

Only what the path needs lives here: ``csrc/`` (hand-written HIP kernels for gfx950
behind the C-ABI of ``include/vcr_hip.h``), ``native`` (ctypes binding of that ABI),
``module`` (the host-side mirror of the reference's ``VCRNet`` nn.Module contract),
``weights`` / ``synth`` (parameter inventory, deterministic weights, synthetic pairs),
``shard`` (process-per-GPU batch sharding) and ``evalmetrics`` (the reference's eval arithmetic).

The directory name carries a hyphen; import it as ``vcrnet_amd`` (see ``vcrnet_amd.py`` at
the repository root).
"""
from . import weights, synth  # noqa: F401

__all__ = ["weights", "synth"]
