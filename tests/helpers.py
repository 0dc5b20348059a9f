"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def set_mismatch(a, b):
    """Number of rows whose index SETS differ, for [.., K] integer arrays."""
    a = np.sort(np.asarray(a).reshape(-1, a.shape[-1]).astype(np.int64), axis=1)
    b = np.sort(np.asarray(b).reshape(-1, b.shape[-1]).astype(np.int64), axis=1)
    return int((a != b).any(axis=1).sum())


def sl(t, cs):
    return t[:, ::cs, :].contiguous().numpy()


def cfg_weights(regime="default", **kw):      # kw: emb_nn / vcp_nn / pointer / n_blocks, and seed / scale overrides
    """regime "default" = generate_weights(1234, lpd fixture): the weights of every fixture recorded before round 4."""
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import weights
    return weights.regime_weights(regime, lpd=weights.load_lpd_fixture(), **kw)


REGIMES = ("seed4321", "trained", "randemb")          # the non-default points of vcrnet_amd.weights.REGIMES
