#!/usr/bin/env python
"""Record what the REFERENCE's own data recipe produces (ModelNet40.__getitem__, util/data.py:247-314, with the
partial crop nearest_neighbor :320-329) for the synthetic base clouds of vcrnet_amd.synth, so that
synth.make_pair / make_batch (and through it the device generator vcr_make_pairs_f32) are pinned to the reference
instead of to themselves.

Runs only in the build container (needs /root/reference).  util/data.py imports h5py (absent here) only to read the
ModelNet40 files that do not exist offline: a stub module satisfies the import, the dataset object is built with
ModelNet40.__new__ and the base clouds are injected as its .data -- __getitem__ itself runs unmodified.

  python tests/golden/gen_data_golden.py      ->  tests/golden/data_recipe.npz
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.modules["h5py"] = types.ModuleType("h5py")              # only used by load_data(), which is never called
sys.path.insert(0, REF)

import util.data as ref_data                                # noqa: E402
import vcrnet_amd                                           # noqa: E402,F401
from vcrnet_amd import synth                                # noqa: E402


def dataset(clouds, num_points, partial):
    ds = ref_data.ModelNet40.__new__(ref_data.ModelNet40)    # skip __init__: it would wget + read the .h5 files
    ds.dataset, ds.num_points, ds.partition = "modelnet40", num_points, "test"
    ds.reserve, ds.gaussian_noise, ds.model = synth.RESERVE_0575, False, "vcrnet"
    ds.factor, ds.partial, ds.unseen = 4.0, partial, False
    ds.data, ds.label = clouds, np.zeros(len(clouds), np.int64)
    return ds


CASES = {  # name: (kind, num_points, partial, items)
    "whole_n1024": ("object", 1024, False, 6),
    "partial_n1024": ("object", 1024, True, 6),
    "whole_n256": ("object", 256, False, 4),
    "uniform_n4096": ("uniform", 4096, False, 2),
    "uniform_partial_n2048": ("uniform", 2048, True, 2),
}

if __name__ == "__main__":
    out = {}
    for name, (kind, n, partial, items) in CASES.items():
        clouds = np.stack([synth.base_cloud(i) if kind == "object" else synth.uniform_cloud(i, max(n, 2048))
                           for i in range(items)])
        ds = dataset(clouds, n, partial)
        cols = list(zip(*[ds[i] for i in range(items)]))
        for key, col in zip(("src", "tgt", "R_ab", "t_ab", "R_ba", "t_ba", "euler_ab", "euler_ba"), cols):
            out[f"{name}.{key}"] = np.stack(col)
        out[f"{name}.meta"] = np.asarray([n, int(partial), items, int(kind == "uniform")], np.int32)
        print(name, out[f"{name}.src"].shape, out[f"{name}.src"].dtype)
    np.savez_compressed(os.path.join(HERE, "data_recipe.npz"), **out)
