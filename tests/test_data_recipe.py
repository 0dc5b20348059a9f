"""The synthetic-pair generator against the REFERENCE's own data recipe: tests/golden/data_recipe.npz holds what
ModelNet40.__getitem__ (util/data.py:247-314, crop :320-329) returned in the build container for the base clouds
of vcrnet_amd.synth (recorded by tests/golden/gen_data_golden.py).  Bit-for-bit: inputs of a discretely chaotic
path must not differ in the last place."""
import numpy as np
import pytest

from helpers import golden

CASES = ["whole_n1024", "partial_n1024", "whole_n256", "uniform_n4096", "uniform_partial_n2048"]


@pytest.mark.parametrize("name", CASES)
def test_make_batch_equals_reference_getitem(name):
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import synth
    g = golden("data_recipe")
    n, partial, items, uniform = (int(x) for x in g[name + ".meta"])
    src, tgt, R, t, eul = synth.make_batch(0, items, n, partial=bool(partial), kind="uniform" if uniform else "object")
    assert src.dtype == np.float32 and src.shape == g[name + ".src"].shape
    assert np.array_equal(src, g[name + ".src"])
    assert np.array_equal(tgt, g[name + ".tgt"])
    assert np.array_equal(R, g[name + ".R_ab"]) and np.array_equal(t, g[name + ".t_ab"])
    assert np.array_equal(eul, g[name + ".euler_ab"])
    # the B -> A labels the loader also returns (util/data.py:278,286,295)
    R_ba, t_ba, eul_ba = synth.inverse_labels(R.astype(np.float64), t.astype(np.float64), eul)
    np.testing.assert_allclose(R_ba, g[name + ".R_ba"], atol=1e-7)
    np.testing.assert_allclose(t_ba, g[name + ".t_ba"], atol=1e-6)
    assert np.array_equal(eul_ba.astype(np.float32), g[name + ".euler_ba"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_generator_equals_reference_getitem(name):
    """vcr_make_pairs_f32 (gather, float64 rigid transform, nearest-to-last-point crop on the device) reproduces the
    reference loader's output bit for bit."""
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import synth
    g = golden("data_recipe")
    n, partial, items, uniform = (int(x) for x in g[name + ".meta"])
    src, tgt, R, t, eul = synth.make_batch_device(0, items, n, partial=bool(partial),
                                                  kind="uniform" if uniform else "object", device="cuda")
    assert np.array_equal(src.cpu().numpy(), g[name + ".src"])
    assert np.array_equal(tgt.cpu().numpy(), g[name + ".tgt"])
    assert np.array_equal(R, g[name + ".R_ab"]) and np.array_equal(t, g[name + ".t_ab"])
