"""the evaluation-pair arithmetic of util/data.py:247-329 on the device
(vcr_make_pairs_f32) against the host generator that restates the reference recipe."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,partial,kind", [(1024, False, "object"), (1024, True, "object"), (256, True, "object"),
                                            (4096, False, "uniform"), (2048, True, "uniform")])
def test_device_pairs_equal_host_pairs(N, partial, kind):
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import synth
    B = 5
    src, tgt, R, t, eul = synth.make_batch(40, B, N, partial=partial, kind=kind)
    dsrc, dtgt, dR, dt_, deul = synth.make_batch_device(40, B, N, partial=partial, kind=kind)
    assert dsrc.shape == src.shape and dtgt.shape == tgt.shape
    # float32 gathers, float64 fma-chain transform (what the host dgemm does), exact stable ranking: bit-equal
    np.testing.assert_array_equal(dsrc.cpu().numpy(), src)
    np.testing.assert_array_equal(dtgt.cpu().numpy(), tgt)
    np.testing.assert_array_equal(dR, R)
    np.testing.assert_array_equal(dt_, t)


def test_crop_keeps_nearest_in_distance_order():
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import synth
    src, tgt, *_ = synth.make_batch_device(3, 2, 512, partial=True)
    full_s, full_t, *_ = synth.make_batch_device(3, 2, 512, partial=False)
    K = int(512 * synth.RESERVE_0575)
    assert src.shape == (2, 3, K)
    for part, full in ((src, full_s), (tgt, full_t)):
        d = ((part - full[:, :, -1:]) ** 2).sum(1)
        assert (d[:, 0] == 0).all()                                  # the query point itself comes first
        assert (d[:, 1:] >= d[:, :-1] - 1e-7).all()                  # ascending distance
        dall = ((full - full[:, :, -1:]) ** 2).sum(1).sort(dim=1).values
        assert torch.allclose(d, dall[:, :K], atol=1e-6)


def test_bad_arguments_are_rejected():
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import native
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device="cuda")
    with pytest.raises(native.VcrHipError):                           # keep > N
        native.make_pairs(z(1, 64, 3), z(1, 3, 3, dt=torch.float64), z(1, 3, dt=torch.float64),
                          z(1, 32, dt=torch.int32), z(1, 32, dt=torch.int32), z(1, 32, dt=torch.int32), 33)
    with pytest.raises(native.VcrHipError):                           # float32 pose
        native.make_pairs(z(1, 64, 3), z(1, 3, 3), z(1, 3), z(1, 32, dt=torch.int32), z(1, 32, dt=torch.int32),
                          z(1, 32, dt=torch.int32), 32)
