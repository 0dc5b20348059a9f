"""bench.py's work model (vcrnet_amd.workmodel): algorithmic flops / bytes per launch name.  Pure arithmetic, no GPU."""
import pytest


def test_source_only_launches_count_half_and_every_name_of_a_trace_is_known():
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import workmodel as wm
    B, N, k = 24, 768, 20
    for name in ("linear:conv3", "linear:encdec.qkv", "linear:enc.wo+dec.self.wo", "sdpa:encdec.self", "edgeconv:dg1_dg2",
                 "knn:feat64+xyz", "gathermax:sn1", "pointwise:src+tgt+dg1_pq", "knn:rank", "linear:sn1_pq", "linear:dg_c3",
                 "edgeconv:dg_chain", "linear:pn_c4", "knn:xyz"):
        full, half = wm.launch_work(name, B, N, k), wm.launch_work(name + "@src", B, N, k)
        assert half[0] == pytest.approx(0.5 * full[0]) and half[1] == pytest.approx(0.5 * full[1]), name
        assert wm.gather_bytes(name + "@src", B, N, k) == pytest.approx(0.5 * wm.gather_bytes(name, B, N, k))
        assert name.split(":")[0] in wm.FAMILY_BOUND
    # SURVEY 8d's per-pair figures are what the model reproduces for the reference formulation
    assert wm.reference_flops_per_pair(1024, 20)["total"] == pytest.approx(44.87e9, rel=2e-3)
