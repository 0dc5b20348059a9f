"""CPU-side checks of the drop-in boundary: libvcr_hip.so loads and exports every entry point
include/vcr_hip.h declares; argument validation returns error codes without touching a GPU;
the ctypes structs match the C layout the header implies."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vcr_hip.h")


@pytest.fixture(scope="module")
def lib():
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import build, native
    build.build()
    return native.lib()


def declared_symbols():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"\b(vcr_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/vcr_hip.h but not exported"


def test_version_and_strerror(lib):
    from vcrnet_amd import native
    assert lib.vcr_abi_version() == native.ABI_VERSION == 27
    assert lib.vcr_strerror(0) == b"ok"
    assert b"invalid" in lib.vcr_strerror(-1)
    assert b"workspace" in lib.vcr_strerror(-2)


def test_ctypes_structs_match_the_c_layout(tmp_path):
    """sizeof + the offset of every field of each args struct, as gcc lays out include/vcr_hip.h, against the ctypes
    mirrors in vcrnet_amd/native.py (a field added on one side only would silently shift everything after it)."""
    import subprocess
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import native
    pairs = {"vcr_pointwise_args": native.PointwiseArgs, "vcr_knn_args": native.KnnArgs,
             "vcr_knn_order_args": native.KnnOrderArgs,
             "vcr_linear_args": native.LinearArgs, "vcr_layernorm_args": native.LayerNormArgs,
             "vcr_rowside_args": native.RowsideArgs, "vcr_edgeconv_args": native.EdgeconvArgs,
             "vcr_gathermax_args": native.GathermaxArgs, "vcr_edgerows_args": native.EdgerowsArgs,
             "vcr_edgechain_args": native.EdgechainArgs,
             "vcr_segmax_args": native.SegmaxArgs, "vcr_sdpa_args": native.SdpaArgs,
             "vcr_keymass_args": native.KeymassArgs, "vcr_softcorr_args": native.SoftcorrArgs,
             "vcr_pairscore_args": native.PairscoreArgs, "vcr_scoremass_args": native.ScoremassArgs,
             "vcr_rankselect_args": native.RankselectArgs, "vcr_gather_args": native.GatherArgs,
             "vcr_rigid_svd_args": native.RigidSvdArgs, "vcr_icp_args": native.IcpArgs,
             "vcr_make_pairs_args": native.MakePairsArgs, "vcr_pose_step_args": native.PoseStepArgs, "vcr_vcrnet_weights": native.VcrnetWeights,
             "vcr_vcrnet_io": native.VcrnetIo, "vcr_trace": native.Trace}
    hdr = open(HEADER).read()
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void) {']
    expect = []
    for cname, ct in pairs.items():
        body = re.search(r"typedef struct[^{]*\{((?:[^{}]|\{[^{}]*\})*)\}\s*%s;" % cname, hdr)
        assert body, cname
        lines.append(f'printf("%zu\\n", sizeof({cname}));')
        expect.append((cname, "sizeof", ctypes.sizeof(ct)))
        for fname, _ in ct._fields_:
            cfield = "in" if fname == "in_" else fname
            lines.append(f'printf("%zu\\n", offsetof({cname}, {cfield}));')
            expect.append((cname, fname, getattr(ct, fname).offset))
    lines.append("return 0; }")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-o", str(exe), str(src)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert len(got) == len(expect)
    bad = [(c, f, e, g) for (c, f, e), g in zip(expect, got) if e != g]
    assert not bad, bad


def test_argument_errors_do_not_need_a_gpu(lib):
    from vcrnet_amd import native
    a = native.LinearArgs()          # all NULL
    assert lib.vcr_linear_f32(ctypes.byref(a), None) == -1
    k = native.KnnArgs()
    assert lib.vcr_knn_f32(ctypes.byref(k), None) == -1
    assert lib.vcr_linear_f32(None, None) == -1
    w = native.VcrnetWeights()
    w.E, w.F, w.heads, w.k = 512, 1024, 4, 20
    n1 = lib.vcr_vcrnet_workspace_bytes(ctypes.byref(w), 1, 1024)
    n16 = lib.vcr_vcrnet_workspace_bytes(ctypes.byref(w), 16, 1024)
    assert 0 < n1 < n16 < (4 << 30)
    assert lib.vcr_vcrnet_workspace_bytes(ctypes.byref(w), 0, 1024) == 0
    # vcrnetIter's bookkeeping step: NULL poses, a cloud without its output, a composition without its destination
    assert lib.vcr_pose_step_f32(None, None) == -1
    ps = native.PoseStepArgs()
    assert lib.vcr_pose_step_f32(ctypes.byref(ps), None) == -1
    ps.R_i, ps.t_i, ps.B, ps.N = 0x1000, 0x2000, 2, 64
    assert lib.vcr_pose_step_f32(ctypes.byref(ps), None) == 0          # nothing asked for: no launch
    ps.in_cf = 0x3000
    assert lib.vcr_pose_step_f32(ctypes.byref(ps), None) == -1
    ps.in_cf, ps.compose = None, 1
    assert lib.vcr_pose_step_f32(ctypes.byref(ps), None) == -1
    ps.compose = 3
    assert lib.vcr_pose_step_f32(ctypes.byref(ps), None) == -1
    # the ordered search's ranking entry point: NULL / missing outputs, then a cloud beyond its 8192 points
    assert lib.vcr_knn_order_f32(None, None) == -1
    o = native.KnnOrderArgs()
    assert lib.vcr_knn_order_f32(ctypes.byref(o), None) == -1
    for f in ("xyz4", "perm", "xyz4_p", "cen4", "cen4_rad", "cen4_sqmax"):
        setattr(o, f, 0x1000)                                # (never dereferenced: the size check comes first)
    o.B, o.N = 2, 9000
    assert lib.vcr_knn_order_f32(ctypes.byref(o), None) == -3 and b"unsupported" in lib.vcr_strerror(-3)
    o.feat_t = 0x2000                                        # features without their norms / outputs
    assert lib.vcr_knn_order_f32(ctypes.byref(o), None) == -1


def test_sized_structs_refuse_what_they_cannot_read(lib):
    """ABI 27: vcr_knn_args / vcr_vcrnet_weights state their size.  Zero (a caller that never heard of the field), less than
    the mandatory part, or more than the library knows is an argument error; a SHORTER struct from an older header is served,
    its missing tail read as zeros -- the library never reads past what the caller said it passed."""
    from vcrnet_amd import native
    lib.vcr_vcrnet_workspace_bytes.restype = ctypes.c_size_t
    lib.vcr_knn_ties_inline.argtypes = [ctypes.POINTER(native.KnnArgs)]
    w = native.VcrnetWeights()
    assert w.struct_bytes == ctypes.sizeof(native.VcrnetWeights)
    w.E, w.F, w.heads, w.k, w.has_pointer = 512, 1024, 4, 20, 1
    full = lib.vcr_vcrnet_workspace_bytes(ctypes.byref(w), 4, 1024)
    assert full > 0 and lib.vcr_vcrnet_pairs(ctypes.byref(w), 1024) == 1024
    mandatory = native.VcrnetWeights.fold_encdec_qkv.offset
    w.partial = 1                                            # lives BEHIND the mandatory part ...
    w.overlap2 = 0.75
    assert lib.vcr_vcrnet_pairs(ctypes.byref(w), 1024) < 1024
    w.struct_bytes = mandatory                               # ... so a struct that ends before it is a whole-mode request
    assert lib.vcr_vcrnet_pairs(ctypes.byref(w), 1024) == 1024
    assert lib.vcr_vcrnet_workspace_bytes(ctypes.byref(w), 4, 1024) == full
    for bad in (0, mandatory - 8, ctypes.sizeof(native.VcrnetWeights) + 8):
        w.struct_bytes = bad
        assert lib.vcr_vcrnet_workspace_bytes(ctypes.byref(w), 4, 1024) == 0
        assert lib.vcr_vcrnet_pairs(ctypes.byref(w), 1024) == 0
        assert lib.vcr_vcrnet_forward_f32(ctypes.byref(w), None, None, 0, None) == -1
        assert lib.vcr_vcrnet_iter_f32(ctypes.byref(w), None, 1, None, 0, None, None) == -1
    a = native.KnnArgs(0x1000, 4, None, 64, 1024, 4, 20, 0x2000, 0x3000, 64 * 1024)
    assert a.struct_bytes == ctypes.sizeof(native.KnnArgs) and a.N == 1024
    assert lib.vcr_knn_ties_inline(ctypes.byref(a)) == 1
    a.struct_bytes = native.KnnArgs.waves.offset             # the mandatory part alone
    assert lib.vcr_knn_ties_inline(ctypes.byref(a)) == 1
    for bad in (0, native.KnnArgs.waves.offset - 4, ctypes.sizeof(native.KnnArgs) + 8):
        a.struct_bytes = bad
        assert lib.vcr_knn_ties_inline(ctypes.byref(a)) == 0
        assert lib.vcr_knn_f32(ctypes.byref(a), None) == -1
        assert lib.vcr_knn_pair_f32(ctypes.byref(a), ctypes.byref(a), None) == -1
        assert lib.vcr_knn_ties_f32(ctypes.byref(a), None, None) == -1


def test_iter_workspace_adds_the_target_cache_where_it_applies(lib):
    """vcr_vcrnet_iter_workspace_bytes: the forward's workspace + 2 B N x 2560 floats (the four buffers whose target rows persist) for a vcrnetIter loop of more
    than one pass (every embedding and pointer); nothing for one pass or with iter_reuse = 1."""
    from vcrnet_amd import native
    w = native.VcrnetWeights()
    w.E, w.F, w.heads, w.k, w.has_pointer = 512, 1024, 4, 20, 1
    base = lib.vcr_vcrnet_workspace_bytes(ctypes.byref(w), 24, 768)
    it = lambda n: lib.vcr_vcrnet_iter_workspace_bytes(ctypes.byref(w), 24, 768, n)
    assert it(1) == base and it(2) == it(3) == base + 2 * 24 * 768 * 2560 * 4
    w.iter_reuse = 1
    assert it(3) == base
    w.iter_reuse, w.emb_kind = 0, 1
    assert it(3) == lib.vcr_vcrnet_workspace_bytes(ctypes.byref(w), 24, 768) + 2 * 24 * 768 * 2560 * 4
    w.emb_kind, w.has_pointer = 0, 2
    assert it(3) == lib.vcr_vcrnet_workspace_bytes(ctypes.byref(w), 24, 768) + 2 * 24 * 768 * 2560 * 4
    assert lib.vcr_vcrnet_iter_workspace_bytes(ctypes.byref(w), 24, 768, 0) == 0


def test_workspace_plan_sizes_of_the_baseline_configs(lib):
    """The forward's workspace is laid out by buffer liveness (forward.hip: Plan): pure host arithmetic, so the
    sizes of the BASELINE configs are pinned here -- a buffer registered with too long a life, or the bump allocator coming
    back, shows up as a size regression without a GPU.  (Round 4's bump layout: 1.5 GB at configs[1], ~12 GB at configs[4].)"""
    from vcrnet_amd import native
    lib.vcr_vcrnet_workspace_bytes.restype = ctypes.c_size_t

    def gib(B, N, k=20, merged=True, **kw):
        w = native.VcrnetWeights()
        w.E, w.F, w.heads, w.k, w.has_pointer = 512, 1024, 4, k, 1
        if merged:                                           # (the merged enc + dec first sublayers: never dereferenced here)
            w.fold_encdec_qkv.w = w.fold_encdec_qkv.colsum = w.fold_encdec_qkv.bias = 0x1000
        for f, v in kw.items():
            setattr(w, f, v)
        return lib.vcr_vcrnet_workspace_bytes(ctypes.byref(w), B, N) / 2.0 ** 30

    c1, c1u = gib(16, 1024), gib(16, 1024, merged=False)
    assert 0.5 < c1 < 0.6 and 0.5 < c1u < 0.6, (c1, c1u)     # GiB: nine [M, 512] buffers live at the peak (+ the small ones)
    assert 0.9 < gib(24, 768, partial=1, overlap2=0.75) < 1.25
    assert 1.0 < gib(16, 2048) < 1.35                        # twice configs[1]'s rows
    assert 4.0 < gib(32, 4096, k=40) < 5.3
    sizes = [gib(B, 1024) for B in (1, 2, 4, 8, 16)]
    assert all(a < b for a, b in zip(sizes, sizes[1:]))
    assert sizes[-1] / sizes[0] < 16.5                       # linear in the batch, no per-call constant of note


def test_host_side_dispatch_logic_without_a_gpu(lib):
    """Pure host logic of the round-3 entry points: which kNN calls replay their ties inside the launch, how much
    scratch the replay of long rows needs, the limits that are refused rather than degraded, argument errors of the
    paired linear and the grouped / indexed attention."""
    from vcrnet_amd import native
    lib.vcr_knn_ties_inline.argtypes = [ctypes.POINTER(native.KnnArgs)]
    lib.vcr_knn_ties_inline.restype = ctypes.c_int
    lib.vcr_knn_tie_work_bytes.argtypes, lib.vcr_knn_tie_work_bytes.restype = [ctypes.c_int], ctypes.c_size_t

    def knn_args(B, N, Cc, k, ties=True, waves=0):
        a = native.KnnArgs()
        a.x, a.idx, a.sq = 0x1000, 0x2000, 0x3000            # (never dereferenced on the host)
        a.ldx, a.B, a.N, a.C, a.k, a.waves = Cc, B, N, Cc, k, waves
        if ties:
            a.tie_scratch, a.tie_cap = 0x4000, B * N
        return a
    inline = lambda *a, **k: lib.vcr_knn_ties_inline(ctypes.byref(knn_args(*a, **k)))
    assert inline(32, 1024, 64, 20) == 1 and inline(32, 1024, 4, 20) == 1      # BASELINE configs[1]: both searches
    assert inline(48, 768, 64, 20) == 1 and inline(32, 2048, 64, 20) == 1
    assert inline(64, 4096, 64, 40) == 0 and inline(64, 4096, 4, 40) == 0      # the row image does not fit the workgroup's LDS
    assert inline(2, 1024, 64, 20) == 0                                        # small grid: 32-query kernel, separate replay
    assert inline(32, 1024, 64, 20, waves=1) == 0 and inline(2, 1024, 64, 20, waves=8) == 1
    assert inline(32, 1024, 64, 20, ties=False) == 0
    assert lib.vcr_knn_tie_work_bytes(1024) == 0 and lib.vcr_knn_tie_work_bytes(10091) == 0
    assert lib.vcr_knn_tie_work_bytes(12000) == 64 * 16 * 12000
    a = knn_args(1, 12000, 4, 20)                          # long rows, replay owed, no scratch: refused, not skipped
    assert lib.vcr_knn_f32(ctypes.byref(a), None) == -3
    assert lib.vcr_knn_f32(ctypes.byref(knn_args(4, 1024, 4, 63)), None) == -3 # library limit k <= 62
    assert lib.vcr_knn_f32(ctypes.byref(knn_args(4, 1024, 4, 20, waves=3)), None) == -1
    lib.vcr_linear_pair_f32.argtypes = [ctypes.POINTER(native.LinearArgs), ctypes.POINTER(native.LinearArgs), ctypes.c_void_p]
    lib.vcr_linear_pair_f32.restype = ctypes.c_int
    la = native.LinearArgs()
    assert lib.vcr_linear_pair_f32(ctypes.byref(la), ctypes.byref(la), None) == -1
    la.x, la.w, la.y, la.ldx, la.ldy, la.M, la.N, la.K = 0x1000, 0x2000, 0x3000, 64, 64, 128, 64, 64
    la.variant = 32                                        # a retired selector
    assert lib.vcr_linear_f32(ctypes.byref(la), None) == -1
    sa = native.SdpaArgs()
    sa.q, sa.k, sa.v, sa.out = 0x1000, 0x2000, 0x3000, 0x4000
    sa.ldq = sa.ldk = sa.ldv = sa.ldo = 512
    sa.nbatch, sa.heads, sa.nq, sa.nk, sa.scale = 2, 4, 64, 64, 0.1
    sa.ngroups, sa.rowstat = 2, 0x5000                     # grouped launches: attention-output form only
    assert lib.vcr_sdpa_f32(ctypes.byref(sa), None) == -1
    sa.ngroups, sa.rowstat, sa.key_index, sa.nk_src = 0, None, 0x6000, 0
    assert lib.vcr_sdpa_f32(ctypes.byref(sa), None) == -1  # indexed keys need nk_src


def test_linear_launch_choice_against_the_recorded_sweep(lib):
    """vcr_linear_config (host-only): the kernel configuration per launch.  The BASELINE shapes take what DESIGN says
    (128-row tiles at configs[1]; 96-row tiles for the residual launches of configs[2]; 32-row tiles and the 16x16x4 shape
    for one pair per call), and replayed against the sweep recorded on the GPU (profiles/rounds1-3/r3z_sweep_bm_after.txt: both
    forced heights timed at 56 shapes) the automatic height never loses more than 6 % to the better one."""
    import os
    import re
    from vcrnet_amd import native
    lib.vcr_linear_config.argtypes, lib.vcr_linear_config.restype = [ctypes.POINTER(native.LinearArgs)], ctypes.c_int

    def cfg(M, N, K, residual, variant=0):
        a = native.LinearArgs()
        a.x, a.w, a.y, a.bias = 0x1000, 0x2000, 0x3000, 0x4000            # (never dereferenced on the host)
        a.ldx, a.ldy, a.M, a.N, a.K, a.variant = K, N, M, N, K, variant
        if residual:
            a.residual, a.ldr, a.stats_out = 0x5000, N, 0x6000
        c = lib.vcr_linear_config(ctypes.byref(a))
        assert c > 0, c
        return c & 0xFF, (c >> 8) & 0xFF, bool(c & (1 << 16))
    assert cfg(32768, 512, 512, True) == (128, 32, True) and cfg(32768, 1536, 512, False) == (128, 16, False)     # configs[1]
    assert cfg(36864, 512, 512, True) == (96, 32, True) and cfg(36864, 512, 1024, True) == (96, 32, True)          # configs[2]
    assert cfg(36864, 1024, 512, False) == (128, 16, False)                                                          # (BK 16: no)
    assert cfg(2048, 512, 512, True) == (32, 32, True) and cfg(2048, 512, 512, False) == (32, 32, True)            # one pair
    assert cfg(2048, 3072, 512, False)[0] in (64, 128) and cfg(2048, 3072, 512, False)[2]
    assert cfg(8192, 512, 512, True) == (128, 32, True)               # 256 tiles: exactly one per CU (the first model took 96: -34 %)
    assert cfg(32768, 512, 512, True, variant=2048)[0] == 96 and cfg(2048, 512, 512, True, variant=4096 | 16)[0] == 128
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "rounds1-3", "r3z_sweep_bm_after.txt")
    worst, n = 0.0, 0
    for line in open(path):
        m = re.match(r"K=(\d+) M=\s*(\d+) tiles128=\s*\d+: auto\s+[\d.]+\s+bm128\s+([\d.]+)\s+bm96\s+([\d.]+)", line)
        if not m:
            continue
        K, M, t128, t96 = int(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4))
        if M < 16384:
            continue                                              # (small problems may take 64 / 32 rows: not in this sweep)
        rows = cfg(M, 512, K, True)[0]
        assert rows in (96, 128)
        worst, n = max(worst, (t96 if rows == 96 else t128) / min(t96, t128) - 1), n + 1
    assert n >= 30 and worst <= 0.06, (n, worst)


def test_module_contract_on_cpu():
    """Constructor / state-dict contract of the reference module without a GPU."""
    from types import SimpleNamespace
    import torch
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import weights
    from vcrnet_amd.module import VCRNet
    args = SimpleNamespace(emb_dims=512, cycle=False, emb_nn="lpdnet", pointer="transformer", vcp_nn="topK",
                           partial=False, overlap2=0.75, t3d=False, tfea=False, n_blocks=1, dropout=0.0,
                           ff_dims=1024, n_heads=4)
    net = VCRNet(args)
    keys = set(net.state_dict().keys())
    assert keys == set(weights.param_shapes().keys())
    assert len(keys) == 59 and sum(p.numel() for p in net.parameters()) == 5625161
    w = weights.generate_weights(1234, lpd=weights.load_lpd_fixture())
    net.load_state_dict({"module." + k: v for k, v in w.items()}, strict=True)   # DataParallel-saved checkpoint
    assert torch.equal(net.emb_nn.conv3_lpd.weight, w["emb_nn.conv3_lpd.weight"])
    # attributes util/initPara.py:38-65 touches
    assert hasattr(net.emb_nn, "convDG1") and net.emb_nn.negative_slope == 0.0
    assert any(isinstance(m, torch.nn.Conv2d) for m in net.emb_nn.modules())
    assert not hasattr(net.head, "linears_emb")
    assert net._get_name() == "VCRNet"
    with pytest.raises(Exception):
        VCRNet(SimpleNamespace(**{**vars(args), "emb_nn": "nope"}))
    with pytest.raises(Exception):
        VCRNet(SimpleNamespace(**{**vars(args), "vcp_nn": "nope"}))
    net.eval()
    with torch.no_grad(), pytest.raises(RuntimeError):   # no CPU fallback by design
        net(torch.zeros(1, 3, 64), torch.zeros(1, 3, 64))


def test_cpp_host_example_builds_and_refuses_a_foreign_file(lib, tmp_path):
    """examples/host_cpp/forward_host.cpp (the C-ABI from C++ without Python) compiles against the header and links the library;
    without a GPU it can still be asked to read a file that is not a model blob."""
    import subprocess
    from vcrnet_amd import build
    exe = build.build_host_example()
    bad = tmp_path / "not_a_blob.bin"
    bad.write_bytes(b"\0" * 256)
    r = subprocess.run([exe, str(bad), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 2 and "not a VCRB blob" in r.stderr
    assert subprocess.run([exe], capture_output=True, text=True).returncode == 2      # usage
