"""ctypes binding of include/vcr_hip.h (libvcr_hip.so).

The product path has NO fallback: if the HIP library is missing or a call fails this module
raises.  ``import torch`` must happen before the library is loaded so that it binds to the HIP
runtime torch already loaded (torch bundles its own libamdhip64 with the same soname).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvcr_hip.so")
VCR_TRACE_MAX = 256

f32p = C.c_void_p  # device pointers travel as integers


class PointwiseArgs(C.Structure):
    _fields_ = [("x_cf", f32p), ("B", C.c_int), ("N", C.c_int), ("w1", f32p), ("b1", f32p), ("w2", f32p),
                ("b2", f32p), ("xyz4", f32p), ("feat64", f32p), ("sq64", f32p), ("x_cf2", f32p), ("B2", C.c_int),
                ("pq_w", f32p), ("pq_b", f32p), ("pq", f32p), ("ldpq", C.c_int), ("feat64t", f32p)]


class _Sized(C.Structure):
    """Args structs with a leading `struct_bytes` (ABI 27): set on construction, positional arguments start at the field
    behind it -- KnnArgs(x, ldx, ...) reads as before."""

    def __init__(self, *args, **kw):
        super().__init__(C.sizeof(type(self)), *args, **kw)


class KnnArgs(_Sized):
    _fields_ = [("struct_bytes", C.c_uint32), ("x", f32p), ("ldx", C.c_int), ("sq", f32p), ("B", C.c_int), ("N", C.c_int), ("C", C.c_int),
                ("k", C.c_int), ("idx", f32p), ("tie_scratch", f32p), ("tie_cap", C.c_int), ("waves", C.c_int),
                ("tie_zeroed", C.c_int),
                ("tie_defer", C.c_int), ("tie_work", C.c_void_p), ("tie_work_bytes", C.c_size_t), ("tie_inline", C.c_int),
                ("xt", f32p),
                ("perm", f32p), ("xp", f32p), ("sqp", f32p), ("cen", f32p), ("cen_sq", f32p), ("cen_rad", f32p),
                ("cen_sqmax", f32p), ("ord_ok", f32p)]


class KnnOrderArgs(C.Structure):
    _fields_ = [("xyz4", f32p), ("feat_t", f32p), ("ldf", C.c_int), ("sq", f32p), ("B", C.c_int), ("N", C.c_int),
                ("perm", f32p), ("xyz4_p", f32p), ("cen4", f32p), ("cen4_rad", f32p), ("cen4_sqmax", f32p),
                ("feat_p", f32p), ("sq_p", f32p), ("cen64", f32p), ("cen64_sq", f32p), ("cen64_rad", f32p),
                ("cen64_sqmax", f32p), ("ord_ok", f32p), ("ord_stat", f32p), ("guard_ratio", C.c_float)]


class LinearArgs(C.Structure):
    _fields_ = [("x", f32p), ("ldx", C.c_int), ("w", f32p), ("bias", f32p), ("residual", f32p), ("ldr", C.c_int),
                ("y", f32p), ("ldy", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("relu", C.c_int),
                ("ln_stats_in", f32p), ("ln_nseg", C.c_int), ("ln_colsum", f32p), ("ln_eps", C.c_float),
                ("stats_out", f32p), ("segmax_out", f32p), ("ld_segmax", C.c_int), ("seg_k", C.c_int),
                ("variant", C.c_int)]


class LayerNormArgs(C.Structure):
    _fields_ = [("x", f32p), ("ldx", C.c_int), ("a", f32p), ("b", f32p), ("eps", C.c_float), ("residual", f32p),
                ("ldr", C.c_int), ("y", f32p), ("ldy", C.c_int), ("M", C.c_int), ("C", C.c_int), ("xyz4", f32p),
                ("side4", f32p)]


class RowsideArgs(C.Structure):
    _fields_ = [("x", f32p), ("ldx", C.c_int), ("M", C.c_int), ("C", C.c_int), ("scale", C.c_float), ("y", f32p),
                ("ldy", C.c_int), ("xyz4", f32p), ("side4", f32p)]


class EdgeconvArgs(C.Structure):
    _fields_ = [("pq", f32p), ("ldpq", C.c_int), ("idx", f32p), ("k", C.c_int), ("M", C.c_int),
                ("n_per_cloud", C.c_int), ("w2", f32p), ("b2", f32p), ("x1", f32p), ("ldx1", C.c_int),
                ("x2", f32p), ("ldx2", C.c_int)]


class GathermaxArgs(C.Structure):
    _fields_ = [("pq", f32p), ("ldpq", C.c_int), ("C", C.c_int), ("idx", f32p), ("k", C.c_int), ("M", C.c_int),
                ("n_per_cloud", C.c_int), ("y", f32p), ("ldy", C.c_int), ("variant", C.c_int), ("order", f32p)]


class SdpaArgs(C.Structure):
    _fields_ = [("q", f32p), ("ldq", C.c_int), ("k", f32p), ("ldk", C.c_int), ("v", f32p), ("ldv", C.c_int),
                ("out", f32p), ("ldo", C.c_int), ("nbatch", C.c_int), ("heads", C.c_int), ("nq", C.c_int),
                ("nk", C.c_int), ("scale", C.c_float), ("kv_batch_shift", C.c_int), ("key_keep", f32p),
                ("rowstat", f32p), ("score_out", f32p), ("ld_score", C.c_int),
                ("ngroups", C.c_int), ("q_group_stride", C.c_long), ("k_group_stride", C.c_long), ("v_group_stride", C.c_long),
                ("out_group_stride", C.c_long), ("key_index", f32p), ("nk_src", C.c_int), ("split_work", f32p),
                ("split_work_floats", C.c_long), ("variant", C.c_int), ("plan_nbatch", C.c_int)]


class KeymassArgs(C.Structure):
    _fields_ = [("score", f32p), ("ld", C.c_int), ("nbatch", C.c_int), ("heads", C.c_int), ("nq", C.c_int),
                ("nk", C.c_int), ("rowstat", f32p), ("q_batch_shift", C.c_int), ("mass", f32p)]


class SoftcorrArgs(C.Structure):
    _fields_ = [("q", f32p), ("ldq", C.c_int), ("k", f32p), ("ldk", C.c_int), ("qside4", f32p), ("kside4", f32p),
                ("corr4", f32p), ("nbatch", C.c_int), ("nq", C.c_int), ("nk", C.c_int), ("E", C.c_int),
                ("mode", C.c_int), ("scale", C.c_float), ("split_work", f32p), ("split_work_floats", C.c_long)]


class PairscoreArgs(C.Structure):
    _fields_ = [("own", f32p), ("ld_own", C.c_int), ("str", f32p), ("ld_str", C.c_int), ("own_side4", f32p),
                ("str_side4", f32p), ("nbatch", C.c_int), ("n_own", C.c_int), ("n_str", C.c_int), ("E", C.c_int),
                ("score", C.c_int), ("scale", C.c_float), ("str_batch_shift", C.c_int), ("op", C.c_int),
                ("corr4", f32p), ("stat2", f32p), ("argmax", f32p), ("str_stat2", f32p),
                ("str_stat_batch_stride", C.c_long), ("mass", f32p), ("accumulate", C.c_int),
                ("score_out", f32p), ("ld_score", C.c_int), ("variant", C.c_int), ("split_work", f32p),
                ("split_work_floats", C.c_long)]


class MakePairsArgs(C.Structure):
    _fields_ = [("cloud", f32p), ("P", C.c_int), ("R_ab", f32p), ("t_ab", f32p), ("pick", f32p), ("perm_src", f32p),
                ("perm_tgt", f32p), ("B", C.c_int), ("N", C.c_int), ("keep", C.c_int), ("src_cf", f32p),
                ("tgt_cf", f32p)]


class ScoremassArgs(C.Structure):
    _fields_ = [("score", f32p), ("ld", C.c_int), ("nbatch", C.c_int), ("n_rows", C.c_int), ("n_cols", C.c_int),
                ("row_stat2", f32p), ("col_stat2", f32p), ("col_mass", f32p), ("row_mass", f32p)]


class RankselectArgs(C.Structure):
    _fields_ = [("values", f32p), ("nbatch", C.c_int), ("n", C.c_int), ("K", C.c_int), ("order", f32p),
                ("mask", f32p), ("largest", C.c_int), ("stride", C.c_int)]


class GatherArgs(C.Structure):
    _fields_ = [("in_", f32p), ("ld_in", C.c_int), ("n_in", C.c_int), ("idx", f32p), ("nbatch", C.c_int),
                ("n_out", C.c_int), ("C", C.c_int), ("out", f32p), ("ld_out", C.c_int), ("via", f32p), ("n_via", C.c_int)]


class EdgerowsArgs(C.Structure):
    _fields_ = [("pq", f32p), ("ldpq", C.c_int), ("C", C.c_int), ("idx", f32p), ("k", C.c_int), ("M", C.c_int),
                ("n_per_cloud", C.c_int), ("h", f32p), ("ldh", C.c_int), ("ymax", f32p), ("ldymax", C.c_int),
                ("zero_to", C.c_int)]


class EdgechainArgs(C.Structure):
    _fields_ = [("pq", f32p), ("ldpq", C.c_int), ("idx", f32p), ("k", C.c_int), ("M", C.c_int), ("n_per_cloud", C.c_int),
                ("w2", f32p), ("b2", f32p), ("w3", f32p), ("b3", f32p), ("w4", f32p), ("b4", f32p),
                ("out", f32p), ("ldo", C.c_int)]


class SegmaxArgs(C.Structure):
    _fields_ = [("x", f32p), ("ldx", C.c_int), ("M", C.c_int), ("k", C.c_int), ("C", C.c_int), ("y", f32p),
                ("ldy", C.c_int)]


class RigidSvdArgs(C.Structure):
    _fields_ = [("src", f32p), ("lds", C.c_int), ("corr", f32p), ("ldc", C.c_int), ("B", C.c_int), ("K", C.c_int),
                ("R", f32p), ("t", f32p), ("R_ba", f32p), ("t_ba", f32p), ("H", f32p)]


class NormW(C.Structure):
    _fields_ = [("ln_a", f32p), ("ln_b", f32p)]


class MhaW(C.Structure):
    _fields_ = [("wqkv", f32p), ("bqkv", f32p), ("wq", f32p), ("bq", f32p), ("wkv", f32p), ("bkv", f32p),
                ("wo", f32p), ("bo", f32p)]


class FfnW(C.Structure):
    _fields_ = [("w1", f32p), ("b1", f32p), ("w2", f32p), ("b2", f32p)]


SPLIT_SITES = ("dg1_pq", "sn1_pq", "c3", "enc_qkv", "enc_wo", "enc_ffn1", "enc_ffn2", "dec_qkv", "dec_self_wo",
               "dec_cross_q", "dec_cross_kv", "dec_cross_wo", "dec_ffn1", "dec_ffn2", "encdec_qkv")


class SplitW(C.Structure):
    _fields_ = [(s, f32p) for s in SPLIT_SITES]


class FoldedW(C.Structure):
    _fields_ = [("w", f32p), ("colsum", f32p), ("bias", f32p)]


class DgcnnW(C.Structure):
    _fields_ = [(n, f32p) for n in ("c1_wpq", "c1_bpq", "c2_w", "c2_b", "c3_w", "c3_b", "c4_w", "c4_b", "c5_w", "c5_b")]


class PointnetW(C.Structure):
    _fields_ = [(n, f32p) for n in ("c3_w", "c3_b", "c4_w", "c4_b", "c5_w", "c5_b")]


class VcrnetWeights(_Sized):
    _fields_ = [("struct_bytes", C.c_uint32), ("c1_w", f32p), ("c1_b", f32p), ("c2_w", f32p), ("c2_b", f32p),
                ("dg1_wpq", f32p), ("dg1_bpq", f32p), ("dg2_w", f32p), ("dg2_b", f32p),
                ("sn1_wpq", f32p), ("sn1_bpq", f32p), ("c3_w", f32p), ("c3_b", f32p),
                ("enc_ln0", NormW), ("enc_ln1", NormW), ("enc_norm", NormW), ("dec_ln0", NormW),
                ("dec_ln1", NormW), ("dec_ln2", NormW), ("dec_norm", NormW),
                ("enc_self", MhaW), ("dec_self", MhaW), ("dec_cross", MhaW),
                ("enc_ffn", FfnW), ("dec_ffn", FfnW),
                ("E", C.c_int), ("F", C.c_int), ("heads", C.c_int), ("k", C.c_int),
                ("has_pointer", C.c_int), ("head_mode", C.c_int), ("linear_mode", C.c_int), ("split", SplitW),
                ("fold_enc_qkv", FoldedW), ("fold_enc_ffn1", FoldedW), ("fold_dec_qkv", FoldedW),
                ("fold_dec_cross_q", FoldedW), ("fold_dec_cross_kv", FoldedW), ("fold_dec_ffn1", FoldedW),
                ("fold_encdec_qkv", FoldedW),
                ("partial", C.c_int), ("overlap2", C.c_double), ("emb_kind", C.c_int), ("dgcnn", DgcnnW), ("pointnet", PointnetW),
                ("att_w0", f32p), ("att_b0", f32p), ("att_w1", f32p), ("att_b1", f32p), ("cycle", C.c_int),
                ("linear_mfma", C.c_int), ("linear_bk", C.c_int), ("linear_bm", C.c_int), ("knn_waves", C.c_int),
                ("xscore_limit_mb", C.c_int), ("sdpa_variant", C.c_int), ("iter_reuse", C.c_int), ("workspace_flat", C.c_int)]


class VcrnetIo(C.Structure):
    _fields_ = [("src_cf", f32p), ("tgt_cf", f32p), ("B", C.c_int), ("N", C.c_int), ("corr4", f32p), ("src4", f32p),
                ("R_ab", f32p), ("t_ab", f32p), ("R_ba", f32p), ("t_ba", f32p), ("emb_out", f32p),
                ("force_keys", f32p), ("force_sel_src", f32p), ("force_sel_tgt", f32p), ("force_argmax", f32p),
                ("force_pairs", f32p), ("out_keys", f32p), ("out_sel_src", f32p), ("out_sel_tgt", f32p),
                ("out_argmax", f32p), ("out_pairs", f32p)]


SELECTION_FIELDS = ("keys", "sel_src", "sel_tgt", "argmax", "pairs")


class Trace(C.Structure):
    _fields_ = [("events", C.POINTER(C.c_void_p)), ("capacity", C.c_int), ("count", C.c_int),
                ("names", C.c_char_p * VCR_TRACE_MAX)]


_SIGS = {
    "vcr_pointwise_f32": PointwiseArgs, "vcr_knn_f32": KnnArgs, "vcr_linear_f32": LinearArgs,
    "vcr_layernorm_f32": LayerNormArgs, "vcr_rowside_f32": RowsideArgs, "vcr_edgeconv_f32": EdgeconvArgs, "vcr_edgeconv_bf16x3_f32": EdgeconvArgs,
    "vcr_gathermax_f32": GathermaxArgs, "vcr_sdpa_f32": SdpaArgs, "vcr_sdpa_bf16x3_f32": SdpaArgs, "vcr_softcorr_f32": SoftcorrArgs,
    "vcr_rigid_svd_f32": RigidSvdArgs, "vcr_pairscore_f32": PairscoreArgs, "vcr_rankselect_f32": RankselectArgs,
    "vcr_gather_rows_f32": GatherArgs, "vcr_scoremass_f32": ScoremassArgs, "vcr_keymass_f32": KeymassArgs, "vcr_make_pairs_f32": MakePairsArgs,
    "vcr_edgerows_f32": EdgerowsArgs, "vcr_segmax_f32": SegmaxArgs, "vcr_edgechain_f32": EdgechainArgs,
}

_lib: Optional[C.CDLL] = None


ABI_VERSION = 27         # include/vcr_hip.h vcr_abi_version(); the ctypes structs below mirror that header


class VcrHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libvcr_hip.so (once).  Raises if it has not been built -- there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VcrHipError(f"{LIB_PATH} not built: run `python vcr-net_amd/build.py` "
                              "(or __graft_entry__.build()); the HIP path has no fallback")
        L = C.CDLL(LIB_PATH)
        for name, st in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = [C.POINTER(st), C.c_void_p]
            fn.restype = C.c_int
        L.vcr_strerror.argtypes = [C.c_int]; L.vcr_strerror.restype = C.c_char_p
        L.vcr_abi_version.restype = C.c_int
        if L.vcr_abi_version() != ABI_VERSION:
            raise VcrHipError(f"{LIB_PATH} exports ABI {L.vcr_abi_version()}, these bindings are for {ABI_VERSION}: "
                              "rebuild with `python vcr-net_amd/build.py`")
        L.vcr_vcrnet_workspace_bytes.argtypes = [C.POINTER(VcrnetWeights), C.c_int, C.c_int]
        L.vcr_vcrnet_workspace_bytes.restype = C.c_size_t
        L.vcr_vcrnet_iter_workspace_bytes.argtypes = [C.POINTER(VcrnetWeights), C.c_int, C.c_int, C.c_int]
        L.vcr_vcrnet_iter_workspace_bytes.restype = C.c_size_t
        L.vcr_vcrnet_forward_f32.argtypes = [C.POINTER(VcrnetWeights), C.POINTER(VcrnetIo), C.c_void_p, C.c_size_t,
                                             C.c_void_p]
        L.vcr_vcrnet_forward_f32.restype = C.c_int
        L.vcr_vcrnet_forward_traced_f32.argtypes = [C.POINTER(VcrnetWeights), C.POINTER(VcrnetIo), C.c_void_p,
                                                    C.c_size_t, C.c_void_p, C.POINTER(Trace)]
        L.vcr_vcrnet_forward_traced_f32.restype = C.c_int
        L.vcr_vcrnet_pairs.argtypes = [C.POINTER(VcrnetWeights), C.c_int]; L.vcr_vcrnet_pairs.restype = C.c_int
        L.vcr_vcrnet_iter_f32.argtypes = [C.POINTER(VcrnetWeights), C.POINTER(VcrnetIo), C.c_int, C.c_void_p,
                                          C.c_size_t, C.c_void_p, C.POINTER(Trace)]
        L.vcr_vcrnet_iter_f32.restype = C.c_int
        L.vcr_event_create.argtypes = [C.POINTER(C.c_void_p)]; L.vcr_event_create.restype = C.c_int
        L.vcr_event_destroy.argtypes = [C.c_void_p]; L.vcr_event_destroy.restype = C.c_int
        L.vcr_event_record.argtypes = [C.c_void_p, C.c_void_p]; L.vcr_event_record.restype = C.c_int
        L.vcr_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        L.vcr_event_elapsed_ms.restype = C.c_int
        _lib = L
    return _lib


class LaunchTrace:
    """A pool of HIP events for vcr_vcrnet_forward_traced_f32: one event before every kernel launch
    and one after the last, recorded on the launch stream (include/vcr_hip.h, vcr_trace)."""

    def __init__(self, capacity: int = VCR_TRACE_MAX):
        L = lib()
        self.capacity = capacity
        self._arr = (C.c_void_p * capacity)()
        for i in range(capacity):
            e = C.c_void_p()
            check(L.vcr_event_create(C.byref(e)), "vcr_event_create")
            self._arr[i] = e.value
        self.trace = Trace(C.cast(self._arr, C.POINTER(C.c_void_p)), capacity, 0)

    def launches(self):
        """[(name, ms)] of the last traced forward -- call after the stream has been synchronised."""
        L = lib()
        n = min(self.trace.count, self.capacity - 1, VCR_TRACE_MAX)
        out = []
        for i in range(n):
            ms = C.c_float()
            check(L.vcr_event_elapsed_ms(self._arr[i], self._arr[i + 1], C.byref(ms)), "vcr_event_elapsed_ms")
            out.append((self.trace.names[i].decode(), ms.value))
        return out

    def close(self):
        L = lib()
        for i in range(self.capacity):
            if self._arr[i]:
                L.vcr_event_destroy(self._arr[i])
                self._arr[i] = None


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise VcrHipError(f"{what}: rc={rc} ({lib().vcr_strerror(rc).decode()})")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a CUDA(HIP) fp32/int32/uint8 tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise VcrHipError("vcr-net_amd kernels take device tensors only (no CPU fallback)")
    return t.data_ptr()


def stream_ptr(device=None) -> int:
    """The current HIP stream of `device` (default: the current device).  Kernels must be enqueued on a stream of the
    device that owns their pointers: every wrapper below runs under on_device(tensor), so `net.to('cuda:1')(x)`
    without torch.cuda.set_device(1) launches on cuda:1, not on the current device."""
    return torch.cuda.current_stream(device).cuda_stream


def on_device(t: torch.Tensor):
    """Context manager: make t's device current (HIP launches go to the CURRENT device's stream)."""
    if not t.is_cuda:
        raise VcrHipError("vcr-net_amd kernels take device tensors only (no CPU fallback)")
    return torch.cuda.device(t.device)


def same_device(*tensors) -> torch.device:
    devs = {t.device for t in tensors if t is not None}
    if len(devs) != 1:
        raise VcrHipError(f"all tensors of one call must live on one device, got {sorted(map(str, devs))}")
    return next(iter(devs))


def call(name: str, args: C.Structure) -> None:
    check(getattr(lib(), name)(C.byref(args), C.c_void_p(stream_ptr())), name)


def _guarded(fn):
    """Run a tensor-level wrapper with its first device tensor's device made current."""
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **kw):
        for x in list(a) + list(kw.values()):
            if isinstance(x, torch.Tensor) and x.is_cuda:
                with torch.cuda.device(x.device):
                    return fn(*a, **kw)
        return fn(*a, **kw)
    return wrapper


# ---- thin tensor-level wrappers (used by the tests and by the module for the non-fused variants) -------------

def _f32(*shape, device):
    return torch.empty(*shape, dtype=torch.float32, device=device)


@_guarded
def pointwise(x_cf, w1, b1, w2, b2, pq_w=None, pq_b=None, feat_t=None):
    """conv1_lpd + conv2_lpd (+ ReLU) -> (xyz4, feat64, sq64); with pq_w [256,64] / pq_b [256] also the P | Q projection
    of the first EdgeConv from the same (MFMA) launch -> (xyz4, feat64, sq64, pq [B*N,256]).  feat_t: a [B,N,64] tensor
    that receives the copy of feat64 with its 16-channel groups transposed (what knn(..., xt=) reads)."""
    B, _, N = x_cf.shape
    x_cf = x_cf.contiguous()
    xyz4, f64, sq = _f32(B, N, 4, device=x_cf.device), _f32(B, N, 64, device=x_cf.device), _f32(B, N, device=x_cf.device)
    a = PointwiseArgs(ptr(x_cf), B, N, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(xyz4), ptr(f64), ptr(sq))
    pq = None
    if pq_w is not None:
        pq = _f32(B * N, 256, device=x_cf.device)
        a.pq_w, a.pq_b, a.pq, a.ldpq = ptr(pq_w), ptr(pq_b), ptr(pq), 256
    a.feat64t = ptr(feat_t)
    call("vcr_pointwise_f32", a)
    return (xyz4, f64, sq) if pq is None else (xyz4, f64, sq, pq)


@_guarded
def knn(x, sq, k, exact_ties=True, waves=0, tie_work=True, xt=None, tie_slots=False):
    """x [B,N,C] rows (C = 64 with sq [B,N], or C = 4 xyz4 rows) -> int32 idx [B,N,k].  exact_ties: rows whose
    (k+1)-th and (k+2)-th distances are equal get Tensor.topk's (libstdc++'s) pick instead of the lower index.
    tie_slots: give the launch vcr_knn_tie_slot_bytes(B, N) of scratch -- it then replays its tied rows itself."""
    B, N, Cc = x.shape
    idx = torch.empty(B, N, k, dtype=torch.int32, device=x.device)
    ties = torch.empty(1 + B * N, dtype=torch.int32, device=x.device) if exact_ties else None   # room for every row
    a = KnnArgs(ptr(x), x.stride(1), ptr(sq), B, N, Cc, k, ptr(idx), ptr(ties), B * N if exact_ties else 0, waves)
    L = lib()
    L.vcr_knn_tie_work_bytes.restype, L.vcr_knn_tie_work_bytes.argtypes = C.c_size_t, [C.c_int]
    need = L.vcr_knn_tie_work_bytes(N) if (exact_ties and tie_work) else 0
    if exact_ties and tie_slots:
        L.vcr_knn_tie_slot_bytes.restype, L.vcr_knn_tie_slot_bytes.argtypes = C.c_size_t, [C.c_int, C.c_int]
        need = L.vcr_knn_tie_slot_bytes(B, N)
    work = torch.empty(need, dtype=torch.uint8, device=x.device) if need else None            # long rows: replay scratch
    a.tie_work, a.tie_work_bytes = ptr(work), need
    a.xt = ptr(xt)               # the rows with their 16-channel groups transposed (pointwise(..., feat_t=)): same result
    call("vcr_knn_f32", a)
    return idx


def _tie_work(a, N, device, keep, slots_for=0):
    """Long rows (vcr_knn_tie_work_bytes(N) > 0, N > ~10 100): the replay's global scratch, as knn() provides it.
    slots_for = B: vcr_knn_tie_slot_bytes(B, N) instead -- the launch replays its own ties (vcr_knn_args.tie_inline 2)."""
    L = lib()
    L.vcr_knn_tie_work_bytes.restype, L.vcr_knn_tie_work_bytes.argtypes = C.c_size_t, [C.c_int]
    L.vcr_knn_tie_slot_bytes.restype, L.vcr_knn_tie_slot_bytes.argtypes = C.c_size_t, [C.c_int, C.c_int]
    need = L.vcr_knn_tie_slot_bytes(slots_for, N) if slots_for else L.vcr_knn_tie_work_bytes(N)
    if need:
        work = torch.empty(need, dtype=torch.uint8, device=device)
        a.tie_work, a.tie_work_bytes = ptr(work), need
        keep.append(work)


@_guarded
def knn_pair(feat, sq, xyz4, k, xt=None, order=None, tie_slots=False):
    """vcr_knn_pair_f32: the feature-space (feat [B,N,64], sq [B,N]) and the Cartesian (xyz4 [B,N,4]) kNN in one launch
    -> (idx_feat, idx_xyz), tie replay included.  order = knn_order()'s dict: the ordered search (vcr_knn_args.perm).
    tie_slots: per-workgroup replay slots (vcr_knn_tie_slot_bytes) -- tied rows are replayed inside the launch."""
    L = lib()
    out, args, keep = [], [], []
    for x, s_ in ((feat, sq), (xyz4, None)):
        B, N, Cc = x.shape
        idx = torch.empty(B, N, k, dtype=torch.int32, device=x.device)
        ties = torch.empty(1 + B * N, dtype=torch.int32, device=x.device)
        args.append(KnnArgs(ptr(x), x.stride(1), ptr(s_), B, N, Cc, k, ptr(idx), ptr(ties), B * N, 0))
        _tie_work(args[-1], N, x.device, keep, slots_for=B if tie_slots else 0)
        out.append(idx); keep.append(ties)
    args[0].xt = ptr(xt)
    if order is not None:
        o = lambda n: ptr(order[n])
        args[0].perm = args[1].perm = o("perm")
        a = args[0]
        a.xp, a.sqp, a.cen, a.cen_sq, a.cen_rad, a.cen_sqmax = o("feat_p"), o("sq_p"), o("cen64"), o("cen64_sq"), o("cen64_rad"), o("cen64_sqmax")
        a.ord_ok = ptr(order.get("ord_ok"))
        a = args[1]
        a.xp, a.cen, a.cen_rad, a.cen_sqmax = o("xyz4_p"), o("cen4"), o("cen4_rad"), o("cen4_sqmax")
    L.vcr_knn_pair_f32.argtypes = [C.POINTER(KnnArgs), C.POINTER(KnnArgs), C.c_void_p]
    L.vcr_knn_pair_f32.restype = C.c_int
    check(L.vcr_knn_pair_f32(C.byref(args[0]), C.byref(args[1]), C.c_void_p(stream_ptr())), "vcr_knn_pair_f32")
    return out[0], out[1]


@_guarded
def knn_pair_deferred(xa, sqa, xb, sqb, k):
    """Two kNN launches with tie_defer and ONE vcr_knn_ties_f32 replay for both (the LPDNet pattern) -> (idx_a, idx_b)."""
    L = lib()
    args, keep = [], []
    for x, sq in ((xa, sqa), (xb, sqb)):
        B, N, Cc = x.shape
        idx = torch.empty(B, N, k, dtype=torch.int32, device=x.device)
        ties = torch.zeros(1 + B * N, dtype=torch.int32, device=x.device)
        a = KnnArgs(ptr(x), x.stride(1), ptr(sq), B, N, Cc, k, ptr(idx), ptr(ties), B * N, 0)
        a.tie_zeroed, a.tie_defer = 1, 1
        work = []
        _tie_work(a, N, x.device, work)
        call("vcr_knn_f32", a)
        args.append(a); keep.append((idx, ties, work))
    L.vcr_knn_ties_f32.argtypes = [C.POINTER(KnnArgs), C.POINTER(KnnArgs), C.c_void_p]
    L.vcr_knn_ties_f32.restype = C.c_int
    check(L.vcr_knn_ties_f32(C.byref(args[0]), C.byref(args[1]), C.c_void_p(stream_ptr())), "vcr_knn_ties_f32")
    return keep[0][0], keep[1][0]


@_guarded
def linear(x, w, bias=None, relu=False, residual=None, out=None, ln=None, want_stats=False, variant=0, segmax=None,
           store=True):
    """y = act(x w^T + bias) (+ residual).  ln = (stats [M,nseg,2], colsum [N], eps) with w / bias folded by
    fold_layernorm(): y = act(LayerNorm(x) w0^T + bias0).  want_stats: also return the [M, N/64, 2]
    (sum, sum of squares) partials of y."""
    M, K = x.shape
    N = w.shape[0]
    y = None if not store else out if out is not None else _f32(M, N, device=x.device)
    stats = _f32(M, N // 64, 2, device=x.device) if want_stats else None
    a = LinearArgs(ptr(x), x.stride(0), ptr(w), ptr(bias), ptr(residual),
                   residual.stride(0) if residual is not None else 0, ptr(y), y.stride(0) if y is not None else N,
                   M, N, K, int(relu))
    if segmax is not None:       # (out [M / seg_k, >= N] row view pre-set to 0, seg_k): fused max over each point's edge rows
        a.segmax_out, a.ld_segmax, a.seg_k = ptr(segmax[0]), segmax[0].stride(0), int(segmax[1])
    if ln is not None:
        a.ln_stats_in, a.ln_nseg, a.ln_colsum, a.ln_eps = ptr(ln[0]), ln[0].shape[1], ptr(ln[1]), ln[2]
    a.stats_out = ptr(stats)
    a.variant = variant
    call("vcr_linear_f32", a)
    return (y, stats) if want_stats else y


@_guarded
def fold_layernorm(w, bias, ln_a, ln_b):
    """vcr_fold_layernorm_f32: (w * a, colsum, bias + w b) for a Linear that consumes LayerNorm(a, b)."""
    L = lib()
    N, K = w.shape
    w = w.contiguous()
    wf, cs, bf = _f32(N, K, device=w.device), _f32(N, device=w.device), _f32(N, device=w.device)
    L.vcr_fold_layernorm_f32.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int] + [C.c_void_p] * 4
    L.vcr_fold_layernorm_f32.restype = C.c_int
    check(L.vcr_fold_layernorm_f32(ptr(w), ptr(bias), ptr(ln_a), ptr(ln_b), N, K, ptr(wf), ptr(cs), ptr(bf),
                                   C.c_void_p(stream_ptr())), "vcr_fold_layernorm_f32")
    return wf, cs, bf


@_guarded
def split_bf16x3(w):
    """fp32 tensor -> int16 [3, numel] bf16 planes (hi, mid, lo) with hi + mid + lo == w exactly."""
    L = lib()
    w = w.contiguous().float()
    planes = torch.empty(3, w.numel(), dtype=torch.int16, device=w.device)
    L.vcr_split_bf16x3_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.vcr_split_bf16x3_f32.restype = C.c_int
    check(L.vcr_split_bf16x3_f32(ptr(w), ptr(planes), w.numel(), C.c_void_p(stream_ptr())), "vcr_split_bf16x3_f32")
    return planes


@_guarded
def linear_bf16x3(x, w_planes, n_out, bias=None, relu=False, residual=None, out=None, ln=None, want_stats=False):
    """vcr_linear_bf16x3_f32; ln / want_stats as in linear() (w_planes = split_bf16x3 of the folded weight)."""
    L = lib()
    M, K = x.shape
    y = out if out is not None else _f32(M, n_out, device=x.device)
    stats = _f32(M, n_out // 64, 2, device=x.device) if want_stats else None
    a = LinearArgs(ptr(x), x.stride(0), None, ptr(bias), ptr(residual),
                   residual.stride(0) if residual is not None else 0, ptr(y), y.stride(0), M, n_out, K, int(relu))
    if ln is not None:
        a.ln_stats_in, a.ln_nseg, a.ln_colsum, a.ln_eps = ptr(ln[0]), ln[0].shape[1], ptr(ln[1]), ln[2]
    a.stats_out = ptr(stats)
    L.vcr_linear_bf16x3_f32.argtypes = [C.POINTER(LinearArgs), C.c_void_p, C.c_void_p]
    L.vcr_linear_bf16x3_f32.restype = C.c_int
    check(L.vcr_linear_bf16x3_f32(C.byref(a), ptr(w_planes), C.c_void_p(stream_ptr())), "vcr_linear_bf16x3_f32")
    return (y, stats) if want_stats else y


@_guarded
def layernorm(x, a, b, eps=1e-6, residual=None, xyz4=None):
    M, Cc = x.shape
    y = _f32(M, Cc, device=x.device)
    side4 = _f32(M, 4, device=x.device) if xyz4 is not None else None
    call("vcr_layernorm_f32", LayerNormArgs(ptr(x), x.stride(0), ptr(a), ptr(b), eps, ptr(residual),
                                            residual.stride(0) if residual is not None else 0, ptr(y), Cc, M, Cc,
                                            ptr(xyz4), ptr(side4)))
    return (y, side4) if xyz4 is not None else y


@_guarded
def rowside(x, xyz4, scale=1.0):
    M, Cc = x.shape
    y, side4 = _f32(M, Cc, device=x.device), _f32(M, 4, device=x.device)
    call("vcr_rowside_f32", RowsideArgs(ptr(x), x.stride(0), M, Cc, scale, ptr(y), Cc, ptr(xyz4), ptr(side4)))
    return y, side4


@_guarded
def edgeconv(pq, idx, n_per_cloud, w2, b2, bf16x3=False):
    M = pq.shape[0]
    k = idx.shape[-1]
    x1, x2 = _f32(M, 128, device=pq.device), _f32(M, 128, device=pq.device)
    call("vcr_edgeconv_bf16x3_f32" if bf16x3 else "vcr_edgeconv_f32", EdgeconvArgs(ptr(pq), pq.stride(0), ptr(idx), k, M, n_per_cloud, ptr(w2), ptr(b2),
                                          ptr(x1), 128, ptr(x2), 128))
    return x1, x2


def knn_order(xyz4, feat_t=None, sq=None, guard=False, guard_ratio=0.0):
    """vcr_knn_order_f32: the clouds' Morton ranking and what the ordered kNN search reads (vcr_knn_args.perm ...), as a dict.
    guard: also the per-cloud verdict ord_ok [B] / statistic ord_stat [B] on whether the FEATURE tiles are compact enough for the
    ordered search to pay (knn_pair(order=) hands ord_ok on; results never depend on it)."""
    B, N, _ = xyz4.shape
    T = (N + 15) // 16
    e = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device=xyz4.device)
    o = {"perm": e(B, N, dt=torch.int32), "xyz4_p": e(B, N, 4), "cen4": e(B, T, 4), "cen4_rad": e(B, T), "cen4_sqmax": e(B, T)}
    if feat_t is not None:
        o.update(feat_p=e(B, N, 64), sq_p=e(B, N), cen64=e(B, T, 64), cen64_sq=e(B, T), cen64_rad=e(B, T), cen64_sqmax=e(B, T))
        if guard:
            o.update(ord_ok=e(B, dt=torch.int32), ord_stat=e(B))
    g = lambda n: ptr(o.get(n))
    a = KnnOrderArgs(ptr(xyz4), ptr(feat_t), feat_t.stride(1) if feat_t is not None else 0, ptr(sq), B, N, g("perm"), g("xyz4_p"),
                     g("cen4"), g("cen4_rad"), g("cen4_sqmax"), g("feat_p"), g("sq_p"), g("cen64"), g("cen64_sq"), g("cen64_rad"),
                     g("cen64_sqmax"), g("ord_ok"), g("ord_stat"), float(guard_ratio))
    call("vcr_knn_order_f32", a)
    return o


@_guarded
def gathermax(pq, Cc, idx, n_per_cloud, variant=0, order=None):
    """variant: 0 automatic, 1 = gathers through L2, 32 / 16 / 8 = out of LDS with that channel slice (vcr_gathermax_args).
    order: int32 [M], per cloud a permutation of its point numbers -- the order in which the L2 form's waves take the points."""
    M = pq.shape[0]
    k = idx.shape[-1]
    y = _f32(M, Cc, device=pq.device)
    call("vcr_gathermax_f32", GathermaxArgs(ptr(pq), pq.stride(0), Cc, ptr(idx), k, M, n_per_cloud, ptr(y), Cc, variant, ptr(order)))
    return y


@_guarded
def sdpa(q, k, v, nbatch, heads, nq, nk, scale, kv_batch_shift=0, key_keep=None, want_rowstat=False, pv=True,
         score_out=None, bf16x3=False, groups=None, key_index=None, nk_src=0, split=False, variant=0):
    """q [nbatch*nq, >=heads*128] (row views allowed), k/v likewise -> out [nbatch*nq, heads*128].
    score_out [nbatch, heads, nq, ld]: also keep the scaled scores (statistics pass of the partial path).
    groups = (ngroups, q_stride, k_stride, v_stride): that many problems in one launch, group g at element offset
    g * stride of q / k / v -> out [ngroups, nbatch*nq, heads*128]."""
    ng = groups[0] if groups else 1
    out = _f32(ng * nbatch * nq, heads * 128, device=q.device) if pv else None
    rs = _f32(nbatch, heads, nq, 2, device=q.device) if want_rowstat else None
    a = SdpaArgs(ptr(q), q.stride(0), ptr(k), k.stride(0), ptr(v) if pv else None, v.stride(0) if pv else 0, ptr(out),
                 heads * 128 if pv else 0, nbatch, heads, nq, nk, scale, kv_batch_shift, ptr(key_keep), ptr(rs),
                 ptr(score_out), score_out.stride(2) if score_out is not None else 0)
    if groups:
        a.ngroups, a.q_group_stride, a.k_group_stride, a.v_group_stride = groups
        a.out_group_stride = nbatch * nq * heads * 128
    if key_index is not None:       # int32 [nbatch, nk]: the keys are these rows of the nk_src rows per key batch
        a.key_index, a.nk_src = ptr(key_index), int(nk_src)
    # scratch for a key split: statistics passes (small), attention-output launches of less than one round (planes)
    work = _f32(4 * ng * (nbatch * nq * heads * 128 * (1 if pv else 0) + nbatch * heads * nq * 2), device=q.device) if split else None
    a.split_work, a.split_work_floats = ptr(work), (work.numel() if split else 0)
    a.variant = int(variant)                              # 1 = the tile kernel, 2 = the persistent kernel (where it applies)
    call("vcr_sdpa_bf16x3_f32" if bf16x3 else "vcr_sdpa_f32", a)
    if groups:
        out = out.view(ng, nbatch * nq, heads * 128)
    return (out, rs) if want_rowstat else out


@_guarded
def keymass(score, rowstat, nk, q_batch_shift):
    """vcr_keymass_f32: score [nbatch, heads, nq, ld], rowstat [nbatch, heads, nq, 2] -> mass [nbatch, nk] by KEY batch."""
    nb, h, nq, ld = score.shape
    mass = _f32(nb, nk, device=score.device)
    call("vcr_keymass_f32", KeymassArgs(ptr(score), ld, nb, h, nq, nk, ptr(rowstat), q_batch_shift, ptr(mass)))
    return mass


@_guarded
def softcorr(q, k, qside4, kside4, nbatch, nq, nk, mode=0, scale=1.0, split=False):
    corr4 = _f32(nbatch * nq, 4, device=q.device)
    work = _f32(4 * nbatch * nq * 8, device=q.device) if split else None
    call("vcr_softcorr_f32", SoftcorrArgs(ptr(q), q.stride(0), ptr(k), k.stride(0), ptr(qside4), ptr(kside4),
                                          ptr(corr4), nbatch, nq, nk, q.shape[1], mode, scale, ptr(work),
                                          work.numel() if split else 0))
    return corr4


@_guarded
def rigid_svd(src, corr, want_h=False):
    """src/corr [B,K,>=3] rows -> R [B,3,3], t [B,3], R_ba, t_ba (and H when asked)."""
    B, K, _ = src.shape
    dev = src.device
    R, t, Rb, tb = _f32(B, 3, 3, device=dev), _f32(B, 3, device=dev), _f32(B, 3, 3, device=dev), _f32(B, 3, device=dev)
    H = _f32(B, 3, 3, device=dev) if want_h else None
    call("vcr_rigid_svd_f32", RigidSvdArgs(ptr(src), src.stride(1), ptr(corr), corr.stride(1), B, K, ptr(R), ptr(t),
                                           ptr(Rb), ptr(tb), ptr(H)))
    return (R, t, Rb, tb, H) if want_h else (R, t, Rb, tb)


@_guarded
def pairscore(own, strm, nbatch, n_own, n_str, op, score=0, scale=1.0, own_side4=None, str_side4=None,
              shift=0, str_stat2=None, str_stat_stride=None, mass=None, accumulate=False, want_argmax=False,
              score_out=None, variant=0, split=False):
    """vcr_pairscore_f32: op 0 -> corr4; op 1 -> (stat2 [nbatch*n_own,2], argmax or None); op 2 -> mass.
    score_out (op 1): [nbatch, n_own, ld] buffer that also receives the scores.  split: give the launch scratch to split
    the streamed side over several workgroups when its grid calls for it (op 1 without argmax)."""
    dev = own.device
    corr4 = _f32(nbatch * n_own, 4, device=dev) if op == 0 else None
    stat2 = _f32(nbatch * n_own, 2, device=dev) if op == 1 else None
    amax = torch.empty(nbatch * n_own, dtype=torch.int32, device=dev) if (op == 1 and want_argmax) else None
    if op == 2 and mass is None:
        mass = _f32(nbatch, n_own, device=dev)
    work = _f32(4 * nbatch * n_own * (8 if op == 0 else 2), device=dev) if split else None
    call("vcr_pairscore_f32", PairscoreArgs(
        ptr(own), own.stride(0), ptr(strm), strm.stride(0), ptr(own_side4), ptr(str_side4), nbatch, n_own, n_str,
        own.shape[1], score, scale, shift, op, ptr(corr4), ptr(stat2), ptr(amax), ptr(str_stat2),
        int(str_stat_stride if str_stat_stride is not None else n_str * 2), ptr(mass), int(accumulate),
        ptr(score_out), score_out.stride(1) if score_out is not None else 0, variant, ptr(work),
        work.numel() if split else 0))
    if op == 0:
        return corr4
    if op == 1:
        return stat2, amax
    return mass


@_guarded
def scoremass(score, n_cols, row_stat2):
    """vcr_scoremass_f32 on score [nbatch, n_rows, ld]: (col_stat2 [nbatch,n_cols,2], col_mass, row_mass)."""
    nb, n_rows, ld = score.shape
    cs, cm, rm = _f32(nb, n_cols, 2, device=score.device), _f32(nb, n_cols, device=score.device), \
        _f32(nb, n_rows, device=score.device)
    call("vcr_scoremass_f32", ScoremassArgs(ptr(score), ld, nb, n_rows, n_cols, ptr(row_stat2), ptr(cs), ptr(cm), ptr(rm)))
    return cs, cm, rm


@_guarded
def make_pairs(cloud, R_ab, t_ab, pick, perm_src, perm_tgt, keep):
    """vcr_make_pairs_f32: cloud [B,P,3] f32, R_ab [B,3,3] / t_ab [B,3] f64, index maps [B,N] int32 ->
    (src [B,3,keep], tgt [B,3,keep])."""
    B, P, _ = cloud.shape
    N = pick.shape[1]
    for x, dt in ((cloud, torch.float32), (R_ab, torch.float64), (t_ab, torch.float64), (pick, torch.int32),
                  (perm_src, torch.int32), (perm_tgt, torch.int32)):
        if x.dtype != dt or not x.is_contiguous():
            raise VcrHipError("make_pairs: wrong dtype or non-contiguous argument")
    src, tgt = _f32(B, 3, keep, device=cloud.device), _f32(B, 3, keep, device=cloud.device)
    call("vcr_make_pairs_f32", MakePairsArgs(ptr(cloud), P, ptr(R_ab), ptr(t_ab), ptr(pick), ptr(perm_src),
                                             ptr(perm_tgt), B, N, keep, ptr(src), ptr(tgt)))
    return src, tgt


class IcpArgs(C.Structure):
    _fields_ = [("src4", f32p), ("dst4", f32p), ("B", C.c_int), ("N", C.c_int), ("M", C.c_int),
                ("max_iterations", C.c_int), ("tolerance", C.c_float), ("final4", f32p), ("R", f32p), ("t", f32p),
                ("R_ba", f32p), ("t_ba", f32p), ("iterations", f32p)]


@_guarded
def to_rows4(x_cf):
    """[B,3,N] channels-first points -> [B,N,4] rows (x, y, z, |p|^2) (layout plumbing for the C-ABI)."""
    L = lib()
    B, _, N = x_cf.shape
    x = x_cf.contiguous().float()
    out = _f32(B, N, 4, device=x.device)
    L.vcr_rows4_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]; L.vcr_rows4_f32.restype = C.c_int
    check(L.vcr_rows4_f32(ptr(x), ptr(out), B, N, C.c_void_p(stream_ptr())), "vcr_rows4_f32")
    return out


@_guarded
def rows4_pq(x_cf, wpq, bpq):
    """vcr_rows4_pq_f32: [B,3,N] -> ([B,N,4] rows, [B*N, C] = W xyz + b with W [C, >= 3])."""
    L = lib()
    B, _, N = x_cf.shape
    x = x_cf.contiguous().float()
    Cc = wpq.shape[0]
    out, pq = _f32(B, N, 4, device=x.device), _f32(B * N, Cc, device=x.device)
    L.vcr_rows4_pq_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_int, C.c_void_p]
    L.vcr_rows4_pq_f32.restype = C.c_int
    check(L.vcr_rows4_pq_f32(ptr(x), ptr(out), B, N, ptr(wpq), wpq.stride(0), ptr(bpq), Cc, ptr(pq), Cc,
                             C.c_void_p(stream_ptr())), "vcr_rows4_pq_f32")
    return out, pq


@_guarded
def icp(src_cf, dst_cf, max_iterations=10, tolerance=0.001):
    """ICP.forward (model/icp_model.py:26-48) on the device: returns (final [B,3,N], R, t, R_ba, t_ba, iters)."""
    L = lib()
    B, _, N = src_cf.shape
    M = dst_cf.shape[2]
    dev = src_cf.device
    src4, dst4 = to_rows4(src_cf.float()), to_rows4(dst_cf.float())
    final4 = _f32(B, N, 4, device=dev)
    R, t, Rb, tb = _f32(B, 3, 3, device=dev), _f32(B, 3, device=dev), _f32(B, 3, 3, device=dev), _f32(B, 3, device=dev)
    iters = torch.zeros(1, dtype=torch.int32, device=dev)
    L.vcr_icp_workspace_bytes.argtypes = [C.c_int, C.c_int]; L.vcr_icp_workspace_bytes.restype = C.c_size_t
    L.vcr_icp_f32.argtypes = [C.POINTER(IcpArgs), C.c_void_p, C.c_size_t, C.c_void_p]; L.vcr_icp_f32.restype = C.c_int
    nbytes = L.vcr_icp_workspace_bytes(B, N)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    a = IcpArgs(ptr(src4), ptr(dst4), B, N, M, max_iterations, tolerance, ptr(final4), ptr(R), ptr(t), ptr(Rb),
                ptr(tb), ptr(iters))
    check(L.vcr_icp_f32(C.byref(a), C.c_void_p(ws.data_ptr() + off), nbytes, C.c_void_p(stream_ptr())), "vcr_icp_f32")
    return final4[:, :, :3].transpose(1, 2).contiguous(), R, t, Rb, tb, iters


class PoseStepArgs(C.Structure):
    _fields_ = [("R_i", f32p), ("t_i", f32p), ("B", C.c_int), ("N", C.c_int), ("in_cf", f32p), ("out_cf", f32p),
                ("compose", C.c_int), ("R_f", f32p), ("t_f", f32p), ("R_ba", f32p), ("t_ba", f32p)]


@_guarded
def pose_step(R_i, t_i, cloud=None, R_f=None, t_f=None):
    """vcr_pose_step_f32: one step of vcrnetIter's bookkeeping (vcrnet_model.py:32-38).  cloud [B,3,N] -> the moved cloud
    R_i cloud + t_i (util/util.py:91-96).  With (R_f, t_f) the composed pose so far: they are REPLACED by (R_i R_f,
    R_i t_f + t_i) (new tensors; the inputs are not modified) and (R_ba, t_ba) is the inverse of the composition; without
    them the composition is (R_i, t_i) itself.  Returns (moved or None, R_f, t_f, R_ba, t_ba)."""
    B = R_i.shape[0]
    dev = R_i.device
    R_i, t_i = R_i.contiguous().float(), t_i.contiguous().float()
    moved = None
    if cloud is not None:
        cloud = cloud.contiguous().float()
        moved = torch.empty_like(cloud)
    if R_f is not None:
        R_f, t_f = R_f.contiguous().float().clone(), t_f.contiguous().float().clone()
    R_ba, t_ba = _f32(B, 3, 3, device=dev), _f32(B, 3, device=dev)
    call("vcr_pose_step_f32", PoseStepArgs(ptr(R_i), ptr(t_i), B, cloud.shape[2] if cloud is not None else 0, ptr(cloud),
                                           ptr(moved), 1 if R_f is not None else 2, ptr(R_f), ptr(t_f), ptr(R_ba), ptr(t_ba)))
    if R_f is None:
        R_f, t_f = R_i, t_i
    return moved, R_f, t_f, R_ba, t_ba


@_guarded
def edgerows(pq, Cc, idx, n_per_cloud, ymax=None, zero_to=0):
    """pq [M, 2C] (P | Q), idx [M,k] -> per-edge rows relu(P[nbr] + Q[i]) as [M*k, C].  ymax (row view [M, >= zero_to]):
    also the max over each point's k rows in columns 0..C-1, and zeros in columns C..zero_to-1."""
    M, k = idx.shape
    h = _f32(M * k, Cc, device=pq.device)
    call("vcr_edgerows_f32", EdgerowsArgs(ptr(pq), pq.stride(0), Cc, ptr(idx), k, M, n_per_cloud, ptr(h), Cc,
                                          ptr(ymax), ymax.stride(0) if ymax is not None else 0, zero_to))
    return h


@_guarded
def edgechain(pq, idx, n_per_cloud, w2, b2, w3, b3, w4, b4, out=None):
    """vcr_edgechain_f32: DGCNN's conv1-gather -> conv2 -> conv3 -> conv4 with the four maxima, one kernel (k = 20 / 40).
    pq [M, >= 128] (P | Q), idx [M, k] -> out [M, 512] = (x1 | x2 | x3 | x4)."""
    M, k = idx.shape
    out = out if out is not None else _f32(M, 512, device=pq.device)
    call("vcr_edgechain_f32", EdgechainArgs(ptr(pq), pq.stride(0), ptr(idx), k, M, n_per_cloud, ptr(w2), ptr(b2), ptr(w3),
                                            ptr(b3), ptr(w4), ptr(b4), ptr(out), out.stride(0)))
    return out


@_guarded
def segmax(x, M, k, out=None):
    """x [M*k, C] edge rows -> [M, C] max over each point's k rows (out may be a strided row view)."""
    Cc = x.shape[1]
    y = out if out is not None else _f32(M, Cc, device=x.device)
    call("vcr_segmax_f32", SegmaxArgs(ptr(x), x.stride(0), M, k, Cc, ptr(y), y.stride(0)))
    return y


@_guarded
def rankselect(values, K, want_order=True, want_mask=False, largest=True):
    """values [nbatch, n] (any element stride along n, e.g. one column of a [nbatch, n, 2] record)."""
    nb, n = values.shape
    order = torch.empty(nb, K, dtype=torch.int32, device=values.device) if want_order else None
    mask = torch.empty(nb, n, dtype=torch.uint8, device=values.device) if want_mask else None
    stride = values.stride(1) if n > 1 else 1
    if values.stride(0) != n * stride:
        values, stride = values.contiguous(), 1
    call("vcr_rankselect_f32", RankselectArgs(ptr(values), nb, n, K, ptr(order), ptr(mask), int(largest), stride))
    return order, mask


@_guarded
def gather_rows(x, idx, nbatch, n_in, via=None):
    """x [nbatch*n_in, C] rows, idx [nbatch, n_out] int32 -> [nbatch*n_out, C]; with via [nbatch, n_via] int32 the
    row taken is via[b][idx[b][r]]."""
    n_out = idx.shape[1]
    Cc = x.shape[1]
    out = _f32(nbatch * n_out, Cc, device=x.device)
    call("vcr_gather_rows_f32", GatherArgs(ptr(x), x.stride(0), n_in, ptr(idx.contiguous()), nbatch, n_out, Cc,
                                           ptr(out), Cc, ptr(via), via.shape[1] if via is not None else 0))
    return out
