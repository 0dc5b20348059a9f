// Whole-forward driver: VCRNet.forward (model/vcrnet_model.py:495-518) as one host call that
// enqueues every kernel of the path on the caller's stream.  No allocation, no synchronisation.
//
// Both clouds are batched as ONE 2B batch (rows 0..B*N-1 = src, B*N.. = tgt): emb_nn has shared
// weights (vcrnet_model.py:499-500) and the Transformer runs both directions with the same weights
// (transformer.py:269-270).  With that ordering the decoder stream for batch b is "the cloud itself"
// and its encoder memory is the OTHER cloud's, i.e. batch (b + B) mod 2B -- a kv_batch_shift in the
// attention kernel, no data movement.
#include "common.h"

namespace {

struct Bump {
  unsigned char* base; size_t off, cap;
  template <class T> T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = reinterpret_cast<T*>(base + off);
    off += n * sizeof(T);
    return p;
  }
};

struct Ws {
  float *xyz4, *feat64, *sq64, *pq1, *cat, *pq3, *emb;
  int32_t *idx1, *idx3;
  float *ln, *qkv, *att, *e1, *e2, *mem, *hid, *d1, *d2, *d3, *qc, *kvc, *embf, *side4;
  float *st_emb, *st_e1, *st_e2, *st_d1, *st_d2;       // [M, E/64, 2] LayerNorm partial sums
  size_t bytes;
};

Ws carve(void* base, int B, int N, int k, int E, int F) {
  Bump bp{reinterpret_cast<unsigned char*>(base), 0, 0};
  const size_t M = (size_t)2 * B * N;
  Ws w;
  w.xyz4 = bp.take<float>(M * 4);   w.feat64 = bp.take<float>(M * 64); w.sq64 = bp.take<float>(M);
  w.idx1 = bp.take<int32_t>(M * k); w.idx3 = bp.take<int32_t>(M * k);
  w.pq1 = bp.take<float>(M * 256);  w.cat = bp.take<float>(M * 512);   w.pq3 = bp.take<float>(M * 512);
  w.emb = bp.take<float>(M * E);
  w.ln = bp.take<float>(M * E);     w.qkv = bp.take<float>(M * 3 * E); w.att = bp.take<float>(M * E);
  w.e1 = bp.take<float>(M * E);     w.e2 = bp.take<float>(M * E);      w.mem = bp.take<float>(M * E);
  w.hid = bp.take<float>(M * F);
  w.d1 = bp.take<float>(M * E);     w.d2 = bp.take<float>(M * E);      w.d3 = bp.take<float>(M * E);
  w.qc = bp.take<float>(M * E);     w.kvc = bp.take<float>(M * 2 * E);
  w.embf = bp.take<float>(M * E);   w.side4 = bp.take<float>(M * 4);
  const size_t sn = M * (E / 64) * 2;
  w.st_emb = bp.take<float>(sn); w.st_e1 = bp.take<float>(sn); w.st_e2 = bp.take<float>(sn);
  w.st_d1 = bp.take<float>(sn);  w.st_d2 = bp.take<float>(sn);
  w.bytes = bp.off + 256;
  return w;
}

struct Runner {
  hipStream_t stream; vcr_trace* tr; int rc = 0;
  void mark(const char* name) {
    if (!tr) return;
    if (tr->count < VCR_TRACE_MAX) tr->names[tr->count] = name;
    if (tr->events && tr->count < tr->capacity) (void)hipEventRecord((hipEvent_t)tr->events[tr->count], stream);
    ++tr->count;
  }
  void finish() {
    if (tr && tr->events && tr->count < tr->capacity) (void)hipEventRecord((hipEvent_t)tr->events[tr->count], stream);
  }
  bool ok(int r) { if (rc == 0 && r != 0) rc = r; return rc == 0; }

  // ln_stats + ln: LayerNorm fused into the A operand (statistics from the producer's epilogue);
  // stats_out: this launch's epilogue writes the (sum, sum^2) partials of its output rows for a later LayerNorm.
  bool linear(const char* nm, const float* x, int ldx, const float* w, const void* wsplit, const float* b, float* y,
              int ldy, int M, int N, int K, int relu, const float* res = nullptr, int ldr = 0,
              const float* ln_stats = nullptr, const vcr_norm_w* ln = nullptr, float* stats_out = nullptr) {
    if (rc) return false;
    mark(nm);
    vcr_linear_args a{x, ldx, w, b, res, ldr, y, ldy, M, N, K, relu};
    if (ln_stats) { a.ln_stats_in = ln_stats; a.ln_nseg = K / 64; a.ln_a = ln->ln_a; a.ln_b = ln->ln_b; a.ln_eps = 1e-6f; }
    a.stats_out = stats_out;
    return ok(wsplit ? vcr_linear_bf16x3_f32(&a, wsplit, stream) : vcr_linear_f32(&a, stream));
  }
  bool norm(const char* nm, const float* x, const vcr_norm_w& n, float* y, int M, int E,
            const float* res = nullptr, const float* xyz4 = nullptr, float* side4 = nullptr) {
    if (rc) return false;
    mark(nm);
    vcr_layernorm_args a{x, E, n.ln_a, n.ln_b, 1e-6f, res, E, y, E, M, E, xyz4, side4};
    return ok(vcr_layernorm_f32(&a, stream));
  }
  bool sdpa(const char* nm, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out,
            int ldo, int nb, int heads, int nq, int nk, int shift) {
    if (rc) return false;
    mark(nm);
    vcr_sdpa_args a{q, ldq, k, ldk, v, ldv, out, ldo, nb, heads, nq, nk, 1.0f / sqrtf(128.f), shift, nullptr, nullptr};
    return ok(vcr_sdpa_f32(&a, stream));
  }
};

int forward_impl(const vcr_vcrnet_weights* W, const vcr_vcrnet_io* io, void* workspace, size_t ws_bytes,
                 vcr_stream_t stream, vcr_trace* tr) {
  if (!W || !io || !workspace || !io->src_cf || !io->tgt_cf || !io->corr4 || !io->src4 || !io->R_ab || !io->t_ab)
    return VCR_EINVAL;
  const int B = io->B, N = io->N, E = W->E, F = W->F, k = W->k;
  if (B <= 0 || N <= 0 || E != 512 || W->heads * 128 != E || k <= 0 || k > 40 || k + 1 > N) return VCR_EINVAL;
  if (W->has_pointer == 1 && (F % 128)) return VCR_EINVAL;
  if (((uintptr_t)workspace) & 255) return VCR_EINVAL;
  Ws w = carve(workspace, B, N, k, E, F);
  if (ws_bytes < w.bytes) return VCR_EWORKSPACE;
  const int M1 = B * N, M2 = 2 * M1;
  if (tr) tr->count = 0;
  Runner R{(hipStream_t)stream, tr};
#define SP(site) (W->linear_mode == 1 ? W->split.site : nullptr)

  // ---- emb_nn = LPDNet on both clouds (lpdnet_model.py:103-137)
  for (int c = 0; c < 2 && R.rc == 0; ++c) {
    R.mark(c ? "pointwise:tgt" : "pointwise:src");
    vcr_pointwise_args a{c ? io->tgt_cf : io->src_cf, B, N, W->c1_w, W->c1_b, W->c2_w, W->c2_b,
                         w.xyz4 + (size_t)c * M1 * 4, w.feat64 + (size_t)c * M1 * 64, w.sq64 + (size_t)c * M1};
    R.ok(vcr_pointwise_f32(&a, R.stream));
  }
  if (R.rc == 0) {
    R.mark("knn:feat64");
    vcr_knn_args a{w.feat64, 64, w.sq64, 2 * B, N, 64, k, w.idx1};
    R.ok(vcr_knn_f32(&a, R.stream));
  }
  R.linear("linear:dg1_pq", w.feat64, 64, W->dg1_wpq, SP(dg1_pq), W->dg1_bpq, w.pq1, 256, M2, 256, 64, 0);
  if (R.rc == 0) {
    R.mark("edgeconv:dg1_dg2");
    vcr_edgeconv_args a{w.pq1, 256, w.idx1, k, M2, N, W->dg2_w, W->dg2_b, w.cat, 512, w.cat + 128, 512};
    R.ok(vcr_edgeconv_f32(&a, R.stream));
  }
  if (R.rc == 0) {
    R.mark("knn:xyz");
    vcr_knn_args a{w.xyz4, 4, nullptr, 2 * B, N, 4, k, w.idx3};
    R.ok(vcr_knn_f32(&a, R.stream));
  }
  R.linear("linear:sn1_pq", w.cat + 128, 512, W->sn1_wpq, SP(sn1_pq), W->sn1_bpq, w.pq3, 512, M2, 512, 128, 0);
  if (R.rc == 0) {
    R.mark("gathermax:sn1");
    vcr_gathermax_args a{w.pq3, 512, 256, w.idx3, k, M2, N, w.cat + 256, 512};
    R.ok(vcr_gathermax_f32(&a, R.stream));
  }
  R.linear("linear:conv3", w.cat, 512, W->c3_w, SP(c3), W->c3_b, w.emb, E, M2, E, 512, 1, nullptr, 0, nullptr, nullptr,
           (W->has_pointer == 1 && W->linear_mode == 0) ? w.st_emb : nullptr);

  // ---- pointer (transformer.py:264-272) + residual (vcrnet_model.py:504-505)
  const float* head_emb = w.embf;
  if (W->has_pointer == 1 && W->linear_mode == 0) {
    // fp32 path with LayerNorm fused into the consuming linears: each producer of a
    // residual stream writes per-row (sum, sum^2) partials from its epilogue, each consumer normalises its A
    // fragments on the fly -- six LayerNorm launches and their 2 x 67 MB round trips are gone.
    const int H = W->heads;
    R.linear("linear:enc.qkv", w.emb, E, W->enc_self.wqkv, nullptr, W->enc_self.bqkv, w.qkv, 3 * E, M2, 3 * E, E, 0,
             nullptr, 0, w.st_emb, &W->enc_ln0);
    R.sdpa("sdpa:enc.self", w.qkv, 3 * E, w.qkv + E, 3 * E, w.qkv + 2 * E, 3 * E, w.att, E, 2 * B, H, N, N, 0);
    R.linear("linear:enc.wo", w.att, E, W->enc_self.wo, nullptr, W->enc_self.bo, w.e1, E, M2, E, E, 0, w.emb, E,
             nullptr, nullptr, w.st_e1);
    R.linear("linear:enc.ffn1", w.e1, E, W->enc_ffn.w1, nullptr, W->enc_ffn.b1, w.hid, F, M2, F, E, 1, nullptr, 0,
             w.st_e1, &W->enc_ln1);
    R.linear("linear:enc.ffn2", w.hid, F, W->enc_ffn.w2, nullptr, W->enc_ffn.b2, w.e2, E, M2, E, F, 0, w.e1, E,
             nullptr, nullptr, w.st_e2);
    // decoder; batch b attends to the encoder memory (= enc.norm(e2), applied inside the K/V projection) of
    // batch (b + B) mod 2B
    R.linear("linear:dec.qkv", w.emb, E, W->dec_self.wqkv, nullptr, W->dec_self.bqkv, w.qkv, 3 * E, M2, 3 * E, E, 0,
             nullptr, 0, w.st_emb, &W->dec_ln0);
    R.sdpa("sdpa:dec.self", w.qkv, 3 * E, w.qkv + E, 3 * E, w.qkv + 2 * E, 3 * E, w.att, E, 2 * B, H, N, N, 0);
    R.linear("linear:dec.self.wo", w.att, E, W->dec_self.wo, nullptr, W->dec_self.bo, w.d1, E, M2, E, E, 0, w.emb, E,
             nullptr, nullptr, w.st_d1);
    R.linear("linear:dec.cross.q", w.d1, E, W->dec_cross.wq, nullptr, W->dec_cross.bq, w.qc, E, M2, E, E, 0, nullptr, 0,
             w.st_d1, &W->dec_ln1);
    R.linear("linear:dec.cross.kv", w.e2, E, W->dec_cross.wkv, nullptr, W->dec_cross.bkv, w.kvc, 2 * E, M2, 2 * E, E, 0,
             nullptr, 0, w.st_e2, &W->enc_norm);
    R.sdpa("sdpa:dec.cross", w.qc, E, w.kvc, 2 * E, w.kvc + E, 2 * E, w.att, E, 2 * B, H, N, N, B);
    R.linear("linear:dec.cross.wo", w.att, E, W->dec_cross.wo, nullptr, W->dec_cross.bo, w.d2, E, M2, E, E, 0, w.d1, E,
             nullptr, nullptr, w.st_d2);
    R.linear("linear:dec.ffn1", w.d2, E, W->dec_ffn.w1, nullptr, W->dec_ffn.b1, w.hid, F, M2, F, E, 1, nullptr, 0,
             w.st_d2, &W->dec_ln2);
    R.linear("linear:dec.ffn2", w.hid, F, W->dec_ffn.w2, nullptr, W->dec_ffn.b2, w.d3, E, M2, E, F, 0, w.d2, E);
    R.norm("layernorm:dec.norm+res", w.d3, W->dec_norm, w.embf, M2, E, w.emb, w.xyz4, w.side4);
  } else if (W->has_pointer == 1) {
    const int H = W->heads;
    // encoder layer (pre-norm residual sublayers, transformer.py:156-166) on [src; tgt]
    R.norm("layernorm:enc.sub0", w.emb, W->enc_ln0, w.ln, M2, E);
    R.linear("linear:enc.qkv", w.ln, E, W->enc_self.wqkv, SP(enc_qkv), W->enc_self.bqkv, w.qkv, 3 * E, M2, 3 * E, E, 0);
    R.sdpa("sdpa:enc.self", w.qkv, 3 * E, w.qkv + E, 3 * E, w.qkv + 2 * E, 3 * E, w.att, E, 2 * B, H, N, N, 0);
    R.linear("linear:enc.wo", w.att, E, W->enc_self.wo, SP(enc_wo), W->enc_self.bo, w.e1, E, M2, E, E, 0, w.emb, E);
    R.norm("layernorm:enc.sub1", w.e1, W->enc_ln1, w.ln, M2, E);
    R.linear("linear:enc.ffn1", w.ln, E, W->enc_ffn.w1, SP(enc_ffn1), W->enc_ffn.b1, w.hid, F, M2, F, E, 1);
    R.linear("linear:enc.ffn2", w.hid, F, W->enc_ffn.w2, SP(enc_ffn2), W->enc_ffn.b2, w.e2, E, M2, E, F, 0, w.e1, E);
    R.norm("layernorm:enc.norm", w.e2, W->enc_norm, w.mem, M2, E);
    // decoder layer (transformer.py:169-185); batch b attends to memory of batch (b + B) mod 2B
    R.norm("layernorm:dec.sub0", w.emb, W->dec_ln0, w.ln, M2, E);
    R.linear("linear:dec.qkv", w.ln, E, W->dec_self.wqkv, SP(dec_qkv), W->dec_self.bqkv, w.qkv, 3 * E, M2, 3 * E, E, 0);
    R.sdpa("sdpa:dec.self", w.qkv, 3 * E, w.qkv + E, 3 * E, w.qkv + 2 * E, 3 * E, w.att, E, 2 * B, H, N, N, 0);
    R.linear("linear:dec.self.wo", w.att, E, W->dec_self.wo, SP(dec_self_wo), W->dec_self.bo, w.d1, E, M2, E, E, 0, w.emb, E);
    R.norm("layernorm:dec.sub1", w.d1, W->dec_ln1, w.ln, M2, E);
    R.linear("linear:dec.cross.q", w.ln, E, W->dec_cross.wq, SP(dec_cross_q), W->dec_cross.bq, w.qc, E, M2, E, E, 0);
    R.linear("linear:dec.cross.kv", w.mem, E, W->dec_cross.wkv, SP(dec_cross_kv), W->dec_cross.bkv, w.kvc, 2 * E, M2, 2 * E, E, 0);
    R.sdpa("sdpa:dec.cross", w.qc, E, w.kvc, 2 * E, w.kvc + E, 2 * E, w.att, E, 2 * B, H, N, N, B);
    R.linear("linear:dec.cross.wo", w.att, E, W->dec_cross.wo, SP(dec_cross_wo), W->dec_cross.bo, w.d2, E, M2, E, E, 0, w.d1, E);
    R.norm("layernorm:dec.sub2", w.d2, W->dec_ln2, w.ln, M2, E);
    R.linear("linear:dec.ffn1", w.ln, E, W->dec_ffn.w1, SP(dec_ffn1), W->dec_ffn.b1, w.hid, F, M2, F, E, 1);
    R.linear("linear:dec.ffn2", w.hid, F, W->dec_ffn.w2, SP(dec_ffn2), W->dec_ffn.b2, w.d3, E, M2, E, F, 0, w.d2, E);
    // final decoder norm, + embedding residual, + the head's side record in one pass
    R.norm("layernorm:dec.norm+res", w.d3, W->dec_norm, w.embf, M2, E, w.emb, w.xyz4, w.side4);
  } else if (R.rc == 0) {
    R.mark("layernorm:rowside");
    vcr_rowside_args a{w.emb, E, M2, E, W->has_pointer == 2 ? 2.f : 1.f, w.embf, E, w.xyz4, w.side4};
    R.ok(vcr_rowside_f32(&a, R.stream));
  }

  // ---- head (whole mode) + SVD
  if (R.rc == 0) {
    R.mark("softcorr:head");
    vcr_softcorr_args a{head_emb, E, head_emb + (size_t)M1 * E, E, w.side4, w.side4 + (size_t)M1 * 4,
                        io->corr4, B, N, N, E, W->head_mode, 1.0f / sqrtf((float)E)};
    R.ok(vcr_softcorr_f32(&a, R.stream));
  }
  if (R.rc == 0) {
    R.mark("rigid_svd:ab");
    (void)hipMemcpyAsync(io->src4, w.xyz4, (size_t)M1 * 4 * sizeof(float), hipMemcpyDeviceToDevice, R.stream);
    vcr_rigid_svd_args a{w.xyz4, 4, io->corr4, 4, B, N, io->R_ab, io->t_ab, io->R_ba, io->t_ba, nullptr};
    R.ok(vcr_rigid_svd_f32(&a, R.stream));
  }
  if (R.rc == 0 && io->emb_out)
    (void)hipMemcpyAsync(io->emb_out, w.embf, (size_t)M2 * E * sizeof(float), hipMemcpyDeviceToDevice, R.stream);
  R.finish();
#undef SP
  return R.rc;
}

}  // namespace

extern "C" size_t vcr_vcrnet_workspace_bytes(const vcr_vcrnet_weights* W, int B, int N) {
  if (!W || B <= 0 || N <= 0) return 0;
  return carve(nullptr, B, N, W->k, W->E, W->F).bytes;
}

extern "C" int vcr_vcrnet_forward_f32(const vcr_vcrnet_weights* W, const vcr_vcrnet_io* io, void* ws, size_t bytes,
                                      vcr_stream_t stream) {
  return forward_impl(W, io, ws, bytes, stream, nullptr);
}

extern "C" int vcr_vcrnet_forward_traced_f32(const vcr_vcrnet_weights* W, const vcr_vcrnet_io* io, void* ws,
                                             size_t bytes, vcr_stream_t stream, vcr_trace* tr) {
  return forward_impl(W, io, ws, bytes, stream, tr);
}

extern "C" const char* vcr_strerror(int code) {
  switch (code) {
    case VCR_OK: return "ok";
    case VCR_EINVAL: return "invalid argument (shape, pitch, alignment or null pointer)";
    case VCR_EWORKSPACE: return "workspace too small";
    case VCR_EUNSUPPORTED: return "unsupported configuration";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

extern "C" int vcr_abi_version(void) { return 3; }

// hipEvent helpers so a host language without HIP bindings can time launches on the SAME runtime
// this library is bound to.
extern "C" int vcr_event_create(void** ev) {
  if (!ev) return VCR_EINVAL;
  hipEvent_t e;
  const hipError_t rc = hipEventCreate(&e);
  *ev = rc == hipSuccess ? (void*)e : nullptr;
  return (int)rc;
}
extern "C" int vcr_event_destroy(void* ev) { return ev ? (int)hipEventDestroy((hipEvent_t)ev) : VCR_EINVAL; }
extern "C" int vcr_event_record(void* ev, vcr_stream_t stream) {
  return ev ? (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) : VCR_EINVAL;
}
extern "C" int vcr_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!start || !stop || !ms) return VCR_EINVAL;
  return (int)hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
}
