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

// Workspace plan (round 5): every buffer of the forward with the launches between which it is LIVE, laid out by a static
// interval allocator -- buffers whose lifetimes do not overlap share memory.  Rounds 1-4 carved the workspace with a bump
// allocator (every buffer its own memory: 1.5 GB at BASELINE configs[1], 12 GB at configs[4]); the plan needs what the
// busiest launch has live (the grouped self-attention: embeddings + enc | dec Q | K | V + both outputs = 9 x [2BN, 512]
// floats).  Times are positions in the launch sequence of forward_impl (the merged and the unmerged first sublayers have
// their own numbering up to 15; from the cross-attention's K | V projection on they agree):
//   0 stem  1 kNN (+ dg1_pq)  2 tie replay  3 EdgeConv / DGCNN chain  4 sn1_pq  5 gathermax  6 conv3
//   merged:   7 encdec.qkv  8 self-attention  9 wo pair  10 enc.ffn1 + dec.cross.q  11 enc.ffn2
//   unmerged: 7 enc.qkv  8 enc.self  9 enc.wo  10 enc.ffn1  11 enc.ffn2  12 dec.qkv  13 dec.self  14 dec.self.wo  15 dec.cross.q
//   16 dec.cross.kv  17 cross-attention  18 dec.cross.wo  19 dec.ffn1  20 dec.ffn2  21 dec.norm / rowside  22 head  23 second head
struct Plan {
  struct Req { void** slot; size_t bytes; int birth, death; size_t off; };
  static constexpr int MAXR = 96;
  Req r[MAXR];
  int n = 0;
  bool overflow = false;                                 // more requests than MAXR: carve() reports an impossible size
  bool flat = false;                                     // vcr_vcrnet_weights.workspace_flat: nothing is overlaid
  template <class T> void want(T*& field, size_t count, int birth, int death) {
    field = nullptr;
    if (count == 0) return;
    if (flat) { birth = 0; death = 1 << 30; }
    if (n >= MAXR) { overflow = true; return; }
    r[n++] = Req{reinterpret_cast<void**>(&field), (count * sizeof(T) + 255) & ~(size_t)255, birth, death, 0};
  }
  // largest first, each at the lowest offset where it collides with no placed buffer that is live at the same time
  size_t solve(unsigned char* base) {
    int order[MAXR];
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 1; i < n; ++i) {                        // insertion sort by size, descending (n ~ 60)
      const int x = order[i];
      int j = i - 1;
      while (j >= 0 && r[order[j]].bytes < r[x].bytes) { order[j + 1] = order[j]; --j; }
      order[j + 1] = x;
    }
    size_t total = 0;
    for (int oi = 0; oi < n; ++oi) {
      Req& q = r[order[oi]];
      size_t off = 0;
      for (bool moved = true; moved;) {
        moved = false;
        for (int pj = 0; pj < oi; ++pj) {
          const Req& p = r[order[pj]];
          if (p.death < q.birth || q.death < p.birth) continue;          // never live together
          if (off < p.off + p.bytes && p.off < off + q.bytes) { off = p.off + p.bytes; moved = true; }
        }
      }
      q.off = off;
      if (off + q.bytes > total) total = off + q.bytes;
    }
    for (int i = 0; i < n; ++i) *r[i].slot = base + r[i].off;
    return total;
  }
};

struct Ws {
  float *xyz4, *feat64, *sq64, *pq1, *cat, *pq3, *emb;
  int32_t *idx1, *idx3, *ties;                         // ties: 2 x (count + one slot per row) for the kNN tie replay
  unsigned char* tie_work; size_t tie_work_each;       // 2 x vcr_knn_tie_work_bytes(N): replay scratch of long rows (else NULL)
  float *qkv, *att, *attx, *e1, *e2, *hid, *d1, *d2, *d3, *qc, *kvc, *embf, *side4, *csplit, *asplit;   // att: self-attention output(s), attx: cross
  float* corr_ba;                                      // cycle: the second head's correspondences [B, N, 4]
  long asplit_floats;
  float *st_emb, *st_e1, *st_e2, *st_d1, *st_d2;       // [M, E/64, 2] LayerNorm partial sums
  // partial-overlap mode
  float *rowstat, *keymass; uint8_t* keep;             // cross-attention: [2B,H,N,2], [2B,N], [2B,N]
  float* xsplit;                                        // vcr_sdpa_args.split_work of the cross-attention statistics pass
  float* xscore;                                       // [2B,H,N,roundup32(N)] scaled scores, NULL above 4 GB
  int32_t* xorder;                                     // [2B, nkeep] kept keys, heaviest first
  float *rstat, *cstat, *colsum, *rowsum, *score;      // selectCom: [B,N,2] x2, [B,N] x2, [B,N,roundup32(N)]
  float* rsplit;                                        // vcr_pairscore_args.split_work of the head's statistics pass
  int32_t *sel_s, *sel_t, *amax, *pick;                // [B,K1] x3, [B,K2]
  float *so_e, *to_e, *so_s, *to_s, *peak;             // overlap sets: [B,K1,E] x2, [B,K1,4] x2; [B,K1,2]
  // DGCNN embedding: per-edge activations [2B*N*k, 64 | 64 | 128]
  float *eh1, *eh2, *eh3;
  // vcrnetIter
  float *cur_cf, *Ri, *ti, *Rb, *tb;
  size_t bytes;
};
// vcrnetIter with target reuse (forward_impl's `pass`): the four buffers whose TARGET halves a later pass reads behind the
// cross-attention -- emb (the final residual), d1 (the cross sublayer's residual), qc (its queries), kvc (its keys | values) --
// then live BEHIND the planned workspace, 2 B N x 5 E floats that nothing else is ever laid over: the first pass writes all
// their rows, a later pass only the source rows, and the target rows simply stay.  (The LayerNorm statistics of the target
// rows are consumed in front of the cross-attention, by launches a later pass runs on the source rows only: nothing of them
// needs to persist.)
inline size_t tgt_cache_floats(int B, int N, int E) { return (size_t)2 * B * N * 5 * E; }

// vcrnet_model.py:208-209 and :284 -- Python truncates float64 products, so do we
inline int overlap_k1(int N, double o2) { return (int)((double)N * 0.84 * o2); }
inline int overlap_k2(int N, double o2) { return (int)((double)overlap_k1(N, o2) * 0.52 * o2); }

Ws carve(void* base, int B, int N, int k, int E, int F, int heads, int partial, double o2, int emb_kind, int xscore_limit_mb,
         int merged, int flat) {
  Plan pl;
  pl.flat = flat != 0;
  const size_t M = (size_t)2 * B * N;
  constexpr int END = 23;                                // (the last launch of a forward; vcrnetIter's state lives across forwards)
  Ws w{};
  pl.want(w.xyz4, M * 4, 0, END);   pl.want(w.feat64, M * 64, 0, 3);   pl.want(w.sq64, M, 0, 2);   // (PointNet: conv3 reads feat64 at 3)
  pl.want(w.idx1, M * k, 1, 3);     pl.want(w.idx3, M * k, 1, 5);      pl.want(w.ties, 2 * (1 + M), 0, 2);   // a slot for every row
  // kNN tie replay scratch per search: k > 20 -- slots for the in-launch replay (the lists leave no room for a row image in
  // LDS), while that stays below 1 GiB (live during the kNN launch only: it lies under the Transformer's buffers) --, else what
  // the replay launch needs for rows beyond 10 091 points
  {
    const size_t slots = vcr_knn_tie_slot_bytes(2 * B, N);
    w.tie_work_each = (k > 20 && slots <= ((size_t)1 << 30)) ? slots : vcr_knn_tie_work_bytes(N);
    w.tie_work_each = (w.tie_work_each + 255) & ~(size_t)255;
  }
  pl.want(w.tie_work, w.tie_work_each ? 2 * w.tie_work_each : 0, 1, 2);
  pl.want(w.pq1, M * 256, 0, 4);    pl.want(w.cat, M * 512, 3, 6);     pl.want(w.pq3, M * 512, 4, 5);   // (PointNet: conv4 reads pq1 at 4)
  pl.want(w.emb, M * E, 0, 21);                          // (0 .. 2: the transposed feat64 rows of the 16-query kNN waves; 6 ..: the embeddings)
  // merged: the encoder's and the decoder's Q|K|V side by side ([M, 6E]) and their attention outputs one after the other
  pl.want(w.qkv, M * 3 * E * (merged ? 2 : 1), 7, merged ? 8 : 13);
  pl.want(w.att, M * E * (merged ? 2 : 1), 8, merged ? 9 : 14);
  pl.want(w.e1, M * E, 9, 11);      pl.want(w.e2, M * E, 11, 16);
  pl.want(w.hid, M * F, 10, 20);                         // (FFN hidden rows of both layers; 17: the dense K | V copy of the exact-split cross-attention)
  pl.want(w.d1, M * E, merged ? 9 : 14, 18);
  pl.want(w.qc, M * E, merged ? 10 : 15, 17);            pl.want(w.kvc, M * 2 * E, 16, 17);
  pl.want(w.attx, M * E, 17, 18);
  pl.want(w.d2, M * E, 18, END);                         // (23: the VcpAtt projections of the second head)
  pl.want(w.d3, M * E, 20, END);                         // (22: the VcpAtt projections of the head)
  pl.want(w.embf, M * E, 21, END);  pl.want(w.side4, M * 4, 21, END);
  pl.want(w.csplit, VCR_PAIRSCORE_MAX_SPLIT * (M / 2) * 8, 22, END);   // vcr_softcorr_args.split_work of the soft heads
  pl.want(w.corr_ba, (size_t)B * N * 4, 23, END);
  {
    // planes of a key-split attention-output launch (vcr_sdpa_args.split_work; the grouped self-attention has 2 M rows):
    // the library only splits while the planes stay below 64 MB, i.e. at small batches -- no more than that is set aside
    const size_t rows = 2 * M, ml = (size_t)VCR_SDPA_MAX_SPLIT * rows * heads * 2;
    const size_t want = (size_t)VCR_SDPA_MAX_SPLIT * rows * E, cap = ((size_t)64 << 20) / 4;
    w.asplit_floats = (long)((want < cap ? want : cap) + ml);
    // ... and only for grids of at most half a round: vcr_sdpa_f32 splits the keys when blocks x split <= slots (the same
    // constants and CU count as its launcher; 512 slots on MI355X); the smallest attention-output launch of the forward is the
    // cross-attention (ceil(N / 128) x 2B x heads blocks)
    if ((long)((N + VCR_SDPA_QROWS - 1) / VCR_SDPA_QROWS) * 2 * B * heads * 2 > (long)vcr_cu_count() * VCR_SDPA_WG_PER_CU) w.asplit_floats = 0;
    pl.want(w.asplit, (size_t)w.asplit_floats, 8, 17);
  }
  const size_t sn = M * (E / 64) * 2;
  pl.want(w.st_emb, sn, 6, 12);
  pl.want(w.st_e1, sn, 9, 10);      pl.want(w.st_e2, sn, 11, 16);
  pl.want(w.st_d1, sn, merged ? 9 : 14, merged ? 10 : 15);             pl.want(w.st_d2, sn, 18, 19);
  const size_t K1 = (size_t)overlap_k1(N, o2), K2 = (size_t)overlap_k2(N, o2), B1 = (size_t)B;
  if (partial) {
    pl.want(w.rowstat, M * heads * 2, 17, 17); pl.want(w.keymass, M, 17, 17); pl.want(w.keep, M, 17, 17);
    pl.want(w.xsplit, VCR_SDPA_MAX_SPLIT * M * heads * 2, 17, 17);
    const size_t xs = M * heads * ((N + 31) & ~31);       // keep the cross-attention scores if they fit 4 GB
    const size_t xlimit = xscore_limit_mb > 0 ? (size_t)xscore_limit_mb << 20 : xscore_limit_mb < 0 ? 0 : (size_t)4 << 30;
    pl.want(w.xscore, xs * 4 <= xlimit ? xs : 0, 17, 17);
    pl.want(w.xorder, M, 17, 17);
    pl.want(w.rstat, B1 * N * 2, 22, END); pl.want(w.cstat, B1 * N * 2, 22, END);
    pl.want(w.rsplit, VCR_PAIRSCORE_MAX_SPLIT * B1 * N * 2, 22, END);
    // source-side block first, target-side block right behind it ([2B, ...] like the embeddings): one rank-select and
    // one gather launch then serve both clouds
    pl.want(w.rowsum, 2 * B1 * N, 22, END);
    pl.want(w.score, B1 * N * ((N + 31) & ~31), 22, END);
    pl.want(w.sel_s, 2 * B1 * K1, 22, END);
    pl.want(w.amax, B1 * K1, 22, END);     pl.want(w.pick, B1 * (K2 ? K2 : 1), 22, END);
    pl.want(w.so_e, 2 * B1 * K1 * E, 22, END);
    pl.want(w.so_s, 2 * B1 * K1 * 4, 22, END);
    pl.want(w.peak, B1 * K1 * 2, 22, END);
  }
  if (emb_kind == 1 && k != 20 && k != 40) {              // (k = 20 / 40 run the chain in one kernel: no per-edge tensor at all)
    const size_t Mk = M * k;
    pl.want(w.eh1, Mk * 64, 3, 5); pl.want(w.eh2, Mk * 64, 3, 5);
    pl.want(w.eh3, Mk * 128, 3, 5);                        // (conv4's [M*k, 256] output is only ever max-reduced: never stored)
  }
  pl.want(w.cur_cf, (size_t)B * 3 * N, 0, END);
  pl.want(w.Ri, (size_t)B * 9, 0, END); pl.want(w.ti, (size_t)B * 3, 0, END);
  pl.want(w.Rb, (size_t)B * 9, 0, END); pl.want(w.tb, (size_t)B * 3, 0, END);
  w.bytes = pl.solve(reinterpret_cast<unsigned char*>(base)) + 256;
  if (pl.overflow) w.bytes = ~(size_t)0;                 // (never with the ~60 buffers above: forward_impl then returns VCR_EWORKSPACE)
  if (partial) {
    w.colsum = w.rowsum + B1 * N;
    w.sel_t = w.sel_s + B1 * K1;
    w.to_e = w.so_e + B1 * K1 * E;
    w.to_s = w.so_s + B1 * K1 * 4;
  }
  return w;
}

// enc.qkv + dec.qkv as one GEMM, the two self-attentions as one grouped launch: needs the stacked folded weight (fp32 mode)
constexpr int KNN_ORDERED_MIN_N = 2048;                 // clouds from this size on take the ordered kNN search (see the LPDNet stage)
inline int merged_encdec(const vcr_vcrnet_weights* W) {
  return W->has_pointer == 1 && (W->linear_mode == 0 || W->split.encdec_qkv) && W->fold_encdec_qkv.w && W->fold_encdec_qkv.colsum &&
         W->fold_encdec_qkv.bias;
}

__global__ __launch_bounds__(256) void zero_i32_kernel(int32_t* p, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0;
}
extern "C" int vcr_linear_shapes_(const vcr_linear_args* a, const vcr_linear_args* b, int* shape_a, int* shape_b);   // linear.hip

struct Runner {
  hipStream_t stream; vcr_trace* tr; int rc = 0;
  // A later vcrnetIter pass with target reuse launches its pre-cross-attention linears on the SOURCE rows only.  The one
  // configuration choice a linear's bits depend on is its MFMA shape, which the library picks from the row count among other
  // things: such a launch pins the shape the full-row launch it stands for would take (shape_rows = that row count; 0 = off).
  int shape_rows = 0;
  int plan_nbatch = 0;                                   // likewise for the attention launches' key split (vcr_sdpa_args.plan_nbatch)
  void pin_shape(vcr_linear_args& a, vcr_linear_args* b = nullptr, int full_rows = 0) {   // full_rows: of the launch this one stands for
    const int fm = full_rows ? full_rows : shape_rows;    // (default: the point rows; DGCNN's per-edge linears pass rows x k)
    if (!shape_rows || a.M >= fm || (a.variant & (16 | 1024))) return;
    vcr_linear_args fa = a, fb = b ? *b : a;
    fa.M = fm; fb.M = fm;
    int sa = 0, sb = 0;
    if (vcr_linear_shapes_(&fa, b ? &fb : nullptr, &sa, &sb) != VCR_OK) return;
    a.variant |= sa ? 16 : 1024;
    if (b) b->variant |= sb ? 16 : 1024;
  }
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

  // ln_stats + ln_colsum: LayerNorm folded into this linear (w, b are the folded weight / bias; the row statistics
  // come from the producer's epilogue);
  // stats_out: this launch's epilogue writes the per-64-column (sum, second moment about the segment mean) partials of its
  // output rows for a later LayerNorm (common.h: ln_row_moments).
  bool linear(const char* nm, const float* x, int ldx, const float* w, const void* wsplit, const float* b, float* y,
              int ldy, int M, int N, int K, int relu, const float* res = nullptr, int ldr = 0,
              const float* ln_stats = nullptr, const float* ln_colsum = nullptr, float* stats_out = nullptr) {
    if (rc) return false;
    mark(nm);
    vcr_linear_args a{x, ldx, w, b, res, ldr, y, ldy, M, N, K, relu};
    if (ln_stats) { a.ln_stats_in = ln_stats; a.ln_nseg = K / 64; a.ln_colsum = ln_colsum; a.ln_eps = 1e-6f; }
    a.stats_out = stats_out;
    a.variant = linear_variant;
    if (!wsplit) pin_shape(a);                           // (the exact-split kernel has one configuration)
    return ok(wsplit ? vcr_linear_bf16x3_f32(&a, wsplit, stream) : vcr_linear_f32(&a, stream));
  }
  // the argument block of linear() without launching it, and two such blocks as one launch (vcr_linear_pair_f32)
  vcr_linear_args linear_args(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M, int N, int K,
                              int relu, const float* res = nullptr, int ldr = 0, const float* ln_stats = nullptr,
                              const float* ln_colsum = nullptr, float* stats_out = nullptr) {
    vcr_linear_args a{x, ldx, w, b, res, ldr, y, ldy, M, N, K, relu};
    if (ln_stats) { a.ln_stats_in = ln_stats; a.ln_nseg = K / 64; a.ln_colsum = ln_colsum; a.ln_eps = 1e-6f; }
    a.stats_out = stats_out;
    a.variant = linear_variant;
    return a;
  }
  bool linear2(const char* nm, vcr_linear_args a, vcr_linear_args b) {
    if (rc) return false;
    mark(nm);
    pin_shape(a, &b);
    return ok(vcr_linear_pair_f32(&a, &b, stream));
  }
  bool norm(const char* nm, const float* x, const vcr_norm_w& n, float* y, int M, int E,
            const float* res = nullptr, const float* xyz4 = nullptr, float* side4 = nullptr) {
    if (rc) return false;
    mark(nm);
    vcr_layernorm_args a{x, E, n.ln_a, n.ln_b, 1e-6f, res, E, y, E, M, E, xyz4, side4};
    return ok(vcr_layernorm_f32(&a, stream));
  }
  bool sdpa(const char* nm, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out,
            int ldo, int nb, int heads, int nq, int nk, int shift, const uint8_t* keep = nullptr,
            float* rowstat = nullptr, float* score_out = nullptr, int ld_score = 0, int ngroups = 1, long in_group_stride = 0,
            long out_group_stride = 0, float* split_work = nullptr, long split_floats = 0) {
    if (rc) return false;
    mark(nm);
    vcr_sdpa_args a{q, ldq, k, ldk, v, ldv, out, ldo, nb, heads, nq, nk, 1.0f / sqrtf(128.f), shift, keep, rowstat,
                    score_out, ld_score};
    // key split: statistics passes bring their own scratch; attention-output launches share the driver's planes (only
    // taken by vcr_sdpa_f32 for grids of less than one round, i.e. small batches)
    a.split_work = split_work ? split_work : out ? pv_split : nullptr;
    a.split_work_floats = split_work ? split_floats : out ? pv_split_floats : 0;
    if (ngroups > 1) {
      a.ngroups = ngroups; a.q_group_stride = a.k_group_stride = a.v_group_stride = in_group_stride;
      a.out_group_stride = out_group_stride;
    }
    a.variant = sdpa_variant;
    if (plan_nbatch > nb) a.plan_nbatch = plan_nbatch;    // (a source-only launch of a later vcrnetIter pass: split as the full one)
    // linear_mode 2: the attention-output launches on the bf16 matrix pipe as exact splits; statistics passes stay fp32
    return ok((sdpa_split && out && !rowstat && !score_out) ? vcr_sdpa_bf16x3_f32(&a, stream) : vcr_sdpa_f32(&a, stream));
  }
  bool pairscore(const char* nm, const vcr_pairscore_args& a) {
    if (rc) return false;
    mark(nm);
    return ok(vcr_pairscore_f32(&a, stream));
  }
  bool rank(const char* nm, const float* values, int stride, int nb, int n, int K, int32_t* order, uint8_t* mask,
            int largest) {
    if (rc) return false;
    mark(nm);
    vcr_rankselect_args a{values, nb, n, K, order, mask, largest, stride};
    return ok(vcr_rankselect_f32(&a, stream));
  }
  bool gather(const char* nm, const float* in, int ld, int n_in, const int32_t* idx, int nb, int n_out, int C,
              float* out, const int32_t* via = nullptr, int n_via = 0) {
    if (rc) return false;
    mark(nm);
    vcr_gather_args a{in, ld, n_in, idx, nb, n_out, C, out, C, via, n_via};
    return ok(vcr_gather_rows_f32(&a, stream));
  }

  // Decoder cross-attention.  Partial mode (transformer.py:35-53): soft-max once, total probability mass every
  // KEY receives over heads and queries, keep the int(nk*overlap2) heaviest keys, soft-max again over those.
  // The first soft-max is never written: a statistics pass leaves (max, sum) per query row, the mass pass
  // streams the queries past each key block (owner = keys of batch b, streamed = queries of batch (b+B) % 2B).
  // kNN launches: the tie counters were zeroed together at the start of the forward; a launch that does not replay its
  // tied rows itself (long rows) is listed in `deferred` and replayed by knn_ties() before the first consumer of the indices
  const vcr_vcrnet_io* io_ = nullptr;
  bool sdpa_split = false;                               // linear_mode 2
  float* pv_split = nullptr; long pv_split_floats = 0;   // vcr_sdpa_args.split_work of the attention-output launches
  int linear_variant = 0;                                // MFMA shape / k-slab forced by vcr_vcrnet_weights.linear_mfma / linear_bk
  int sdpa_variant = 0;                                  // vcr_vcrnet_weights.sdpa_variant
  vcr_knn_args deferred[2];                              // kNN launches whose tie replay is still owed (knn_ties)
  int n_deferred = 0;
  void knn(const char* nm, vcr_knn_args a, int which, bool defer = false) {
    if (rc) return;
    mark(nm);
    a.tie_zeroed = 1;
    if (defer && n_deferred < 2 && !vcr_knn_ties_inline(&a)) {          // (inline: the launch replays its own ties)
      a.tie_defer = 1;
      deferred[n_deferred++] = a;
    }
    ok(vcr_knn_f32(&a, stream));
  }
  // LPDNet's two independent searches as one launch (vcr_knn_pair_f32), their tie replays deferred to knn_ties()
  void knn_pair(const char* nm, vcr_knn_args a64, vcr_knn_args a3) {
    if (rc) return;
    mark(nm);
    a64.tie_zeroed = a3.tie_zeroed = 1;
    a64.tie_defer = a3.tie_defer = 1;
    n_deferred = 0;                                      // (a launch that replays its ties itself owes nothing)
    if (!vcr_knn_ties_inline(&a64)) deferred[n_deferred++] = a64;
    if (!vcr_knn_ties_inline(&a3)) deferred[n_deferred++] = a3;
    ok(vcr_knn_pair_f32(&a64, &a3, stream));
  }
  // one replay launch for every kNN deferred so far: to be called before the first consumer of any of their indices
  void knn_ties() {
    if (rc || n_deferred == 0) return;
    mark("knn:ties");
    ok(vcr_knn_ties_f32(&deferred[0], n_deferred == 2 ? &deferred[1] : nullptr, stream));
    n_deferred = 0;
  }
  // device-to-device copy of a forced / reported selection (tiny; stays on the stream)
  void copy_idx(const char* nm, int32_t* dst, const int32_t* src, size_t n) {
    if (rc) return;
    mark(nm);
    ok(vcr_copy_d2d(dst, src, n * sizeof(int32_t), stream));
  }

  void cross_attention(const vcr_vcrnet_weights* W, const vcr_vcrnet_io* io, const Ws& w, int B, int N) {
    const int E = W->E, H = W->heads, nb = 2 * B, dk = E / H;
    const uint8_t* keep = nullptr;
    if (W->partial) {
      const int nkeep = (int)((double)N * W->overlap2);
      if (io->force_keys) {
        // teacher forcing: the caller's kept keys replace the mass ranking (both passes that only feed it are skipped)
        if (sdpa_split && W->F < 2 * E) { ok(VCR_EUNSUPPORTED); return; }   // (the exact-split kernel needs the dense copy)
        copy_idx("select:dec.cross.keys.forced", w.xorder, io->force_keys, (size_t)nb * nkeep);
      } else if (w.xscore) {
        // statistics pass that also keeps the scaled scores; the key mass is then one HBM-bound pass over them
        const int ldS = (N + 31) & ~31;
        sdpa("sdpa:dec.cross.stats", w.qc, E, w.kvc, 2 * E, nullptr, 0, nullptr, 0, nb, H, N, N, B, nullptr, w.rowstat,
             w.xscore, ldS, 1, 0, 0, w.xsplit, (long)VCR_SDPA_MAX_SPLIT * nb * H * N * 2);
        if (rc == 0) {
          mark("scoremass:dec.cross.keymass");
          vcr_keymass_args a{w.xscore, ldS, nb, H, N, N, w.rowstat, B, w.keymass};
          ok(vcr_keymass_f32(&a, stream));
        }
      } else {                                           // score matrix too large to keep: recompute it per head
        sdpa("sdpa:dec.cross.stats", w.qc, E, w.kvc, 2 * E, nullptr, 0, nullptr, 0, nb, H, N, N, B, nullptr, w.rowstat, nullptr, 0,
             1, 0, 0, w.xsplit, (long)VCR_SDPA_MAX_SPLIT * nb * H * N * 2);
        for (int h = 0; h < H; ++h) {
          vcr_pairscore_args a{};
          a.own = w.kvc + h * dk; a.ld_own = 2 * E; a.str = w.qc + h * dk; a.ld_str = E;
          a.nbatch = nb; a.n_own = N; a.n_str = N; a.E = dk; a.score = 1; a.scale = 1.0f / sqrtf((float)dk);
          a.str_batch_shift = B; a.op = 2; a.str_stat2 = w.rowstat + (size_t)h * N * 2;
          a.str_stat_batch_stride = (long)H * N * 2; a.mass = w.keymass; a.accumulate = h > 0;
          pairscore("pairscore:dec.cross.keymass", a);
        }
      }
      // the second soft-max runs unmasked over the nkeep kept keys only: the same set as masked_fill(-1e9) + softmax, 23 %
      // fewer score / PV MFMAs at overlap2 = 0.766 and no per-score mask lookups.  fp32: the attention kernel reads the
      // kept K|V rows THROUGH the list (vcr_sdpa_args.key_index; round 2 gathered them into a dense buffer first: 0.039 ms
      // and 230 MB per pass at configs[2]); the exact-split attention kernel still takes the dense copy (hid [2B*N, F] is
      // free until the FFN and holds it when F >= 2E), else the masked form
      if (!io->force_keys) rank("select:dec.cross.keys", w.keymass, 1, nb, N, nkeep, w.xorder, w.keep, 1);
      if (io->out_keys) copy_idx("select:dec.cross.keys.out", io->out_keys, w.xorder, (size_t)nb * nkeep);
      if (!sdpa_split && nkeep <= 16384) {               // (vcr_sdpa_f32 holds the index list in LDS: <= 16 384 kept keys;
        if (rc) return;                                  //  longer lists take the dense copy / the masked form below)
        mark("sdpa:dec.cross");
        vcr_sdpa_args a{w.qc, E, w.kvc, 2 * E, w.kvc + E, 2 * E, w.attx, E, nb, H, N, nkeep, 1.0f / sqrtf(128.f), B};
        a.key_index = w.xorder; a.nk_src = N;
        ok(vcr_sdpa_f32(&a, stream));
        return;
      }
      if (W->F >= 2 * E) {
        gather("select:gather.kv", w.kvc, 2 * E, N, w.xorder, nb, nkeep, 2 * E, w.hid);
        sdpa("sdpa:dec.cross", w.qc, E, w.hid, 2 * E, w.hid + E, 2 * E, w.attx, E, nb, H, N, nkeep, B);
        return;
      }
      keep = w.keep;
    }
    sdpa("sdpa:dec.cross", w.qc, E, w.kvc, 2 * E, w.kvc + E, 2 * E, w.attx, E, nb, H, N, N, B, keep);
  }

  // VcpTopK partial mode: selectCom (vcrnet_model.py:190-262) + getCopair (:264-332) + SVD.  The N x N score
  // matrix is written once, its two soft-maxes never: rankselect on the column / row probability sums gives the
  // overlap sets, one more STATS pass (with arg-max) on the reduced sets the hard correspondences.
  void partial_head(const vcr_vcrnet_weights* W, const vcr_vcrnet_io* io, const Ws& w, int B, int N) {
    const int E = W->E, M1 = B * N;
    const int K1 = overlap_k1(N, W->overlap2), K2 = overlap_k2(N, W->overlap2);
    const float *se = w.embf, *te = w.embf + (size_t)M1 * E, *ss = w.side4, *ts = w.side4 + (size_t)M1 * 4;
    auto stats = [&](const char* nm, const float* own, const float* str, const float* os, const float* sd, int no,
                     int ns, float* stat2, int32_t* amax, float* score_out, int ld_score) {
      vcr_pairscore_args a{};
      a.own = own; a.ld_own = E; a.str = str; a.ld_str = E; a.own_side4 = os; a.str_side4 = sd;
      a.nbatch = B; a.n_own = no; a.n_str = ns; a.E = E; a.score = 0; a.scale = 1.f; a.op = 1;
      a.stat2 = stat2; a.argmax = amax; a.score_out = score_out; a.ld_score = ld_score;
      a.split_work = amax ? nullptr : w.rsplit;
      a.split_work_floats = amax ? 0 : (long)VCR_PAIRSCORE_MAX_SPLIT * B * N * 2;     // (sized in plan(): B1 * N * 2 per split)
      pairscore(nm, a);
    };
    // score_ij = (-|s_i|^2 + 2 s_i.t_j) - |t_j|^2 (:211-216), computed and stored once with the row soft-max
    // statistics (dim=2); the column statistics (dim=1) and both probability masses come from two HBM-bound
    // passes over it
    const int ldS = (N + 31) & ~31;
    const bool forced_sets = io->force_sel_src && io->force_sel_tgt;
    if (!forced_sets) {
      stats("pairscore:head.scores", se, te, ss, ts, N, N, w.rstat, nullptr, w.score, ldS);
      if (rc == 0) {
        mark("scoremass:head");
        vcr_scoremass_args a{w.score, ldS, B, N, N, w.rstat, w.cstat, w.colsum, w.rowsum};           // :222, :244
        ok(vcr_scoremass_f32(&a, stream));
      }
    }
    if (forced_sets) {
      copy_idx("select:head.tgt.forced", w.sel_t, io->force_sel_tgt, (size_t)B * K1);
      copy_idx("select:head.src.forced", w.sel_s, io->force_sel_src, (size_t)B * K1);
    } else {
      // :245 (sources by row mass) and :223 (targets by column mass) as one launch over the [2B, N] block rowsum | colsum
      rank("select:head.src+tgt", w.rowsum, 1, 2 * B, N, K1, w.sel_s, nullptr, 1);
    }
    if (io->out_sel_tgt) copy_idx("select:head.tgt.out", io->out_sel_tgt, w.sel_t, (size_t)B * K1);
    if (io->out_sel_src) copy_idx("select:head.src.out", io->out_sel_src, w.sel_s, (size_t)B * K1);
    gather("select:gather.emb", se, E, N, w.sel_s, 2 * B, K1, E, w.so_e);    // :251-260 (source) and :235-238 (target), one launch
    gather("select:gather.xyz", ss, 4, N, w.sel_s, 2 * B, K1, 4, w.so_s);
    // getCopair on the overlap sets: peak soft-max probability = 1/l and its arg-max target (:295-298)
    if (!(io->force_argmax && io->force_pairs))
      stats("pairscore:head.copair", w.so_e, w.to_e, w.so_s, w.to_s, K1, K1, w.peak, w.amax, nullptr, 0);
    if (io->force_argmax) copy_idx("select:head.argmax.forced", w.amax, io->force_argmax, (size_t)B * K1);
    if (io->force_pairs) copy_idx("select:head.pairs.forced", w.pick, io->force_pairs, (size_t)B * K2);
    else rank("select:head.pairs", w.peak + 1, 2, B, K1, K2, w.pick, nullptr, 0);   // largest peak prob == smallest l (:312)
    if (io->out_argmax) copy_idx("select:head.argmax.out", io->out_argmax, w.amax, (size_t)B * K1);
    if (io->out_pairs) copy_idx("select:head.pairs.out", io->out_pairs, w.pick, (size_t)B * K2);
    gather("select:gather.srcK", w.so_s, 4, K1, w.pick, B, K2, 4, io->src4);                         // :328-330
    gather("select:gather.corrK", w.to_s, 4, K1, w.pick, B, K2, 4, io->corr4, w.amax, K1);           // :325 (weights == 1)
    if (rc) return;
    mark("rigid_svd:ab");
    vcr_rigid_svd_args a{io->src4, 4, io->corr4, 4, B, K2, io->R_ab, io->t_ab, io->R_ba, io->t_ba, nullptr};
    ok(vcr_rigid_svd_f32(&a, stream));
  }
};

// pass: 0 = a forward on its own.  Target reuse inside vcrnetIter (the target cloud does not change between the passes of
// vcrnet_model.py:21-43, so everything computed from it ALONE is loop-invariant: its LPDNet embedding, the encoder on its rows,
// the decoder's self-attention sublayer and cross-attention query on its rows, the K | V projection of its encoder memory):
// 1 = the first pass, which keeps the four buffers those rows are read from behind the workspace; 2 = a later pass, whose
// launches in front of the cross-attention run on the SOURCE rows only (names end in "@src") and leave the target rows alone.
// Every launch computes what the full-row launch would (linears pin its MFMA shape): the loop's results do not change by a bit.
int forward_impl(const vcr_vcrnet_weights* W, const vcr_vcrnet_io* io, void* workspace, size_t ws_bytes,
                 vcr_stream_t stream, vcr_trace* tr, bool last = true, int pass = 0) {
  if (!W || !io || !workspace || !io->src_cf || !io->tgt_cf || !io->corr4 || !io->src4 || !io->R_ab || !io->t_ab)
    return VCR_EINVAL;
  const int B = io->B, N = io->N, E = W->E, F = W->F, k = W->k;
  if (B <= 0 || N <= 0 || E != 512 || W->heads * 128 != E || k <= 0 || k + 1 > N) return VCR_EINVAL;
  if (k > 62 || N > 65535) return VCR_EUNSUPPORTED;     // k: vcr_knn_f32's limit; N: the forward's own (the kNN kernels alone take N <= 131 072)
  if (W->emb_kind < 0 || W->emb_kind > 2) return VCR_EINVAL;
  if (W->emb_kind == 2 && !(W->c1_w && W->c1_b && W->c2_w && W->c2_b && W->pointnet.c3_w && W->pointnet.c3_b && W->pointnet.c4_w &&
                            W->pointnet.c4_b && W->pointnet.c5_w && W->pointnet.c5_b))
    return VCR_EINVAL;
  if (W->has_pointer == 1 && (F % 128)) return VCR_EINVAL;
  if (W->has_pointer == 1 &&
      !(W->fold_enc_qkv.w && W->fold_enc_ffn1.w && W->fold_dec_qkv.w && W->fold_dec_cross_q.w &&
        W->fold_dec_cross_kv.w && W->fold_dec_ffn1.w))
    return VCR_EINVAL;
  if (((uintptr_t)workspace) & 255) return VCR_EINVAL;
  if (W->head_mode < 0 || W->head_mode > 2 || W->linear_mode < 0 || W->linear_mode > 2) return VCR_EINVAL;
  for (int ms : {W->linear_mfma, W->linear_bk})
    if (ms != 0 && ms != 16 && ms != 32) return VCR_EINVAL;
  if ((W->linear_bm != 0 && W->linear_bm != 96 && W->linear_bm != 128) || (W->linear_bm == 96 && W->linear_mfma == 32)) return VCR_EINVAL;
  if (W->knn_waves != 0 && W->knn_waves != 1 && W->knn_waves != 8) return VCR_EINVAL;
  if (W->partial) {                                      // key pruning in the decoder (+ hard pairs for the topK head)
    if (W->has_pointer != 1 || (W->cycle && W->head_mode == 0)) return VCR_EUNSUPPORTED;
    if (!(W->overlap2 > 0.0 && W->overlap2 <= 1.0) || (int)((double)N * W->overlap2) < 1) return VCR_EINVAL;
    if (W->head_mode == 0 && overlap_k2(N, W->overlap2) < 3) return VCR_EINVAL;
  }
  if (!W->partial && (io->force_keys || io->force_sel_src || io->force_sel_tgt || io->force_argmax || io->force_pairs))
    return VCR_EINVAL;                                   // there is nothing discrete to force in whole mode
  if ((io->force_sel_src != nullptr) != (io->force_sel_tgt != nullptr)) return VCR_EINVAL;
  Ws w = carve(workspace, B, N, k, E, F, W->heads, W->partial, W->overlap2, W->emb_kind, W->xscore_limit_mb, merged_encdec(W), W->workspace_flat);
  if (ws_bytes < w.bytes) return VCR_EWORKSPACE;
  const int M1 = B * N, M2 = 2 * M1;
  if (pass != 0) {
    if (ws_bytes < w.bytes + tgt_cache_floats(B, N, E) * sizeof(float)) return VCR_EINVAL;
    float* c = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + w.bytes);
    w.emb = c;       c += (size_t)M2 * E;                // (the planned copies of these four stay unused in such a loop)
    w.d1 = c;        c += (size_t)M2 * E;
    w.qc = c;        c += (size_t)M2 * E;
    w.kvc = c;
  }
  const bool half = pass == 2;                           // launches in front of the cross-attention: source rows only
  const int Bq = half ? B : 2 * B, Mq = half ? M1 : M2;
#define NM(site) (half ? site "@src" : site)
  Runner R{(hipStream_t)stream, tr};
  R.io_ = io;
  R.shape_rows = half ? M2 : 0;
  R.plan_nbatch = half ? 2 * B : 0;
  // both tie counters (and the first block).  A kernel, not hipMemsetAsync: the forward then records into a HIP graph of
  // kernel nodes only
  hipLaunchKernelGGL(zero_i32_kernel, dim3((unsigned)((M2 + 2 + 255) / 256)), dim3(256), 0, R.stream, w.ties, (long)(M2 + 2));
  R.ok(VCR_LAUNCH_RC());
#define SP(site) (W->linear_mode != 0 ? W->split.site : nullptr)
  R.sdpa_split = W->linear_mode == 2;
  R.sdpa_variant = W->sdpa_variant;
  R.pv_split = w.asplit; R.pv_split_floats = w.asplit_floats;
  R.linear_variant = (W->linear_mfma == 16 ? 16 : W->linear_mfma == 32 ? 1024 : 0) | (W->linear_bk == 16 ? 64 : W->linear_bk == 32 ? 8 : 0) |
                     (W->linear_bm == 96 ? 2048 : W->linear_bm == 128 ? 4096 : 0);

  const float* stats_for_ln = W->has_pointer == 1 ? w.st_emb : nullptr;
  if (W->emb_kind == 1) {
    // ---- emb_nn = DGCNN on both clouds (vcrnet_model.py:104-123): one Cartesian kNN, conv1 via the neighbour/centre
    // split (per-point P/Q + gather), conv2..conv4 as N*k-row GEMMs, max over the k edges after each, conv5 on the
    // 512-wide concatenation.  BatchNorm (eval mode) is folded into the weights by the host.  (linear_mode 1 / 2: the
    // embedding's own GEMMs stay fp32 MFMA -- the chain kernel has no split variant -- the Transformer takes the mode.)
    for (int c = 0; c < (half ? 1 : 2) && R.rc == 0; ++c) {   // rows (x, y, z, |p|^2) and conv1's per-point (P | Q), one pass
      R.mark(c ? "pointwise:tgt" : "pointwise:src");
      R.ok(vcr_rows4_pq_f32(c ? io->tgt_cf : io->src_cf, w.xyz4 + (size_t)c * M1 * 4, B, N, W->dgcnn.c1_wpq, 32,
                            W->dgcnn.c1_bpq, 128, w.pq1 + (size_t)c * M1 * 128, 128, R.stream));
    }
    {
      vcr_knn_args a3{(uint32_t)sizeof(vcr_knn_args), w.xyz4, 4, nullptr, Bq, N, 4, k, w.idx3, w.ties + 1 + M2, M2};
      a3.tie_work = w.tie_work; a3.tie_work_bytes = w.tie_work_each;
      R.knn(NM("knn:xyz"), a3, 1);
    }
    const int Mk = Mq * k;
    // Every x.max(dim=-1) of vcrnet_model.py:109-118 rides on the kernel that produces the per-edge rows: the edge-row
    // builder writes x1 (and the zero base of x2..x4), conv2..conv4 fold the max over each point's k rows into their
    // epilogues (integer atomic max on post-ReLU values), and conv4's 256-wide per-edge activations are never written.
    auto conv_max = [&](const char* nm, const float* x, int K, const float* wt, const float* bias, float* y, int Nout,
                        int col) {
      if (R.rc) return;
      R.mark(nm);
      vcr_linear_args a{x, K, wt, bias, nullptr, 0, y, Nout, Mk, Nout, K, 1};
      a.segmax_out = w.cat + col; a.ld_segmax = 512; a.seg_k = k;
      a.variant = R.linear_variant;                      // (vcr_vcrnet_weights.linear_mfma / linear_bk / linear_bm reach every fp32 linear)
      R.pin_shape(a, nullptr, M2 * k);                   // (a source-only pass: the MFMA shape of the launch over both clouds' edges)
      R.ok(vcr_linear_f32(&a, R.stream));
    };
    if (k == 20 || k == 40) {
      // the path's k: the whole chain in one kernel, the per-edge activations stay in LDS (edgechain.hip)
      if (R.rc == 0) {
        R.mark(NM("edgeconv:dg_chain"));
        vcr_edgechain_args a{w.pq1, 128, w.idx3, k, Mq, N, W->dgcnn.c2_w, W->dgcnn.c2_b, W->dgcnn.c3_w, W->dgcnn.c3_b,
                             W->dgcnn.c4_w, W->dgcnn.c4_b, w.cat, 512};
        R.ok(vcr_edgechain_f32(&a, R.stream));
      }
    } else {
      if (R.rc == 0) {
        R.mark(NM("gathermax:dg_c1"));
        vcr_edgerows_args a{w.pq1, 128, 64, w.idx3, k, Mq, N, w.eh1, 64, w.cat, 512, 512};
        R.ok(vcr_edgerows_f32(&a, R.stream));
      }
      conv_max(NM("linear:dg_c2"), w.eh1, 64, W->dgcnn.c2_w, W->dgcnn.c2_b, w.eh2, 64, 64);
      conv_max(NM("linear:dg_c3"), w.eh2, 64, W->dgcnn.c3_w, W->dgcnn.c3_b, w.eh3, 128, 128);
      conv_max(NM("linear:dg_c4"), w.eh3, 128, W->dgcnn.c4_w, W->dgcnn.c4_b, nullptr, 256, 256);
    }
    R.linear(NM("linear:conv3"), w.cat, 512, W->dgcnn.c5_w, nullptr, W->dgcnn.c5_b, w.emb, E, Mq, E, 512, 1, nullptr, 0, nullptr,
             nullptr, const_cast<float*>(stats_for_ln));
  } else if (W->emb_kind == 2) {
    // ---- emb_nn = PointNet on both clouds (vcrnet_model.py:81-87): five pointwise convs + BatchNorm (eval mode, folded
    // into weight and bias by the host) + ReLU, no graph.  conv1 / conv2 have the shape of LPDNet's stem: same kernel.
    if (R.rc == 0) {
      R.mark(NM("pointwise:src+tgt"));
      vcr_pointwise_args a{io->src_cf, B, N, W->c1_w, W->c1_b, W->c2_w, W->c2_b, w.xyz4, w.feat64, w.sq64, half ? nullptr : io->tgt_cf,
                           half ? 0 : B};
      R.ok(vcr_pointwise_f32(&a, R.stream));
    }
    R.linear(NM("linear:pn_c3"), w.feat64, 64, W->pointnet.c3_w, nullptr, W->pointnet.c3_b, w.pq1, 64, Mq, 64, 64, 1);
    R.linear(NM("linear:pn_c4"), w.pq1, 64, W->pointnet.c4_w, nullptr, W->pointnet.c4_b, w.cat, 128, Mq, 128, 64, 1);
    R.linear(NM("linear:pn_c5"), w.cat, 128, W->pointnet.c5_w, nullptr, W->pointnet.c5_b, w.emb, E, Mq, E, 128, 1, nullptr, 0, nullptr,
             nullptr, const_cast<float*>(stats_for_ln));
  } else {
  // ---- emb_nn = LPDNet on both clouds (lpdnet_model.py:103-137)
  // fp32 mode: the first EdgeConv's per-point projection P | Q (K = 64) rides on the stem launch (conv2 and the projection
  // on the matrix pipe, pointwise.hip); the split modes keep the separate launch on their pre-split weight.  (Round 6 tried the
  // fused stem in the split modes too -- one launch less, 0.4 % of that line --: the fp32 chain is a little LESS accurate than
  // the exact split, and the accuracy ledger's `trained` k = 40 fixture left its bound (R error vs the float64 twin 4.3e-6 ->
  // 5.9e-6 against 1.5 x the reference's + 3e-6 = 4.9e-6).  Reverted.)
  const bool pq_fused = W->linear_mode == 0;
  if (R.rc == 0) {                                       // both clouds in one launch: rows 0..M1-1 = src, then tgt
    R.mark(pq_fused ? NM("pointwise:src+tgt+dg1_pq") : NM("pointwise:src+tgt"));
    vcr_pointwise_args a{io->src_cf, B, N, W->c1_w, W->c1_b, W->c2_w, W->c2_b, w.xyz4, w.feat64, w.sq64, half ? nullptr : io->tgt_cf,
                         half ? 0 : B};
    if (pq_fused) { a.pq_w = W->dg1_wpq; a.pq_b = W->dg1_bpq; a.pq = w.pq1; a.ldpq = 256; }
    a.feat64t = W->E >= 64 ? w.emb : nullptr;                                   // operand layout of the 16-query kNN waves (w.emb is free until conv3)
    R.ok(vcr_pointwise_f32(&a, R.stream));
  }
  // The feature-space and the Cartesian kNN (lpdnet_model.py:113,129) are independent: one launch for both, and one
  // tie replay for both right before the first consumer of the indices.
  const int32_t* rank_perm = nullptr;                    // the clouds' Morton ranking, when the kNN took the ordered search
  {
    vcr_knn_args a64{(uint32_t)sizeof(vcr_knn_args), w.feat64, 64, w.sq64, Bq, N, 64, k, w.idx1, w.ties, M2, W->knn_waves};
    vcr_knn_args a3{(uint32_t)sizeof(vcr_knn_args), w.xyz4, 4, nullptr, Bq, N, 4, k, w.idx3, w.ties + 1 + M2, M2};
    a64.tie_work = w.tie_work; a3.tie_work = w.tie_work ? w.tie_work + w.tie_work_each : nullptr;
    a64.tie_work_bytes = a3.tie_work_bytes = w.tie_work_each;
    a64.xt = W->E >= 64 ? w.emb : nullptr;
    // The ORDERED search (vcr_knn_args.perm) for the larger clouds: ranking the points along a Morton curve lets a 16-query
    // wave skip the tiles whose balls cannot hold a neighbour -- measured (profiles/rounds4-5/r5o_knn_ordered.txt) 416 -> 295 + 45 us
    // (ranking) at 32 x 2048, 2970 -> 1850 + 93 at 64 x 4096, k = 40; at 1024 points the plain scan is faster (120 vs 143 + 30).
    // Its arrays live in the unused part of w.emb (free until conv3; feat64t is its first M2 x 64 floats).
    if (R.rc == 0 && W->E >= 256 && N >= KNN_ORDERED_MIN_N && N <= 8192 && (k == 20 || k == 40) && W->knn_waves == 0 &&
        (long)Bq * ((N + 15) / 16) >= 1024) {            // (fewer query groups: vcr_knn_pair_f32 takes its small-grid kernels)
      const size_t m = (size_t)M2, mt = (size_t)2 * B * ((N + 15) / 16);
      float* base = w.emb + m * 64;
      float* feat_p = base;                 base += m * 64;
      float* xyz4_p = base;                 base += m * 4;
      float* cen64 = base;                  base += mt * 64;
      float* cen4 = base;                   base += mt * 4;
      float* sq_p = base;                   base += m;
      int32_t* perm = reinterpret_cast<int32_t*>(base);   base += m;
      float* c64_sq = base;  base += mt;  float* c64_rad = base;  base += mt;  float* c64_max = base;  base += mt;
      float* c4_rad = base;  base += mt;  float* c4_max = base;  base += mt;
      int32_t* ord_ok = reinterpret_cast<int32_t*>(base);                  // the ranking's per-cloud verdict on the feature tiles
      R.mark(NM("knn:rank"));
      vcr_knn_order_args o{w.xyz4, w.emb, 64, w.sq64, Bq, N, perm, xyz4_p, cen4, c4_rad, c4_max, feat_p, sq_p, cen64, c64_sq,
                           c64_rad, c64_max, ord_ok, nullptr, 0.f};
      R.ok(vcr_knn_order_f32(&o, R.stream));
      a64.perm = a3.perm = perm;
      a64.ord_ok = ord_ok;
      rank_perm = perm;                                  // (lives in w.emb until conv3 writes the embeddings: gathermax runs before)
      a64.xp = feat_p; a64.sqp = sq_p; a64.cen = cen64; a64.cen_sq = c64_sq; a64.cen_rad = c64_rad; a64.cen_sqmax = c64_max;
      a3.xp = xyz4_p; a3.cen = cen4; a3.cen_rad = c4_rad; a3.cen_sqmax = c4_max;
    }
    R.knn_pair(NM("knn:feat64+xyz"), a64, a3);
  }
  if (!pq_fused) R.linear(NM("linear:dg1_pq"), w.feat64, 64, W->dg1_wpq, SP(dg1_pq), W->dg1_bpq, w.pq1, 256, Mq, 256, 64, 0);
  R.knn_ties();                                          // both tie replays in one launch (one latency instead of two)
  if (R.rc == 0) {
    R.mark(NM("edgeconv:dg1_dg2"));
    vcr_edgeconv_args a{w.pq1, 256, w.idx1, k, Mq, N, W->dg2_w, W->dg2_b, w.cat, 512, w.cat + 128, 512};
    // (linear_mode 1 / 2: convDG2 as exact bf16 splits at the path's k; other k keep the fp32 kernel)
    R.ok(W->linear_mode != 0 && (k == 20 || k == 40) ? vcr_edgeconv_bf16x3_f32(&a, R.stream) : vcr_edgeconv_f32(&a, R.stream));
  }
  R.linear(NM("linear:sn1_pq"), w.cat + 128, 512, W->sn1_wpq, SP(sn1_pq), W->sn1_bpq, w.pq3, 512, Mq, 512, 128, 0);
  if (R.rc == 0) {
    R.mark(NM("gathermax:sn1"));
    vcr_gathermax_args a{w.pq3, 512, 256, w.idx3, k, Mq, N, w.cat + 256, 512};
    a.order = rank_perm;                                 // (clouds that were ranked for the kNN: the L2 form walks them in rank order)
    R.ok(vcr_gathermax_f32(&a, R.stream));
  }
  R.linear(NM("linear:conv3"), w.cat, 512, W->c3_w, SP(c3), W->c3_b, w.emb, E, Mq, E, 512, 1, nullptr, 0, nullptr, nullptr,
           W->has_pointer == 1 ? w.st_emb : nullptr);

  }
  if (W->has_pointer != 1) { R.shape_rows = 0; R.plan_nbatch = 0; }     // (no Transformer: nothing else runs on half the rows)

  // ---- pointer (transformer.py:264-272) + residual (vcrnet_model.py:504-505)
  const float* head_emb = w.embf;
  if (W->has_pointer == 1) {
    // LayerNorm folded into the consuming linears: each producer of a
    // residual stream writes per-row (sum, sum^2) partials from its epilogue, each consumer runs the plain GEMM on
    // the folded weight and applies (mean, 1/(std+eps)) in its epilogue -- six LayerNorm launches and their
    // 2 x 67 MB round trips are gone.
    const int H = W->heads;
    const bool merged = merged_encdec(W) != 0;
    const float* att_dec = w.att;                        // the decoder's self-attention output
    if (merged) {
      // both first sublayers read the embedding rows with the same row statistics: one [M, 6E] projection, and the two
      // independent self-attentions as one grouped launch (group 0 = encoder, 1 = decoder)
      R.linear(NM("linear:encdec.qkv"), w.emb, E, W->fold_encdec_qkv.w, SP(encdec_qkv), W->fold_encdec_qkv.bias, w.qkv, 6 * E, Mq, 6 * E, E, 0,
               nullptr, 0, w.st_emb, W->fold_encdec_qkv.colsum);
      R.sdpa(NM("sdpa:encdec.self"), w.qkv, 6 * E, w.qkv + E, 6 * E, w.qkv + 2 * E, 6 * E, w.att, E, Bq, H, N, N, 0, nullptr, nullptr,
             nullptr, 0, 2, 3 * E, (long)M2 * E);
      att_dec = w.att + (size_t)M2 * E;
    } else {
    R.linear(NM("linear:enc.qkv"), w.emb, E, W->fold_enc_qkv.w, SP(enc_qkv), W->fold_enc_qkv.bias, w.qkv, 3 * E, Mq, 3 * E, E, 0,
             nullptr, 0, w.st_emb, W->fold_enc_qkv.colsum);
    R.sdpa(NM("sdpa:enc.self"), w.qkv, 3 * E, w.qkv + E, 3 * E, w.qkv + 2 * E, 3 * E, w.att, E, Bq, H, N, N, 0);
    }
    if (merged && W->linear_mode != 0) {
      // the split-arithmetic linears have no paired launcher: the same sequence, one launch each
      R.linear(NM("linear:enc.wo"), w.att, E, W->enc_self.wo, SP(enc_wo), W->enc_self.bo, w.e1, E, Mq, E, E, 0, w.emb, E,
               nullptr, nullptr, w.st_e1);
      R.linear(NM("linear:dec.self.wo"), att_dec, E, W->dec_self.wo, SP(dec_self_wo), W->dec_self.bo, w.d1, E, Mq, E, E, 0, w.emb, E,
               nullptr, nullptr, w.st_d1);
      R.linear(NM("linear:enc.ffn1"), w.e1, E, W->fold_enc_ffn1.w, SP(enc_ffn1), W->fold_enc_ffn1.bias, w.hid, F, Mq, F, E, 1, nullptr, 0,
               w.st_e1, W->fold_enc_ffn1.colsum);
      R.linear(NM("linear:dec.cross.q"), w.d1, E, W->fold_dec_cross_q.w, SP(dec_cross_q), W->fold_dec_cross_q.bias, w.qc, E, Mq, E, E, 0, nullptr, 0,
               w.st_d1, W->fold_dec_cross_q.colsum);
      R.linear(NM("linear:enc.ffn2"), w.hid, F, W->enc_ffn.w2, SP(enc_ffn2), W->enc_ffn.b2, w.e2, E, Mq, E, F, 0, w.e1, E,
               nullptr, nullptr, w.st_e2);
    } else if (merged) {
      // independent launches of one kernel configuration run as pairs: the two output projections (inputs = the two
      // attention outputs, residual = the embedding), then the encoder's FFN-in beside the decoder's cross-attention query
      R.linear2(NM("linear:enc.wo+dec.self.wo"),
                R.linear_args(w.att, E, W->enc_self.wo, W->enc_self.bo, w.e1, E, Mq, E, E, 0, w.emb, E, nullptr, nullptr, w.st_e1),
                R.linear_args(att_dec, E, W->dec_self.wo, W->dec_self.bo, w.d1, E, Mq, E, E, 0, w.emb, E, nullptr, nullptr, w.st_d1));
      R.linear2(NM("linear:enc.ffn1+dec.cross.q"),
                R.linear_args(w.e1, E, W->fold_enc_ffn1.w, W->fold_enc_ffn1.bias, w.hid, F, Mq, F, E, 1, nullptr, 0, w.st_e1,
                              W->fold_enc_ffn1.colsum),
                R.linear_args(w.d1, E, W->fold_dec_cross_q.w, W->fold_dec_cross_q.bias, w.qc, E, Mq, E, E, 0, nullptr, 0, w.st_d1,
                              W->fold_dec_cross_q.colsum));
      R.linear(NM("linear:enc.ffn2"), w.hid, F, W->enc_ffn.w2, nullptr, W->enc_ffn.b2, w.e2, E, Mq, E, F, 0, w.e1, E,
               nullptr, nullptr, w.st_e2);
    } else {
    R.linear(NM("linear:enc.wo"), w.att, E, W->enc_self.wo, SP(enc_wo), W->enc_self.bo, w.e1, E, Mq, E, E, 0, w.emb, E,
             nullptr, nullptr, w.st_e1);
    R.linear(NM("linear:enc.ffn1"), w.e1, E, W->fold_enc_ffn1.w, SP(enc_ffn1), W->fold_enc_ffn1.bias, w.hid, F, Mq, F, E, 1, nullptr, 0,
             w.st_e1, W->fold_enc_ffn1.colsum);
    R.linear(NM("linear:enc.ffn2"), w.hid, F, W->enc_ffn.w2, SP(enc_ffn2), W->enc_ffn.b2, w.e2, E, Mq, E, F, 0, w.e1, E,
             nullptr, nullptr, w.st_e2);
    // decoder; batch b attends to the encoder memory (= enc.norm(e2), applied inside the K/V projection) of
    // batch (b + B) mod 2B
    R.linear(NM("linear:dec.qkv"), w.emb, E, W->fold_dec_qkv.w, SP(dec_qkv), W->fold_dec_qkv.bias, w.qkv, 3 * E, Mq, 3 * E, E, 0,
             nullptr, 0, w.st_emb, W->fold_dec_qkv.colsum);
    R.sdpa(NM("sdpa:dec.self"), w.qkv, 3 * E, w.qkv + E, 3 * E, w.qkv + 2 * E, 3 * E, w.att, E, Bq, H, N, N, 0);
    R.linear(NM("linear:dec.self.wo"), att_dec, E, W->dec_self.wo, SP(dec_self_wo), W->dec_self.bo, w.d1, E, Mq, E, E, 0, w.emb, E,
             nullptr, nullptr, w.st_d1);
    R.linear(NM("linear:dec.cross.q"), w.d1, E, W->fold_dec_cross_q.w, SP(dec_cross_q), W->fold_dec_cross_q.bias, w.qc, E, Mq, E, E, 0, nullptr, 0,
             w.st_d1, W->fold_dec_cross_q.colsum);
    }
    // batch b attends to the encoder memory (= enc.norm(e2), applied inside the K/V projection) of batch (b + B) mod 2B
    R.linear(NM("linear:dec.cross.kv"), w.e2, E, W->fold_dec_cross_kv.w, SP(dec_cross_kv), W->fold_dec_cross_kv.bias, w.kvc, 2 * E, Mq, 2 * E, E, 0,
             nullptr, 0, w.st_e2, W->fold_dec_cross_kv.colsum);
    R.shape_rows = 0; R.plan_nbatch = 0;                 // from here on every launch has its own, full row count (the VcpAtt head's
                                                         // one-cloud linears included: nothing stands for a larger launch any more)
    R.cross_attention(W, io, w, B, N);
    R.linear("linear:dec.cross.wo", w.attx, E, W->dec_cross.wo, SP(dec_cross_wo), W->dec_cross.bo, w.d2, E, M2, E, E, 0, w.d1, E,
             nullptr, nullptr, w.st_d2);
    R.linear("linear:dec.ffn1", w.d2, E, W->fold_dec_ffn1.w, SP(dec_ffn1), W->fold_dec_ffn1.bias, w.hid, F, M2, F, E, 1, nullptr, 0,
             w.st_d2, W->fold_dec_ffn1.colsum);
    R.linear("linear:dec.ffn2", w.hid, F, W->dec_ffn.w2, SP(dec_ffn2), W->dec_ffn.b2, w.d3, E, M2, E, F, 0, w.d2, E);
    R.norm("layernorm:dec.norm+res", w.d3, W->dec_norm, w.embf, M2, E, w.emb, w.xyz4, w.side4);
  } else if (R.rc == 0) {
    R.mark("layernorm:rowside");
    vcr_rowside_args a{w.emb, E, M2, E, W->has_pointer == 2 ? 2.f : 1.f, w.embf, E, w.xyz4, w.side4};
    R.ok(vcr_rowside_f32(&a, R.stream));
  }

  // ---- head + SVD
  const bool hard_pairs = W->partial && W->head_mode == 0;       // VcpTopK in partial mode: selectCom + getCopair
  const float* side = w.side4;
  if (W->head_mode == 2) {
    // VcpAtt (vcrnet_model.py:444-449): one Linear per cloud on the final embeddings, then the topK whole-mode scoring
    // on the projected embeddings (their |.|^2 recomputed); d3 is free once the final LayerNorm has consumed it
    if (!W->att_w0 || !W->att_w1) return VCR_EINVAL;
    R.linear("linear:head.att.src", w.embf, E, W->att_w0, nullptr, W->att_b0, w.d3, E, M1, E, E, 0);
    R.linear("linear:head.att.tgt", w.embf + (size_t)M1 * E, E, W->att_w1, nullptr, W->att_b1, w.d3 + (size_t)M1 * E, E,
             M1, E, E, 0);
    if (R.rc == 0) {
      R.mark("layernorm:rowside.att");
      vcr_rowside_args a{w.d3, E, M2, E, 1.f, nullptr, E, w.xyz4, w.side4};
      R.ok(vcr_rowside_f32(&a, R.stream));
    }
    head_emb = w.d3;
  }
  auto soft_head = [&](const char* nm, size_t q0, size_t k0, float* corr) {   // rows q0.. are the queries, k0.. the keys
    if (R.rc) return;
    R.mark(nm);
    vcr_softcorr_args a{head_emb + q0 * E, E, head_emb + k0 * E, E, side + q0 * 4, side + k0 * 4,
                        corr, B, N, N, E, W->head_mode == 1 ? 1 : 0, 1.0f / sqrtf((float)E), w.csplit,
                        (long)VCR_PAIRSCORE_MAX_SPLIT * B * N * 8};
    R.ok(vcr_softcorr_f32(&a, R.stream));
  };
  if (hard_pairs) {
    R.partial_head(W, io, w, B, N);
  } else {
    soft_head("softcorr:head", 0, (size_t)M1, io->corr4);
    if (R.rc == 0) {
      R.mark("rigid_svd:ab");
      R.ok(vcr_copy_d2d(io->src4, w.xyz4, (size_t)M1 * 4 * sizeof(float), R.stream));
      vcr_rigid_svd_args a{w.xyz4, 4, io->corr4, 4, B, N, io->R_ab, io->t_ab, io->R_ba, io->t_ba, nullptr};
      R.ok(vcr_rigid_svd_f32(&a, R.stream));
    }
    if (W->cycle) {
      // cycle consistency (vcrnet_model.py:511-513): a second head + solve with the roles swapped gives (R_ba, t_ba)
      // instead of the inverse of (R_ab, t_ab)
      if (!io->R_ba || !io->t_ba) return VCR_EINVAL;
      if (W->head_mode == 2) {
        // VcpAtt with the roles swapped, head(tgt_emb, src_emb, tgt, src) (vcrnet_model.py:511-513 -> :444-445):
        // linears_emb[0] now projects the TARGET embeddings (the queries) and linears_emb[1] the SOURCE embeddings
        // (the keys) -- not the ab pass's projections with the roles exchanged.  d2 is free after the decoder.
        R.linear("linear:head.att.ba.src", w.embf, E, W->att_w1, nullptr, W->att_b1, w.d2, E, M1, E, E, 0);
        R.linear("linear:head.att.ba.tgt", w.embf + (size_t)M1 * E, E, W->att_w0, nullptr, W->att_b0, w.d2 + (size_t)M1 * E,
                 E, M1, E, E, 0);
        if (R.rc == 0) {
          R.mark("layernorm:rowside.att.ba");
          vcr_rowside_args a{w.d2, E, M2, E, 1.f, nullptr, E, w.xyz4, w.side4};
          R.ok(vcr_rowside_f32(&a, R.stream));
        }
        head_emb = w.d2;
      }
      soft_head("softcorr:head.ba", (size_t)M1, 0, w.corr_ba);
      if (R.rc == 0) {
        R.mark("rigid_svd:ba");
        vcr_rigid_svd_args a{w.xyz4 + (size_t)M1 * 4, 4, w.corr_ba, 4, B, N, io->R_ba, io->t_ba, nullptr, nullptr, nullptr};
        R.ok(vcr_rigid_svd_f32(&a, R.stream));
      }
    }
  }
  if (R.rc == 0 && io->emb_out) R.ok(vcr_copy_d2d(io->emb_out, w.embf, (size_t)M2 * E * sizeof(float), R.stream));
  if (last) R.finish();
#undef SP
#undef NM
  return R.rc;
}

// One step of vcrnetIter's bookkeeping (vcrnet_model.py:32-38) for sample blockIdx.y:
//   out = R_i in + t_i              transform_point_cloud (util/util.py:91-96), skipped when out == NULL
//   R_f <- R_i R_f,  t_f <- R_i t_f + t_i,  R_ba = R_f^T,  t_ba = -R_ba t_f      (block x == 0, when compose == 1)
//   R_ba = R_i^T, t_ba = -R_ba t_i                                                (block x == 0, when compose == 2)
// Products are k-ascending fma chains like the reference's CPU matmul.
__global__ __launch_bounds__(256) void pose_step_kernel(const float* __restrict__ Ri, const float* __restrict__ ti,
                                                        const float* in_cf, float* out_cf, int N, int compose,
                                                        float* Rf, float* tf, float* Rba, float* tba) {
  const int b = blockIdx.y;
  float r[9], t[3];
  for (int i = 0; i < 9; ++i) r[i] = Ri[b * 9 + i];
  for (int i = 0; i < 3; ++i) t[i] = ti[b * 3 + i];
  if (out_cf) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n < N) {
      const float* p = in_cf + (size_t)b * 3 * N + n;
      const float x = p[0], y = p[N], z = p[2 * (size_t)N];
      float* o = out_cf + (size_t)b * 3 * N + n;
      for (int c = 0; c < 3; ++c) o[(size_t)c * N] = fmaf(r[c * 3 + 2], z, fmaf(r[c * 3 + 1], y, r[c * 3] * x)) + t[c];
    }
  }
  if (compose == 2 && blockIdx.x == 0 && threadIdx.x == 0) {          // (R_ba, t_ba) = inverse of (R_i, t_i) only
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) Rba[b * 9 + i * 3 + j] = r[j * 3 + i];
      tba[b * 3 + i] = -fmaf(r[6 + i], t[2], fmaf(r[3 + i], t[1], r[i] * t[0]));
    }
  } else if (compose && blockIdx.x == 0 && threadIdx.x == 0) {
    float f[9], g[3], nf[9], ng[3];
    for (int i = 0; i < 9; ++i) f[i] = Rf[b * 9 + i];
    for (int i = 0; i < 3; ++i) g[i] = tf[b * 3 + i];
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) nf[i * 3 + j] = fmaf(r[i * 3 + 2], f[6 + j], fmaf(r[i * 3 + 1], f[3 + j], r[i * 3] * f[j]));
      ng[i] = fmaf(r[i * 3 + 2], g[2], fmaf(r[i * 3 + 1], g[1], r[i * 3] * g[0])) + t[i];
    }
    for (int i = 0; i < 9; ++i) Rf[b * 9 + i] = nf[i];
    for (int i = 0; i < 3; ++i) tf[b * 3 + i] = ng[i];
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) Rba[b * 9 + i * 3 + j] = nf[j * 3 + i];
      tba[b * 3 + i] = -fmaf(nf[6 + i], ng[2], fmaf(nf[3 + i], ng[1], nf[i] * ng[0]));
    }
  }
}

}  // namespace

// the part of vcr_vcrnet_weights every caller must pass: up to the first optional extension (fold_encdec_qkv)
static int weights_take(const vcr_vcrnet_weights* user, vcr_vcrnet_weights* mine) {
  return vcr_take_args(user, mine, offsetof(vcr_vcrnet_weights, fold_encdec_qkv));
}

extern "C" size_t vcr_vcrnet_workspace_bytes(const vcr_vcrnet_weights* UW, int B, int N) {
  vcr_vcrnet_weights Wn;
  const vcr_vcrnet_weights* W = &Wn;
  if (weights_take(UW, &Wn) || B <= 0 || N <= 0) return 0;
  return carve(nullptr, B, N, W->k, W->E, W->F, W->heads, W->partial, W->overlap2, W->emb_kind, W->xscore_limit_mb, merged_encdec(W), W->workspace_flat).bytes;
}

// vcrnetIter with target reuse: more than one pass, not switched off (vcr_vcrnet_weights.iter_reuse == 1).  Every embedding
// (its stage runs on the source clouds only in the later passes) and every pointer (with the Transformer also the encoder, the
// decoder's first sublayer and the K | V projection)
static bool iter_reuse_applies(const vcr_vcrnet_weights* W, int iters) { return iters > 1 && W->iter_reuse != 1; }
extern "C" size_t vcr_vcrnet_iter_workspace_bytes(const vcr_vcrnet_weights* UW, int B, int N, int iters) {
  vcr_vcrnet_weights Wn;
  const vcr_vcrnet_weights* W = &Wn;
  if (weights_take(UW, &Wn) || B <= 0 || N <= 0 || iters < 1) return 0;
  const size_t base = carve(nullptr, B, N, W->k, W->E, W->F, W->heads, W->partial, W->overlap2, W->emb_kind, W->xscore_limit_mb, merged_encdec(W), W->workspace_flat).bytes;
  return base + (iter_reuse_applies(W, iters) ? tgt_cache_floats(B, N, W->E) * sizeof(float) : 0);
}

extern "C" int vcr_vcrnet_pairs(const vcr_vcrnet_weights* UW, int N) {
  vcr_vcrnet_weights Wn;
  const vcr_vcrnet_weights* W = &Wn;
  if (weights_take(UW, &Wn) || N <= 0) return 0;
  return (W->partial && W->head_mode == 0) ? overlap_k2(N, W->overlap2) : N;
}

extern "C" int vcr_vcrnet_forward_f32(const vcr_vcrnet_weights* UW, const vcr_vcrnet_io* io, void* ws, size_t bytes,
                                      vcr_stream_t stream) {
  vcr_vcrnet_weights Wn;
  if (weights_take(UW, &Wn)) return VCR_EINVAL;
  return forward_impl(&Wn, io, ws, bytes, stream, nullptr);
}

extern "C" int vcr_vcrnet_forward_traced_f32(const vcr_vcrnet_weights* UW, const vcr_vcrnet_io* io, void* ws,
                                             size_t bytes, vcr_stream_t stream, vcr_trace* tr) {
  vcr_vcrnet_weights Wn;
  if (weights_take(UW, &Wn)) return VCR_EINVAL;
  if (tr) tr->count = 0;
  return forward_impl(&Wn, io, ws, bytes, stream, tr);
}

extern "C" int vcr_vcrnet_iter_f32(const vcr_vcrnet_weights* UW, const vcr_vcrnet_io* io, int iters, void* ws,
                                   size_t bytes, vcr_stream_t stream, vcr_trace* tr) {
  vcr_vcrnet_weights Wn;
  const vcr_vcrnet_weights* W = &Wn;
  if (weights_take(UW, &Wn)) return VCR_EINVAL;
  if (!W || !io || iters < 1 || !io->R_ba || !io->t_ba) return VCR_EINVAL;
  if (tr) tr->count = 0;
  const int B = io->B, N = io->N;
  if (B <= 0 || N <= 0) return VCR_EINVAL;
  if (iters == 1) {
    // vcrnetIter ALWAYS returns the inverse of the composed pose (vcrnet_model.py:40-41), also when args.cycle made
    // the forward itself return the second head's (R_ba, t_ba)
    const int rc = forward_impl(W, io, ws, bytes, stream, tr, !W->cycle);
    if (rc || !W->cycle) return rc;
    Runner R{(hipStream_t)stream, tr};
    R.mark("pose:inverse");
    hipLaunchKernelGGL(pose_step_kernel, dim3(1, B), dim3(64), 0, (hipStream_t)stream, io->R_ab, io->t_ab, nullptr, nullptr, N,
                       2, nullptr, nullptr, io->R_ba, io->t_ba);
    const int lrc = VCR_LAUNCH_RC();
    R.finish();
    return lrc;
  }
  const Ws w = carve(ws, B, N, W->k, W->E, W->F, W->heads, W->partial, W->overlap2, W->emb_kind, W->xscore_limit_mb, merged_encdec(W), W->workspace_flat);
  if (bytes < w.bytes) return VCR_EWORKSPACE;
  // target reuse (forward_impl, `pass`): taken when the caller sized the workspace with vcr_vcrnet_iter_workspace_bytes
  const bool reuse = iter_reuse_applies(W, iters) && bytes >= w.bytes + tgt_cache_floats(B, N, W->E) * sizeof(float);
  const size_t nkeys = W->partial ? (size_t)2 * B * (int)((double)N * W->overlap2) : 0;
  const size_t nsel = W->partial ? (size_t)B * overlap_k1(N, W->overlap2) : 0;
  const size_t npair = W->partial ? (size_t)B * overlap_k2(N, W->overlap2) : 0;
  for (int it = 0; it < iters; ++it) {
    vcr_vcrnet_io step = *io;
    if (it > 0) { step.src_cf = w.cur_cf; step.R_ab = w.Ri; step.t_ab = w.ti; step.R_ba = w.Rb; step.t_ba = w.tb; }
    if (it + 1 < iters) step.emb_out = nullptr;
    // forced / reported selections: one block per iteration
    auto at = [it](auto* p, size_t n) { return p ? p + (size_t)it * n : p; };
    step.force_keys = at(io->force_keys, nkeys);       step.out_keys = at(io->out_keys, nkeys);
    step.force_sel_src = at(io->force_sel_src, nsel);  step.out_sel_src = at(io->out_sel_src, nsel);
    step.force_sel_tgt = at(io->force_sel_tgt, nsel);  step.out_sel_tgt = at(io->out_sel_tgt, nsel);
    step.force_argmax = at(io->force_argmax, nsel);    step.out_argmax = at(io->out_argmax, nsel);
    step.force_pairs = at(io->force_pairs, npair);     step.out_pairs = at(io->out_pairs, npair);
    const bool last = it + 1 == iters;
    const int rc = forward_impl(W, &step, ws, bytes, stream, tr, false, reuse ? (it == 0 ? 1 : 2) : 0);
    if (rc) return rc;
    if (last && it == 0) break;
    Runner R{(hipStream_t)stream, tr};
    R.mark("pose:step");
    hipLaunchKernelGGL(pose_step_kernel, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, step.R_ab,
                       step.t_ab, step.src_cf, last ? nullptr : w.cur_cf, N, it > 0 ? 1 : 0, io->R_ab, io->t_ab,
                       io->R_ba, io->t_ba);
    const int lrc = VCR_LAUNCH_RC();
    if (lrc) return lrc;
    if (last) R.finish();
  }
  return VCR_OK;
}

extern "C" int vcr_pose_step_f32(const vcr_pose_step_args* a, vcr_stream_t stream) {
  if (!a || !a->R_i || !a->t_i || a->B <= 0 || a->compose < 0 || a->compose > 2) return VCR_EINVAL;
  if ((a->out_cf != nullptr) != (a->in_cf != nullptr) || (a->out_cf && a->N <= 0)) return VCR_EINVAL;
  if (a->compose && (!a->R_ba || !a->t_ba)) return VCR_EINVAL;
  if (a->compose == 1 && (!a->R_f || !a->t_f)) return VCR_EINVAL;
  if (!a->compose && !a->out_cf) return VCR_OK;
  const int gx = a->out_cf ? (a->N + 255) / 256 : 1;
  hipLaunchKernelGGL(pose_step_kernel, dim3(gx, a->B), dim3(a->out_cf ? 256 : 64), 0, (hipStream_t)stream, a->R_i, a->t_i,
                     a->in_cf, a->out_cf, a->N, a->compose, a->R_f, a->t_f, a->R_ba, a->t_ba);
  return VCR_LAUNCH_RC();
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

extern "C" int vcr_abi_version(void) { return VCR_ABI_VERSION; }

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
