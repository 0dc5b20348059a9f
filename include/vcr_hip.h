/*
 * vcr_hip.h -- C-ABI of libvcr_hip.so: the MI355X (gfx950) implementation of the VCR-Net
 * per-batch registration hot path.
 *
 * The reference (qiaozhijian/VCR-Net) is pure Python/PyTorch and has no FFI layer; its
 * boundary for this path is the nn.Module protocol of VCRNet (model/vcrnet_model.py:463-518).
 * This header is therefore the NEW native boundary underneath that protocol:
 * one entry point per kernel family plus whole-forward drivers.  Each declaration cites the
 * reference code whose arithmetic it replaces.
 *
 * Conventions (all entry points):
 *   - plain C types only; every pointer is a DEVICE pointer unless the name ends in _host;
 *   - the caller owns every buffer (inputs, outputs, workspace); the library never allocates,
 *     frees or retains pointers, never synchronises, never exits;
 *   - work is enqueued asynchronously on `stream` (a hipStream_t passed as void*);
 *   - return 0 on success, <0 for an argument error (VCR_E*), >0 = a hipError_t from the launch;
 *   - re-entrant: no global mutable state (safe under nn.DataParallel's thread-per-device); tuning / test selectors
 *     travel in the args structs (`variant`, `waves`), 0 = the automatic choice;
 *   - activations are fp32, point-major ("channels-last"): a [B,N,C] tensor is row-major with one
 *     row per point and an explicit row pitch `ld*` in floats.  Indices are int32.
 */
#ifndef VCR_HIP_H
#define VCR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VCR_OK 0
#define VCR_EINVAL (-1)      /* bad shape / null pointer / unsupported size          */
#define VCR_EWORKSPACE (-2)  /* workspace too small (see *_workspace_bytes)           */
#define VCR_EUNSUPPORTED (-3)/* valid request this build does not implement            */

typedef void* vcr_stream_t;  /* hipStream_t */

const char* vcr_strerror(int code);
#define VCR_ABI_VERSION 27
int vcr_abi_version(void);   /* == VCR_ABI_VERSION of the header the library was built from; bumped on any signature / layout change */

/* ---- K-a: conv1_lpd + conv2_lpd (model/lpdnet_model.py:111-112), ReLU == LeakyReLU(0.0) ----
 * x_cf  [B,3,N] channels-first input (the module's forward() argument layout)
 * xyz4  [B,N,4]  = (x, y, z, x^2+y^2+z^2)          feat64 [B,N,64]     sq64 [B,N] = sum_c feat64^2 */
typedef struct {
  const float* x_cf; int B, N;
  const float* w1; const float* b1;   /* [64,3], [64]  */
  const float* w2; const float* b2;   /* [64,64] (16-B aligned, like pq_w below), [64] */
  float* xyz4; float* feat64; float* sq64;
  const float* x_cf2; int B2;         /* optional second block of B2 clouds (same N) processed by the same launch; its rows
                                         follow the first block's in xyz4 / feat64 / sq64 (the forward: src, then tgt) */
  /* Optional (pq != NULL): LPDNet's first EdgeConv projection per point from the same launch,
   * pq[row][0:256] = feat64[row] pq_w^T + pq_b  (pq_w [256,64], pq_b [256]: the neighbour | centre halves of convDG1 after
   * the SURVEY-F7 split, lpdnet_model.py:122; what vcr_linear_f32 computes from feat64).  The launch then runs conv2 and
   * this projection on the matrix pipe; xyz4 / feat64 / sq64 are bit-identical to the plain launch's. */
  const float* pq_w; const float* pq_b; float* pq; int ldpq;
  /* Optional: a second copy of feat64 in the layout the 16-query kNN waves consume without shuffling (vcr_knn_args.xt):
   * every group of 16 channels transposed as a 4 x 4 block, feat64t[row][16 g + 4 a + b] = feat64[row][16 g + 4 b + a]. */
  float* feat64t;
} vcr_pointwise_args;
int vcr_pointwise_f32(const vcr_pointwise_args*, vcr_stream_t);

/* x_cf [B,3,N] channels-first -> xyz4 [B,N,4] rows (x, y, z, x^2+y^2+z^2): layout change only. */
int vcr_rows4_f32(const float* x_cf, float* xyz4, int B, int N, vcr_stream_t);
/* rows4 plus a K = 3 per-point linear in the same pass: pq[i][0:C] = W xyz_i + b, W [C, ldw >= 3] (DGCNN's first EdgeConv
 * through the neighbour / centre split, vcrnet_model.py:108).  C % 4 == 0, C / 4 divides 256. */
int vcr_rows4_pq_f32(const float* x_cf, float* xyz4, int B, int N, const float* wpq, int ldw, const float* bpq, int C,
                     float* pq, int ldpq, vcr_stream_t);

/* ---- kernel 1: fused pairwise-distance + top-k (util/util.py:143-160) ----
 * D_ij = (-sq_j + 2 x_i.x_j) - sq_i ; idx = indices of the k largest D per row after dropping
 * rank 0 ("topk(k+1)[:, :, 1:]").  C == 64 (feature space, fp32 MFMA) or C == 4 (xyz4 rows; Cartesian, VALU).
 * Limits of this library (the reference's knn has none): k <= 62 (lists of k + 2 <= 64 values per query; the tie replay keeps
 * topk(k + 1)'s heap in the 64 lanes of a wave) and N <= 131 072 with B * N < 2^31 (the largest size validated: sampled rows at N = 70 001 and 131 072), VCR_EUNSUPPORTED beyond; the whole forward
 * (vcr_forward_f32) keeps N <= 65535: its other stages were never validated beyond.
 * Exact ties at the (k+1)-th value: with tie_scratch the kept SET equals what Tensor.topk (libstdc++ nth_element /
 * partial_sort on the CPU) keeps; without it one of the tied candidates is kept (deterministically, but not by a
 * documented rule).  A best value shared by two or more entries (copies of a point; a neighbour whose distance rounds to
 * the point's own) is the other tie that decides a set: "rank 0" is the entry Tensor.topk returns FIRST, an outcome of its
 * sort; with tie_scratch such rows are replayed too and the copy the reference drops is dropped, without it the first
 * logged one is.  The replay holds a row's N distances in LDS up to N = 10 091; longer rows need tie_work
 * (vcr_knn_tie_work_bytes(N) bytes of 16-B aligned device scratch) -- with tie_scratch set and tie_work missing such a
 * call returns VCR_EUNSUPPORTED: the replay is never skipped silently. */
typedef struct {
  uint32_t struct_bytes;              /* sizeof(vcr_knn_args) as the CALLER was compiled (ABI 27): fewer bytes than this header's
                                         = a caller built against an older header, the missing tail is read as zeros; 0, less than
                                         the mandatory part (through tie_cap) or more than this library knows: VCR_EINVAL */
  const float* x; int ldx;            /* [B,N,C] rows                                  */
  const float* sq;                    /* [B,N] squared norms (C==64); ignored for C==4 */
  int B, N, C, k;
  int32_t* idx;                       /* [B,N,k], neighbour index within the cloud     */
  int32_t* tie_scratch; int tie_cap;  /* optional: [1 + tie_cap] ints of scratch (count, then the rows with a tie: ~1 row in
                                         10^4 has a boundary tie, and EVERY copy of a repeated point is such a row; rows
                                         beyond tie_cap are not replayed -- B * N entries are always enough) */
  int waves;                          /* tuning / tests, never changes a result.  0 = chosen from the shape.  C == 64: 8 = the
                                         16-query-wave kernel (v_mfma_f32_16x16x4_f32, four lanes per query); 1 / 2 / 4 = the
                                         32-query-wave kernel with that many waves sharing one group of queries and splitting
                                         its candidates.  C == 4: 1 / 2 / 4 likewise. */
  /* tie_zeroed != 0: the caller guarantees tie_scratch[0] == 0 on entry (no zeroing kernel is enqueued here). */
  int tie_zeroed;
  /* tie_defer != 0: the tied rows are only listed; idx is final after a later vcr_knn_ties_f32 on these same args. */
  int tie_defer;
  void* tie_work; size_t tie_work_bytes;   /* see above: NULL / 0 unless vcr_knn_tie_work_bytes(N) > 0 */
  int tie_inline;                     /* the library's own (set on its copy of the struct; callers' value is ignored) */
  /* Optional, C == 64: the same rows as x (same ldx) with every group of 16 channels stored as its 4 x 4 transpose,
   * xt[row][16 g + 4 a + b] = x[row][16 g + 4 b + a] (vcr_pointwise_args.feat64t writes it).  The 16-query-wave kernel
   * feeds v_mfma_f32_16x16x4_f32 one k per lane row; from x it fetches 16-B chunks and transposes them between lane rows
   * for every (query wave, candidate tile) -- 14 of a wave's 87 us at N = 1024; from xt the chunks ARE the operands.
   * Same operands, same k order: the result does not depend on whether xt is given.  x is still required (tie replay,
   * the 32-query kernel). */
  const float* xt;
  /* Optional: the ORDERED search (all seven set, by vcr_knn_order_f32; vcr_knn_pair_f32 with 16-query waves only -- any other
   * launch ignores them).  The points of a cloud are ranked along a Morton curve of their coordinates: perm[b][r] = the point at
   * rank r; xp / sqp = the rows of xt (C == 64; of x for C == 4) and of sq in rank order (pitch ldx); per tile of 16 consecutive
   * ranks: cen = the centroid row (same layout; C == 4: (x, y, z, |c|^2)), cen_sq its squared norm (C == 64), cen_rad an upper
   * bound of the distance of the tile's rows to it, cen_sqmax their largest squared norm.  A wave of 16 queries scans the 9 tiles
   * around its own rank, then only the tiles whose ball (centroid, radius) can still hold a neighbour of one of its queries --
   * decided with the fp32 rounding of the scores priced in, so the neighbour SETS are those of the plain search (rank-0 and
   * boundary-tie rules included: indices are translated back before they are applied).  Ranked clouds up to 8192 points. */
  const int32_t* perm; const float* xp; const float* sqp;
  const float* cen; const float* cen_sq; const float* cen_rad; const float* cen_sqmax;
  /* Optional, with perm: ord_ok[b] == 0 sends cloud b's search over the plain scan although it is ranked (vcr_knn_order_args.ord_ok:
   * the ranking's own verdict on whether this cloud's rows are compact enough along the curve for the pruning to pay).  Decided on
   * the device, cloud by cloud; the neighbour sets do not depend on it. */
  const int32_t* ord_ok;
} vcr_knn_args;

/* Ranks the points of every cloud along a Morton curve of their coordinates and writes what the ordered search reads
 * (vcr_knn_args.perm ...): perm [B,N]; xyz4_p [B,N,4]; cen4 [B,T,4], cen4_rad / cen4_sqmax [B,T] with T = (N + 15) / 16; and,
 * when feat_t is given ([B,N,64] rows in vcr_knn_args.xt's layout, pitch ldf, with sq [B,N]): feat_p [B,N,64] (pitch 64),
 * sq_p [B,N], cen64 [B,T,64], cen64_sq / cen64_rad / cen64_sqmax [B,T].  N <= 8192 (VCR_EUNSUPPORTED beyond). */
typedef struct {
  const float* xyz4;                  /* [B,N,4] rows (x, y, z, |p|^2) */
  const float* feat_t; int ldf; const float* sq;
  int B, N;
  int32_t* perm; float* xyz4_p; float* cen4; float* cen4_rad; float* cen4_sqmax;
  float* feat_p; float* sq_p; float* cen64; float* cen64_sq; float* cen64_rad; float* cen64_sqmax;
  /* Optional guard of the FEATURE-space search (needs feat_t): the ordered search assumes that the features are a smooth function of
   * the coordinates, so that 16 Morton neighbours are close in feature space too.  Per cloud, ord_stat[b] = mean squared tile radius /
   * mean squared distance of the tile centroids from their mean (small: compact tiles far apart -- the pruning pays; unrelated
   * features: ~30), and ord_ok[b] = ord_stat[b] < guard_ratio (0 = the library's threshold, from measurement: profiles/NOTES.md).
   * Hand ord_ok to vcr_knn_args.ord_ok.  Costs one small launch; never changes a result. */
  int32_t* ord_ok; float* ord_stat; float guard_ratio;
} vcr_knn_order_args;
int vcr_knn_order_f32(const vcr_knn_order_args*, vcr_stream_t);
int vcr_knn_f32(const vcr_knn_args*, vcr_stream_t);
size_t vcr_knn_tie_work_bytes(int N);
/* Optional, k > 20 (the launches whose workgroups have no room for a row image in LDS beside their lists): with tie_work of at
 * least this many bytes -- one 16 N-byte slot per workgroup of 64 queries -- such a launch still replays its tied rows ITSELF,
 * the image in the finder's slot, instead of leaving them to a replay launch (vcr_knn_ties_inline() then says 1).  Same sets
 * either way.  (k <= 20: the LDS image up to ~2400 points, the replay launch beyond.) */
size_t vcr_knn_tie_slot_bytes(int B, int N);
/* 1 when vcr_knn_f32 / vcr_knn_pair_f32 with these args replays the tied rows INSIDE the kNN launch (workgroups of 4 x 16
 * queries whose LDS holds a row's replay image: N <= ~2400): idx is then final when the launch ends, whatever tie_defer
 * says, and a later vcr_knn_ties_f32 on these args has nothing to do. */
int vcr_knn_ties_inline(const vcr_knn_args*);
/* The tie replay of one (b == NULL) or two earlier vcr_knn_f32 calls made with tie_defer, as ONE launch: the replay is
 * latency-bound (~25 us whatever the number of tied rows), so two kNN launches whose indices are consumed later -- the
 * Cartesian and the feature-space kNN of LPDNet -- pay for it once. */
int vcr_knn_ties_f32(const vcr_knn_args* a, const vcr_knn_args* b, vcr_stream_t);
/* LPDNet's two independent searches (lpdnet_model.py:113,129) -- a64: C == 64 feature space, a3: C == 4 Cartesian -- as ONE
 * launch when they have the same k and the automatic kernel choice (waves == 0): the unsplit kernels side by side
 * at >= 1024 query groups each (the regime of the path; k <= 40), the candidate-split kernels side by side on small
 * grids (a few pairs per call; k <= 20); otherwise exactly the two vcr_knn_f32 calls.  Same results either way. */
int vcr_knn_pair_f32(const vcr_knn_args* a64, const vcr_knn_args* a3, vcr_stream_t);

/* ---- pointwise linear / 1x1 conv: Y = act(X W^T + bias) (+ residual) ----
 * replaces nn.Conv1d/Conv2d(kernel 1) and nn.Linear on the path (lpdnet_model.py:123-135 after the
 * SURVEY-F7 split, transformer.py:210-212,224,237-238).  fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * K % 32 == 0. */
typedef struct {
  const float* x; int ldx;            /* [M,K] */
  const float* w;                     /* [N,K] row-major (PyTorch weight layout) */
  const float* bias;                  /* [N] or NULL */
  const float* residual; int ldr;     /* [M,N] or NULL, added AFTER the activation */
  float* y; int ldy;                  /* [M,N] */
  int M, N, K;
  int relu;
  /* Optional LayerNorm fusion (transformer.py:141-144; all NULL/0 = plain linear; vcr_linear_f32 only):
   * stats_out   [M, N/64, 2]: the epilogue also writes, per row and 64-column segment (N % 64 == 0), (sum y, sum of
   *             (y - segment mean)^2): moments that combine without cancellation (Chan et al.);
   * ln_stats_in [M, ln_nseg, 2]: such partials over the K columns of x, K / ln_nseg columns per segment.  The operation becomes
   *             Y = act(LayerNorm(X) W0^T + bias0) for the Linear (W0, bias0) and LayerNorm (a, b) that were folded by
   *             vcr_fold_layernorm_f32 into w, bias and ln_colsum [N]:
   *             y = inv_m * (sum_k x[m,k] w[n,k] - mean_m * ln_colsum[n]) + bias[n],  inv = 1 / (std_unbiased + ln_eps). */
  const float* ln_stats_in; int ln_nseg; const float* ln_colsum; float ln_eps;
  float* stats_out;
  /* Optional EdgeConv max-pool fused into the epilogue (DGCNN's x.max(dim=-1), vcrnet_model.py:112-118): the rows are
   * edges, seg_k consecutive rows per point; segmax_out[row / seg_k][n] = max(segmax_out[..][n], y[row][n]) by integer
   * atomic max -- needs relu != 0 (values >= 0) and segmax_out pre-set to 0 (vcr_edgerows_f32 does that).  y may then
   * be NULL: the per-edge activations of the last conv are never written.  LDS-DMA kernels (variant 0) only. */
  float* segmax_out; int ld_segmax; int seg_k;
  int variant;                        /* tuning / tests, 0 = automatic (LDS-DMA staging, one 128x128 tile per workgroup:
                                         BK 16, 32x32x2 MFMAs and four workgroups per CU without a residual; BK 32 and
                                         16x16x4 MFMAs with one -- those on 96x128 tiles when M-tiles of 128 rows
                                         would leave a mostly empty last round of workgroups on the 256 CUs, e.g.
                                         M = 36 864 at BASELINE configs[2]).
                                         bit11 (2048) force 96-row tiles (16x16x4 MFMAs; not with bit10 / bit2:
                                         VCR_EINVAL), bit12 (4096) force 128-row tiles: the tile height does not change
                                         a shape's results; bit13 (8192) / bit14 (16384) force 64- / 32-row tiles (BK 32,
                                         16x16x4 MFMAs; taken automatically, with variant == 0, by launches of far less
                                         than one round of workgroups -- one or two pairs per call);
                                         bit3 (8) force BK 32, bit6 (64) force BK 16; bit4 (16) force the 16x16x4 MFMA
                                         shape, bit10 (1024) force 32x32x2; bit2 (4) the register-staged kernel without
                                         alignment requirements on y / bias / residual (taken automatically when they
                                         are not 16-B aligned).  BK and staging do not change results; the two MFMA shapes
                                         sum k in different orders (fp32-rounding apart).  Any other bit: VCR_EINVAL. */
} vcr_linear_args;
int vcr_linear_f32(const vcr_linear_args*, vcr_stream_t);
/* Host-only query (nothing is launched, no device needed): the kernel configuration vcr_linear_f32 would pick for these
 * arguments -- tile rows | k-slab << 8 | (16x16x4 MFMAs ? 1 << 16 : 0) | (LDS-DMA kernel ? 1 << 17 : 0) -- or a negative
 * VCR_E* code.  For tests and for reading a profile. */
int vcr_linear_config(const vcr_linear_args*);
/* Two independent linears as ONE launch when both resolve to the same kernel configuration (k-slab, MFMA shape,
 * LayerNorm-in, statistics-out; no fused max), else exactly the two calls: fewer, fuller rounds of workgroups for e.g. the
 * encoder's and the decoder's output projections.  The tile height (128 / 96 rows) is chosen for the combined grid unless
 * a variant bit forces it.  Every element is computed as in its own launch with that MFMA shape. */
int vcr_linear_pair_f32(const vcr_linear_args* a, const vcr_linear_args* b, vcr_stream_t);

/* Fold LayerNorm(a, b) into the Linear (w [N,K], bias [N] or NULL) that consumes it, once per weight:
 * w_out[n,k] = w[n,k] a[k];  colsum[n] = sum_k w_out[n,k];  bias_out[n] = bias[n] + sum_k w[n,k] b[k]. */
int vcr_fold_layernorm_f32(const float* w, const float* bias, const float* ln_a, const float* ln_b, int N, int K,
                           float* w_out, float* colsum, float* bias_out, vcr_stream_t);

/* Same operation on the bf16 matrix pipe with fp32-equivalent products: every fp32 operand is split exactly
 * into three bf16 pieces and the six leading partial products are accumulated in fp32 (error <= 2^-24
 * relative per product, i.e. fp32-GEMM accuracy at 2.67x the fp32 matrix rate).  `w_planes` = the weight
 * [N,K] pre-split by vcr_split_bf16x3_f32 (3 consecutive bf16 planes of N*K elements); args->w is ignored.
 * Needs N % 4 == 0 and 16-byte aligned x / y / bias / residual.  ln_stats_in / ln_colsum / stats_out as in
 * vcr_linear_f32 (w_planes then = the split of the FOLDED weight); segmax_out is not offered here. */
int vcr_split_bf16x3_f32(const float* x, void* planes, size_t n, vcr_stream_t);
int vcr_linear_bf16x3_f32(const vcr_linear_args*, const void* w_planes, vcr_stream_t);

/* ---- LayerNorm of model/transformer.py:141-144: a*(x-mean)/(std_unbiased+eps)+b over C ----
 * optional: y += residual; side4[row] = (xyz4[row].xyz, sum_c y^2) for the correspondence head. */
typedef struct {
  const float* x; int ldx; const float* a; const float* b; float eps;
  const float* residual; int ldr;     /* or NULL */
  float* y; int ldy; int M, C;        /* C == 512 */
  const float* xyz4; float* side4;    /* both NULL, or both [M,4] */
} vcr_layernorm_args;
int vcr_layernorm_f32(const vcr_layernorm_args*, vcr_stream_t);

/* y = scale*x (y may be NULL or == x) and side4[row] = (xyz4[row].xyz, |scale*x|^2): the head's side
 * record when the embedding does not pass a LayerNorm (pointer None / Identity, vcrnet_model.py:477-482). */
typedef struct {
  const float* x; int ldx; int M, C; float scale; float* y; int ldy;
  const float* xyz4; float* side4;
} vcr_rowside_args;
int vcr_rowside_f32(const vcr_rowside_args*, vcr_stream_t);

/* ---- kernel 2a: EdgeConv convDG1 -> max -> convDG2 -> max (lpdnet_model.py:122-126, util.py:176-199)
 * pq [M,2*Cmid]: P = X Wn^T (cols 0..Cmid-1), Q = X Wc^T + b (cols Cmid..), from vcr_linear_f32.
 * h_ij = relu(P[nbr_ij] + Q[i]);  x1[i] = max_j h_ij;  x2[i] = relu(max_j (W2 h_ij) + b2).
 * Cmid == Cout == 128.  idx [M,k] cloud-local; n_per_cloud = N. */
typedef struct {
  const float* pq; int ldpq; const int32_t* idx; int k; int M; int n_per_cloud;
  const float* w2; const float* b2;   /* [128,128], [128] */
  float* x1; int ldx1; float* x2; int ldx2;
} vcr_edgeconv_args;
int vcr_edgeconv_f32(const vcr_edgeconv_args*, vcr_stream_t);
/* The same operation with convDG2 (the N*k GEMM on the per-edge rows) on the bf16 matrix pipe as exact 3-way splits
 * (see vcr_linear_bf16x3_f32): k = 20 / 40 only (VCR_EUNSUPPORTED otherwise), w2 / pq / x1 16-B aligned.  x1 is
 * bit-identical to vcr_edgeconv_f32's, x2 agrees to fp32-GEMM rounding.  Used by the driver in linear_mode 1 / 2. */
int vcr_edgeconv_bf16x3_f32(const vcr_edgeconv_args*, vcr_stream_t);

/* ---- kernel 2b: gather + max EdgeConv for a single 1x1 conv (convSN1, lpdnet_model.py:129-132)
 * y[i] = relu(max_j P[nbr_ij] + Q[i]),  C % 4 == 0, C <= 256. */
typedef struct {
  const float* pq; int ldpq; int C; const int32_t* idx; int k; int M; int n_per_cloud;
  float* y; int ldy;
  int variant;   /* tuning / tests, 0 = automatic: 1 = neighbour rows gathered through L2 (one wave per point); 32 / 16 / 8 =
                    gathered out of LDS with that many channels per workgroup slice (VCR_EUNSUPPORTED when the slice of one
                    cloud does not fit a workgroup's LDS, k is not 20 / 40, or pq / y / idx are not 16-B aligned) */
  /* Optional, the L2 form only: order [M] int32, cloud by cloud a permutation of 0 .. n_per_cloud - 1 (vcr_knn_args.perm: the
   * clouds' Morton ranking).  Wave w then serves the point order[w] of its cloud instead of point w: waves that run side by side
   * gather the rows of NEIGHBOURING points, most of which they share -- the gathers hit in the CU's L1 instead of L2.  Every
   * point is still served exactly once, by the same arithmetic: the output does not depend on it. */
  const int32_t* order;
} vcr_gathermax_args;
/* (automatic: k = 20 / 40, n_per_cloud <= 2048, >= 192 (cloud, 32-channel slice) workgroups, 16-B aligned pq / y / idx: the
 * neighbour rows are gathered out of LDS -- a workgroup stages its slice of one cloud's P rows once (32 channels up to
 * 1066 points, 16 beyond) -- instead of through L2; same bits either way.) */
int vcr_gathermax_f32(const vcr_gathermax_args*, vcr_stream_t);

/* ---- EdgeConv chains (DGCNN, vcrnet_model.py:104-118): per-edge rows h[(i,j)] = relu(P[nbr_ij] + Q[i])
 * ([M*k, C], feeding vcr_linear_f32 for conv2..conv4) and the max over each point's k edge rows. */
typedef struct {
  const float* pq; int ldpq; int C; const int32_t* idx; int k; int M; int n_per_cloud; float* h; int ldh;
  /* optional (C == 64): ymax[i][0:C] = max_j h[(i,j)] written by the same pass (x1 of vcrnet_model.py:109), and
   * ymax[i][C:zero_to] = 0 -- the base the fused maxima of the following convs accumulate into (vcr_linear_args.segmax_out) */
  float* ymax; int ldymax; int zero_to;
} vcr_edgerows_args;
int vcr_edgerows_f32(const vcr_edgerows_args*, vcr_stream_t);
/* The whole chain in one kernel (k = 20 or 40, else VCR_EUNSUPPORTED -- use the pieces above): pq = conv1's per-point
 * (P | Q) rows [M, >= 128]; w2 [64,64], w3 [128,64], w4 [256,128] row-major [out][in] with their biases (BatchNorm
 * folded); out [M, >= 512] = (x1 | x2 | x3 | x4) of vcrnet_model.py:109-118.  No per-edge tensor is written. */
typedef struct {
  const float* pq; int ldpq; const int32_t* idx; int k; int M; int n_per_cloud;
  const float *w2, *b2, *w3, *b3, *w4, *b4;
  float* out; int ldo;
} vcr_edgechain_args;
int vcr_edgechain_f32(const vcr_edgechain_args*, vcr_stream_t);
typedef struct {
  const float* x; int ldx; int M, k, C; float* y; int ldy;
} vcr_segmax_args;
int vcr_segmax_f32(const vcr_segmax_args*, vcr_stream_t);

/* ---- kernel 3: scaled-dot-product attention, flash-style (transformer.py:29-34,55) ----
 * q,k,v: [nbatch*n, h*128] rows with pitches; head hh uses columns hh*128..+127.
 * out = softmax(q k^T * scale) v, never materialising the n x n scores.
 * kv_batch_shift: keys/values of batch b come from batch (b + shift) % nbatch (decoder cross-attn
 * over the 2B-batched encoder memory).  key_keep: NULL, or uint8 [nbatch, nk] -- keys with 0 are
 * masked out (transformer.py:46-53; exp underflows to exactly 0 for masked_fill(-1e9)).
 * rowstat: NULL or [nbatch,h,nq,2] (max, sum) of the scaled scores -- for the partial path.
 * score_out: NULL or [nbatch,h,nq,ld_score] -- the scaled scores themselves (needs rowstat; ld_score % 4 == 0 and
 *            >= nk rounded up to 32, the pad receives -inf), for vcr_keymass_f32. */
typedef struct {
  const float* q; int ldq; const float* k; int ldk; const float* v; int ldv;
  float* out; int ldo;
  int nbatch, heads, nq, nk; float scale; int kv_batch_shift;
  const uint8_t* key_keep; float* rowstat;
  float* score_out; int ld_score;
  /* Optional: ngroups > 1 runs that many independent attention problems of identical shape as ONE launch (the
   * encoder's and the decoder's self-attention of the forward: fewer, fuller rounds of workgroups).  Group g reads
   * q / k / v at q + g * q_group_stride ... (element offsets) and writes out + g * out_group_stride; kv_batch_shift acts
   * inside a group.  Attention-output form only (no key_keep / rowstat / score_out); 0 or 1 = one group; vcr_sdpa_f32 only. */
  int ngroups; long q_group_stride, k_group_stride, v_group_stride, out_group_stride;
  /* Optional: key_index int32 [nbatch, nk] -- the keys of KEY batch kb are the rows key_index[kb][0..nk-1] of its nk_src
   * rows of k / v (duplicates allowed; vcr_sdpa_f32 only, not with key_keep / score_out / ngroups).  The same set of
   * keys as a key_keep mask, with no masked scores computed and no dense copy of the kept rows. */
  const int32_t* key_index; int nk_src;
  /* Optional scratch of split_work_floats floats.  Statistics passes (out == NULL) need VCR_SDPA_MAX_SPLIT * nbatch *
   * heads * nq * 2: with it the launch may deal the key tiles to up to that many workgroups per query block when its grid
   * would otherwise end in a mostly empty round of workgroups (1152 on 512 resident at BASELINE configs[2]); the partial
   * (max, sum) pairs are merged in a fixed order by a second small kernel.  Attention-output launches of less than one
   * round of workgroups (small batches) split the keys too when nsplit * G * (nbatch * nq * ldo + nbatch * heads * nq * 2)
   * floats fit (G = max(ngroups, 1); no key_keep / key_index / rowstat): partial outputs merged by sdpa_merge_kernel.
   * Scores are unaffected; sums and outputs merge in a different order.  A scratch too small for a split: no split. */
  float* split_work; long split_work_floats;
  /* tuning / tests, never changes a bit of the result: 0 = the library's choice (the persistent kernel where it applies), 1 = the
   * tile kernel (one workgroup per query block and batch x head), 2 = the persistent kernel (2 x CUs workgroups walk the same
   * items; the fast attention-output form with at least two items per workgroup -- any other call runs the tile kernel).
   * vcr_sdpa_f32 only. */
  int variant;
  /* 0, or the number of batches of the launch this one stands for (>= nbatch): the key-split decision -- the one choice of
   * this entry point that changes the summation order of its results -- is then taken for a grid of that many batches.  The
   * forward's source-only attention launches of a later vcrnetIter pass (nbatch = B) give 2 B here and so compute every row
   * exactly as the full launch of the first pass did. */
  int plan_nbatch;
} vcr_sdpa_args;
#define VCR_SDPA_MAX_SPLIT 4
int vcr_sdpa_f32(const vcr_sdpa_args*, vcr_stream_t);
/* The attention-output form of vcr_sdpa_f32 (out != NULL, rowstat == score_out == NULL, scale > 0) with Q, K, V and the
 * soft-max probabilities split exactly into three bf16 pieces and every product evaluated as six bf16 MFMAs with fp32
 * accumulation ("bf16x3", see vcr_linear_bf16x3_f32): fp32-GEMM accuracy on the bf16 matrix pipe.  Opt-in (linear_mode 2
 * of the drivers); VCR_EUNSUPPORTED for the statistics forms, which stay on vcr_sdpa_f32. */
int vcr_sdpa_bf16x3_f32(const vcr_sdpa_args*, vcr_stream_t);

/* ---- key mass of the partial-mode decoder (transformer.py:40): mass[kb][key] = sum over heads and queries of the
 * soft-max probability that key receives, from the scores and row statistics a statistics-only vcr_sdpa_f32 pass
 * stored (queries of batch (kb + q_batch_shift) % nbatch attend to the keys of batch kb).  One HBM-bound pass
 * instead of recomputing the 128-d scores per head. */
typedef struct {
  const float* score; int ld; int nbatch, heads, nq, nk;
  const float* rowstat; int q_batch_shift; float* mass;
} vcr_keymass_args;
int vcr_keymass_f32(const vcr_keymass_args*, vcr_stream_t);

/* ---- virtual-correspondence head, whole mode (vcrnet_model.py:334-347, :402-421, dcp_model.py:138-142)
 * corr[i] = sum_j softmax_j(score_ij) * kside4[j].xyz
 * mode 0: score = (-|q_i|^2 + 2 q_i.k_j) - |k_j|^2 (VcpTopK/VcpAtt); mode 1: q_i.k_j * scale (VcpByDis/DCP).
 * qside4/kside4 [nbatch*n,4] = (x,y,z,|emb|^2).  E % 64 == 0.  corr4 [nbatch*nq,4] (w = 0). */
typedef struct {
  const float* q; int ldq; const float* k; int ldk; const float* qside4; const float* kside4;
  float* corr4; int nbatch, nq, nk, E; int mode; float scale;
  float* split_work;                  /* optional: vcr_pairscore_args.split_work of the underlying op-0 launch ... */
  long split_work_floats;             /* ... and its capacity in floats (see there) */
} vcr_softcorr_args;
int vcr_softcorr_f32(const vcr_softcorr_args*, vcr_stream_t);

/* ---- general pair-score kernel (partial-overlap path): everything the reference derives from an
 * n_own x n_str score matrix it materialises (vcrnet_model.py:190-332 selectCom/getCopair, transformer.py:35-53)
 * op 0 SOFTMAX_PV: corr4[o] = sum_s softmax_s(score) * str_side4[s].xyz            (== vcr_softcorr_f32)
 * op 1 STATS     : stat2[o] = (max_s score, sum_s exp(score - max)), argmax[o] = first arg-max s (optional)
 * op 2 MASS      : mass[o] (+)= sum_s exp(score - m_s) / l_s with (m_s, l_s) = str_stat2[s]  -- the column sums
 *                  of a row soft-max whose rows are the STREAMED index
 * score 0: (-|own|^2 + 2 own.str) - |str|^2;  2: (-|str|^2 + 2 own.str) - |own|^2;  1: own.str * scale.
 * The streamed batch of owner batch b is (b + str_batch_shift) % nbatch; str_stat2 is indexed
 * [streamed batch * str_stat_batch_stride + s*2].  E % 128 == 0, E <= 1024. */
typedef struct {
  const float* own; int ld_own; const float* str; int ld_str;
  const float* own_side4; const float* str_side4;   /* [nbatch*n,4] = (x,y,z,|emb|^2); may be NULL for score 1 */
  int nbatch, n_own, n_str, E; int score; float scale; int str_batch_shift; int op;
  float* corr4; float* stat2; int32_t* argmax;
  const float* str_stat2; long str_stat_batch_stride; float* mass; int accumulate;
  float* score_out; int ld_score;     /* op 1 only, optional: also keep the scores, S[(b*n_own + o)*ld_score + s];
                                         ld_score % 4 == 0 and >= n_str rounded up to 32 (the pad receives -inf) */
  int variant;                        /* tuning / tests, 0 = automatic: bit0 one owner tile (32 owners) per block; bit2 (4)
                                         never split the streamed side */
  /* op 0, or op 1 without argmax, optional: VCR_PAIRSCORE_MAX_SPLIT * nbatch * n_own * 8 floats of scratch (op 1: * 2).
   * With it the launch may deal the streamed rows to up to that many workgroups per owner block when its grid would
   * otherwise leave a mostly empty last round on the chip (e.g. 288 or 320 workgroups on 256 CUs) or fill only part of it;
   * the partial (max, sum[, weighted xyz]) records are merged in a fixed order by a second small kernel.  Scores and
   * arg-max are unaffected; the sums merge in a different order.  split_work_floats = the buffer's capacity in floats: a
   * split whose records would not fit is not taken (the launch then runs unsplit), so an under-sized buffer costs speed,
   * never an out-of-bounds write. */
  float* split_work; long split_work_floats;
} vcr_pairscore_args;
#define VCR_PAIRSCORE_MAX_SPLIT 4
int vcr_pairscore_f32(const vcr_pairscore_args*, vcr_stream_t);

/* ---- both probability masses of selectCom (vcrnet_model.py:217-248) from a STORED score matrix
 * score [nbatch, n_rows, ld] and its row statistics row_stat2 [nbatch,n_rows,2] = (max_j, sum_j exp(. - max)):
 *   col_stat2[j] = (max_i S_ij, sum_i exp(S_ij - max))        soft-max over dim=1
 *   col_mass[j]  = sum_i exp(S_ij - m_i) / l_i                scoresColSum (:222)
 *   row_mass[i]  = sum_j exp(S_ij - cm_j) / cl_j              scoresRowSum (:244)
 * Two HBM-bound passes over S instead of three more N x N x E score GEMMs. */
typedef struct {
  const float* score; int ld; int nbatch, n_rows, n_cols;
  const float* row_stat2; float* col_stat2; float* col_mass; float* row_mass;
} vcr_scoremass_args;
int vcr_scoremass_f32(const vcr_scoremass_args*, vcr_stream_t);

/* ---- per-sample top-K of n scores in descending order, ties -> lower index (Tensor.topk at
 * transformer.py:42, vcrnet_model.py:223,245,312).  order [nbatch,K] int32 and/or mask [nbatch,n] uint8. */
typedef struct {
  const float* values; int nbatch, n, K; int32_t* order; uint8_t* mask;
  int largest;                        /* 1: K largest, descending; 0: K smallest, ascending */
  int stride;                         /* element stride between consecutive values (0 or 1: dense); sample b starts
                                         at values + b*n*stride -- lets column 1 of a [nbatch,n,2] record be ranked */
} vcr_rankselect_args;
int vcr_rankselect_f32(const vcr_rankselect_args*, vcr_stream_t);

/* ---- out[b][r][0:C] = in[b][idx[b][r]][0:C]  (index gathers of vcrnet_model.py:230-260,305-330).
 * via (optional, [nbatch,n_via] int32): the row taken is via[b][idx[b][r]] -- the arg-max target of the r-th
 * kept source in getCopair (vcrnet_model.py:325). */
typedef struct {
  const float* in; int ld_in; int n_in; const int32_t* idx; int nbatch, n_out, C; float* out; int ld_out;
  const int32_t* via; int n_via;
} vcr_gather_args;
int vcr_gather_rows_f32(const vcr_gather_args*, vcr_stream_t);

/* ---- kernel 4: weighted-covariance + 3x3 SVD rigid solve (vcrnet_model.py:356-399) ----
 * src/corr: [B,K,ld] rows (first 3 floats used).  R [B,9] row-major acting on column vectors,
 * t [B,3]; R_ba = R^T, t_ba = -R^T t (vcrnet_model.py:515-516) written when non-NULL. */
typedef struct {
  const float* src; int lds; const float* corr; int ldc; int B, K;
  float* R; float* t; float* R_ba; float* t_ba; float* H; /* H [B,9] optional diagnostic */
} vcr_rigid_svd_args;
int vcr_rigid_svd_f32(const vcr_rigid_svd_args*, vcr_stream_t);

/* ---- ICP refinement, the --iter 0 path (ICP.forward, model/icp_model.py:26-48; used by vcrnetIcpNet,
 * model/vcrnet_model.py:46-62): <= max_iterations rounds of {nearest neighbour, best-fit transform, apply},
 * stopping when the BATCH-mean error changes by < tolerance -- tested on the device, no host sync.
 * src4 / dst4: [B,N,4] / [B,M,4] rows (x,y,z,|p|^2).  final4 [B,N,4]: the moved source.  R,t: the transform
 * src -> final (icp_model.py:42).  iterations: optional device int = rounds executed. */
typedef struct {
  const float* src4; const float* dst4; int B, N, M; int max_iterations; float tolerance;
  float* final4; float* R; float* t; float* R_ba; float* t_ba; int* iterations;
} vcr_icp_args;
size_t vcr_icp_workspace_bytes(int B, int N);
int vcr_icp_f32(const vcr_icp_args*, void* workspace, size_t workspace_bytes, vcr_stream_t);

/* ---- evaluation-pair construction (ModelNet40.__getitem__, util/data.py:247-314; partial crop :320-329).
 * The host draws the random numbers (the reference seeds legacy NumPy with the item index, :255-256); the
 * device applies them to base clouds resident in HBM:
 *   src_i = cloud[pick[perm_src[i]]];   tgt_i = R_ab cloud[pick[perm_tgt[i]]] + t_ab  (float64, then float32)
 * keep == N: no crop; keep < N: the `keep` points nearest to the LAST point, in order of distance.
 * cloud [B,P,3]; R_ab [B,9], t_ab [B,3] doubles; pick / perm_src / perm_tgt [B,N] int32; outputs [B,3,keep]. */
typedef struct {
  const float* cloud; int P;
  const double* R_ab; const double* t_ab;
  const int32_t* pick; const int32_t* perm_src; const int32_t* perm_tgt;
  int B, N, keep;
  float* src_cf; float* tgt_cf;
} vcr_make_pairs_args;
int vcr_make_pairs_f32(const vcr_make_pairs_args*, vcr_stream_t);

/* ---- whole forward: VCRNet.forward (vcrnet_model.py:495-518), LPDNet + Transformer + VcpTopK(whole)/
 * VcpByDis + SVD, both clouds batched as 2B.  Weight pointers are the packed device tensors the host
 * module prepares once; all [N,K] row-major. */
typedef struct {
  const float *ln_a, *ln_b;
} vcr_norm_w;
typedef struct {
  const float *wqkv, *bqkv;           /* [3E,E],[3E]  linears.0,1,2 stacked (self-attention) */
  const float *wq, *bq, *wkv, *bkv;   /* cross-attention split: [E,E],[E],[2E,E],[2E]        */
  const float *wo, *bo;               /* linears.3 */
} vcr_mha_w;
typedef struct {
  const float *w1, *b1, *w2, *b2;     /* [F,E],[F],[E,F],[E] */
} vcr_ffn_w;
typedef struct {
  uint32_t struct_bytes;                           /* sizeof(vcr_vcrnet_weights) as the CALLER was compiled (ABI 27; see vcr_knn_args):
                                                      the mandatory part ends before fold_encdec_qkv; a shorter, zero or longer
                                                      value is VCR_EINVAL (vcr_vcrnet_workspace_bytes / vcr_vcrnet_pairs: 0) */
  /* LPDNet */
  const float *c1_w, *c1_b, *c2_w, *c2_b;          /* conv1_lpd, conv2_lpd                         */
  const float *dg1_wpq, *dg1_bpq;                  /* [256,64]: rows 0..127 = W[:, :64] (neighbour), 128.. = W[:, 64:] (centre); bias [256] = (0, b) */
  const float *dg2_w, *dg2_b;                      /* [128,128]                                     */
  const float *sn1_wpq, *sn1_bpq;                  /* [512,128], [512]                              */
  const float *c3_w, *c3_b;                        /* [E,512]                                       */
  /* Transformer (n_blocks == 1) */
  vcr_norm_w enc_ln0, enc_ln1, enc_norm, dec_ln0, dec_ln1, dec_ln2, dec_norm;
  vcr_mha_w enc_self, dec_self, dec_cross;
  vcr_ffn_w enc_ffn, dec_ffn;
  int E, F, heads, k;                              /* 512, 1024, 4, 20 */
  int has_pointer;                                 /* 1 transformer, 0 none, 2 identity (emb*2)     */
  int head_mode;                                   /* 0 VcpTopK (neg-distance), 1 VcpByDis / DCP (dot/sqrt(E)), 2 VcpAtt */
  /* linear_mode 0: every 1x1 conv / Linear on v_mfma_f32_32x32x2_f32 (vcr_linear_f32).
   * linear_mode 1: the same products as exact 3-way bf16 splits on the bf16 matrix pipe (vcr_linear_bf16x3_f32);
   * then `split` holds the weights pre-split by vcr_split_bf16x3_f32, in the order of the sites below -- for the six
   * sites that consume a LayerNorm (enc_qkv, enc_ffn1, dec_qkv, dec_cross_q, dec_cross_kv, dec_ffn1) the split of the
   * FOLDED weight fold_<site>.w.
   * linear_mode 2: mode 1, and the attention-output launches through vcr_sdpa_bf16x3_f32 as well. */
  int linear_mode;
  struct {
    const void *dg1_pq, *sn1_pq, *c3, *enc_qkv, *enc_wo, *enc_ffn1, *enc_ffn2, *dec_qkv, *dec_self_wo, *dec_cross_q,
               *dec_cross_kv, *dec_cross_wo, *dec_ffn1, *dec_ffn2;
    const void* encdec_qkv;                          /* optional: the split of fold_encdec_qkv.w [6E,E] (the merged first sublayers, below) */
  } split;
  /* has_pointer 1 (every linear_mode): the six Linears that consume a LayerNorm, folded with it by
   * vcr_fold_layernorm_f32 (w [N,E], colsum [N], bias [N]).  dec_cross_kv is folded with the ENCODER's final norm. */
  struct vcr_folded { const float *w, *colsum, *bias; } fold_enc_qkv, fold_enc_ffn1, fold_dec_qkv, fold_dec_cross_q,
      fold_dec_cross_kv, fold_dec_ffn1;
  /* optional (linear_mode 0; 1 / 2 with split.encdec_qkv): fold_enc_qkv and fold_dec_qkv stacked, w [6E,E], colsum [6E], bias [6E] -- both consume the
   * embedding rows with the same row statistics, so the two projections run as ONE GEMM and the encoder's and the
   * decoder's self-attention as ONE grouped launch (fewer, fuller rounds of workgroups).  w == NULL: two launches each. */
  struct vcr_folded fold_encdec_qkv;
  /* partial-overlap mode (args.partial, vcrnet_model.py:178-187 + transformer.py:35-53): the decoder's
   * cross-attention keeps the int(N*overlap2) keys with the largest soft-max mass; with head_mode 0 the head is
   * selectCom + getCopair and the outputs hold vcr_vcrnet_pairs() hard pairs per sample instead of N soft ones.
   * overlap2 is a double because the reference truncates float64 products of it (vcrnet_model.py:208,284). */
  int partial;
  double overlap2;
  /* emb_kind 1: DGCNN embedding (vcrnet_model.py:90-123 = dcp_model.py:46-79) instead of LPDNet; eval-mode BatchNorm
   * folded into each bias-free 1x1 conv by the host (w' = w*g/sqrt(var+eps), b' = beta - mean*g/sqrt(var+eps)).
   * c1_wpq [128,32]: rows 0..63 = conv1 columns 0..2 (neighbour part), rows 64..127 = columns 3..5 (centre part),
   * K padded 3 -> 32 with zeros; c1_bpq [128] = (0, b1').  c2 [64,64], c3 [128,64], c4 [256,128], c5 [E,512]. */
  int emb_kind;
  struct {
    const float *c1_wpq, *c1_bpq, *c2_w, *c2_b, *c3_w, *c3_b, *c4_w, *c4_b, *c5_w, *c5_b;
  } dgcnn;
  /* emb_kind 2: PointNet embedding (vcrnet_model.py:65-87): five pointwise convs + eval-mode BatchNorm (folded by the host
   * like DGCNN's) + ReLU, no graph (k is ignored).  conv1 / conv2 travel in c1_w [64,3], c1_b, c2_w [64,64], c2_b above
   * (the LPDNet stem has the same shape); c3 [64,64], c4 [128,64], c5 [E,128] here.  These three linears stay fp32 in
   * linear_mode 1 / 2. */
  struct {
    const float *c3_w, *c3_b, *c4_w, *c4_b, *c5_w, *c5_b;
  } pointnet;
  /* head_mode 2: VcpAtt's two Linear(E,E) on the source / target embeddings (head.linears_emb.0/1, [E,E] + [E]) */
  const float *att_w0, *att_b0, *att_w1, *att_b1;
  /* args.cycle (vcrnet_model.py:511-513): (R_ba, t_ba) from a second head + solve with the clouds swapped
   * (soft heads only) instead of the inverse of (R_ab, t_ab) */
  int cycle;
  /* MFMA shape of the fp32 linears of the forward: 0 = the library's choice per launch, 16 = v_mfma_f32_16x16x4_f32,
   * 32 = v_mfma_f32_32x32x2_f32 (benchmarks / tests; results agree to fp32 rounding) */
  int linear_mfma;
  int linear_bk;                                   /* 0 = the library's choice, 16 / 32 = that k-slab for every fp32 linear (benchmarks) */
  int linear_bm;                                   /* 0 = the library's choice, 96 / 128 = that tile height (variant bits 11 / 12; benchmarks) */
  int knn_waves;                                   /* vcr_knn_args.waves of the feature-space kNN (0 = automatic; benchmarks) */
  /* partial mode: the decoder's cross-attention scores ([2B,H,N,N] fp32) are kept between the statistics pass and the
   * key-mass pass when they fit this many MiB of workspace (0 = 4096), and recomputed per head otherwise (< 0: never
   * kept).  Same kept-key set either way up to summation order. */
  int xscore_limit_mb;
  int sdpa_variant;                                /* vcr_sdpa_args.variant of the fp32 attention-output launches (0 = automatic; benchmarks) */
  /* vcr_vcrnet_iter_f32 with iters > 1: 0 = the passes after the first reuse what the first computed from the TARGET cloud alone
   * (it does not change between passes: its embedding, the encoder on its rows, the decoder's self-attention sublayer and
   * cross-attention query on its rows, the K | V projection of its encoder memory) -- when the workspace was sized with
   * vcr_vcrnet_iter_workspace_bytes; 1 = every pass recomputes both clouds, as the reference does.  Bit-identical either way. */
  int iter_reuse;
  /* tests: != 0 lays the forward's workspace out with EVERY buffer live from the first launch to the last (no two buffers
   * share memory; vcr_vcrnet_workspace_bytes grows ~2.5x).  The default layout overlays buffers by their (first, last) launch;
   * a lifetime registered too short would let one launch overwrite what a later one still reads -- comparing the two
   * layouts bit for bit is the test for that.  Results never depend on it. */
  int workspace_flat;
} vcr_vcrnet_weights;

typedef struct {
  const float* src_cf; const float* tgt_cf;        /* [B,3,N] channels-first, as VCRNet.forward gets them */
  int B, N;
  float* corr4;                                    /* [B,K,4] src_corr (x,y,z,0); K = vcr_vcrnet_pairs(): N in whole mode */
  float* src4;                                     /* [B,K,4] the matched src points as rows (x,y,z,|p|^2) */
  float* R_ab; float* t_ab; float* R_ba; float* t_ba;  /* [B,9],[B,3],[B,9],[B,3] */
  float* emb_out;                                  /* optional [2B*N,E] final embeddings (src then tgt), may be NULL */
  /* Partial-overlap mode only, every pointer optional (NULL = free-running / not wanted): the discrete selections of
   * the path, in the reference's own order and index spaces.  `force_*` REPLACE the device's ranking with the
   * caller's (teacher forcing: SURVEY F5 -- one flipped near-tie moves (R, t) by 1e-2, so the 1e-4 / 1e-5 tolerance
   * is assertable per iteration only on identical selections); `out_*` report what was used.  For
   * vcr_vcrnet_iter_f32 every array holds one block per iteration, iteration i at offset i * (block size).
   *   keys    [2B, nkeep]  kept keys of the decoder cross-attention per KEY cloud (src clouds, then tgt clouds),
   *                        nkeep = int(N*overlap2)                                     (transformer.py:41-42)
   *   sel_src / sel_tgt [B, K1]  overlap sets of selectCom, K1 = int(N*0.84*overlap2) (vcrnet_model.py:223,245)
   *   argmax  [B, K1]      arg-max target (position in sel_tgt) of each selected source (vcrnet_model.py:297)
   *   pairs   [B, K2]      positions in sel_src of the kept sources, K2 = vcr_vcrnet_pairs() (vcrnet_model.py:312) */
  const int32_t *force_keys, *force_sel_src, *force_sel_tgt, *force_argmax, *force_pairs;
  int32_t *out_keys, *out_sel_src, *out_sel_tgt, *out_argmax, *out_pairs;
} vcr_vcrnet_io;

size_t vcr_vcrnet_workspace_bytes(const vcr_vcrnet_weights*, int B, int N);
/* The same for vcr_vcrnet_iter_f32(iters): with iters > 1 it adds room behind the forward's workspace (2 B N x 2560 floats) for the
 * four buffers whose target-cloud rows the later passes reuse; a workspace of only vcr_vcrnet_workspace_bytes still works -- every
 * pass then recomputes both clouds. */
size_t vcr_vcrnet_iter_workspace_bytes(const vcr_vcrnet_weights*, int B, int N, int iters);
/* Correspondences per sample in corr4/src4: N, or int(int(N*0.84*overlap2)*0.52*overlap2) for partial + head_mode 0. */
int vcr_vcrnet_pairs(const vcr_vcrnet_weights*, int N);
int vcr_vcrnet_forward_f32(const vcr_vcrnet_weights*, const vcr_vcrnet_io*, void* workspace,
                           size_t workspace_bytes, vcr_stream_t);

/* Optional per-launch timing of the whole forward.  `events` holds `capacity` hipEvent_t handles
 * created by the caller; event i is recorded on `stream` immediately before launch i and one more
 * after the last launch, so launch i took elapsed(events[i], events[i+1]).  names[i] is a static
 * string "family:site" (families: pointwise, knn, linear, edgeconv, gathermax, layernorm, sdpa,
 * softcorr, rigid_svd, select).  count = number of launches recorded. */
#define VCR_TRACE_MAX 256
typedef struct {
  void** events; int capacity; int count;
  const char* names[VCR_TRACE_MAX];
} vcr_trace;
int vcr_vcrnet_forward_traced_f32(const vcr_vcrnet_weights*, const vcr_vcrnet_io*, void* workspace,
                                  size_t workspace_bytes, vcr_stream_t, vcr_trace*);

/* vcrnetIter (vcrnet_model.py:21-43): `iters` forwards, each on the source moved by the previous pose
 * (transform_point_cloud, util/util.py:91-96), poses composed on the device (R_f <- R_i R_f, t_f <- R_i t_f + t_i);
 * io->R_ab/t_ab receive the composed pose, R_ba/t_ba its inverse, corr4/src4 the LAST iteration's pairs.
 * No host synchronisation between iterations.  trace may be NULL; launches of all iterations are appended. */
int vcr_vcrnet_iter_f32(const vcr_vcrnet_weights*, const vcr_vcrnet_io*, int iters, void* workspace,
                        size_t workspace_bytes, vcr_stream_t, vcr_trace* trace);

/* One step of vcrnetIter's / vcrnetIcpNet's bookkeeping (vcrnet_model.py:32-38,52-59; transform_point_cloud, util/util.py:91-96)
 * for hosts that drive the passes themselves (what vcr_vcrnet_iter_f32 does between its passes):
 *   out_cf = R_i in_cf + t_i                       when out_cf != NULL ([B,3,N] channels-first; in place is allowed)
 *   compose 1: R_f <- R_i R_f, t_f <- R_i t_f + t_i (in place), then (R_ba, t_ba) = (R_f^T, -R_f^T t_f)
 *   compose 2: (R_ba, t_ba) = (R_i^T, -R_i^T t_i)   (the inverse of this pass's pose only; R_f / t_f unused)
 *   compose 0: neither.
 * Products are k-ascending fma chains, as the reference's CPU matmul computes 3 x 3 products. */
typedef struct {
  const float* R_i; const float* t_i;              /* [B,9], [B,3] */
  int B, N;
  const float* in_cf; float* out_cf;               /* optional pair */
  int compose;
  float* R_f; float* t_f;                          /* compose 1 */
  float* R_ba; float* t_ba;                        /* compose 1 / 2 */
} vcr_pose_step_args;
int vcr_pose_step_f32(const vcr_pose_step_args*, vcr_stream_t);

/* hipEvent helpers (create / destroy / record / elapsed) bound to the same HIP runtime as the
 * kernels, for hosts without HIP bindings.  Elapsed needs both events completed (synchronise first). */
int vcr_event_create(void** ev);
int vcr_event_destroy(void* ev);
int vcr_event_record(void* ev, vcr_stream_t stream);
int vcr_event_elapsed_ms(void* start, void* stop, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* VCR_HIP_H */
