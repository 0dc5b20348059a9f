// Kernel 1: fused pairwise-distance + top-k (util/util.py:143-160).  The N x N distance matrix is
// never written: distances are produced tile by tile in registers and filtered against each
// query's current (k+2)-th best value; survivors are logged per query in LDS.
//
//   D_ij = (-sq_j + 2 x_i.x_j) - sq_i      (same association as util.py:157-158)
//   idx  = top-(k+1) of D_i. by value, rank 0 dropped (util.py:159); the k kept indices are written as a SET
//          (unordered) -- every consumer is a max over neighbours.
//          Exact ties at the (k+1)-th value: Tensor.topk on the CPU is libstdc++'s std::nth_element (or
//          std::partial_sort when (k+1)*64 <= N) with a value-only comparator, so WHICH of the tied candidates it
//          keeps is an artefact of introselect's pivoting / the heap's shape.  The value lists here carry one entry
//          more than needed, which makes such a tie visible (about 1 row in 10^4 in fp32); those rows are re-done by
//          knn_tiebreak_kernel, a replica of the libstdc++ algorithms, so that the neighbour SETS equal the
//          reference's on every row (validated against torch.topk on tie-heavy inputs).  A row WITHOUT a boundary tie
//          has a set that depends on the values only; without tie_scratch a boundary tie keeps the tied candidates
//          that were scanned first.
//          A shared BEST value (copies of a point, or a neighbour so close that its distance rounds to the point's own --
//          fp32 self-distances are not exactly 0) is the other tie that matters: util.py:159 drops whichever entry topk
//          returns first, and that is position 0 after ATen's sort of the selected entries (std::sort of the first k after
//          nth_element, or partial_sort's heap sort), not the lowest index.  Such rows are listed and replayed as well; the
//          replay ends with a port of that sort.  (Found in round 6 by the vcrnetIter reuse soak: two launch forms of the
//          Cartesian search logged such a pair in different orders and kept different copies.)
//
// Selection, round 2 (measured on the round-1 kernels: 40 % of their time went into the sorted-insert network that
// moved (value, index) pairs through 22-42 register slots, ~100 issue slots per insertion):
//   * registers hold the sorted top-KS VALUES only: an insertion is one v_med3_f32 per slot, no compares, no index
//     traffic.  The list of a query is spread over the lanes that share the query (2 for the MFMA layout, 4 in the
//     Cartesian kernel); lane segment s takes min(d, last value of segment s-1) -- what falls off the segment above,
//     known before the insertion -- so the segments need one cross-lane move per insertion and no chain;
//   * every candidate that passed the filter stays in the query's LDS log as (value, index).  The log is compacted
//     in place against the current KS-th best value whenever it runs out of room (entries strictly above it, at most
//     KS-1, plus as many equal ones as the list itself holds), and once more at the end against the (k+2)-th best
//     value: what is left ARE the k+1 neighbours.
//
// C == 64: v_mfma_f32_32x32x2_f32 with candidates as MFMA rows and queries as MFMA columns: lanes l and l+32 own one
//          query column and 16 candidate rows each (cross-lane: v_permlane32_swap).  The k order of the MFMA chain is
//          the natural one (step s multiplies k = 2s, 2s+1) and -sq_j/2 rides along as a 33rd k-step: together with
//          the pointwise kernel's reference-ordered features and norms the distance matrix is BIT-IDENTICAL to the
//          reference's (CPU sgemm = k-ascending fma chain; verified), so the feature-space neighbour sets never flip.
//          S waves of a workgroup may share a query tile and split the candidate tiles (k <= 20; S = 2 puts two waves
//          on every SIMD at BASELINE configs[1]); their value lists and logs are folded at the end.
// C == 4 : Cartesian xyz4 rows on the VALU: four lanes (one DPP quad) per query, 16 queries per wave, each lane
//          scanning every fourth candidate (cross-lane: DPP quad_perm).
#include "common.h"

namespace {

constexpr int TILE = 32;            // candidates per MFMA tile
// log entries per query: room for the KS-1 entries a compaction can leave, the <= 16 a step adds, and slack so that
// compactions stay rare
// Log capacity per query.  Measured for the MFMA kernel at k = 20 (MI355X, 32 clouds): 96 / 128 entries make the kernel
// alone 7 % faster at N = 1024 (fewer compactions) but cost the second workgroup per CU at N = 2048 (+20 %) and the
// co-residency of the one-launch kNN pair (+20 %): 64 stays.
constexpr int KNN_MAX_N = 131072;                       // points per cloud the kNN entry points accept: the largest size that is validated
                                                         // (sampled rows at N = 70 001 and 131 072, tests/test_hip_kernels.py; beyond: VCR_EUNSUPPORTED
                                                         // rather than an unvalidated result).  The whole forward keeps its own, lower limit: forward.hip
static bool knn_rows_overflow(const vcr_knn_args* a) { return (long long)a->B * a->N >= (1ll << 31); }   // row = b * N + q is an int
constexpr int KNN_PEND_MFMA = 64;
struct GeomMfma;
struct GeomCol16;
// 16-query kernels (k <= 20): 72 -- the most that keeps four workgroups per CU (4 x (76 rows x 16 queries x 8 B x 4 waves
// + the tie list) = 157 KB of the 160): a compaction then frees 35 slots instead of 27.  Measured against 64: the pair
// launch 151.4 -> 147.7 us at BASELINE configs[1], 147.5 -> 136.7 at N = 768 (configs[2]), 439 -> 432 at N = 2048.
constexpr int KNN_ORD_NEAR = 4;                          // tiles on either side of the own one scanned first
constexpr int KNN_ORD_MAX_TILES = 512;                   // the ordered search's wave-uniform tile mask: clouds of up to 8192 points
constexpr int KNN_ORD_MODE = 0;                          // timing ablations of the ordered search: 1 = its loop over ALL tiles, 2 = the plain loop over the ranked rows
constexpr int KNN_PEND_K40 = 96;                         // k = 21 .. 40 (lists of 42)
constexpr int KNN_PEND_COL16 = 72;                       // (the sweeps: profiles/experiments/probe_build.py --set NAME=VALUE)
template <class G, int KS> constexpr int pend_of() {
  return KS > 22 ? KNN_PEND_K40 : std::is_same<G, GeomMfma>::value ? KNN_PEND_MFMA : std::is_same<G, GeomCol16>::value ? KNN_PEND_COL16 : 64;
}

// "//@probe ..." lines: inert here, uncommented by profiles/experiments/probe_build.py (phase clocks of wave 0).

// In-kernel tie replay (vcr_knn_args.tie_inline, set by the host when a row's replay image fits the workgroup's LDS): the
// rows of a workgroup whose (k+1)-th and (k+2)-th values tie are listed in LDS and replayed by that workgroup itself
// once its four waves have written their results -- the separate, latency-bound replay launch (36 us at BASELINE
// configs[1] for a handful of rows) disappears; only the few workgroups that own a tied row run ~20 us longer.
constexpr int BLK_TIES = 64;                             // a 64-query workgroup cannot list more
constexpr int TB_LDS_PAD = 8;                            // see tiebreak_row (8: N = 10 091 still fits 160 KB)
__host__ __device__ constexpr size_t tiebreak_lds(int N) { return TB_LDS_PAD + (size_t)N * 16 + 256 + (16 + 2 * 256 + 2) * 4; }
// the list sits behind whichever is larger, the waves' logs or the replay's LDS image of a row
__host__ __device__ constexpr size_t inline_tie_offset(size_t log_bytes, int N) {
  return ((log_bytes > tiebreak_lds(N) ? log_bytes : tiebreak_lds(N)) + 15) & ~(size_t)15;
}
constexpr size_t INLINE_TIE_MAX_LDS = 40 * 1024;         // four workgroups per CU must still fit
__device__ void tiebreak_row(const vcr_knn_args& a, int row, unsigned char* smem, unsigned char* gwork);
// gwork (tie_inline == 2): the row image lives in THIS workgroup's 16 N-byte slot of vcr_knn_args.tie_work instead of LDS --
// rows too long for an LDS image beside three or four resident workgroups (N > ~2400) are then replayed by the workgroup
// that found them too, under the other workgroups' scans, instead of by a separate launch (0.16 ms at 64 x 4096, k = 40).
__device__ __forceinline__ void replay_block_ties(const vcr_knn_args& a, int* blk_ties, unsigned char* smem, unsigned char* gwork = nullptr) {
  __syncthreads();                                       // every wave is done with its log: the LDS is free
  const int n = min(blk_ties[0], BLK_TIES);
  int rows[4];                                           // (the list itself lies behind the replay's LDS image)
  for (int t0 = 0; t0 < n; t0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) rows[u] = t0 + u < n ? blk_ties[1 + t0 + u] : -1;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (rows[u] >= 0) tiebreak_row(a, rows[u], smem, gwork);
  }
}

// Row whose (k+1)-th and (k+2)-th best values are equal: hand it to knn_tiebreak_kernel (ties[0] = count).
__device__ __forceinline__ void report_tie(int32_t* ties, int cap, int row) {
  if (!ties) return;
  const int pos = atomicAdd(&ties[0], 1);
  if (pos < cap) ties[1 + pos] = row;
}

// ---- lane geometry of a query's lanes.  Every cross-lane move is issued with all lanes active and only its RESULT
// is selected per lane (DPP / permlane reads of switched-off lanes return 0).
struct GeomMfma {                    // 32 query columns, lanes l and l+32 share one: segment = lane >> 5
  static constexpr int COLS = 32, LPQ = 2;
  static constexpr bool SPLIT_COMPACT = false;
  __device__ static __forceinline__ int ord(int) { return 0; }
  __device__ static __forceinline__ int prefix(int x, int, int& total) { total = x; return 0; }
  __device__ static __forceinline__ int col(int lane) { return lane & 31; }
  __device__ static __forceinline__ int seg(int lane) { return lane >> 5; }
  // x of segment `which` (0 / 1), in every lane of the column (v_permlane32_swap: result 0 = the lower half's values
  // in both halves, result 1 = the upper half's)
  __device__ static __forceinline__ int from_seg(int x, int which) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return which ? r[1] : r[0];
  }
  __device__ static __forceinline__ int from_prev(int x, int, int) { return from_seg(x, 0); }   // only segment 1 has a predecessor
  __device__ static __forceinline__ int prev_addr(int) { return 0; }
  __device__ static __forceinline__ int col_sum(int x, int sg) { return x + from_seg(x, sg ^ 1); }
};
struct GeomCol16 {                   // v_mfma_f32_16x16x4_f32 layout: 16 query columns, lanes c, c+16, c+32, c+48 share one.
  // The value list of a query runs through its four lanes in the row order 0 -> 1 -> 3 -> 2 (seg 0..3), chosen so that
  // every segment's predecessor is ONE row swap away: v_permlane16_swap exchanges rows (0,1) and (2,3),
  // v_permlane32_swap rows (0,2) and (1,3).  With both operands = x, swap16 returns {even row of the pair, odd row of
  // the pair} in every lane of the pair, swap32 {row of the lower half, row of the upper half} in both halves.
  static constexpr int COLS = 16, LPQ = 4;
  __device__ static __forceinline__ int col(int lane) { return lane & 15; }
  __device__ static __forceinline__ int seg(int lane) { const int q = lane >> 4; return q ^ (q >> 1); }   // 0,1,3,2
  __device__ static __forceinline__ int from_seg(int x, int which) {      // `which` is wave-uniform
    const int q = which ^ (which >> 1);                                   // the row that holds segment `which`
    const auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    const int v = (q & 1) ? a[1] : a[0];
    const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (q >> 1) ? b[1] : b[0];
  }
  // segment sg - 1's value (sg == 0: unused): row 1 <- row 0; row 3 <- row 1; row 2 <- row 3.  ONE ds_bpermute_b32 on the
  // otherwise idle LDS crossbar instead of both row swaps, their operand copies and the selects (9 VALU instructions of
  // the 17 an insertion cost -- the drains are bound by VALU issue, four waves per SIMD cover the longer latency).
  // (The same exchange for the per-tile / per-compaction-round prefixes was measured and is slower: those chains are
  // short and wait for the crossbar.)
  __device__ static __forceinline__ int prev_addr(int lane) {
    const int q = lane >> 4, pq = q == 1 ? 0 : q == 3 ? 1 : q == 2 ? 3 : 0;
    return 4 * (16 * pq + (lane & 15));
  }
  __device__ static __forceinline__ int from_prev(int x, int, int pa) { return __builtin_amdgcn_ds_bpermute(pa, x); }
  __device__ static __forceinline__ int col_sum(int x, int sg) {
    const int q = sg ^ (sg >> 1);
    const auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    x += (q & 1) ? a[0] : a[1];
    const auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return x + ((q >> 1) ? b[0] : b[1]);
  }
  // exclusive prefix of x over the four lanes of a column in ROW order (row = lane >> 4), and the total
  static constexpr bool SPLIT_COMPACT = true;            // log rows PEND .. PEND + 3 exist (one trash row per lane row)
  __device__ static __forceinline__ int ord(int lane) { return lane >> 4; }
  __device__ static __forceinline__ int prefix(int x, int row, int& total) {
    const auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);      // {even row, odd row} of the pair
    const int pair_total = a[0] + a[1];
    const auto b = __builtin_amdgcn_permlane32_swap(pair_total, pair_total, false, false);   // {rows 0+1, rows 2+3}
    total = b[0] + b[1];
    return ((row & 1) ? a[0] : 0) + ((row >> 1) ? b[0] : 0);
  }
};
struct GeomQuad {                    // 16 queries, one DPP quad each: segment = lane & 3
  static constexpr int COLS = 16, LPQ = 4;
  static constexpr bool SPLIT_COMPACT = false;
  __device__ static __forceinline__ int ord(int) { return 0; }
  __device__ static __forceinline__ int prefix(int x, int, int& total) { total = x; return 0; }
  __device__ static __forceinline__ int col(int lane) { return lane >> 2; }
  __device__ static __forceinline__ int seg(int lane) { return lane & 3; }
  __device__ static __forceinline__ int from_seg(int x, int which) {
    const int a = __builtin_amdgcn_mov_dpp(x, 0x00, 0xF, 0xF, true), b = __builtin_amdgcn_mov_dpp(x, 0x55, 0xF, 0xF, true);
    const int c = __builtin_amdgcn_mov_dpp(x, 0xAA, 0xF, 0xF, true), d = __builtin_amdgcn_mov_dpp(x, 0xFF, 0xF, 0xF, true);
    return which == 0 ? a : which == 1 ? b : which == 2 ? c : d;
  }
  __device__ static __forceinline__ int from_prev(int x, int, int) { return __builtin_amdgcn_mov_dpp(x, 0x90, 0xF, 0xF, true); }
  __device__ static __forceinline__ int prev_addr(int) { return 0; }
  __device__ static __forceinline__ int col_sum(int x, int) {
    x += __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    x += __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    return x;
  }
};
template <class G> __device__ __forceinline__ float gf_from_seg(float x, int which) {
  return __int_as_float(G::from_seg(__float_as_int(x), which));
}

// ---- per-query selection state of one wave: sorted top-KS values in registers (T per lane), (value, index) log in LDS
// v_med3_f32 a, b, (+-inf in an SGPR): see Selector::insert
__device__ __forceinline__ float med3_inf(float a, float b, float inf) {
  float r;
  asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(inf));
  return r;
}
template <class G, int KS>
struct Selector {
  static constexpr int PEND = pend_of<G, KS>();
  static constexpr int T = (KS + G::LPQ - 1) / G::LPQ;   // values per lane; the list holds LPQ*T >= KS values
  static constexpr int TL = (KS - 1) / T, TS = (KS - 1) % T;   // segment / slot of rank KS-1: the filter threshold
  float v[T];
  float* lv; int* li;                                    // log [PEND + 1][COLS]; row PEND swallows the writes of lanes
                                                         // that have nothing to log (branch-free appends)
  int cnt, done;                                         // entries logged / already inserted (same in a query's lanes)
  float thr;                                             // max(thr0, rank KS-1 value): nothing <= thr can be a neighbour
  float thr0;                                            // filter floor taken from a sample of the candidates (see sample_floor)
  int col, sg, pa;

  __device__ __forceinline__ void init(float* lv_, int* li_, int lane, float floor0 = VCR_NEG_INF) {
    lv = lv_; li = li_; cnt = 0; done = 0; thr = thr0 = floor0; col = G::col(lane); sg = G::seg(lane); pa = G::prev_addr(lane);
#pragma unroll
    for (int t = 0; t < T; ++t) v[t] = VCR_NEG_INF;
  }
  // one value into the query's list, all segments at once (inserting -inf or anything <= the last value is a no-op)
  __device__ __forceinline__ void insert(float d) {
    const float pb = __int_as_float(G::from_prev(__float_as_int(v[T - 1]), sg, pa));
    // (v_med3_f32 with an infinite third operand: min / max in ONE instruction.  Spelled as inline assembly: hipcc folds
    // the builtin with an infinite constant back into v_min / v_max plus a NaN-quieting v_max x, x per operand -- four
    // instructions for the clamp, three for the head of the list)
    d = med3_inf(d, sg ? pb : __builtin_huge_valf(), VCR_NEG_INF);       // min(d, predecessor's last); segment 0 has none
#pragma unroll
    for (int t = T - 1; t >= 1; --t) v[t] = __builtin_amdgcn_fmed3f(v[t - 1], d, v[t]);
    v[0] = med3_inf(v[0], d, __builtin_huge_valf());
  }
  __device__ __forceinline__ void refresh_thr() {
    const float mine = v[TS];
    thr = fmaxf(thr0, gf_from_seg<G>(mine, TL));
  }
  // The floor came from a sample: it is only valid if at least KS candidates lie above it.  False -> the list is not
  // full although a floor was used: the caller scans again without one.
  __device__ __forceinline__ bool floor_held() const {
    const float last = gf_from_seg<G>(v[TS], TL);
    return !(thr0 > VCR_NEG_INF) || last > VCR_NEG_INF;
  }
  // value at global rank r (wave-uniform r) in every lane of the column
  __device__ __forceinline__ float rank_value(int r) const {
    const int rs = r / T, rt = r % T;
    int bits = 0;                                        // (an OR of masked words: a select chain over v[] would be
#pragma unroll                                           // turned into a dynamically indexed scratch array)
    for (int t = 0; t < T; ++t) bits |= (rt == t ? -1 : 0) & __float_as_int(v[t]);
    return gf_from_seg<G>(__int_as_float(bits), rs);
  }
  // insert the values logged since the last drain.  Four log reads are in flight per round trip: the loop is bound by
  // LDS latency, not by the 1-med3-per-slot network.
  __device__ __forceinline__ void drain() {
    int i = done;
    while (__any(i < cnt)) {
      float d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) d[u] = lv[min(i + u, PEND - 1) * G::COLS + col];
#pragma unroll
      for (int u = 0; u < 4; ++u) insert(i + u < cnt ? d[u] : VCR_NEG_INF);      // idle lanes insert -inf: a no-op
      i += 4;
    }
    done = cnt;
    refresh_thr();
  }
  // keep the log entries above x, plus at most `emax` equal to x (the earliest logged); cnt = done = kept
  __device__ __forceinline__ void compact(float x, int emax) {
    if constexpr (G::SPLIT_COMPACT) {
      // the four lanes of a column take one entry each per round (the loop below has every lane walk all four: four
      // times the LDS instructions): keep flags and write positions come from prefixes over the lanes in row order, so
      // the kept entries stay in logging order and the "at most emax equal to x, the earliest" rule is unchanged
      int w = 0, ne = 0;
      const int od = G::ord((int)__lane_id());            // (recomputed here: a register less across the scan)
      for (int i = 0; __any(i < cnt); i += 4) {
        const int ii = i + od;
        const bool valid = ii < cnt;
        const int ic = min(ii, PEND - 1);
        const float d = lv[ic * G::COLS + col];
        const int j = li[ic * G::COLS + col];
        const bool gt = valid && d > x, eq = valid && d == x;
        // ONE prefix for both counts (packed: entries above x in the low half, entries equal to x in the high half); of
        // the equal ones the first `cap` still wanted are kept, so their kept-prefix is min(prefix, cap)
        int tot;
        const int pre = G::prefix((gt ? 1 : 0) | (eq ? 0x10000 : 0), od, tot);
        const int cap = max(emax - ne, 0), epre = pre >> 16, etot = tot >> 16;
        const bool keep = gt || (eq && epre < cap);
        const int kpre = (pre & 0xffff) + min(epre, cap);
        const int wr = keep ? w + kpre : PEND + od;      // w + kpre <= i + od: in place; reads of the round precede its writes
        lv[wr * G::COLS + col] = d;
        li[wr * G::COLS + col] = j;
        w += (tot & 0xffff) + min(etot, cap);
        ne += min(etot, cap);
      }
      cnt = done = w;
      return;
    }
    int w = 0, ne = 0;
    for (int i = 0; __any(i < cnt); i += 4) {
      float d[4];
      int j[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {                      // all four entries are in registers before any is rewritten
        const int ii = min(i + u, PEND - 1);
        d[u] = lv[ii * G::COLS + col];
        j[u] = li[ii * G::COLS + col];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool eq = d[u] == x && ne < emax;
        const bool keep = i + u < cnt && (d[u] > x || eq);
        const int wr = keep ? w : PEND;                  // w <= i + u: in place (the lanes of a column write the same words)
        lv[wr * G::COLS + col] = d[u];
        li[wr * G::COLS + col] = j[u];
        w += keep ? 1 : 0;
        ne += (keep && eq) ? 1 : 0;
      }
    }
    cnt = done = w;
  }
  __device__ __forceinline__ int count_above(float x) const {
    int c = 0;
#pragma unroll
    for (int t = 0; t < T; ++t) c += v[t] > x ? 1 : 0;
    return G::col_sum(c, sg);
  }
  // make room for the next step (<= 16 new entries per query)
  __device__ __forceinline__ void make_room() {
    if (__any(cnt > PEND - 16)) {
      drain();
      compact(thr, KS - count_above(thr));
    }
  }
};

// Filter floor from a SAMPLE of the candidates.  A streaming top-k logs k (1 + ln(n / k)) candidates per query because
// its threshold starts at -inf; most of those are entries the first few hundred candidates push through a list that
// later ones empty again.  A values-only pre-pass (R v_med3 per candidate) over a lane's share of the first 256
// candidates keeps its R best; the smallest of the lanes' R-th values is a floor with at least LPQ * R - 1 sample
// values strictly above it, and R is chosen so that this is >= KS: the floor is below the final KS-th best value by
// construction, for ANY ordering of the cloud.  The scan proper then starts with a useful threshold: at N = 1024,
// k = 20 it logs ~55 candidates per query instead of ~125, and the log rarely needs compacting.  (Exact ties AT the
// floor value could still leave fewer than KS values strictly above it; that is checked at the end -- floor_held() --
// and such a wave scans again without a floor.)
constexpr int SAMPLE = 256;
template <int R>
struct SampleNet {
  float s[R];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int t = 0; t < R; ++t) s[t] = VCR_NEG_INF;
  }
  __device__ __forceinline__ void insert(float d) {
#pragma unroll
    for (int t = R - 1; t >= 1; --t) s[t] = __builtin_amdgcn_fmed3f(s[t - 1], d, s[t]);
    s[0] = med3_inf(s[0], d, __builtin_huge_valf());
  }
};
template <class G> __device__ __forceinline__ float col_min(float x, int sg) {   // min over the lanes of a query
  float m = x;
#pragma unroll
  for (int w = 0; w < G::LPQ; ++w) m = fminf(m, gf_from_seg<G>(x, w));
  (void)sg;
  return m;
}

// Final stage shared by both kernels: the log holds every candidate above the (k+2)-th best value; fold the lists /
// logs of the S waves of a query group into wave part 0, reduce the log to the k+1 best, drop rank 0, write the set.
// perm (ordered search): q and the logged indices are RANKS of the cloud's Morton order; the kept entries are translated to
// point indices before the rank-0 rule and the output, which goes to the row of the point at rank q
template <class G, int KS, int S>
__device__ __forceinline__ void finish(Selector<G, KS>& sel, const vcr_knn_args& a, int b, int q, int wave, int part,
                                       unsigned char* smem, int* blk_ties = nullptr, const int32_t* perm = nullptr) {
  constexpr int T = Selector<G, KS>::T;
  constexpr int PEND = Selector<G, KS>::PEND;
  constexpr int AREA = 2 * (PEND + 1) * G::COLS;         // floats per wave
  sel.drain();
  if (S > 1) {
    // every wave first shrinks its log to its own top-KS and parks its sorted values behind it (KS <= 22: 22 + 24 <= 64)
    sel.compact(sel.thr, KS - sel.count_above(sel.thr));
#pragma unroll
    for (int t = 0; t < T; ++t) sel.lv[(PEND - G::LPQ * T + sel.sg * T + t) * G::COLS + sel.col] = sel.v[t];
    if (sel.sg == 0) sel.li[(PEND - 1) * G::COLS + sel.col] = sel.cnt;
    __syncthreads();
    if (part == 0) {
      for (int p = 1; p < S; ++p) {
        const float* ov = reinterpret_cast<const float*>(smem) + (size_t)(wave + p) * AREA;
        for (int t = 0; t < KS; ++t) sel.insert(ov[(PEND - G::LPQ * T + t) * G::COLS + sel.col]);
      }
      sel.refresh_thr();
    }
  }
  if (part != 0) return;
  const float vk = sel.rank_value(a.k), vk1 = sel.rank_value(a.k + 1);       // ranks k+1 and k+2 (KS >= k+2)
  const int need = a.k + 1 - sel.count_above(vk1);      // neighbours that EQUAL the (k+2)-th value: 0 unless tied
  sel.compact(vk1, need);
  if (S > 1) {                                           // append the other waves' qualifying entries
    int ne = 0;
    for (int i = 0; i < sel.cnt; ++i) ne += sel.lv[i * G::COLS + sel.col] == vk1 ? 1 : 0;
    for (int p = 1; p < S; ++p) {
      const float* ov = reinterpret_cast<const float*>(smem) + (size_t)(wave + p) * AREA;
      const int* oi = reinterpret_cast<const int*>(ov + (PEND + 1) * G::COLS);
      const int oc = oi[(PEND - 1) * G::COLS + sel.col];
      for (int i = 0; __any(i < oc); ++i) {
        const int ii = min(i, PEND - 1);
        const float d = ov[ii * G::COLS + sel.col];
        const int j = oi[ii * G::COLS + sel.col];
        const bool eq = d == vk1 && ne < need;
        const bool keep = i < oc && (d > vk1 || eq) && sel.cnt < PEND;
        if (keep) { sel.lv[sel.cnt * G::COLS + sel.col] = d; sel.li[sel.cnt * G::COLS + sel.col] = j; }
        sel.cnt += keep ? 1 : 0;
        ne += (keep && eq) ? 1 : 0;
      }
    }
  }
  // rank 0 = the largest value (the point itself): dropped.  When that value is SHARED (duplicate points, or a neighbour so
  // close that its distance rounds to the point's own), WHICH of the tied entries Tensor.topk returns first is an outcome of
  // its sort (util.py:159 then drops that one and keeps the others): such a row is replayed like a boundary tie (best_shared
  // below; tiebreak_row sorts the kept entries the way ATen does).  Without tie_scratch: the first logged is dropped.
  // (Whether it is shared is read off the sorted value list: its two best entries are equal.)
  const bool best_shared = sel.rank_value(0) == sel.rank_value(1);
  int imax = 0;
  float vmax = VCR_NEG_INF;
  if (perm) {
    // (the plain scan logs in index order, so "the first logged" is the LOWEST point index among the largest values)
    for (int i = sel.sg; i < sel.cnt; i += G::LPQ) sel.li[i * G::COLS + sel.col] = perm[sel.li[i * G::COLS + sel.col]];
    int jmax = 0x7fffffff;
    for (int i = 0; __any(i < sel.cnt); ++i) {
      const int ic = min(i, PEND - 1);
      const float d = i < sel.cnt ? sel.lv[ic * G::COLS + sel.col] : VCR_NEG_INF;
      const int j = sel.li[ic * G::COLS + sel.col];
      if (d > vmax || (d == vmax && i < sel.cnt && j < jmax)) { vmax = d; imax = i; jmax = j; }
    }
  } else {
  for (int i = 0; __any(i < sel.cnt); ++i) {
    const float d = i < sel.cnt ? sel.lv[min(i, PEND - 1) * G::COLS + sel.col] : VCR_NEG_INF;
    if (d > vmax) { vmax = d; imax = i; }
  }
  }
  if (q < a.N) {
    if (perm) q = perm[q];
    int32_t* o = a.idx + ((size_t)b * a.N + q) * a.k;
    for (int i = sel.sg; i < sel.cnt && i <= a.k; i += G::LPQ)
      if (i != imax) o[i - (i > imax ? 1 : 0)] = sel.li[i * G::COLS + sel.col];
    if (sel.sg == 0 && ((vk1 == vk && vk1 > VCR_NEG_INF) || best_shared)) {
      if (blk_ties) {                                    // replayed by this very workgroup (replay_block_ties)
        const int pos = atomicAdd(&blk_ties[0], 1);
        if (pos < BLK_TIES) blk_ties[1 + pos] = b * a.N + q;
      } else {
        report_tie(a.tie_scratch, a.tie_cap, b * a.N + q);
      }
    }
  }
}

// ---------------------------------------------------------------- C == 64 (MFMA)
// Workgroup = W waves = W/S query tiles of 32 queries; wave (qt, part) scans candidate tiles part, part+S, ...
// (W = 4, or 2 for k > 20 whose longer logs would otherwise leave one workgroup per CU)
template <int KS, int S, int W>
__device__ __forceinline__ void knn64_body(const vcr_knn_args& a, int bx, int b) {   // block bx of cloud b
  using G = GeomMfma;
  constexpr int PEND = pend_of<G, KS>();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int qt = wave / S, part = wave % S;
  const int q0 = (bx * (W / S) + qt) * 32;       // this wave's 32 queries (may lie beyond N: clamped, not written)
  float* lv = reinterpret_cast<float*>(smem) + wave * (2 * (PEND + 1) * 32);
  Selector<G, KS> sel;

  const float* xb = a.x + (size_t)b * a.N * a.ldx;
  const float* sqb = a.sq + (size_t)b * a.N;
  const int q = min(q0 + col, a.N - 1);
  // query fragment: this lane supplies B[k][col] with k = 2s + half for MFMA step s, i.e. the NATURAL k
  // order: the MFMA result is then bit-for-bit the k-ascending fma chain that the reference's CPU sgemm
  // produces (verified against torch.matmul), and with the exact |x|^2 association of the pointwise kernel
  // the whole distance matrix -- hence every top-k set -- equals the reference's.
  // Loads stay 16 B wide: both lanes of a row fetch the whole row and each keeps its parity.
  auto pick = [&](const f32x4* raw, float* dst) {
#pragma unroll
    for (int st = 0; st < 32; ++st) dst[st] = half ? raw[st >> 1][(st & 1) * 2 + 1] : raw[st >> 1][(st & 1) * 2];
  };
  float qf[32];
  {
    f32x4 raw[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) raw[m] = ld4(xb + (size_t)q * a.ldx + 4 * m);
    pick(raw, qf);
  }
  const float sq_q = sqb[q];

  const int ntiles = (a.N + TILE - 1) / TILE;
  //@probe VCR_PROBE_ACC_DECL;
  // walk candidate tiles t0, t0 + S, ... < t1: operands prefetched one tile ahead as raw rows (64 VGPRs), the MFMA
  // chain, then body(tile, acc)
  auto scan_tiles = [&](int t0, int t1, auto&& body) {
    float cf[32];
    float csq = 0.f;
    if (t0 < t1) {
      const int c = min(t0 * TILE + col, a.N - 1);
      f32x4 raw[16];
#pragma unroll
      for (int m = 0; m < 16; ++m) raw[m] = ld4(xb + (size_t)c * a.ldx + 4 * m);
      pick(raw, cf);
      csq = sqb[c];
    }
    for (int tile = t0; tile < t1; tile += S) {
      f32x4 nraw[16];
      float nsq = 0.f;
      if (tile + S < t1) {
        const int c = min((tile + S) * TILE + col, a.N - 1);
#pragma unroll
        for (int m = 0; m < 16; ++m) nraw[m] = ld4(xb + (size_t)c * a.ldx + 4 * m);
        nsq = sqb[c];
      }
      f32x16 acc = {0};
#pragma unroll
      for (int st = 0; st < 32; ++st) acc = mfma32(cf[st], qf[st], acc);
      // 33rd k-step: A[cand][k*] = -sq_cand/2 (half 0), B[k*][q] = 1  ->  acc = dot - sq_j/2, rounded once
      acc = mfma32(half == 0 ? -0.5f * csq : 0.f, half == 0 ? 1.f : 0.f, acc);
      // hipcc (ROCm 7.2) under-pads the MFMA -> v_accvgpr_read hazard of this 16-pass instruction when the
      // accumulator lands in AGPRs (seen in the k = 40 build: register 15, the last one written, was read stale).
      // Tie the wait states to the accumulator itself so they cannot be scheduled away.
      if (KS > 22) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc));
      //@probe VCR_PROBE_ACC(0);                                            // prefetch issue + MFMA chain
      body(tile, acc);
      if (tile + S < t1) {
        pick(nraw, cf);
        csq = nsq;
      }
      //@probe VCR_PROBE_ACC(3);                                            // operand pick (waits for the prefetched rows)
    }
  };
  // the selection proper: filter against sel.thr, log, drain
  auto select_body = [&](int tile, const f32x16& acc) {
    const int jbase = tile * TILE;
    const bool ragged = jbase + TILE > a.N;              // only the last tile can hold rows beyond N
    // two steps of 16 rows per query (8 per lane): the log has room for 16 new entries, never for 32
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
      sel.make_room();
      //@probe VCR_PROBE_ACC(2);
      float dd[8];
      unsigned m = 0;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        dd[r] = 2.f * acc[8 * hs + r] - sq_q;            // (-sq_j + 2 dot) - sq_i
        m |= dd[r] > sel.thr ? (1u << r) : 0u;
      }
      if (ragged) {
#pragma unroll
        for (int r = 0; r < 8; ++r) m &= (jbase + acc_row(8 * hs + r, half) < a.N) ? ~0u : ~(1u << r);
      }
      const unsigned om = (unsigned)G::from_seg((int)m, half ^ 1);
      //@probe VCR_PROBE_ACC(1);
      if (__any(m != 0)) {                               // the two lanes of a column append to ONE log: upper half first
        const int base = sel.cnt + (half ? __popc(om) : 0);
#pragma unroll
        for (int r = 0; r < 8; ++r) {                    // branch-free: a lane without a survivor in row r writes the trash row
          const int pos = (m & (1u << r)) ? base + __popc(m & ((1u << r) - 1u)) : PEND;
          sel.lv[pos * 32 + col] = dd[r];
          sel.li[pos * 32 + col] = jbase + acc_row(8 * hs + r, half);
        }
      }
      sel.cnt += __popc(m) + __popc(om);
      //@probe VCR_PROBE_ACC(4);
      if (__any(sel.cnt - sel.done > 16)) sel.drain();   // keep the threshold fresh
      //@probe VCR_PROBE_ACC(5);
    }
  };
  // sample pre-pass over this wave's first SAMPLE candidates (full tiles only), when it has at least twice as many
  constexpr int T0 = SAMPLE / TILE, R0 = (KS + 1 + G::LPQ - 1) / G::LPQ;   // LPQ * R0 - 1 >= KS
  const int my_tiles = part < ntiles ? (ntiles - part + S - 1) / S : 0;
  float floor0 = VCR_NEG_INF;
  // Measured (profiles/timeline_knn.py, N = 1024, k = 20): the floor cuts make_room 28 -> 13 us and the drains
  // 12.5 -> 8 us per wave, and the pre-pass -- eight more tiles of loads + 33 MFMAs + pick -- costs the same 17 us back;
  // at N = 2048 it loses 5 %.  Distances are too expensive here to compute a quarter of them twice: off in this kernel
  // (the Cartesian kernel, whose distances are three FMAs, keeps it: 75 -> 65 us).
  constexpr bool SAMPLE_FLOOR = false;
  if (SAMPLE_FLOOR && my_tiles >= 2 * T0 + 1) {          // (+1: the last, possibly ragged, tile is never sampled)
    SampleNet<R0> net;
    net.init();
    scan_tiles(part, part + S * T0, [&](int, const f32x16& acc) {
#pragma unroll
      for (int r = 0; r < 16; ++r) net.insert(2.f * acc[r] - sq_q);
    });
    floor0 = col_min<G>(net.s[R0 - 1], half);
  }
  sel.init(lv, reinterpret_cast<int*>(lv + (PEND + 1) * 32), lane, floor0);
  scan_tiles(part, ntiles, select_body);
  sel.drain();
  if (__any(!sel.floor_held())) {                        // the sample misjudged some query of this wave: scan without a floor
    sel.init(lv, reinterpret_cast<int*>(lv + (PEND + 1) * 32), lane);
    scan_tiles(part, ntiles, select_body);
  }
  //@probe VCR_PROBE_ACC_FLUSH(threadIdx.x == 0 && blockIdx.y == 0, blockIdx.x);
  finish<G, KS, S>(sel, a, b, q0 + col, wave, part, smem);
}
template <int KS, int S, int W>
__global__ __launch_bounds__(64 * W, 2) void knn64_kernel(vcr_knn_args a) {
  int bx, b;
  xcd_chunk2(bx, b);                                     // a cloud's query tiles share one XCD's L2
  knn64_body<KS, S, W>(a, bx, b);
}

// ---------------------------------------------------------------- C == 64 on v_mfma_f32_16x16x4_f32 (round 3)
// Half-size waves: 16 queries per wave (MFMA columns), candidate tiles of 16 (MFMA rows), four lanes per query.  Twice as
// many, half as wide waves as knn64_body with the SAME selection work per query (one value list, one log), ~100 VGPRs and
// 8 KB of log per wave: four waves per SIMD instead of one or two, so one wave's 17-MFMA distance chain and its LDS /
// scalar-branch latencies run under the other waves' selection code (rocprofv3 counters of the 32-query kernel in the
// one-launch pair: VALU issuing in 30 % of its wave-cycles, 26 % parked at a waitcnt, 32 % waiting to issue).
// k order: MFMA step s multiplies k = 4s .. 4s+3 (lane row q supplies k = 4s + q), i.e. the natural ascending order, so
// the distance is bit-for-bit the same k-ascending fma chain as in knn64_body and in the reference's CPU sgemm; the
// -sq_j/2 term rides as a 17th step.  Operands are read from the natural [N][64] rows as 16-B chunks and transposed
// across the four lane rows of a column in registers (see load_raw / transpose).
// C == 4 (the Cartesian search, rows (x, y, z, |p|^2)): the whole distance is ONE MFMA -- lane row q4 < 3 supplies
// coordinate q4 of its candidate row (A) and of its query (B), lane row 3 supplies -|c|^2 / 2 and 1: the hardware's
// k-ascending chain is ((x x' + y y') + z z') - |c|^2 / 2, exactly the fma chain + norm step of the C = 64 case and of the
// VALU kernel knn3_body (whose (2 dot - |c|^2) - |q|^2 is the same rounding: doubling is exact).  The Cartesian
// search's distances then cost the idle matrix pipe one instruction per 16 x 16 tile instead of 16 VALU FMAs per wave.
template <int KS, int W, int C = 64, bool XT = false, bool ORD = false>
__device__ __forceinline__ void knn64c_body(const vcr_knn_args& a, int bx, int b) {
  using G = GeomCol16;
  static_assert(C == 64 || C == 4, "feature rows of 64 floats or xyz4 rows");
  static_assert(!ORD || C == 4 || XT, "the ranked rows are stored in xt's layout");
  constexpr int NST = C == 64 ? 16 : 1;                  // MFMA steps of the dot product
  constexpr int PEND = pend_of<G, KS>();
  constexpr int CT = 16;                                 // candidates per tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q4 = lane >> 4, col = lane & 15;
  const int q0 = (bx * W + wave) * 16;
  // log [PEND + 4][16]: rows PEND .. PEND+3 swallow the branch-free appends of lanes without a survivor, one row per
  // lane row q4 (the four lanes of a column would otherwise hit one LDS word with four different values)
  constexpr int LROWS = PEND + 4;
  float* lv = reinterpret_cast<float*>(smem) + wave * (2 * LROWS * 16);
  Selector<G, KS> sel;
  // (ORD: the rows in rank order -- vcr_knn_args.xp / sqp; everything below then counts ranks, finish() translates)
  // (an ORD launch may still run ONE of its two searches the plain way -- or single clouds of one: vcr_knn_args.ord_ok)
  const bool ord = ORD && a.perm != nullptr && (C != 64 || !a.ord_ok || a.ord_ok[b] != 0);
  const float* xb = (ord ? a.xp : a.x) + (size_t)b * a.N * a.ldx;
  // (C == 64) rows whose 16-channel groups are stored transposed (vcr_knn_args.xt): lane row q4's chunks 4 g + q4 are its
  // operands of steps 4 g .. 4 g + 3 as they lie -- same addresses, no shuffles
  constexpr bool pre = C == 64 && XT;                  // (a compile-time variant: the launcher picks it when a.xt is set)
  const float* xl = ord ? xb : pre ? a.xt + (size_t)b * a.N * a.ldx : xb;
  const float* sqb = C == 64 ? (ord ? a.sqp : a.sq) + (size_t)b * a.N : nullptr;
  const int q = min(q0 + col, a.N - 1);
  // Operand fetch (C == 64).  MFMA step s needs x[row][4 s + q4] in lane row q4: every fourth float of the row.  Fetched as such
  // (16 x global_load_dword) the texture path sees 4-byte requests -- 8x the requests of knn64_body per byte, and the
  // kernel is bound by them (measured: 145 of a wave's 208 us at N = 1024).  Instead lane row q4 loads the 16-B chunks
  // 4 g + q4 (g = 0..3; four global_load_dwordx4) and a 4 x 4 transpose between lane rows and vector components --
  // v_permlane16_swap on the register pairs (0,1) (2,3), then v_permlane32_swap on (0,2) (1,3) -- leaves
  // component j of chunk register g = x[row][4 (4 g + j) + q4], i.e. the operand of step s = 4 g + j.
  // (C == 4: one float per lane -- element q4 of the xyz4 row; raw[0][0] carries it, raw[0][1] the row's |p|^2)
  auto load_raw = [&](int row, f32x4* raw) {
    if constexpr (C == 64) {
      const float* rp = xl + (size_t)row * a.ldx + 4 * q4;
#pragma unroll
      for (int g = 0; g < 4; ++g) raw[g] = ld4(rp + 16 * g);
    } else {
      const float* rp = xb + (size_t)row * a.ldx;
      raw[0][0] = rp[q4];
    }
  };
  auto swap16 = [](float& x, float& y) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(x), __float_as_int(y), false, false);
    x = __int_as_float(r[0]); y = __int_as_float(r[1]);
  };
  auto swap32 = [](float& x, float& y) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(x), __float_as_int(y), false, false);
    x = __int_as_float(r[0]); y = __int_as_float(r[1]);
  };
  auto transpose = [&](const f32x4* raw, float* dst) {   // dst[4 g + j] = operand of MFMA step 4 g + j
    if constexpr (C == 4) {
      dst[0] = q4 == 3 ? -0.5f * raw[0][0] : raw[0][0];  // candidate side: (x, y, z, -|c|^2 / 2)
      return;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float r0 = raw[g][0], r1 = raw[g][1], r2 = raw[g][2], r3 = raw[g][3];
      if constexpr (!pre) {
        swap16(r0, r1); swap16(r2, r3);
        swap32(r0, r2); swap32(r1, r3);
      }
      dst[4 * g] = r0; dst[4 * g + 1] = r1; dst[4 * g + 2] = r2; dst[4 * g + 3] = r3;
    }
  };
  float qf[NST];
  {
    f32x4 raw[4];
    load_raw(q, raw);
    transpose(raw, qf);
    if constexpr (C == 4) qf[0] = q4 == 3 ? 1.f : raw[0][0];              // query side: (x, y, z, 1)
  }
  const float sq_q = C == 64 ? sqb[q] : xb[(size_t)q * a.ldx + 3];
  const int ntiles = (a.N + CT - 1) / CT;

  sel.init(lv, reinterpret_cast<int*>(lv + LROWS * 16), lane);
  // (tie_inline 2: the replay's row image is in global scratch, only its query row + block scratch need LDS)
  int* blk_ties = a.tie_inline ? reinterpret_cast<int*>(smem + inline_tie_offset((size_t)W * 2 * LROWS * 16 * 4, (KS > 22 && a.tie_inline == 2) ? 0 : a.N)) : nullptr;
  if (blk_ties) {
    if (threadIdx.x == 0) blk_ties[0] = 0;
    __syncthreads();
  }
  //@probe VCR_PROBE_ACC_DECL;
  // Two candidate tiles per step: their two 17-MFMA chains are independent and issue alternately (a dependent
  // v_mfma_f32_16x16x4_f32 chain leaves 8 of every 40 cycles empty), and the next two tiles' rows are in flight meanwhile.
  float cf[2][NST];
  f32x4 nraw[2][C == 64 ? 4 : 1];
  float csq[2] = {0.f, 0.f}, nsq[2] = {0.f, 0.f};
  auto scan_prologue = [&]() {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = min(u * CT + col, a.N - 1);
      load_raw(c, nraw[u]);
      if constexpr (C == 64) csq[u] = sqb[c];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      transpose(nraw[u], cf[u]);                         // (VALU consumers: the first tiles have arrived before the loop)
      asm volatile("" : "+v"(csq[u]));
    }
  };
  // the selection proper for one tile: 16 new distances per query (4 per lane: candidate rows 4 q4 + r)
  auto select_tile = [&](int tile, const f32x4& acc) {
    sel.make_room();
    //@probe VCR_PROBE_ACC(2);
    const int jbase = tile * CT + 4 * q4;
    float dd[4];
    unsigned m = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dd[r] = fmaf(2.f, acc[r], -sq_q);                  // (-sq_j + 2 dot) - sq_i: 2 x is exact, so one fma rounds like mul + sub
      m |= (dd[r] > sel.thr && jbase + r < a.N) ? (1u << r) : 0u;
    }
    // the four lanes of a column append to ONE log, in row order 0, 1, 2, 3: exclusive prefix of their survivor counts
    const int c0 = __popc(m);
    const auto pa = __builtin_amdgcn_permlane16_swap(c0, c0, false, false);      // {even row, odd row} of the pair
    const int pair_total = pa[0] + pa[1];
    const auto pb = __builtin_amdgcn_permlane32_swap(pair_total, pair_total, false, false);   // {rows 0+1, rows 2+3}
    const int pre = ((q4 & 1) ? pa[0] : 0) + ((q4 >> 1) ? pb[0] : 0);
    //@probe VCR_PROBE_ACC(1);                                              // distances + filter + prefix
    if (__any(m != 0)) {
      const int base = sel.cnt + pre;
#pragma unroll
      for (int r = 0; r < 4; ++r) {                      // branch-free: a lane without a survivor in row r writes its trash row
        const int pos = (m & (1u << r)) ? base + __popc(m & ((1u << r) - 1u)) : PEND + q4;
        sel.lv[pos * 16 + col] = dd[r];
        sel.li[pos * 16 + col] = jbase + r;
      }
    }
    sel.cnt += pb[0] + pb[1];
    //@probe VCR_PROBE_ACC(4);                                              // log push
    if (__any(sel.cnt - sel.done > 16)) sel.drain();     // keep the threshold fresh
    //@probe VCR_PROBE_ACC(5);
  };
  auto scan_all = [&]() {
  scan_prologue();
  for (int tile = 0; tile < ntiles; tile += 2) {
    const bool more = tile + 2 < ntiles;
    if (more) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {                      // (a tile beyond the last one re-reads the last row: masked in select_tile)
        const int c = min((tile + 2 + u) * CT + col, a.N - 1);
        load_raw(c, nraw[u]);
        if constexpr (C == 64) nsq[u] = sqb[c];
      }
    }
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      acc[0] = mfma16(cf[0][st], qf[st], acc[0]);
      acc[1] = mfma16(cf[1][st], qf[st], acc[1]);
    }
    // 17th k-step: A[cand][k*] = -sq_cand/2 (row 0 of the lanes), B[k*][q] = 1  ->  acc = dot - sq_j/2, rounded once
    // (C == 4: the norm is the 4th k of the one MFMA above)
    if constexpr (C == 64) {
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = mfma16(q4 == 0 ? -0.5f * csq[u] : 0.f, q4 == 0 ? 1.f : 0.f, acc[u]);
    }
    //@probe VCR_PROBE_ACC(0);                                              // prefetch issue + MFMA chains
    select_tile(tile, acc[0]);
    if (tile + 1 < ntiles) select_tile(tile + 1, acc[1]);
    if (more) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        transpose(nraw[u], cf[u]);                       // the wait for the prefetch sits HERE
        csq[u] = nsq[u];
        asm volatile("" : "+v"(csq[u]));
      }
    }
    //@probe VCR_PROBE_ACC(3);                                              // wait for the prefetched rows + transpose
  }
  };
  // ---- ORD: the ordered search.  Tiles = 16 consecutive RANKS of the cloud's Morton order (compact in coordinate AND feature
  // space).  (1) the 2 NEAR + 1 tiles around the wave's own rank are scanned: the list then holds a valid lower bound thr of every
  // query's final (k + 2)-th value.  (2) the tiles' centroids go through the same MFMA chain as candidates: s = -|q - c|^2; with
  // the tile's radius rho, no row of the tile can score above -(max(0, |q - c| - rho))^2 -- evaluated with the fp32 rounding of
  // the scores priced in (m: 3e-5 of the norms involved, ~4x the worst case of a 64-term fp32 dot product), so a tile is dropped
  // only when every one of its rows would fail the filter `score > thr` for every query of the wave.  (3) the remaining tiles
  // are scanned like any others.  Visiting order and omissions of entries below the final threshold do not change the kept set.
  [[maybe_unused]] auto ordered_scan_impl = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr int NEAR = KNN_ORD_NEAR;
    const int T = ntiles, g = q0 / CT;
    int lo = max(0, g - NEAR), hi = min(T, g + NEAR + 1);
    if (hi - lo < 2 * NEAR + 1) { if (lo == 0) hi = min(T, 2 * NEAR + 1); else lo = max(0, T - (2 * NEAR + 1)); }
    // the two-tile step of scan_all over an arbitrary tile sequence (next() = the next tile, -1 at the end; wave-uniform)
    auto scan_seq = [&](auto&& next) {
      int tA = next(), tB = tA >= 0 ? next() : -1;
      if (tA < 0) return;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c = min((u ? max(tB, 0) : tA) * CT + col, a.N - 1);
        load_raw(c, nraw[u]);
        if constexpr (C == 64) csq[u] = sqb[c];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        transpose(nraw[u], cf[u]);
        asm volatile("" : "+v"(csq[u]));
      }
      while (tA >= 0) {
        const int nA = next(), nB = nA >= 0 ? next() : -1;
        if (nA >= 0) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int c = min((u ? max(nB, 0) : nA) * CT + col, a.N - 1);
            load_raw(c, nraw[u]);
            if constexpr (C == 64) nsq[u] = sqb[c];
          }
        }
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int st = 0; st < NST; ++st) {
          acc[0] = mfma16(cf[0][st], qf[st], acc[0]);
          acc[1] = mfma16(cf[1][st], qf[st], acc[1]);
        }
        if constexpr (C == 64) {
#pragma unroll
          for (int u = 0; u < 2; ++u) acc[u] = mfma16(q4 == 0 ? -0.5f * csq[u] : 0.f, q4 == 0 ? 1.f : 0.f, acc[u]);
        }
        select_tile(tA, acc[0]);
        if (tB >= 0) select_tile(tB, acc[1]);
        if (nA >= 0) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            transpose(nraw[u], cf[u]);
            csq[u] = nsq[u];
            asm volatile("" : "+v"(csq[u]));
          }
        }
        tA = nA; tB = nB;
      }
    };
    if constexpr (FULL) { lo = 0; hi = T; }
    if constexpr (FULL) {
      int t = lo;
      scan_seq([&]() { return t < hi ? t++ : -1; });
    } else {
      // own tile first, then outwards (g + 1, g - 1, g + 2, ...): along the curve the tiles get closer as the scan approaches
      // the wave's rank -- in rank order every candidate would beat the threshold the previous ones left (a streaming top-k's
      // worst case: the plain loop over ranked rows takes 1.9x its time over unranked ones)
      int i = 0;
      const int gc = min(max(g, lo), hi - 1);
      scan_seq([&]() {
        while (i < 2 * (2 * NEAR + 1) + 2) {
          const int d = (i + 1) >> 1, t = (i & 1) ? gc + d : gc - d;
          ++i;
          if (t >= lo && t < hi && !(d == 0 && (i & 1) == 0)) return t;
        }
        return -1;
      });
    }
    if (FULL || hi - lo >= T) return;
    sel.drain();                                         // thr = the (k + 2)-th best of the near candidates, exactly
    constexpr int NEEDW = KNN_ORD_MAX_TILES / 64;                          // tiles still to visit: one bit each (T <= 512)
    unsigned long long need[NEEDW];
#pragma unroll
    for (int i = 0; i < NEEDW; ++i) need[i] = 0ull;
    const float* cen = a.cen + (size_t)b * T * a.ldx;
    const float* crad = a.cen_rad + (size_t)b * T;
    const float* cmax = a.cen_sqmax + (size_t)b * T;
    // (two centroid tiles in flight: fetched one iteration ahead, unconditionally -- past the last one the last again --, so that
    //  the loop is not a chain of exposed memory round trips; 64 x 4096, k = 40: 1846 -> 1795 us)
    struct CenTile { f32x4 raw[C == 64 ? 4 : 1]; float cs, rd[4], mx[4]; };
    auto cfetch = [&](int ct, CenTile& f) {
      const int crow = min(ct * CT + col, T - 1);
      if constexpr (C == 64) {
        const float* rp = cen + (size_t)crow * a.ldx + 4 * q4;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) f.raw[gq] = ld4(rp + 16 * gq);
        f.cs = a.cen_sq[(size_t)b * T + crow];
      } else {
        f.raw[0][0] = cen[(size_t)crow * a.ldx + q4];
        f.cs = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tc = min(ct * CT + 4 * q4 + r, T - 1);
        f.rd[r] = crad[tc];
        f.mx[r] = cmax[tc];
      }
    };
    auto ceval = [&](int ct, const CenTile& f) {
      float ccf[NST];
      transpose(f.raw, ccf);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < NST; ++st) acc = mfma16(ccf[st], qf[st], acc);
      if constexpr (C == 64) acc = mfma16(q4 == 0 ? -0.5f * f.cs : 0.f, q4 == 0 ? 1.f : 0.f, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = ct * CT + 4 * q4 + r;
        const float sc = fmaf(2.f, acc[r], -sq_q);       // -|q - centroid|^2 as the kernel computes scores
        const float m = 3e-5f * (sq_q + f.mx[r]) + 1e-30f;
        const float dl = __builtin_sqrtf(fmaxf(0.f, -sc - m)) * 0.999999f;
        const float lbd = fmaxf(0.f, dl - f.rd[r]);
        const float ub = m - lbd * lbd;                  // no row of tile t scores above ub for this query
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(t < T && !(ub < sel.thr));
#pragma unroll
        for (int qg = 0; qg < 4; ++qg)
          if ((bal >> (16 * qg)) & 0xffffull) {
            const int tt = ct * CT + 4 * qg + r;
            need[tt >> 6] |= 1ull << (tt & 63);
          }
      }
    };
    if constexpr (KS > 22) {
      const int nct = (T + CT - 1) / CT;
      CenTile fa, fb;
      cfetch(0, fa);
      for (int ct = 0; ct < nct; ct += 2) {
        cfetch(min(ct + 1, nct - 1), fb);
        ceval(ct, fa);
        cfetch(min(ct + 2, nct - 1), fa);
        if (ct + 1 < nct) ceval(ct + 1, fb);
      }
    } else {
      // (k <= 20: ONE centroid tile at a time -- the second one in flight costs 25 registers, and with them the fourth
      //  workgroup per CU; the lists of 22 are what makes 128 registers possible at all)
      const int nct = (T + CT - 1) / CT;
      CenTile fa;
      for (int ct = 0; ct < nct; ++ct) {
        cfetch(ct, fa);
        ceval(ct, fa);
      }
    }
    for (int t = lo; t < hi; ++t) need[t >> 6] &= ~(1ull << (t & 63));
    {
      // (ascending rank order; nearest-first on either side of the near range measured 10-14 % slower: the two tiles of a step
      // then lie far apart and the scalar bookkeeping per step grows)
      int w = 0;
      unsigned long long cur = need[0];
      scan_seq([&]() {
        while (cur == 0ull && w < NEEDW - 1) cur = need[++w];
        if (cur == 0ull) return -1;
        const int bit = __builtin_ctzll(cur);
        cur &= cur - 1ull;
        return w * 64 + bit;
      });
    }
  };
  [[maybe_unused]] auto ordered_scan = [&]() { ordered_scan_impl(std::false_type{}); };
  [[maybe_unused]] auto ordered_scan_full = [&]() { ordered_scan_impl(std::true_type{}); };
  // C == 4: filter floor from a sample (see SampleNet): the first 256 candidates' values only -- 16 MFMAs -- give a
  // threshold the scan proper starts with (the VALU kernel: 75 -> 65 us; distances are one MFMA per tile here, so the
  // pre-pass costs next to nothing.  C == 64 recomputes 17 MFMAs per sampled tile: measured a wash, off).
  if constexpr (C == 4) {
    constexpr int T0 = SAMPLE / CT, R0 = (KS + 1 + G::LPQ - 1) / G::LPQ;     // LPQ * R0 - 1 >= KS
    if (!ord && ntiles >= 2 * T0 + 1) {
      SampleNet<R0> net;
      net.init();
      for (int t = 0; t < T0; ++t) {                     // (full tiles: 16 T0 <= N)
        const float cv = xb[(size_t)(t * CT + col) * a.ldx + q4];
        const f32x4 acc = mfma16(q4 == 3 ? -0.5f * cv : cv, qf[0], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int r = 0; r < 4; ++r) net.insert(fmaf(2.f, acc[r], -sq_q));
      }
      sel.init(lv, reinterpret_cast<int*>(lv + LROWS * 16), lane, col_min<G>(net.s[R0 - 1], sel.sg));
    }
  }
  if constexpr (ORD) {
    if (!ord) scan_all();
    else if constexpr (KNN_ORD_MODE == 0) ordered_scan();
    else if constexpr (KNN_ORD_MODE == 1) ordered_scan_full();
    else scan_all();
  } else {
    scan_all();
  }
  if constexpr (C == 4) {
    sel.drain();
    if (!ord && __any(!sel.floor_held())) {                      // the sample misjudged some query of this wave: scan without a floor
      sel.init(lv, reinterpret_cast<int*>(lv + LROWS * 16), lane);
      scan_all();
    }
  }
  //@probe VCR_PROBE_ACC_FLUSH(threadIdx.x == 0 && blockIdx.y == 0, blockIdx.x);
  finish<G, KS, 1>(sel, a, b, q0 + col, wave, 0, smem, blk_ties, ord ? a.perm + (size_t)b * a.N : nullptr);
  if (blk_ties) {
    // (the slot form exists in the k > 20 kernels only: they run three workgroups per CU and have the registers for it; with it
    //  the k <= 20 kernels, held to 128 registers for their fourth workgroup, spill 32 of them -- 124 -> 142 us at configs[1])
    unsigned char* gwork = nullptr;
    if constexpr (KS > 22) {
      if (a.tie_inline == 2)                               // this workgroup's slot: (cloud, query block) in launch-independent order
        gwork = reinterpret_cast<unsigned char*>(a.tie_work) + ((size_t)b * ((a.N + 16 * W - 1) / (16 * W)) + bx) * 16 * (size_t)a.N;
    }
    replay_block_ties(a, blk_ties, smem, gwork);
  }
}
// (k > 20: the logs of a workgroup take 51 KB, so three workgroups share a CU whatever the registers allow -- the bound
//  says so and the lists of 42 keep their registers: at four waves per SIMD, 128 VGPRs, two spilled to scratch)
template <int KS, int W, bool XT>
__global__ __launch_bounds__(64 * W, (KS > 22 ? 3 : 4)) void knn64c_kernel(vcr_knn_args a) {
  int bx, b;
  xcd_chunk2(bx, b);
  knn64c_body<KS, W, 64, XT>(a, bx, b);
}
// the Cartesian search on the same body (distances = one MFMA per tile): the unsplit (S = 1) kernel of C == 4
template <int KS, int W>
__global__ __launch_bounds__(64 * W, (KS > 22 ? 3 : 4)) void knn3c_kernel(vcr_knn_args a) {
  int bx, b;
  xcd_chunk2(bx, b);
  knn64c_body<KS, W, 4>(a, bx, b);
}

// ---------------------------------------------------------------- C == 4 (xyz4, VALU)
// Wave = 16 queries x 4 lanes (DPP quad = query); lane s of the quad computes the distances of candidates j = 4u + s.
// S waves of a workgroup may split the candidates of a query group (small grids, k <= 20).
template <int KS, int S>
__device__ __forceinline__ void knn3_body(const vcr_knn_args& a, int bx, int b) {
  using G = GeomQuad;
  constexpr int PEND = pend_of<G, KS>();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = lane & 3, qd = lane >> 2;
  const int grp = wave / S, part = wave % S;
  const int q0 = (bx * (4 / S) + grp) * 16;
  float* lv = reinterpret_cast<float*>(smem) + wave * (2 * (PEND + 1) * 16);
  Selector<G, KS> sel;
  const float* xb = a.x + (size_t)b * a.N * a.ldx;
  const int qi = q0 + qd;
  const f32x4 qv = ld4(xb + (size_t)min(qi, a.N - 1) * a.ldx);
  // 16 candidates per step, 4 per lane: j = j0 + 4u + s (the quad reads 64 contiguous bytes per load); the next
  // step's rows are in flight while this one is filtered
  const int nsteps = (a.N + 15) / 16;
  //@probe VCR_PROBE_ACC_DECL;
  auto scan_steps = [&](int s0, int s1, auto&& body) {
    f32x4 c[4], cn[4];
    auto load = [&](f32x4* dst, int st) {
#pragma unroll
      for (int u = 0; u < 4; ++u) dst[u] = ld4(xb + (size_t)min(st * 16 + 4 * u + s, a.N - 1) * a.ldx);
    };
    if (s0 < s1) load(c, s0);
    int it = 0;
    for (int st = s0; st < s1; st += S, ++it) {
      if (st + S < s1) load(cn, st + S);
      float dd[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float dot = fmaf(qv[2], c[u][2], fmaf(qv[1], c[u][1], qv[0] * c[u][0]));
        dd[u] = (2.f * dot - c[u][3]) - qv[3];
      }
      body(st, it, dd);
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = cn[u];
    }
  };
  auto select_body = [&](int st, int it, const float (&dd)[4]) {
    const int j0 = st * 16;
    sel.make_room();
    unsigned m = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) m |= (dd[u] > sel.thr && j0 + 4 * u + s < a.N) ? (1u << u) : 0u;
    //@probe VCR_PROBE_ACC(0);                                              // distances + filter
    // the quad appends to ONE log: exclusive prefix of the lanes' survivor counts
    const int c0 = __popc(m);
    const int p1 = __builtin_amdgcn_mov_dpp(c0, 0x90, 0xF, 0xF, true);     // lane s <- lane s-1 (lane 0: itself)
    int inc = c0 + (s >= 1 ? p1 : 0);
    const int p2 = __builtin_amdgcn_mov_dpp(inc, 0x44, 0xF, 0xF, true);    // lane s <- lane s-2 (lanes 0, 1: themselves)
    inc += s >= 2 ? p2 : 0;
    const int total = G::from_seg(inc, 3);
    if (__any(m != 0)) {
      const int base = sel.cnt + inc - c0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {                      // branch-free: lanes without a survivor write the trash row
        const int pos = (m & (1u << u)) ? base + __popc(m & ((1u << u) - 1u)) : PEND;
        sel.lv[pos * 16 + qd] = dd[u];
        sel.li[pos * 16 + qd] = j0 + 4 * u + s;
      }
    }
    sel.cnt += total;
    if ((it < 3 && !(sel.thr0 > VCR_NEG_INF)) || __any(sel.cnt - sel.done > 12)) sel.drain();   // no floor: settle the threshold quickly
    //@probe VCR_PROBE_ACC(1);                                              // log + drains
  };
  constexpr int ST0 = SAMPLE / 16, R0 = (KS + 1 + G::LPQ - 1) / G::LPQ;    // LPQ * R0 - 1 >= KS
  const int my_steps = part < nsteps ? (nsteps - part + S - 1) / S : 0;
  float floor0 = VCR_NEG_INF;
  if (my_steps >= 2 * ST0 + 1) {
    SampleNet<R0> net;
    net.init();
    scan_steps(part, part + S * ST0, [&](int, int, const float (&dd)[4]) {
#pragma unroll
      for (int u = 0; u < 4; ++u) net.insert(dd[u]);
    });
    floor0 = col_min<G>(net.s[R0 - 1], s);
  }
  int* blk_ties = (S == 1 && a.tie_inline) ? reinterpret_cast<int*>(smem + inline_tie_offset((size_t)4 * 2 * (PEND + 1) * 16 * 4, a.N)) : nullptr;
  if (blk_ties) {
    if (threadIdx.x == 0) blk_ties[0] = 0;
    __syncthreads();
  }
  sel.init(lv, reinterpret_cast<int*>(lv + (PEND + 1) * 16), lane, floor0);
  scan_steps(part, nsteps, select_body);
  sel.drain();
  if (__any(!sel.floor_held())) {
    sel.init(lv, reinterpret_cast<int*>(lv + (PEND + 1) * 16), lane);
    scan_steps(part, nsteps, select_body);
  }
  //@probe VCR_PROBE_ACC_FLUSH(threadIdx.x == 0 && blockIdx.y == 0, blockIdx.x);
  finish<G, KS, S>(sel, a, b, qi, wave, part, smem, blk_ties);
  if (blk_ties) replay_block_ties(a, blk_ties, smem);
}
template <int KS, int S>
__global__ __launch_bounds__(256, 2) void knn3_kernel(vcr_knn_args a) {
  int bx, b;
  xcd_chunk2(bx, b);
  knn3_body<KS, S>(a, bx, b);
}

// Both kNN graphs of an LPDNet pass in ONE launch (lpdnet_model.py:113,129 -- the feature-space and the Cartesian search
// are independent): the first n64 workgroups run the MFMA kernel's body, the rest the Cartesian one.  Either kernel
// alone is latency-bound at one wave per SIMD (1024 waves on 1024 SIMDs); launched together the second fills the
// first one's idle issue slots, and the pair costs little more than the longer of the two.
template <int KS, bool COL16, bool XT = false, bool ORD = false>
__global__ __launch_bounds__(256, (COL16 ? (KS > 22 || ORD ? 3 : 4) : 2)) void knn_pair_kernel(vcr_knn_args a64, vcr_knn_args a3, int n64, int gx64, int gx3) {
  const int bid = (int)blockIdx.x;
  if (bid < n64) {
    const int lin = xcd_chunk(bid, n64);
    if constexpr (COL16) knn64c_body<KS, 4, 64, XT, ORD>(a64, lin % gx64, lin / gx64);
    else knn64_body<KS, 1, 4>(a64, lin % gx64, lin / gx64);
  } else {
    const int lin = xcd_chunk(bid - n64, (int)gridDim.x - n64);
    knn64c_body<KS, 4, 4, false, ORD>(a3, lin % gx3, lin / gx3);
  }
}

// The same for SMALL grids (a few pairs per call), where both searches run their candidate-split kernels (S64 / S3 waves
// share a group of queries): launched one after the other each is a latency-bound handful of workgroups (63 + 44 us at one
// pair); together they take as long as the longer one.
template <int KS, int S64, int S3>
__global__ __launch_bounds__(256, 2) void knn_pair_small_kernel(vcr_knn_args a64, vcr_knn_args a3, int n64, int gx64, int gx3) {
  const int bid = (int)blockIdx.x;
  if (bid < n64) {
    const int lin = xcd_chunk(bid, n64);
    knn64_body<KS, S64, 4>(a64, lin % gx64, lin / gx64);
  } else {
    const int lin = xcd_chunk(bid - n64, (int)gridDim.x - n64);
    if constexpr (S3 == 1) knn64c_body<KS, 4, 4>(a3, lin % gx3, lin / gx3);   // (unsplit: the MFMA body, as vcr_knn_f32 launches it)
    else knn3_body<KS, S3>(a3, lin % gx3, lin / gx3);
  }
}

// ---------------------------------------------------------------- exact replica of Tensor.topk's tie-breaking
// Sequential port of libstdc++'s std::nth_element (__introselect: median-of-three to first, unguarded partition,
// depth limit 2 log2 n with __heap_select fallback, final insertion sort) and of std::partial_sort's __heap_select,
// on (value, index) pairs ordered by VALUE ONLY, exactly as ATen's CPU topk runs them (TopKImpl: queue[j] = (x[j], j);
// partial_sort when k*64 <= n, else nth_element(k-1) + sort of the first k-1).  Only the SET of the first K entries
// matters here.  One thread per tied row; rows are rare.
struct PairArr {
  float* v; int* id;
  __device__ __forceinline__ bool gt(int a, int b) const { return v[a] > v[b]; }
  __device__ __forceinline__ void swap(int a, int b) {
    const float tv = v[a]; v[a] = v[b]; v[b] = tv;
    const int ti = id[a]; id[a] = id[b]; id[b] = ti;
  }
};

__device__ void tb_push_heap(PairArr& q, int first, int hole, int top, float val, int vid) {
  int parent = (hole - 1) / 2;
  while (hole > top && q.v[first + parent] > val) {
    q.v[first + hole] = q.v[first + parent]; q.id[first + hole] = q.id[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  q.v[first + hole] = val; q.id[first + hole] = vid;
}

__device__ void tb_adjust_heap(PairArr& q, int first, int hole, int len, float val, int vid) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (q.v[first + child] > q.v[first + child - 1]) --child;
    q.v[first + hole] = q.v[first + child]; q.id[first + hole] = q.id[first + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    q.v[first + hole] = q.v[first + child - 1]; q.id[first + hole] = q.id[first + child - 1];
    hole = child - 1;
  }
  tb_push_heap(q, first, hole, top, val, vid);
}

__device__ void tb_heap_select(PairArr& q, int first, int middle, int last) {
  const int len = middle - first;
  if (len >= 2) {                                        // std::__make_heap
    for (int parent = (len - 2) / 2;; --parent) {
      tb_adjust_heap(q, first, parent, len, q.v[first + parent], q.id[first + parent]);
      if (parent == 0) break;
    }
  }
  for (int i = middle; i < last; ++i) {
    if (q.v[i] > q.v[first]) {                           // std::__pop_heap(first, middle, i)
      const float val = q.v[i]; const int vid = q.id[i];
      q.v[i] = q.v[first]; q.id[i] = q.id[first];
      tb_adjust_heap(q, first, 0, len, val, vid);
    }
  }
}

__device__ void tb_nth_element(PairArr& q, int first, int last, int nth, int depth) {
  while (last - first > 3) {
    if (depth == 0) {
      tb_heap_select(q, first, nth + 1, last);
      q.swap(first, nth);
      return;
    }
    --depth;
    // __unguarded_partition_pivot: median of (first+1, mid, last-1) to first, then partition [first+1, last)
    const int mid = first + (last - first) / 2, a = first + 1, b = mid, c = last - 1;
    if (q.gt(a, b)) {
      if (q.gt(b, c)) q.swap(first, b);
      else if (q.gt(a, c)) q.swap(first, c);
      else q.swap(first, a);
    } else if (q.gt(a, c)) q.swap(first, a);
    else if (q.gt(b, c)) q.swap(first, c);
    else q.swap(first, b);
    int lo = first + 1, hi = last;
    for (;;) {
      while (q.gt(lo, first)) ++lo;
      --hi;
      while (q.gt(first, hi)) --hi;
      if (!(lo < hi)) break;
      q.swap(lo, hi);
      ++lo;
    }
    if (lo <= nth) first = lo; else last = lo;
  }
  for (int i = first + 1; i < last; ++i) {               // std::__insertion_sort(first, last)
    const float val = q.v[i]; const int vid = q.id[i];
    if (val > q.v[first]) {
      for (int j = i; j > first; --j) { q.v[j] = q.v[j - 1]; q.id[j] = q.id[j - 1]; }
      q.v[first] = val; q.id[first] = vid;
    } else {
      int j = i;
      while (val > q.v[j - 1]) { q.v[j] = q.v[j - 1]; q.id[j] = q.id[j - 1]; --j; }
      q.v[j] = val; q.id[j] = vid;
    }
  }
}

// std::__sort_heap(first, first + len): what std::partial_sort runs on its heap, and std::sort when its depth limit runs out
__device__ void tb_sort_heap(PairArr& q, int first, int len) {
  while (len > 1) {
    --len;                                               // std::__pop_heap(first, last, last)
    const float val = q.v[first + len]; const int vid = q.id[first + len];
    q.v[first + len] = q.v[first]; q.id[first + len] = q.id[first];
    tb_adjust_heap(q, first, 0, len, val, vid);
  }
}

__device__ void tb_unguarded_linear_insert(PairArr& q, int last) {
  const float val = q.v[last]; const int vid = q.id[last];
  int next = last - 1;
  while (val > q.v[next]) { q.v[last] = q.v[next]; q.id[last] = q.id[next]; last = next; --next; }
  q.v[last] = val; q.id[last] = vid;
}
__device__ void tb_insertion_sort(PairArr& q, int first, int last) {
  for (int i = first + 1; i < last; ++i) {
    if (q.v[i] > q.v[first]) {
      const float val = q.v[i]; const int vid = q.id[i];
      for (int j = i; j > first; --j) { q.v[j] = q.v[j - 1]; q.id[j] = q.id[j - 1]; }
      q.v[first] = val; q.id[first] = vid;
    } else {
      tb_unguarded_linear_insert(q, i);
    }
  }
}
// Sequential port of libstdc++'s std::sort on [first, last) (__introsort_loop: median-of-three to first + unguarded partition
// while a range is longer than 16, depth limit 2 log2 n with the heap sort fallback; then __final_insertion_sort) with the
// value-only comparator: the ORDER it leaves equal values in is what decides Tensor.topk's rank 0 among tied best values.
// Ranges here are the <= 62 kept entries of a row.  The recursion on the right-hand parts: only a part of more than 16 entries
// has work left, the parts are disjoint (the order they are finished in does not matter) -- at most three are ever pending,
// kept packed (first | last << 8 | depth << 16) in the caller's LDS scratch (stk[0..2]).
__device__ void tb_sort(PairArr& q, int first, int last, int* stk) {
  if (last - first < 2) return;
  int depth0 = 0;
  for (int m = last - first; m > 1; m >>= 1) ++depth0;
  depth0 *= 2;
  int sp = 1;
  stk[0] = first | (last << 8) | (depth0 << 16);
  while (sp > 0) {
    --sp;
    const int e = stk[sp];
    int f = e & 255, l = (e >> 8) & 255, depth = e >> 16;
    while (l - f > 16) {
      if (depth == 0) {                                  // std::__partial_sort(f, l, l): heap sort of the range
        tb_heap_select(q, f, l, l);
        tb_sort_heap(q, f, l - f);
        break;
      }
      --depth;
      const int mid = f + (l - f) / 2, a = f + 1, b = mid, c = l - 1;
      if (q.gt(a, b)) {
        if (q.gt(b, c)) q.swap(f, b);
        else if (q.gt(a, c)) q.swap(f, c);
        else q.swap(f, a);
      } else if (q.gt(a, c)) q.swap(f, a);
      else if (q.gt(b, c)) q.swap(f, c);
      else q.swap(f, b);
      int lo = f + 1, hi = l;
      for (;;) {
        while (q.gt(lo, f)) ++lo;
        --hi;
        while (q.gt(f, hi)) --hi;
        if (!(lo < hi)) break;
        q.swap(lo, hi);
        ++lo;
      }
      if (l - lo > 16 && sp < 3) {                       // __introsort_loop(cut, last, depth_limit): later
        stk[sp++] = lo | (l << 8) | (depth << 16);
      }
      l = lo;
    }
  }
  if (last - first > 16) {                               // std::__final_insertion_sort
    tb_insertion_sort(q, first, first + 16);
    for (int i = first + 16; i < last; ++i) tb_unguarded_linear_insert(q, i);
  } else {
    tb_insertion_sort(q, first, last);
  }
}

// std::partial_sort's __heap_select(first = 0, middle = K, last = n) with the K-entry heap held ACROSS THE LANES of one
// wave (lane j = heap[j]; K <= 64): every heap access is a v_readlane / v_writelane with a scalar index instead of a
// dependent LDS round trip, and the scan over the n - K remaining values tests 64 of them per step.  Same compares,
// same moves as libstdc++ (__make_heap, then __pop_heap for every v[i] > heap[0]); the values evicted to positions
// >= K are not written back: nothing reads them again.  Returns with (hv, hid) = the kept set in lanes 0..K-1.
struct LaneHeap {
  float hv; int hid; int lane;
  __device__ __forceinline__ float val(int i) const {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hv), __builtin_amdgcn_readfirstlane(i)));
  }
  __device__ __forceinline__ int idx(int i) const {
    return __builtin_amdgcn_readlane(hid, __builtin_amdgcn_readfirstlane(i));
  }
  __device__ __forceinline__ void set(int i, float v, int id) {
    const int si = __builtin_amdgcn_readfirstlane(i);   // (v, id) are wave-uniform: a lane-select is a writelane
    hv = lane == si ? v : hv;
    hid = lane == si ? id : hid;
  }
  __device__ void push(int hole, int top, float v, int id) {          // std::__push_heap
    int parent = (hole - 1) / 2;
    while (hole > top && val(parent) > v) {
      set(hole, val(parent), idx(parent));
      hole = parent;
      parent = (hole - 1) / 2;
    }
    set(hole, v, id);
  }
  __device__ void adjust(int hole, int len, float v, int id) {        // std::__adjust_heap
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
      child = 2 * (child + 1);
      if (val(child) > val(child - 1)) --child;
      set(hole, val(child), idx(child));
      hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
      child = 2 * (child + 1);
      set(hole, val(child - 1), idx(child - 1));
      hole = child - 1;
    }
    push(hole, top, v, id);
  }
};

__device__ void tb_heap_select_wave(const float* v, int n, int K, LaneHeap& h, int lane) {
  h.hv = lane < K ? v[lane] : VCR_NEG_INF;
  h.hid = lane;
  h.lane = lane;
  if (K >= 2) {
    for (int parent = (K - 2) / 2;; --parent) {
      h.adjust(parent, K, h.val(parent), h.idx(parent));
      if (parent == 0) break;
    }
  }
  float top = h.val(0);
  for (int base = K; base < n; base += 64) {
    const int x = base + lane;
    const float c = x < n ? v[x] : VCR_NEG_INF;
    unsigned long long mask = __builtin_amdgcn_ballot_w64(c > top);
    while (mask) {
      const int i = __builtin_ctzll(mask);
      const float cv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), __builtin_amdgcn_readfirstlane(i)));
      h.adjust(0, K, cv, base + i);                      // __pop_heap: the candidate replaces the root
      top = h.val(0);
      mask = __builtin_amdgcn_ballot_w64(c > top) & ~((2ull << i) - 1ull);
    }
  }
}

// One __unguarded_partition_pivot pass of introselect on [first, last), run by the whole block with the SAME result
// as the sequential loop.  With pivot p = v[first] after the median-of-three, the left scan stops at the elements
// <= p and the right scan at the elements >= p, in order: if A lists the positions > first with v <= p (ascending)
// and Bd the positions > first with v >= p (descending), the loop swaps A[i] <-> Bd[i] while A[i] < Bd[i] (m swaps)
// and returns cut = min(A[m], Bd[m-1]) (A[0] when m = 0).  A / Bd are built by an ordered block compaction.
__device__ int tb_partition_parallel(PairArr& q, int first, int last, int* A, int* Bd, int* red) {
  const int t = threadIdx.x, nt = blockDim.x;
  if (t == 0) {
    const int mid = first + (last - first) / 2, a = first + 1, b = mid, c = last - 1;
    if (q.gt(a, b)) {
      if (q.gt(b, c)) q.swap(first, b);
      else if (q.gt(a, c)) q.swap(first, c);
      else q.swap(first, a);
    } else if (q.gt(a, c)) q.swap(first, a);
    else if (q.gt(b, c)) q.swap(first, c);
    else q.swap(first, b);
  }
  __syncthreads();
  const float pv = q.v[first];
  const int n = last - (first + 1);
  const int per = (n + nt - 1) / nt;
  const int x0 = first + 1 + t * per, x1 = min(last, x0 + per);
  int ca = 0, cb = 0;
  for (int x = x0; x < x1; ++x) { ca += q.v[x] <= pv ? 1 : 0; cb += q.v[x] >= pv ? 1 : 0; }
  // exclusive prefix of ca over ascending threads, exclusive SUFFIX of cb (threads to the right come first in Bd):
  // wave-level shuffles + four wave totals through LDS
  const int lane = t & 63, wv = t >> 6, nwv = nt >> 6;
  int ia = ca, ib = cb;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int ua = __shfl_up(ia, o, 64), ub = __shfl_down(ib, o, 64);
    if (lane >= o) ia += ua;
    if (lane + o < 64) ib += ub;
  }
  if (lane == 63) red[wv] = ia;                          // wave totals
  if (lane == 0) red[8 + wv] = ib;
  __syncthreads();
  int offa = ia - ca, offb = ib - cb, sa = 0, sb = 0;
  for (int i = 0; i < nwv; ++i) {
    if (i < wv) offa += red[i];
    if (i > wv) offb += red[8 + i];
    sa += red[i]; sb += red[8 + i];
  }
  __syncthreads();
  red[16 + t] = offa; red[16 + nt + t] = offb;
  if (t == 0) { red[2 * nt + 16] = sa; red[2 * nt + 17] = sb; }
  __syncthreads();
  const int na = red[2 * nt + 16], nb = red[2 * nt + 17];
  {
    int oa = red[16 + t];
    for (int x = x0; x < x1; ++x) if (q.v[x] <= pv) A[oa++] = x;
    int ob = red[16 + nt + t];                           // descending order: this chunk's elements from the right
    for (int x = x1 - 1; x >= x0; --x) if (q.v[x] >= pv) Bd[ob++] = x;
  }
  __syncthreads();
  const int lim = min(na, nb);
  int mloc = 0;
  for (int i = t; i < lim; i += nt) mloc += A[i] < Bd[i] ? 1 : 0;   // monotone in i: the count is the first failure
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mloc += __shfl_xor(mloc, o, 64);
  if (lane == 0) red[wv] = mloc;
  __syncthreads();
  int m = 0;
  for (int i = 0; i < nwv; ++i) m += red[i];
  int cut;
  if (m == 0) cut = A[0];
  else cut = min(m < na ? A[m] : 0x7fffffff, Bd[m - 1]);
  __syncthreads();                                       // everyone has read A / Bd / red before the swaps reuse LDS
  for (int i = t; i < m; i += nt) q.swap(A[i], Bd[i]);
  __syncthreads();
  return cut;
}

constexpr size_t TB_LDS_MAX = 160 * 1024;
constexpr int TB_BLOCKS = 64;

// One block per tied row: all threads recompute the row's N distances with the SAME arithmetic as the main kernels
// (C == 64: the k-ascending fma chain the MFMA produces, then the -sq_j/2 step, then 2 acc - sq_i; C == 4: the VALU
// expression of knn3_kernel), thread 0 replays the selection and rewrites the row's k indices.
__device__ void tiebreak_row(const vcr_knn_args& a, int row, unsigned char* smem, unsigned char* gwork) {
  float *val, *qrow;
  int *id, *A, *Bd, *red;
  if (!gwork) {
    // (8 bytes of padding in front: the arrays the sequential ports walk downwards must not start at LDS offset 0.  This code
    // reaches LDS through flat instructions wherever val / id may also be global (see below); the compiler turns the `v[j - 1]`
    // of a descending loop into (base - 4) + an immediate offset of 4, and a flat address below the LDS aperture faults
    // whatever the offset -- MEMORY_APERTURE_VIOLATION in the replay launch, located with rocgdb when the rank-0 sort was
    // added.  The ports never index more than one entry below their position.)
    val = reinterpret_cast<float*>(smem) + TB_LDS_PAD / 4;
    id = reinterpret_cast<int*>(val + a.N);
    qrow = reinterpret_cast<float*>(id + a.N);           // [64]
    A = reinterpret_cast<int*>(qrow + 64);               // [N] left stoppers, [N] right stoppers, block scratch
    Bd = A + a.N;
    red = Bd + a.N;                                      // [16 + 2*256 + 2]
  } else {
    // rows too long for an LDS image (N > ~10 100): the four row-sized arrays live in the caller's tie_work, one 16 N-byte
    // slice per block (the host checked that it is there); __syncthreads() orders a block's global accesses as well
    val = reinterpret_cast<float*>(gwork);
    id = reinterpret_cast<int*>(val + a.N);
    A = id + a.N;
    Bd = A + a.N;
    qrow = reinterpret_cast<float*>(smem);
    red = reinterpret_cast<int*>(qrow + 64);
  }
  {
    const int b = row / a.N, qi = row - b * a.N;
    const float* xb = a.x + (size_t)b * a.N * a.ldx;
    __syncthreads();
    if (a.C == 64 && threadIdx.x < 64) qrow[threadIdx.x] = xb[(size_t)qi * a.ldx + threadIdx.x];
    __syncthreads();
    for (int j = threadIdx.x; j < a.N; j += blockDim.x) {
      float d;
      if (a.C == 64) {
        const float* c = xb + (size_t)j * a.ldx;
        f32x4 cr[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) cr[m] = ld4(c + 4 * m);    // the whole row in flight, then the chain
        float acc = 0.f;
#pragma unroll
        for (int kk = 0; kk < 64; ++kk) acc = fmaf(cr[kk >> 2][kk & 3], qrow[kk], acc);
        acc = fmaf(-0.5f * a.sq[(size_t)b * a.N + j], 1.f, acc);
        d = 2.f * acc - a.sq[(size_t)b * a.N + qi];
      } else {
        const f32x4 qv = ld4(xb + (size_t)qi * a.ldx), cv = ld4(xb + (size_t)j * a.ldx);
        const float dot = fmaf(qv[2], cv[2], fmaf(qv[1], cv[1], qv[0] * cv[0]));
        d = (2.f * dot - cv[3]) - qv[3];
      }
      val[j] = d; id[j] = j;
    }
    __syncthreads();
    PairArr q{val, id};
    const int K = a.k + 1;                               // topk(k + 1)
    const bool use_heap = (long)K * 64 <= a.N;
    if (!use_heap) {
      // std::nth_element(K-1): the partition passes over long ranges run on the whole block (see below); the tail
      // (range <= 24, depth exhaustion, final insertion sort) is finished by thread 0 with the sequential port.
      int first = 0, last = a.N, depth = 0;
      for (int m = a.N; m > 1; m >>= 1) ++depth;
      depth *= 2;
      while (last - first > 24 && depth > 0) {
        --depth;
        const int cut = tb_partition_parallel(q, first, last, A, Bd, red);
        if (cut <= K - 1) first = cut; else last = cut;
      }
      if (threadIdx.x == 0) tb_nth_element(q, first, last, K - 1, depth);
    } else if (threadIdx.x < 64) {
      // std::partial_sort branch ((k+1)*64 <= N): heap across the lanes of wave 0
      LaneHeap h;
      const int lane = threadIdx.x;
      tb_heap_select_wave(val, a.N, K, h, lane);
      if (lane < K) { val[lane] = h.hv; id[lane] = h.hid; }           // the kept set, like the sequential port leaves it
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      // rank 0 = the largest of the K kept.  Shared by two or more of them (duplicate points ...): the one Tensor.topk
      // returns FIRST, i.e. position 0 after what ATen does next with the selected entries -- std::sort of the first K - 1
      // (the nth_element branch; the K-th is not above any of them) or partial_sort's __sort_heap of the K-entry heap
      int best = 0, nbest = 1;
      for (int i = 1; i < K; ++i) {
        if (val[i] > val[best]) { best = i; nbest = 1; }
        else if (val[i] == val[best]) ++nbest;
      }
      if (nbest > 1) {
        if (use_heap) tb_sort_heap(q, 0, K); else tb_sort(q, 0, K - 1, red);
        best = 0;
      }
      int32_t* o = a.idx + (size_t)row * a.k;
      int w = 0;
      for (int i = 0; i < K; ++i)
        if (i != best) o[w++] = id[i];
    }
  }
}

__device__ __forceinline__ void tiebreak_body(const vcr_knn_args& a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // rows too long for an LDS image (N > ~10 100): the four row-sized arrays live in the caller's tie_work, one 16 N-byte
  // slice per block (the host checked that it is there); __syncthreads() orders a block's global accesses as well
  unsigned char* gwork = tiebreak_lds(a.N) <= TB_LDS_MAX ? nullptr
                                                         : reinterpret_cast<unsigned char*>(a.tie_work) + (size_t)blockIdx.x * 16 * a.N;
  const int count = min(a.tie_scratch[0], a.tie_cap);
  for (int t = blockIdx.x; t < count; t += gridDim.x) tiebreak_row(a, a.tie_scratch[1 + t], smem, gwork);
}

__global__ __launch_bounds__(256) void knn_tiebreak_kernel(vcr_knn_args a) { tiebreak_body(a); }
// the replays of two kNN launches in one launch (blockIdx.y picks the launch): one latency instead of two
__global__ __launch_bounds__(256) void knn_tiebreak2_kernel(vcr_knn_args a, vcr_knn_args b) {
  if (blockIdx.y == 0) tiebreak_body(a); else tiebreak_body(b);
}

template <auto Kernel, class... Args>
int launch(dim3 grid, dim3 block, size_t lds, hipStream_t s, const Args&... a) {
  VCR_DYN_LDS(Kernel, (int)lds);                         // one cache per kernel: Kernel is a template argument
  hipLaunchKernelGGL(Kernel, grid, block, lds, s, a...);
  return VCR_LAUNCH_RC();
}

}  // namespace

// the tie counter is zeroed by a kernel, not hipMemsetAsync: a memset NODE in a captured HIP graph made replays on the
// default stream hang on this ROCm build (see forward.hip)
__global__ void zero_count_kernel(int32_t* p) { *p = 0; }
static int zero_count(int32_t* p, hipStream_t s) {
  hipLaunchKernelGGL(zero_count_kernel, dim3(1), dim3(1), 0, s, p);
  return VCR_LAUNCH_RC();
}

// LDS of a replay launch: the row image when it fits, else only the query row + block scratch (the image is in tie_work)
static size_t tiebreak_launch_lds(int N) { return tiebreak_lds(N) <= TB_LDS_MAX ? tiebreak_lds(N) : tiebreak_lds(0); }
extern "C" size_t vcr_knn_tie_work_bytes(int N) {
  return (N > 0 && tiebreak_lds(N) > TB_LDS_MAX) ? (size_t)TB_BLOCKS * 16 * (size_t)N : 0;
}
// one 16 N-byte slot per workgroup of the 16-query-wave kernels (64 queries each): with this much tie_work a launch whose rows
// (or lists: k > 20) leave no room for an LDS image replays its tied rows itself (tie_inline 2)
extern "C" size_t vcr_knn_tie_slot_bytes(int B, int N) {
  return (B > 0 && N > 0) ? (size_t)B * ((N + 63) / 64) * 16 * (size_t)N : 0;
}
static int ties_inline(const vcr_knn_args* a);
// a replay is owed (tie_scratch) but the rows need global scratch that the caller did not provide
static bool tie_work_missing(const vcr_knn_args* a) {
  const size_t need = vcr_knn_tie_work_bytes(a->N);
  if (ties_inline(a) == 2) return false;                 // (the launch replays its ties itself, in its workgroups' slots)
  return a->tie_scratch && need && (!a->tie_work || a->tie_work_bytes < need || ((uintptr_t)a->tie_work & 15));
}

// the part of vcr_knn_args every caller must pass: through tie_cap (everything behind it is optional, zero = automatic)
static int knn_take(const vcr_knn_args* user, vcr_knn_args* mine) {
  return vcr_take_args(user, mine, offsetof(vcr_knn_args, waves));
}

extern "C" int vcr_knn_ties_f32(const vcr_knn_args* ua, const vcr_knn_args* ub, vcr_stream_t stream) {
  vcr_knn_args na, nb;
  if (knn_take(ua, &na) || (ub && knn_take(ub, &nb))) return VCR_EINVAL;
  const vcr_knn_args *a = &na, *b = ub ? &nb : nullptr;
  if (!a->x || !a->idx || !a->tie_scratch || a->tie_cap < 1) return VCR_EINVAL;
  if (b && (!b->x || !b->idx || !b->tie_scratch || b->tie_cap < 1)) return VCR_EINVAL;
  if (tie_work_missing(a) || (b && tie_work_missing(b))) return VCR_EUNSUPPORTED;
  const size_t la = tiebreak_launch_lds(a->N), lb = b ? tiebreak_launch_lds(b->N) : 0, lds = la > lb ? la : lb;
  if (b) return launch<knn_tiebreak2_kernel>(dim3(TB_BLOCKS, 2), dim3(256), lds, (hipStream_t)stream, *a, *b);
  return launch<knn_tiebreak_kernel>(dim3(TB_BLOCKS), dim3(256), lds, (hipStream_t)stream, *a);
}

// Which feature-space kernel: 16-query waves on 16x16x4 MFMAs (knn64c_body) or 32-query waves on 32x32x2 (knn64_body).
// vcr_knn_args.waves: 8 forces the former; 1 / 2 / 4 the latter with that candidate split; 0 = the half-size waves as soon
// as there are 1024 groups of 16 queries (a wave for every SIMD), by measurement on MI355X (profiles/rounds1-3/r3i_bench_knn.txt;
// 16- vs 32-query waves): one-launch pair 32 clouds x 1024: 148 vs 176 us inside the forward; pair 32 x 2048: 452 vs 477;
// 64 x 4096, k = 40: 2.39 vs 2.92 ms; alone they are level at k = 20 (32 x 1024: 112 vs 123 us, 32 x 2048: 315 vs 303).
// Smaller grids keep the 32-query kernels, whose S = 2 / 4 waves split the candidates of a query group.  Results are
// identical either way (same k-ascending fma chain, same selection).
static bool use_col16(const vcr_knn_args* a) {
  if (a->waves == 8 || a->k > 40) return true;           // (k = 41 .. 62: lists of 64, built for the 16-query bodies only)
  if (a->waves != 0) return false;
  return (long)((a->N + 15) / 16) * a->B >= 1024;
}

// In-kernel tie replay: the launch's workgroups are 4 waves of 16 queries (knn64c_body / knn3_body with S = 1) and a row's
// replay image fits beside nothing else in <= 40 KB of LDS (N <= ~2400): see replay_block_ties.
static int knn_s(const vcr_knn_args* a);
static size_t knn_log_bytes(const vcr_knn_args* a) {       // LDS of the four logs of such a workgroup
  const bool k20 = a->k <= 20;
  const size_t col16 = (size_t)4 * 2 * ((k20 ? pend_of<GeomCol16, 22>() : pend_of<GeomCol16, 42>()) + 4) * 16 * 4;
  if (a->C == 64) return col16;
  // the Cartesian search: unsplit = the MFMA body (same log as the feature-space kernel), split = the quad kernel
  return knn_s(a) == 1 ? col16 : (size_t)4 * 2 * ((k20 ? pend_of<GeomQuad, 22>() : pend_of<GeomQuad, 42>()) + 1) * 16 * 4;
}
static int knn_s(const vcr_knn_args* a) {                  // candidate split of the 32-query / Cartesian kernels (vcr_knn_f32)
  if (a->k > 20) return 1;
  if (a->waves == 1 || a->waves == 2 || a->waves == 4) return a->waves;
  const long groups = (long)((a->N + (a->C == 64 ? 31 : 15)) / (a->C == 64 ? 32 : 16)) * a->B;
  return groups >= 1024 ? 1 : groups >= 512 ? 2 : 4;
}
// 0: the tied rows go to a replay launch.  1: replayed by the workgroup that found them, row image in its LDS (N <= ~2400).
// 2 (k > 20): likewise, row image in the workgroup's slot of tie_work (the caller gave vcr_knn_tie_slot_bytes(B, N) bytes of it).
static int ties_inline(const vcr_knn_args* a) {
  if (!a->tie_scratch) return 0;
  if (a->C == 64 ? !use_col16(a) : knn_s(a) != 1) return 0;
  if (inline_tie_offset(knn_log_bytes(a), a->N) + (1 + BLK_TIES) * 4 <= INLINE_TIE_MAX_LDS) return 1;
  if (a->k <= 20) return 0;                              // (the slot form is compiled into the k > 20 kernels only)
  const size_t slots = vcr_knn_tie_slot_bytes(a->B, a->N);
  return (slots && a->tie_work && a->tie_work_bytes >= slots && !((uintptr_t)a->tie_work & 15)) ? 2 : 0;
}
static size_t knn_lds_bytes(const vcr_knn_args* a, int inl) {
  return inl ? inline_tie_offset(knn_log_bytes(a), inl == 2 ? 0 : a->N) + (1 + BLK_TIES) * 4 : knn_log_bytes(a);
}
extern "C" int vcr_knn_ties_inline(const vcr_knn_args* ua) {
  vcr_knn_args na;
  const vcr_knn_args* a = &na;
  return (knn_take(ua, &na) == 0 && a->x && a->idx && a->B > 0 && a->N > 0 && a->k > 0 && a->k <= 62 && (a->C == 64 || a->C == 4) && ties_inline(a)) ? 1 : 0;
}

// Feature-space (a64: C == 64) and Cartesian (a3: C == 4) kNN of the same pass as one launch (see knn_pair_kernel) when
// both are in the one-list-per-query regime the path runs in (k <= 20, >= 1024 query groups each); any other shape
// simply makes the two self-contained calls.  Tie handling as in vcr_knn_f32 (tie_defer honoured; a
// replay that is not deferred serves both launches at once).
// the ordered search's inputs are all there and the cloud is small enough for the wave's tile mask (256 tiles of 16 ranks)
// (vcr_knn_order_f32 writes the ranked rows and the centroids at pitch C: a padded x keeps the plain scan)
static bool knn_ordered(const vcr_knn_args* a) {
  return a->perm && a->xp && a->cen && a->cen_rad && a->cen_sqmax && (a->C == 4 || (a->sqp && a->cen_sq)) && a->N <= 16 * KNN_ORD_MAX_TILES &&
         a->ldx == a->C &&
         !(((uintptr_t)a->xp | (uintptr_t)a->cen) & 15);
}
extern "C" int vcr_knn_pair_f32(const vcr_knn_args* u64, const vcr_knn_args* u3, vcr_stream_t stream) {
  vcr_knn_args n64_, n3_;
  if (knn_take(u64, &n64_) || knn_take(u3, &n3_)) return VCR_EINVAL;
  const vcr_knn_args *a64 = &n64_, *a3 = &n3_;
  vcr_stream_scope scope_(stream);
  if (!a64->x || !a3->x || !a64->idx || !a3->idx || a64->C != 64 || a3->C != 4) return VCR_EINVAL;
  const bool col16 = use_col16(a64);
  const bool fusable = a64->k == a3->k && (a64->k <= 20 || (col16 && a64->k <= 40)) &&      // (k > 20: the 16-query bodies only)
                       (a64->waves == 0 || a64->waves == 1 || a64->waves == 8) && a3->waves == 0 &&
                       (long)((a64->N + (col16 ? 15 : 31)) / (col16 ? 16 : 32)) * a64->B >= 1024 && (long)((a3->N + 15) / 16) * a3->B >= 1024 &&
                       (a64->tie_scratch != nullptr) == (a3->tie_scratch != nullptr) && a64->tie_defer == a3->tie_defer;
  // small grids: the candidate-split kernels of both searches as one launch (k <= 20, automatic kernel choice)
  const bool small = !fusable && a64->k == a3->k && a64->k <= 20 && a64->waves == 0 && a3->waves == 0 && !col16 &&
                     (a64->tie_scratch != nullptr) == (a3->tie_scratch != nullptr) && a64->tie_defer == a3->tie_defer &&
                     a64->B > 0 && a3->B > 0 && a64->N > 0 && a3->N > 0 && a64->k > 0 && a64->k + 1 <= a64->N && a3->k + 1 <= a3->N &&
                     a64->N <= KNN_MAX_N && a3->N <= KNN_MAX_N && !knn_rows_overflow(a64) && !knn_rows_overflow(a3) && a64->sq && a64->ldx >= 64 && !(a64->ldx & 3) && a3->ldx >= 4 && !(a3->ldx & 3) &&
                     !(a64->tie_scratch && (a64->tie_cap < 1 || a3->tie_cap < 1)) && !tie_work_missing(a64) && !tie_work_missing(a3) &&
                     knn_s(a64) != 1;                     // (an unsplit feature-space search on a small grid: the separate launches)
  if (small) {
    hipStream_t s = (hipStream_t)stream;
    for (const vcr_knn_args* a : {a64, a3})
      if (a->tie_scratch && !a->tie_zeroed) {
        const int e = zero_count(a->tie_scratch, s);
        if (e != 0) return e;
      }
    const int S64 = knn_s(a64), S3 = knn_s(a3);
    const int gx64 = (a64->N + 32 * (4 / S64) - 1) / (32 * (4 / S64)), gx3 = (a3->N + 16 * (4 / S3) - 1) / (16 * (4 / S3));
    const int n64 = gx64 * a64->B, n3 = gx3 * a3->B;
    vcr_knn_args k64 = *a64, k3 = *a3;
    k64.tie_inline = 0; k3.tie_inline = ties_inline(a3);               // (only an unsplit Cartesian search replays in place)
    const size_t lds64 = (size_t)4 * 2 * (pend_of<GeomMfma, 22>() + 1) * 32 * 4;
    const size_t lds3 = knn_lds_bytes(a3, k3.tie_inline), lds = lds64 > lds3 ? lds64 : lds3;
    const dim3 grid(n64 + n3);
    int rc = VCR_EUNSUPPORTED;
#define VCR_KPS(A, B_) rc = launch<knn_pair_small_kernel<22, A, B_>>(grid, dim3(256), lds, s, k64, k3, n64, gx64, gx3)
    if (S64 == 4) { if (S3 == 4) VCR_KPS(4, 4); else if (S3 == 2) VCR_KPS(4, 2); else VCR_KPS(4, 1); }
    else if (S64 == 2) { if (S3 == 4) VCR_KPS(2, 4); else if (S3 == 2) VCR_KPS(2, 2); else VCR_KPS(2, 1); }
#undef VCR_KPS
    if (rc == 0 && a64->tie_scratch && !a64->tie_defer) rc = vcr_knn_ties_f32(a64, k3.tie_inline ? nullptr : a3, stream);
    return rc;
  }
  if (!fusable) {
    const int rc = vcr_knn_f32(a3, stream);
    return rc ? rc : vcr_knn_f32(a64, stream);
  }
  for (const vcr_knn_args* a : {a64, a3}) {
    if (a->B <= 0 || a->N <= 0 || a->k <= 0 || a->k + 1 > a->N || a->ldx < a->C || (a->ldx & 3)) return VCR_EINVAL;
    if (a->N > KNN_MAX_N || knn_rows_overflow(a)) return VCR_EUNSUPPORTED;
    if (a->tie_scratch && a->tie_cap < 1) return VCR_EINVAL;
    if (tie_work_missing(a)) return VCR_EUNSUPPORTED;
  }
  if (!a64->sq) return VCR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  for (const vcr_knn_args* a : {a64, a3})
    if (a->tie_scratch && !a->tie_zeroed) {
      const int e = zero_count(a->tie_scratch, s);
      if (e != 0) return e;
    }
  const int gx64 = col16 ? (a64->N + 63) / 64 : (a64->N + 127) / 128, gx3 = (a3->N + 63) / 64;   // 4 waves x 16 (or 32) queries
  const int n64 = gx64 * a64->B, n3 = gx3 * a3->B;
  vcr_knn_args k64 = *a64, k3 = *a3;                     // (tie_inline is the library's own field)
  k64.tie_inline = col16 ? ties_inline(a64) : 0; k3.tie_inline = ties_inline(a3);
  if (!knn_ordered(a64)) k64.perm = nullptr;             // (incomplete ordered inputs: that search runs the plain way)
  if (!knn_ordered(a3)) k3.perm = nullptr;
  const size_t lds64 = col16 ? knn_lds_bytes(a64, k64.tie_inline) : (size_t)4 * 2 * (pend_of<GeomMfma, 22>() + 1) * 32 * 4;
  const size_t lds3 = knn_lds_bytes(a3, k3.tie_inline), lds = lds64 > lds3 ? lds64 : lds3;
  const dim3 grid(n64 + n3);
  int rc;
  if (!col16) rc = launch<knn_pair_kernel<22, false>>(grid, dim3(256), lds, s, k64, k3, n64, gx64, gx3);
  else if ((knn_ordered(a64) || knn_ordered(a3)) && a64->xt)           // searches over the ranked rows (see vcr_knn_args.perm)
    rc = a64->k <= 20 ? launch<knn_pair_kernel<22, true, true, true>>(grid, dim3(256), lds, s, k64, k3, n64, gx64, gx3)
                      : launch<knn_pair_kernel<42, true, true, true>>(grid, dim3(256), lds, s, k64, k3, n64, gx64, gx3);
  else if (a64->k <= 20) rc = a64->xt ? launch<knn_pair_kernel<22, true, true>>(grid, dim3(256), lds, s, k64, k3, n64, gx64, gx3)
                                      : launch<knn_pair_kernel<22, true>>(grid, dim3(256), lds, s, k64, k3, n64, gx64, gx3);
  else rc = a64->xt ? launch<knn_pair_kernel<42, true, true>>(grid, dim3(256), lds, s, k64, k3, n64, gx64, gx3)
                    : launch<knn_pair_kernel<42, true>>(grid, dim3(256), lds, s, k64, k3, n64, gx64, gx3);
  // whatever was not replayed inside the launch: one replay launch, now or (tie_defer) when the caller asks for it
  if (rc == 0 && a64->tie_scratch && !a64->tie_defer) {
    if (!k64.tie_inline && !k3.tie_inline) rc = vcr_knn_ties_f32(a64, a3, stream);
    else if (!k64.tie_inline) rc = vcr_knn_ties_f32(a64, nullptr, stream);
    else if (!k3.tie_inline) rc = vcr_knn_ties_f32(a3, nullptr, stream);
  }
  return rc;
}

extern "C" int vcr_knn_f32(const vcr_knn_args* ua, vcr_stream_t stream) {
  vcr_knn_args na;
  if (knn_take(ua, &na)) return VCR_EINVAL;
  const vcr_knn_args* a = &na;
  vcr_stream_scope scope_(stream);
  if (!a->x || !a->idx) return VCR_EINVAL;
  if (a->B <= 0 || a->N <= 0 || a->k <= 0 || a->k + 1 > a->N) return VCR_EINVAL;
  if (a->k > 62 || a->N > KNN_MAX_N || knn_rows_overflow(a)) return VCR_EUNSUPPORTED;   // limits of this library, not of the operation (see vcr_hip.h):
                                                            // the tie replay keeps topk(k + 1)'s heap in the 64 lanes of a wave
  if (a->waves != 0 && a->waves != 1 && a->waves != 2 && a->waves != 4 && a->waves != 8) return VCR_EINVAL;
  if (a->C != 64 && a->C != 4) return VCR_EUNSUPPORTED;
  if (a->C == 64 ? (!a->sq || a->ldx < 64 || (a->ldx & 3)) : (a->ldx < 4 || (a->ldx & 3))) return VCR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (a->tie_scratch) {
    if (a->tie_cap < 1) return VCR_EINVAL;
    if (tie_work_missing(a)) return VCR_EUNSUPPORTED;     // refuse loudly rather than skip the replay silently
    if (!a->tie_zeroed) {
      const int e = zero_count(a->tie_scratch, s);
      if (e != 0) return e;
    }
  }
  int rc = VCR_EUNSUPPORTED;
  const bool k20 = a->k <= 20;                           // list of k+2 entries (one more than topk(k+1): exposes boundary ties)
  // S waves of a workgroup share one group of queries and split its candidates.  The selection work grows with the
  // number of lists (S per query), and measured on MI355X a second wave per SIMD bought with S = 2 only breaks even, so
  // S stays 1 as soon as that gives every SIMD (1024 of them) one wave; smaller grids split to fill the chip.
  // (the fold of S > 1 waves parks the value lists behind the logs: there is room for that with k <= 20 only)
  auto pick_s = [&](long) { return knn_s(a); };
  vcr_knn_args ka = *a;                                  // (tie_inline is the library's own field)
  const int inl = ties_inline(a);
  ka.tie_inline = inl;
  if (a->C == 64 && use_col16(a)) {
    // the half-size-wave kernel (see use_col16)
    if (!a->sq || a->ldx < 64 || (a->ldx & 3)) return VCR_EINVAL;
    const dim3 grid((a->N + 63) / 64, a->B);
    const size_t lds = knn_lds_bytes(a, inl);
    const bool k40 = a->k <= 40;                         // lists of k + 2: 22 / 42 / 64 entries
    if (a->xt) rc = k20 ? launch<knn64c_kernel<22, 4, true>>(grid, dim3(256), lds, s, ka)
                  : k40 ? launch<knn64c_kernel<42, 4, true>>(grid, dim3(256), lds, s, ka)
                        : launch<knn64c_kernel<64, 4, true>>(grid, dim3(256), lds, s, ka);
    else rc = k20 ? launch<knn64c_kernel<22, 4, false>>(grid, dim3(256), lds, s, ka)
              : k40 ? launch<knn64c_kernel<42, 4, false>>(grid, dim3(256), lds, s, ka)
                    : launch<knn64c_kernel<64, 4, false>>(grid, dim3(256), lds, s, ka);
  } else if (a->C == 64) {
    if (!a->sq || a->ldx < 64 || (a->ldx & 3)) return VCR_EINVAL;
    const int S = pick_s((long)((a->N + 31) / 32) * a->B), W = k20 ? 4 : 2;
    const dim3 grid((a->N + 32 * (W / S) - 1) / (32 * (W / S)), a->B);
    const size_t lds = (size_t)W * 2 * ((k20 ? pend_of<GeomMfma, 22>() : pend_of<GeomMfma, 42>()) + 1) * 32 * 4;
    rc = !k20 ? launch<knn64_kernel<42, 1, 2>>(grid, dim3(128), lds, s, *a)
         : S == 1 ? launch<knn64_kernel<22, 1, 4>>(grid, dim3(256), lds, s, *a)
         : S == 2 ? launch<knn64_kernel<22, 2, 4>>(grid, dim3(256), lds, s, *a)
                  : launch<knn64_kernel<22, 4, 4>>(grid, dim3(256), lds, s, *a);
  } else if (a->C == 4) {
    if (a->ldx < 4 || (a->ldx & 3)) return VCR_EINVAL;
    const int S = pick_s((long)((a->N + 15) / 16) * a->B);
    const dim3 grid((a->N + 16 * (4 / S) - 1) / (16 * (4 / S)), a->B);
    const size_t lds = knn_lds_bytes(a, inl);
    rc = a->k > 40 ? launch<knn3c_kernel<64, 4>>(grid, dim3(256), lds, s, ka)
         : !k20 ? launch<knn3c_kernel<42, 4>>(grid, dim3(256), lds, s, ka)
         : S == 1 ? launch<knn3c_kernel<22, 4>>(grid, dim3(256), lds, s, ka)
         : S == 2 ? launch<knn3_kernel<22, 2>>(grid, dim3(256), lds, s, *a)
                  : launch<knn3_kernel<22, 4>>(grid, dim3(256), lds, s, *a);
  }
  if (rc != 0) return rc;
  // rows with an exact tie at the (k+1)-th value: replay libstdc++'s selection on them (see knn_tiebreak_kernel)
  const size_t tb_lds = tiebreak_launch_lds(a->N);
  if (a->tie_scratch && !a->tie_defer && !inl) {         // (inl: the launch replayed its ties itself)
    rc = launch<knn_tiebreak_kernel>(dim3(TB_BLOCKS), dim3(256), tb_lds, s, *a);
  }
  return rc;
}
