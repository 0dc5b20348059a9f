// Kernel 1: fused pairwise-distance + top-k (util/util.py:143-160).  The N x N distance matrix is
// never written: distances are produced tile by tile in registers and filtered against each
// query's current k-th best; survivors are parked in a per-query LDS list and merged into a
// register-resident sorted list in wave-synchronous batches, so the insertion network is paid per
// survivor (~k ln(N/k) per query), not per candidate.
//
//   D_ij = (-sq_j + 2 x_i.x_j) - sq_i      (same association as util.py:157-158)
//   idx  = top-(k+1) of D_i. by (value desc, index asc), rank 0 dropped (util.py:159); the k kept
//          indices are written as a SET (unordered) -- every consumer is a max over neighbours.
//          Exact ties at the (k+1)-th value: Tensor.topk on the CPU is libstdc++'s std::nth_element (or
//          std::partial_sort when (k+1)*64 <= N) with a value-only comparator, so WHICH of the tied candidates it
//          keeps is an artefact of introselect's pivoting / the heap's shape.  The lists here carry one entry more
//          than needed, which makes such a tie visible (about 1 row in 10^4 in fp32); those rows are re-done by
//          knn_tiebreak_kernel, a sequential replica of the libstdc++ algorithms, so that the neighbour SETS equal
//          the reference's on every row (validated against torch.topk on tie-heavy inputs).
//
// C == 64: v_mfma_f32_32x32x2_f32 with candidates as MFMA rows and queries as MFMA columns, so every lane
//          owns ONE query column (lanes l and l+32 share a query and split the candidates, their two sorted
//          lists are merged at the end).  The k order of the MFMA chain is the natural one (step s multiplies
//          k = 2s, 2s+1) and -sq_j/2 rides along as a 33rd k-step: together with the pointwise kernel's
//          reference-ordered features and norms the distance matrix is BIT-IDENTICAL to the reference's
//          (CPU sgemm = k-ascending fma chain; verified), so the feature-space neighbour sets never flip.
//          Operands go global -> registers (candidate tiles are L2-resident); LDS holds the survivor lists.
// C == 4 : Cartesian xyz4 rows, one lane per query, candidates broadcast from LDS 16 at a time, VALU.
//          A block's 2-8 waves split the candidates in interleaved groups; their sorted lists are combined by a
//          tree of bitonic sorted merges (merge_sorted).
#include "common.h"

namespace {

constexpr int PEND = 32;            // survivor slots per (lane) sub-list between merges
constexpr int TILE = 32;            // candidates per MFMA tile
template <int KS> constexpr int knn3_slots() { return KS > 24 ? KS : 24; }   // Cartesian kernel: small areas = more waves per CU

__device__ __forceinline__ bool lex_gt(float d, int j, float v, int id) { return d > v || (d == v && j < id); }

template <int KS>
struct TopList {
  float v[KS];
  int id[KS];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int t = 0; t < KS; ++t) { v[t] = VCR_NEG_INF; id[t] = 0x7fffffff; }
  }
  // Sorted-descending insert.  Values move with ONE v_med3_f32 per slot: for v[t-1] >= v[t] the new
  // slot value is median(v[t-1], d, v[t]).  Indices follow with one compare per slot (the compare of
  // slot t-1 is the "shift" condition of slot t).  LEX = exact (value desc, index asc) order for streams
  // that are not index-sorted; otherwise strict '>' keeps the earlier (= lower index) entry on ties.
  // Inserting -inf is a no-op, which lets callers run the network unconditionally (no divergent branch
  // around 2*KS live registers).
  template <bool LEX>
  __device__ __forceinline__ void insert(float d, int j) {
    bool c_hi = LEX ? lex_gt(d, j, v[KS - 1], id[KS - 1]) : d > v[KS - 1];
#pragma unroll
    for (int t = KS - 1; t >= 1; --t) {
      const bool c_lo = LEX ? lex_gt(d, j, v[t - 1], id[t - 1]) : d > v[t - 1];
      id[t] = c_lo ? id[t - 1] : (c_hi ? j : id[t]);
      v[t] = __builtin_amdgcn_fmed3f(v[t - 1], d, v[t]);
      c_hi = c_lo;
    }
    id[0] = c_hi ? j : id[0];
    v[0] = fmaxf(v[0], d);
  }
};

// Row whose (k+1)-th and (k+2)-th best values are equal: hand it to knn_tiebreak_kernel (ties[0] = count).
__device__ __forceinline__ void report_tie(int32_t* ties, int cap, int row) {
  if (!ties) return;
  const int pos = atomicAdd(&ties[0], 1);
  if (pos < cap) ties[1 + pos] = row;
}

// Survivor list of one wave: [slot][lane] so a wave's pushes hit 64 consecutive words.
struct Pending {
  float* pv; int* pi; int cnt;
  __device__ __forceinline__ void push(float d, int j, int lane) {
    pv[cnt * 64 + lane] = d; pi[cnt * 64 + lane] = j; ++cnt;
  }
  template <int KS>
  __device__ __forceinline__ float drain(TopList<KS>& L, int lane) {
    float dn = pv[lane];
    int jn = pi[lane];
    for (int i = 0; __any(i < cnt); ++i) {               // branch-free body: idle lanes insert -inf (a no-op)
      const float d = i < cnt ? dn : VCR_NEG_INF;
      const int j = jn;
      const int nx = min(i + 1, PEND - 1);
      dn = pv[nx * 64 + lane];
      jn = pi[nx * 64 + lane];
      L.template insert<false>(d, j);
    }
    cnt = 0;
    return L.v[KS - 1];
  }
};

// ---------------------------------------------------------------- C == 64 (MFMA)
template <int KS>
__global__ __launch_bounds__(256, 2) void knn64_kernel(vcr_knn_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int b = blockIdx.y;
  const int q0 = (blockIdx.x * 4 + wave) * 32;          // this wave's 32 queries
  if (q0 >= a.N) return;                                // wave-uniform
  Pending pend;
  pend.pv = reinterpret_cast<float*>(smem) + wave * (2 * PEND * 64);
  pend.pi = reinterpret_cast<int*>(pend.pv + PEND * 64);
  pend.cnt = 0;

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

  TopList<KS> L;
  L.init();
  float thr = VCR_NEG_INF;

  const int ntiles = (a.N + TILE - 1) / TILE;
  float cf[32];
  float csq;
  {
    const int c = min(col, a.N - 1);
    f32x4 raw[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) raw[m] = ld4(xb + (size_t)c * a.ldx + 4 * m);
    pick(raw, cf);
    csq = sqb[c];
  }
  for (int tile = 0; tile < ntiles; ++tile) {
    // k <= 20: the next candidate tile is prefetched as raw rows (64 VGPRs) across the MFMA + selection phase.
    // k = 40: the 41-entry list leaves no room for that at two waves per SIMD, and two waves hide the load latency
    // better than one wave with a prefetch (measured), so the tile is loaded after the selection phase instead.
    constexpr bool PREFETCH = KS <= 22;
    f32x4 nraw[PREFETCH ? 16 : 1];
    float nsq = 0.f;
    if (PREFETCH && tile + 1 < ntiles) {                // prefetch next candidate tile (raw rows stay in flight)
      const int c = min((tile + 1) * TILE + col, a.N - 1);
#pragma unroll
      for (int m = 0; m < (PREFETCH ? 16 : 1); ++m) nraw[m] = ld4(xb + (size_t)c * a.ldx + 4 * m);
      nsq = sqb[c];
    }
    f32x16 acc = {0};
#pragma unroll
    for (int st = 0; st < 32; ++st) acc = mfma32(cf[st], qf[st], acc);
    // 33rd k-step: A[cand][k*] = -sq_cand/2 (half 0), B[k*][q] = 1  ->  acc = dot - sq_j/2, rounded once
    acc = mfma32(half == 0 ? -0.5f * csq : 0.f, half == 0 ? 1.f : 0.f, acc);
    // hipcc (ROCm 7.2) under-pads the MFMA -> v_accvgpr_read hazard of this 16-pass instruction when the
    // accumulator lands in AGPRs (seen only in the 512-register KS=41 build: register 15, the last one written,
    // was read stale).  Tie the wait states to the accumulator itself so they cannot be scheduled away.
    if (KS > 22) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc));

    if (__any(pend.cnt > PEND - 16)) thr = pend.drain(L, lane);
    const int jbase = tile * TILE;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = jbase + acc_row(r, half);
      const float d = 2.f * acc[r] - sq_q;              // (-sq_j + 2 dot) - sq_i
      if (d > thr && j < a.N) pend.push(d, j, lane);
    }
    if (tile + 1 < ntiles) {
      if (PREFETCH) {
        pick(nraw, cf);
        csq = nsq;
      } else {
        const int c = min((tile + 1) * TILE + col, a.N - 1);
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) {                  // four 64-B quarters of the row: 16 temporaries, not 64
          f32x4 raw[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) raw[m] = ld4(xb + (size_t)c * a.ldx + 16 * m4 + 4 * m);
#pragma unroll
          for (int st = 0; st < 8; ++st)
            cf[8 * m4 + st] = half ? raw[st >> 1][(st & 1) * 2 + 1] : raw[st >> 1][(st & 1) * 2];
        }
        csq = sqb[c];
      }
    }
  }
  pend.drain(L, lane);

  // merge the two half-lists of each query: half 1 hands its list over through LDS
  float* mv = pend.pv;                                  // reuse: [KS][32]
  int* mi = pend.pi;
  if (half == 1) {
#pragma unroll
    for (int t = 0; t < KS; ++t) { mv[t * 32 + col] = L.v[t]; mi[t * 32 + col] = L.id[t]; }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): same-wave LDS hand-off
  __builtin_amdgcn_wave_barrier();
  if (half == 0) {
    for (int t = 0; t < KS; ++t) {
      const float d = mv[t * 32 + col];
      const int j = mi[t * 32 + col];
      if (d > L.v[KS - 1] || (d == L.v[KS - 1] && j < L.id[KS - 1])) L.template insert<true>(d, j);
    }
    if (q0 + col < a.N) {
      int32_t* o = a.idx + ((size_t)b * a.N + q0 + col) * a.k;
      bool tie = false;
#pragma unroll
      for (int t = 1; t < KS; ++t) {
        if (t <= a.k) o[t - 1] = L.id[t];
        if (t == a.k + 1) tie = L.v[t] == L.v[t - 1] && L.v[t] > VCR_NEG_INF;   // rank k+2 equals rank k+1
      }
      if (tie) report_tie(a.tie_scratch, a.tie_cap, b * a.N + q0 + col);
    }
  }
}

// Sorted merge of this lane's list with another sorted list held in LDS as [KS][64] (column `lane`):
// X[i] = max(A'[i], B'[H-1-i]) over the lists padded with -inf to H entries is a bitonic sequence that holds
// the top-H of the union; a log2(H)-stage bitonic network sorts it, the first KS entries are the new list.
// Order: value descending, index ascending (exact, so the result does not depend on how candidates were split).
template <int KS>
__device__ __forceinline__ void merge_sorted(TopList<KS>& L, const float* bv, const int* bi, int lane) {
  constexpr int H = KS <= 32 ? 32 : 64;
  float xv[H];
  int xi[H];
#pragma unroll
  for (int i = 0; i < H; ++i) {
    const int jb = H - 1 - i;
    const float av = i < KS ? L.v[i] : VCR_NEG_INF;
    const int ai = i < KS ? L.id[i] : 0x7fffffff;
    const float ov = jb < KS ? bv[jb * 64 + lane] : VCR_NEG_INF;
    const int oi = jb < KS ? bi[jb * 64 + lane] : 0x7fffffff;
    const bool o = lex_gt(ov, oi, av, ai);
    xv[i] = o ? ov : av;
    xi[i] = o ? oi : ai;
  }
#pragma unroll
  for (int d = H / 2; d >= 1; d >>= 1) {
#pragma unroll
    for (int i = 0; i < H; ++i) {
      if ((i & d) == 0) {
        const bool sw = lex_gt(xv[i + d], xi[i + d], xv[i], xi[i]);
        const float hv = sw ? xv[i + d] : xv[i], lv = sw ? xv[i] : xv[i + d];
        const int hi = sw ? xi[i + d] : xi[i], li = sw ? xi[i] : xi[i + d];
        xv[i] = hv; xi[i] = hi; xv[i + d] = lv; xi[i + d] = li;
      }
    }
  }
#pragma unroll
  for (int t = 0; t < KS; ++t) { L.v[t] = xv[t]; L.id[t] = xi[t]; }
}

// ---------------------------------------------------------------- C == 4 (xyz4, VALU)
// Block = 2, 4 or 8 waves over the same 64 queries (one lane per query); wave s scans candidate groups
// s, s + nw, ... (16 candidates each, fetched as wave-uniform 16-B reads: one L1/L2 broadcast per candidate, no
// LDS copy of the cloud -- the kernel is latency-bound and LDS is what limits the waves per CU).  The waves'
// sorted lists are combined by a binary tree of merge_sorted() steps through LDS; wave 0 writes ranks 1..k.
template <int KS>
__global__ __launch_bounds__(512) void knn3_kernel(vcr_knn_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, s = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int b = blockIdx.y;
  constexpr int PEND3 = knn3_slots<KS>();               // survivor slots of this kernel (>= KS for the merge hand-off)
  constexpr int AS = PEND3 * 64;
  float* pv = reinterpret_cast<float*>(smem) + s * (2 * AS);
  int* pi = reinterpret_cast<int*>(pv + AS);
  int cnt = 0;
  const float* xb = a.x + (size_t)b * a.N * a.ldx;
  const int qi = blockIdx.x * 64 + lane;
  const f32x4 qv = ld4(xb + (size_t)min(qi, a.N - 1) * a.ldx);
  TopList<KS> L;
  L.init();
  float thr = VCR_NEG_INF;
  auto drain = [&]() {
    float dn = pv[lane];
    int jn = pi[lane];
    for (int i = 0; __any(i < cnt); ++i) {
      const float d = i < cnt ? dn : VCR_NEG_INF;
      const int j = jn;
      const int nx = min(i + 1, PEND3 - 1);
      dn = pv[nx * 64 + lane];
      jn = pi[nx * 64 + lane];
      L.template insert<false>(d, j);                    // this lane's stream is index-sorted
    }
    cnt = 0;
    thr = L.v[KS - 1];
  };
  const int ngroups = (a.N + 15) / 16;
  int it = 0;
  for (int grp = s; grp < ngroups; grp += nw, ++it) {
    const int j0 = grp * 16;
    if (it < 3 || __any(cnt > PEND3 - 16)) drain();       // early groups: settle the threshold quickly
    f32x4 c[16];
#pragma unroll
    for (int u = 0; u < 16; ++u)                          // 16 wave-uniform 16-B reads in flight (L1/L2 broadcasts)
      c[u] = ld4(xb + (size_t)min(j0 + u, a.N - 1) * a.ldx);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = j0 + u;
      const float dot = fmaf(qv[2], c[u][2], fmaf(qv[1], c[u][1], qv[0] * c[u][0]));
      const float d = (2.f * dot - c[u][3]) - qv[3];
      if (d > thr && j < a.N) { pv[cnt * 64 + lane] = d; pi[cnt * 64 + lane] = j; ++cnt; }
    }
  }
  drain();
  for (int step = 1; step < nw; step <<= 1) {             // tree merge: wave s absorbs wave s + step
    if ((s & (2 * step - 1)) == step) {
#pragma unroll
      for (int t = 0; t < KS; ++t) { pv[t * 64 + lane] = L.v[t]; pi[t * 64 + lane] = L.id[t]; }
    }
    __syncthreads();
    if ((s & (2 * step - 1)) == 0 && s + step < nw) {
      const float* ov = reinterpret_cast<const float*>(smem) + (s + step) * (2 * AS);
      merge_sorted<KS>(L, ov, reinterpret_cast<const int*>(ov + AS), lane);
    }
  }
  if (s == 0 && qi < a.N) {
    int32_t* o = a.idx + ((size_t)b * a.N + qi) * a.k;
    bool tie = false;
#pragma unroll
    for (int t = 1; t < KS; ++t) {
      if (t <= a.k) o[t - 1] = L.id[t];
      if (t == a.k + 1) tie = L.v[t] == L.v[t - 1] && L.v[t] > VCR_NEG_INF;
    }
    if (tie) report_tie(a.tie_scratch, a.tie_cap, b * a.N + qi);
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

// One block per tied row: all threads recompute the row's N distances with the SAME arithmetic as the main kernels
// (C == 64: the k-ascending fma chain the MFMA produces, then the -sq_j/2 step, then 2 acc - sq_i; C == 4: the VALU
// expression of knn3_kernel), thread 0 replays the selection and rewrites the row's k indices.
__global__ __launch_bounds__(256) void knn_tiebreak_kernel(vcr_knn_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* val = reinterpret_cast<float*>(smem);
  int* id = reinterpret_cast<int*>(val + a.N);
  float* qrow = reinterpret_cast<float*>(id + a.N);      // [64]
  int* A = reinterpret_cast<int*>(qrow + 64);            // [N] left stoppers, [N] right stoppers, block scratch
  int* Bd = A + a.N;
  int* red = Bd + a.N;                                   // [16 + 2*256 + 2]
  const int count = min(a.tie_scratch[0], a.tie_cap);
  for (int t = blockIdx.x; t < count; t += gridDim.x) {
    const int row = a.tie_scratch[1 + t];
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
      int best = 0;                                      // rank 0 = the largest of the K kept (lowest index on ties)
      for (int i = 1; i < K; ++i)
        if (val[i] > val[best] || (val[i] == val[best] && id[i] < id[best])) best = i;
      int32_t* o = a.idx + (size_t)row * a.k;
      int w = 0;
      for (int i = 0; i < K; ++i)
        if (i != best) o[w++] = id[i];
    }
  }
}

template <auto Kernel>
int launch(dim3 grid, dim3 block, size_t lds, hipStream_t s, const vcr_knn_args& a) {
  VCR_DYN_LDS(Kernel, (int)lds);                         // one cache per kernel: Kernel is a template argument
  hipLaunchKernelGGL(Kernel, grid, block, lds, s, a);
  return VCR_LAUNCH_RC();
}

}  // namespace

extern "C" int vcr_knn_f32(const vcr_knn_args* a, vcr_stream_t stream) {
  if (!a || !a->x || !a->idx) return VCR_EINVAL;
  if (a->B <= 0 || a->N <= 0 || a->k <= 0 || a->k > 40 || a->k + 1 > a->N || a->N > 65535) return VCR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int ks = a->k <= 20 ? 22 : 42;                   // k+1 kept entries + one more that exposes boundary ties
  if (a->tie_scratch) {
    if (a->tie_cap < 1) return VCR_EINVAL;
    const hipError_t e = hipMemsetAsync(a->tie_scratch, 0, sizeof(int32_t), s);
    if (e != hipSuccess) return (int)e;
  }
  int rc = VCR_EUNSUPPORTED;
  if (a->C == 64) {
    if (!a->sq || a->ldx < 64 || (a->ldx & 3)) return VCR_EINVAL;
    dim3 grid((a->N + 127) / 128, a->B);
    const size_t lds64 = (size_t)4 * 2 * PEND * 64 * 4;
    rc = a->k <= 20 ? launch<knn64_kernel<22>>(grid, dim3(256), lds64, s, *a)
                    : launch<knn64_kernel<42>>(grid, dim3(256), lds64, s, *a);
  } else if (a->C == 4) {
    if (a->ldx < 4 || (a->ldx & 3)) return VCR_EINVAL;
    // Waves per 64 queries.  Splitting the candidates over more waves shortens each wave's serial scan + insert
    // chain but adds inserts and merge steps in total, so it only pays while the chip is under-filled: keep 2 waves
    // unless that leaves at most one wave per SIMD (MI355X: 256 CUs x 4 SIMDs), 8 only for very small grids.
    const long blocks = (long)((a->N + 63) / 64) * a->B;
    int nw = blocks * 4 <= 1024 ? 8 : blocks * 2 <= 1024 ? 4 : 2;
    if (a->waves == 2 || a->waves == 4 || a->waves == 8) nw = a->waves;   // caller's override (tests / tuning)
    while (nw > 2 && (size_t)nw * 2 * (ks > 24 ? ks : 24) * 64 * 4 > 160 * 1024) nw >>= 1;
    const size_t lds = (size_t)nw * 2 * (ks > 24 ? ks : 24) * 64 * 4;
    dim3 grid((a->N + 63) / 64, a->B);
    rc = a->k <= 20 ? launch<knn3_kernel<22>>(grid, dim3(64 * nw), lds, s, *a)
                    : launch<knn3_kernel<42>>(grid, dim3(64 * nw), lds, s, *a);
  }
  if (rc != 0) return rc;
  // rows with an exact tie at the (k+1)-th value: replay libstdc++'s selection on them (see knn_tiebreak_kernel)
  const size_t tb_lds = (size_t)a->N * 16 + 256 + (16 + 2 * 256 + 2) * 4;
  if (a->tie_scratch && tb_lds <= 160 * 1024) rc = launch<knn_tiebreak_kernel>(dim3(64), dim3(256), tb_lds, s, *a);
  return rc;
}
