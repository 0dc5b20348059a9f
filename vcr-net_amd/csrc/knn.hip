// Kernel 1: fused pairwise-distance + top-k (util/util.py:143-160).  The N x N distance matrix is
// never written: distances are produced tile by tile in registers and filtered against each
// query's current k-th best; survivors are parked in a per-query LDS list and merged into a
// register-resident sorted list in wave-synchronous batches, so the insertion network is paid per
// survivor (~k ln(N/k) per query), not per candidate.
//
//   D_ij = (-sq_j + 2 x_i.x_j) - sq_i      (same association as util.py:157-158)
//   idx  = top-(k+1) of D_i. by value, rank 0 dropped (util.py:159); the k kept indices are written as a SET
//          (unordered) -- every consumer is a max over neighbours.
//          Exact ties at the (k+1)-th value: Tensor.topk on the CPU is libstdc++'s std::nth_element (or
//          std::partial_sort when (k+1)*64 <= N) with a value-only comparator, so WHICH of the tied candidates it
//          keeps is an artefact of introselect's pivoting / the heap's shape.  The lists here carry one entry more
//          than needed, which makes such a tie visible (about 1 row in 10^4 in fp32); those rows are re-done by
//          knn_tiebreak_kernel, a replica of the libstdc++ algorithms, so that the neighbour SETS equal the
//          reference's on every row (validated against torch.topk on tie-heavy inputs).  A row WITHOUT a boundary tie
//          has a set that depends on the values only, so the order in which equal values entered the list is
//          irrelevant; without tie_scratch a boundary tie keeps one of the tied candidates, deterministically.
//
// Round 2: ONE sorted list per query, spread over the lanes that share the query, instead of one full list per lane
// and a candidate split that multiplied the lists (and with them the k ln(N/k) insertions: 2 lists per query in the
// feature-space kernel, 4-8 in the Cartesian one).  Lane segment s holds ranks [s*T, (s+1)*T); an insertion of d runs
// on every segment at once, segment s taking min(d, last element of segment s-1) -- the element that falls off the
// segment above, known BEFORE the insertion -- so the segments need one cross-lane move per insertion and no chain.
//
// C == 64: v_mfma_f32_32x32x2_f32 with candidates as MFMA rows and queries as MFMA columns: lanes l and l+32 own one
//          query column and 16 candidate rows each; they hold the two halves of the query's list (exchange:
//          v_permlane32_swap).  The k order of the MFMA chain is the natural one (step s multiplies k = 2s, 2s+1) and
//          -sq_j/2 rides along as a 33rd k-step: together with the pointwise kernel's reference-ordered features and
//          norms the distance matrix is BIT-IDENTICAL to the reference's (CPU sgemm = k-ascending fma chain;
//          verified), so the feature-space neighbour sets never flip.  S waves of a workgroup share a query tile and
//          split the candidate tiles (S = 2 fills two waves per SIMD at BASELINE configs[1]); their lists are folded
//          into one at the end.
// C == 4 : Cartesian xyz4 rows on the VALU: four lanes (one DPP quad) per query, 16 queries per wave, each lane
//          scanning every fourth candidate; segments exchange through DPP quad_perm.
#include "common.h"

namespace {

constexpr int TILE = 32;            // candidates per MFMA tile

#ifdef VCR_TIMELINE
// Experiment-only (profiles/timeline_knn.py, -DVCR_TIMELINE builds): wave 0 of every workgroup accumulates the 100 MHz
// wall clock over the phases of its scan.
__device__ unsigned long long vcr_tl_knn[4096 * 8];
#define KTL_DECL unsigned long long ktl_t = wall_clock64(), ktl_acc[6] = {0, 0, 0, 0, 0, 0}
#define KTL(slot) do { const unsigned long long n_ = wall_clock64(); ktl_acc[slot] += n_ - ktl_t; ktl_t = n_; } while (0)
#define KTL_FLUSH do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096) for (int i_ = 0; i_ < 6; ++i_) vcr_tl_knn[blockIdx.x * 8 + i_] = ktl_acc[i_]; } while (0)
#else
#define KTL_DECL ((void)0)
#define KTL(slot) ((void)0)
#define KTL_FLUSH ((void)0)
#endif

// A sorted-descending segment of T (value, index) entries in registers.
template <int T>
struct Seg {
  float v[T];
  int id[T];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int t = 0; t < T; ++t) { v[t] = VCR_NEG_INF; id[t] = 0x7fffffff; }
  }
  // Insert d: values move with ONE v_med3_f32 per slot (for v[t-1] >= v[t] the new slot value is
  // median(v[t-1], d, v[t])), indices follow with one compare per slot (the compare of slot t-1 is the "shift"
  // condition of slot t).  Strict '>': an equal value goes behind the entry already there.  Inserting -inf (or any
  // value <= the last entry) is a no-op, which lets callers run the network unconditionally (no divergent branch
  // around 2*T live registers).
  __device__ __forceinline__ void insert(float d, int j) {
    bool c_hi = d > v[T - 1];
#pragma unroll
    for (int t = T - 1; t >= 1; --t) {
      const bool c_lo = d > v[t - 1];
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

// values of the other 32-lane half (v_permlane32_swap: result 0 = the lower half's values in both halves, 1 = the upper's)
__device__ __forceinline__ int other_half(int x, int half) {
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return half ? r[0] : r[1];
}
__device__ __forceinline__ float other_half(float x, int half) {
  return __int_as_float(other_half(__float_as_int(x), half));
}

// ---------------------------------------------------------------- C == 64 (MFMA)
// Workgroup = 4 waves = 4/S query tiles of 32 queries; wave (qt, part) scans candidate tiles part, part+S, ...
// KS = list length (k+2 rounded to 22 / 42), one half of it per lane.
template <int KS, int S>
__global__ __launch_bounds__(256, 2) void knn64_kernel(vcr_knn_args a) {
  constexpr int T = KS / 2;                              // list entries per lane (lane half h holds ranks h*T ..)
  constexpr int PEND = 64;                               // survivor slots per query between drains (a tile adds <= 32)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int b = blockIdx.y;
  const int qt = wave / S, part = wave % S;
  const int q0 = (blockIdx.x * (4 / S) + qt) * 32;       // this wave's 32 queries (may lie beyond N: clamped, not written)
  float* pv = reinterpret_cast<float*>(smem) + wave * (2 * PEND * 32);       // [slot][32 queries]
  int* pi = reinterpret_cast<int*>(pv + PEND * 32);
  int cnt = 0;                                           // survivors parked for this lane's query (same in both halves)

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

  Seg<T> L;
  L.init();
  float thr = VCR_NEG_INF;                               // the list's last value: nothing <= thr can enter
  // One insertion into the query's list, both halves at once.  When d displaces the upper segment's last entry
  // (d > that entry, known before either insertion), that entry falls into the lower segment -- at its FRONT,
  // unconditionally: it is >= everything there, and a compare-based insertion would drop it on an exact tie with the
  // lower segment's own entries while the upper segment has already let go of it.
  auto insert_one = [&](float d, int j) {
    const float ob = other_half(L.v[T - 1], half);
    const int oj = other_half(L.id[T - 1], half);
    const bool take = half && d > ob;
    L.insert(take ? __builtin_huge_valf() : d, take ? oj : j);
    if (take) L.v[0] = ob;
  };
  auto drain = [&]() {
    float dn = pv[col];
    int jn = pi[col];
    for (int i = 0; __any(i < cnt); ++i) {               // branch-free body: idle lanes insert -inf (a no-op)
      const float d = i < cnt ? dn : VCR_NEG_INF;
      const int j = jn;
      const int nx = min(i + 1, PEND - 1);
      dn = pv[nx * 32 + col];
      jn = pi[nx * 32 + col];
      insert_one(d, j);
    }
    cnt = 0;
    const float ol = other_half(L.v[T - 1], half);
    thr = half ? L.v[T - 1] : ol;                        // the LOWER segment's last value, in both halves
  };

  const int ntiles = (a.N + TILE - 1) / TILE;
  float cf[32];
  float csq = 0.f;
  if (part < ntiles) {
    const int c = min(part * TILE + col, a.N - 1);
    f32x4 raw[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) raw[m] = ld4(xb + (size_t)c * a.ldx + 4 * m);
    pick(raw, cf);
    csq = sqb[c];
  }
  KTL_DECL;
  for (int tile = part; tile < ntiles; tile += S) {
    // k <= 20: the next candidate tile is prefetched as raw rows (64 VGPRs) across the MFMA + selection phase.
    // k = 40: the longer list leaves no room for that at two waves per SIMD; the tile is loaded after the selection.
    constexpr bool PREFETCH = KS <= 22;
    f32x4 nraw[PREFETCH ? 16 : 1];
    float nsq = 0.f;
    if (PREFETCH && tile + S < ntiles) {
      const int c = min((tile + S) * TILE + col, a.N - 1);
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
    // accumulator lands in AGPRs (seen in the k = 40 build: register 15, the last one written, was read stale).
    // Tie the wait states to the accumulator itself so they cannot be scheduled away.
    if (KS > 22) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc));
#ifdef VCR_TIMELINE
    asm volatile("" : "+v"(acc));
    { const float fence_ = acc[15]; asm volatile("" :: "v"(fence_)); }
#endif
    KTL(0);                                              // prefetch issue + MFMA chain

    const int jbase = tile * TILE;
    float dd[16];
    unsigned m = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dd[r] = 2.f * acc[r] - sq_q;                       // (-sq_j + 2 dot) - sq_i
      m |= (dd[r] > thr && jbase + acc_row(r, half) < a.N) ? (1u << r) : 0u;
    }
    const unsigned om = (unsigned)other_half((int)m, half);
    if (__any(m != 0)) {                                 // both halves of a column append to ONE list: upper half first
      const int base = cnt + (half ? __popc(om) : 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (m & (1u << r)) {
          const int pos = base + __popc(m & ((1u << r) - 1u));
          pv[pos * 32 + col] = dd[r];
          pi[pos * 32 + col] = jbase + acc_row(r, half);
        }
      }
    }
    cnt += __popc(m) + __popc(om);
    KTL(1);                                              // filter + push
    if (__any(cnt > PEND - 32)) drain();
    KTL(2);                                              // drain
    if (tile + S < ntiles) {
      if (PREFETCH) {
        pick(nraw, cf);
        csq = nsq;
      } else {
        const int c = min((tile + S) * TILE + col, a.N - 1);
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) {                  // four 64-B quarters of the row: 16 temporaries, not 64
          f32x4 raw[4];
#pragma unroll
          for (int mm = 0; mm < 4; ++mm) raw[mm] = ld4(xb + (size_t)c * a.ldx + 16 * m4 + 4 * mm);
#pragma unroll
          for (int st = 0; st < 8; ++st)
            cf[8 * m4 + st] = half ? raw[st >> 1][(st & 1) * 2 + 1] : raw[st >> 1][(st & 1) * 2];
        }
        csq = sqb[c];
      }
    }
    KTL(3);                                              // operand pick (waits for the prefetched rows)
  }
  drain();
  KTL(4);
  KTL_FLUSH;

  // ---- the S waves of a query tile fold their lists into wave part 0's, which writes ranks 1..k
  float* lv = pv;                                        // reuse the wave's survivor area: [KS][32]
  int* li = pi;
  auto dump = [&]() {
#pragma unroll
    for (int t = 0; t < T; ++t) { lv[(half * T + t) * 32 + col] = L.v[t]; li[(half * T + t) * 32 + col] = L.id[t]; }
  };
  if (S > 1) {
    if (part != 0) dump();
    __syncthreads();
    if (part == 0) {
      for (int p = 1; p < S; ++p) {
        const float* ov = reinterpret_cast<const float*>(smem) + (wave + p) * (2 * PEND * 32);
        const int* oi = reinterpret_cast<const int*>(ov + PEND * 32);
        for (int t = 0; t < KS; ++t) insert_one(ov[t * 32 + col], oi[t * 32 + col]);
      }
    }
  }
  if (part == 0) {
    dump();
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): same-wave LDS hand-off
    __builtin_amdgcn_wave_barrier();
    if (q0 + col < a.N) {
      int32_t* o = a.idx + ((size_t)b * a.N + q0 + col) * a.k;
      for (int t = 1 + half; t <= a.k; t += 2) o[t - 1] = li[t * 32 + col];    // rank 0 dropped (util.py:159)
      if (half == 0) {
        const float vk = lv[a.k * 32 + col], vk1 = lv[(a.k + 1) * 32 + col];   // ranks k+1 and k+2
        if (vk1 == vk && vk1 > VCR_NEG_INF) report_tie(a.tie_scratch, a.tie_cap, b * a.N + q0 + col);
      }
    }
  }
}

// ---------------------------------------------------------------- C == 4 (xyz4, VALU)
// Wave = 16 queries x 4 lanes (DPP quad = query); lane s of the quad computes the distances of candidates j = 4u + s
// and holds ranks [s*T, (s+1)*T) of the query's list (T = 6 for k <= 20: 24 entries, T = 11 for k <= 40: 44).
// S waves of a workgroup may split the candidates of a query group (small grids); their lists are folded at the end.
__device__ __forceinline__ int quad_from_prev(int x) {   // lane s <- lane s-1 of its quad (lane 0: itself)
  return __builtin_amdgcn_mov_dpp(x, 0x90, 0xF, 0xF, true);
}
__device__ __forceinline__ int quad_from_prev2(int x) {  // lane s <- lane s-2 (lanes 0, 1: themselves)
  return __builtin_amdgcn_mov_dpp(x, 0x44, 0xF, 0xF, true);
}
__device__ __forceinline__ int quad_bcast3(int x) { return __builtin_amdgcn_mov_dpp(x, 0xFF, 0xF, 0xF, true); }

template <int KS, int S>
__global__ __launch_bounds__(256, 2) void knn3_kernel(vcr_knn_args a) {
  constexpr int T = (KS + 3) / 4;                        // entries per lane; the list holds 4T >= KS entries
  constexpr int PEND = 64;                               // survivor slots per query between drains (a step adds <= 16)
  constexpr int TL = (KS - 1) / T, TS = (KS - 1) % T;    // lane / slot of rank KS-1: the threshold
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = lane & 3, qd = lane >> 2;
  const int b = blockIdx.y;
  const int grp = wave / S, part = wave % S;
  const int q0 = (blockIdx.x * (4 / S) + grp) * 16;
  float* pv = reinterpret_cast<float*>(smem) + wave * (2 * PEND * 16);       // [slot][16 queries]
  int* pi = reinterpret_cast<int*>(pv + PEND * 16);
  int cnt = 0;
  const float* xb = a.x + (size_t)b * a.N * a.ldx;
  const int qi = q0 + qd;
  const f32x4 qv = ld4(xb + (size_t)min(qi, a.N - 1) * a.ldx);
  Seg<T> L;
  L.init();
  float thr = VCR_NEG_INF;
  // segment s takes min(d, last entry of segment s-1); when that is the entry falling off the segment above it goes to
  // the FRONT unconditionally (see knn64_kernel: a compare-based insertion would lose it on an exact tie)
  auto insert_one = [&](float d, int j) {
    const float pb = __int_as_float(quad_from_prev(__float_as_int(L.v[T - 1])));
    const int pj = quad_from_prev(L.id[T - 1]);
    const bool take = s && d > pb;
    L.insert(take ? __builtin_huge_valf() : d, take ? pj : j);
    if (take) L.v[0] = pb;
  };
  auto drain = [&]() {
    float dn = pv[qd];
    int jn = pi[qd];
    for (int i = 0; __any(i < cnt); ++i) {
      const float d = i < cnt ? dn : VCR_NEG_INF;
      const int j = jn;
      const int nx = min(i + 1, PEND - 1);
      dn = pv[nx * 16 + qd];
      jn = pi[nx * 16 + qd];
      insert_one(d, j);
    }
    cnt = 0;
    // rank KS-1 lives in lane TL of the quad, slot TS: broadcast it
    float tv = L.v[TS];
    if (TL == 3) tv = __int_as_float(quad_bcast3(__float_as_int(tv)));
    else tv = __shfl(tv, (lane & ~3) + TL, 64);
    thr = tv;
  };
  // 16 candidates per step, 4 per lane: j = j0 + 4u + s (the quad reads 64 contiguous bytes per load)
  const int nsteps = (a.N + 15) / 16;
  int it = 0;
  KTL_DECL;
  for (int st = part; st < nsteps; st += S, ++it) {
    const int j0 = st * 16;
    if (it < 3 || __any(cnt > PEND - 16)) drain();       // early steps: settle the threshold quickly
    KTL(2);
    f32x4 c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = ld4(xb + (size_t)min(j0 + 4 * u + s, a.N - 1) * a.ldx);
    float dd[4];
    unsigned m = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float dot = fmaf(qv[2], c[u][2], fmaf(qv[1], c[u][1], qv[0] * c[u][0]));
      dd[u] = (2.f * dot - c[u][3]) - qv[3];
      m |= (dd[u] > thr && j0 + 4 * u + s < a.N) ? (1u << u) : 0u;
    }
    // the quad appends to ONE list: exclusive prefix of the lanes' survivor counts
    // (DPP reads of lanes that are switched off return 0: every cross-lane move is issued with all lanes active and
    // only its RESULT is selected per lane)
    const int c0 = __popc(m);
    const int p1 = quad_from_prev(c0);
    int inc = c0 + (s >= 1 ? p1 : 0);
    const int p2 = quad_from_prev2(inc);
    inc += s >= 2 ? p2 : 0;
    const int total = quad_bcast3(inc);
    if (m) {
      const int base = cnt + inc - c0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (m & (1u << u)) {
          const int pos = base + __popc(m & ((1u << u) - 1u));
          pv[pos * 16 + qd] = dd[u];
          pi[pos * 16 + qd] = j0 + 4 * u + s;
        }
      }
    }
    cnt += total;
    KTL(1);                                              // prefix + push
  }
  drain();
  KTL(4);
  KTL_FLUSH;

  float* lv = pv;                                        // [4T][16]
  int* li = pi;
  auto dump = [&]() {
#pragma unroll
    for (int t = 0; t < T; ++t) { lv[(s * T + t) * 16 + qd] = L.v[t]; li[(s * T + t) * 16 + qd] = L.id[t]; }
  };
  if (S > 1) {
    if (part != 0) dump();
    __syncthreads();
    if (part == 0) {
      for (int p = 1; p < S; ++p) {
        const float* ov = reinterpret_cast<const float*>(smem) + (wave + p) * (2 * PEND * 16);
        const int* oi = reinterpret_cast<const int*>(ov + PEND * 16);
        for (int t = 0; t < KS; ++t) insert_one(ov[t * 16 + qd], oi[t * 16 + qd]);
      }
    }
  }
  if (part == 0) {
    dump();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (qi < a.N) {
      int32_t* o = a.idx + ((size_t)b * a.N + qi) * a.k;
      for (int t = 1 + s; t <= a.k; t += 4) o[t - 1] = li[t * 16 + qd];
      if (s == 0) {
        const float vk = lv[a.k * 16 + qd], vk1 = lv[(a.k + 1) * 16 + qd];
        if (vk1 == vk && vk1 > VCR_NEG_INF) report_tie(a.tie_scratch, a.tie_cap, b * a.N + qi);
      }
    }
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

#ifdef VCR_TIMELINE
extern "C" int vcr_dbg_timeline_knn(unsigned long long* host_dst, int clear) {
  if (clear) {
    static unsigned long long zeros[4096 * 8];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(vcr_tl_knn), zeros, sizeof(zeros));
  }
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(vcr_tl_knn), sizeof(unsigned long long) * 4096 * 8);
}
#endif

extern "C" int vcr_knn_f32(const vcr_knn_args* a, vcr_stream_t stream) {
  if (!a || !a->x || !a->idx) return VCR_EINVAL;
  if (a->B <= 0 || a->N <= 0 || a->k <= 0 || a->k > 40 || a->k + 1 > a->N || a->N > 65535) return VCR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (a->tie_scratch) {
    if (a->tie_cap < 1) return VCR_EINVAL;
    const hipError_t e = hipMemsetAsync(a->tie_scratch, 0, sizeof(int32_t), s);
    if (e != hipSuccess) return (int)e;
  }
  int rc = VCR_EUNSUPPORTED;
  const bool k20 = a->k <= 20;                           // list of k+2 entries (one more than topk(k+1): exposes boundary ties)
  // S waves of a workgroup share one group of queries and split its candidates.  One list per query keeps the
  // insertion work minimal, so S stays 1 whenever that alone gives every SIMD (MI355X: 1024) two waves.
  auto pick_s = [&](long groups) { return a->waves == 1 || a->waves == 2 || a->waves == 4 ? a->waves
                                          : groups >= 2048 ? 1 : groups >= 1024 ? 2 : 4; };
  if (a->C == 64) {
    if (!a->sq || a->ldx < 64 || (a->ldx & 3)) return VCR_EINVAL;
    const int S = pick_s((long)((a->N + 31) / 32) * a->B);
    const dim3 grid((a->N + 32 * (4 / S) - 1) / (32 * (4 / S)), a->B);
    const size_t lds = (size_t)4 * 2 * 64 * 32 * 4;
#define VCR_KNN64(SV) (k20 ? launch<knn64_kernel<22, SV>>(grid, dim3(256), lds, s, *a) : launch<knn64_kernel<42, SV>>(grid, dim3(256), lds, s, *a))
    rc = S == 1 ? VCR_KNN64(1) : S == 2 ? VCR_KNN64(2) : VCR_KNN64(4);
#undef VCR_KNN64
  } else if (a->C == 4) {
    if (a->ldx < 4 || (a->ldx & 3)) return VCR_EINVAL;
    const int S = pick_s((long)((a->N + 15) / 16) * a->B);
    const dim3 grid((a->N + 16 * (4 / S) - 1) / (16 * (4 / S)), a->B);
    const size_t lds = (size_t)4 * 2 * 64 * 16 * 4;
#define VCR_KNN3(SV) (k20 ? launch<knn3_kernel<22, SV>>(grid, dim3(256), lds, s, *a) : launch<knn3_kernel<42, SV>>(grid, dim3(256), lds, s, *a))
    rc = S == 1 ? VCR_KNN3(1) : S == 2 ? VCR_KNN3(2) : VCR_KNN3(4);
#undef VCR_KNN3
  }
  if (rc != 0) return rc;
  // rows with an exact tie at the (k+1)-th value: replay libstdc++'s selection on them (see knn_tiebreak_kernel)
  const size_t tb_lds = (size_t)a->N * 16 + 256 + (16 + 2 * 256 + 2) * 4;
  if (a->tie_scratch && tb_lds <= 160 * 1024) rc = launch<knn_tiebreak_kernel>(dim3(64), dim3(256), tb_lds, s, *a);
  return rc;
}
