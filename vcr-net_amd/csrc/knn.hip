// Kernel 1: fused pairwise-distance + top-k (util/util.py:143-160).  The N x N distance matrix is
// never written: distances are produced tile by tile in registers and filtered against each
// query's current k-th best; survivors are parked in a per-lane LDS list and merged into a
// register-resident sorted list in wave-synchronous batches, so the (divergent) insertion cost is
// paid per survivor, not per candidate.
//
//   D_ij = (-sq_j + 2 x_i.x_j) - sq_i      (same association as util.py:157-158)
//   idx  = top-(k+1) of D_i. by (value desc, index asc), rank 0 dropped (util.py:159)
//
// C == 64: v_mfma_f32_32x32x2_f32 with candidates as MFMA rows and queries as MFMA columns, so
//          every lane owns ONE query column (lanes l and l+32 share a query and split the
//          candidates); -sq_j/2 rides along as a 33rd k-step, which reproduces the reference's
//          rounding of (-sq_j + 2 dot) exactly.  Operands go global -> registers (candidate tiles are
//          L2-resident), LDS holds only the survivor lists.
// C == 4 : Cartesian xyz4 rows, one lane per query, candidates broadcast from LDS, VALU.
#include "common.h"

namespace {

constexpr int PEND = 32;            // survivor slots per lane between merges
constexpr int TILE = 32;            // candidates per MFMA tile

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
  // slot t-1 is the "shift" condition of slot t).  Strict '>' : an equal value never displaces an
  // earlier entry, and candidates arrive in increasing index order per lane, so ties resolve to the
  // lower index.
  __device__ __forceinline__ void insert(float d, int j) {
    bool c_hi = d > v[KS - 1];
#pragma unroll
    for (int t = KS - 1; t >= 1; --t) {
      const bool c_lo = d > v[t - 1];
      id[t] = c_lo ? id[t - 1] : (c_hi ? j : id[t]);
      v[t] = __builtin_amdgcn_fmed3f(v[t - 1], d, v[t]);
      c_hi = c_lo;
    }
    id[0] = c_hi ? j : id[0];
    v[0] = fmaxf(v[0], d);
  }
  // lexicographic (value desc, index asc) for merging two lists with interleaved indices
  __device__ __forceinline__ void insert_lex(float d, int j) {
    bool c_hi = d > v[KS - 1] || (d == v[KS - 1] && j < id[KS - 1]);
#pragma unroll
    for (int t = KS - 1; t >= 1; --t) {
      const bool c_lo = d > v[t - 1] || (d == v[t - 1] && j < id[t - 1]);
      id[t] = c_lo ? id[t - 1] : (c_hi ? j : id[t]);
      v[t] = __builtin_amdgcn_fmed3f(v[t - 1], d, v[t]);
      c_hi = c_lo;
    }
    id[0] = c_hi ? j : id[0];
    v[0] = fmaxf(v[0], d);
  }
};

// Survivor list of one wave: [slot][lane] so a wave's pushes hit 64 consecutive words.
struct Pending {
  float* pv; int* pi; int cnt;
  __device__ __forceinline__ void push(float d, int j, int lane) {
    pv[cnt * 64 + lane] = d; pi[cnt * 64 + lane] = j; ++cnt;
  }
  template <int KS>
  __device__ __forceinline__ float drain(TopList<KS>& L, int lane) {
    for (int i = 0; __any(i < cnt); ++i) {
      if (i < cnt) {
        const float d = pv[i * 64 + lane];
        const int j = pi[i * 64 + lane];
        if (d > L.v[KS - 1]) L.insert(d, j);
      }
    }
    cnt = 0;
    return L.v[KS - 1];
  }
};

// ---------------------------------------------------------------- C == 64 (MFMA)
template <int KS>
__global__ __launch_bounds__(256, (KS > 21 ? 1 : 2)) void knn64_kernel(vcr_knn_args a) {
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
  // query fragment: this lane supplies B[k][col] for k = 8m + 4*half + s  (k permuted identically on A)
  f32x4 qf[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) qf[m] = ld4(xb + (size_t)q * a.ldx + 8 * m + 4 * half);
  const float sq_q = sqb[q];

  TopList<KS> L;
  L.init();
  float thr = VCR_NEG_INF;

  const int ntiles = (a.N + TILE - 1) / TILE;
  f32x4 cf[8];
  float csq;
  {
    const int c = min(col, a.N - 1);
#pragma unroll
    for (int m = 0; m < 8; ++m) cf[m] = ld4(xb + (size_t)c * a.ldx + 8 * m + 4 * half);
    csq = sqb[c];
  }
  for (int tile = 0; tile < ntiles; ++tile) {
    f32x4 nf[8];
    float nsq = 0.f;
    if (tile + 1 < ntiles) {                            // prefetch next candidate tile (registers)
      const int c = min((tile + 1) * TILE + col, a.N - 1);
#pragma unroll
      for (int m = 0; m < 8; ++m) nf[m] = ld4(xb + (size_t)c * a.ldx + 8 * m + 4 * half);
      nsq = sqb[c];
    }
    f32x16 acc = {0};
#pragma unroll
    for (int m = 0; m < 8; ++m) {
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = mfma32(cf[m][s], qf[m][s], acc);
    }
    // 33rd k-step: A[cand][k*] = -sq_cand/2 (half 0), B[k*][q] = 1  ->  acc = dot - sq_j/2, rounded once
    acc = mfma32(half == 0 ? -0.5f * csq : 0.f, half == 0 ? 1.f : 0.f, acc);

    if (__any(pend.cnt > PEND - 16)) thr = pend.drain(L, lane);
    const int jbase = tile * TILE;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = jbase + acc_row(r, half);
      const float d = 2.f * acc[r] - sq_q;              // (-sq_j + 2 dot) - sq_i
      if (d > thr && j < a.N) pend.push(d, j, lane);
    }
    if (tile + 1 < ntiles) {
#pragma unroll
      for (int m = 0; m < 8; ++m) cf[m] = nf[m];
      csq = nsq;
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
      if (d > L.v[KS - 1] || (d == L.v[KS - 1] && j < L.id[KS - 1])) L.insert_lex(d, j);
    }
    if (q0 + col < a.N) {
      int32_t* o = a.idx + ((size_t)b * a.N + q0 + col) * a.k;
#pragma unroll
      for (int t = 1; t < KS; ++t)
        if (t <= a.k) o[t - 1] = L.id[t];
    }
  }
}

// ---------------------------------------------------------------- C == 4 (xyz4, VALU)
template <int KS>
__global__ __launch_bounds__(64) void knn3_kernel(vcr_knn_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  const int b = blockIdx.y;
  Pending pend;
  pend.pv = reinterpret_cast<float*>(smem);
  pend.pi = reinterpret_cast<int*>(pend.pv + PEND * 64);
  pend.cnt = 0;
  f32x4* cand = reinterpret_cast<f32x4*>(smem + 2 * PEND * 64 * 4);
  const float* xb = a.x + (size_t)b * a.N * a.ldx;
  for (int i = lane; i < a.N; i += 64) cand[i] = ld4(xb + (size_t)i * a.ldx);
  __syncthreads();
  const int qi = blockIdx.x * 64 + lane;
  const f32x4 qv = cand[min(qi, a.N - 1)];
  TopList<KS> L;
  L.init();
  float thr = VCR_NEG_INF;
  for (int j0 = 0; j0 < a.N; j0 += 16) {
    if (__any(pend.cnt > PEND - 16)) thr = pend.drain(L, lane);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = j0 + u;
      if (j < a.N) {                                    // wave-uniform
        const f32x4 c = cand[j];                        // LDS broadcast
        const float dot = fmaf(qv[2], c[2], fmaf(qv[1], c[1], qv[0] * c[0]));
        const float d = (2.f * dot - c[3]) - qv[3];
        if (d > thr) pend.push(d, j, lane);
      }
    }
  }
  pend.drain(L, lane);
  if (qi < a.N) {
    int32_t* o = a.idx + ((size_t)b * a.N + qi) * a.k;
#pragma unroll
    for (int t = 1; t < KS; ++t)
      if (t <= a.k) o[t - 1] = L.id[t];
  }
}

}  // namespace

extern "C" int vcr_knn_f32(const vcr_knn_args* a, vcr_stream_t stream) {
  if (!a || !a->x || !a->idx) return VCR_EINVAL;
  if (a->B <= 0 || a->N <= 0 || a->k <= 0 || a->k > 40 || a->k + 1 > a->N || a->N > 65535) return VCR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const size_t pend_bytes = (size_t)2 * PEND * 64 * 4;
  if (a->C == 64) {
    if (!a->sq || a->ldx < 64 || (a->ldx & 3)) return VCR_EINVAL;
    dim3 grid((a->N + 127) / 128, a->B);
    if (a->k <= 20) hipLaunchKernelGGL(knn64_kernel<21>, grid, dim3(256), 4 * pend_bytes, s, *a);
    else hipLaunchKernelGGL(knn64_kernel<41>, grid, dim3(256), 4 * pend_bytes, s, *a);
  } else if (a->C == 4) {
    if (a->ldx < 4 || (a->ldx & 3)) return VCR_EINVAL;
    const size_t lds = pend_bytes + (size_t)a->N * 16;
    if (lds > 160 * 1024) return VCR_EUNSUPPORTED;
    dim3 grid((a->N + 63) / 64, a->B);
    if (a->k <= 20) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(knn3_kernel<21>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(knn3_kernel<21>, grid, dim3(64), lds, s, *a);
    } else {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(knn3_kernel<41>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(knn3_kernel<41>, grid, dim3(64), lds, s, *a);
    }
  } else {
    return VCR_EUNSUPPORTED;
  }
  return VCR_LAUNCH_RC();
}
