// What the ORDERED kNN search reads (vcr_knn_args.perm ...; util/util.py:143-160 is order-free: any visiting order gives the same
// neighbour sets): the points of every cloud ranked along a Morton curve of their coordinates, the rows in rank order, and per
// tile of 16 consecutive ranks a ball (centroid, radius) that holds its rows -- in coordinate space and in feature space.  The
// features are a smooth function of the coordinates (lpdnet_model.py:111-112: two pointwise convs), so a tile of Morton
// neighbours is compact in BOTH spaces: with the tiles' balls a 16-query wave needs 0.4-0.5 of a 1024-point cloud's tiles and
// 0.2-0.3 of a 4096-point cloud's (profiles/rounds4-5/r5n_knn_feat_prune_potential.txt).
#include "common.h"

namespace {

constexpr int ORDER_MAX_N = 8192;                     // = 16 x KNN_ORD_MAX_TILES (knn.hip): the search's tile mask

__device__ __forceinline__ unsigned spread10(unsigned v) {           // 10 bits -> every third bit
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// one workgroup per cloud: bounding box, 30-bit Morton codes, and the rank of every (code, index) key by BUCKETS of the code's
// top 12 bits -- histogram (LDS atomics: the counts do not depend on their order), exclusive prefix, the keys dealt to their
// bucket's segment (slot order arbitrary), and a key's rank = its bucket's start + the number of smaller keys in that segment
// (keys are distinct: the ranking is the sort's, whatever order the slots were handed out in).  Four barriers instead of the
// ~70 steps of a bitonic sort in LDS (32 x 2048: 49 -> ~12 us, 64 x 4096: 107 -> ~20); a cloud whose points share few buckets
// (all points equal: one) degrades to the quadratic count of one segment, ~50 us at 4096 points.
constexpr int ORDER_BUCKETS = 4096;
__global__ __launch_bounds__(1024) void knn_morton_kernel(const float* xyz4, int N, int32_t* perm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char mo_smem[];
  unsigned long long* seg = reinterpret_cast<unsigned long long*>(mo_smem);           // [N] keys, grouped by bucket
  int* cnt = reinterpret_cast<int*>(seg + N);                                          // [BUCKETS] counts -> next free slot
  int* start = cnt + ORDER_BUCKETS;                                                    // [BUCKETS] first slot of a bucket
  float* red = reinterpret_cast<float*>(start + ORDER_BUCKETS);                        // [12][16]: lo, hi, sum, sum of squares x 3 axes
  int* wsum = reinterpret_cast<int*>(red + 192);                                       // [16] wave totals of the prefix
  const int t = threadIdx.x, nt = blockDim.x, b = blockIdx.x;
  const float* rows = xyz4 + (size_t)b * N * 4;
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  float s1[3] = {0.f, 0.f, 0.f}, s2[3] = {0.f, 0.f, 0.f};
  const f32x4 v0 = ld4(rows);                               // sums about the first point: no cancellation for clouds far from 0
  for (int i = t; i < N; i += nt) {
    const f32x4 v = ld4(rows + (size_t)i * 4);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      lo[d] = fminf(lo[d], v[d]); hi[d] = fmaxf(hi[d], v[d]);
      const float c = v[d] - v0[d];
      s1[d] += c; s2[d] += c * c;
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    lo[d] = -wave_max(-lo[d]); hi[d] = wave_max(hi[d]);
    s1[d] = wave_sum(s1[d]); s2[d] = wave_sum(s2[d]);
    if ((t & 63) == 0) {
      red[d * 16 + (t >> 6)] = lo[d]; red[(3 + d) * 16 + (t >> 6)] = hi[d];
      red[(6 + d) * 16 + (t >> 6)] = s1[d]; red[(9 + d) * 16 + (t >> 6)] = s2[d];
    }
  }
  for (int i = t; i < ORDER_BUCKETS; i += nt) cnt[i] = 0;
  __syncthreads();
  const int nw = nt >> 6;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float l = red[d * 16], h = red[(3 + d) * 16], a1 = red[(6 + d) * 16], a2 = red[(9 + d) * 16];
    for (int w = 1; w < nw; ++w) {
      l = fminf(l, red[d * 16 + w]); h = fmaxf(h, red[(3 + d) * 16 + w]);
      a1 += red[(6 + d) * 16 + w]; a2 += red[(9 + d) * 16 + w];
    }
    // A far outlier (one LiDAR return at 100 m) stretches the box and squeezes every other point into a handful of the
    // 4096 buckets, whose in-bucket ranking is quadratic.  When an axis' extent exceeds 16 standard deviations --
    // never for a cloud without outliers: a uniform cloud spans 3.5, 4096 gaussian points ~8 -- the box becomes mean +- 4
    // sigma (inside the true one) and the points beyond it take the boundary's code.  The ranking only steers the visiting
    // order: any box gives the same neighbour sets.
    const float mc = a1 / (float)N, var = fmaxf(a2 / (float)N - mc * mc, 0.f), sd = __builtin_sqrtf(var), mean = v0[d] + mc;
    if (h - l > 16.f * sd && sd > 0.f) { l = fmaxf(l, mean - 4.f * sd); h = fminf(h, mean + 4.f * sd); }
    lo[d] = l; hi[d] = h;
  }
  auto key_of = [&](int i) {
    const f32x4 v = ld4(rows + (size_t)i * 4);
    unsigned code = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float ext = hi[d] - lo[d];
      const float u = ext > 0.f ? (v[d] - lo[d]) / ext : 0.f;
      const unsigned qd = (unsigned)fminf(fmaxf(u * 1024.f, 0.f), 1023.f);
      code |= spread10(qd) << d;
    }
    return ((unsigned long long)code << 32) | (unsigned)i;
  };
  for (int i = t; i < N; i += nt) atomicAdd(&cnt[(int)(key_of(i) >> 50)], 1);          // bucket = the code's top 12 of 30 bits
  __syncthreads();
  // exclusive prefix of the ORDER_BUCKETS counts: ORDER_BUCKETS / nt consecutive buckets per thread, wave scan, wave totals
  {
    const int per = ORDER_BUCKETS / nt;                   // (nt divides ORDER_BUCKETS: 256 / 512 / 1024 threads)
    int local = 0;
    for (int u = 0; u < per; ++u) local += cnt[t * per + u];
    int inc = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(inc, o, 64);
      if ((t & 63) >= o) inc += up;
    }
    if ((t & 63) == 63) wsum[t >> 6] = inc;
    __syncthreads();
    int base = inc - local;
    for (int w = 0; w < (t >> 6); ++w) base += wsum[w];
    for (int u = 0; u < per; ++u) {
      const int c = cnt[t * per + u];
      start[t * per + u] = base;
      cnt[t * per + u] = base;                            // becomes the bucket's next free slot
      base += c;
    }
  }
  __syncthreads();
  for (int i = t; i < N; i += nt) {
    const unsigned long long k = key_of(i);
    seg[atomicAdd(&cnt[(int)(k >> 50)], 1)] = k;
  }
  __syncthreads();
  for (int i = t; i < N; i += nt) {
    const unsigned long long k = key_of(i);
    const int bk = (int)(k >> 50), s0 = start[bk], s1 = cnt[bk];
    int r = s0;
    for (int j = s0; j < s1; ++j) r += seg[j] < k ? 1 : 0;
    perm[(size_t)b * N + r] = i;
  }
}

// one wave per tile of 16 ranks: rows into rank order, the tile's ball.  lane = (row = lane >> 2, quarter = lane & 3)
__global__ __launch_bounds__(256) void knn_rank_rows_kernel(vcr_knn_order_args a, int T) {
  const int lane = threadIdx.x & 63;
  const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gw >= (long)a.B * T) return;
  const int b = (int)(gw / T), t = (int)(gw % T);
  const int row = lane >> 2, qd = lane & 3;
  const int r = t * 16 + row;
  const bool valid = r < a.N;
  const int o = a.perm[(size_t)b * a.N + min(r, a.N - 1)];
  const float inv = 1.f / (float)min(16, a.N - t * 16);
  auto rows_sum = [](float x) {                           // over the 16 rows (lanes of one quarter)
    x += __shfl_xor(x, 4, 64); x += __shfl_xor(x, 8, 64); x += __shfl_xor(x, 16, 64); x += __shfl_xor(x, 32, 64);
    return x;
  };
  auto rows_max = [](float x) {
    x = fmaxf(x, __shfl_xor(x, 4, 64)); x = fmaxf(x, __shfl_xor(x, 8, 64));
    x = fmaxf(x, __shfl_xor(x, 16, 64)); x = fmaxf(x, __shfl_xor(x, 32, 64));
    return x;
  };
  // ---- coordinates
  {
    const f32x4 v = ld4(a.xyz4 + ((size_t)b * a.N + o) * 4);
    if (valid && qd == 0) st4(a.xyz4_p + ((size_t)b * a.N + r) * 4, v);
    float c[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) c[d] = rows_sum(valid ? v[d] : 0.f) * inv;
    const float dx = v[0] - c[0], dy = v[1] - c[1], dz = v[2] - c[2];
    const float d2 = rows_max(valid ? dx * dx + dy * dy + dz * dz : 0.f);
    const float nmax = rows_max(valid ? v[3] : 0.f);
    if (lane == 0) {
      st4(a.cen4 + ((size_t)b * T + t) * 4, f32x4{c[0], c[1], c[2], c[0] * c[0] + c[1] * c[1] + c[2] * c[2]});
      a.cen4_rad[(size_t)b * T + t] = __builtin_sqrtf(d2 * 1.00002f) * 1.000002f + 1e-30f;      // rounded UP: a bound
      a.cen4_sqmax[(size_t)b * T + t] = nmax;
    }
  }
  if (!a.feat_t) return;
  // ---- features: this lane's 16 floats of its row (any layout: the centroid is taken position by position)
  f32x4 v[4];
  const float* src = a.feat_t + ((size_t)b * a.N + o) * a.ldf + 16 * qd;
#pragma unroll
  for (int g = 0; g < 4; ++g) v[g] = ld4(src + 4 * g);
  if (valid) {
    float* dst = a.feat_p + ((size_t)b * a.N + r) * 64 + 16 * qd;
#pragma unroll
    for (int g = 0; g < 4; ++g) st4(dst + 4 * g, v[g]);
  }
  const float sqv = a.sq[(size_t)b * a.N + o];
  if (valid && qd == 0) a.sq_p[(size_t)b * a.N + r] = sqv;
  float d2 = 0.f, cn = 0.f;
  f32x4 c[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float m = rows_sum(valid ? v[g][e] : 0.f) * inv;
      c[g][e] = m;
      const float df = v[g][e] - m;
      d2 += df * df;
      cn += m * m;
    }
  d2 += __shfl_xor(d2, 1, 64); d2 += __shfl_xor(d2, 2, 64);       // the row's four quarters
  cn += __shfl_xor(cn, 1, 64); cn += __shfl_xor(cn, 2, 64);
  d2 = rows_max(valid ? d2 : 0.f);
  const float nmax = rows_max(valid ? sqv : 0.f);
  if (row == 0) {
    float* dst = a.cen64 + ((size_t)b * T + t) * 64 + 16 * qd;
#pragma unroll
    for (int g = 0; g < 4; ++g) st4(dst + 4 * g, c[g]);
  }
  if (lane == 0) {
    a.cen64_sq[(size_t)b * T + t] = cn;
    a.cen64_rad[(size_t)b * T + t] = __builtin_sqrtf(d2 * 1.00002f) * 1.000002f + 1e-30f;
    a.cen64_sqmax[(size_t)b * T + t] = nmax;
  }
}

// The guard of the feature-space search (vcr_knn_order_args.ord_ok): one workgroup of 16 waves per cloud over its T tile balls,
// ONE pass.  lane = position in the 64-float centroid row (any layout: only distances between centroids are taken), wave w takes
// the tiles w, w + 16, ...; sums and sums of squares are taken about the FIRST tile's centroid (post-ReLU features have a mean
// of the order of their spread: the shift keeps the one-pass variance well conditioned; the verdict's margin is a factor 2-3).
constexpr float ORDER_GUARD_RATIO = 0.8f;
// measured (profiles/r6b_knn_guard.txt): the stem's features score 0.2-0.36 (LPD-pretrained and random weights), with noise of
// 0.3 sigma 0.29-0.39 -- ordered still 0.66-0.84 of the plain time --, with 1 sigma 1.24-1.35 -- level at 4096 points, 1.10x at
// 2048 --, unrelated features 23
__global__ __launch_bounds__(1024) void knn_order_guard_kernel(const float* cen64, const float* rad, int T, float ratio,
                                                               int32_t* ok, float* stat) {
  __shared__ float ps[16][64];                              // per wave: sum of (c - c0) per position
  __shared__ float red[32];                                 // per wave: sum of |c - c0|^2, sum of rad^2
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const float* c = cen64 + (size_t)b * T * 64;
  const float c0 = c[lane];
  float s = 0.f, q = 0.f;
  for (int i = w; i < T; i += 16) { const float d = c[(size_t)i * 64 + lane] - c0; s += d; q += d * d; }
  float r2 = 0.f;
  for (int i = t; i < T; i += 1024) { const float r = rad[(size_t)b * T + i]; r2 += r * r; }
  ps[w][lane] = s;
  q = wave_sum(q); r2 = wave_sum(r2);
  if (lane == 0) { red[w] = q; red[16 + w] = r2; }
  __syncthreads();
  if (w == 0) {
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sm += ps[i][lane];
    const float mean = sm / (float)T;                       // of (c - c0) at this position
    const float m2 = wave_sum(mean * mean);
    if (lane == 0) {
      float qs = 0.f, rs = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { qs += red[i]; rs += red[16 + i]; }
      const float between = fmaxf(qs / (float)T - m2, 0.f);
      const float within = rs / (float)T;
      const float v = within / fmaxf(between, 1e-37f);
      ok[b] = v < ratio ? 1 : 0;
      if (stat) stat[b] = v;
    }
  }
}

}  // namespace

extern "C" int vcr_knn_order_f32(const vcr_knn_order_args* a, vcr_stream_t stream) {
  vcr_stream_scope bound(stream);
  if (!a || !a->xyz4 || !a->perm || !a->xyz4_p || !a->cen4 || !a->cen4_rad || !a->cen4_sqmax) return VCR_EINVAL;
  if (a->B <= 0 || a->N <= 0) return VCR_EINVAL;
  if (a->feat_t && (!a->sq || !a->feat_p || !a->sq_p || !a->cen64 || !a->cen64_sq || !a->cen64_rad || !a->cen64_sqmax ||
                    a->ldf < 64 || (a->ldf & 3)))
    return VCR_EINVAL;
  if (a->ord_ok && !a->feat_t) return VCR_EINVAL;         // (the guard judges the feature-space tiles)
  if (((uintptr_t)a->xyz4 | (uintptr_t)a->xyz4_p | (uintptr_t)a->cen4 | (uintptr_t)a->feat_t | (uintptr_t)a->feat_p |
       (uintptr_t)a->cen64) & 15)
    return VCR_EINVAL;
  if (a->N > ORDER_MAX_N) return VCR_EUNSUPPORTED;
  const int threads = a->N >= 2048 ? 1024 : a->N >= 512 ? 512 : 256;
  const size_t lds = (size_t)a->N * 8 + 2 * ORDER_BUCKETS * 4 + 192 * 4 + 16 * 4;
  VCR_DYN_LDS(knn_morton_kernel, (int)lds);               // (64.9 KB at 4096 points, 97.7 KB at 8192)
  hipLaunchKernelGGL(knn_morton_kernel, dim3(a->B), dim3(threads), lds, (hipStream_t)stream, a->xyz4, a->N, a->perm);
  int rc = VCR_LAUNCH_RC();
  if (rc != 0) return rc;
  const int T = (a->N + 15) / 16;
  const long waves = (long)a->B * T;
  hipLaunchKernelGGL(knn_rank_rows_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a, T);
  rc = VCR_LAUNCH_RC();
  if (rc != 0 || !a->ord_ok) return rc;
  hipLaunchKernelGGL(knn_order_guard_kernel, dim3(a->B), dim3(1024), 0, (hipStream_t)stream, a->cen64, a->cen64_rad, T,
                     a->guard_ratio > 0.f ? a->guard_ratio : ORDER_GUARD_RATIO, a->ord_ok, a->ord_stat);
  return VCR_LAUNCH_RC();
}
