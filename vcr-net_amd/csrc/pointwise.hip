// K-a: conv1_lpd (3->64) + ReLU + conv2_lpd (64->64) + ReLU  (model/lpdnet_model.py:111-112)
// plus the side products the later kernels need: xyz4 rows and squared feature norms.
// 0.009 GF per cloud: VALU, one launch, weights staged once per block in LDS.
#include "common.h"

namespace {

constexpr int PTS = 64;  // points per 256-thread block; 4 threads per point, 16 channels each

__global__ __launch_bounds__(256) void pointwise12_kernel(vcr_pointwise_args a) {
  __shared__ float w2t[64][64];       // [k][c]
  __shared__ float h1s[PTS][65];
  __shared__ float w1s[64][4];        // (w0,w1,w2,b1)
  __shared__ float b2s[64];
  const int t = threadIdx.x;
  for (int i = t; i < 64 * 64; i += 256) w2t[i & 63][i >> 6] = a.w2[i];   // w2[c][k] -> [k][c]
  if (t < 64) {
    w1s[t][0] = a.w1[t * 3 + 0]; w1s[t][1] = a.w1[t * 3 + 1]; w1s[t][2] = a.w1[t * 3 + 2];
    w1s[t][3] = a.b1[t]; b2s[t] = a.b2[t];
  }
  const int b = blockIdx.y;
  const int p = t >> 2, g = t & 3;
  const int n = blockIdx.x * PTS + p;
  const bool live = n < a.N;
  const int nc = live ? n : a.N - 1;
  const float* xb = b < a.B ? a.x_cf + (size_t)b * 3 * a.N : a.x_cf2 + (size_t)(b - a.B) * 3 * a.N;   // (second block of clouds)
  const float x = xb[nc], y = xb[a.N + nc], z = xb[2 * a.N + nc];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int c = g * 16 + j;
    // rounding order of the reference's (multi-threaded oneDNN) 1x1 conv, established bit-for-bit against
    // F.conv1d: accumulator starts at the bias, one fma per input channel in ascending order
    const float v = fmaf(w1s[c][2], z, fmaf(w1s[c][1], y, fmaf(w1s[c][0], x, w1s[c][3])));
    h1s[p][c] = fmaxf(v, 0.f);
  }
  __syncthreads();
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = b2s[g * 16 + j];   // same order for conv2: bias first, k ascending
  for (int k = 0; k < 64; ++k) {
    const float h = h1s[p][k];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 w = ld4(&w2t[k][g * 16 + q * 4]);
      acc[q * 4 + 0] = fmaf(w[0], h, acc[q * 4 + 0]);
      acc[q * 4 + 1] = fmaf(w[1], h, acc[q * 4 + 1]);
      acc[q * 4 + 2] = fmaf(w[2], h, acc[q * 4 + 2]);
      acc[q * 4 + 3] = fmaf(w[3], h, acc[q * 4 + 3]);
    }
  }
  // |feat|^2 with torch.sum(x**2, dim=1)'s exact association (ATen cascade sum, 64 rows -> level step 16):
  // rounded squares, four sequential 16-channel block sums b0..b3 starting from 0, then ((b0 + b1) + b2) + b3.
  // Each thread owns exactly one block.  Together with the k-ordered MFMA chain of the kNN kernel this makes
  // the feature-space distance matrix bit-identical to the reference's, so its top-k sets cannot flip.
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    acc[j] = fmaxf(acc[j], 0.f);
    ss = ss + acc[j] * acc[j];
  }
  const int base = (threadIdx.x & 63) & ~3;
  {
    const float b0 = __shfl(ss, base, 64), b1 = __shfl(ss, base + 1, 64);
    const float b2 = __shfl(ss, base + 2, 64), b3 = __shfl(ss, base + 3, 64);
    ss = ((b0 + b1) + b2) + b3;
  }
  // Ragged N: ATen reduces the last N % 32 points of a cloud (the columns its 32-wide vector loop leaves over) in a
  // scalar loop with FOUR interleaved accumulators, a_l = sum of channels l, l+4, l+8, ... in ascending order,
  // combined as ((a0 + a1) + a2) + a3 (established bit-for-bit like the rule above).  The chain runs through the
  // four threads of a point in turn.
  const bool tail = live && n >= (a.N & ~31);
  if (__builtin_amdgcn_ballot_w64(tail) != 0) {
    float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sgrp = 0; sgrp < 4; ++sgrp) {
      if (g == sgrp) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a4[j & 3] = a4[j & 3] + acc[j] * acc[j];
      }
#pragma unroll
      for (int l = 0; l < 4; ++l) a4[l] = __shfl(a4[l], base + sgrp, 64);
    }
    if (tail) ss = ((a4[0] + a4[1]) + a4[2]) + a4[3];
  }
  if (live) {
    const size_t row = (size_t)b * a.N + n;
    float* o = a.feat64 + row * 64 + g * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) st4(o + q * 4, f32x4{acc[q * 4], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]});
    if (g == 0) {
      a.sq64[row] = ss;
      st4(a.xyz4 + row * 4, f32x4{x, y, z, (x * x + y * y) + z * z});   // torch.sum order for 3 rows: sequential
    }
  }
}

// [B,3,N] channels-first points -> [B,N,4] rows (x, y, z, |p|^2): the layout the kNN / ICP kernels read.
__global__ __launch_bounds__(256) void rows4_kernel(const float* x_cf, float* xyz4, int B, int N) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * N) return;
  const int b = (int)(i / N), n = (int)(i % N);
  const float* xb = x_cf + (size_t)b * 3 * N;
  const float x = xb[n], y = xb[N + n], z = xb[2 * N + n];
  st4(xyz4 + i * 4, f32x4{x, y, z, (x * x + y * y) + z * z});
}

// rows4 + DGCNN's first EdgeConv projection per point (model/vcrnet_model.py:108 through the neighbour / centre split):
// pq[i] = W xyz_i + b with W [C,ldw] (only its first three columns are non-zero: K = 3), C / 4 lanes per point.
// The fp32 fma chain in k order is what the K-padded MFMA GEMM it replaces computes.
__global__ __launch_bounds__(256) void rows4_pq_kernel(const float* x_cf, float* xyz4, int B, int N, const float* wpq,
                                                       int ldw, const float* bpq, int C, float* pq, int ldpq) {
  const int per = C / 4, cg = threadIdx.x % per;
  const long i = (long)blockIdx.x * (256 / per) + threadIdx.x / per;
  if (i >= (long)B * N) return;
  const int b = (int)(i / N), n = (int)(i % N);
  const float* xb = x_cf + (size_t)b * 3 * N;
  const float x = xb[n], y = xb[N + n], z = xb[2 * N + n];
  if (cg == 0) st4(xyz4 + i * 4, f32x4{x, y, z, (x * x + y * y) + z * z});
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float* wr = wpq + (size_t)(4 * cg + e) * ldw;
    o[e] = fmaf(wr[2], z, fmaf(wr[1], y, wr[0] * x)) + bpq[4 * cg + e];
  }
  st4(pq + i * ldpq + 4 * cg, o);
}

}  // namespace

extern "C" int vcr_rows4_pq_f32(const float* x_cf, float* xyz4, int B, int N, const float* wpq, int ldw, const float* bpq,
                                int C, float* pq, int ldpq, vcr_stream_t stream) {
  if (!x_cf || !xyz4 || !wpq || !bpq || !pq || B <= 0 || N <= 0) return VCR_EINVAL;
  if (C <= 0 || (C % 4) || 256 % (C / 4) || ldw < 3 || ldpq < C || (ldpq & 3) || ((uintptr_t)pq & 15)) return VCR_EINVAL;
  const int pts = 256 / (C / 4);
  hipLaunchKernelGGL(rows4_pq_kernel, dim3((unsigned)(((long)B * N + pts - 1) / pts)), dim3(256), 0, (hipStream_t)stream,
                     x_cf, xyz4, B, N, wpq, ldw, bpq, C, pq, ldpq);
  return VCR_LAUNCH_RC();
}

extern "C" int vcr_rows4_f32(const float* x_cf, float* xyz4, int B, int N, vcr_stream_t stream) {
  if (!x_cf || !xyz4 || B <= 0 || N <= 0) return VCR_EINVAL;
  hipLaunchKernelGGL(rows4_kernel, dim3((unsigned)(((long)B * N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_cf,
                     xyz4, B, N);
  return VCR_LAUNCH_RC();
}

extern "C" int vcr_pointwise_f32(const vcr_pointwise_args* a, vcr_stream_t stream) {
  if (!a || !a->x_cf || !a->w1 || !a->b1 || !a->w2 || !a->b2 || !a->xyz4 || !a->feat64 || !a->sq64) return VCR_EINVAL;
  if (a->B <= 0 || a->N <= 0 || a->B2 < 0 || (a->B2 > 0 && !a->x_cf2)) return VCR_EINVAL;
  dim3 grid((a->N + PTS - 1) / PTS, a->B + a->B2);
  hipLaunchKernelGGL(pointwise12_kernel, grid, dim3(256), 0, (hipStream_t)stream, *a);
  return VCR_LAUNCH_RC();
}
