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
  {                                                     // w2[c][k] -> [k][c]: four 16-B loads per thread, all issued before any is consumed
    f32x4 wv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) wv[u] = ld4(a.w2 + 4 * (t + 256 * u));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = 4 * (t + 256 * u), c = i >> 6, k = i & 63;
#pragma unroll
      for (int e = 0; e < 4; ++e) w2t[k + e][c] = wv[u][e];
    }
  }
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
    if (a.feat64t) {                                     // the 4 x 4 transposed copy of this lane's 16 channels (vcr_knn_args.xt)
      float* ot = a.feat64t + row * 64 + g * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) st4(ot + q * 4, f32x4{acc[q], acc[4 + q], acc[8 + q], acc[12 + q]});
    }
    if (g == 0) {
      a.sq64[row] = ss;
      st4(a.xyz4 + row * 4, f32x4{x, y, z, (x * x + y * y) + z * z});   // torch.sum order for 3 rows: sequential
    }
  }
}

// The same outputs, bit for bit, with conv2 on the matrix pipe, and the first EdgeConv projection of LPDNet
// (lpdnet_model.py:122 after the SURVEY-F7 split: P | Q = feat64 Wpq^T + bpq, K = 64, N = 256) computed from the tile while it
// is in LDS -- the separate pointwise launch + `linear:dg1_pq` launch (K = 64: 52 TFLOP/s) of the forward become one.
//   v_mfma_f32_16x16x4_f32 is a k-ascending fma chain that STARTS at its accumulator operand: with the accumulator preset
//   to the bias and k = 4 s + (lane >> 4) ascending over the 16 steps, D is exactly "bias first, one fma per input channel
//   in ascending order" -- the rounding order of the reference's conv that pointwise12_kernel spells out on the VALU.
// Wave = tiles of 16 points (MFMA rows).  conv1 (K = 3) stays on the VALU: lane (q4, l15) computes h1[point l15][k = 4 s + q4],
// which IS its A operand of step s -- no staging.  conv2's weights sit in 64 VGPRs as B fragments; D goes through a
// per-wave LDS tile [16][64] (ReLU applied), from which (a) four lanes per point read 16 consecutive channels each and run
// pointwise12_kernel's epilogue unchanged (|feat|^2 in ATen's association, 16-B stores) and (b) the lanes re-read A
// fragments for the P | Q projection, whose 64 KB of weights lie in LDS in fragment order (one ds_read_b128 per four MFMAs).
// "//@probe ..." lines: inert here, uncommented by profiles/experiments/probe_build.py (phase clocks of waves 0 and 7).
constexpr int TP = 68;                                   // tile pitch: the 64 lanes' A-fragment reads hit 64 distinct banks
constexpr int PQW = 8;                                   // waves per workgroup (two per SIMD: one wave's LDS / store phases under the other's MFMAs)
__global__ __launch_bounds__(64 * PQW, 1) void pointwise12_pq_kernel(vcr_pointwise_args a, int tiles_per_cloud, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) float pw_smem[];
  //@probe VCR_PROBE_ACC_DECL;
  float* wfrag = pw_smem;                                // [16 col tiles][4 step quads][64 lanes][4 steps]: Wpq as B fragments
  float* w1s = wfrag + 16 * 4 * 64 * 4;                  // [64][4] = (w0, w1, w2, b1)
  float* pqb = w1s + 64 * 4;                             // [256] P | Q bias (an LDS read per accumulator instead of an L2 round trip)
  float* tiles = pqb + 256;                              // [PQW waves][16][TP]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int q4 = lane >> 4, l15 = lane & 15;
  // The prologue is a third of this launch (every wave runs ONE tile at the path's sizes): all its loads are issued
  // back to back as 16-B requests -- conv2's B fragments first, then this thread's eight chunks of Wpq -- and only then
  // consumed (the first version loaded Wpq dword by dword with an LDS write behind every load, and w2 as 64 strided
  // dwords per lane: 15 us before the first MFMA).
  f32x4 w2c[4][4];                                       // w2[16 j + l15][16 g + 4 q4 .. + 3]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) w2c[j][g] = ld4(a.w2 + (16 * j + l15) * 64 + 16 * g + 4 * q4);
  float b2r[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) b2r[j] = a.b2[16 * j + l15];
  f32x4 wq[8];                                           // Wpq [256][64] as 4096 chunks of 16 B: chunks t, t + 512, ...
#pragma unroll
  for (int it = 0; it < 8; ++it) wq[it] = ld4(a.pq_w + 4 * (t + 64 * PQW * it));
  if (t >= 256) pqb[t - 256] = a.pq_b[t - 256];
  if (t < 64) {
    w1s[t * 4 + 0] = a.w1[t * 3 + 0]; w1s[t * 4 + 1] = a.w1[t * 3 + 1]; w1s[t * 4 + 2] = a.w1[t * 3 + 2];
    w1s[t * 4 + 3] = a.b1[t];
  }
  // conv2 B fragments w2[16 j + l15][4 s + q4]: the 4 x 4 transpose between lane rows and chunk components that the kNN
  // kernel describes (v_permlane16_swap on (0,1) (2,3), v_permlane32_swap on (0,2) (1,3))
  float w2r[4][16];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int r0 = __float_as_int(w2c[j][g][0]), r1 = __float_as_int(w2c[j][g][1]);
      int r2 = __float_as_int(w2c[j][g][2]), r3 = __float_as_int(w2c[j][g][3]);
      auto s01 = __builtin_amdgcn_permlane16_swap(r0, r1, false, false); r0 = s01[0]; r1 = s01[1];
      auto s23 = __builtin_amdgcn_permlane16_swap(r2, r3, false, false); r2 = s23[0]; r3 = s23[1];
      auto s02 = __builtin_amdgcn_permlane32_swap(r0, r2, false, false); r0 = s02[0]; r2 = s02[1];
      auto s13 = __builtin_amdgcn_permlane32_swap(r1, r3, false, false); r1 = s13[0]; r3 = s13[1];
      w2r[j][4 * g] = __int_as_float(r0); w2r[j][4 * g + 1] = __int_as_float(r1);
      w2r[j][4 * g + 2] = __int_as_float(r2); w2r[j][4 * g + 3] = __int_as_float(r3);
    }
  // Wpq into fragment order: chunk c holds Wpq[n = c / 16][k = 4 (c % 16) .. + 3], i.e. MFMA step st = c % 16, lane rows 0..3
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int c = t + 64 * PQW * it, n = c >> 4, st = c & 15, j = n >> 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) wfrag[(((j * 4 + (st >> 2)) * 64) + e * 16 + (n & 15)) * 4 + (st & 3)] = wq[it][e];
  }
  //@probe VCR_PROBE_ACC(0);                                                // weight loads issued
  __syncthreads();
  //@probe VCR_PROBE_ACC(1);                                                // ... arrived, LDS filled
  float* T = tiles + wave * 16 * TP;
  const int p = lane >> 2, g = lane & 3;                 // epilogue mapping: four lanes per point, 16 channels each
  for (int tile = (int)blockIdx.x * PQW + wave; tile < total_tiles; tile += (int)gridDim.x * PQW) {
    const int b = tile / tiles_per_cloud, n0 = (tile - b * tiles_per_cloud) * 16;
    const float* xb = b < a.B ? a.x_cf + (size_t)b * 3 * a.N : a.x_cf2 + (size_t)(b - a.B) * 3 * a.N;
    float h[16];
    {
      const int nc = min(n0 + l15, a.N - 1);
      const float x = xb[nc], y = xb[a.N + nc], z = xb[2 * a.N + nc];
#pragma unroll
      for (int st = 0; st < 16; ++st) {                  // conv1 + ReLU for this lane's k = 4 st + q4 (same chain as pointwise12_kernel)
        const f32x4 w = ld4(&w1s[(4 * st + q4) * 4]);
        h[st] = fmaxf(fmaf(w[2], z, fmaf(w[1], y, fmaf(w[0], x, w[3]))), 0.f);
      }
    }
    //@probe VCR_PROBE_ACC(2);                                              // x loads + conv1
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{b2r[j], b2r[j], b2r[j], b2r[j]};   // bias first
#pragma unroll
    for (int st = 0; st < 16; ++st)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = mfma16(h[st], w2r[j][st], acc[j]);     // k ascending
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(4 * q4 + r) * TP + 16 * j + l15] = fmaxf(acc[j][r], 0.f);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    //@probe VCR_PROBE_ACC(3);                                              // conv2 MFMAs + tile to LDS
    // The two waves of a SIMD (w, w + 4) take the two halves of the tile's work in opposite order -- the feature epilogue
    // (LDS reads, global stores) and the P | Q product (256 MFMAs) -- so that one wave's MFMAs run under the other's stores
    // instead of all eight waves queueing for the matrix pipe at once.  Both halves only read the tile.
    auto feature_epilogue = [&]() {
      const int n = n0 + p;
      const bool live = n < a.N;
      const int nc = live ? n : a.N - 1;
      const float x = xb[nc], y = xb[a.N + nc], z = xb[2 * a.N + nc];
      float f[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = ld4(&T[p * TP + 16 * g + 4 * q]);
        f[4 * q] = v[0]; f[4 * q + 1] = v[1]; f[4 * q + 2] = v[2]; f[4 * q + 3] = v[3];
      }
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) ss = ss + f[j] * f[j];
      const int base = lane & ~3;
      {
        const float s0 = __shfl(ss, base, 64), s1 = __shfl(ss, base + 1, 64);
        const float s2 = __shfl(ss, base + 2, 64), s3 = __shfl(ss, base + 3, 64);
        ss = ((s0 + s1) + s2) + s3;
      }
      const bool tail = live && n >= (a.N & ~31);        // (ATen's scalar tail loop: see pointwise12_kernel)
      if (__builtin_amdgcn_ballot_w64(tail) != 0) {
        float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sgrp = 0; sgrp < 4; ++sgrp) {
          if (g == sgrp) {
#pragma unroll
            for (int j = 0; j < 16; ++j) a4[j & 3] = a4[j & 3] + f[j] * f[j];
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
        for (int q = 0; q < 4; ++q) st4(o + q * 4, f32x4{f[q * 4], f[q * 4 + 1], f[q * 4 + 2], f[q * 4 + 3]});
        if (a.feat64t) {                                 // the 4 x 4 transposed copy of this lane's 16 channels (vcr_knn_args.xt)
          float* ot = a.feat64t + row * 64 + g * 16;
#pragma unroll
          for (int q = 0; q < 4; ++q) st4(ot + q * 4, f32x4{f[q], f[4 + q], f[8 + q], f[12 + q]});
        }
        if (g == 0) {
          a.sq64[row] = ss;
          st4(a.xyz4 + row * 4, f32x4{x, y, z, (x * x + y * y) + z * z});
        }
      }
    };
    auto pq_product = [&]() {
    // ---- P | Q = feat64 Wpq^T + bpq for the tile's 16 points (bias first, k ascending)
    float af[16];
#pragma unroll
    for (int st = 0; st < 16; ++st) af[st] = T[l15 * TP + 4 * st + q4];
#pragma unroll 1
    for (int jg = 0; jg < 4; ++jg) {
      f32x4 pa[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float bq = pqb[16 * (4 * jg + jj) + l15];
        pa[jj] = f32x4{bq, bq, bq, bq};
      }
#pragma unroll
      for (int sq = 0; sq < 4; ++sq)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const f32x4 wf = ld4(&wfrag[(((4 * jg + jj) * 4 + sq) * 64 + lane) * 4]);
#pragma unroll
          for (int e = 0; e < 4; ++e) pa[jj] = mfma16(af[4 * sq + e], wf[e], pa[jj]);
        }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + 4 * q4 + r;
          if (n < a.N) a.pq[((size_t)b * a.N + n) * a.ldpq + 16 * (4 * jg + jj) + l15] = pa[jj][r];
        }
    }
    };
    if (wave < PQW / 2) {
      feature_epilogue();
      //@probe VCR_PROBE_ACC(4);
      pq_product();
    } else {
      pq_product();
      //@probe VCR_PROBE_ACC(4);
      feature_epilogue();
    }
    //@probe VCR_PROBE_ACC(5);                                              // P | Q MFMAs + stores issued
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // the tile is rewritten by the next iteration
    __builtin_amdgcn_wave_barrier();
    //@probe __builtin_amdgcn_s_waitcnt(0);              // (probe builds: the stores' acknowledgement is a phase of its own)
    //@probe VCR_PROBE_ACC(6);
  }
  //@probe VCR_PROBE_ACC_FLUSH(threadIdx.x == 0 || threadIdx.x == 448, blockIdx.x * 2 + (threadIdx.x != 0));
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
  vcr_stream_scope bound(stream);
  if (!a || !a->x_cf || !a->w1 || !a->b1 || !a->w2 || !a->b2 || !a->xyz4 || !a->feat64 || !a->sq64) return VCR_EINVAL;
  if (((uintptr_t)a->w2 | (uintptr_t)a->pq_w) & 15) return VCR_EINVAL;   // the weights are fetched as 16-B chunks
  if (a->B <= 0 || a->N <= 0 || a->B2 < 0 || (a->B2 > 0 && !a->x_cf2)) return VCR_EINVAL;
  if (a->pq) {                                           // fused with the P | Q projection: the MFMA kernel
    if (!a->pq_w || !a->pq_b || a->ldpq < 256 || (a->ldpq & 3)) return VCR_EINVAL;
    const int tpc = (a->N + 15) / 16, total = tpc * (a->B + a->B2);
    const int cus = vcr_cu_count(), blocks = (total + PQW - 1) / PQW < cus ? (total + PQW - 1) / PQW : cus;
    const size_t lds = (size_t)(16 * 4 * 64 * 4 + 64 * 4 + 256 + PQW * 16 * TP) * sizeof(float);
    VCR_DYN_LDS(pointwise12_pq_kernel, lds);
    hipLaunchKernelGGL(pointwise12_pq_kernel, dim3(blocks), dim3(64 * PQW), lds, (hipStream_t)stream, *a, tpc, total);
    return VCR_LAUNCH_RC();
  }
  dim3 grid((a->N + PTS - 1) / PTS, a->B + a->B2);
  hipLaunchKernelGGL(pointwise12_kernel, grid, dim3(256), 0, (hipStream_t)stream, *a);
  return VCR_LAUNCH_RC();
}
