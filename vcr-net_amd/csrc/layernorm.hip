// LayerNorm of model/transformer.py:141-144:  a_2 * (x - mean) / (std + eps) + b_2, with the
// UNBIASED std (torch.Tensor.std default) and eps added to the std, not the variance.
// One wave per 512-wide row (2 x 16 B per lane), two-pass in registers, HBM-bound.
// Optional fused tails: + residual (vcrnet_model.py:504-505) and the head's side record
// (x, y, z, |y|^2) so the correspondence kernel never re-reads the embedding for its norm.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void layernorm512_kernel(vcr_layernorm_args p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const float* x = p.x + (size_t)row * p.ldx;
  f32x4 v0 = ld4(x + lane * 4), v1 = ld4(x + 256 + lane * 4);
  float s = (v0[0] + v0[1]) + (v0[2] + v0[3]) + (v1[0] + v1[1]) + (v1[2] + v1[3]);
  const float mean = wave_sum(s) * (1.f / 512.f);
  float d[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { d[i] = v0[i] - mean; d[4 + i] = v1[i] - mean; }
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) ss = fmaf(d[i], d[i], ss);
  const float var = wave_sum(ss) * (1.f / 511.f);
  const float den = sqrtf(var) + p.eps;
  const f32x4 a0 = ld4(p.a + lane * 4), a1 = ld4(p.a + 256 + lane * 4);
  const f32x4 b0 = ld4(p.b + lane * 4), b1 = ld4(p.b + 256 + lane * 4);
  f32x4 y0, y1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    y0[i] = a0[i] * d[i] / den + b0[i];
    y1[i] = a1[i] * d[4 + i] / den + b1[i];
  }
  if (p.residual) {
    const float* r = p.residual + (size_t)row * p.ldr;
    const f32x4 r0 = ld4(r + lane * 4), r1 = ld4(r + 256 + lane * 4);
    y0 = r0 + y0; y1 = r1 + y1;
  }
  float* y = p.y + (size_t)row * p.ldy;
  st4(y + lane * 4, y0);
  st4(y + 256 + lane * 4, y1);
  if (p.side4) {
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { q = fmaf(y0[i], y0[i], q); q = fmaf(y1[i], y1[i], q); }
    q = wave_sum(q);
    if (lane == 0) {
      const f32x4 xyz = ld4(p.xyz4 + (size_t)row * 4);
      st4(p.side4 + (size_t)row * 4, f32x4{xyz[0], xyz[1], xyz[2], q});
    }
  }
}

// y = scale * x (optional, in place allowed) and side4 = (xyz, |y|^2): the head's side record when the
// embedding does not come out of a LayerNorm (pointer == None / Identity, vcrnet_model.py:477-482).
__global__ __launch_bounds__(256) void rowside_kernel(vcr_rowside_args p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const float* x = p.x + (size_t)row * p.ldx;
  float q = 0.f;
  for (int c = lane * 4; c < p.C; c += 256) {
    f32x4 v = ld4(x + c) * p.scale;
    if (p.y) st4(p.y + (size_t)row * p.ldy + c, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) q = fmaf(v[i], v[i], q);
  }
  q = wave_sum(q);
  if (lane == 0) {
    const f32x4 xyz = ld4(p.xyz4 + (size_t)row * 4);
    st4(p.side4 + (size_t)row * 4, f32x4{xyz[0], xyz[1], xyz[2], q});
  }
}

// Fold a LayerNorm's affine into the Linear that consumes it (once per weight, at pack time):
//   w_out[n,k] = w[n,k] * a[k];  colsum[n] = sum_k w_out[n,k];  bias_out[n] = bias[n] + sum_k w[n,k] * b[k]
// One wave per output row; the two sums are accumulated in fp64 and rounded once.
__global__ __launch_bounds__(256) void fold_layernorm_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                             const float* __restrict__ a, const float* __restrict__ b,
                                                             int N, int K, float* w_out, float* colsum, float* bias_out) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  double c = 0.0, d = 0.0;
  for (int k = lane; k < K; k += 64) {
    const float wv = w[(size_t)n * K + k];
    const float wf = wv * a[k];
    w_out[(size_t)n * K + k] = wf;
    c += (double)wf;
    d += (double)wv * (double)b[k];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { c += __shfl_xor(c, off, 64); d += __shfl_xor(d, off, 64); }
  if (lane == 0) { colsum[n] = (float)c; bias_out[n] = (float)((bias ? (double)bias[n] : 0.0) + d); }
}

}  // namespace

extern "C" int vcr_fold_layernorm_f32(const float* w, const float* bias, const float* ln_a, const float* ln_b, int N,
                                      int K, float* w_out, float* colsum, float* bias_out, vcr_stream_t stream) {
  if (!w || !ln_a || !ln_b || !w_out || !colsum || !bias_out || N <= 0 || K <= 0) return VCR_EINVAL;
  hipLaunchKernelGGL(fold_layernorm_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, w, bias, ln_a, ln_b,
                     N, K, w_out, colsum, bias_out);
  return VCR_LAUNCH_RC();
}

extern "C" int vcr_rowside_f32(const vcr_rowside_args* a, vcr_stream_t stream) {
  if (!a || !a->x || !a->xyz4 || !a->side4 || a->M <= 0 || a->C <= 0 || (a->C & 3) || (a->ldx & 3)) return VCR_EINVAL;
  hipLaunchKernelGGL(rowside_kernel, dim3((a->M + 3) / 4), dim3(256), 0, (hipStream_t)stream, *a);
  return VCR_LAUNCH_RC();
}

extern "C" int vcr_layernorm_f32(const vcr_layernorm_args* a, vcr_stream_t stream) {
  if (!a || !a->x || !a->a || !a->b || !a->y || a->M <= 0) return VCR_EINVAL;
  if (a->C != 512) return VCR_EUNSUPPORTED;
  if ((a->ldx & 3) || (a->ldy & 3) || (a->residual && (a->ldr & 3))) return VCR_EINVAL;
  if ((a->side4 == nullptr) != (a->xyz4 == nullptr)) return VCR_EINVAL;
  hipLaunchKernelGGL(layernorm512_kernel, dim3((a->M + 3) / 4), dim3(256), 0, (hipStream_t)stream, *a);
  return VCR_LAUNCH_RC();
}
