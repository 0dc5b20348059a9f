// Discrete selection steps of the partial-overlap path:
//   rankselect : per-sample top-K of n scalar scores, in descending order (ties -> lower index), the
//                arithmetic of Tensor.topk at model/transformer.py:42 and model/vcrnet_model.py:223,245,312.
//                Every item computes its exact rank against the whole row held in LDS (n <= 16384: no
//                sort, deterministic); a sample is spread over n/64 one-wave blocks so the chip is filled even at
//                small batch (24 samples x 12 blocks at config 3).
//   gather_rows: out[b][r] = in[b][idx[b][r]]  (vcrnet_model.py:230-260,305-330 index gathers).
#include "common.h"

namespace {

// One wave per 64 candidates j (a sample is spread over n/64 single-wave blocks: 288 at BASELINE configs[2], one per CU);
// the row sits in LDS padded to a multiple of 4 with values that never outrank anything and is read as broadcast
// ds_read_b128.  The tie rule (equal values: lower index first) only needs the index comparison inside the wave's own
// 64-wide window of i: below it "before" is >=, above it >, one compare + one add-with-carry per candidate pair
// (the loop is bound by VALU issue: the first version spent ~8 instructions per pair, 25 us per launch at n = 768).
template <bool LARGEST>
__global__ __launch_bounds__(64) void rankselect_kernel(vcr_rankselect_args p) {
  extern __shared__ __attribute__((aligned(16))) float vals[];
  const int b = blockIdx.y, t = threadIdx.x;
  const int stride = p.stride > 1 ? p.stride : 1;
  const float* v = p.values + (size_t)b * p.n * stride;
  const int n4 = (p.n + 3) & ~3;
  for (int i = t; i < n4; i += 64) vals[i] = i < p.n ? v[(size_t)i * stride] : (LARGEST ? VCR_NEG_INF : -VCR_NEG_INF);
  __syncthreads();
  const int j0 = blockIdx.x * 64, j = j0 + t;            // j0: block-uniform, a multiple of 4
  const float vj = vals[min(j, p.n - 1)];
  int rank = 0;
#pragma unroll 4
  for (int i = 0; i < j0; i += 4) {                      // i < every j of this wave: ties count
    const f32x4 vi = ld4(&vals[i]);
#pragma unroll
    for (int e = 0; e < 4; ++e) rank += (LARGEST ? vi[e] >= vj : vi[e] <= vj) ? 1 : 0;
  }
  const int j1 = min(j0 + 64, n4);
  for (int i = j0; i < j1; i += 4) {                     // the wave's own window: the full rule
    const f32x4 vi = ld4(&vals[i]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool before = LARGEST ? vi[e] > vj : vi[e] < vj;
      rank += (before || (vi[e] == vj && i + e < j)) ? 1 : 0;
    }
  }
#pragma unroll 4
  for (int i = j1; i < n4; i += 4) {                     // i > every j of this wave: ties do not count
    const f32x4 vi = ld4(&vals[i]);
#pragma unroll
    for (int e = 0; e < 4; ++e) rank += (LARGEST ? vi[e] > vj : vi[e] < vj) ? 1 : 0;
  }
  if (j >= p.n) return;
  if (p.order && rank < p.K) p.order[(size_t)b * p.K + rank] = j;
  if (p.mask) p.mask[(size_t)b * p.n + j] = rank < p.K ? 1 : 0;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(vcr_gather_args p) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)p.nbatch * p.n_out) return;
  const int b = (int)(row / p.n_out);
  int src = p.idx[row];
  if (p.via) src = p.via[(size_t)b * p.n_via + src];
  const float* in = p.in + ((size_t)b * p.n_in + src) * p.ld_in;
  float* out = p.out + (size_t)row * p.ld_out;
  for (int c = lane * 4; c < p.C; c += 256) st4(out + c, ld4(in + c));
}

}  // namespace

extern "C" int vcr_rankselect_f32(const vcr_rankselect_args* a, vcr_stream_t stream) {
  if (!a || !a->values || (!a->order && !a->mask)) return VCR_EINVAL;
  if (a->nbatch <= 0 || a->n <= 0 || a->K <= 0 || a->K > a->n || a->n > 16384 || a->stride < 0) return VCR_EINVAL;
  const dim3 grid((a->n + 63) / 64, a->nbatch);
  const size_t lds = (size_t)((a->n + 3) & ~3) * 4;
  if (a->largest) hipLaunchKernelGGL(rankselect_kernel<true>, grid, dim3(64), lds, (hipStream_t)stream, *a);
  else hipLaunchKernelGGL(rankselect_kernel<false>, grid, dim3(64), lds, (hipStream_t)stream, *a);
  return VCR_LAUNCH_RC();
}

extern "C" int vcr_gather_rows_f32(const vcr_gather_args* a, vcr_stream_t stream) {
  if (!a || !a->in || !a->idx || !a->out) return VCR_EINVAL;
  if (a->nbatch <= 0 || a->n_in <= 0 || a->n_out <= 0 || a->C <= 0 || (a->C & 3) || (a->ld_in & 3) || (a->ld_out & 3))
    return VCR_EINVAL;
  if (a->via && a->n_via <= 0) return VCR_EINVAL;
  const long rows = (long)a->nbatch * a->n_out;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a);
  return VCR_LAUNCH_RC();
}
