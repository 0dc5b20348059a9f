// Discrete selection steps of the partial-overlap path:
//   rankselect : per-sample top-K of n scalar scores, in descending order (ties -> lower index), the
//                arithmetic of Tensor.topk at model/transformer.py:42 and model/vcrnet_model.py:223,245,312.
//                Every item computes its exact rank against the whole row held in LDS (n <= 16384: no
//                sort, deterministic); a sample is spread over n/256 blocks so the chip is filled even at
//                small batch (24 samples x 3 chunks at config 3).
//   gather_rows: out[b][r] = in[b][idx[b][r]]  (vcrnet_model.py:230-260,305-330 index gathers).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void rankselect_kernel(vcr_rankselect_args p) {
  extern __shared__ __attribute__((aligned(16))) float vals[];
  const int b = blockIdx.y, t = threadIdx.x;
  const int stride = p.stride > 1 ? p.stride : 1;
  const float* v = p.values + (size_t)b * p.n * stride;
  for (int i = t; i < p.n; i += 256) vals[i] = v[(size_t)i * stride];
  __syncthreads();
  const int j = blockIdx.x * 256 + t;
  if (j >= p.n) return;
  const float vj = vals[j];
  int rank = 0;
  if (p.largest) {
    for (int i = 0; i < p.n; ++i) { const float vi = vals[i]; rank += (vi > vj || (vi == vj && i < j)) ? 1 : 0; }
  } else {
    for (int i = 0; i < p.n; ++i) { const float vi = vals[i]; rank += (vi < vj || (vi == vj && i < j)) ? 1 : 0; }
  }
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
  hipLaunchKernelGGL(rankselect_kernel, dim3((a->n + 255) / 256, a->nbatch), dim3(256), (size_t)a->n * 4,
                     (hipStream_t)stream, *a);
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
