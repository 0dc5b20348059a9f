// ICP refinement (the --iter 0 path): ICP.forward of model/icp_model.py:26-48 as ONE host call with no
// host synchronisation.  The reference tests convergence on the host every iteration
// (`torch.abs(prev_error - mean_error) < tol`, :37 -- a device->host sync per iteration); here the test
// runs on the device: the kernels of iteration i execute iff i <= state.stop, and the apply kernel of the
// iteration that converges sets state.stop = i (so that iteration's transform IS applied, like the
// reference's `break` after `src = transform(...)`).
//   nearest neighbour : arg-max_j (-|s_i|^2 + 2 s_i.d_j) - |d_j|^2  (:56-63), one lane per source point,
//                       destination points broadcast from LDS; block partial sums of the best values
//   best fit          : vcr_rigid_svd_f32's kernel (:75-108)
//   apply + converge  : s_i <- R s_i + t, batch-mean error from the partial sums in a fixed order
#include "common.h"

extern "C" int vcr_rigid_svd_f32(const vcr_rigid_svd_args* a, vcr_stream_t stream);

namespace {

struct IcpState { float prev_err; int stop; int iters; int pad; };

__global__ void icp_init_kernel(IcpState* st) { st->prev_err = 0.f; st->stop = 0x7fffffff; st->iters = 0; }

__global__ __launch_bounds__(256) void icp_nn_kernel(const float* src4, const float* dst4, float* cand4, float* partial,
                                                    const IcpState* st, int iter, int N, int M) {
  if (iter > st->stop) return;
  extern __shared__ __attribute__((aligned(16))) f32x4 dl[];
  __shared__ float red[4];
  const int b = blockIdx.y, t = threadIdx.x;
  const int i = blockIdx.x * 256 + t;
  const f32x4 s = ld4(src4 + ((size_t)b * N + min(i, N - 1)) * 4);
  float best = VCR_NEG_INF;
  int bidx = 0;
  for (int j0 = 0; j0 < M; j0 += 2048) {
    const int cnt = min(2048, M - j0);
    __syncthreads();
    for (int j = t; j < cnt; j += 256) dl[j] = ld4(dst4 + ((size_t)b * M + j0 + j) * 4);
    __syncthreads();
    for (int j = 0; j < cnt; ++j) {
      const f32x4 d = dl[j];
      const float dot = fmaf(s[2], d[2], fmaf(s[1], d[1], s[0] * d[0]));
      const float v = (2.f * dot - s[3]) - d[3];
      if (v > best) { best = v; bidx = j0 + j; }
    }
  }
  float val = 0.f;
  if (i < N) {
    st4(cand4 + ((size_t)b * N + i) * 4, ld4(dst4 + ((size_t)b * M + bidx) * 4));
    val = best;
  }
  val = wave_sum(val);
  if ((t & 63) == 0) red[t >> 6] = val;
  __syncthreads();
  if (t == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void icp_apply_kernel(float* src4, const float* R, const float* tr, const float* partial,
                                                       int npartial, IcpState* st, int iter, float tol, int B, int N) {
  if (iter > st->stop) return;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx < (long)B * N) {
    const int b = (int)(idx / N);
    const float* r = R + b * 9;
    const f32x4 p = ld4(src4 + idx * 4);
    const float x = fmaf(r[2], p[2], fmaf(r[1], p[1], r[0] * p[0])) + tr[b * 3 + 0];
    const float y = fmaf(r[5], p[2], fmaf(r[4], p[1], r[3] * p[0])) + tr[b * 3 + 1];
    const float z = fmaf(r[8], p[2], fmaf(r[7], p[1], r[6] * p[0])) + tr[b * 3 + 2];
    st4(src4 + idx * 4, f32x4{x, y, z, (x * x + y * y) + z * z});
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < npartial; ++i) s += partial[i];
    const float err = (float)(s / ((double)B * N));
    st->iters = iter + 1;
    if (fabsf(st->prev_err - err) < tol) st->stop = iter;     // this iteration's transform is still applied
    st->prev_err = err;
  }
}

__global__ void icp_finish_kernel(const IcpState* st, int* iters_out) { if (iters_out) *iters_out = st->iters; }

}  // namespace

extern "C" size_t vcr_icp_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  const size_t pts = (size_t)B * N * 4 * sizeof(float);
  return pts + 256 + (size_t)B * ((N + 255) / 256) * sizeof(float) + 256 + (size_t)B * 12 * sizeof(float) + 256 + 256;
}

extern "C" int vcr_icp_f32(const vcr_icp_args* a, void* workspace, size_t workspace_bytes, vcr_stream_t stream) {
  if (!a || !a->src4 || !a->dst4 || !a->final4 || !a->R || !a->t || !workspace) return VCR_EINVAL;
  if (a->B <= 0 || a->N < 3 || a->M < 1 || a->max_iterations < 1) return VCR_EINVAL;
  if (workspace_bytes < vcr_icp_workspace_bytes(a->B, a->N)) return VCR_EWORKSPACE;
  if (((uintptr_t)workspace) & 15) return VCR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const int nbx = (a->N + 255) / 256, npartial = a->B * nbx;
  float* cand4 = reinterpret_cast<float*>(w);                 w += up((size_t)a->B * a->N * 16);
  float* partial = reinterpret_cast<float*>(w);               w += up((size_t)npartial * 4);
  float* Rt = reinterpret_cast<float*>(w);                    w += up((size_t)a->B * 12 * 4);
  IcpState* st = reinterpret_cast<IcpState*>(w);
  float* cur4 = a->final4;                                    // iterate in place in the caller's output buffer
  { const int rc0 = vcr_copy_d2d(cur4, a->src4, (size_t)a->B * a->N * 16, s); if (rc0) return rc0; }
  hipLaunchKernelGGL(icp_init_kernel, dim3(1), dim3(1), 0, s, st);
  const size_t lds = (size_t)((a->M < 2048 ? a->M : 2048)) * 16;
  for (int it = 0; it < a->max_iterations; ++it) {
    hipLaunchKernelGGL(icp_nn_kernel, dim3(nbx, a->B), dim3(256), lds, s, cur4, a->dst4, cand4, partial, st, it, a->N, a->M);
    vcr_rigid_svd_args sv{cur4, 4, cand4, 4, a->B, a->N, Rt, Rt + a->B * 9, nullptr, nullptr, nullptr};
    const int rc = vcr_rigid_svd_f32(&sv, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(icp_apply_kernel, dim3((unsigned)(((long)a->B * a->N + 255) / 256)), dim3(256), 0, s, cur4, Rt,
                       Rt + a->B * 9, partial, npartial, st, it, a->tolerance, a->B, a->N);
  }
  vcr_rigid_svd_args fin{a->src4, 4, cur4, 4, a->B, a->N, a->R, a->t, a->R_ba, a->t_ba, nullptr};   // icp_model.py:42
  const int rc = vcr_rigid_svd_f32(&fin, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(icp_finish_kernel, dim3(1), dim3(1), 0, s, st, a->iterations);
  return VCR_LAUNCH_RC();
}
