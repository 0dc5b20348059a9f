// Evaluation-pair construction on the device: the arithmetic of ModelNet40.__getitem__
// (util/data.py:247-314, test partition) and its partial crop nearest_neighbor (util/data.py:320-329).
//
// The random draws themselves (Euler angles, translation, three permutations) stay on the host: the
// reference seeds legacy NumPy with the item index (util/data.py:255-256) and a Mersenne-Twister replay on the
// GPU would buy nothing -- they are 3 N int32 + 12 doubles per item.  Everything that touches the POINTS runs
// here, so a batch goes from base clouds resident in HBM to network inputs without a host round trip:
//   src_i = cloud[pick[perm_src[i]]]                              float32                (:289, :298)
//   tgt_i = R_ab cloud[pick[perm_tgt[i]]] + t_ab                  float64 fma chain like the host dgemm (:290-291, :301)
//   crop  : keep the `keep` points nearest to the LAST point, ordered by distance (stable)   (:320-329);
//           distances in the cloud's own precision (float32 for src, float64 for tgt), exact rank by counting.
// One block per (side, item); the points and their distances live in LDS (N <= 4096).
#include "common.h"

namespace {

template <typename T>
__device__ void build_side(const vcr_make_pairs_args& p, int b, bool is_tgt, unsigned char* smem) {
  T* px = reinterpret_cast<T*>(smem);
  T* py = px + p.N; T* pz = py + p.N; T* d = pz + p.N;
  const int t = threadIdx.x;
  const int32_t* perm = (is_tgt ? p.perm_tgt : p.perm_src) + (size_t)b * p.N;
  const int32_t* pick = p.pick + (size_t)b * p.N;
  const float* cloud = p.cloud + (size_t)b * p.P * 3;
  double r[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tr[3] = {0, 0, 0};
  if (is_tgt) {
    for (int i = 0; i < 9; ++i) r[i] = p.R_ab[b * 9 + i];
    for (int i = 0; i < 3; ++i) tr[i] = p.t_ab[b * 3 + i];
  }
  for (int i = t; i < p.N; i += blockDim.x) {
    const float* c = cloud + (size_t)pick[perm[i]] * 3;
    if (is_tgt) {
      const double x = c[0], y = c[1], z = c[2];
      px[i] = (T)(fma(r[2], z, fma(r[1], y, r[0] * x)) + tr[0]);
      py[i] = (T)(fma(r[5], z, fma(r[4], y, r[3] * x)) + tr[1]);
      pz[i] = (T)(fma(r[8], z, fma(r[7], y, r[6] * x)) + tr[2]);
    } else {
      px[i] = (T)c[0]; py[i] = (T)c[1]; pz[i] = (T)c[2];
    }
  }
  __syncthreads();
  float* out = (is_tgt ? p.tgt_cf : p.src_cf) + (size_t)b * 3 * p.keep;
  if (p.keep == p.N) {
    for (int i = t; i < p.N; i += blockDim.x) {
      out[i] = (float)px[i]; out[p.keep + i] = (float)py[i]; out[2 * p.keep + i] = (float)pz[i];
    }
    return;
  }
  const T qx = px[p.N - 1], qy = py[p.N - 1], qz = pz[p.N - 1];
  for (int i = t; i < p.N; i += blockDim.x) {
    const T dx = px[i] - qx, dy = py[i] - qy, dz = pz[i] - qz;
    d[i] = (dx * dx + dy * dy) + dz * dz;                     // np.sum over 3 terms, no contraction
  }
  __syncthreads();
  for (int j = t; j < p.N; j += blockDim.x) {
    const T dj = d[j];
    int rank = 0;
    for (int i = 0; i < p.N; ++i) { const T di = d[i]; rank += (di < dj || (di == dj && i < j)) ? 1 : 0; }
    if (rank < p.keep) {
      out[rank] = (float)px[j]; out[p.keep + rank] = (float)py[j]; out[2 * p.keep + rank] = (float)pz[j];
    }
  }
}

__global__ __launch_bounds__(256) void make_pairs_kernel(vcr_make_pairs_args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (blockIdx.x == 0) build_side<float>(p, blockIdx.y, false, smem);
  else build_side<double>(p, blockIdx.y, true, smem);
}

}  // namespace

extern "C" int vcr_make_pairs_f32(const vcr_make_pairs_args* a, vcr_stream_t stream) {
  if (!a || !a->cloud || !a->R_ab || !a->t_ab || !a->pick || !a->perm_src || !a->perm_tgt || !a->src_cf || !a->tgt_cf)
    return VCR_EINVAL;
  if (a->B <= 0 || a->N < 2 || a->N > 4096 || a->P < a->N || a->keep < 1 || a->keep > a->N) return VCR_EINVAL;
  const int lds = a->N * 4 * (int)sizeof(double);
  VCR_DYN_LDS(make_pairs_kernel, lds);
  hipLaunchKernelGGL(make_pairs_kernel, dim3(2, a->B), dim3(256), lds, (hipStream_t)stream, *a);
  return VCR_LAUNCH_RC();
}
