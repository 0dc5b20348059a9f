// Kernel 4: centred 3x3 cross-covariance + SVD rigid solve (model/vcrnet_model.py:356-399).
// The reference loops over the batch in Python with one LAPACK call and one host sync (det < 0) per
// sample; here one 256-thread block per sample reduces the means and the covariance (fp64 accumulation of the fp32
// inputs; all six / all nine sums cross the block together: four barriers in total), and the first quad of wave 0 runs
// the one-sided Jacobi SVD in fp64 cooperatively: lane i owns row i of A and V, the column inner products of a
// rotation are summed over the quad with DPP, every lane rotates its own row.  Lane 0 gathers the rows for the tail
// (ordering, rank completion, R = V U^T, determinant rule).
//   H = sum_k (s_k - s_mean)(c_k - c_mean)^T,  H = U S V^T,  R = V U^T,
//   det R < 0  ->  flip the column of V that belongs to the SMALLEST singular value (torch.svd sorts
//   descending, so the reference's V @ diag(1,1,-1) is exactly that; :382-386),  t = -R s_mean + c_mean.
// R = V U^T does not depend on the SVD's sign/order conventions while the singular values are distinct,
// so LAPACK-vs-Jacobi differences do not leak into (R, t).
#include "common.h"

namespace {

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// NV sums across the block at once: wave trees, one LDS hand-off, fixed order over the four waves
template <int NV>
__device__ __forceinline__ void block_sums(double (&v)[NV], double* red) {
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < NV; ++e) v[e] = wave_sum_f64(v[e]);
  __syncthreads();
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int e = 0; e < NV; ++e) red[w * NV + e] = v[e];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < NV; ++e) v[e] = ((red[e] + red[NV + e]) + red[2 * NV + e]) + red[3 * NV + e];
}
// sum over the four lanes of a DPP quad (two butterflies on the 32-bit halves of a double); same value in every lane
__device__ __forceinline__ double quad_xor(double v, int ctrl_is_1) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  if (ctrl_is_1) { lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true); }
  else { lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true); }
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_sum(double v) {
  v += quad_xor(v, 1);                                   // lanes (0,1) (2,3)
  v += quad_xor(v, 0);                                   // + the other pair: (x0 + x1) + (x2 + x3) in every lane
  return v;
}
__device__ __forceinline__ double quad_get(double v, int lane_in_quad) {   // lane_in_quad: compile-time 0..2
  int lo = __double2loint(v), hi = __double2hiint(v);
  const int c = lane_in_quad == 0 ? 0x00 : lane_in_quad == 1 ? 0x55 : 0xAA;
  if (lane_in_quad == 0) { lo = __builtin_amdgcn_mov_dpp(lo, 0x00, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x00, 0xF, 0xF, true); }
  else if (lane_in_quad == 1) { lo = __builtin_amdgcn_mov_dpp(lo, 0x55, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x55, 0xF, 0xF, true); }
  else { lo = __builtin_amdgcn_mov_dpp(lo, 0xAA, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xAA, 0xF, 0xF, true); }
  (void)c;
  return __hiloint2double(hi, lo);
}

// One-sided Jacobi sweeps on the quad: lane i < 3 holds a[0..2] = row i of A (initially H) and v[0..2] = row i of V
// (initially I); lane 3 holds zeros.  The control flow is uniform over the quad (the inner products are quad sums).
__device__ __forceinline__ void jacobi_sweeps_quad(double (&a)[3], double (&v)[3]) {
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = 0.0;
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = (pq == 2) ? 1 : 0, q = (pq == 0) ? 1 : 2;
      const double al = quad_sum(a[p] * a[p]), be = quad_sum(a[q] * a[q]), ga = quad_sum(a[p] * a[q]);
      if (fabs(ga) <= 1e-30 + 1e-17 * sqrt(al * be)) continue;
      off = fmax(off, fabs(ga) / sqrt(al * be + 1e-300));
      const double zeta = (be - al) / (2.0 * ga);
      const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
      const double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
      const double ap = a[p], aq = a[q];
      a[p] = c * ap - s * aq; a[q] = s * ap + c * aq;
      const double vp = v[p], vq = v[q];
      v[p] = c * vp - s * vq; v[q] = s * vp + c * vq;
    }
    if (off < 1e-15) break;
  }
}

// Tail of the solve on one lane, from the converged A (columns = singular vectors times singular values) and V
__device__ void svd3_finish(const double A[3][3], const double V[3][3], double R[9]) {
  double sig[3];
  int ord[3] = {0, 1, 2};
  for (int j = 0; j < 3; ++j) sig[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2 - i; ++j)
      if (sig[ord[j]] < sig[ord[j + 1]]) { const int tmp = ord[j]; ord[j] = ord[j + 1]; ord[j + 1] = tmp; }
  double U[3][3], W[3][3];
  for (int j = 0; j < 3; ++j) {
    const int c = ord[j];
    const double inv = sig[c] > 0 ? 1.0 / sig[c] : 0.0;
    for (int i = 0; i < 3; ++i) { U[i][j] = A[i][c] * inv; W[i][j] = V[i][c]; }
  }
  // Vanishing singular values leave their left vectors undefined: complete the frame so that R is always a proper
  // rotation.  Rank 2 (coplanar pairs): u2 = u0 x u1, either sign gives the same R after the determinant rule
  // below.  Rank 1 / rank 0 (all pairs on a line / one point; happens when tiny partial clouds collapse in later
  // vcrnetIter passes): R is not unique -- LAPACK's choice in the reference is arbitrary too -- we take the
  // completion closest to the coordinate axes, and R = I for H = 0.
  const double s0 = sig[ord[0]];
  if (!(s0 > 1e-300)) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) U[i][j] = W[i][j];
  } else {
    if (sig[ord[1]] <= 1e-12 * s0) {
      int e = 0;
      for (int i = 1; i < 3; ++i) if (fabs(U[i][0]) < fabs(U[e][0])) e = i;
      double v[3] = {0, 0, 0}, n2 = 0;
      v[e] = 1.0;
      const double d = U[e][0];
      for (int i = 0; i < 3; ++i) { v[i] -= d * U[i][0]; n2 += v[i] * v[i]; }
      const double inv = 1.0 / sqrt(n2);
      for (int i = 0; i < 3; ++i) U[i][1] = v[i] * inv;
    }
    if (sig[ord[2]] <= 1e-12 * s0) {
      U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
      U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
      U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    }
  }
  auto build = [&]() {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) R[3 * i + j] = W[i][0] * U[j][0] + W[i][1] * U[j][1] + W[i][2] * U[j][2];
  };
  build();
  const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
                     R[2] * (R[3] * R[7] - R[4] * R[6]);
  if (det < 0) {
    for (int i = 0; i < 3; ++i) W[i][2] = -W[i][2];
    build();
  }
}

__global__ __launch_bounds__(256) void rigid_svd_kernel(vcr_rigid_svd_args p) {
  __shared__ double red[4 * 9];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* S = p.src + (size_t)b * p.K * p.lds;
  const float* C = p.corr + (size_t)b * p.K * p.ldc;
  double mean[6] = {0, 0, 0, 0, 0, 0};
  for (int i = t; i < p.K; i += 256)
    for (int c = 0; c < 3; ++c) { mean[c] += S[(size_t)i * p.lds + c]; mean[3 + c] += C[(size_t)i * p.ldc + c]; }
  block_sums<6>(mean, red);
  // the reference centres in fp32 (src - src.mean): round the means to fp32 like it does
  float smf[3], cmf[3];
  for (int c = 0; c < 3; ++c) { smf[c] = (float)(mean[c] / p.K); cmf[c] = (float)(mean[3 + c] / p.K); }
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = t; i < p.K; i += 256) {
    float sc[3], cc[3];
    for (int c = 0; c < 3; ++c) { sc[c] = S[(size_t)i * p.lds + c] - smf[c]; cc[c] = C[(size_t)i * p.ldc + c] - cmf[c]; }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) H[3 * r + c] += (double)sc[r] * (double)cc[c];
  }
  block_sums<9>(H, red);
  if (t >= 64) return;                                   // wave 0 finishes; its first quad runs the Jacobi sweeps together
  const int li = t & 3;
  double a[3], v[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    a[j] = li == 0 ? H[j] : li == 1 ? H[3 + j] : li == 2 ? H[6 + j] : 0.0;
    v[j] = li == j ? 1.0 : 0.0;
  }
  jacobi_sweeps_quad(a, v);
  double A[3][3], V[3][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {                          // gather the rows (lane i of the quad = row i)
    A[0][j] = quad_get(a[j], 0); A[1][j] = quad_get(a[j], 1); A[2][j] = quad_get(a[j], 2);
    V[0][j] = quad_get(v[j], 0); V[1][j] = quad_get(v[j], 1); V[2][j] = quad_get(v[j], 2);
  }
  if (t == 0) {
    double R[9];
    svd3_finish(A, V, R);
    float* Ro = p.R + (size_t)b * 9;
    float* to = p.t + (size_t)b * 3;
    float Rf[9], tf[3];
    for (int e = 0; e < 9; ++e) { Rf[e] = (float)R[e]; Ro[e] = Rf[e]; }
    for (int r = 0; r < 3; ++r) {
      tf[r] = (float)(-(R[3 * r] * smf[0] + R[3 * r + 1] * smf[1] + R[3 * r + 2] * smf[2]) + cmf[r]);
      to[r] = tf[r];
    }
    if (p.R_ba) for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) p.R_ba[(size_t)b * 9 + 3 * r + c] = Rf[3 * c + r];
    if (p.t_ba) for (int r = 0; r < 3; ++r)
      p.t_ba[(size_t)b * 3 + r] = -(Rf[r] * tf[0] + Rf[3 + r] * tf[1] + Rf[6 + r] * tf[2]);
    if (p.H) for (int e = 0; e < 9; ++e) p.H[(size_t)b * 9 + e] = (float)H[e];
  }
}

}  // namespace

extern "C" int vcr_rigid_svd_f32(const vcr_rigid_svd_args* a, vcr_stream_t stream) {
  if (!a || !a->src || !a->corr || !a->R || !a->t) return VCR_EINVAL;
  if (a->B <= 0 || a->K < 3 || a->lds < 3 || a->ldc < 3) return VCR_EINVAL;
  hipLaunchKernelGGL(rigid_svd_kernel, dim3(a->B), dim3(256), 0, (hipStream_t)stream, *a);
  return VCR_LAUNCH_RC();
}
