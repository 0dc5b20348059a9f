// EdgeConv convDG1 -> max -> convDG2 -> max (model/lpdnet_model.py:122-126, util/util.py:176-199) with the one real
// N*k GEMM of the block -- convDG2 on the post-ReLU per-edge rows -- on the bf16 matrix pipe as exact 3-way splits
// (bf16x3.h): the opt-in companion of edgeconv.hip's packed kernel (k = 20 / 40), same interface (vcr_edgeconv_args),
// same geometry, selected by the driver in linear_mode 1 / 2.
//
//   per group of G = 160 / k points: five tiles of 32 edge rows H = relu(P[nbr] + Q[i]) (fp32, exactly the values of the
//   fp32 kernel), split into three bf16 planes while they are committed to LDS ([3][32][128], pitch 272 B);
//   Y = H W2^T as 8 k-steps x 6 v_mfma_f32_32x32x16_bf16 per tile and wave (48 x 32 cycles against 64 x 64 cycles of
//   v_mfma_f32_32x32x2_f32), wave w owning output channels 32w..32w+31 with its W2 slice split once into 96 VGPRs of
//   packed planes; the max over a point's edges folds into per-point registers through the static row -> point map.
//   x1 = max_j H is taken from the fp32 rows in registers on their way to LDS (same static map per 8-row slab; one
//   cross-lane reduction per point, as soon as its last row has passed), so it is bit-identical to the fp32 kernel's and
//   needs no column pass over LDS.
#include "bf16x3.h"

namespace {

constexpr int HPB = 136;   // H plane row pitch in bf16 (272 B): the 16-lane ds_read_b128 groups are conflict-free

template <int KE>
__global__ __launch_bounds__(256, 2) void edgeconv_dg_packed_bf16x3_kernel(vcr_edgeconv_args p) {
  constexpr int G = 160 / KE;                            // points per group
  __shared__ __attribute__((aligned(16))) short Hs[2][3][32][HPB];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int rs = lane >> 3, cg = lane & 7;              // build mapping: rows rs + 8i, channels 32w + 4cg .. +3
  const int ch = 32 * w + 4 * cg;

  bf16x8 wh[8], wm[8], wl[8];                            // W2[32w + l31][16 s + 8 half + 0..7], three planes
  {
    const float* wr = p.w2 + (size_t)(32 * w + l31) * 128 + 8 * half;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const f32x4 a = ld4(wr + 16 * s), c = ld4(wr + 16 * s + 4);
      const float x[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
      split3x8(x, wh[s], wm[s], wl[s]);
    }
  }
  const float bias2 = p.b2[32 * w + l31];

  const int ngroups = (p.M + G - 1) / G;
  const xcd_slice_t sl = xcd_slice(ngroups);             // this block's groups, all inside its XCD's run of clouds
  if (sl.count == 0) return;
  const int my_groups = sl.count;

  // Gathers run AHEAD of the tile being multiplied, slab by slab (a slab = the 8 rows rs + 8 i of a tile): at the start of tile
  // t the rows of tile t + 1 are in flight (hp / hq) and the indices of tile t + 2 have been requested (nb).  In k-step 2 i + 1 of
  // tile t, slab i of tile t + 1 is split and committed to the other LDS buffer IN THE SHADOW of this wave's own MFMAs (beside the
  // partner wave's MFMAs the same ~40 vector instructions per slab crawl at one per ~20 cycles: profiles/rounds4-5/r4f_mfma_valu_coissue.txt),
  // and its registers are re-used at once for slab i of tile t + 2, whose indices came in a tile ago; the indices of tile t + 3
  // follow.  Every request is unconditional (past the block's last group it re-reads that group: never committed).
  f32x4 hp[4], hq[4];                                    // neighbour P rows / centre Q rows of the tile after the current one
  int nb[4];
  auto idx_slab = [&](int grp, int t, int i) {
    const int e = 32 * t + rs + 8 * i;                   // edge row inside the group, 0..159
    const int pl = e / KE, j = e - pl * KE;
    const int pt = min(grp * G + pl, p.M - 1);
    nb[i] = p.idx[(size_t)pt * KE + j];
  };
  auto row_slab = [&](int grp, int t, int i) {
    const int e = 32 * t + rs + 8 * i;
    const int pt = min(grp * G + e / KE, p.M - 1);
    const int base = (pt / p.n_per_cloud) * p.n_per_cloud;
    hp[i] = ld4(p.pq + (size_t)(base + nb[i]) * p.ldpq + ch);
    hq[i] = ld4(p.pq + (size_t)pt * p.ldpq + 128 + ch);
  };
  float cm[G][4];                                        // x1: running max of this thread's rows, per point (H >= 0)
#pragma unroll
  for (int q = 0; q < G; ++q)
#pragma unroll
    for (int c = 0; c < 4; ++c) cm[q][c] = 0.f;
  // registers -> three bf16 planes in LDS, and the x1 maxima: the 8 rows rs + 8i (rs = 0..7) of slab i of tile t belong
  // to at most two points, split at a compile-time row (t, i are constants after unrolling)
  auto commit_slab = [&](int buf, int t, int i) {
    const f32x4 v = hp[i] + hq[i];
    const f32x4 h = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
    unsigned h0, m0, l0, h1, m1, l1;
    split3x2(h[0], h[1], h0, m0, l0);
    split3x2(h[2], h[3], h1, m1, l1);
    *reinterpret_cast<u32x2*>(&Hs[buf][0][rs + 8 * i][ch]) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(&Hs[buf][1][rs + 8 * i][ch]) = u32x2{m0, m1};
    *reinterpret_cast<u32x2*>(&Hs[buf][2][rs + 8 * i][ch]) = u32x2{l0, l1};
    const int e0 = 32 * t + 8 * i, p0 = e0 / KE, p1 = (e0 + 7) / KE;
    if (p0 == p1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) cm[p0][c] = fmaxf(cm[p0][c], h[c]);
    } else {
      const bool second = rs >= p1 * KE - e0;            // this lane's row belongs to point p1
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        cm[p0][c] = fmaxf(cm[p0][c], second ? 0.f : h[c]);
        cm[p1][c] = fmaxf(cm[p1][c], second ? h[c] : 0.f);
      }
    }
  };
  // a point whose last row lies in tile t: its maxima reduced over the 8 row residues (lanes with the same cg: three
  // exchanges) and written
  auto finish_points = [&](int t, int grp) {
#pragma unroll
    for (int q = 0; q < G; ++q) {
      if (KE * q + KE - 1 < 32 * t || KE * q + KE - 1 > 32 * t + 31) continue;     // (compile-time)
      const int pt = grp * G + q;
      f32x4 m1;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float x = cm[q][c];
        x = fmaxf(x, __shfl_xor(x, 8, 64));
        x = fmaxf(x, __shfl_xor(x, 16, 64));
        x = fmaxf(x, __shfl_xor(x, 32, 64));
        m1[c] = x;
        cm[q][c] = 0.f;
      }
      if (rs == 0 && pt < p.M) st4(p.x1 + (size_t)pt * p.ldx1 + ch, m1);
    }
  };

#pragma unroll
  for (int i = 0; i < 4; ++i) idx_slab(sl.first, 0, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) row_slab(sl.first, 0, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) idx_slab(sl.first, 1, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) commit_slab(0, 0, i);
  finish_points(0, sl.first);
#pragma unroll
  for (int i = 0; i < 4; ++i) row_slab(sl.first, 1, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) idx_slab(sl.first, 2, i);
  __syncthreads();
  int cur = 0;
  for (int gi = 0; gi < my_groups; ++gi) {
    const int grp = sl.first + gi * sl.stride;
    const bool more = gi + 1 < my_groups;
    const int grp_after = more ? grp + sl.stride : grp;  // (no group after the last: its rows again, never committed)
    float pm[G];                                         // x2: per-point maxima of the MFMA rows
#pragma unroll
    for (int q = 0; q < G; ++q) pm[q] = VCR_NEG_INF;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const bool has_next = t < 4 || more;
      const int tn = (t + 1) % 5, t2 = (t + 2) % 5, t3 = (t + 3) % 5;
      const int grp2 = t + 2 < 5 ? grp : grp_after, grp3 = t + 3 < 5 ? grp : grp_after;
      f32x16 acc = {0};
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        bf16x8 fa[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fa[pl] = *reinterpret_cast<const bf16x8*>(&Hs[cur][pl][l31][16 * s + 8 * half]);
        acc = mfma6(fa, wh[s], wm[s], wl[s], acc);
        if (s & 1) {
          const int i = s >> 1;
          if (has_next) commit_slab(cur ^ 1, tn, i);
          row_slab(grp2, t2, i);
          idx_slab(grp3, t3, i);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {                     // fold the tile into the per-point maxima (static map)
        const int row0 = 32 * t + acc_row(r, 0), row1 = row0 + 4;
        const int p0 = row0 / KE, p1 = row1 / KE;
        if (p0 == p1) {
          pm[p0] = fmaxf(pm[p0], acc[r]);
        } else {
          pm[p0] = fmaxf(pm[p0], half ? VCR_NEG_INF : acc[r]);
          pm[p1] = fmaxf(pm[p1], half ? acc[r] : VCR_NEG_INF);
        }
      }
      if (t == 4) {
        // the group's five tiles are done: x2 from the MFMA maxima
#pragma unroll
        for (int q = 0; q < G; ++q) {
          const int pt = grp * G + q;
          const float v = fmaxf(pm[q], xhalf(pm[q]));
          if (half == 0 && pt < p.M) p.x2[(size_t)pt * p.ldx2 + 32 * w + l31] = fmaxf(v + bias2, 0.f);
        }
      }
      if (has_next) finish_points(tn, t < 4 ? grp : grp + sl.stride);
      lds_barrier();                                     // (LDS only: the rows / indices requested ahead stay in flight across it)
      cur ^= 1;
    }
  }
}

}  // namespace

// Same contract as vcr_edgeconv_f32 for k = 20 / 40 (the path's values); other k: VCR_EUNSUPPORTED (the caller keeps
// the fp32 kernel).  x1 is bit-identical to vcr_edgeconv_f32's, x2 agrees to fp32-GEMM rounding.
extern "C" int vcr_edgeconv_bf16x3_f32(const vcr_edgeconv_args* a, vcr_stream_t stream) {
  if (!a || !a->pq || !a->idx || !a->w2 || !a->b2 || !a->x1 || !a->x2) return VCR_EINVAL;
  if (a->M <= 0 || a->k <= 0 || a->k > 64 || a->n_per_cloud <= 0 || (a->M % a->n_per_cloud)) return VCR_EINVAL;
  if (a->ldpq < 256 || (a->ldpq & 3) || (a->ldx1 & 3) || a->ldx1 < 128 || a->ldx2 < 128) return VCR_EINVAL;
  if (((uintptr_t)a->w2 | (uintptr_t)a->pq | (uintptr_t)a->x1) & 15) return VCR_EINVAL;
  if (a->k != 20 && a->k != 40) return VCR_EUNSUPPORTED;
  const int G = 160 / a->k, ngroups = (a->M + G - 1) / G;
  const int grid = ngroups < 1024 ? ngroups : 1024;
  if (a->k == 20) hipLaunchKernelGGL(edgeconv_dg_packed_bf16x3_kernel<20>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else hipLaunchKernelGGL(edgeconv_dg_packed_bf16x3_kernel<40>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  return VCR_LAUNCH_RC();
}
