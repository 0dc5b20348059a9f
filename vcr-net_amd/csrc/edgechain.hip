// DGCNN's EdgeConv chain in ONE kernel (model/vcrnet_model.py:104-118): per point i and neighbour j
//     h1 = relu(P[nbr_ij] + Q[i])            (conv1 through the neighbour / centre split, 64 channels)
//     h2 = relu(W2 h1 + b2)  (64)      h3 = relu(W3 h2 + b3)  (128)      h4 = relu(W4 h3 + b4)  (256)
//     x_l[i] = max_j h_l[(i,j)],   out[i] = (x1 | x2 | x3 | x4)          (512 columns)
// The per-edge activations (1.3 GB per step at BASELINE configs[1] when they were written by separate GEMMs) never
// leave the CU: they live in LDS one 32-row MFMA tile at a time.
//
// Pipeline.  A workgroup (8 waves, one per CU) walks a contiguous run of edge rows in 32-row tiles.  In step S
//     waves 6-7 gather tile S into H1,          waves 4-5 run conv2 on tile S-1 (H1 -> H2, one 32-channel tile each),
//     waves 0-3 run conv3 on tile S-2 (H2 -> H3, one 32-channel tile each),
//     and ALL eight run conv4 on tile S-3 (H3 -> registers, wave w = channels 32w..32w+31),
// one barrier per step, every buffer double buffered (69 KB of LDS).  Per step the four SIMDs carry 192 / 192 / 160 /
// 160 MFMAs of the tile's 704, so the chain runs at 0.92 of what its 59 GFLOP per step cost on the fp32 matrix pipe.
// Weights stay in registers for the kernel's lifetime (W4 slice 64 VGPRs + one 32-VGPR slice of W2 or W3).
//
// Maxima without atomics.  k = 20 or 40: 160 edge rows = 5 tiles = 8 or 4 whole points, so the row -> point map of a
// tile is a compile-time function of (tile mod 5, accumulator register, lane half).  A wave sees every row of its
// channels, tile after tile, and folds them into per-point registers with static indexing; the point's value is
// stored when its group's fifth tile has passed.  MFMA accumulators have the channel on the lane, so the stores are
// 128-B rows.
#include "common.h"

namespace {

constexpr int P1 = 68;    // H1 / H2 row pitch (floats): 16-lane ds_read_b128 groups conflict-free
constexpr int P3 = 132;   // H3 row pitch

struct Tiles {
  float h1[2][32][P1];
  float h2[2][32][P1];
  float h3[2][32][P3];
};

template <int N, class F, int... I>
__device__ __forceinline__ void unroll_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void unroll(F&& f) { unroll_impl<N>(f, std::make_integer_sequence<int, N>{}); }

// fold one accumulator tile (rows = edges, lane = channel) into the per-point maxima of its group; TT = tile mod 5
template <int KE, int TT, int G>
__device__ __forceinline__ void fold_tile(const f32x16& v, int half, float lowest, float (&pm)[G]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int e0 = 32 * TT + (r & 3) + 8 * (r >> 2), e1 = e0 + 4;   // this register's edge row in lane half 0 / 1
    const int p0 = e0 / KE, p1 = e1 / KE;
    if (p0 == p1) {
      pm[p0] = fmaxf(pm[p0], v[r]);
    } else {
      pm[p0] = fmaxf(pm[p0], half ? lowest : v[r]);
      pm[p1] = fmaxf(pm[p1], half ? v[r] : lowest);
    }
  }
}

template <int KE>
__global__ __launch_bounds__(512, 1) void edgechain_kernel(vcr_edgechain_args p, int groups_per_block) {
  constexpr int G = 160 / KE;                              // points per 5-tile group
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Tiles& T = *reinterpret_cast<Tiles*>(smem);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int ngroups = (p.M + G - 1) / G;
  const int g0 = xcd_chunk((int)blockIdx.x, (int)gridDim.x) * groups_per_block;   // contiguous run: a cloud stays on one XCD
  const int my_groups = min(groups_per_block, ngroups - g0);
  if (my_groups <= 0) return;
  const int ntiles = my_groups * 5;

  // ---- weights: conv4 slice for everyone, one conv3 (waves 0-3) or conv2 (waves 4-5) slice
  f32x4 w4[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) w4[g] = ld4(p.w4 + (size_t)(32 * w + l31) * 128 + 8 * g + 4 * half);
  const float bias4 = p.b4[32 * w + l31];
  f32x4 ws[8];
  float bias_s = 0.f;
  if (w < 6) {
    const float* wsrc = w < 4 ? p.w3 + (size_t)(32 * w + l31) * 64 : p.w2 + (size_t)(32 * (w - 4) + l31) * 64;
#pragma unroll
    for (int g = 0; g < 8; ++g) ws[g] = ld4(wsrc + 8 * g + 4 * half);
    bias_s = w < 4 ? p.b3[32 * w + l31] : p.b2[32 * (w - 4) + l31];
  } else {
#pragma unroll
    for (int g = 0; g < 8; ++g) ws[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- gather role (waves 6-7): lane = (row sub-group rs, 4-channel group cg); rows 16 (w-6) + rs + 4 i of the tile
  const int rs = lane >> 4, cg = lane & 15;
  int nbr[4], gpt[4];                                      // neighbour row / own point of the tile fetched NEXT
  auto fetch_idx = [&](int tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = (g0 * G) * KE + 32 * tile + 16 * (w - 6) + rs + 4 * i;     // global edge row
      const int pt = min(e / KE, p.M - 1), j = e % KE;
      gpt[i] = pt;
      nbr[i] = (pt / p.n_per_cloud) * p.n_per_cloud + p.idx[(size_t)pt * KE + j];
    }
  };
  f32x4 hp[4], hq[4];
  auto fetch_rows = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      hp[i] = ld4(p.pq + (size_t)nbr[i] * p.ldpq + 4 * cg);
      hq[i] = ld4(p.pq + (size_t)gpt[i] * p.ldpq + 64 + 4 * cg);
    }
  };
  auto write_h1 = [&](int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 v = hp[i] + hq[i];
      st4(&T.h1[slot][16 * (w - 6) + rs + 4 * i][4 * cg],
          f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)});
    }
  };

  float pm4[G], pms[G];                                    // running per-point maxima: conv4 / this wave's second role
#pragma unroll
  for (int q = 0; q < G; ++q) { pm4[q] = VCR_NEG_INF; pms[q] = 0.f; }

  auto emit = [&](int grp, const float (&pm)[G], int col, bool relu_bias, float bias) {
#pragma unroll
    for (int q = 0; q < G; ++q) {
      const int pt = (g0 + grp) * G + q;
      float m = fmaxf(pm[q], xhalf(pm[q]));
      if (relu_bias) m = fmaxf(m + bias, 0.f);
      if (half == 0 && pt < p.M) p.out[(size_t)pt * p.ldo + col + l31] = m;
    }
  };

  if (w >= 6) fetch_idx(0);
  // one step: S = base + U with base % 5 == 0, so every role's tile index mod 5 is the compile-time (U - depth) mod 5
  auto step = [&](int base, auto Uc) {
    constexpr int U = decltype(Uc)::value;
    const int S = base + U;
    // -- second role, first half: issue the memory work
    if (w >= 6) {
      if (S < ntiles) fetch_rows();
      if (S + 1 < ntiles) fetch_idx(S + 1);
    } else if (w >= 4) {
      constexpr int TT = (U + 4) % 5;                      // conv2 on tile S-1
      const int tau = S - 1;
      if (tau >= 0 && tau < ntiles) {
        const int slot = tau & 1, c = w - 4;
        f32x16 acc = {0};
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const f32x4 af = ld4(&T.h1[slot][l31][8 * g + 4 * half]);
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = mfma32(af[s], ws[g][s], acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[r] = fmaxf(acc[r] + bias_s, 0.f);
          T.h2[slot][acc_row(r, half)][32 * c + l31] = acc[r];
        }
        if (TT == 0) {
#pragma unroll
          for (int q = 0; q < G; ++q) pms[q] = 0.f;
        }
        fold_tile<KE, TT, G>(acc, half, 0.f, pms);
        if (TT == 4) emit(tau / 5, pms, 64 + 32 * c, false, 0.f);
      }
    } else {
      constexpr int TT = (U + 3) % 5;                      // conv3 on tile S-2
      const int tau = S - 2;
      if (tau >= 0 && tau < ntiles) {
        const int slot = tau & 1;
        f32x16 acc = {0};
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const f32x4 af = ld4(&T.h2[slot][l31][8 * g + 4 * half]);
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = mfma32(af[s], ws[g][s], acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[r] = fmaxf(acc[r] + bias_s, 0.f);
          T.h3[slot][acc_row(r, half)][32 * w + l31] = acc[r];
        }
        if (TT == 0) {
#pragma unroll
          for (int q = 0; q < G; ++q) pms[q] = 0.f;
        }
        fold_tile<KE, TT, G>(acc, half, 0.f, pms);
        if (TT == 4) emit(tau / 5, pms, 128 + 32 * w, false, 0.f);
      }
    }
    // -- x1 = max over a point's h1 rows: a column pass by wave 6 (lane = channel) over the tile conv2 is reading
    if (w == 6) {
      constexpr int TT = (U + 4) % 5;
      const int tau = S - 1;
      if (tau >= 0 && tau < ntiles) {
        const int slot = tau & 1;
        if (TT == 0) {
#pragma unroll
          for (int q = 0; q < G; ++q) pms[q] = 0.f;
        }
#pragma unroll
        for (int rr = 0; rr < 32; ++rr) pms[(32 * TT + rr) / KE] = fmaxf(pms[(32 * TT + rr) / KE], T.h1[slot][rr][lane]);
        if (TT == 4) {
#pragma unroll
          for (int q = 0; q < G; ++q) {
            const int pt = (g0 + tau / 5) * G + q;
            if (pt < p.M) p.out[(size_t)pt * p.ldo + lane] = pms[q];
          }
        }
      }
    }
    // -- conv4 on tile S-3, every wave
    {
      constexpr int TT = (U + 2) % 5;
      const int tau = S - 3;
      if (tau >= 0 && tau < ntiles) {
        const int slot = tau & 1;
        f32x16 acc = {0};
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const f32x4 af = ld4(&T.h3[slot][l31][8 * g + 4 * half]);
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = mfma32(af[s], w4[g][s], acc);
        }
        if (TT == 0) {
#pragma unroll
          for (int q = 0; q < G; ++q) pm4[q] = VCR_NEG_INF;
        }
        fold_tile<KE, TT, G>(acc, half, VCR_NEG_INF, pm4);
        if (TT == 4) emit(tau / 5, pm4, 256 + 32 * w, true, bias4);
      }
    }
    // -- second role, second half: the gathered rows have had the conv4 MFMAs to arrive
    if (w >= 6 && S < ntiles) write_h1(S & 1);
    __syncthreads();
  };
  for (int base = 0; base < ntiles + 3; base += 5) unroll<5>([&](auto Uc) { step(base, Uc); });
}

}  // namespace

extern "C" int vcr_edgechain_f32(const vcr_edgechain_args* a, vcr_stream_t stream) {
  vcr_stream_scope bound(stream);
  if (!a || !a->pq || !a->idx || !a->w2 || !a->b2 || !a->w3 || !a->b3 || !a->w4 || !a->b4 || !a->out) return VCR_EINVAL;
  if (a->M <= 0 || a->n_per_cloud <= 0 || a->ldpq < 128 || (a->ldpq & 3) || a->ldo < 512) return VCR_EINVAL;
  if (((uintptr_t)a->pq | (uintptr_t)a->w2 | (uintptr_t)a->w3 | (uintptr_t)a->w4) & 15) return VCR_EINVAL;
  if (a->k != 20 && a->k != 40) return VCR_EUNSUPPORTED;   // the static row -> point maps (see the kernel) exist for these
  const int G = 160 / a->k, ngroups = (a->M + G - 1) / G;
  const int cus = vcr_cu_count();
  const int gpb = (ngroups + cus - 1) / cus;               // groups per workgroup: one workgroup per CU, contiguous runs
  const int grid = (ngroups + gpb - 1) / gpb;
  const int lds = (int)sizeof(Tiles);
  if (a->k == 20) {
    VCR_DYN_LDS(edgechain_kernel<20>, lds);
    hipLaunchKernelGGL(edgechain_kernel<20>, dim3(grid), dim3(512), lds, (hipStream_t)stream, *a, gpb);
  } else {
    VCR_DYN_LDS(edgechain_kernel<40>, lds);
    hipLaunchKernelGGL(edgechain_kernel<40>, dim3(grid), dim3(512), lds, (hipStream_t)stream, *a, gpb);
  }
  return VCR_LAUNCH_RC();
}
