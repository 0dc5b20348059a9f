// RECORD OF AN EXPERIMENT (round 4) -- not part of the library build.
//
// A persistent, hand-scheduled fp32 linear: correct (GEMM part bit-identical to linear.hip's 32x32x2 kernels on nine shapes
// incl. ragged M / N, LayerNorm-in, statistics-out, residual: profiles/experiments/check_linear_stream.py ran against it while
// it was wired into vcr_linear_f32 behind variant bits 15 / 16), and SLOWER than the one-tile-per-workgroup kernels wherever
// those run four workgroups per CU:  stacked QKV projection (M 32768, N 3072, K 512) 866 us = 119 TFLOP/s against 785 us = 131;
// level on the residual shapes (wo 158 vs 161 us, ffn2 282 vs 282).  Timing ablations of the same launch (profiles/rounds4-5/r4k_linear_stream.txt):
// everything 913 us; without the LDS-DMA slab requests 787; without the k-step barriers 885; without the drain 838; with
// none of them 743 (138.7 TFLOP/s = 0.88: the MFMA + fragment-read loop of two waves per SIMD by itself).  What it shows: at
// two waves per SIMD (two 64-register accumulator sets need 256 VGPRs) the slab requests (~100 cycles of the issuing wave
// each) and the vmcnt-ordered operand loads of the drain cannot be hidden as well as two MORE waves per SIMD hide them --
// for fp32 MFMAs (64 cycles each, 64 accumulator registers per wave tile) occupancy beats hand scheduling.  The same
// scheduling idea DID pay in EdgeConv (csrc/edgeconv.hip: no LDS-DMA, one accumulator set).
//
// Pointwise linear, persistent and hand-scheduled:  Y = act(X W^T + bias) (+ residual), LayerNorm-in / statistics-out as
// in linear.hip (model/transformer.py:141-144,210-212,224,237-238; model/lpdnet_model.py:133-135).
//
// Why a second kernel.  linear.hip runs one 128 x 128 tile per workgroup, 2-4 workgroups per CU: a workgroup's prologue
// and epilogue (LDS transpose, bias / LayerNorm / residual, stores: ~400 vector instructions per wave) run beside OTHER
// workgroups' k loops -- and beside a wave that issues fp32 MFMAs back to back, a SIMD lets another wave's vector / LDS
// instructions through at one per ~20 cycles (profiles/rounds4-5/r4f_mfma_valu_coissue.txt), while a wave's OWN instructions issue
// freely in the 64-cycle shadow of its own MFMA.  Measured with in-kernel stamps (profiles/rounds4-5/r4f_timeline_linear.txt): the
// stacked QKV projection spends 9.8 + 23 us of a workgroup's 134 us in prologue + epilogue, the k loop's own clock is
// 2.25-2.39 GHz (profiles/rounds4-5/r4c_clock_probe_linear.txt), and a bare MFMA loop sustains 153 TFLOP/s on the same boxes: the
// 0.80 of peak is issue structure, not power.  So here
//   * workgroups are PERSISTENT (two per CU, 256 threads, BK 32): tile after tile, the LDS-DMA slab pipeline runs across
//     tile boundaries (the next tile's first slab is requested during the last k-step of the current one);
//   * the MFMA operands are SWAPPED -- A = W fragment, B = X fragment, D[n][m] -- so that a lane owns ONE output row
//     (m = lane & 31) and its accumulator registers are runs of four consecutive COLUMNS: bias / LayerNorm / ReLU /
//     residual / 16-B stores straight from registers (no LDS transpose, no barrier), the LayerNorm (mean, inv) of a lane's row
//     in two registers, the statistics-out sums lane-local;
//   * the epilogue of tile t is drained out of a second accumulator set DURING the k loop of tile t+1, one 16-B piece at
//     a time, inside chunks fenced by sched_barrier(0): per k-group the next group's fragment reads, 16 MFMAs and a slice
//     of everything else.
// Products are commutative and the k order is that of linear.hip's 32x32x2 kernels: the GEMM part is bit-identical to
// them; the statistics-out partials are summed in a different (shifted one-pass) order, fp32 rounding apart.
#include "common.h"
#include <type_traits>

namespace {

constexpr int SM = 128, SN = 128, SK = 32;
constexpr int STREAM_ABLATE = 0;   // TIMING experiments only (probe_build.py --set): 1 no slab fills after the first, 2 no
                                   // k-step barriers, 4 no drain (outputs never written): results are wrong with any of them
struct STile { float a[SM][SK]; float b[SN][SK]; };      // 32 KB per stage, two stages

__device__ __forceinline__ void glds16s(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

struct SProb { vcr_linear_args p; int tiles_m, tiles_n; };

// tile number (in a problem's own order) -> tile origin: M-major inside column groups of <= 12 tiles (linear.hip)
__device__ __forceinline__ void tile_origin(int tiles_m, int tiles_n, int id, int& m0, int& n0) {
  constexpr int GW = 12;
  const int grp = id / (GW * tiles_m), wg = min(GW, tiles_n - grp * GW), loc = id - grp * GW * tiles_m;
  m0 = (loc / wg) * SM;
  n0 = (grp * GW + loc % wg) * SN;
}

// The fields of a problem that the kernel uses, selected field by field between the two kernel arguments (uniform selects
// into SGPRs; a POINTER to one of two by-value kernel arguments makes the compiler copy both to scratch).
struct TileP {
  const float *x, *w, *bias, *residual, *ln_stats_in, *ln_colsum;
  float *y, *stats_out;
  int ldx, ldr, ldy, M, N, K, relu, ln_nseg;
  float ln_eps;
};
__device__ __forceinline__ TileP pick_problem(const SProb& a, const SProb& b, bool second) {
#define VCR_SEL(f) (second ? b.p.f : a.p.f)
  return TileP{VCR_SEL(x), VCR_SEL(w), VCR_SEL(bias), VCR_SEL(residual), VCR_SEL(ln_stats_in), VCR_SEL(ln_colsum),
               VCR_SEL(y), VCR_SEL(stats_out), VCR_SEL(ldx), VCR_SEL(ldr), VCR_SEL(ldy), VCR_SEL(M), VCR_SEL(N), VCR_SEL(K),
               VCR_SEL(relu), VCR_SEL(ln_nseg), VCR_SEL(ln_eps)};
#undef VCR_SEL
}

// state of one tile's epilogue (the tile whose accumulators are being drained)
struct Drain {
  TileP p;
  int m0, n0;
};

template <bool PAIR>
__global__ __launch_bounds__(256, 2) void linear_stream_kernel(SProb q0, SProb q1, int ntiles0, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  STile* tile = reinterpret_cast<STile*>(smem);         // [2]
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int G = (int)gridDim.x;
  // this workgroup's tiles: round r covers ids [r G, (r + 1) G); inside a round the XCDs take contiguous eighths
  const int bperm = (G % 8 == 0) ? ((int)blockIdx.x % 8) * (G / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
  const int my_n = bperm < ntiles ? (ntiles - bperm + G - 1) / G : 0;
  if (my_n == 0) return;

  // fill mapping (linear.hip, BK 32): one wave instruction lands 8 rows x 128 B; wave w covers rows 32 w + 8 i + lane / 8,
  // physical 16-B chunk lane % 8, logical chunk ^ ((row >> 1) & 7)
  const int frow = lane >> 3, fpc = lane & 7;
  const float* xa[4];
  const float* wb[4];
  // a tile is (which problem, m0, n0): the problem's fields are re-selected from the kernel arguments where they are used
  // (three resident copies of them overflowed the scalar registers)
  auto set_tile = [&](int id, bool& second, int& m0, int& n0) {
    second = PAIR && id >= ntiles0;
    const int tiles_m = second ? q1.tiles_m : q0.tiles_m, tiles_n = second ? q1.tiles_n : q0.tiles_n;
    tile_origin(tiles_m, tiles_n, second ? id - ntiles0 : id, m0, n0);
  };
  auto set_fill = [&](bool second, int m0, int n0) {
    const float* x = second ? q1.p.x : q0.p.x;
    const float* w = second ? q1.p.w : q0.p.w;
    const int ldx = second ? q1.p.ldx : q0.p.ldx, M = second ? q1.p.M : q0.p.M, N = second ? q1.p.N : q0.p.N,
              K = second ? q1.p.K : q0.p.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 32 + 8 * i + frow;
      const int lc = fpc ^ ((row >> 1) & 7);
      xa[i] = x + (size_t)min(m0 + row, M - 1) * ldx + 4 * lc;
      wb[i] = w + (size_t)min(n0 + row, N - 1) * K + 4 * lc;
    }
  };
  auto fill = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16s(xa[i] + k0, &tile[buf].a[wave * 32 + 8 * i][0]);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16s(wb[i] + k0, &tile[buf].b[wave * 32 + 8 * i][0]);
  };
  auto fill_part = [&](int buf, int k0, int g) {         // a quarter of a slab's requests (one per chunk: each LDS-DMA issue
    glds16s(xa[g] + k0, &tile[buf].a[wave * 32 + 8 * g][0]);   // takes ~100 cycles of the wave, so it goes into an MFMA shadow)
    glds16s(wb[g] + k0, &tile[buf].b[wave * 32 + 8 * g][0]);
  };

  // fragment addressing (linear.hip, MS 32): row r of the wave's 64, logical chunk 2 g + half, swizzled by (row >> 1) & 7
  int ra_[2], rb_[2], sa[2], sb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ra_[i] = wm * 64 + i * 32 + l31; sa[i] = (ra_[i] >> 1) & 7;
    rb_[i] = wn * 64 + i * 32 + l31; sb[i] = (rb_[i] >> 1) & 7;
  }

  f32x16 acc[2][2], accp[2][2];                          // acc[i][j]: D[n = 32 j + acc_row(r, half)][m = 32 i + l31]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) { acc[i][j] = f32x16{0}; accp[i][j] = f32x16{0}; }

  // ---- the drain of one finished tile, in pieces with compile-time register indices --------------------------------
  Drain dr{};
  float ln_mean[2], ln_inv[2];                           // LayerNorm-in: this lane's rows (i = 0, 1)
  float st_v0[2], st_s1[2], st_s2[2];                    // statistics-out: shifted one-pass sums of this lane's rows
  f32x4 d_bias[2], d_csum[2], d_res[2];                  // operands of the two pieces of the NEXT k-step
  auto dr_row = [&](int i) { return dr.m0 + wm * 64 + 32 * i + l31; };
  auto dr_col = [&](int j, int qd) { return dr.n0 + wn * 64 + 32 * j + 8 * qd + 4 * half; };
  auto drain_moments = [&](int i) {                      // (LN_IN) mean / inv of row i from the producer's partial sums
    const TileP& p = dr.p;
    if (p.ln_stats_in) {
      float mean, var;
      ln_row_moments(p.ln_stats_in + (size_t)min(dr_row(i), p.M - 1) * p.ln_nseg * 2, p.ln_nseg, p.K, mean, var);
      ln_mean[i] = mean;
      ln_inv[i] = 1.f / (sqrtf(var) + p.ln_eps);
    }
  };
  auto drain_load = [&](int slot, int i, int j, int qd) {   // request a piece's column / residual operands
    const TileP& p = dr.p;
    const int col = dr_col(j, qd), row = dr_row(i);
    const bool ok = col < p.N;
    d_bias[slot] = (p.bias && ok) ? ld4(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    d_csum[slot] = (p.ln_stats_in && ok) ? ld4(p.ln_colsum + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    d_res[slot] = (p.residual && ok && row < p.M) ? ld4(p.residual + (size_t)row * p.ldr + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto drain_piece = [&](int slot, int i, int j, int qd) {   // accp[i][j][4 qd .. 4 qd + 3]: four consecutive columns of one row
    const TileP& p = dr.p;
    const int col = dr_col(j, qd), row = dr_row(i);
    f32x4 v = {accp[i][j][4 * qd], accp[i][j][4 * qd + 1], accp[i][j][4 * qd + 2], accp[i][j][4 * qd + 3]};
    if (p.ln_stats_in) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf(ln_inv[i], fmaf(-ln_mean[i], d_csum[slot][e], v[e]), d_bias[slot][e]);
    } else {
      v = v + d_bias[slot];
    }
    if (p.relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
    if (p.residual) v = v + d_res[slot];
    if (row < p.M && col < p.N) st4(p.y + (size_t)row * p.ldy + col, v);
    if (p.stats_out) {
      if (j == 0 && qd == 0) { st_v0[i] = v[0]; st_s1[i] = 0.f; st_s2[i] = 0.f; }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[e] - st_v0[i];
        st_s1[i] += d;
        st_s2[i] = fmaf(d, d, st_s2[i]);
      }
    }
  };
  auto drain_stats = [&](int i) {                        // (STATS_OUT) this wave's 64-column segment of row i
    const TileP& p = dr.p;
    if (p.stats_out) {
      // each half holds 32 of the 64 columns: (mean, M2) per half, combined as in Chan et al.
      const float mh = st_v0[i] + st_s1[i] * (1.f / 32.f);
      const float m2h = st_s2[i] - st_s1[i] * st_s1[i] * (1.f / 32.f);
      const float mo = xhalf(mh), m2o = xhalf(m2h);
      const float dlt = mh - mo;
      const int row = dr_row(i);
      if (half == 0 && row < p.M && dr.n0 + wn * 64 < p.N) {
        float* so = p.stats_out + ((size_t)row * (p.N / 64) + (dr.n0 + wn * 64) / 64) * 2;
        so[0] = 32.f * (mh + mo);
        so[1] = m2h + m2o + dlt * dlt * 16.f;
      }
    }
  };
  // Drain schedule inside the eight k-steps of the next tile's first block.  vmcnt retires in order, so a piece's operand
  // loads must be OLDER than the slab requests that are still in flight when the piece needs them: the two pieces of
  // k-step k8 are requested at the top of k-step k8 - 1 (before its slab requests) and consumed in chunks 1 and 3 of k-step
  // k8 (k8 = 0: requested before the block).  The row moments ride in front of the block, the statistics behind it.
  auto piece_ijq = [](int n, int& i, int& j, int& qd) { i = n >> 3; j = (n >> 2) & 1; qd = n & 3; };
  auto drain_top = [&](int k8) {                         // top of k-step k8: request the operands of k-step k8 + 1's pieces
    if (k8 + 1 < 8) {
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) { int i, j, qd; piece_ijq(2 * (k8 + 1) + sl, i, j, qd); drain_load(sl, i, j, qd); }
    }
  };
  auto drain_chunk = [&](int k8, int g) {
    if (g == 1 || g == 3) { int i, j, qd; piece_ijq(2 * k8 + (g >> 1), i, j, qd); drain_piece(g >> 1, i, j, qd); }
  };
  auto drain_begin = [&]() {                             // before the block: row moments, operands of k-step 0's pieces
    drain_moments(0);
    drain_moments(1);
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) { int i, j, qd; piece_ijq(sl, i, j, qd); drain_load(sl, i, j, qd); }
  };
  auto drain_end = [&]() { drain_stats(0); drain_stats(1); };

  // ---- main loop ---------------------------------------------------------------------------------------------------
  bool pc;                                               // current tile: which problem, origin
  int m0c, n0c;
  set_tile(bperm, pc, m0c, n0c);
  set_fill(pc, m0c, n0c);
  fill(0, 0);
  __syncthreads();
  int s = 0;                                             // global k-step counter: stage = s & 1
  bool have_prev = false;
  for (int ti = 0; ti < my_n; ++ti) {
    const int nk = (pc ? q1.p.K : q0.p.K) / SK;
    // the tile after this one (its first slab is requested during this tile's last k-step)
    const bool more = ti + 1 < my_n;
    bool pn = pc;
    int m0n = m0c, n0n = n0c;
    if (more) set_tile(bperm + (ti + 1) * G, pn, m0n, n0n);
    // eight unrolled k-steps; DRAIN: with the previous tile's epilogue pieces in the chunks (a second copy of the code, so
    // that the plain block carries no branches and no drain state)
    auto kblock = [&](auto drain_tag, int kb) {
      constexpr bool DRAIN = decltype(drain_tag)::value;
#pragma unroll
      for (int k8 = 0; k8 < 8; ++k8) {
        const int ks = kb + k8, cur = s & 1;
        const bool next_same = ks + 1 < nk;              // the next slab: this tile's, or the first one of the next tile
        if (!next_same && more) set_fill(pn, m0n, n0n);
        const int k0n = next_same ? (ks + 1) * SK : 0;
        const bool do_fill = !(STREAM_ABLATE & 1) && (next_same || more);
        if constexpr (DRAIN) drain_top(k8);
        const STile& T = tile[cur];
        f32x4 fa[2][2], fb[2][2];                        // [set][i / j]: the next k-group's fragments are requested
#pragma unroll                                           // before this group's MFMAs
        for (int i = 0; i < 2; ++i) {
          fa[0][i] = ld4(&T.a[ra_[i]][4 * ((0 + half) ^ sa[i])]);
          fb[0][i] = ld4(&T.b[rb_[i]][4 * ((0 + half) ^ sb[i])]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (g + 1 < 4) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              fa[(g + 1) & 1][i] = ld4(&T.a[ra_[i]][4 * ((2 * (g + 1) + half) ^ sa[i])]);
              fb[(g + 1) & 1][i] = ld4(&T.b[rb_[i]][4 * ((2 * (g + 1) + half) ^ sb[i])]);
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fb[g & 1][j][e], fa[g & 1][i][e], acc[i][j]);
          if (do_fill) fill_part(cur ^ 1, k0n, g);
          if constexpr (DRAIN) drain_chunk(k8, g);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (!(STREAM_ABLATE & 2)) __syncthreads();       // drains the LDS-DMA (vmcnt(0)) and orders the two stages
        ++s;
      }
    };
    if (have_prev && !(STREAM_ABLATE & 4)) {
      drain_begin();
      kblock(std::true_type{}, 0);
      drain_end();
    } else {
      kblock(std::false_type{}, 0);
    }
    for (int kb = 8; kb < nk; kb += 8) kblock(std::false_type{}, kb);
    // this tile's accumulators become the set that the next tile's k loop drains
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) { accp[i][j] = acc[i][j]; acc[i][j] = f32x16{0}; }
    dr = Drain{pick_problem(q0, q1, pc), m0c, n0c};
    have_prev = true;
    pc = pn; m0c = m0n; n0c = n0n;
  }
  // the last tile: nothing left to hide its epilogue under
  drain_begin();
#pragma unroll
  for (int k8 = 0; k8 < 8; ++k8) {
    drain_top(k8);
    drain_chunk(k8, 1);
    drain_chunk(k8, 3);
  }
  drain_end();
}

}  // namespace

// Launch the persistent kernel for one linear (b == nullptr) or two independent ones.  The caller (linear.hip) has validated
// the arguments and decided eligibility (linear_stream_eligible below).
int vcr_linear_stream_launch(const vcr_linear_args* a, const vcr_linear_args* b, hipStream_t stream) {
  SProb q0{*a, (a->M + SM - 1) / SM, (a->N + SN - 1) / SN};
  SProb q1 = q0;
  int n0 = q0.tiles_m * q0.tiles_n, n = n0;
  if (b) {
    q1 = SProb{*b, (b->M + SM - 1) / SM, (b->N + SN - 1) / SN};
    n += q1.tiles_m * q1.tiles_n;
  }
  const int slots = 2 * vcr_cu_count();
  const int grid = n < slots ? n : slots;
  const int lds = 2 * (int)sizeof(STile);
  if (b) {
    VCR_DYN_LDS(linear_stream_kernel<true>, lds);
    hipLaunchKernelGGL(linear_stream_kernel<true>, dim3(grid), dim3(256), lds, stream, q0, q1, n0, n);
  } else {
    VCR_DYN_LDS(linear_stream_kernel<false>, lds);
    hipLaunchKernelGGL(linear_stream_kernel<false>, dim3(grid), dim3(256), lds, stream, q0, q1, n0, n);
  }
  return VCR_LAUNCH_RC();
}

// Shapes the persistent kernel takes: 16-B aligned operands (the LDS-DMA / 16-B store requirements of linear.hip's fast
// kernels), K a multiple of 256 (eight unrolled k-steps per block), no fused max.
bool vcr_linear_stream_eligible(const vcr_linear_args* a) {
  if (a->segmax_out || !a->y || (a->K % 256) != 0 || (a->N % 4) != 0) return false;
  if ((a->ldx & 3) || (a->ldy & 3) || (((uintptr_t)a->x | (uintptr_t)a->w | (uintptr_t)a->y) & 15)) return false;
  if (a->bias && ((uintptr_t)a->bias & 15)) return false;
  if (a->residual && ((a->ldr & 3) || ((uintptr_t)a->residual & 15))) return false;
  if (a->ln_stats_in && (((uintptr_t)a->ln_colsum & 15) || a->ln_nseg <= 0 || (a->K % a->ln_nseg))) return false;
  if (a->stats_out && (a->N % 64)) return false;
  return true;
}
