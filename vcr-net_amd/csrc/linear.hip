// Pointwise linear / 1x1 convolution on fp32 MFMA:  Y = act(X W^T + bias) (+ residual).
// Replaces nn.Conv1d/Conv2d(k=1)/nn.Linear on the path (model/lpdnet_model.py:123-135 after the F7
// split, model/transformer.py:210-212,224,237-238).
//
// 128x128x32 block tile, 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x2_f32 tiles.
// Both operands are K-contiguous ([M,K] and [N,K]), so the LDS image of each is [rows][32+4 pad]
// and a lane fetches its fragment as ONE ds_read_b128 = 4 consecutive k, fed to 4 successive
// MFMAs; the k order inside the instruction's (lane-half, step) grid is permuted the same way on A
// and B (k = 8g + 4*half + s), which a sum over k does not care about.  Pitch 36 floats (144 B)
// makes the 16-lane b128 groups conflict-free.  Register-staged double buffering: the next
// K-slab's global loads are issued before the MFMA block and written to the other LDS buffer after it.
#include "common.h"
#include <type_traits>
#include <utility>

namespace {

constexpr int BM = 128, BN = 128;

#ifdef VCR_TIMELINE
// Experiment-only instrumentation (profiles/timeline_linear.py builds a scratch library with -DVCR_TIMELINE): wave 0 of
// every workgroup stamps the 100 MHz wall clock at its start, after every tile's k loop and at its end.
__device__ unsigned long long vcr_tl[4096 * 16];
__device__ __forceinline__ void tl_mark(int slot) {
  if (threadIdx.x == 0 && blockIdx.x < 4096 && slot < 16) vcr_tl[blockIdx.x * 16 + slot] = wall_clock64();
}
#define TL(slot) tl_mark(slot)
#define VCR_TL_EXTRA_LDS(variant, stage) (((variant) & 256) ? 90 * 1024 - (stage) : 0)   /* experiment: one workgroup per CU */
#else
#define TL(slot) ((void)0)
#define VCR_TL_EXTRA_LDS(variant, stage) 0
#endif

template <int BK> struct TileT { float a[BM][BK + 4]; float b[BN][BK + 4]; };

// BK = 32: 73.7 KB LDS, 2 blocks/CU.  BK = 16: 40 KB LDS, 3 blocks/CU (VGPR-limited): the third block's MFMAs
// cover the other blocks' prologue / epilogue bubbles.
template <int BK>
__global__ __launch_bounds__(256, (BK == 32 ? 2 : 3)) void linear_kernel(vcr_linear_args p, int tiles_m, int tiles_n,
                                                                        int vec_epilogue) {
  using Tile = TileT<BK>;
  constexpr int CPR = BK / 4;                            // 16-B chunks per row
  constexpr int RPP = 256 / CPR;                         // rows covered per staging pass
  constexpr int NPASS = BM / RPP;                        // staging passes per operand
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Tile* tile = reinterpret_cast<Tile*>(smem);           // [2]
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);   // scalar: LDS-DMA targets via SALU
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware mapping: consecutive block ids land on different XCDs (round robin), so give every
  // XCD a contiguous run of tiles; tiles that share an X panel (same tm) then share an L2.
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  bid = xcd_chunk(bid, nblk);
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // global -> register staging: thread owns NPASS rows (r0 + RPP i) x one 16-B column chunk of each operand
  const int r0 = t / CPR, c4 = (t % CPR) * 4;
  const float* xa[NPASS];
  const float* wb[NPASS];
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    xa[i] = p.x + (size_t)min(m0 + r0 + RPP * i, p.M - 1) * p.ldx + c4;
    wb[i] = p.w + (size_t)min(n0 + r0 + RPP * i, p.N - 1) * p.K + c4;
  }
  f32x4 ra[NPASS], rb[NPASS];
#pragma unroll
  for (int i = 0; i < NPASS; ++i) { ra[i] = ld4(xa[i]); rb[i] = ld4(wb[i]); }
#pragma unroll
  for (int i = 0; i < NPASS; ++i) { st4(&tile[0].a[r0 + RPP * i][c4], ra[i]); st4(&tile[0].b[r0 + RPP * i][c4], rb[i]); }
  __syncthreads();

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0};

  const int nk = p.K / BK;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
#pragma unroll
      for (int i = 0; i < NPASS; ++i) { ra[i] = ld4(xa[i] + (kt + 1) * BK); rb[i] = ld4(wb[i] + (kt + 1) * BK); }
    }
    const Tile& T = tile[cur];
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      f32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = ld4(&T.a[wm * 64 + i * 32 + l31][8 * g + 4 * half]);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = ld4(&T.b[wn * 64 + j * 32 + l31][8 * g + 4 * half]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fa[i][s], fb[j][s], acc[i][j]);
    }
    if (kt + 1 < nk) {
      Tile& Nx = tile[cur ^ 1];
#pragma unroll
      for (int i = 0; i < NPASS; ++i) { st4(&Nx.a[r0 + RPP * i][c4], ra[i]); st4(&Nx.b[r0 + RPP * i][c4], rb[i]); }
    }
    __syncthreads();
  }

  // epilogue.  Fast path: transpose the wave's 64x64 tile through its slice of the (now free) LDS so that
  // every lane owns 4 consecutive columns: bias / ReLU / residual / store all move 16 B per lane and a
  // wave-instruction covers 4 rows x 256 contiguous bytes (the accumulator layout itself only offers
  // 4-byte accesses at a 32-lane stride per row).
  if (vec_epilogue) {
    constexpr int EP = 68;
    float* ot = reinterpret_cast<float*>(smem) + wave * 32 * EP;     // 8.7 KB per wave, reused for both row halves
    const int c4e = (lane & 15) * 4, col = n0 + wn * 64 + c4e;
    const f32x4 bias = (p.bias && col < p.N) ? ld4(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[acc_row(r, half) * EP + j * 32 + l31] = acc[i][j][r];
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      if (col < p.N) {
#pragma unroll 4
        for (int ps = 0; ps < 8; ++ps) {
          const int rl = ps * 4 + (lane >> 4);
          const int row = m0 + wm * 64 + i * 32 + rl;
          if (row < p.M) {
            f32x4 v = ld4(&ot[rl * EP + c4e]) + bias;
            if (p.relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
            if (p.residual) v = v + ld4(p.residual + (size_t)row * p.ldr + col);
            st4(p.y + (size_t)row * p.ldy + col, v);
          }
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  // generic path: D[row = (r&3)+8(r>>2)+4 half][col = l31]; a half-wave writes 128 contiguous bytes per row
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + l31;
    if (col >= p.N) continue;
    const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + acc_row(r, half);
        if (row < p.M) {
          float v = acc[i][j][r] + bias;
          if (p.relu) v = fmaxf(v, 0.f);
          if (p.residual) v += p.residual[(size_t)row * p.ldr + col];
          p.y[(size_t)row * p.ldy + col] = v;
        }
      }
    }
  }
}

// ---- direct-to-LDS variant: global_load_lds_dwordx4 (LDS-DMA) instead of register staging.
// One wave-instruction lands 1 KiB = 8 rows x 128 B CONTIGUOUSLY (wave-uniform base + lane*16), so the LDS
// image cannot be padded; bank conflicts are avoided by an XOR swizzle of the 16-B chunk index,
// pc = lc ^ ((row >> 1) & 7), applied to the per-lane GLOBAL source address when filling (the LDS side stays
// linear) and to the ds_read_b128 address when reading (both sides or neither).  With 128-B rows two rows
// share a 256-B bank row, so a 16-lane b128 group (16 distinct rows) hits 16 distinct slots.
// No VGPRs hold the in-flight slab and there is no ds_write pass.
template <int BK> struct TileGT { float a[BM][BK]; float b[BN][BK]; };
using TileG = TileGT<32>;

__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// LN_IN : the A operand is LayerNorm(x) of model/transformer.py:141-144 without ever forming it.  With
//         x_hat = (x - mean) / (std + eps),  W (a * x_hat + b) + bias = inv * (W' x - mean * c) + d  where
//         W' = W diag(a), c_n = sum_k W'_nk, d = bias + W b are folded once per weight (vcr_fold_layernorm_f32):
//         the main loop is the plain GEMM on W', the epilogue applies the per-row (mean, inv) rebuilt from the
//         per-64-column partial sums (sum, sum of squares) that the PRODUCING linear wrote from its epilogue
//         (STATS_OUT).  The separate LayerNorm launch and its 2 x M x 512 x 4 B round trip disappear
//          at no cost in the MFMA loop.
// STATS_OUT: epilogue also writes, per row and per 64-column segment, (sum y, sum y^2) of the final outputs.
// BK    : k-slab per stage.  32: 128-B LDS rows (8 rows per KiB of LDS-DMA, swizzle pc = lc ^ ((row >> 1) & 7)), 64 KB
//         of staging, two workgroups per CU; the residual tile (if any) is prefetched across the GEMM loop (64 VGPRs).
//         16: 64-B rows (16 rows per KiB, pc = lc ^ ((row >> 2) & 3)), 35 KB and ~100 VGPRs, FOUR workgroups per CU (4
//         waves per SIMD from independent workgroups cover each other's k-step barriers): +2-3 % on the launches
//         without a residual, which is what the launcher uses it for.  Same k order: results are bit-identical.
template <int BK, bool LN_IN, bool STATS_OUT>
__global__ __launch_bounds__(256, (BK == 32 ? 2 : 4)) void linear_glds_kernel(vcr_linear_args p, int tiles_m, int tiles_n) {
  using Tile = TileGT<BK>;
  constexpr int CPR = BK / 4;                            // 16-B chunks per LDS row
  constexpr int RPK = 64 / CPR;                          // rows per 1-KiB LDS-DMA wave instruction
  constexpr int NF = 32 / RPK;                           // fills per operand per wave (32 rows each)
  constexpr int SWS = BK == 32 ? 1 : 2;                  // swizzle: chunk ^= (row >> SWS) & (CPR - 1)
  constexpr int EP = 68;
  constexpr int STAGE_BYTES = 2 * sizeof(Tile) > 4 * 32 * EP * 4 ? 2 * sizeof(Tile) : 4 * 32 * EP * 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Tile* tile = reinterpret_cast<Tile*>(smem);            // [2]
  float* rowst = reinterpret_cast<float*>(smem + STAGE_BYTES);   // [BM][2] (mean, inv) per row (LN_IN only)
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);   // scalar: LDS-DMA targets via SALU
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  bid = xcd_chunk(bid, nblk);
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  TL(0);

  // fill mapping: wave w covers rows w*32 + RPK*i + lane / CPR, physical chunk lane % CPR
  const int frow = lane / CPR, fpc = lane % CPR;
  const float* xa[NF];
  const float* wb[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    const int row = wave * 32 + RPK * i + frow;
    const int lc = fpc ^ ((row >> SWS) & (CPR - 1));
    xa[i] = p.x + (size_t)min(m0 + row, p.M - 1) * p.ldx + 4 * lc;
    wb[i] = p.w + (size_t)min(n0 + row, p.N - 1) * p.K + 4 * lc;
  }
  auto fill = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      glds16(xa[i] + k0, &tile[buf].a[wave * 32 + RPK * i][0]);
      glds16(wb[i] + k0, &tile[buf].b[wave * 32 + RPK * i][0]);
    }
  };
  fill(0, 0);
  if (LN_IN && t < BM) {
    const float* sp = p.ln_stats_in + (size_t)min(m0 + t, p.M - 1) * p.ln_nseg * 2;
    float s1 = 0.f, s2 = 0.f;
    for (int sg = 0; sg < p.ln_nseg; ++sg) { s1 += sp[2 * sg]; s2 += sp[2 * sg + 1]; }     // fixed order
    const float mean = s1 / (float)p.K;
    const float var = fmaxf((s2 - s1 * mean) / (float)(p.K - 1), 0.f);                      // unbiased, like x.std()
    rowst[2 * t] = mean;
    rowst[2 * t + 1] = 1.f / (sqrtf(var) + p.ln_eps);
  }
  __syncthreads();
  TL(1);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0};
  int ra_[2], rb_[2], sa[2], sb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ra_[i] = wm * 64 + i * 32 + l31; sa[i] = (ra_[i] >> SWS) & (CPR - 1);
    rb_[i] = wn * 64 + i * 32 + l31; sb[i] = (rb_[i] >> SWS) & (CPR - 1);
  }
  // BK 32: the residual tile does not depend on the GEMM: fetch this lane's 16 chunks now, so that the epilogue's
  // load -> add -> store chain does not start with an HBM round trip (64 VGPRs; the kernel runs 2 waves per SIMD).
  // BK 16 (four workgroups per CU, ~100 VGPRs): no room for that; a residual is read in the epilogue.
  constexpr bool PREFETCH_RES = BK == 32;
  f32x4 resv[PREFETCH_RES ? 2 : 1][PREFETCH_RES ? 8 : 1];
  if (PREFETCH_RES) {
    const int colr = n0 + wn * 64 + (lane & 15) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        const int row = m0 + wm * 64 + i * 32 + ps * 4 + (lane >> 4);
        resv[PREFETCH_RES ? i : 0][PREFETCH_RES ? ps : 0] =
            (p.residual && row < p.M && colr < p.N) ? ld4(p.residual + (size_t)row * p.ldr + colr) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  }
  const int nk = p.K / BK;
#ifdef VCR_TIMELINE
  if (p.variant & 512) {                                 // experiment: fragments of group g+1 requested before the MFMAs of g
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) fill(cur ^ 1, (kt + 1) * BK);
      const Tile& T = tile[cur];
      f32x4 fa[2][2], fb[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[0][i] = ld4(&T.a[ra_[i]][4 * ((half) ^ sa[i])]);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[0][j] = ld4(&T.b[rb_[j]][4 * ((half) ^ sb[j])]);
#pragma unroll
      for (int g = 0; g < BK / 8; ++g) {
        if (g + 1 < BK / 8) {
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[(g + 1) & 1][i] = ld4(&T.a[ra_[i]][4 * ((2 * (g + 1) + half) ^ sa[i])]);
#pragma unroll
          for (int j = 0; j < 2; ++j) fb[(g + 1) & 1][j] = ld4(&T.b[rb_[j]][4 * ((2 * (g + 1) + half) ^ sb[j])]);
        }
        __builtin_amdgcn_sched_barrier(0);               // keep the reads ABOVE this group's MFMAs
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fa[g & 1][i][s], fb[g & 1][j][s], acc[i][j]);
      }
      __syncthreads();
    }
  } else
#endif
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) fill(cur ^ 1, (kt + 1) * BK);
    const Tile& T = tile[cur];
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      f32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = ld4(&T.a[ra_[i]][4 * ((2 * g + half) ^ sa[i])]);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = ld4(&T.b[rb_[j]][4 * ((2 * g + half) ^ sb[j])]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fa[i][s], fb[j][s], acc[i][j]);
    }
    __syncthreads();                                     // drains the LDS-DMA (vmcnt(0)) and orders the buffers
  }
  TL(2);

  // epilogue: transpose the wave's 64x64 tile through its slice of the (now free) staging LDS so that every lane owns 4
  // consecutive columns: bias / LayerNorm / ReLU / residual / store all move 16 B per lane
  float* ot = reinterpret_cast<float*>(smem) + wave * 32 * EP;
  const int c4e = (lane & 15) * 4, col = n0 + wn * 64 + c4e;
  const f32x4 bias = (p.bias && col < p.N) ? ld4(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 csum = (LN_IN && col < p.N) ? ld4(p.ln_colsum + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  int run_pt = -1;                                       // fused max-pool state (segmax_out only)
  f32x4 run_mx = {0.f, 0.f, 0.f, 0.f};
  auto seg_flush = [&]() {
    if (run_pt >= 0 && (lane >> 4) == 0 && col < p.N) {
      int* o = reinterpret_cast<int*>(p.segmax_out + (size_t)run_pt * p.ld_segmax + col);
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicMax(o + e, __float_as_int(run_mx[e]));
    }
    run_pt = -1;
  };
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[acc_row(r, half) * EP + j * 32 + l31] = acc[i][j][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (p.segmax_out) {
      // (uniform) fused EdgeConv max-pool, DGCNN conv2..conv4 -- no LayerNorm / residual / statistics here.  Rows are
      // edges, seg_k consecutive rows per point.  The wave walks its 64 rows in ascending order (i, then ps, then
      // lane >> 4); the maximum of a point is kept in registers while the point lasts -- folded over the four lane
      // groups of a ps step when they share the point -- and goes out as ONE integer atomic max per column and run
      // (post-ReLU values are >= 0: their bit patterns order like ints; the target was pre-set to 0 by vcr_edgerows_f32).
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        const int rl = ps * 4 + (lane >> 4);
        const int row0 = m0 + wm * 64 + i * 32 + ps * 4, row = row0 + (lane >> 4);   // row0: wave-uniform
        f32x4 v = ld4(&ot[rl * EP + c4e]) + bias;
        v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
        if (row >= p.M || col >= p.N) v = f32x4{0.f, 0.f, 0.f, 0.f};
        else if (p.y) st4(p.y + (size_t)row * p.ldy + col, v);
        if (row0 >= p.M) continue;
        const int pa = row0 / p.seg_k, pb = min(row0 + 3, p.M - 1) / p.seg_k;
        if (pa == pb) {                                  // the four rows of this step belong to one point
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = fmaxf(v[e], __shfl_xor(v[e], 16, 64));
            v[e] = fmaxf(v[e], __shfl_xor(v[e], 32, 64));
          }
          if (pa != run_pt) { seg_flush(); run_pt = pa; run_mx = v; }
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) run_mx[e] = fmaxf(run_mx[e], v[e]);
          }
        } else {                                         // a point boundary inside the step: every row for itself
          seg_flush();
          if (row < p.M && col < p.N) {
            int* o = reinterpret_cast<int*>(p.segmax_out + (size_t)(row / p.seg_k) * p.ld_segmax + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicMax(o + e, __float_as_int(v[e]));
          }
        }
      }
      if (i == 1) seg_flush();
    } else if (col < p.N) {
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        const int rl = ps * 4 + (lane >> 4);
        const int row = m0 + wm * 64 + i * 32 + rl;
        if (row < p.M) {
          f32x4 v = ld4(&ot[rl * EP + c4e]);
          if (LN_IN) {
            const float mean = rowst[2 * (wm * 64 + i * 32 + rl)], inv = rowst[2 * (wm * 64 + i * 32 + rl) + 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(inv, fmaf(-mean, csum[e], v[e]), bias[e]);
          } else {
            v = v + bias;
          }
          if (p.relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
          if (p.residual) v = v + (PREFETCH_RES ? resv[PREFETCH_RES ? i : 0][PREFETCH_RES ? ps : 0]
                                                : ld4(p.residual + (size_t)row * p.ldr + col));
          st4(p.y + (size_t)row * p.ldy + col, v);
          if (STATS_OUT) {                               // the 16 lanes of a row group (= one DPP row) hold this
            float s1 = (v[0] + v[1]) + (v[2] + v[3]);    // wave's 64 columns of the row
            float s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            s1 = row16_sum(s1); s2 = row16_sum(s2);
            if ((lane & 15) == 0) {
              float* so = p.stats_out + ((size_t)row * (p.N / 64) + (n0 + wn * 64) / 64) * 2;
              so[0] = s1; so[1] = s2;
            }
          }
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
#ifdef VCR_TIMELINE
  __builtin_amdgcn_s_waitcnt(0);                         // stores acknowledged
  TL(3);
#endif
}

// ---- persistent variant with a DEFERRED epilogue (round 2).
// Measured on the round-1 kernels: the MFMA loop itself runs at 0.95 of the matrix peak (ffn2 vs wo: 3.6 us per
// 32-wide k-slab against 3.41 at peak), but every "round" of co-resident workgroups pays ~20 us on top -- all
// workgroups start together, run in lockstep and reach their epilogues at the same moment, so 33-67 MB of stores
// (+ residual reads) hit the memory system while no MFMA work is available anywhere on the chip.  25 rounds per
// forward = 0.5 of the 3.0 ms the linear family took.
// Here a workgroup walks over its tiles (grid = 2 per CU), the LDS-DMA slab pipeline runs straight across tile
// boundaries (the first slab of the next tile is requested during the last k-step of the current one), and the
// epilogue of tile i is issued from the accumulator registers in eight slices DURING the first eight k-steps of tile
// i+1: bias / LayerNorm / ReLU / residual / store / row statistics all ride under the next tile's MFMAs.  Only the
// last tile of a workgroup pays for its epilogue.  No LDS transpose: a lane stores its accumulator elements directly
// (32 lanes x 4 B = one full 128-B line per row), which also frees the LDS slice the transposed epilogue needed.
// The k order inside a tile is the one of the kernels above, so the GEMM results are bit-identical to theirs; the row
// statistics are summed in a different (still fixed) order.
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }   // f(0) .. f(N-1), index a constant

__device__ __forceinline__ float xor16_sum(float v) {    // + the value 16 lanes away (within each 32-lane half)
  return v + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}

template <bool LN_IN, bool STATS_OUT>
__global__ __launch_bounds__(256, 2) void linear_persist_kernel(vcr_linear_args p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  TileG* tile = reinterpret_cast<TileG*>(smem);          // [2]
  float* rowst = reinterpret_cast<float*>(smem + 2 * sizeof(TileG));   // [3][BM][2] (mean, inv): previous / current / next tile
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int nblk = tiles_m * tiles_n, G = gridDim.x;
  const int nk = p.K / 32;

  // virtual block id -> tile, XCD-aware: ids congruent mod 8 run on one XCD (G is a multiple of 8 or == nblk), and
  // every XCD owns a contiguous run of tiles, so tiles sharing an X panel share an L2
  auto tile_of = [&](int vb, int& m0, int& n0) {
    const int bid = xcd_chunk(vb, nblk);
    m0 = (bid / tiles_n) * BM; n0 = (bid % tiles_n) * BN;
  };
  const int frow = lane >> 3, fpc = lane & 7;
  const float* xa[4];
  const float* wb[4];
  auto set_tile = [&](int m0, int n0) {
#ifdef VCR_TIMELINE
    if (p.variant & 128) { m0 = (m0 / BM % 2) * BM; }    // experiment: every X tile comes from the first 256 rows (L2 hits)
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 32 + 8 * i + frow;
      const int lc = fpc ^ ((row >> 1) & 7);
      xa[i] = p.x + (size_t)min(m0 + row, p.M - 1) * p.ldx + 4 * lc;
      wb[i] = p.w + (size_t)min(n0 + row, p.N - 1) * p.K + 4 * lc;
    }
  };
  auto fill = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(xa[i] + k0, &tile[buf].a[wave * 32 + 8 * i][0]);
      glds16(wb[i] + k0, &tile[buf].b[wave * 32 + 8 * i][0]);
    }
  };
  auto row_stats = [&](int m0, int rs) {                 // LayerNorm (mean, 1/(std+eps)) of the tile's 128 rows
    if (LN_IN && t < BM) {
      const float* sp = p.ln_stats_in + (size_t)min(m0 + t, p.M - 1) * p.ln_nseg * 2;
      float s1 = 0.f, s2 = 0.f;
      for (int sg = 0; sg < p.ln_nseg; ++sg) { s1 += sp[2 * sg]; s2 += sp[2 * sg + 1]; }   // fixed order
      const float mean = s1 / (float)p.K;
      const float var = fmaxf((s2 - s1 * mean) / (float)(p.K - 1), 0.f);                    // unbiased, like x.std()
      rowst[(rs * BM + t) * 2] = mean;
      rowst[(rs * BM + t) * 2 + 1] = 1.f / (sqrtf(var) + p.ln_eps);
    }
  };

  int vb = blockIdx.x;
  if (vb >= nblk) return;
  int m0, n0;
  TL(0);
  tile_of(vb, m0, n0);
  set_tile(m0, n0);
  fill(0, 0);
  row_stats(m0, 0);
  __syncthreads();
  TL(1);
  int tl_slot = 2;

  f32x16 acc[2][2], pacc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) { acc[i][j] = f32x16{0}; pacc[i][j] = f32x16{0}; }
  int ra_[2], rb_[2], sa[2], sb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ra_[i] = wm * 64 + i * 32 + l31; sa[i] = (ra_[i] >> 1) & 7;
    rb_[i] = wn * 64 + i * 32 + l31; sb[i] = (rb_[i] >> 1) & 7;
  }

  // ---- deferred epilogue of the PREVIOUS tile, slice ph of 8: rows i*32 + 8*rq + 4*half + (0..3), both j
  int pm0 = 0, pn0 = 0, prs = 0;
  float pbias[2] = {0.f, 0.f}, pcsum[2] = {0.f, 0.f};
  float res[2][4];
  auto epi_load = [&](auto PH) {                         // residual elements of the slice: requested before the MFMAs
    constexpr int ph = decltype(PH)::value, i = ph >> 2, rq = ph & 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = pn0 + wn * 64 + j * 32 + l31;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = pm0 + wm * 64 + i * 32 + 8 * rq + 4 * half + e;
        res[j][e] = (p.residual && row < p.M && col < p.N) ? p.residual[(size_t)row * p.ldr + col] : 0.f;
      }
    }
  };
  // values of the slice, final (bias / LayerNorm / ReLU / residual applied), parked until the k-step's barrier has
  // passed: every barrier drains vmcnt(0) for the LDS-DMA, so a store (or load) issued just BEFORE one would be waited
  // for at once -- stores go out right AFTER a barrier and have a whole k-step of MFMAs to complete
  float pend[2][4];
  auto epi_compute = [&](auto PH) {
    constexpr int ph = decltype(PH)::value, i = ph >> 2, rq = ph & 3;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rl = wm * 64 + i * 32 + 8 * rq + 4 * half + e;      // == acc_row(rq*4 + e, half) within the 32-row tile
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float x = pacc[i][j][rq * 4 + e];
        if (LN_IN) {
          const float mean = rowst[(prs * BM + rl) * 2], inv = rowst[(prs * BM + rl) * 2 + 1];
          x = fmaf(inv, fmaf(-mean, pcsum[j], x), pbias[j]);
        } else {
          x = x + pbias[j];
        }
        if (p.relu) x = fmaxf(x, 0.f);
        if (p.residual) x = x + res[j][e];
        pend[j][e] = x;
      }
    }
  };
  auto epi_flush = [&](auto PH) {
    constexpr int ph = decltype(PH)::value, i = ph >> 2, rq = ph & 3;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = pm0 + wm * 64 + i * 32 + 8 * rq + 4 * half + e;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = pn0 + wn * 64 + j * 32 + l31;
        if (row < p.M && col < p.N) p.y[(size_t)row * p.ldy + col] = pend[j][e];
      }
      if (STATS_OUT) {                                   // this wave's 64 columns of the row = 32 lanes x 2 j-tiles
        float s1 = pend[0][e] + pend[1][e], s2 = pend[0][e] * pend[0][e] + pend[1][e] * pend[1][e];
        s1 = xor16_sum(row16_sum(s1)); s2 = xor16_sum(row16_sum(s2));
        if (l31 == 0 && row < p.M) {
          float* so = p.stats_out + ((size_t)row * (p.N / 64) + (pn0 + wn * 64) / 64) * 2;
          so[0] = s1; so[1] = s2;
        }
      }
    }
  };
  auto retire = [&]() {                                  // the finished tile becomes "previous"
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) { pacc[i][j] = acc[i][j]; acc[i][j] = f32x16{0}; }
    pm0 = m0; pn0 = n0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
      pbias[j] = (p.bias && col < p.N) ? p.bias[col] : 0.f;
      pcsum[j] = (LN_IN && col < p.N) ? p.ln_colsum[col] : 0.f;
    }
  };

  int buf = 0, rs = 0;
  bool have_prev = false;
  int nm0 = 0, nn0 = 0;
  bool has_next = false;
  // one k-step: request the next slab (possibly the next tile's first), MFMAs on the current one
  auto kstep_begin = [&](int kt) {
    if (kt + 1 < nk) {
      fill(buf ^ 1, (kt + 1) * 32);
    } else if (has_next) {                               // last k-step: the slab pipeline crosses into the next tile
      set_tile(nm0, nn0);
      fill(buf ^ 1, 0);
    }
  };
  auto kstep_mfma = [&]() {
    const TileG& T = tile[buf];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = ld4(&T.a[ra_[i]][4 * ((2 * g + half) ^ sa[i])]);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = ld4(&T.b[rb_[j]][4 * ((2 * g + half) ^ sb[j])]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fa[i][s], fb[j][s], acc[i][j]);
    }
  };
  auto kstep_end = [&]() {
    __syncthreads();                                     // drains the LDS-DMA (vmcnt(0)) and orders the buffers
    buf ^= 1;
  };
  for (;;) {
    const int vbn = vb + G;
    has_next = vbn < nblk;
    if (has_next) {
      tile_of(vbn, nm0, nn0);
      row_stats(nm0, rs == 2 ? 0 : rs + 1);              // next tile's LayerNorm rows: ready long before its epilogue
    }
    // the first eight k-steps carry the previous tile's epilogue, one slice each (straight-line code: the slices'
    // addresses must not become loop invariants that the compiler keeps live across the whole k loop)
    static_for<8>([&](auto PH) {
      constexpr int ph = decltype(PH)::value;
      if (ph < nk) {                                     // uniform
        if (ph > 0 && have_prev) epi_flush(std::integral_constant<int, (ph > 0 ? ph - 1 : 0)>{});
        kstep_begin(ph);
        if (have_prev) epi_load(PH);
        kstep_mfma();
        if (have_prev) epi_compute(PH);
        kstep_end();
      } else if (have_prev) {                            // short K: a slice that found no k-step to hide under
        if (ph == nk) epi_flush(std::integral_constant<int, (ph > 0 ? ph - 1 : 0)>{});
        epi_load(PH);
        epi_compute(PH);
        epi_flush(PH);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    if (have_prev && nk >= 8) epi_flush(std::integral_constant<int, 7>{});
    for (int kt = 8; kt < nk; ++kt) {
      kstep_begin(kt);
      kstep_mfma();
      kstep_end();
    }
    TL(tl_slot); ++tl_slot;
    prs = rs;
    retire();
    have_prev = true;
    if (!has_next) break;
    vb = vbn; m0 = nm0; n0 = nn0; rs = rs == 2 ? 0 : rs + 1;
  }
  // the last tile's epilogue is the only exposed one
  static_for<8>([&](auto PH) {
    epi_load(PH);
    epi_compute(PH);
    epi_flush(PH);
    __builtin_amdgcn_sched_barrier(0);                   // one slice in flight: keeps the register budget of the main loop
  });
#ifdef VCR_TIMELINE
  __builtin_amdgcn_s_waitcnt(0);
  TL(tl_slot);
#endif
}


}  // namespace

#ifdef VCR_TIMELINE
extern "C" int vcr_dbg_timeline(unsigned long long* host_dst, int clear) {
  if (clear) {
    static unsigned long long zeros[4096 * 16];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(vcr_tl), zeros, sizeof(zeros));
  }
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(vcr_tl), sizeof(unsigned long long) * 4096 * 16);
}
#endif

extern "C" int vcr_linear_f32(const vcr_linear_args* a, vcr_stream_t stream) {
  if (!a || !a->x || !a->w || (!a->y && !a->segmax_out)) return VCR_EINVAL;
  if (a->segmax_out && (a->seg_k <= 0 || !a->relu || a->residual || a->ln_stats_in || a->stats_out || (a->ld_segmax & 3) ||
                        a->ld_segmax < a->N || ((uintptr_t)a->segmax_out & 15) || (a->variant & (1 | 4 | 32))))
    return VCR_EINVAL;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || (a->K % 32) != 0) return VCR_EINVAL;
  const int variant = a->variant;                    // tuning / test selector carried by the call (see vcr_hip.h)
  if ((a->ldx & 3) || a->ldx < a->K || (a->y && a->ldy < a->N) || (a->residual && a->ldr < a->N)) return VCR_EINVAL;
  if (((uintptr_t)a->x | (uintptr_t)a->w) & 15) return VCR_EINVAL;
  const int tiles_m = (a->M + BM - 1) / BM, tiles_n = (a->N + BN - 1) / BN;
  const int lds32 = 2 * sizeof(TileT<32>), lds16 = 2 * sizeof(TileT<16>);
  static_assert(2 * sizeof(TileT<16>) >= 4 * 32 * 68 * 4, "epilogue slice fits the staging buffers");
  const int vec = (a->N % 4 == 0) && (!a->y || ((a->ldy % 4 == 0) && (((uintptr_t)a->y & 15) == 0))) &&
                  (!a->bias || ((uintptr_t)a->bias & 15) == 0) &&
                  (!a->residual || ((a->ldr % 4 == 0) && ((uintptr_t)a->residual & 15) == 0));
  const bool persist = (variant & 32) != 0;
  if (a->ln_stats_in || a->stats_out) {                  // fused LayerNorm prologue / statistics epilogue
    if (a->ln_stats_in && (!a->ln_colsum || !a->bias || a->ln_nseg <= 0 || a->K < 2)) return VCR_EINVAL;
    if (a->stats_out && (a->N % 64)) return VCR_EINVAL;
    if (!persist) {                                      // the one-tile-per-workgroup kernels move 16 B per lane
      if (!vec || (variant & (4 | 1))) return VCR_EUNSUPPORTED;
      if (a->ln_stats_in && ((uintptr_t)a->ln_colsum & 15)) return VCR_EINVAL;
    }
  }
  if (persist) {
    // opt-in (bit 5): persistent workgroups (two per CU), LDS-DMA staging, epilogue deferred under the next tile's
    // MFMAs.  Measured SLOWER than the default on every shape of the path, kept as the tested
    // record of that experiment and because it has no 16-byte alignment requirement on y / bias / residual.
    const bool ln_in = a->ln_stats_in != nullptr, st_out = a->stats_out != nullptr;
    const int nblk = tiles_m * tiles_n, slots = 2 * vcr_cu_count();
    const int grid = nblk < slots ? nblk : slots;
    const int ldsp = 2 * sizeof(TileG) + 3 * BM * 2 * 4;
#define VCR_LINP_LAUNCH(LI, SO)                                                                                         \
  do {                                                                                                                   \
    VCR_DYN_LDS((linear_persist_kernel<LI, SO>), ldsp);                                                                  \
    hipLaunchKernelGGL((linear_persist_kernel<LI, SO>), dim3(grid), dim3(256), ldsp, (hipStream_t)stream, *a, tiles_m,   \
                       tiles_n);                                                                                         \
  } while (0)
    if (ln_in && st_out) VCR_LINP_LAUNCH(true, true);
    else if (ln_in) VCR_LINP_LAUNCH(true, false);
    else if (st_out) VCR_LINP_LAUNCH(false, true);
    else VCR_LINP_LAUNCH(false, false);
#undef VCR_LINP_LAUNCH
    return VCR_LAUNCH_RC();
  }
  if (a->segmax_out && !vec) return VCR_EUNSUPPORTED;
  if (!(variant & 4) && vec) {   // LDS-DMA staging, one tile per workgroup
    const bool ln_in = a->ln_stats_in != nullptr, st_out = a->stats_out != nullptr;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(tiles_m * tiles_n);
#define VCR_LIN_LAUNCH(BKV, LI, SO)                                                                                     \
  do {                                                                                                                   \
    const int stage = 2 * (int)sizeof(TileGT<BKV>) > 4 * 32 * 68 * 4 ? 2 * (int)sizeof(TileGT<BKV>) : 4 * 32 * 68 * 4;    \
    const int lds = stage + (LI ? BM * 2 * 4 : 0) + VCR_TL_EXTRA_LDS(variant, stage);                                   \
    VCR_DYN_LDS((linear_glds_kernel<BKV, LI, SO>), lds);                                                                 \
    hipLaunchKernelGGL((linear_glds_kernel<BKV, LI, SO>), grid, dim3(256), lds, s, *a, tiles_m, tiles_n);               \
  } while (0)
#define VCR_LIN_PICK(BKV)                                                                                               \
  do {                                                                                                                   \
    if (ln_in && st_out) VCR_LIN_LAUNCH(BKV, true, true);                                                                \
    else if (ln_in) VCR_LIN_LAUNCH(BKV, true, false);                                                                    \
    else if (st_out) VCR_LIN_LAUNCH(BKV, false, true);                                                                   \
    else VCR_LIN_LAUNCH(BKV, false, false);                                                                              \
  } while (0)
    // Without a residual: BK = 16, four workgroups per CU (measured +2-3 % on the qkv / ffn1 / kv projections).
    // With one: BK = 32 and the residual tile prefetched across the GEMM loop (bit 3 forces BK 32, bit 6 BK 16).
    if ((!a->residual || (variant & 64)) && !(variant & 8)) VCR_LIN_PICK(16);
    else VCR_LIN_PICK(32);
#undef VCR_LIN_PICK
#undef VCR_LIN_LAUNCH
  } else if (variant & 1) {
    VCR_DYN_LDS(linear_kernel<16>, lds16);
    hipLaunchKernelGGL(linear_kernel<16>, dim3(tiles_m * tiles_n), dim3(256), lds16, (hipStream_t)stream, *a, tiles_m,
                       tiles_n, vec);
  } else {
    VCR_DYN_LDS(linear_kernel<32>, lds32);
    hipLaunchKernelGGL(linear_kernel<32>, dim3(tiles_m * tiles_n), dim3(256), lds32, (hipStream_t)stream, *a, tiles_m,
                       tiles_n, vec);
  }
  return VCR_LAUNCH_RC();
}
