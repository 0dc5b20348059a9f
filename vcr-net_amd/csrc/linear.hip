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
// MFMA shape of the LDS-DMA kernels when the call does not force one (vcr_linear_args.variant bits 4 / 10)
constexpr int LINEAR_MS_DEFAULT = 0;   // 0 = per launch (see vcr_linear_f32), 16 / 32 = that shape everywhere (probe_build.py --set)

// Lines that start with "//@probe " are inert here: profiles/experiments/probe_build.py uncomments them in a scratch copy
// built against profiles/experiments/probes.h (in-kernel clock stamps; DESIGN.md, measurement protocol).

// OPTIONAL block accumulation (compile-time, off in the product build).  1: every linear kernel adds its MFMA accumulators into
// a second register set at each multiple of 128 k and starts again from zero -- what a blocked CPU sgemm does (ATen's, i.e. the
// reference's), 3-5x closer to the exact dot product than ONE k-ascending chain of 512 / 1024 steps; the block boundaries are the
// same for every k-slab, MFMA shape, tile height and staging variant, so a shape's results still do not depend on the kernel
// configuration.  2: only the BK 32 kernels (a shape's bits then depend on the configuration).  0: one chain (rounds 1-4).
// Measured on one box (round 5, profiles/NOTES.md): final embeddings vs the reference's float64 twin, rms relative to the fp32
// reference's own: 0 -> 1.35x, 2 -> 1.10x, 1 -> 1.0x; pairs/s at BASELINE configs[1]: 3174-3179 / 3144-3150 / 3122-3127 -- the
// second accumulator set costs the BK 16 kernels their fourth workgroup per CU.  The head's scores, where a discrete decision
// hangs on the last bits, are block-accumulated unconditionally (pairscore.hip); here speed wins: 0.
constexpr int LINEAR_BLOCKED_ACC = 0;

template <int BK> struct TileT { float a[BM][BK + 4]; float b[BN][BK + 4]; };

// Register-staged kernel: the ALIGNMENT-FREE fallback of vcr_linear_f32 (any N, any row pitch of y / residual; the
// LDS-DMA kernels below need 16-B aligned rows).  BK = 32: 73.7 KB LDS, 2 blocks/CU.
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

  f32x16 tot[2][2];                                      // block accumulation (head of this file)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) tot[i][j] = f32x16{0};
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
    if (LINEAR_BLOCKED_ACC && (((kt + 1) * BK) % 128 == 0 || kt + 1 == nk)) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { tot[i][j] += acc[i][j]; acc[i][j] = f32x16{0}; }
    }
    if (kt + 1 < nk) {
      Tile& Nx = tile[cur ^ 1];
#pragma unroll
      for (int i = 0; i < NPASS; ++i) { st4(&Nx.a[r0 + RPP * i][c4], ra[i]); st4(&Nx.b[r0 + RPP * i][c4], rb[i]); }
    }
    __syncthreads();
  }
  if (LINEAR_BLOCKED_ACC) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = tot[i][j];
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
template <int BK, int BMV = BM> struct TileGT { float a[BMV][BK]; float b[BN][BK]; };

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
// BMV   : rows per block tile.  128 (wave tile 64 x 64), or 96 with MS = 16 (wave tile 48 x 64 = 3 x 4 tiles of 16 x 16): the
//         launcher picks 96 when 128-row tiles would leave a mostly empty last round of workgroups (M = 36 864 at BASELINE
//         configs[2]: 1152 tiles on 512 or 1024 slots = 2.25 / 1.125 rounds; 1536 tiles of 96 rows = 3.0 / 1.5).  Every
//         output element is the same k-ordered chain as in the 128-row MS = 16 kernel: bit-identical results.  64 and 32 rows
//         (wave tiles 32 x 64 / 16 x 64, BK 32 only) serve launches of far less than one round of workgroups -- one or two
//         pairs per call -- where a launch takes as long as ONE workgroup: four times as many workgroups of a quarter the
//         length.
// MS    : MFMA shape.  32 = v_mfma_f32_32x32x2_f32 (wave tile = 2x2 tiles, k = 8g + 4*half + s); 16 =
//         v_mfma_f32_16x16x4_f32 (wave tile = 4x4 tiles of 16x16, a lane's quarter q = lane >> 4 owns chunk q of a 16-wide
//         k group: k = 16g + 4q + s).  Same flops per cycle; the chip sustains a higher clock on the 16x16 shape under
//         the matrix pipe's power limit (profiles/rounds1-3/r2e_mfma_sustained_rates.txt: 132-140 vs 127-136 TFLOP/s).  The two
//         shapes sum k in different orders: results agree to fp32 rounding, not bitwise.
// (LN_IN = p.ln_stats_in != NULL and STATS_OUT = p.stats_out != NULL are run-time properties of the launch -- a uniform
//  branch in the prologue and two in the epilogue -- not template parameters: a quarter of the instantiations)
template <int BK, int MS, int BMV = BM>
__device__ __forceinline__ void linear_glds_body(const vcr_linear_args& p, int tiles_m, int tiles_n, int blk) {
  const bool LN_IN = p.ln_stats_in != nullptr, STATS_OUT = p.stats_out != nullptr;
  static_assert(BMV == 128 || ((BMV == 96 || BMV == 64 || BMV == 32) && MS == 16), "the lower tiles are built from 16 x 16 MFMA tiles");
  using Tile = TileGT<BK, BMV>;
  constexpr int WR = BMV / 2;                            // rows per wave tile
  constexpr int EPR = BMV == 128 ? 32 : 16;              // rows per epilogue pass
  constexpr int NPASS = WR / EPR, NPS = EPR / 4;
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
  const int bid = xcd_chunk(blk, nblk);
  // Tile order: M-major inside column GROUPS of <= 12 tiles (a 3 MB slice of W for K = 512), so that the weight slice an
  // XCD works on stays in its 4 MB L2 while the X panels stream past it.  Up to N = 1536 that is the plain M-major
  // order; for the stacked [6E, E] projection (24 column tiles, W = 6 MB) the plain order fetched 1.2 GB per launch
  // (rocprofv3 FETCH_SIZE, profiles/rounds1-3/r3f_pmc_summary.json) against 70 MB of operands.  A renumbering: same results.
  constexpr int GW = 12;
  const int grp = bid / (GW * tiles_m), wg = min(GW, tiles_n - grp * GW), loc = bid - grp * GW * tiles_m;
  const int tm = loc / wg, tn = grp * GW + loc % wg;
  const int m0 = tm * BMV, n0 = tn * BN;
  //@probe VCR_PROBE_STAMP(0);
  // fill mapping: wave w covers rows w*32 + RPK*i + lane / CPR, physical chunk lane % CPR
  // (BMV = 96: the A panel is BMV / RPK wave instructions, dealt to the waves round robin -- 3 each at BK 32; 2, 2, 1, 1 at BK 16)
  const int frow = lane / CPR, fpc = lane % CPR;
  constexpr int NFA = BMV == 128 ? NF : (BMV / RPK + 3) / 4;
  const float* xa[NFA];
  const float* wb[NF];
  int arow[NFA];
#pragma unroll
  for (int i = 0; i < NFA; ++i) {
    arow[i] = BMV == 128 ? wave * 32 + RPK * i : (wave + 4 * i) * RPK;      // wave-uniform
    const int row = min(arow[i], BMV - RPK) + frow;
    const int lc = fpc ^ ((row >> SWS) & (CPR - 1));
    xa[i] = p.x + (size_t)min(m0 + row, p.M - 1) * p.ldx + 4 * lc;
  }
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    const int row = wave * 32 + RPK * i + frow;
    const int lc = fpc ^ ((row >> SWS) & (CPR - 1));
    wb[i] = p.w + (size_t)min(n0 + row, p.N - 1) * p.K + 4 * lc;
  }
  auto fill = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < NFA; ++i)
      if (BMV == 128 || arow[i] < BMV) glds16(xa[i] + k0, &tile[buf].a[min(arow[i], BMV - RPK)][0]);
#pragma unroll
    for (int i = 0; i < NF; ++i) glds16(wb[i] + k0, &tile[buf].b[wave * 32 + RPK * i][0]);
  };
  fill(0, 0);
  if (LN_IN && t < BMV) {
    float mean, var;
    ln_row_moments(p.ln_stats_in + (size_t)min(m0 + t, p.M - 1) * p.ln_nseg * 2, p.ln_nseg, p.K, mean, var);
    rowst[2 * t] = mean;
    rowst[2 * t + 1] = 1.f / (sqrtf(var) + p.ln_eps);
  }
  __syncthreads();
  //@probe VCR_PROBE_STAMP(1);

  constexpr int NT = MS == 32 ? 2 : 4;                   // MFMA tiles per 64-wide wave-tile edge
  constexpr int NTM = WR / MS;                           // MFMA tile rows per wave tile (BMV 96: 3)
  const int qtr = lane >> 4, l15 = lane & 15;            // MS 16: quarter q supplies k = 4q + s of a 16-wide k group
  f32x16 acc[2][2];                                      // MS 32
  f32x4 acc4[MS == 16 ? NTM : 4][4];                     // MS 16 (the unused set is dead code)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0};
#pragma unroll
  for (int i = 0; i < (MS == 16 ? NTM : 4); ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int ra_[NT], rb_[NT], sa[NT], sb[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    ra_[i] = wm * WR + i * MS + (MS == 32 ? l31 : l15); sa[i] = (ra_[i] >> SWS) & (CPR - 1);     // (i < NTM used)
    rb_[i] = wn * 64 + i * MS + (MS == 32 ? l31 : l15); sb[i] = (rb_[i] >> SWS) & (CPR - 1);
  }
  // BK 32: the residual tile does not depend on the GEMM: fetch this lane's 16 chunks now, so that the epilogue's
  // load -> add -> store chain does not start with an HBM round trip (64 VGPRs; the kernel runs 2 waves per SIMD).
  // BK 16 (four workgroups per CU, ~100 VGPRs): no room for that; a residual is read in the epilogue.
  constexpr bool PREFETCH_RES = BK == 32;
  f32x4 resv[PREFETCH_RES ? NPASS : 1][PREFETCH_RES ? NPS : 1];
  if (PREFETCH_RES) {
    const int colr = n0 + wn * 64 + (lane & 15) * 4;
#pragma unroll
    for (int i = 0; i < NPASS; ++i)
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        const int row = m0 + wm * WR + i * EPR + ps * 4 + (lane >> 4);
        resv[PREFETCH_RES ? i : 0][PREFETCH_RES ? ps : 0] =
            (p.residual && row < p.M && colr < p.N) ? ld4(p.residual + (size_t)row * p.ldr + colr) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  }
  // optional block accumulation (head of this file)
  constexpr bool BLOCKED = LINEAR_BLOCKED_ACC == 1 || (LINEAR_BLOCKED_ACC == 2 && BK == 32);
  f32x16 tot[BLOCKED ? 2 : 1][BLOCKED ? 2 : 1];
  f32x4 tot4[BLOCKED ? (MS == 16 ? NTM : 4) : 1][BLOCKED ? 4 : 1];
  if constexpr (BLOCKED) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) tot[i][j] = f32x16{0};
#pragma unroll
    for (int i = 0; i < (MS == 16 ? NTM : 4); ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) tot4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto bank = [&]() {                                     // accumulators -> block totals
    if constexpr (BLOCKED) {
      if constexpr (MS == 32) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) { tot[i][j] += acc[i][j]; acc[i][j] = f32x16{0}; }
      } else {
#pragma unroll
        for (int i = 0; i < NTM; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) { tot4[i][j] += acc4[i][j]; acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      }
    }
  };
  const int nk = p.K / BK;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) fill(cur ^ 1, (kt + 1) * BK);
    const Tile& T = tile[cur];
    if constexpr (MS == 32) {
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
    } else {
#pragma unroll
      for (int g = 0; g < BK / 16; ++g) {
        f32x4 fa[NTM], fb[4];
#pragma unroll
        for (int i = 0; i < NTM; ++i) fa[i] = ld4(&T.a[ra_[i]][4 * ((4 * g + qtr) ^ sa[i])]);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = ld4(&T.b[rb_[j]][4 * ((4 * g + qtr) ^ sb[j])]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < NTM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc4[i][j] = mfma16(fa[i][s], fb[j][s], acc4[i][j]);
      }
    }
    if (BLOCKED && ((kt + 1) * BK) % 128 == 0 && kt + 1 < nk) bank();
    __syncthreads();                                     // drains the LDS-DMA (vmcnt(0)) and orders the buffers
  }
  if constexpr (BLOCKED) {                               // the last block, and the totals back into the epilogue's registers
    bank();
    if constexpr (MS == 32) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = tot[i][j];
    } else {
#pragma unroll
      for (int i = 0; i < NTM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc4[i][j] = tot4[i][j];
    }
  }
  //@probe VCR_PROBE_STAMP(2);

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
  for (int i = 0; i < NPASS; ++i) {
    if constexpr (MS == 32) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[acc_row(r, half) * EP + j * 32 + l31] = acc[i][j][r];
    } else {                                             // D[row = 4q + r][col = l15] of tile (EPR / 16 * i + ih, jt)
#pragma unroll
      for (int ih = 0; ih < EPR / 16; ++ih)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) ot[(16 * ih + 4 * qtr + r) * EP + jt * 16 + l15] = acc4[EPR / 16 * i + ih][jt][r];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    //@probe if (i == 0) VCR_PROBE_STAMP(4);                 // the wave's accumulators are complete and in LDS
    if (p.segmax_out) {
      // (uniform) fused EdgeConv max-pool, DGCNN conv2..conv4 -- no LayerNorm / residual / statistics here.  Rows are
      // edges, seg_k consecutive rows per point.  The wave walks its 64 rows in ascending order (i, then ps, then
      // lane >> 4); the maximum of a point is kept in registers while the point lasts -- folded over the four lane
      // groups of a ps step when they share the point -- and goes out as ONE integer atomic max per column and run
      // (post-ReLU values are >= 0: their bit patterns order like ints; the target was pre-set to 0 by vcr_edgerows_f32).
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        const int rl = ps * 4 + (lane >> 4);
        const int row0 = m0 + wm * WR + i * EPR + ps * 4, row = row0 + (lane >> 4);   // row0: wave-uniform
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
      if (i == NPASS - 1) seg_flush();
    } else if (col < p.N) {
      //@probe if (i == 0) { __builtin_amdgcn_s_waitcnt(0); VCR_PROBE_STAMP(9); }   // (bias / column sums have arrived)
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        //@probe if (i == 0 && ps == 1) VCR_PROBE_STAMP(7);
        //@probe if (i == 0 && ps == 4) VCR_PROBE_STAMP(8);
        const int rl = ps * 4 + (lane >> 4);
        const int row = m0 + wm * WR + i * EPR + rl;
        if (row < p.M) {
          f32x4 v = ld4(&ot[rl * EP + c4e]);
          if (LN_IN) {
            const float mean = rowst[2 * (wm * WR + i * EPR + rl)], inv = rowst[2 * (wm * WR + i * EPR + rl) + 1];
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
            s1 = row16_sum(s1);
            const float ms = s1 * (1.f / 64.f);          // segment mean; second moment ABOUT it (no cancellation when
            const float d0 = v[0] - ms, d1 = v[1] - ms, d2 = v[2] - ms, d3 = v[3] - ms;   // |mean| >> std)
            float s2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            s2 = row16_sum(s2);
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
    //@probe VCR_PROBE_STAMP(5 + i);                         // pass i: stores issued
  }
  //@probe __builtin_amdgcn_s_waitcnt(0); VCR_PROBE_STAMP(3);     // (stores acknowledged)
}

template <int BK, int MS, int BMV>
__global__ __launch_bounds__(256, (BK == 32 ? 2 : LINEAR_BLOCKED_ACC == 1 ? 3 : 4)) void linear_glds_kernel(vcr_linear_args p, int tiles_m, int tiles_n) {
  linear_glds_body<BK, MS, BMV>(p, tiles_m, tiles_n, (int)blockIdx.x);
}
// Two independent linears of the same kernel configuration as ONE launch (the first n0 workgroups work on p0, the rest
// on p1): the encoder's and the decoder's output projections, or enc.ffn1 beside dec.cross.q -- fewer, fuller rounds of
// workgroups; each tile is computed exactly as in its own launch.
template <int BK, int MS, int BMV>
__global__ __launch_bounds__(256, (BK == 32 ? 2 : LINEAR_BLOCKED_ACC == 1 ? 3 : 4)) void linear_glds_pair_kernel(vcr_linear_args p0, vcr_linear_args p1, int tm0,
                                                                                 int tn0, int tm1, int tn1) {
  const int n0 = tm0 * tn0;
  if ((int)blockIdx.x < n0) linear_glds_body<BK, MS, BMV>(p0, tm0, tn0, (int)blockIdx.x);
  else linear_glds_body<BK, MS, BMV>(p1, tm1, tn1, (int)blockIdx.x - n0);
}

}  // namespace

namespace {
struct LinearPlan { bool glds, bk16, ms16, bm_free, small_free, ln_in, st_out; int bm, tiles_m, tiles_n, vec, lds; long t96, t128; };

// Launches of less than one round (t128 < 2 x CUs): tile height in {128, 96, 64, 32} minimising g(ceil(tiles / CUs)) x
// (rows + 24) -- workgroups spread one per CU first; a workgroup costs its rows plus a fixed prologue / epilogue /
// pipeline-fill share (~24 rows' worth); two co-resident ones take 1.75 x one (profiles/rounds1-3/r3z_sweep_bm.txt), a third
// waits for a slot.  tiles_by_h: the launch's (or the pair's) tile counts at the four heights.
int small_grid_rows(const long* tiles_by_h) {
  static const int hs[4] = {128, 96, 64, 32};
  int best = 128;
  double bc = 1e30;
  const int cus = vcr_cu_count();
  for (int i = 0; i < 4; ++i) {
    const long per_cu = (tiles_by_h[i] + cus - 1) / cus;
    const double c = ((double)(per_cu / 2) * 1.75 + (double)(per_cu & 1)) * (hs[i] + 24);
    if (c < bc - 1e-9) { bc = c; best = hs[i]; }
  }
  return best;
}

// Relative cost of a launch of `tiles` workgroups of `rows`-row tiles on `slots` resident workgroups (two per CU), fitted to
// a sweep of M at both tile heights (profiles/rounds1-3/r3z_sweep_bm.txt: 56 shapes; the choice it makes loses 0.2 % on average and
// 5 % at worst against the better height, always-128 loses 8.9 % on average):  full rounds cost 1 each;  a LAST partial
// round filled to f costs half a round up to f = 0.3 (its workgroups have a CU to themselves), a whole one from f = 0.7
// (the dispatcher pairs them up on the CUs that free first), linear in between;  a launch of less than one round costs
// 0.6 while every workgroup gets its own CU (f <= 0.5) and 1 beyond.
double launch_cost(long tiles, int slots, int rows) {
  const long full = tiles / slots;
  const double f = (double)(tiles - full * slots) / slots;
  double c;
  if (full == 0) c = f <= 0.0 ? 0.0 : f <= 0.5 ? 0.6 : 1.0;
  else c = (double)full + (f <= 0.0 ? 0.0 : f <= 0.3 ? 0.5 : f >= 0.7 ? 1.0 : 0.5 + (f - 0.3) * 1.25);
  return c * rows;
}

// validation + kernel choice of one linear (shared by vcr_linear_f32 and vcr_linear_pair_f32)
// (bm_override: 0 = decide here, else the tile rows a paired launch decided for both of its halves; 64 / 32 also mean the
//  small-grid configuration BK 32 + 16x16x4)
int linear_plan(const vcr_linear_args* a, LinearPlan* pl, int bm_override = 0, bool in_pair = false) {
  if (!a || !a->x || !a->w || (!a->y && !a->segmax_out)) return VCR_EINVAL;
  if (a->segmax_out && (a->seg_k <= 0 || !a->relu || a->residual || a->ln_stats_in || a->stats_out || (a->ld_segmax & 3) ||
                        a->ld_segmax < a->N || ((uintptr_t)a->segmax_out & 15) || (a->variant & 4)))
    return VCR_EINVAL;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || (a->K % 32) != 0) return VCR_EINVAL;
  const int variant = a->variant;                    // tuning / test selector carried by the call (see vcr_hip.h)
  if (variant & ~(4 | 8 | 16 | 64 | 1024 | 2048 | 4096 | 8192 | 16384)) return VCR_EINVAL;   // retired selectors (1, 32, 128, 256, 512) are refused, not ignored
  if ((variant & 2048) && (variant & (4096 | 1024 | 4))) return VCR_EINVAL;    // 96-row tiles exist on the LDS-DMA 16x16x4 kernels only
  if ((variant & (8192 | 16384)) && ((variant & (4096 | 2048 | 1024 | 64 | 4)) || (variant & (8192 | 16384)) == (8192 | 16384)))
    return VCR_EINVAL;                                   // 64- / 32-row tiles: LDS-DMA, BK 32, 16x16x4 only
  if ((a->ldx & 3) || a->ldx < a->K || (a->y && a->ldy < a->N) || (a->residual && a->ldr < a->N)) return VCR_EINVAL;
  if (((uintptr_t)a->x | (uintptr_t)a->w) & 15) return VCR_EINVAL;
  pl->tiles_n = (a->N + BN - 1) / BN;
  pl->vec = (a->N % 4 == 0) && (!a->y || ((a->ldy % 4 == 0) && (((uintptr_t)a->y & 15) == 0))) &&
            (!a->bias || ((uintptr_t)a->bias & 15) == 0) &&
            (!a->residual || ((a->ldr % 4 == 0) && ((uintptr_t)a->residual & 15) == 0));
  if (a->ln_stats_in || a->stats_out) {                  // fused LayerNorm prologue / statistics epilogue
    if (a->ln_stats_in && (!a->ln_colsum || !a->bias || a->ln_nseg <= 0 || a->K < 2 || (a->K % a->ln_nseg))) return VCR_EINVAL;
    if (a->stats_out && (a->N % 64)) return VCR_EINVAL;
    if (!pl->vec || (variant & 4)) return VCR_EUNSUPPORTED;   // the LDS-DMA kernels move 16 B per lane
    if (a->ln_stats_in && ((uintptr_t)a->ln_colsum & 15)) return VCR_EINVAL;
  }
  if (a->segmax_out && !pl->vec) return VCR_EUNSUPPORTED;
  pl->glds = !(variant & 4) && pl->vec;                  // LDS-DMA staging, one tile per workgroup
  pl->ln_in = a->ln_stats_in != nullptr; pl->st_out = a->stats_out != nullptr;
  // Without a residual: BK = 16, four workgroups per CU (measured +2-3 % on the qkv / ffn1 / kv projections).
  // With one: BK = 32 and the residual tile prefetched across the GEMM loop (bit 3 forces BK 32, bit 6 BK 16).
  // MFMA shape: bit 4 (16) forces 16x16x4, bit 10 (1024) forces 32x32x2.  Automatic: 16x16x4 for the launches with a
  // residual (the BK 32 kernels: measured in the pipeline at BASELINE configs[1], wo 155 -> 151 us, ffn2 276 -> 266 us),
  // 32x32x2 for the BK 16 kernels (qkv / ffn1 / kv / q / conv3: equal within 1 %); DESIGN.md 5.1.
  // ... except a launch of its own whose grid is at most ONE round of the BK 16 kernel's 1024 slots (conv3 at BASELINE
  // configs[1]: 1024 tiles): every workgroup then stores its tile at the same time with no k loop left to run under the
  // stores; two rounds of the BK 32 kernel's 512 slots drift apart instead.  Measured inside the forward on one box
  // (profiles/rounds1-3/r3r_ab_conv3_bk32.txt): conv3 0.148 -> 0.1425 ms.  K >= 512 only (sn1_pq, K = 128: 0.047 -> 0.049 with BK 32).
  const bool one_round16 = !in_pair && variant == 0 && a->M >= 16384 && a->K >= 512 &&
                           (long)((a->M + 127) / 128) * pl->tiles_n <= 4L * vcr_cu_count();
  pl->bk16 = ((!a->residual && !one_round16) || (variant & 64)) && !(variant & 8);
  pl->ms16 = (variant & 16) ? true : (variant & 1024) ? false : (LINEAR_MS_DEFAULT == 16 || (LINEAR_MS_DEFAULT == 0 && !pl->bk16));
  // Tile rows: bit 11 (2048) forces 96, bit 12 (4096) forces 128.  Automatic: 96 when the launch cost model above
  // prefers it by > 2 %, on the BK 32 kernels only (two workgroups per CU; measured at BASELINE configs[2], M = 36 864:
  // ffn2 0.335 -> 0.309 ms, cross.wo 0.184 -> 0.170, the wo pair 0.357 -> 0.340.  The BK 16 kernels run four
  // workgroups per CU, their last round costs little, and 96-row tiles measured 0-6 % SLOWER there).
  const int slots = vcr_cu_count() * (pl->bk16 ? 4 : 2); // CUs (MI355X: 256) x resident workgroups per CU
  const long t128 = (long)((a->M + 127) / 128) * pl->tiles_n, t96 = (long)((a->M + 95) / 96) * pl->tiles_n;
  pl->t96 = t96; pl->t128 = t128;
  pl->bm_free = pl->glds && !(variant & (4096 | 2048 | 1024 | 8192 | 16384));   // nothing forces the tile rows or the 32x32x2 shape
  // Small problems (M < 16 384 rows: up to 7 pairs of 1024 points per call), nothing forced (tests and benchmarks that
  // pin BK / the MFMA shape keep what they ask for): ALWAYS the 16x16x4 shape, so that the results of a launch do not
  // depend on which tile height its grid makes it take (all heights are bit-identical on that shape) -- the merged and
  // the separate projections of one forward, for example, give the same bits at any size.
  pl->small_free = pl->glds && variant == 0 && a->M < 16384;
  if (pl->small_free) pl->ms16 = true;
  bool bm96 = pl->glds && !(variant & (4096 | 1024 | 8192 | 16384)) &&
              ((variant & 2048) || (!pl->bk16 && 1.02 * launch_cost(t96, slots, 96) < launch_cost(t128, slots, 128)));
  if ((variant & 2048) && !bm96) return VCR_EUNSUPPORTED;
  int bm = bm96 ? 96 : BM;
  if (variant & (8192 | 16384)) {
    if (!pl->glds) return VCR_EUNSUPPORTED;
    bm = (variant & 8192) ? 64 : 32;
  } else if (bm_override && pl->bm_free && (bm_override >= 96 || pl->small_free)) {
    bm = bm_override;
  } else if (!bm_override && pl->small_free && t128 < 2L * vcr_cu_count()) {
    const long th[4] = {t128, t96, (long)((a->M + 63) / 64) * pl->tiles_n, (long)((a->M + 31) / 32) * pl->tiles_n};
    bm = small_grid_rows(th);
    if (bm >= 96) bm = bm96 ? 96 : 128;                  // (>= 96 rows: the regular choice above, with its k-slab and shape)
  }
  if (bm < 96) pl->bk16 = false;                         // 64 / 32 rows: the small-grid configuration
  if (bm < 128) pl->ms16 = true;
  pl->bm = bm;
  pl->tiles_m = (a->M + bm - 1) / bm;
  static_assert(2 * sizeof(TileGT<16>) == 2 * (BM + BN) * 16 * 4 && 2 * sizeof(TileGT<32, 96>) == 2 * (96 + BN) * 32 * 4, "stage size below");
  const int bkv = pl->bk16 ? 16 : 32;
  const int stage = 2 * (bm + BN) * bkv * 4 > 4 * 32 * 68 * 4 ? 2 * (bm + BN) * bkv * 4 : 4 * 32 * 68 * 4;   // 2 x TileGT<BK, BMV> or the epilogue slices
  pl->lds = stage + (pl->ln_in ? bm * 2 * 4 : 0);
  return VCR_OK;
}

// dispatch over the template grid (BK, MS, BMV): F is a generic lambda taking three integral_constants
template <class F>
void linear_dispatch(const LinearPlan& pl, F&& f) {
  using I16 = std::integral_constant<int, 16>;
  using I32 = std::integral_constant<int, 32>;
  using I96 = std::integral_constant<int, 96>;
  using I128 = std::integral_constant<int, 128>;
  using I64 = std::integral_constant<int, 64>;
  if (pl.bm == 64) f(I32{}, I16{}, I64{});
  else if (pl.bm == 32) f(I32{}, I16{}, I32{});
  else if (pl.bm == 96) { if (pl.bk16) f(I16{}, I16{}, I96{}); else f(I32{}, I16{}, I96{}); }
  else if (pl.bk16) { if (pl.ms16) f(I16{}, I16{}, I128{}); else f(I16{}, I32{}, I128{}); }
  else { if (pl.ms16) f(I32{}, I16{}, I128{}); else f(I32{}, I32{}, I128{}); }
}
}  // namespace

// Host-only: the kernel configuration vcr_linear_f32 would launch for these arguments, nothing launched.  Returns
// rows | k-slab << 8 | (16x16x4 ? 1 << 16 : 0) | (LDS-DMA ? 1 << 17 : 0), or a negative VCR_E* code.
extern "C" int vcr_linear_config(const vcr_linear_args* a) {
  LinearPlan pl{};
  const int rc = linear_plan(a, &pl);
  if (rc != VCR_OK) return rc;
  if (!pl.glds) return 128 | (32 << 8);
  return pl.bm | ((pl.bk16 ? 16 : 32) << 8) | (pl.ms16 ? 1 << 16 : 0) | (1 << 17);
}

extern "C" int vcr_linear_f32(const vcr_linear_args* a, vcr_stream_t stream) {
  vcr_stream_scope bound(stream);
  LinearPlan pl{};
  const int rc = linear_plan(a, &pl);
  if (rc != VCR_OK) return rc;
  if (pl.glds) {
    linear_dispatch(pl, [&](auto bk, auto ms, auto bm) {
      constexpr int BKV = decltype(bk)::value, MSV = decltype(ms)::value, BMV = decltype(bm)::value;
      VCR_DYN_LDS((linear_glds_kernel<BKV, MSV, BMV>), pl.lds);
      hipLaunchKernelGGL((linear_glds_kernel<BKV, MSV, BMV>), dim3(pl.tiles_m * pl.tiles_n), dim3(256), pl.lds,
                         (hipStream_t)stream, *a, pl.tiles_m, pl.tiles_n);
    });
  } else {
    // alignment-free fallback (odd N, unaligned y / bias / residual; bit 2 forces it): register staging, scalar epilogue
    const int lds32 = 2 * sizeof(TileT<32>);
    VCR_DYN_LDS(linear_kernel<32>, lds32);
    hipLaunchKernelGGL(linear_kernel<32>, dim3(pl.tiles_m * pl.tiles_n), dim3(256), lds32, (hipStream_t)stream, *a, pl.tiles_m,
                       pl.tiles_n, pl.vec);
  }
  return VCR_LAUNCH_RC();
}

// Two independent linears as one launch when they resolve to the same LDS-DMA kernel configuration (k-slab, MFMA shape,
// tile rows; neither with a fused max); otherwise exactly the two vcr_linear_f32 calls.  Same results.
// the planning half of vcr_linear_pair_f32: both plans as a paired launch would take them, and whether it IS one launch
static int linear_pair_plan(const vcr_linear_args* a, const vcr_linear_args* b, LinearPlan& pa, LinearPlan& pb, bool& same) {
  int rc = linear_plan(a, &pa, 0, true);
  if (rc == VCR_OK) rc = linear_plan(b, &pb, 0, true);
  if (rc != VCR_OK) return rc;
  int joint = 0;                                         // tile rows from the COMBINED grid (the two halves share the rounds)
  if (pa.small_free && pb.small_free && pa.t128 + pb.t128 < 2L * vcr_cu_count()) {     // a small grid even together
    const long th[4] = {pa.t128 + pb.t128, pa.t96 + pb.t96,
                        (long)((a->M + 63) / 64) * pa.tiles_n + (long)((b->M + 63) / 64) * pb.tiles_n,
                        (long)((a->M + 31) / 32) * pa.tiles_n + (long)((b->M + 31) / 32) * pb.tiles_n};
    joint = small_grid_rows(th);
  }
  if (joint >= 96 || (joint == 0 && pa.bm_free && pb.bm_free)) {
    const bool r = a->residual != nullptr, rb = b->residual != nullptr;     // (the regular k-slab: BK 32 iff a residual)
    joint = 0;
    if (r == rb) {
      const int slots = vcr_cu_count() * (r ? 2 : 4);
      joint = r && 1.02 * launch_cost(pa.t96 + pb.t96, slots, 96) < launch_cost(pa.t128 + pb.t128, slots, 128) ? 96 : 128;
    }
  }
  if (joint) {
    linear_plan(a, &pa, joint, true);
    linear_plan(b, &pb, joint, true);
  }
  same = pa.glds && pb.glds && pa.bk16 == pb.bk16 && pa.ms16 == pb.ms16 && pa.bm == pb.bm &&
         !a->segmax_out && !b->segmax_out;             // (LayerNorm-in / statistics-out may differ: run-time flags of each half)
  return VCR_OK;
}

// Host-only, library-internal (forward.hip): the MFMA shape -- 1 = 16x16x4, 0 = 32x32x2, the one choice of a linear's
// configuration that its BITS depend on -- that vcr_linear_f32 (b == NULL) or vcr_linear_pair_f32 would compute these arguments
// with.  The forward's src-only launches of a later vcrnetIter pass pin the shape of the full-row launch they stand for.
extern "C" int vcr_linear_shapes_(const vcr_linear_args* a, const vcr_linear_args* b, int* shape_a, int* shape_b) {
  LinearPlan pa{}, pb{};
  if (!b) {
    const int rc = linear_plan(a, &pa);
    if (rc == VCR_OK && shape_a) *shape_a = pa.glds ? (pa.ms16 ? 1 : 0) : 0;
    return rc;
  }
  bool same = false;
  int rc = linear_pair_plan(a, b, pa, pb, same);
  if (rc != VCR_OK) return rc;
  if (!same) {                                           // two separate launches, each planned on its own
    rc = linear_plan(a, &pa);
    if (rc == VCR_OK) rc = linear_plan(b, &pb);
    if (rc != VCR_OK) return rc;
  }
  if (shape_a) *shape_a = pa.glds ? (pa.ms16 ? 1 : 0) : 0;
  if (shape_b) *shape_b = pb.glds ? (pb.ms16 ? 1 : 0) : 0;
  return VCR_OK;
}

extern "C" int vcr_linear_pair_f32(const vcr_linear_args* a, const vcr_linear_args* b, vcr_stream_t stream) {
  vcr_stream_scope bound(stream);
  LinearPlan pa{}, pb{};
  bool same = false;
  int rc = linear_pair_plan(a, b, pa, pb, same);
  if (rc != VCR_OK) return rc;
  if (pb.lds > pa.lds) pa.lds = pb.lds;
  if (!same) {
    rc = vcr_linear_f32(a, stream);
    return rc ? rc : vcr_linear_f32(b, stream);
  }
  linear_dispatch(pa, [&](auto bk, auto ms, auto bm) {
    constexpr int BKV = decltype(bk)::value, MSV = decltype(ms)::value, BMV = decltype(bm)::value;
    VCR_DYN_LDS((linear_glds_pair_kernel<BKV, MSV, BMV>), pa.lds);
    hipLaunchKernelGGL((linear_glds_pair_kernel<BKV, MSV, BMV>), dim3(pa.tiles_m * pa.tiles_n + pb.tiles_m * pb.tiles_n),
                       dim3(256), pa.lds, (hipStream_t)stream, *a, *b, pa.tiles_m, pa.tiles_n, pb.tiles_m, pb.tiles_n);
  });
  return VCR_LAUNCH_RC();
}
