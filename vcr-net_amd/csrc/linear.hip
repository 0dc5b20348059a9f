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

namespace {

constexpr int BM = 128, BN = 128;

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
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, i = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
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
struct TileG { float a[BM][32]; float b[BN][32]; };

// Sum over the 16 lanes of a DPP row with VALU-rate DPP moves (quad_perm xor 1, xor 2, then row_ror 4 and 8);
// every lane ends with the full sum, in a fixed order.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
  return v;
}

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
template <bool LN_IN, bool STATS_OUT>
__global__ __launch_bounds__(256, 2) void linear_glds_kernel(vcr_linear_args p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  TileG* tile = reinterpret_cast<TileG*>(smem);          // [2]
  float* rowst = reinterpret_cast<float*>(smem + 2 * sizeof(TileG));   // [BM][2] (mean, inv) per row (LN_IN only)
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);   // scalar: LDS-DMA targets via SALU
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, i = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // fill mapping: wave w covers rows w*32 + 8i + (lane>>3), physical chunk lane&7
  const int frow = lane >> 3, fpc = lane & 7;
  const float* xa[4];
  const float* wb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + 8 * i + frow;
    const int lc = fpc ^ ((row >> 1) & 7);
    xa[i] = p.x + (size_t)min(m0 + row, p.M - 1) * p.ldx + 4 * lc;
    wb[i] = p.w + (size_t)min(n0 + row, p.N - 1) * p.K + 4 * lc;
  }
  auto fill = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(xa[i] + k0, &tile[buf].a[wave * 32 + 8 * i][0]);
      glds16(wb[i] + k0, &tile[buf].b[wave * 32 + 8 * i][0]);
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

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0};
  int ra_[2], rb_[2], sa[2], sb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ra_[i] = wm * 64 + i * 32 + l31; sa[i] = (ra_[i] >> 1) & 7;
    rb_[i] = wn * 64 + i * 32 + l31; sb[i] = (rb_[i] >> 1) & 7;
  }
  // The residual tile does not depend on the GEMM: fetch this lane's 16 chunks now, so that the epilogue's
  // load -> add -> store chain does not start with an HBM round trip (64 VGPRs; the kernel runs 2 waves per SIMD).
  f32x4 resv[2][8];
  {
    const int colr = n0 + wn * 64 + (lane & 15) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        const int row = m0 + wm * 64 + i * 32 + ps * 4 + (lane >> 4);
        resv[i][ps] = (p.residual && row < p.M && colr < p.N) ? ld4(p.residual + (size_t)row * p.ldr + colr)
                                                              : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  }
  const int nk = p.K / 32;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) fill(cur ^ 1, (kt + 1) * 32);
    const TileG& T = tile[cur];
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
    __syncthreads();                                     // drains the LDS-DMA (vmcnt(0)) and orders the buffers
  }

  constexpr int EP = 68;
  float* ot = reinterpret_cast<float*>(smem) + wave * 32 * EP;
  const int c4e = (lane & 15) * 4, col = n0 + wn * 64 + c4e;
  const f32x4 bias = (p.bias && col < p.N) ? ld4(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 csum = (LN_IN && col < p.N) ? ld4(p.ln_colsum + col) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[acc_row(r, half) * EP + j * 32 + l31] = acc[i][j][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (col < p.N) {
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
          if (p.residual) v = v + resv[i][ps];
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
}

// ---- BK = 16 variant of the LDS-DMA kernel for launches WITHOUT a residual: half-size stages (64-B rows, swizzle
// pc = lc ^ ((row >> 2) & 3)), 35 KB of LDS per workgroup and ~100 VGPRs, so FOUR workgroups share a CU (4 waves
// per SIMD from independent workgroups) and the k-step barrier of one is covered by the other three.  Same k order
// as the BK = 32 kernel: results are bit-identical.
struct TileG16 { float a[BM][16]; float b[BN][16]; };

template <bool LN_IN, bool STATS_OUT>
__global__ __launch_bounds__(256, 4) void linear_glds16_kernel(vcr_linear_args p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  TileG16* tile = reinterpret_cast<TileG16*>(smem);      // [2]
  float* rowst = reinterpret_cast<float*>(smem + 4 * 32 * 68 * 4);     // [BM][2] (mean, inv) per row (LN_IN only)
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);   // scalar: LDS-DMA targets via SALU
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, i = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // fill mapping: wave w covers rows w*32 + 16i + (lane>>2), physical chunk lane&3 (64-B rows, 16 rows per KiB)
  const int frow = lane >> 2, fpc = lane & 3;
  const float* xa[2];
  const float* wb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wave * 32 + 16 * i + frow;
    const int lc = fpc ^ ((row >> 2) & 3);
    xa[i] = p.x + (size_t)min(m0 + row, p.M - 1) * p.ldx + 4 * lc;
    wb[i] = p.w + (size_t)min(n0 + row, p.N - 1) * p.K + 4 * lc;
  }
  auto fill = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      glds16(xa[i] + k0, &tile[buf].a[wave * 32 + 16 * i][0]);
      glds16(wb[i] + k0, &tile[buf].b[wave * 32 + 16 * i][0]);
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

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0};
  int ra_[2], rb_[2], sa[2], sb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ra_[i] = wm * 64 + i * 32 + l31; sa[i] = (ra_[i] >> 2) & 3;
    rb_[i] = wn * 64 + i * 32 + l31; sb[i] = (rb_[i] >> 2) & 3;
  }
  const int nk = p.K / 16;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) fill(cur ^ 1, (kt + 1) * 16);
    const TileG16& T = tile[cur];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
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

  constexpr int EP = 68;
  float* ot = reinterpret_cast<float*>(smem) + wave * 32 * EP;
  const int c4e = (lane & 15) * 4, col = n0 + wn * 64 + c4e;
  const f32x4 bias = (p.bias && col < p.N) ? ld4(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 csum = (LN_IN && col < p.N) ? ld4(p.ln_colsum + col) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[acc_row(r, half) * EP + j * 32 + l31] = acc[i][j][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (col < p.N) {
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
}

}  // namespace

extern "C" int vcr_linear_f32(const vcr_linear_args* a, vcr_stream_t stream) {
  if (!a || !a->x || !a->w || !a->y) return VCR_EINVAL;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || (a->K % 32) != 0) return VCR_EINVAL;
  const int variant = a->variant;                    // tuning / test selector carried by the call (see vcr_hip.h)
  if ((a->ldx & 3) || a->ldx < a->K || a->ldy < a->N || (a->residual && a->ldr < a->N)) return VCR_EINVAL;
  if (((uintptr_t)a->x | (uintptr_t)a->w) & 15) return VCR_EINVAL;
  const int tiles_m = (a->M + BM - 1) / BM, tiles_n = (a->N + BN - 1) / BN;
  const int lds32 = 2 * sizeof(TileT<32>), lds16 = 2 * sizeof(TileT<16>);
  static_assert(2 * sizeof(TileT<16>) >= 4 * 32 * 68 * 4, "epilogue slice fits the staging buffers");
  const int vec = (a->N % 4 == 0) && (a->ldy % 4 == 0) && (((uintptr_t)a->y & 15) == 0) &&
                  (!a->bias || ((uintptr_t)a->bias & 15) == 0) &&
                  (!a->residual || ((a->ldr % 4 == 0) && ((uintptr_t)a->residual & 15) == 0));
  if (a->ln_stats_in || a->stats_out) {                  // fused LayerNorm prologue / statistics epilogue
    if (!vec || (variant & 4)) return VCR_EUNSUPPORTED;
    if (a->ln_stats_in && (!a->ln_colsum || !a->bias || a->ln_nseg <= 0 || a->K < 2 ||
                           ((uintptr_t)a->ln_colsum & 15))) return VCR_EINVAL;
    if (a->stats_out && (a->N % 64)) return VCR_EINVAL;
  }
  if (!(variant & 4) && vec) {   // default: LDS-DMA staging (bit2 of the debug variant selects register staging)
    const bool ln_in = a->ln_stats_in != nullptr, st_out = a->stats_out != nullptr;
    const int ldsg = 2 * sizeof(TileG) + (ln_in ? BM * 2 * 4 : 0);
#define VCR_LIN_LAUNCH(LI, SO)                                                                                          \
  do {                                                                                                                   \
    VCR_DYN_LDS((linear_glds_kernel<LI, SO>), ldsg);                                                                     \
    hipLaunchKernelGGL((linear_glds_kernel<LI, SO>), dim3(tiles_m * tiles_n), dim3(256), ldsg, (hipStream_t)stream, *a, \
                       tiles_m, tiles_n);                                                                                \
  } while (0)
#define VCR_LIN16_LAUNCH(LI, SO)                                                                                        \
  do {                                                                                                                   \
    const int lds16g = 4 * 32 * 68 * 4 + (ln_in ? BM * 2 * 4 : 0);                                                       \
    VCR_DYN_LDS((linear_glds16_kernel<LI, SO>), lds16g);                                                                 \
    hipLaunchKernelGGL((linear_glds16_kernel<LI, SO>), dim3(tiles_m * tiles_n), dim3(256), lds16g, (hipStream_t)stream,  \
                       *a, tiles_m, tiles_n);                                                                            \
  } while (0)
    // Without a residual: BK = 16, four workgroups per CU (measured +2-3 % on the qkv / ffn1 / kv projections).
    // With one: BK = 32 and the residual tile prefetched across the GEMM loop (bit 3 of the debug variant forces BK 32).
    if (!a->residual && !(variant & 8)) {
      if (ln_in && st_out) VCR_LIN16_LAUNCH(true, true);
      else if (ln_in) VCR_LIN16_LAUNCH(true, false);
      else if (st_out) VCR_LIN16_LAUNCH(false, true);
      else VCR_LIN16_LAUNCH(false, false);
    } else if (ln_in && st_out) VCR_LIN_LAUNCH(true, true);
    else if (ln_in) VCR_LIN_LAUNCH(true, false);
    else if (st_out) VCR_LIN_LAUNCH(false, true);
    else VCR_LIN_LAUNCH(false, false);
#undef VCR_LIN16_LAUNCH
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
