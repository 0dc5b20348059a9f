// RECORD (not built): the bf16x3 linear as it stood until round 4 (256 x 128 tile, 8 waves, one workgroup per CU, 32x32x16 MFMAs,
// LDS-DMA weight planes).  Replaced by vcr-net_amd/csrc/linear_bf16x3.hip (128 x 128 tile, 16x16x32 MFMAs, two workgroups per CU);
// profiles/rounds4-5/r4t_* are the measurements that led there, profiles/rounds4-5/r4w_* the comparison.

// Pointwise linear on the bf16 matrix pipe with fp32-equivalent products ("bf16x3"):
//   every fp32 operand is split EXACTLY into three bf16 pieces, x = x1 + x2 + x3 (8 + 8 + 8 significand bits),
//   and a.b is evaluated as the six partial products whose weight is >= 2^-16 of the leading one,
//       a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a3 b1 + a2 b2),
//   each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped terms are
//   <= 2^-24 relative, i.e. the result carries fp32-GEMM accuracy, while the six MFMAs cost 6 x 32 cycles
//   for a 32x32x16 block against 8 x 64 cycles of v_mfma_f32_32x32x2_f32: 2.67x the fp32 matrix rate.
//   Same interface, epilogue and output layout as linear.hip -- including the folded LayerNorm (ln_stats_in: the
//   main loop runs on the folded weight, the epilogue applies the per-row mean / 1/(std+eps)) and the statistics
//   epilogue (stats_out) -- weights are pre-split once (vcr_split_bf16x3_f32), activations are split on the fly
//   while they are staged into LDS.
//
// 256 x 128 x 32 block tile, 8 waves (4 x 2), wave tile 64 x 64, one block per CU (144 KB LDS, double buffered).
// LDS image per operand: three planes [rows][32 bf16] (64-B rows, no padding) with the 16-B chunk index
// XOR-swizzled by ((row >> 2) & 3): a 16-lane ds_read_b128 group (16 consecutive rows) then touches 16
// distinct slots.  B planes arrive by LDS-DMA (swizzle applied to the global source address), A is
// register-staged because it has to pass through the VALU split.
#include "common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 256, TN = 128, TK = 32;

struct Stage3 {
  short a[3][TM][TK];   // 3 x 16 KB
  short b[3][TN][TK];   // 3 x  8 KB
};

// fp32 -> bf16 round-to-nearest-even.  A plain cast compiles to v_cvt_pk_bf16_f32 on gfx950 (two elements per
// instruction); widening back is a 16-bit shift.  Both subtractions below are exact in fp32.
__device__ __forceinline__ unsigned short bf16_rn(float x) {
  const __bf16 b = (__bf16)x;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf16_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  h = bf16_rn(x);
  const float r1 = x - bf16_f32(h);       // exact
  m = bf16_rn(r1);
  const float r2 = r1 - bf16_f32(m);      // exact
  l = bf16_rn(r2);
}

__device__ __forceinline__ void glds16b(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(512, 2) void linear_bf16x3_kernel(vcr_linear_args p, const short* wsplit, int tiles_m,
                                                               int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stage3* st = reinterpret_cast<Stage3*>(smem);          // [2]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  bid = xcd_chunk(bid, nblk);
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;
  const size_t plane = (size_t)p.N * p.K;                 // elements per weight plane

  // A staging: thread owns rows (t >> 3) + 64 i, float4 group c = t & 7 (k = 4c .. 4c+3)
  const int ar0 = t >> 3, ac = t & 7;
  const float* xa[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) xa[i] = p.x + (size_t)min(m0 + ar0 + 64 * i, p.M - 1) * p.ldx + 4 * ac;
  // B fill by LDS-DMA: one wave-instruction = 16 rows x 64 B of one plane; wave w covers rows 16 w .. 16 w + 15
  const int brow = wave * 16 + (lane >> 2), bpc = lane & 3;
  const int blc = bpc ^ ((brow >> 2) & 3);
  const short* wb = wsplit + (size_t)min(n0 + brow, p.N - 1) * p.K + 8 * blc;

  f32x4 ra[4];
  auto load_a = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = ld4(xa[i] + k0);
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = ar0 + 64 * i;
      unsigned short h[4], m[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split3(ra[i][e], h[e], m[e], l[e]);
      const int off = ((ac >> 1) ^ ((row >> 2) & 3)) * 8 + (ac & 1) * 4;   // in bf16 elements within the 32-wide row
      *reinterpret_cast<s16x4*>(&st[buf].a[0][row][off]) = s16x4{(short)h[0], (short)h[1], (short)h[2], (short)h[3]};
      *reinterpret_cast<s16x4*>(&st[buf].a[1][row][off]) = s16x4{(short)m[0], (short)m[1], (short)m[2], (short)m[3]};
      *reinterpret_cast<s16x4*>(&st[buf].a[2][row][off]) = s16x4{(short)l[0], (short)l[1], (short)l[2], (short)l[3]};
    }
  };
  auto fill_b = [&](int buf, int k0) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) glds16b(wb + pl * plane + k0, &st[buf].b[pl][wave * 16][0]);
  };

  //@probe VCR_PROBE_STAMP(0);
  load_a(0);
  fill_b(0, 0);
  store_a(0);
  float* rowst = reinterpret_cast<float*>(smem + 2 * sizeof(Stage3));   // [TM][2] (mean, 1/(std+eps)) of this block's rows
  if (p.ln_stats_in && t < TM) {
    float mean, var;
    ln_row_moments(p.ln_stats_in + (size_t)min(m0 + t, p.M - 1) * p.ln_nseg * 2, p.ln_nseg, p.K, mean, var);
    rowst[2 * t] = mean;
    rowst[2 * t + 1] = 1.f / (sqrtf(var) + p.ln_eps);
  }
  __syncthreads();
  //@probe VCR_PROBE_STAMP(1);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0};
  int arow[2], brow_[2], asw[2], bsw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    arow[i] = wm * 64 + i * 32 + l31; asw[i] = (arow[i] >> 2) & 3;
    brow_[i] = wn * 64 + i * 32 + l31; bsw[i] = (brow_[i] >> 2) & 3;
  }

  // One slab = 2 k-steps of 16 x (2 x 2 output tiles) x 6 MFMAs.  The stream is laid out by hand (sched_barrier fences one
  // CHUNK = one output tile's six MFMAs = 192 cycles of the matrix pipe): beside a wave that issues MFMAs back to back a
  // SIMD lets the other wave's vector / LDS instructions through at one per 20-36 cycles, while a wave's own instructions
  // issue in the shadow of its own MFMAs (profiles/rounds4-5/r4f_mfma_valu_coissue.txt) -- and a slab carries ~150 of them per wave
  // (24 fragment reads, the 3-way split of 16 activations = ~110 VALU, 12 LDS stores, the next slab's requests) against 48
  // MFMAs.  hipcc grouped them in front of and behind the MFMA block, where both waves of a SIMD (one workgroup per CU: they
  // run in phase) crawled through them together.  Here k-step 0's chunks carry the fragment reads of k-step 1 and the
  // next slab's weight-plane requests, k-step 1's chunks the split + LDS stores of the next slab's activations, one
  // quarter each.  Same MFMA order per output element: bit-identical results.
  const int nk = p.K / TK;
  auto frag = [&](const Stage3& S, int sstep, bf16x8 (&fa)[2][3], bf16x8 (&fb)[2][3], int i) {   // row block i of both operands
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      fa[i][pl] = *reinterpret_cast<const bf16x8*>(&S.a[pl][arow[i]][((2 * sstep + half) ^ asw[i]) * 8]);
      fb[i][pl] = *reinterpret_cast<const bf16x8*>(&S.b[pl][brow_[i]][((2 * sstep + half) ^ bsw[i]) * 8]);
    }
  };
  auto store_a_row = [&](int buf, int i) {               // split + store the staged activations of row block i
    const int row = ar0 + 64 * i;
    unsigned short h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split3(ra[i][e], h[e], m[e], l[e]);
    const int off = ((ac >> 1) ^ ((row >> 2) & 3)) * 8 + (ac & 1) * 4;
    *reinterpret_cast<s16x4*>(&st[buf].a[0][row][off]) = s16x4{(short)h[0], (short)h[1], (short)h[2], (short)h[3]};
    *reinterpret_cast<s16x4*>(&st[buf].a[1][row][off]) = s16x4{(short)m[0], (short)m[1], (short)m[2], (short)m[3]};
    *reinterpret_cast<s16x4*>(&st[buf].a[2][row][off]) = s16x4{(short)l[0], (short)l[1], (short)l[2], (short)l[3]};
  };
  auto tile6 = [&](const bf16x8 (&fa)[2][3], const bf16x8 (&fb)[2][3], int i, int j) {
    f32x16 c = acc[i][j];
    c = mfma_bf16(fa[i][1], fb[j][1], c);               // smallest terms first
    c = mfma_bf16(fa[i][0], fb[j][2], c);
    c = mfma_bf16(fa[i][2], fb[j][0], c);
    c = mfma_bf16(fa[i][0], fb[j][1], c);
    c = mfma_bf16(fa[i][1], fb[j][0], c);
    c = mfma_bf16(fa[i][0], fb[j][0], c);
    acc[i][j] = c;
  };
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) load_a((kt + 1) * TK);
    const Stage3& S = st[cur];
    bf16x8 fa0[2][3], fb0[2][3], fa1[2][3], fb1[2][3];
    frag(S, 0, fa0, fb0, 0);
    frag(S, 0, fa0, fb0, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {                         // k-step 0: output tile (c >> 1, c & 1)
      if (c < 2) frag(S, 1, fa1, fb1, c);                 // (six 16-B reads in each of the first two chunks)
      tile6(fa0, fb0, c >> 1, c & 1);
      if (more && c < 3) glds16b(wb + c * plane + (kt + 1) * TK, &st[cur ^ 1].b[c][wave * 16][0]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {                         // k-step 1
      tile6(fa1, fb1, c >> 1, c & 1);
      if (more) store_a_row(cur ^ 1, c);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  //@probe VCR_PROBE_STAMP(2);
  // epilogue: identical to linear.hip (the 32x32 accumulator layout does not depend on the input dtype)
  constexpr int EP = 68;
  float* ot = reinterpret_cast<float*>(smem) + wave * 32 * EP;
  const int c4e = (lane & 15) * 4, col = n0 + wn * 64 + c4e;
  const f32x4 bias = (p.bias && col < p.N) ? ld4(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  const bool ln_in = p.ln_stats_in != nullptr;           // block-uniform
  const f32x4 csum = (ln_in && col < p.N) ? ld4(p.ln_colsum + col) : f32x4{0.f, 0.f, 0.f, 0.f};
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
          f32x4 v = ld4(&ot[rl * EP + c4e]);
          if (ln_in) {
            const float mean = rowst[2 * (wm * 64 + i * 32 + rl)], inv = rowst[2 * (wm * 64 + i * 32 + rl) + 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(inv, fmaf(-mean, csum[e], v[e]), bias[e]);
          } else {
            v = v + bias;
          }
          if (p.relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
          if (p.residual) v = v + ld4(p.residual + (size_t)row * p.ldr + col);
          st4(p.y + (size_t)row * p.ldy + col, v);
          if (p.stats_out) {                             // the 16 lanes of a row group hold this wave's 64 columns of the row
            float s1 = (v[0] + v[1]) + (v[2] + v[3]);
            s1 = row16_sum(s1);
            const float ms = s1 * (1.f / 64.f);          // (sum, second moment about the segment mean): linear.hip
            const float d0 = v[0] - ms, d1 = v[1] - ms, d2 = v[2] - ms, d3 = v[3] - ms;
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
  }
  //@probe __builtin_amdgcn_s_waitcnt(0); VCR_PROBE_STAMP(3);     // (stores acknowledged)
}

__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* x, short* out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned short h, m, l;
  split3(x[i], h, m, l);
  out[i] = (short)h; out[n + i] = (short)m; out[2 * n + i] = (short)l;
}

}  // namespace

extern "C" int vcr_split_bf16x3_f32(const float* x, void* planes, size_t n, vcr_stream_t stream) {
  if (!x || !planes || n == 0) return VCR_EINVAL;
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     reinterpret_cast<short*>(planes), n);
  return VCR_LAUNCH_RC();
}

extern "C" int vcr_linear_bf16x3_f32(const vcr_linear_args* a, const void* w_planes, vcr_stream_t stream) {
  if (!a || !a->x || !w_planes || !a->y) return VCR_EINVAL;
  if (a->ln_stats_in && (!a->ln_colsum || !a->bias || a->ln_nseg <= 0 || a->K < 2 || (a->K % a->ln_nseg) ||
                         ((uintptr_t)a->ln_colsum & 15)))
    return VCR_EINVAL;                                   // (K % ln_nseg: ln_row_moments needs equal segments, as linear_plan checks)
  if (a->stats_out && (a->N % 64)) return VCR_EINVAL;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || (a->K % TK) != 0) return VCR_EINVAL;
  if ((a->ldx & 3) || a->ldx < a->K || a->ldy < a->N || (a->residual && a->ldr < a->N)) return VCR_EINVAL;
  if ((a->N % 4) || (a->ldy % 4) || ((uintptr_t)a->y & 15) || ((uintptr_t)a->x & 15) || ((uintptr_t)w_planes & 15))
    return VCR_EINVAL;
  if ((a->bias && ((uintptr_t)a->bias & 15)) || (a->residual && ((a->ldr % 4) || ((uintptr_t)a->residual & 15))))
    return VCR_EINVAL;
  const int tiles_m = (a->M + TM - 1) / TM, tiles_n = (a->N + TN - 1) / TN;
  const int lds = 2 * sizeof(Stage3) + TM * 2 * sizeof(float);
  static_assert(2 * sizeof(Stage3) >= 8 * 32 * 68 * 4, "epilogue slices fit");
  VCR_DYN_LDS(linear_bf16x3_kernel, lds);
  hipLaunchKernelGGL(linear_bf16x3_kernel, dim3(tiles_m * tiles_n), dim3(512), lds, (hipStream_t)stream, *a,
                     reinterpret_cast<const short*>(w_planes), tiles_m, tiles_n);
  return VCR_LAUNCH_RC();
}
