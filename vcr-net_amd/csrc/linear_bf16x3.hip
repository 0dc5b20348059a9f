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
// 128 x 128 x 32 block tile, 4 waves (2 x 2), wave tile 64 x 64 = 4 x 4 tiles of v_mfma_f32_16x16x32_bf16, TWO workgroups
// per CU (49 KB LDS and <= 256 registers each).  Round 4 measured what held the first kernel (256 x 128 tile, 8 waves, one
// workgroup per CU, 32x32x16 MFMAs; profiles/experiments/linear_bf16x3_round4a.hip) at 0.37 of 417 TFLOP/s-equivalent
// (profiles/rounds4-5/r4t_*): (1) the bf16 matrix pipe is POWER limited -- the shader clock sits at 1.8 GHz inside the k loop (2.2-2.4
// outside), and the 16x16x32 form delivers 1.14x the 32x32x16 form in the same MFMA-only loop; (2) with one workgroup per
// CU nothing covers a tile's prologue and epilogue (5-10 us of 45-50); (3) the activation path (loads, 3-way split, LDS
// stores) costs 15 % of the loop where both waves of a SIMD run it in phase.  Here a slab is ONE k-step: after barrier 1 a
// wave requests its 24 fragments (12 of A, 12 of B) and starts its 96 MFMAs as they arrive; the first half of them carries
// the 3-way split of the next slab's activations (registers only), barrier 2 in the middle -- every wave has long had its
// fragments -- frees the single-buffered LDS image, and the second half carries the LDS stores of the next slab (activations
// and weight planes, both staged in registers one slab ahead).  The other workgroup of the CU, out of phase, has the
// matrix pipe while this one reads, waits at a barrier or runs its epilogue.
// LDS image per operand: three planes [rows][32 bf16] (64-B rows, no padding) with the 16-B chunk index XOR-swizzled by
// swz(row) = {0, 2, 3, 1}[(row >> 2) & 3].  A ds_read_b128 is served in four NON-contiguous 16-lane groups ({0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31} and the same + 32: MI355X_MICROARCH.md, LDS): with the 16x16x32 fragment (row = lane & 15, chunk =
// lane >> 4) a group reads rows 0-3 and 12-15 of one chunk and rows 4-11 of the next, and this table puts those on 16
// distinct 16-B slots of the 256-B bank row (the plain (row >> 2) & 3 is 2-way there: SQ_LDS_BANK_CONFLICT was a third of the
// LDS cycles, profiles/rounds4-5/r4u_pmc_bx3_3.txt).  The stores (16 / 8 contiguous lanes = 128 contiguous bytes) are conflict-free.
#include <type_traits>

#include "common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 128, TN = 128, TK = 32;

struct Stage3 {
  short a[3][TM][TK];   // 3 x 8 KB
  short b[3][TN][TK];   // 3 x 8 KB
};

// fp32 -> bf16 round-to-nearest-even.  A plain cast compiles to v_cvt_pk_bf16_f32 on gfx950 (two elements per
// instruction); widening back is a 16-bit shift.  Both subtractions below are exact in fp32.
__device__ __forceinline__ unsigned short bf16_rn(float x) {
  const __bf16 b = (__bf16)x;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf16_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

__device__ __forceinline__ int swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  h = bf16_rn(x);
  const float r1 = x - bf16_f32(h);       // exact
  m = bf16_rn(r1);
  const float r2 = r1 - bf16_f32(m);      // exact
  l = bf16_rn(r2);
}

// D[4 (lane >> 4) + r][lane & 15] += sum_k A[lane & 15][8 (lane >> 4) + k] B[8 (lane >> 4) + k][lane & 15]
__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <bool RES>   // RES: the launch has a residual operand (its tile is requested during the last slab's MFMAs)
__global__ __launch_bounds__(256, 2) void linear_bf16x3_kernel(vcr_linear_args p, const short* wsplit, int tiles_m,
                                                               int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stage3& S = *reinterpret_cast<Stage3*>(smem);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  bid = xcd_chunk(bid, nblk);
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;
  const size_t plane = (size_t)p.N * p.K;                 // elements per weight plane

  //@probe VCR_PROBE_STAMP(0);
  // A staging: thread owns rows (t >> 3) + 32 i, float4 group ac = t & 7 (k = 4 ac .. 4 ac + 3)
  const int ar0 = t >> 3, ac = t & 7;
  const float* xa[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) xa[i] = p.x + (size_t)min(m0 + ar0 + 32 * i, p.M - 1) * p.ldx + 4 * ac;
  // B staging: 16-B chunk u of a thread = plane (u >> 1), row (t >> 2) + 64 (u & 1), chunk bc = t & 3 of the 64-B row
  const int br0 = t >> 2, bc = t & 3;
  const short* wb[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) wb[h] = wsplit + (size_t)min(n0 + br0 + 64 * h, p.N - 1) * p.K + 8 * bc;

  f32x4 ra[4];
  bf16x8 rb[6];
  auto load_a = [&](int k0, int i) { ra[i] = ld4(xa[i] + k0); };
  auto load_b = [&](int k0, int u) { rb[u] = *reinterpret_cast<const bf16x8*>(wb[u & 1] + (u >> 1) * plane + k0); };
  // split of one staged float4 in two halves (elements 2 e2, 2 e2 + 1 -> one packed dword per plane), so that the ~8 vector
  // instructions of a half fit the shadow of one chunk's MFMAs; the row goes to LDS once both halves are done
  unsigned pk[4][3][2];
  auto split_half = [&](int i, int e2) {
    unsigned short h[2], m[2], l[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) split3(ra[i][2 * e2 + e], h[e], m[e], l[e]);
    pk[i][0][e2] = (unsigned)h[0] | ((unsigned)h[1] << 16);
    pk[i][1][e2] = (unsigned)m[0] | ((unsigned)m[1] << 16);
    pk[i][2][e2] = (unsigned)l[0] | ((unsigned)l[1] << 16);
  };
  auto store_a = [&](int i) {
    const int row = ar0 + 32 * i;
    const int off = ((ac >> 1) ^ swz(row)) * 8 + (ac & 1) * 4;   // in bf16 elements within the 32-wide row
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(&S.a[pl][row][off]) = uint2{pk[i][pl][0], pk[i][pl][1]};
  };
  auto store_b = [&](int u) {
    const int row = br0 + 64 * (u & 1);
    *reinterpret_cast<bf16x8*>(&S.b[u >> 1][row][(bc ^ swz(row)) * 8]) = rb[u];
  };

  const int nk = p.K / TK;
#pragma unroll
  for (int i = 0; i < 4; ++i) load_a(0, i);
#pragma unroll
  for (int u = 0; u < 6; ++u) load_b(0, u);
  float* rowst = reinterpret_cast<float*>(smem + sizeof(Stage3));   // [TM][2] (mean, 1/(std+eps)) of this block's rows
  if (p.ln_stats_in && t < TM) {
    float mean, var;
    ln_row_moments(p.ln_stats_in + (size_t)min(m0 + t, p.M - 1) * p.ln_nseg * 2, p.ln_nseg, p.K, mean, var);
    rowst[2 * t] = mean;
    rowst[2 * t + 1] = 1.f / (sqrtf(var) + p.ln_eps);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    split_half(i, 0);
    split_half(i, 1);
    store_a(i);
  }
#pragma unroll
  for (int u = 0; u < 6; ++u) store_b(u);
  if (nk > 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) load_a(TK, i);
#pragma unroll
    for (int u = 0; u < 6; ++u) load_b(TK, u);
  }
  //@probe VCR_PROBE_STAMP(1);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment addresses: row block i of A / j of B = rows w * 64 + 16 i + (lane & 15), 16-B chunk (lane >> 4)
  const short* fap[4];
  const short* fbp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ar = wm * 64 + i * 16 + l15, br = wn * 64 + i * 16 + l15;
    fap[i] = &S.a[0][ar][(quad ^ swz(ar)) * 8];
    fbp[i] = &S.b[0][br][(quad ^ swz(br)) * 8];
  }

  // The MFMA stream of a slab is laid out by hand (sched_barrier fences one CHUNK = one output tile's six MFMAs = 96 cycles
  // of the matrix pipe): a wave's own vector / LDS instructions issue in the shadow of its own MFMAs, while beside another
  // wave that issues MFMAs back to back they get one slot per 20-36 cycles (profiles/rounds4-5/r4f_mfma_valu_coissue.txt) -- and a
  // 16x16x32 MFMA leaves room for about two of them (MI355X_MICROARCH.md: it holds the vector issue for 8 of its 16 cycles).
  // Chunks 0-7 carry the split of the next slab's activations and the requests for the slab after, chunks 8-13 the LDS
  // stores.  Same MFMA order per output element in every build: bit-identical results.
  // (the body is instantiated three times -- steady state, last but one, last slab -- so that it stays ONE basic block: behind
  // a branch hipcc waits for every outstanding load, the one issued a chunk earlier included)
  f32x4 res[16];                                         // (RES) residual[row 32 ps + 4 q + quad of the wave tile][4 columns], index 8 ps + q
  auto slab = [&](auto more_t, auto more2_t, int kt) {
    constexpr bool more = decltype(more_t)::value, more2 = decltype(more2_t)::value;
    lds_barrier();                                       // barrier 1: slab kt is in LDS
    bf16x8 fa[4][3], fb[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        fa[i][pl] = *reinterpret_cast<const bf16x8*>(fap[i] + pl * TM * TK);
        fb[i][pl] = *reinterpret_cast<const bf16x8*>(fbp[i] + pl * TN * TK);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int i = c >> 2, j = c & 3;
      if (c == 8) lds_barrier();                         // barrier 2: every wave has had its fragments for 8 chunks, the image is free
      f32x4 d = acc[i][j];
      d = mfma_bf16(fa[i][1], fb[j][1], d);             // smallest terms first
      d = mfma_bf16(fa[i][0], fb[j][2], d);
      d = mfma_bf16(fa[i][2], fb[j][0], d);
      d = mfma_bf16(fa[i][0], fb[j][1], d);
      d = mfma_bf16(fa[i][1], fb[j][0], d);
      d = mfma_bf16(fa[i][0], fb[j][0], d);
      acc[i][j] = d;
      if constexpr (RES && !more) {                      // last slab: the staging registers are free -- the residual tile, 16 B per chunk
        const int row = min(m0 + wm * 64 + (c >> 3) * 32 + (c & 7) * 4 + quad, p.M - 1);
        res[c] = ld4(p.residual + (size_t)row * p.ldr + min(n0 + wn * 64 + l15 * 4, p.N - 4));
      }
      if constexpr (more) {
        if (c < 8) {                                     // chunks 0-7: half a staged float4 each (registers only)
          split_half(c >> 1, c & 1);
          if constexpr (more2)
            if (c & 1) load_a((kt + 2) * TK, c >> 1);
        } else if (c < 14) {                             // chunks 8-13: the LDS stores -- a row of activations (8-11), a weight piece
          if (c < 12) store_a(c - 8);
          store_b(c - 8);
          if constexpr (more2) load_b((kt + 2) * TK, c - 8);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  int kt = 0;
  for (; kt + 2 < nk; ++kt) slab(std::true_type{}, std::true_type{}, kt);
  if (kt + 1 < nk) slab(std::true_type{}, std::false_type{}, kt++);
  slab(std::false_type{}, std::false_type{}, kt);
  //@probe VCR_PROBE_STAMP(2);

  // epilogue: as linear.hip -- a wave transposes 32 rows x 64 columns of its accumulators through LDS (the image is free: every
  // wave passed barrier 2 of the last slab after its last fragment read) and stores 256-B row pieces
  constexpr int EP = 68;
  float* ot = reinterpret_cast<float*>(smem) + wave * 32 * EP;
  const int c4e = l15 * 4, col = n0 + wn * 64 + c4e;
  const f32x4 bias = (p.bias && col < p.N) ? ld4(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  const bool ln_in = p.ln_stats_in != nullptr;           // block-uniform
  const f32x4 csum = (ln_in && col < p.N) ? ld4(p.ln_colsum + col) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {                       // rows 32 ps .. 32 ps + 31 of the wave tile
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) ot[(ii * 16 + 4 * quad + r) * EP + j * 16 + l15] = acc[2 * ps + ii][j][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (col < p.N) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int rl = q * 4 + quad;
        const int row = m0 + wm * 64 + ps * 32 + rl;
        if (row < p.M) {
          f32x4 v = ld4(&ot[rl * EP + c4e]);
          if (ln_in) {
            const float mean = rowst[2 * (wm * 64 + ps * 32 + rl)], inv = rowst[2 * (wm * 64 + ps * 32 + rl) + 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(inv, fmaf(-mean, csum[e], v[e]), bias[e]);
          } else {
            v = v + bias;
          }
          if (p.relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
          if constexpr (RES) v = v + res[8 * ps + q];
          st4(p.y + (size_t)row * p.ldy + col, v);
          if (p.stats_out) {                             // the 16 lanes of a row group hold this wave's 64 columns of the row
            float s1 = (v[0] + v[1]) + (v[2] + v[3]);
            s1 = row16_sum(s1);
            const float ms = s1 * (1.f / 64.f);          // (sum, second moment about the segment mean): linear.hip
            const float d0 = v[0] - ms, d1 = v[1] - ms, d2 = v[2] - ms, d3 = v[3] - ms;
            float s2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            s2 = row16_sum(s2);
            if (l15 == 0) {
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
  const int lds = sizeof(Stage3) + TM * 2 * sizeof(float);
  static_assert(sizeof(Stage3) >= 4 * 32 * 68 * 4, "epilogue slices fit");
  if (a->residual) {
    VCR_DYN_LDS(linear_bf16x3_kernel<true>, lds);
    hipLaunchKernelGGL(linear_bf16x3_kernel<true>, dim3(tiles_m * tiles_n), dim3(256), lds, (hipStream_t)stream, *a,
                       reinterpret_cast<const short*>(w_planes), tiles_m, tiles_n);
  } else {
    VCR_DYN_LDS(linear_bf16x3_kernel<false>, lds);
    hipLaunchKernelGGL(linear_bf16x3_kernel<false>, dim3(tiles_m * tiles_n), dim3(256), lds, (hipStream_t)stream, *a,
                       reinterpret_cast<const short*>(w_planes), tiles_m, tiles_n);
  }
  return VCR_LAUNCH_RC();
}
