// Flash-style attention (model/transformer.py:29-34,55) on the bf16 matrix pipe with fp32-equivalent products:
// the opt-in "bf16x3" companion of attention.hip, same interface (vcr_sdpa_args) and the same orientation.
//
//   Every fp32 operand -- Q, K, V and the soft-max probabilities P -- is split EXACTLY into three bf16 pieces,
//   x = x1 + x2 + x3, and each dot product is evaluated as the six partial products of weight >= 2^-16,
//       a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a3 b1 + a2 b2),
//   on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (see linear_bf16x3.hip): fp32-GEMM accuracy at 2.67x the
//   fp32 matrix rate.  The soft-max itself (max, exp2, sum, rescale) is the fp32 code of attention.hip.
//
// Orientation: S^T = K Q^T (keys = MFMA rows, queries = columns), so a lane owns one query column and the
// probability accumulator is already laid out as the B operand of O^T += V^T P^T.  With the 16-deep bf16 MFMA a
// lane supplies 8 consecutive k per step; the accumulator hands lane half h the keys 4h + (j&3) + 8(j>>2) (+16 per
// step), so V^T is stored in LDS with the key bits 2 and 3 swapped -- the same permutation on both operands, which a
// sum over keys does not see.
//
// Block = 8 waves = 256 queries of one (batch, head): one K/V tile of 32 keys is split once and shared by all eight
// (the VALU cost of the split is what limits this scheme, so it is amortised over as many queries as the register
// file allows).  Q lives in registers as 3 x 8 packed fragments (96 VGPRs); K planes [32][128] and V^T planes
// [128][32] are double buffered in LDS (111 KB), one workgroup per CU, two waves per SIMD.
//
// Measured (MI355X, 32 x 4 heads, N = 1024): 370 us against 516 us for attention.hip, 186 TFLOP/s fp32-equivalent.
// The bound is not the nominal 2.5 PFLOP/s: a register-only v_mfma_f32_32x32x16_bf16 loop on all 256 CUs sustains
// 1.45-1.68 PFLOP/s on random operands (2.29 on zeros, i.e. the chip clocks down under the matrix pipe's power), so
// six MFMAs per product cap the scheme at ~2x the fp32 matrix pipe (itself at 0.84 of ITS nominal rate); the MFMAs
// alone take 266 of the 370 us.  Running the two waves of a SIMD half a tile apart (one in the score MFMAs while its
// partner does soft-max and splits) measured the same 370 us and was dropped for this simpler loop.
#include "bf16x3.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int KPB = 136;   // K plane row pitch in bf16 (272 B): the 16-lane ds_read_b128 groups are conflict-free
constexpr int VPB = 40;    // V^T plane row pitch in bf16 (80 B): likewise
constexpr int EP = 68;     // epilogue row pitch in floats

struct Stage3 {
  short k[3][32][KPB];     // [plane][key][d]
  short vt[3][128][VPB];   // [plane][d][permuted key]
};

template <bool HAS_MASK>
__global__ __launch_bounds__(512, 1) void sdpa_bf16x3_kernel(vcr_sdpa_args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stage3* st = reinterpret_cast<Stage3*>(smem);          // [2]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  // XCD-aware block order (as attention.hip): all query blocks of a (batch, head) pair stream K/V through one L2
  const int nqb = (p.nq + 255) / 256, nbh = p.nbatch * p.heads * (p.ngroups > 1 ? p.ngroups : 1);
  int qb, bh;
  if ((nbh & 7) == 0) {
    const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
    qb = i % nqb; bh = (i / nqb) * 8 + xcd;
  } else {
    qb = blockIdx.x % nqb; bh = blockIdx.x / nqb;
  }
  const int head = bh % p.heads, grp = (bh / p.heads) / p.nbatch, b = (bh / p.heads) % p.nbatch;   // (grp == 0 unless p.ngroups > 1)
  p.q += (size_t)grp * p.q_group_stride; p.k += (size_t)grp * p.k_group_stride;
  p.v += (size_t)grp * p.v_group_stride; p.out += (size_t)grp * p.out_group_stride;
  const int kvb = (b + p.kv_batch_shift) % p.nbatch;
  const int q = qb * 256 + w * 32 + l31;
  const int qc = min(q, p.nq - 1);

  //@probe VCR_PROBE_STAMP(0);
  bf16x8 qh[8], qm[8], ql[8];                            // Q[query l31][16 step + 8 half + 0..7], three planes
  {
    const float* qp = p.q + ((size_t)b * p.nq + qc) * p.ldq + head * 128 + 8 * half;
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const f32x4 a = ld4(qp + 16 * s8), c = ld4(qp + 16 * s8 + 4);
      const float x[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
      split3x8(x, qh[s8], qm[s8], ql[s8]);
    }
  }
  const float* kbase = p.k + (size_t)kvb * p.nk * p.ldk + head * 128;
  const float* vbase = p.v + (size_t)kvb * p.nk * p.ldv + head * 128;
  const int skey = t >> 4, sc = (t & 15) * 4;            // K staging: key row, floats sc..sc+3 and 64+sc..
  // V staging: wave w owns keys 4w..4w+3 (adjacent after the permutation), lane owns head dims lane and lane + 64
  const int vpos = 16 * (w >> 2) + 8 * (w & 1) + 4 * ((w >> 1) & 1);
  const int ntiles = (p.nk + 31) / 32;

  f32x4 rk[2];
  float rv[4][2];
  auto load_k = [&](int tile, int i) {
    const int key = min(tile * 32 + skey, p.nk - 1);
    rk[i] = ld4(kbase + (size_t)key * p.ldk + 64 * i + sc);
  };
  auto load_v = [&](int tile, int e) {
#pragma unroll
    for (int c = 0; c < 4; ++c) rv[c][e] = vbase[(size_t)min(tile * 32 + 4 * w + c, p.nk - 1) * p.ldv + lane + 64 * e];
  };
  // Staging of the NEXT tile in eight pieces (one per 16-deep step of the score MFMAs, in whose shadow they issue): piece
  // 2 i / 2 i + 1 = the two halves of K chunk i (split, then the three 8-B plane stores and the request for the tile after
  // next), pieces 4 + 2 e / 5 + 2 e likewise for head dims lane + 64 e of V^T.
  unsigned hp[2], mp[2], lp[2];
  auto stage_piece = [&](int buf, int tile2, int piece) {
    const int x = (piece >> 1) & 1, second = piece & 1;
    if (piece < 4) split3x2(rk[x][2 * second], rk[x][2 * second + 1], hp[second], mp[second], lp[second]);
    else split3x2(rv[2 * second][x], rv[2 * second + 1][x], hp[second], mp[second], lp[second]);
    if (second) {
      if (piece < 4) {
        *reinterpret_cast<u32x2*>(&st[buf].k[0][skey][sc + 64 * x]) = u32x2{hp[0], hp[1]};
        *reinterpret_cast<u32x2*>(&st[buf].k[1][skey][sc + 64 * x]) = u32x2{mp[0], mp[1]};
        *reinterpret_cast<u32x2*>(&st[buf].k[2][skey][sc + 64 * x]) = u32x2{lp[0], lp[1]};
        load_k(tile2, x);
      } else {
        *reinterpret_cast<u32x2*>(&st[buf].vt[0][lane + 64 * x][vpos]) = u32x2{hp[0], hp[1]};
        *reinterpret_cast<u32x2*>(&st[buf].vt[1][lane + 64 * x][vpos]) = u32x2{mp[0], mp[1]};
        *reinterpret_cast<u32x2*>(&st[buf].vt[2][lane + 64 * x][vpos]) = u32x2{lp[0], lp[1]};
        load_v(tile2, x);
      }
    }
  };

  f32x16 o[4];                                           // o[dt][r] = O[query l31][d = 32 dt + acc_row(r, half)]
#pragma unroll
  for (int d = 0; d < 4; ++d) o[d] = f32x16{0};
  float m = VCR_NEG_INF, l = 0.f;
  const float c2 = p.scale * LOG2E;

  load_k(0, 0); load_k(0, 1); load_v(0, 0); load_v(0, 1);
#pragma unroll
  for (int piece = 0; piece < 8; ++piece) stage_piece(0, 1, piece);   // tile 0 -> LDS, tile 1 requested (rows clamped: always valid)
  __syncthreads();
  //@probe VCR_PROBE_STAMP(1);
  int cur = 0;
  // One tile.  The vector work that does not depend on this tile's scores -- the 3-way split and LDS stores of the NEXT
  // tile's K and V (its loads were issued a tile ago) -- rides in the shadow of the 48 score MFMAs, a piece per step; the
  // split of the second 16 keys' probabilities rides in the P V MFMAs of the first 16.  Before round 4 both sat between
  // the MFMA phases, where the two waves of a SIMD (one workgroup per CU: in phase) went through them together with the
  // matrix pipe idle (~1500 of a tile's ~9500 cycles).  sched_barrier fences keep hipcc from regrouping.  Past the last
  // tile the pieces stage clamped rows into the buffer nobody reads: no branch in the loop body.
  for (int tile = 0; tile < ntiles; ++tile) {
    const Stage3& S = st[cur];
    f32x16 s = {0};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      bf16x8 kf[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) kf[pl] = *reinterpret_cast<const bf16x8*>(&S.k[pl][l31][16 * s8 + 8 * half]);
      s = mfma6(kf, qh[s8], qm[s8], ql[s8], s);
      stage_piece(cur ^ 1, tile + 2, s8);
      __builtin_amdgcn_sched_barrier(0);
    }
    // soft-max of attention.hip's fast path: running maximum kept in log2 units, one fma + exp2 per score
    float mt = VCR_NEG_INF, ls = 0.f;
    if (!HAS_MASK && tile * 32 + 32 <= p.nk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[r]);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = tile * 32 + acc_row(r, half);
        bool ok = key < p.nk;
        if (HAS_MASK) ok = ok && p.key_keep[(size_t)kvb * p.nk + min(key, p.nk - 1)] != 0;
        s[r] = ok ? s[r] : VCR_NEG_INF;
        mt = fmaxf(mt, s[r]);
      }
    }
    mt = fmaxf(mt, xhalf(mt)) * c2;
    const float m_new = fmaxf(m, mt);
    const float mref = (m_new == VCR_NEG_INF) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m - mref);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -mref));
      ls += s[r];
    }
    m = m_new;
    l = l * alpha + ls;
    if (__any(alpha != 1.f)) {
#pragma unroll
      for (int d = 0; d < 4; ++d) o[d] = o[d] * alpha;
    }
    // two 16-key steps; registers 8 sp .. 8 sp + 7 are this step's keys
    unsigned PH[2][4], PM[2][4], PL[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split3x2(s[2 * i], s[2 * i + 1], PH[0][i], PM[0][i], PL[0][i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      const bf16x8 ph = __builtin_bit_cast(bf16x8, u32x4{PH[sp][0], PH[sp][1], PH[sp][2], PH[sp][3]}),
                   pm = __builtin_bit_cast(bf16x8, u32x4{PM[sp][0], PM[sp][1], PM[sp][2], PM[sp][3]}),
                   pl3 = __builtin_bit_cast(bf16x8, u32x4{PL[sp][0], PL[sp][1], PL[sp][2], PL[sp][3]});
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x8 vf[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          vf[pl] = *reinterpret_cast<const bf16x8*>(&S.vt[pl][32 * dt + l31][16 * sp + 8 * half]);
        o[dt] = mfma6(vf, ph, pm, pl3, o[dt]);
        if (sp == 0) split3x2(s[8 + 2 * dt], s[9 + 2 * dt], PH[1][dt], PM[1][dt], PL[1][dt]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    cur ^= 1;
  }

  //@probe VCR_PROBE_STAMP(2);
  // O^T (d rows in registers, query on the lane) -> [query][d] rows through this wave's LDS slice, 64 head dims per
  // round, then 256-B contiguous row stores.  The stage buffers are free (all waves passed the last barrier).
  const float inv = 1.f / (l + xhalf(l));
  float* ot = reinterpret_cast<float*>(smem) + (size_t)w * 32 * EP;
#pragma unroll
  for (int round = 0; round < 2; ++round) {
#pragma unroll
    for (int dd = 0; dd < 2; ++dd)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x16& a = o[2 * round + dd];
        st4(&ot[l31 * EP + 32 * dd + 8 * g + 4 * half],
            f32x4{a[4 * g] * inv, a[4 * g + 1] * inv, a[4 * g + 2] * inv, a[4 * g + 3] * inv});
      }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = 4 * i + (lane >> 4), c4 = (lane & 15) * 4;
      const int qq = qb * 256 + w * 32 + row;
      if (qq < p.nq)
        st4(p.out + ((size_t)b * p.nq + qq) * p.ldo + head * 128 + 64 * round + c4, ld4(&ot[row * EP + c4]));
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

// Same contract as vcr_sdpa_f32 for the attention-output form (out != NULL, no row statistics / score dump):
// the statistics-only passes of the partial-overlap path stay on vcr_sdpa_f32.
extern "C" int vcr_sdpa_bf16x3_f32(const vcr_sdpa_args* a, vcr_stream_t stream) {
  if (!a || !a->q || !a->k || !a->v || !a->out) return VCR_EINVAL;
  if (a->key_index) return VCR_EUNSUPPORTED;             // indexed-key launches: vcr_sdpa_f32 only
  if (a->rowstat || a->score_out || !(a->scale > 0.f)) return VCR_EUNSUPPORTED;
  if (a->nbatch <= 0 || a->heads <= 0 || a->nq <= 0 || a->nk <= 0) return VCR_EINVAL;
  if ((a->ldq & 3) || (a->ldk & 3) || (a->ldo & 3)) return VCR_EINVAL;
  if (a->ldq < a->heads * 128 || a->ldk < a->heads * 128 || a->ldv < a->heads * 128) return VCR_EINVAL;
  if (((uintptr_t)a->q & 15) || ((uintptr_t)a->k & 15) || ((uintptr_t)a->out & 15)) return VCR_EINVAL;
  dim3 grid(((a->nq + 255) / 256) * a->heads * a->nbatch * (a->ngroups > 1 ? a->ngroups : 1));
  const int lds = 2 * sizeof(Stage3);
  static_assert(2 * sizeof(Stage3) >= 8 * 32 * EP * 4, "epilogue slices fit");
  hipStream_t s = (hipStream_t)stream;
  if (a->key_keep) {
    VCR_DYN_LDS(sdpa_bf16x3_kernel<true>, lds);
    hipLaunchKernelGGL(sdpa_bf16x3_kernel<true>, grid, dim3(512), lds, s, *a);
  } else {
    VCR_DYN_LDS(sdpa_bf16x3_kernel<false>, lds);
    hipLaunchKernelGGL(sdpa_bf16x3_kernel<false>, grid, dim3(512), lds, s, *a);
  }
  return VCR_LAUNCH_RC();
}
