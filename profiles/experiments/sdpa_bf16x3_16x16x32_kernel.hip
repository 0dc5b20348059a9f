// RECORD (not built): round-4 attempt at vcr_sdpa_bf16x3_f32 on v_mfma_f32_16x16x32_bf16 (correct: tests pass except the tight error-vs-fp32
// bound at N = 256, 1.12e-5 against 1.05e-5).  A timing ablation of the 32x32x16 kernel with every MFMA replaced by two 16x16x32 ran
// 7828 cycles per tile at 1.86 GHz (8400 at 1.66 GHz for the product kernel, profiles/rounds4-5/r4y_timeline_sdpa_bf16x3.txt) = 1.22x; this real
// kernel needs more registers than a wave has at two waves per SIMD (Q 96 + O 64 + two query blocks of soft-max state): 10-13 VGPRs
// spill, the V^T fragments are read twice, and it measured 9044 cycles per tile at 1.85 GHz = 365 us against 380 us: +4 %, not adopted.
// What it would take: one wave per SIMD (512 registers: all 24 fragments of a tile resident, four query blocks software-pipelined).

// Flash-style attention (model/transformer.py:29-34,55) on the bf16 matrix pipe with fp32-equivalent products:
// the opt-in "bf16x3" companion of attention.hip, same interface (vcr_sdpa_args) and the same orientation.
//
//   Every fp32 operand -- Q, K, V and the soft-max probabilities P -- is split EXACTLY into three bf16 pieces,
//   x = x1 + x2 + x3, and each dot product is evaluated as the six partial products of weight >= 2^-16,
//       a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a3 b1 + a2 b2),
//   on the bf16 matrix pipe with fp32 accumulation (see linear_bf16x3.hip): fp32-GEMM accuracy at 2.67x the
//   fp32 matrix rate.  The soft-max itself (max, exp2, sum, rescale) is the fp32 code of attention.hip.
//
// Orientation: S^T = K Q^T (keys = MFMA rows, queries = columns) on v_mfma_f32_16x16x32_bf16: a wave owns 32 queries = two
// 16-query blocks, a tile is 32 keys = two 16-key blocks, a step is 32 head dims.  A lane holds, per query block, the
// scores of query (lane & 15) against keys 16 kb + 4 (lane >> 4) + r -- eight values that are exactly the B operand of
// O^T += V^T P^T (one 32-key MFMA step) once V^T is stored with its keys permuted the same way (position 8 quad + 4 kb + r
// holds key 16 kb + 4 quad + r): a sum over keys does not see the order.
//
// Block = 8 waves = 256 queries of one (batch, head): one K/V tile of 32 keys is split once and shared by all eight
// (the VALU cost of the split is what limits this scheme, so it is amortised over as many queries as the register
// file allows).  Q lives in registers as 2 x 4 x 3 packed fragments (96 VGPRs); K planes [32][128] and V^T planes
// [128][32] are double buffered in LDS (96 KB), one workgroup per CU, two waves per SIMD.  Both images are unpadded with the
// 16-B chunk index XOR-swizzled (K: by key & 15; V^T: by {0,2,3,1}[(d >> 2) & 3]) -- conflict-free for the NON-contiguous
// 16-lane groups a ds_read_b128 is served in (MI355X_MICROARCH.md, LDS).
//
// The bf16 matrix pipe is POWER limited here: round 4 measured the 32x32x16 form of this kernel at 1.65 GHz in the tile loop
// and exactly 8400 cycles per tile (6144 of MFMAs), and the same loop on 16x16x32 MFMAs at 1.86 GHz and 7828 cycles
// (profiles/rounds4-5/r4y_timeline_sdpa_bf16x3.txt): hence this shape.  The vector work that does not depend on a tile's scores -- the
// 3-way split and LDS stores of the NEXT tile's K and V (loads issued a tile ago) -- rides in the shadow of the score
// MFMAs, a piece per (key block, step).  (Splitting the second query block's probabilities in the shadow of the first's P V
// MFMAs needs 12 more registers than the wave has: it spilled.)
#include "bf16x3.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int EP = 68;     // epilogue row pitch in floats

struct Stage3 {
  short k[3][32][128];     // [plane][key][d], chunk ^= key & 15
  short vt[3][128][32];    // [plane][d][permuted key], chunk ^= vswz(d)
};

__device__ __forceinline__ int vswz(int d) { return (0x78 >> (2 * ((d >> 2) & 3))) & 3; }

template <bool HAS_MASK>
__global__ __launch_bounds__(512, 1) void sdpa_bf16x3_kernel(vcr_sdpa_args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stage3* st = reinterpret_cast<Stage3*>(smem);          // [2]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int quad = lane >> 4, l15 = lane & 15;
  // XCD-aware block order (as attention.hip): all query blocks of a (batch, head) pair stream K/V through one L2
  const int nqb = (p.nq + 255) / 256, nbh = p.nbatch * p.heads;
  int qb, bh;
  if ((nbh & 7) == 0) {
    const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
    qb = i % nqb; bh = (i / nqb) * 8 + xcd;
  } else {
    qb = blockIdx.x % nqb; bh = blockIdx.x / nqb;
  }
  const int head = bh % p.heads, b = bh / p.heads;
  const int kvb = (b + p.kv_batch_shift) % p.nbatch;

  //@probe VCR_PROBE_STAMP(0);
  bf16x8 qf[2][4][3];                                    // Q[query 16 qi + l15][32 step + 8 quad + 0..7], three planes
#pragma unroll
  for (int qi = 0; qi < 2; ++qi) {
    const int qc = min(qb * 256 + w * 32 + 16 * qi + l15, p.nq - 1);
    const float* qp = p.q + ((size_t)b * p.nq + qc) * p.ldq + head * 128 + 8 * quad;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const f32x4 a = ld4(qp + 32 * s4), c = ld4(qp + 32 * s4 + 4);
      const float x[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
      split3x8(x, qf[qi][s4][0], qf[qi][s4][1], qf[qi][s4][2]);
    }
  }
  const float* kbase = p.k + (size_t)kvb * p.nk * p.ldk + head * 128;
  const float* vbase = p.v + (size_t)kvb * p.nk * p.ldv + head * 128;
  const int skey = t >> 4, sc = (t & 15) * 4;            // K staging: key row, floats sc..sc+3 and 64+sc..
  const int koff = ((((t & 15) >> 1) ^ (skey & 15)) * 8) + 4 * (t & 1);   // (+64 floats = chunk + 8: koff ^ 64)
  // V staging: wave w owns keys 4w..4w+3 (positions 8 (w & 3) + 4 (w >> 2) + c), lane owns head dims lane and lane + 64
  const int voff = ((w & 3) ^ vswz(lane)) * 8 + 4 * (w >> 2);             // (vswz(lane + 64) == vswz(lane))
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
  // Staging of the NEXT tile in eight pieces (one per (key block, step) of the score MFMAs, in whose shadow they issue):
  // piece 2 i / 2 i + 1 = the two halves of K chunk i (split, then the three 8-B plane stores and the request for the tile
  // after next), pieces 4 + 2 e / 5 + 2 e likewise for head dims lane + 64 e of V^T.
  unsigned hp[2], mp[2], lp[2];
  auto stage_piece = [&](int buf, int tile2, int piece) {
    const int x = (piece >> 1) & 1, second = piece & 1;
    if (piece < 4) split3x2(rk[x][2 * second], rk[x][2 * second + 1], hp[second], mp[second], lp[second]);
    else split3x2(rv[2 * second][x], rv[2 * second + 1][x], hp[second], mp[second], lp[second]);
    if (second) {
      if (piece < 4) {
        *reinterpret_cast<u32x2*>(&st[buf].k[0][skey][koff ^ (64 * x)]) = u32x2{hp[0], hp[1]};
        *reinterpret_cast<u32x2*>(&st[buf].k[1][skey][koff ^ (64 * x)]) = u32x2{mp[0], mp[1]};
        *reinterpret_cast<u32x2*>(&st[buf].k[2][skey][koff ^ (64 * x)]) = u32x2{lp[0], lp[1]};
        load_k(tile2, x);
      } else {
        *reinterpret_cast<u32x2*>(&st[buf].vt[0][lane + 64 * x][voff]) = u32x2{hp[0], hp[1]};
        *reinterpret_cast<u32x2*>(&st[buf].vt[1][lane + 64 * x][voff]) = u32x2{mp[0], mp[1]};
        *reinterpret_cast<u32x2*>(&st[buf].vt[2][lane + 64 * x][voff]) = u32x2{lp[0], lp[1]};
        load_v(tile2, x);
      }
    }
  };

  f32x4 o[8][2];                                         // o[db][qi][r] = O[query 16 qi + l15][d = 16 db + 4 quad + r]
#pragma unroll
  for (int d = 0; d < 8; ++d) o[d][0] = o[d][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m[2] = {VCR_NEG_INF, VCR_NEG_INF}, l[2] = {0.f, 0.f};
  const float c2 = p.scale * LOG2E;

  load_k(0, 0); load_k(0, 1); load_v(0, 0); load_v(0, 1);
#pragma unroll
  for (int piece = 0; piece < 8; ++piece) stage_piece(0, 1, piece);   // tile 0 -> LDS, tile 1 requested (rows clamped: always valid)
  __syncthreads();
  //@probe VCR_PROBE_STAMP(1);
  int cur = 0;
  // One tile.  sched_barrier fences keep hipcc from regrouping the hand-placed pieces.  Past the last tile the pieces stage
  // clamped rows into the buffer nobody reads: no branch in the loop body.
  for (int tile = 0; tile < ntiles; ++tile) {
    const Stage3& S = st[cur];
    f32x4 s[2][2];                                       // s[kb][qi][r]: key 16 kb + 4 quad + r, query 16 qi + l15
    s[0][0] = s[0][1] = s[1][0] = s[1][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // score MFMAs, plane by plane (a2, then a1, then a0 products, both query blocks each): a K plane's register is free after its
    // last product and takes the NEXT chunk's fragment at once -- a rolling prefetch of 6-10 MFMAs' lead without a second
    // register set (the wave has none to spare)
    auto kfrag = [&](int c, int pl) {
      return *reinterpret_cast<const bf16x8*>(&S.k[pl][16 * (c >> 2) + l15][((4 * (c & 3) + quad) ^ l15) * 8]);
    };
    bf16x8 kf[3] = {kfrag(0, 0), kfrag(0, 1), kfrag(0, 2)};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int kb = c >> 2, s4 = c & 3;
      f32x4 a = s[kb][0], b = s[kb][1];
      a = mfma16_bf16(kf[2], qf[0][s4][0], a); b = mfma16_bf16(kf[2], qf[1][s4][0], b);
      if (c < 7) kf[2] = kfrag(c + 1, 2);
      __builtin_amdgcn_sched_barrier(0);
      a = mfma16_bf16(kf[1], qf[0][s4][1], a); b = mfma16_bf16(kf[1], qf[1][s4][1], b);
      a = mfma16_bf16(kf[1], qf[0][s4][0], a); b = mfma16_bf16(kf[1], qf[1][s4][0], b);
      if (c < 7) kf[1] = kfrag(c + 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      a = mfma16_bf16(kf[0], qf[0][s4][2], a); b = mfma16_bf16(kf[0], qf[1][s4][2], b);
      a = mfma16_bf16(kf[0], qf[0][s4][1], a); b = mfma16_bf16(kf[0], qf[1][s4][1], b);
      a = mfma16_bf16(kf[0], qf[0][s4][0], a); b = mfma16_bf16(kf[0], qf[1][s4][0], b);
      if (c < 7) kf[0] = kfrag(c + 1, 0);
      s[kb][0] = a; s[kb][1] = b;
      stage_piece(cur ^ 1, tile + 2, c);
      __builtin_amdgcn_sched_barrier(0);
    }
    // soft-max of attention.hip's fast path: running maximum kept in log2 units, one fma + exp2 per score
    float mt[2] = {VCR_NEG_INF, VCR_NEG_INF}, ls[2] = {0.f, 0.f};
    if (!HAS_MASK && tile * 32 + 32 <= p.nk) {
#pragma unroll
      for (int qi = 0; qi < 2; ++qi)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) mt[qi] = fmaxf(mt[qi], s[kb][qi][r]);
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = tile * 32 + 16 * kb + 4 * quad + r;
          bool ok = key < p.nk;
          if (HAS_MASK) ok = ok && p.key_keep[(size_t)kvb * p.nk + min(key, p.nk - 1)] != 0;
#pragma unroll
          for (int qi = 0; qi < 2; ++qi) {
            s[kb][qi][r] = ok ? s[kb][qi][r] : VCR_NEG_INF;
            mt[qi] = fmaxf(mt[qi], s[kb][qi][r]);
          }
        }
    }
    float alpha[2], mref[2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) mt[qi] = fmaxf(mt[qi], __shfl_xor(mt[qi], 16, 64));
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
      mt[qi] = fmaxf(mt[qi], xhalf(mt[qi])) * c2;
      const float m_new = fmaxf(m[qi], mt[qi]);
      mref[qi] = (m_new == VCR_NEG_INF) ? 0.f : m_new;
      alpha[qi] = __builtin_amdgcn_exp2f(m[qi] - mref[qi]);
      m[qi] = m_new;
    }
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[kb][qi][r] = __builtin_amdgcn_exp2f(fmaf(s[kb][qi][r], c2, -mref[qi]));
          ls[qi] += s[kb][qi][r];
        }
      l[qi] = l[qi] * alpha[qi] + ls[qi];
    }
    if (__any(alpha[0] != 1.f || alpha[1] != 1.f)) {
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        o[d][0] = o[d][0] * alpha[0];
        o[d][1] = o[d][1] * alpha[1];
      }
    }
    // P^T of query block qi as the B operand: k positions 8 quad + 4 kb + r <- s[kb][qi][r].  P V MFMAs plane by plane with the
    // same rolling fragment prefetch (chunk = one 16-row block of V^T for one query block).
    auto vfrag = [&](int db, int pl) {
      return *reinterpret_cast<const bf16x8*>(&S.vt[pl][16 * (db & 7) + l15][(quad ^ vswz(l15)) * 8]);
    };
    bf16x8 vf[3] = {vfrag(0, 0), vfrag(0, 1), vfrag(0, 2)};
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
      const float x[8] = {s[0][qi][0], s[0][qi][1], s[0][qi][2], s[0][qi][3], s[1][qi][0], s[1][qi][1], s[1][qi][2], s[1][qi][3]};
      bf16x8 ph, pm, pl3;
      split3x8(x, ph, pm, pl3);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int db = 0; db < 8; ++db) {
        const bool more = qi == 0 || db < 7;             // (the chunk after (0, 7) is (1, 0): block 0 again)
        f32x4 a = o[db][qi];
        a = mfma16_bf16(vf[2], ph, a);
        if (more) vf[2] = vfrag(db + 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        a = mfma16_bf16(vf[1], pm, a);
        a = mfma16_bf16(vf[1], ph, a);
        if (more) vf[1] = vfrag(db + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        a = mfma16_bf16(vf[0], pl3, a);
        a = mfma16_bf16(vf[0], pm, a);
        a = mfma16_bf16(vf[0], ph, a);
        if (more) vf[0] = vfrag(db + 1, 0);
        o[db][qi] = a;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    cur ^= 1;
  }

  //@probe VCR_PROBE_STAMP(2);
  // O^T (d rows in registers, query on the lane) -> [query][d] rows through this wave's LDS slice, 64 head dims per
  // round, then 256-B contiguous row stores.  The stage buffers are free (all waves passed the last barrier).
  float inv[2];
#pragma unroll
  for (int qi = 0; qi < 2; ++qi) {
    float lt = l[qi] + __shfl_xor(l[qi], 16, 64);
    lt += xhalf(lt);
    inv[qi] = 1.f / lt;
  }
  float* ot = reinterpret_cast<float*>(smem) + (size_t)w * 32 * EP;
#pragma unroll
  for (int round = 0; round < 2; ++round) {
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) st4(&ot[(16 * qi + l15) * EP + 16 * dd + 4 * quad], o[4 * round + dd][qi] * inv[qi]);
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
  if (a->ngroups > 1 || a->key_index) return VCR_EUNSUPPORTED;   // grouped / indexed-key launches: vcr_sdpa_f32 only
  if (a->rowstat || a->score_out || !(a->scale > 0.f)) return VCR_EUNSUPPORTED;
  if (a->nbatch <= 0 || a->heads <= 0 || a->nq <= 0 || a->nk <= 0) return VCR_EINVAL;
  if ((a->ldq & 3) || (a->ldk & 3) || (a->ldo & 3)) return VCR_EINVAL;
  if (a->ldq < a->heads * 128 || a->ldk < a->heads * 128 || a->ldv < a->heads * 128) return VCR_EINVAL;
  if (((uintptr_t)a->q & 15) || ((uintptr_t)a->k & 15) || ((uintptr_t)a->out & 15)) return VCR_EINVAL;
  dim3 grid(((a->nq + 255) / 256) * a->heads * a->nbatch);
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
