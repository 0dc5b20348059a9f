// Pair-score kernels over E-wide embeddings (E = 512 for the correspondence head, 128 per attention
// head): everything the reference computes from an n_own x n_str score matrix that it materialises
// and we do not (model/vcrnet_model.py:190-347,402-460, model/transformer.py:35-53).
//
// Geometry (same swapped-QK^T scheme as attention.hip): a block owns 64 (E >= 256) or 32 "owner" points of one
// sample; their embeddings sit in LDS ([64][E+4], conflict-free ds_read_b128) and are the MFMA B operand, so
// each lane owns ONE owner column per owner tile.  The "streamed" points are MFMA rows: wave w walks streamed
// tiles w, w+8, ... with their embeddings going global -> registers in double-buffered 64-wide chunks, each
// fragment used against both owner tiles.  The waves' per-owner partials are merged through LDS.
//
//   op 0  SOFTMAX_PV : corr_o = sum_s softmax_s(score) * xyz_s              (getCopairALL / VcpByDis / DCP)
//   op 1  STATS      : (max_s score, sum_s exp(score - max), argmax_s)       (row / column soft-max statistics)
//   op 2  MASS       : mass_o = sum_s exp(score - m_s) / l_s                 (column sums of a ROW soft-max whose
//                      statistics (m_s, l_s) belong to the STREAMED index: selectCom's scoresColSum/RowSum,
//                      and the key mass of transformer.py:40)
//   score 0: (-|own|^2 + 2 own.str) - |str|^2   score 2: (-|str|^2 + 2 own.str) - |own|^2   (the reference's
//            association with the owner / the streamed side as ITS row index; the first norm rides the
//            MFMA chain as an extra k-step so that (2 dot - norm) is rounded once, like `-xx - inner`)
//   score 1: own.str * scale
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

// OT = owner tiles (of 32) per block.  Every block streams ALL streamed rows past its owners, so the L2 traffic per
// MAC is 1/(32 OT): with one tile the kernel is L2-bandwidth-bound (16 flop/B -> ~6 TB/s at 96 TFLOP/s measured);
// two tiles, shared by 8 waves, halve that.  Each streamed fragment (MFMA A operand, registers) is then used
// against both owner tiles (B operand, LDS).
// nsplit > 1 (OP 1 without arg-max, chosen by the launcher): the streamed tiles are dealt to nsplit workgroups per owner
// block in contiguous runs -- a grid just above a multiple of the resident workgroups (BASELINE configs[2]: 288
// two-tile workgroups on 256 CUs = two rounds for 1.125 rounds of work) becomes nsplit times as many workgroups of
// 1 / nsplit the length.  Each writes its (max, sum) partial to split_work[sp][row]; statmerge_kernel combines them in
// split order.  The scores themselves (score_out) do not depend on the split.
template <int OP, int OT>
__global__ __launch_bounds__(128 * OT * 2, (OT == 1 ? 2 : 1)) void pairscore_kernel(vcr_pairscore_args p, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NW = 4 * OT;                             // waves per block
  constexpr int NO = 32 * OT;                            // owners per block
  const int QP = p.E + 4;
  float* Os = reinterpret_cast<float*>(smem);            // [NO][QP] owner embeddings
  float* mg = Os + NO * QP;                              // [NW waves][NO owners][5]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  int bx, b;
  xcd_chunk2(bx, b);                                     // the owner tiles of one cloud stream the same rows: one XCD's L2
  const int sp = bx % nsplit;                            // (nsplit == 1: sp = 0)
  bx /= nsplit;
  const int o0 = bx * NO;
  const int sb = (b + p.str_batch_shift) % p.nbatch;
  const int chunks = p.E / 64;

  {
    const int per_row = p.E / 4;
    for (int i = t; i < NO * per_row; i += 64 * NW) {
      const int row = i / per_row, c4 = (i % per_row) * 4;
      const int orow = min(o0 + row, p.n_own - 1);
      st4(&Os[row * QP + c4], ld4(p.own + ((size_t)b * p.n_own + orow) * p.ld_own + c4));
    }
  }
  size_t own_row[OT];
  float own_norm[OT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) {
    own_row[ot] = (size_t)b * p.n_own + min(o0 + 32 * ot + l31, p.n_own - 1);
    own_norm[ot] = p.score == 1 ? 0.f : p.own_side4[own_row[ot] * 4 + 3];
  }
  __syncthreads();

  const int ntiles_all = (p.n_str + 31) / 32;
  const int per_split = (ntiles_all + nsplit - 1) / nsplit, t0 = sp * per_split;
  const int ntiles = max(0, min(ntiles_all, t0 + per_split) - t0);   // this workgroup's run of streamed tiles
  const int my_tiles = max(0, (ntiles - w + NW - 1) / NW);           // tiles t0 + w, t0 + w + NW, ...
  const int nflat = my_tiles * chunks;
  const float* sbase = p.str + (size_t)sb * p.n_str * p.ld_str;
  const float* sside = p.str_side4 ? p.str_side4 + (size_t)sb * p.n_str * 4 : nullptr;
  const float* sstat = (OP == 2) ? p.str_stat2 + (size_t)sb * p.str_stat_batch_stride : nullptr;

  f32x4 bufA[8], bufB[8];
  auto load_chunk = [&](int flat, f32x4* dst) {
    const int tile = t0 + w + NW * (flat / chunks), c = flat % chunks;
    const int row = min(tile * 32 + l31, p.n_str - 1);
    const float* kp = sbase + (size_t)row * p.ld_str + 64 * c + 4 * half;
#pragma unroll
    for (int g = 0; g < 8; ++g) dst[g] = ld4(kp + 8 * g);
  };

  float m[OT], l[OT], ox[OT], oy[OT], oz[OT];            // OP 0/1 state
  float best[OT]; int bidx[OT];                          // OP 1 argmax
  float mass[OT];                                        // OP 2
  f32x16 s[OT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) {
    m[ot] = VCR_NEG_INF; l[ot] = ox[ot] = oy[ot] = oz[ot] = 0.f; best[ot] = VCR_NEG_INF; bidx[ot] = 0x7fffffff;
    mass[ot] = 0.f; s[ot] = f32x16{0};
  }
  // The 512-d dot products are accumulated in BLOCKS: every 64-channel chunk is its own k-ascending MFMA chain from zero,
  // and the chunk sums are added up in fp32 -- what a blocked CPU sgemm does (ATen's matmul, vcrnet_model.py:213,289,339),
  // and ~3-5x closer to the exact dot product than ONE 512-step chain, whose late additions each round at the full
  // magnitude of the sum.  Measured against the reference's float64 twin (profiles/accuracy_ledger.txt, round 5): with the
  // single chain the hard-pair selections of the random-feature regime flipped 49-57 of 392 against the twin where the
  // fp32 reference flips 22-37.
  f32x16 tot[OT];
  auto compute = [&](int flat, const f32x4* kf) {
    const int tile = t0 + w + NW * (flat / chunks), c = flat % chunks;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) s[ot] = f32x16{0};
#pragma unroll
    for (int g = 0; g < 8; ++g) {
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const f32x4 qv = ld4(&Os[(32 * ot + l31) * QP + 64 * c + 8 * g + 4 * half]);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[ot] = mfma32(kf[g][e], qv[e], s[ot]);
      }
    }
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      if (c == 0) tot[ot] = s[ot];
      else tot[ot] += s[ot];
    }
    if (c != chunks - 1) return;
    float rn = 0.f;
    if (p.score == 2) rn = sside[(size_t)min(tile * 32 + l31, p.n_str - 1) * 4 + 3];
    f32x4 side[16];
    if (OP == 0 || p.score == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) side[r] = ld4(sside + (size_t)min(tile * 32 + acc_row(r, half), p.n_str - 1) * 4);
    }
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      f32x16& sc16 = tot[ot];
      if (p.score == 0) sc16 = mfma32(half == 0 ? 1.f : 0.f, half == 0 ? -0.5f * own_norm[ot] : 0.f, sc16);
      else if (p.score == 2) sc16 = mfma32(half == 0 ? -0.5f * rn : 0.f, half == 0 ? 1.f : 0.f, sc16);
      float mt = VCR_NEG_INF;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = tile * 32 + acc_row(r, half);
        float sc;
        if (p.score == 0) sc = 2.f * sc16[r] - side[r][3];
        else if (p.score == 2) sc = 2.f * sc16[r] - own_norm[ot];
        else sc = sc16[r] * p.scale;
        sc16[r] = j < p.n_str ? sc : VCR_NEG_INF;
        mt = fmaxf(mt, sc16[r]);
      }
      if (OP == 1 && p.score_out && o0 + 32 * ot + l31 < p.n_own) {
        // keep the scores for the light column / row passes of vcr_scoremass_f32 (the pad past n_str holds -inf)
        float* srow = p.score_out + (own_row[ot] * (size_t)p.ld_score) + tile * 32 + 4 * half;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
          st4(srow + 8 * r4, f32x4{sc16[4 * r4], sc16[4 * r4 + 1], sc16[4 * r4 + 2], sc16[4 * r4 + 3]});
      }
      if (OP == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int jc = min(tile * 32 + acc_row(r, half), p.n_str - 1);
          const float ms = sstat[(size_t)jc * 2], ls = sstat[(size_t)jc * 2 + 1];
          mass[ot] += __builtin_amdgcn_exp2f((sc16[r] - ms) * LOG2E) / ls;   // exp2(-inf) = 0 past the tail
        }
        continue;
      }
      if (OP == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = tile * 32 + acc_row(r, half);
          if (sc16[r] > best[ot]) { best[ot] = sc16[r]; bidx[ot] = j; }
        }
      }
      mt = fmaxf(mt, xhalf(mt));
      const float m_new = fmaxf(m[ot], mt);
      const float alpha = __builtin_amdgcn_exp2f((m[ot] - m_new) * LOG2E);
      float ls = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f((sc16[r] - m_new) * LOG2E);
        ls += pr;
        if (OP == 0) { ax = fmaf(pr, side[r][0], ax); ay = fmaf(pr, side[r][1], ay); az = fmaf(pr, side[r][2], az); }
      }
      l[ot] = l[ot] * alpha + ls;
      if (OP == 0) { ox[ot] = ox[ot] * alpha + ax; oy[ot] = oy[ot] * alpha + ay; oz[ot] = oz[ot] * alpha + az; }
      m[ot] = m_new;
    }
  };

  // nflat is even (chunks = E/64 is even: E % 128 == 0).  The requests of the NEXT chunk are fenced in front of the current
  // chunk's MFMAs, and the steady-state body issues them unconditionally: left alone hipcc sinks the eight loads to the end of
  // the chunk before their use (a register double buffer that hides nothing: vmcnt(7) right after the requests), and behind a
  // branch it cannot count how many loads are in flight and waits for the newest possible one
  if (nflat > 0) {
    load_chunk(0, bufA);
    int f = 0;
    for (; f + 2 < nflat; f += 2) {
      load_chunk(f + 1, bufB);
      __builtin_amdgcn_sched_barrier(0);
      compute(f, bufA);
      load_chunk(f + 2, bufA);
      __builtin_amdgcn_sched_barrier(0);
      compute(f + 1, bufB);
    }
    load_chunk(f + 1, bufB);
    __builtin_amdgcn_sched_barrier(0);
    compute(f, bufA);
    compute(f + 1, bufB);
  }

  // the two halves of a wave hold disjoint streamed rows of the same owner (same running max)
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) {
    float* g = mg + (w * NO + 32 * ot + l31) * 5;
    if (OP == 2) {
      mass[ot] += xhalf(mass[ot]);
      if (half == 0) g[0] = mass[ot];
    } else {
      l[ot] += xhalf(l[ot]);
      if (OP == 0) { ox[ot] += xhalf(ox[ot]); oy[ot] += xhalf(oy[ot]); oz[ot] += xhalf(oz[ot]); }
      if (OP == 1) {
        const float ob = xhalf(best[ot]);
        const int oi = __shfl_xor(bidx[ot], 32, 64);
        if (ob > best[ot] || (ob == best[ot] && oi < bidx[ot])) { best[ot] = ob; bidx[ot] = oi; }
      }
      if (half == 0) {
        g[0] = m[ot]; g[1] = l[ot];
        if (OP == 0) { g[2] = ox[ot]; g[3] = oy[ot]; g[4] = oz[ot]; }
        else { g[2] = best[ot]; g[3] = __int_as_float(bidx[ot]); }
      }
    }
  }
  __syncthreads();
  if (t < NO && o0 + t < p.n_own) {
    const size_t orow = (size_t)b * p.n_own + o0 + t;
    if (OP == 2) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < NW; ++i) acc += mg[(i * NO + t) * 5];
      p.mass[orow] = p.accumulate ? p.mass[orow] + acc : acc;
      return;
    }
    float M = VCR_NEG_INF;
#pragma unroll
    for (int i = 0; i < NW; ++i) M = fmaxf(M, mg[(i * NO + t) * 5]);
    float L = 0.f, X = 0.f, Y = 0.f, Z = 0.f, Bv = VCR_NEG_INF;
    int Bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const float* g = mg + (i * NO + t) * 5;
      const float a = g[0] == VCR_NEG_INF ? 0.f : __builtin_amdgcn_exp2f((g[0] - M) * LOG2E);   // a wave with no tiles
      L = fmaf(g[1], a, L);
      if (OP == 0) { X = fmaf(g[2], a, X); Y = fmaf(g[3], a, Y); Z = fmaf(g[4], a, Z); }
      else {
        const int gi = __float_as_int(g[3]);
        if (g[2] > Bv || (g[2] == Bv && gi < Bi)) { Bv = g[2]; Bi = gi; }
      }
    }
    if (OP == 0) {
      if (nsplit > 1) {                                  // partial (max, sum, weighted xyz) of this run of streamed tiles
        float* pw = p.split_work + ((size_t)sp * p.nbatch * p.n_own + orow) * 8;
        st4(pw, f32x4{M, L, X, Y});
        pw[4] = Z;
      } else {
        st4(p.corr4 + orow * 4, f32x4{X / L, Y / L, Z / L, 0.f});
      }
    } else {
      float* st = nsplit > 1 ? p.split_work + (size_t)sp * p.nbatch * p.n_own * 2 : p.stat2;
      st[orow * 2] = M; st[orow * 2 + 1] = L;
      if (p.argmax) p.argmax[orow] = Bi;
    }
  }
}

// corr4[row] = merge over the nsplit partial (max, sum, weighted xyz) records of a row, in split order
__global__ __launch_bounds__(256) void corrmerge_kernel(const float* part, int nsplit, long rows, float* corr4) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float M = VCR_NEG_INF;
  for (int s = 0; s < nsplit; ++s) M = fmaxf(M, part[((size_t)s * rows + r) * 8]);
  float L = 0.f, X = 0.f, Y = 0.f, Z = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float* g = part + ((size_t)s * rows + r) * 8;
    const float a = g[0] == VCR_NEG_INF ? 0.f : __builtin_amdgcn_exp2f((g[0] - M) * LOG2E);
    L = fmaf(g[1], a, L); X = fmaf(g[2], a, X); Y = fmaf(g[3], a, Y); Z = fmaf(g[4], a, Z);
  }
  st4(corr4 + r * 4, f32x4{X / L, Y / L, Z / L, 0.f});
}

// stat2[row] = merge over the nsplit partial (max, sum) pairs of a row, in split order
__global__ __launch_bounds__(256) void statmerge_kernel(const float* part, int nsplit, long rows, float* stat2) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float M = VCR_NEG_INF;
  for (int s = 0; s < nsplit; ++s) M = fmaxf(M, part[((size_t)s * rows + r) * 2]);
  float L = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float* g = part + ((size_t)s * rows + r) * 2;
    L = fmaf(g[1], g[0] == VCR_NEG_INF ? 0.f : __builtin_amdgcn_exp2f((g[0] - M) * LOG2E), L);
  }
  stat2[r * 2] = M; stat2[r * 2 + 1] = L;
}

// ---- light passes over a stored score matrix S[b][i][j] (selectCom, vcrnet_model.py:217-248) ----
// column pass: per column j   cm_j = max_i S_ij,  cl_j = sum_i exp(S_ij - cm_j)      (soft-max over dim=1)
//                             colmass_j = sum_i exp(S_ij - m_i) / l_i                (column sums of soft-max dim=2)
// 64 columns per block (lanes), rows split over the 4 waves, merged through LDS in wave order.
__global__ __launch_bounds__(256) void score_colpass_kernel(vcr_scoremass_args p) {
  __shared__ float mg[4][64][3];
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane, jc = min(j, p.n_cols - 1);
  const float* S = p.score + (size_t)b * p.n_rows * p.ld + jc;
  const float* rs = p.row_stat2 + (size_t)b * p.n_rows * 2;
  const int per = (p.n_rows + 3) / 4, i0 = w * per, i1 = min(p.n_rows, i0 + per);
  float cm = VCR_NEG_INF, cl = 0.f, mass = 0.f;
  auto step = [&](float v, float mi, float li) {
    mass += __builtin_amdgcn_exp2f((v - mi) * LOG2E) / li;
    if (v > cm) { cl *= __builtin_amdgcn_exp2f((cm - v) * LOG2E); cm = v; }
    cl += __builtin_amdgcn_exp2f((v - cm) * LOG2E);
  };
  int i = i0;
  for (; i + 8 <= i1; i += 8) {                          // eight rows in flight per lane, consumed in row order
    float v[8], m8[8], l8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u] = S[(size_t)(i + u) * p.ld]; m8[u] = rs[2 * (i + u)]; l8[u] = rs[2 * (i + u) + 1]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) step(v[u], m8[u], l8[u]);
  }
  for (; i < i1; ++i) step(S[(size_t)i * p.ld], rs[2 * i], rs[2 * i + 1]);
  mg[w][lane][0] = cm; mg[w][lane][1] = cl; mg[w][lane][2] = mass;
  __syncthreads();
  if (w == 0 && j < p.n_cols) {
    float M = VCR_NEG_INF, L = 0.f, A = 0.f;
    for (int q = 0; q < 4; ++q) M = fmaxf(M, mg[q][lane][0]);
    for (int q = 0; q < 4; ++q) {
      L = fmaf(mg[q][lane][1], __builtin_amdgcn_exp2f((mg[q][lane][0] - M) * LOG2E), L);
      A += mg[q][lane][2];
    }
    const size_t o = (size_t)b * p.n_cols + j;
    p.col_stat2[o * 2] = M; p.col_stat2[o * 2 + 1] = L;
    p.col_mass[o] = A;
  }
}

// row pass: rowmass_i = sum_j exp(S_ij - cm_j) / cl_j  (row sums of the dim=1 soft-max); one wave per row.
__global__ __launch_bounds__(256) void score_rowpass_kernel(vcr_scoremass_args p) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= p.n_rows) return;
  const float* S = p.score + ((size_t)b * p.n_rows + i) * p.ld;
  const float* cs = p.col_stat2 + (size_t)b * p.n_cols * 2;
  float acc = 0.f;
#pragma unroll 4
  for (int j = lane; j < p.n_cols; j += 64)
    acc += __builtin_amdgcn_exp2f((S[j] - cs[2 * j]) * LOG2E) / cs[2 * j + 1];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) p.row_mass[(size_t)b * p.n_rows + i] = acc;
}

int launch(const vcr_pairscore_args* a, vcr_stream_t stream) {
  vcr_stream_scope bound(stream);
  if (!a || !a->own || !a->str) return VCR_EINVAL;
  if (a->nbatch <= 0 || a->n_own <= 0 || a->n_str <= 0 || a->E <= 0 || (a->E % 128) || a->E > 1024) return VCR_EINVAL;
  if ((a->ld_own & 3) || (a->ld_str & 3) || a->ld_own < a->E || a->ld_str < a->E) return VCR_EINVAL;
  if (a->score < 0 || a->score > 2 || a->op < 0 || a->op > 2) return VCR_EINVAL;
  if (a->score != 1 && (!a->own_side4 || !a->str_side4)) return VCR_EINVAL;
  if (a->op == 0 && (!a->corr4 || !a->str_side4)) return VCR_EINVAL;
  if (a->op == 1 && !a->stat2) return VCR_EINVAL;
  if (a->op == 2 && (!a->mass || !a->str_stat2)) return VCR_EINVAL;
  if (a->score_out && (a->op != 1 || (a->ld_score & 3) || a->ld_score < ((a->n_str + 31) & ~31))) return VCR_EINVAL;
  // two owner tiles per block (8 waves) when their embeddings fit the LDS, one (4 waves) otherwise
  const int lds2 = (64 * (a->E + 4) + 8 * 64 * 5) * 4, lds1 = (32 * (a->E + 4) + 4 * 32 * 5) * 4;
  // (narrow scores, E = 128 per attention head: too little MFMA work per block to pay for the half-size grid)
  const int ot = (lds2 <= 160 * 1024 && a->E >= 256 && !(a->variant & 1)) ? 2 : 1;
  const int lds = ot == 2 ? lds2 : lds1;
  if (lds > 160 * 1024) return VCR_EUNSUPPORTED;
  // Split the streamed side when the grid leaves a mostly empty last round of workgroups (statistics passes without an
  // arg-max, caller scratch given): nsplit in 1..VCR_PAIRSCORE_MAX_SPLIT minimising rounds / nsplit, every wave keeping
  // at least one streamed tile, taken only when it saves a fifth of the rounds (each workgroup re-reads its owners).
  const int owner_blocks = (a->n_own + 32 * ot - 1) / (32 * ot);
  int nsplit = 1;
  if ((a->op == 0 || (a->op == 1 && !a->argmax)) && a->split_work && !(a->variant & 4)) {
    const long blocks = (long)owner_blocks * a->nbatch, slots = (long)vcr_cu_count() * (ot == 2 ? 1 : 2);
    const int ntiles = (a->n_str + 31) / 32;
    double best = (double)((blocks + slots - 1) / slots);
    const double base = best;
    for (int sp = 2; sp <= VCR_PAIRSCORE_MAX_SPLIT; ++sp) {
      if (ntiles / sp < 4 * ot || (sp - 1) * ((ntiles + sp - 1) / sp) >= ntiles) break;   // a tile per wave, no empty run
      const double c = (double)((blocks * sp + slots - 1) / slots) / sp;
      const long need = (long)sp * a->nbatch * a->n_own * (a->op == 0 ? 8 : 2);      // floats of partial records
      if (need > a->split_work_floats) break;            // the caller's scratch decides how far the split may go
      if (c < 0.8 * base && c < best - 1e-9) { best = c; nsplit = sp; }
    }
  }
  dim3 grid(owner_blocks * nsplit, a->nbatch);
  hipStream_t s = (hipStream_t)stream;
#define VCR_PS_LAUNCH(OPV, OTV)                                                                                         \
  do {                                                                                                                   \
    VCR_DYN_LDS((pairscore_kernel<OPV, OTV>), lds);                                                                      \
    hipLaunchKernelGGL((pairscore_kernel<OPV, OTV>), grid, dim3(256 * OTV), lds, s, *a, nsplit);                        \
  } while (0)
  if (ot == 2) { if (a->op == 0) VCR_PS_LAUNCH(0, 2); else if (a->op == 1) VCR_PS_LAUNCH(1, 2); else VCR_PS_LAUNCH(2, 2); }
  else         { if (a->op == 0) VCR_PS_LAUNCH(0, 1); else if (a->op == 1) VCR_PS_LAUNCH(1, 1); else VCR_PS_LAUNCH(2, 1); }
#undef VCR_PS_LAUNCH
  if (nsplit > 1) {
    const long rows = (long)a->nbatch * a->n_own;
    const dim3 mg((unsigned)((rows + 255) / 256));
    if (a->op == 0) hipLaunchKernelGGL(corrmerge_kernel, mg, dim3(256), 0, s, a->split_work, nsplit, rows, a->corr4);
    else hipLaunchKernelGGL(statmerge_kernel, mg, dim3(256), 0, s, a->split_work, nsplit, rows, a->stat2);
  }
  return VCR_LAUNCH_RC();
}

}  // namespace

extern "C" int vcr_pairscore_f32(const vcr_pairscore_args* a, vcr_stream_t stream) { return launch(a, stream); }

extern "C" int vcr_scoremass_f32(const vcr_scoremass_args* a, vcr_stream_t stream) {
  if (!a || !a->score || !a->row_stat2 || !a->col_stat2 || !a->col_mass || !a->row_mass) return VCR_EINVAL;
  if (a->nbatch <= 0 || a->n_rows <= 0 || a->n_cols <= 0 || a->ld < a->n_cols) return VCR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(score_colpass_kernel, dim3((a->n_cols + 63) / 64, a->nbatch), dim3(256), 0, s, *a);
  hipLaunchKernelGGL(score_rowpass_kernel, dim3((a->n_rows + 3) / 4, a->nbatch), dim3(256), 0, s, *a);
  return VCR_LAUNCH_RC();
}

extern "C" int vcr_softcorr_f32(const vcr_softcorr_args* a, vcr_stream_t stream) {
  if (!a || !a->qside4 || !a->kside4 || !a->corr4) return VCR_EINVAL;
  if (a->mode != 0 && a->mode != 1) return VCR_EINVAL;
  vcr_pairscore_args p{};
  p.own = a->q; p.ld_own = a->ldq; p.str = a->k; p.ld_str = a->ldk;
  p.own_side4 = a->qside4; p.str_side4 = a->kside4;
  p.nbatch = a->nbatch; p.n_own = a->nq; p.n_str = a->nk; p.E = a->E;
  p.score = a->mode; p.scale = a->scale; p.str_batch_shift = 0; p.op = 0; p.corr4 = a->corr4;
  p.split_work = a->split_work; p.split_work_floats = a->split_work_floats;
  return launch(&p, stream);
}
