// Kernel 3: flash-style attention on fp32 MFMA (model/transformer.py:29-34,55).  The
// virtual-correspondence head built on the same scheme lives in pairscore.hip.
// The n x n score matrix is never written to memory.
//
// Orientation ("swapped QK^T"): scores are computed TRANSPOSED, S^T = K Q^T, with KEYS as MFMA rows
// and QUERIES as MFMA columns.  With v_mfma_f32_32x32x2_f32 a lane then owns one query column
// (lanes l and l+32 split the 32 keys of a tile), so the soft-max max / sum / rescale are lane-local
// plus one exchange with lane^32, and the accumulator P^T[key][query] is ALREADY the B operand of the
// second product  O^T[d][query] += sum_key V^T[d][key] P^T[key][query]:  register r of lane (half, q)
// holds key (r&3)+8(r>>2)+4*half, the A operand for that k-step is V[that key][d] -- no LDS round trip
// and no cross-lane movement between the two MFMA chains.
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int KP = 132;                                  // K/V tile row pitch (floats)

struct Stage { float k[32][KP]; float v[32][KP]; };

// ------------------------------------------------------------------------------------------------
// sdpa: d_k = d_v = 128 per head.  Block = 4 waves = 128 queries of one (batch, head); K/V tiles of
// 32 keys are staged global -> registers -> LDS (double buffered, loads issued before the MFMA block
// and written after it) and shared by the 4 waves; Q (32 x 128 per wave) lives in 64 VGPRs.
// nsplit > 1 (statistics passes only, chosen by the launcher): the key tiles are dealt to nsplit workgroups per query
// block in contiguous runs; each writes its (max, sum) partial to split_work[sp][row], rowstat_merge_kernel combines them
// in split order.  The stored scores do not depend on the split.
template <bool HAS_MASK, bool DO_PV>
__global__ __launch_bounds__(256, 2) void sdpa_kernel(vcr_sdpa_args p, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stage* st = reinterpret_cast<Stage*>(smem);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  // XCD-aware block order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD whole
  // (batch, head) pairs -- all query blocks of a pair then stream the same K/V through ONE L2 instead of
  // eight (measured before: 1.14 GB fetched per launch = 8 x the K/V bytes).  Speed only, never correctness.
  const int nqb = (p.nq + 127) / 128, nbh = p.nbatch * p.heads * (p.ngroups > 1 ? p.ngroups : 1);
  const int nqs = nqb * nsplit;                          // (query block, key run) pairs per (batch, head)
  int qb, bh;
  if ((nbh & 7) == 0) {
    const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
    qb = i % nqs; bh = (i / nqs) * 8 + xcd;
  } else {
    qb = blockIdx.x % nqs; bh = blockIdx.x / nqs;
  }
  const int sp = qb % nsplit;
  qb /= nsplit;
  const int head = bh % p.heads;
  const int grp = (bh / p.heads) / p.nbatch, b = (bh / p.heads) % p.nbatch;   // (grp == 0 unless p.ngroups > 1; grid covers them)
  p.q += (size_t)grp * p.q_group_stride; p.k += (size_t)grp * p.k_group_stride;
  if (DO_PV) { p.v += (size_t)grp * p.v_group_stride; p.out += (size_t)grp * p.out_group_stride; }
  const int kvb = (b + p.kv_batch_shift) % p.nbatch;
  const int q = qb * 128 + w * 32 + l31;
  const int qc = min(q, p.nq - 1);

  //@probe VCR_PROBE_STAMP(0);
  f32x4 qf[16];
  {
    const float* qp = p.q + ((size_t)b * p.nq + qc) * p.ldq + head * 128 + 4 * half;
#pragma unroll
    for (int g = 0; g < 16; ++g) qf[g] = ld4(qp + 8 * g);
  }
  // keys: rows 0..nk-1 of the key batch, or -- key_index -- the rows key_index[kvb][0..nk-1] of its nk_src rows (the
  // decoder's kept keys in partial-overlap mode: no dense copy of the kept K|V rows); the list sits in LDS behind the stages
  const int krows = p.key_index ? p.nk_src : p.nk;
  const float* kbase = p.k + (size_t)kvb * krows * p.ldk + head * 128;
  const float* vbase = p.v + (size_t)kvb * krows * p.ldv + head * 128;
  const int srow = t >> 5, sc4 = (t & 31) * 4;           // staging: rows srow + 8i, one 16-B chunk
  const int ntiles_all = (p.nk + 31) / 32, per_split = (ntiles_all + nsplit - 1) / nsplit;
  const int t0 = sp * per_split, ntiles = min(ntiles_all, t0 + per_split);   // this workgroup's key tiles: t0 .. ntiles - 1
  int* kidx = reinterpret_cast<int*>(smem + 2 * sizeof(Stage));
  if (p.key_index) {
    for (int i = t; i < p.nk; i += 256) kidx[i] = p.key_index[(size_t)kvb * p.nk + i];
    __syncthreads();
  }

  f32x4 rk[4], rv[4];
  auto stage_load = [&](int tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int key = min(tile * 32 + srow + 8 * i, p.nk - 1);
      if (p.key_index) key = kidx[key];
      rk[i] = ld4(kbase + (size_t)key * p.ldk + sc4);
      if (DO_PV) rv[i] = ld4(vbase + (size_t)key * p.ldv + sc4);
    }
  };
  auto stage_write = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      st4(&st[buf].k[srow + 8 * i][sc4], rk[i]);
      if (DO_PV) st4(&st[buf].v[srow + 8 * i][sc4], rv[i]);
    }
  };

  f32x16 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) o[d] = f32x16{0};
  float m = VCR_NEG_INF, l = 0.f;
  const bool fast = DO_PV && p.rowstat == nullptr && p.scale > 0.f;   // wave-uniform
  const float c2 = p.scale * LOG2E;

  stage_load(t0);
  stage_write(0);
  __syncthreads();
  //@probe VCR_PROBE_STAMP(1);
  int cur = 0;
  for (int tile = t0; tile < ntiles; ++tile) {
    if (tile + 1 < ntiles) stage_load(tile + 1);
    f32x16 s = {0};
    __builtin_amdgcn_s_setprio(2);                       // matrix phases outrank the other workgroup's soft-max VALU
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const f32x4 kf = ld4(&st[cur].k[l31][8 * g + 4 * half]);
#pragma unroll
      for (int e = 0; e < 4; ++e) s = mfma32(kf[e], qf[g][e], s);
    }
    __builtin_amdgcn_s_setprio(0);
    float mt = VCR_NEG_INF, alpha, ls = 0.f;
    if (fast) {
      // no row statistics requested: keep the running maximum in log2 units, m2 = max(raw score) * (scale log2 e),
      // so that a probability is ONE fma + exp2 of the raw MFMA result (3 VALU ops per score instead of 5)
      if (!HAS_MASK && tile * 32 + 32 <= p.nk) {         // interior tile: no key to mask (wave-uniform)
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
      alpha = __builtin_amdgcn_exp2f(m - mref);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -mref));
        ls += s[r];
      }
      m = m_new;
    } else {
      if (!HAS_MASK && tile * 32 + 32 <= p.nk) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[r] = s[r] * p.scale;
          mt = fmaxf(mt, s[r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = tile * 32 + acc_row(r, half);
          bool ok = key < p.nk;
          if (HAS_MASK) ok = ok && p.key_keep[(size_t)kvb * p.nk + min(key, p.nk - 1)] != 0;
          s[r] = ok ? s[r] * p.scale : VCR_NEG_INF;
          mt = fmaxf(mt, s[r]);
        }
      }
      if (p.score_out && q < p.nq) {                     // keep the scaled scores for vcr_keymass_f32
        float* srow = p.score_out + ((((size_t)b * p.heads + head) * p.nq + q) * p.ld_score) + tile * 32 + 4 * half;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) st4(srow + 8 * r4, f32x4{s[4 * r4], s[4 * r4 + 1], s[4 * r4 + 2], s[4 * r4 + 3]});
      }
      mt = fmaxf(mt, xhalf(mt));
      const float m_new = fmaxf(m, mt);
      const float mref = (m_new == VCR_NEG_INF) ? 0.f : m_new;
      alpha = __builtin_amdgcn_exp2f((m - mref) * LOG2E);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f((s[r] - mref) * LOG2E);
        ls += s[r];
      }
      m = m_new;
    }
    l = l * alpha + ls;
    if (DO_PV) {
      if (__any(alpha != 1.f)) {
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = o[d] * alpha;
      }
      // MFMA group dg owns the head dims d = 4*i + dg (i = the A-operand lane), so ONE ds_read_b128 of
      // V[key][4*l31 .. 4*l31+3] feeds the four groups of a k-step: 16 wide LDS reads per tile instead of 64 narrow ones.
      __builtin_amdgcn_s_setprio(2);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const f32x4 vf = ld4(&st[cur].v[acc_row(r, half)][4 * l31]);
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = mfma32(vf[d], s[r], o[d]);
      }
      __builtin_amdgcn_s_setprio(0);
    }
    if (tile + 1 < ntiles) stage_write(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  //@probe VCR_PROBE_STAMP(2);
  const float lt = l + xhalf(l);
  if (p.rowstat && half == 0 && q < p.nq) {
    float* rs = (nsplit > 1 ? p.split_work + (size_t)sp * p.nbatch * p.heads * p.nq * 2 : p.rowstat) +
                (((size_t)b * p.heads + head) * p.nq + q) * 2;
    rs[0] = m; rs[1] = lt;
  }
  if (DO_PV) {
    // O^T (d rows in registers, query on the lane) -> [query][d] rows through this wave's LDS slice,
    // then 512-B contiguous row stores.  The stage buffers are free (all waves passed the last barrier).
    // nsplit > 1 (fast path only): the UNNORMALISED partial output of this run of keys goes to plane sp of the scratch,
    // its (running max in log2 units, sum) next to it; sdpa_merge_kernel combines the planes.
    const float inv = nsplit > 1 ? 1.f : 1.f / lt;
    const size_t grows = (size_t)(p.ngroups > 1 ? p.ngroups : 1) * p.nbatch * p.nq;      // rows of one plane
    if (nsplit > 1) {
      p.out = p.split_work + ((size_t)sp * grows + (size_t)grp * p.nbatch * p.nq) * p.ldo;   // (the group offset was added above:
      if (half == 0 && q < p.nq) {                                                           //  undone by the caller, see launcher)
        float* ml = p.split_work + (size_t)nsplit * grows * p.ldo +
                    (((size_t)sp * (p.ngroups > 1 ? p.ngroups : 1) + grp) * p.nbatch * p.heads * p.nq + ((size_t)b * p.heads + head) * p.nq + q) * 2;
        ml[0] = m; ml[1] = lt;
      }
    }
    float* ot = reinterpret_cast<float*>(smem) + (size_t)w * 32 * KP;
#pragma unroll
    for (int r = 0; r < 16; ++r)   // o[dg][r] = O[query l31][d = 4*acc_row(r, half) + dg]
      st4(&ot[l31 * KP + 4 * acc_row(r, half)], f32x4{o[0][r] * inv, o[1][r] * inv, o[2][r] * inv, o[3][r] * inv});
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = 2 * i + half;
      const int qq = qb * 128 + w * 32 + row;
      if (qq < p.nq) {
        const f32x4 v = ld4(&ot[row * KP + l31 * 4]);
        st4(p.out + ((size_t)b * p.nq + qq) * p.ldo + head * 128 + l31 * 4, v);
      }
    }
  }
  //@probe __builtin_amdgcn_s_waitcnt(0); VCR_PROBE_STAMP(3);
}

// ------------------------------------------------------------------------------------------------
// The attention-output fast path (no mask, no statistics, no split, no key list) as a PERSISTENT kernel: 2 x CUs workgroups, each
// walks the work items (query block, batch x head) it would have been dispatched for in the tile kernel's own order (item =
// workgroup + round x grid: the XCD of an item is the XCD the workgroup runs on, the items in flight at any time are the ones
// the dispatcher would have in flight).  What it buys: the next item's Q rows and first K / V tile are requested BEFORE the
// current item's epilogue (their registers are dead by then), so an item boundary costs the epilogue's LDS round trip and two
// barriers instead of a workgroup's teardown, dispatch, Q fetch and first-tile fetch; the output stores drain under the next
// item's first tiles.  Same arithmetic in the same order: bit-identical to sdpa_kernel<false, true>.
__global__ __launch_bounds__(256, 2) void sdpa_persist_kernel(vcr_sdpa_args p, int nitems) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stage* st = reinterpret_cast<Stage*>(smem);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int nqb = (p.nq + 127) / 128, nbh = p.nbatch * p.heads * (p.ngroups > 1 ? p.ngroups : 1);
  const int srow = t >> 5, sc4 = (t & 31) * 4;
  const int ntiles = (p.nk + 31) / 32;
  const float c2 = p.scale * LOG2E;

  struct Item { int qb, b, head; const float *q, *kbase, *vbase; float* out; };
  auto item_of = [&](int i) {
    int qb, bh;
    if ((nbh & 7) == 0) {
      const int xcd = i & 7, j = i >> 3;
      qb = j % nqb; bh = (j / nqb) * 8 + xcd;
    } else {
      qb = i % nqb; bh = i / nqb;
    }
    Item it;
    it.qb = qb;
    it.head = bh % p.heads;
    const int grp = (bh / p.heads) / p.nbatch;
    it.b = (bh / p.heads) % p.nbatch;
    const int kvb = (it.b + p.kv_batch_shift) % p.nbatch;
    it.q = p.q + (size_t)grp * p.q_group_stride;
    it.kbase = p.k + (size_t)grp * p.k_group_stride + (size_t)kvb * p.nk * p.ldk + it.head * 128;
    it.vbase = p.v + (size_t)grp * p.v_group_stride + (size_t)kvb * p.nk * p.ldv + it.head * 128;
    it.out = p.out + (size_t)grp * p.out_group_stride;
    return it;
  };
  f32x4 qf[16];
  f32x4 rk[4], rv[4];
  auto q_load = [&](const Item& it) {
    const int q = min(it.qb * 128 + w * 32 + l31, p.nq - 1);
    const float* qp = it.q + ((size_t)it.b * p.nq + q) * p.ldq + it.head * 128 + 4 * half;
#pragma unroll
    for (int g = 0; g < 16; ++g) qf[g] = ld4(qp + 8 * g);
  };
  auto stage_load = [&](const Item& it, int tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = min(tile * 32 + srow + 8 * i, p.nk - 1);
      rk[i] = ld4(it.kbase + (size_t)key * p.ldk + sc4);
      rv[i] = ld4(it.vbase + (size_t)key * p.ldv + sc4);
    }
  };
  auto stage_write = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      st4(&st[buf].k[srow + 8 * i][sc4], rk[i]);
      st4(&st[buf].v[srow + 8 * i][sc4], rv[i]);
    }
  };

  int i = blockIdx.x;
  if (i >= nitems) return;
  Item it = item_of(i);
  q_load(it);
  stage_load(it, 0);
  stage_write(0);
  __syncthreads();
  for (;;) {
    f32x16 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = f32x16{0};
    float m = VCR_NEG_INF, l = 0.f;
    int cur = 0;
    for (int tile = 0; tile < ntiles; ++tile) {
      if (tile + 1 < ntiles) stage_load(it, tile + 1);
      f32x16 s = {0};
      __builtin_amdgcn_s_setprio(2);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const f32x4 kf = ld4(&st[cur].k[l31][8 * g + 4 * half]);
#pragma unroll
        for (int e = 0; e < 4; ++e) s = mfma32(kf[e], qf[g][e], s);
      }
      __builtin_amdgcn_s_setprio(0);
      float mt = VCR_NEG_INF, alpha, ls = 0.f;
      if (tile * 32 + 32 <= p.nk) {
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = tile * 32 + acc_row(r, half);
          s[r] = key < p.nk ? s[r] : VCR_NEG_INF;
          mt = fmaxf(mt, s[r]);
        }
      }
      mt = fmaxf(mt, xhalf(mt)) * c2;
      const float m_new = fmaxf(m, mt);
      const float mref = (m_new == VCR_NEG_INF) ? 0.f : m_new;
      alpha = __builtin_amdgcn_exp2f(m - mref);
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
      __builtin_amdgcn_s_setprio(2);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const f32x4 vf = ld4(&st[cur].v[acc_row(r, half)][4 * l31]);
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = mfma32(vf[d], s[r], o[d]);
      }
      __builtin_amdgcn_s_setprio(0);
      if (tile + 1 < ntiles) stage_write(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
    // ---- item boundary: the next item's rows are requested first, then this item's output leaves through the (free) stages
    const Item done = it;
    const int nxt = i + (int)gridDim.x;
    const bool more = nxt < nitems;                        // (workgroup-uniform)
    if (more) {
      it = item_of(nxt);
      q_load(it);
      stage_load(it, 0);
    }
    const float inv = 1.f / (l + xhalf(l));
    float* ot = reinterpret_cast<float*>(smem) + (size_t)w * 32 * KP;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      st4(&ot[l31 * KP + 4 * acc_row(r, half)], f32x4{o[0][r] * inv, o[1][r] * inv, o[2][r] * inv, o[3][r] * inv});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (the LDS writes only: the next item's global loads stay in flight)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r2 = 0; r2 < 16; ++r2) {
      const int row = 2 * r2 + half;
      const int qq = done.qb * 128 + w * 32 + row;
      if (qq < p.nq) {
        const f32x4 v = ld4(&ot[row * KP + l31 * 4]);
        st4(done.out + ((size_t)done.b * p.nq + qq) * p.ldo + done.head * 128 + l31 * 4, v);
      }
    }
    if (!more) break;
    __syncthreads();                                       // every wave has read its slice back: the stages are free again
    stage_write(0);
    __syncthreads();
    i = nxt;
  }
}

// rowstat[row] = merge over the nsplit partial (max, sum) pairs of a row, in split order
__global__ __launch_bounds__(256) void rowstat_merge_kernel(const float* part, int nsplit, long rows, float* rowstat) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float M = VCR_NEG_INF;
  for (int s = 0; s < nsplit; ++s) M = fmaxf(M, part[((size_t)s * rows + r) * 2]);
  float L = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float* g = part + ((size_t)s * rows + r) * 2;
    L = fmaf(g[1], g[0] == VCR_NEG_INF ? 0.f : __builtin_amdgcn_exp2f((g[0] - M) * LOG2E), L);
  }
  rowstat[r * 2] = M; rowstat[r * 2 + 1] = L;
}

// out[g][row][c] = sum_s O_s[g][row][c] 2^(m_s - M) / sum_s l_s 2^(m_s - M): the planes of a key-split attention-output
// launch (running maxima in log2 units), merged in split order; one wave per row, 16 B per lane.
__global__ __launch_bounds__(256) void sdpa_merge_kernel(const float* work, int nsplit, int ngroups, int nbatch, int heads, int nq,
                                                         int ldo, float* out, long out_group_stride) {
  const int lane = threadIdx.x & 63;
  const long grows = (long)ngroups * nbatch * nq;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);            // (group, batch, query)
  if (row >= grows) return;
  const int g = (int)(row / ((long)nbatch * nq)), b = (int)((row / nq) % nbatch), q = (int)(row % nq);
  const float* ml = work + (size_t)nsplit * grows * ldo;
  for (int c = lane * 4; c < heads * 128; c += 256) {
    const int head = c >> 7;
    float M = VCR_NEG_INF;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, ml[((((size_t)s * ngroups + g) * nbatch + b) * heads + head) * nq * 2 + (size_t)q * 2]);
    float L = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsplit; ++s) {
      const float* r2 = ml + ((((size_t)s * ngroups + g) * nbatch + b) * heads + head) * nq * 2 + (size_t)q * 2;
      const float a = r2[0] == VCR_NEG_INF ? 0.f : __builtin_amdgcn_exp2f(r2[0] - M);
      L = fmaf(r2[1], a, L);
      const f32x4 o = ld4(work + ((size_t)s * grows + row) * ldo + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaf(o[e], a, acc[e]);
    }
    const float inv = 1.f / L;
    st4(out + (size_t)g * out_group_stride + ((size_t)b * nq + q) * ldo + c, f32x4{acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv});
  }
}

// mass[kb][key] = sum_h sum_q exp(S[qb][h][q][key] - m) / l, qb = (kb + shift) % nbatch; 64 keys per block (lanes),
// the (head, query) rows split over the 4 waves in a fixed order, merged through LDS.  (Row pitch not a multiple of 4.)
__global__ __launch_bounds__(256) void keymass_kernel(vcr_keymass_args p) {
  __shared__ float mg[4][64];
  const int kb = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int key = blockIdx.x * 64 + lane, kc = min(key, p.nk - 1);
  const int qb = (kb + p.q_batch_shift) % p.nbatch;
  const int rows = p.heads * p.nq;                       // (head, query) rows of this query batch, contiguous
  const float* S = p.score + (size_t)qb * rows * p.ld + kc;
  const float* rs = p.rowstat + (size_t)qb * rows * 2;
  const int per = (rows + 3) / 4, r0 = w * per, r1 = min(rows, r0 + per);
  float acc = 0.f;
  int r = r0;
  for (; r + 8 <= r1; r += 8) {                          // eight rows in flight per lane; summed in row order
    float v[8], m8[8], l8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u] = S[(size_t)(r + u) * p.ld]; m8[u] = rs[2 * (r + u)]; l8[u] = rs[2 * (r + u) + 1]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += __builtin_amdgcn_exp2f((v[u] - m8[u]) * LOG2E) / l8[u];
  }
  for (; r < r1; ++r) acc += __builtin_amdgcn_exp2f((S[(size_t)r * p.ld] - rs[2 * r]) * LOG2E) / rs[2 * r + 1];
  mg[w][lane] = acc;
  __syncthreads();
  if (w == 0 && key < p.nk) p.mass[(size_t)kb * p.nk + key] = ((mg[0][lane] + mg[1][lane]) + mg[2][lane]) + mg[3][lane];
}

// The same sums with 256 keys per block (16 B per lane: every row contributes one contiguous KiB, where the kernel above
// reads 256-B pieces 3 KB apart and leaves the DRAM pages to whichever block comes next -- 2.9 TB/s at BASELINE
// configs[2]); the rows are split over 16 waves in a fixed order, merged through LDS in wave order.
constexpr int KM_WAVES = 16;
__global__ __launch_bounds__(64 * KM_WAVES) void keymass4_kernel(vcr_keymass_args p) {
  __shared__ f32x4 mg[KM_WAVES][64];
  const int kb = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int key = blockIdx.x * 256 + lane * 4, kc = min(key, (p.nk - 1) & ~3);
  const int qb = (kb + p.q_batch_shift) % p.nbatch;
  const int rows = p.heads * p.nq;
  const float* S = p.score + (size_t)qb * rows * p.ld + kc;
  const float* rs = p.rowstat + (size_t)qb * rows * 2;
  const int per = (rows + KM_WAVES - 1) / KM_WAVES, r0 = min(rows, w * per), r1 = min(rows, r0 + per);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int r = r0;
  for (; r + 8 <= r1; r += 8) {                          // eight rows (8 KiB per wave) in flight; summed in row order
    f32x4 v[8];
    float m8[8], l8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u] = ld4(S + (size_t)(r + u) * p.ld); m8[u] = rs[2 * (r + u)]; l8[u] = rs[2 * (r + u) + 1]; }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += __builtin_amdgcn_exp2f((v[u][e] - m8[u]) * LOG2E) / l8[u];
  }
  for (; r < r1; ++r) {
    const f32x4 v = ld4(S + (size_t)r * p.ld);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += __builtin_amdgcn_exp2f((v[e] - rs[2 * r]) * LOG2E) / rs[2 * r + 1];
  }
  mg[w][lane] = acc;
  __syncthreads();
  if (w == 0) {
    f32x4 tot = mg[0][lane];
#pragma unroll
    for (int i = 1; i < KM_WAVES; ++i) tot = tot + mg[i][lane];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (key + e < p.nk) p.mass[(size_t)kb * p.nk + key + e] = tot[e];
  }
}

}  // namespace

extern "C" int vcr_keymass_f32(const vcr_keymass_args* a, vcr_stream_t stream) {
  if (!a || !a->score || !a->rowstat || !a->mass) return VCR_EINVAL;
  if (a->nbatch <= 0 || a->heads <= 0 || a->nq <= 0 || a->nk <= 0 || a->ld < a->nk) return VCR_EINVAL;
  // 16 B per lane when every row is 16-B aligned and padded to a multiple of 4 keys (vcr_sdpa_f32's score_out is)
  if ((a->ld & 3) == 0 && (((uintptr_t)a->score) & 15) == 0 && ((a->nk + 3) & ~3) <= a->ld)
    hipLaunchKernelGGL(keymass4_kernel, dim3((a->nk + 255) / 256, a->nbatch), dim3(64 * KM_WAVES), 0, (hipStream_t)stream, *a);
  else
    hipLaunchKernelGGL(keymass_kernel, dim3((a->nk + 63) / 64, a->nbatch), dim3(256), 0, (hipStream_t)stream, *a);
  return VCR_LAUNCH_RC();
}

extern "C" int vcr_sdpa_f32(const vcr_sdpa_args* a, vcr_stream_t stream) {
  vcr_stream_scope bound(stream);
  if (!a || !a->q || !a->k) return VCR_EINVAL;
  const bool pv = a->out != nullptr;
  if (pv && !a->v) return VCR_EINVAL;
  if (!pv && !a->rowstat) return VCR_EINVAL;
  if (a->score_out && (!a->rowstat || (a->ld_score & 3) || a->ld_score < ((a->nk + 31) & ~31))) return VCR_EINVAL;
  if (a->nbatch <= 0 || a->heads <= 0 || a->nq <= 0 || a->nk <= 0) return VCR_EINVAL;
  if ((a->ldq & 3) || (a->ldk & 3) || (pv && ((a->ldv & 3) || (a->ldo & 3)))) return VCR_EINVAL;
  if (a->ldq < a->heads * 128 || a->ldk < a->heads * 128) return VCR_EINVAL;
  const int ng = a->ngroups > 1 ? a->ngroups : 1;
  if (ng > 1 && (!pv || a->key_keep || a->rowstat || a->score_out)) return VCR_EINVAL;
  if (a->key_index && (a->nk_src < 1 || a->key_keep || a->score_out || ng > 1 || a->nk > 16384)) return VCR_EINVAL;
  // Statistics passes (no P V, caller scratch given) split the keys over nsplit workgroups per query block when that
  // shortens the launch by a tenth in a simple round model (a partial last round filled to f costs 0.35 + 0.65 f of a round):
  // 1152 workgroups on 512 slots at BASELINE configs[2] = 2.25 rounds -> four times as many of a quarter the length.
  const long blocks = (long)((a->nq + VCR_SDPA_QROWS - 1) / VCR_SDPA_QROWS) * a->heads * a->nbatch * ng;
  if (a->plan_nbatch != 0 && a->plan_nbatch < a->nbatch) return VCR_EINVAL;
  // (the split decisions below count the blocks of the launch this one stands for: vcr_sdpa_args.plan_nbatch)
  const long plan_blocks = a->plan_nbatch ? blocks / a->nbatch * a->plan_nbatch : blocks;
  int nsplit = 1;
  if (!pv && a->split_work && a->split_work_floats >= (long)VCR_SDPA_MAX_SPLIT * a->nbatch * a->heads * a->nq * 2) {
    const long slots = (long)vcr_cu_count() * 2;
    const int ntiles = (a->nk + 31) / 32;
    auto cost = [&](int sp) {
      const long t = plan_blocks * sp, full = t / slots;
      const double f = (double)(t - full * slots) / slots;
      return ((double)full + (f > 0.0 ? 0.35 + 0.65 * f : 0.0)) / sp;
    };
    double best = cost(1);
    const double base = best;
    for (int sp = 2; sp <= VCR_SDPA_MAX_SPLIT; ++sp) {
      if (ntiles / sp < 4 || (sp - 1) * ((ntiles + sp - 1) / sp) >= ntiles) break;   // >= 4 tiles per run, no empty run
      const double c = cost(sp);
      if (c < 0.9 * base && c < best - 1e-9) { best = c; nsplit = sp; }
    }
  }
  // Attention-output launches of LESS than one round of workgroups (small batches: 64 .. 512 on 512 resident) split the
  // keys too: every workgroup writes its unnormalised partial output and (max, sum) to a plane of the scratch, one more
  // kernel merges the planes (partial outputs: nsplit x the output bytes -- only worth it while that is a few MB).
  if (pv && a->split_work && !a->rowstat && !a->score_out && a->scale > 0.f && !a->key_keep && !a->key_index) {
    const long slots = (long)vcr_cu_count() * VCR_SDPA_WG_PER_CU;
    const int ntiles = (a->nk + 31) / 32;
    const size_t plane = (size_t)ng * a->nbatch * a->nq * a->ldo * 4;
    const size_t plan_plane = plane / a->nbatch * (a->plan_nbatch ? a->plan_nbatch : a->nbatch);     // (of the launch this one stands for)
    const size_t plan_ml = (size_t)ng * (a->plan_nbatch ? a->plan_nbatch : a->nbatch) * a->heads * a->nq * 8;
    for (int sp = VCR_SDPA_MAX_SPLIT; sp >= 2; sp >>= 1)
      if (plan_blocks * sp <= slots && ntiles / sp >= 4 && (sp - 1) * ((ntiles + sp - 1) / sp) < ntiles && plan_plane * sp <= ((size_t)64 << 20) &&
          (size_t)a->split_work_floats * 4 >= sp * (plan_plane + plan_ml)) {
        nsplit = sp;
        break;
      }
  }
  dim3 grid((unsigned)(blocks * nsplit));
  const int lds = 2 * sizeof(Stage) + (a->key_index ? ((a->nk * 4 + 15) & ~15) : 0);
  hipStream_t s = (hipStream_t)stream;
  // The persistent kernel where it applies (the fast attention-output form with at least two items per workgroup) unless
  // vcr_sdpa_args.variant asks for the tile kernel: same bits; measured inside the forward (profiles/r6c_sdpa_variant_bench.txt,
  // alternated on one box) sdpa 1.570 -> 1.559 ms per step at configs[1], 6.18 -> 6.07 at configs[3]'s share, 4.86 -> 4.80 at
  // configs[2], level at configs[4] (48.1 ms: 128 key tiles per item, the item boundary is 1 % of an item there)
  if (a->variant != 0 && a->variant != 1 && a->variant != 2) return VCR_EINVAL;
  if (a->variant != 1 && pv && nsplit == 1 && !a->key_keep && !a->key_index && !a->rowstat && !a->score_out && a->scale > 0.f &&
      blocks >= 2 * (long)vcr_cu_count() * 2) {
    const int lds_p = 2 * sizeof(Stage);
    VCR_DYN_LDS(sdpa_persist_kernel, lds_p);
    hipLaunchKernelGGL(sdpa_persist_kernel, dim3((unsigned)(vcr_cu_count() * 2)), dim3(256), lds_p, s, *a, (int)blocks);
    return VCR_LAUNCH_RC();
  }
#define VCR_SDPA_LAUNCH(M, P)                                                                                         \
  do {                                                                                                                 \
    VCR_DYN_LDS((sdpa_kernel<M, P>), lds);                                                                             \
    hipLaunchKernelGGL((sdpa_kernel<M, P>), grid, dim3(256), lds, s, *a, nsplit);                                     \
  } while (0)
  if (a->key_keep) { if (pv) VCR_SDPA_LAUNCH(true, true); else VCR_SDPA_LAUNCH(true, false); }
  else             { if (pv) VCR_SDPA_LAUNCH(false, true); else VCR_SDPA_LAUNCH(false, false); }
#undef VCR_SDPA_LAUNCH
  if (nsplit > 1 && pv) {
    const long grows = (long)ng * a->nbatch * a->nq;
    hipLaunchKernelGGL(sdpa_merge_kernel, dim3((unsigned)((grows + 3) / 4)), dim3(256), 0, s, a->split_work, nsplit, ng, a->nbatch, a->heads,
                       a->nq, a->ldo, a->out, ng > 1 ? a->out_group_stride : 0L);
  } else if (nsplit > 1) {
    const long rows = (long)a->nbatch * a->heads * a->nq;
    hipLaunchKernelGGL(rowstat_merge_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, a->split_work, nsplit, rows, a->rowstat);
  }
  return VCR_LAUNCH_RC();
}
