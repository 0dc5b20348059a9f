// EXPERIMENT RECORD -- not part of the build (moved out of vcr-net_amd/csrc/attention.hip in round 3).
//
// sdpa16_kernel: the flash-style attention of attention.hip on v_mfma_f32_16x16x4_f32 instead of 32x32x2 (the round-2 review
// asked for the second fp32 MFMA shape in linear AND sdpa, "keep the faster by wall").  Correct (held to fp64 like the
// 32x32x2 kernel, masked / statistics / stored-scores forms included) and SLOWER on every shape, 32x32x2 / 16x16x4:
//   2B x H = 32 x 4, N = 1024: 583 / 588 us     48 x 4, N = 768: 517 / 524 us
//   32 x 4, N = 2048: 2032 / 2160 us            64 x 4, N = 4096: 16.0 / 16.9 ms        (profiles/rounds1-3/r3b_bench_sdpa.txt)
//   whole step at BASELINE configs[1]: 5.209 vs 5.307 ms (profiles: r3b_bench_l32_s32 / l32_s16)
// Why: a query column is spread over four lanes (two exchanges per soft-max statistic instead of one), a wave carries two
// 16-query column groups (two sets of running max / sum, twice the accumulator rescales), and the kernel was already at
// the matrix pipe's sustained rate.  It compiled inside attention.hip's anonymous namespace (Stage, KP, LOG2E, mfma16,
// ld4 / st4) and was selected by vcr_sdpa_args.variant bit 4 (field removed with it).

// ------------------------------------------------------------------------------------------------
// The same kernel on v_mfma_f32_16x16x4_f32 (vcr_sdpa_args.variant bit 4; the chip holds a higher clock on this shape
// under the matrix pipe's power limit).  Same orientation: keys are MFMA rows, queries MFMA columns.  A wave still owns
// 32 queries -- two column groups jq of 16 -- and a 32-key tile is two row groups ik; lane (quarter qt = lane >> 4,
// c = lane & 15) holds, of score tile (ik, jq), keys 16 ik + 4 qt + r (r = 0..3) of query 16 jq + c, and that register IS
// the B operand of the k-step whose A operand is V[that key][d]: no LDS round trip between the two chains here either.
// A query column is spread over four lanes (c, c+16, c+32, c+48): the soft-max statistics take two exchanges.
// Output dims: MFMA group (h, dg) owns d = 64 h + 4 i + dg (i = A-operand lane), so one ds_read_b128 of
// V[key][64 h + 4 c ..] feeds four groups.
template <bool HAS_MASK, bool DO_PV>
__global__ __launch_bounds__(256, 2) void sdpa16_kernel(vcr_sdpa_args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stage* st = reinterpret_cast<Stage*>(smem);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int half = lane >> 5, l31 = lane & 31, qt = lane >> 4, l15 = lane & 15;
  const int nqb = (p.nq + 127) / 128, nbh = p.nbatch * p.heads * (p.ngroups > 1 ? p.ngroups : 1);
  int qb, bh;
  if ((nbh & 7) == 0) {
    const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
    qb = i % nqb; bh = (i / nqb) * 8 + xcd;
  } else {
    qb = blockIdx.x % nqb; bh = blockIdx.x / nqb;
  }
  const int head = bh % p.heads;
  const int grp = (bh / p.heads) / p.nbatch, b = (bh / p.heads) % p.nbatch;
  p.q += (size_t)grp * p.q_group_stride; p.k += (size_t)grp * p.k_group_stride;
  if (DO_PV) { p.v += (size_t)grp * p.v_group_stride; p.out += (size_t)grp * p.out_group_stride; }
  const int kvb = (b + p.kv_batch_shift) % p.nbatch;
  int q[2];
  f32x4 qf[2][8];                                        // [jq][g]: dims 16 g + 4 qt .. + 3 of query 16 jq + l15
#pragma unroll
  for (int jq = 0; jq < 2; ++jq) {
    q[jq] = qb * 128 + w * 32 + 16 * jq + l15;
    const float* qp = p.q + ((size_t)b * p.nq + min(q[jq], p.nq - 1)) * p.ldq + head * 128 + 4 * qt;
#pragma unroll
    for (int g = 0; g < 8; ++g) qf[jq][g] = ld4(qp + 16 * g);
  }
  const float* kbase = p.k + (size_t)kvb * p.nk * p.ldk + head * 128;
  const float* vbase = p.v + (size_t)kvb * p.nk * p.ldv + head * 128;
  const int srow = t >> 5, sc4 = (t & 31) * 4;
  const int ntiles = (p.nk + 31) / 32;

  f32x4 rk[4], rv[4];
  auto stage_load = [&](int tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = min(tile * 32 + srow + 8 * i, p.nk - 1);
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
  auto xq = [](float v) {                                // max over the four lanes of a query column
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
  };

  f32x4 o[2][4][2];                                      // [h][dg][jq]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int jq = 0; jq < 2; ++jq) o[h][d][jq] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m[2] = {VCR_NEG_INF, VCR_NEG_INF}, l[2] = {0.f, 0.f};
  const bool fast = DO_PV && p.rowstat == nullptr && p.scale > 0.f;   // wave-uniform
  const float c2 = p.scale * LOG2E;

  stage_load(0);
  stage_write(0);
  __syncthreads();
  int cur = 0;
  for (int tile = 0; tile < ntiles; ++tile) {
    if (tile + 1 < ntiles) stage_load(tile + 1);
    f32x4 s[2][2];                                       // [ik][jq]
#pragma unroll
    for (int ik = 0; ik < 2; ++ik)
#pragma unroll
      for (int jq = 0; jq < 2; ++jq) s[ik][jq] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      f32x4 kf[2];
#pragma unroll
      for (int ik = 0; ik < 2; ++ik) kf[ik] = ld4(&st[cur].k[16 * ik + l15][16 * g + 4 * qt]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ik = 0; ik < 2; ++ik)
#pragma unroll
          for (int jq = 0; jq < 2; ++jq) s[ik][jq] = mfma16(kf[ik][e], qf[jq][g][e], s[ik][jq]);
    }
    __builtin_amdgcn_s_setprio(0);
    const bool interior = !HAS_MASK && tile * 32 + 32 <= p.nk;          // no key to mask (wave-uniform)
    bool ok[2][4];
    if (!interior) {
#pragma unroll
      for (int ik = 0; ik < 2; ++ik)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = tile * 32 + 16 * ik + 4 * qt + r;
          ok[ik][r] = key < p.nk;
          if (HAS_MASK) ok[ik][r] = ok[ik][r] && p.key_keep[(size_t)kvb * p.nk + min(key, p.nk - 1)] != 0;
        }
    }
    float alpha[2];
#pragma unroll
    for (int jq = 0; jq < 2; ++jq) {
      float mt = VCR_NEG_INF, ls = 0.f;
      if (fast) {
        // running maximum in log2 units, m2 = max(raw score) * (scale log2 e): a probability is ONE fma + exp2
#pragma unroll
        for (int ik = 0; ik < 2; ++ik)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (!interior) s[ik][jq][r] = ok[ik][r] ? s[ik][jq][r] : VCR_NEG_INF;
            mt = fmaxf(mt, s[ik][jq][r]);
          }
        mt = xq(mt) * c2;
        const float m_new = fmaxf(m[jq], mt);
        const float mref = (m_new == VCR_NEG_INF) ? 0.f : m_new;
        alpha[jq] = __builtin_amdgcn_exp2f(m[jq] - mref);
#pragma unroll
        for (int ik = 0; ik < 2; ++ik)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s[ik][jq][r] = __builtin_amdgcn_exp2f(fmaf(s[ik][jq][r], c2, -mref));
            ls += s[ik][jq][r];
          }
        m[jq] = m_new;
      } else {
#pragma unroll
        for (int ik = 0; ik < 2; ++ik)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s[ik][jq][r] = (interior || ok[ik][r]) ? s[ik][jq][r] * p.scale : VCR_NEG_INF;
            mt = fmaxf(mt, s[ik][jq][r]);
          }
        if (p.score_out && q[jq] < p.nq) {               // keep the scaled scores for vcr_keymass_f32
          float* srow_ = p.score_out + ((((size_t)b * p.heads + head) * p.nq + q[jq]) * p.ld_score) + tile * 32 + 4 * qt;
#pragma unroll
          for (int ik = 0; ik < 2; ++ik) st4(srow_ + 16 * ik, s[ik][jq]);
        }
        mt = xq(mt);
        const float m_new = fmaxf(m[jq], mt);
        const float mref = (m_new == VCR_NEG_INF) ? 0.f : m_new;
        alpha[jq] = __builtin_amdgcn_exp2f((m[jq] - mref) * LOG2E);
#pragma unroll
        for (int ik = 0; ik < 2; ++ik)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s[ik][jq][r] = __builtin_amdgcn_exp2f((s[ik][jq][r] - mref) * LOG2E);
            ls += s[ik][jq][r];
          }
        m[jq] = m_new;
      }
      l[jq] = l[jq] * alpha[jq] + ls;
    }
    if (DO_PV) {
      if (__any(alpha[0] != 1.f || alpha[1] != 1.f)) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int jq = 0; jq < 2; ++jq) o[h][d][jq] = o[h][d][jq] * alpha[jq];
      }
      __builtin_amdgcn_s_setprio(2);
#pragma unroll
      for (int ik = 0; ik < 2; ++ik)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f32x4 vf[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) vf[h] = ld4(&st[cur].v[16 * ik + 4 * qt + r][64 * h + 4 * l15]);
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
              for (int jq = 0; jq < 2; ++jq) o[h][d][jq] = mfma16(vf[h][d], s[ik][jq][r], o[h][d][jq]);
        }
      __builtin_amdgcn_s_setprio(0);
    }
    if (tile + 1 < ntiles) stage_write(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  float lt[2];
#pragma unroll
  for (int jq = 0; jq < 2; ++jq) {
    float x = l[jq] + __shfl_xor(l[jq], 16, 64);         // fixed order: (q0 + q1) + (q2 + q3)
    lt[jq] = x + __shfl_xor(x, 32, 64);
    if (p.rowstat && qt == 0 && q[jq] < p.nq) {
      float* rs = p.rowstat + (((size_t)b * p.heads + head) * p.nq + q[jq]) * 2;
      rs[0] = m[jq]; rs[1] = lt[jq];
    }
  }
  if (DO_PV) {
    float* ot = reinterpret_cast<float*>(smem) + (size_t)w * 32 * KP;
#pragma unroll
    for (int jq = 0; jq < 2; ++jq) {
      const float inv = 1.f / lt[jq];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r)   // o[h][dg][jq][r] = O[query 16 jq + l15][d = 64 h + 4 (4 qt + r) + dg]
          st4(&ot[(16 * jq + l15) * KP + 64 * h + 16 * qt + 4 * r],
              f32x4{o[h][0][jq][r] * inv, o[h][1][jq][r] * inv, o[h][2][jq][r] * inv, o[h][3][jq][r] * inv});
    }
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
}

