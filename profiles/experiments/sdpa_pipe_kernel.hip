// RECORD OF AN EXPERIMENT (round 4) -- not part of the library build.
//
// sdpa as a software pipeline (QK^T of tile t+1 interleaved with the soft-max of tile t inside sched_barrier-fenced chunks;
// K one tile ahead of V in the two LDS stages).  It replaced sdpa_kernel<false, true> on the forward's attention-output
// launches for one build: outputs BIT-IDENTICAL (checksums of five shapes, profiles/experiments/bench_sdpa_ab.py), and the
// launch times IDENTICAL too -- 32 x 1024: 577-587 us both builds, 32 x 2048: 2072-2083, 64 x 4096: 16.05-16.10 ms -- as
// was a lighter variant that only requested the K / V fragments one group ahead (-0..1 %).  The key loop's 8.2 us per tile
// against 6.8 at the matrix peak (profiles/rounds4-5/r4n_timeline_sdpa.txt: prologue 5.4, key loop 263, epilogue 7.7 us of a workgroup
// at 32 x 1024) is therefore neither exposed LDS latency nor the soft-max beside the partner's MFMAs.  It matches the 0.88
// that the persistent linear's MFMA + fragment-read loop reached with everything else ablated away
// (profiles/rounds4-5/r4k_linear_stream.txt): LDS-fed fp32 MFMA loops at two waves per SIMD top out there.
// (kernel body as it was wired into attention.hip; Stage, KP, LOG2E, mfma32, acc_row, xhalf are that file's)

// The attention-OUTPUT launches of the forward (no mask, no statistics, no stored scores, one run of keys per workgroup)
// as a software pipeline.  In sdpa_kernel a tile is  QK^T (64 MFMAs) -> soft-max (~100 vector instructions, more with the
// rescale of O) -> P V (64 MFMAs): the soft-max of one wave runs beside its SIMD partner's MFMAs, where a vector instruction
// gets through once per ~20 cycles (profiles/rounds4-5/r4f_mfma_valu_coissue.txt), and the key loop takes 8.2 us per tile against
// 6.8 at the matrix peak (profiles/rounds4-5/r4n_timeline_sdpa.txt).  Here the scores of tile t+1 are computed WHILE the soft-max of
// tile t runs -- QK^T(t+1)'s MFMAs and soft-max(t)'s instructions alternate inside chunks fenced by sched_barrier(0) --,
// then P V(t) carries the LDS writes of the rows staged for later tiles.  K therefore runs one tile ahead of V in the two
// LDS stages: during iteration t stage[cur] holds V[t] (and receives K[t+2]), stage[cur ^ 1] holds K[t+1] (and receives
// V[t+1]).  Same MFMA order per accumulator and the same soft-max arithmetic as sdpa_kernel<false, true>'s fast path: the
// outputs are bit-identical (the rescale of O is unconditional here: a multiplication by 1.0f is exact).
__global__ __launch_bounds__(256, 2) void sdpa_pipe_kernel(vcr_sdpa_args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Stage* st = reinterpret_cast<Stage*>(smem);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int half = lane >> 5, l31 = lane & 31;
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
  p.v += (size_t)grp * p.v_group_stride; p.out += (size_t)grp * p.out_group_stride;
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
  const int krows = p.key_index ? p.nk_src : p.nk;
  const float* kbase = p.k + (size_t)kvb * krows * p.ldk + head * 128;
  const float* vbase = p.v + (size_t)kvb * krows * p.ldv + head * 128;
  const int srow = t >> 5, sc4 = (t & 31) * 4;
  const int ntiles = (p.nk + 31) / 32;
  int* kidx = reinterpret_cast<int*>(smem + 2 * sizeof(Stage));
  if (p.key_index) {
    for (int i = t; i < p.nk; i += 256) kidx[i] = p.key_index[(size_t)kvb * p.nk + i];
    __syncthreads();
  }
  f32x4 rk[4], rv[4];
  auto rowkey = [&](int tile, int i) {
    int key = min(tile * 32 + srow + 8 * i, p.nk - 1);
    if (p.key_index) key = kidx[key];
    return key;
  };
  auto load_k = [&](int tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) rk[i] = ld4(kbase + (size_t)rowkey(tile, i) * p.ldk + sc4);
  };
  auto load_v = [&](int tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) rv[i] = ld4(vbase + (size_t)rowkey(tile, i) * p.ldv + sc4);
  };
  auto qk_tile = [&](int buf, f32x16& s) {               // s = K[tile in stage buf] Q^T, fragments one k-group ahead
    const float* krow = &st[buf].k[l31][4 * half];
    f32x4 kf[2];
    kf[0] = ld4(krow);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if (g + 1 < 16) kf[(g + 1) & 1] = ld4(krow + 8 * (g + 1));
#pragma unroll
      for (int e = 0; e < 4; ++e) s = mfma32(kf[g & 1][e], qf[g][e], s);
    }
  };

  f32x16 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) o[d] = f32x16{0};
  float m = VCR_NEG_INF, l = 0.f;
  const float c2 = p.scale * LOG2E;

  // prologue: K[0], V[0] -> stage 0, K[1] -> stage 1; S_0 = K[0] Q^T
  load_k(0); load_v(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) { st4(&st[0].k[srow + 8 * i][sc4], rk[i]); st4(&st[0].v[srow + 8 * i][sc4], rv[i]); }
  if (ntiles > 1) {
    load_k(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) st4(&st[1].k[srow + 8 * i][sc4], rk[i]);
  }
  __syncthreads();
  //@probe VCR_PROBE_STAMP(1);
  f32x16 sc = {0};
  qk_tile(0, sc);
  int cur = 0;
  for (int tile = 0; tile < ntiles; ++tile) {
    const bool has1 = tile + 1 < ntiles, has2 = tile + 2 < ntiles;
    if (has2) load_k(tile + 2);
    if (has1) load_v(tile + 1);
    if (tile * 32 + 32 > p.nk) {                         // the last, partial tile: keys past nk do not exist (wave-uniform)
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = (tile * 32 + acc_row(r, half) < p.nk) ? sc[r] : VCR_NEG_INF;
    }
    // ---- phase A: S_{t+1} = K[t+1] Q^T  ||  soft-max of S_t and the rescale of O
    f32x16 sn = {0};
    float mt = VCR_NEG_INF, alpha = 1.f, mref = 0.f, ls = 0.f;
    {
      const float* krow = &st[cur ^ 1].k[l31][4 * half];
      f32x4 kf[2];
      if (has1) kf[0] = ld4(krow);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        if (has1) {
          if (g + 1 < 16) kf[(g + 1) & 1] = ld4(krow + 8 * (g + 1));
#pragma unroll
          for (int e = 0; e < 4; ++e) sn = mfma32(kf[g & 1][e], qf[g][e], sn);
        }
        if (g < 4) {
#pragma unroll
          for (int r = 4 * g; r < 4 * g + 4; ++r) mt = fmaxf(mt, sc[r]);
        } else if (g == 4) {
          mt = fmaxf(mt, xhalf(mt)) * c2;
          const float m_new = fmaxf(m, mt);
          mref = (m_new == VCR_NEG_INF) ? 0.f : m_new;
          alpha = __builtin_amdgcn_exp2f(m - mref);
          m = m_new;
        } else if (g < 13) {                             // g = 5 .. 12: two probabilities per chunk, summed in register order
#pragma unroll
          for (int r = 2 * (g - 5); r < 2 * (g - 5) + 2; ++r) {
            sc[r] = __builtin_amdgcn_exp2f(fmaf(sc[r], c2, -mref));
            ls += sc[r];
          }
        }
        if (g >= 5 && g < 13) {                          // ... and an eighth of O's rescale (x 1.0f when the maximum stood)
          const int d = (g - 5) >> 1, r0 = ((g - 5) & 1) * 8;
#pragma unroll
          for (int r = r0; r < r0 + 8; ++r) o[d][r] *= alpha;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    l = l * alpha + ls;
    // ---- phase B: O += V[t]^T P_t  ||  the staged rows of later tiles into their LDS stages
    {
      f32x4 vf[2];
      vf[0] = ld4(&st[cur].v[acc_row(0, half)][4 * l31]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (r + 1 < 16) vf[(r + 1) & 1] = ld4(&st[cur].v[acc_row(r + 1, half)][4 * l31]);
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = mfma32(vf[r & 1][d], sc[r], o[d]);
        if (r >= 4 && r < 8 && has2) st4(&st[cur].k[srow + 8 * (r - 4)][sc4], rk[r - 4]);          // K[t+2] over K[t]
        if (r >= 8 && r < 12 && has1) st4(&st[cur ^ 1].v[srow + 8 * (r - 8)][sc4], rv[r - 8]);     // V[t+1] over V[t-1]
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    sc = sn;
    cur ^= 1;
  }
  //@probe VCR_PROBE_STAMP(2);
  const float lt = l + xhalf(l);
  const float inv = 1.f / lt;
  float* ot = reinterpret_cast<float*>(smem) + (size_t)w * 32 * KP;
#pragma unroll
  for (int r = 0; r < 16; ++r)
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
  //@probe __builtin_amdgcn_s_waitcnt(0); VCR_PROBE_STAMP(3);
}

