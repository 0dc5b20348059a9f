// EXPERIMENT RECORD (round 2) -- not part of the build (vcr-net_amd/build.py compiles csrc/*.hip only).
// linear_pipe_kernel: the persistent / deferred-epilogue structure of linear_persist_kernel (csrc/linear.hip) for ONE
// workgroup per CU with the k loop pipelined by hand.  To reproduce: paste into csrc/linear.hip after
// linear_persist_kernel and launch it with grid = min(tiles, CUs), 256 threads, 3 * sizeof(TileG) + 3 * BM * 8 bytes of
// LDS (K >= 64).  Results are bit-identical to the product kernels; a K = 512 tile takes 39-40 us (100 TFLOP/s over a
// whole launch) against 37 us per tile for the product's two co-resident workgroups -- DESIGN.md section 5.1 has the
// ablation that explains why (the matrix pipe is not decoupled from LDS / VMEM / register-file traffic).
//
// hipcc waits lgkmcnt(0) in front of every MFMA group once an LDS-DMA is pending and puts vmcnt(0) in front of every
// barrier.  Here the fragment reads are inline asm with their own s_waitcnt (tied to the fragment registers so that no
// MFMA can move above it):
//   * fragments of group g+1 are requested before the MFMAs of group g (two register sets);
//   * three LDS stages: the DMA of slab S+2 is issued at the top of slab S, and the only vmcnt wait -- "all but the
//     newest 8 VMEM operations", i.e. slab S+1 has landed -- sits with the barrier in front of the LAST MFMA group of
//     slab S, followed at once by the reads of slab S+1's first group, which those 16 MFMAs cover.
template <bool LN_IN, bool STATS_OUT>
__global__ __launch_bounds__(256, 1) void linear_pipe_kernel(vcr_linear_args p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  TileG* tile = reinterpret_cast<TileG*>(smem);          // [3]: the slab being multiplied, the next one, the one in flight
  float* rowst = reinterpret_cast<float*>(smem + 3 * sizeof(TileG));   // [3][BM][2] (mean, inv): previous / current / next tile
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int nblk = tiles_m * tiles_n, G = gridDim.x;
  const int nk = p.K / 32;

  // virtual block id -> tile, XCD-aware: ids congruent mod 8 run on one XCD (G is a multiple of 8 or == nblk), and
  // every XCD owns a contiguous run of tiles, so tiles sharing an X panel share an L2
  auto tile_of = [&](int vb, int& m0, int& n0) {
    const int q = nblk / 8, r = nblk % 8, xcd = vb % 8, i = vb / 8;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    m0 = (bid / tiles_n) * BM; n0 = (bid % tiles_n) * BN;
  };
  const int frow = lane >> 3, fpc = lane & 7;
  const float* xa[4];
  const float* wb[4];
  auto set_tile = [&](int m0, int n0) {
#ifdef VCR_TIMELINE
    if (p.variant & 128) { m0 = (m0 / BM % 2) * BM; }    // experiment: every X tile comes from the first 256 rows (L2 hits)
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 32 + 8 * i + frow;
      const int lc = fpc ^ ((row >> 1) & 7);
      xa[i] = p.x + (size_t)min(m0 + row, p.M - 1) * p.ldx + 4 * lc;
      wb[i] = p.w + (size_t)min(n0 + row, p.N - 1) * p.K + 4 * lc;
    }
  };
  auto fill = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(xa[i] + k0, &tile[buf].a[wave * 32 + 8 * i][0]);
      glds16(wb[i] + k0, &tile[buf].b[wave * 32 + 8 * i][0]);
    }
  };
  auto row_stats = [&](int m0, int rs) {                 // LayerNorm (mean, 1/(std+eps)) of the tile's 128 rows
    if (LN_IN && t < BM) {
      const float* sp = p.ln_stats_in + (size_t)min(m0 + t, p.M - 1) * p.ln_nseg * 2;
      float s1 = 0.f, s2 = 0.f;
      for (int sg = 0; sg < p.ln_nseg; ++sg) { s1 += sp[2 * sg]; s2 += sp[2 * sg + 1]; }   // fixed order
      const float mean = s1 / (float)p.K;
      const float var = fmaxf((s2 - s1 * mean) / (float)(p.K - 1), 0.f);                    // unbiased, like x.std()
      rowst[(rs * BM + t) * 2] = mean;
      rowst[(rs * BM + t) * 2 + 1] = 1.f / (sqrtf(var) + p.ln_eps);
    }
  };

  int vb = blockIdx.x;
  if (vb >= nblk) return;
  int m0, n0;
  TL(0);
  tile_of(vb, m0, n0);
  set_tile(m0, n0);
  fill(0, 0);
  fill(1, 32);                                           // nk >= 2 (launcher): two slabs in flight from the start
  row_stats(m0, 0);
  __syncthreads();                                       // (drains both slabs once; from here on the waits are by hand)
  TL(1);
  int tl_slot = 2;

  f32x16 acc[2][2], pacc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) { acc[i][j] = f32x16{0}; pacc[i][j] = f32x16{0}; }
  int ra_[2], rb_[2], sa[2], sb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ra_[i] = wm * 64 + i * 32 + l31; sa[i] = (ra_[i] >> 1) & 7;
    rb_[i] = wn * 64 + i * 32 + l31; sb[i] = (rb_[i] >> 1) & 7;
  }

  // ---- deferred epilogue of the PREVIOUS tile, slice ph of 8: rows i*32 + 8*rq + 4*half + (0..3), both j
  int pm0 = 0, pn0 = 0, prs = 0;
  float pbias[2] = {0.f, 0.f}, pcsum[2] = {0.f, 0.f};
  float res[2][4];
  auto epi_load = [&](auto PH) {                         // residual elements of the slice: requested before the MFMAs
    constexpr int ph = decltype(PH)::value, i = ph >> 2, rq = ph & 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = pn0 + wn * 64 + j * 32 + l31;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = pm0 + wm * 64 + i * 32 + 8 * rq + 4 * half + e;
        res[j][e] = (p.residual && row < p.M && col < p.N) ? p.residual[(size_t)row * p.ldr + col] : 0.f;
      }
    }
  };
  // values of the slice, final (bias / LayerNorm / ReLU / residual applied), parked until the k-step's barrier has
  // passed: every barrier drains vmcnt(0) for the LDS-DMA, so a store (or load) issued just BEFORE one would be waited
  // for at once -- stores go out right AFTER a barrier and have a whole k-step of MFMAs to complete
  float pend[2][4];
  auto epi_compute = [&](auto PH) {
    constexpr int ph = decltype(PH)::value, i = ph >> 2, rq = ph & 3;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rl = wm * 64 + i * 32 + 8 * rq + 4 * half + e;      // == acc_row(rq*4 + e, half) within the 32-row tile
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float x = pacc[i][j][rq * 4 + e];
        if (LN_IN) {
          const float mean = rowst[(prs * BM + rl) * 2], inv = rowst[(prs * BM + rl) * 2 + 1];
          x = fmaf(inv, fmaf(-mean, pcsum[j], x), pbias[j]);
        } else {
          x = x + pbias[j];
        }
        if (p.relu) x = fmaxf(x, 0.f);
        if (p.residual) x = x + res[j][e];
        pend[j][e] = x;
      }
    }
  };
  auto epi_flush = [&](auto PH) {
    constexpr int ph = decltype(PH)::value, i = ph >> 2, rq = ph & 3;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = pm0 + wm * 64 + i * 32 + 8 * rq + 4 * half + e;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = pn0 + wn * 64 + j * 32 + l31;
        if (row < p.M && col < p.N) p.y[(size_t)row * p.ldy + col] = pend[j][e];
      }
      if (STATS_OUT) {                                   // this wave's 64 columns of the row = 32 lanes x 2 j-tiles
        float s1 = pend[0][e] + pend[1][e], s2 = pend[0][e] * pend[0][e] + pend[1][e] * pend[1][e];
        s1 = xor16_sum(row16_sum(s1)); s2 = xor16_sum(row16_sum(s2));
        if (l31 == 0 && row < p.M) {
          float* so = p.stats_out + ((size_t)row * (p.N / 64) + (pn0 + wn * 64) / 64) * 2;
          so[0] = s1; so[1] = s2;
        }
      }
    }
  };
  auto retire = [&]() {                                  // the finished tile becomes "previous"
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) { pacc[i][j] = acc[i][j]; acc[i][j] = f32x16{0}; }
    pm0 = m0; pn0 = n0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
      pbias[j] = (p.bias && col < p.N) ? p.bias[col] : 0.f;
      pcsum[j] = (LN_IN && col < p.N) ? p.ln_colsum[col] : 0.f;
    }
  };

  // ---- the k loop, software-pipelined by hand (see the header of this file)
  int lda[2][4], ldb[2][4];                              // byte offsets inside a stage of this lane's fragment of group g
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      lda[i][g] = (ra_[i] * 32 + 4 * ((2 * g + half) ^ sa[i])) * 4;
      ldb[i][g] = (int)sizeof(TileG) / 2 + (rb_[i] * 32 + 4 * ((2 * g + half) ^ sb[i])) * 4;
    }
  f32x4 fa[2][2], fb[2][2];
  auto frag_read = [&](int set, int stage, int g) {
    const int base = stage * (int)sizeof(TileG);
    // (no "memory" clobber: with one, hipcc assumes the asm may read what a pending LDS-DMA writes and drains vmcnt(0)
    // in front of it; the statements are volatile, which keeps their order among themselves and around the barrier)
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[set][i]) : "v"(base + lda[i][g]));
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(fb[set][j]) : "v"(base + ldb[j][g]));
  };
  auto frag_wait4 = [&](int set) {                       // the four reads issued BEFORE the newest four have returned
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fb[set][0]), "+v"(fb[set][1]));
  };
  auto group_mfma = [&](int set) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fa[set][i][s], fb[set][j][s], acc[i][j]);
  };
  int buf = 0, rs = 0;                                   // buf: stage of the slab being multiplied
  bool have_prev = false;
  int nm0 = 0, nn0 = 0;
  bool has_next = false;
  auto stage_after = [](int st, int n) { st += n; return st >= 3 ? st - 3 : st; };
  // top of k-step kt: request slab kt + 2 (of this tile, or slab 0 / 1 of the next tile) into the stage freed last step
  auto kstep_begin = [&](int kt) {
    const int tgt = stage_after(buf, 2);
    if (kt + 2 < nk) {
      fill(tgt, (kt + 2) * 32);
    } else if (has_next) {
      if (kt + 2 == nk) set_tile(nm0, nn0);              // this tile's slabs are all requested: switch the pointers
      fill(tgt, (kt + 2 - nk) * 32);
    }
  };
  // groups 0..3 of the slab in stage `buf`; entering, the fragments of group 0 are already requested into set 0
  auto kstep_mfma = [&]() {
    frag_read(1, buf, 1);
    frag_wait4(0);
    group_mfma(0);
    frag_read(0, buf, 2);
    frag_wait4(1);
    group_mfma(1);
    frag_read(1, buf, 3);
    frag_wait4(0);
    group_mfma(0);
    // last group: its fragments (and every earlier LDS read of this stage) are waited for, the next slab must have
    // landed (all VMEM operations but the newest 8 -- the DMA requested at the top of this step), everyone meets, and
    // the next slab's first fragments are requested under the cover of this group's MFMAs
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" : "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fb[1][0]), "+v"(fb[1][1]));
    __builtin_amdgcn_s_barrier();
    buf = stage_after(buf, 1);
    frag_read(0, buf, 0);                                // (unconditional: after the last slab it reads a stale stage, unused)
    group_mfma(1);
  };
  frag_read(0, 0, 0);                                    // slab 0 has landed (the __syncthreads above)
  for (;;) {
    const int vbn = vb + G;
    has_next = vbn < nblk;
    if (has_next) {
      tile_of(vbn, nm0, nn0);
      row_stats(nm0, rs == 2 ? 0 : rs + 1);
    }
    static_for<8>([&](auto PH) {
      constexpr int ph = decltype(PH)::value;
      if (ph < nk) {
        kstep_begin(ph);                                 // DMA first: the vmcnt(8) of this step never waits for the epilogue
        if (ph > 0 && have_prev) epi_flush(std::integral_constant<int, (ph > 0 ? ph - 1 : 0)>{});
        if (have_prev) epi_load(PH);
        kstep_mfma();
        if (have_prev) epi_compute(PH);
      } else if (have_prev) {
        if (ph == nk) epi_flush(std::integral_constant<int, (ph > 0 ? ph - 1 : 0)>{});
        epi_load(PH);
        epi_compute(PH);
        epi_flush(PH);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    if (have_prev && nk >= 8) epi_flush(std::integral_constant<int, 7>{});
    for (int kt = 8; kt < nk; ++kt) {
      kstep_begin(kt);
      kstep_mfma();
    }
    TL(tl_slot); ++tl_slot;
    prs = rs;
    retire();
    have_prev = true;
    if (!has_next) break;
    vb = vbn; m0 = nm0; n0 = nn0; rs = rs == 2 ? 0 : rs + 1;
  }
  // the last tile's epilogue is the only exposed one
  static_for<8>([&](auto PH) {
    epi_load(PH);
    epi_compute(PH);
    epi_flush(PH);
    __builtin_amdgcn_sched_barrier(0);                   // one slice in flight: keeps the register budget of the main loop
  });
#ifdef VCR_TIMELINE
  __builtin_amdgcn_s_waitcnt(0);
  TL(tl_slot);
#endif
}

