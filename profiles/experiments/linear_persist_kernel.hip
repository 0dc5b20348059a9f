// EXPERIMENT RECORD -- not part of the build (moved out of vcr-net_amd/csrc/linear.hip in round 3).
//
// linear_persist_kernel: persistent workgroups (two per CU) that walk over their 128x128 tiles with the LDS-DMA slab
// pipeline running across tile boundaries and the epilogue of tile i issued from the accumulator registers under the
// first eight k-steps of tile i+1.  Bit-identical results to linear_glds_kernel, measured SLOWER on every shape of the
// path (profiles/rounds1-3/r2b_bench_linear_shapes.txt: qkv 479 vs 402 us, wo 156 vs 148, ffn2 281 vs 273; DESIGN.md 5.1 item 1:
// persistent workgroups of equal work stay in phase and share the matrix pipe the whole time).  Known defect, never
// fixed because the kernel was retired: with K < 256 and ln_stats_in the 3-slot rowst ring is read by the trailing
// "short K" epilogue slices without a barrier before the next tile's row_stats() overwrites the slot.
// It compiled inside linear.hip's anonymous namespace (TileG, glds16, BM/BN, mfma32, acc_row, row16_sum, xcd_chunk, TL).
// The variant-128 ("every X tile from the first 256 rows") and variant-512 ("fragments of group g+1 requested before
// the MFMAs of g") timing experiments of DESIGN.md 5.1 items 2 and 4 lived in the same file under -DVCR_TIMELINE.

// ---- persistent variant with a DEFERRED epilogue (round 2).
// Measured on the round-1 kernels: the MFMA loop itself runs at 0.95 of the matrix peak (ffn2 vs wo: 3.6 us per
// 32-wide k-slab against 3.41 at peak), but every "round" of co-resident workgroups pays ~20 us on top -- all
// workgroups start together, run in lockstep and reach their epilogues at the same moment, so 33-67 MB of stores
// (+ residual reads) hit the memory system while no MFMA work is available anywhere on the chip.  25 rounds per
// forward = 0.5 of the 3.0 ms the linear family took.
// Here a workgroup walks over its tiles (grid = 2 per CU), the LDS-DMA slab pipeline runs straight across tile
// boundaries (the first slab of the next tile is requested during the last k-step of the current one), and the
// epilogue of tile i is issued from the accumulator registers in eight slices DURING the first eight k-steps of tile
// i+1: bias / LayerNorm / ReLU / residual / store / row statistics all ride under the next tile's MFMAs.  Only the
// last tile of a workgroup pays for its epilogue.  No LDS transpose: a lane stores its accumulator elements directly
// (32 lanes x 4 B = one full 128-B line per row), which also frees the LDS slice the transposed epilogue needed.
// The k order inside a tile is the one of the kernels above, so the GEMM results are bit-identical to theirs; the row
// statistics are summed in a different (still fixed) order.
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }   // f(0) .. f(N-1), index a constant

__device__ __forceinline__ float xor16_sum(float v) {    // + the value 16 lanes away (within each 32-lane half)
  return v + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}

template <bool LN_IN, bool STATS_OUT>
__global__ __launch_bounds__(256, 2) void linear_persist_kernel(vcr_linear_args p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  TileG* tile = reinterpret_cast<TileG*>(smem);          // [2]
  float* rowst = reinterpret_cast<float*>(smem + 2 * sizeof(TileG));   // [3][BM][2] (mean, inv): previous / current / next tile
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int nblk = tiles_m * tiles_n, G = gridDim.x;
  const int nk = p.K / 32;

  // virtual block id -> tile, XCD-aware: ids congruent mod 8 run on one XCD (G is a multiple of 8 or == nblk), and
  // every XCD owns a contiguous run of tiles, so tiles sharing an X panel share an L2
  auto tile_of = [&](int vb, int& m0, int& n0) {
    const int bid = xcd_chunk(vb, nblk);
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
  row_stats(m0, 0);
  __syncthreads();
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

  int buf = 0, rs = 0;
  bool have_prev = false;
  int nm0 = 0, nn0 = 0;
  bool has_next = false;
  // one k-step: request the next slab (possibly the next tile's first), MFMAs on the current one
  auto kstep_begin = [&](int kt) {
    if (kt + 1 < nk) {
      fill(buf ^ 1, (kt + 1) * 32);
    } else if (has_next) {                               // last k-step: the slab pipeline crosses into the next tile
      set_tile(nm0, nn0);
      fill(buf ^ 1, 0);
    }
  };
  auto kstep_mfma = [&]() {
    const TileG& T = tile[buf];
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
  };
  auto kstep_end = [&]() {
    __syncthreads();                                     // drains the LDS-DMA (vmcnt(0)) and orders the buffers
    buf ^= 1;
  };
  for (;;) {
    const int vbn = vb + G;
    has_next = vbn < nblk;
    if (has_next) {
      tile_of(vbn, nm0, nn0);
      row_stats(nm0, rs == 2 ? 0 : rs + 1);              // next tile's LayerNorm rows: ready long before its epilogue
    }
    // the first eight k-steps carry the previous tile's epilogue, one slice each (straight-line code: the slices'
    // addresses must not become loop invariants that the compiler keeps live across the whole k loop)
    static_for<8>([&](auto PH) {
      constexpr int ph = decltype(PH)::value;
      if (ph < nk) {                                     // uniform
        if (ph > 0 && have_prev) epi_flush(std::integral_constant<int, (ph > 0 ? ph - 1 : 0)>{});
        kstep_begin(ph);
        if (have_prev) epi_load(PH);
        kstep_mfma();
        if (have_prev) epi_compute(PH);
        kstep_end();
      } else if (have_prev) {                            // short K: a slice that found no k-step to hide under
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
      kstep_end();
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


