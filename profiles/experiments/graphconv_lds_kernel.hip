// Experiment, NOT in the build (end of round 3): LPDNet's convSN1 as ONE launch -- the workgroup that stages a 32-channel
// slice of a cloud's P rows for the LDS gathers computes that slice itself (and the Q rows tile by tile), so that P | Q
// never reach HBM.  Correct (max|diff| 2.9e-6 against vcr_linear_f32 + vcr_gathermax_f32: the 16x16x4 chain rounds
// differently from the linear's 32x32x2 one) and NOT faster: 32 clouds x 1024 points, k = 20, C = 256: 75.9 us against
// 75.5 us for the two launches (81.0 before the neighbour indices went through LDS slots); 48 x 768: 107.8 against 90.9.
// Two waves per SIMD run their 2 x 512 MFMAs in lockstep phases, and no gather can start before the whole cloud's P
// slice exists.  Kept as a record; it dropped into edgeconv.hip next to gathermax_lds_kernel, with
//   typedef struct { const float* x; int ldx; int K; const float* wpq; const float* bpq; int C;
//                    const int32_t* idx; int k; int M; int n_per_cloud; float* y; int ldy; } vcr_graphconv_args;

// A whole graph convolution with a single conv (LPDNet's convSN1, lpdnet_model.py:129-132, after the neighbour / centre
// split): y[i] = relu(max_j (x[nbr_j] Wp^T + bp) + x[i] Wq^T + bq).  The workgroup that stages a 32-channel slice of a
// cloud's P rows for the LDS gathers (gathermax_lds_kernel) COMPUTES that slice itself -- a [N x 128] x [128 x 32] product
// on v_mfma_f32_16x16x4_f32 straight into LDS -- and the Q rows tile by tile in the accumulator layout while it gathers:
// P | Q never reach HBM (vcr_linear_f32 wrote and vcr_gathermax_f32 re-read 134 MB per step at BASELINE configs[1]) and
// one launch goes.  Same arithmetic as vcr_linear_f32 + vcr_gathermax_f32: a k-ascending fma chain from 0, then the bias.
// Wave w takes the 16-point tiles w, w + 8, ...; lane (q4, l15): A[point l15][k = 4 s + q4] (16-B chunks of the row,
// transposed between lane rows as in the kNN kernel), B[k][channel 16 j + l15]; D[point 4 q4 + r][channel 16 j + l15].
template <int KQ>
__global__ __launch_bounds__(512, 1) void graphconv_lds_kernel(vcr_graphconv_args p, int slices) {
  constexpr int CS = 32, PITCH = CS + 1, K = 128, NST = K / 4;   // (dword accesses only: an odd pitch spreads the rows over the banks)
  extern __shared__ __attribute__((aligned(16))) float gc_smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int q4 = lane >> 4, l15 = lane & 15;
  const int cloud = (int)blockIdx.x / slices, sl = (int)blockIdx.x - cloud * slices;
  const int N = p.n_per_cloud, c0 = sl * CS;
  const size_t base = (size_t)cloud * N;
  const int ntiles = (N + 15) / 16;
  auto transpose4 = [](f32x4& c) {                       // chunk components <-> lane rows (see knn64c_body)
    int r0 = __float_as_int(c[0]), r1 = __float_as_int(c[1]), r2 = __float_as_int(c[2]), r3 = __float_as_int(c[3]);
    auto s01 = __builtin_amdgcn_permlane16_swap(r0, r1, false, false); r0 = s01[0]; r1 = s01[1];
    auto s23 = __builtin_amdgcn_permlane16_swap(r2, r3, false, false); r2 = s23[0]; r3 = s23[1];
    auto s02 = __builtin_amdgcn_permlane32_swap(r0, r2, false, false); r0 = s02[0]; r2 = s02[1];
    auto s13 = __builtin_amdgcn_permlane32_swap(r1, r3, false, false); r1 = s13[0]; r3 = s13[1];
    c = f32x4{__int_as_float(r0), __int_as_float(r1), __int_as_float(r2), __int_as_float(r3)};
  };
  // operand rows: 8 chunks of 16 floats; lane row q4 loads floats 16 g + 4 q4 .. + 3, after the transpose component e of
  // chunk g is element 4 (4 g + e) + q4 = the operand of MFMA step 4 g + e
  auto load_row = [&](const float* row, f32x4* c) {
#pragma unroll
    for (int g = 0; g < 8; ++g) c[g] = ld4(row + 16 * g + 4 * q4);
  };
  f32x4 wf[2][8];                                        // B fragments of the current half (P, then Q)
  auto load_weights = [&](int row0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      load_row(p.wpq + (size_t)(row0 + c0 + 16 * j + l15) * K, wf[j]);
#pragma unroll
      for (int g = 0; g < 8; ++g) transpose4(wf[j][g]);
    }
  };
  auto tile_product = [&](const f32x4* a, f32x4* acc) {
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0] = mfma16(a[g][e], wf[0][g][e], acc[0]);
        acc[1] = mfma16(a[g][e], wf[1][g][e], acc[1]);
      }
  };
  static_assert(NST == 32, "8 chunks x 4 steps");
  // ---- phase 1: this slice of P for every point of the cloud -> LDS
  load_weights(0);
  float bia[2] = {p.bpq ? p.bpq[c0 + l15] : 0.f, p.bpq ? p.bpq[c0 + 16 + l15] : 0.f};
  f32x4 an[8];
  if (wave < ntiles) load_row(p.x + (base + min(16 * wave + l15, N - 1)) * p.ldx, an);
  for (int tile = wave; tile < ntiles; tile += 8) {
    f32x4 a[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) { a[g] = an[g]; transpose4(a[g]); }
    if (tile + 8 < ntiles) load_row(p.x + (base + min(16 * (tile + 8) + l15, N - 1)) * p.ldx, an);
    f32x4 acc[2];
    tile_product(a, acc);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 16 * tile + 4 * q4 + r;
        if (n < N) gc_smem[n * PITCH + 16 * j + l15] = acc[j][r] + bia[j];
      }
  }
  __syncthreads();
  // ---- phase 2: Q per tile in the accumulators, the k neighbour rows of P out of LDS, y = relu(max + Q)
  load_weights(p.C);
  bia[0] = p.bpq ? p.bpq[p.C + c0 + l15] : 0.f; bia[1] = p.bpq ? p.bpq[p.C + c0 + 16 + l15] : 0.f;
  using i32x4 = __attribute__((ext_vector_type(4))) int;
  // the tile's 16 x k neighbour indices go through a per-wave LDS slot, fetched one tile ahead with coalesced loads (read
  // straight from global they are a chain of L2 round trips: four per tile)
  int* islot = reinterpret_cast<int*>(gc_smem + (((size_t)N * PITCH + 3) & ~(size_t)3)) + wave * (16 * 4 * KQ);
  constexpr int IPL = (16 * 4 * KQ + 63) / 64;           // index words per lane and tile
  int inx[IPL];
  auto fetch_idx = [&](int tile) {
    const int32_t* src = p.idx + (base + 16 * tile) * (4 * KQ);
    const int lim = (min(N, 16 * tile + 16) - 16 * tile) * 4 * KQ;
#pragma unroll
    for (int u = 0; u < IPL; ++u) inx[u] = lane + 64 * u < lim ? src[lane + 64 * u] : 0;
  };
  if (wave < ntiles) { load_row(p.x + (base + min(16 * wave + l15, N - 1)) * p.ldx, an); fetch_idx(wave); }
  for (int tile = wave; tile < ntiles; tile += 8) {
    f32x4 a[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) { a[g] = an[g]; transpose4(a[g]); }
#pragma unroll
    for (int u = 0; u < IPL; ++u) if (lane + 64 * u < 16 * 4 * KQ) islot[lane + 64 * u] = inx[u];
    if (tile + 8 < ntiles) { load_row(p.x + (base + min(16 * (tile + 8) + l15, N - 1)) * p.ldx, an); fetch_idx(tile + 8); }
    f32x4 acc[2];
    tile_product(a, acc);
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // the slot's writes have landed (lgkmcnt 0); one wave: no barrier needed
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = 16 * tile + 4 * q4 + r;
      const i32x4* id = reinterpret_cast<const i32x4*>(islot + (4 * q4 + r) * (4 * KQ));
      float m0 = VCR_NEG_INF, m1 = VCR_NEG_INF;
#pragma unroll
      for (int u = 0; u < KQ; ++u) {
        const i32x4 q = id[u];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          m0 = fmaxf(m0, gc_smem[q[e] * PITCH + l15]);
          m1 = fmaxf(m1, gc_smem[q[e] * PITCH + 16 + l15]);
        }
      }
      if (n < N) {
        float* o = p.y + (base + n) * p.ldy + c0 + l15;
        o[0] = fmaxf(m0 + (acc[0][r] + bia[0]), 0.f);
        o[16] = fmaxf(m1 + (acc[1][r] + bia[1]), 0.f);
      }
    }
    __builtin_amdgcn_wave_barrier();                     // the slot is rewritten by the next tile
  }
}

extern "C" int vcr_graphconv_f32(const vcr_graphconv_args* a, vcr_stream_t stream) {
  if (!a || !a->x || !a->wpq || !a->idx || !a->y) return VCR_EINVAL;
  if (a->M <= 0 || a->k <= 0 || a->C <= 0 || a->n_per_cloud <= 0 || (a->M % a->n_per_cloud) || a->ldx < a->K || a->ldy < a->C) return VCR_EINVAL;
  const int N = a->n_per_cloud;
  if (a->K != 128 || (a->C & 31) || (a->k != 20 && a->k != 40) || (size_t)N * 36 * 4 > 150 * 1024 || (a->ldx & 3) ||
      (((uintptr_t)a->x | (uintptr_t)a->wpq | (uintptr_t)a->idx) & 15))
    return VCR_EUNSUPPORTED;                               // (vcr_linear_f32 + vcr_gathermax_f32 serve every other shape)
  const int slices = a->C / 32, clouds = a->M / N;
  const size_t lds = (((size_t)N * 33 + 3) & ~(size_t)3) * 4 + (size_t)8 * 16 * a->k * 4;   // the P slice + eight index slots
  const dim3 grid(clouds * slices);
  if (a->k == 20) { VCR_DYN_LDS(graphconv_lds_kernel<5>, lds); hipLaunchKernelGGL(graphconv_lds_kernel<5>, grid, dim3(512), lds, (hipStream_t)stream, *a, slices); }
  else { VCR_DYN_LDS(graphconv_lds_kernel<10>, lds); hipLaunchKernelGGL(graphconv_lds_kernel<10>, grid, dim3(512), lds, (hipStream_t)stream, *a, slices); }
  return VCR_LAUNCH_RC();
}

