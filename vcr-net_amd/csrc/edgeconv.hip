// Kernel 2: EdgeConv on the kNN graph (model/lpdnet_model.py:122-132, util/util.py:176-199).
//
// The reference materialises [B, 2C, N, k] = cat(x_j, x_i) and runs 1x1 convs over it.  Because the
// conv is 1x1, W [x_j; x_i] + b = Wn x_j + (Wc x_i + b)  (SURVEY F7), so the first conv of each
// EdgeConv is a per-POINT linear (P = X Wn^T, Q = X Wc^T + b, done by vcr_linear_f32) followed by a
// neighbour gather; only convDG2, which acts on the post-ReLU per-EDGE features, is a real N*k GEMM.
//
// edgeconv_dg_kernel (convDG1 -> max -> convDG2 -> max):
//   per point i: H[j] = relu(P[nbr_ij] + Q[i])  (k rows of 128, padded to 32-row MFMA tiles by
//   repeating the last neighbour, which cannot change a max);  x1[i] = max_j H[j];
//   Y = H W2^T on v_mfma_f32_32x32x2_f32 with EDGES as MFMA rows, so the max over a point's edges is a
//   max over accumulator registers (+ one cross-half exchange);  x2[i] = relu(max_j Y[j] + b2).
//   A 4-wave block shares one H tile in LDS (double buffered, the next point's gather is in flight
//   during the MFMAs); wave w owns output channels 32w..32w+31 and keeps its 32x128 slice of W2 in
//   64 VGPRs for the whole kernel.
// gathermax_kernel (convSN1 -> max): y[i] = relu(max_j P[nbr_ij] + Q[i]); one wave per point,
//   whole 16-B-per-lane row loads, L2-bound.
#include "common.h"

namespace {

constexpr int HP = 132;  // H row pitch in floats: 528 B -> 16-lane ds_read_b128 groups conflict-free

__global__ __launch_bounds__(256, 2) void edgeconv_dg_kernel(vcr_edgeconv_args p) {
  __shared__ __attribute__((aligned(16))) float Hs[2][32][HP];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int rs = lane >> 3, cg = lane & 7;              // build mapping: rows rs+8i, channels 32w+4cg..+3
  const int ch = 32 * w + 4 * cg;

  f32x4 wf[16];                                          // W2[32w + l31][8g + 4 half + s]
#pragma unroll
  for (int g = 0; g < 16; ++g) wf[g] = ld4(p.w2 + (size_t)(32 * w + l31) * 128 + 8 * g + 4 * half);
  const float bias2 = p.b2[32 * w + l31];

  // this block's work items: points sl.first, +sl.stride, ... (inside its XCD's run of clouds); each point =
  // row_tiles consecutive items
  const int row_tiles = (p.k + 31) / 32;
  const xcd_slice_t sl = xcd_slice(p.M);
  if (sl.count == 0) return;
  const int my_pts = sl.count;
  const int n_items = my_pts * row_tiles;

  f32x4 hr[4];                                           // staged rows of the item being built
  auto gather = [&](int it) {
    const int pt = sl.first + (it / row_tiles) * sl.stride, rt = it % row_tiles;
    const int base = (pt / p.n_per_cloud) * p.n_per_cloud;
    const f32x4 q = ld4(p.pq + (size_t)pt * p.ldpq + 128 + ch);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = min(rt * 32 + rs + 8 * i, p.k - 1);
      const int nb = p.idx[(size_t)pt * p.k + r];
      const f32x4 v = ld4(p.pq + (size_t)(base + nb) * p.ldpq + ch) + q;
      hr[i] = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
    }
  };
  f32x4 x1m = f32x4{0.f, 0.f, 0.f, 0.f};                 // H >= 0, so 0 is the identity of this max
  auto commit = [&](int it, int buf) {                   // registers -> LDS, and the x1 running max
    const int pt = sl.first + (it / row_tiles) * sl.stride, rt = it % row_tiles;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      st4(&Hs[buf][rs + 8 * i][ch], hr[i]);
#pragma unroll
      for (int c = 0; c < 4; ++c) x1m[c] = fmaxf(x1m[c], hr[i][c]);
    }
    if (rt == row_tiles - 1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float v = x1m[c];
        v = fmaxf(v, __shfl_xor(v, 8, 64));
        v = fmaxf(v, __shfl_xor(v, 16, 64));
        v = fmaxf(v, __shfl_xor(v, 32, 64));
        x1m[c] = v;
      }
      if (rs == 0) st4(p.x1 + (size_t)pt * p.ldx1 + ch, x1m);
      x1m = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  gather(0);
  commit(0, 0);
  __syncthreads();
  int cur = 0;
  float x2m = VCR_NEG_INF;
  for (int item = 0; item < n_items; ++item) {
    const int nxt = item + 1;
    if (nxt < n_items) gather(nxt);
    f32x16 acc = {0};
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const f32x4 af = ld4(&Hs[cur][l31][8 * g + 4 * half]);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = mfma32(af[s], wf[g][s], acc);
    }
    float m = acc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
    x2m = fmaxf(x2m, m);
    const int pt = sl.first + (item / row_tiles) * sl.stride, rt = item % row_tiles;
    if (rt == row_tiles - 1) {
      const float v = fmaxf(x2m, xhalf(x2m));
      if (half == 0) p.x2[(size_t)pt * p.ldx2 + 32 * w + l31] = fmaxf(v + bias2, 0.f);
      x2m = VCR_NEG_INF;
    }
    if (nxt < n_items) commit(nxt, cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
}

// Packed variant for k = 20 / 40 (the path's values): instead of padding each point's k edge rows to a
// multiple of 32, G consecutive points (G*k = 160 rows = 5 MFMA tiles exactly; G = 8 or 4) share the tiles,
// which removes 37.5 % of the MFMA work.  A tile then spans several points: the row -> point map is a
// compile-time function of (tile, accumulator register, lane half), so the max over a point's edges folds
// into G per-point registers with static indexing; x1 (max over H rows) is collected by a column pass over the
// staged tile with the same static map.
template <int KE>
__global__ __launch_bounds__(256, 2) void edgeconv_dg_packed_kernel(vcr_edgeconv_args p) {
  constexpr int G = 160 / KE;                            // points per group
  __shared__ __attribute__((aligned(16))) float Hs[2][32][HP];
  __shared__ __attribute__((aligned(16))) float x1half[2][G][128];   // per-point column maxima of the two row halves
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int rs = lane >> 3, cg = lane & 7;
  const int ch = 32 * w + 4 * cg;
  const int colc = threadIdx.x & 127, rowh = threadIdx.x >> 7;       // x1 pass: one channel, 16 rows of the tile

  f32x4 wf[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) wf[g] = ld4(p.w2 + (size_t)(32 * w + l31) * 128 + 8 * g + 4 * half);
  const float bias2 = p.b2[32 * w + l31];

  const int ngroups = (p.M + G - 1) / G;
  const xcd_slice_t sl = xcd_slice(ngroups);             // this block's groups, all inside its XCD's run of clouds
  if (sl.count == 0) return;
  const int my_groups = sl.count;

  f32x4 hr[4];
  auto gather = [&](int grp, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = 32 * t + rs + 8 * i;                 // edge row inside the group, 0..159
      const int pl = e / KE, j = e - pl * KE;
      const int pt = min(grp * G + pl, p.M - 1);
      const int base = (pt / p.n_per_cloud) * p.n_per_cloud;
      const int nb = p.idx[(size_t)pt * KE + j];
      const f32x4 v = ld4(p.pq + (size_t)(base + nb) * p.ldpq + ch) + ld4(p.pq + (size_t)pt * p.ldpq + 128 + ch);
      hr[i] = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) st4(&Hs[buf][rs + 8 * i][ch], hr[i]);
  };

  gather(sl.first, 0);
  commit(0);
  __syncthreads();
  int cur = 0;
  for (int gi = 0; gi < my_groups; ++gi) {
    const int grp = sl.first + gi * sl.stride;
    const bool more = gi + 1 < my_groups;
    float pm[G], cm[G];                                  // per-point maxima: x2 (MFMA rows) and x1 (this thread's channel)
#pragma unroll
    for (int q = 0; q < G; ++q) { pm[q] = VCR_NEG_INF; cm[q] = 0.f; }   // H rows are >= 0 after ReLU
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const bool has_next = t < 4 || more;
      if (has_next) gather(t < 4 ? grp : grp + sl.stride, t < 4 ? t + 1 : 0);
      // x1 = max over a point's H rows: a conflict-free column pass over the staged tile (the row -> point map is
      // static per tile and row half), instead of LDS atomics that collide on (point, channel)
      if (rowh == 0) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) cm[(32 * t + rr) / KE] = fmaxf(cm[(32 * t + rr) / KE], Hs[cur][rr][colc]);
      } else {
#pragma unroll
        for (int rr = 16; rr < 32; ++rr) cm[(32 * t + rr) / KE] = fmaxf(cm[(32 * t + rr) / KE], Hs[cur][rr][colc]);
      }
      f32x16 acc = {0};
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const f32x4 af = ld4(&Hs[cur][l31][8 * g + 4 * half]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma32(af[e], wf[g][e], acc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {                     // fold the tile into the per-point maxima (static map)
        const int row0 = 32 * t + acc_row(r, 0), row1 = row0 + 4;
        const int p0 = row0 / KE, p1 = row1 / KE;
        if (p0 == p1) {
          pm[p0] = fmaxf(pm[p0], acc[r]);
        } else {
          pm[p0] = fmaxf(pm[p0], half ? VCR_NEG_INF : acc[r]);
          pm[p1] = fmaxf(pm[p1], half ? acc[r] : VCR_NEG_INF);
        }
      }
      if (t == 4) {
        // the group's five tiles are done: emit x2 from the MFMA maxima, x1 from the two row halves' column maxima
#pragma unroll
        for (int q = 0; q < G; ++q) {
          const int pt = grp * G + q;
          const float v = fmaxf(pm[q], xhalf(pm[q]));
          if (half == 0 && pt < p.M) p.x2[(size_t)pt * p.ldx2 + 32 * w + l31] = fmaxf(v + bias2, 0.f);
          x1half[rowh][q][colc] = cm[q];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < G * 32; i += 256) {
          const int q = i >> 5, c4 = (i & 31) * 4, pt = grp * G + q;
          const f32x4 lo = ld4(&x1half[0][q][c4]), hi = ld4(&x1half[1][q][c4]);
          if (pt < p.M)
            st4(p.x1 + (size_t)pt * p.ldx1 + c4, f32x4{fmaxf(lo[0], hi[0]), fmaxf(lo[1], hi[1]), fmaxf(lo[2], hi[2]),
                                                        fmaxf(lo[3], hi[3])});
        }
      }
      if (has_next) commit(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }
}

// The packed kernel again, with every wave's instruction stream laid out by hand for the matrix pipe it shares.
//
// Measured (profiles/rounds4-5/r4f_mfma_valu_coissue.txt): beside a wave that issues fp32 MFMAs back to back, another wave of the
// same SIMD gets ONE vector / LDS instruction through per ~20 cycles (alone: 2.5-6), while a wave's OWN vector
// instructions issue freely in the 64-cycle shadow of its own MFMA.  The kernel above runs two workgroups per CU; per tile
// a wave has 64 MFMAs (4096 cycles of the pipe) and ~200 other instructions, which hipcc groups in front of and behind
// the MFMA block -- where they crawl beside the partner's MFMAs, ~4000 cycles: the pipe is busy 0.73 of the time.  Here a
// tile is 16 CHUNKS fenced by sched_barrier(0) -- one k-group each: the next group's fragment read, four MFMAs, and a
// sixteenth of everything else --, which takes straight-line tiles:
//   * the index loads run two tiles ahead and the row loads one tile ahead of the tile being multiplied, without
//     conditions (past the block's last group they re-read it);
//   * the x1 column pass splits the tile's rows by PARITY between the two thread halves: k is even, so rows 2m and 2m+1
//     belong to the same point and both halves follow one compile-time row -> point map (no branch); one row per chunk,
//     folded one chunk after it was requested;
//   * the previous tile's accumulator is folded into the per-point maxima one register per chunk (two accumulators).
// Results are bit-identical to the kernel above (same MFMA order per output; max is exact).
constexpr int FRAG_AHEAD = 1;                            // 1 or 2 (profiles/rounds4-5/r4i_edgeconv_ab.txt)
template <int KE>
__global__ __launch_bounds__(256, 2) void edgeconv_dg_pipe_kernel(vcr_edgeconv_args p) {
  static_assert(KE % 2 == 0 && 160 % KE == 0, "parity split of the x1 pass");
  constexpr int G = 160 / KE;                            // points per group
  __shared__ __attribute__((aligned(16))) float Hs[2][32][HP];
  __shared__ __attribute__((aligned(16))) float x1half[2][G][128];   // per-point column maxima of the two row parities
  __shared__ __attribute__((aligned(16))) float x2half[2][G][128];   // per-point maxima of the MFMA rows held by the two lane halves
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int rs = lane >> 3, cg = lane & 7;
  const int ch = 32 * w + 4 * cg;
  const int colc = threadIdx.x & 127, rowp = threadIdx.x >> 7;       // x1 pass: one channel, the 16 rows of one parity

  f32x4 wf[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) wf[g] = ld4(p.w2 + (size_t)(32 * w + l31) * 128 + 8 * g + 4 * half);
  const f32x4 bias4 = ld4(p.b2 + 4 * (threadIdx.x & 31));           // the group-end pass: this thread's four channels

  const int ngroups = (p.M + G - 1) / G;
  const xcd_slice_t sl = xcd_slice(ngroups);
  if (sl.count == 0) return;
  const int my_groups = sl.count;
  const int grp_last = sl.first + (my_groups - 1) * sl.stride;

  int nb[4], nbn[4];                                     // neighbour indices of the next tile / the one after
  f32x4 hr[4], hq[4];                                    // the next tile's gathered P rows and centre Q rows
  auto load_idx1 = [&](int grp, int t, int i, int (&dst)[4]) {
    const int e = 32 * t + rs + 8 * i;
    const int pl = e / KE, j = e - pl * KE;
    const int pt = min(grp * G + pl, p.M - 1);
    dst[i] = p.idx[(size_t)pt * KE + j];
  };
  auto load_row1 = [&](int grp, int t, int i) {
    const int e = 32 * t + rs + 8 * i;
    const int pt = min(grp * G + e / KE, p.M - 1);
    const int base = (pt / p.n_per_cloud) * p.n_per_cloud;
    hr[i] = ld4(p.pq + (size_t)(base + nb[i]) * p.ldpq + ch);
    hq[i] = ld4(p.pq + (size_t)pt * p.ldpq + 128 + ch);
  };
  auto commit1 = [&](int buf, int i) {
    const f32x4 v = hr[i] + hq[i];
    st4(&Hs[buf][rs + 8 * i][ch], f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)});
  };
  float pm[G], cm[G];                                    // per-point maxima: x2 (MFMA rows) and x1 (this thread's channel)
  auto fold1 = [&](const f32x16& a, int t, int r) {      // accumulator register r of tile t -> the per-point maxima
    const int row0 = 32 * t + acc_row(r, 0), row1 = row0 + 4;
    const int p0 = row0 / KE, p1 = row1 / KE;
    if (p0 == p1) {
      pm[p0] = fmaxf(pm[p0], a[r]);
    } else {
      pm[p0] = fmaxf(pm[p0], half ? VCR_NEG_INF : a[r]);
      pm[p1] = fmaxf(pm[p1], half ? a[r] : VCR_NEG_INF);
    }
  };

#pragma unroll
  for (int i = 0; i < 4; ++i) load_idx1(sl.first, 0, i, nb);
#pragma unroll
  for (int i = 0; i < 4; ++i) load_row1(sl.first, 0, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) load_idx1(sl.first, 1, i, nbn);
#pragma unroll
  for (int i = 0; i < 4; ++i) commit1(0, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) nb[i] = nbn[i];
  lds_barrier();
  int cur = 0;
  for (int gi = 0; gi < my_groups; ++gi) {
    const int grp = sl.first + gi * sl.stride;
    const int grp_n = min(grp + sl.stride, grp_last);    // (the block's last group is re-read instead of branching)
#pragma unroll
    for (int q = 0; q < G; ++q) { pm[q] = VCR_NEG_INF; cm[q] = 0.f; }
    f32x16 accp = {0};                                   // the previous tile's accumulator (folded during this tile)
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const float* hcol = &Hs[cur][rowp][colc];
      const float* hrow = &Hs[cur][l31][4 * half];
      f32x16 acc = {0};
      f32x4 af[3];                                       // fragments are requested FRAG_AHEAD k-groups ahead
      float xv[3];
#pragma unroll
      for (int g = 0; g < FRAG_AHEAD; ++g) { af[g] = ld4(hrow + 8 * g); xv[g] = hcol[2 * g * HP]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        if (g + FRAG_AHEAD < 16) {
          af[(g + FRAG_AHEAD) % 3] = ld4(hrow + 8 * (g + FRAG_AHEAD));
          xv[(g + FRAG_AHEAD) % 3] = hcol[2 * (g + FRAG_AHEAD) * HP];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma32(af[g % 3][e], wf[g][e], acc);
        cm[(32 * t + 2 * g) / KE] = fmaxf(cm[(32 * t + 2 * g) / KE], xv[g % 3]);     // x1 pass, row 2g + parity
        if (t > 0) fold1(accp, t - 1, g);                                           // previous tile, register g
        if (g < 4) load_row1(t < 4 ? grp : grp_n, (t + 1) % 5, g);                  // tile s+1: rows (indices: tile s-1)
        else if (g < 8) load_idx1(t < 3 ? grp : grp_n, (t + 2) % 5, g - 4, nbn);    // tile s+2: indices
        else if (g >= 10 && g < 14) commit1(cur ^ 1, g - 10);                       // tile s+1 -> the other LDS buffer
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) nb[i] = nbn[i];
      if (t == 4) {
#pragma unroll
        for (int r = 0; r < 16; ++r) fold1(acc, 4, r);   // (the group's last tile: nothing left to hide it under)
        // group end: both maxima leave through LDS -- no cross-half shuffles, no per-point 4-B stores: one 16-B store of
        // x1 and one of x2 per thread
#pragma unroll
        for (int q = 0; q < G; ++q) {
          x2half[half][q][32 * w + l31] = pm[q];
          x1half[rowp][q][colc] = cm[q];
        }
        lds_barrier();
        for (int i = threadIdx.x; i < G * 32; i += 256) {
          const int q = i >> 5, c4 = (i & 31) * 4, pt = grp * G + q;
          if (pt < p.M) {
            const f32x4 lo = ld4(&x1half[0][q][c4]), hi = ld4(&x1half[1][q][c4]);
            st4(p.x1 + (size_t)pt * p.ldx1 + c4, f32x4{fmaxf(lo[0], hi[0]), fmaxf(lo[1], hi[1]), fmaxf(lo[2], hi[2]),
                                                        fmaxf(lo[3], hi[3])});
            const f32x4 a0 = ld4(&x2half[0][q][c4]), a1 = ld4(&x2half[1][q][c4]);
            st4(p.x2 + (size_t)pt * p.ldx2 + c4,
                f32x4{fmaxf(fmaxf(a0[0], a1[0]) + bias4[0], 0.f), fmaxf(fmaxf(a0[1], a1[1]) + bias4[1], 0.f),
                      fmaxf(fmaxf(a0[2], a1[2]) + bias4[2], 0.f), fmaxf(fmaxf(a0[3], a1[3]) + bias4[3], 0.f)});
          }
        }
      }
      accp = acc;
      lds_barrier();
      cur ^= 1;
    }
  }
}

constexpr int EDGECONV_PIPE = 1;                         // 1: the kernel above; 0: edgeconv_dg_packed_kernel (A/B builds: probe_build.py --set)

__global__ __launch_bounds__(256) void gathermax_kernel(vcr_gathermax_args p) {
  const int lane = threadIdx.x & 63;
  int pt = xcd_chunk((int)blockIdx.x, (int)gridDim.x) * 4 + (threadIdx.x >> 6);
  if (pt >= p.M) return;
  const int c = lane * 4;
  if (c >= p.C) return;
  const int base = (pt / p.n_per_cloud) * p.n_per_cloud;
  if (p.order) pt = base + p.order[pt];                  // (wave-uniform: the point at this rank of the cloud's Morton order)
  const int32_t* id = p.idx + (size_t)pt * p.k;
  f32x4 m;
  int j;
  if (p.k == 20 || p.k == 40) {
    // the path's k: all 20 neighbour rows in flight at once (the kernel is bound by the latency of these L2 gathers)
    f32x4 a[20];
#pragma unroll
    for (int u = 0; u < 20; ++u) a[u] = ld4(p.pq + (size_t)(base + id[u]) * p.ldpq + c);
    m = a[0];
#pragma unroll
    for (int u = 1; u < 20; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) m[i] = fmaxf(m[i], a[u][i]);
    j = 20;
    if (p.k == 40) {                                     // (BASELINE configs[4]: the second twenty likewise)
#pragma unroll
      for (int u = 0; u < 20; ++u) a[u] = ld4(p.pq + (size_t)(base + id[20 + u]) * p.ldpq + c);
#pragma unroll
      for (int u = 0; u < 20; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = fmaxf(m[i], a[u][i]);
      j = 40;
    }
  } else {
    m = ld4(p.pq + (size_t)(base + id[0]) * p.ldpq + c);
    j = 1;
  }
  for (; j + 3 < p.k; j += 4) {
    const f32x4 a0 = ld4(p.pq + (size_t)(base + id[j]) * p.ldpq + c);
    const f32x4 a1 = ld4(p.pq + (size_t)(base + id[j + 1]) * p.ldpq + c);
    const f32x4 a2 = ld4(p.pq + (size_t)(base + id[j + 2]) * p.ldpq + c);
    const f32x4 a3 = ld4(p.pq + (size_t)(base + id[j + 3]) * p.ldpq + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = fmaxf(fmaxf(fmaxf(m[i], a0[i]), fmaxf(a1[i], a2[i])), a3[i]);
  }
  for (; j < p.k; ++j) {
    const f32x4 a0 = ld4(p.pq + (size_t)(base + id[j]) * p.ldpq + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = fmaxf(m[i], a0[i]);
  }
  const f32x4 q = ld4(p.pq + (size_t)pt * p.ldpq + p.C + c);
  f32x4 y;
#pragma unroll
  for (int i = 0; i < 4; ++i) y[i] = fmaxf(m[i] + q[i], 0.f);
  st4(p.y + (size_t)pt * p.ldy + c, y);
}

// The same result with the gathers served by LDS instead of L2: a workgroup owns one cloud and a slice of CS channels,
// stages the slice of all the cloud's P rows ([N][CS], pitch CS + 4 floats: 16-B aligned, rows spread over the banks) with
// coalesced 16-B loads, and every point then picks its k neighbour rows out of LDS (CS / 4 lanes per point, one
// ds_read_b128 per neighbour).  gathermax_kernel moves N k C floats through L2 per cloud (655 MB per step at BASELINE
// configs[1], 19 TB/s: the launch is bound by exactly that); here L2 / HBM see every P row once, and the chip's LDS
// delivers ~4x the L2's gather rate.  Max is exact, so the result is bit-identical.
// (KQ = k / 4 index quads per point, read as 16-B loads one point AHEAD of the gathers that use them: the loop is a chain
// of L2 round trips otherwise)
template <int CS, int KQ>
__global__ __launch_bounds__(512, 1) void gathermax_lds_kernel(vcr_gathermax_args p, int slices) {
  constexpr int PITCH = CS + 4, LPP = CS / 4;            // floats per staged row; lanes per point
  extern __shared__ __attribute__((aligned(16))) float gm_smem[];
  const int t = threadIdx.x;
  const int cloud = (int)blockIdx.x / slices, sl = (int)blockIdx.x - cloud * slices;
  const int N = p.n_per_cloud, c0 = sl * CS;
  const size_t base = (size_t)cloud * N;
  const int sub = t % LPP, c4 = 4 * sub;
  if (c0 + c4 >= p.C) {                                  // (C not a multiple of CS: the lanes beyond it only attend the barrier)
    __syncthreads();
    return;
  }
  for (int n = t / LPP; n < N; n += 512 / LPP)
    st4(&gm_smem[n * PITCH + c4], ld4(p.pq + (base + n) * p.ldpq + c0 + c4));
  __syncthreads();
  using i32x4 = __attribute__((ext_vector_type(4))) int;
  i32x4 idn[KQ];
  f32x4 qn;
  auto fetch = [&](int n) {
    const i32x4* id = reinterpret_cast<const i32x4*>(p.idx + (base + n) * (4 * KQ));
#pragma unroll
    for (int u = 0; u < KQ; ++u) idn[u] = id[u];
    qn = ld4(p.pq + (base + n) * p.ldpq + p.C + c0 + c4);
  };
  int n = t / LPP;
  if (n < N) fetch(n);
  for (; n < N; n += 512 / LPP) {
    i32x4 idc[KQ];
#pragma unroll
    for (int u = 0; u < KQ; ++u) idc[u] = idn[u];
    const f32x4 q = qn;
    if (n + 512 / LPP < N) fetch(n + 512 / LPP);
    f32x4 m = ld4(&gm_smem[idc[0][0] * PITCH + c4]);
#pragma unroll
    for (int u = 0; u < KQ; ++u)
#pragma unroll
      for (int e = (u == 0 ? 1 : 0); e < 4; ++e) {
        const f32x4 a0 = ld4(&gm_smem[idc[u][e] * PITCH + c4]);
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = fmaxf(m[i], a0[i]);
      }
    f32x4 y;
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = fmaxf(m[i] + q[i], 0.f);
    st4(p.y + (base + n) * p.ldy + c0 + c4, y);
  }
}

// Per-edge feature rows for EdgeConv CHAINS (DGCNN, model/vcrnet_model.py:104-118): conv2..conv4 act on the
// post-ReLU per-edge tensor, so every layer is a true N*k GEMM (vcr_linear_f32 over [M*k, C] edge rows) and
// only the first conv enjoys the F7 split.  h[(i,j)] = relu(P[nbr_ij] + Q[i]).
__global__ __launch_bounds__(256) void edge_rows_kernel(vcr_edgerows_args p) {
  const int lane = threadIdx.x & 63;
  const long e = (long)xcd_chunk((int)blockIdx.x, (int)gridDim.x) * 4 + (threadIdx.x >> 6);
  if (e >= (long)p.M * p.k) return;
  const int pt = (int)(e / p.k);
  const int base = (pt / p.n_per_cloud) * p.n_per_cloud;
  const int nb = p.idx[e];
  for (int c = lane * 4; c < p.C; c += 256) {
    const f32x4 v = ld4(p.pq + (size_t)(base + nb) * p.ldpq + c) + ld4(p.pq + (size_t)pt * p.ldpq + p.C + c);
    st4(p.h + (size_t)e * p.ldh + c, f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)});
  }
}

// C == 64: one wave per POINT, four edge rows per step (16 lanes x 16 B each); the same pass writes the max over the
// point's k rows (x1, vcrnet_model.py:109) and zeroes the columns that the fused maxima of conv2..conv4 will
// accumulate into with atomic max (vcr_linear_args.segmax_out)
__global__ __launch_bounds__(256) void edge_rows64_kernel(vcr_edgerows_args p) {
  const int lane = threadIdx.x & 63;
  const int pt = xcd_chunk((int)blockIdx.x, (int)gridDim.x) * 4 + (threadIdx.x >> 6);
  if (pt >= p.M) return;
  const int base = (pt / p.n_per_cloud) * p.n_per_cloud;
  const int c = (lane & 15) * 4, rg = lane >> 4;
  const f32x4 q = ld4(p.pq + (size_t)pt * p.ldpq + 64 + c);
  f32x4 m = {0.f, 0.f, 0.f, 0.f};                        // post-ReLU values are >= 0
  for (int j0 = 0; j0 < p.k; j0 += 4) {
    const int j = j0 + rg;
    if (j < p.k) {
      const int nb = p.idx[(size_t)pt * p.k + j];
      const f32x4 v = ld4(p.pq + (size_t)(base + nb) * p.ldpq + c) + q;
      const f32x4 h = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
      st4(p.h + ((size_t)pt * p.k + j) * p.ldh + c, h);
#pragma unroll
      for (int i = 0; i < 4; ++i) m[i] = fmaxf(m[i], h[i]);
    }
  }
  if (p.ymax) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      m[i] = fmaxf(m[i], __shfl_xor(m[i], 16, 64));
      m[i] = fmaxf(m[i], __shfl_xor(m[i], 32, 64));
    }
    float* yr = p.ymax + (size_t)pt * p.ldymax;
    if (rg == 0) st4(yr + c, m);
    for (int z = 64 + lane * 4; z < p.zero_to; z += 256) st4(yr + z, f32x4{0.f, 0.f, 0.f, 0.f});
  }
}

// y[i] = max_j x[(i,j)]   (x.max(dim=-1) of vcrnet_model.py:109-118)
__global__ __launch_bounds__(256) void segmax_kernel(vcr_segmax_args p) {
  const int lane = threadIdx.x & 63;
  const int pt = xcd_chunk((int)blockIdx.x, (int)gridDim.x) * 4 + (threadIdx.x >> 6);
  if (pt >= p.M) return;
  for (int c = lane * 4; c < p.C; c += 256) {
    f32x4 m = ld4(p.x + ((size_t)pt * p.k) * p.ldx + c);
    for (int j = 1; j < p.k; ++j) {
      const f32x4 v = ld4(p.x + ((size_t)pt * p.k + j) * p.ldx + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) m[i] = fmaxf(m[i], v[i]);
    }
    st4(p.y + (size_t)pt * p.ldy + c, m);
  }
}

}  // namespace

extern "C" int vcr_edgerows_f32(const vcr_edgerows_args* a, vcr_stream_t stream) {
  if (!a || !a->pq || !a->idx || !a->h) return VCR_EINVAL;
  if (a->M <= 0 || a->k <= 0 || a->C <= 0 || (a->C & 3) || (a->ldpq & 3) || (a->ldh & 3) || a->ldpq < 2 * a->C) return VCR_EINVAL;
  if (a->n_per_cloud <= 0 || (a->M % a->n_per_cloud)) return VCR_EINVAL;
  if (a->ymax && (a->C != 64 || (a->ldymax & 3) || a->ldymax < a->zero_to || (a->zero_to & 3) ||
                  (a->zero_to && a->zero_to < 64))) return VCR_EINVAL;
  const long rows = (long)a->M * a->k;
  if (a->C == 64)
    hipLaunchKernelGGL(edge_rows64_kernel, dim3((unsigned)((a->M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a);
  else
    hipLaunchKernelGGL(edge_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a);
  return VCR_LAUNCH_RC();
}

extern "C" int vcr_segmax_f32(const vcr_segmax_args* a, vcr_stream_t stream) {
  if (!a || !a->x || !a->y || a->M <= 0 || a->k <= 0 || a->C <= 0 || (a->C & 3) || (a->ldx & 3) || (a->ldy & 3)) return VCR_EINVAL;
  hipLaunchKernelGGL(segmax_kernel, dim3((a->M + 3) / 4), dim3(256), 0, (hipStream_t)stream, *a);
  return VCR_LAUNCH_RC();
}

extern "C" int vcr_edgeconv_f32(const vcr_edgeconv_args* a, vcr_stream_t stream) {
  vcr_stream_scope bound(stream);
  if (!a || !a->pq || !a->idx || !a->w2 || !a->b2 || !a->x1 || !a->x2) return VCR_EINVAL;
  if (a->M <= 0 || a->k <= 0 || a->k > 64 || a->n_per_cloud <= 0 || (a->M % a->n_per_cloud)) return VCR_EINVAL;
  if (a->ldpq < 256 || (a->ldpq & 3) || (a->ldx1 & 3) || a->ldx1 < 128 || a->ldx2 < 128) return VCR_EINVAL;
  if (a->k == 20 || a->k == 40) {                        // packed tiles: no padding rows
    const int G = 160 / a->k, ngroups = (a->M + G - 1) / G;
    // one resident round: two workgroups per CU (each keeps a 32 x 128 slice of W2 in registers and pays one gather chain
    // of two dependent round trips before its first tile; 1024 workgroups on 512 slots paid both twice)
    const int slots = 2 * vcr_cu_count();
    const int grid = ngroups < slots ? ngroups : slots;
    // the hand-scheduled kernel stores x2 as 16-B pieces; rows that are not 16-B aligned keep the kernel above
    const bool al16 = !(a->ldx2 & 3) && !(((uintptr_t)a->x1 | (uintptr_t)a->x2 | (uintptr_t)a->b2) & 15);
    if (EDGECONV_PIPE && al16) {
      if (a->k == 20) hipLaunchKernelGGL(edgeconv_dg_pipe_kernel<20>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
      else hipLaunchKernelGGL(edgeconv_dg_pipe_kernel<40>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
    } else if (a->k == 20) hipLaunchKernelGGL(edgeconv_dg_packed_kernel<20>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL(edgeconv_dg_packed_kernel<40>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
    return VCR_LAUNCH_RC();
  }
  const int grid = a->M < 2048 ? a->M : 2048;
  hipLaunchKernelGGL(edgeconv_dg_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  return VCR_LAUNCH_RC();
}

extern "C" int vcr_gathermax_f32(const vcr_gathermax_args* a, vcr_stream_t stream) {
  if (!a || !a->pq || !a->idx || !a->y) return VCR_EINVAL;
  if (a->M <= 0 || a->k <= 0 || a->C <= 0 || a->C > 256 || (a->C & 3)) return VCR_EINVAL;
  if (a->n_per_cloud <= 0 || (a->M % a->n_per_cloud) || a->ldpq < 2 * a->C || (a->ldpq & 3) || (a->ldy & 3)) return VCR_EINVAL;
  // LDS-staged gathers when a 32-channel slice of one cloud's P rows fits a workgroup's LDS (N <= 1066), the grid fills
  // most of the chip (one workgroup per CU: >= 192 of them) and the 16-B accesses are aligned.  Measured on MI355X
  // (profiles/experiments/bench_gathermax.py): 32 clouds x 1024, k = 20, C = 256: 35.4 -> 23.0 us; 48 x 768: 39.5 -> 31.6;
  // 4 clouds x 1024 x 128 channels (16 workgroups): 11.1 -> 14.9, and 8-channel slices at N = 2048: 69 -> 121 -- those
  // keep the L2 gathers.  a->variant forces a form (1 = L2; 32 / 16 / 8 = LDS with that slice), or refuses.
  const int N = a->n_per_cloud;
  const bool aligned = !(((uintptr_t)a->pq | (uintptr_t)a->y | (uintptr_t)a->idx) & 15) && (a->k == 20 || a->k == 40);
  const size_t budget = 150 * 1024;
  int cs = 0;
  if (a->variant == 0) {
    // 32-channel slices up to N = 1066; 16-channel slices up to N = 2048 (a whole CU's LDS per workgroup; measured at 32
    // clouds x 2048: 62.8 -> 57.2 us, profiles/rounds4-5/r4j_bench_gathermax.txt; 8-channel slices lose everywhere: 121 us there)
    if (aligned && (long)(a->M / N) * ((a->C + 31) / 32) >= 192)
      cs = (size_t)N * 36 * 4 <= budget ? 32 : (size_t)N * 20 * 4 <= 160 * 1024 ? 16 : 0;
  } else if (a->variant == 32 || a->variant == 16 || a->variant == 8) {
    if (!aligned || (size_t)N * (a->variant + 4) * 4 > 160 * 1024) return VCR_EUNSUPPORTED;
    cs = a->variant;
  } else if (a->variant != 1) {
    return VCR_EINVAL;
  }
  if (cs) {
    const int slices = (a->C + cs - 1) / cs, clouds = a->M / N;
    const size_t lds = (size_t)N * (cs + 4) * 4;
    const dim3 grid(clouds * slices);
#define VCR_GM(CS_, KQ_) do { VCR_DYN_LDS((gathermax_lds_kernel<CS_, KQ_>), lds); \
      hipLaunchKernelGGL((gathermax_lds_kernel<CS_, KQ_>), grid, dim3(512), lds, (hipStream_t)stream, *a, slices); } while (0)
    if (a->k == 20) { if (cs == 32) VCR_GM(32, 5); else if (cs == 16) VCR_GM(16, 5); else VCR_GM(8, 5); }
    else { if (cs == 32) VCR_GM(32, 10); else if (cs == 16) VCR_GM(16, 10); else VCR_GM(8, 10); }
#undef VCR_GM
    return VCR_LAUNCH_RC();
  }
  hipLaunchKernelGGL(gathermax_kernel, dim3((a->M + 3) / 4), dim3(256), 0, (hipStream_t)stream, *a);
  return VCR_LAUNCH_RC();
}
