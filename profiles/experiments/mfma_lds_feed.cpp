// How fast can fp32 MFMAs be issued when their operands come out of LDS?  Every wave runs the same loop:
//   R reads of 16 B per lane (ds_read_b128) per 4 MFMAs, the MFMA A operands taken FROM the data just read (mode dep) or from
//   registers with the reads' results only summed at the end (mode indep); prefetch distance one group.
// 1, 2 or 4 waves per SIMD, every CU.  Prints shader cycles per MFMA per SIMD (64 = the pipe's rate) and TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/mfma_lds_feed profiles/experiments/mfma_lds_feed.cpp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// AGPR: the accumulators are forced into the accumulator half of the register file (inline assembly, "a" constraint)
template <int READS, bool DEP, int THREADS, bool AGPR = false>
__global__ __launch_bounds__(THREADS, 1) void k(const float* in, float* out, unsigned long long* st, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += THREADS) lds[i] = in[i & 4095];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float b[4];
  for (int i = 0; i < 4; ++i) b[i] = in[lane * 4 + i];
  f32x16 c[4] = {f32x16{0}, f32x16{0}, f32x16{0}, f32x16{0}};
  f32x4 cur[READS > 0 ? READS : 1], nxt[READS > 0 ? READS : 1];
  f32x4 sink = {0, 0, 0, 0};
  const float* base = &lds[(lane * 4) & 8188];
  for (int r = 0; r < READS; ++r) cur[r] = *reinterpret_cast<const f32x4*>(base + 256 * r);
  if (READS == 0) cur[0] = f32x4{b[0], b[1], b[2], b[3]};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
#pragma unroll
      for (int r = 0; r < READS; ++r) nxt[r] = *reinterpret_cast<const f32x4*>(base + ((256 * (r + READS * (g + 1)) + 4 * it) & 4095));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = DEP ? cur[0][e] : b[e];
        if (AGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c[e]) : "v"(a), "v"(b[(e + g) & 3]));
        else c[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[(e + g) & 3], c[e], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < READS; ++r) { if (!DEP || r > 0) sink += cur[r]; cur[r] = nxt[r]; }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = sink[0] + sink[1] + sink[2] + sink[3];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  out[blockIdx.x * THREADS + threadIdx.x] = s;
  if (lane == 0) st[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int READS, bool DEP, int THREADS, bool AGPR = false> void run(const float* in, float* out, unsigned long long* st) {
  const int blocks = 256, iters = 2000, waves = THREADS / 64;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 300; ++i) k<READS, DEP, THREADS, AGPR><<<blocks, THREADS>>>(in, out, st, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 10; ++i) k<READS, DEP, THREADS, AGPR><<<blocks, THREADS>>>(in, out, st, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  std::vector<unsigned long long> h(blocks * 16);
  hipMemcpy(h.data(), st, blocks * 128, hipMemcpyDeviceToHost);
  std::vector<double> cyc;
  for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) cyc.push_back((double)h[b * 16 + w]);
  std::sort(cyc.begin(), cyc.end());
  const double mfmas = (double)iters * 32;                       // per wave
  const double per_simd = cyc[cyc.size() / 2] / (mfmas * (waves / 4.0));
  const double tf = (double)blocks * waves * mfmas * 4096.0 / (ms * 1e-3) / 1e12;
  printf("%s%d ds_read_b128 per 4 MFMAs (%s), %d wave(s) per SIMD: %5.1f shader cycles per MFMA per SIMD (64 = pipe rate)  %6.1f TFLOP/s (%.3f)\n",
         AGPR ? "[acc in AGPRs] " : "", READS, READS == 0 ? "none" : DEP ? "operands from the reads" : "reads beside", waves / 4, per_simd, tf, tf / 157.3);
}

int main() {
  float *in, *out; unsigned long long* st;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&st, 256 * 128);
  static float h[4096]; unsigned s = 7;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffffff) / 16777216.f * 2.f - 1.f; }
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  run<0, false, 512>(in, out, st);
  run<1, false, 512>(in, out, st); run<1, true, 512>(in, out, st);
  run<2, false, 512>(in, out, st); run<4, false, 512>(in, out, st); run<4, true, 512>(in, out, st);
  run<1, true, 256>(in, out, st); run<4, true, 256>(in, out, st);
  run<1, true, 1024>(in, out, st); run<4, true, 1024>(in, out, st);
  run<0, false, 512, true>(in, out, st); run<1, true, 512, true>(in, out, st); run<1, false, 512, true>(in, out, st);
  run<2, false, 512, true>(in, out, st); run<4, true, 512, true>(in, out, st);
  return 0;
}
