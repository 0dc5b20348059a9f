// In-kernel clock of register-only fp32 MFMA loops on every CU (MI355X_MICROARCH.md, DVFS item 6): stamps of
// s_memtime (shader clock) and s_memrealtime (100 MHz) once around the loop, after >= 2 s of back-to-back launches of
// the same kernel on random operands; median over workgroups.  Stamps go to a buffer of their own.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/mfma_f32_clock profiles/experiments/mfma_f32_clock.cpp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND, int CHAINS, int THREADS = 512>
__global__ __launch_bounds__(THREADS, 1) void k(const float* in, float* out, unsigned long long* stamps, int iters) {
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[threadIdx.x * 16 + i]; b[i] = in[threadIdx.x * 16 + 8 + i]; }
  float s = 0;
  unsigned long long t0 = 0, r0 = 0, t1 = 0, r1 = 0;
  if (KIND == 0) {
    f32x16 c[CHAINS];
    for (int i = 0; i < CHAINS; ++i) c[i] = f32x16{0};
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) c[u % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u & 7], b[(u >> 1) & 7], c[u % CHAINS], 0, 0, 0);
    }
    for (int i = 0; i < CHAINS; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
    t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
  } else {
    f32x4 c[CHAINS];
    for (int i = 0; i < CHAINS; ++i) c[i] = f32x4{0};
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u) c[u % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u & 7], b[(u >> 1) & 7], c[u % CHAINS], 0, 0, 0);
    }
    for (int i = 0; i < CHAINS; ++i) for (int r = 0; r < 4; ++r) s += c[i][r];
    t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
  }
  out[blockIdx.x * THREADS + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}
template <int KIND, int CHAINS, int THREADS = 512> void run(const float* in, float* out, unsigned long long* st, int blocks, const char* nm) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 3000;
  auto w0 = std::chrono::steady_clock::now();
  int launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < 2.0) {   // >= 2 s of load first
    for (int i = 0; i < 20; ++i) k<KIND, CHAINS, THREADS><<<blocks, THREADS>>>(in, out, st, iters);
    hipDeviceSynchronize(); launches += 20;
  }
  hipEventRecord(e0, 0);
  for (int i = 0; i < 10; ++i) k<KIND, CHAINS, THREADS><<<blocks, THREADS>>>(in, out, st, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  std::vector<unsigned long long> h(blocks * 2);
  hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
  std::vector<double> ghz, cyc;
  for (int b = 0; b < blocks; ++b) { ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1); cyc.push_back((double)h[2 * b]); }
  std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
  const double nm_per_wave = (double)iters * (KIND == 0 ? 16 : 32);                 // MFMAs per wave
  const double cyc_per_mfma_simd = cyc[blocks / 2] / (nm_per_wave * (THREADS / 256));   // THREADS / 256 waves per SIMD share the pipe
  double fl = (double)blocks * (THREADS / 64) * iters * (KIND == 0 ? 16 * 4096.0 : 32 * 2048.0);
  printf("%-7s %s chains=%d blocks=%d waves/SIMD=%d: %.3f ms/launch  %.1f TF/s (%.3f of 157.3)  in-kernel clock median %.3f GHz (min %.3f max %.3f)  "
         "%.1f shader cycles per MFMA per SIMD (nominal %d)  [%d warm-up launches]\n", nm, KIND == 0 ? "32x32x2" : "16x16x4", CHAINS, blocks, THREADS / 256,
         ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3, ghz[blocks / 2], ghz.front(), ghz.back(), cyc_per_mfma_simd, KIND == 0 ? 64 : 32, launches);
}
int main() {
  float *in, *out; unsigned long long* st;
  hipMalloc(&in, 512 * 16 * 4); hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&st, 1024 * 16);
  static float h[512 * 16]; unsigned s = 7;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffffff) / 16777216.f * 2.f - 1.f; }
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  run<0, 4>(in, out, st, 256, "random"); run<1, 8>(in, out, st, 256, "random");
  run<0, 1>(in, out, st, 256, "random"); run<1, 2>(in, out, st, 256, "random");
  run<0, 4>(in, out, st, 32, "rnd32CU");
  run<0, 4, 256>(in, out, st, 256, "random"); run<1, 8, 256>(in, out, st, 256, "random");      // ONE wave per SIMD
  run<0, 4, 1024>(in, out, st, 256, "random");                                                  // four waves per SIMD
  hipMemset(in, 0, 512 * 16 * 4);
  run<0, 4>(in, out, st, 256, "zeros"); run<1, 8>(in, out, st, 256, "zeros");
  return 0;
}
