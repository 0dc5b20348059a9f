#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ __launch_bounds__(512, 1) void k(const bf16x8* in, float* out, int iters) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x * 8 + i]; b[i] = in[threadIdx.x * 8 + 4 + i]; }
  f32x16 c[CHAINS];
  for (int i = 0; i < CHAINS; ++i) c[i] = f32x16{0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) c[u % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u & 3], b[(u >> 2) & 3], c[u % CHAINS], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < CHAINS; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int CHAINS> void run(const bf16x8* in, float* out, int blocks, int threads, const char* nm) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  k<CHAINS><<<blocks, threads>>>(in, out, 100);
  hipEventRecord(e0, 0);
  k<CHAINS><<<blocks, threads>>>(in, out, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * (threads / 64) * iters * 16 * 32768.0;
  double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 16 * (threads / 64 / 4.0) * ((blocks + 255) / 256));
  printf("%s chains=%d blocks=%d threads=%d: %.2f ms  %.0f TF/s  (%.1f cyc/MFMA/SIMD at 2.4 GHz)\n", nm, CHAINS, blocks, threads, ms, fl / ms / 1e9, cyc);
}
int main() {
  bf16x8* in; float* out;
  hipMalloc(&in, 512 * 8 * 16); hipMalloc(&out, 1024 * 512 * 4);
  static short h[512 * 8 * 8]; unsigned s = 7;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (short)((0x3c00 | ((s >> 9) & 0x3ff)) ^ ((s >> 3) & 0x8000)); }
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  run<1>(in, out, 256, 512, "random");
  run<4>(in, out, 256, 512, "random");
  run<1>(in, out, 256, 256, "random");
  run<4>(in, out, 256, 256, "random");
  run<4>(in, out, 32, 512, "random 32 CUs");
  hipMemset(in, 0, 512 * 8 * 16);
  run<1>(in, out, 256, 512, "zeros");
  run<4>(in, out, 256, 512, "zeros");
  return 0;
}
