#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND, int CHAINS>
__global__ __launch_bounds__(512, 1) void k(const float* in, float* out, int iters) {
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[threadIdx.x * 16 + i]; b[i] = in[threadIdx.x * 16 + 8 + i]; }
  float s = 0;
  if (KIND == 0) {
    f32x16 c[CHAINS];
    for (int i = 0; i < CHAINS; ++i) c[i] = f32x16{0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) c[u % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u & 7], b[(u >> 1) & 7], c[u % CHAINS], 0, 0, 0);
    }
    for (int i = 0; i < CHAINS; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  } else {
    f32x4 c[CHAINS];
    for (int i = 0; i < CHAINS; ++i) c[i] = f32x4{0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u) c[u % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u & 7], b[(u >> 1) & 7], c[u % CHAINS], 0, 0, 0);
    }
    for (int i = 0; i < CHAINS; ++i) for (int r = 0; r < 4; ++r) s += c[i][r];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int KIND, int CHAINS> void run(const float* in, float* out, int blocks, const char* nm) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 3000;
  k<KIND, CHAINS><<<blocks, 512>>>(in, out, 100);
  hipEventRecord(e0, 0);
  k<KIND, CHAINS><<<blocks, 512>>>(in, out, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * 8 * iters * (KIND == 0 ? 16 * 4096.0 : 32 * 2048.0);
  printf("%s %s chains=%d blocks=%d: %.2f ms  %.1f TF/s (%.3f of 157.3)\n", nm, KIND == 0 ? "32x32x2" : "16x16x4", CHAINS, blocks, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3);
}
int main() {
  float *in, *out;
  hipMalloc(&in, 512 * 16 * 4); hipMalloc(&out, 1024 * 512 * 4);
  static float h[512 * 16]; unsigned s = 7;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffffff) / 16777216.f * 2.f - 1.f; }
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  run<0, 4>(in, out, 256, "random"); run<1, 8>(in, out, 256, "random");
  run<0, 1>(in, out, 256, "random"); run<1, 2>(in, out, 256, "random");
  run<0, 4>(in, out, 32, "random 32 CUs"); run<1, 8>(in, out, 32, "random 32 CUs");
  hipMemset(in, 0, 512 * 16 * 4);
  run<0, 4>(in, out, 256, "zeros"); run<1, 8>(in, out, 256, "zeros");
  return 0;
}
