#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// RUN = consecutive MFMAs on one accumulator before switching to the next of 4
template <int RUN>
__global__ __launch_bounds__(512, 1) void k(const float* in, float* out, int iters) {
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[threadIdx.x * 16 + i]; b[i] = in[threadIdx.x * 16 + 8 + i]; }
  f32x16 c[4];
  for (int i = 0; i < 4; ++i) c[i] = f32x16{0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 64; ++u) {
      const int ch = (u / RUN) % 4;
      c[ch] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u & 7], b[(u >> 1) & 7], c[ch], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int RUN> void run(const float* in, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 800;
  k<RUN><<<256, 512>>>(in, out, 50);
  hipEventRecord(e0, 0);
  k<RUN><<<256, 512>>>(in, out, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = 256.0 * 8 * iters * 64 * 4096.0;
  printf("4 accumulators, %2d consecutive MFMAs each: %.2f ms  %.1f TF/s (%.3f)\n", RUN, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3);
}
int main() {
  float *in, *out;
  hipMalloc(&in, 512 * 16 * 4); hipMalloc(&out, 1024 * 512 * 4);
  static float h[512 * 16]; unsigned s = 7;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffffff) / 16777216.f * 2.f - 1.f; }
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  run<1>(in, out); run<4>(in, out); run<16>(in, out); run<1>(in, out); run<4>(in, out); run<16>(in, out);
  return 0;
}
