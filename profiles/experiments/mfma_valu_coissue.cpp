// Do fp32-input MFMAs and ordinary VALU instructions share an execution resource on gfx950?
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run a register-only MFMA loop, waves 4-7 (their SIMD
// partners) a register-only VALU loop of one kind.  Each role is timed alone and together (s_memtime around the loop,
// median over workgroups).  If the two were independent pipes, "together" would cost each role what it costs alone.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/mfma_valu_coissue profiles/experiments/mfma_valu_coissue.cpp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MK: 0 = v_mfma_f32_32x32x2_f32, 1 = v_mfma_f32_16x16x4_f32, 2 = v_mfma_f32_32x32x16_bf16
// VK: 0 = v_fma_f32, 1 = v_add_u32 / v_xor (integer), 2 = v_max_f32 (compare class), 3 = ds_read_b128 (LDS)
template <int MK, int VK>
__global__ __launch_bounds__(512, 1) void k(const float* in, float* out, unsigned long long* st, int mfma_iters, int valu_iters,
                                            int run_mfma, int run_valu) {
  __shared__ float lds[4096];
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = in[i & 1023];
  __syncthreads();
  float s = 0;
  unsigned long long t0 = 0, t1 = 0;
  if (wave < 4) {
    if (run_mfma) {
      float a[8], b[8];
      for (int i = 0; i < 8; ++i) { a[i] = in[threadIdx.x * 16 + i]; b[i] = in[threadIdx.x * 16 + 8 + i]; }
      if (MK == 0) {
        f32x16 c[4] = {f32x16{0}, f32x16{0}, f32x16{0}, f32x16{0}};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
          for (int u = 0; u < 16; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u & 7], b[(u >> 1) & 7], c[u & 3], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
        t1 = __builtin_amdgcn_s_memtime();
      } else if (MK == 1) {
        f32x4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = f32x4{0, 0, 0, 0};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
          for (int u = 0; u < 32; ++u) c[u & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u & 7], b[(u >> 1) & 7], c[u & 7], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += c[i][r];
        t1 = __builtin_amdgcn_s_memtime();
      } else {
        bf16x8 av, bv;
        for (int i = 0; i < 8; ++i) { av[i] = (__bf16)a[i]; bv[i] = (__bf16)b[i]; }
        f32x16 c[4] = {f32x16{0}, f32x16{0}, f32x16{0}, f32x16{0}};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
          for (int u = 0; u < 32; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c[u & 3], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
        t1 = __builtin_amdgcn_s_memtime();
      }
    }
  } else if (run_valu) {
    float x[8];
    unsigned xi[8];
    for (int i = 0; i < 8; ++i) { x[i] = in[threadIdx.x * 8 + i]; xi[i] = __float_as_uint(x[i]); }
    const float m = in[3], ad = in[5];
    f32x4 acc4 = {0, 0, 0, 0};
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 64; ++u) {
        if (VK == 0) x[u & 7] = __builtin_fmaf(x[u & 7], m, ad);
        else if (VK == 1) xi[u & 7] = (xi[u & 7] + 0x9e3779b9u) ^ (unsigned)u;
        else if (VK == 2) x[u & 7] = __builtin_fmaxf(x[u & 7], x[(u + 1) & 7]);
        else if (u < 16) { const f32x4 v = *reinterpret_cast<const f32x4*>(&lds[((threadIdx.x & 63) * 4 + u * 256 + it * 4) & 4092]); acc4 += v; }
      }
    }
    for (int i = 0; i < 8; ++i) s += x[i] + (float)xi[i];
    s += acc4[0] + acc4[1] + acc4[2] + acc4[3];
    t1 = __builtin_amdgcn_s_memtime();
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) st[blockIdx.x * 8 + wave] = t1 - t0;
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

template <int MK, int VK> void run(const float* in, float* out, unsigned long long* st, const char* mn, const char* vn, double mfma_cyc_nominal, int mfma_per_iter) {
  const int blocks = 256, mi = 1500, vi = MK == 2 ? 700 : 1400;
  double res[3][2];
  for (int mode = 0; mode < 3; ++mode) {       // 0 both, 1 MFMA alone, 2 VALU alone
    const int rm = mode != 2, rv = mode != 1;
    for (int rep = 0; rep < 200; ++rep) k<MK, VK><<<blocks, 512>>>(in, out, st, mi, vi, rm, rv);   // ~1 s of load first
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), st, blocks * 64, hipMemcpyDeviceToHost);
    std::vector<double> m, v;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4; ++w) { m.push_back((double)h[b * 8 + w]); v.push_back((double)h[b * 8 + 4 + w]); }
    res[mode][0] = med(m) / ((double)mi * mfma_per_iter);
    res[mode][1] = med(v) / ((double)vi * (VK == 3 ? 16 : 64));
  }
  printf("%-22s + %-12s | MFMA cycles each: alone %6.1f  beside %6.1f (nominal %.0f) | %s cycles each: alone %6.2f  beside %6.2f\n", mn, vn,
         res[1][0], res[0][0], mfma_cyc_nominal, vn, res[2][1], res[0][1]);
}

int main() {
  float *in, *out; unsigned long long* st;
  hipMalloc(&in, 512 * 16 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 256 * 64);
  static float h[512 * 16]; unsigned s = 7;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffffff) / 16777216.f * 2.f - 1.f; }
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  printf("one MFMA wave + one VALU wave per SIMD, 256 CUs, register-only loops; cycles = s_memtime ticks per instruction of the role's own wave\n");
  run<0, 0>(in, out, st, "v_mfma_f32_32x32x2_f32", "v_fma_f32", 64, 16);
  run<0, 1>(in, out, st, "v_mfma_f32_32x32x2_f32", "v_add+v_xor", 64, 16);
  run<0, 2>(in, out, st, "v_mfma_f32_32x32x2_f32", "v_max_f32", 64, 16);
  run<0, 3>(in, out, st, "v_mfma_f32_32x32x2_f32", "ds_read_b128", 64, 16);
  run<1, 0>(in, out, st, "v_mfma_f32_16x16x4_f32", "v_fma_f32", 32, 32);
  run<1, 1>(in, out, st, "v_mfma_f32_16x16x4_f32", "v_add+v_xor", 32, 32);
  run<1, 3>(in, out, st, "v_mfma_f32_16x16x4_f32", "ds_read_b128", 32, 32);
  run<2, 0>(in, out, st, "v_mfma_f32_32x32x16_bf16", "v_fma_f32", 32, 32);
  run<2, 1>(in, out, st, "v_mfma_f32_32x32x16_bf16", "v_add+v_xor", 32, 32);
  return 0;
}
