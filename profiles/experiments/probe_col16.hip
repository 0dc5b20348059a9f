// Hardware probe behind knn.hip's GeomCol16 (round 3): (1) what v_permlane16_swap / v_permlane32_swap return when both
// operands are the same register; (2) that v_mfma_f32_16x16x4_f32 accumulates k = 4s..4s+3 (lane row q = k mod 4) as a
// k-ascending fmaf chain, bit for bit, and where accumulator register r of lane l lands (row 4*(l>>4)+r, col l&15).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 profiles/experiments/probe_col16.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void swaps(int* o) {
  const int x = threadIdx.x;
  const auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  o[x] = a[0]; o[64 + x] = a[1]; o[128 + x] = b[0]; o[192 + x] = b[1];
}
// D[i][j] = sum_k A[i][k] B[k][j] over K = 64 through 16 MFMA steps, lane l supplies A[l&15][4s + (l>>4)], B[4s + (l>>4)][l&15]
__global__ void mfma(const float* A, const float* B, float* D) {
  const int l = threadIdx.x, q = l >> 4, c = l & 15;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c * 64 + 4 * s + q], B[(4 * s + q) * 16 + c], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + c] = acc[r];
}
int main() {
  int *o, ho[256];
  hipMalloc(&o, sizeof(ho));
  hipLaunchKernelGGL(swaps, dim3(1), dim3(64), 0, 0, o);
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  const char* nm[4] = {"permlane16_swap(x,x)[0]", "permlane16_swap(x,x)[1]", "permlane32_swap(x,x)[0]", "permlane32_swap(x,x)[1]"};
  int bad = 0;
  for (int v = 0; v < 4; ++v) {
    printf("%s: source row per destination row:", nm[v]);
    for (int row = 0; row < 4; ++row) {
      const int src = ho[v * 64 + row * 16] >> 4;
      printf(" %d<-%d", row, src);
      for (int c = 0; c < 16; ++c) bad += ho[v * 64 + row * 16 + c] != src * 16 + c;
    }
    printf("\n");
  }
  const int want[4][4] = {{0, 0, 2, 2}, {1, 1, 3, 3}, {0, 1, 0, 1}, {2, 3, 2, 3}};
  for (int v = 0; v < 4; ++v) for (int row = 0; row < 4; ++row) bad += (ho[v * 64 + row * 16] >> 4) != want[v][row];
  float hA[16 * 64], hB[64 * 16], hD[256], *A, *B, *D;
  srand(1);
  for (auto& x : hA) x = (float)rand() / RAND_MAX - 0.5f;
  for (auto& x : hB) x = (float)rand() / RAND_MAX - 0.5f;
  hipMalloc(&A, sizeof(hA)); hipMalloc(&B, sizeof(hB)); hipMalloc(&D, sizeof(hD));
  hipMemcpy(A, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(B, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(mfma, dim3(1), dim3(64), 0, 0, A, B, D);
  hipMemcpy(hD, D, sizeof(hD), hipMemcpyDeviceToHost);
  int diff = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    float acc = 0.f;
    for (int k = 0; k < 64; ++k) acc = fmaf(hA[i * 64 + k], hB[k * 16 + j], acc);
    diff += acc != hD[i * 16 + j];
  }
  printf("16x16x4 MFMA vs k-ascending fmaf chain: %d of 256 elements differ\n", diff);
  printf(bad || diff ? "PROBE FAILED (%d swap mismatches)\n" : "PROBE OK\n", bad);
  return bad || diff;
}
