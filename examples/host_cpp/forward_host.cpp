// A host with no Python and no PyTorch in the process: the C-ABI of include/vcr_hip.h driven from C++ with memory from
// hipMalloc and a stream of its own -- what a maintainer binding the library from another runtime does.
//
//   forward_host <model.blob> <out.bin> [iters]
//
// model.blob (written by vcrnet_amd.export_blob.write_blob from a module's packed weights):
//   u32 magic 'VCRB', u32 abi, u32 sizeof(vcr_vcrnet_weights), u32 n_patches, u32 B, u32 N, u64 blob_bytes,
//   u64 src_offset, u64 tgt_offset,  n_patches x (u32 struct_offset, u32 pad, u64 blob_offset),
//   the struct image (pointers zero), then the blob (packed weight tensors + the two clouds [B,3,N] fp32).
// Every pointer field of the weights struct named by a patch becomes device_base + blob_offset.  Output: B x (9 + 3 + 9 + 3)
// floats (R_ab, t_ab, R_ba, t_ba) + the first 3 x min(K, 8) floats of corr4 -- compared with the Python module's by the test.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "vcr_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)
#define CHECK_VCR(x) do { int e_ = (x); if (e_ != 0) { fprintf(stderr, "%s: rc=%d (%s)\n", #x, e_, vcr_strerror(e_)); return 4; } } while (0)

struct Header { uint32_t magic, abi, weights_bytes, n_patches, B, N; uint64_t blob_bytes, src_off, tgt_off; };
struct Patch { uint32_t struct_off, pad; uint64_t blob_off; };

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s <model.blob> <out.bin> [iters]\n", argv[0]); return 2; }
  const int iters = argc > 3 ? atoi(argv[3]) : 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  Header h;
  if (fread(&h, sizeof h, 1, f) != 1 || h.magic != 0x42524356u) { fprintf(stderr, "not a VCRB blob\n"); return 2; }
  if ((int)h.abi != vcr_abi_version() || h.abi != VCR_ABI_VERSION) {
    fprintf(stderr, "blob written for ABI %u, header %d, library %d\n", h.abi, VCR_ABI_VERSION, vcr_abi_version());
    return 2;
  }
  if (h.weights_bytes != sizeof(vcr_vcrnet_weights)) { fprintf(stderr, "struct size %u != %zu\n", h.weights_bytes, sizeof(vcr_vcrnet_weights)); return 2; }
  std::vector<Patch> patches(h.n_patches);
  if (fread(patches.data(), sizeof(Patch), h.n_patches, f) != h.n_patches) return 2;
  vcr_vcrnet_weights W;
  if (fread(&W, sizeof W, 1, f) != 1) return 2;
  std::vector<unsigned char> blob(h.blob_bytes);
  if (fread(blob.data(), 1, h.blob_bytes, f) != h.blob_bytes) return 2;
  fclose(f);

  unsigned char* dev = nullptr;
  CHECK_HIP(hipMalloc((void**)&dev, h.blob_bytes));
  CHECK_HIP(hipMemcpy(dev, blob.data(), h.blob_bytes, hipMemcpyHostToDevice));
  for (const Patch& p : patches) {
    if (p.struct_off + sizeof(void*) > sizeof W || p.blob_off >= h.blob_bytes) { fprintf(stderr, "bad patch\n"); return 2; }
    void* ptr = dev + p.blob_off;
    memcpy(reinterpret_cast<unsigned char*>(&W) + p.struct_off, &ptr, sizeof ptr);
  }
  W.struct_bytes = (uint32_t)sizeof W;

  const int B = (int)h.B, N = (int)h.N;
  const int K = vcr_vcrnet_pairs(&W, N);
  if (K <= 0) { fprintf(stderr, "vcr_vcrnet_pairs refused the weights\n"); return 4; }
  const size_t ws_bytes = vcr_vcrnet_workspace_bytes(&W, B, N);
  void* ws = nullptr;
  float *corr4, *src4, *pose;                              // pose: R_ab [B,9] | t_ab [B,3] | R_ba [B,9] | t_ba [B,3]
  CHECK_HIP(hipMalloc(&ws, ws_bytes));
  CHECK_HIP(hipMalloc((void**)&corr4, (size_t)B * K * 4 * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&src4, (size_t)B * K * 4 * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&pose, (size_t)B * 24 * sizeof(float)));
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  vcr_vcrnet_io io;
  memset(&io, 0, sizeof io);
  io.src_cf = reinterpret_cast<const float*>(dev + h.src_off);
  io.tgt_cf = reinterpret_cast<const float*>(dev + h.tgt_off);
  io.B = B; io.N = N; io.corr4 = corr4; io.src4 = src4;
  io.R_ab = pose; io.t_ab = pose + (size_t)B * 9; io.R_ba = pose + (size_t)B * 12; io.t_ba = pose + (size_t)B * 21;
  if (iters == 1) CHECK_VCR(vcr_vcrnet_forward_f32(&W, &io, ws, ws_bytes, stream));
  else CHECK_VCR(vcr_vcrnet_iter_f32(&W, &io, iters, ws, ws_bytes, stream, nullptr));
  CHECK_HIP(hipStreamSynchronize(stream));

  std::vector<float> out((size_t)B * 24 + (size_t)B * 4 * (K < 8 ? K : 8));
  CHECK_HIP(hipMemcpy(out.data(), pose, (size_t)B * 24 * sizeof(float), hipMemcpyDeviceToHost));
  for (int b = 0; b < B; ++b)
    CHECK_HIP(hipMemcpy(out.data() + (size_t)B * 24 + (size_t)b * 4 * (K < 8 ? K : 8), corr4 + (size_t)b * K * 4,
                        (size_t)4 * (K < 8 ? K : 8) * sizeof(float), hipMemcpyDeviceToHost));
  FILE* o = fopen(argv[2], "wb");
  if (!o) { perror(argv[2]); return 2; }
  fwrite(out.data(), sizeof(float), out.size(), o);
  fclose(o);
  printf("forward_host: ABI %d, B=%d N=%d pairs=%d workspace %.1f MB, iters %d: R_ab[0] = %.6f %.6f %.6f ...\n", vcr_abi_version(), B, N, K,
         ws_bytes / 1048576.0, iters, out[0], out[1], out[2]);
  (void)hipFree(ws); (void)hipFree(corr4); (void)hipFree(src4); (void)hipFree(pose); (void)hipFree(dev);
  (void)hipStreamDestroy(stream);
  return 0;
}
