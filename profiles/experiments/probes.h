// In-kernel probes for DIAGNOSTIC builds of the kernel sources (never part of libvcr_hip.so).
//
// The product sources under vcr-net_amd/csrc carry lines of the form `//@probe <statement>`: comments there.
// profiles/experiments/probe_build.py copies the sources to scratch/probed/, strips the `//@probe ` prefix, compiles each
// file with `-include probes.h -DVCR_PROBE_TU_<file stem>` and links scratch/libvcr_probe.so.  Stamp values go to a buffer
// of their own that no kernel reads (MI355X_MICROARCH.md, DVFS item 6); outputs are bit-identical to the product build.
//
//   VCR_PROBE_STAMP(slot)            lane 0 of every workgroup: the 100 MHz wall clock (s_memrealtime) to [blk][slot] and the
//                                    shader clock (s_memtime) to [blk][16 + slot]   (slot < 16)
//   VCR_PROBE_ACC_DECL / _ACC(slot) / _ACC_FLUSH(pred, row)
//                                    per-wave phase clocks: ACC adds the wall-clock time since the previous ACC to slot
//                                    (slot < 8); FLUSH writes the eight sums to row `row` when `pred`
//   vcr_dbg_probe_<stem>(host_dst, clear)   copy out / zero the TU's buffer (4096 x 32 u64)
#pragma once
#include <hip/hip_runtime.h>

#define VCR_PROBE_ROWS 4096
#define VCR_PROBE_COLS 32

#if defined(VCR_PROBE_TU_linear)
#define VCR_PROBE_BUF vcr_probe_buf_linear
#define VCR_PROBE_READER vcr_dbg_probe_linear
#elif defined(VCR_PROBE_TU_linear_bf16x3)
#define VCR_PROBE_BUF vcr_probe_buf_linear_bf16x3
#define VCR_PROBE_READER vcr_dbg_probe_linear_bf16x3
#elif defined(VCR_PROBE_TU_attention_bf16x3)
#define VCR_PROBE_BUF vcr_probe_buf_attention_bf16x3
#define VCR_PROBE_READER vcr_dbg_probe_attention_bf16x3
#elif defined(VCR_PROBE_TU_knn)
#define VCR_PROBE_BUF vcr_probe_buf_knn
#define VCR_PROBE_READER vcr_dbg_probe_knn
#elif defined(VCR_PROBE_TU_pointwise)
#define VCR_PROBE_BUF vcr_probe_buf_pointwise
#define VCR_PROBE_READER vcr_dbg_probe_pointwise
#elif defined(VCR_PROBE_TU_attention)
#define VCR_PROBE_BUF vcr_probe_buf_attention
#define VCR_PROBE_READER vcr_dbg_probe_attention
#elif defined(VCR_PROBE_TU_edgeconv)
#define VCR_PROBE_BUF vcr_probe_buf_edgeconv
#define VCR_PROBE_READER vcr_dbg_probe_edgeconv
#endif

#ifdef VCR_PROBE_BUF
__device__ unsigned long long VCR_PROBE_BUF[VCR_PROBE_ROWS * VCR_PROBE_COLS];

#define VCR_PROBE_STAMP(slot)                                                                              \
  do {                                                                                                     \
    if (threadIdx.x == 0 && blockIdx.x < VCR_PROBE_ROWS && (slot) < 16) {                                  \
      VCR_PROBE_BUF[blockIdx.x * VCR_PROBE_COLS + (slot)] = __builtin_amdgcn_s_memrealtime();              \
      VCR_PROBE_BUF[blockIdx.x * VCR_PROBE_COLS + 16 + (slot)] = __builtin_amdgcn_s_memtime();             \
    }                                                                                                      \
  } while (0)
#define VCR_PROBE_ACC_DECL \
  unsigned long long probe_t_ = __builtin_amdgcn_s_memrealtime(), probe_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define VCR_PROBE_ACC(slot)                                             \
  do {                                                                  \
    const unsigned long long n_ = __builtin_amdgcn_s_memrealtime();     \
    probe_acc_[slot] += n_ - probe_t_;                                  \
    probe_t_ = n_;                                                      \
  } while (0)
#define VCR_PROBE_ACC_FLUSH(pred, row)                                                                     \
  do {                                                                                                     \
    if ((pred) && (row) < VCR_PROBE_ROWS)                                                                  \
      for (int i_ = 0; i_ < 8; ++i_) VCR_PROBE_BUF[(row) * VCR_PROBE_COLS + i_] = probe_acc_[i_];          \
  } while (0)

extern "C" int VCR_PROBE_READER(unsigned long long* host_dst, int clear) {
  if (clear) {
    static unsigned long long zeros[VCR_PROBE_ROWS * VCR_PROBE_COLS];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(VCR_PROBE_BUF), zeros, sizeof(zeros));
  }
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(VCR_PROBE_BUF), sizeof(unsigned long long) * VCR_PROBE_ROWS * VCR_PROBE_COLS);
}
#endif
