// Shared device helpers for the gfx950 kernels (wave64, fp32 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <atomic>
#include <utility>
#include <type_traits>
#include "../../include/vcr_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define VCR_LAUNCH_RC() ((int)hipGetLastError())

// Args structs that grow with the ABI carry a leading `struct_bytes` (ABI 27): the entry point works on a zero-filled copy of
// what the caller really passed, so a caller built against an older, shorter header is served (its missing tail = "not
// given") and a struct that does not say its size, or says more than this build knows, is refused instead of over-read.
#include <string.h>
template <class T>
static inline int vcr_take_args(const T* user, T* mine, size_t mandatory_bytes) {
  if (!user) return VCR_EINVAL;
  const size_t n = user->struct_bytes;
  if (n < mandatory_bytes || n > sizeof(T)) return VCR_EINVAL;
  memset((void*)mine, 0, sizeof(T));
  memcpy((void*)mine, (const void*)user, n);
  mine->struct_bytes = (uint32_t)sizeof(T);
  return VCR_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is sticky per (kernel, device): raise it when a launch needs more than
// any earlier launch on that device did, not on every launch.  The cache is a per-call-site static of atomics (a
// monotone maximum), so the entry points stay re-entrant: two threads may both issue the same idempotent
// hipFuncSetAttribute, neither can lower the limit.
struct vcr_lds_cache { std::atomic<int> cur[16]; };
// The device an entry point's stream belongs to (vcr_stream_scope below binds it for the duration of the call; -1 = none
// bound: the thread's current device).
inline int& vcr_bound_device() {
  static thread_local int dev = -1;
  return dev;
}
// The limit is raised ON THE DEVICE THE LAUNCH GOES TO: the stream's when the entry point bound one -- hipFuncSetAttribute
// acts on the current device, so a stream of another device makes that device current for the one call.
inline void vcr_raise_dyn_lds(const void* kernel, int bytes, vcr_lds_cache& c) {
  int cur = 0;
  const bool have_cur = hipGetDevice(&cur) == hipSuccess;
  const int dev = vcr_bound_device() >= 0 ? vcr_bound_device() : cur;
  if (!have_cur || dev < 0 || dev >= 16) {
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return;
  }
  int seen = c.cur[dev].load(std::memory_order_acquire);
  if (seen >= bytes) return;
  if (dev != cur) (void)hipSetDevice(dev);
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (dev != cur) (void)hipSetDevice(cur);
  while (seen < bytes && !c.cur[dev].compare_exchange_weak(seen, bytes, std::memory_order_release)) {}
}
#define VCR_DYN_LDS(kernel, bytes)                                                   \
  do {                                                                               \
    static vcr_lds_cache vcr_lds_cache_;                                             \
    vcr_raise_dyn_lds(reinterpret_cast<const void*>(kernel), (bytes), vcr_lds_cache_); \
  } while (0)
#define VCR_NEG_INF (-__builtin_huge_valf())

// v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// accumulator register r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
// (verified on hardware with exact integer data).
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
// v_mfma_f32_16x16x4_f32: lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; accumulator register r of
// lane l is D[row = 4*(l>>4) + r][col = l&15].  Like the 32x32x2 form the result is bitwise a k-ordered fmaf chain.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Row moments for the folded LayerNorm from per-segment partials [nseg][2] = (sum of the segment's K/nseg values, sum of
// their squared deviations from the SEGMENT mean): combined as in Chan et al. -- M2 = sum_s (M2_s + n_s (mean_s - mean)^2)
// -- so a row with |mean| >> std loses nothing to cancellation (the one-pass sum(x^2) - sum(x)^2 / n form does).
// var is the unbiased one of Tensor.std() (transformer.py:142).
// nseg = 8 (K = 512, the path's case): the row's 64 B arrive as four 16-B loads issued together -- ONE memory round trip.
// The scalar loop below compiles to one load per iteration, each waited for before the next is issued: eight round
// trips, 16-20 us of a workgroup's prologue under load (profiles/rounds4-5/r4c_timeline_linear.txt).  Same values, same order.
__device__ __forceinline__ void ln_row_moments(const float* sp, int nseg, int K, float& mean, float& var) {
  if (nseg == 8 && (((uintptr_t)sp) & 15) == 0) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4 a = *reinterpret_cast<const v4*>(sp), b = *reinterpret_cast<const v4*>(sp + 4),
             c = *reinterpret_cast<const v4*>(sp + 8), d = *reinterpret_cast<const v4*>(sp + 12);
    const float su[8] = {a[0], a[2], b[0], b[2], c[0], c[2], d[0], d[2]};
    const float sq[8] = {a[1], a[3], b[1], b[3], c[1], c[3], d[1], d[3]};
    float s1 = 0.f;
#pragma unroll
    for (int sg = 0; sg < 8; ++sg) s1 += su[sg];
    mean = s1 / (float)K;
    const float nsg = (float)(K / 8), rn = 1.f / nsg;
    float m2 = 0.f;
#pragma unroll
    for (int sg = 0; sg < 8; ++sg) {
      const float dd = su[sg] * rn - mean;
      m2 += sq[sg] + nsg * (dd * dd);
    }
    var = m2 / (float)(K - 1);
    return;
  }
  float s1 = 0.f;
  for (int sg = 0; sg < nseg; ++sg) s1 += sp[2 * sg];                  // fixed order
  mean = s1 / (float)K;
  const float nsg = (float)(K / nseg), rn = 1.f / nsg;
  float m2 = 0.f;
  for (int sg = 0; sg < nseg; ++sg) {
    const float d = sp[2 * sg] * rn - mean;
    m2 += sp[2 * sg + 1] + nsg * (d * d);
  }
  var = m2 / (float)(K - 1);
}

// CUs of the device the launch goes to: the STREAM's device when an entry point bound one (vcr_stream_scope: a caller may
// pass a stream of a device that is not the thread's current one), else the current device.  Cached per device.
struct vcr_stream_scope {                                 // first statement of every entry point that sizes a grid by the CU count
  int saved;
  explicit vcr_stream_scope(vcr_stream_t s) : saved(vcr_bound_device()) {
    int dev = -1;
    if (s && hipStreamGetDevice((hipStream_t)s, &dev) == hipSuccess && dev >= 0) vcr_bound_device() = dev;
  }
  ~vcr_stream_scope() { vcr_bound_device() = saved; }
};
// vcr_sdpa_f32: rows of a query block, resident workgroups per CU of its kernels -- shared with the forward's workspace plan
// (forward.hip carve(): the key-split planes are only set aside where the launcher can take a split)
constexpr int VCR_SDPA_QROWS = 128;
constexpr int VCR_SDPA_WG_PER_CU = 2;
inline int vcr_cu_count() {
  static std::atomic<int> cache[16];
  int dev = vcr_bound_device();
  if (dev < 0) (void)hipGetDevice(&dev);
  const bool ok = dev >= 0 && dev < 16;
  int n = ok ? cache[dev].load(std::memory_order_relaxed) : 0;
  if (n <= 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    if (ok) cache[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// Sum over the 16 lanes of a DPP row with VALU-rate DPP moves (quad_perm xor 1, xor 2, then row_ror 4 and 8);
// every lane ends with the full sum, in a fixed order.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
  return v;
}

// Workgroups are dealt round-robin over the chip's 8 XCDs, each with its own L2.  xcd_chunk() renumbers a launch so
// that XCD x works on the x-th CONTIGUOUS eighth of the blocks: neighbouring blocks (the rows and neighbour gathers
// of one cloud) then share one L2 instead of pulling the same lines into eight.  A pure renumbering of independent
// blocks -- it changes speed and HBM traffic, never results.
__device__ __forceinline__ int xcd_chunk(int bid, int nblk) {
  const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, i = bid / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}
// The same for a 2-D launch whose blockIdx.y is the cloud: (x, y) of the renumbered block.
__device__ __forceinline__ void xcd_chunk2(int& bx, int& by) {
  const int lin = xcd_chunk((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  bx = lin % (int)gridDim.x; by = lin / (int)gridDim.x;
}
// ... and for a grid-stride loop over `n` items: this block takes first, first + stride, ... (count items), all from
// its XCD's contiguous eighth of the items.
struct xcd_slice_t { int first, stride, count; };
__device__ __forceinline__ xcd_slice_t xcd_slice(int n) {
  const int g = (int)gridDim.x, bid = (int)blockIdx.x;
  if (g % 8) return xcd_slice_t{bid, g, bid < n ? (n - bid + g - 1) / g : 0};
  const int xcd = bid % 8, slot = bid / 8, slots = g / 8;
  const int q = n / 8, r = n % 8;
  const int lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, len = q + (xcd < r ? 1 : 0);
  return xcd_slice_t{lo + slot, slots, slot < len ? (len - slot + slots - 1) / slots : 0};
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence over ALL address spaces plus
// s_barrier: hipcc emits s_waitcnt vmcnt(0) in front of it, so every global load still in flight (index / row prefetches
// issued tiles ahead) and every global STORE (a group's outputs: a full write round trip, ~3 us under load) is waited for
// at each barrier.  Here only the LDS accesses are fenced (s_waitcnt lgkmcnt(0); s_barrier): use it where the barrier
// guards LDS buffers and nothing in global memory is exchanged between the workgroup's threads.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// exchange with the lane 32 away (the other half of the wave)
__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// Device-to-device copies on the path (the driver's selections, the soft heads' srcK, emb_out; ICP's start cloud) are
// KERNELS, like the tie-counter zeroing:
// the forward then captures into a HIP graph of kernel nodes only (a memset node made replays hang on this ROCm build;
// memcpy nodes are the same kind of node).  16-B words when both pointers and the size allow, 4-B words otherwise.
static __global__ __launch_bounds__(256) void vcr_copy_words_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, long n4,
                                                         int vec) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (vec) {
    if (i < n4 / 4) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
  } else if (i < n4) {
    dst[i] = src[i];
  }
}
static inline int vcr_copy_d2d(void* dst, const void* src, size_t bytes, hipStream_t s) {     // bytes % 4 == 0
  if (bytes == 0) return 0;
  const long n4 = (long)(bytes / 4);
  const int vec = ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0 && (n4 & 3) == 0) ? 1 : 0;
  const long items = vec ? n4 / 4 : n4;
  hipLaunchKernelGGL(vcr_copy_words_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<const uint32_t*>(src), reinterpret_cast<uint32_t*>(dst), n4, vec);
  return VCR_LAUNCH_RC();
}
