// Internal helpers shared by the libtnhip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/tnh.h"

namespace tnh {

// ---- error plumbing --------------------------------------------------------
void set_error(const char* fmt, ...);
hipStream_t stream();       // process stream (nullptr before tnh_init)
bool initialised();
bool capturing();        // between tnh_graph_begin and tnh_graph_end: no host synchronisation is legal on the stream
int num_cus();
int stream_wgs_per_cu(int64_t bytes);   // grid-stride streaming kernels: workgroups per CU for a buffer of this size

#define TNH_HIP(call)                                                        \
  do {                                                                       \
    hipError_t _e = (call);                                                  \
    if (_e != hipSuccess) {                                                  \
      tnh::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e),  \
                     __FILE__, __LINE__);                                    \
      return TNH_ERR_HIP;                                                    \
    }                                                                        \
  } while (0)

#define TNH_REQUIRE(cond, ...)                                               \
  do {                                                                       \
    if (!(cond)) {                                                           \
      tnh::set_error(__VA_ARGS__);                                           \
      return TNH_ERR_INVALID;                                                \
    }                                                                        \
  } while (0)

#define TNH_NEED_INIT()                                                      \
  do {                                                                       \
    if (!tnh::initialised()) {                                               \
      tnh::set_error("tnh_init() has not been called");                      \
      return TNH_ERR_NOT_INIT;                                               \
    }                                                                        \
  } while (0)

#define TNH_LAUNCH_CHECK()                                                   \
  do {                                                                       \
    hipError_t _e = hipGetLastError();                                       \
    if (_e != hipSuccess) {                                                  \
      tnh::set_error("kernel launch failed: %s (%s:%d)",                     \
                     hipGetErrorString(_e), __FILE__, __LINE__);             \
      return TNH_ERR_HIP;                                                    \
    }                                                                        \
  } while (0)

inline int dtype_size(int dt) {
  switch (dt) {
    case TNH_F32: return 4;
    case TNH_F64: return 8;
    case TNH_BF16: return 2;
    case TNH_F16: return 2;
    case TNH_C64: return 8;
    case TNH_C128: return 16;
    case TNH_I32: return 4;
    case TNH_I64: return 8;
    default: return 0;
  }
}

// ---- device-side scalar helpers ---------------------------------------------
struct cf32 { float re, im; };
struct cf64 { double re, im; };

__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
  return __uint_as_float(((uint32_t)h) << 16);
}
// round-to-nearest-even, NaN preserved (same rule the host shim uses).
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float f16_to_f32(uint16_t h) {
  return __half2float(__ushort_as_half(h));
}
__device__ __forceinline__ uint16_t f32_to_f16(float f) {
  return __half_as_ushort(__float2half_rn(f));
}

// 64-lane wavefront sum via DPP-free shuffles.
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// K9 fast path (f32, m >= n, n % 16 == 0): 16-wide panels factored by the band SVD's panel kernels (tnh_svd_band.hip).
// *status_host != 0: a panel was numerically rank-deficient, the caller must use the column-by-column path.
bool qr_panel16_supported(int dtype, int64_t m, int64_t n);
size_t qr_panel16_work_bytes(int dtype, int64_t m, int64_t n);
int qr_panel16(int dtype, int64_t m, int64_t n, const void* A, void* Q, void* R, void* work, int* status_host);

struct DimVec {
  int64_t v[TNH_MAX_RANK];
};

}  // namespace tnh
