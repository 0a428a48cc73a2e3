// dtype traits: storage type S, compute type C, load/store with conversion.
#pragma once
#include "tnh_internal.h"

namespace tnh {

template <int DT> struct Tr;

template <> struct Tr<TNH_F32> {
  using S = float; using C = float; using R = float;
  static constexpr int REAL_DT = TNH_F32;
  __device__ static C ld(const S* p, int64_t i) { return p[i]; }
  __device__ static void st(S* p, int64_t i, C v) { p[i] = v; }
};
template <> struct Tr<TNH_F64> {
  using S = double; using C = double; using R = double;
  static constexpr int REAL_DT = TNH_F64;
  __device__ static C ld(const S* p, int64_t i) { return p[i]; }
  __device__ static void st(S* p, int64_t i, C v) { p[i] = v; }
};
template <> struct Tr<TNH_BF16> {
  using S = uint16_t; using C = float; using R = float;
  static constexpr int REAL_DT = TNH_BF16;
  __device__ static C ld(const S* p, int64_t i) { return bf16_to_f32(p[i]); }
  __device__ static void st(S* p, int64_t i, C v) { p[i] = f32_to_bf16(v); }
};
template <> struct Tr<TNH_F16> {
  using S = uint16_t; using C = float; using R = float;
  static constexpr int REAL_DT = TNH_F16;
  __device__ static C ld(const S* p, int64_t i) { return f16_to_f32(p[i]); }
  __device__ static void st(S* p, int64_t i, C v) { p[i] = f32_to_f16(v); }
};
template <> struct Tr<TNH_C64> {
  using S = cf32; using C = cf32; using R = float;
  static constexpr int REAL_DT = TNH_F32;
  __device__ static C ld(const S* p, int64_t i) { return p[i]; }
  __device__ static void st(S* p, int64_t i, C v) { p[i] = v; }
};
template <> struct Tr<TNH_C128> {
  using S = cf64; using C = cf64; using R = double;
  static constexpr int REAL_DT = TNH_F64;
  __device__ static C ld(const S* p, int64_t i) { return p[i]; }
  __device__ static void st(S* p, int64_t i, C v) { p[i] = v; }
};

// int32 / int64: exact two's-complement arithmetic (NumPy's wrap-around semantics); the reference
// passes any NumPy dtype straight through tensordot / sum / trace (numpy_backend.py:35-54, 603-607).
template <> struct Tr<TNH_I32> {
  using S = int32_t; using C = int32_t; using R = int32_t;
  static constexpr int REAL_DT = TNH_I32;
  __device__ static C ld(const S* p, int64_t i) { return p[i]; }
  __device__ static void st(S* p, int64_t i, C v) { p[i] = v; }
};
template <> struct Tr<TNH_I64> {
  using S = int64_t; using C = int64_t; using R = int64_t;
  static constexpr int REAL_DT = TNH_I64;
  __device__ static C ld(const S* p, int64_t i) { return p[i]; }
  __device__ static void st(S* p, int64_t i, C v) { p[i] = v; }
};

// ---- arithmetic on compute types -------------------------------------------
__device__ __forceinline__ cf32 operator+(cf32 a, cf32 b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cf32 operator-(cf32 a, cf32 b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cf32 operator*(cf32 a, cf32 b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ __forceinline__ cf64 operator+(cf64 a, cf64 b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cf64 operator-(cf64 a, cf64 b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cf64 operator*(cf64 a, cf64 b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
template <typename Z, typename R>
__device__ __forceinline__ Z cdiv(Z a, Z b) {
  // Smith's algorithm (what numpy's complex division uses).
  if (fabs((double)b.re) >= fabs((double)b.im)) {
    const R r = b.im / b.re, d = b.re + b.im * r;
    return {(a.re + a.im * r) / d, (a.im - a.re * r) / d};
  }
  const R r = b.re / b.im, d = b.im + b.re * r;
  return {(a.re * r + a.im) / d, (a.im * r - a.re) / d};
}
__device__ __forceinline__ cf32 operator/(cf32 a, cf32 b) { return cdiv<cf32, float>(a, b); }
__device__ __forceinline__ cf64 operator/(cf64 a, cf64 b) { return cdiv<cf64, double>(a, b); }

__device__ __forceinline__ float zero_of(float) { return 0.f; }
__device__ __forceinline__ double zero_of(double) { return 0.0; }
__device__ __forceinline__ cf32 zero_of(cf32) { return {0.f, 0.f}; }
__device__ __forceinline__ cf64 zero_of(cf64) { return {0.0, 0.0}; }
__device__ __forceinline__ int32_t zero_of(int32_t) { return 0; }
__device__ __forceinline__ int64_t zero_of(int64_t) { return 0; }

__device__ __forceinline__ float abs2(float a) { return a * a; }
__device__ __forceinline__ double abs2(double a) { return a * a; }
__device__ __forceinline__ float abs2(cf32 a) { return a.re * a.re + a.im * a.im; }
__device__ __forceinline__ double abs2(cf64 a) { return a.re * a.re + a.im * a.im; }

__device__ __forceinline__ cf32 shfl_xor_t(cf32 v, int off) {
  return {__shfl_xor(v.re, off, 64), __shfl_xor(v.im, off, 64)};
}
__device__ __forceinline__ cf64 shfl_xor_t(cf64 v, int off) {
  return {__shfl_xor(v.re, off, 64), __shfl_xor(v.im, off, 64)};
}
__device__ __forceinline__ float shfl_xor_t(float v, int off) { return __shfl_xor(v, off, 64); }
__device__ __forceinline__ double shfl_xor_t(double v, int off) { return __shfl_xor(v, off, 64); }
__device__ __forceinline__ int32_t shfl_xor_t(int32_t v, int off) { return __shfl_xor(v, off, 64); }
__device__ __forceinline__ int64_t shfl_xor_t(int64_t v, int off) {
  return (int64_t)__shfl_xor((long long)v, off, 64);
}

template <typename T>
__device__ __forceinline__ T wave_sum_t(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = v + shfl_xor_t(v, off);
  return v;
}

#define TNH_DISPATCH_FLOAT(dt, ...)                                   \
  switch (dt) {                                                       \
    case TNH_F32: { constexpr int DT = TNH_F32; __VA_ARGS__; } break; \
    case TNH_F64: { constexpr int DT = TNH_F64; __VA_ARGS__; } break; \
    case TNH_BF16: { constexpr int DT = TNH_BF16; __VA_ARGS__; } break; \
    case TNH_F16: { constexpr int DT = TNH_F16; __VA_ARGS__; } break; \
    case TNH_C64: { constexpr int DT = TNH_C64; __VA_ARGS__; } break; \
    case TNH_C128: { constexpr int DT = TNH_C128; __VA_ARGS__; } break; \
    default:                                                          \
      tnh::set_error("unsupported dtype %d", (int)(dt));              \
      return TNH_ERR_UNSUPPORTED;                                     \
  }

// float dtypes + int32 / int64 (arithmetic that is exact on integers: add / sub / mul, sums, products)
#define TNH_DISPATCH_NUM(dt, ...)                                     \
  switch (dt) {                                                       \
    case TNH_F32: { constexpr int DT = TNH_F32; __VA_ARGS__; } break; \
    case TNH_F64: { constexpr int DT = TNH_F64; __VA_ARGS__; } break; \
    case TNH_BF16: { constexpr int DT = TNH_BF16; __VA_ARGS__; } break; \
    case TNH_F16: { constexpr int DT = TNH_F16; __VA_ARGS__; } break; \
    case TNH_C64: { constexpr int DT = TNH_C64; __VA_ARGS__; } break; \
    case TNH_C128: { constexpr int DT = TNH_C128; __VA_ARGS__; } break; \
    case TNH_I32: { constexpr int DT = TNH_I32; __VA_ARGS__; } break; \
    case TNH_I64: { constexpr int DT = TNH_I64; __VA_ARGS__; } break; \
    default:                                                          \
      tnh::set_error("unsupported dtype %d", (int)(dt));              \
      return TNH_ERR_UNSUPPORTED;                                     \
  }

}  // namespace tnh
