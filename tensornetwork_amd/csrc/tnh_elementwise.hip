// K5/K6: elementwise, broadcasting binary ops, fills and casts (HBM-bound).
// Algorithmic bytes: (inputs + output) * itemsize per element.
#include "tnh_types.h"

namespace tnh {

// ---- scalar math on compute types --------------------------------------------
template <typename R> struct Cx { R re, im; };

__device__ __forceinline__ float m_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double m_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float m_exp(float x) { return expf(x); }
__device__ __forceinline__ double m_exp(double x) { return exp(x); }
__device__ __forceinline__ float m_log(float x) { return logf(x); }
__device__ __forceinline__ double m_log(double x) { return log(x); }
__device__ __forceinline__ float m_sin(float x) { return sinf(x); }
__device__ __forceinline__ double m_sin(double x) { return sin(x); }
__device__ __forceinline__ float m_cos(float x) { return cosf(x); }
__device__ __forceinline__ double m_cos(double x) { return cos(x); }
__device__ __forceinline__ float m_sinh(float x) { return sinhf(x); }
__device__ __forceinline__ double m_sinh(double x) { return sinh(x); }
__device__ __forceinline__ float m_cosh(float x) { return coshf(x); }
__device__ __forceinline__ double m_cosh(double x) { return cosh(x); }
__device__ __forceinline__ float m_hypot(float a, float b) { return hypotf(a, b); }
__device__ __forceinline__ double m_hypot(double a, double b) { return hypot(a, b); }
__device__ __forceinline__ float m_atan2(float a, float b) { return atan2f(a, b); }
__device__ __forceinline__ double m_atan2(double a, double b) { return atan2(a, b); }
__device__ __forceinline__ float m_pow(float a, float b) { return powf(a, b); }
__device__ __forceinline__ double m_pow(double a, double b) { return pow(a, b); }
__device__ __forceinline__ float m_abs(float a) { return fabsf(a); }
__device__ __forceinline__ double m_abs(double a) { return fabs(a); }

template <typename R>
__device__ __forceinline__ R real_unary(int op, R x) {
  switch (op) {
    case TNH_OP_SQRT: return m_sqrt(x);
    case TNH_OP_CONJ: return x;
    case TNH_OP_ABS: return m_abs(x);
    case TNH_OP_SIGN: return (x > R(0)) ? R(1) : ((x < R(0)) ? R(-1) : x);  // keeps 0 / nan
    case TNH_OP_EXP: return m_exp(x);
    case TNH_OP_LOG: return m_log(x);
    case TNH_OP_SIN: return m_sin(x);
    case TNH_OP_COS: return m_cos(x);
    case TNH_OP_NEG: return -x;
    case TNH_OP_REAL: return x;
    case TNH_OP_IMAG: return R(0);
    default: return x;
  }
}

template <typename Z, typename R>
__device__ __forceinline__ Z cplx_unary(int op, Z z) {
  switch (op) {
    case TNH_OP_SQRT: {
      // principal branch, numerically stable form
      const R m = m_hypot(z.re, z.im);
      if (m == R(0)) return {R(0), z.im};
      R a = m_sqrt((m + m_abs(z.re)) / R(2));
      R b = z.im / (R(2) * a);
      if (z.re >= R(0)) return {a, b};
      return {m_abs(b), (z.im < R(0)) ? -a : a};
    }
    case TNH_OP_CONJ: return {z.re, -z.im};
    case TNH_OP_SIGN: {
      // numpy >= 2: z / |z|
      const R m = m_hypot(z.re, z.im);
      if (m == R(0)) return {R(0), R(0)};
      return {z.re / m, z.im / m};
    }
    case TNH_OP_EXP: {
      const R e = m_exp(z.re);
      return {e * m_cos(z.im), e * m_sin(z.im)};
    }
    case TNH_OP_LOG: return {m_log(m_hypot(z.re, z.im)), m_atan2(z.im, z.re)};
    case TNH_OP_SIN: return {m_sin(z.re) * m_cosh(z.im), m_cos(z.re) * m_sinh(z.im)};
    case TNH_OP_COS: return {m_cos(z.re) * m_cosh(z.im), -m_sin(z.re) * m_sinh(z.im)};
    case TNH_OP_NEG: return {-z.re, -z.im};
    default: return z;
  }
}

__device__ __forceinline__ float apply_unary(int op, float x) { return real_unary<float>(op, x); }
__device__ __forceinline__ double apply_unary(int op, double x) { return real_unary<double>(op, x); }
__device__ __forceinline__ cf32 apply_unary(int op, cf32 x) { return cplx_unary<cf32, float>(op, x); }
__device__ __forceinline__ cf64 apply_unary(int op, cf64 x) { return cplx_unary<cf64, double>(op, x); }

template <typename Z, typename R>
__device__ __forceinline__ Z cplx_pow(Z a, Z b) {
  if (a.re == R(0) && a.im == R(0)) {
    if (b.re == R(0) && b.im == R(0)) return {R(1), R(0)};
    return {R(0), R(0)};
  }
  const Z l = cplx_unary<Z, R>(TNH_OP_LOG, a);
  const Z e = {b.re * l.re - b.im * l.im, b.re * l.im + b.im * l.re};
  return cplx_unary<Z, R>(TNH_OP_EXP, e);
}

__device__ __forceinline__ float apply_binary(int op, float a, float b) {
  switch (op) {
    case TNH_OP_ADD: return a + b;
    case TNH_OP_SUB: return a - b;
    case TNH_OP_MUL: return a * b;
    case TNH_OP_DIV: return a / b;
    default: return m_pow(a, b);
  }
}
__device__ __forceinline__ double apply_binary(int op, double a, double b) {
  switch (op) {
    case TNH_OP_ADD: return a + b;
    case TNH_OP_SUB: return a - b;
    case TNH_OP_MUL: return a * b;
    case TNH_OP_DIV: return a / b;
    default: return m_pow(a, b);
  }
}
__device__ __forceinline__ cf32 apply_binary(int op, cf32 a, cf32 b) {
  switch (op) {
    case TNH_OP_ADD: return a + b;
    case TNH_OP_SUB: return a - b;
    case TNH_OP_MUL: return a * b;
    case TNH_OP_DIV: return a / b;
    default: return cplx_pow<cf32, float>(a, b);
  }
}
__device__ __forceinline__ cf64 apply_binary(int op, cf64 a, cf64 b) {
  switch (op) {
    case TNH_OP_ADD: return a + b;
    case TNH_OP_SUB: return a - b;
    case TNH_OP_MUL: return a * b;
    case TNH_OP_DIV: return a / b;
    default: return cplx_pow<cf64, double>(a, b);
  }
}

// integers: add / sub / mul wrap like NumPy's; division truncates toward zero and x / 0 = 0 (the host
// shim routes true division through float64 as NumPy does; this only keeps the kernel total);
// pow by repeated squaring for non-negative exponents.
template <typename I, typename U>
__device__ __forceinline__ I int_binary(int op, I a, I b) {
  switch (op) {
    case TNH_OP_ADD: return (I)((U)a + (U)b);
    case TNH_OP_SUB: return (I)((U)a - (U)b);
    case TNH_OP_MUL: return (I)((U)a * (U)b);
    case TNH_OP_DIV: return (b == 0 || (b == -1 && a == (I)((U)1 << (sizeof(I) * 8 - 1)))) ? 0 : a / b;
    default: {
      if (b < 0) return 0;
      U r = 1, x = (U)a;
      for (U e = (U)b; e; e >>= 1) {
        if (e & 1) r *= x;
        x *= x;
      }
      return (I)r;
    }
  }
}
__device__ __forceinline__ int32_t apply_binary(int op, int32_t a, int32_t b) { return int_binary<int32_t, uint32_t>(op, a, b); }
__device__ __forceinline__ int64_t apply_binary(int op, int64_t a, int64_t b) { return int_binary<int64_t, uint64_t>(op, a, b); }
template <typename I>
__device__ __forceinline__ I int_unary(int op, I x) {
  switch (op) {
    case TNH_OP_ABS: return x < 0 ? (I)(0 - x) : x;
    case TNH_OP_SIGN: return (x > 0) - (x < 0);
    case TNH_OP_NEG: return (I)(0 - x);
    case TNH_OP_IMAG: return 0;
    default: return x;   // conj / real / copy; transcendental ops are promoted to float on the host
  }
}
__device__ __forceinline__ int32_t apply_unary(int op, int32_t x) { return int_unary<int32_t>(op, x); }
__device__ __forceinline__ int64_t apply_unary(int op, int64_t x) { return int_unary<int64_t>(op, x); }
__device__ __forceinline__ int32_t from_scalar(int32_t, double re, double) { return (int32_t)(int64_t)re; }
__device__ __forceinline__ int64_t from_scalar(int64_t, double re, double) { return (int64_t)re; }

__device__ __forceinline__ float from_scalar(float, double re, double) { return (float)re; }
__device__ __forceinline__ double from_scalar(double, double re, double) { return re; }
__device__ __forceinline__ cf32 from_scalar(cf32, double re, double im) { return {(float)re, (float)im}; }
__device__ __forceinline__ cf64 from_scalar(cf64, double re, double im) { return {re, im}; }

// ---- kernels ---------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void unary_kernel(int op, typename Tr<DT>::S* __restrict__ dst,
                                                    const typename Tr<DT>::S* __restrict__ src,
                                                    int64_t n) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step)
    Tr<DT>::st(dst, i, apply_unary(op, Tr<DT>::ld(src, i)));
}

// f32 fast paths (round 3): 16-byte accesses, two vectors per lane in flight, grid-stride.
// The generic kernels above move 4 bytes per lane per access and reach 5.2-5.9 TB/s; a copy reaches 6.3.
//   MODE 0: dst = op(src)            MODE 1: dst = src (op) scalar / scalar (op) src
//   MODE 2: dst[r, c] = a[r, c] (op) v[c]   (cols % 4 == 0)        MODE 3: dst[r, c] = a[r, c] (op) v[r]
template <int MODE>
__global__ __launch_bounds__(256) void f32v_kernel(int op, float4* __restrict__ dst, const float4* __restrict__ src,
                                                   const float* __restrict__ vec, float scalar, int left, int64_t nvec,
                                                   int64_t cols4) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  auto one = [&](float x, float s) -> float {
    if (MODE == 0) return apply_unary(op, x);
    return left ? apply_binary(op, s, x) : apply_binary(op, x, s);
  };
  auto vec4 = [&](int64_t i, float4 x) -> float4 {
    float4 s = make_float4(scalar, scalar, scalar, scalar);
    if (MODE == 2) s = *reinterpret_cast<const float4*>(vec + 4 * (i % cols4));
    if (MODE == 3) { const float t = vec[i / cols4]; s = make_float4(t, t, t, t); }
    return make_float4(one(x.x, s.x), one(x.y, s.y), one(x.z, s.z), one(x.w, s.w));
  };
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (MODE == 0 && op == TNH_OP_SQRT) {       // the one unary op on the hot path (sqrt(s) of split_node): no op switch per element
    for (; i + step < nvec; i += 2 * step) {
      const float4 x0 = src[i], x1 = src[i + step];
      dst[i] = make_float4(m_sqrt(x0.x), m_sqrt(x0.y), m_sqrt(x0.z), m_sqrt(x0.w));
      dst[i + step] = make_float4(m_sqrt(x1.x), m_sqrt(x1.y), m_sqrt(x1.z), m_sqrt(x1.w));
    }
    if (i < nvec) {
      const float4 x0 = src[i];
      dst[i] = make_float4(m_sqrt(x0.x), m_sqrt(x0.y), m_sqrt(x0.z), m_sqrt(x0.w));
    }
    return;
  }
  for (; i + step < nvec; i += 2 * step) {
    const float4 x0 = src[i], x1 = src[i + step];
    dst[i] = vec4(i, x0);
    dst[i + step] = vec4(i + step, x1);
  }
  if (i < nvec) dst[i] = vec4(i, src[i]);
}

static inline bool f32v_ok(const void* a, const void* b, int64_t n) {
  return n >= 4096 && (n % 4) == 0 && (((uintptr_t)a | (uintptr_t)b) % 16) == 0;
}

// complex -> real outputs (abs / real / imag)
template <int DT>
__global__ __launch_bounds__(256) void unary_to_real_kernel(int op, typename Tr<DT>::R* __restrict__ dst,
                                                            const typename Tr<DT>::S* __restrict__ src,
                                                            int64_t n) {
  using R = typename Tr<DT>::R;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const auto z = Tr<DT>::ld(src, i);
    R r;
    if (op == TNH_OP_ABS) r = m_hypot(z.re, z.im);
    else if (op == TNH_OP_REAL) r = z.re;
    else r = z.im;
    dst[i] = r;
  }
}

struct BinParams {
  int rank;
  int64_t total;
  int64_t shape[TNH_MAX_RANK];
  int64_t sa[TNH_MAX_RANK];
  int64_t sb[TNH_MAX_RANK];
};

template <int DT>
__global__ __launch_bounds__(256) void binary_kernel(int op, typename Tr<DT>::S* __restrict__ dst,
                                                     const typename Tr<DT>::S* __restrict__ a,
                                                     const typename Tr<DT>::S* __restrict__ b,
                                                     BinParams p) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.total; i += step) {
    int64_t rem = i, oa = 0, ob = 0;
#pragma unroll 1
    for (int d = p.rank - 1; d >= 0; --d) {
      const int64_t q = rem / p.shape[d];
      const int64_t c = rem - q * p.shape[d];
      oa += c * p.sa[d];
      ob += c * p.sb[d];
      rem = q;
    }
    Tr<DT>::st(dst, i, apply_binary(op, Tr<DT>::ld(a, oa), Tr<DT>::ld(b, ob)));
  }
}

// Fast path: dst[r, c] = a[r, c] (op) b[c]   or   a[r, c] (op) b[r]
// (row / column scaling: broadcast_right / broadcast_left multiplication).
template <int DT, bool BY_COL>
__global__ __launch_bounds__(256) void binary_rowcol_kernel(int op, typename Tr<DT>::S* __restrict__ dst,
                                                            const typename Tr<DT>::S* __restrict__ a,
                                                            const typename Tr<DT>::S* __restrict__ v,
                                                            int64_t rows, int64_t cols, int vec_left) {
  const int64_t total = rows * cols;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
    const int64_t r = i / cols;
    const int64_t c = i - r * cols;
    const auto x = Tr<DT>::ld(a, i);
    const auto s = Tr<DT>::ld(v, BY_COL ? c : r);
    Tr<DT>::st(dst, i, vec_left ? apply_binary(op, s, x) : apply_binary(op, x, s));
  }
}

template <int DT>
__global__ __launch_bounds__(256) void binary_scalar_kernel(int op, typename Tr<DT>::S* __restrict__ dst,
                                                            const typename Tr<DT>::S* __restrict__ src,
                                                            double re, double im, int left, int64_t n) {
  using C = typename Tr<DT>::C;
  const C s = from_scalar(C{}, re, im);
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const C x = Tr<DT>::ld(src, i);
    Tr<DT>::st(dst, i, left ? apply_binary(op, s, x) : apply_binary(op, x, s));
  }
}

// Comparisons (real dtypes) -> int32 0/1 mask, and the select that consumes it: the two
// halves of `tensor[tensor <= eps] = value` (AbstractBackend.index_update with a scalar
// assignee, numpy_backend.py:548-552; caller infinite_mps.py:237-241).
template <int DT>
__global__ __launch_bounds__(256) void compare_kernel(int op, int32_t* __restrict__ dst,
                                                      const typename Tr<DT>::S* __restrict__ a,
                                                      const typename Tr<DT>::S* __restrict__ b, double scalar,
                                                      int64_t n) {
  using C = typename Tr<DT>::C;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const C x = Tr<DT>::ld(a, i);
    const C y = b ? Tr<DT>::ld(b, i) : (C)scalar;
    bool r;
    switch (op) {
      case TNH_CMP_LT: r = x < y; break;
      case TNH_CMP_LE: r = x <= y; break;
      case TNH_CMP_GT: r = x > y; break;
      case TNH_CMP_GE: r = x >= y; break;
      case TNH_CMP_EQ: r = x == y; break;
      default: r = x != y; break;
    }
    dst[i] = r ? 1 : 0;
  }
}

// ---- index_update with a TENSOR assignee (round 6): t = copy(tensor); t[mask] = values, the k-th set position of the
// mask (row-major order) takes values[k] (numpy_backend.py:548-552).  Three small launches, no host round trip of the
// mask: set entries per block of 1024, an exclusive scan of the block counts by one workgroup, and the scatter, where
// every block rebuilds its local ranks from wave ballots.  ESZ = element bytes (2 ... 16: the values are moved, not read).
constexpr int MS_BLOCK = 1024;      // elements per workgroup (256 threads x 4 consecutive elements)

__global__ __launch_bounds__(256) void mask_count_kernel(const int32_t* __restrict__ mask, int64_t n,
                                                         int64_t* __restrict__ counts) {
  __shared__ int wsum[4];
  const int tid = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.x * MS_BLOCK + 4 * tid;
  int c = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) c += (i0 + e < n && mask[i0 + e] != 0) ? 1 : 0;
  c = wave_sum(c);
  if ((tid & 63) == 0) wsum[tid >> 6] = c;
  __syncthreads();
  if (tid == 0) counts[blockIdx.x] = (int64_t)wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// counts[b] -> exclusive prefix (in place); total[0] = number of set entries.  One workgroup walks the blocks 1024 at a time.
__global__ __launch_bounds__(1024) void mask_scan_kernel(int64_t* __restrict__ counts, int64_t nb, int64_t* __restrict__ total) {
  __shared__ int64_t wsum[16];
  __shared__ int64_t carry;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int64_t b0 = 0; b0 < nb; b0 += 1024) {
    const int64_t b = b0 + tid;
    const int64_t v = b < nb ? counts[b] : 0;
    int64_t incl = v;                                  // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int64_t up = __shfl_up(incl, off, 64);
      if (lane >= off) incl += up;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int64_t before = carry;
    for (int ww = 0; ww < w; ++ww) before += wsum[ww];
    if (b < nb) counts[b] = before + incl - v;
    __syncthreads();
    if (tid == 1023) carry = before + incl;
    __syncthreads();
  }
  if (tid == 0) total[0] = carry;
}

template <int ESZ>
struct alignas(ESZ) RawElem {
  unsigned char b[ESZ];
};
template <int ESZ>
__global__ __launch_bounds__(256) void masked_scatter_kernel(RawElem<ESZ>* __restrict__ dst, const RawElem<ESZ>* __restrict__ src,
                                                             const int32_t* __restrict__ mask,
                                                             const RawElem<ESZ>* __restrict__ values, int64_t nvalues,
                                                             int64_t n, const int64_t* __restrict__ offsets) {
  __shared__ int wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t i0 = (int64_t)blockIdx.x * MS_BLOCK + 4 * tid;
  bool m[4];
  int c = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    m[e] = i0 + e < n && mask[i0 + e] != 0;
    c += m[e] ? 1 : 0;
  }
  int incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int64_t rank = offsets[blockIdx.x] + (incl - c);
  for (int ww = 0; ww < w; ++ww) rank += wsum[ww];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (i0 + e >= n) break;
    if (m[e]) {
      dst[i0 + e] = values[rank < nvalues ? rank : nvalues - 1];     // (a count mismatch is reported by the caller)
      ++rank;
    } else {
      dst[i0 + e] = src[i0 + e];
    }
  }
}

template <int DT>
__global__ __launch_bounds__(256) void masked_fill_kernel(typename Tr<DT>::S* __restrict__ dst,
                                                          const typename Tr<DT>::S* __restrict__ src,
                                                          const int32_t* __restrict__ mask, double re, double im,
                                                          int64_t n) {
  using C = typename Tr<DT>::C;
  const C s = from_scalar(C{}, re, im);
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step)
    Tr<DT>::st(dst, i, mask[i] ? s : Tr<DT>::ld(src, i));
}

template <int DT>
__global__ __launch_bounds__(256) void fill_kernel(typename Tr<DT>::S* __restrict__ dst, double re,
                                                   double im, int64_t n) {
  using C = typename Tr<DT>::C;
  const C s = from_scalar(C{}, re, im);
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step)
    Tr<DT>::st(dst, i, s);
}

template <int DT>
__global__ __launch_bounds__(256) void eye_kernel(typename Tr<DT>::S* __restrict__ dst, int64_t rows,
                                                  int64_t cols) {
  using C = typename Tr<DT>::C;
  const int64_t n = rows * cols;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const int64_t r = i / cols, c = i - r * cols;
    Tr<DT>::st(dst, i, from_scalar(C{}, r == c ? 1.0 : 0.0, 0.0));
  }
}


// Counter-based generator for synthetic benchmark operands (NOT the backend's
// randn: that one reproduces NumPy's stream on the host).  splitmix64 of
// (seed, index) -> two 24-bit uniforms -> Box-Muller.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

template <int DT>
__global__ __launch_bounds__(256) void random_kernel(typename Tr<DT>::S* __restrict__ dst, int64_t n,
                                                     uint64_t seed, int normal, double a, double b) {
  using C = typename Tr<DT>::C;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const uint64_t h = splitmix64(seed * 0x100000001b3ull + (uint64_t)i);
    const float u1 = ((float)((h >> 40) & 0xffffff) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)((h >> 8) & 0xffffff) + 0.5f) * (1.0f / 16777216.0f);
    double re, im = 0.0;
    if (normal) {
      const float r = sqrtf(-2.0f * logf(u1));
      re = a + b * (double)(r * cosf(6.2831853071795864f * u2));
      im = a + b * (double)(r * sinf(6.2831853071795864f * u2));
    } else {
      re = a + (b - a) * (double)u1;
      im = a + (b - a) * (double)u2;
    }
    Tr<DT>::st(dst, i, from_scalar(C{}, re, im));
  }
}

__device__ __forceinline__ double to_double(float x) { return (double)x; }
__device__ __forceinline__ double to_double(double x) { return x; }
__device__ __forceinline__ double to_double(int32_t x) { return (double)x; }
__device__ __forceinline__ double to_double(int64_t x) { return (double)x; }

template <int SRC, int DST>
__global__ __launch_bounds__(256) void cast_kernel(typename Tr<DST>::S* __restrict__ dst,
                                                   const typename Tr<SRC>::S* __restrict__ src,
                                                   int64_t n) {
  using CS = typename Tr<SRC>::C;
  using CD = typename Tr<DST>::C;
  constexpr bool src_cplx = (SRC == TNH_C64 || SRC == TNH_C128);
  constexpr bool dst_cplx = (DST == TNH_C64 || DST == TNH_C128);
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const CS x = Tr<SRC>::ld(src, i);
    double re, im;
    if constexpr (src_cplx) { re = (double)x.re; im = (double)x.im; }
    else { re = to_double(x); im = 0.0; }
    CD y;
    if constexpr (dst_cplx) y = from_scalar(CD{}, re, im);
    else y = from_scalar(CD{}, re, 0.0);
    Tr<DST>::st(dst, i, y);
  }
}

static inline unsigned grid_for(int64_t n, int64_t bytes_per_item = 4) {
  int64_t blocks = (n + 255) / 256;
  const int64_t cap = (int64_t)num_cus() * stream_wgs_per_cu(n * bytes_per_item);
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

template <int SRC>
static int cast_from(void* dst, int dst_dtype, const void* src, int64_t n) {
  const unsigned g = grid_for(n);
#define TNH_CAST_CASE(D)                                                                          \
  case D:                                                                                         \
    hipLaunchKernelGGL((cast_kernel<SRC, D>), dim3(g), dim3(256), 0, stream(),                    \
                       (typename Tr<D>::S*)dst, (const typename Tr<SRC>::S*)src, n);              \
    break;
  switch (dst_dtype) {
    TNH_CAST_CASE(TNH_F32)
    TNH_CAST_CASE(TNH_F64)
    TNH_CAST_CASE(TNH_BF16)
    TNH_CAST_CASE(TNH_F16)
    TNH_CAST_CASE(TNH_C64)
    TNH_CAST_CASE(TNH_C128)
    TNH_CAST_CASE(TNH_I32)
    TNH_CAST_CASE(TNH_I64)
    default:
      set_error("unsupported cast target %d", dst_dtype);
      return TNH_ERR_UNSUPPORTED;
  }
#undef TNH_CAST_CASE
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

}  // namespace tnh

using namespace tnh;

namespace tnh {
// int64 storage of a narrow / unsigned / bool tensor brought back to the canonical value of its NumPy dtype:
// mode 0: the low `bits` zero-extended, 1: sign-extended, 2: x != 0.
__global__ __launch_bounds__(256) void wrap_int_kernel(int64_t* __restrict__ dst, const int64_t* __restrict__ src, int64_t n,
                                                       int bits, int mode) {
  const int64_t step = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += step) {
    const int64_t x = src[i];
    int64_t y;
    if (mode == 2) y = (x != 0) ? 1 : 0;
    else if (bits == 64) y = x;
    else if (mode == 0) y = (int64_t)((uint64_t)x & (((uint64_t)1 << bits) - 1));
    else y = (int64_t)((uint64_t)x << (64 - bits)) >> (64 - bits);
    dst[i] = y;
  }
}
}  // namespace tnh

extern "C" {

int tnh_unary(int op, void* dst, const void* src, int64_t n, int dtype) {
  TNH_NEED_INIT();
  TNH_REQUIRE(op >= TNH_OP_SQRT && op <= TNH_OP_IMAG, "bad unary op %d", op);
  TNH_REQUIRE(n >= 0, "negative size");
  if (n == 0) return TNH_OK;
  TNH_REQUIRE(dst && src, "null pointer");
  const unsigned g = grid_for(n);
  const bool cplx = (dtype == TNH_C64 || dtype == TNH_C128);
  if (cplx && (op == TNH_OP_ABS || op == TNH_OP_REAL || op == TNH_OP_IMAG)) {
    if (dtype == TNH_C64)
      hipLaunchKernelGGL((unary_to_real_kernel<TNH_C64>), dim3(g), dim3(256), 0, stream(), op,
                         (float*)dst, (const cf32*)src, n);
    else
      hipLaunchKernelGGL((unary_to_real_kernel<TNH_C128>), dim3(g), dim3(256), 0, stream(), op,
                         (double*)dst, (const cf64*)src, n);
    TNH_LAUNCH_CHECK();
    return TNH_OK;
  }
  if (dtype == TNH_F32 && f32v_ok(dst, src, n)) {
    hipLaunchKernelGGL((f32v_kernel<0>), dim3(grid_for(n / 8)), dim3(256), 0, stream(), op, (float4*)dst, (const float4*)src,
                       (const float*)nullptr, 0.f, 0, n / 4, (int64_t)1);
    TNH_LAUNCH_CHECK();
    return TNH_OK;
  }
  TNH_DISPATCH_NUM(dtype, hipLaunchKernelGGL((unary_kernel<DT>), dim3(g), dim3(256), 0, stream(), op,
                                               (typename Tr<DT>::S*)dst,
                                               (const typename Tr<DT>::S*)src, n));
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

int tnh_binary(int op, void* dst, const void* a, const void* b, int rank, const int64_t* shape,
               const int64_t* a_strides, const int64_t* b_strides, int dtype) {
  TNH_NEED_INIT();
  TNH_REQUIRE(op >= TNH_OP_ADD && op <= TNH_OP_POW, "bad binary op %d", op);
  TNH_REQUIRE(rank >= 0 && rank <= TNH_MAX_RANK, "rank %d out of range", rank);
  BinParams p;
  p.rank = 0;
  p.total = 1;
  for (int d = 0; d < rank; ++d) {
    TNH_REQUIRE(shape[d] >= 0, "negative dimension");
    p.total *= shape[d];
    if (shape[d] == 1) continue;
    if (p.rank > 0 && p.sa[p.rank - 1] == a_strides[d] * shape[d] &&
        p.sb[p.rank - 1] == b_strides[d] * shape[d]) {
      p.shape[p.rank - 1] *= shape[d];
      p.sa[p.rank - 1] = a_strides[d];
      p.sb[p.rank - 1] = b_strides[d];
    } else {
      p.shape[p.rank] = shape[d];
      p.sa[p.rank] = a_strides[d];
      p.sb[p.rank] = b_strides[d];
      ++p.rank;
    }
  }
  if (p.total == 0) return TNH_OK;
  TNH_REQUIRE(dst && a && b, "null pointer");
  const unsigned g = grid_for(p.total);
  // (rows, cols) op vector fast paths
  if (p.rank == 2 && p.sa[0] == p.shape[1] && p.sa[1] == 1 && p.sb[0] == 0 && p.sb[1] == 1 && dtype == TNH_F32 &&
      p.shape[1] % 4 == 0 && f32v_ok(dst, a, p.total) && ((uintptr_t)b % 16) == 0) {
    hipLaunchKernelGGL((f32v_kernel<2>), dim3(grid_for(p.total / 8)), dim3(256), 0, stream(), op, (float4*)dst,
                       (const float4*)a, (const float*)b, 0.f, 0, p.total / 4, p.shape[1] / 4);
    TNH_LAUNCH_CHECK();
    return TNH_OK;
  }
  if (p.rank == 2 && p.sa[0] == p.shape[1] && p.sa[1] == 1 && p.sb[0] == 0 && p.sb[1] == 1) {
    TNH_DISPATCH_NUM(dtype, hipLaunchKernelGGL((binary_rowcol_kernel<DT, true>), dim3(g), dim3(256), 0,
                                                 stream(), op, (typename Tr<DT>::S*)dst,
                                                 (const typename Tr<DT>::S*)a,
                                                 (const typename Tr<DT>::S*)b, p.shape[0], p.shape[1], 0));
    TNH_LAUNCH_CHECK();
    return TNH_OK;
  }
  if (p.rank == 2 && p.sb[0] == p.shape[1] && p.sb[1] == 1 && p.sa[0] == 1 && p.sa[1] == 0 && dtype == TNH_F32 &&
      p.shape[1] % 4 == 0 && f32v_ok(dst, b, p.total)) {
    hipLaunchKernelGGL((f32v_kernel<3>), dim3(grid_for(p.total / 8)), dim3(256), 0, stream(), op, (float4*)dst,
                       (const float4*)b, (const float*)a, 0.f, 1, p.total / 4, p.shape[1] / 4);
    TNH_LAUNCH_CHECK();
    return TNH_OK;
  }
  if (p.rank == 2 && p.sb[0] == p.shape[1] && p.sb[1] == 1 && p.sa[0] == 1 && p.sa[1] == 0) {
    TNH_DISPATCH_NUM(dtype, hipLaunchKernelGGL((binary_rowcol_kernel<DT, false>), dim3(g), dim3(256), 0,
                                                 stream(), op, (typename Tr<DT>::S*)dst,
                                                 (const typename Tr<DT>::S*)b,
                                                 (const typename Tr<DT>::S*)a, p.shape[0], p.shape[1], 1));
    TNH_LAUNCH_CHECK();
    return TNH_OK;
  }
  if (p.rank == 2 && p.sa[0] == p.shape[1] && p.sa[1] == 1 && p.sb[0] == 1 && p.sb[1] == 0 && dtype == TNH_F32 &&
      p.shape[1] % 4 == 0 && f32v_ok(dst, a, p.total)) {
    hipLaunchKernelGGL((f32v_kernel<3>), dim3(grid_for(p.total / 8)), dim3(256), 0, stream(), op, (float4*)dst,
                       (const float4*)a, (const float*)b, 0.f, 0, p.total / 4, p.shape[1] / 4);
    TNH_LAUNCH_CHECK();
    return TNH_OK;
  }
  if (p.rank == 2 && p.sa[0] == p.shape[1] && p.sa[1] == 1 && p.sb[0] == 1 && p.sb[1] == 0) {
    TNH_DISPATCH_NUM(dtype, hipLaunchKernelGGL((binary_rowcol_kernel<DT, false>), dim3(g), dim3(256), 0,
                                                 stream(), op, (typename Tr<DT>::S*)dst,
                                                 (const typename Tr<DT>::S*)a,
                                                 (const typename Tr<DT>::S*)b, p.shape[0], p.shape[1], 0));
    TNH_LAUNCH_CHECK();
    return TNH_OK;
  }
  if (p.rank == 0) {  // scalar (op) scalar
    p.rank = 1;
    p.shape[0] = 1;
    p.sa[0] = 0;
    p.sb[0] = 0;
  }
  TNH_DISPATCH_NUM(dtype, hipLaunchKernelGGL((binary_kernel<DT>), dim3(g), dim3(256), 0, stream(), op,
                                               (typename Tr<DT>::S*)dst, (const typename Tr<DT>::S*)a,
                                               (const typename Tr<DT>::S*)b, p));
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

int tnh_binary_scalar(int op, void* dst, const void* src, double re, double im, int scalar_left,
                      int64_t n, int dtype) {
  TNH_NEED_INIT();
  TNH_REQUIRE(op >= TNH_OP_ADD && op <= TNH_OP_POW, "bad binary op %d", op);
  TNH_REQUIRE(n >= 0, "negative size");
  if (n == 0) return TNH_OK;
  TNH_REQUIRE(dst && src, "null pointer");
  const unsigned g = grid_for(n);
  if (dtype == TNH_F32 && f32v_ok(dst, src, n)) {
    hipLaunchKernelGGL((f32v_kernel<1>), dim3(grid_for(n / 8)), dim3(256), 0, stream(), op, (float4*)dst,
                       (const float4*)src, (const float*)nullptr, (float)re, scalar_left, n / 4, (int64_t)1);
    TNH_LAUNCH_CHECK();
    return TNH_OK;
  }
  TNH_DISPATCH_NUM(dtype, hipLaunchKernelGGL((binary_scalar_kernel<DT>), dim3(g), dim3(256), 0, stream(),
                                               op, (typename Tr<DT>::S*)dst,
                                               (const typename Tr<DT>::S*)src, re, im, scalar_left, n));
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

int tnh_compare(int op, void* dst, const void* a, const void* b, double scalar, int64_t n, int dtype) {
  TNH_NEED_INIT();
  TNH_REQUIRE(op >= TNH_CMP_LT && op <= TNH_CMP_NE, "bad comparison op %d", op);
  TNH_REQUIRE(dtype == TNH_F32 || dtype == TNH_F64 || dtype == TNH_BF16 || dtype == TNH_F16,
              "comparisons need a real floating dtype (got %d)", dtype);
  TNH_REQUIRE(n >= 0, "negative size");
  if (n == 0) return TNH_OK;
  TNH_REQUIRE(dst && a, "null pointer");
  const unsigned g = grid_for(n);
  switch (dtype) {
    case TNH_F32: hipLaunchKernelGGL((compare_kernel<TNH_F32>), dim3(g), dim3(256), 0, stream(), op, (int32_t*)dst, (const float*)a, (const float*)b, scalar, n); break;
    case TNH_F64: hipLaunchKernelGGL((compare_kernel<TNH_F64>), dim3(g), dim3(256), 0, stream(), op, (int32_t*)dst, (const double*)a, (const double*)b, scalar, n); break;
    case TNH_BF16: hipLaunchKernelGGL((compare_kernel<TNH_BF16>), dim3(g), dim3(256), 0, stream(), op, (int32_t*)dst, (const uint16_t*)a, (const uint16_t*)b, scalar, n); break;
    default: hipLaunchKernelGGL((compare_kernel<TNH_F16>), dim3(g), dim3(256), 0, stream(), op, (int32_t*)dst, (const uint16_t*)a, (const uint16_t*)b, scalar, n); break;
  }
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

int tnh_masked_fill(void* dst, const void* src, const void* mask, double re, double im, int64_t n, int dtype) {
  TNH_NEED_INIT();
  TNH_REQUIRE(n >= 0, "negative size");
  if (n == 0) return TNH_OK;
  TNH_REQUIRE(dst && src && mask, "null pointer");
  const unsigned g = grid_for(n);
  TNH_DISPATCH_NUM(dtype, hipLaunchKernelGGL((masked_fill_kernel<DT>), dim3(g), dim3(256), 0, stream(),
                                               (typename Tr<DT>::S*)dst, (const typename Tr<DT>::S*)src,
                                               (const int32_t*)mask, re, im, n));
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

int tnh_masked_scatter(void* dst, const void* src, const void* mask, const void* values, int64_t nvalues, int64_t n,
                       int itemsize, int64_t* count_out) {
  TNH_NEED_INIT();
  TNH_REQUIRE(n >= 0 && nvalues >= 0, "negative size");
  TNH_REQUIRE(itemsize == 2 || itemsize == 4 || itemsize == 8 || itemsize == 16, "tnh_masked_scatter: item size %d", itemsize);
  TNH_REQUIRE(!(count_out && capturing()), "tnh_masked_scatter: the count read-back synchronises the stream (graph capture)");
  if (count_out) *count_out = 0;
  if (n == 0) return TNH_OK;
  TNH_REQUIRE(dst && src && mask && (values || nvalues == 0), "null pointer");
  const int64_t nb = (n + MS_BLOCK - 1) / MS_BLOCK;
  void* work = nullptr;
  int rc = tnh_malloc(&work, (size_t)(nb + 1) * sizeof(int64_t));
  if (rc) return rc;
  int64_t* counts = (int64_t*)work;
  hipLaunchKernelGGL(mask_count_kernel, dim3((unsigned)nb), dim3(256), 0, stream(), (const int32_t*)mask, n, counts);
  hipLaunchKernelGGL(mask_scan_kernel, dim3(1), dim3(1024), 0, stream(), counts, nb, counts + nb);
  const int64_t nv = nvalues > 0 ? nvalues : 1;
  switch (itemsize) {
    case 2: hipLaunchKernelGGL((masked_scatter_kernel<2>), dim3((unsigned)nb), dim3(256), 0, stream(), (RawElem<2>*)dst, (const RawElem<2>*)src, (const int32_t*)mask, (const RawElem<2>*)(values ? values : src), nv, n, (const int64_t*)counts); break;
    case 4: hipLaunchKernelGGL((masked_scatter_kernel<4>), dim3((unsigned)nb), dim3(256), 0, stream(), (RawElem<4>*)dst, (const RawElem<4>*)src, (const int32_t*)mask, (const RawElem<4>*)(values ? values : src), nv, n, (const int64_t*)counts); break;
    case 8: hipLaunchKernelGGL((masked_scatter_kernel<8>), dim3((unsigned)nb), dim3(256), 0, stream(), (RawElem<8>*)dst, (const RawElem<8>*)src, (const int32_t*)mask, (const RawElem<8>*)(values ? values : src), nv, n, (const int64_t*)counts); break;
    default: hipLaunchKernelGGL((masked_scatter_kernel<16>), dim3((unsigned)nb), dim3(256), 0, stream(), (RawElem<16>*)dst, (const RawElem<16>*)src, (const int32_t*)mask, (const RawElem<16>*)(values ? values : src), nv, n, (const int64_t*)counts); break;
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && count_out) {
    e = hipMemcpyAsync(count_out, counts + nb, sizeof(int64_t), hipMemcpyDeviceToHost, stream());
    if (e == hipSuccess) e = hipStreamSynchronize(stream());
  }
  tnh_free(work);
  if (e != hipSuccess) {
    set_error("tnh_masked_scatter failed: %s", hipGetErrorString(e));
    return TNH_ERR_HIP;
  }
  return TNH_OK;
}

int tnh_fill(void* dst, double re, double im, int64_t n, int dtype) {
  TNH_NEED_INIT();
  TNH_REQUIRE(n >= 0, "negative size");
  if (n == 0) return TNH_OK;
  TNH_REQUIRE(dst != nullptr, "null pointer");
  const unsigned g = grid_for(n);
  TNH_DISPATCH_NUM(dtype, hipLaunchKernelGGL((fill_kernel<DT>), dim3(g), dim3(256), 0, stream(),
                                               (typename Tr<DT>::S*)dst, re, im, n));
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

int tnh_eye(void* dst, int64_t rows, int64_t cols, int dtype) {
  TNH_NEED_INIT();
  TNH_REQUIRE(rows >= 0 && cols >= 0, "negative size");
  if (rows * cols == 0) return TNH_OK;
  TNH_REQUIRE(dst != nullptr, "null pointer");
  const unsigned g = grid_for(rows * cols);
  TNH_DISPATCH_NUM(dtype, hipLaunchKernelGGL((eye_kernel<DT>), dim3(g), dim3(256), 0, stream(),
                                               (typename Tr<DT>::S*)dst, rows, cols));
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

int tnh_random(void* dst, int64_t n, int dtype, uint64_t seed, int normal, double a, double b) {
  TNH_NEED_INIT();
  TNH_REQUIRE(n >= 0, "negative size");
  if (n == 0) return TNH_OK;
  TNH_REQUIRE(dst != nullptr, "null pointer");
  const unsigned g = grid_for(n);
  TNH_DISPATCH_FLOAT(dtype, hipLaunchKernelGGL((random_kernel<DT>), dim3(g), dim3(256), 0, stream(),
                                               (typename Tr<DT>::S*)dst, n, seed, normal, a, b));
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

int tnh_wrap_int(void* dst, const void* src, int64_t n, int bits, int mode) {
  TNH_NEED_INIT();
  TNH_REQUIRE(n >= 0, "negative size");
  TNH_REQUIRE(mode >= 0 && mode <= 2, "tnh_wrap_int: mode %d", mode);
  TNH_REQUIRE(mode == 2 || bits == 8 || bits == 16 || bits == 32 || bits == 64, "tnh_wrap_int: %d bits", bits);
  if (n == 0) return TNH_OK;
  TNH_REQUIRE(dst && src, "null pointer");
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(tnh::wrap_int_kernel, dim3((unsigned)(blocks < 65536 * 16 ? blocks : 65536 * 16)), dim3(256), 0,
                     tnh::stream(), (int64_t*)dst, (const int64_t*)src, n, bits, mode);
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

int tnh_cast(void* dst, int dst_dtype, const void* src, int src_dtype, int64_t n) {
  TNH_NEED_INIT();
  TNH_REQUIRE(n >= 0, "negative size");
  if (n == 0) return TNH_OK;
  TNH_REQUIRE(dst && src, "null pointer");
  switch (src_dtype) {
    case TNH_F32: return cast_from<TNH_F32>(dst, dst_dtype, src, n);
    case TNH_F64: return cast_from<TNH_F64>(dst, dst_dtype, src, n);
    case TNH_BF16: return cast_from<TNH_BF16>(dst, dst_dtype, src, n);
    case TNH_F16: return cast_from<TNH_F16>(dst, dst_dtype, src, n);
    case TNH_C64: return cast_from<TNH_C64>(dst, dst_dtype, src, n);
    case TNH_C128: return cast_from<TNH_C128>(dst, dst_dtype, src, n);
    case TNH_I32: return cast_from<TNH_I32>(dst, dst_dtype, src, n);
    case TNH_I64: return cast_from<TNH_I64>(dst, dst_dtype, src, n);
    default:
      set_error("unsupported cast source %d", src_dtype);
      return TNH_ERR_UNSUPPORTED;
  }
}

}  // extern "C"
