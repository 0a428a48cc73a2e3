// K3/K4: trace over the last two axes, axis sums and the Frobenius norm.
// HBM-bound: every input element is read once (algorithmic bytes =
// numel_in * itemsize + numel_out * itemsize).  Reductions accumulate in f32
// (f32/bf16/f16/c64) or f64 (f64/c128); lanes reduce with wavefront shuffles,
// long reductions are split over workgroups with partials in a runtime-owned
// scratch buffer and combined by a second pass (deterministic, no atomics).
#include "tnh_types.h"

namespace tnh {

static void* g_scratch = nullptr;
static const size_t kScratchBytes = size_t(8) << 20;

static int ensure_scratch() {
  if (g_scratch) return TNH_OK;
  TNH_HIP(hipMalloc(&g_scratch, kScratchBytes));
  return TNH_OK;
}

template <typename C> struct AccStore;  // accumulate-type storage in scratch
template <> struct AccStore<float> { using S = float; };
template <> struct AccStore<double> { using S = double; };
template <> struct AccStore<cf32> { using S = cf32; };
template <> struct AccStore<cf64> { using S = cf64; };
template <> struct AccStore<int32_t> { using S = int32_t; };
template <> struct AccStore<int64_t> { using S = int64_t; };

__device__ __forceinline__ float tf_abs2(float x) { return x * x; }
__device__ __forceinline__ double tf_abs2(double x) { return x * x; }
__device__ __forceinline__ cf32 tf_abs2(cf32 x) { return {x.re * x.re + x.im * x.im, 0.f}; }
__device__ __forceinline__ cf64 tf_abs2(cf64 x) { return {x.re * x.re + x.im * x.im, 0.0}; }
__device__ __forceinline__ int32_t tf_abs2(int32_t x) { return x * x; }
__device__ __forceinline__ int64_t tf_abs2(int64_t x) { return x * x; }
__device__ __forceinline__ int32_t tf_sqrt(int32_t x) { return x; }   // norm of integer tensors is taken in float on the host
__device__ __forceinline__ int64_t tf_sqrt(int64_t x) { return x; }
__device__ __forceinline__ float tf_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double tf_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ cf32 tf_sqrt(cf32 x) { return {sqrtf(x.re), 0.f}; }
__device__ __forceinline__ cf64 tf_sqrt(cf64 x) { return {sqrt(x.re), 0.0}; }

// Row reduction: out[row * nsplit + s] = sum over r in split s of in[row*R*stride + r*stride].
// One wave per (row, split); 4 waves per workgroup.
//   IN_ACC : input already holds accumulate-type partials (second pass)
//   OUT_ACC: write accumulate-type partials (first pass of a split reduction)
template <int DT, bool IN_ACC, bool OUT_ACC, bool SQUARE, bool ROOT>
__global__ __launch_bounds__(256) void reduce_rows_kernel(void* __restrict__ out,
                                                          const void* __restrict__ in, int64_t rows,
                                                          int64_t R, int64_t stride, int64_t row_stride,
                                                          int nsplit) {
  using C = typename Tr<DT>::C;
  using S = typename Tr<DT>::S;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwork = rows * nsplit;
  if (wave >= nwork) return;
  const int64_t row = wave / nsplit;
  const int s = (int)(wave - row * nsplit);
  const int64_t chunk = (R + nsplit - 1) / nsplit;
  const int64_t r0 = (int64_t)s * chunk;
  int64_t r1 = r0 + chunk;
  if (r1 > R) r1 = R;
  C acc = zero_of(C{});
  for (int64_t r = r0 + lane; r < r1; r += 64) {
    C v;
    if (IN_ACC) v = ((const C*)in)[row * row_stride + r * stride];
    else v = Tr<DT>::ld((const S*)in, row * row_stride + r * stride);
    if (SQUARE) v = tf_abs2(v);
    acc = acc + v;
  }
  acc = wave_sum_t(acc);
  if (lane == 0) {
    if (ROOT) acc = tf_sqrt(acc);
    if (OUT_ACC) ((C*)out)[wave] = acc;
    else Tr<DT>::st((S*)out, wave, acc);
  }
}

// f32 fast paths (16-byte loads, two accumulators per lane): contiguous rows, and axis sums whose
// inner extent is a small power of two (then (r, i) of one outer index is ONE contiguous run:
// lanes stream it as float4 and the lanes that hold the same i are combined with shuffles).
// Round-1 kernels read 4 bytes per lane: 1.3 TB/s (full sum) / 2.4 TB/s (axis sum, inner = 16).
__global__ __launch_bounds__(256) void reduce_rows_f32v_kernel(float* __restrict__ out, const float* __restrict__ in,
                                                                int64_t rows, int64_t R, int64_t row_stride,
                                                                int nsplit, int square, int root) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= rows * nsplit) return;
  const int64_t row = wave / nsplit;
  const int s = (int)(wave - row * nsplit);
  // Block-cyclic split: wave s takes the 4-KiB blocks s, s + nsplit, s + 2 nsplit, ... of the row, so that all waves
  // together sweep the row as one front instead of streaming 4096 separate segments (1 GiB: 5.1-5.3 -> 5.44 TB/s;
  // 1-KiB blocks with one load in flight per lane: 5.25).
  const float* p = in + row * row_stride;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const int64_t full = R & ~(int64_t)1023;           // whole 1024-float blocks
  const int64_t step = (int64_t)nsplit * 1024;
  for (int64_t r = (int64_t)s * 1024 + 4 * lane; r < full; r += step) {   // four float4 per lane per iteration
    const float4 x = *(const float4*)(p + r), y = *(const float4*)(p + r + 256);
    const float4 z = *(const float4*)(p + r + 512), w = *(const float4*)(p + r + 768);
    if (square) {
      a0 += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
      a1 += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
      a2 += z.x * z.x + z.y * z.y + z.z * z.z + z.w * z.w;
      a3 += w.x * w.x + w.y * w.y + w.z * w.z + w.w * w.w;
    } else {
      a0 += (x.x + x.y) + (x.z + x.w);
      a1 += (y.x + y.y) + (y.z + y.w);
      a2 += (z.x + z.y) + (z.z + z.w);
      a3 += (w.x + w.y) + (w.z + w.w);
    }
  }
  a0 += a2;
  a1 += a3;
  if (s == nsplit - 1) {                              // the ragged tail of the row
    for (int64_t r = full + lane; r < R; r += 64) {
      const float v = p[r];
      a0 += square ? v * v : v;
    }
  }
  float acc = wave_sum_t(a0 + a1);
  if (lane == 0) out[wave] = root ? sqrtf(acc) : acc;
}

// grid: outer * nsplit waves; inner in {4, 8, ..., 256}; out[(o * nsplit + s) * inner + i]
__global__ __launch_bounds__(256) void reduce_mid_f32v_kernel(float* __restrict__ out, const float* __restrict__ in,
                                                               int64_t outer, int64_t R, int inner, int nsplit) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= outer * nsplit) return;
  const int64_t o = wave / nsplit;
  const int s = (int)(wave - o * nsplit);
  const int rows_per_pass = 256 / inner;             // r rows covered by one wave-wide float4 load
  int64_t chunk = (R + nsplit - 1) / nsplit;
  chunk = (chunk + rows_per_pass - 1) / rows_per_pass * rows_per_pass;
  const int64_t r0 = (int64_t)s * chunk;
  int64_t r1 = r0 + chunk;
  if (r1 > R) r1 = R;
  const float* p = in + o * R * inner;
  const int64_t e1 = r1 * inner;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  int64_t e = r0 * inner + 4 * lane;
  for (; e + 256 < e1; e += 512) {
    const float4 x = *(const float4*)(p + e), y = *(const float4*)(p + e + 256);
    a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
    b.x += y.x; b.y += y.y; b.z += y.z; b.w += y.w;
  }
  for (; e < e1; e += 256) {
    const float4 x = *(const float4*)(p + e);
    a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
  }
  a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  // lanes l and l' hold the same four i when l == l' mod (inner / 4)
  for (int off = 32; off >= inner / 4; off >>= 1) {
    a.x += __shfl_xor(a.x, off, 64);
    a.y += __shfl_xor(a.y, off, 64);
    a.z += __shfl_xor(a.z, off, 64);
    a.w += __shfl_xor(a.w, off, 64);
  }
  if (lane < inner / 4) *(float4*)(out + wave * inner + 4 * lane) = a;
}

// Column reduction over an (outer, R, inner) view: thread per (o, i), i fastest.
template <int DT, bool IN_ACC, bool OUT_ACC>
__global__ __launch_bounds__(256) void reduce_mid_kernel(void* __restrict__ out,
                                                         const void* __restrict__ in, int64_t outer,
                                                         int64_t R, int64_t inner, int nsplit) {
  using C = typename Tr<DT>::C;
  using S = typename Tr<DT>::S;
  const int64_t cols = outer * inner;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= cols) return;
  const int s = blockIdx.y;
  const int64_t o = idx / inner, i = idx - o * inner;
  const int64_t chunk = (R + nsplit - 1) / nsplit;
  const int64_t r0 = (int64_t)s * chunk;
  int64_t r1 = r0 + chunk;
  if (r1 > R) r1 = R;
  C acc = zero_of(C{});
  const int64_t base = o * R * inner + i;
  for (int64_t r = r0; r < r1; ++r) {
    C v;
    if (IN_ACC) v = ((const C*)in)[base + r * inner];
    else v = Tr<DT>::ld((const S*)in, base + r * inner);
    acc = acc + v;
  }
  if (OUT_ACC) ((C*)out)[(int64_t)s * cols + idx] = acc;
  else Tr<DT>::st((S*)out, idx, acc);
}

template <int DT, bool SQUARE, bool ROOT>
static int reduce_rows(void* dst, const void* src, int64_t rows, int64_t R, int64_t stride,
                       int64_t row_stride) {
  using C = typename Tr<DT>::C;
  if constexpr (DT == TNH_F32) {
    if (stride == 1 && R >= 1024 && row_stride % 4 == 0 && ((uintptr_t)src % 16) == 0) {
      // enough waves in flight to cover HBM latency: 16 per CU, each with >= 2 KiB of work
      int64_t ns = ((int64_t)num_cus() * 16 + rows - 1) / (rows > 0 ? rows : 1);
      if (ns > R / 512) ns = R / 512;
      const int64_t cap = (int64_t)(kScratchBytes / sizeof(float)) / (rows > 0 ? rows : 1);
      if (ns > cap) ns = cap;
      if (ns < 1) ns = 1;
      if (ns > 1) {
        int rc = ensure_scratch();
        if (rc) return rc;
      }
      float* first = ns > 1 ? (float*)g_scratch : (float*)dst;
      hipLaunchKernelGGL(reduce_rows_f32v_kernel, dim3((unsigned)((rows * ns + 3) / 4)), dim3(256), 0, stream(), first,
                         (const float*)src, rows, R, row_stride, (int)ns, SQUARE ? 1 : 0, (ROOT && ns == 1) ? 1 : 0);
      TNH_LAUNCH_CHECK();
      if (ns > 1) {
        hipLaunchKernelGGL((reduce_rows_kernel<TNH_F32, true, false, false, ROOT>), dim3((unsigned)((rows + 3) / 4)),
                           dim3(256), 0, stream(), dst, (const void*)g_scratch, rows, ns, (int64_t)1, ns, 1);
        TNH_LAUNCH_CHECK();
      }
      return TNH_OK;
    }
  }
  // split long rows so that there are enough waves to fill 256 CUs
  int nsplit = 1;
  const int64_t want_waves = (int64_t)num_cus() * 8;
  if (rows < want_waves && R >= 8192) {
    int64_t s = want_waves / (rows > 0 ? rows : 1);
    const int64_t max_by_len = R / 2048;
    if (s > max_by_len) s = max_by_len;
    const int64_t max_by_scratch = (int64_t)(kScratchBytes / sizeof(C)) / (rows > 0 ? rows : 1);
    if (s > max_by_scratch) s = max_by_scratch;
    if (s > 1) nsplit = (int)s;
  }
  if (nsplit == 1) {
    const int64_t blocks = (rows + 3) / 4;
    hipLaunchKernelGGL((reduce_rows_kernel<DT, false, false, SQUARE, ROOT>), dim3((unsigned)blocks),
                       dim3(256), 0, stream(), dst, src, rows, R, stride, row_stride, 1);
    TNH_LAUNCH_CHECK();
    return TNH_OK;
  }
  int rc = ensure_scratch();
  if (rc) return rc;
  const int64_t blocks = (rows * nsplit + 3) / 4;
  hipLaunchKernelGGL((reduce_rows_kernel<DT, false, true, SQUARE, false>), dim3((unsigned)blocks),
                     dim3(256), 0, stream(), g_scratch, src, rows, R, stride, row_stride, nsplit);
  TNH_LAUNCH_CHECK();
  const int64_t blocks2 = (rows + 3) / 4;
  hipLaunchKernelGGL((reduce_rows_kernel<DT, true, false, false, ROOT>), dim3((unsigned)blocks2),
                     dim3(256), 0, stream(), dst, (const void*)g_scratch, rows, (int64_t)nsplit,
                     (int64_t)1, (int64_t)nsplit, 1);
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

template <int DT>
static int reduce_mid(void* dst, const void* src, int64_t outer, int64_t R, int64_t inner) {
  using C = typename Tr<DT>::C;
  const int64_t cols = outer * inner;
  if constexpr (DT == TNH_F32) {
    if (inner >= 4 && inner <= 256 && (inner & (inner - 1)) == 0 && R * inner >= 4096 && ((uintptr_t)src % 16) == 0 &&
        ((uintptr_t)dst % 16) == 0) {
      int64_t ns = ((int64_t)num_cus() * 16 + outer - 1) / outer;
      if (ns > R * inner / 2048) ns = R * inner / 2048;
      const int64_t cap = (int64_t)(kScratchBytes / sizeof(float)) / cols;
      if (ns > cap) ns = cap;
      if (ns < 1) ns = 1;
      if (ns > 1) {
        int rc = ensure_scratch();
        if (rc) return rc;
      }
      float* first = ns > 1 ? (float*)g_scratch : (float*)dst;
      hipLaunchKernelGGL(reduce_mid_f32v_kernel, dim3((unsigned)((outer * ns + 3) / 4)), dim3(256), 0, stream(), first,
                         (const float*)src, outer, R, (int)inner, (int)ns);
      TNH_LAUNCH_CHECK();
      if (ns > 1) {   // partials [o][s][i] -> sum over s
        const unsigned bx = (unsigned)((cols + 255) / 256);
        hipLaunchKernelGGL((reduce_mid_kernel<TNH_F32, true, false>), dim3(bx, 1), dim3(256), 0, stream(), dst,
                           (const void*)g_scratch, outer, ns, inner, 1);
        TNH_LAUNCH_CHECK();
      }
      return TNH_OK;
    }
  }
  int nsplit = 1;
  const int64_t want_threads = (int64_t)num_cus() * 1024;
  if (cols < want_threads && R >= 256) {
    int64_t s = want_threads / cols;
    if (s > R / 64) s = R / 64;
    const int64_t max_by_scratch = (int64_t)(kScratchBytes / sizeof(C)) / cols;
    if (s > max_by_scratch) s = max_by_scratch;
    if (s > 1024) s = 1024;
    if (s > 1) nsplit = (int)s;
  }
  const unsigned bx = (unsigned)((cols + 255) / 256);
  if (nsplit == 1) {
    hipLaunchKernelGGL((reduce_mid_kernel<DT, false, false>), dim3(bx, 1), dim3(256), 0, stream(), dst,
                       src, outer, R, inner, 1);
    TNH_LAUNCH_CHECK();
    return TNH_OK;
  }
  int rc = ensure_scratch();
  if (rc) return rc;
  hipLaunchKernelGGL((reduce_mid_kernel<DT, false, true>), dim3(bx, (unsigned)nsplit), dim3(256), 0,
                     stream(), g_scratch, src, outer, R, inner, nsplit);
  TNH_LAUNCH_CHECK();
  hipLaunchKernelGGL((reduce_mid_kernel<DT, true, false>), dim3(bx, 1), dim3(256), 0, stream(), dst,
                     (const void*)g_scratch, (int64_t)1, (int64_t)nsplit, cols, 1);
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

}  // namespace tnh

using namespace tnh;

extern "C" {

int tnh_trace_last2(void* dst, const void* src, int64_t outer, int64_t n, int64_t m, int64_t offset,
                    int dtype) {
  TNH_NEED_INIT();
  TNH_REQUIRE(outer >= 0 && n >= 0 && m >= 0, "negative size");
  if (outer == 0) return TNH_OK;
  TNH_REQUIRE(dst != nullptr, "null pointer");
  // diagonal k = offset: elements (i, i + offset); start and length as numpy.
  int64_t start, len;
  if (offset >= 0) {
    start = offset;
    len = (m - offset < n) ? (m - offset) : n;
  } else {
    start = -offset * m;
    len = (n + offset < m) ? (n + offset) : m;
  }
  if (len < 0) len = 0;
  const int esz = dtype_size(dtype);
  TNH_REQUIRE(esz > 0, "bad dtype %d", dtype);
  if (len == 0) {
    TNH_HIP(hipMemsetAsync(dst, 0, (size_t)outer * esz, stream()));
    return TNH_OK;
  }
  TNH_REQUIRE(src != nullptr, "null pointer");
  const char* base = (const char*)src + start * esz;
  TNH_DISPATCH_NUM(dtype, return (reduce_rows<DT, false, false>(dst, base, outer, len, m + 1, n * m)));
  return TNH_OK;
}

int tnh_sum_mid(void* dst, const void* src, int64_t outer, int64_t reduce, int64_t inner, int dtype) {
  TNH_NEED_INIT();
  TNH_REQUIRE(outer >= 0 && reduce >= 0 && inner >= 0, "negative size");
  if (outer * inner == 0) return TNH_OK;
  TNH_REQUIRE(dst != nullptr, "null pointer");
  const int esz = dtype_size(dtype);
  TNH_REQUIRE(esz > 0, "bad dtype %d", dtype);
  if (reduce == 0) {
    TNH_HIP(hipMemsetAsync(dst, 0, (size_t)(outer * inner) * esz, stream()));
    return TNH_OK;
  }
  TNH_REQUIRE(src != nullptr, "null pointer");
  if (inner == 1) {
    TNH_DISPATCH_NUM(dtype, return (reduce_rows<DT, false, false>(dst, src, outer, reduce, 1, reduce)));
  } else {
    TNH_DISPATCH_NUM(dtype, return (reduce_mid<DT>(dst, src, outer, reduce, inner)));
  }
  return TNH_OK;
}

int tnh_norm(void* dst, const void* src, int64_t n, int dtype) {
  TNH_NEED_INIT();
  TNH_REQUIRE(n >= 0, "negative size");
  TNH_REQUIRE(dst != nullptr, "null pointer");
  const int esz = dtype_size(dtype);
  TNH_REQUIRE(esz > 0, "bad dtype %d", dtype);
  if (n == 0) {
    TNH_HIP(hipMemsetAsync(dst, 0, (size_t)esz, stream()));
    return TNH_OK;
  }
  TNH_REQUIRE(src != nullptr, "null pointer");
  // complex inputs write (norm, 0) in the complex dtype; the host shim takes
  // the real part (numpy returns a real scalar).
  TNH_DISPATCH_FLOAT(dtype, return (reduce_rows<DT, true, true>(dst, src, 1, n, 1, n)));
  return TNH_OK;
}

}  // extern "C"
