// HBM bandwidth ceilings of the MI355X for the access shapes the helper kernels use (round 3): what a plain read
// (reduction-like) and a plain copy (permute-like) can reach at 1 GiB, as a function of bytes in flight per lane,
// workgroups per CU and store flavour.  The helper rooflines in DESIGN are quoted against these, next to the 8 TB/s spec.
//   build: tools/bw_probe/build.sh    run: python tools/bw_probe/run.py   (GPU box)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

static void* g_a = nullptr;
static void* g_b = nullptr;
static size_t g_bytes = 0;

__global__ __launch_bounds__(256) void fill_kernel(uint32_t* __restrict__ p, int64_t n, uint32_t seed) {
  const int64_t step = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += step) {
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    p[i] = x;
  }
}

template <int U>
__global__ __launch_bounds__(256) void read_kernel(const uint4* __restrict__ src, int64_t n, uint4* __restrict__ sink) {
  const int64_t step = (int64_t)gridDim.x * 256;
  uint4 acc = make_uint4(0, 0, 0, 0);
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * step < n; i += U * step) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src[i + u * step];
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
  }
  for (; i < n; i += step) { const uint4 v = src[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc;
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  const int64_t step = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * step < n; i += U * step) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src[i + u * step];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(*(v4u*)&v[u], (v4u*)(dst + i + u * step));
      else dst[i + u * step] = v[u];
    }
  }
  for (; i < n; i += step) dst[i] = src[i];
}

extern "C" {
int bw_setup(int64_t bytes) {
  if ((size_t)bytes != g_bytes) {
    if (g_a) { (void)hipFree(g_a); (void)hipFree(g_b); g_a = g_b = nullptr; }
    if (hipMalloc(&g_a, bytes) != hipSuccess || hipMalloc(&g_b, bytes) != hipSuccess) return 1;
    // pseudo-random contents: constant buffers read 10-15 % faster than real data on this part
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t*)g_a, (int64_t)(bytes / 4), 0x9E3779B9u);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t*)g_b, (int64_t)(bytes / 4), 0x85EBCA6Bu);
    (void)hipDeviceSynchronize();
    g_bytes = bytes;
  }
  return 0;
}
// mode 0: read, 1: copy, 2: copy with non-temporal stores, 3: hipMemcpyAsync D2D.  unroll in {1, 2, 4, 8}
int bw_run(int mode, int unroll, int grid, int reps, float* ms_out) {
  const int64_t n = (int64_t)(g_bytes / 16);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto launch = [&]() {
    const uint4* a = (const uint4*)g_a; uint4* b = (uint4*)g_b;
#define RUN(U)                                                                                      \
    if (mode == 0) hipLaunchKernelGGL((read_kernel<U>), dim3(grid), dim3(256), 0, 0, a, n, b);       \
    else if (mode == 1) hipLaunchKernelGGL((copy_kernel<U, false>), dim3(grid), dim3(256), 0, 0, a, b, n); \
    else hipLaunchKernelGGL((copy_kernel<U, true>), dim3(grid), dim3(256), 0, 0, a, b, n);
    if (mode == 3) { (void)hipMemcpyAsync(g_b, g_a, g_bytes, hipMemcpyDeviceToDevice, 0); return; }
    switch (unroll) { case 1: RUN(1) break; case 2: RUN(2) break; case 4: RUN(4) break; default: RUN(8) break; }
#undef RUN
  };
  launch(); launch();
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *ms_out = ms / reps;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
}
