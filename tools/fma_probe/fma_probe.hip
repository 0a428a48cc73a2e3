// Issue rate of v_fma_f64 / v_fma_f32 / v_accvgpr moves on gfx950 (round 6): one wave per SIMD (grid = 1024 waves of 64),
// 8 independent accumulator chains per lane, N iterations; cycles per wave instruction = elapsed / (instructions per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <typename T, int CH>
__global__ __launch_bounds__(64) void fma_kernel(T* out, int iters, T a, T b) {
  T acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = (T)(threadIdx.x + c);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = acc[c] * a + b;
  }
  T s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += acc[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <typename T, int CH>
static void run(const char* name, int waves) {
  T* out;
  hipMalloc(&out, (size_t)waves * 64 * sizeof(T));
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  fma_kernel<T, CH><<<waves, 64>>>(out, 100, (T)1.0000001, (T)1e-9);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  fma_kernel<T, CH><<<waves, 64>>>(out, iters, (T)1.0000001, (T)1e-9);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)iters * 8 * CH;
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("%s waves=%d chains=%d: %.3f ms, %.2f ns per wave instruction = %.2f cycles at %d MHz (max clock)\n", name, waves, CH, ms,
         ms * 1e6 / instr, ms * 1e6 / instr * clk * 1e-6, clk / 1000);
  hipFree(out);
}
int main() {
  run<double, 8>("v_fma_f64", 1024);
  run<double, 8>("v_fma_f64", 2048);
  run<double, 16>("v_fma_f64", 1024);
  run<float, 8>("v_fma_f32", 1024);
  run<float, 8>("v_fma_f32", 2048);
  return 0;
}
