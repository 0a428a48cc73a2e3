// Where do the joules of the bf16 GEMM go?  (round 3; built by tools/energy_probe/build.sh, driven by run.py)
//
// Four kernels with the ping-pong GEMM's geometry (512 threads = 8 waves per CU, 32 v_mfma_f32_16x16x32_bf16 per wave per
// "K-tile", all operands random bf16 so that the datapath toggles like the real product):
//   mode 0  MFMA only: operand fragments live in registers (8 A and 8 B fragments, constant: a LOWER bound of the
//           matrix pipe's power on random data -- consecutive MFMAs still see different operands)
//   mode 1  + the LDS fragment reads of the real kernel: 24 ds_read_b128 per 64 MFMA from a random-filled 128 KiB image
//   mode 2  + the global -> LDS traffic of the real kernel, L2 / Infinity-Cache resident (32 MiB source)
//   mode 3  + the same traffic streamed from a 16 GiB source (HBM)
// Board power and clock are sampled by the host while a mode runs; TFLOP/s, W, MHz -> pJ / flop per component.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

__device__ __forceinline__ uint32_t lcg(uint32_t& s) {
  s = s * 1664525u + 1013904223u;
  return s;
}
// two bf16 in [1, 2) with random sign and mantissa
__device__ __forceinline__ uint32_t rnd2(uint32_t& s) {
  const uint32_t r = lcg(s);
  return 0x3F803F80u | (r & 0x807F807Fu);
}

template <int MODE>
__global__ __launch_bounds__(512) void probe_kernel(const uint4* __restrict__ src, int64_t src_vecs, int ktiles,
                                                    float* __restrict__ sink) {
  __shared__ __attribute__((aligned(16))) uint4 lds[8192];      // 128 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  uint32_t seed = (blockIdx.x * 512u + tid) * 2654435761u + 12345u;
  uint4 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = make_uint4(rnd2(seed), rnd2(seed), rnd2(seed), rnd2(seed));
    b[i] = make_uint4(rnd2(seed), rnd2(seed), rnd2(seed), rnd2(seed));
  }
  if (MODE >= 1) {
    for (int i = tid; i < 8192; i += 512) lds[i] = make_uint4(rnd2(seed), rnd2(seed), rnd2(seed), rnd2(seed));
    __syncthreads();
  }
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // global source walk: every workgroup streams its own 64 KiB per K-tile (A and B halves), 8 x 16 B per lane
  int64_t gpos = ((int64_t)blockIdx.x * 4096 + tid) % src_vecs;
  for (int kt = 0; kt < ktiles; ++kt) {
    if (MODE >= 2) {
      uint4 g[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        g[q] = src[gpos];
        gpos += 512;
        if (gpos >= src_vecs) gpos -= src_vecs;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) lds[(kt & 1) * 4096 + q * 512 + tid] = g[q];
      gpos += (int64_t)(gridDim.x - 1) * 4096;
      if (gpos >= src_vecs) gpos %= src_vecs;
      __syncthreads();
    }
    if (MODE >= 1) {
      // 12 fragment reads per 32 MFMA (the real kernel: 24 ds_read_b128 per 64 MFMA)
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        const uint4 v = lds[((kt * 12 + q) * 64 + lane * 3 + q) & 8191];
        if (q < 6) a[q] = v; else b[q - 6] = v;       // static register indices (no scratch)
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[ks * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&a[ks * 4 + i],
                                                                      *(const bf16x8*)&b[ks * 4 + j],
                                                                      acc[ks * 4 + i][j], 0, 0, 0);
    if ((kt & 63) == 63) {       // keep the accumulators finite
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] *= 1e-3f;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 123.456f) sink[0] = s;
}

static uint4* g_src = nullptr;
static int64_t g_src_vecs = 0;
static float* g_sink = nullptr;

__global__ void fill_kernel(uint4* p, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    uint32_t s = (uint32_t)i * 2654435761u + 99u;
    p[i] = make_uint4(rnd2(s), rnd2(s), rnd2(s), rnd2(s));
  }
}

extern "C" int probe_setup(int64_t src_bytes) {
  if (g_src) { (void)hipFree(g_src); g_src = nullptr; }
  g_src_vecs = src_bytes / 16;
  if (hipMalloc(&g_src, (size_t)g_src_vecs * 16) != hipSuccess) return 1;
  if (!g_sink && hipMalloc(&g_sink, 64) != hipSuccess) return 2;
  fill_kernel<<<4096, 256>>>(g_src, g_src_vecs);
  return hipDeviceSynchronize() == hipSuccess ? 0 : 3;
}

// runs `launches` launches of `ktiles` K-tiles each; returns ms per launch; flops per launch = grid * 8 waves * ktiles * 32 * 16384
extern "C" int probe_run(int mode, int grid, int ktiles, int launches, float* ms_out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  for (int l = 0; l < launches; ++l) {
    switch (mode) {
      case 0: probe_kernel<0><<<grid, 512>>>(g_src, g_src_vecs, ktiles, g_sink); break;
      case 1: probe_kernel<1><<<grid, 512>>>(g_src, g_src_vecs, ktiles, g_sink); break;
      case 2: probe_kernel<2><<<grid, 512>>>(g_src, g_src_vecs, ktiles, g_sink); break;
      default: probe_kernel<3><<<grid, 512>>>(g_src, g_src_vecs, ktiles, g_sink); break;
    }
  }
  (void)hipEventRecord(e1, 0);
  if (hipEventSynchronize(e1) != hipSuccess) return 1;
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *ms_out = ms / launches;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return 0;
}
