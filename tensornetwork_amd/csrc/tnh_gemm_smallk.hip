// K2 (smallest-K corner of the bf16 / f16 path): C[M, N] = A[M, K] * B[N, K]^T with K <= 16 -- what contracting ONE small
// bond between two tensors lowers to (D = 12: 1728 x 248 832 x 12, a 860 MB result from 6 MB of operands; the reference
// reaches it through tensordot, numpy_backend.py:35-37).  Such a product is a pure STORE stream (2 M N bytes against
// 2 M N K flop, K <= 16), and the tile kernels spend it badly: a 128 x 128 x 64 tile is one mostly-zero K step between
// a load phase and a 32 KB epilogue, 10 us per tile (1.6 TB/s: profiles/r04_rr64_D12_final_kernel_seq.txt; 484 us for
// the product above, this kernel: 174 us = 4.9 TB/s, profiles/r04_smallk_check.jsonl).  Here a
// thread owns 8 consecutive columns: it keeps its 8 rows of B in registers as f32 (8 K values), walks 64 rows of A that
// the workgroup staged in LDS as f32 (every lane reads the same address: a broadcast), and leaves one 16-byte store per
// row -- 4 KB contiguous per wave and row.  f32 FMAs in k order; results agree with the matrix-core kernels to the
// rounding of the fp32 sums (not bit for bit: the MFMA adds its 32 products in another order).
//
// Roofline: HBM (stores).  Algorithmic bytes 2 * (M * N + M * K + N * K).
#include "tnh_gemm_nt.h"

namespace tnh {

template <bool IS_BF16>
__device__ __forceinline__ float half_to_f32(uint16_t h) {
  if constexpr (IS_BF16) return __uint_as_float((uint32_t)h << 16);
  else return f16_to_f32(h);
}

struct SmallKArgs {
  const uint16_t* A;
  const uint16_t* B;
  uint16_t* C;
  int64_t lda, ldb, ldc;
  int64_t M, N;
};

template <bool IS_BF16, int K4>
__global__ __launch_bounds__(256) void gemm_smallk_kernel(SmallKArgs p) {
  constexpr int K = 4 * K4;
  constexpr int MB = 64;                       // rows of A (and of C) per workgroup
  __shared__ __attribute__((aligned(16))) float sA[MB][K];
  const int tid = threadIdx.x;
  const int64_t n8 = ((int64_t)blockIdx.x * 256 + tid) * 8;
  const int64_t m0 = (int64_t)blockIdx.y * MB;
  for (int idx = tid; idx < MB * K; idx += 256) {
    const int r = idx / K, k = idx - r * K;
    sA[r][k] = (m0 + r < p.M) ? half_to_f32<IS_BF16>(p.A[(m0 + r) * p.lda + k]) : 0.f;
  }
  const bool active = n8 < p.N;                // N % 8 == 0 (host-checked): all eight columns or none
  float b[8][K];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint16_t* row = p.B + (active ? (n8 + j) * p.ldb : 0);
#pragma unroll
    for (int q = 0; q < K4; ++q) {
      const uint2 v = *(const uint2*)(row + 4 * q);      // rows of B are 8-byte aligned (host-checked)
      b[j][4 * q + 0] = half_to_f32<IS_BF16>((uint16_t)(v.x & 0xffffu));
      b[j][4 * q + 1] = half_to_f32<IS_BF16>((uint16_t)(v.x >> 16));
      b[j][4 * q + 2] = half_to_f32<IS_BF16>((uint16_t)(v.y & 0xffffu));
      b[j][4 * q + 3] = half_to_f32<IS_BF16>((uint16_t)(v.y >> 16));
    }
  }
  __syncthreads();
  if (!active) return;
  const int rows = (p.M - m0 < MB) ? (int)(p.M - m0) : MB;
  uint16_t* c = p.C + m0 * p.ldc + n8;
  for (int r = 0; r < rows; ++r) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float a = sA[r][k];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(a, b[j][k], acc[j]);
    }
    uint4 o;
    o.x = pack2<IS_BF16>(acc[0], acc[1]);
    o.y = pack2<IS_BF16>(acc[2], acc[3]);
    o.z = pack2<IS_BF16>(acc[4], acc[5]);
    o.w = pack2<IS_BF16>(acc[6], acc[7]);
    *(uint4*)(c + (int64_t)r * p.ldc) = o;
  }
}

// true when the product is in this kernel's range and large enough to be a store stream
bool gemm_bf16_smallk_wanted(int out_dt, int64_t M, int64_t N, int64_t K, int64_t batch, const void* A, int64_t lda,
                             const void* B, int64_t ldb, const void* C, int64_t ldc) {
  static const bool on = []() { const char* e = getenv("TNH_GEMM_SMALLK"); return !(e && e[0] == '0'); }();   // "0": tile kernels
  if (!on || batch != 1 || out_dt == TNH_F32 || K < 4 || K > 16 || K % 4 != 0) return false;
  if (M < 16 || N < 256 || N % 8 != 0 || M * N < (int64_t(1) << 22)) return false;
  return lda % 4 == 0 && ldb % 4 == 0 && ldc % 8 == 0 && lda >= K && ldb >= K && ldc >= N && ((uintptr_t)A % 8) == 0 &&
         ((uintptr_t)B % 8) == 0 && ((uintptr_t)C % 16) == 0 && (N + 2047) / 2048 < (int64_t(1) << 31) &&
         (M + 63) / 64 < 65536;
}

int gemm_bf16_smallk(int in_dt, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                     void* C, int64_t ldc, const char** name) {
  SmallKArgs p;
  p.A = (const uint16_t*)A;
  p.B = (const uint16_t*)B;
  p.C = (uint16_t*)C;
  p.lda = lda;
  p.ldb = ldb;
  p.ldc = ldc;
  p.M = M;
  p.N = N;
  const dim3 grid((unsigned)((N + 2047) / 2048), (unsigned)((M + 63) / 64));
  const bool is_bf16 = in_dt == TNH_BF16;
  auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, dim3(256), 0, stream(), p); };
  switch ((int)(K / 4)) {
    case 1: is_bf16 ? go(gemm_smallk_kernel<true, 1>) : go(gemm_smallk_kernel<false, 1>); break;
    case 2: is_bf16 ? go(gemm_smallk_kernel<true, 2>) : go(gemm_smallk_kernel<false, 2>); break;
    case 3: is_bf16 ? go(gemm_smallk_kernel<true, 3>) : go(gemm_smallk_kernel<false, 3>); break;
    default: is_bf16 ? go(gemm_smallk_kernel<true, 4>) : go(gemm_smallk_kernel<false, 4>); break;
  }
  TNH_LAUNCH_CHECK();
  *name = "bf16_smallk_64x2048";
  return TNH_OK;
}

}  // namespace tnh
