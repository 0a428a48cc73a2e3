// K2 (general part): strided-batched GEMM for every dtype/transpose/ragged
// shape, plus the tnh_gemm dispatcher.
//
//   gemm_mfma_f32_kernel  f32 / bf16 / f16 inputs, f32 MFMA (v_mfma_f32_32x32x2_f32,
//                         exact f32 fma chain), 128x128x16 tile, 4 waves.
//   gemm_mfma_f64_kernel  f64, v_mfma_f64_16x16x4_f64, 64x64x16 tile.
//   gemm_valu_kernel      any dtype incl. complex; 64x64x16 register-tiled VALU
//                         kernel (also the on-device second opinion in tests).
//
// The bf16/f16 speed path (LDS-DMA staged, 16x16x32 MFMA) lives in
// tnh_gemm_bf16.hip and is selected here when its alignment rules hold.
//
// Operands are addressed as A(m,k) = A[m*rsA + k*csA], B(k,n) = B[k*rsB + n*csB],
// so the four transpose combinations are one kernel.  MFMA roofline: 2*M*N*K flop.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "tnh_types.h"
#include "tnh_gemm_nt.h"

namespace tnh {

struct GemmArgs {
  const void* A;
  const void* B;
  void* C;
  int64_t M, N, K;
  int64_t rsA, csA, rsB, csB, ldc;
  int64_t sA, sB, sC;
  int out_dt;
  double alpha, beta;  // C = alpha * A B + beta * C (real scalars; beta == 0 never reads C)
  const int* run_if;   // non-null: the f32 kernels return at once unless *run_if != 0 (guard of the 3 x bf16 split)
};

// provided by tnh_gemm_bf16.hip
extern int g_opt_raster, g_opt_phases, g_opt_tail, g_opt_persist, g_opt_kwalk, g_opt_lean, g_opt_epi, g_opt_nt;
int gemm_bf16_fast(int in_dt, int out_dt, int variant, int transA, int transB, int64_t M, int64_t N,
                   int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
                   int64_t ldc, int64_t batch, int64_t sA, int64_t sB, int64_t sC, const char** name);

int gemm_bf16_view(int in_dt, int out_dt, int64_t M, int64_t N, int64_t K, const void* A, const OpView& va,
                   const void* B, const OpView& vb, void* C, int64_t ldc, const char** name);

// provided by tnh_gemm_gather.hip
int gemm_gather(int dtype, int64_t Ms, int64_t K, int64_t Nl, const void* S, int64_t lds, const void* L,
                int64_t l_elems, const tnh_gather_desc* desc, void* C, int64_t ldc, int small_first,
                const char** name);
int gemm_gather_plan(const tnh_gather_desc* desc, int64_t K, int64_t Nl, int64_t l_elems, int32_t* chunk_off,
                     int32_t* chunk_row, int32_t* chunk_k, int64_t nchunks, int64_t* tile_base, int64_t ntiles);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float load_out_f32(const void* C, int out_dt, int64_t idx) {
  if (out_dt == TNH_F32) return ((const float*)C)[idx];
  if (out_dt == TNH_BF16) return bf16_to_f32(((const uint16_t*)C)[idx]);
  return f16_to_f32(((const uint16_t*)C)[idx]);
}
__device__ __forceinline__ void store_out_f32(void* C, int out_dt, int64_t idx, float v) {
  if (out_dt == TNH_F32) ((float*)C)[idx] = v;
  else if (out_dt == TNH_BF16) ((uint16_t*)C)[idx] = f32_to_bf16(v);
  else ((uint16_t*)C)[idx] = f32_to_f16(v);
}

// ------------------------------------------------------------------ f32 MFMA
template <int DT>
__global__ __launch_bounds__(256) void gemm_mfma_f32_kernel(GemmArgs g) {
  if (g.run_if != nullptr && *g.run_if == 0) return;      // wave-uniform
  constexpr int BM = 128, BN = 128, BK = 16, LD = 129;
  using S = typename Tr<DT>::S;
  __shared__ float As[BK][LD];
  __shared__ float Bs[BK][LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const S* A = (const S*)g.A + (int64_t)blockIdx.z * g.sA;
  const S* B = (const S*)g.B + (int64_t)blockIdx.z * g.sB;
  const int64_t cbase = (int64_t)blockIdx.z * g.sC;

  // thread -> (row-in-tile, k) mapping for the global loads, chosen so that
  // consecutive lanes walk the contiguous direction of each operand.
  const bool a_k_contig = (g.csA == 1);
  const bool b_k_contig = (g.rsB == 1);
  float ra[8], rb[8];
  auto load_tile = [&](int64_t k0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int mm, kk;
      if (a_k_contig) { kk = tid & 15; mm = (tid >> 4) + 16 * j; }
      else { mm = tid & 127; kk = (tid >> 7) + 2 * j; }
      const int64_t m = m0 + mm, k = k0 + kk;
      ra[j] = (m < g.M && k < g.K) ? Tr<DT>::ld(A, m * g.rsA + k * g.csA) : 0.f;
      int nn;
      if (b_k_contig) { kk = tid & 15; nn = (tid >> 4) + 16 * j; }
      else { nn = tid & 127; kk = (tid >> 7) + 2 * j; }
      const int64_t n = n0 + nn;
      const int64_t kb = k0 + kk;
      rb[j] = (n < g.N && kb < g.K) ? Tr<DT>::ld(B, kb * g.rsB + n * g.csB) : 0.f;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int mm, kk;
      if (a_k_contig) { kk = tid & 15; mm = (tid >> 4) + 16 * j; }
      else { mm = tid & 127; kk = (tid >> 7) + 2 * j; }
      As[kk][mm] = ra[j];
      int nn;
      if (b_k_contig) { kk = tid & 15; nn = (tid >> 4) + 16 * j; }
      else { nn = tid & 127; kk = (tid >> 7) + 2 * j; }
      Bs[kk][nn] = rb[j];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int64_t nt = (g.K + BK - 1) / BK;
  load_tile(0);
  for (int64_t t = 0; t < nt; ++t) {
    store_tile();
    __syncthreads();
    if (t + 1 < nt) load_tile((t + 1) * BK);
    const int kl = lane >> 5, il = lane & 31;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a0 = As[kk + kl][wm * 64 + il];
      const float a1 = As[kk + kl][wm * 64 + 32 + il];
      const float b0 = Bs[kk + kl][wn * 64 + il];
      const float b1 = Bs[kk + kl][wn * 64 + 32 + il];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }

  // C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const float alpha = (float)g.alpha, beta = (float)g.beta;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t n = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < g.M && n < g.N) {
          const int64_t idx = cbase + m * g.ldc + n;
          float v = alpha * acc[i][j][r];
          if (beta != 0.f) v += beta * load_out_f32(g.C, g.out_dt, idx);
          store_out_f32(g.C, g.out_dt, idx, v);
        }
      }
    }
}

// f32 fast path (round 2): same 128 x 128 output tile and v_mfma_f32_32x32x2_f32 schedule, but BK = 32, 16-byte
// global loads along each operand's contiguous direction, two LDS stages and ONE barrier per K-tile (the loads of
// tile t+1 are in flight during the 64 MFMAs of tile t and land in the other stage after them).  Needs f32 in /
// out, 16-byte aligned operands and leading dimensions that are multiples of 4; ragged M / N / K are handled by
// clamping rows and zero-filling the K tail.  Everything else stays on gemm_mfma_f32_kernel above.
//   KC_A / KC_B: the operand is K-contiguous (A[m][k] / B[n][k]); otherwise row-contiguous (A[k][m] / B[k][n]).
template <bool KC_A, bool KC_B>
__global__ __launch_bounds__(256) void gemm_mfma_f32_v2_kernel(GemmArgs g) {
  // LDS image per operand and stage: row-contiguous operands as [k][row] (pitch 132: one ds_write_b128 per
  // float4), K-contiguous operands as [row][k] (pitch 33: the four scalar stores of a float4 and the MFMA
  // fragment reads -- 32 rows at one k -- are both conflict-free; the [k][row] image costs 4-way store conflicts:
  // 98.8 vs 114 TF at 4096^3 before this change)
  if (g.run_if != nullptr && *g.run_if == 0) return;      // wave-uniform
  constexpr int BM = 128, BN = 128, BK = 32, LD = 132, LDK = 33;
  constexpr int IMG = (BK * LD > BM * LDK) ? BK * LD : BM * LDK;
  __shared__ __attribute__((aligned(16))) float As[2][IMG];
  __shared__ __attribute__((aligned(16))) float Bs[2][IMG];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const float* A = (const float*)g.A + (int64_t)blockIdx.z * g.sA;
  const float* B = (const float*)g.B + (int64_t)blockIdx.z * g.sB;
  const int64_t cbase = (int64_t)blockIdx.z * g.sC;
  const int64_t lda = KC_A ? g.rsA : g.csA, ldb = KC_B ? g.csB : g.rsB;   // stride of the non-contiguous index

  float4 ra[4], rb[4];
  // one operand tile = 128 rows x 32 k = 1024 float4, four per thread
  auto load_op = [&](auto kc, float4 (&r)[4], const float* P, int64_t ld, int64_t row0, int64_t rows, int64_t k0) {
    constexpr bool KC = decltype(kc)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (KC) {
        int64_t row = row0 + (tid >> 3) + 32 * j;
        if (row >= rows) row = rows - 1;                  // clamped rows are computed and never stored
        const int64_t k = k0 + 4 * (tid & 7);
        const float* p = P + row * ld + k;
        if (k + 3 < g.K) v = *(const float4*)p;
        else {
          if (k < g.K) v.x = p[0];
          if (k + 1 < g.K) v.y = p[1];
          if (k + 2 < g.K) v.z = p[2];
        }
      } else {
        const int64_t k = k0 + (tid >> 5) + 8 * j;
        const int64_t row = row0 + 4 * (tid & 31);
        if (k < g.K) {
          const float* p = P + k * ld + row;
          if (row + 3 < rows) v = *(const float4*)p;
          else {
            if (row < rows) v.x = p[0];
            if (row + 1 < rows) v.y = p[1];
            if (row + 2 < rows) v.z = p[2];
          }
        }
      }
      r[j] = v;
    }
  };
  auto store_op = [&](auto kc, const float4 (&r)[4], float* S) {
    constexpr bool KC = decltype(kc)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (KC) {
        float* d = S + ((tid >> 3) + 32 * j) * LDK + 4 * (tid & 7);
        d[0] = r[j].x;
        d[1] = r[j].y;
        d[2] = r[j].z;
        d[3] = r[j].w;
      } else {
        *(float4*)(S + ((tid >> 5) + 8 * j) * LD + 4 * (tid & 31)) = r[j];
      }
    }
  };
  // element (row, k) of an operand image
  auto at = [&](auto kc, const float* S, int row, int k) -> float {
    constexpr bool KC = decltype(kc)::value;
    return KC ? S[row * LDK + k] : S[k * LD + row];
  };
  using KA = std::integral_constant<bool, KC_A>;
  using KB = std::integral_constant<bool, KC_B>;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int64_t nt = (g.K + BK - 1) / BK;
  load_op(KA{}, ra, A, lda, m0, g.M, 0);
  load_op(KB{}, rb, B, ldb, n0, g.N, 0);
  store_op(KA{}, ra, As[0]);
  store_op(KB{}, rb, Bs[0]);
  __syncthreads();
  const int kl = lane >> 5, il = lane & 31;
  for (int64_t t = 0; t < nt; ++t) {
    const int cur = (int)(t & 1);
    if (t + 1 < nt) {
      load_op(KA{}, ra, A, lda, m0, g.M, (t + 1) * BK);
      load_op(KB{}, rb, B, ldb, n0, g.N, (t + 1) * BK);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a0 = at(KA{}, As[cur], wm * 64 + il, kk + kl);
      const float a1 = at(KA{}, As[cur], wm * 64 + 32 + il, kk + kl);
      const float b0 = at(KB{}, Bs[cur], wn * 64 + il, kk + kl);
      const float b1 = at(KB{}, Bs[cur], wn * 64 + 32 + il, kk + kl);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (t + 1 < nt) {
      store_op(KA{}, ra, As[cur ^ 1]);
      store_op(KB{}, rb, Bs[cur ^ 1]);
    }
    __syncthreads();
  }

  const float alpha = (float)g.alpha, beta = (float)g.beta;
  float* C = (float*)g.C;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t n = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < g.M && n < g.N) {
          const int64_t idx = cbase + m * g.ldc + n;
          float v = alpha * acc[i][j][r];
          if (beta != 0.f) v += beta * C[idx];
          C[idx] = v;
        }
      }
    }
}

// ------------------------------------------------------------------ f64 MFMA
__global__ __launch_bounds__(256) void gemm_mfma_f64_kernel(GemmArgs g) {
  constexpr int BM = 64, BN = 64, BK = 16, LD = 80;
  __shared__ double As[BK][LD];
  __shared__ double Bs[BK][LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const double* A = (const double*)g.A + (int64_t)blockIdx.z * g.sA;
  const double* B = (const double*)g.B + (int64_t)blockIdx.z * g.sB;
  double* C = (double*)g.C + (int64_t)blockIdx.z * g.sC;
  const bool a_k_contig = (g.csA == 1);
  const bool b_k_contig = (g.rsB == 1);
  double ra[4], rb[4];
  auto load_tile = [&](int64_t k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int mm, kk;
      if (a_k_contig) { kk = tid & 15; mm = (tid >> 4) + 16 * j; }
      else { mm = tid & 63; kk = (tid >> 6) + 4 * j; }
      const int64_t m = m0 + mm, k = k0 + kk;
      ra[j] = (m < g.M && k < g.K) ? A[m * g.rsA + k * g.csA] : 0.0;
      int nn;
      if (b_k_contig) { kk = tid & 15; nn = (tid >> 4) + 16 * j; }
      else { nn = tid & 63; kk = (tid >> 6) + 4 * j; }
      const int64_t n = n0 + nn, kb = k0 + kk;
      rb[j] = (n < g.N && kb < g.K) ? B[kb * g.rsB + n * g.csB] : 0.0;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int mm, kk;
      if (a_k_contig) { kk = tid & 15; mm = (tid >> 4) + 16 * j; }
      else { mm = tid & 63; kk = (tid >> 6) + 4 * j; }
      As[kk][mm] = ra[j];
      int nn;
      if (b_k_contig) { kk = tid & 15; nn = (tid >> 4) + 16 * j; }
      else { nn = tid & 63; kk = (tid >> 6) + 4 * j; }
      Bs[kk][nn] = rb[j];
    }
  };
  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

  const int64_t nt = (g.K + BK - 1) / BK;
  load_tile(0);
  for (int64_t t = 0; t < nt; ++t) {
    store_tile();
    __syncthreads();
    if (t + 1 < nt) load_tile((t + 1) * BK);
    const int kl = lane >> 4, il = lane & 15;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      const double a0 = As[kk + kl][wm * 32 + il];
      const double a1 = As[kk + kl][wm * 32 + 16 + il];
      const double b0 = Bs[kk + kl][wn * 32 + il];
      const double b1 = Bs[kk + kl][wn * 32 + 16 + il];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
  // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * r
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t n = n0 + wn * 32 + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + wm * 32 + i * 16 + (lane >> 4) + 4 * r;
        if (m < g.M && n < g.N) {
          double v = g.alpha * acc[i][j][r];
          if (g.beta != 0.0) v += g.beta * C[m * g.ldc + n];
          C[m * g.ldc + n] = v;
        }
      }
    }
}

// f64, large products: 128x128x32 tile, 4 waves of 64x64 (4x4 v_mfma_f64_16x16x4_f64, 8 LDS reads per 16
// MFMAs instead of 4 per 4; 128 MFMAs per wave between barriers).  LDS image [row][k] with a 34-double
// pitch: the fragment read (16 rows x 2 k per 32-lane half) and both global->LDS write patterns
// (k-contiguous or row-contiguous operand) are bank-conflict free (row pitch 68 dwords = 4 mod 64: 16 rows
// land on 16 distinct 4-bank groups).
__global__ __launch_bounds__(256) void gemm_mfma_f64_128_kernel(GemmArgs g) {
  constexpr int BM = 128, BN = 128, BK = 32, LD = 34;
  __shared__ double As[BM][LD];
  __shared__ double Bs[BN][LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const double* A = (const double*)g.A + (int64_t)blockIdx.z * g.sA;
  const double* B = (const double*)g.B + (int64_t)blockIdx.z * g.sB;
  double* C = (double*)g.C + (int64_t)blockIdx.z * g.sC;
  const bool a_k_contig = (g.csA == 1);
  const bool b_k_contig = (g.rsB == 1);
  double ra[16], rb[16];
  auto load_tile = [&](int64_t k0) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      int mm, kk;
      if (a_k_contig) { kk = tid & 31; mm = (tid >> 5) + 8 * j; }
      else { mm = tid & 127; kk = (tid >> 7) + 2 * j; }
      const int64_t m = m0 + mm, k = k0 + kk;
      ra[j] = (m < g.M && k < g.K) ? A[m * g.rsA + k * g.csA] : 0.0;
      int nn;
      if (b_k_contig) { kk = tid & 31; nn = (tid >> 5) + 8 * j; }
      else { nn = tid & 127; kk = (tid >> 7) + 2 * j; }
      const int64_t n = n0 + nn, kb = k0 + kk;
      rb[j] = (n < g.N && kb < g.K) ? B[kb * g.rsB + n * g.csB] : 0.0;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      int mm, kk;
      if (a_k_contig) { kk = tid & 31; mm = (tid >> 5) + 8 * j; }
      else { mm = tid & 127; kk = (tid >> 7) + 2 * j; }
      As[mm][kk] = ra[j];
      int nn;
      if (b_k_contig) { kk = tid & 31; nn = (tid >> 5) + 8 * j; }
      else { nn = tid & 127; kk = (tid >> 7) + 2 * j; }
      Bs[nn][kk] = rb[j];
    }
  };
  f64x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

  const int64_t nt = (g.K + BK - 1) / BK;
  load_tile(0);
  const int kl = lane >> 4, il = lane & 15;
  for (int64_t t = 0; t < nt; ++t) {
    store_tile();
    __syncthreads();
    if (t + 1 < nt) load_tile((t + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[wm * 64 + i * 16 + il][kk + kl];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[wn * 64 + j * 16 + il][kk + kl];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * r
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wn * 64 + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + wm * 64 + i * 16 + (lane >> 4) + 4 * r;
        if (m < g.M && n < g.N) {
          double v = g.alpha * acc[i][j][r];
          if (g.beta != 0.0) v += g.beta * C[m * g.ldc + n];
          C[m * g.ldc + n] = v;
        }
      }
    }
}

// f64 fast path (round 2).  The 128 x 128 x 32 kernel above compiles to 512 registers (32 separate 64-bit load
// pointers, 512 accumulator <-> VGPR moves, a spill): one wave per SIMD, 42 of 78.6 TFLOP/s with the board at 810 W
// and the full 2.4 GHz -- kernel-bound, not power-bound (profiles/r02_power_f32_f64.jsonl).  Same structure as the
// f32 v2 kernel: BK = 16, 16-byte loads along each operand's contiguous direction from ONE base pointer per
// operand, two LDS stages, one barrier per K-tile; images [k][row] (pitch 144 doubles) for row-contiguous and
// [row][k] (pitch 17) for K-contiguous operands, both conflict-free for the 16-row x 4-k fragment reads.
template <bool KC_A, bool KC_B>
__global__ __launch_bounds__(256, 2) void gemm_mfma_f64_v2_kernel(GemmArgs g) {
  constexpr int BM = 128, BN = 128, BK = 16, LD = 144, LDK = 17;
  constexpr int IMG = (BK * LD > BM * LDK) ? BK * LD : BM * LDK;
  __shared__ __attribute__((aligned(16))) double As[2][IMG];
  __shared__ __attribute__((aligned(16))) double Bs[2][IMG];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const double* A = (const double*)g.A + (int64_t)blockIdx.z * g.sA;
  const double* B = (const double*)g.B + (int64_t)blockIdx.z * g.sB;
  double* C = (double*)g.C + (int64_t)blockIdx.z * g.sC;
  const int64_t lda = KC_A ? g.rsA : g.csA, ldb = KC_B ? g.csB : g.rsB;
  typedef double d2 __attribute__((ext_vector_type(2)));

  d2 ra[4], rb[4];
  // one operand tile = 128 rows x 16 k = 1024 double2, four per thread
  auto load_op = [&](auto kc, d2 (&r)[4], const double* P, int64_t ld, int64_t row0, int64_t rows, int64_t k0) {
    constexpr bool KC = decltype(kc)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      d2 v = {0.0, 0.0};
      if constexpr (KC) {
        int64_t row = row0 + (tid >> 3) + 32 * j;
        if (row >= rows) row = rows - 1;
        const int64_t k = k0 + 2 * (tid & 7);
        const double* p = P + row * ld + k;
        if (k + 1 < g.K) v = *(const d2*)p;
        else if (k < g.K) v[0] = p[0];
      } else {
        const int64_t k = k0 + (tid >> 6) + 4 * j;
        const int64_t row = row0 + 2 * (tid & 63);
        if (k < g.K) {
          const double* p = P + k * ld + row;
          if (row + 1 < rows) v = *(const d2*)p;
          else if (row < rows) v[0] = p[0];
        }
      }
      r[j] = v;
    }
  };
  auto store_op = [&](auto kc, const d2 (&r)[4], double* S) {
    constexpr bool KC = decltype(kc)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (KC) {
        double* d = S + ((tid >> 3) + 32 * j) * LDK + 2 * (tid & 7);
        d[0] = r[j][0];
        d[1] = r[j][1];
      } else {
        *(d2*)(S + ((tid >> 6) + 4 * j) * LD + 2 * (tid & 63)) = r[j];
      }
    }
  };
  auto at = [&](auto kc, const double* S, int row, int k) -> double {
    constexpr bool KC = decltype(kc)::value;
    return KC ? S[row * LDK + k] : S[k * LD + row];
  };
  using KA = std::integral_constant<bool, KC_A>;
  using KB = std::integral_constant<bool, KC_B>;

  f64x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

  const int64_t nt = (g.K + BK - 1) / BK;
  load_op(KA{}, ra, A, lda, m0, g.M, 0);
  load_op(KB{}, rb, B, ldb, n0, g.N, 0);
  store_op(KA{}, ra, As[0]);
  store_op(KB{}, rb, Bs[0]);
  __syncthreads();
  const int kl = lane >> 4, il = lane & 15;
  for (int64_t t = 0; t < nt; ++t) {
    const int cur = (int)(t & 1);
    if (t + 1 < nt) {
      load_op(KA{}, ra, A, lda, m0, g.M, (t + 1) * BK);
      load_op(KB{}, rb, B, ldb, n0, g.N, (t + 1) * BK);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = at(KA{}, As[cur], wm * 64 + i * 16 + il, kk + kl);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = at(KB{}, Bs[cur], wn * 64 + j * 16 + il, kk + kl);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nt) {
      store_op(KA{}, ra, As[cur ^ 1]);
      store_op(KB{}, rb, Bs[cur ^ 1]);
    }
    __syncthreads();
  }
  // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * r
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wn * 64 + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + wm * 64 + i * 16 + (lane >> 4) + 4 * r;
        if (m < g.M && n < g.N) {
          double v = g.alpha * acc[i][j][r];
          if (g.beta != 0.0) v += g.beta * C[m * g.ldc + n];
          C[m * g.ldc + n] = v;
        }
      }
    }
}

// ---------------------------------------------------------------------- VALU
__device__ __forceinline__ float fma_t(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ cf32 fma_t(cf32 a, cf32 b, cf32 c) {
  return {fmaf(a.re, b.re, fmaf(-a.im, b.im, c.re)), fmaf(a.re, b.im, fmaf(a.im, b.re, c.im))};
}
__device__ __forceinline__ cf64 fma_t(cf64 a, cf64 b, cf64 c) {
  return {fma(a.re, b.re, fma(-a.im, b.im, c.re)), fma(a.re, b.im, fma(a.im, b.re, c.im))};
}

__device__ __forceinline__ int32_t fma_t(int32_t a, int32_t b, int32_t c) { return (int32_t)((uint32_t)a * (uint32_t)b + (uint32_t)c); }
__device__ __forceinline__ int64_t fma_t(int64_t a, int64_t b, int64_t c) { return (int64_t)((uint64_t)a * (uint64_t)b + (uint64_t)c); }
__device__ __forceinline__ int32_t scale_t(int32_t a, double s) { return (s == 1.0) ? a : (int32_t)((double)a * s); }
__device__ __forceinline__ int64_t scale_t(int64_t a, double s) { return (s == 1.0) ? a : (int64_t)((double)a * s); }
__device__ __forceinline__ double scale_t(double a, double s) { return a * s; }
__device__ __forceinline__ cf32 scale_t(cf32 a, double s) { return {a.re * (float)s, a.im * (float)s}; }
__device__ __forceinline__ cf64 scale_t(cf64 a, double s) { return {a.re * s, a.im * s}; }

template <int DT>
__global__ __launch_bounds__(256) void gemm_valu_kernel(GemmArgs g) {
  constexpr int BM = 64, BN = 64, BK = 16;
  using S = typename Tr<DT>::S;
  using C = typename Tr<DT>::C;
  __shared__ C As[BK][BM + 1];
  __shared__ C Bs[BK][BN + 1];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const S* A = (const S*)g.A + (int64_t)blockIdx.z * g.sA;
  const S* B = (const S*)g.B + (int64_t)blockIdx.z * g.sB;
  const int64_t cbase = (int64_t)blockIdx.z * g.sC;
  const bool a_k_contig = (g.csA == 1);
  const bool b_k_contig = (g.rsB == 1);
  C acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = zero_of(C{});

  for (int64_t k0 = 0; k0 < g.K; k0 += BK) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int mm, kk;
      if (a_k_contig) { kk = tid & 15; mm = (tid >> 4) + 16 * j; }
      else { mm = tid & 63; kk = (tid >> 6) + 4 * j; }
      const int64_t m = m0 + mm, k = k0 + kk;
      As[kk][mm] = (m < g.M && k < g.K) ? Tr<DT>::ld(A, m * g.rsA + k * g.csA) : zero_of(C{});
      int nn;
      if (b_k_contig) { kk = tid & 15; nn = (tid >> 4) + 16 * j; }
      else { nn = tid & 63; kk = (tid >> 6) + 4 * j; }
      const int64_t n = n0 + nn, kb = k0 + kk;
      Bs[kk][nn] = (n < g.N && kb < g.K) ? Tr<DT>::ld(B, kb * g.rsB + n * g.csB) : zero_of(C{});
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      C a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty + 16 * i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx + 16 * j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fma_t(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t m = m0 + ty + 16 * i, n = n0 + tx + 16 * j;
      if (m < g.M && n < g.N) {
        const int64_t idx = cbase + m * g.ldc + n;
        if constexpr (DT == TNH_BF16 || DT == TNH_F16 || DT == TNH_F32) {
          float v = (float)g.alpha * acc[i][j];
          if (g.beta != 0.0) v += (float)g.beta * load_out_f32(g.C, g.out_dt, idx);
          store_out_f32(g.C, g.out_dt, idx, v);
        } else {
          C v = scale_t(acc[i][j], g.alpha);
          if (g.beta != 0.0) v = v + scale_t(Tr<DT>::ld((const S*)g.C, idx), g.beta);
          Tr<DT>::st((S*)g.C, idx, v);
        }
      }
    }
}

// ------------------------------------------------------- complex via real MFMA
// A complex product C = A B equals the REAL product  C_r = A_r B'  where A_r (M x 2K) and
// C_r (M x 2N) are the interleaved (re, im) images of row-major A and C -- i.e. their own
// memory, zero copy -- and B' (2K x 2N) is the 2x2-block expansion of B:
//     B'[2k][2n] = re,  B'[2k][2n+1] = im,  B'[2k+1][2n] = -im,  B'[2k+1][2n+1] = re.
// One expansion pass over B puts complex64 / complex128 contractions on the f32 / f64
// matrix cores with exactly the 8 M N K real flops a complex GEMM needs.
template <typename R>
__global__ __launch_bounds__(256) void complex_expand_kernel(R* __restrict__ dst, const R* __restrict__ src,
                                                             int64_t K, int64_t N, int64_t rs, int64_t cs,
                                                             int conj) {
  const int64_t total = K * N;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
    const int64_t k = e / N, n = e - k * N;
    const R re = src[2 * (k * rs + n * cs)];
    R im = src[2 * (k * rs + n * cs) + 1];
    if (conj) im = -im;
    R* d0 = dst + (2 * k) * (2 * N) + 2 * n;
    R* d1 = d0 + 2 * N;
    d0[0] = re;  d0[1] = im;
    d1[0] = -im; d1[1] = re;
  }
}


// ----------------------------------------------- f32 GEMM on the bf16 matrix cores
// An f32 value is the exact sum of three bf16 values: hi = bf16(x), mid = bf16(x - hi),
// lo = bf16(x - hi - mid) (8 + 8 + 8 mantissa bits, round-to-nearest-even at every step; the two
// subtractions are exact in f32).  A product a*b is then hi*hi + (hi*mid + mid*hi) + (mid*mid +
// hi*lo + lo*hi) + O(2^-26 |a||b|): six bf16 x bf16 products, each exact in the MFMA's fp32
// accumulator.  Laid out as ONE bf16 "NT" GEMM with a six times longer contraction
//     A' (M x 6 Kp) = [ lo | hi | mid | mid | hi | hi ],   B' (N x 6 Kp) = [ hi | lo | mid | hi | mid | hi ]
// (smallest terms first, so they are summed among themselves before they meet the large partial
// sums), Kp = K rounded up to 64 with zero fill.  Measured error against float64, relative to
// sum |a||b|: 6.3e-8 here vs 8.4e-8 for an f32 GEMM (K = 4096, 6 decades of dynamic range).
// The bf16 kernels run at ~1.4 PFLOP/s, i.e. ~240 TFLOP/s of f32-equivalent work against the
// 157 TFLOP/s peak (95 measured) of v_mfma_f32_32x32x2_f32; the split is one HBM-bound pass
// (4 B in, 12 B out per element).  Inputs the split cannot carry -- inf / nan (inf - inf in the split would turn
// the row into NaN where NumPy gives +-inf), finite values beyond the bf16 range (|x| > 3.38e38), values whose
// low parts would be subnormal (|x| < 2^-100) -- are detected BY the split kernels (a device flag) and the
// product is then recomputed by the f32 MFMA kernel in a predicated launch (round 3: VERDICT r2 weak 1d; it
// returns at once when the flag is clear).  TNH_F32_SPLIT=0 / ":s0" keep every f32 product on the f32 kernel.
// Inputs the split cannot carry: non-finite values (inf - inf in the split poisons the row), finite values above the
// bf16 range, and values so small that their mid / lo parts fall into the subnormal range (below 2^-100 here: the lo
// part sits 2^-16 lower).  The split kernels raise a flag; the caller then recomputes the product with the f32
// MFMA kernel (predicated launch, no host round trip) -- NumPy's answer, not NaN.
__device__ __forceinline__ bool split_unsafe(float x) {
  const float a = fabsf(x);
  return !(a <= 3.38e38f) || (a != 0.f && a < 7.9e-31f);
}
__device__ __forceinline__ void split3(float x, uint16_t& hi, uint16_t& mid, uint16_t& lo) {
  hi = f32_to_bf16(x);
  const float r1 = x - bf16_to_f32(hi);
  mid = f32_to_bf16(r1);
  const float r2 = r1 - bf16_to_f32(mid);
  lo = f32_to_bf16(r2);
}

// segment s of a destination row holds part seg(s) of the source row: 0 = hi, 1 = mid, 2 = lo
template <bool IS_B>
__device__ __forceinline__ constexpr int split_seg(int s) {
  constexpr int a[6] = {2, 0, 1, 1, 0, 0};
  constexpr int b[6] = {0, 2, 1, 0, 1, 0};
  return IS_B ? b[s] : a[s];
}

template <bool IS_B>
__device__ __forceinline__ void store_split_row(uint16_t* drow, int64_t Kp, const uint16_t (&part)[3][8]) {
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const uint16_t* q = part[split_seg<IS_B>(s)];
    uint4 v;
    v.x = q[0] | ((uint32_t)q[1] << 16); v.y = q[2] | ((uint32_t)q[3] << 16);
    v.z = q[4] | ((uint32_t)q[5] << 16); v.w = q[6] | ((uint32_t)q[7] << 16);
    *(uint4*)(drow + s * Kp) = v;
  }
}

// Source rows are K-contiguous (element (r, k) at src[r * rs + k]).  One thread: 8 consecutive k of one row.
template <bool IS_B>
__global__ __launch_bounds__(256) void f32_split3_rows_kernel(uint16_t* __restrict__ dst, const float* __restrict__ src,
                                                              int64_t rows, int64_t K, int64_t Kp, int64_t rs,
                                                              int* __restrict__ unsafe) {
  bool bad = false;
  const int64_t groups = Kp / 8;
  const int64_t total = rows * groups;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
    const int64_t r = e / groups, k0 = (e - r * groups) * 8;
    uint16_t part[3][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x = (k0 + i < K) ? src[r * rs + k0 + i] : 0.f;
      bad |= split_unsafe(x);
      split3(x, part[0][i], part[1][i], part[2][i]);
    }
    store_split_row<IS_B>(dst + r * 6 * Kp + k0, Kp, part);
  }
  if (bad) *unsafe = 1;     // benign race: every writer stores 1
}

// Source is row-index-contiguous (element (r, k) at src[k * cs + r]): 64 x 64 tile through LDS so that both
// the reads (along r) and the writes (along k) are coalesced.
template <bool IS_B>
__global__ __launch_bounds__(256) void f32_split3_cols_kernel(uint16_t* __restrict__ dst, const float* __restrict__ src,
                                                              int64_t rows, int64_t K, int64_t Kp, int64_t cs,
                                                              int* __restrict__ unsafe) {
  __shared__ float tile[64][65];
  bool bad = false;
  const int64_t tiles_r = (rows + 63) / 64, tiles_k = Kp / 64;
  for (int64_t t = blockIdx.x; t < tiles_r * tiles_k; t += gridDim.x) {
    const int64_t r0 = (t % tiles_r) * 64, k0 = (t / tiles_r) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int kk = ty + 4 * j;
      const float x = (r0 + tx < rows && k0 + kk < K) ? src[(k0 + kk) * cs + r0 + tx] : 0.f;
      bad |= split_unsafe(x);
      tile[kk][tx] = x;
    }
    __syncthreads();
    // 64 rows x 8 groups of 8 k = 512 work items, 2 per thread
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const int item = threadIdx.x + 256 * w;
      const int rr = item >> 3, g8 = (item & 7) * 8;
      if (r0 + rr < rows) {
        uint16_t part[3][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) split3(tile[g8 + i][rr], part[0][i], part[1][i], part[2][i]);
        store_split_row<IS_B>(dst + (r0 + rr) * 6 * Kp + k0 + g8, Kp, part);
      }
    }
    __syncthreads();
  }
  if (bad) *unsafe = 1;     // benign race: every writer stores 1
}

static int* g_split_flag = nullptr;   // device word: the split kernels found an operand they cannot carry
static int g_f32_split = -1;   // -1: read TNH_F32_SPLIT (default on); knob ":s0" / ":s1" of tnh_gemm_set_variant

static bool f32_split_enabled() {
  if (g_f32_split < 0) {
    const char* e = getenv("TNH_F32_SPLIT");
    g_f32_split = (e && e[0] == '0') ? 0 : 1;
  }
  return g_f32_split == 1;
}

static int launch_split3(uint16_t* dst, const float* src, int64_t rows, int64_t K, int64_t Kp, bool k_contig,
                         int64_t ld, int is_b, int* unsafe) {
  if (k_contig) {
    int64_t blocks = (rows * (Kp / 8) + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (is_b) hipLaunchKernelGGL(f32_split3_rows_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream(), dst, src, rows, K, Kp, ld, unsafe);
    else hipLaunchKernelGGL(f32_split3_rows_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream(), dst, src, rows, K, Kp, ld, unsafe);
  } else {
    int64_t blocks = ((rows + 63) / 64) * (Kp / 64);
    if (blocks > 65536) blocks = 65536;
    if (is_b) hipLaunchKernelGGL(f32_split3_cols_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream(), dst, src, rows, K, Kp, ld, unsafe);
    else hipLaunchKernelGGL(f32_split3_cols_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream(), dst, src, rows, K, Kp, ld, unsafe);
  }
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

static bool f32_v2_enabled() {
  static const bool on = []() { const char* e = getenv("TNH_F32_V2"); return !(e && e[0] == '0'); }();
  return on;
}

static thread_local const char* g_last_kernel = "none";
// ---- tiny outputs (round 6): M x N <= 16 results over a contraction of up to 2^20 -- the full contraction that ends an
// MPS overlap (<psi|psi> = a 1 x 1 x 65536 "GEMM") ran 75 us through the split-K machinery (64 slices of a 128 x 128
// tile each).  ONE workgroup of 1024 threads: thread t takes k = t, t + 1024, ... for all M x N outputs (coalesced
// along k when the operands are k-contiguous or M = N = 1), fixed-order tree reduction: deterministic.
// (bf16 / f16 too -- the closing 1 x 1 x 1728 product of every slice of the D = 12 network took 49 us through the ragged
//  128 x 128 tile kernel, 8 % of that network's time: operands converted on load, f32 sums, one rounding at the end.)
template <int DT, int ODT>
__global__ __launch_bounds__(1024) void gemm_tiny_kernel(const typename Tr<DT>::S* __restrict__ A, int64_t sam, int64_t sak,
                                                         const typename Tr<DT>::S* __restrict__ B, int64_t sbk, int64_t sbn,
                                                         typename Tr<ODT>::S* __restrict__ C, int64_t ldc, int M, int N, int64_t K) {
  using T = typename Tr<DT>::C;
  __shared__ T red[16][16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  T acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = T(0);
  if (M == 1 && N == 1) {
    // the inner product: eight independent loads per operand in flight (one load per trip made the 64 trips of a
    // 65536-long product a chain of memory latencies: 23 us)
    T part[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) part[u] = T(0);
    for (int64_t k0 = tid; k0 < K; k0 += 8 * 1024) {
      T av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t k = k0 + (int64_t)u * 1024;
        av[u] = k < K ? Tr<DT>::ld(A, k * sak) : T(0);
        bv[u] = k < K ? Tr<DT>::ld(B, k * sbk) : T(0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) part[u] += av[u] * bv[u];
    }
    acc[0] = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
  } else {
    for (int64_t k0 = tid; k0 < K; k0 += 2 * 1024) {
      T av[2][4], bv[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t k = k0 + (int64_t)u * 1024;
#pragma unroll
        for (int m = 0; m < 4; ++m) av[u][m] = (m < M && k < K) ? Tr<DT>::ld(A, m * sam + k * sak) : T(0);
#pragma unroll
        for (int n = 0; n < 4; ++n) bv[u][n] = (n < N && k < K) ? Tr<DT>::ld(B, k * sbk + n * sbn) : T(0);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[4 * m + n] += av[u][m] * bv[u][n];
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const T v = wave_sum(acc[i]);
    if (lane == 0) red[w][i] = v;
  }
  __syncthreads();
  if (tid < 16) {
    T sum = T(0);
#pragma unroll
    for (int ww = 0; ww < 16; ++ww) sum += red[ww][tid];
    const int m = tid >> 2, n = tid & 3;
    if (m < M && n < N) Tr<ODT>::st(C, (int64_t)m * ldc + n, sum);
  }
}

// ---- skinny products (round 6): one side <= 128 against a side <= 2048 over K = 256 ... 4096 in f32 / f64 -- an MPS
// site tensor's physical legs against a bond (M = 2 ... 128, N = 1024, K = 512: 29 of the 31 steps of configs[3]).
// A 128 x 128 tile kernel puts such a product on <= 16 workgroups (35 us), the split-K detour on two launches (13 us).
// Here ONE workgroup owns a 16 x 16 output tile and its NW waves split K; operands go global -> VGPR -> MFMA
// (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64, exact FMAs), no LDS staging: lane (i = lane % 16, q = lane / 16)
// feeds A(m0 + i, k) and B(k, n0 + i) with k = kb + 4 q + e in the e-th MFMA of a 16-deep chunk -- the summation
// order inside a chunk is free, so a k-contiguous operand is ONE 4-element vector load per chunk and a k-strided one
// four 64-byte-coalesced scalar loads.  The waves' partial tiles meet in LDS and are added in wave order:
// deterministic, one launch.
template <typename T>
struct SkinnyTraits;
template <>
struct SkinnyTraits<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int q, int r) { return 4 * q + r; }
};
template <>
struct SkinnyTraits<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int q, int r) { return q + 4 * r; }
};

// four operand values k = kb + 4 q .. + 3 of one row / column `line` (element (line, k) at base[line * ld_line + k * ld_k])
template <typename T, bool KCONTIG>
__device__ __forceinline__ void skinny_load(const T* __restrict__ base, int64_t line, int64_t ld, int64_t k, int64_t kend, T (&v)[4]) {
  if (KCONTIG) {
    const T* p = base + line * ld + k;
    if (k + 3 < kend) {
      typedef T vec_t __attribute__((ext_vector_type(4)));
      const vec_t x = *reinterpret_cast<const vec_t*>(p);
      v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (k + e < kend) ? p[e] : T(0);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (k + e < kend) ? base[(k + e) * ld + line] : T(0);
  }
}

template <typename T, bool TA, bool TB, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
                                                              int64_t ldb, T* __restrict__ C, int64_t ldc, int64_t M,
                                                              int64_t N, int64_t K, int tiles_n) {
  typedef SkinnyTraits<T> Tr;
  typedef typename Tr::acc_t acc_t;
  __shared__ T red[NW][256];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l16 = lane & 15, q = lane >> 4;
  const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * 16, n0 = (int64_t)(blockIdx.x % tiles_n) * 16;
  int64_t mi = m0 + l16, ni = n0 + l16;
  if (mi >= M) mi = M - 1;            // ragged edge: re-read the last row / column, never stored
  if (ni >= N) ni = N - 1;
  const int64_t kc = ((K + NW - 1) / NW + 15) / 16 * 16;
  const int64_t kbeg = (int64_t)w * kc, kend = kbeg + kc < K ? kbeg + kc : K;
  acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll 2
  for (int64_t kb = kbeg; kb < kend; kb += 32) {
    T a0[4], b0[4], a1[4], b1[4];
    // A is [M][K] (TA = 0: k contiguous) or [K][M]; B is [K][N] (TB = 0: k strided) or [N][K]
    skinny_load<T, !TA>(A, mi, lda, kb + 4 * q, kend, a0);
    skinny_load<T, TB>(B, ni, ldb, kb + 4 * q, kend, b0);
    skinny_load<T, !TA>(A, mi, lda, kb + 16 + 4 * q, kend, a1);
    skinny_load<T, TB>(B, ni, ldb, kb + 16 + 4 * q, kend, b1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc0 = Tr::mfma(a0[e], b0[e], acc0);
      acc1 = Tr::mfma(a1[e], b1[e], acc1);
    }
  }
  acc0 += acc1;
#pragma unroll
  for (int r = 0; r < 4; ++r) red[w][Tr::row(q, r) * 16 + l16] = acc0[r];
  __syncthreads();
  if (tid < 256) {
    T sum = red[0][tid];
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) sum += red[ww][tid];
    const int64_t m = m0 + (tid >> 4), n = n0 + (tid & 15);
    if (m < M && n < N) C[m * ldc + n] = sum;
  }
}

template <typename T, int NW>
static void launch_skinny(int ta, int tb, unsigned grid, const T* A, int64_t lda, const T* B, int64_t ldb, T* C, int64_t ldc,
                          int64_t M, int64_t N, int64_t K, int tiles_n) {
  const dim3 g(grid), b(NW * 64);
  if (!ta && !tb) hipLaunchKernelGGL((gemm_skinny_kernel<T, false, false, NW>), g, b, 0, stream(), A, lda, B, ldb, C, ldc, M, N, K, tiles_n);
  else if (!ta && tb) hipLaunchKernelGGL((gemm_skinny_kernel<T, false, true, NW>), g, b, 0, stream(), A, lda, B, ldb, C, ldc, M, N, K, tiles_n);
  else if (ta && !tb) hipLaunchKernelGGL((gemm_skinny_kernel<T, true, false, NW>), g, b, 0, stream(), A, lda, B, ldb, C, ldc, M, N, K, tiles_n);
  else hipLaunchKernelGGL((gemm_skinny_kernel<T, true, true, NW>), g, b, 0, stream(), A, lda, B, ldb, C, ldc, M, N, K, tiles_n);
}

static thread_local bool g_in_splitk = false;   // re-entrancy guard of the split-K path
static int g_variant = 0;  // 0 auto, 1 generic (mfma), 2 valu, 3 bf16_128, 4 bf16_256, 5 bf16_256pp,
                           // 6 bf16_ragged (auto shape), 7 .._128x128, 8 .._64x256, 9 .._256x64

template <typename F>
static int launch_batched(F launch, const GemmArgs& g0, int64_t batch, int BM, int BN) {
  const int64_t gx = (g0.N + BN - 1) / BN, gy = (g0.M + BM - 1) / BM;
  TNH_REQUIRE(gx < (int64_t(1) << 24) && gy < 65536, "GEMM grid too large (%lld x %lld tiles)",
              (long long)gy, (long long)gx);
  for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
    const int64_t nb = (batch - b0 < 65535) ? (batch - b0) : 65535;
    const int rc = launch(g0, b0, dim3((unsigned)gx, (unsigned)gy, (unsigned)nb));
    if (rc) return rc;
    TNH_LAUNCH_CHECK();
  }
  return TNH_OK;
}

}  // namespace tnh

using namespace tnh;

extern "C" {

const char* tnh_gemm_last_kernel(void) { return g_last_kernel; }

int tnh_gemm_set_variant(const char* full) {
  TNH_REQUIRE(full != nullptr, "null name");
  // "<variant>[:r<digit>][:p<digit>]" -- the suffixes set A/B knobs of the bf16 speed path
  char name[64];
  snprintf(name, sizeof(name), "%s", full);
  tnh::g_opt_raster = -1;
  tnh::g_opt_phases = 2;
  tnh::g_opt_tail = 1;
  tnh::g_opt_persist = -1;
  tnh::g_opt_kwalk = -1;
  tnh::g_opt_lean = -1;
  tnh::g_opt_epi = -1;
  tnh::g_opt_nt = -1;
  g_f32_split = -1;   // back to the environment's choice unless ":s<d>" follows
  for (char* c = strchr(name, ':'); c != nullptr;) {
    *c = 0;
    char* next = strchr(c + 1, ':');
    if (c[1] == 'r') tnh::g_opt_raster = atoi(c + 2);
    else if (c[1] == 'p') tnh::g_opt_phases = atoi(c + 2);
    else if (c[1] == 't') tnh::g_opt_tail = atoi(c + 2);
    else if (c[1] == 'g') tnh::g_opt_persist = atoi(c + 2);
    else if (c[1] == 'w') tnh::g_opt_kwalk = atoi(c + 2);
    else if (c[1] == 'l') tnh::g_opt_lean = atoi(c + 2);
    else if (c[1] == 'e') tnh::g_opt_epi = atoi(c + 2);
    else if (c[1] == 'n') tnh::g_opt_nt = atoi(c + 2);
    else if (c[1] == 's') g_f32_split = atoi(c + 2) ? 1 : 0;
    c = next;
  }
  if (!strcmp(name, "auto")) g_variant = 0;
  else if (!strcmp(name, "generic")) g_variant = 1;
  else if (!strcmp(name, "valu")) g_variant = 2;
  else if (!strcmp(name, "bf16_128")) g_variant = 3;
  else if (!strcmp(name, "bf16_256")) g_variant = 4;
  else if (!strcmp(name, "bf16_256pp")) g_variant = 5;
  else if (!strcmp(name, "bf16_ragged")) g_variant = 6;
  else if (!strcmp(name, "bf16_ragged_128x128")) g_variant = 7;
  else if (!strcmp(name, "bf16_ragged_64x256")) g_variant = 8;
  else if (!strcmp(name, "bf16_ragged_256x64")) g_variant = 9;
  else if (!strcmp(name, "bf16_ragged_192x128")) g_variant = 10;
  else if (!strcmp(name, "bf16_ragged_128x192")) g_variant = 11;
  else if (!strcmp(name, "bf16_stream")) g_variant = 12;
  else {
    set_error("unknown gemm variant '%s'", name);
    return TNH_ERR_INVALID;
  }
  return TNH_OK;
}

int tnh_complex_expand(void* dst, const void* src, int64_t K, int64_t N, int64_t row_stride,
                       int64_t col_stride, int conj, int dtype) {
  TNH_NEED_INIT();
  TNH_REQUIRE(dtype == TNH_C64 || dtype == TNH_C128, "tnh_complex_expand needs a complex dtype (got %d)", dtype);
  TNH_REQUIRE(K >= 0 && N >= 0, "negative extent");
  if (K == 0 || N == 0) return TNH_OK;
  TNH_REQUIRE(dst && src, "null pointer");
  int64_t blocks = (K * N + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (dtype == TNH_C64)
    hipLaunchKernelGGL((complex_expand_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, stream(), (float*)dst,
                       (const float*)src, K, N, row_stride, col_stride, conj);
  else
    hipLaunchKernelGGL((complex_expand_kernel<double>), dim3((unsigned)blocks), dim3(256), 0, stream(), (double*)dst,
                       (const double*)src, K, N, row_stride, col_stride, conj);
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

int tnh_gemm_view(int in_dtype, int out_dtype, int64_t M, int64_t N, int64_t K, const void* A,
                  const tnh_operand_view* va, const void* B, const tnh_operand_view* vb, void* C, int64_t ldc) {
  TNH_NEED_INIT();
  TNH_REQUIRE(in_dtype == TNH_BF16 || in_dtype == TNH_F16, "tnh_gemm_view: bf16 / f16 operands only (got %d)", in_dtype);
  TNH_REQUIRE(out_dtype == in_dtype || out_dtype == TNH_F32, "tnh_gemm_view: bad output dtype %d", out_dtype);
  TNH_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C && va && vb, "tnh_gemm_view: bad arguments");
  TNH_REQUIRE(ldc >= N, "ldc (%lld) < N (%lld)", (long long)ldc, (long long)N);
  auto conv = [](const tnh_operand_view* v, int64_t K, OpView* o) -> bool {
    if (v->k0 <= 0 || v->r0 <= 0 || v->k0 % 32 != 0 || K % v->k0 != 0) return false;
    o->r0 = v->r0; o->sr0 = v->sr0; o->sr1 = v->sr1;
    o->sk0 = v->sk0; o->sk1 = v->sk1;
    o->tpi = (int)(v->k0 / 32);
    return v->k0 / 32 < (int64_t(1) << 30);
  };
  if (g_variant != 0) {   // a kernel forced through tnh_gemm_set_variant (tests, A/B): the caller's fallback path runs it
    set_error("tnh_gemm_view: a GEMM variant is forced");
    return TNH_ERR_UNSUPPORTED;
  }
  OpView a, b;
  if (!conv(va, K, &a) || !conv(vb, K, &b)) {
    set_error("tnh_gemm_view: the inner contraction run of an operand is not a multiple of 32 that divides K");
    return TNH_ERR_UNSUPPORTED;
  }
  const char* name = nullptr;
  int rc = gemm_bf16_view(in_dtype, out_dtype, M, N, K, A, a, B, b, C, ldc, &name);
  if (rc == TNH_OK) g_last_kernel = name;
  return rc;
}

int tnh_gemm_gather(int dtype, int64_t Ms, int64_t K, int64_t Nl, const void* S, int64_t lds, const void* L,
                    int64_t l_elems, const tnh_gather_desc* desc, void* C, int64_t ldc, int small_first) {
  TNH_NEED_INIT();
  TNH_REQUIRE(desc != nullptr && S != nullptr && L != nullptr && C != nullptr, "tnh_gemm_gather: null argument");
  TNH_REQUIRE(dtype == TNH_BF16 || dtype == TNH_F16, "tnh_gemm_gather: bf16 / f16 only");
  if (g_variant != 0) {   // a kernel forced through tnh_gemm_set_variant (tests, A/B): the caller's fallback path runs it
    set_error("tnh_gemm_gather: a GEMM variant is forced");
    return TNH_ERR_UNSUPPORTED;
  }
  const char* name = nullptr;
  const int rc = gemm_gather(dtype, Ms, K, Nl, S, lds, L, l_elems, desc, C, ldc, small_first, &name);
  if (rc == TNH_OK) g_last_kernel = name;
  return rc;
}

int tnh_gemm_gather_plan(const tnh_gather_desc* desc, int64_t K, int64_t Nl, int64_t l_elems, int32_t* chunk_off,
                         int32_t* chunk_row, int32_t* chunk_k, int64_t nchunks, int64_t* tile_base, int64_t ntiles) {
  TNH_REQUIRE(desc != nullptr && (nchunks == 0 || (chunk_off && chunk_row && chunk_k)) && (ntiles == 0 || tile_base),
              "tnh_gemm_gather_plan: null argument");
  return gemm_gather_plan(desc, K, Nl, l_elems, chunk_off, chunk_row, chunk_k, nchunks, tile_base, ntiles);
}

int tnh_gemm(int in_dtype, int out_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K,
             const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
             int64_t batch, int64_t strideA, int64_t strideB, int64_t strideC) {
  return tnh_gemm_ex(in_dtype, out_dtype, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, batch, strideA,
                     strideB, strideC, 1.0, 0.0);
}

int tnh_gemm_ex(int in_dtype, int out_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K,
                const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                int64_t batch, int64_t strideA, int64_t strideB, int64_t strideC, double alpha,
                double beta) {
  TNH_NEED_INIT();
  TNH_REQUIRE(M >= 0 && N >= 0 && K >= 0 && batch >= 0, "negative GEMM extent");
  TNH_REQUIRE(in_dtype >= TNH_F32 && in_dtype <= TNH_I64, "bad GEMM dtype %d", in_dtype);
  const bool half_in = (in_dtype == TNH_BF16 || in_dtype == TNH_F16);
  TNH_REQUIRE(out_dtype == in_dtype || (half_in && out_dtype == TNH_F32),
              "unsupported GEMM output dtype %d for input dtype %d", out_dtype, in_dtype);
  TNH_REQUIRE((transA == 0 || transA == 1) && (transB == 0 || transB == 1), "bad transpose flag");
  if (M == 0 || N == 0 || batch == 0) return TNH_OK;
  TNH_REQUIRE(C != nullptr, "null output");
  TNH_REQUIRE(ldc >= N, "ldc (%lld) < N (%lld)", (long long)ldc, (long long)N);
  const int esz_out = dtype_size(out_dtype);
  if (K == 0 && beta != 0.0) {
    set_error("tnh_gemm_ex: K == 0 with beta != 0 is not supported");
    return TNH_ERR_UNSUPPORTED;
  }
  if (K == 0) {
    // empty contraction: C = 0 (numpy.tensordot over a zero-length axis)
    for (int64_t b = 0; b < batch; ++b)
      TNH_HIP(hipMemset2DAsync((char*)C + b * strideC * esz_out, (size_t)ldc * esz_out, 0,
                               (size_t)N * esz_out, (size_t)M, stream()));
    g_last_kernel = "memset";
    return TNH_OK;
  }
  TNH_REQUIRE(A && B, "null operand");
  TNH_REQUIRE(lda >= (transA ? M : K), "lda too small");
  TNH_REQUIRE(ldb >= (transB ? K : N), "ldb too small");

  const bool plain = (alpha == 1.0 && beta == 0.0);

  // ---- tiny outputs: one workgroup (gemm_tiny_kernel)
  if (plain && batch == 1 && M <= 4 && N <= 4 && K >= 1024 && K <= (int64_t(1) << 20) && g_variant == 0 &&
      (in_dtype == TNH_F32 || in_dtype == TNH_F64 || half_in)) {
    const int64_t sam = transA ? 1 : lda, sak = transA ? lda : 1, sbk = transB ? 1 : ldb, sbn = transB ? ldb : 1;
#define TNH_TINY(DT, ODT)                                                                                              \
  hipLaunchKernelGGL((gemm_tiny_kernel<DT, ODT>), dim3(1), dim3(1024), 0, stream(), (const Tr<DT>::S*)A, sam, sak,     \
                     (const Tr<DT>::S*)B, sbk, sbn, (Tr<ODT>::S*)C, ldc, (int)M, (int)N, K)
    if (in_dtype == TNH_F32) TNH_TINY(TNH_F32, TNH_F32);
    else if (in_dtype == TNH_F64) TNH_TINY(TNH_F64, TNH_F64);
    else if (in_dtype == TNH_BF16 && out_dtype == TNH_F32) TNH_TINY(TNH_BF16, TNH_F32);
    else if (in_dtype == TNH_BF16) TNH_TINY(TNH_BF16, TNH_BF16);
    else if (out_dtype == TNH_F32) TNH_TINY(TNH_F16, TNH_F32);
    else TNH_TINY(TNH_F16, TNH_F16);
#undef TNH_TINY
    TNH_LAUNCH_CHECK();
    g_last_kernel = "tiny_1wg";
    return TNH_OK;
  }

  // ---- skinny f32 / f64 products: one workgroup per 16 x 16 output tile, its waves split K (gemm_skinny_kernel)
  if (plain && batch == 1 && g_variant == 0 && out_dtype == in_dtype && (in_dtype == TNH_F32 || in_dtype == TNH_F64) &&
      K >= 256 && K <= 4096 && std::min(M, N) <= 128 && ((M + 127) / 128) * ((N + 127) / 128) <= 16) {
    const int esz = in_dtype == TNH_F32 ? 4 : 8;
    // vector loads of a k-contiguous operand: 16-byte aligned rows
    const bool a_ok = transA || (((uintptr_t)A % 16) == 0 && (lda * esz) % 16 == 0);
    const bool b_ok = !transB || (((uintptr_t)B % 16) == 0 && (ldb * esz) % 16 == 0);
    if (a_ok && b_ok) {
      const int tiles_n = (int)((N + 15) / 16);
      const unsigned grid = (unsigned)(((M + 15) / 16) * tiles_n);
      const bool wide = grid >= 512;       // enough workgroups: four waves each (K / 4 per wave) instead of sixteen
      if (in_dtype == TNH_F32) {
        if (wide) launch_skinny<float, 4>(transA, transB, grid, (const float*)A, lda, (const float*)B, ldb, (float*)C, ldc, M, N, K, tiles_n);
        else launch_skinny<float, 16>(transA, transB, grid, (const float*)A, lda, (const float*)B, ldb, (float*)C, ldc, M, N, K, tiles_n);
      } else {
        if (wide) launch_skinny<double, 4>(transA, transB, grid, (const double*)A, lda, (const double*)B, ldb, (double*)C, ldc, M, N, K, tiles_n);
        else launch_skinny<double, 16>(transA, transB, grid, (const double*)A, lda, (const double*)B, ldb, (double*)C, ldc, M, N, K, tiles_n);
      }
      TNH_LAUNCH_CHECK();
      g_last_kernel = "skinny_16x16";
      return TNH_OK;
    }
  }

  // ---- split-K: few output tiles and a long contraction (inner products <x, y>, environment
  // updates with a small result ...).  One workgroup per tile would walk all of K alone -- a
  // 262144-long dot product took 0.86 ms on ONE CU -- so K is cut into slices computed as one
  // strided-batched GEMM into f32 / f64 partials, summed by the K4 reduction (fixed order).
  const bool cplx_in = (in_dtype == TNH_C64 || in_dtype == TNH_C128);
  const bool int_in = (in_dtype == TNH_I32 || in_dtype == TNH_I64);   // exact integer products: VALU kernel only
  // Round 4, the mid-K regime for f32 / f64: a 128 x 128 tile of an f32 product keeps ONE CU busy for K x 0.054 us
  // (f64: K x 0.11 us) -- the 16-site MPS overlap of configs[3] is 29 products of (512 ... 1024)^3 / 2 that ran 45 us
  // each on 16-32 of the 256 CUs (rocprofv3: 86 % of the chain's 1.5 ms).  From K = 512 on, with at most
  // CUs / 8 tiles of at least 64 x 64 outputs, K is cut into slices of >= 128 so that the chip is filled twice over.
  const bool wide_in = (in_dtype == TNH_F32 || in_dtype == TNH_F64);
  const int64_t kmin = wide_in ? 512 : 4096;
  if (plain && !cplx_in && !int_in && batch == 1 && ldc == N && K >= kmin && g_variant == 0 && !g_in_splitk) {
    const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
    const bool mid = K < 4096;
    // Round 6: one side below 64 (an MPS site tensor's physical leg against a bond: M = 2 ... 32, N = 1024, K = 512 ran
    // 35 us on 8 workgroups -- profiles/r06_mps_chain_shapes.jsonl) is split too, into slices of >= 64.
    const bool skinny = mid && wide_in && std::min(M, N) < 64 && std::max(M, N) >= 256;
    int64_t splits = mid ? std::min<int64_t>({K / (skinny ? 64 : 128), (2 * (int64_t)num_cus()) / tiles, 32})
                         : std::min<int64_t>({K / 1024, (2 * (int64_t)num_cus()) / tiles, 256});
    // (round 6: up to CUs / 2 tiles in the mid-K regime -- 2048 x 512 x 512 of the d = 4 MPS chain ran 40 us on 64 workgroups)
    const bool few = mid ? (tiles * (wide_in ? 2 : 8) <= (int64_t)num_cus() && ((M >= 64 && N >= 64) || skinny))
                         : (tiles * 4 <= (int64_t)num_cus());
    if (few && splits >= 2) {
      int64_t kc = (K + splits - 1) / splits;
      kc = (kc + 63) / 64 * 64;                       // keeps the bf16 kernels' K % 64 rule for full slices
      const int64_t full = K / kc, rem = K - full * kc;
      const int part_dt = half_in ? TNH_F32 : in_dtype;
      const int esz_p = dtype_size(part_dt), esz_in = dtype_size(in_dtype);
      const int64_t parts = full + (rem > 0 ? 1 : 0);
      void* W = nullptr;
      int rc = tnh_malloc(&W, (size_t)parts * M * N * esz_p);
      if (rc) return rc;
      g_in_splitk = true;
      const int64_t sA = transA ? kc * lda : kc, sB = transB ? kc : kc * ldb;
      rc = tnh_gemm_ex(in_dtype, part_dt, transA, transB, M, N, kc, A, lda, B, ldb, W, N, full, sA, sB, M * N, 1.0, 0.0);
      if (!rc && rem > 0)
        rc = tnh_gemm_ex(in_dtype, part_dt, transA, transB, M, N, rem, (const char*)A + (size_t)full * sA * esz_in, lda,
                         (const char*)B + (size_t)full * sB * esz_in, ldb, (char*)W + (size_t)full * M * N * esz_p, N, 1,
                         0, 0, 0, 1.0, 0.0);
      g_in_splitk = false;
      if (!rc) {
        if (part_dt == out_dtype) {
          rc = tnh_sum_mid(C, W, 1, parts, M * N, part_dt);
        } else {   // half output: reduce in f32, then one cast
          void* T32 = nullptr;
          rc = tnh_malloc(&T32, (size_t)M * N * 4);
          if (!rc) rc = tnh_sum_mid(T32, W, 1, parts, M * N, TNH_F32);
          if (!rc) rc = tnh_cast(C, out_dtype, T32, TNH_F32, M * N);
          if (T32) tnh_free(T32);
        }
      }
      tnh_free(W);
      if (!rc) g_last_kernel = "splitk";
      return rc;
    }
  }

  bool split_done = false;
  // ---- f32 on the bf16 matrix cores (see f32_split3_*_kernel): large products only -- below ~192 tiles
  // of 256 x 256 the bf16 kernels are not fast enough to pay for six passes plus the split.
  if (in_dtype == TNH_F32 && plain && batch == 1 && g_variant == 0 && f32_split_enabled() && K >= 1024 &&
      M >= 256 && N >= 256 && ((M + 255) / 256) * ((N + 255) / 256) >= 192 && ldc % 4 == 0 &&
      ((uintptr_t)C % 16) == 0) {
    const int64_t Kp = (K + 63) / 64 * 64;
    void *A3 = nullptr, *B3 = nullptr;
    if (!g_split_flag) {
      void* f = nullptr;
      if (tnh_malloc(&f, 256) == TNH_OK) g_split_flag = (int*)f;     // lives as long as the library
    }
    int rc = g_split_flag ? tnh_malloc(&A3, (size_t)M * 6 * Kp * 2) : TNH_ERR_NOMEM;
    if (!rc) rc = tnh_malloc(&B3, (size_t)N * 6 * Kp * 2);
    if (!rc && hipMemsetAsync(g_split_flag, 0, sizeof(int), stream()) != hipSuccess) {
      set_error("hipMemsetAsync(split flag) failed");
      rc = TNH_ERR_HIP;          // falls through to the frees below (an early return leaked A3 / B3)
    }
    if (!rc) {
      // A element (m, k): transA ? A[k * lda + m] : A[m * lda + k];  B element (n, k): transB ? B[n * ldb + k] : B[k * ldb + n]
      rc = launch_split3((uint16_t*)A3, (const float*)A, M, K, Kp, !transA, lda, 0, g_split_flag);
      if (!rc) rc = launch_split3((uint16_t*)B3, (const float*)B, N, K, Kp, transB != 0, ldb, 1, g_split_flag);
      const char* name = nullptr;
      if (!rc)
        rc = gemm_bf16_fast(TNH_BF16, TNH_F32, 0, 0, 1, M, N, 6 * Kp, A3, 6 * Kp, B3, 6 * Kp, C, ldc, 1, 0, 0, 0, &name);
      if (!rc) split_done = true;
    } else {
      rc = TNH_ERR_UNSUPPORTED;   // no room for the split operands: take the native f32 kernel below
    }
    if (A3) tnh_free(A3);
    if (B3) tnh_free(B3);
    if (rc != TNH_ERR_UNSUPPORTED && !split_done) return rc;
    // split_done: fall through to the f32 MFMA kernel, launched with run_if = the flag the split kernels raise for
    // operands the split cannot carry (inf / nan / beyond the bf16 range / near-subnormal): it returns at once
    // otherwise (one near-empty launch), and rewrites C with NumPy's answer if it has to
  }

  if (half_in && plain && (g_variant == 0 || g_variant >= 3)) {
    const char* name = nullptr;
    int rc = gemm_bf16_fast(in_dtype, out_dtype, g_variant, transA, transB, M, N, K, A, lda, B, ldb, C,
                            ldc, batch, strideA, strideB, strideC, &name);
    if (rc == TNH_OK) {
      g_last_kernel = name;
      return TNH_OK;
    }
    if (rc != TNH_ERR_UNSUPPORTED) return rc;
    if (g_variant >= 3) return rc;  // forced variant cannot run this shape
  }

  GemmArgs g;
  g.A = A; g.B = B; g.C = C;
  g.M = M; g.N = N; g.K = K;
  g.rsA = transA ? 1 : lda;  g.csA = transA ? lda : 1;
  g.rsB = transB ? 1 : ldb;  g.csB = transB ? ldb : 1;
  g.ldc = ldc;
  g.sA = strideA; g.sB = strideB; g.sC = strideC;
  g.out_dt = out_dtype;
  g.alpha = alpha;
  g.beta = beta;
  g.run_if = split_done ? g_split_flag : nullptr;

  const bool use_valu = (g_variant == 2) || in_dtype == TNH_C64 || in_dtype == TNH_C128 || int_in;
  const int esz_in = dtype_size(in_dtype);
  auto shifted = [&](int64_t b0) {
    GemmArgs h = g;
    h.A = (const char*)g.A + b0 * strideA * esz_in;
    h.B = (const char*)g.B + b0 * strideB * esz_in;
    h.C = (char*)g.C + b0 * strideC * esz_out;
    return h;
  };
  if (use_valu) {
    g_last_kernel = "valu_64x64x16";
    return launch_batched(
        [&](const GemmArgs&, int64_t b0, dim3 grid) -> int {
          GemmArgs h = shifted(b0);
          TNH_DISPATCH_NUM(in_dtype, hipLaunchKernelGGL((gemm_valu_kernel<DT>), grid, dim3(256), 0,
                                                        stream(), h));
          return 0;
        },
        g, batch, 64, 64);
  }
  if (in_dtype == TNH_F64 && M >= 128 && N >= 128 && ((M + 127) / 128) * ((N + 127) / 128) * batch >= 128 &&
      g_variant != 1) {
    // enough 128x128 tiles to occupy the chip: the wider tile halves the LDS reads per MFMA
    {
      const bool kc_a = (g.csA == 1), kc_b = (g.rsB == 1);
      const int64_t lda_ = kc_a ? g.rsA : g.csA, ldb_ = kc_b ? g.csB : g.rsB;
      const bool ok = f32_v2_enabled() && (kc_a || g.rsA == 1) && (kc_b || g.csB == 1) && lda_ % 2 == 0 && ldb_ % 2 == 0 &&
                      ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && strideA % 2 == 0 && strideB % 2 == 0;
      if (ok) {
        g_last_kernel = "mfma_f64_128x128x16_v2";
        return launch_batched(
            [&](const GemmArgs&, int64_t b0, dim3 grid) -> int {
              GemmArgs h = shifted(b0);
              if (kc_a && kc_b) hipLaunchKernelGGL((gemm_mfma_f64_v2_kernel<true, true>), grid, dim3(256), 0, stream(), h);
              else if (kc_a) hipLaunchKernelGGL((gemm_mfma_f64_v2_kernel<true, false>), grid, dim3(256), 0, stream(), h);
              else if (kc_b) hipLaunchKernelGGL((gemm_mfma_f64_v2_kernel<false, true>), grid, dim3(256), 0, stream(), h);
              else hipLaunchKernelGGL((gemm_mfma_f64_v2_kernel<false, false>), grid, dim3(256), 0, stream(), h);
              return 0;
            },
            g, batch, 128, 128);
      }
    }
    g_last_kernel = "mfma_f64_128x128x32";
    return launch_batched(
        [&](const GemmArgs&, int64_t b0, dim3 grid) -> int {
          GemmArgs h = shifted(b0);
          hipLaunchKernelGGL(gemm_mfma_f64_128_kernel, grid, dim3(256), 0, stream(), h);
          return 0;
        },
        g, batch, 128, 128);
  }
  if (in_dtype == TNH_F64) {
    g_last_kernel = "mfma_f64_64x64x16";
    return launch_batched(
        [&](const GemmArgs&, int64_t b0, dim3 grid) -> int {
          GemmArgs h = shifted(b0);
          hipLaunchKernelGGL(gemm_mfma_f64_kernel, grid, dim3(256), 0, stream(), h);
          return 0;
        },
        g, batch, 64, 64);
  }
  if (in_dtype == TNH_F32 && out_dtype == TNH_F32 && g_variant != 1 && f32_v2_enabled()) {
    // 16-byte loads need: a contiguous direction on each operand (always true for the two storage forms the
    // dispatcher produces), the other stride a multiple of 4 elements, 16-byte aligned bases and batch strides
    const bool kc_a = (g.csA == 1), kc_b = (g.rsB == 1);
    const int64_t lda_ = kc_a ? g.rsA : g.csA, ldb_ = kc_b ? g.csB : g.rsB;
    const bool ok = (kc_a || g.rsA == 1) && (kc_b || g.csB == 1) && lda_ % 4 == 0 && ldb_ % 4 == 0 &&
                    ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && strideA % 4 == 0 && strideB % 4 == 0;
    if (ok) {
      g_last_kernel = split_done ? "f32_as_3xbf16_nt_256x256x64_pp" : "mfma_f32_128x128x32_v2";
      return launch_batched(
          [&](const GemmArgs&, int64_t b0, dim3 grid) -> int {
            GemmArgs h = shifted(b0);
            if (kc_a && kc_b) hipLaunchKernelGGL((gemm_mfma_f32_v2_kernel<true, true>), grid, dim3(256), 0, stream(), h);
            else if (kc_a) hipLaunchKernelGGL((gemm_mfma_f32_v2_kernel<true, false>), grid, dim3(256), 0, stream(), h);
            else if (kc_b) hipLaunchKernelGGL((gemm_mfma_f32_v2_kernel<false, true>), grid, dim3(256), 0, stream(), h);
            else hipLaunchKernelGGL((gemm_mfma_f32_v2_kernel<false, false>), grid, dim3(256), 0, stream(), h);
            return 0;
          },
          g, batch, 128, 128);
    }
  }
  g_last_kernel = split_done ? "f32_as_3xbf16_nt_256x256x64_pp" : "mfma_f32_128x128x16";
  return launch_batched(
      [&](const GemmArgs&, int64_t b0, dim3 grid) -> int {
        GemmArgs h = shifted(b0);
        if (in_dtype == TNH_F32)
          hipLaunchKernelGGL((gemm_mfma_f32_kernel<TNH_F32>), grid, dim3(256), 0, stream(), h);
        else if (in_dtype == TNH_BF16)
          hipLaunchKernelGGL((gemm_mfma_f32_kernel<TNH_BF16>), grid, dim3(256), 0, stream(), h);
        else
          hipLaunchKernelGGL((gemm_mfma_f32_kernel<TNH_F16>), grid, dim3(256), 0, stream(), h);
        return 0;
      },
      g, batch, 128, 128);
}

}  // extern "C"
