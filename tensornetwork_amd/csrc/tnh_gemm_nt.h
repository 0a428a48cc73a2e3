// Pieces shared by the bf16 / f16 "NT" GEMM kernels (C[M,N] = A[M,K] * B[N,K]^T, both
// operands K-contiguous): argument block, MFMA wrapper, XCD-aware tile order, epilogue.
#pragma once
#include "tnh_internal.h"

namespace tnh {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// One operand of a "view" GEMM (tnh_gemm_view): a rank-<=4 tensor read in place as a matrix whose
// row index and contraction index are each split over up to two memory levels (elements):
//   row r = r1 * r0 + r0'  ->  r1 * sr1 + r0' * sr0        k = k1 * (32 * tpi) + k0'  ->  k1 * sk1 + k0' * sk0
// with exactly one of sk0 / sr0 equal to 1: "K-contiguous" (the NT form generalised) or "k-major"
// (rows contiguous: what a [K][N] operand is).  A half K-tile (32 indices) never straddles a k1 boundary.
struct OpView {
  int64_t r0, sr0, sr1;
  int64_t sk0, sk1;
  int tpi;   // half K-tiles (32 contraction indices) per inner contraction run (k0 / 32)
};

struct NtArgs {
  OpView va, vb;   // view kernels only
  int64_t m_off;     // ping-pong kernels: row index of A's first row (tail launches of a two-level-row view), else 0
  int kslice_tiles;  // view kernels, split-K launches: blockIdx.y owns K-tiles [y * kslice_tiles, ...) and writes
                     // its partial product to C + y * sC (0: one slice = all of K)
  const uint16_t* A;
  const uint16_t* B;
  void* C;
  int64_t M, N, K;
  int64_t lda, ldb, ldc;
  int64_t sA, sB, sC;
  int tiles_m, tiles_n;
  int c_vec;   // 1: C rows are 8-B (half out) / 16-B (f32 out) aligned at every n % 4 == 0 -> vector stores;
               // 3 (ping-pong kernels, half out): the same with non-temporal 16-B stores (C of 64 MiB and more: it is
               //   not read again before it has left the L2; +1 % at 32768^2 x 1024..4096);
               // 2 (ragged kernel, half out): also 16-B aligned at every n % 8 == 0 -> LDS-staged row stores
  int a_vw, b_vw;  // ragged kernel only: widest aligned load (elements: 8, 4, 2, 1) on rows of A / B
  int lean_rel_a, lean_rel_b;   // lean loop: 1 = a lane's row offsets are taken from the tile's first row (rows ascend in
                                // memory, a tile spans < 4 GiB), 0 = from the operand's base (the whole operand spans < 4 GiB)
  int epi_early;   // ping-pong kernels: 1 = a persistent workgroup starts its next tile's MFMAs while the stores of the tile it
                   // just finished are still draining (the wait at the top of a tile covers the prologue's loads only)
  int raster;  // 1 (default): 16x16 super-tiles shared by the 8 XCDs (3x less HBM traffic, +2%); 0: per-XCD ranges, M-grouped
};

template <bool IS_BF16>
__device__ __forceinline__ f32x4 mma16(const uint4& a, const uint4& b, f32x4 c) {
  if constexpr (IS_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&a, *(const bf16x8*)&b, c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const f16x8*)&a, *(const f16x8*)&b, c, 0, 0, 0);
}

template <bool IS_BF16>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  if constexpr (IS_BF16) {
    // v_cvt_pk_bf16_f32: round-to-nearest-even in one VALU op (the software rule costs ~6 per value)
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2_t v = {lo, hi};
    const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    return *(const uint32_t*)&b;
  } else {
    return (uint32_t)f32_to_f16(lo) | ((uint32_t)f32_to_f16(hi) << 16);
  }
}

// XCD-aware, M-grouped tile order.  `bid` -> (tile_m, tile_n), bijective for
// any grid size.
__device__ __forceinline__ void tile_of_block(int bid, int tiles_m, int tiles_n, int raster, int& tm,
                                              int& tn) {
  const int nwg = tiles_m * tiles_n;
  if (raster == 1 && (tiles_m & 15) == 0 && (tiles_n & 15) == 0) {
    // The 256 workgroups resident at one time (32 per XCD) cover one 16x16
    // block of tiles; XCD x owns the 4x8 sub-block (x >> 1, x & 1) so its private
    // L2 sees 4 A-panels x 8 B-panels, while the other XCDs' fetches of the same
    // panels hit in the memory-side Infinity Cache.
    const int xcd = bid & 7, j = bid >> 3;
    const int sb = j >> 5, w = j & 31;
    const int sbm = tiles_m >> 4;
    const int sm = sb % sbm, sn = sb / sbm;
    tm = sm * 16 + (xcd >> 1) * 4 + (w & 3);
    tn = sn * 16 + (xcd & 1) * 8 + (w >> 2);
    return;
  }
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, local = bid >> 3;
  const int pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  const int GROUP_M = raster == 2 ? 4 : (raster == 3 ? 16 : (raster == 4 ? 32 : 8));   // A/B knobs :r2 / :r3 / :r4
  const int per_group = GROUP_M * tiles_n;
  const int group = pid / per_group;
  const int first_m = group * GROUP_M;
  const int gsize = (tiles_m - first_m < GROUP_M) ? (tiles_m - first_m) : GROUP_M;
  const int in_group = pid - group * per_group;
  tm = first_m + in_group % gsize;
  tn = in_group / gsize;
}

// Epilogue shared by the speed-path kernels.  With the swapped-operand MFMA a
// lane holds C[m = l & 15][n = 4*(l >> 4) .. +3] of each 16x16 tile: 8-B (bf16 /
// f16) or 16-B (f32) stores, 64 contiguous bytes per row per tile.
template <bool IS_BF16, bool OUT_F32, int FM, int FN>
__device__ __forceinline__ void store_wave_tile(const f32x4 (&acc)[FM][FN], const NtArgs& p, char* Cb,
                                                int64_t m0, int64_t n0, int BM, int BN, int wave_m,
                                                int wave_n, int lane) {
  const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.N);
  if constexpr (!OUT_F32 && FN % 4 == 0) {
    // Half output, full tile, 16-B aligned rows: 16-B stores that cover whole 128-B lines.  A lane holds 4 consecutive
    // n (8 B) of one m per fragment and the lane 16 further on the next 4.  Step 1, v_permlane16_swap on the fragment
    // pairs (j, j + 1): lane rows 0 / 2 get columns 0-7 / 8-15 of fragment j, rows 1 / 3 those of fragment j + 1 --
    // 16 B per lane, 64 B per m.  Step 2, a rotation by 8 inside the 16-lane rows between the pairs (j, j + 1) and
    // (j + 2, j + 3): lanes 0-7 keep the first 32 columns, lanes 8-15 the second 32, of m = lane & 7 (first store)
    // and 8 + (lane & 7) (second store) -- 8 rows x 128 B per store instruction.  The 8-byte stores this replaces
    // (16 rows x 32 B each) kept the CU's address path busy for 7.3 us per 256 x 256 tile, the 64-B-per-row form for
    // 3.9 us (profiles/r03_gemm_epilogue.md).
    if (p.c_vec && full && (p.ldc & 7) == 0 && (((uintptr_t)Cb) & 15) == 0) {
      typedef unsigned v4u __attribute__((ext_vector_type(4)));
      const int r = lane >> 4, c = lane & 15;
      const bool lo = c < 8;
      auto ror8 = [](unsigned x) -> unsigned { return (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x128, 0xf, 0xf, true); };
#pragma unroll
      for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int j = 0; j < FN; j += 4) {
          v4u v[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x4 va = acc[i][j + 2 * h], vb = acc[i][j + 2 * h + 1];
            const auto sx = __builtin_amdgcn_permlane16_swap(pack2<IS_BF16>(va[0], va[1]), pack2<IS_BF16>(vb[0], vb[1]), false, false);
            const auto sy = __builtin_amdgcn_permlane16_swap(pack2<IS_BF16>(va[2], va[3]), pack2<IS_BF16>(vb[2], vb[3]), false, false);
            v[h] = v4u{sx[0], sy[0], sx[1], sy[1]};
          }
          v4u w0, w1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned a8 = ror8(v[0][e]), b8 = ror8(v[1][e]);
            w0[e] = lo ? v[0][e] : b8;
            w1[e] = lo ? a8 : v[1][e];
          }
          const int64_t n = n0 + wave_n + j * 16 + 32 * (c >> 3) + 16 * (r & 1) + 8 * (r >> 1);
          const int64_t m = m0 + wave_m + i * 16 + (c & 7);
          char* q0 = Cb + (m * p.ldc + n) * 2;
          char* q1 = q0 + 8 * p.ldc * 2;
          if (p.c_vec == 3) {      // large C: streaming stores.  (__builtin_nontemporal_store on these vector types compiles
                                   // to a plain global_store_dwordx4 with this toolchain -- no `nt` in the ISA -- hence asm; the
                                   // s_nop is the wait state a > 8-byte store needs before its data registers are rewritten)
            asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(q0), "v"(w0) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(q1), "v"(w1) : "memory");
          } else {
            *(v4u*)q0 = w0;
            *(v4u*)q1 = w1;
          }
        }
      }
      return;
    }
  } else if constexpr (!OUT_F32 && FN % 2 == 0) {
    // two fragments per wave row only: step 1 alone (16 rows x 64 B per store)
    if (p.c_vec && full && (p.ldc & 7) == 0 && (((uintptr_t)Cb) & 15) == 0) {
      const int r = lane >> 4;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int64_t m = m0 + wave_m + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < FN; j += 2) {
          const f32x4 va = acc[i][j], vb = acc[i][j + 1];
          const auto sx = __builtin_amdgcn_permlane16_swap(pack2<IS_BF16>(va[0], va[1]), pack2<IS_BF16>(vb[0], vb[1]), false, false);
          const auto sy = __builtin_amdgcn_permlane16_swap(pack2<IS_BF16>(va[2], va[3]), pack2<IS_BF16>(vb[2], vb[3]), false, false);
          const int64_t n = n0 + wave_n + (j + (r & 1)) * 16 + (r >> 1) * 8;
          *(uint4*)(Cb + (m * p.ldc + n) * 2) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int64_t m = m0 + wave_m + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int64_t n = n0 + wave_n + j * 16 + (lane >> 4) * 4;
      const f32x4 v = acc[i][j];
      if (p.c_vec && (full || (m < p.M && n + 3 < p.N))) {
        if constexpr (OUT_F32) {
          *(float4*)(Cb + (m * p.ldc + n) * 4) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          uint2 o;
          o.x = pack2<IS_BF16>(v[0], v[1]);
          o.y = pack2<IS_BF16>(v[2], v[3]);
          *(uint2*)(Cb + (m * p.ldc + n) * 2) = o;
        }
      } else if (m < p.M) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r < p.N) {
            if constexpr (OUT_F32) ((float*)Cb)[m * p.ldc + n + r] = v[r];
            else ((uint16_t*)Cb)[m * p.ldc + n + r] = IS_BF16 ? f32_to_bf16(v[r]) : f32_to_f16(v[r]);
          }
        }
      }
    }
  }
}

}  // namespace tnh
