// K2 (ragged part of the bf16 / f16 matrix-core path):
//
//   C[M,N] = A[M,K] * B[N,K]^T  for ANY M, N, K >= 1 and any element alignment.
//
// Tensor-network bond dimensions are rarely multiples of 64 (D = 12 gives K = 12, 144,
// 1728, M = 144 ...), so the LDS-DMA kernels of tnh_gemm_bf16.hip (K % 64 == 0, 16-B
// aligned rows) cover only part of a contraction path.  This kernel keeps the same LDS
// image and the same v_mfma_f32_16x16x32 fragment scheme but stages HBM -> VGPR -> LDS,
// which lets it
//   * zero-fill the K tail and skip 32-wide k-steps that are entirely past K,
//   * clamp ragged M / N rows (computed, never stored),
//   * load with the widest vector the row alignment allows (16 / 8 / 4 / 2 bytes per
//     load, wave-uniform choice) instead of requiring 16-B aligned rows.
// Pipeline: the global loads of K-tile t+1 are issued before the MFMA work on tile t and
// written to the other LDS stage after it; one workgroup barrier per K-tile.
// Three tile shapes (256 threads each) so that skinny products (M = 144 against
// N = 3*10^6) do not pad the short side to 128: 128x128, 64x256, 256x64.
//
// MFMA roofline: 2*M*N*K flop; the skinny shapes are HBM-bound: 2*(M*K + N*K + M*N) bytes.
#include "tnh_gemm_nt.h"

namespace tnh {

extern int g_opt_phases;  // A/B knob (tnh_gemm_set_variant ":p<d>"): p1 disables the small-K variant
#define g_opt_smallk (g_opt_phases != 1)
#define g_opt_persist (g_opt_phases != 5)   // ":p5" = one block per tile (A/B)
extern int g_opt_raster;  // ":r2" disables the LDS-staged epilogue (A/B)
#define g_opt_wide192 (g_opt_raster != 3)   // ":r3" keeps 128 x 128 tiles for short sides of 129 .. 192 (A/B)

// 8 consecutive k of one operand row -> one 16-B register chunk (zero past K).
// `vw` (elements per aligned load) is wave-uniform.
__device__ __forceinline__ uint4 load_chunk8(const uint16_t* p, int64_t krem, int vw) {
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (krem >= 8) {
    if (vw == 8) {
      v = *(const uint4*)p;
    } else if (vw == 4) {
      const uint2 lo = *(const uint2*)p, hi = *(const uint2*)(p + 4);
      v = make_uint4(lo.x, lo.y, hi.x, hi.y);
    } else if (vw == 2) {
      const uint32_t* q = (const uint32_t*)p;
      v = make_uint4(q[0], q[1], q[2], q[3]);
    } else {
      v.x = (uint32_t)p[0] | ((uint32_t)p[1] << 16);
      v.y = (uint32_t)p[2] | ((uint32_t)p[3] << 16);
      v.z = (uint32_t)p[4] | ((uint32_t)p[5] << 16);
      v.w = (uint32_t)p[6] | ((uint32_t)p[7] << 16);
    }
  } else if (krem > 0) {
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < krem) w[j >> 1] |= (uint32_t)p[j] << (16 * (j & 1));
    v = make_uint4(w[0], w[1], w[2], w[3]);
  }
  return v;
}

//
// SMALLK (K <= 192, i.e. at most 3 K-tiles -- one or two small bonds): such products are
// pure streaming (HBM-bound) and a block's life is a few dependent memory latencies, so
// ALL K-tiles are requested up front into registers and passed through ONE LDS stage
// (half the LDS, twice the resident blocks) instead of the load / compute ring.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool IS_BF16, bool OUT_F32, bool SMALLK>
__device__ __forceinline__ void ragged_one_tile(const NtArgs& p, int tile, char* smem) {
  static_assert(WAVES_M * WAVES_N == 4, "256 threads");
  constexpr int BK = 64;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int FM = WTM / 16, FN = WTN / 16;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int A_CH = BM * 8 / 256, B_CH = BN * 8 / 256;  // 16-B chunks per thread per K-tile
  constexpr int KT_MAX = 3;  // SMALLK: K-tiles held in registers
  // epilogue staging image (half-precision output): BM rows of BN elements + 16 B pad
  constexpr int EPI_PITCH = BN * 2 + 16;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;

  int tm, tn;
  tile_of_block(tile, p.tiles_m, p.tiles_n, 0, tm, tn);
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const uint16_t* A = p.A + (int64_t)blockIdx.y * p.sA;
  const uint16_t* B = p.B + (int64_t)blockIdx.y * p.sB;

  // thread -> (row, chunk) of the tile image: 8 lanes cover the 128 B of one row.
  const int crow = tid >> 3, cchunk = tid & 7;
  const uint16_t* ga[A_CH];
  const uint16_t* gb[B_CH];
#pragma unroll
  for (int j = 0; j < A_CH; ++j) {
    int64_t row = m0 + crow + 32 * j;
    if (row >= p.M) row = p.M - 1;  // ragged edge: re-read the last row, never stored
    ga[j] = A + row * p.lda + cchunk * 8;
  }
#pragma unroll
  for (int j = 0; j < B_CH; ++j) {
    int64_t row = n0 + crow + 32 * j;
    if (row >= p.N) row = p.N - 1;
    gb[j] = B + row * p.ldb + cchunk * 8;
  }
  // LDS image: 128-B rows, 16-B chunk c of row r lives in slot c ^ (r & 7)
  // (r & 7 == crow & 7 for every j because rows advance by 32).
  const int st_off = crow * 128 + ((cchunk ^ (crow & 7)) * 16);

  uint4 ra[A_CH], rb[B_CH];
  const int a_vw = p.a_vw, b_vw = p.b_vw;
  auto load_tile = [&](int64_t k0) {
    const int64_t krem = p.K - k0 - cchunk * 8;
#pragma unroll
    for (int j = 0; j < A_CH; ++j) ra[j] = load_chunk8(ga[j] + k0, krem, a_vw);
#pragma unroll
    for (int j = 0; j < B_CH; ++j) rb[j] = load_chunk8(gb[j] + k0, krem, b_vw);
  };
  auto store_tile = [&](int s) {
    char* base = smem + s * STAGE_BYTES + st_off;
#pragma unroll
    for (int j = 0; j < A_CH; ++j) *(uint4*)(base + j * 32 * 128) = ra[j];
#pragma unroll
    for (int j = 0; j < B_CH; ++j) *(uint4*)(base + A_BYTES + j * 32 * 128) = rb[j];
  };

  int frag_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
    frag_off[ks] = (lane & 15) * 128 + (((ks * 4 + (lane >> 4)) ^ (lane & 7)) * 16);

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto mma_tile = [&](int s, int64_t kvalid) {
    const char* sa = smem + s * STAGE_BYTES + (wm * WTM) * 128;
    const char* sb = smem + s * STAGE_BYTES + A_BYTES + (wn * WTN) * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks * 32 < kvalid) {  // wave-uniform: skip a k-step that is all zero padding
        uint4 af[FM], bf[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = *(const uint4*)(sa + i * 2048 + frag_off[ks]);
#pragma unroll
        for (int j = 0; j < FN; ++j) bf[j] = *(const uint4*)(sb + j * 2048 + frag_off[ks]);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = mma16<IS_BF16>(bf[j], af[i], acc[i][j]);
      }
    }
  };

  const int nt = (int)((p.K + BK - 1) / BK);
  if constexpr (SMALLK) {
    uint4 qa[KT_MAX][A_CH], qb[KT_MAX][B_CH];
#pragma unroll
    for (int t = 0; t < KT_MAX; ++t) {
      if (t < nt) {
        const int64_t krem = p.K - (int64_t)t * BK - cchunk * 8;
#pragma unroll
        for (int j = 0; j < A_CH; ++j) qa[t][j] = load_chunk8(ga[j] + t * BK, krem, a_vw);
#pragma unroll
        for (int j = 0; j < B_CH; ++j) qb[t][j] = load_chunk8(gb[j] + t * BK, krem, b_vw);
      }
    }
#pragma unroll
    for (int t = 0; t < KT_MAX; ++t) {
      if (t < nt) {
        if (t > 0) __syncthreads();  // everyone is done reading tile t-1
        char* base = smem + st_off;
#pragma unroll
        for (int j = 0; j < A_CH; ++j) *(uint4*)(base + j * 32 * 128) = qa[t][j];
#pragma unroll
        for (int j = 0; j < B_CH; ++j) *(uint4*)(base + A_BYTES + j * 32 * 128) = qb[t][j];
        __syncthreads();
        mma_tile(0, p.K - (int64_t)t * BK);
      }
    }
  } else {
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
      if (t + 1 < nt) load_tile((int64_t)(t + 1) * BK);
      mma_tile(t & 1, p.K - (int64_t)t * BK);
      if (t + 1 < nt) store_tile((t + 1) & 1);
      __syncthreads();
    }
  }

  char* Cb = (char*)p.C + (int64_t)blockIdx.y * p.sC * (OUT_F32 ? 4 : 2);
  if constexpr (!OUT_F32) {
    if (p.c_vec == 2) {
      // Half-precision output through LDS: the MFMA layout gives a lane 4 consecutive n of
      // one row (32-B pieces, 16 rows per store); streaming products live on their C
      // writes, so rows are re-assembled in LDS and written as 16 B per lane, a whole
      // BN-wide row segment (up to 512 B) per 32 lanes.
      __syncthreads();  // (SMALLK: the K loop does not end on a barrier) all fragment reads done
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int row = wm * WTM + i * 16 + (lane & 15);
          const int col = wn * WTN + j * 16 + (lane >> 4) * 4;
          uint2 o;
          o.x = pack2<IS_BF16>(acc[i][j][0], acc[i][j][1]);
          o.y = pack2<IS_BF16>(acc[i][j][2], acc[i][j][3]);
          *(uint2*)(smem + row * EPI_PITCH + col * 2) = o;
        }
      __syncthreads();
      constexpr int ROW_CH = BN / 8;  // 16-B chunks per row
#pragma unroll
      for (int it = 0; it < BM * ROW_CH / 256; ++it) {
        const int idx = it * 256 + tid;
        const int row = idx / ROW_CH, ch = idx % ROW_CH;
        const int64_t m = m0 + row, n = n0 + ch * 8;
        if (m < p.M && n < p.N) {
          const uint4 v = *(const uint4*)(smem + row * EPI_PITCH + ch * 16);
          uint16_t* dst = (uint16_t*)Cb + m * p.ldc + n;
          if (n + 8 <= p.N) {
            *(uint4*)dst = v;
          } else {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (n + e < p.N) dst[e] = (uint16_t)(w[e >> 1] >> (16 * (e & 1)));
          }
        }
      }
      return;
    }
  }
  store_wave_tile<IS_BF16, OUT_F32, FM, FN>(acc, p, Cb, m0, n0, BM, BN, wm * WTM, wn * WTN, lane);
}


// Persistent launcher: block b takes tiles b, b + gridDim.x, ...  The launch sizes the grid to the
// tile count, or -- SMALLK -- to what is resident at once: blocks that live a few microseconds are
// otherwise bound by the workgroup dispatch rate (measured on 144 x 3e6 x 144: ~3 resident waves
// per CU with one block per tile).
template <int BM, int BN, int WAVES_M, int WAVES_N, bool IS_BF16, bool OUT_F32, bool SMALLK>
__global__ __launch_bounds__(256) void gemm_nt_ragged_kernel(NtArgs p) {
  constexpr int BK = 64;
  constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
  constexpr int EPI_BYTES = OUT_F32 ? 0 : BM * (BN * 2 + 16);
  constexpr int LOOP_BYTES = (SMALLK ? 1 : 2) * STAGE_BYTES;
  __shared__ __attribute__((aligned(1024))) char smem[LOOP_BYTES > EPI_BYTES ? LOOP_BYTES : EPI_BYTES];
  const int ntiles = p.tiles_m * p.tiles_n;
#pragma clang loop unroll(disable)
  for (int tile = blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
    ragged_one_tile<BM, BN, WAVES_M, WAVES_N, IS_BF16, OUT_F32, SMALLK>(p, tile, smem);
    __syncthreads();  // this tile's LDS reads are done before the next tile's stores
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool SMALLK>
static int launch_ragged(bool is_bf16, bool out_f32, NtArgs p, int64_t batch) {
  p.tiles_m = (int)((p.M + BM - 1) / BM);
  p.tiles_n = (int)((p.N + BN - 1) / BN);
  const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  TNH_REQUIRE(nwg < (int64_t(1) << 24), "GEMM grid too large");
  const int esz_out = out_f32 ? 4 : 2;
  for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
    const int64_t nb = (batch - b0 < 65535) ? (batch - b0) : 65535;
    NtArgs q = p;
    q.A = p.A + b0 * p.sA;
    q.B = p.B + b0 * p.sB;
    q.C = (char*)p.C + b0 * p.sC * esz_out;
    // SMALLK blocks live a few microseconds: a grid of one block per tile is bound by the workgroup
    // dispatch rate, so the grid is what the occupancy API says is resident at once (rounded down to
    // a multiple of 8, which keeps tile index and XCD congruent) and blocks loop over tiles.
    auto launch = [&](auto kernel) -> int {
      int64_t gx = nwg;
      if (SMALLK && g_opt_persist) {
        static int per_cu = 0;   // one query per kernel instantiation (generic lambda)
        if (per_cu == 0) TNH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0));
        int64_t resident = (int64_t)num_cus() * (per_cu > 0 ? per_cu : 1);
        resident -= resident % 8;
        if (resident >= 8 && gx > resident) gx = resident;
      }
      hipLaunchKernelGGL(kernel, dim3((unsigned)gx, (unsigned)nb), dim3(256), 0, stream(), q);
      return TNH_OK;
    };
    int rc;
    if (is_bf16) {
      if (out_f32) rc = launch(gemm_nt_ragged_kernel<BM, BN, WAVES_M, WAVES_N, true, true, SMALLK>);
      else rc = launch(gemm_nt_ragged_kernel<BM, BN, WAVES_M, WAVES_N, true, false, SMALLK>);
    } else {
      if (out_f32) rc = launch(gemm_nt_ragged_kernel<BM, BN, WAVES_M, WAVES_N, false, true, SMALLK>);
      else rc = launch(gemm_nt_ragged_kernel<BM, BN, WAVES_M, WAVES_N, false, false, SMALLK>);
    }
    if (rc) return rc;
    TNH_LAUNCH_CHECK();
  }
  return TNH_OK;
}

// widest power-of-two vector (in 2-byte elements, <= 8) that every row start of an
// operand is aligned to.
static int row_vector_width(const void* base, int64_t ld, int64_t batch_stride, int64_t batch) {
  int vw = 8;
  while (vw > 1 && (((uintptr_t)base % (2 * vw)) != 0 || ld % vw != 0 || (batch > 1 && batch_stride % vw != 0)))
    vw >>= 1;
  return vw;
}

// NT product of any shape / alignment on the matrix cores (called by gemm_bf16_fast
// when the LDS-DMA kernels' alignment rules do not hold).  shape: 0 auto, 1 128x128,
// 2 64x256, 3 256x64, 4 192x128, 5 128x192.
int gemm_bf16_ragged(int in_dt, int out_dt, int shape, int64_t M, int64_t N, int64_t K, const void* A,
                     int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t batch,
                     int64_t sA, int64_t sB, int64_t sC, const char** name) {
  NtArgs p;
  p.A = (const uint16_t*)A;
  p.B = (const uint16_t*)B;
  p.C = C;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.sA = sA; p.sB = sB; p.sC = sC;
  p.raster = 0;
  p.a_vw = row_vector_width(A, lda, sA, batch);
  p.b_vw = row_vector_width(B, ldb, sB, batch);
  p.c_vec = (ldc % 4 == 0) && ((uintptr_t)C % 16 == 0) && (batch == 1 || sC % 4 == 0);
  // 2: rows are 16-B aligned at every n % 8 == 0 -> LDS-staged full-row stores (half output)
  if (p.c_vec && ldc % 8 == 0 && (batch == 1 || sC % 8 == 0) && g_opt_raster != 2) p.c_vec = 2;
  const bool is_bf16 = (in_dt == TNH_BF16), out_f32 = (out_dt == TNH_F32);
  const bool smallk = (K <= 192) && g_opt_smallk;
  if (shape == 0) {
    if (smallk) {
      // streaming products: the square tile keeps the most blocks resident (measured:
      // 144 x 3e6 x 144 0.64 ms vs 0.73 ms for 64x256); only a really short side switches
      shape = (M <= 64 && N > 64) ? 2 : ((N <= 64 && M > 64) ? 3 : 1);
      // a short M of 129 .. 192 (D = 12 / 13: D^2 = 144 / 169) fits ONE 192-row tile: with 128 x 128 tiles the
      // long operand is streamed through L2 / LDS twice and the second tile row is mostly padding.  Measured on
      // MI355X (profiles/r02_skinny_ab.txt): 144 x 2985984 x 144 0.758 -> 0.663-0.696 ms; it loses on short grids
      // (144 x 248832 x 144: 0.055 -> 0.061 ms, one block per CU at 280 registers) and in the transposed
      // orientation (128 x 192 tiles: 0.649 -> 0.704 ms), so only the long-N case takes it.
      if (g_opt_wide192 && M > 128 && M <= 192 && N >= (int64_t(1) << 20)) shape = 4;
    } else {
      // least padded work; ties go to the square tile
      auto padded = [&](int64_t bm, int64_t bn) {
        return (double)((M + bm - 1) / bm * bm) * (double)((N + bn - 1) / bn * bn);
      };
      const double sq = padded(128, 128), wide = padded(64, 256), tall = padded(256, 64);
      shape = 1;
      if (wide < sq && wide <= tall) shape = 2;
      else if (tall < sq && tall < wide) shape = 3;
    }
  }
  if (shape == 2) {
    *name = smallk ? "bf16_nt_ragged_64x256x64_smallk" : "bf16_nt_ragged_64x256x64";
    return smallk ? launch_ragged<64, 256, 1, 4, true>(is_bf16, out_f32, p, batch)
                  : launch_ragged<64, 256, 1, 4, false>(is_bf16, out_f32, p, batch);
  }
  if (shape == 3) {
    *name = smallk ? "bf16_nt_ragged_256x64x64_smallk" : "bf16_nt_ragged_256x64x64";
    return smallk ? launch_ragged<256, 64, 4, 1, true>(is_bf16, out_f32, p, batch)
                  : launch_ragged<256, 64, 4, 1, false>(is_bf16, out_f32, p, batch);
  }
  if (shape == 4) {
    *name = smallk ? "bf16_nt_ragged_192x128x64_smallk" : "bf16_nt_ragged_192x128x64";
    return smallk ? launch_ragged<192, 128, 2, 2, true>(is_bf16, out_f32, p, batch)
                  : launch_ragged<192, 128, 2, 2, false>(is_bf16, out_f32, p, batch);
  }
  if (shape == 5) {
    *name = smallk ? "bf16_nt_ragged_128x192x64_smallk" : "bf16_nt_ragged_128x192x64";
    return smallk ? launch_ragged<128, 192, 2, 2, true>(is_bf16, out_f32, p, batch)
                  : launch_ragged<128, 192, 2, 2, false>(is_bf16, out_f32, p, batch);
  }
  *name = smallk ? "bf16_nt_ragged_128x128x64_smallk" : "bf16_nt_ragged_128x128x64";
  return smallk ? launch_ragged<128, 128, 2, 2, true>(is_bf16, out_f32, p, batch)
                : launch_ragged<128, 128, 2, 2, false>(is_bf16, out_f32, p, batch);
}

}  // namespace tnh
