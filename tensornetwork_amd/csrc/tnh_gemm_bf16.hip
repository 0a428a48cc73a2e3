// K2 (speed path): bf16 / f16 GEMM on the gfx950 matrix cores.
//
//   C[M,N] = A[M,K] * B[N,K]^T   ("NT": both operands K-contiguous, which is
//   what the transpose+reshape lowering of tensordot produces when a's
//   contracted axes are trailing and b's are trailing; other layouts are
//   brought here by one K1 permute on the host side).
//
// Structure (per workgroup, BM x BN x 64 tile, WAVES_M x WAVES_N wavefronts):
//   * HBM -> LDS with global_load_lds_dwordx4 (16 B/lane, no VGPR round trip),
//     two LDS stages; the load of tile t+1 is issued right after the barrier
//     that publishes tile t, so it overlaps the MFMA work on tile t.
//   * LDS image: rows of 64 bf16 (128 B); the eight 16-B chunks of a row are
//     XOR-swizzled with (row & 7).  LDS-DMA writes lane-linear, so the swizzle
//     is applied to the per-lane SOURCE chunk and again on the ds_read_b128
//     address (same involution) -> conflict-free fragment reads.
//   * v_mfma_f32_16x16x32_{bf16,f16}, fp32 accumulate, operands swapped
//     (mfma(Bfrag, Afrag)) so every lane ends up with 4 consecutive n for one
//     m: the epilogue stores 8-B (bf16/f16) or 16-B (f32) vectors.
//   * workgroup ids are remapped XCD-aware (ids congruent mod 8 share an L2)
//     and then grouped along M so that neighbouring tiles reuse B panels.
//
// MFMA roofline: 2*M*N*K flop against 2.5 PFLOP/s dense bf16.
#include "tnh_gemm_nt.h"
#include <type_traits>
#include <utility>

namespace tnh {

int g_opt_raster = -1;  // A/B knobs, set through tnh_gemm_set_variant("name:r<d>:p<d>"); -1 = by shape (pick_raster)

// Tile order of the 256 x 256 kernels.  Raster 1 (16 x 16 super-tiles shared by the 8 XCDs) wins wherever there are
// many tiles (65536^3: 1477 vs 1464 TF, 36864^3: 1545 vs 1482); a grid of at most four resident waves of tiles with
// a very long K prefers the per-XCD M-grouped order (8192 x 8192 x 262144, the D = 512 row: 1530 vs 1505 TF;
// profiles/r03_raster_ab.txt).  Round 6: so does a TALL result (eight times more tile rows than tile columns) with a short
// contraction -- 262144 x 32768 x 1024: 1216 vs 1150 TF, 524288 x 16384 x 2048: 1416 vs 1352, 262144 x 8192 x 2048: 1418 vs
// 1349, the MERA layer's 1048576 x 32768 x 1024: 1181 vs 1167, the chi = 64 slices' 262144 x 4096 x 4096: 1497 vs 1475; a WIDE
// one does not (32768 x 1048576 x 1024: 1168 vs 1189), nor a square one (131072^2 x 512: 811 vs 902)
// -- and so does a large grid (>= 8192 tiles, at least as many tile rows as columns) with a contraction of 1536 ... 4096:
// 32768^2 x 1536 / 2048 / 3072 / 4096: +3.5 / +4.4 / +4.3 / +2.5 %, 65536^2 x 2048 / 4096: +4.5 / +3 %, 65536 x 16384 x 2048 /
// 4096: +4.2 / +3.6 %; below (32768^2 x 1024: -3 %) and above (x 6144 / 8192: -2.8 / -4.2 %, 65536^2 x 8192: -6 %) the
// super-tiles win, as they do at 16384 x 65536 x 2048 (profiles/r06_raster_tall_ab.txt; tools/raster_ab.sh, raster_ab2.sh)
// Short contractions (K <= 1536) on such grids take the M-grouped order with groups of SIXTEEN tile rows (raster 3):
// 65536^2 x 1024: 1222 against 1155 (super-tiles) / 1188 (groups of 8), 1048576 x 32768 x 1024 (the MERA layer): 1232 / 1167 /
// 1191, 262144 x 32768 x 512: 944 / 888 / 926, 32768^2 x 1536: 1310 / 1240 / 1290; at K = 4096 groups of 16 lose 12 %.
static int pick_raster(int64_t M, int64_t N, int64_t K) {
  if (g_opt_raster >= 0) return g_opt_raster;
  const int64_t tm = (M + 255) / 256, tn = (N + 255) / 256, tiles = tm * tn;
  if (tiles <= 1024 && K >= 65536) return 0;       // 8192 x 8192 x 65536: 1527 vs 1487 (profiles/r03_gemm_pj_per_flop.md)
  if (tiles >= 8192 || tm >= 8 * tn) {             // large or tall grids
    if (K <= 1536) return 3;
    if (K <= 4096 && tm >= tn) return 0;
  }
  return 1;
}
int g_opt_phases = 2;  // ping-pong kernel: MFMA clusters per K-tile (2 = 32-MFMA clusters, default; 4)
int g_opt_tail = 1;    // view GEMM: split-K launch for the last, mostly empty wave of tiles (":t0" switches it off)
int g_opt_persist = -1;  // ping-pong kernels: one workgroup per CU looping over tiles (":g0" / ":g1"; -1: when tiles > CUs)

// grid.x of a ping-pong launch over `tiles` output tiles
static unsigned pp_grid_x(int64_t tiles, unsigned grid_y) {
  const int cus = num_cus() > 0 ? num_cus() : 256;
  static const int env = []() { const char* e = getenv("TNH_GEMM_PERSIST"); return e ? atoi(e) : -1; }();
  const int mode = g_opt_persist >= 0 ? g_opt_persist : env;
  const bool on = (mode != 0) && grid_y == 1 && tiles > cus;
  return on ? (unsigned)cus : (unsigned)tiles;
}
static bool g_pp_default = true;  // ping-pong kernel won the A/B on MI355X (profiles/r01_sweep_v2.jsonl)

#define TNH_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define TNH_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// One 1-KiB LDS-DMA piece: lane l copies 16 B from its own global address to
// LDS byte (lds_dst + 16*l); lds_dst is wave-uniform and travels in M0.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool IS_BF16, bool OUT_F32>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_nt_kernel(NtArgs p) {
  constexpr int BK = 64;
  constexpr int NWAVES = WAVES_M * WAVES_N;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int FM = WTM / 16, FN = WTN / 16;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / 8 / NWAVES;  // 1 KiB LDS-DMA pieces per wave per tile
  constexpr int B_INSTR = BN / 8 / NWAVES;
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;

  int tm, tn;
  tile_of_block(blockIdx.x, p.tiles_m, p.tiles_n, p.raster, tm, tn);
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const uint16_t* A = p.A + (int64_t)blockIdx.y * p.sA;
  const uint16_t* B = p.B + (int64_t)blockIdx.y * p.sB;

  // ---- per-lane global source pointers of the LDS-DMA pieces -----------------
  // piece (i, wave): rows rb*8 .. rb*8+7 of the tile, rb = i*NWAVES + wid; lane l
  // lands at LDS (row rb*8 + (l>>3), chunk l&7) and fetches global chunk
  // (l&7) ^ (l>>3)   [row & 7 == l >> 3].
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ lrow;
  const uint16_t* ga[A_INSTR];
  const uint16_t* gb[B_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    int64_t row = m0 + (i * NWAVES + wid) * 8 + lrow;
    if (row >= p.M) row = p.M - 1;  // ragged edge: re-read the last row, never stored
    ga[i] = A + row * p.lda + lchunk * 8;
  }
#pragma unroll
  for (int i = 0; i < B_INSTR; ++i) {
    int64_t row = n0 + (i * NWAVES + wid) * 8 + lrow;
    if (row >= p.N) row = p.N - 1;
    gb[i] = B + row * p.ldb + lchunk * 8;
  }

  // LDS-DMA is issued from inline asm so that hipcc's waitcnt pass does not
  // see it: with the builtin it drains vmcnt(0) in front of the first ds_read
  // of every tile, which serialises the prefetch behind the MFMA work.  The
  // waits are therefore ours: s_waitcnt vmcnt(0) + barrier before a stage is read.
  const unsigned lds0 = (unsigned)(size_t)TNH_LDS_PTR(smem);
  auto stage = [&](int s, int64_t k0) {
    const unsigned base = lds0 + s * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i)
      glds16(ga[i] + k0, __builtin_amdgcn_readfirstlane(base + (i * NWAVES + wid) * 1024));
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i)
      glds16(gb[i] + k0, __builtin_amdgcn_readfirstlane(base + A_BYTES + (i * NWAVES + wid) * 1024));
  };

  // ---- fragment read offsets (bytes inside a tile image) ------------------------
  // fragment row = 16*f + (l & 15): row & 7 == l & 7, so the swizzled chunk is a
  // per-lane constant for each of the two k-steps.
  int frag_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
    frag_off[ks] = (lane & 15) * 128 + (((ks * 4 + (lane >> 4)) ^ (lane & 7)) * 16);

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nt = (int)(p.K / BK);
  stage(0, 0);
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile t visible to every wave; stage (t+1)&1 no longer read
    if (t + 1 < nt) stage((t + 1) & 1, (int64_t)(t + 1) * BK);
    const char* sa = smem + (t & 1) * STAGE_BYTES + (wm * WTM) * 128;
    const char* sb = smem + (t & 1) * STAGE_BYTES + A_BYTES + (wn * WTN) * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
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

  store_wave_tile<IS_BF16, OUT_F32, FM, FN>(acc, p, (char*)p.C + (int64_t)blockIdx.y * p.sC * (OUT_F32 ? 4 : 2),
                                            m0, n0, BM, BN, wm * WTM, wn * WTN, lane);
}

// ---------------------------------------------------------------------------
// 256x256x64 "ping-pong" kernel: the two halves of the workgroup (waves 0-3 own
// rows 0-127, waves 4-7 rows 128-255) run the same phase sequence ONE barrier
// interval apart, so that on every SIMD one wave is inside its MFMA cluster
// while the other issues its ds_reads / LDS-DMA.  A K-tile is consumed in 4
// phases (one 64x32 quadrant of the wave's 128x64 output per phase); the tile
// is staged as 4 half-tiles (A rows 0-127 / 128-255, B rows 0-127 / 128-255,
// 16 KiB each), one per phase, 2 LDS buffers:
//
//   phase of tile t | ds_read (tile t)   | LDS-DMA issued          | MFMA cluster
//   ----------------+--------------------+-------------------------+-------------
//        0          | A sub0 (8) B sub0 (4) | A-half0 of tile t+1  | A0 x B0
//        1          | B sub1 (4)         | A-half1 of tile t+1     | A0 x B1
//        2          | A sub1 (8)         | B-half0 of tile t+2     | A1 x B1
//        3          | -                  | B-half1 of tile t+2     | A1 x B0
//
// Hazards (interval = time between two workgroup barriers; group 1 lags group 0
// by one interval):
//   RAW  every wave waits `vmcnt(4)` at the end of phase 3's load segment (the 4
//        pieces of tile t+2's B halves stay in flight), then a barrier is passed
//        by everyone before the first ds_read of tile t+1.
//   WAR  a half-tile buffer is restaged at least one full interval after the
//        last ds_read of its previous content, and every load segment ends with
//        lgkmcnt(0) BEFORE its barrier, so those reads have retired.
//
// M32 = false: v_mfma_f32_16x16x32 (16 per phase), chunk swizzle row & 7.
// M32 = true : v_mfma_f32_32x32x16 ( 8 per phase; 4x fewer operand-register
//              reads per MAC), chunk swizzle (row >> 1) & 7 -- conflict-free for
//              the 32-row x 2-chunk ds_read_b128 pattern.
// ---------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <bool IS_BF16>
__device__ __forceinline__ f32x16 mma32(const uint4& a, const uint4& b, f32x16 c) {
  if constexpr (IS_BF16)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&a, *(const bf16x8*)&b, c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)&a, *(const f16x8*)&b, c, 0, 0, 0);
}

// VIEW = true (tnh_gemm_view; needs TWO, not M32): operands are read IN PLACE through an OpView --
// two-level row / contraction strides, and per operand either K-contiguous (the LDS image and the
// ds_read_b128 fragments above) or k-major (A_KM / B_KN: the operand is stored [k][row], rows
// contiguous -- what tensordot's [K][N] operand is).  A k-major half-tile is staged as [64 k][128 rows]
// (256-B LDS rows, 4 k-rows per LDS-DMA piece) and its MFMA fragments -- 8 consecutive k of one row per
// lane -- come from two ds_read_b64_tr_b16 (the LDS transpose read: each 16-lane group fetches a
// [4 k][16 rows] block and lane i receives column i; semantics pinned by tools/tr_probe.hip).  Bank
// conflicts: a 32-lane half of one transpose read touches 8 k-rows (k & 3, bit 3 of k) x 32 B; the
// 32-B unit index of every row is XOR-ed with h(k) = (k & 3) | ((k >> 3) & 1) << 2, on the LDS-DMA
// source side and again on the read address, so the 8 segments fall on 8 different 32-B bank windows.
// Same MFMA sequence as the NT kernel, so results are bit-identical to permute + NT.
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;
typedef __attribute__((address_space(3))) char lds_char;

// walks the element offsets of consecutive HALF K-tiles (32 contraction indices) of one operand (wave-uniform, SALU
// only): the inner contraction run of a view is a multiple of 32, so a 64-deep K-tile may take its two halves from
// two different runs (D = 96: runs of 96 = 3 halves; chi = 32: every half its own run).
struct KWalk {
  int64_t off, step, wrap;
  int in, hpr;
  __device__ __forceinline__ void init(const OpView& v, int tile) {
    hpr = v.tpi;
    step = 32 * v.sk0;
    wrap = v.sk1 - (int64_t)v.tpi * step;
    const int half = 2 * tile;
    in = half % v.tpi;
    off = (int64_t)(half / v.tpi) * v.sk1 + (int64_t)in * step;
  }
  __device__ __forceinline__ void advance_half() {
    ++in;
    off += step;
    if (in == hpr) {
      in = 0;
      off += wrap;
    }
  }
  // KW = 1 (inner contraction runs that are multiples of 64): the same walk in whole K-tiles
  __device__ __forceinline__ void init_tiles(const OpView& v, int tile) {
    hpr = v.tpi >> 1;
    step = 64 * v.sk0;
    wrap = v.sk1 - (int64_t)hpr * step;
    in = tile % hpr;
    off = (int64_t)(tile / hpr) * v.sk1 + (int64_t)in * step;
  }
  __device__ __forceinline__ void next_tile(int64_t& h0) {
    h0 = off;
    ++in;
    off += step;
    if (in == hpr) {
      in = 0;
      off += wrap;
    }
  }
  // offsets of the two halves of the next K-tile
  __device__ __forceinline__ void next(int64_t& h0, int64_t& h1) {
    h0 = off;
    advance_half();
    h1 = off;
    advance_half();
  }
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// ---- helpers of the lean main loop (SCHED = 3) ---------------------------------------------------------------------
// One 1-KiB LDS-DMA piece in the SADDR form: lane address = 64-bit wave-uniform base (SGPR pair) + 32-bit per-lane
// byte offset -- no 64-bit VALU add per piece; M0 = LDS destination = wave-uniform base + an immediate, written by
// ONE scalar add inside the asm block (M0 is not saved / restored: the compiler sets M0 itself right before each of
// its own uses and nothing in these kernels keeps a value in it).
template <int IMM>
__device__ __forceinline__ void glds16s(unsigned voff, const void* sbase, unsigned lds_base) {
  asm volatile(
      "s_add_u32 m0, %1, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %3"
      :
      : "v"(voff), "s"(lds_base), "n"(IMM), "s"(sbase)
      : "memory", "scc");       // s_add_u32 writes SCC
}
// The same piece in two statements, for callers that put another instruction (a fragment read) between them: that
// instruction is the wait state the M0 write needs before an LDS-DMA reads it, and the s_nop goes away.
template <int IMM>
__device__ __forceinline__ void glds_m0(unsigned lds_base) {
  asm volatile("s_add_u32 m0, %0, %1" : : "s"(lds_base), "n"(IMM) : "scc");
}
__device__ __forceinline__ void glds_go(unsigned voff, const void* sbase) {
  asm volatile("global_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase) : "memory");
}
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
// A fragment read the compiler does not track (no s_waitcnt lgkmcnt of its own in front of every MFMA: the load
// segments end with our lgkmcnt(0) anyway)
template <int OFF>
__device__ __forceinline__ void lds_frag(uint4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(*(u32x4_t*)&dst) : "v"(addr), "n"(OFF));
}

// A k-major fragment (8 consecutive k of one row per lane) as two untracked transposing reads: k rows 0-3 and 4-7 of
// the lane's group (the second 1 KiB = four 256-byte k rows further on)
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
template <int OFF>
__device__ __forceinline__ void lds_frag_tr(uint4& dst, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(*(u32x2_t*)&dst.x) : "v"(addr), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(*(u32x2_t*)&dst.z) : "v"(addr), "n"(OFF + 1024));
}

// SCHED = 1 (round 5, A/B knob ":p7"): ONE 64-MFMA cluster per K-tile -- every fragment of the K-tile (24 ds_read_b128
// = 96 registers) is read in one load segment, two barriers per K-tile instead of four.  LDS: A double-buffered
// (2 x 32 KiB), B triple-buffered (3 x 32 KiB) = 160 KiB, the whole CU.  Group g (waves 4g .. 4g+3) loads what it
// alone reads of A -- half g, four pieces per wave -- and half g of B, always in its LOAD segment:
//
//   load segment of K-tile t (group g, interval 2t + g):  24 fragment reads of tile t;
//        A-half g of tile t+1 -> A buffer (t+1) & 1   (last read by this group in its load segment of tile t-1)
//        B-half g of tile t+2 -> B slot (t+2) % 3     (tile t-1's slot, last read by group 1 in interval 2t-1)
//   RAW  group 1 ends its load segment with vmcnt(8): its B-half 1 of tile t+1 (issued one K-tile ago) has landed
//        before group 0 reads tile t+1 in the next interval; every wave ends its MFMA cluster with vmcnt(4): its A
//        pieces of tile t+1 and its older B pieces have landed, only B of tile t+2 stays in flight.
//   distances: A two intervals (~2 x 1100 cycles), B three to four.
// (The first form of this schedule -- whole buffers restaged two intervals ahead, group 1 issuing at the head of its
// MFMA cluster -- lost 8-20 % to the default: profiles/r05_gemm_variants.md.)
// KW (view kernels with K-contiguous operands; round 5): how the contraction index walks through memory.
//   0  general: inner runs that are multiples of 32 -- the walk advances in HALF K-tiles and every lane picks the half
//      its chunk belongs to (62 more instructions per K-tile and wave than the plain NT loop);
//   1  inner runs that are multiples of 64 on both operands: the walk advances in whole K-tiles, lanes carry their
//      full chunk offset;
//   2  one contiguous run on both operands: offset = K-tile x 64, the plain NT loop.  Measured on the headline shape
//      (65536^3, the same buffers): the general walk 1422-1425 TFLOP/s, the plain loop 1489-1496
//      (profiles/r05_gemm_headline_plain_vs_view.jsonl) -- which is why the view launcher picks the cheapest walk
//      the operands allow.  Same MFMA sequence in all three: bit-identical results.
template <bool IS_BF16, bool OUT_F32, bool TWO, bool M32 = false, bool VIEW = false, bool A_KM = false,
          bool B_KN = false, int SCHED = 0, int KW = 0>
__global__ __launch_bounds__(512) void gemm_nt_pp_kernel(NtArgs p) {
  static_assert(KW == 0 || (VIEW && !A_KM && (!B_KN || (SCHED == 3 && KW == 1))),
                "the tile-granular K walks: K-contiguous view operands, or (lean loop, KW = 1) a k-major B");
  static_assert(!VIEW || (TWO && !M32), "view kernels use the 2-phase 16x16x32 schedule");
  static_assert(SCHED == 0 || (TWO && !M32), "the round-5 schedules are 16x16x32 schedules");
  static_assert(SCHED == 0 || SCHED == 3 || (!A_KM && !B_KN), "one-cluster / snake: K-contiguous operands");
  // B_KT (round 6): a k-major B whose contraction runs are multiples of 64, next to a K-contiguous A, in the lean loop --
  // the whole-K-tile walk (both pieces of a half-tile hang off ONE scalar base: a lane's k row 0 .. 63 sits in its 32-bit
  // offset), the 16 transposing fragment reads as untracked asm from precomputed addresses (one VGPR per 32-B unit of
  // the fragment columns and LDS buffer; k-step and half as immediate offsets), and the K-contiguous loop's
  // interleaved issue of the LDS-DMA pieces: 8 more instructions per K-tile and wave than the NT loop (the 8 extra
  // transposing reads) instead of 94.
  constexpr bool B_KT = B_KN && KW == 1;
  // SCHED = 3: the default two-cluster schedule with a LEAN main loop -- the same fragment reads, LDS-DMA pieces, MFMAs
  // and barriers, and about 50 fewer bookkeeping instructions per K-tile and wave (205 -> ~150): SADDR-form LDS-DMA
  // (one 64-bit scalar add per operand and K-tile instead of a 64-bit VALU add per piece; M0 written by one scalar add,
  // not saved / restored), fragment reads as untracked asm (no compiler lgkmcnt waits between the MFMAs), one s_setprio
  // pair per cluster, the K loop unrolled by two so that both LDS buffers have static addresses (no per-K-tile VALU).
  // Why: on this kernel every instruction the load segment issues costs ~0.06 % (the 62 extra instructions of the
  // general view walk cost 4.3 %: profiles/r05_gemm_headline_plain_vs_view.jsonl).  Needs single-level rows (plain NT,
  // or a view with KW = 2 whose rows are linear) and row offsets inside a tile below 4 GiB (host check).
  constexpr bool LEAN = (SCHED == 3);
  // (the lean loop without the s_setprio pair around its clusters measured the same to 0.1 %:
  //  profiles/r05_gemm_setprio_ab.jsonl -- the pair stays)
  // (KW = 0 in the lean loop: the full chunk offset sits in the lane offsets as for KW >= 1, and in a K-tile whose halves
  //  come from two runs the lanes of the second half add the distance between the runs beyond 32 elements -- one v_and
  //  + four v_add per operand, in those K-tiles only)
  constexpr bool ONE = (SCHED == 1);
  // SCHED = 2 (A/B knob ":p8"): the default schedule with the MFMAs of a quadrant in snake order -- every issue
  // changes exactly one of the two operand registers (fewer operand-bus toggles; same sums, bit-identical)
  constexpr bool SNAKE = (SCHED == 2);
  constexpr int ASUBS = ONE ? 2 : 1;
  static_assert(VIEW || (!A_KM && !B_KN), "k-major operands need the view kernel");
  // M32 (32x32x16 MFMA) lost the A/B with 4 phases per K-tile (only 2 independent accumulators per
  // phase); variant ":p3" re-tests it with the 2-phase schedule (4 independent accumulators).
  constexpr int BM = 256, BN = 256, BK = 64;
  constexpr int HALF_BYTES = 128 * BK * 2;   // 16 KiB
  constexpr int BUF_BYTES = 4 * HALF_BYTES;  // A0 A1 B0 B1
  constexpr int KS = M32 ? 4 : 2;            // MFMA k-steps per K-tile
  constexpr int FA = M32 ? 2 : 4;            // row fragments per 64-row A sub-tile
  constexpr int FB = M32 ? 1 : 2;            // row fragments per 32-row B sub-tile
  constexpr int FROWS = M32 ? 32 : 16;
  __shared__ __attribute__((aligned(1024))) char smem[ONE ? 10 * HALF_BYTES : 2 * BUF_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;

  // Persistent over tiles: workgroup b computes tiles b, b + gridDim.x, ... (gridDim.x = tiles: one each, as before;
  // gridDim.x = CUs: the loads of the next tile's first K-tiles are issued BEFORE the epilogue of the current one,
  // and the workgroup hand-over on the CU -- which can hold only one of these 128-KiB-LDS workgroups -- is gone).
  // The hardware deals consecutive workgroup ids round-robin to the XCDs, so tile t still runs on XCD t & 7 and
  // the rasters of tile_of_block keep their meaning.
  const int ntiles = p.tiles_m * p.tiles_n;
  int64_t m0 = 0, n0 = 0;
  const uint16_t* A = p.A + (int64_t)blockIdx.y * p.sA;
  const uint16_t* B = p.B + (int64_t)blockIdx.y * p.sB;

  // LDS-DMA source pointers: half-tile h, piece i -> rows h*128 + (i*8 + wid)*8 + (lane >> 3);
  // the lane that lands in chunk slot (lane & 7) fetches global chunk slot ^ swz(row).
  const int lrow = lane >> 3;
  const uint16_t* ga[2][2];
  const uint16_t* gb[2][2];
  // K-contiguous operand: piece (h, i) -> tile rows h*128 + (i*8 + wid)*8 + (lane >> 3), 16-B chunk (lane & 7) ^ swz.
  // k-major operand:      piece (h, i) -> k rows (i*8 + wid)*4 + (lane >> 4) of the K-tile, and the 16-B chunk of
  //                       the half-tile's 128 rows that lands in slot (lane & 15): 32-B unit ((lane & 15) >> 1) ^ h(k).
  auto src_ptr = [&](const uint16_t* base, const OpView& v, bool kmajor, int64_t first, int64_t limit, int64_t ld,
                     int h, int i) -> const uint16_t* {
    if (kmajor) {
      const int krow = (i * 8 + wid) * 4 + (lane >> 4);
      const int s16 = lane & 15;
      const int hsw = (lane >> 4) | (((wid >> 1) & 1) << 2);                 // h(krow): see the kernel header
      const int chunk = ((((s16 >> 1) ^ hsw)) << 1) | (s16 & 1);
      int64_t col = first + h * 128 + chunk * 8;
      if (col + 8 > limit) col = limit - 8;                                  // ragged edge: valid memory, never stored
      const uint32_t c1 = (uint32_t)col / (uint32_t)v.r0, c0 = (uint32_t)col - c1 * (uint32_t)v.r0;   // rows < 2^31 (host check)
      return base + (int64_t)c1 * v.sr1 + c0 + (int64_t)(krow & 31) * v.sk0;      // piece i = half i of the K-tile
    }
    const int trow = ONE ? (i * 4 + wc) * 8 + lrow      // one-cluster schedule: the 4 waves of a group cover a half-tile
                         : (i * 8 + wid) * 8 + lrow;    // row inside the half-tile
    const int swz = M32 ? ((trow >> 1) & 7) : (trow & 7);
    const int lchunk = (lane & 7) ^ swz;
    int64_t row = first + h * 128 + trow;
    if (row >= limit) row = limit - 1;
    if constexpr (VIEW) {
      const uint32_t r1 = (uint32_t)row / (uint32_t)v.r0, r0 = (uint32_t)row - r1 * (uint32_t)v.r0;
      return base + (int64_t)r1 * v.sr1 + (int64_t)r0 * v.sr0 + (KW >= 1 ? lchunk : (lchunk & 3)) * 8;   // KW 0: chunks 4-7 = second half, offset in k1
    }
    return base + row * ld + lchunk * 8;
  };
  const uint16_t* a_next = nullptr;     // lean loop with a contiguous K: wave-uniform address of the next K-tile A / B stage
  const uint16_t* b_next = nullptr;
  uint32_t oa[2][2], ob[2][2];          // lean loop: byte offset of this lane's piece (h, i) from row 0, k 0 of the tile
  const uint16_t* abase = nullptr;      // lean loop: wave-uniform address of the tile's first row of A / B
  const uint16_t* bbase = nullptr;
  auto setup_tile = [&](int t) {
    int tm, tn;
    tile_of_block(t, p.tiles_m, p.tiles_n, p.raster, tm, tn);
    m0 = (int64_t)tm * BM;
    n0 = (int64_t)tn * BN;
    if constexpr (LEAN) {
      // element offset of row r of an operand from the operand's base (two-level rows of a view; r < 2^31: host check)
      auto row_elems = [&](const OpView& v, int64_t ld, int64_t r) -> int64_t {
        if constexpr (VIEW) {
          const uint32_t r1 = (uint32_t)r / (uint32_t)v.r0, r0 = (uint32_t)r - r1 * (uint32_t)v.r0;
          return (int64_t)r1 * v.sr1 + (int64_t)r0 * v.sr0;
        }
        return r * ld;
      };
      auto row_elems_km = [&](const OpView& v, int64_t c) -> int64_t {       // k-major: rows are the contiguous direction
        const uint32_t c1 = (uint32_t)c / (uint32_t)v.r0, c0 = (uint32_t)c - c1 * (uint32_t)v.r0;
        return (int64_t)c1 * v.sr1 + c0;
      };
      const int64_t fa = m0 + p.m_off, la = p.M + p.m_off;
      const int64_t a0 = !p.lean_rel_a ? 0 : (A_KM ? row_elems_km(p.va, fa) : row_elems(p.va, p.lda, fa));
      const int64_t b0 = !p.lean_rel_b ? 0 : (B_KN ? row_elems_km(p.vb, n0) : row_elems(p.vb, p.ldb, n0));
      // (the tile's first row is the same for every lane: made scalar here so that the SADDR operand stays in SGPRs)
      abase = A + (((int64_t)__builtin_amdgcn_readfirstlane((int)(a0 >> 32)) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a0));
      bbase = B + (((int64_t)__builtin_amdgcn_readfirstlane((int)(b0 >> 32)) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b0));
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int trow = (i * 8 + wid) * 8 + lrow;
          const int lchunk = (lane & 7) ^ (trow & 7);
          // k-major operand (stored [k][row], rows contiguous): piece i = k rows (i * 8 + wid) * 4 + (lane >> 4) of the
          // K-tile (i.e. half i), 16-byte chunk of the half-tile's 128 rows as in src_ptr; its K offset comes per piece
          const int krow_tile = (i * 8 + wid) * 4 + (lane >> 4);
          const int krow = krow_tile & 31;
          const int s16 = lane & 15;
          const int hsw = (lane >> 4) | (((wid >> 1) & 1) << 2);
          const int kchunk = ((((s16 >> 1) ^ hsw)) << 1) | (s16 & 1);
          if constexpr (A_KM) {
            int64_t col = fa + h * 128 + kchunk * 8;
            if (col + 8 > la) col = la - 8;
            oa[h][i] = (uint32_t)((row_elems_km(p.va, col) - a0 + (int64_t)krow * p.va.sk0) * 2);
          } else {
            int64_t ra = fa + h * 128 + trow;
            if (ra >= la) ra = la - 1;              // ragged edge: re-read the last row, never stored
            oa[h][i] = (uint32_t)((row_elems(p.va, p.lda, ra) - a0 + lchunk * 8) * 2);
          }
          if constexpr (B_KN) {
            int64_t col = n0 + h * 128 + kchunk * 8;
            if (col + 8 > p.N) col = p.N - 8;
            ob[h][i] = (uint32_t)((row_elems_km(p.vb, col) - b0 + (int64_t)(B_KT ? krow_tile : krow) * p.vb.sk0) * 2);
          } else {
            int64_t rb = n0 + h * 128 + trow;
            if (rb >= p.N) rb = p.N - 1;
            ob[h][i] = (uint32_t)((row_elems(p.vb, p.ldb, rb) - b0 + lchunk * 8) * 2);
          }
        }
      return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if constexpr (ONE) {     // four pieces of the half this wave's group owns (piece 2 h + i)
          ga[h][i] = src_ptr(A, p.va, false, m0 + p.m_off, p.M + p.m_off, p.lda, wr, 2 * h + i);
          gb[h][i] = src_ptr(B, p.vb, false, n0, p.N, p.ldb, wr, 2 * h + i);
        } else {
          ga[h][i] = src_ptr(A, p.va, A_KM, m0 + p.m_off, p.M + p.m_off, p.lda, h, i);
          gb[h][i] = src_ptr(B, p.vb, B_KN, n0, p.N, p.ldb, h, i);
        }
      }
  };
  const unsigned lds0 = (unsigned)(size_t)TNH_LDS_PTR(smem);
  // which: 0 = A-half0, 1 = A-half1, 2 = B-half0, 3 = B-half1 (halves of the tile's ROWS).  k0 / k1: element offsets of
  // the two 32-deep halves of the K-tile (plain NT kernels: k1 unused, the pointers carry the full chunk offset).
  // A K-contiguous row keeps chunks 0-3 of its 128-B LDS row in the first half and 4-7 in the second (the chunk a
  // lane fetches is (lane & 7) ^ swizzle(row), the swizzle being lane >> 3 for every piece); a k-major piece i is
  // half i.
  const bool hi_half = ((((lane & 7) ^ (lane >> 3)) >> 2) & 1) != 0;
  auto issue = [&](int buf, int which, int64_t k0, int64_t k1) {
    const unsigned base = lds0 + buf * BUF_BYTES + which * HALF_BYTES;
    const bool kmajor = (which < 2) ? A_KM : B_KN;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint16_t* g = (which < 2) ? ga[which & 1][i] : gb[which & 1][i];
      int64_t k = k0;
      if constexpr (VIEW && KW == 0) k = kmajor ? (i == 0 ? k0 : k1) : (hi_half ? k1 : k0);
      glds16(g + k, __builtin_amdgcn_readfirstlane(base + (i * 8 + wid) * 1024));
    }
  };

  // one-cluster schedule: this wave's four pieces of its group's half of A (is_a) or B -> LDS half-tile at `base`
  auto issue_own = [&](unsigned base, bool is_a, int64_t k0, int64_t k1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint16_t* g = is_a ? ga[q >> 1][q & 1] : gb[q >> 1][q & 1];
      int64_t k = k0;
      if constexpr (VIEW && KW == 0) k = hi_half ? k1 : k0;
      glds16(g + k, __builtin_amdgcn_readfirstlane(base + (q * 4 + wc) * 1024));
    }
  };
  const unsigned own_a = lds0 + wr * HALF_BYTES;                    // + (tile & 1) * 2 * HALF_BYTES
  const unsigned own_b = lds0 + 4 * HALF_BYTES + wr * HALF_BYTES;   // + (tile % 3) * 2 * HALF_BYTES

  // lean loop: the 2 pieces of half-tile WHICH (0 = A-half0 .. 3 = B-half1) of the K-tile at element offset k -> buffer
  const unsigned lean_dst0 = __builtin_amdgcn_readfirstlane(lds0 + wid * 1024);
  // piece I (0 / 1) of half-tile WHICH in two steps (M0, then the load) for the interleaved main loop
  auto piece_m0 = [&](auto bufc, auto whichc, auto ic) {
    constexpr int BUF = decltype(bufc)::value, WHICH = decltype(whichc)::value, I = decltype(ic)::value;
    glds_m0<BUF * BUF_BYTES + WHICH * HALF_BYTES + I * 8 * 1024>(lean_dst0);
  };
  auto piece_go = [&](uint32_t voff, const void* sb) { glds_go(voff, sb); };
  const uint32_t himask = hi_half ? 0xffffffffu : 0u;      // lanes whose chunk lies in the second half of a K-tile
  // prologue form: both pieces of half-tile WHICH of the K-tile whose halves start at element offsets k0 / k1
  auto stage_lean = [&](auto bufc, auto whichc, int64_t k0, int64_t k1) {
    constexpr int BUF = decltype(bufc)::value, WHICH = decltype(whichc)::value;
    constexpr bool KMAJOR = (WHICH < 2) ? A_KM : (B_KN && !B_KT);     // (B_KT: one base per half-tile, as K-contiguous)
    const uint16_t* xb = (WHICH < 2) ? abase : bbase;
    const uint32_t o0 = (WHICH < 2) ? oa[WHICH & 1][0] : ob[WHICH & 1][0];
    const uint32_t o1 = (WHICH < 2) ? oa[WHICH & 1][1] : ob[WHICH & 1][1];
    if constexpr (KMAJOR) {       // piece i is half i of the K-tile: its own scalar base
      glds16s<BUF * BUF_BYTES + WHICH * HALF_BYTES>(o0, (const void*)(xb + k0), lean_dst0);
      glds16s<BUF * BUF_BYTES + WHICH * HALF_BYTES + 8 * 1024>(o1, (const void*)(xb + k1), lean_dst0);
    } else {
      uint32_t extra = 0;
      if constexpr (VIEW && KW == 0) extra = himask & (uint32_t)((k1 - k0 - 32) * 2);
      const void* sb = (const void*)(xb + k0);
      glds16s<BUF * BUF_BYTES + WHICH * HALF_BYTES>(o0 + extra, sb, lean_dst0);
      glds16s<BUF * BUF_BYTES + WHICH * HALF_BYTES + 8 * 1024>(o1 + extra, sb, lean_dst0);
    }
  };

  // fragment read offsets inside a half-tile image (row base is a multiple of FROWS)
  int frag_off[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if constexpr (M32)
      frag_off[ks] = (lane & 31) * 128 + (((ks * 2 + (lane >> 5)) ^ ((lane >> 1) & 7)) * 16);
    else
      frag_off[ks] = (lane & 15) * 128 + (((ks * 4 + (lane >> 4)) ^ (lane & 7)) * 16);
  }

  f32x4 acc[M32 ? 1 : 8][M32 ? 1 : 4];
  f32x16 acc32[M32 ? 4 : 1][M32 ? 2 : 1];
  auto zero_acc = [&]() {
  if constexpr (M32) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  };

  uint4 af[ASUBS][KS][FA];   // [sub (one-cluster schedule: both)][k-step][row fragment] of the current A sub-tile
  uint4 bf[2][KS][FB];  // [sub][k-step][row fragment] of both B sub-tiles

  // lean loop: LDS byte addresses of this lane's fragment rows, wave's sub-tile origin included, per k-step; buffer 1
  // is BUF_BYTES further on (beyond the 16-bit offset field, hence its own registers)
  unsigned a_rd[2][KS], b_rd[2][KS];
  if constexpr (LEAN) {
#pragma unroll
    for (int bq = 0; bq < 2; ++bq)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a_rd[bq][ks] = lds0 + bq * BUF_BYTES + wr * HALF_BYTES + frag_off[ks];
        b_rd[bq][ks] = lds0 + bq * BUF_BYTES + (2 + (wc >> 1)) * HALF_BYTES + ((wc & 1) * 64) * 128 + frag_off[ks];
      }
  }
  auto lean_read_a = [&](auto bufc, auto subc) {
    constexpr int BUF = decltype(bufc)::value, SUB = decltype(subc)::value;
    static_for<0, 2>([&](auto ks) {
      static_for<0, 4>([&](auto f) {
        lds_frag<SUB * 64 * 128 + decltype(f)::value * 16 * 128>(af[0][decltype(ks)::value][decltype(f)::value],
                                                                 a_rd[BUF][decltype(ks)::value]);
      });
    });
  };
  // one fragment: Q = 0..7 -> A (sub, ks = Q >> 2, f = Q & 3); used by the interleaved issue below
  auto lean_read_a1 = [&](auto bufc, auto subc, auto qc) {
    constexpr int BUF = decltype(bufc)::value, SUB = decltype(subc)::value, Q = decltype(qc)::value;
    lds_frag<SUB * 64 * 128 + (Q & 3) * 16 * 128>(af[0][Q >> 2][Q & 3], a_rd[BUF][Q >> 2]);
  };
  auto lean_read_b = [&](auto bufc, auto subc) {
    constexpr int BUF = decltype(bufc)::value, SUB = decltype(subc)::value;
    static_for<0, 2>([&](auto ks) {
      static_for<0, 2>([&](auto f) {
        lds_frag<SUB * 32 * 128 + decltype(f)::value * 16 * 128>(bf[SUB][decltype(ks)::value][decltype(f)::value],
                                                                 b_rd[BUF][decltype(ks)::value]);
      });
    });
  };

  // k-major images: per-lane part of the transpose-read address.  Lane (g = lane >> 4, i = lane & 15)
  // reads k row 32 ks + 8 g + 4 r + (i >> 2), bytes 8 (i & 3) .. +7 of 32-B unit u ^ h:
  //   byte = [8 g + (i >> 2)] * 256 + ((u ^ h) << 5) + 8 (i & 3) + ks * 8192 + r * 1024,   h = (i >> 2) | (g & 1) << 2
  // the row term, the unit term and the in-unit term occupy disjoint bits, so (base | h << 5) ^ (u << 5) is the address.
  const unsigned tr_lane = (unsigned)((8 * (lane >> 4) + ((lane & 15) >> 2)) * 256 + 8 * (lane & 3)) |
                           (unsigned)((((lane & 15) >> 2) | (((lane >> 4) & 1) << 2)) << 5);
  const unsigned tr_a = tr_lane;                                  // A units: sub * 4 + f
  const unsigned tr_b = tr_lane ^ (unsigned)(((wc & 1) * 4) << 5);  // B units: (wc & 1) * 4 + sub * 2 + f
  // B_KT: LDS byte address of this lane's transposing reads per buffer and 32-B unit u = sub * 2 + f of the wave's 64
  // columns; the k-step (8 KiB) and the second half of a fragment (k rows + 4: 1 KiB) are immediate offsets
  unsigned b_tr[2][4];
  if constexpr (B_KT) {
#pragma unroll
    for (int bq = 0; bq < 2; ++bq)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        b_tr[bq][u] = lds0 + bq * BUF_BYTES + (2 + (wc >> 1)) * HALF_BYTES + (tr_b ^ (unsigned)(u << 5));
  }
  auto lean_read_b_tr = [&](auto bufc, auto subc) {
    constexpr int BUF = decltype(bufc)::value, SUB = decltype(subc)::value;
    static_for<0, 2>([&](auto ks) {
      static_for<0, 2>([&](auto f) {
        constexpr int KS_ = decltype(ks)::value, F_ = decltype(f)::value;
        lds_frag_tr<KS_ * 8192>(bf[SUB][KS_][F_], b_tr[BUF][SUB * 2 + F_]);
      });
    });
  };
  auto tr_read = [&](unsigned lds_byte) -> uint4 {
    const v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)((lds_char*)TNH_LDS_PTR(smem) + lds_byte));
    const v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)((lds_char*)TNH_LDS_PTR(smem) + lds_byte + 1024));
    const uint2 l2 = *(const uint2*)&lo, h2 = *(const uint2*)&hi;
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
  };
  auto read_a = [&](const char* buf_base, int sub) {
    if constexpr (A_KM) {
      const unsigned base = (unsigned)(buf_base - smem) + wr * HALF_BYTES;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int f = 0; f < FA; ++f) af[sub % ASUBS][ks][f] = tr_read(base + (tr_a ^ (unsigned)((sub * 4 + f) << 5)) + ks * 8192);
    } else {
      const char* sa = buf_base + wr * HALF_BYTES + (sub * 64) * 128;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int f = 0; f < FA; ++f) af[sub % ASUBS][ks][f] = *(const uint4*)(sa + f * FROWS * 128 + frag_off[ks]);
    }
  };
  auto read_b = [&](const char* buf_base, int sub) {
    if constexpr (B_KN) {
      const unsigned base = (unsigned)(buf_base - smem) + (2 + (wc >> 1)) * HALF_BYTES;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int f = 0; f < FB; ++f) bf[sub][ks][f] = tr_read(base + (tr_b ^ (unsigned)((sub * 2 + f) << 5)) + ks * 8192);
    } else {
      const char* sb = buf_base + (2 + (wc >> 1)) * HALF_BYTES + ((wc & 1) * 64 + sub * 32) * 128;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int f = 0; f < FB; ++f) bf[sub][ks][f] = *(const uint4*)(sb + f * FROWS * 128 + frag_off[ks]);
    }
  };
  auto mma_quadrant = [&](int sa, int sb) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FB; ++j) {
          if constexpr (M32)
            acc32[sa * 2 + i][sb] = mma32<IS_BF16>(bf[sb][ks][j], af[sa % ASUBS][ks][i], acc32[sa * 2 + i][sb]);
          else {
            const int jj = (SNAKE && (i & 1)) ? FB - 1 - j : j;
            acc[sa * 4 + i][sb * 2 + jj] = mma16<IS_BF16>(bf[sb][ks][jj], af[sa % ASUBS][ks][i], acc[sa * 4 + i][sb * 2 + jj]);
          }
        }
    __builtin_amdgcn_s_setprio(0);
  };
  // lean loop: both quadrants of A sub-tile `sa` under ONE s_setprio pair (same MFMA order as two mma_quadrant calls)
  auto mma_cluster = [&](int sa, int sb_first) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int sb = sb_first ^ q;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < FA; ++i)
#pragma unroll
          for (int j = 0; j < FB; ++j)
            if constexpr (!M32)
              acc[sa * 4 + i][sb * 2 + j] = mma16<IS_BF16>(bf[sb][ks][j], af[0][ks][i], acc[sa * 4 + i][sb * 2 + j]);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // end of a load segment: retire this wave's LDS reads, then meet the other waves
#define TNH_SEG_LOAD_END()                                  \
  do {                                                      \
    __builtin_amdgcn_sched_barrier(0);                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
    __builtin_amdgcn_s_barrier();                           \
    __builtin_amdgcn_sched_barrier(0);                      \
  } while (0)
#define TNH_SEG_MMA_END()                                   \
  do {                                                      \
    __builtin_amdgcn_sched_barrier(0);                      \
    __builtin_amdgcn_s_barrier();                           \
    asm volatile("" ::: "memory");                          \
    __builtin_amdgcn_sched_barrier(0);                      \
  } while (0)

  int nt = (int)(p.K / BK);
  int kfirst = 0;
  if constexpr (VIEW) {
    if (p.kslice_tiles > 0) {            // split-K launch: this workgroup's slice of the contraction
      kfirst = (int)blockIdx.y * p.kslice_tiles;
      nt = (nt - kfirst < p.kslice_tiles) ? (nt - kfirst) : p.kslice_tiles;
    }
  }
  // element offset of the K-tile each operand stages next: A runs one tile ahead, B two
  KWalk wa, wb;
  auto walk = [&](KWalk& w, int64_t& h0, int64_t& h1, int kt) {      // element offsets of K-tile kt of this launch's slice
    if constexpr (KW == 2 || !VIEW) {
      h0 = (int64_t)(kfirst + kt) * BK;
      h1 = 0;
    } else if constexpr (KW == 1) {
      w.next_tile(h0);
      h1 = 0;
    } else {
      w.next(h0, h1);
    }
  };
  // prologue of an output tile (its LDS-DMA source pointers set by setup_tile): K-tile 0 complete, B halves of K-tile 1
  auto start_tile = [&]() {
    if constexpr (LEAN) {       // K-tile 0 complete -> buffer 0, the B halves of K-tile 1 -> buffer 1
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      if constexpr (VIEW && KW == 1) {
        wa.init_tiles(p.va, kfirst);
        wb.init_tiles(p.vb, kfirst);
      } else if constexpr (VIEW && KW == 0) {
        wa.init(p.va, kfirst);
        wb.init(p.vb, kfirst);
      }
      int64_t ka = 0, ka1 = 0, kb = 0, kb1 = 0, kc = 0, kc1 = 0;
      walk(wa, ka, ka1, 0);
      walk(wb, kb, kb1, 0);
      stage_lean(I0{}, I0{}, ka, ka1);
      stage_lean(I0{}, I1{}, ka, ka1);
      stage_lean(I0{}, std::integral_constant<int, 2>{}, kb, kb1);
      stage_lean(I0{}, std::integral_constant<int, 3>{}, kb, kb1);
      if (nt > 1) {
        walk(wb, kc, kc1, 1);
        stage_lean(I1{}, std::integral_constant<int, 2>{}, kc, kc1);
        stage_lean(I1{}, std::integral_constant<int, 3>{}, kc, kc1);
      }
      if constexpr (!VIEW || KW == 2) {       // running stage pointers: A one K-tile ahead, B two
        a_next = abase + (int64_t)(kfirst + 1) * BK;
        b_next = bbase + (int64_t)(kfirst + 2) * BK;
      }
      return;
    }
    int64_t ka0 = 0, ka1 = 0, kb0 = 0, kb1 = 0;
    if constexpr (VIEW) {
      if constexpr (KW == 1) {
        wa.init_tiles(p.va, kfirst);
        wb.init_tiles(p.vb, kfirst);
      } else if constexpr (KW == 0) {
        wa.init(p.va, kfirst);
        wb.init(p.vb, kfirst);
      }
      walk(wa, ka0, ka1, 0);
      walk(wb, kb0, kb1, 0);
    }
    if constexpr (ONE) {        // A of K-tiles 0, 1 and B of K-tiles 0, 1, 2 (this group's halves)
      issue_own(own_a, true, ka0, ka1);
      issue_own(own_b, false, kb0, kb1);
      if (nt > 1) {
        int64_t a0 = (int64_t)BK, a1 = 0, b0 = (int64_t)BK, b1 = 0;
        if constexpr (VIEW) {
          walk(wa, a0, a1, 1);
          walk(wb, b0, b1, 1);
        }
        issue_own(own_a + 2 * HALF_BYTES, true, a0, a1);
        issue_own(own_b + 2 * HALF_BYTES, false, b0, b1);
      }
      if (nt > 2) {
        int64_t b0 = (int64_t)2 * BK, b1 = 0;
        if constexpr (VIEW) walk(wb, b0, b1, 2);
        issue_own(own_b + 4 * HALF_BYTES, false, b0, b1);
      }
      return;
    }
    issue(0, 0, ka0, ka1);
    issue(0, 1, ka0, ka1);
    issue(0, 2, kb0, kb1);
    issue(0, 3, kb0, kb1);
    if (nt > 1) {
      int64_t kc0 = (int64_t)BK, kc1 = 0;
      if constexpr (VIEW) walk(wb, kc0, kc1, 1);
      issue(1, 2, kc0, kc1);
      issue(1, 3, kc0, kc1);
    } else if constexpr (VIEW) {
      int64_t d0, d1;
      walk(wb, d0, d1, 1);
    }
  };
  setup_tile((int)blockIdx.x);
  start_tile();

  // Stores of the previous tile's epilogue that may still be in flight at the top of a tile (p.epi_early).  VMEM operations
  // retire in issue order (what the compiler's own vmcnt bookkeeping relies on for global loads / stores on gfx9), and the
  // prologue's LDS-DMA pieces were issued BEFORE those stores: vmcnt(pend) = every prologue load has landed.  The first
  // K-tile's fragment reads and its first MFMA cluster then run while the store path (16 B/clk: ~3.8 us per 128-KiB tile,
  // profiles/r03_gemm_epilogue.md) drains; the loop's own vmcnt waits retire the stores with the loads queued behind them.
  int pend = 0;
  for (int tile = (int)blockIdx.x;;) {
  zero_acc();
  if (pend == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (pend == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's prologue loads (and the previous tile's stores)
  __syncthreads();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one interval behind group 0
  __builtin_amdgcn_sched_barrier(0);

  if constexpr (LEAN) {
    // the two-cluster schedule of the default loop (table in the kernel header), K-tile t in buffer B = t & 1, unrolled by two
    // k-major operand(s): the same schedule, fragment reads through the tracked transposing / plain readers, the pieces
    // of a k-major operand from two scalar bases (one per half of the K-tile), nothing interleaved
    auto body_km = [&](auto bufc, int t) {
      constexpr int B = decltype(bufc)::value;
      using IB = std::integral_constant<int, B>;
      using IO = std::integral_constant<int, B ^ 1>;
      const char* cur = smem + B * BUF_BYTES;
      const int left = nt - t;
      read_a(cur, 0);
      read_b(cur, 0);
      read_b(cur, 1);
      if (left > 1) {
        int64_t k0 = 0, k1 = 0;
        walk(wa, k0, k1, t + 1);
        stage_lean(IO{}, std::integral_constant<int, 0>{}, k0, k1);
        stage_lean(IO{}, std::integral_constant<int, 1>{}, k0, k1);
      }
      TNH_SEG_LOAD_END();
      mma_cluster(0, 0);
      TNH_SEG_MMA_END();
      read_a(cur, 1);
      if (left > 2) {
        int64_t k0 = 0, k1 = 0;
        walk(wb, k0, k1, t + 2);
        stage_lean(IB{}, std::integral_constant<int, 2>{}, k0, k1);
        stage_lean(IB{}, std::integral_constant<int, 3>{}, k0, k1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      TNH_SEG_LOAD_END();
      mma_cluster(1, 1);
      TNH_SEG_MMA_END();
    };
    auto body = [&](auto bufc, int t) {
      constexpr int B = decltype(bufc)::value;
      if constexpr (A_KM || (B_KN && !B_KT)) {
        body_km(bufc, t);
        return;
      }
      using IB = std::integral_constant<int, B>;
      using IO = std::integral_constant<int, B ^ 1>;
      using C0 = std::integral_constant<int, 0>;
      using C1 = std::integral_constant<int, 1>;
      using C2 = std::integral_constant<int, 2>;
      using C3 = std::integral_constant<int, 3>;
      const int left = nt - t;                 // K-tiles from this one on (scalar compares below, no lane masks)
      int64_t unused = 0;
      // ---- load segment 1: B sub-tiles 0 / 1 (8 reads), A sub-tile 0 (8 reads); A halves of K-tile t + 1 -> other buffer.
      // Every LDS-DMA piece is [M0 write] [one fragment read] [load]: the read is the wait state the M0 write needs.
      if constexpr (B_KT) {
        lean_read_b_tr(IB{}, C0{});
        lean_read_b_tr(IB{}, C1{});
      } else {
        lean_read_b(IB{}, C0{});
        lean_read_b(IB{}, C1{});
      }
      if (left > 1) {
        const void* sb;
        uint32_t v0 = oa[0][0], v1 = oa[0][1], v2 = oa[1][0], v3 = oa[1][1];
        if constexpr (!VIEW || KW == 2) {
          sb = (const void*)a_next;
          a_next += BK;
        } else {
          int64_t ka = 0, ka1 = 0;
          walk(wa, ka, ka1, t + 1);
          sb = (const void*)(abase + ka);
          if constexpr (KW == 0) {
            const uint32_t d = (uint32_t)((ka1 - ka - 32) * 2);     // wave-uniform; 0 unless the halves come from two runs
            if (d != 0) {
              const uint32_t e = himask & d;
              v0 += e; v1 += e; v2 += e; v3 += e;
            }
          }
        }
        piece_m0(IO{}, C0{}, C0{}); lean_read_a1(IB{}, C0{}, C0{}); piece_go(v0, sb); lean_read_a1(IB{}, C0{}, C1{});
        piece_m0(IO{}, C0{}, C1{}); lean_read_a1(IB{}, C0{}, C2{}); piece_go(v1, sb); lean_read_a1(IB{}, C0{}, C3{});
        piece_m0(IO{}, C1{}, C0{}); lean_read_a1(IB{}, C0{}, std::integral_constant<int, 4>{}); piece_go(v2, sb);
        lean_read_a1(IB{}, C0{}, std::integral_constant<int, 5>{});
        piece_m0(IO{}, C1{}, C1{}); lean_read_a1(IB{}, C0{}, std::integral_constant<int, 6>{}); piece_go(v3, sb);
        lean_read_a1(IB{}, C0{}, std::integral_constant<int, 7>{});
      } else {
        lean_read_a(IB{}, C0{});
      }
      TNH_SEG_LOAD_END();
      mma_cluster(0, 0);
      TNH_SEG_MMA_END();
      // ---- load segment 2: A sub-tile 1 (8 reads); B halves of K-tile t + 2 -> this buffer
      if (left > 2) {
        const void* sb;
        uint32_t v0 = ob[0][0], v1 = ob[0][1], v2 = ob[1][0], v3 = ob[1][1];
        if constexpr (!VIEW || KW == 2) {
          sb = (const void*)b_next;
          b_next += BK;
        } else {
          int64_t kb = 0, kb1 = 0;
          walk(wb, kb, kb1, t + 2);
          sb = (const void*)(bbase + kb);
          if constexpr (KW == 0) {
            const uint32_t d = (uint32_t)((kb1 - kb - 32) * 2);
            if (d != 0) {
              const uint32_t e = himask & d;
              v0 += e; v1 += e; v2 += e; v3 += e;
            }
          }
        }
        piece_m0(IB{}, C2{}, C0{}); lean_read_a1(IB{}, C1{}, C0{}); piece_go(v0, sb); lean_read_a1(IB{}, C1{}, C1{});
        piece_m0(IB{}, C2{}, C1{}); lean_read_a1(IB{}, C1{}, C2{}); piece_go(v1, sb); lean_read_a1(IB{}, C1{}, C3{});
        piece_m0(IB{}, C3{}, C0{}); lean_read_a1(IB{}, C1{}, std::integral_constant<int, 4>{}); piece_go(v2, sb);
        lean_read_a1(IB{}, C1{}, std::integral_constant<int, 5>{});
        piece_m0(IB{}, C3{}, C1{}); lean_read_a1(IB{}, C1{}, std::integral_constant<int, 6>{}); piece_go(v3, sb);
        lean_read_a1(IB{}, C1{}, std::integral_constant<int, 7>{});
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        lean_read_a(IB{}, C1{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      TNH_SEG_LOAD_END();
      mma_cluster(1, 1);
      TNH_SEG_MMA_END();
    };
    // (peeling the last two K-tiles off to get a condition-free body was tried: four copies of the body made the
    //  register allocator spill 300 bytes per lane into the loop -- not kept)
    for (int t = 0; t < nt; t += 2) {
      body(std::integral_constant<int, 0>{}, t);
      if (t + 1 < nt) body(std::integral_constant<int, 1>{}, t + 1);
    }
  } else {
  int b3 = 0;      // one-cluster schedule: K-tile % 3 (B slot)
  for (int t = 0; t < nt; ++t) {
    const int b = t & 1;
    const char* cur = smem + b * BUF_BYTES;
    const bool n1 = (t + 1 < nt), n2 = (t + 2 < nt);
    if constexpr (ONE) {
      // one 64-MFMA cluster per K-tile (schedule in the kernel header)
      const char* curA = smem + (t & 1) * 2 * HALF_BYTES;
      const char* curB = smem + 2 * HALF_BYTES + b3 * 2 * HALF_BYTES;     // read_b adds (2 + (wc >> 1)) half-tiles
      read_a(curA, 0);
      read_b(curB, 0);
      read_b(curB, 1);
      read_a(curA, 1);
      const bool ia = (t >= 1) && n1, ib = (t >= 1) && n2;
      if (ia) {
        int64_t ka = (int64_t)(t + 1) * BK, ka1 = 0;
        if constexpr (VIEW) walk(wa, ka, ka1, t + 1);
        issue_own(own_a + ((t + 1) & 1) * 2 * HALF_BYTES, true, ka, ka1);
      }
      if (ib) {
        int64_t kb = (int64_t)(t + 2) * BK, kb1 = 0;
        if constexpr (VIEW) walk(wb, kb, kb1, t + 2);
        const int slot = (b3 == 0) ? 2 : b3 - 1;       // (t + 2) % 3
        issue_own(own_b + slot * 2 * HALF_BYTES, false, kb, kb1);
      }
      if (wr == 1) {            // group 1's half of B of K-tile t + 1: group 0 reads it in the next interval
        if (ia && ib) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      TNH_SEG_LOAD_END();
      mma_quadrant(0, 0);
      mma_quadrant(0, 1);
      mma_quadrant(1, 1);
      mma_quadrant(1, 0);
      if (ib) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // A of K-tile t + 1 (and older pieces) landed
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      TNH_SEG_MMA_END();
      b3 = (b3 == 2) ? 0 : b3 + 1;
      continue;
    }
    if constexpr (TWO) {
      // two phases per K-tile, 32-MFMA clusters: half as many barriers per flop
      read_a(cur, 0);
      read_b(cur, 0);
      read_b(cur, 1);
      if (n1) {
        int64_t ka = (int64_t)(t + 1) * BK, ka1 = 0;
        if constexpr (VIEW) walk(wa, ka, ka1, t + 1);
        issue(b ^ 1, 0, ka, ka1);
        issue(b ^ 1, 1, ka, ka1);
      }
      TNH_SEG_LOAD_END();
      mma_quadrant(0, 0);
      mma_quadrant(0, 1);
      TNH_SEG_MMA_END();
      read_a(cur, 1);
      if (n2) {
        int64_t kb = (int64_t)(t + 2) * BK, kb1 = 0;
        if constexpr (VIEW) walk(wb, kb, kb1, t + 2);
        issue(b, 2, kb, kb1);
        issue(b, 3, kb, kb1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      TNH_SEG_LOAD_END();
      mma_quadrant(1, 1);
      mma_quadrant(1, 0);
      TNH_SEG_MMA_END();
      continue;
    }
    // ---- phase 0
    read_a(cur, 0);
    read_b(cur, 0);
    if (n1) issue(b ^ 1, 0, (int64_t)(t + 1) * BK, 0);
    TNH_SEG_LOAD_END();
    mma_quadrant(0, 0);
    TNH_SEG_MMA_END();
    // ---- phase 1
    read_b(cur, 1);
    if (n1) issue(b ^ 1, 1, (int64_t)(t + 1) * BK, 0);
    TNH_SEG_LOAD_END();
    mma_quadrant(0, 1);
    TNH_SEG_MMA_END();
    // ---- phase 2
    read_a(cur, 1);
    if (n2) issue(b, 2, (int64_t)(t + 2) * BK, 0);
    TNH_SEG_LOAD_END();
    mma_quadrant(1, 1);
    TNH_SEG_MMA_END();
    // ---- phase 3
    if (n2) {
      issue(b, 3, (int64_t)(t + 2) * BK, 0);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // tile t+1 landed; B halves of t+2 in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    TNH_SEG_LOAD_END();
    mma_quadrant(1, 0);
    TNH_SEG_MMA_END();
  }
  }   // (default / one-cluster loops)
  if (wr == 0) __builtin_amdgcn_s_barrier();  // pairs with group 1's last barrier
  __builtin_amdgcn_sched_barrier(0);

  // every wave is past its last LDS read of this tile: start the next tile's loads, then store this one
  const int64_t em0 = m0, en0 = n0;
  const int next = tile + (int)gridDim.x;
  const bool more = next < ntiles;
  if (more) {
    setup_tile(next);
    start_tile();
  }
  {
  const int64_t m0 = em0, n0 = en0;
  char* Cb = (char*)p.C + (int64_t)blockIdx.y * p.sC * (OUT_F32 ? 4 : 2);
  if constexpr (!M32) {
    store_wave_tile<IS_BF16, OUT_F32, 8, 4>(acc, p, Cb, m0, n0, BM, BN, wr * 128, wc * 64, lane);
    // how many store instructions that was, where the count is static (store_wave_tile: the 16-byte half path, the
    // float4 path of a full f32 tile); anything else is waited for in full
    const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.N);
    const bool vec = p.epi_early && p.c_vec && full;
    if constexpr (OUT_F32) pend = vec ? 32 : 0;
    else pend = (vec && (p.ldc & 7) == 0 && (((uintptr_t)Cb) & 15) == 0) ? 16 : 0;
  } else {
    // 32x32 C/D layout (operands swapped): lane holds, for m = lane & 31,
    // n = 8*q + 4*(lane >> 5) + (0..3) for q = 0..3  (registers 4q .. 4q+3).
    const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.N);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m0 + wr * 128 + i * 32 + (lane & 31);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t n = n0 + wc * 64 + j * 32 + 8 * q + 4 * (lane >> 5);
          const float v0 = acc32[i][j][4 * q], v1 = acc32[i][j][4 * q + 1], v2 = acc32[i][j][4 * q + 2],
                      v3 = acc32[i][j][4 * q + 3];
          if (full || (m < p.M && n + 3 < p.N)) {
            if constexpr (OUT_F32) {
              *(float4*)(Cb + (m * p.ldc + n) * 4) = make_float4(v0, v1, v2, v3);
            } else {
              uint2 o;
              o.x = pack2<IS_BF16>(v0, v1);
              o.y = pack2<IS_BF16>(v2, v3);
              *(uint2*)(Cb + (m * p.ldc + n) * 2) = o;
            }
          } else if (m < p.M) {
            const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (n + r < p.N) {
                if constexpr (OUT_F32) ((float*)Cb)[m * p.ldc + n + r] = vv[r];
                else ((uint16_t*)Cb)[m * p.ldc + n + r] = IS_BF16 ? f32_to_bf16(vv[r]) : f32_to_f16(vv[r]);
              }
            }
          }
        }
    }
  }
  }
  if (!more) break;
  tile = next;
  }
#undef TNH_SEG_LOAD_END
#undef TNH_SEG_MMA_END
}


// ---------------------------------------------------------------------------
// 256x256x64 "w4" kernel (A/B variant, knob ":p6", NOT the default): FOUR wavefronts (2 x 2),
// each owning a 128x128 output block = 64 accumulator tiles = all 256 AGPRs of the lane, one wave
// per SIMD.  Idea: 128x128 wave tiles read 4 x 256 rows x 128 B = 128 KiB of fragments per K-tile
// instead of the 192 KiB of the 8-wave kernels (128x64 wave tiles), and have no second wave to
// pay registers for.  Measured on MI355X (tools/w4_probe.py; bit-identical results to the
// ping-pong kernel): 8192^3 zero-filled 1.71 PF vs 1.95 PF, random 1.44-1.47 vs 1.48-1.53;
// 8192 x 8192 x 65536 (row stride 128 KiB) 1.17 PF vs 1.46-1.64 PF, identical for zero and random
// operands -- i.e. bound by the latency of the K-tile fetch, which one wave per SIMD exposes
// directly (a variant with four 32-deep stages and twice the prefetch distance was slower still:
// 1.05 PF, the extra barrier per k-step costs more than the distance buys).  The ping-pong kernel,
// whose partner wave keeps the matrix pipe busy across those waits, stays the default.
//
// One wave per SIMD means nobody else fills the matrix pipe while this wave waits,
// so the overlap is software pipelining inside the wave, scheduled by hand: the
// MFMAs and ds_reads are inline asm in the order they must issue (hipcc given the
// builtins spreads the 256 accumulators over VGPRs and AGPRs and shuttles them with
// ~200 v_accvgpr moves per K-tile; the "+a" constraint pins them).  Two fragment
// register sets X / Y: while the 64 MFMAs of one k-step run on X, the 16
// ds_read_b128 of the next k-step land in Y (1 read : 3 MFMA, last 16 MFMAs cover
// the latency of the final read).  The workgroup barrier that publishes the next
// K-tile sits between the two k-steps of the current one, where the second
// k-step's fragments are already in registers -- only the barrier skew is exposed.
//
//   prologue: DMA tile 0 -> buf0, tile 1 -> buf1; wait tile 0; barrier; X <- (0, ks0)
//   tile t:   Y <- (t, ks1)                      ||  MFMA X
//             wait tile t+1; barrier
//             X <- (t+1, ks0), DMA tile t+2 -> buf t&1 (its content is in X/Y)  ||  MFMA Y
//
// Waits are ours (the compiler does not see asm loads): lgkmcnt(0) after each
// batch, before the fragments it fetched are used; vmcnt(0) before the barrier.
// LDS: 2 stages x 64 KiB (dynamic), same XOR-swizzled image as the kernels above.
// Needs M % 256 == 0 and N % 256 == 0 (no edge handling: every row/col is real).
// ---------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;   // asm operands must be plain vectors


template <int OFF>
__device__ __forceinline__ void lds_read16_asm(u32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF));
}

template <bool IS_BF16>
__device__ __forceinline__ void mma16_pinned(f32x4& c, const u32x4& a, const u32x4& b) {
  if constexpr (IS_BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <bool IS_BF16, bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_nt_w4_kernel(NtArgs p) {
  constexpr int BM = 256, BN = 256, BK = 64;
  constexpr int A_BYTES = BM * BK * 2, STAGE_BYTES = 2 * A_BYTES;
  extern __shared__ __attribute__((aligned(1024))) char smem_w4[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;

  int tm, tn;
  tile_of_block(blockIdx.x, p.tiles_m, p.tiles_n, p.raster, tm, tn);
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const uint16_t* A = p.A + (int64_t)blockIdx.y * p.sA;
  const uint16_t* B = p.B + (int64_t)blockIdx.y * p.sB;

  // LDS-DMA piece (i, wave): tile rows (4 i + wid) * 8 .. +7; lane l -> row + (l >> 3), source chunk
  // (l & 7) ^ (l >> 3) (the swizzle, see gemm_nt_kernel).  Rows step by 32 between pieces.
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ lrow;
  const uint16_t* ga = A + (m0 + wid * 8 + lrow) * p.lda + lchunk * 8;
  const uint16_t* gb = B + (n0 + wid * 8 + lrow) * p.ldb + lchunk * 8;
  const int64_t step_a = 32 * p.lda, step_b = 32 * p.ldb;
  const unsigned lds0 = (unsigned)(size_t)TNH_LDS_PTR(smem_w4);

  // piece q of a stage: q < 8 -> A rows, else B rows
  auto dma_piece = [&](int q, int s, int64_t k0) {
    const unsigned dst = lds0 + s * STAGE_BYTES + wid * 1024 + (q & 7) * 4096 + (q >> 3) * A_BYTES;
    const uint16_t* src = (q < 8) ? ga + (q & 7) * step_a : gb + (q & 7) * step_b;
    glds16(src + k0, __builtin_amdgcn_readfirstlane(dst));
  };

  // fragment read addresses (LDS bytes, stage 0): row 16 f + (l & 15), swizzled chunk per k-step
  unsigned fa_addr[2], fb_addr[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const unsigned off = (lane & 15) * 128 + (((ks * 4 + (lane >> 4)) ^ (lane & 7)) * 16);
    fa_addr[ks] = lds0 + wm * 16384 + off;
    fb_addr[ks] = lds0 + A_BYTES + wn * 16384 + off;
  }

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 xa[8], xb[8], ya[8], yb[8];
  // fragment #g of a set: g < 8 -> B rows (needed by every MFMA of the first A row), else A rows
  auto read_frag = [&](auto g, u32x4 (&fa)[8], u32x4 (&fb)[8], unsigned aaddr, unsigned baddr) {
    constexpr int G = decltype(g)::value;
    if constexpr (G < 8) lds_read16_asm<G * 2048>(fb[G], baddr);
    else lds_read16_asm<(G - 8) * 2048>(fa[G - 8], aaddr);
  };
  auto mma4 = [&](auto q, const u32x4 (&fa)[8], const u32x4 (&fb)[8]) {   // MFMA #q of 64: acc[q / 8][q % 8]
    constexpr int Q = decltype(q)::value;
    mma16_pinned<IS_BF16>(acc[Q / 8][Q % 8], fb[Q % 8], fa[Q / 8]);
  };

  const int nt = (int)(p.K / BK);
#pragma unroll
  for (int q = 0; q < 16; ++q) dma_piece(q, 0, 0);
#pragma unroll
  for (int q = 0; q < 16; ++q) dma_piece(q, 1, (int64_t)(nt > 1 ? 1 : 0) * BK);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // tile 0 landed; the 16 pieces of tile 1 stay in flight
  __syncthreads();
  static_for<0, 16>([&](auto g) { read_frag(g, xa, xb, fa_addr[0], fb_addr[0]); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  for (int t = 0; t + 1 < nt; ++t) {
    const unsigned cur = (t & 1) * STAGE_BYTES, nxt = STAGE_BYTES - cur;
    // ---- k-step 0 of tile t on X, fetching k-step 1 into Y
    static_for<0, 16>([&](auto g) {
      constexpr int G = decltype(g)::value;
      read_frag(g, ya, yb, fa_addr[1] + cur, fb_addr[1] + cur);
      static_for<3 * G, 3 * G + 3>([&](auto q) { mma4(q, xa, xb); });
    });
    static_for<48, 64>([&](auto q) { mma4(q, xa, xb); });
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // tile t+1 landed; Y complete; no reads of `cur` pending
    __syncthreads();
    // ---- k-step 1 of tile t on Y, fetching (t+1, ks0) into X and issuing the DMA of tile t+2 into `cur`
    const int64_t k2 = (int64_t)((t + 2 < nt) ? t + 2 : nt - 1) * BK;   // tail: harmless re-load keeps the body branch-free
    static_for<0, 16>([&](auto g) {
      constexpr int G = decltype(g)::value;
      read_frag(g, xa, xb, fa_addr[0] + nxt, fb_addr[0] + nxt);
      dma_piece(G, t & 1, k2);
      static_for<3 * G, 3 * G + 3>([&](auto q) { mma4(q, ya, yb); });
    });
    static_for<48, 64>([&](auto q) { mma4(q, ya, yb); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  {
    const unsigned cur = ((nt - 1) & 1) * STAGE_BYTES;
    static_for<0, 16>([&](auto g) {
      constexpr int G = decltype(g)::value;
      read_frag(g, ya, yb, fa_addr[1] + cur, fb_addr[1] + cur);
      static_for<3 * G, 3 * G + 3>([&](auto q) { mma4(q, xa, xb); });
    });
    static_for<48, 64>([&](auto q) { mma4(q, xa, xb); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    static_for<0, 64>([&](auto q) { mma4(q, ya, yb); });
  }
  // MFMA results -> VALU reads: the compiler cannot see the asm MFMAs, so the wait states are ours;
  // and no DMA may outlive the workgroup's LDS allocation.
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");

  store_wave_tile<IS_BF16, OUT_F32, 8, 8>(acc, p, (char*)p.C + (int64_t)blockIdx.y * p.sC * (OUT_F32 ? 4 : 2), m0, n0,
                                          BM, BN, wm * 128, wn * 128, lane);
}

static int launch_w4(bool is_bf16, bool out_f32, NtArgs p, int64_t batch) {
  constexpr int LDS_BYTES = 2 * 2 * 256 * 64 * 2;   // 128 KiB of the CU's 160 KiB
  p.tiles_m = (int)(p.M / 256);
  p.tiles_n = (int)(p.N / 256);
  const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  TNH_REQUIRE(nwg < (int64_t(1) << 24), "GEMM grid too large");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipSuccess;
#define TNH_W4_ATTR(B16, O32)                                                                                   \
  if (e == hipSuccess)                                                                                          \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_w4_kernel<B16, O32>),                         \
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)
    TNH_W4_ATTR(true, true); TNH_W4_ATTR(true, false); TNH_W4_ATTR(false, true); TNH_W4_ATTR(false, false);
#undef TNH_W4_ATTR
    TNH_REQUIRE(e == hipSuccess, "hipFuncSetAttribute(128 KiB LDS) failed: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const int esz_out = out_f32 ? 4 : 2;
  for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
    const int64_t nb = (batch - b0 < 65535) ? (batch - b0) : 65535;
    NtArgs q = p;
    q.A = p.A + b0 * p.sA;
    q.B = p.B + b0 * p.sB;
    q.C = (char*)p.C + b0 * p.sC * esz_out;
    const dim3 grid((unsigned)nwg, (unsigned)nb), block(256);
#define TNH_W4_LAUNCH(B16, O32)                                                                                  \
  do {                                                                                                          \
    hipLaunchKernelGGL((gemm_nt_w4_kernel<B16, O32>), grid, block, LDS_BYTES, stream(), q);                      \
  } while (0)
    if (is_bf16) {
      if (out_f32) TNH_W4_LAUNCH(true, true);
      else TNH_W4_LAUNCH(true, false);
    } else {
      if (out_f32) TNH_W4_LAUNCH(false, true);
      else TNH_W4_LAUNCH(false, false);
    }
#undef TNH_W4_LAUNCH
    TNH_LAUNCH_CHECK();
  }
  return TNH_OK;
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_nt(bool is_bf16, bool out_f32, NtArgs p, int64_t batch) {
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
    const dim3 grid((unsigned)nwg, (unsigned)nb), block(WAVES_M * WAVES_N * 64);
    if (is_bf16) {
      if (out_f32) hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WAVES_M, WAVES_N, true, true>), grid, block, 0, stream(), q);
      else hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WAVES_M, WAVES_N, true, false>), grid, block, 0, stream(), q);
    } else {
      if (out_f32) hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WAVES_M, WAVES_N, false, true>), grid, block, 0, stream(), q);
      else hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WAVES_M, WAVES_N, false, false>), grid, block, 0, stream(), q);
    }
    TNH_LAUNCH_CHECK();
  }
  return TNH_OK;
}

int g_opt_nt = -1;       // A/B knob ":n<d>": non-temporal epilogue stores for results of 64 MiB and more (-1 / 1: on; 0: off)
int g_opt_epi = -1;      // A/B knob ":e<d>": next tile's MFMAs under the draining epilogue stores (-1 / 1: on; 0: off)
int g_opt_lean = -1;     // A/B knob ":l<d>": the lean main loop of the ping-pong kernels (-1: on where it applies)

// The lean loop addresses a tile's rows as a 32-bit byte offset from the tile's first row
static bool lean_wanted(const NtArgs& q) {
  static const int env = []() { const char* e = getenv("TNH_GEMM_LEAN"); return e ? atoi(e) : -1; }();
  const int mode = g_opt_lean >= 0 ? g_opt_lean : env;
  if (mode == 0) return false;
  const int64_t span_a = 255 * q.lda + 64, span_b = 255 * q.ldb + 64;
  return span_a > 0 && span_b > 0 && span_a * 2 < (int64_t(1) << 32) && span_b * 2 < (int64_t(1) << 32);
}

// View operands (K-contiguous): can the lean loop address the rows?  Sets q.lean_rel_a / _b.
//   1  rows ascend in memory (sr0 > 0 and, with two levels, sr1 >= the extent of an inner run of rows) and any 256
//      consecutive rows plus a K-tile span < 4 GiB: offsets from the tile's first row;
//   0  otherwise, if both strides are non-negative and the whole operand spans < 4 GiB: offsets from the operand's base.
static bool lean_view_rows(NtArgs& q, bool half_walk = false, bool a_km = false, bool b_kn = false, bool km_tile = false) {
  static const int env = []() { const char* e = getenv("TNH_GEMM_LEAN"); return e ? atoi(e) : -1; }();
  const int mode = g_opt_lean >= 0 ? g_opt_lean : env;
  if (mode == 0) return false;
  const int64_t lim = int64_t(1) << 31;      // elements (2 bytes each)
  auto one = [&](const OpView& v, int64_t rows, int& rel, bool kmajor) -> bool {
    if (v.sr0 <= 0) return false;
    // K-contiguous operand, half-K-tile walk: the lanes of a K-tile's second half add the distance between two
    // contraction runs; k-major operand: a lane's k row inside a half K-tile (each half has its own scalar base)
    const bool one_run = (int64_t)v.tpi * 32 >= q.K;          // (sk1 is meaningless then: the walk never wraps)
    // (km_tile: the whole-K-tile walk of a k-major B -- a lane's k row 0 .. 63 in its offset)
    const int64_t jump = kmajor ? (km_tile ? 64 : 32) * v.sk0 : ((half_walk && !one_run) ? (v.sk1 - (int64_t)v.tpi * 32) : 0);
    if (jump < 0 || jump >= lim / 2) return false;
    const bool single = v.r0 >= rows;
    const int64_t inner = (v.r0 - 1) * v.sr0;
    if (mode != 2 && (single || v.sr1 > inner)) {           // ascending rows (knob ":l2" prefers the operand-base form: tests)
      const int64_t steps = single ? 0 : (255 / v.r0 + 2);
      const int64_t span = (single ? 255 * v.sr0 : inner + steps * v.sr1) + 64 + jump;
      if (span < lim) { rel = 1; return true; }
    }
    if (!single && v.sr1 < 0) return false;
    const int64_t whole = (single ? (rows - 1) * v.sr0 : inner + ((rows - 1) / v.r0) * v.sr1) + 64 + jump;
    if (whole < lim) { rel = 0; return true; }
    return false;
  };
  return one(q.va, q.M + q.m_off, q.lean_rel_a, a_km) && one(q.vb, q.N, q.lean_rel_b, b_kn);
}

static int launch_pp(bool is_bf16, bool out_f32, bool two, NtArgs p, int64_t batch, bool m32 = false) {
  p.tiles_m = (int)((p.M + 255) / 256);
  p.tiles_n = (int)((p.N + 255) / 256);
  const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  TNH_REQUIRE(nwg < (int64_t(1) << 24), "GEMM grid too large");
  const int esz_out = out_f32 ? 4 : 2;
  for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
    const int64_t nb = (batch - b0 < 65535) ? (batch - b0) : 65535;
    NtArgs q = p;
    q.A = p.A + b0 * p.sA;
    q.B = p.B + b0 * p.sB;
    q.C = (char*)p.C + b0 * p.sC * esz_out;
    q.m_off = 0;
    q.lean_rel_a = q.lean_rel_b = 1;
    const dim3 grid(pp_grid_x(nwg, (unsigned)nb), (unsigned)nb), block(512);
    const bool lean = lean_wanted(q) && two && !m32 && g_opt_phases != 8;
#define TNH_PP_LAUNCH(B16, O32)                                                                               \
  do {                                                                                                       \
    if (g_opt_phases == 7) hipLaunchKernelGGL((gemm_nt_pp_kernel<B16, O32, true, false, false, false, false, 1>), grid, block, 0, stream(), q); \
    else if (lean) hipLaunchKernelGGL((gemm_nt_pp_kernel<B16, O32, true, false, false, false, false, 3>), grid, block, 0, stream(), q); \
    else if (g_opt_phases == 8) hipLaunchKernelGGL((gemm_nt_pp_kernel<B16, O32, true, false, false, false, false, 2>), grid, block, 0, stream(), q); \
    else if (m32) hipLaunchKernelGGL((gemm_nt_pp_kernel<B16, O32, true, true>), grid, block, 0, stream(), q);      \
    else if (two) hipLaunchKernelGGL((gemm_nt_pp_kernel<B16, O32, true>), grid, block, 0, stream(), q);      \
    else hipLaunchKernelGGL((gemm_nt_pp_kernel<B16, O32, false>), grid, block, 0, stream(), q);              \
  } while (0)
    if (is_bf16) {
      if (out_f32) TNH_PP_LAUNCH(true, true);
      else TNH_PP_LAUNCH(true, false);
    } else {
      if (out_f32) TNH_PP_LAUNCH(false, true);
      else TNH_PP_LAUNCH(false, false);
    }
#undef TNH_PP_LAUNCH
    TNH_LAUNCH_CHECK();
  }
  return TNH_OK;
}

// ---- view GEMM: operands read in place through two-level strides (tnh_gemm_view) ----------------
int g_opt_kwalk = -1;    // A/B knob ":w<d>": cap on the K-walk form of the view kernel (-1: the cheapest the operands allow)

template <bool A_KM, bool B_KN>
static void launch_pp_view_t(bool is_bf16, bool out_f32, dim3 grid, const NtArgs& q, int kw = 0) {
  const dim3 block(512);
  if constexpr (!A_KM && !B_KN) {
    if (g_opt_phases != 7) {      // K-contiguous operands: the cheapest K walk they allow, in the lean loop where it applies
      // the lean loop: rows addressable as 32-bit offsets from a wave-uniform base (lean_view_rows)
      NtArgs ql = q;
      const bool lean = lean_view_rows(ql, kw == 0);
#define TNH_VIEW_KW(B16, O32)                                                                                              \
  do {                                                                                                                     \
    if (lean && kw == 2) hipLaunchKernelGGL((gemm_nt_pp_kernel<B16, O32, true, false, true, false, false, 3, 2>), grid, block, 0, stream(), ql); \
    else if (lean && kw == 1) hipLaunchKernelGGL((gemm_nt_pp_kernel<B16, O32, true, false, true, false, false, 3, 1>), grid, block, 0, stream(), ql); \
    else if (lean) hipLaunchKernelGGL((gemm_nt_pp_kernel<B16, O32, true, false, true, false, false, 3, 0>), grid, block, 0, stream(), ql); \
    else if (kw == 0) hipLaunchKernelGGL((gemm_nt_pp_kernel<B16, O32, true, false, true, false, false, 0, 0>), grid, block, 0, stream(), q); \
    else if (kw == 2) hipLaunchKernelGGL((gemm_nt_pp_kernel<B16, O32, true, false, true, false, false, 0, 2>), grid, block, 0, stream(), q); \
    else hipLaunchKernelGGL((gemm_nt_pp_kernel<B16, O32, true, false, true, false, false, 0, 1>), grid, block, 0, stream(), q);         \
  } while (0)
      if (is_bf16) {
        if (out_f32) TNH_VIEW_KW(true, true);
        else TNH_VIEW_KW(true, false);
      } else {
        if (out_f32) TNH_VIEW_KW(false, true);
        else TNH_VIEW_KW(false, false);
      }
#undef TNH_VIEW_KW
      return;
    }
    if (g_opt_phases == 7) {      // A/B knob ":p7": one 64-MFMA cluster per K-tile
      if (is_bf16) {
        if (out_f32) hipLaunchKernelGGL((gemm_nt_pp_kernel<true, true, true, false, true, false, false, 1>), grid, block, 0, stream(), q);
        else hipLaunchKernelGGL((gemm_nt_pp_kernel<true, false, true, false, true, false, false, 1>), grid, block, 0, stream(), q);
      } else {
        if (out_f32) hipLaunchKernelGGL((gemm_nt_pp_kernel<false, true, true, false, true, false, false, 1>), grid, block, 0, stream(), q);
        else hipLaunchKernelGGL((gemm_nt_pp_kernel<false, false, true, false, true, false, false, 1>), grid, block, 0, stream(), q);
      }
      return;
    }
  }
  if constexpr (!A_KM && B_KN) {       // k-major B with whole-K-tile runs next to a K-contiguous A: the interleaved lean loop
    NtArgs ql = q;
    if (kw >= 1 && g_opt_phases != 7 && lean_view_rows(ql, false, false, true, true)) {
      if (is_bf16) {
        if (out_f32) hipLaunchKernelGGL((gemm_nt_pp_kernel<true, true, true, false, true, false, true, 3, 1>), grid, block, 0, stream(), ql);
        else hipLaunchKernelGGL((gemm_nt_pp_kernel<true, false, true, false, true, false, true, 3, 1>), grid, block, 0, stream(), ql);
      } else {
        if (out_f32) hipLaunchKernelGGL((gemm_nt_pp_kernel<false, true, true, false, true, false, true, 3, 1>), grid, block, 0, stream(), ql);
        else hipLaunchKernelGGL((gemm_nt_pp_kernel<false, false, true, false, true, false, true, 3, 1>), grid, block, 0, stream(), ql);
      }
      return;
    }
  }
  if constexpr (A_KM || B_KN) {
    NtArgs ql = q;
    if (g_opt_phases != 7 && lean_view_rows(ql, true, A_KM, B_KN)) {
      if (is_bf16) {
        if (out_f32) hipLaunchKernelGGL((gemm_nt_pp_kernel<true, true, true, false, true, A_KM, B_KN, 3, 0>), grid, block, 0, stream(), ql);
        else hipLaunchKernelGGL((gemm_nt_pp_kernel<true, false, true, false, true, A_KM, B_KN, 3, 0>), grid, block, 0, stream(), ql);
      } else {
        if (out_f32) hipLaunchKernelGGL((gemm_nt_pp_kernel<false, true, true, false, true, A_KM, B_KN, 3, 0>), grid, block, 0, stream(), ql);
        else hipLaunchKernelGGL((gemm_nt_pp_kernel<false, false, true, false, true, A_KM, B_KN, 3, 0>), grid, block, 0, stream(), ql);
      }
      return;
    }
  }
  if (is_bf16) {
    if (out_f32) hipLaunchKernelGGL((gemm_nt_pp_kernel<true, true, true, false, true, A_KM, B_KN>), grid, block, 0, stream(), q);
    else hipLaunchKernelGGL((gemm_nt_pp_kernel<true, false, true, false, true, A_KM, B_KN>), grid, block, 0, stream(), q);
  } else {
    if (out_f32) hipLaunchKernelGGL((gemm_nt_pp_kernel<false, true, true, false, true, A_KM, B_KN>), grid, block, 0, stream(), q);
    else hipLaunchKernelGGL((gemm_nt_pp_kernel<false, false, true, false, true, A_KM, B_KN>), grid, block, 0, stream(), q);
  }
}

static bool view_ok(const OpView& v, int64_t rows, int64_t K, const void* base, const char* which) {
  const bool kcontig = (v.sk0 == 1), kmajor = (v.sr0 == 1 && v.sk0 != 1);
  if (!kcontig && !kmajor) {
    set_error("tnh_gemm_view: operand %s has no contiguous direction (sk0 %lld, sr0 %lld)", which, (long long)v.sk0,
              (long long)v.sr0);
    return false;
  }
  const int64_t k0 = (int64_t)v.tpi * 32;
  bool ok = v.tpi >= 1 && v.r0 >= 1 && K % k0 == 0 && ((uintptr_t)base % 16) == 0 && v.sr1 % 8 == 0 && v.sk1 % 8 == 0;
  if (kcontig) ok = ok && v.sr0 % 8 == 0;
  else ok = ok && v.sk0 % 8 == 0 && v.r0 % 8 == 0 && rows % 8 == 0;
  if (!ok) set_error("tnh_gemm_view: operand %s breaks the 16-byte / 64-deep alignment rules of the LDS-DMA loaders", which);
  return ok;
}

int gemm_bf16_view(int in_dt, int out_dt, int64_t M, int64_t N, int64_t K, const void* A, const OpView& va,
                   const void* B, const OpView& vb, void* C, int64_t ldc, const char** name) {
  const bool big = (M >= 256 && N >= 256) && (((M + 255) / 256) * ((N + 255) / 256) >= 192);
  if (!big || K % 64 != 0 || K < 128 || ldc % 4 != 0 || ((uintptr_t)C % 16) != 0 || M >= (int64_t(1) << 31) ||
      N >= (int64_t(1) << 31)) {
    set_error("tnh_gemm_view: shape outside the 256x256 ping-pong kernel's range");
    return TNH_ERR_UNSUPPORTED;
  }
  if (!view_ok(va, M, K, A, "A") || !view_ok(vb, N, K, B, "B")) return TNH_ERR_UNSUPPORTED;
  NtArgs p;
  p.va = va;
  p.vb = vb;
  if (p.va.r0 > M) p.va.r0 = M;   // single-level rows: keep the 32-bit row split in range
  if (p.vb.r0 > N) p.vb.r0 = N;
  p.A = (const uint16_t*)A;
  p.B = (const uint16_t*)B;
  p.C = C;
  p.M = M; p.N = N; p.K = K;
  p.lda = va.sr0; p.ldb = vb.sr0; p.ldc = ldc;
  p.sA = p.sB = p.sC = 0;
  p.raster = pick_raster(M, N, K);
  p.epi_early = g_opt_epi != 0;
  p.c_vec = ((int64_t)M * N * 2 >= (int64_t(64) << 20) && g_opt_nt != 0) ? 3 : 1;
  p.a_vw = p.b_vw = 8;
  p.tiles_m = (int)((M + 255) / 256);
  p.tiles_n = (int)((N + 255) / 256);
  const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  TNH_REQUIRE(nwg < (int64_t(1) << 24), "GEMM grid too large");
  const bool a_km = (va.sk0 != 1), b_kn = (vb.sk0 != 1);
  const bool is_bf16 = (in_dt == TNH_BF16), out_f32 = (out_dt == TNH_F32);
  p.kslice_tiles = 0;
  p.m_off = 0;
  // The cheapest K walk both operands allow (kernel header, KW): 2 = one contiguous contraction run each (a run that
  // continues seamlessly into the next counts), 1 = runs that are multiples of 64, 0 = half-K-tile walk.
  int kw = 0;
  if (!a_km && !b_kn) {
    auto contiguous = [&](OpView& v) {
      const bool yes = (int64_t)v.tpi * 32 >= K || v.sk1 == (int64_t)v.tpi * 32;
      if (yes) {                       // normalise to ONE run (tile-granular walks read tpi / sk1)
        v.tpi = (int)(K / 32);
        v.sk1 = K;
      }
      return yes;
    };
    const bool ca = contiguous(p.va), cb = contiguous(p.vb);
    kw = (ca && cb) ? 2 : ((p.va.tpi % 2 == 0 && p.vb.tpi % 2 == 0) ? 1 : 0);
    if (g_opt_kwalk >= 0 && kw > g_opt_kwalk) kw = g_opt_kwalk;
  } else if (!a_km && b_kn) {
    // k-major B: the whole-K-tile walk (kernel header, B_KT) when every contraction run of both operands is a
    // multiple of 64; a single run is normalised as above (k-major: the run's stride is sk0)
    auto one_run = [&](OpView& v, int64_t ks) {
      if ((int64_t)v.tpi * 32 >= K || v.sk1 == (int64_t)v.tpi * 32 * ks) {
        v.tpi = (int)(K / 32);
        v.sk1 = K * ks;
      }
    };
    one_run(p.va, 1);
    one_run(p.vb, p.vb.sk0);
    kw = (p.va.tpi % 2 == 0 && p.vb.tpi % 2 == 0) ? 1 : 0;
    if (g_opt_kwalk >= 0 && kw > g_opt_kwalk) kw = g_opt_kwalk;
  }
  auto launch = [&](const NtArgs& q, unsigned gy, bool f32_out) {
    const dim3 grid(pp_grid_x((int64_t)q.tiles_m * q.tiles_n, gy), gy);
    if (a_km && b_kn) launch_pp_view_t<true, true>(is_bf16, f32_out, grid, q);
    else if (a_km) launch_pp_view_t<true, false>(is_bf16, f32_out, grid, q);
    else if (b_kn) launch_pp_view_t<false, true>(is_bf16, f32_out, grid, q, kw);
    else launch_pp_view_t<false, false>(is_bf16, f32_out, grid, q, kw);
  };
  *name = a_km ? (b_kn ? "bf16_view_tt_256x256x64_pp" : "bf16_view_tn_256x256x64_pp")
               : (b_kn ? "bf16_view_nn_256x256x64_pp" : "bf16_view_nt_256x256x64_pp");

  // ---- tail split.  Every CU runs one 256 x 256 tile at a time, so a launch takes ceil(tiles / CUs) tile times:
  // 1296 tiles (D = 96: 36 x 36) = 5.06 "waves" pay for 6.  When the last wave is mostly empty, the last tile rows are
  // cut off and computed by a SPLIT-K launch of the same kernel (blockIdx.y = K-slice, f32 partial slabs, summed in a
  // fixed order by K4): 35 x 36 tiles in 5 waves + 36 tiles x 7 slices in 1/7 of a tile time.
  static const bool tail_env = []() { const char* e = getenv("TNH_GEMM_TAIL_SPLIT"); return !(e && e[0] == '0'); }();
  const bool tail_on = tail_env && g_opt_tail != 0;     // knob ":t0" of tnh_gemm_set_variant (tests, A/B)
  const int cus = num_cus() > 0 ? num_cus() : 256;
  const int nkt = (int)(K / 64);
  // (measured, tools/tail_probe.py: 9216^3 +5.6 %, 9216 x 9216 x 2048 -3.7 % -- the partial-slab reduction does not
  // shrink with K -- hence only for K >= 6144)
  if (tail_on && nwg > cus && ldc == N && nkt >= 96) {
    const double base = (double)((nwg + cus - 1) / cus);
    double best = base;
    int best_rt = 0, best_ks = 0, best_s = 0;
    // (up to 32 tile rows: 81 x 7 tiles -- the 20736 x 1728 x 20736 product of the D = 12 network -- are 2.2 waves, and
    //  cutting 8 rows leaves 511 tiles = 2 waves + 56 tiles x 4 K-slices: 3 tile times -> 2.35)
    for (int rt = 1; rt <= 32 && rt < p.tiles_m; ++rt) {
      const int64_t main_tiles = (int64_t)(p.tiles_m - rt) * p.tiles_n, tail_tiles = (int64_t)rt * p.tiles_n;
      int s = (int)(cus / tail_tiles);
      if (s < 2) continue;
      if (s > 16) s = 16;
      const int ks = (nkt + s - 1) / s;
      if (ks < 8) continue;
      const int s_eff = (nkt + ks - 1) / ks;
      const double cost = (double)((main_tiles + cus - 1) / cus) +
                          (double)((tail_tiles * s_eff + cus - 1) / cus) * ks / nkt + 0.10;   // + reduction passes
      if (cost < best) { best = cost; best_rt = rt; best_ks = ks; best_s = s_eff; }
    }
    if (best_rt > 0 && best <= 0.94 * base) {
      const int64_t m_main = (int64_t)(p.tiles_m - best_rt) * 256, m_tail = M - m_main;
      void* W = nullptr;
      int rc = tnh_malloc(&W, (size_t)best_s * m_tail * N * 4);
      if (rc == TNH_OK) {
        NtArgs q = p;                     // main part: rows [0, m_main)
        q.M = m_main;
        q.tiles_m = p.tiles_m - best_rt;
        launch(q, 1, out_f32);
        NtArgs t = p;                     // tail rows [m_main, M) of A (two-level rows included: the row index
        t.m_off = m_main;                 // keeps its origin), split over K
        t.M = m_tail;
        t.tiles_m = (int)((m_tail + 255) / 256);
        t.C = W;
        t.ldc = N;
        t.sC = m_tail * N;
        t.kslice_tiles = best_ks;
        launch(t, (unsigned)best_s, true);
        TNH_LAUNCH_CHECK();
        char* c_tail = (char*)C + (size_t)m_main * N * (out_f32 ? 4 : 2);
        if (out_f32) {
          rc = tnh_sum_mid(c_tail, W, 1, best_s, m_tail * N, TNH_F32);
        } else {
          void* T32 = nullptr;
          rc = tnh_malloc(&T32, (size_t)m_tail * N * 4);
          if (!rc) rc = tnh_sum_mid(T32, W, 1, best_s, m_tail * N, TNH_F32);
          if (!rc) rc = tnh_cast(c_tail, out_dt, T32, TNH_F32, m_tail * N);
          if (T32) tnh_free(T32);
        }
        tnh_free(W);
        *name = a_km ? (b_kn ? "bf16_view_tt_256x256x64_pp+tail_splitk" : "bf16_view_tn_256x256x64_pp+tail_splitk")
                     : (b_kn ? "bf16_view_nn_256x256x64_pp+tail_splitk" : "bf16_view_nt_256x256x64_pp+tail_splitk");
        return rc;
      }
      (void)hipGetLastError();            // no room for the partial slabs: plain launch below
    }
  }
  launch(p, 1, out_f32);
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

// Returns TNH_ERR_UNSUPPORTED (without setting an error the caller must
// surface) when the layout is not NT; the dispatcher then falls back to the
// general (strided) MFMA kernel.
int gemm_bf16_ragged(int in_dt, int out_dt, int shape, int64_t M, int64_t N, int64_t K, const void* A,
                     int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t batch,
                     int64_t sA, int64_t sB, int64_t sC, const char** name);

// K <= 16, large results (tnh_gemm_smallk.hip)
bool gemm_bf16_smallk_wanted(int out_dt, int64_t M, int64_t N, int64_t K, int64_t batch, const void* A, int64_t lda,
                             const void* B, int64_t ldb, const void* C, int64_t ldc);
int gemm_bf16_smallk(int in_dt, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                     void* C, int64_t ldc, const char** name);

// small x very long products (tnh_gemm_stream.hip)
bool gemm_bf16_stream_wanted(int out_dt, int64_t M, int64_t N, int64_t K, int64_t batch);
int gemm_bf16_stream(int in_dt, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                     int64_t ldb, void* C, int64_t ldc, const char** name);

int gemm_bf16_fast(int in_dt, int out_dt, int variant, int transA, int transB, int64_t M, int64_t N,
                   int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
                   int64_t ldc, int64_t batch, int64_t sA, int64_t sB, int64_t sC, const char** name) {
  if (!(transA == 0 && transB == 1)) {
    set_error("bf16 matrix-core path needs the NT layout (both operands K-contiguous)");
    return TNH_ERR_UNSUPPORTED;
  }
  if (variant == 12) {   // "bf16_stream" forced (tests, A/B)
    if (batch != 1 || out_dt == TNH_F32 || K > 192 || (M > 192 && N > 192)) {
      set_error("bf16_stream needs batch 1, half output, K <= 192 and one side <= 192");
      return TNH_ERR_UNSUPPORTED;
    }
    return gemm_bf16_stream(in_dt, M, N, K, A, lda, B, ldb, C, ldc, name);
  }
  if (variant == 0 && gemm_bf16_stream_wanted(out_dt, M, N, K, batch)) {
    const int rc = gemm_bf16_stream(in_dt, M, N, K, A, lda, B, ldb, C, ldc, name);
    if (rc != TNH_ERR_UNSUPPORTED) return rc;   // odd alignment: the tile kernels below
  }
  // K <= 16 with a large result: a store stream (tnh_gemm_smallk.hip)
  if (variant == 0 && gemm_bf16_smallk_wanted(out_dt, M, N, K, batch, A, lda, B, ldb, C, ldc))
    return gemm_bf16_smallk(in_dt, M, N, K, A, lda, B, ldb, C, ldc, name);
  bool dma_ok = K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 &&
                ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0 &&
                sA % 8 == 0 && sB % 8 == 0 && sC % 4 == 0 && M >= 16 && N >= 16;
  // a short side of <= 64 against a very long one, K <= 192: the 64 x 256 / 256 x 64 small-K tile kernel streams it
  // better than 128 x 128 LDS-DMA tiles that are half padding (64 x 4e6 x 64: 0.20 vs 0.29 ms)
  if (variant == 0 && K <= 192 && (M < N ? M : N) <= 64 && (M < N ? N : M) >= (int64_t(1) << 16)) dma_ok = false;
  if (variant >= 6 || !dma_ok) {
    if (variant >= 3 && variant <= 5) {
      set_error("LDS-DMA bf16 kernels need K%%64==0 and 16-B aligned rows");
      return TNH_ERR_UNSUPPORTED;
    }
    // ragged shapes / odd alignment: register-staged kernel (tnh_gemm_ragged.hip)
    return gemm_bf16_ragged(in_dt, out_dt, variant >= 6 ? variant - 6 : 0, M, N, K, A, lda, B, ldb, C, ldc,
                            batch, sA, sB, sC, name);
  }
  NtArgs p;
  p.A = (const uint16_t*)A;
  p.B = (const uint16_t*)B;
  p.C = C;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.sA = sA; p.sB = sB; p.sC = sC;
  p.raster = pick_raster(M, N, K);
  p.epi_early = g_opt_epi != 0;
  p.c_vec = ((int64_t)M * N * 2 >= (int64_t(64) << 20) && g_opt_nt != 0) ? 3 : 1;
  p.a_vw = p.b_vw = 8;
  const bool is_bf16 = (in_dt == TNH_BF16), out_f32 = (out_dt == TNH_F32);
  // 256x256 tiles once there are enough of them to fill the 256 CUs.
  bool big = (M >= 256 && N >= 256) && (((M + 255) / 256) * ((N + 255) / 256) * batch >= 192);
  if (variant == 3) big = false;
  if (variant == 4) big = true;
  if (g_opt_phases == 6 && (big || variant == 5) && M % 256 == 0 && N % 256 == 0 && K >= 128) {   // A/B knob ":p6"
    *name = "bf16_nt_256x256x64_w4";
    return launch_w4(is_bf16, out_f32, p, batch);
  }
  if (variant == 5 || (big && variant == 0 && g_pp_default)) {
    *name = "bf16_nt_256x256x64_pp";
    return launch_pp(is_bf16, out_f32, g_opt_phases != 4, p, batch, g_opt_phases == 3);
  }
  if (big) {
    *name = "bf16_nt_256x256x64";
    return launch_nt<256, 256, 2, 4>(is_bf16, out_f32, p, batch);
  }
  *name = "bf16_nt_128x128x64";
  return launch_nt<128, 128, 2, 2>(is_bf16, out_f32, p, batch);
}

}  // namespace tnh
