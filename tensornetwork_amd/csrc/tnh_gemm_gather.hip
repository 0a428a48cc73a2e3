// K2 (gather part of the bf16 / f16 streaming path): a SMALL matrix against a very long operand that is read
// WHERE IT LIES -- a many-axis intermediate whose contracted axes sit anywhere among its free ones,
//
//   C[Ms, Nl] = S[Ms, K] * L[Nl, K]^T          (small_first: the small operand is tensordot's `a`)
//   C[Nl, Ms] = L[Nl, K] * S[Ms, K]^T
//
// where row n of L is the n-th index tuple of the long tensor's FREE axes (row-major, natural axis order) and
// column k the k-th tuple of its CONTRACTED axes (row-major, memory order).  This is what `ncon` / the greedy
// contractor produce when a small tensor takes one or two bonds off a big intermediate
// (ncon_interface.py:336-343, 483-489 -> tensordot, numpy_backend.py:35-37): bond D = 12 gives 144 x 2 985 984
// x 144 against a rank-9 tensor with contracted axes (3, 8) or (1, 6).  The classic lowering permutes the
// intermediate to [free..., contracted...] first (K1: 2 x its bytes) and streams it again in the product
// (tnh_gemm_stream.hip); here the product's own loader does the gathering, so the intermediate is read once.
//
// A tile is a BOX of the long tensor: every contracted axis in full, and BN consecutive free tuples (all of
// the innermost free axes and T indices of the next one, BN = 48 or 64).  The box is a handful of
// contiguous pieces of memory (D = 12: twelve pieces of 1.1 - 1.5 KB); threads walk it in MEMORY order in
// 8-byte chunks (four elements of the innermost axis) and scatter them into the [row][k] LDS image the MFMA
// fragments are read from: one 8-byte LDS store when the innermost axis is contracted, four 2-byte stores
// into four consecutive rows when it is free.  Where a chunk comes from (offset in the box) and where it goes
// (row, k) depends only on the chunk number, not on the tile: every thread decodes its <= 12 chunks ONCE
// (mixed-radix over the box digits, `gather_chunk` below -- the same function the host uses to validate a
// descriptor and that tests call through tnh_gemm_gather_plan), then only the box origin changes per tile.
// Everything after the image (MFMA sequence per output element, LDS-staged epilogue, persistent grid,
// register prefetch of the next tile) is the streaming kernel's, so results are bit-identical to
// permute + tnh_gemm.
//
// Roofline: HBM.  Algorithmic bytes 2 * (Nl * K + Ms * K + Ms * Nl).
#include "tnh_gemm_nt.h"
#include <mutex>

namespace tnh {

// ---- descriptor arithmetic shared by the kernel, the host-side validation and tnh_gemm_gather_plan
// Chunk e of a box (8 bytes: four elements of the innermost digit), e counted in memory order: e written in the
// mixed radix of the box digits gives the digit indices, hence the element offset in the box and the image position.
// A thread owns chunks tid, tid + 256, tid + 512, ...: tid is decoded with one division per digit, the step of
// 256 is added digit by digit with carries afterwards (the divisions of a direct decode of every chunk were ~10 us
// of prologue per launch: profiles/r04_gather_gemm.md).  Chunks past the end of the box repeat the last one.
template <int NB>
__host__ __device__ __forceinline__ void gather_thread_chunks(const tnh_gather_desc& g, int tid, int total,
                                                              int (&off)[NB], int (&row)[NB], int (&kcol)[NB]) {
  unsigned radix[TNH_GATHER_MAX_DIGITS], digit[TNH_GATHER_MAX_DIGITS], step[TNH_GATHER_MAX_DIGITS];
  {
    unsigned e = (unsigned)tid, inc = 256u;
#pragma unroll
    for (int d = 0; d < TNH_GATHER_MAX_DIGITS; ++d) {
      // digits the box does not have count modulo 1 (always 0, carries fall through); the top digit is unbounded
      radix[d] = 1u;
      digit[d] = 0u;
      step[d] = 0u;
      if (d < g.nd) {
        const unsigned ext = d == 0 ? (unsigned)g.ext[0] >> 2 : (unsigned)g.ext[d];
        radix[d] = ext;
        const unsigned qe = e / ext, qi = inc / ext;
        digit[d] = e - qe * ext;
        step[d] = inc - qi * ext;
        e = qe;
        inc = qi;
      }
    }
  }
  int last_off = 0, last_row = 0, last_k = 0;     // chunk total - 1: every digit at its maximum
#pragma unroll
  for (int d = 0; d < TNH_GATHER_MAX_DIGITS; ++d) {
    if (d < g.nd) {
      const int top = d == 0 ? g.ext[0] - 4 : g.ext[d] - 1;
      last_off += top * g.stride[d];
      if ((g.k_mask >> d) & 1) last_k += top * g.mult[d];
      else last_row += top * g.mult[d];
    }
  }
#pragma unroll
  for (int it = 0; it < NB; ++it) {
    if (it * 256 + tid >= total) {
      off[it] = last_off;
      row[it] = last_row;
      kcol[it] = last_k;
    } else {
      int o = 0, r = 0, k = 0;
#pragma unroll
      for (int d = 0; d < TNH_GATHER_MAX_DIGITS; ++d) {
        if (d < g.nd) {
          const int idx = d == 0 ? (int)(digit[0] << 2) : (int)digit[d];
          o += idx * g.stride[d];
          if ((g.k_mask >> d) & 1) k += idx * g.mult[d];
          else r += idx * g.mult[d];
        }
      }
      off[it] = o;
      row[it] = r;
      kcol[it] = k;
    }
    unsigned carry = 0;                            // next chunk of this thread: + 256 in the mixed radix
#pragma unroll
    for (int d = 0; d < TNH_GATHER_MAX_DIGITS; ++d) {
      unsigned v = digit[d] + step[d] + carry;
      carry = v >= radix[d] ? 1u : 0u;
      if (carry) v -= radix[d];
      digit[d] = v;
    }
  }
}

__host__ __device__ __forceinline__ int64_t gather_tile_base(const tnh_gather_desc& g, int tile) {
  int64_t base = 0;
  unsigned t = (unsigned)tile;
#pragma unroll
  for (int d = 0; d < TNH_GATHER_MAX_TILE_DIGITS; ++d) {
    if (d < g.nt) {
      const unsigned ext = (unsigned)g.text[d];
      const unsigned q = t / ext;
      base += (int64_t)(t - q * ext) * g.tstride[d];
      t = q;
    }
  }
  return base;
}

// Zero, but not to the optimiser: added to an address computation inside a loop body it keeps that computation IN the
// body (hoisted out, the per-thread addresses of the K-loop kernel's small slice and epilogue cost ~50 registers and
// pushed the slice's prefetch registers into scratch).
__device__ __forceinline__ int opaque_zero() {
  int z = 0;
  asm volatile("" : "+s"(z));
  return z;
}

struct GatherArgs {
  tnh_gather_desc g;
  const uint16_t* S;   // small operand, [Ms][K] rows lds apart
  const uint16_t* L;   // the long tensor as it lies
  uint16_t* C;
  int64_t lds, ldc;
  int Ms, K;
  int ntiles;
};

// LDS as in tnh_gemm_stream.hip: [ small image: msf*16 rows x PA ][ R: max(long image BN x PA, staging) ].
// KIN: the innermost digit of the box is contracted (8-byte image stores); otherwise it is free (2-byte stores).
template <int BN, bool IS_BF16, bool SWAP, bool KIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_gather_kernel(GatherArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int FN = BN / 16;                       // long-operand fragments per tile (every wave takes all of them)
  constexpr int MAX_MI = 3;                         // small-operand fragments per wave: 12 / 4
  constexpr int NB = (BN * 48 + 255) / 256;         // 8-B chunks of the box per thread at K = 192
  constexpr int NST = (192 * BN / 8 + 255) / 256;   // 16-B output chunks per thread at Ms = 192
  static_assert(BN == 48 || BN == 64, "tile rows");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = p.K, Ms = p.Ms;
  const int Kp = (K + 31) & ~31, cpr = Kp >> 3, kch = K >> 3, ksteps = Kp >> 5;
  const int PA = Kp * 2 + 16;
  const int msf = (Ms + 15) >> 4;
  const int msp = msf * 16;
  char* sS = smem;
  char* sR = smem + msp * PA;
  const int EP = (SWAP ? msp : BN) * 2 + 16;
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);

  // ---- small operand -> LDS, once (zero rows / zero k padding)
  for (int idx = tid; idx < msp * cpr; idx += 256) {
    const int row = idx / cpr, c = idx - row * cpr;
    uint4 v = zero4;
    if (row < Ms && c < kch) v = *(const uint4*)(p.S + (int64_t)row * p.lds + c * 8);
    *(uint4*)(sS + row * PA + c * 16) = v;
  }

  // ---- this thread's chunks of a box: (offset in the box, image position) is the same for every tile
  int loff[NB], lsm[NB];
  {
    int crow[NB], ccol[NB];
    gather_thread_chunks<NB>(p.g, tid, BN * (K >> 2), loff, crow, ccol);
#pragma unroll
    for (int it = 0; it < NB; ++it) lsm[it] = crow[it] * PA + ccol[it] * 2;
  }
  // Two tiles of prefetch in registers (the streaming kernel keeps one): a box is 14 - 18 KB, and one box per
  // workgroup in flight is 7 - 9 MB over the chip -- less than HBM latency x rate (measured: 48-row boxes at
  // 2.6 TB/s with one, see profiles/r04_gather_gemm.md).
  uint2 rl0[NB], rl1[NB];
  auto load_long = [&](int tile, uint2 (&rl)[NB]) {
    const uint16_t* base = p.L + gather_tile_base(p.g, tile);
#pragma unroll
    for (int it = 0; it < NB; ++it) rl[it] = *(const uint2*)(base + loff[it]);
  };
  const int padch = cpr - kch;                     // 16-B chunks of zero k-padding per image row (0 ... 3)
  auto store_long = [&](const uint2 (&rl)[NB]) {
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      if constexpr (KIN) {
        *(uint2*)(sR + lsm[it]) = rl[it];
      } else {
        char* q = sR + lsm[it];
        *(uint16_t*)(q) = (uint16_t)(rl[it].x & 0xffffu);
        *(uint16_t*)(q + PA) = (uint16_t)(rl[it].x >> 16);
        *(uint16_t*)(q + 2 * PA) = (uint16_t)(rl[it].y & 0xffffu);
        *(uint16_t*)(q + 3 * PA) = (uint16_t)(rl[it].y >> 16);
      }
    }
    // the staging area overwrote the padding columns: zero them again (LDS only -- no branch around a global access)
    for (int idx = tid; idx < BN * padch; idx += 256) {
      const int row = idx / padch, c = idx - row * padch;
      *(uint4*)(sR + row * PA + (kch + c) * 16) = zero4;
    }
  };

  // ---- this thread's output chunks (staging byte offset, element offset in C from the tile origin)
  int esm[NST], eoff[NST];
  {
    const int total = SWAP ? BN * (Ms >> 3) : Ms * (BN / 8);
#pragma unroll
    for (int it = 0; it < NST; ++it) {
      int idx = it * 256 + tid;
      if (idx >= total) idx = total - 1;
      if constexpr (SWAP) {
        const int rch = Ms >> 3;
        const int row = idx / rch, ch = idx - row * rch;
        esm[it] = row * EP + ch * 16;
        eoff[it] = row * (int)p.ldc + ch * 8;
      } else {
        constexpr int CH = BN / 8;
        const int row = idx / CH, ch = idx - row * CH;
        esm[it] = row * EP + ch * 16;
        eoff[it] = row * (int)p.ldc + ch * 8;   // ldc * Ms < 2^30 is host-checked
      }
    }
  }

  const int frag_row = lane & 15, frag_chk = lane >> 4;
  const int last = p.ntiles - 1;
  const int step = (int)gridDim.x;             // the host launches at most ntiles workgroups

  // One tile: `hold` has the NEXT tile's box (requested one body earlier), `fill` takes the one after it.  The image
  // of the next tile is written at the END of the body, right after this tile's stores were issued: there the wait
  // for `hold` has one predecessor path with static counts (the loads into `fill`, then NST stores: all younger).
  auto body = [&](int tile, uint2 (&hold)[NB], uint2 (&fill)[NB]) {
    __syncthreads();                            // long image (and, first time, the small image) complete
    {
      const int far = tile + 2 * step;
      load_long(far <= last ? far : last, fill);   // in flight during this tile and the next
    }

    f32x4 acc[MAX_MI][FN];
#pragma unroll
    for (int i = 0; i < MAX_MI; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int ks = 0; ks < ksteps; ++ks) {
      const int koff = (ks * 4 + frag_chk) * 16;
      uint4 lf[FN];
#pragma unroll
      for (int j = 0; j < FN; ++j) lf[j] = *(const uint4*)(sR + (j * 16 + frag_row) * PA + koff);
#pragma unroll
      for (int i = 0; i < MAX_MI; ++i) {
        const int mi = wid + 4 * i;
        if (mi < msf) {                         // wave-uniform
          const uint4 sf = *(const uint4*)(sS + (mi * 16 + frag_row) * PA + koff);
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            // first argument: the index a lane holds 4 consecutive values of
            if constexpr (SWAP) acc[i][j] = mma16<IS_BF16>(sf, lf[j], acc[i][j]);
            else acc[i][j] = mma16<IS_BF16>(lf[j], sf, acc[i][j]);
          }
        }
      }
    }
    __syncthreads();                            // long image consumed: R becomes the staging area

#pragma unroll
    for (int i = 0; i < MAX_MI; ++i) {
      const int mi = wid + 4 * i;
      if (mi < msf) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          uint2 o;
          o.x = pack2<IS_BF16>(acc[i][j][0], acc[i][j][1]);
          o.y = pack2<IS_BF16>(acc[i][j][2], acc[i][j][3]);
          if constexpr (SWAP) {   // row = long index (lane & 15), 4 consecutive small indices
            *(uint2*)(sR + (j * 16 + frag_row) * EP + (mi * 16 + frag_chk * 4) * 2) = o;
          } else {                // row = small index (lane & 15), 4 consecutive long indices
            *(uint2*)(sR + (mi * 16 + frag_row) * EP + (j * 16 + frag_chk * 4) * 2) = o;
          }
        }
      }
    }
    __syncthreads();

    {
      // SWAP: C rows = long index (tile origin n0 * ldc); else C rows = small index (tile origin n0)
      uint16_t* cbase = p.C + (SWAP ? (int64_t)tile * BN * p.ldc : (int64_t)tile * BN);
#pragma unroll
      for (int it = 0; it < NST; ++it) {
        const uint4 v = *(const uint4*)(sR + esm[it]);
        *(uint4*)(cbase + eoff[it]) = v;
      }
    }
    __syncthreads();                            // staging consumed: R takes the next long image
    store_long(hold);
  };

  int tile = blockIdx.x;
  load_long(tile, rl0);
  store_long(rl0);
  load_long(tile + step <= last ? tile + step : last, rl0);
  for (;;) {
    body(tile, rl0, rl1);
    tile += step;
    if (tile > last) break;
    body(tile, rl1, rl0);
    tile += step;
    if (tile > last) break;
  }
}

// The same product with a K LOOP (kl_ext > 1): K = kl_ext * Kbox with only the Kbox innermost contracted indices inside
// a box; the outermost contracted digit is walked step by step (p.g.kl_stride elements per step) with the accumulators
// staying in registers, and the Ms x Kbox slice of the small operand that belongs to the step is re-staged into LDS
// (it comes from L2: Ms x K x 2 bytes in all, read once per tile).  Steps of all tiles of a workgroup form ONE sequence
// through which the boxes are prefetched two deep and the small slices one deep; the epilogue runs after a tile's last
// step.  D = 12: 144 x 248 832 x 1728 against contracted axes (1, 7, 8) of a rank-9 tensor = 12 steps of 144.
template <int BN, bool IS_BF16, bool SWAP, bool KIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_gather_kloop_kernel(GatherArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int FN = BN / 16;
  constexpr int MAX_MI = 3;
  constexpr int NB = (BN * 48 + 255) / 256;
  constexpr int NST = (192 * BN / 8 + 255) / 256;
  constexpr int NS = 11;                            // 16-B chunks of a small slice per thread (host: Ms * Kbox / 8 <= 2816)
  static_assert(BN == 48 || BN == 64, "tile rows");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = p.K, Ms = p.Ms;                    // K: indices inside a box (Kbox)
  const int Kp = (K + 31) & ~31, cpr = Kp >> 3, kch = K >> 3, ksteps = Kp >> 5;
  const int PA = Kp * 2 + 16;
  const int msf = (Ms + 15) >> 4;
  const int msp = msf * 16;
  char* sS = smem;
  char* sR = smem + msp * PA;
  const int EP = (SWAP ? msp : BN) * 2 + 16;
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  const int klx = p.g.kl_ext;
  const int64_t kls = p.g.kl_stride;

  // ---- small image: zero rows / zero k padding once; the slices overwrite the Ms x Kbox part only
  for (int idx = tid; idx < msp * cpr; idx += 256) {
    const int row = idx / cpr, c = idx - row * cpr;
    *(uint4*)(sS + row * PA + c * 16) = zero4;
  }
  int spos[NS];                                    // (row << 8) | chunk of this thread's pieces of a slice
  {
    const int total = Ms * kch;
#pragma unroll
    for (int it = 0; it < NS; ++it) {
      int idx = it * 256 + tid;
      if (idx >= total) idx = total - 1;
      const int row = idx / kch;
      spos[it] = (row << 8) | (idx - row * kch);
    }
  }
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  v4u sreg[NS];
  auto load_small = [&](int kl) {
    const uint16_t* base = p.S + (int64_t)kl * K;
    const int z = opaque_zero();
#pragma unroll
    for (int it = 0; it < NS; ++it) {
      const int pos = spos[it] + z;
      sreg[it] = *(const v4u*)(base + (pos >> 8) * (int)p.lds + (pos & 255) * 8);
    }
  };
  auto store_small = [&]() {
    const int z = opaque_zero();
#pragma unroll
    for (int it = 0; it < NS; ++it) {
      const int pos = spos[it] + z;
      *(v4u*)(sS + (pos >> 8) * PA + (pos & 255) * 16) = sreg[it];
    }
  };

  int loff[NB], lsm[NB];
  {
    int crow[NB], ccol[NB];
    gather_thread_chunks<NB>(p.g, tid, BN * (K >> 2), loff, crow, ccol);
#pragma unroll
    for (int it = 0; it < NB; ++it) lsm[it] = crow[it] * PA + ccol[it] * 2;
  }
  // Boxes of register prefetch: two at 48 rows, one at 64 (the small slice needs 44 registers; 229 of 256 are taken
  // at 64 rows with one).  With one, a step sees most of a load latency: 4.3 us per step against 2.7 us per tile of
  // the single-box kernel (profiles/r04_gather_gemm.md).
  constexpr int DEPTH = BN == 48 ? 2 : 1;
  uint2 rl0[NB], rl1[DEPTH == 2 ? NB : 1];
  auto load_long = [&](int tile, int kl, uint2 (&rl)[NB]) {
    const uint16_t* base = p.L + gather_tile_base(p.g, tile) + (int64_t)kl * kls;
#pragma unroll
    for (int it = 0; it < NB; ++it) rl[it] = *(const uint2*)(base + loff[it]);
  };
  const int padch = cpr - kch;
  auto store_long = [&](const uint2 (&rl)[NB]) {
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      if constexpr (KIN) {
        *(uint2*)(sR + lsm[it]) = rl[it];
      } else {
        char* q = sR + lsm[it];
        *(uint16_t*)(q) = (uint16_t)(rl[it].x & 0xffffu);
        *(uint16_t*)(q + PA) = (uint16_t)(rl[it].x >> 16);
        *(uint16_t*)(q + 2 * PA) = (uint16_t)(rl[it].y & 0xffffu);
        *(uint16_t*)(q + 3 * PA) = (uint16_t)(rl[it].y >> 16);
      }
    }
    for (int idx = tid; idx < BN * padch; idx += 256) {
      const int row = idx / padch, c = idx - row * padch;
      *(uint4*)(sR + row * PA + (kch + c) * 16) = zero4;
    }
  };

  const int frag_row = lane & 15, frag_chk = lane >> 4;
  const int last = p.ntiles - 1;
  const int step = (int)gridDim.x;
  // the step after (tile, kl) in this workgroup's sequence; past the end it stays on the last step (a harmless re-read)
  auto advance = [&](int& tile, int& kl) {
    if (kl + 1 < klx) {
      ++kl;
    } else if (tile + step <= last) {
      tile += step;
      kl = 0;
    }
  };

  f32x4 acc[MAX_MI][FN];
#pragma unroll
  for (int i = 0; i < MAX_MI; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // One step.  `hold` receives (DEPTH 1) or already has (DEPTH 2) the NEXT step's box; with DEPTH 2 `fill` takes the
  // one after it.  The next step's slice of the small operand is requested here too.
  auto body = [&](int tile, int kl, uint2 (&hold)[NB], uint2 (&fill)[NB]) {
    __syncthreads();                            // both images complete
    {
      int t1 = tile, k1 = kl;
      advance(t1, k1);
      if constexpr (DEPTH == 2) {
        int t2 = t1, k2 = k1;
        advance(t2, k2);
        load_long(t2, k2, fill);                // in flight during this step and the next
      } else {
        load_long(t1, k1, hold);                // in flight during the MFMA work (and the epilogue) of this step
      }
      load_small(k1);
    }
    for (int ks = 0; ks < ksteps; ++ks) {
      const int koff = (ks * 4 + frag_chk) * 16;
      uint4 lf[FN];
#pragma unroll
      for (int j = 0; j < FN; ++j) lf[j] = *(const uint4*)(sR + (j * 16 + frag_row) * PA + koff);
#pragma unroll
      for (int i = 0; i < MAX_MI; ++i) {
        const int mi = wid + 4 * i;
        if (mi < msf) {
          const uint4 sf = *(const uint4*)(sS + (mi * 16 + frag_row) * PA + koff);
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            if constexpr (SWAP) acc[i][j] = mma16<IS_BF16>(sf, lf[j], acc[i][j]);
            else acc[i][j] = mma16<IS_BF16>(lf[j], sf, acc[i][j]);
          }
        }
      }
    }
    __syncthreads();                            // images consumed
    if (kl == klx - 1) {                        // (uniform) the tile's last step: results out, accumulators cleared
#pragma unroll
      for (int i = 0; i < MAX_MI; ++i) {
        const int mi = wid + 4 * i;
        if (mi < msf) {
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            uint2 o;
            o.x = pack2<IS_BF16>(acc[i][j][0], acc[i][j][1]);
            o.y = pack2<IS_BF16>(acc[i][j][2], acc[i][j][3]);
            if constexpr (SWAP) *(uint2*)(sR + (j * 16 + frag_row) * EP + (mi * 16 + frag_chk * 4) * 2) = o;
            else *(uint2*)(sR + (mi * 16 + frag_row) * EP + (j * 16 + frag_chk * 4) * 2) = o;
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      }
      __syncthreads();
      {
        uint16_t* cbase = p.C + (SWAP ? (int64_t)tile * BN * p.ldc : (int64_t)tile * BN);
        const int total = SWAP ? BN * (Ms >> 3) : Ms * (BN / 8);
        const int z = opaque_zero();
#pragma unroll
        for (int it = 0; it < NST; ++it) {
          int idx = it * 256 + tid + z;
          if (idx >= total) idx = total - 1;
          const int per_row = SWAP ? (Ms >> 3) : BN / 8;
          const int row = idx / per_row, ch = idx - row * per_row;
          const uint4 v = *(const uint4*)(sR + row * EP + ch * 16);
          *(uint4*)(cbase + (int64_t)row * p.ldc + ch * 8) = v;
        }
      }
      __syncthreads();                          // staging consumed
    }
    store_long(hold);
    store_small();
  };

  int tile = blockIdx.x, kl = 0;
  load_long(tile, 0, rl0);
  load_small(0);
  __syncthreads();                              // the zero fill of the small image is complete
  store_long(rl0);
  store_small();
  if constexpr (DEPTH == 2) {
    {
      int t1 = tile, k1 = kl;
      advance(t1, k1);
      load_long(t1, k1, rl0);
    }
    for (;;) {
      body(tile, kl, rl0, rl1);
      if (kl == klx - 1 && tile + step > last) break;
      advance(tile, kl);
      body(tile, kl, rl1, rl0);
      if (kl == klx - 1 && tile + step > last) break;
      advance(tile, kl);
    }
  } else {
    for (;;) {
      body(tile, kl, rl0, rl0);
      if (kl == klx - 1 && tile + step > last) break;
      advance(tile, kl);
    }
  }
}

// ---- host side
// Everything the kernel relies on, checked by brute force over the <= 3072 chunks of a box (rows and columns of the
// image are hit exactly once, offsets stay inside the tensor).  Returns the tile height BN (48 / 64) or 0.
static int gather_validate(const tnh_gather_desc& g, int64_t K, int64_t Nl, int64_t l_elems, const char** why) {
  *why = nullptr;
  if (g.nd < 1 || g.nd > TNH_GATHER_MAX_DIGITS || g.nt < 0 || g.nt > TNH_GATHER_MAX_TILE_DIGITS) {
    *why = "digit counts";
    return 0;
  }
  int64_t rows = 1, cols = 1, span = 0;
  for (int d = 0; d < g.nd; ++d) {
    if (g.ext[d] < 1 || g.stride[d] < 1 || g.mult[d] < 1) { *why = "non-positive extent / stride / weight"; return 0; }
    if (d == 0 ? (g.stride[0] != 1 || g.mult[0] != 1 || g.ext[0] % 4) : (g.stride[d] % 4 != 0)) {
      *why = "the innermost digit must be contiguous with a multiple of 4 elements, the others 8-byte aligned";
      return 0;
    }
    if ((g.k_mask >> d) & 1) cols *= g.ext[d];
    else rows *= g.ext[d];
    span += (int64_t)(g.ext[d] - 1) * g.stride[d];
    if (rows > 64 || cols > 192 || span >= (int64_t(1) << 31)) { *why = "box too large"; return 0; }
  }
  if (cols != K || (rows != 48 && rows != 64)) { *why = "box is not BN x K with BN = 48 or 64"; return 0; }
  int64_t tiles = 1, tspan = 0;
  for (int d = 0; d < g.nt; ++d) {
    if (g.text[d] < 1 || g.tstride[d] < 0 || g.tstride[d] % 4) { *why = "tile digits"; return 0; }
    tiles *= g.text[d];
    tspan += (int64_t)(g.text[d] - 1) * g.tstride[d];
    if (tiles >= (int64_t(1) << 30)) { *why = "too many tiles"; return 0; }
  }
  if (tiles * rows != Nl) { *why = "tiles x BN != long rows"; return 0; }
  if (g.kl_ext < 1 || g.kl_ext > 4096 || (g.kl_ext > 1 && (g.kl_stride < 4 || g.kl_stride % 4))) {
    *why = "K loop";
    return 0;
  }
  if (tspan + span + (int64_t)(g.kl_ext - 1) * g.kl_stride + 1 > l_elems) { *why = "the last box leaves the tensor"; return 0; }
  // the image is covered exactly once
  const int total = (int)(rows * (K >> 2));
  const bool kin = g.k_mask & 1;
  unsigned char seen[64 * 48];
  memset(seen, 0, sizeof(seen));
  for (int tid = 0; tid < 256; ++tid) {
    int coff[12], crow[12], ccol[12];
    gather_thread_chunks<12>(g, tid, total, coff, crow, ccol);      // the kernel's own decode, thread by thread
    for (int it = 0; it < 12 && it * 256 + tid < total; ++it) {
      const int off = coff[it], row = crow[it], kcol = ccol[it];
      if (off < 0 || off > span || row < 0 || kcol < 0) { *why = "chunk out of the box"; return 0; }
      for (int i = 0; i < 4; ++i) {
        const int r = kin ? row : row + i, c = kin ? kcol + i : kcol;
        if (r >= rows || c >= K) { *why = "chunk leaves the image"; return 0; }
        unsigned char& cell = seen[r * 48 + (c >> 2)];
        const unsigned char bit = (unsigned char)(1u << (c & 3));
        if (cell & bit) { *why = "two chunks write one image element"; return 0; }
        cell |= bit;
      }
    }
  }
  return (int)rows;
}

// The check costs ~0.1 ms of host time and a contraction path asks for the same few descriptors slice after slice:
// the last 32 accepted (descriptor, K, Nl, elements) are remembered.
static int gather_validate_cached(const tnh_gather_desc& g, int64_t K, int64_t Nl, int64_t l_elems, const char** why) {
  struct Entry {
    tnh_gather_desc g;
    int64_t K, Nl, l_elems;
    int bn;
  };
  static std::mutex mu;
  static Entry seen[32];
  static int count = 0, next = 0;
  *why = nullptr;
  {
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < count; ++i)
      if (seen[i].K == K && seen[i].Nl == Nl && seen[i].l_elems == l_elems && !memcmp(&seen[i].g, &g, sizeof(g)))
        return seen[i].bn;
  }
  const int bn = gather_validate(g, K, Nl, l_elems, why);
  if (bn) {
    std::lock_guard<std::mutex> lock(mu);
    Entry& e = seen[next];
    e.g = g;
    e.K = K;
    e.Nl = Nl;
    e.l_elems = l_elems;
    e.bn = bn;
    next = (next + 1) % 32;
    if (count < 32) ++count;
  }
  return bn;
}

static bool gather_enabled() {
  static const bool on = []() { const char* e = getenv("TNH_GATHER_GEMM"); return !(e && e[0] == '0'); }();
  return on;
}

template <int BN, bool SWAP, bool KIN>
static int launch_gather(bool is_bf16, const GatherArgs& p, size_t lds_bytes) {
  static const int attr_rc = []() -> int {
    for (const void* k : {reinterpret_cast<const void*>(gemm_gather_kernel<BN, true, SWAP, KIN>),
                          reinterpret_cast<const void*>(gemm_gather_kernel<BN, false, SWAP, KIN>),
                          reinterpret_cast<const void*>(gemm_gather_kloop_kernel<BN, true, SWAP, KIN>),
                          reinterpret_cast<const void*>(gemm_gather_kloop_kernel<BN, false, SWAP, KIN>)})
      if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 1;
    return 0;
  }();
  if (attr_rc) {
    set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the gather GEMM");
    return TNH_ERR_HIP;
  }
  auto go = [&](auto kernel) -> int {
    int per_cu = 0;
    TNH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds_bytes));
    if (per_cu < 1) per_cu = 1;
    int64_t gx = (int64_t)num_cus() * per_cu;
    if (gx > p.ntiles) gx = p.ntiles;
    hipLaunchKernelGGL(kernel, dim3((unsigned)gx), dim3(256), lds_bytes, stream(), p);
    return TNH_OK;
  };
  int rc;
  if (p.g.kl_ext > 1) rc = is_bf16 ? go(gemm_gather_kloop_kernel<BN, true, SWAP, KIN>) : go(gemm_gather_kloop_kernel<BN, false, SWAP, KIN>);
  else rc = is_bf16 ? go(gemm_gather_kernel<BN, true, SWAP, KIN>) : go(gemm_gather_kernel<BN, false, SWAP, KIN>);
  if (rc) return rc;
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

template <int BN>
static int launch_gather_bn(bool swap, bool kin, bool is_bf16, const GatherArgs& p, size_t lds_bytes) {
  if (swap) return kin ? launch_gather<BN, true, true>(is_bf16, p, lds_bytes) : launch_gather<BN, true, false>(is_bf16, p, lds_bytes);
  return kin ? launch_gather<BN, false, true>(is_bf16, p, lds_bytes) : launch_gather<BN, false, false>(is_bf16, p, lds_bytes);
}

// C[Ms, Nl] (small_first) or C[Nl, Ms]: see the file header.  TNH_ERR_UNSUPPORTED (nothing launched) when the
// descriptor or an alignment rule is outside the kernel's range: the caller permutes and calls tnh_gemm.
int gemm_gather(int dtype, int64_t Ms, int64_t K, int64_t Nl, const void* S, int64_t lds, const void* L,
                int64_t l_elems, const tnh_gather_desc* desc, void* C, int64_t ldc, int small_first,
                const char** name) {
  const bool swap = !small_first;
  const char* why = nullptr;
  int bn = 0;
  const int64_t steps = desc->kl_ext >= 1 ? desc->kl_ext : 0;       // K = steps * Kbox (steps == 1: no K loop)
  const int64_t Kbox = steps && K % steps == 0 ? K / steps : 0;
  const bool shape_ok = gather_enabled() && Ms >= 1 && Ms <= 192 && Kbox >= 8 && Kbox <= 192 && Kbox % 8 == 0 && Nl >= 48 &&
                        lds % 8 == 0 && ldc % 8 == 0 && lds >= K && ((uintptr_t)S % 16) == 0 &&
                        ((uintptr_t)L % 8) == 0 && ((uintptr_t)C % 16) == 0 && (!swap || Ms % 8 == 0) &&
                        ldc >= (swap ? Ms : Nl) && Ms * lds < (int64_t(1) << 30) &&
                        (steps == 1 || Ms * (Kbox / 8) <= 11 * 256);
  if (shape_ok) bn = gather_validate_cached(*desc, Kbox, Nl, l_elems, &why);
  if (shape_ok && bn && (swap ? (int64_t)bn * ldc : Ms * ldc) >= (int64_t(1) << 30)) {
    bn = 0;
    why = "32-bit offsets inside one output tile";
  }
  if (!bn) {
    set_error("tnh_gemm_gather: outside the gather kernel's range (%s)",
              why ? why : "shape / alignment / TNH_GATHER_GEMM=0");
    return TNH_ERR_UNSUPPORTED;
  }
  GatherArgs p;
  p.g = *desc;
  p.S = (const uint16_t*)S;
  p.L = (const uint16_t*)L;
  p.C = (uint16_t*)C;
  p.lds = lds;
  p.ldc = ldc;
  p.Ms = (int)Ms;
  p.K = (int)Kbox;
  p.ntiles = (int)(Nl / bn);
  const int Kp = (int)((Kbox + 31) & ~31), PA = Kp * 2 + 16;
  const int msp = (p.Ms + 15) / 16 * 16;
  const size_t image = (size_t)bn * PA;
  const size_t staging = swap ? (size_t)bn * (msp * 2 + 16) : (size_t)msp * (bn * 2 + 16);
  const size_t lds_bytes = (size_t)msp * PA + (image > staging ? image : staging);
  const bool kin = desc->k_mask & 1;
  const bool is_bf16 = dtype == TNH_BF16;
  if (bn == 48) {
    *name = steps > 1 ? (swap ? "bf16_gather_kloop_48xS" : "bf16_gather_kloop_Sx48") : (swap ? "bf16_gather_48xS" : "bf16_gather_Sx48");
    return launch_gather_bn<48>(swap, kin, is_bf16, p, lds_bytes);
  }
  *name = steps > 1 ? (swap ? "bf16_gather_kloop_64xS" : "bf16_gather_kloop_Sx64") : (swap ? "bf16_gather_64xS" : "bf16_gather_Sx64");
  return launch_gather_bn<64>(swap, kin, is_bf16, p, lds_bytes);
}

// Host-only (no device needed): the chunk plan of one box and the origins of the first `ntiles` tiles, exactly as
// the kernel computes them.  Returns the tile height BN or a negative error code.
int gemm_gather_plan(const tnh_gather_desc* desc, int64_t K, int64_t Nl, int64_t l_elems, int32_t* chunk_off,
                     int32_t* chunk_row, int32_t* chunk_k, int64_t nchunks, int64_t* tile_base, int64_t ntiles) {
  const char* why = nullptr;
  const int64_t steps = desc->kl_ext >= 1 ? desc->kl_ext : 0;
  const int64_t Kbox = steps && K % steps == 0 ? K / steps : 0;      // the plan is that of ONE step of the K loop
  const int bn = Kbox >= 4 && Kbox % 4 == 0 ? gather_validate(*desc, Kbox, Nl, l_elems, &why) : 0;
  if (!bn) {
    set_error("tnh_gemm_gather_plan: %s", why ? why : "invalid descriptor");
    return TNH_ERR_UNSUPPORTED;
  }
  const int64_t total = (int64_t)bn * (Kbox >> 2);
  TNH_REQUIRE(nchunks >= 0 && nchunks <= total && ntiles >= 0 && ntiles <= Nl / bn,
              "tnh_gemm_gather_plan: more chunks / tiles asked than there are");
  for (int tid = 0; tid < 256; ++tid) {
    int coff[12], crow[12], ccol[12];
    gather_thread_chunks<12>(*desc, tid, (int)total, coff, crow, ccol);
    for (int it = 0; it < 12; ++it) {
      const int64_t e = (int64_t)it * 256 + tid;
      if (e >= nchunks) break;
      chunk_off[e] = coff[it];
      chunk_row[e] = crow[it];
      chunk_k[e] = ccol[it];
    }
  }
  for (int64_t t = 0; t < ntiles; ++t) tile_base[t] = gather_tile_base(*desc, (int)t);
  return bn;
}

}  // namespace tnh
