// K1: index-permutation / strided gather-scatter kernels (HBM-bound, bit-exact).
//
//   tnh_permute        numpy.transpose semantics; host-side dimension merging,
//                      then (a) memcpy, (b) row-preserving gather, or
//                      (c) LDS-tiled 2-D transpose with the remaining dims as
//                      batch, coalesced on both the read and the write side.
//   tnh_strided_copy   dst contiguous <- arbitrary-strided src (slice,
//                      diagonal, broadcast).
//   tnh_strided_scatter  arbitrary-strided dst <- contiguous src (diagflat).
//
// Algorithmic HBM bytes: 2 * numel * itemsize per call.
#include <stdlib.h>
#include <algorithm>
#include "tnh_internal.h"

namespace tnh {

struct GatherParams {
  int rank;
  int64_t total;
  int64_t shape[TNH_MAX_RANK];    // output (iteration) shape, slowest first
  int64_t stride[TNH_MAX_RANK];   // element stride on the strided side
  int64_t offset;
  uint32_t magic[TNH_MAX_RANK];   // 32-bit path: division by shape[d] as mulhi + shifts
  uint32_t shift[TNH_MAX_RANK];   //   (q = mulhi(n, magic); ((n - q) >> 1) + q) >> shift)
};

// Unsigned division by an invariant d >= 2 (round-up method with a 33-bit multiplier):
// exact for every 32-bit n.  The index chain of the gather kernel is ALU-bound with hardware
// divisions (a rank-8 chain per 16-B element capped the kernel at ~1 TB/s).
static void fastdiv_gen(uint32_t d, uint32_t* magic, uint32_t* shift) {
  const uint32_t k = 31 - (uint32_t)__builtin_clz(d);
  if ((d & (d - 1)) == 0) {
    *magic = 0;
    *shift = k - 1;
    return;
  }
  const uint64_t num = (uint64_t)1 << (32 + k);
  uint64_t m = num / d;
  const uint64_t rem = num % d;
  m += m;
  if (rem + rem >= d) m += 1;
  *magic = (uint32_t)(1 + m);
  *shift = k;
}
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, uint32_t magic, uint32_t shift) {
  const uint32_t q = __umulhi(n, magic);
  return (((n - q) >> 1) + q) >> shift;
}

// One element per thread-iteration; the contiguous side is indexed linearly so
// it is always coalesced.  SCATTER=false: dst[i] = src[f(i)]; true: dst[f(i)] = src[i].
template <typename T, typename IDX, bool SCATTER>
__global__ __launch_bounds__(256) void gather_kernel(T* __restrict__ dst,
                                                     const T* __restrict__ src,
                                                     GatherParams p) {
  const IDX total = (IDX)p.total;
  const IDX step = (IDX)gridDim.x * blockDim.x;
  for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
    IDX rem = i;
    int64_t off = p.offset;
#pragma unroll 1
    for (int d = p.rank - 1; d >= 0; --d) {
      const IDX s = (IDX)p.shape[d];
      IDX q;
      if constexpr (sizeof(IDX) == 4) q = (IDX)fastdiv((uint32_t)rem, p.magic[d], p.shift[d]);
      else q = rem / s;
      const IDX c = rem - q * s;
      off += (int64_t)c * p.stride[d];
      rem = q;
    }
    if (SCATTER) dst[off] = src[i];
    else dst[i] = src[off];
  }
}

// Row-preserving permute whose rows are shorter than a 128-B line: the lines of the source are
// shared by g = 128 / row_bytes rows that sit far apart in the output, so a linear walk of the
// output touches every source line from g different workgroups (different XCDs, i.e. different
// L2s) and the kernel reads g x the bytes (4.2 TB/s algorithmic = 6.4 TB/s of traffic on the
// (16,)^6 f32 case of configs[2]).  Here the walk is over (outer..., line-mate, row): the g rows
// of one source line are taken by neighbouring lanes of one wave, and because the dim just
// outside the line-mate is the output's next-fastest, the same wave also fills whole output
// lines.  Both sides are strided, so each element carries two offsets.
struct Gather2Params {
  int rank;
  uint32_t total;
  uint32_t shape[TNH_MAX_RANK + 1];
  uint32_t sstride[TNH_MAX_RANK + 1];
  uint32_t dstride[TNH_MAX_RANK + 1];
  uint32_t magic[TNH_MAX_RANK + 1];
  uint32_t shift[TNH_MAX_RANK + 1];
};

template <typename T>
__global__ __launch_bounds__(256) void gather2_kernel(T* __restrict__ dst, const T* __restrict__ src,
                                                      Gather2Params p) {
  const uint32_t step = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < p.total; i += step) {
    uint32_t rem = i, so = 0, dof = 0;
#pragma unroll 1
    for (int d = p.rank - 1; d >= 0; --d) {
      const uint32_t q = fastdiv(rem, p.magic[d], p.shift[d]);
      const uint32_t c = rem - q * p.shape[d];
      so += c * p.sstride[d];
      dof += c * p.dstride[d];
      rem = q;
    }
    dst[dof] = src[so];
  }
}

struct TiledParams {
  int64_t Na, Nb;          // a: fastest dim of src, b: fastest dim of dst
  int64_t a_out_stride;    // stride of a in dst
  int64_t b_in_stride;     // stride of b in src
  int64_t tiles_a, tiles_b;
  int64_t nblocks;
  int nbatch;
  int order = 1;           // permute_tiled16: consecutive workgroups along b (1 = a fastest)
  int64_t bshape[TNH_MAX_RANK];
  int64_t bin[TNH_MAX_RANK];
  int64_t bout[TNH_MAX_RANK];
};

// 2-D transpose of a TILE x TILE patch through LDS (padded rows: no bank
// conflicts on the transposed read).  Reads run along a (src-contiguous),
// writes run along b (dst-contiguous): both sides coalesced.
template <typename T, int TILE>
__global__ __launch_bounds__(256) void permute_tiled_kernel(T* __restrict__ dst,
                                                            const T* __restrict__ src,
                                                            TiledParams p) {
  __shared__ T tile[TILE][TILE + 1];
  constexpr int ROWS_PER_PASS = 256 / TILE;
  // grid-stride over tiles: gridDim.x * blockDim.x must stay below 2^32 on this runtime
  for (int64_t blk = blockIdx.x; blk < p.nblocks; blk += gridDim.x) {
  int64_t bid = blk;
  int64_t ta, tb;          // p.order consecutive workgroups along b (see permute_tiled16_kernel)
  if (p.order > 1) {
    const int64_t tl = bid % p.order;
    bid /= p.order;
    ta = bid % p.tiles_a;
    bid /= p.tiles_a;
    const int64_t groups = p.tiles_b / p.order;
    tb = (bid % groups) * p.order + tl;
    bid /= groups;
  } else {
    ta = bid % p.tiles_a;
    bid /= p.tiles_a;
    tb = bid % p.tiles_b;
    bid /= p.tiles_b;
  }
  int64_t in_base = 0, out_base = 0;
#pragma unroll 1
  for (int d = p.nbatch - 1; d >= 0; --d) {
    const int64_t q = bid / p.bshape[d];
    const int64_t c = bid - q * p.bshape[d];
    in_base += c * p.bin[d];
    out_base += c * p.bout[d];
    bid = q;
  }
  const int tx = threadIdx.x % TILE;
  const int ty = threadIdx.x / TILE;
  const int64_t a0 = ta * TILE, b0 = tb * TILE;
  {
    const int64_t a = a0 + tx;
#pragma unroll
    for (int k = 0; k < TILE / ROWS_PER_PASS; ++k) {
      const int r = ty + k * ROWS_PER_PASS;
      const int64_t b = b0 + r;
      if (a < p.Na && b < p.Nb) tile[r][tx] = src[in_base + b * p.b_in_stride + a];
    }
  }
  __syncthreads();
  {
    const int64_t b = b0 + tx;
#pragma unroll
    for (int k = 0; k < TILE / ROWS_PER_PASS; ++k) {
      const int r = ty + k * ROWS_PER_PASS;
      const int64_t a = a0 + r;
      if (a < p.Na && b < p.Nb) dst[out_base + a * p.a_out_stride + b] = tile[tx][r];
    }
  }
  __syncthreads();
  }
}

// 2-byte elements (bf16 / f16), full tiles, 16-byte aligned rows on both sides:
// 64 (a) x 128 (b) tile.  A thread loads 8 a-consecutive elements of two
// neighbouring b rows (2 x 16 B), transposes the eight 2x2 micro-blocks in
// registers (v_perm) into dwords that hold (b, b+1) for one a, and stores them
// to a dword tile T[a][b/2] of 64 x 64 dwords whose 16-byte chunks are XOR-swizzled
// with (a >> 3) & 7: the ds_write_b32 of a 32-lane half (8 a-chunks q x 4 columns)
// then lands on 32 different banks, and the write side fetches its 4 consecutive
// dwords with ONE conflict-free ds_read_b128 (the 16 lanes of a group read the 16
// chunks of one row, permuted).  Round 1's padded layout (row pitch 65 dwords, four
// ds_read_b32 per store) measured 50 % LDS bank-conflict cycles (c and c + 8 on one
// bank).  Stores are 16 B (8 b-consecutive elements): 256-B runs per output row.
// TBV = 64 (round 6): 64 x 64 tiles for a destination-fastest extent that is a multiple of 64 but not of 128 -- the
// [..., c, d, e] -> [..., d, .., e, c] passes of the chi = 64 MERA slices ran through the scalar 64 x 64 tiles at 2.5
// TB/s (8 % of a slice).  Same register transposes and swizzle; the dword tile is 64 x 32, a 16-lane group of the write
// side reads two rows (conflict-free: consecutive rows start 32 banks apart), 128-byte runs per output row.
template <int NA, int TBV = 128>
__global__ __launch_bounds__(256) void permute_tiled16_kernel(uint16_t* __restrict__ dst,
                                                              const uint16_t* __restrict__ src,
                                                              TiledParams p) {
  // NA = 2 (round-3 experiment, not dispatched): two neighbouring a-tiles per workgroup = 256-byte source runs and
  // eight loads per thread in flight.  Measured: no change (4.24 vs 4.22 TB/s) -- what helps is the ORDER of the
  // tiles (p.order below), i.e. the write side.
  constexpr int TA = 64, TB = TBV, LD = TB / 2;
  __shared__ __attribute__((aligned(16))) uint32_t T[NA][TA * LD];
  const int64_t tiles_a = p.tiles_a / NA;          // NA == 2 only with an even tile count
  const int64_t nblocks = p.nblocks / NA;
  for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
  int64_t bid = blk;
  // tile order: p.order consecutive workgroups take neighbouring b-tiles (their 256-byte output runs are adjacent in
  // every output row: longer DRAM bursts on the write side), then a runs, then the remaining b-tiles
  int64_t ta, tb;
  if (p.order > 1) {
    const int64_t tl = bid % p.order;
    bid /= p.order;
    ta = bid % tiles_a;
    bid /= tiles_a;
    const int64_t groups = p.tiles_b / p.order;
    tb = (bid % groups) * p.order + tl;
    bid /= groups;
  } else {
    ta = bid % tiles_a;
    bid /= tiles_a;
    tb = bid % p.tiles_b;
    bid /= p.tiles_b;
  }
  int64_t in_base = 0, out_base = 0;
#pragma unroll 1
  for (int d = p.nbatch - 1; d >= 0; --d) {
    const int64_t q = bid / p.bshape[d];
    const int64_t c = bid - q * p.bshape[d];
    in_base += c * p.bin[d];
    out_base += c * p.bout[d];
    bid = q;
  }
  const int tid = threadIdx.x;
  const int64_t a0 = ta * (TA * NA), b0 = tb * TB;
  {
    const int q = tid & 7;  // a chunk: a = 8 q .. 8 q + 7
    constexpr int NH = TB / 64;     // b pairs per thread
    uint4 xv[NA][NH], yv[NA][NH];
#pragma unroll
    for (int t = 0; t < NA; ++t)
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int j = (tid >> 3) + 32 * h;  // b pair: rows 2 j, 2 j + 1
        const uint16_t* s0 = src + in_base + (b0 + 2 * j) * p.b_in_stride + a0 + t * TA + 8 * q;
        xv[t][h] = *(const uint4*)s0;
        yv[t][h] = *(const uint4*)(s0 + p.b_in_stride);
      }
#pragma unroll
    for (int t = 0; t < NA; ++t)
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int j = (tid >> 3) + 32 * h;
        const uint4 x = xv[t][h], y = yv[t][h];
        const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // xs[e] = (a = 8q+2e, a+1) of row 2j ; ys[e] the same of row 2j+1
          const uint32_t lo = (xs[e] & 0xffffu) | (ys[e] << 16);          // a = 8q+2e:   (b=2j, 2j+1)
          const uint32_t hi = (xs[e] >> 16) | (ys[e] & 0xffff0000u);      // a = 8q+2e+1
          const int col = (((j >> 2) ^ q) << 2) | (j & 3);                 // (a >> 3) & 7 == q for both rows
          T[t][(8 * q + 2 * e) * LD + col] = lo;
          T[t][(8 * q + 2 * e + 1) * LD + col] = hi;
        }
      }
  }
  __syncthreads();
  {
    constexpr int CH = TB / 8;             // 16-byte chunks per output row of the tile
    const int c = tid & (CH - 1);  // 16-byte chunk along b: dwords 4 c .. 4 c + 3
#pragma unroll
    for (int t = 0; t < NA; ++t)
#pragma unroll
      for (int h = 0; h < TA * CH / 256; ++h) {
        const int a = tid / CH + (256 / CH) * h;
        const uint4 o = *(const uint4*)&T[t][a * LD + ((c ^ ((a >> 3) & 7)) << 2)];
        *(uint4*)(dst + out_base + (a0 + t * TA + a) * p.a_out_stride + b0 + 8 * c) = o;
      }
  }
  __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// Brick permute: for tensors whose fastest source / destination dims are SMALL
// (bond-dimension-16 networks: every dim is 16), where a 2-D tile over one (a, b)
// pair would be mostly empty.  A brick is the product of index ranges over the
// few fastest source dims (volume >= TA: contiguous source runs) and the few
// fastest destination dims (volume >= TB: contiguous destination runs); all
// other dims are batch.  The block reads a brick in source order (coalesced),
// scatters it into LDS at its destination-order position (padded against bank
// conflicts) and streams it out in destination order (coalesced).  Index
// tables are built once per block and reused for every brick it owns.
struct BrickParams {
  int nd;                       // dims after merging, destination order (slowest first)
  int64_t ext[TNH_MAX_RANK];    // brick extent per dim (1 = batch dim)
  int64_t cnt[TNH_MAX_RANK];    // oshape / ext
  int64_t istr[TNH_MAX_RANK];   // source element stride
  int64_t ostr[TNH_MAX_RANK];   // destination element stride
  int nb;                       // brick dims (ext > 1)
  int sdim[TNH_MAX_RANK];       // brick dims, source-fastest first
  int ddim[TNH_MAX_RANK];       // brick dims, destination-fastest first
  int jstr[TNH_MAX_RANK];       // destination-order stride (inside the brick) of sdim[q]
  int sext[TNH_MAX_RANK];       // ext[sdim[q]]
  uint32_t smagic[TNH_MAX_RANK];  // ceil(2^32 / sext[q]) (exact quotients for indices < V <= 2^15)
  int V, runA, runB;            // brick volume; contiguous run length in source / destination
  uint32_t magicA, magicB;      // ceil(2^32 / run)
  int pr_shift;                 // log2 of the padding period in bytes, < 0: no padding
  int64_t nbricks;
};

template <typename T>
__device__ __forceinline__ int brick_pad(int j, int pr_shift) {
  if (pr_shift < 0) return j;
  return j + (int)(((uint32_t)j * (uint32_t)sizeof(T)) >> pr_shift) * (sizeof(T) == 2 ? 2 : 1);
}

// RG / WG: elements per thread on the read / write side: 1, or a vector of 4 / 16 bytes (2-byte T: 2 / 8
// elements, 4-byte T: 4 elements).  Round 2: the 16-byte forms.  With one dword per lane the kernel is bound by
// its LDS instruction count (per 4 bytes: run-offset lookup, two position lookups, two 2-byte scatters on the
// read side, offset lookup + dword read on the write side -- ~8 LDS operations; 4.3e8 bf16 elements with 288-byte
// source runs: 0.57 ms = 3.0 TB/s); a 16-byte vector shares the lookups between 8 elements (~15 LDS operations
// per 16 bytes instead of 32).
template <typename T, int RG, int WG>
__global__ __launch_bounds__(256) void permute_brick_kernel(T* __restrict__ dst, const T* __restrict__ src,
                                                            BrickParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int V = p.V;
  const int nrunA = V / p.runA, nrunB = V / p.runB;
  int64_t* offS = (int64_t*)smem;                 // [nrunA]
  int64_t* offD = offS + nrunA;                   // [nrunB]
  T* data = (T*)(smem + (((size_t)(nrunA + nrunB) * 8 + 15) & ~size_t(15)));
  const int tid = threadIdx.x;
  for (int k = tid; k < nrunA; k += 256) {
    int rem = k * p.runA;
    int64_t off = 0;
    for (int q = 0; q < p.nb; ++q) {
      const int d = p.sdim[q];
      const int e = (int)p.ext[d];
      off += (int64_t)(rem % e) * p.istr[d];
      rem /= e;
    }
    offS[k] = off;
  }
  for (int k = tid; k < nrunB; k += 256) {
    int rem = k * p.runB;
    int64_t off = 0;
    for (int q = 0; q < p.nb; ++q) {
      const int d = p.ddim[q];
      const int e = (int)p.ext[d];
      off += (int64_t)(rem % e) * p.ostr[d];
      rem /= e;
    }
    offD[k] = off;
  }
  __syncthreads();
  // Destination-order position (inside the brick) of source-order element i, and its coordinate along the
  // source-fastest brick dim: a few multiply-highs per VECTOR; the elements of a vector advance by jstr[0] and
  // re-decode only when they wrap along that (fused, so long) dim.  Round 1 read every element's position from a
  // V-entry LDS table: one more LDS read per element and as much LDS as the payload.  (A per-run table plus a
  // carry chain inside the run was tried and is slower: the predicated carries cost more VALU than the decode.)
  const int e0 = p.sext[0], j0 = p.jstr[0];
  auto decode = [&](int i, int& c0) -> int {
    int rem = i, j = 0;
    for (int q = 0; q < p.nb; ++q) {
      const int quo = (int)(((uint64_t)(uint32_t)rem * p.smagic[q]) >> 32);
      const int c = rem - quo * p.sext[q];
      if (q == 0) c0 = c;
      j += c * p.jstr[q];
      rem = quo;
    }
    return j;
  };
  for (int64_t brick = blockIdx.x; brick < p.nbricks; brick += gridDim.x) {
    int64_t rem = brick, bs = 0, bd = 0;
    for (int d = p.nd - 1; d >= 0; --d) {
      const int64_t c = rem % p.cnt[d];
      rem /= p.cnt[d];
      bs += c * p.ext[d] * p.istr[d];
      bd += c * p.ext[d] * p.ostr[d];
    }
    for (int i = tid * RG; i < V; i += 256 * RG) {
      const int run = (int)(((uint64_t)(uint32_t)i * p.magicA) >> 32);
      const int w = i - run * p.runA;
      const T* g = src + bs + offS[run] + w;
      T v[RG];
      if constexpr (RG > 1 && RG * sizeof(T) == 16) {
        *(uint4*)v = *(const uint4*)g;
      } else if constexpr (RG > 1 && RG * sizeof(T) == 8) {
        *(uint2*)v = *(const uint2*)g;
      } else if constexpr (RG == 2) {
        *(uint32_t*)v = *(const uint32_t*)g;
      } else {
        v[0] = *g;
      }
      int c0;
      int j = decode(i, c0);
      if (c0 + RG <= e0) {                          // the whole vector lies along the source-fastest dim (the usual case)
#pragma unroll
        for (int e = 0; e < RG; ++e) data[brick_pad<T>(j + e * j0, p.pr_shift)] = v[e];
      } else {
#pragma unroll
        for (int e = 0; e < RG; ++e) {
          data[brick_pad<T>(j, p.pr_shift)] = v[e];
          if (e + 1 < RG) {
            ++c0;
            j += j0;
            if (c0 == e0) j = decode(i + e + 1, c0);   // wrapped: next index of the slower dims
          }
        }
      }
    }
    __syncthreads();
    for (int j = tid * WG; j < V; j += 256 * WG) {
      const int run = (int)(((uint64_t)(uint32_t)j * p.magicB) >> 32);
      const int w = j - run * p.runB;
      T* g = dst + bd + offD[run] + w;
      const int pj = brick_pad<T>(j, p.pr_shift);
      if constexpr (WG > 1 && WG * sizeof(T) == 16) {
        // the padding period is >= 16 bytes (host-checked), so the 16 bytes at j are contiguous in LDS, dword-aligned
        const uint32_t* q = (const uint32_t*)(data + pj);
        *(uint4*)g = make_uint4(q[0], q[1], q[2], q[3]);
      } else if constexpr (WG > 1 && WG * sizeof(T) == 8) {
        const uint32_t* q = (const uint32_t*)(data + pj);
        *(uint2*)g = make_uint2(q[0], q[1]);
      } else if constexpr (WG == 2) {
        *(uint32_t*)g = *(const uint32_t*)(data + pj);
      } else {
        *g = data[pj];
      }
    }
    __syncthreads();
  }
}

// Plans and launches the brick kernel; returns TNH_ERR_UNSUPPORTED when the shape does not
// suit it (the caller then falls through to the tiled / gather kernels).
template <typename T>
static int try_brick(void* dst, const void* src, int r, const int64_t* oshape, const int64_t* istride,
                     const int64_t* ostride, int64_t total) {
  constexpr int ISZ = (int)sizeof(T);
  // payload per brick: 32 KiB; round 1: 16 KiB (the position table doubled the LDS)
  static const int max_kb = []() { const char* e = getenv("TNH_BRICK_MAXKB"); return e ? atoi(e) : 32; }();
  const int maxV = (max_kb * 1024 / ISZ) < 32768 ? (max_kb * 1024 / ISZ) : 32768;
  // (source run, destination run) targets in elements, best first; the first plan whose brick
  // fits in maxV wins (long source runs matter most: 256-element runs measured 3.7 TB/s vs 2.8)
  // round 2 (computed positions, vector accesses, bricks up to 32 KiB): 512-element runs on both sides first --
  // tools/brick_target_probe.sh: 12^8 (0,3,7,8,4,2,5,6,1) 0.58 -> 0.43 ms, 8^9 0.197 -> 0.134 ms; 1024 x 1024 is slower
  // again (41 KiB bricks, one or two workgroups per CU)
  static const int cand2[][2] = {{512, 512}, {512, 256}, {256, 256}, {256, 128}, {128, 128}, {64, 128}, {64, 64}, {32, 64}, {32, 32}};
  static const int cand4[][2] = {{256, 256}, {256, 128}, {128, 64}, {64, 64}, {32, 64}, {32, 32}, {16, 32}};
  static const int cand8[][2] = {{128, 128}, {64, 32}, {32, 32}, {16, 32}, {16, 16}};
  const int (*cand)[2] = ISZ == 2 ? cand2 : (ISZ == 4 ? cand4 : cand8);
  int ncand = ISZ == 2 ? 9 : (ISZ == 4 ? 7 : 5);
  int envc[1][2];
  if (getenv("TNH_BRICK_TA") && getenv("TNH_BRICK_TB")) {   // tuning knobs (tools/permute_one.py)
    envc[0][0] = atoi(getenv("TNH_BRICK_TA"));
    envc[0][1] = atoi(getenv("TNH_BRICK_TB"));
    cand = envc;
    ncand = 1;
  }
  // smallest divisor of n that is >= need, searched in a bounded window (0: none there)
  auto smallest_divisor_ge = [](int64_t n, int64_t need) -> int64_t {
    if (need >= n) return n;
    for (int64_t e = need; e <= n && e <= 64 * need; ++e)
      if (n % e == 0) return e;
    return 0;
  };
  int order_s[TNH_MAX_RANK];
  for (int d = 0; d < r; ++d) order_s[d] = d;
  std::sort(order_s, order_s + r, [&](int x, int y) { return istride[x] < istride[y]; });
  BrickParams p;
  p.nd = r;
  for (int d = 0; d < r; ++d) {
    p.istr[d] = istride[d];
    p.ostr[d] = ostride[d];
  }
  bool planned = false;
  for (int ci = 0; ci < ncand && !planned; ++ci) {
    const int TA = cand[ci][0], TB = cand[ci][1];
    for (int d = 0; d < r; ++d) p.ext[d] = 1;
    bool ok = true;
    int64_t vol = 1;
    for (int q = 0; q < r && vol < TA && ok; ++q) {   // source-fastest dims
      const int d = order_s[q];
      p.ext[d] = smallest_divisor_ge(oshape[d], (TA + vol - 1) / vol);
      if (p.ext[d] == 0) ok = false;
      else vol *= p.ext[d];
    }
    vol = 1;
    for (int d = r - 1; d >= 0 && vol < TB && ok; --d) {   // destination-fastest dims
      const int64_t need = (TB + vol - 1) / vol;
      if (p.ext[d] < need) p.ext[d] = smallest_divisor_ge(oshape[d], need);
      if (p.ext[d] == 0) ok = false;
      else vol *= p.ext[d];
    }
    if (!ok) continue;
    int64_t V = 1;
    for (int d = 0; d < r; ++d) V *= p.ext[d];
    if (V > maxV || V < 64) continue;
    p.V = (int)V;
    planned = true;
  }
  if (!planned) return TNH_ERR_UNSUPPORTED;
  p.nbricks = 1;
  for (int d = 0; d < r; ++d) {
    p.cnt[d] = oshape[d] / p.ext[d];
    p.nbricks *= p.cnt[d];
  }
  // brick dims in both orders
  p.nb = 0;
  for (int q = 0; q < r; ++q)
    if (p.ext[order_s[q]] > 1) p.sdim[p.nb++] = order_s[q];
  int nbd = 0;
  for (int d = r - 1; d >= 0; --d)
    if (p.ext[d] > 1) p.ddim[nbd++] = d;
  // destination-order stride inside the brick
  int dstr_of[TNH_MAX_RANK];
  int acc = 1;
  for (int q = 0; q < nbd; ++q) {
    dstr_of[p.ddim[q]] = acc;
    acc *= (int)p.ext[p.ddim[q]];
  }
  for (int q = 0; q < p.nb; ++q) {
    p.jstr[q] = dstr_of[p.sdim[q]];
    p.sext[q] = (int)p.ext[p.sdim[q]];
    p.smagic[q] = (uint32_t)((((uint64_t)1 << 32) + p.sext[q] - 1) / p.sext[q]);
  }
  // contiguous runs: leading brick dims while they are whole and adjacent in memory
  auto run_len = [&](const int* dims, const int64_t* str, int max_dims, int* ndims) -> int {
    int64_t run = 1, expect = 1;
    int n = 0;
    for (int q = 0; q < p.nb && n < max_dims; ++q) {
      const int d = dims[q];
      if (str[d] != expect) break;
      run *= p.ext[d];
      ++n;
      if (p.ext[d] != oshape[d]) break;   // a partial dim ends the run
      expect = str[d] * oshape[d];
    }
    if (ndims) *ndims = n;
    return (int)run;
  };
  p.runA = run_len(p.sdim, p.istr, TNH_MAX_RANK, nullptr);
  p.runB = run_len(p.ddim, p.ostr, TNH_MAX_RANK, nullptr);
  if (p.runA < 1 || p.runB < 1 || p.V % p.runA || p.V % p.runB) return TNH_ERR_UNSUPPORTED;
  p.magicA = (uint32_t)((((uint64_t)1 << 32) + p.runA - 1) / p.runA);
  p.magicB = (uint32_t)((((uint64_t)1 << 32) + p.runB - 1) / p.runB);
  if (p.runA == 1 || p.runB == 1) return TNH_ERR_UNSUPPORTED;  // degenerate: leave to the gather kernel
  // padding period: the byte stride (in destination order) between source-consecutive elements
  const int64_t sa_bytes = (int64_t)p.jstr[0] * ISZ;
  p.pr_shift = -1;
  if (sa_bytes >= 8) {
    int sh = 3;
    while (((int64_t)1 << (sh + 1)) <= sa_bytes) ++sh;
    p.pr_shift = sh;
  }
  const size_t padV = (size_t)p.V + (p.pr_shift < 0 ? 0 : (((size_t)p.V * ISZ) >> p.pr_shift) * (ISZ == 2 ? 2 : 1)) + 8;
  const size_t tables = (((size_t)(p.V / p.runA + p.V / p.runB) * 8 + 15) & ~size_t(15));
  const size_t smem = tables + padV * ISZ;
  if (smem > 64 * 1024) return TNH_ERR_UNSUPPORTED;
  int64_t grid = (int64_t)num_cus() * 16;
  if (const char* e = getenv("TNH_BRICK_GRID")) grid = (int64_t)num_cus() * atoi(e);
  if (grid > p.nbricks) grid = p.nbricks;
  // widest access per side: every run start must be aligned to it (base pointer, every stride but the unit one)
  auto width = [&](int run, const void* base, const int64_t* str, int want) -> int {
    int wdt = want;
    while (wdt > 1) {
      bool ok = (run % wdt == 0) && ((uintptr_t)base % ((size_t)wdt * ISZ) == 0);
      for (int d = 0; d < r && ok; ++d)
        if (str[d] != 1 && str[d] % wdt != 0) ok = false;
      if (ok) break;
      wdt >>= 1;
    }
    return wdt;
  };
  static const bool vec16 = []() { const char* e = getenv("TNH_BRICK_VEC16"); return !(e && e[0] == '0'); }();
  const int vmax = (ISZ <= 4 && vec16) ? 16 / ISZ : (ISZ == 2 ? 2 : 1);
  const int rg = width(p.runA, src, istride, vmax);
  int wg = width(p.runB, dst, ostride, vmax);
  // a vector write reads wg * ISZ contiguous LDS bytes: the padding period must not cut them
  while (wg * ISZ > 4 && p.pr_shift >= 0 && (1 << p.pr_shift) < wg * ISZ) wg >>= 1;
  (void)total;
#define TNH_BRICK_LAUNCH(RG_, WG_)                                                                                \
  hipLaunchKernelGGL((permute_brick_kernel<T, RG_, WG_>), dim3((unsigned)grid), dim3(256), smem, stream(), (T*)dst, \
                     (const T*)src, p)
#define TNH_BRICK_WG(RG_)                            \
  switch (wg) {                                      \
    case 8: TNH_BRICK_LAUNCH(RG_, 8); break;         \
    case 4: TNH_BRICK_LAUNCH(RG_, 4); break;         \
    case 2: TNH_BRICK_LAUNCH(RG_, 2); break;         \
    default: TNH_BRICK_LAUNCH(RG_, 1); break;        \
  }
#define TNH_BRICK_WG4(RG_)                           \
  switch (wg) {                                      \
    case 4: TNH_BRICK_LAUNCH(RG_, 4); break;         \
    case 2: TNH_BRICK_LAUNCH(RG_, 2); break;         \
    default: TNH_BRICK_LAUNCH(RG_, 1); break;        \
  }
  if constexpr (ISZ == 2) {
    switch (rg) {
      case 8: TNH_BRICK_WG(8); break;
      case 4: TNH_BRICK_WG(4); break;
      case 2: TNH_BRICK_WG(2); break;
      default: TNH_BRICK_WG(1); break;
    }
  } else if constexpr (ISZ == 4) {
    switch (rg) {
      case 4: TNH_BRICK_WG4(4); break;
      case 2: TNH_BRICK_WG4(2); break;
      default: TNH_BRICK_WG4(1); break;
    }
  } else {
    TNH_BRICK_LAUNCH(1, 1);
  }
#undef TNH_BRICK_WG
#undef TNH_BRICK_WG4
#undef TNH_BRICK_LAUNCH
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

#ifndef TNH_BRICK_RAGGED_DEFAULT
#define TNH_BRICK_RAGGED_DEFAULT true
#endif

template <typename T, bool SCATTER>
static int launch_gather(void* dst, const void* src, const GatherParams& p) {
  if (p.total == 0) return TNH_OK;
  int64_t blocks = (p.total + 255) / 256;
  // (narrower elements do more index arithmetic per byte and want the full 16: 24-byte rows of a D = 12 tensor, 8-byte
  //  elements: 4.07 TB/s with 16 workgroups per CU, 2.96 with 4)
  const int64_t cap = (int64_t)num_cus() * (sizeof(T) == 16 ? stream_wgs_per_cu(p.total * (int64_t)sizeof(T)) : 16);
  if (blocks > cap) blocks = cap;
  // 32-bit index math when every offset fits (the common case).
  bool small = p.total < (int64_t(1) << 31);
  int64_t span = p.offset;
  for (int d = 0; d < p.rank; ++d) span += (p.shape[d] - 1) * (p.stride[d] < 0 ? -p.stride[d] : p.stride[d]);
  if (span >= (int64_t(1) << 31)) small = false;
  for (int d = 0; d < p.rank; ++d)
    if (p.shape[d] < 2) small = false;   // fastdiv needs d >= 2 (unit dims are squeezed before; be safe)
  if (small) {
    GatherParams q = p;
    for (int d = 0; d < q.rank; ++d) fastdiv_gen((uint32_t)q.shape[d], &q.magic[d], &q.shift[d]);
    hipLaunchKernelGGL((gather_kernel<T, uint32_t, SCATTER>), dim3((unsigned)blocks), dim3(256), 0,
                       stream(), (T*)dst, (const T*)src, q);
  }
  else
    hipLaunchKernelGGL((gather_kernel<T, int64_t, SCATTER>), dim3((unsigned)blocks), dim3(256), 0,
                       stream(), (T*)dst, (const T*)src, p);
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

template <bool SCATTER>
static int dispatch_gather(void* dst, const void* src, const GatherParams& p, int itemsize) {
  switch (itemsize) {
    case 1: return launch_gather<uint8_t, SCATTER>(dst, src, p);
    case 2: return launch_gather<uint16_t, SCATTER>(dst, src, p);
    case 4: return launch_gather<uint32_t, SCATTER>(dst, src, p);
    case 8: return launch_gather<uint64_t, SCATTER>(dst, src, p);
    case 16: return launch_gather<uint4, SCATTER>(dst, src, p);
    default:
      set_error("unsupported itemsize %d", itemsize);
      return TNH_ERR_UNSUPPORTED;
  }
}

template <typename T, int TILE>
static int launch_tiled(void* dst, const void* src, TiledParams p, int64_t nblocks) {
  p.nblocks = nblocks;
  {
    const char* eo = getenv("TNH_PERMUTE_ORDER_T");      // default 1 until measured per element size
    int order = eo ? atoi(eo) : 1;
    if (order < 1) order = 1;
    while (order > 1 && p.tiles_b % order != 0) order >>= 1;
    p.order = order;
  }
  const int64_t grid = nblocks < (int64_t(1) << 22) ? nblocks : (int64_t(1) << 22);
  hipLaunchKernelGGL((permute_tiled_kernel<T, TILE>), dim3((unsigned)grid), dim3(256), 0, stream(),
                     (T*)dst, (const T*)src, p);
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

// Squeeze unit dims and fuse runs of output dims that are also adjacent (and
// in order) in the input.  On return `shape`/`stride` describe the OUTPUT dims
// (size, src element stride), slowest first.
static int simplify(int rank, const int64_t* in_shape, const int32_t* perm, int64_t* shape,
                    int64_t* stride) {
  int64_t in_stride[TNH_MAX_RANK];
  int64_t acc = 1;
  for (int d = rank - 1; d >= 0; --d) {
    in_stride[d] = acc;
    acc *= in_shape[d];
  }
  int r = 0;
  int prev_axis = -2;
  for (int d = 0; d < rank; ++d) {
    const int ax = perm[d];
    if (in_shape[ax] == 1) continue;
    if (r > 0 && ax == prev_axis + 1) {
      // contiguous continuation of the previous output dim (unit dims between
      // them were skipped, which keeps the stride relation intact).
      shape[r - 1] *= in_shape[ax];
      stride[r - 1] = in_stride[ax];
    } else {
      shape[r] = in_shape[ax];
      stride[r] = in_stride[ax];
      ++r;
    }
    prev_axis = ax;
    // skip over unit input dims that directly follow `ax`
    while (prev_axis + 1 < rank && in_shape[prev_axis + 1] == 1) ++prev_axis;
  }
  return r;
}

}  // namespace tnh

using namespace tnh;

extern "C" {

int tnh_strided_copy(void* dst, const void* src, int rank, const int64_t* shape,
                     const int64_t* src_strides, int64_t src_offset, int itemsize) {
  TNH_NEED_INIT();
  TNH_REQUIRE(rank >= 0 && rank <= TNH_MAX_RANK, "rank %d out of range", rank);
  GatherParams p;
  p.rank = 0;
  p.total = 1;
  p.offset = src_offset;
  for (int d = 0; d < rank; ++d) {
    TNH_REQUIRE(shape[d] >= 0, "negative dimension");
    p.total *= shape[d];
    if (shape[d] == 1) continue;
    // fuse with previous dim when contiguous on the strided side too
    if (p.rank > 0 && p.stride[p.rank - 1] == src_strides[d] * shape[d]) {
      p.shape[p.rank - 1] *= shape[d];
      p.stride[p.rank - 1] = src_strides[d];
    } else {
      p.shape[p.rank] = shape[d];
      p.stride[p.rank] = src_strides[d];
      ++p.rank;
    }
  }
  if (p.total == 0) return TNH_OK;
  TNH_REQUIRE(dst && src, "null pointer");
  if (p.rank == 0 || (p.rank == 1 && p.stride[0] == 1)) {
    TNH_HIP(hipMemcpyAsync(dst, (const char*)src + src_offset * itemsize,
                           (size_t)p.total * itemsize, hipMemcpyDeviceToDevice, stream()));
    return TNH_OK;
  }
  return dispatch_gather<false>(dst, src, p, itemsize);
}

int tnh_strided_scatter(void* dst, const void* src, int rank, const int64_t* shape,
                        const int64_t* dst_strides, int64_t dst_offset, int itemsize) {
  TNH_NEED_INIT();
  TNH_REQUIRE(rank >= 0 && rank <= TNH_MAX_RANK, "rank %d out of range", rank);
  GatherParams p;
  p.rank = 0;
  p.total = 1;
  p.offset = dst_offset;
  for (int d = 0; d < rank; ++d) {
    TNH_REQUIRE(shape[d] >= 0, "negative dimension");
    p.total *= shape[d];
    if (shape[d] == 1) continue;
    p.shape[p.rank] = shape[d];
    p.stride[p.rank] = dst_strides[d];
    ++p.rank;
  }
  if (p.total == 0) return TNH_OK;
  TNH_REQUIRE(dst && src, "null pointer");
  return dispatch_gather<true>(dst, src, p, itemsize);
}

int tnh_permute(void* dst, const void* src, int rank, const int64_t* shape, const int32_t* perm,
                int itemsize) {
  TNH_NEED_INIT();
  TNH_REQUIRE(rank >= 0 && rank <= TNH_MAX_RANK, "rank %d out of range (max %d)", rank,
              TNH_MAX_RANK);
  TNH_REQUIRE(itemsize == 1 || itemsize == 2 || itemsize == 4 || itemsize == 8 || itemsize == 16,
              "unsupported itemsize %d", itemsize);
  int64_t total = 1;
  {
    bool seen[TNH_MAX_RANK] = {false};
    for (int d = 0; d < rank; ++d) {
      TNH_REQUIRE(perm[d] >= 0 && perm[d] < rank && !seen[perm[d]], "perm is not a permutation");
      seen[perm[d]] = true;
      TNH_REQUIRE(shape[d] >= 0, "negative dimension");
      total *= shape[d];
    }
  }
  if (total == 0) return TNH_OK;
  TNH_REQUIRE(dst && src, "null pointer");

  int64_t oshape[TNH_MAX_RANK], istride[TNH_MAX_RANK];
  const int r = simplify(rank, shape, perm, oshape, istride);

  // (a) identity after simplification
  if (r <= 1) {
    TNH_HIP(hipMemcpyAsync(dst, src, (size_t)total * itemsize, hipMemcpyDeviceToDevice, stream()));
    return TNH_OK;
  }

  // (b) innermost dim preserved: rows stay contiguous -> widened gather.
  if (istride[r - 1] == 1) {
    GatherParams p;
    p.rank = r;
    p.offset = 0;
    int64_t row = oshape[r - 1];
    int wide = itemsize;
    const uintptr_t align = (uintptr_t)dst | (uintptr_t)src;
    while (wide < 16 && row % 2 == 0 && (align % (2 * wide)) == 0) {
      // every other stride must stay a multiple of the widened element
      bool ok = true;
      for (int d = 0; d < r - 1; ++d)
        if (istride[d] % 2 != 0) ok = false;
      if (!ok) break;
      for (int d = 0; d < r - 1; ++d) istride[d] /= 2;
      row /= 2;
      wide *= 2;
    }
    p.total = 1;
    for (int d = 0; d < r; ++d) {
      p.shape[d] = (d == r - 1) ? row : oshape[d];
      p.stride[d] = istride[d];
      p.total *= p.shape[d];
    }
    // rows shorter than a line: walk line-mates together (see gather2_kernel)
    static const int linemate = [] { const char* e = getenv("TNH_PERMUTE_LINEMATE"); return e ? atoi(e) : 1; }();
    const int64_t row_bytes = row * wide;
    if (linemate && wide == 16 && row >= 2 && row_bytes < 128 && 128 % row_bytes == 0 && r >= 3 && r <= TNH_MAX_RANK &&
        p.total < (int64_t(1) << 31)) {
      int ds = -1;  // the source dim whose stride is one row: its neighbours share lines
      for (int d = 0; d < r - 1; ++d)
        if (istride[d] == row) ds = d;
      int64_t g = 128 / row_bytes;
      while (ds >= 0 && g > 1 && oshape[ds] % g != 0) g /= 2;
      if (ds >= 0 && g > 1 && oshape[ds] / g >= 1) {
        Gather2Params q;
        int64_t ost[TNH_MAX_RANK];
        {
          int64_t acc = 1;
          for (int d = r - 1; d >= 0; --d) {
            ost[d] = acc;
            acc *= p.shape[d];
          }
        }
        int n = 0;
        bool ok = true;
        auto push = [&](int64_t sz, int64_t ss, int64_t dd) {
          if (sz == 1) return;
          if (ss * (sz - 1) >= (int64_t(1) << 31) || dd * (sz - 1) >= (int64_t(1) << 31)) ok = false;
          q.shape[n] = (uint32_t)sz;
          q.sstride[n] = (uint32_t)ss;
          q.dstride[n] = (uint32_t)dd;
          fastdiv_gen((uint32_t)sz, &q.magic[n], &q.shift[n]);
          ++n;
        };
        for (int d = 0; d < r - 1; ++d) {
          if (d == ds) push(oshape[d] / g, istride[d] * g, ost[d] * g);
          else push(oshape[d], istride[d], ost[d]);
        }
        push(g, istride[ds], ost[ds]);
        push(row, 1, 1);
        if (ok) {
          q.rank = n;
          q.total = (uint32_t)p.total;
          int64_t blocks = (p.total + 255) / 256;
          const int64_t cap = (int64_t)num_cus() * stream_wgs_per_cu(p.total * 16);
          if (blocks > cap) blocks = cap;
          hipLaunchKernelGGL((gather2_kernel<uint4>), dim3((unsigned)blocks), dim3(256), 0, stream(), (uint4*)dst,
                             (const uint4*)src, q);
          TNH_LAUNCH_CHECK();
          return TNH_OK;
        }
      }
    }
    return dispatch_gather<false>(dst, src, p, wide);
  }

  // (c) the fastest src dim `a` moves: LDS-tiled transpose over (a, b).
  int ia = -1;  // output position of the src-fastest dim
  for (int d = 0; d < r; ++d)
    if (istride[d] == 1) ia = d;
  const int ib = r - 1;
  int64_t ostride[TNH_MAX_RANK];
  {
    int64_t acc = 1;
    for (int d = r - 1; d >= 0; --d) {
      ostride[d] = acc;
      acc *= oshape[d];
    }
  }
  // (c0) small fastest dims on either side: brick kernel
  // (round 5) also for 2-byte tensors whose (a, b) extents the 64 x 128 fast path below cannot tile (D = 96, 160, ...):
  // the scalar 64 x 64 fallback moved the [K][N] -> [N][K] pass of a (96,)^4 tensor at 1.48 TB/s (measured)
  static const bool brick_ragged = []() { const char* e = getenv("TNH_BRICK_RAGGED"); return e ? atoi(e) != 0 : TNH_BRICK_RAGGED_DEFAULT; }();
  // (not for [64][N] -> [N][64] with whole 64 x 64 tiles: their output is one contiguous 8 KB piece per tile and the
  //  scalar tiles stay ahead, 2.66 against 2.06 TB/s -- profiles/r05_ragged_brick_probe.txt)
  const bool ragged16 = brick_ragged && ia >= 0 && itemsize == 2 && (oshape[ia] % 64 != 0 || oshape[ib] % 128 != 0) &&
                        !(oshape[ib] == 64 && ia == r - 2 && oshape[ia] % 64 == 0);
  if (ia >= 0 && (oshape[ia] < 64 || oshape[ib] < 64 || ragged16) && itemsize >= 2 && !getenv("TNH_PERMUTE_NOBRICK")) {
    int rc = TNH_ERR_UNSUPPORTED;
    switch (itemsize) {
      case 2: rc = try_brick<uint16_t>(dst, src, r, oshape, istride, ostride, total); break;
      case 4: rc = try_brick<uint32_t>(dst, src, r, oshape, istride, ostride, total); break;
      case 8: rc = try_brick<uint64_t>(dst, src, r, oshape, istride, ostride, total); break;
      case 16: rc = try_brick<uint4>(dst, src, r, oshape, istride, ostride, total); break;
    }
    if (rc != TNH_ERR_UNSUPPORTED) return rc;
  }
  if (ia >= 0 && oshape[ia] >= 16 && oshape[ib] >= 16) {
    TiledParams p;
    p.Na = oshape[ia];
    p.Nb = oshape[ib];
    p.a_out_stride = ostride[ia];
    p.b_in_stride = istride[ib];
    // 2-byte fast path: full 64 x 128 tiles, every row start 16-byte aligned on both sides
    if (itemsize == 2 && p.Na % 64 == 0 && p.Nb % 64 == 0 && p.b_in_stride % 8 == 0 &&
        p.a_out_stride % 8 == 0 && ((uintptr_t)dst % 16) == 0 && ((uintptr_t)src % 16) == 0) {
      bool ok = true;
      TiledParams f = p;
      const bool tb64 = p.Nb % 128 != 0;      // 64 x 64 tiles (round 6)
      f.tiles_a = p.Na / 64;
      f.tiles_b = p.Nb / (tb64 ? 64 : 128);
      f.nbatch = 0;
      int64_t nblocks = f.tiles_a * f.tiles_b;
      for (int d = 0; d < r; ++d) {
        if (d == ia || d == ib) continue;
        if (istride[d] % 8 != 0 || ostride[d] % 8 != 0) ok = false;
        f.bshape[f.nbatch] = oshape[d];
        f.bin[f.nbatch] = istride[d];
        f.bout[f.nbatch] = ostride[d];
        ++f.nbatch;
        nblocks *= oshape[d];
      }
      if (ok && !getenv("TNH_PERMUTE_NO16")) {
        f.nblocks = nblocks;
        {
          // consecutive workgroups take neighbouring b-tiles in groups of up to 16 (measured on the [K][N] -> [N][K]
          // permute of a (128,)^4 bf16 tensor: 4.08 TB/s a-fastest, 4.22 / 4.27 / 4.36 / 4.75 / 4.58 / 4.28 with groups
          // of 2 / 4 / 8 / 16 / 32 / 64); TNH_PERMUTE_ORDER overrides
          const char* eo = getenv("TNH_PERMUTE_ORDER");
          int order = eo ? atoi(eo) : 16;
          if (order < 1) order = 1;
          while (order > 1 && f.tiles_b % order != 0) order >>= 1;
          f.order = order;
        }
        int64_t grid = nblocks < (int64_t(1) << 22) ? nblocks : (int64_t(1) << 22);
        if (const char* eg = getenv("TNH_PERMUTE_TILED_WGS")) {      // A/B: persistent grid of this many workgroups per CU
          const int64_t cap = (int64_t)num_cus() * atoi(eg);
          if (cap > 0 && grid > cap) grid = cap;
        }
        if (tb64)
          hipLaunchKernelGGL((permute_tiled16_kernel<1, 64>), dim3((unsigned)grid), dim3(256), 0, stream(),
                             (uint16_t*)dst, (const uint16_t*)src, f);
        else
          hipLaunchKernelGGL((permute_tiled16_kernel<1, 128>), dim3((unsigned)grid), dim3(256), 0, stream(),
                             (uint16_t*)dst, (const uint16_t*)src, f);
        TNH_LAUNCH_CHECK();
        return TNH_OK;
      }
    }
    const int TILE = (itemsize <= 4) ? 64 : 32;
    p.tiles_a = (p.Na + TILE - 1) / TILE;
    p.tiles_b = (p.Nb + TILE - 1) / TILE;
    p.nbatch = 0;
    int64_t nblocks = p.tiles_a * p.tiles_b;
    for (int d = 0; d < r; ++d) {
      if (d == ia || d == ib) continue;
      p.bshape[p.nbatch] = oshape[d];
      p.bin[p.nbatch] = istride[d];
      p.bout[p.nbatch] = ostride[d];
      ++p.nbatch;
      nblocks *= oshape[d];
    }
    {
      switch (itemsize) {
        case 1: return launch_tiled<uint8_t, 64>(dst, src, p, nblocks);
        case 2: return launch_tiled<uint16_t, 64>(dst, src, p, nblocks);
        case 4: return launch_tiled<uint32_t, 64>(dst, src, p, nblocks);
        case 8: return launch_tiled<uint64_t, 32>(dst, src, p, nblocks);
        case 16: return launch_tiled<uint4, 32>(dst, src, p, nblocks);
      }
    }
  }

  // (d) general gather: coalesced writes, strided reads.
  GatherParams p;
  p.rank = r;
  p.offset = 0;
  p.total = total;
  for (int d = 0; d < r; ++d) {
    p.shape[d] = oshape[d];
    p.stride[d] = istride[d];
  }
  return dispatch_gather<false>(dst, src, p, itemsize);
}

}  // extern "C"
