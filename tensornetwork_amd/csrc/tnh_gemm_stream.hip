// K2 (streaming part of the bf16 / f16 matrix-core path): a SMALL matrix against a very long one,
//
//   C[Ms, Nl] = S[Ms, K] * L[Nl, K]^T          (SWAP = false: the small operand is A)
//   C[Nl, Ms] = L[Nl, K] * S[Ms, K]^T          (SWAP = true:  the small operand is B)
//
// with Ms <= 192, K <= 192 and Nl in the millions: what contracting one or two small bonds of a large
// intermediate with a small tensor lowers to (D = 12: 144 x 2 985 984 x 144; the reference reaches it
// through tensordot, numpy_backend.py:35-37).  Such a product is pure HBM streaming -- 2 * Nl * (K + Ms)
// bytes against 2 * Ms * Nl * K flop -- and the tile kernels of tnh_gemm_ragged.hip spend it badly: every
// tile re-reads the small operand from L2 (more bytes than its slice of the long one) and its load, MFMA and
// store phases run back to back.  Here
//   * the small operand is staged ONCE per workgroup into LDS (zero-padded to 16 rows x 32 k) and stays
//     there; workgroups are persistent and walk the long operand in tiles of BN rows;
//   * the NEXT tile of the long operand is requested into registers before the MFMA work on the current
//     one, so the HBM latency hides behind compute + epilogue; it goes to LDS (one image, padded pitch,
//     conflict-free fragment reads) when the current tile is done with it;
//   * the output tile is re-assembled in LDS and leaves as 16-byte stores of whole row segments
//     (SWAP: the tile is one contiguous block of C when ldc == Ms).
// Two workgroups of 256 threads fit a CU (LDS: small image + one BN-row image), so one tile's stores
// overlap the other's loads.  K % 8 == 0 (zero-filled to a multiple of 32 in LDS) and 16-byte aligned rows on
// all three matrices (host-checked: anything else keeps the tile kernels), any Ms <= 192, ragged Nl (the
// remainder of < BN long rows goes to the tile kernel).
//
// Roofline: HBM.  Algorithmic bytes 2 * (Nl * K + Ms * K + Ms * Nl).
#include "tnh_gemm_nt.h"

namespace tnh {

struct StreamArgs {
  const uint16_t* S;   // small operand, [Ms][K] rows lds apart
  const uint16_t* L;   // long operand,  [Nl][K] rows ldl apart
  uint16_t* C;
  int64_t lds, ldl, ldc;
  int Ms, K;
  int ntiles;          // FULL tiles of BN long rows (the host hands a ragged remainder to the tile kernel)
};

// LDS: [ small image: msf*16 rows x PA ][ R: max(long image BN x PA, staging) ],  PA = Kp*2 + 16 bytes.
// PA / 4 = 4 * (odd) dwords, so the 16 rows a b128 fragment read touches start in 16 different bank quads.
//
// Every row start is 16-byte aligned and K % 8 == 0 (host-checked), so the loop body has no branch around a
// global load or store: the compiler can count the stores issued after the prefetch and wait for the
// prefetch alone (s_waitcnt vmcnt(N)) -- with a data-dependent store count it waits for every store of the
// previous tile to be acknowledged, which serialises the whole pipeline on the write latency.
template <int BN, bool IS_BF16, bool SWAP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_nt_stream_kernel(StreamArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int FN = BN / 16;                 // long-operand fragments per tile (every wave takes all of them)
  constexpr int MAX_MI = 3;                   // small-operand fragments per wave: 12 / 4
  constexpr int NB = BN * 24 / 256;           // 16-B chunks of the long image per thread at Kp = 192
  constexpr int NST = 6;                      // 16-B output chunks per thread at Ms = 192 (192 * BN / 8 / 256, BN = 64)
  static_assert(BN == 64, "the static trip counts assume BN = 64");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = p.K, Ms = p.Ms;
  const int Kp = (K + 31) & ~31, cpr = Kp >> 3, kch = K >> 3, ksteps = Kp >> 5;
  const int PA = Kp * 2 + 16;
  const int msf = (Ms + 15) >> 4;
  const int msp = msf * 16;
  char* sS = smem;
  char* sR = smem + msp * PA;
  // staging pitch: !SWAP rows are small-operand rows holding BN outputs; SWAP rows are long rows holding msp
  const int EP = (SWAP ? msp : BN) * 2 + 16;
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);

  // ---- small operand -> LDS, once (zero rows / zero k padding)
  for (int idx = tid; idx < msp * cpr; idx += 256) {
    const int row = idx / cpr, c = idx - row * cpr;
    uint4 v = zero4;
    if (row < Ms && c < kch) v = *(const uint4*)(p.S + (int64_t)row * p.lds + c * 8);
    *(uint4*)(sS + row * PA + c * 16) = v;
  }

  // ---- this thread's chunks of a long-operand tile: (row, chunk) is the same for every tile
  int loff[NB], lsm[NB];                       // element offset from the tile's first row / byte offset in the image
  bool lzero[NB];
#pragma unroll
  for (int it = 0; it < NB; ++it) {
    int idx = it * 256 + tid;
    if (idx >= BN * cpr) idx = BN * cpr - 1;   // duplicates of the last chunk: same address, same data
    const int row = idx / cpr, c = idx - row * cpr;
    lzero[it] = (c >= kch);
    loff[it] = row * (int)p.ldl + (lzero[it] ? 0 : c * 8);
    lsm[it] = row * PA + c * 16;
  }
  uint4 rl[NB];
  auto load_long = [&](int tile) {
    const uint16_t* base = p.L + (int64_t)tile * BN * p.ldl;
#pragma unroll
    for (int it = 0; it < NB; ++it) rl[it] = *(const uint4*)(base + loff[it]);
  };
  auto store_long = [&]() {
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      uint4 v = rl[it];
      if (lzero[it]) v = make_uint4(0u, 0u, 0u, 0u);   // (a value select: a conditional lvalue keeps rl[] in scratch)
      *(uint4*)(sR + lsm[it]) = v;
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
        const int row = idx >> 3, ch = idx & 7;
        esm[it] = row * EP + ch * 16;
        eoff[it] = row * (int)p.ldc + ch * 8;   // ldc * Ms < 2^31 is host-checked
      }
    }
  }

  const int frag_row = lane & 15, frag_chk = lane >> 4;
  const int last = p.ntiles - 1;
  int tile = blockIdx.x;                       // the host launches at most ntiles workgroups
  load_long(tile);
  store_long();
  // The image of the NEXT tile is written at the END of the loop body, right after this tile's stores were
  // issued: there the wait for the prefetch has one predecessor path (6 loads, then NST stores), so it is
  // s_waitcnt vmcnt(NST) -- at the loop head it would merge with the preheader path and wait for everything.

  for (; tile <= last; tile += (int)gridDim.x) {
    __syncthreads();                            // long image (and, first time, the small image) complete
    {
      const int next = tile + (int)gridDim.x;
      load_long(next <= last ? next : last);   // in flight during the MFMA work and the epilogue
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
    store_long();
  }
}

static bool stream_enabled() {
  static const bool on = []() { const char* e = getenv("TNH_GEMM_STREAM"); return !(e && e[0] == '0'); }();
  return on;
}

// true when the product is in this kernel's range AND long enough for the persistent grid to pay
bool gemm_bf16_stream_wanted(int out_dt, int64_t M, int64_t N, int64_t K, int64_t batch) {
  if (!stream_enabled() || batch != 1 || out_dt == TNH_F32 || K < 16 || K > 192 || K % 8 != 0) return false;
  const int64_t small = M < N ? M : N, lng = M < N ? N : M;
  // a short side of <= 64 keeps the 64 x 256 / 256 x 64 tile kernels (measured: 64 x 4e6 x 64 0.18 vs 0.26 ms)
  return small > 64 && small <= 192 && lng >= (int64_t(1) << 16);
}

template <int BN, bool SWAP>
static int launch_stream(bool is_bf16, const StreamArgs& p, size_t lds_bytes) {
  // the attribute is per KERNEL: the bf16 and f16 instantiations have the same function-pointer type, so a flag
  // inside the generic lambda below would be shared between them -- both are set here, once
  static const int attr_rc = []() -> int {
    for (const void* k : {reinterpret_cast<const void*>(gemm_nt_stream_kernel<BN, true, SWAP>),
                          reinterpret_cast<const void*>(gemm_nt_stream_kernel<BN, false, SWAP>)})
      if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 1;
    return 0;
  }();
  if (attr_rc) {
    set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the streaming GEMM");
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
  const int rc = is_bf16 ? go(gemm_nt_stream_kernel<BN, true, SWAP>) : go(gemm_nt_stream_kernel<BN, false, SWAP>);
  if (rc) return rc;
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

int gemm_bf16_ragged(int in_dt, int out_dt, int shape, int64_t M, int64_t N, int64_t K, const void* A,
                     int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t batch,
                     int64_t sA, int64_t sB, int64_t sC, const char** name);

// NT product C[M,N] = A[M,K] * B[N,K]^T with min(M, N) <= 192, K <= 192, K % 8 == 0 (half-precision in and out).
// Returns TNH_ERR_UNSUPPORTED when an alignment rule of the streaming kernel does not hold (the caller then
// takes the tile kernels).
int gemm_bf16_stream(int in_dt, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                     int64_t ldb, void* C, int64_t ldc, const char** name) {
  const bool swap = N < M;   // the small operand is B
  StreamArgs p;
  p.S = (const uint16_t*)(swap ? B : A);
  p.L = (const uint16_t*)(swap ? A : B);
  p.lds = swap ? ldb : lda;
  p.ldl = swap ? lda : ldb;
  p.C = (uint16_t*)C;
  p.ldc = ldc;
  p.Ms = (int)(swap ? N : M);
  const int64_t Nl = swap ? M : N;
  p.K = (int)K;
  constexpr int BN = 64;
  const bool aligned = K % 8 == 0 && K >= 8 && K <= 192 && p.Ms >= 1 && p.Ms <= 192 && lda % 8 == 0 && ldb % 8 == 0 &&
                       ldc % 8 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0 &&
                       (!swap || p.Ms % 8 == 0) &&
                       // 32-bit offsets inside one tile of the long operand / of C
                       (int64_t)BN * p.ldl < (int64_t(1) << 30) &&
                       (swap ? (int64_t)BN * ldc : (int64_t)p.Ms * ldc) < (int64_t(1) << 30) && Nl >= BN;
  if (!aligned) {
    set_error("gemm_bf16_stream: operands break the 16-byte alignment rules of the streaming kernel");
    return TNH_ERR_UNSUPPORTED;
  }
  const int Kp = (int)((K + 31) & ~31), PA = Kp * 2 + 16;
  const int msp = (p.Ms + 15) / 16 * 16;
  const size_t image = (size_t)BN * PA;
  const size_t staging = swap ? (size_t)BN * (msp * 2 + 16) : (size_t)msp * (BN * 2 + 16);
  const size_t lds_bytes = (size_t)msp * PA + (image > staging ? image : staging);
  TNH_REQUIRE(Nl / BN < (int64_t(1) << 30), "gemm_bf16_stream: too many tiles");
  p.ntiles = (int)(Nl / BN);
  const bool is_bf16 = (in_dt == TNH_BF16);
  int rc;
  if (swap) {
    *name = "bf16_nt_stream_64xS";
    rc = launch_stream<BN, true>(is_bf16, p, lds_bytes);
  } else {
    *name = "bf16_nt_stream_Sx64";
    rc = launch_stream<BN, false>(is_bf16, p, lds_bytes);
  }
  if (rc) return rc;
  // ragged remainder (< BN long rows): the tile kernel on the tail sub-problem -- same MFMA sequence per element
  const int64_t done = (int64_t)p.ntiles * BN, rest = Nl - done;
  if (rest > 0) {
    const char* tail_name = nullptr;
    if (swap)
      rc = gemm_bf16_ragged(in_dt, in_dt, 0, rest, N, K, (const uint16_t*)A + done * lda, lda, B, ldb,
                            (uint16_t*)C + done * ldc, ldc, 1, 0, 0, 0, &tail_name);
    else
      rc = gemm_bf16_ragged(in_dt, in_dt, 0, M, rest, K, A, lda, (const uint16_t*)B + done * ldb, ldb,
                            (uint16_t*)C + done, ldc, 1, 0, 0, 0, &tail_name);
  }
  return rc;
}

}  // namespace tnh
