// K7b: "band + spectrum slicing" SVD for large f32 matrices (round 3; tools/svd_band_model.py is the NumPy
// statement of the same algorithm, step for step).  Backs AbstractBackend.svd (abstract_backend.py:79-137; oracle
// backends/numpy/decompositions.py:21-74 = LAPACK gesdd) for min(m, n) >= 1024 where the block Jacobi of
// tnh_svd_block.hip needs 16-37 sweeps of 127 dependent rounds: this path touches the matrix a fixed number of
// times whatever its spectrum.
//
//   stage 1  A (m x n, m >= n, n % 16 == 0)  ->  Q_L^T A Q_R = B, upper triangular with 16 super-diagonals, by
//            alternating 16-wide column panels (QR) and row panels (LQ).  A panel is NOT factored column by column:
//            its 16 x 16 Gram matrix (f64) is Cholesky-factored, the thin Q is turned into Householder form by the
//            LU of Q1 - S (Ballard et al.), and the block reflector is I - V T V^T with
//            T^-1 = striu(V^T V) + diag(V^T V) / 2  (orthogonal for ANY V) -- one small kernel per panel.
//            Trailing updates are rank-16 streaming kernels (HBM / Infinity-Cache bound, f32 VALU).
//   stage 2  T = B^T B (f64, symmetric, half bandwidth 16).  All n singular values by spectrum slicing:
//            nu(sigma) = #negative pivots of LDL^T(T - sigma^2 I); a grid of n shifts, then multi-section per value.
//            A 16-lane group holds the 16 x 16 active window in registers (row per lane, column mod 16 per
//            register); pivots and pivot columns travel by DPP row broadcasts.
//   stage 3  k leading right vectors of B by inverse iteration on the same LDL^T (factor stored), left vectors
//            u = B v / s, all f64 on the band; back-transformation through the stage-1 reflectors with the same
//            rank-16 streaming kernels.
// Anything the path cannot do accurately (rank-deficient panel, clustered kept values, s_k < 1e-6 s_1) is REPORTED
// (status word) and the caller re-runs the Jacobi path: no silent loss of accuracy.
#include <math.h>
#include <stdlib.h>
#include <map>
#include <utility>
#include <type_traits>
#include "tnh_types.h"

namespace tnh {
namespace svdb {

// band width = panel width = 16 throughout (one 16-lane DPP row per shift)
constexpr int TP = 18;          // row pitch (doubles) of the rotated band image read by the LDL^T kernels

enum StatusBits { ST_PANEL = 1, ST_PIVOT = 2, ST_CLUSTER = 4, ST_RESID = 8, ST_RANGE = 16,
                  ST_FASTCOND = 1 << 20 };   // internal: a panel too ill-conditioned for the raw-panel passes (round 6)

// ------------------------------------------------------------------------------------------------ small helpers
template <int SRC>
__device__ __forceinline__ double bcast16_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x150 + SRC, 0xf, 0xf, true);   // row_newbcast:SRC (gfx90a+): every lane has a source
  hi = __builtin_amdgcn_mov_dpp(hi, 0x150 + SRC, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int SRC, bool DPP>
__device__ __forceinline__ double bcast16(double v) {
  if (DPP) return bcast16_dpp<SRC>(v);
  return __shfl(v, SRC, 16);
}
// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>)  (DPP lane selectors
// are instruction immediates, so the step number has to be a constant expression)
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}
template <int ROT>
__device__ __forceinline__ double ror16(double v) {      // lane a of each 16-lane row reads lane (a + ROT) % 16
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x120 + ROT, 0xf, 0xf, true);   // row_ror:ROT
  hi = __builtin_amdgcn_mov_dpp(hi, 0x120 + ROT, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <bool DPP>
__device__ __forceinline__ double sum16(double v) {      // every lane of the row gets the row's sum
  if (DPP) {
    v += ror16<8>(v);
    v += ror16<4>(v);
    v += ror16<2>(v);
    v += ror16<1>(v);
    return v;
  }
  v += __shfl_xor(v, 8, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 1, 16);
  return v;
}

// f32 reductions without the LDS crossbar (ds_bpermute needs a wait per step):
//   rowsum16: sum over the 16 lanes of a DPP row, every lane gets it (row_ror 8 / 4 / 2 / 1);
//   rowsx4:   sum over the four rows of the wave, lane % 16 kept (gfx950 v_permlane16_swap / v_permlane32_swap: with
//             both operands the same register the two results are the two halves exchanged, so their sum is the
//             pair sum in every lane).
template <int ROT>
__device__ __forceinline__ float ror16f(float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0x120 + ROT, 0xf, 0xf, true));
}
__device__ __forceinline__ float rowsum16(float v) {
  v += ror16f<8>(v);
  v += ror16f<4>(v);
  v += ror16f<2>(v);
  v += ror16f<1>(v);
  return v;
}
__device__ __forceinline__ float rowsx4(float v) {
  const unsigned u = __float_as_uint(v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const unsigned t = __float_as_uint(s);
  auto b = __builtin_amdgcn_permlane32_swap(t, t, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// 1 / d and 1 / sqrt(x) to full double precision from the hardware seeds (2^-26-ish) and two Newton steps: a
// fraction of the IEEE division / square-root sequences, which sit on every pivot's critical path here.
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}

// ------------------------------------------------------------------------------------------------ element types
// Round 4: the stage-1 kernels, the start of stage 3 and the back-transformation are templates over the element type
// T (float: the round-3 path, unchanged; double: f64 inputs -- the reference's default dtype).  A lane still owns FOUR
// consecutive elements (one 16-byte load for float, two for double), so the lane geometry -- 16 lanes x 4 = 64 columns
// per workgroup row -- and every index computation are the same for both.
template <typename T>
struct alignas(sizeof(T) * 4) V4 {
  T x, y, z, w;
};
template <typename T>
struct alignas(sizeof(T) * 2) V2 {
  T x, y;
};
template <typename T>
__device__ __forceinline__ V4<T> vzero() {
  return V4<T>{T(0), T(0), T(0), T(0)};
}
__device__ __forceinline__ float fmaT(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double fmaT(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ double rowsum16(double v) { return sum16<true>(v); }
__device__ __forceinline__ double rowsx4(double v) {     // sum over the four 16-lane rows of the wave, lane % 16 kept
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
// rows of a W-pass chunk (wpass_kernel stages them in LDS: 16 floats or doubles per row)
template <typename T>
constexpr int w_rc() {
  return sizeof(T) == 4 ? 256 : 128;
}

// ------------------------------------------------------------------------------------------------ stage 1: Gram
// Panel element (r, i), r = 0 .. rows-1, i = 0 .. 15:  P[r * sr + i * si].
//   column panel: sr = lda, si = 1      row panel (the LQ is the QR of the transpose): sr = 1, si = lda
// Gpart[b][i][c] = sum over the block's 256 panel rows of P(r, i) P(r, c), f64 (products of floats are exact).
template <bool ROWPANEL, typename T>
__global__ __launch_bounds__(256) void gram_kernel(const T* __restrict__ P, int64_t lda, int64_t rows,
                                                   double* __restrict__ Gpart) {
  __shared__ T tile[256][17];
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * 256;
  if (!ROWPANEL) {
    const int64_t r = r0 + tid;
    V4<T> v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = vzero<T>();
    if (r < rows) {
      const V4<T>* src = reinterpret_cast<const V4<T>*>(P + r * lda);
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = src[q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      tile[tid][4 * q + 0] = v[q].x; tile[tid][4 * q + 1] = v[q].y;
      tile[tid][4 * q + 2] = v[q].z; tile[tid][4 * q + 3] = v[q].w;
    }
  } else {
    const int64_t r = r0 + tid;
#pragma unroll
    for (int i = 0; i < 16; ++i) tile[tid][i] = (r < rows) ? P[(int64_t)i * lda + r] : T(0);
  }
  __syncthreads();
  const int i = tid >> 4, c = tid & 15;
  double acc = 0.0;
#pragma unroll 8
  for (int r = 0; r < 256; ++r) acc += (double)tile[r][i] * (double)tile[r][c];
  Gpart[(int64_t)blockIdx.x * 256 + tid] = acc;
}

// ------------------------------------------------------------------------------------------------ stage 1: factor
// The 16 x 16 algebra of a panel, from its Gram matrix G = P^T P (f64) and its top block Pt:
//   G = R^T R (Cholesky);   P_top - S R = L Ut  (LU, S_jj = -sign of the pivot candidate, so |pivot| >= |R_jj|);
//   X = Ut^-1:  V = [L; P_below X]  are the Householder vectors of the panel (Ballard et al., "Reconstructing
//   Householder vectors from TSQR"), new top block R_h = S R;
//   N = V^T V = X^T (2 G - Pt^T S R - (S R)^T Pt) X,   T^-1 = striu(N) + diag(N) / 2.
// T^-1 is handed on as it is (diagonal stored as its reciprocal): the consumers solve with it (16 dependent steps
// per thread, thread-parallel over columns) instead of this kernel inverting it (16 dependent steps on ONE panel's
// critical path).  The sequential parts (Cholesky, LU, the triangular inverse) run in wave 0 only, synchronised by
// LDS program order; the matrix products use all 256 threads.
struct FactorShared {
  double G0[16][17], G[16][17], R[16][17], Pt[16][17], W[16][17], L[16][17], Ut[16][17], X[16][17], Z[16][17],
      ZX[16][17], Ti[16][17];
  double S[16];
  int bad;
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Cholesky G = R^T R, right-looking, in registers + DPP row broadcasts (no LDS round trip per step): lane r (of each
// 16-lane row; the four rows of the wave do the same work) holds ROW r of the matrix, register c its column c.
// G is symmetric, so R(j, r) = G(r, j) / r_jj is lane r's own register j.  Row r of G stops changing at step r, so
// R(r, c) = G^(r)(r, c) / r_rr is formed after the loop from the lane's own registers and its own 1 / r_rr.
// A pivot that is not above `thresh` is replaced by it and reported (bad).
__device__ __forceinline__ void cholesky_rows(double (&Gr)[16], int r, double thresh, int& bad, double (&Rr)[16]) {
  double myrinv = 0.0, mydiag = 0.0;
  static_for<16>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    double piv = bcast16_dpp<j>(Gr[j]);
    if (!(piv > thresh)) {
      piv = thresh;
      bad = 1;
    }
    const double rinv = fast_rsqrt(piv);
    if (r == j) {
      myrinv = rinv;
      mydiag = piv * rinv;
    }
    const double rowv = (r > j) ? Gr[j] * rinv : 0.0;       // R(j, r) for the rows still being reduced
    static_for<16>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      if (c > j) Gr[c] = fma(-rowv, bcast16_dpp<c>(rowv), Gr[c]);
    });
  });
#pragma unroll
  for (int c = 0; c < 16; ++c) Rr[c] = (c > r) ? Gr[c] * myrinv : (c == r ? mydiag : 0.0);
}

// Inverse of an upper triangular matrix held row per lane (Ur = row r of U): on return Xr = row r of U^-1 (entries
// left of the diagonal are not touched: the caller masks them).  Row-oriented back substitution, last row first:
// X(r, :) = (e_r - sum_{k > r} U(r, k) X(k, :)) / U(r, r) -- U(r, k) is the lane's OWN register, the finished row k
// travels by DPP.  (The column-oriented form of rounds 3-5 broadcast the 120 entries of U instead; none of those
// broadcasts depends on the solve, the compiler issued them all up front and the kernel held 368 registers.)
__device__ __forceinline__ void upper_inverse_rows(const double (&Ur)[16], int r, double (&Xr)[16]) {
  double diag = 1.0;
  static_for<16>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    Xr[c] = (c == r) ? 1.0 : 0.0;
    if (c == r) diag = Ur[c];
  });
  const double mydinv = fast_rcp(diag);
  static_for<16>([&](auto kc) {
    constexpr int k = 15 - decltype(kc)::value;
    const double f = (r < k) ? Ur[k] : 0.0;
    static_for<16>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      if (c >= k) {
        if (r == k) Xr[c] *= mydinv;
        Xr[c] = fma(-f, bcast16_dpp<k>(Xr[c]), Xr[c]);
      }
    });
  });
}

// All 256 threads.  In: sh.G0 = sh.G (Gram, f64), sh.Pt.  Out: sh.X, sh.L (V_top), sh.Ti (T^-1, reciprocal diagonal),
// sh.R and sh.S (R_h = S R), sh.bad.
// degenerate: the panel has exactly 16 rows, so its last column has nothing below the diagonal.  LAPACK's larfg
// then returns tau = 0 (H = I, the entry keeps its sign); here: S_15 = +sign, reciprocal diagonal of T^-1 = 0
// (so that T w has a zero last component), and R matches np.linalg.qr also in its last diagonal entry.
// rel_thresh: a panel whose Gram matrix has a pivot below rel_thresh of its largest diagonal entry loses
// orthogonality in the Cholesky-QR (eps64 * cond^2): reported, the caller takes the Jacobi path.  1e-9 for f32
// input (eps64 cond^2 ~ 1e-7); the second pass of the f64 path sees G = I + O(eps64 cond^2) and uses 1e-3.
__device__ __forceinline__ void panel_factor(FactorShared& sh, bool degenerate, double rel_thresh) {
  const int tid = threadIdx.x, i = tid >> 4, c = tid & 15;
  if (tid == 0) sh.bad = 0;
  __syncthreads();
  if (tid < 64) {
    const int r = tid & 15;
    double Gr[16], Wr[16], Rr[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) Gr[k] = sh.G[r][k];
    double gmax = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) gmax = fmax(gmax, sh.G[k][k]);
    const double thresh = rel_thresh * fmax(gmax, 1e-300);
    int bad = 0;
    cholesky_rows(Gr, r, thresh, bad, Rr);
    // ---- in-place LU of Pt - S R: S_jj = -sign of the pivot candidate (so |pivot| >= R_jj)
#pragma unroll
    for (int k = 0; k < 16; ++k) Wr[k] = sh.Pt[r][k];
    static_for<16>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const double wjj = bcast16_dpp<j>(Wr[j]);
      const double rjj = bcast16_dpp<j>(Rr[j]);
      const bool noop = degenerate && j == 15;
      const double sj = ((wjj >= 0.0) != noop) ? -1.0 : 1.0;
      if (tid == 0) sh.S[j] = sj;
      const double pj = noop ? 1.0 : fma(-sj, rjj, wjj);
      const double lrj = (r > j) ? Wr[j] * fast_rcp(pj) : 0.0;
      Wr[j] = (r == j) ? pj : ((r > j) ? lrj : Wr[j]);
      const double sown = (r == j) ? sj : 0.0;                // only row j takes the S R term
      static_for<16>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if (c > j) {
          const double mine = fma(-sown, Rr[c], Wr[c]);       // lane j: Ut(j, c); other lanes: unchanged
          Wr[c] = fma(-lrj, bcast16_dpp<j>(mine), mine);      // lrj = 0 for rows <= j
        }
      });
    });
    // ---- X = Ut^-1, row r in lane r
    double x[16];
    upper_inverse_rows(Wr, r, x);
    if (tid < 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        sh.X[r][k] = (k >= r) ? x[k] : 0.0;
        sh.R[r][k] = Rr[k];
        sh.L[r][k] = (k < r) ? Wr[k] : (k == r ? 1.0 : 0.0);
      }
      if (tid == 0 && bad) sh.bad = 1;
    }
  }
  __syncthreads();
  // ---- N = X^T (2 G0 - Pt^T S R - (S R)^T Pt) X with all threads
  {
    double z = 2.0 * sh.G0[i][c];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      z -= sh.Pt[k][i] * sh.S[k] * sh.R[k][c];
      z -= sh.S[k] * sh.R[k][i] * sh.Pt[k][c];
    }
    sh.Z[i][c] = z;
  }
  __syncthreads();
  {
    double z = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) z += sh.Z[i][k] * sh.X[k][c];
    sh.ZX[i][c] = z;
  }
  __syncthreads();
  {
    double z = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) z += sh.X[k][i] * sh.ZX[k][c];
    sh.Ti[i][c] = (c > i) ? z : (c == i ? ((degenerate && i == 15) ? 0.0 : 2.0 / z) : 0.0);   // diagonal: 1 / (N_ii / 2)
  }
  __syncthreads();
}

// sum of `nparts` 16 x 16 partial Gram matrices, entry `tid`: all partials requested before the first add (clamped
// index, no branch around a load)
__device__ __forceinline__ double gram_sum(const double* __restrict__ Gpart, int nparts, int tid) {
  double g = 0.0;
  for (int b0 = 0; b0 < nparts; b0 += 16) {
    double pv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int b = b0 + q;
      pv[q] = Gpart[(int64_t)(b < nparts ? b : nparts - 1) * 256 + tid];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) g += (b0 + q < nparts) ? pv[q] : 0.0;
  }
  return g;
}

// ONE workgroup per panel: the 16 x 16 algebra and the panel's small outputs.
//   P: panel origin (see gram_kernel).  Xout: 16 x 16 f64 (V_below = P_below X).
//   Vout: rows 0 .. 15 of the panel's V (rows x 16 row-major) = V_top;  VtOut (row panels): the same, transposed.
//   Tout: 16 x 16 f64 = T^-1 (upper, reciprocal diagonal).  Blk: band block: column panel -> R_h (upper triangular);
//   row panel -> R_h^T (lower).
//   R1 (f64 path, second pass of the Cholesky-QR2): the panel in memory is Q1 = P0 R1^-1 of the original panel P0;
//   the reflector H that triangularises Q1 ([S R2; 0]) triangularises P0 = Q1 R1 to [S R2 R1; 0]: Blk = S R2 R1.
// (Round-3 measurement: the first version repeated this factorisation in every V-forming workgroup; with more than 8
// workgroups the launch took 40-60 us instead of 12.)
// cond_limit > 0 (round 6, the raw-panel passes of tnh_svd_band_fast.inc): a panel whose Cholesky factor has
// max R_jj / min R_jj above it raises ST_FASTCOND -- the passes' rounding errors are amplified by cond(panel).
template <bool ROWPANEL, typename T>
__device__ __forceinline__ void factor_body(FactorShared& sh, const T* __restrict__ P, int64_t lda, int64_t rows,
                                            const double* __restrict__ Gpart, int nparts,
                                            double* __restrict__ Xout, T* __restrict__ Vout,
                                            T* __restrict__ VtOut, int64_t vt_pitch,
                                            double* __restrict__ Tout, double* __restrict__ Blk,
                                            const double* __restrict__ R1, double rel_thresh, double cond_limit,
                                            int* __restrict__ status) {
  const int tid = threadIdx.x, i = tid >> 4, c = tid & 15;
  {
    const double g = gram_sum(Gpart, nparts, tid);
    sh.G[i][c] = g;
    sh.G0[i][c] = g;
    sh.Pt[i][c] = (double)(ROWPANEL ? P[(int64_t)c * lda + i] : P[(int64_t)i * lda + c]);
  }
  __syncthreads();
  panel_factor(sh, rows == 16, rel_thresh);
  Xout[tid] = sh.X[i][c];
  Tout[tid] = sh.Ti[i][c];
  double rh;
  if (R1 != nullptr) {
    sh.W[i][c] = R1[tid];
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += sh.R[i][k] * sh.W[k][c];      // upper x upper: zero below the diagonal
    rh = (c >= i) ? sh.S[i] * acc : 0.0;
  } else {
    rh = (c >= i) ? sh.S[i] * sh.R[i][c] : 0.0;                       // R_h = S R, upper
  }
  if (!ROWPANEL) Blk[tid] = rh;
  else Blk[c * 16 + i] = rh;                                    // transposed: lower triangular
  const T vt = (T)((i >= c) ? sh.L[i][c] : 0.0);                // V_top, unit lower
  Vout[(int64_t)i * 16 + c] = vt;
  if (ROWPANEL && VtOut != nullptr) VtOut[(int64_t)c * vt_pitch + i] = vt;
  if (tid == 0 && sh.bad) atomicOr(status, (int)ST_PANEL);
  if (cond_limit > 0.0 && tid == 0) {
    double rmax = 0.0, rmin = 1e300;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      rmax = fmax(rmax, sh.R[k][k]);
      rmin = fmin(rmin, sh.R[k][k]);
    }
    if (!(rmax <= cond_limit * rmin)) atomicOr(status, (int)ST_FASTCOND);
  }
}

template <bool ROWPANEL, typename T>
__global__ __launch_bounds__(256) void factor_kernel(const T* __restrict__ P, int64_t lda, int64_t rows,
                                                     const double* __restrict__ Gpart, int nparts,
                                                     double* __restrict__ Xout, T* __restrict__ Vout,
                                                     T* __restrict__ VtOut, int64_t vt_pitch,
                                                     double* __restrict__ Tout, double* __restrict__ Blk,
                                                     const double* __restrict__ R1, double rel_thresh,
                                                     int* __restrict__ status) {
  __shared__ FactorShared sh;
  factor_body<ROWPANEL, T>(sh, P, lda, rows, Gpart, nparts, Xout, Vout, VtOut, vt_pitch, Tout, Blk, R1, rel_thresh, 0.0,
                           status);
}

// f64 path, first pass of the Cholesky-QR2: G1 = P^T P (partials) -> R1 (upper, f64) and R1^-1.  One workgroup (one
// wave does the work).  The Gram matrix of an f64 panel carries eps64 cond^2 of error, which a single Cholesky-QR
// turns into that much backward error of the panel's factorisation (f32 input: 1e-16 cond^2 against an eps of 6e-8
// -- invisible; f64 input: visible from cond ~ 30 on).  The second pass factors Q1 = P R1^-1, whose Gram matrix is
// I + O(eps64 cond^2): its Cholesky-QR is accurate to eps64 as long as eps64 cond^2 << 1 (pivots below 1e-13 of the
// largest are reported: cond > 3e6).
__global__ __launch_bounds__(256) void chol_kernel(const double* __restrict__ Gpart, int nparts, double* __restrict__ R1,
                                                   double* __restrict__ R1inv, int* __restrict__ status) {
  __shared__ double G[16][17];
  const int tid = threadIdx.x, r = tid & 15;
  G[tid >> 4][tid & 15] = gram_sum(Gpart, nparts, tid);      // one entry per thread: all partials in flight at once
  __syncthreads();
  if (tid >= 64) return;
  double Gr[16], Rr[16], x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) Gr[k] = G[r][k];
  double gmax = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) gmax = fmax(gmax, G[k][k]);
  int bad = 0;
  cholesky_rows(Gr, r, 1e-13 * fmax(gmax, 1e-300), bad, Rr);
  upper_inverse_rows(Rr, r, x);
  if (tid < 16) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      R1[r * 16 + k] = Rr[k];
      R1inv[r * 16 + k] = (k >= r) ? x[k] : 0.0;       // lane r holds row r of the inverse
    }
    if (tid == 0 && bad) atomicOr(status, (int)ST_PANEL);
  }
}

// f64 path: Q1 = P R1^-1 in place (256 panel rows per workgroup) and the partial Gram matrices of Q1.
template <bool ROWPANEL, typename T>
__global__ __launch_bounds__(256) void scaleq_kernel(T* __restrict__ P, int64_t lda, int64_t rows,
                                                     const double* __restrict__ R1inv, double* __restrict__ Gpart) {
  __shared__ double xs[16][17];
  __shared__ T tile[256][17];
  const int tid = threadIdx.x;
  xs[tid >> 4][tid & 15] = R1inv[tid];
  const int64_t r = (int64_t)blockIdx.x * 256 + tid;
  T p[16];
#pragma unroll
  for (int l = 0; l < 16; ++l) p[l] = T(0);
  if (r < rows) {
    if (!ROWPANEL) {
      const V4<T>* src = reinterpret_cast<const V4<T>*>(P + r * lda);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const V4<T> v = src[q];
        p[4 * q] = v.x; p[4 * q + 1] = v.y; p[4 * q + 2] = v.z; p[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int l = 0; l < 16; ++l) p[l] = P[(int64_t)l * lda + r];
    }
  }
  __syncthreads();
  T qv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    double acc = 0.0;
#pragma unroll
    for (int l = 0; l <= k; ++l) acc += (double)p[l] * xs[l][k];      // R1^-1 is upper triangular
    qv[k] = (T)acc;
    tile[tid][k] = qv[k];
  }
  if (r < rows) {
    if (!ROWPANEL) {
      V4<T>* dst = reinterpret_cast<V4<T>*>(P + r * lda);
#pragma unroll
      for (int q = 0; q < 4; ++q) dst[q] = V4<T>{qv[4 * q], qv[4 * q + 1], qv[4 * q + 2], qv[4 * q + 3]};
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) P[(int64_t)k * lda + r] = qv[k];
    }
  }
  __syncthreads();
  const int i = tid >> 4, c = tid & 15;
  double acc = 0.0;
#pragma unroll 8
  for (int rr = 0; rr < 256; ++rr) acc += (double)tile[rr][i] * (double)tile[rr][c];
  Gpart[(int64_t)blockIdx.x * 256 + tid] = acc;
}

// V rows below the top block: V[r][:] = P[r][:] X (f64 accumulate), r = 16 .. rows - 1, 256 rows per workgroup.
template <bool ROWPANEL, typename T>
__device__ __forceinline__ void formv_body(double (*xs)[17], const T* __restrict__ P, int64_t lda, int64_t rows,
                                           const double* __restrict__ X, T* __restrict__ Vout,
                                           T* __restrict__ VtOut, int64_t vt_pitch, int64_t blk) {
  const int tid = threadIdx.x;
  xs[tid >> 4][tid & 15] = X[tid];
  const int64_t r = 16 + blk * 256 + tid;
  T p[16];
#pragma unroll
  for (int l = 0; l < 16; ++l) p[l] = T(0);
  if (r < rows) {
    if (!ROWPANEL) {
      const V4<T>* src = reinterpret_cast<const V4<T>*>(P + r * lda);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const V4<T> v = src[q];
        p[4 * q] = v.x; p[4 * q + 1] = v.y; p[4 * q + 2] = v.z; p[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int l = 0; l < 16; ++l) p[l] = P[(int64_t)l * lda + r];
    }
  }
  __syncthreads();
  if (r < rows) {
    T v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      double acc = 0.0;
#pragma unroll
      for (int l = 0; l < 16; ++l) acc += (double)p[l] * xs[l][k];
      v[k] = (T)acc;
    }
    V4<T>* dst = reinterpret_cast<V4<T>*>(Vout + r * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = V4<T>{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    if (ROWPANEL && VtOut != nullptr) {
#pragma unroll
      for (int k = 0; k < 16; ++k) VtOut[(int64_t)k * vt_pitch + r] = v[k];
    }
  }
}
template <bool ROWPANEL, typename T>
__global__ __launch_bounds__(256) void formv_kernel(const T* __restrict__ P, int64_t lda, int64_t rows,
                                                    const double* __restrict__ X, T* __restrict__ Vout,
                                                    T* __restrict__ VtOut, int64_t vt_pitch) {
  __shared__ double xs[16][17];
  formv_body<ROWPANEL, T>(xs, P, lda, rows, X, Vout, VtOut, vt_pitch, (int64_t)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------ rank-16 streaming
// Lane geometry shared by the streaming kernels: 256 threads = 4 waves; lane = (g = lane / 16, t = lane % 16);
// a wave covers 4 rows (g) x 64 columns (4 elements per t), a workgroup 16 rows x 64 columns per iteration.

// Wpart[chunk][i][c] = sum over the chunk's RC rows of V[r][i] C[r][c]       (W = V^T C, partial over row chunks)
// The chunk's V rows are staged once in LDS (every lane group reads whole rows: broadcast reads); the C loads of
// four 16-row steps are requested before the first FMA.
template <typename T>
constexpr int w_unroll() {
  return 4;      // C rows requested ahead per lane
}
// columns per lane of the W pass: one 16-byte load per row (4 floats, 2 doubles) -- with 4 doubles per lane the 64
// accumulators alone are 128 registers and the kernel ran at one wave per SIMD (38 us against 12 for float)
template <typename T>
constexpr int w_cpl() {
  return 16 / (int)sizeof(T);
}
template <typename T, int N>
struct alignas(sizeof(T) * N) VN {
  T e[N];
};
template <typename T>
__global__ __launch_bounds__(256) void wpass_kernel(const T* __restrict__ C, int64_t ldc, int64_t rows, int64_t nc,
                                                    const T* __restrict__ V, T* __restrict__ Wpart) {
  constexpr int RC = w_rc<T>();
  constexpr int W_UNROLL = w_unroll<T>();
  constexpr int CPL = w_cpl<T>();
  constexpr int TILE = 16 * CPL;    // columns per workgroup
  __shared__ V4<T> vs[RC][4];       // V rows of the chunk
  __shared__ T red[4][16][TILE + 1];      // [wave][i][column]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, t = lane & 15;
  const int64_t c0 = (int64_t)blockIdx.x * TILE + CPL * t;
  const int64_t rbeg = (int64_t)blockIdx.y * RC;
  const bool col_ok = c0 < nc;
  if (tid < RC) {
    const int64_t r = rbeg + tid;
    const V4<T>* vp = reinterpret_cast<const V4<T>*>(V + r * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) vs[tid][q] = (r < rows) ? vp[q] : vzero<T>();
  }
  __syncthreads();
  T acc[16][CPL];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int q = 0; q < CPL; ++q) acc[i][q] = T(0);
  auto step = [&](int it0) {
    VN<T, CPL> cv[W_UNROLL];
#pragma unroll
    for (int u = 0; u < W_UNROLL; ++u) {
      const int64_t r = rbeg + (it0 + u) * 16 + w * 4 + g;
      if (r < rows && col_ok) {
        cv[u] = *reinterpret_cast<const VN<T, CPL>*>(C + r * ldc + c0);
      } else {
#pragma unroll
        for (int q = 0; q < CPL; ++q) cv[u].e[q] = T(0);
      }
    }
#pragma unroll
    for (int u = 0; u < W_UNROLL; ++u) {
      const int lr = (it0 + u) * 16 + w * 4 + g;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const V4<T> x = vs[lr][q];
        const T ve[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < CPL; ++c) acc[4 * q + e][c] = fmaT(ve[e], cv[u].e[c], acc[4 * q + e][c]);
      }
    }
  };
  if constexpr (sizeof(T) == 4) {
    for (int it0 = 0; it0 < RC / 16; it0 += W_UNROLL) step(it0);        // (the compiler's own unrolling: 132 registers)
  } else {
#pragma unroll 1                                                        // f64: unrolled, the body hoists every LDS read
    for (int it0 = 0; it0 < RC / 16; it0 += W_UNROLL) step(it0);        // of the chunk (256 + 44 registers); rolled: 110
  }
  // rows of one wave: the four lane groups g hold the same columns
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const T v = rowsx4(acc[i][q]);
      if (g == 0) red[w][i][CPL * t + q] = v;
    }
  __syncthreads();
  // 16 x TILE outputs, CPL per thread: fixed summation order over the waves
  const int i = tid >> 4;
  const int64_t cc = (int64_t)blockIdx.x * TILE + CPL * (tid & 15);
  if (cc < nc) {
    VN<T, CPL> o;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      T sacc = T(0);
#pragma unroll
      for (int k = 0; k < 4; ++k) sacc += red[k][i][CPL * (tid & 15) + q];
      o.e[q] = sacc;
    }
    *reinterpret_cast<VN<T, CPL>*>(Wpart + ((int64_t)blockIdx.y * 16 + i) * nc + cc) = o;
  }
}

// Solves with the upper triangular T^-1 = Ti (diagonal stored as its reciprocal), per thread, in registers.
//   trans = 1:  w = T^T s  <=>  Ti^T w = s  (forward substitution)      trans = 0:  w = T s  <=>  Ti w = s  (backward)
template <typename TS, typename T>
__device__ __forceinline__ void tsolve(const TS (*Ti)[17], int trans, const T* s, T* w) {
  if (trans) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      T acc = s[i];
#pragma unroll
      for (int l = 0; l < i; ++l) acc = fmaT(-(T)Ti[l][i], w[l], acc);
      w[i] = acc * (T)Ti[i][i];
    }
  } else {
#pragma unroll
    for (int i = 15; i >= 0; --i) {
      T acc = s[i];
#pragma unroll
      for (int l = i + 1; l < 16; ++l) acc = fmaT(-(T)Ti[i][l], w[l], acc);
      w[i] = acc * (T)Ti[i][i];
    }
  }
}

// Wt[:, c] = T^T (trans = 1: H^T C) or T (trans = 0: H C) times the sum over chunks of Wpart[chunk][:, c].
// 64 columns per workgroup; the four waves take the chunks ch % 4 (their loads are in flight together), wave 0
// finishes.  Fixed summation order: deterministic.
template <typename T>
__global__ __launch_bounds__(256) void wreduce_kernel(const T* __restrict__ Wpart, int nchunks, int64_t nc,
                                                      const double* __restrict__ Tinv, int trans, T* __restrict__ Wt) {
  __shared__ T Ts[16][17];
  __shared__ T part[4][16][65];
  const int tid = threadIdx.x, cg = tid >> 6, cl = tid & 63;
  Ts[tid >> 4][tid & 15] = (T)Tinv[tid];
  const int64_t c = (int64_t)blockIdx.x * 64 + cl;
  T s[16];
#pragma unroll
  for (int l = 0; l < 16; ++l) s[l] = T(0);
  if (c < nc) {
    for (int ch = cg; ch < nchunks; ch += 4) {
#pragma unroll
      for (int l = 0; l < 16; ++l) s[l] += Wpart[((int64_t)ch * 16 + l) * nc + c];
    }
  }
#pragma unroll
  for (int l = 0; l < 16; ++l) part[cg][l][cl] = s[l];
  __syncthreads();
  if (cg == 0 && c < nc) {
#pragma unroll
    for (int l = 0; l < 16; ++l) s[l] = (part[0][l][cl] + part[1][l][cl]) + (part[2][l][cl] + part[3][l][cl]);
    T w[16];
    tsolve(Ts, trans, s, w);
#pragma unroll
    for (int i = 0; i < 16; ++i) Wt[(int64_t)i * nc + c] = w[i];
  }
}

// C[r][c] -= sum_i V[r][i] Wt[i][c]          (tiles of U_RR rows x 64 columns)
// GRAM = 1: the workgroups of the first column tile also write the partial Gram matrices (over their 128 rows) of the
//           first 16 columns of the UPDATED C -- the next column panel -- to Gpart[blockIdx.y]   (gridDim.y partials);
// GRAM = 2: the workgroups of the first row tile write the partial Gram matrices (over their 64 columns) of the first
//           16 rows of the updated C -- the next row panel -- to Gpart[blockIdx.x]                (gridDim.x partials).
//           (rows below gram_row0 only: the QR keeps its R rows at the top of the block);
// Either saves the separate pass of gram_kernel over the panel and its launch.
constexpr int U_RR = 128;
template <int GRAM, typename T>
__global__ __launch_bounds__(256) void update_kernel(T* __restrict__ C, int64_t ldc, int64_t rows, int64_t nc,
                                                     const T* __restrict__ V, const T* __restrict__ Wt,
                                                     int64_t wt_pitch, double* __restrict__ Gpart, int64_t gram_row0) {
  __shared__ T tile[(GRAM == 1) ? U_RR : 16][(GRAM == 1) ? 17 : 65];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, t = lane & 15;
  const int64_t c0 = (int64_t)blockIdx.x * 64 + 4 * t;
  const bool col_ok = c0 < nc;
  const bool gram_wg = (GRAM == 1) ? (blockIdx.x == 0) : ((GRAM == 2) ? (blockIdx.y == 0) : false);
  V4<T> wt[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
    wt[i] = col_ok ? *reinterpret_cast<const V4<T>*>(Wt + (int64_t)i * wt_pitch + c0) : vzero<T>();
  const int64_t rbeg = (int64_t)blockIdx.y * U_RR;
#pragma unroll 2
  for (int it = 0; it < U_RR / 16; ++it) {
    const int lr = it * 16 + w * 4 + g;
    const int64_t r = rbeg + lr;
    V4<T> cv = vzero<T>();
    if (r < rows && col_ok) {
      V4<T>* cp = reinterpret_cast<V4<T>*>(C + r * ldc + c0);
      cv = *cp;
      const V4<T>* vp = reinterpret_cast<const V4<T>*>(V + r * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const V4<T> x = vp[q];
        const T vv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const V4<T> ww = wt[4 * q + e];
          cv.x = fmaT(-vv[e], ww.x, cv.x);
          cv.y = fmaT(-vv[e], ww.y, cv.y);
          cv.z = fmaT(-vv[e], ww.z, cv.z);
          cv.w = fmaT(-vv[e], ww.w, cv.w);
        }
      }
      *cp = cv;
    }
    if (GRAM == 1 && gram_wg && t < 4) {      // rows above gram_row0 belong to R, not to the next panel (QR)
      const bool in = r >= gram_row0;
      tile[lr][4 * t] = in ? cv.x : T(0); tile[lr][4 * t + 1] = in ? cv.y : T(0);
      tile[lr][4 * t + 2] = in ? cv.z : T(0); tile[lr][4 * t + 3] = in ? cv.w : T(0);
    }
    if (GRAM == 2 && gram_wg && it == 0) {
      tile[lr][4 * t] = cv.x; tile[lr][4 * t + 1] = cv.y; tile[lr][4 * t + 2] = cv.z; tile[lr][4 * t + 3] = cv.w;
    }
  }
  if (GRAM != 0 && gram_wg) {       // uniform per workgroup
    __syncthreads();
    const int i = tid >> 4, c = tid & 15;
    double acc = 0.0;
    if (GRAM == 1) {
#pragma unroll 8
      for (int rr = 0; rr < U_RR; ++rr) acc += (double)tile[rr][i] * (double)tile[rr][c];
      Gpart[(int64_t)blockIdx.y * 256 + tid] = acc;
    } else {
#pragma unroll 8
      for (int cc = 0; cc < 64; ++cc) acc += (double)tile[i][cc] * (double)tile[c][cc];
      Gpart[(int64_t)blockIdx.x * 256 + tid] = acc;
    }
  }
}

// Row panel, fused:  Y = C V (rows x 16, row-local),  Z = Y T,  C -= Z V^T.   Vt: 16 x nc (V transposed);
// T comes as T^-1 (see panel_factor).  f32 only (A/B variant, TNH_SVDB_ROWFUSED=1).
// A workgroup owns 16 rows (lane = (row-in-wave g, column chunk t)) and sweeps all columns twice.
constexpr int RU_TILES = 8;   // column tiles of 64 per staging step (8 loads per lane in flight)
__global__ __launch_bounds__(256) void rowupdate_kernel(float* __restrict__ C, int64_t ldc, int64_t rows, int64_t nc,
                                                        const float* __restrict__ Vt, int64_t vt_pitch,
                                                        const double* __restrict__ T) {
  __shared__ float4 vts[RU_TILES][16][16];   // [tile][i][t]: Vt[i][c0 + 4 t ..]
  __shared__ float Ts[16][17];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, t = lane & 15;
  Ts[tid >> 4][tid & 15] = (float)T[tid];
  const int64_t r = (int64_t)blockIdx.x * 16 + w * 4 + g;
  const bool row_ok = r < rows;
  const int64_t ntile = (nc + 63) / 64;
  float y[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) y[i] = 0.f;
  // ---- sweep 1: y[i] = sum_c C[r][c] Vt[i][c] over this lane's column chunks
  for (int64_t k0 = 0; k0 < ntile; k0 += RU_TILES) {
    float4 cv[RU_TILES];
#pragma unroll
    for (int u = 0; u < RU_TILES; ++u) {
      const int64_t c = (k0 + u) * 64 + 4 * t;
      cv[u] = (row_ok && c < nc) ? *reinterpret_cast<const float4*>(C + r * ldc + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < RU_TILES; ++u) {
      const int64_t c = (k0 + u) * 64 + 4 * (tid & 15);
      vts[u][tid >> 4][tid & 15] = (c < nc) ? *reinterpret_cast<const float4*>(Vt + (int64_t)(tid >> 4) * vt_pitch + c)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < RU_TILES; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float4 v = vts[u][i][t];
        y[i] = fmaf(cv[u].x, v.x, y[i]);
        y[i] = fmaf(cv[u].y, v.y, y[i]);
        y[i] = fmaf(cv[u].z, v.z, y[i]);
        y[i] = fmaf(cv[u].w, v.w, y[i]);
      }
  }
  // combine the 16 column-chunk lanes of each row
#pragma unroll
  for (int i = 0; i < 16; ++i) y[i] = rowsum16(y[i]);
  float z[16];
  tsolve(Ts, 1, y, z);        // Z = Y T: row-wise z = T^T y
  // ---- sweep 2: C[r][c] -= sum_i z[i] Vt[i][c]
  for (int64_t k0 = 0; k0 < ntile; k0 += RU_TILES) {
    float4 cv[RU_TILES];
#pragma unroll
    for (int u = 0; u < RU_TILES; ++u) {
      const int64_t c = (k0 + u) * 64 + 4 * t;
      cv[u] = (row_ok && c < nc) ? *reinterpret_cast<const float4*>(C + r * ldc + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < RU_TILES; ++u) {
      const int64_t c = (k0 + u) * 64 + 4 * (tid & 15);
      vts[u][tid >> 4][tid & 15] = (c < nc) ? *reinterpret_cast<const float4*>(Vt + (int64_t)(tid >> 4) * vt_pitch + c)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < RU_TILES; ++u) {
      float4 o = cv[u];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float4 v = vts[u][i][t];
        o.x = fmaf(-z[i], v.x, o.x);
        o.y = fmaf(-z[i], v.y, o.y);
        o.z = fmaf(-z[i], v.z, o.z);
        o.w = fmaf(-z[i], v.w, o.w);
      }
      const int64_t c = (k0 + u) * 64 + 4 * t;
      if (row_ok && c < nc) *reinterpret_cast<float4*>(C + r * ldc + c) = o;
    }
  }
}


// Row panel in three launches (more workgroups than the fused kernel can offer: it owns whole rows):
//   ypass    Ypart[chunk][r][i] = sum over the chunk's 128 columns of C[r][c] Vt[i][c]      (Y_ROWS-row x 128-column tiles)
//   yreduce  Z[r][:] = T^T (sum over chunks of Ypart[chunk][r][:])                            (Z = Y T)
//   update_kernel(C, V := Z, Wt := Vt)                                                        (C -= Z V^T)
constexpr int Y_COLS = 128;
template <typename T>
constexpr int y_rows() {
  return sizeof(T) == 4 ? 64 : 16;      // accumulators: Y_ROWS / 16 x 16 per lane (f64: 32 rows needed 256 + 86 registers)
}
template <typename T>
__global__ __launch_bounds__(256) void ypass_kernel(const T* __restrict__ C, int64_t ldc, int64_t rows, int64_t nc,
                                                    const T* __restrict__ Vt, int64_t vt_pitch,
                                                    T* __restrict__ Ypart) {
  constexpr int YR = y_rows<T>();
  __shared__ V4<T> vts[Y_COLS / 64][16][16];    // [tile][i][t]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, t = lane & 15;
  const int64_t cbeg = (int64_t)blockIdx.x * Y_COLS;
  const int64_t rbeg = (int64_t)blockIdx.y * YR;
  // all C loads of the tile set first: YR / 16 row steps x 2 column tiles
  V4<T> cv[YR / 16][Y_COLS / 64];
#pragma unroll
  for (int it = 0; it < YR / 16; ++it) {
    const int64_t r = rbeg + it * 16 + w * 4 + g;
#pragma unroll
    for (int u = 0; u < Y_COLS / 64; ++u) {
      const int64_t c = cbeg + u * 64 + 4 * t;
      cv[it][u] = (r < rows && c < nc) ? *reinterpret_cast<const V4<T>*>(C + r * ldc + c) : vzero<T>();
    }
  }
#pragma unroll
  for (int u = 0; u < Y_COLS / 64; ++u) {
    const int64_t c = cbeg + u * 64 + 4 * (tid & 15);
    vts[u][tid >> 4][tid & 15] = (c < nc) ? *reinterpret_cast<const V4<T>*>(Vt + (int64_t)(tid >> 4) * vt_pitch + c)
                                          : vzero<T>();
  }
  __syncthreads();
  T y[YR / 16][16];
#pragma unroll
  for (int it = 0; it < YR / 16; ++it)
#pragma unroll
    for (int i = 0; i < 16; ++i) y[it][i] = T(0);
#pragma unroll
  for (int u = 0; u < Y_COLS / 64; ++u)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const V4<T> v = vts[u][i][t];
#pragma unroll
      for (int it = 0; it < YR / 16; ++it) {
        y[it][i] = fmaT(cv[it][u].x, v.x, y[it][i]);
        y[it][i] = fmaT(cv[it][u].y, v.y, y[it][i]);
        y[it][i] = fmaT(cv[it][u].z, v.z, y[it][i]);
        y[it][i] = fmaT(cv[it][u].w, v.w, y[it][i]);
      }
    }
  // combine the 16 column-chunk lanes of each row; lane t keeps entry i == t
#pragma unroll
  for (int it = 0; it < YR / 16; ++it) {
    T mine = T(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const T v = rowsum16(y[it][i]);
      if (t == i) mine = v;
    }
    const int64_t r = rbeg + it * 16 + w * 4 + g;
    if (r < rows) Ypart[((int64_t)blockIdx.x * rows + r) * 16 + t] = mine;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void yreduce_kernel(const T* __restrict__ Ypart, int nchunks, int64_t rows,
                                                      const double* __restrict__ Tinv, T* __restrict__ Z) {
  __shared__ T Ts[16][17];
  __shared__ T part[4][64][17];
  const int tid = threadIdx.x, cg = tid >> 6, rl = tid & 63;
  Ts[tid >> 4][tid & 15] = (T)Tinv[tid];
  const int64_t r = (int64_t)blockIdx.x * 64 + rl;
  T s[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = T(0);
  if (r < rows) {
    for (int ch = cg; ch < nchunks; ch += 4) {
      const V4<T>* yp = reinterpret_cast<const V4<T>*>(Ypart + ((int64_t)ch * rows + r) * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const V4<T> v = yp[q];
        s[4 * q] += v.x; s[4 * q + 1] += v.y; s[4 * q + 2] += v.z; s[4 * q + 3] += v.w;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) part[cg][rl][i] = s[i];
  __syncthreads();
  if (cg == 0 && r < rows) {
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = (part[0][rl][i] + part[1][rl][i]) + (part[2][rl][i] + part[3][rl][i]);
    T z[16];
    tsolve(Ts, 1, s, z);
    V4<T>* zp = reinterpret_cast<V4<T>*>(Z + r * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) zp[q] = V4<T>{z[4 * q], z[4 * q + 1], z[4 * q + 2], z[4 * q + 3]};
  }
}

// ------------------------------------------------------------------------------------------------ stage 2: band
// Bd[i][d] = B(i, i + d), d = 0 .. 16 (f64), from the panels' 16 x 16 blocks: Dblk[p] = diagonal block (upper
// triangular), Eblk[p] = super-diagonal block (lower triangular).
__global__ __launch_bounds__(256) void band_kernel(const double* __restrict__ Dblk, const double* __restrict__ Eblk,
                                                   int64_t n, double* __restrict__ Bd) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * 17) return;
  const int64_t i = e / 17;
  const int d = (int)(e % 17);
  const int64_t col = i + d;
  double v = 0.0;
  if (col < n) {
    const int64_t p = i / 16;
    const int a = (int)(i % 16);
    if (col < 16 * (p + 1)) v = Dblk[p * 256 + a * 16 + (col - 16 * p)];
    else v = Eblk[p * 256 + a * 16 + (col - 16 * (p + 1))];
  }
  Bd[e] = v;
}

// Tb[i][d] = (B^T B)(i, i + d), d = 0 .. 16
__global__ __launch_bounds__(256) void tband_kernel(const double* __restrict__ Bd, int64_t n, double* __restrict__ Tb) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * 17) return;
  const int64_t i = e / 17;
  const int d = (int)(e % 17);
  const int64_t c = i + d;
  double s = 0.0;
  if (c < n) {
    int64_t r0 = c - 16;
    if (r0 < 0) r0 = 0;
    for (int64_t r = r0; r <= i; ++r) s += Bd[r * 17 + (i - r)] * Bd[r * 17 + (c - r)];
  }
  Tb[e] = s;
}

// Rotated image read by the LDL^T kernels: Trot[i][c], c = 0 .. 15 = T(i, col) for the column col in [i - 15, i]
// with col % 16 == c (c == i % 16: the diagonal); Trot[i][16] = T(i, i - 16).  Rows n .. n + 31: zero.
__global__ __launch_bounds__(256) void trot_kernel(const double* __restrict__ Tb, int64_t n, double* __restrict__ Trot) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (n + 32) * TP) return;
  const int64_t i = e / TP;
  const int c = (int)(e % TP);
  double v = 0.0;
  if (i < n && c <= 16) {
    const int64_t delta = (c == 16) ? 16 : ((i - c) & 15);
    const int64_t col = i - delta;
    if (col >= 0) v = Tb[col * 17 + delta];
  }
  Trot[e] = v;
}

// scal[0] = sigma_max bound (Gershgorin on T, + margin), scal[1] = pivmin
__global__ __launch_bounds__(1024) void smax_kernel(const double* __restrict__ Tb, int64_t n, double* __restrict__ scal) {
  __shared__ double red[1024];
  double best = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    double s = 0.0;
    for (int d = 0; d <= 16; ++d) s += fabs(Tb[i * 17 + d]);
    for (int d = 1; d <= 16; ++d)
      if (i - d >= 0) s += fabs(Tb[(i - d) * 17 + d]);
    best = fmax(best, s);
  }
  red[threadIdx.x] = best;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double t = red[0];
    scal[0] = sqrt(t) * (1.0 + 1e-9) + 1e-300;
    scal[1] = fmax(t * 2.3e-16, 1e-290);
  }
}

// ------------------------------------------------------------------------------------------------ stage 2: LDL^T
// One 16-lane group per shift (4 per wave).  Lane a holds the active row i (i % 16 == a) of the window
// j .. j + 15; register c holds its entry in the column col (col % 16 == c) -- only the lower triangle and the
// diagonal are ever read.  Step j: pivot d from lane j % 16; that lane takes over row j + 16 (prefetched once
// per 16 steps); every row i > j subtracts l_i * (pivot column entry of row c) from its register c.
//   COUNT: counts[s] = # pivots < 0 = # eigenvalues of T below shift^2, flags[s] = 1 if a pivot was tiny.
//   STORE: Lc[s][j][e] = L(j + 1 + e, j), Dd[s][j] = d_j for the inverse iteration.
template <bool STORE, bool DPP>
__global__ __launch_bounds__(64) void ldl_kernel(const double* __restrict__ Trot, int64_t n,
                                                 const double* __restrict__ shifts, int64_t ns,
                                                 const double* __restrict__ scal, double tau_rel,
                                                 int* __restrict__ counts, int* __restrict__ flags,
                                                 double* __restrict__ Lc, double* __restrict__ Dd) {
  const int lane = threadIdx.x, grp = lane >> 4, l16 = lane & 15;
  int64_t sidx = (int64_t)blockIdx.x * 4 + grp;
  const bool valid = sidx < ns;
  if (!valid) sidx = ns - 1;
  const double sg = shifts[sidx];
  const double s2 = sg * sg;
  const double pivmin = scal[1];
  // A small pivot d injects an error of about eps |l_i colv_c| = eps colv^2 / |d| into the window; the count at sigma
  // then belongs to a matrix that is off by that much, i.e. to sigma off by error / (2 sigma).  Flag the count (the
  // bracket update skips flagged points) when that exceeds a fifth of the target resolution tau = tau_rel sigma_max
  // (2^-23 in the f32 path's rounds).
  const double tau = scal[0] * tau_rel;
  const double flagbound = fmax(sg, tau) * tau * (0.4 / 2.3e-16);
  double S[16];
  {
    const double* row = Trot + (int64_t)l16 * TP;
#pragma unroll
    for (int c = 0; c < 16; ++c) S[c] = row[c] - ((c == l16) ? s2 : 0.0);
  }
  int cnt = 0, flag = 0;
  for (int64_t j0 = 0; j0 < n; j0 += 16) {
    // the row that enters this lane during the block: j0 + 16 + l16
    const int64_t rn = j0 + 16 + l16;
    double nr[17];
    {
      const double* row = Trot + rn * TP;
#pragma unroll
      for (int c = 0; c < 17; ++c) nr[c] = row[c];
    }
    const double dsub = (rn < n) ? s2 : -1.0;     // padding rows: diagonal 0 - (-1) = 1 > 0, no coupling
    static_for<16>([&](auto jc) {
      constexpr int jj = decltype(jc)::value;
      double d = bcast16<jj, DPP>(S[jj]);
      if (fabs(d) < pivmin) d = -pivmin;
      cnt += (d < 0.0) ? 1 : 0;
      const double rinv = fast_rcp(d);
      const bool me = (l16 == jj);
      const double colv = me ? nr[16] : S[jj];
      if (me) {
#pragma unroll
        for (int c = 0; c < 16; ++c) S[c] = nr[c];
        S[jj] = nr[jj] - dsub;
      }
      const double l = colv * rinv;
      if (!STORE && fabs(l * colv) > flagbound) flag = 1;
      if (STORE) {
        const int64_t j = j0 + jj;
        Lc[(sidx * n + j) * 16 + ((l16 - jj - 1) & 15)] = l;
        if (me) Dd[sidx * n + j] = d;
      }
      static_for<16>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        const double bc = bcast16<c, DPP>(colv);
        S[c] = fma(-l, bc, S[c]);
      });
    });
  }
  if (!STORE) {
    flag = (sum16<DPP>((double)flag) > 0.0) ? 1 : 0;
    if (valid && l16 == 0) {
      counts[sidx] = cnt;
      flags[sidx] = flag;
    }
  }
}

// Throughput form of the count: ONE shift per lane, the whole 16 x 16 window (lower triangle: 136 values) in that
// lane's registers, slot [i % 16][col % 16] for i >= col.  No cross-lane traffic; the rows of T entering the window
// are wave-uniform (scalar loads).  136 f64 FMAs per pivot for 64 shifts, against 16 x 16 FMAs + 34 DPP moves for 4
// shifts in ldl_kernel -- but a wave needs n * ~700 cycles whatever the number of shifts, so this kernel is for
// rounds with tens of thousands of shifts (the grid and the multi-section rounds), ldl_kernel for the factor.
template <bool FLAGS>      // FLAGS = false: the uniform grid round (bracket_init only reads the counts)
__global__ __launch_bounds__(64) void sturm_lane_kernel(const double* __restrict__ Trot, int64_t n,
                                                        const double* __restrict__ shifts, int64_t ns,
                                                        const double* __restrict__ scal, double tau_rel,
                                                        int* __restrict__ counts, int* __restrict__ flags) {
  int64_t sidx = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const bool valid = sidx < ns;
  if (!valid) sidx = ns - 1;
  const double sg = shifts[sidx];
  const double s2 = sg * sg;
  const double pivmin = scal[1];
  const double tau = scal[0] * tau_rel;
  const double flagbound = fmax(sg, tau) * tau * (0.4 / 2.3e-16);     // see ldl_kernel
  double W[16][16];
  static_for<16>([&](auto ac) {
    constexpr int a = decltype(ac)::value;
    static_for<16>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      if (c <= a) W[a][c] = Trot[a * TP + c] - ((c == a) ? s2 : 0.0);
    });
  });
  int cnt = 0;
  double lmax2d = 0.0;        // max over the pivots of l_max^2 |d| = the largest |l colv|
  // the row entering the window at step j is j + 16; it is requested one step ahead (scalar loads: the wait would
  // otherwise sit in front of every pivot)
  double nrow[17];
#pragma unroll
  for (int c = 0; c < 17; ++c) nrow[c] = Trot[16 * TP + c];
  for (int64_t j0 = 0; j0 < n; j0 += 16) {
    static_for<16>([&](auto jc) {
      constexpr int jj = decltype(jc)::value;
      const int64_t rn = j0 + jj + 16;
      const double* __restrict__ ahead = Trot + (rn + 1) * TP;       // wave-uniform
      double nxt[17];
#pragma unroll
      for (int c = 0; c < 17; ++c) nxt[c] = ahead[c];
      const double dsub = (rn < n) ? s2 : -1.0;
      double d = W[jj][jj];
      if (fabs(d) < pivmin) d = -pivmin;
      cnt += (d < 0.0) ? 1 : 0;
      const double rinv = fast_rcp(d);
      // Rows j + 1 .. j + 15: their pivot-column entries W[.][jj] stay untouched during the step, so they serve as
      // colv directly and l_r is a temporary.  The entering row j + 16 goes last and takes its start values straight
      // from the scalar registers: the window never holds more than its 136 entries.
      double lmax = 0.0;
      static_for<15>([&](auto rc) {
        constexpr int r = decltype(rc)::value + 1;
        constexpr int a = (jj + r) & 15;
        const double lr = W[a][jj] * rinv;
        if (FLAGS) lmax = fmax(lmax, fabs(lr));
        static_for<r>([&](auto qc) {
          constexpr int q = decltype(qc)::value + 1;
          constexpr int b = (jj + q) & 15;
          W[a][b] = fma(-lr, W[b][jj], W[a][b]);
        });
      });
      {
        const double l16 = nrow[16] * rinv;
        if (FLAGS) lmax = fmax(lmax, fabs(l16));
        static_for<15>([&](auto qc) {
          constexpr int q = decltype(qc)::value + 1;
          constexpr int b = (jj + q) & 15;
          W[jj][b] = fma(-l16, W[b][jj], nrow[b]);
        });
        W[jj][jj] = fma(-l16, nrow[16], nrow[jj] - dsub);
      }
      if (FLAGS) lmax2d = fmax(lmax2d, lmax * lmax * fabs(d));
#pragma unroll
      for (int c = 0; c < 17; ++c) nrow[c] = nxt[c];
    });
  }
  if (valid) {
    counts[sidx] = cnt;
    flags[sidx] = (FLAGS && lmax2d > flagbound) ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------------ brackets
// Values are numbered ascending: q = 0 .. n - 1 (singular value number i in descending order is q = n - 1 - i).
// grid[t] = smax (t + 1) / ng
__global__ __launch_bounds__(256) void grid_kernel(const double* __restrict__ scal, int64_t ng, double* __restrict__ shifts) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < ng) shifts[t] = scal[0] * (double)(t + 1) / (double)ng;
}

// bracket of value q from the grid counts: smallest t with counts[t] > q  ->  (grid[t - 1], grid[t]]
__global__ __launch_bounds__(256) void bracket_init_kernel(const double* __restrict__ scal, const int* __restrict__ counts,
                                                           int64_t ng, int64_t n, double* __restrict__ lo,
                                                           double* __restrict__ hi) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= n) return;
  int64_t a = 0, b = ng;                  // first t in [0, ng) with counts[t] > q; ng if none
  while (a < b) {
    const int64_t mid = (a + b) >> 1;
    if (counts[mid] > q) b = mid; else a = mid + 1;
  }
  const double smax = scal[0];
  if (a >= ng) a = ng - 1;
  lo[q] = (a == 0) ? 0.0 : smax * (double)a / (double)ng;
  hi[q] = smax * (double)(a + 1) / (double)ng;
}

// P section points per value of the range [q0, q0 + nq): shifts[(q - q0) * P + t] = lo + (hi - lo) f_t,
// f_t = (t + 1 + skew) / (P + 1 + 2 skew)  (skew != 0 keeps a re-tried point off the previous one)
__global__ __launch_bounds__(256) void section_kernel(const double* __restrict__ lo, const double* __restrict__ hi,
                                                      int64_t q0, int64_t nq, int P, double skew,
                                                      double* __restrict__ shifts) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= nq * P) return;
  const int64_t q = q0 + e / P;
  const int t = (int)(e % P);
  const double f = ((double)(t + 1) + skew) / ((double)(P + 1) + 2.0 * skew);
  shifts[e] = lo[q] + (hi[q] - lo[q]) * f;
}

// new bracket of value q from its P counts (flagged counts are ignored)
__global__ __launch_bounds__(256) void bracket_update_kernel(double* __restrict__ lo, double* __restrict__ hi, int64_t q0,
                                                             int64_t nq, int P, const double* __restrict__ shifts,
                                                             const int* __restrict__ counts,
                                                             const int* __restrict__ flags) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= nq) return;
  const int64_t q = q0 + e;
  double l = lo[q], h = hi[q];
  for (int t = 0; t < P; ++t) {
    if (flags[e * P + t]) continue;
    const double x = shifts[e * P + t];
    if (counts[e * P + t] > q) {      // more than q values below x: value q is below x
      if (x < h) h = x;
    } else {
      if (x > l) l = x;
    }
  }
  if (l > h) l = h;
  lo[q] = l;
  hi[q] = h;
}

// S_out[i] (descending, f32) = midpoint of the bracket of q = n - 1 - i
template <typename T>
__global__ __launch_bounds__(256) void values_out_kernel(const double* __restrict__ lo, const double* __restrict__ hi,
                                                         int64_t n, T* __restrict__ S) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t q = n - 1 - i;
  S[i] = (T)(0.5 * (lo[q] + hi[q]));
}

// the k kept values again, from their REFINED brackets (2^-32 sigma_max instead of the 20 bits every value gets)
// (f64 path: `rayleigh` != NULL holds |B x_v| of the kept vectors -- the Rayleigh quotient of the band, accurate to
// the SQUARE of the vector's error, i.e. to eps64 where the bracket is good to 2^-44 sigma_max at best.)
template <typename T>
__global__ __launch_bounds__(256) void values_kept_kernel(const double* __restrict__ lo, const double* __restrict__ hi,
                                                          const double* __restrict__ rayleigh, int64_t n, int64_t k,
                                                          T* __restrict__ S) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= k) return;
  const int64_t q = n - 1 - i;
  S[i] = (T)(rayleigh ? rayleigh[i] : 0.5 * (lo[q] + hi[q]));
}

// shifts of the inverse iteration: vector v (0 = largest) uses the midpoint of its refined bracket
// (shifts[k + v] = half the bracket's width: two values whose brackets overlap cannot be told apart -- see
// cluster_mgs_kernel)
// push_rel > 0 (f64 path): a value with a neighbour it cannot be told apart from (closer than push_rel sigma_max, or
// overlapping brackets) is shifted UP by push_rel sigma_max.  Why: the f64 brackets are refined to the resolution of
// the LDL^T itself (2^-44), where rounding has split an exact multiplet into copies ~1e-13 sigma_max apart; a shift
// that happens to sit 1000x closer to one copy makes EVERY start vector converge to that copy's vector (measured on
// the real embedding of a complex128 matrix: collapsed pairs, status 12).  From push_rel away all copies are
// amplified alike and different starts give independent vectors of the eigenspace, which cluster_mgs_kernel
// orthonormalises; the neighbours outside the multiplet are >> push_rel away, so convergence is unaffected.
// The smallest kept value is also checked against the path's range here (range_rel sigma_max): a call that keeps
// values below it is going to be refused by uv_init_kernel anyway, and knowing it BEFORE the vector stages lets
// cluster_mgs_kernel step aside -- numerically low-rank inputs (two-site DMRG splits) put hundreds of such values into
// one "cluster", whose serial Gram-Schmidt took 250 ms per refused call (measured, round 4).
__global__ __launch_bounds__(256) void vshift_kernel(const double* __restrict__ lo, const double* __restrict__ hi,
                                                     const double* __restrict__ scal, int64_t n, int64_t k,
                                                     double push_rel, double range_rel, double* __restrict__ shifts,
                                                     int* __restrict__ status) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= k) return;
  const int64_t q = n - 1 - v;
  const double mid = 0.5 * (lo[q] + hi[q]), hw = 0.5 * (hi[q] - lo[q]);
  if (v == k - 1 && !(mid > range_rel * scal[0])) atomicOr(status, (int)ST_RANGE);
  double sh = mid;
  if (push_rel > 0.0) {
    const double tol = push_rel * scal[0];
    bool close = false;
    if (q + 1 < n) close = close || (0.5 * (lo[q + 1] + hi[q + 1]) - mid <= tol + hw + 0.5 * (hi[q + 1] - lo[q + 1]));
    if (q >= 1) close = close || (mid - 0.5 * (lo[q - 1] + hi[q - 1]) <= tol + hw + 0.5 * (hi[q - 1] - lo[q - 1]));
    if (close) sh = mid + tol;
  }
  shifts[v] = sh;
  shifts[k + v] = hw;
}

// ------------------------------------------------------------------------------------------------ stage 3: solves
__device__ __forceinline__ double start_value(int64_t v, int64_t i) {
  uint64_t x = (uint64_t)(v + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)(i + 1) * 0xC2B2AE3D27D4EB4Full;
  x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; x *= 0x94D049BB133111EBull; x ^= x >> 29;
  return (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
}

// Inverse iteration with the stored factor: `iters` times x <- (L D L^T)^-1 x, normalised.  One 16-lane group per
// vector; X[v][n] in / out (the start vector is generated here).
template <bool DPP>
__global__ __launch_bounds__(64) void solve_kernel(const double* __restrict__ Lc, const double* __restrict__ Dd,
                                                   int64_t n, int64_t k, int iters, double* __restrict__ X) {
  const int lane = threadIdx.x, grp = lane >> 4, l16 = lane & 15;
  int64_t v = (int64_t)blockIdx.x * 4 + grp;
  const bool valid = v < k;
  if (!valid) v = k - 1;
  const double* L = Lc + v * n * 16;
  const double* D = Dd + v * n;
  double* x = X + v * n;
  if (valid)
    for (int64_t i = l16; i < n; i += 16) x[i] = start_value(v, i);
  double scale = 1.0;
  for (int itn = 0; itn < iters; ++itn) {
    // ---- forward: L y = scale * x (column oriented).  Lane a carries the pending value of entry i, i % 16 == a.
    double cur = x[l16] * scale;
    // (round 6) the factor columns of the NEXT block of 16 steps are requested before the current block's chain of
    // broadcasts starts: one un-prefetched load per block was ~130 of the 184 cycles a step took (64 waves on the
    // whole chip: nothing else hides it)
    double lvn[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) lvn[jj] = L[(int64_t)jj * 16 + ((l16 - jj - 1) & 15)];
    double nbn = (16 + l16 < n) ? x[16 + l16] : 0.0;
    for (int64_t j0 = 0; j0 < n; j0 += 16) {
      const double nb = nbn * scale;
      double lv[16];
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) lv[jj] = lvn[jj];
      {
        const int64_t jn = (j0 + 16 < n) ? j0 + 16 : j0;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) lvn[jj] = L[(jn + jj) * 16 + ((l16 - jj - 1) & 15)];
        const int64_t inext2 = j0 + 32 + l16;
        nbn = (inext2 < n) ? x[inext2] : 0.0;
      }
      static_for<16>([&](auto jc) {
        constexpr int jj = decltype(jc)::value;
        const double yj = bcast16<jj, DPP>(cur);
        if (l16 == jj) {
          if (valid) x[j0 + jj] = yj;
          cur = nb;
        }
        // entry i = row of this lane; rows beyond n have l = 0 (padding columns of the factor)
        cur = fma(-lv[jj], yj, cur);
      });
    }
    // ---- backward: L^T z = D^-1 y (row oriented): z_j = y_j / d_j - sum_e L(j+1+e, j) z_(j+1+e)
    double nrm = 0.0;
    double zc = 0.0;      // lane a: z_i for the window entry i (i % 16 == a), zero beyond n
    double lvb[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) lvb[jj] = L[(n - 16 + jj) * 16 + ((l16 - jj - 1) & 15)];
    double xb = x[n - 16 + l16], db = D[n - 16 + l16];
    for (int64_t j0 = n - 16; j0 >= 0; j0 -= 16) {
      double lv[16];
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) lv[jj] = lvb[jj];
      const double yd = xb / db;
      {
        const int64_t jp = (j0 >= 16) ? j0 - 16 : j0;       // the block below, requested before this block's chain
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) lvb[jj] = L[(jp + jj) * 16 + ((l16 - jj - 1) & 15)];
        xb = x[jp + l16];
        db = D[jp + l16];
      }
      static_for<16>([&](auto jc) {
        constexpr int jj = 15 - decltype(jc)::value;
        // the whole window j + 1 .. j + 16 contributes; lane jj still holds z_(j+16) (band offset 15)
        const double term = sum16<DPP>(lv[jj] * zc);
        const double yj = bcast16<jj, DPP>(yd);
        const double zj = yj - term;
        if (l16 == jj) {
          zc = zj;
          nrm = fma(zj, zj, nrm);
          if (valid) x[j0 + jj] = zj;
        }
      });
    }
    nrm = sum16<DPP>(nrm);
    scale = 1.0 / sqrt(fmax(nrm, 1e-300));
  }
  if (valid)
    for (int64_t i = l16; i < n; i += 16) x[i] *= scale;
}

// Kept values that the f32 input cannot tell apart (closer than ctol * sigma_max: exact multiplets of symmetric
// states, the doubled spectrum of a complex matrix's real embedding) share one invariant subspace; inverse iteration
// from different start vectors returns independent but not orthogonal vectors of it.  Modified Gram-Schmidt inside
// each such run (vectors are sorted by value, so a cluster is a contiguous range) makes them an orthonormal basis of
// that subspace -- all the SVD defines there.  One workgroup walks the k vectors in order; a vector without close
// predecessors costs one comparison.  A vector that collapses (norm < 1e-3 after the projections: the iteration
// returned a dependent vector) raises ST_CLUSTER and the caller takes the Jacobi path.
__global__ __launch_bounds__(1024) void cluster_mgs_kernel(double* __restrict__ X, const double* __restrict__ shifts,
                                                           const double* __restrict__ scal, int64_t n, int64_t k,
                                                           double ctol, int* __restrict__ status) {
  __shared__ double red[16];
  __shared__ double bc;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const double tol = ctol * scal[0];
  if (*status != 0) return;        // the call is already refused (range / panel): nothing here will be used
  auto block_sum = [&](double v) -> double {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) s += red[i];
      bc = s;
    }
    __syncthreads();
    return bc;
  };
  for (int64_t v = 1; v < k; ++v) {
    // close = closer than tol OR brackets that overlap (round 4: a bracket whose section points were all flagged stays
    // wider than tol; two members of an exact multiplet then sit at different midpoints of overlapping brackets)
    int64_t first = v;
    while (first > 0 && shifts[first - 1] - shifts[v] <= tol + shifts[k + first - 1] + shifts[k + v]) --first;   // uniform
    // (f64 path: every member of such a run was pushed up by the same tol in vshift_kernel, the ends of a chain may
    // differ by it -- the comparison above is on the pushed values, a chain's neighbours still differ by <= tol + widths)
    if (first == v) continue;
    if (v - first >= 256) {           // a run this long is a numerically rank-deficient block, not a multiplet: refuse
      if (tid == 0) atomicOr(status, (int)ST_CLUSTER);      // (the serial Gram-Schmidt is quadratic in the run's length)
      return;
    }
    double* xv = X + v * n;
    for (int pass = 0; pass < 2; ++pass)            // "twice is enough"
      for (int64_t u = first; u < v; ++u) {
        const double* xu = X + u * n;
        double d = 0.0;
        for (int64_t i = tid; i < n; i += 1024) d = fma(xu[i], xv[i], d);
        d = block_sum(d);
        for (int64_t i = tid; i < n; i += 1024) xv[i] = fma(-d, xu[i], xv[i]);
        __syncthreads();
      }
    double nn = 0.0;
    for (int64_t i = tid; i < n; i += 1024) nn = fma(xv[i], xv[i], nn);
    nn = block_sum(nn);
    if (!(nn > 1e-6)) {
      if (tid == 0) atomicOr(status, (int)ST_CLUSTER);
      nn = 1.0;
    }
    const double sc = 1.0 / sqrt(nn);
    for (int64_t i = tid; i < n; i += 1024) xv[i] *= sc;
    __syncthreads();
  }
}

// Checks on the band (status bits) and the start of the back-transformation:
//   Vv[i][v] = x_v[i]  (n x k),   Uu[i][v] = (B x_v)[i] / s_v  (rows < n; rows n .. m-1 zero)
// One workgroup per vector.  f64 path (rayleigh != NULL): s_v is replaced by |B x_v| -- the Rayleigh quotient of the
// pair on the band (u is then a unit vector by construction); the |u| = 1 check becomes a check of that value against
// the bracket it came from, at the bracket's accuracy (resid_tol).
template <typename T>
__global__ __launch_bounds__(256) void uv_init_kernel(const double* __restrict__ Bd, const double* __restrict__ X,
                                                      const double* __restrict__ shifts, const double* __restrict__ scal,
                                                      int64_t m, int64_t n, int64_t k, T* __restrict__ Uu,
                                                      T* __restrict__ Vv, double* __restrict__ rayleigh,
                                                      double* __restrict__ Ub, double resid_tol, double orth_tol,
                                                      double range_tol, int* __restrict__ status) {
  __shared__ double red[256];
  __shared__ double sc_inv;
  const int64_t v = blockIdx.x;
  const double* x = X + v * n;
  const double sv = shifts[v];
  double un = 0.0, dot1 = 0.0, dot2 = 0.0;
  // pass 1: |B x|^2 and the neighbour dot products
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    double u = 0.0;
#pragma unroll
    for (int d = 0; d <= 16; ++d)
      if (i + d < n) u += Bd[i * 17 + d] * x[i + d];
    un = fma(u, u, un);
    if (v + 1 < k) dot1 = fma(x[i], X[(v + 1) * n + i], dot1);
    if (v + 2 < k) dot2 = fma(x[i], X[(v + 2) * n + i], dot2);
  }
  double vals[3] = {un, dot1, dot2};
  double out[3];
  for (int w = 0; w < 3; ++w) {
    __syncthreads();
    red[threadIdx.x] = vals[w];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    out[w] = red[0];
  }
  if (threadIdx.x == 0) {
    const double bx = sqrt(out[0]);                 // |B x_v|
    const double s_used = rayleigh ? bx : sv;
    if (rayleigh) rayleigh[v] = bx;
    sc_inv = (s_used > 0.0) ? 1.0 / s_used : 0.0;
    int st = 0;
    // (s_v, x_v) is a singular pair of the band iff |B x_v| = s_v
    if (!(fabs(bx - sv) <= resid_tol * fmax(sv, 1e-300))) st |= ST_RESID;
    if (!(fabs(out[1]) < orth_tol) || !(fabs(out[2]) < orth_tol)) st |= ST_CLUSTER;
    if (!(sv > range_tol * scal[0])) st |= ST_RANGE;
    if (st) atomicOr(status, st);
  }
  __syncthreads();
  const double inv = sc_inv;
  for (int64_t i = threadIdx.x; i < m; i += 256) {
    double u = 0.0;
    if (i < n) {
#pragma unroll
      for (int d = 0; d <= 16; ++d)
        if (i + d < n) u += Bd[i * 17 + d] * x[i + d];
      u *= inv;
      Vv[i * k + v] = (T)x[i];
      if (Ub) Ub[v * n + i] = u;        // f64 path: the left vectors on the band go through a Newton-Schulz step first
    }
    if (!Ub) Uu[i * k + v] = (T)u;
  }
}

// f64 path: Uu[i][v] = Ub[v][i] (rows < n), 0 below
template <typename T>
__global__ __launch_bounds__(256) void ub_out_kernel(const double* __restrict__ Ub, int64_t m, int64_t n, int64_t k,
                                                     T* __restrict__ Uu) {
  const int64_t v = blockIdx.x;
  for (int64_t i = threadIdx.x; i < m; i += 256) Uu[i * k + v] = (i < n) ? (T)Ub[v * n + i] : T(0);
}

// Back-transformation X <- H_0 H_1 ... H_(np-1) X with H_p = I - V_p T_p V_p^T (reflectors of ONE side, applied last
// to first).  The columns of X transform independently, so a workgroup OWNS BT_COLS columns, keeps them in LDS
// (rows x BT_COLS elements) and walks through all panels without any inter-workgroup step; the panels' V (rows_p x 16,
// row-major) stream from L2 / Infinity Cache ONCE per panel: a thread keeps its (up to KEEP) V rows in registers
// between w = V^T x and x -= V (T w).
//   X: ldx-pitched, rows x ncols, updated in place.  Panel p covers rows row0 + 16 p ... rows - 1.
constexpr int BT_COLS = 2;
constexpr int BT_THREADS = 512;
template <typename T>
constexpr int bt_keep() {
  return sizeof(T) == 4 ? 8 : 4;     // rows per thread held in registers (16 values each): 4096 rows (f32) / 2048 (f64)
}
template <typename T>
struct BtSide {
  T* X;
  int64_t ldx, rows;
  const T* Vall;
  const double* Tall;
  int64_t npanels, row0, vrows0;
  int identity;            // X is [I; 0] on entry (forming Q): column c is untouched by the panels beyond c / 16
};
template <typename T>
struct BtArgs {
  BtSide<T> side[2];       // blockIdx.y: 0 = U (column-panel reflectors), 1 = V (row-panel reflectors)
};
template <typename T>
__global__ __launch_bounds__(BT_THREADS) void backtransform_kernel(BtArgs<T> args) {
  constexpr int KEEP = bt_keep<T>();
  const BtSide<T> sd = args.side[blockIdx.y];
  T* __restrict__ X = sd.X;
  const int64_t ldx = sd.ldx, rows = sd.rows, npanels = sd.npanels, row0 = sd.row0, vrows0 = sd.vrows0;
  const T* __restrict__ Vall = sd.Vall;
  const double* __restrict__ Tall = sd.Tall;
  extern __shared__ __align__(16) unsigned char bt_smem[];
  T* xs = reinterpret_cast<T*>(bt_smem);         // [rows][BT_COLS]
  __shared__ T red[BT_THREADS / 64][16][BT_COLS];
  __shared__ T wv[16][BT_COLS];
  __shared__ T Ts[16][17];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t c0 = (int64_t)blockIdx.x * BT_COLS;
  for (int64_t r = tid; r < rows; r += BT_THREADS)
    *reinterpret_cast<V2<T>*>(xs + r * BT_COLS) = *reinterpret_cast<const V2<T>*>(X + r * ldx + c0);
  __syncthreads();
  int64_t pfirst = npanels - 1;
  if (sd.identity && (c0 + BT_COLS - 1) / 16 < pfirst) pfirst = (c0 + BT_COLS - 1) / 16;
  for (int64_t p = pfirst; p >= 0; --p) {        // npanels = 0 (n = 16 on the V side): nothing to do
    const int64_t rbeg = row0 + 16 * p;          // first row of X the panel touches
    const int64_t vrows = vrows0 - 16 * p;       // rows of V_p  (= rows - rbeg)
    // V_p starts at 16 * (p * vrows0 - 8 p (p - 1)) elements (see vl_offset / vr_offset)
    const T* V = Vall + 16 * (p * vrows0 - 8 * p * (p - 1));
    if (tid < 256) Ts[tid >> 4][tid & 15] = (T)Tall[p * 256 + tid];
    V4<T> vreg[KEEP][4];
#pragma unroll
    for (int it = 0; it < KEEP; ++it) {
      const int64_t r = tid + (int64_t)BT_THREADS * it;
      const V4<T>* vp = reinterpret_cast<const V4<T>*>(V + r * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) vreg[it][q] = (r < vrows) ? vp[q] : vzero<T>();
    }
    T acc[16][BT_COLS];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int q = 0; q < BT_COLS; ++q) acc[i][q] = T(0);
#pragma unroll
    for (int it = 0; it < KEEP; ++it) {
      const int64_t r = tid + (int64_t)BT_THREADS * it;
      if (r < vrows) {
        const V2<T> x = *reinterpret_cast<const V2<T>*>(xs + (rbeg + r) * BT_COLS);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const T ve[4] = {vreg[it][q4].x, vreg[it][q4].y, vreg[it][q4].z, vreg[it][q4].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[4 * q4 + e][0] = fmaT(ve[e], x.x, acc[4 * q4 + e][0]);
            acc[4 * q4 + e][1] = fmaT(ve[e], x.y, acc[4 * q4 + e][1]);
          }
        }
      }
    }
    for (int64_t r = tid + (int64_t)BT_THREADS * KEEP; r < vrows; r += BT_THREADS) {     // taller inputs only
      const V4<T>* vp = reinterpret_cast<const V4<T>*>(V + r * 16);
      const V2<T> x = *reinterpret_cast<const V2<T>*>(xs + (rbeg + r) * BT_COLS);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const V4<T> vv = vp[q4];
        const T ve[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[4 * q4 + e][0] = fmaT(ve[e], x.x, acc[4 * q4 + e][0]);
          acc[4 * q4 + e][1] = fmaT(ve[e], x.y, acc[4 * q4 + e][1]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int q = 0; q < BT_COLS; ++q) {
        const T v = rowsx4(rowsum16(acc[i][q]));
        if (lane == 0) red[w][i][q] = v;
      }
    __syncthreads();
    if (tid < BT_COLS) {
      T sv[16], wo[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        T a = T(0);
#pragma unroll
        for (int k = 0; k < BT_THREADS / 64; ++k) a += red[k][i][tid];
        sv[i] = a;
      }
      tsolve(Ts, 0, sv, wo);                     // w = T (V^T x)
#pragma unroll
      for (int i = 0; i < 16; ++i) wv[i][tid] = wo[i];
    }
    __syncthreads();
    T wr[16][BT_COLS];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int q = 0; q < BT_COLS; ++q) wr[i][q] = wv[i][q];
#pragma unroll
    for (int it = 0; it < KEEP; ++it) {
      const int64_t r = tid + (int64_t)BT_THREADS * it;
      if (r < vrows) {
        V2<T> x = *reinterpret_cast<const V2<T>*>(xs + (rbeg + r) * BT_COLS);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const T ve[4] = {vreg[it][q4].x, vreg[it][q4].y, vreg[it][q4].z, vreg[it][q4].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            x.x = fmaT(-ve[e], wr[4 * q4 + e][0], x.x);
            x.y = fmaT(-ve[e], wr[4 * q4 + e][1], x.y);
          }
        }
        *reinterpret_cast<V2<T>*>(xs + (rbeg + r) * BT_COLS) = x;
      }
    }
    for (int64_t r = tid + (int64_t)BT_THREADS * KEEP; r < vrows; r += BT_THREADS) {
      const V4<T>* vp = reinterpret_cast<const V4<T>*>(V + r * 16);
      V2<T> x = *reinterpret_cast<const V2<T>*>(xs + (rbeg + r) * BT_COLS);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const V4<T> vv = vp[q4];
        const T ve[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          x.x = fmaT(-ve[e], wr[4 * q4 + e][0], x.x);
          x.y = fmaT(-ve[e], wr[4 * q4 + e][1], x.y);
        }
      }
      *reinterpret_cast<V2<T>*>(xs + (rbeg + r) * BT_COLS) = x;
    }
    __syncthreads();
  }
  for (int64_t r = tid; r < rows; r += BT_THREADS)
    *reinterpret_cast<V2<T>*>(X + r * ldx + c0) = *reinterpret_cast<const V2<T>*>(xs + r * BT_COLS);
}

// Vh (k x n) = Vv^T
template <typename T>
__global__ __launch_bounds__(256) void transpose_out_kernel(const T* __restrict__ Vv, int64_t n, int64_t k,
                                                            T* __restrict__ Vh) {
  __shared__ T tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
  const int64_t i0 = (int64_t)blockIdx.x * 32, v0 = (int64_t)blockIdx.y * 32;
  for (int r = ty; r < 32; r += 8)
    tile[r][tx] = (i0 + r < n && v0 + tx < k) ? Vv[(i0 + r) * k + v0 + tx] : T(0);
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (v0 + r < k && i0 + tx < n) Vh[(v0 + r) * n + i0 + tx] = tile[tx][r];
}

// Newton-Schulz coefficient matrix of the f64 path: G <- 1.5 I - 0.5 G  (k x k)
__global__ __launch_bounds__(256) void ns_coeff_kernel(double* __restrict__ G, int64_t k) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= k * k) return;
  G[e] = ((e / k == e % k) ? 1.5 : 0.0) - 0.5 * G[e];
}

// offsets (elements) of panel p's V inside the Vl / Vr blocks: column panel p has m - 16 p rows, row panel p n - 16 p - 16
__host__ __device__ static inline int64_t vl_offset(int64_t m, int64_t p) { return 16 * (p * m - 8 * p * (p - 1)); }
__host__ __device__ static inline int64_t vr_offset(int64_t n, int64_t p) { return 16 * (p * n - 8 * p * (p + 1)); }

#include "tnh_svd_band_fast.inc"

// ------------------------------------------------------------------------------------------------ host side
static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
  size_t Af, Vl, Vr, Vt, Zr, Tl, Tr, Dblk, Eblk, Gpart, Gpart2, Gq, R1, R1inv, Xbuf, Xl, Xr, Ff, Wpart, Wt, Bd, Tb, Trot, scal,
      shifts, counts, flags, lo, hi, status, ray, Lc, Dd, X, X2, X3, Gns, Uu, Vv, total;
  int64_t np, kcap, nshift;
  int esz;
};


// esz: 4 (f32 input) or 8 (f64 input).  Everything up to `status` does not depend on kcap except through the shift
// buffers' size; the f64-only blocks (Gq .. R1inv, ray, X2, X3, Gns) are empty for f32.
static Layout make_layout(int64_t m, int64_t n, int64_t kcap, int esz) {
  Layout L;
  const int64_t np = n / 16;
  const bool f64 = esz == 8;
  const size_t e = (size_t)esz;
  L.np = np;
  L.kcap = kcap;
  L.esz = esz;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
  L.Af = take((size_t)m * n * e);
  L.Vl = take((size_t)vl_offset(m, np) * e);
  L.Vr = take((size_t)(vr_offset(n, np - 1) + 16) * e);
  L.Vt = take((size_t)16 * n * e);
  L.Zr = take((size_t)16 * m * e);
  L.Tl = take((size_t)np * 256 * 8);
  L.Tr = take((size_t)np * 256 * 8);
  L.Dblk = take((size_t)np * 256 * 8);
  L.Eblk = take((size_t)np * 256 * 8);
  const int64_t maxparts = (m + 63) / 64 + 1;        // gram_kernel: 256 rows per part; update_kernel<1>: 128; rowreduce: 64
  L.Gpart = take((size_t)maxparts * 256 * 8);
  L.Gpart2 = take((size_t)((n + 63) / 64 + 1) * 256 * 8);      // update_kernel<2>: one part per 64 columns
  L.Gq = take(f64 ? (size_t)((m + 255) / 256 + 1) * 256 * 8 : 0);   // Cholesky-QR2: Gram partials of Q1 (256 rows per part)
  L.R1 = take(f64 ? 256 * 8 : 0);
  L.R1inv = take(f64 ? 256 * 8 : 0);
  L.Xbuf = take(256 * 8);
  L.Xl = take((size_t)np * 256 * 8);       // fast stage 1: X of every panel, for the batched V
  L.Xr = take((size_t)np * 256 * 8);
  L.Ff = take(2 * 512 * e);                // X and T^-1 (in the input type) of the panel being reduced (column side, row side)
  const int64_t wide = n > kcap ? n : kcap;
  const int64_t wrc = f64 ? w_rc<double>() : w_rc<float>();
  const int64_t chunks = (m + wrc - 1) / wrc;
  size_t wpart = (size_t)chunks * 16 * wide * e;
  const size_t ypart = (size_t)((n + Y_COLS - 1) / Y_COLS) * m * 16 * e;     // the row panels' Y partials share the buffer
  if (ypart > wpart) wpart = ypart;
  {       // fast stage 1: raw-pass partials per 64-row block (16 x n each) and per column tile of 64 (f64: 32) (m x 16 each)
    const size_t colparts = (size_t)((m + 63) / 64 + 1) * 16 * (size_t)n;
    const int64_t cwmin = f64 ? 32 : 64;
    const size_t rowparts = (size_t)((n + cwmin - 1) / cwmin + 1) * 16 * (size_t)m;
    const size_t fast = (colparts > rowparts ? colparts : rowparts) * e;
    if (fast > wpart) wpart = fast;
  }
  L.Wpart = take(wpart);
  L.Wt = take((size_t)16 * wide * e);
  L.Bd = take((size_t)n * 17 * 8);
  L.Tb = take((size_t)n * 17 * 8);
  L.Trot = take((size_t)(n + 32) * TP * 8);
  L.scal = take(64);
  // shifts per round: the grid has up to 16 n points, a section round up to 15 n, a refinement round 255 kcap
  int64_t ns = 16 * n;
  if (256 * kcap > ns) ns = 256 * kcap;
  L.nshift = ns;
  L.shifts = take((size_t)ns * 8);
  L.counts = take((size_t)ns * 4);
  L.flags = take((size_t)ns * 4);
  L.lo = take((size_t)n * 8);
  L.hi = take((size_t)n * 8);
  L.status = take(64);
  L.ray = take(f64 ? (size_t)kcap * 8 : 0);
  L.Lc = take((size_t)kcap * n * 16 * 8);
  L.Dd = take((size_t)kcap * n * 8);
  L.X = take((size_t)kcap * n * 8);
  L.X2 = take(f64 ? (size_t)kcap * n * 8 : 0);
  L.X3 = take(f64 ? (size_t)kcap * n * 8 : 0);
  L.Gns = take(f64 ? (size_t)kcap * kcap * 8 : 0);
  L.Uu = take((size_t)m * kcap * e);
  L.Vv = take((size_t)n * kcap * e);
  L.total = off;
  return L;
}

// ---- fast stage 1 (f32): tnh_svd_band_fast.inc
static int g_last_fast = 0;
static bool g_fast = true;          // TNH_SVDB_FAST=0: the loop of rounds 3-5 for every panel
static int g_fast_switch = 128;     // hand over when the trailing block has this many columns or fewer (TNH_SVDB_FAST_SWITCH)
static double g_fast_cond = 16.0;   // panels with max / min Cholesky diagonal above this: ST_FASTCOND, stage 1 is repeated (TNH_SVDB_FAST_COND)
static int g_fast_cw = 0;           // column tile of the fused kernels: 0 = by size, 64, 128 (TNH_SVDB_FAST_CW)
static int g_fast_wgs = 1024;       // workgroups a fused launch aims at (TNH_SVDB_FAST_WGS)
static bool g_fast64 = true;        // f64 input through the fast stage too (TNH_SVDB_FAST64=0: the fourteen-launch loop)
static bool g_dpp = true;
static double g_cluster_tol = 4e-7;  // kept values closer than this (relative to sigma_max: a few eps_f32) are one cluster (TNH_SVDB_CTOL)
static double g_cluster_tol64 = 1e-10;  // f64 input: closer than this is a cluster; wider neighbours are separated by the
                                     // inverse iteration to eps64 / gap <= 1e-6 and made orthonormal by ONE Newton-Schulz step
static bool g_row_fused = false;  // row panels by the row-owning fused kernel (TNH_SVDB_ROWFUSED=1; f32 only) or ypass / yreduce / update
static bool g_bt_fused = true;   // back-transformation by column-owning workgroups (TNH_SVDB_BT=0: per-panel launches)
static bool g_lane = true;       // counts through sturm_lane_kernel (TNH_SVDB_LANE=0: the 16-lane ldl_kernel)
static bool g_lane_auto = true;  // ... unless the round is small enough for the 16-lane kernel to be faster (TNH_SVDB_LANE_AUTO=0)
// Schedule of the spectrum slicing.  A count costs 139 f64 FMAs per pivot (8 cycles each on this chip's vector pipe),
// so the schedule is about the NUMBER of counts and of dependent rounds:
//   all values: a uniform grid of 16 n shifts (capped at 65536 = one lane-kernel wave per SIMD) gives every value 16
//   bits at once, one 16-way section round per value adds 4 (bracket 1e-6 sigma_max: s_rest is good to 5e-7);
//   kept values: g_refine_rounds 16-way rounds on the 16-lane kernel (a few thousand shifts: latency 0.8 ms instead
//   of the lane kernel's 3 ms) take their brackets to 2^-32 sigma_max so that inverse iteration separates neighbours.
//   f64 input: every value to g_bits64 bits (default 28: s_rest good to 2e-9 sigma_max; two more lane rounds),
//   kept values to 2^-44 sigma_max (three more 16-lane rounds) -- beyond that the un-pivoted LDL^T of the band does not
//   resolve (eps64 n |T| / 2 sigma); the kept VALUES then come from the Rayleigh quotient |B x| of their vectors.
static int g_grid_mult = 16;
static int g_sect_p = 15;
static int g_sect_rounds = 1;
static int g_refine_p = 15;
static int g_refine_rounds = 3;
static int g_bits64 = 28;
static bool g_ns64 = true;      // f64 input: Newton-Schulz step on the kept band vectors (TNH_SVDB_NS=0: off, diagnostics)

static void read_env() {
  const char* e = getenv("TNH_SVDB_DPP");
  g_dpp = !(e && e[0] == '0');
  e = getenv("TNH_SVDB_CTOL");
  if (e && atof(e) >= 0.0) g_cluster_tol = atof(e);
  e = getenv("TNH_SVDB_CTOL64");
  if (e && atof(e) >= 0.0) g_cluster_tol64 = atof(e);
  e = getenv("TNH_SVDB_ROWFUSED");
  g_row_fused = (e && e[0] == '1');
  e = getenv("TNH_SVDB_BT");
  g_bt_fused = !(e && e[0] == '0');
  e = getenv("TNH_SVDB_LANE");
  g_lane = !(e && e[0] == '0');
  e = getenv("TNH_SVDB_LANE_AUTO");
  g_lane_auto = !(e && e[0] == '0');
  g_grid_mult = g_lane ? 16 : 1;
  g_sect_p = g_lane ? 15 : 3;
  g_sect_rounds = g_lane ? 1 : 5;
  g_refine_p = 15;
  g_refine_rounds = 3;
  g_bits64 = 28;
  e = getenv("TNH_SVDB_GRID");
  if (e && atoi(e) > 0) g_grid_mult = atoi(e);
  e = getenv("TNH_SVDB_SECT");
  if (e && atoi(e) > 0) g_sect_p = atoi(e);
  e = getenv("TNH_SVDB_ROUNDS");
  if (e && atoi(e) >= 0) g_sect_rounds = atoi(e);
  e = getenv("TNH_SVDB_REFINE_P");
  if (e && atoi(e) > 0) g_refine_p = atoi(e);
  e = getenv("TNH_SVDB_REFINE");
  if (e && atoi(e) >= 0) g_refine_rounds = atoi(e);
  e = getenv("TNH_SVDB_NS");
  g_ns64 = !(e && e[0] == '0');
  e = getenv("TNH_SVDB_FAST");
  g_fast = !(e && e[0] == '0');
  e = getenv("TNH_SVDB_FAST_SWITCH");
  g_fast_switch = (e && atoi(e) >= 32) ? atoi(e) : 128;
  e = getenv("TNH_SVDB_FAST_COND");
  g_fast_cond = (e && atof(e) > 1.0) ? atof(e) : 16.0;
  e = getenv("TNH_SVDB_FAST_CW");
  g_fast_cw = (e && (atoi(e) == 64 || atoi(e) == 128)) ? atoi(e) : 0;
  e = getenv("TNH_SVDB_FAST64");
  g_fast64 = !(e && e[0] == '0');
  e = getenv("TNH_SVDB_FAST_WGS");
  g_fast_wgs = (e && atoi(e) >= 64) ? atoi(e) : 1024;
  e = getenv("TNH_SVDB_BITS64");
  if (e && atoi(e) >= 20 && atoi(e) <= 44) g_bits64 = atoi(e);
}

// tau_rel: resolution (relative to sigma_max) the round is after -- a count whose small pivots injected more than a
// fifth of it is flagged and ignored by the bracket update (1.2e-7 = 2^-23: the f32 path's rounds)
static int launch_counts(const Layout& L, char* base, int64_t n, int64_t ns, bool lane, bool need_flags, double tau_rel) {
  // Which kernel (round 6): a wave of the lane kernel carries 64 shifts through the n pivots at ~1600 cycles each, a wave
  // of the 16-lane kernel 4 shifts at ~400 (profiles/r06_svd_band_f32_kernel_stats_trip4.txt: 3.2 ms for 65536 shifts,
  // 0.79 ms for 3840, n = 4096).  Both are latency chains, so a round costs (waves per SIMD, rounded up) x n x that:
  // the 16-lane kernel wins while the shifts fit in about two of its waves per SIMD -- the rounds of n <= 512.
  if (lane && g_lane_auto) {
    const int64_t simds = 4 * (int64_t)(num_cus() > 0 ? num_cus() : 256);
    const int64_t w_lane = ((ns + 63) / 64 + simds - 1) / simds, w_ldl = ((ns + 3) / 4 + simds - 1) / simds;
    if (w_ldl * 400 < w_lane * 1600) lane = false;
  }
  if (lane) {
    if (need_flags)
      hipLaunchKernelGGL(sturm_lane_kernel<true>, dim3((unsigned)((ns + 63) / 64)), dim3(64), 0, stream(),
                         (const double*)(base + L.Trot), n, (const double*)(base + L.shifts), ns,
                         (const double*)(base + L.scal), tau_rel, (int*)(base + L.counts), (int*)(base + L.flags));
    else
      hipLaunchKernelGGL(sturm_lane_kernel<false>, dim3((unsigned)((ns + 63) / 64)), dim3(64), 0, stream(),
                         (const double*)(base + L.Trot), n, (const double*)(base + L.shifts), ns,
                         (const double*)(base + L.scal), tau_rel, (int*)(base + L.counts), (int*)(base + L.flags));
    TNH_LAUNCH_CHECK();
    return TNH_OK;
  }
  const unsigned blocks = (unsigned)((ns + 3) / 4);
  if (g_dpp)
    hipLaunchKernelGGL((ldl_kernel<false, true>), dim3(blocks), dim3(64), 0, stream(), (const double*)(base + L.Trot), n,
                       (const double*)(base + L.shifts), ns, (const double*)(base + L.scal), tau_rel,
                       (int*)(base + L.counts), (int*)(base + L.flags), (double*)nullptr, (double*)nullptr);
  else
    hipLaunchKernelGGL((ldl_kernel<false, false>), dim3(blocks), dim3(64), 0, stream(), (const double*)(base + L.Trot), n,
                       (const double*)(base + L.shifts), ns, (const double*)(base + L.scal), tau_rel,
                       (int*)(base + L.counts), (int*)(base + L.flags), (double*)nullptr, (double*)nullptr);
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

// one multi-section round over the values [q0, q0 + nq)
static int section_round(const Layout& L, char* base, int64_t n, int64_t q0, int64_t nq, int P, int round, bool lane,
                         double tau_rel) {
  if (nq <= 0) return TNH_OK;
  const int64_t ns = nq * P;
  const double skew = 0.07 * (double)((round % 3) - 1);
  hipLaunchKernelGGL(section_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, stream(),
                     (const double*)(base + L.lo), (const double*)(base + L.hi), q0, nq, P, skew,
                     (double*)(base + L.shifts));
  int rc = launch_counts(L, base, n, ns, lane, true, tau_rel);
  if (rc) return rc;
  hipLaunchKernelGGL(bracket_update_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream(),
                     (double*)(base + L.lo), (double*)(base + L.hi), q0, nq, P, (const double*)(base + L.shifts),
                     (const int*)(base + L.counts), (const int*)(base + L.flags));
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

// p_begin > 0: the panels before it were reduced by stage1_fast, which also left the partial Gram matrices of column
// panel p_begin (parts_in of them) in Gpart.
template <typename T>
static int stage1(const Layout& L, char* base, int64_t m, int64_t n, int64_t p_begin = 0, int parts_in = 0) {
  constexpr bool QR2 = sizeof(T) == 8;       // f64 input: two Cholesky-QR passes per panel (see chol_kernel)
  constexpr int WRC = w_rc<T>();
  constexpr int YR = y_rows<T>();
  T* Af = (T*)(base + L.Af);
  double* Gc = (double*)(base + L.Gpart);                 // partial Grams of the next column panel
  double* Gr = (double*)(base + L.Gpart2);                // ... of the next row panel
  double* Gq = (double*)(base + L.Gq);
  double* R1 = (double*)(base + L.R1);
  double* R1inv = (double*)(base + L.R1inv);
  double* Xb = (double*)(base + L.Xbuf);
  T* Wpart = (T*)(base + L.Wpart);
  T* Wt = (T*)(base + L.Wt);
  T* Vt = (T*)(base + L.Vt);
  int* status = (int*)(base + L.status);
  const int64_t np = L.np;
  int parts_c = (int)((m + 255) / 256), parts_r = 0;
  if (p_begin == 0) hipLaunchKernelGGL((gram_kernel<false, T>), dim3(parts_c), dim3(256), 0, stream(), (const T*)Af, n, m, Gc);
  else parts_c = parts_in;
  for (int64_t p = p_begin; p < np; ++p) {
    const int64_t j = 16 * p;
    const int64_t mj = m - j, nc = n - j - 16, mr = m - j - 16;
    // ---- column panel: rows j .., columns j .. j + 15
    {
      T* P = Af + j * n + j;
      T* V = (T*)(base + L.Vl) + vl_offset(m, p);
      double* Tp = (double*)(base + L.Tl) + p * 256;
      if (QR2) {
        const int qparts = (int)((mj + 255) / 256);
        hipLaunchKernelGGL(chol_kernel, dim3(1), dim3(256), 0, stream(), (const double*)Gc, parts_c, R1, R1inv, status);
        hipLaunchKernelGGL((scaleq_kernel<false, T>), dim3(qparts), dim3(256), 0, stream(), P, n, mj,
                           (const double*)R1inv, Gq);
        hipLaunchKernelGGL((factor_kernel<false, T>), dim3(1), dim3(256), 0, stream(), (const T*)P, n, mj,
                           (const double*)Gq, qparts, Xb, V, (T*)nullptr, (int64_t)0, Tp,
                           (double*)(base + L.Dblk) + p * 256, (const double*)R1, 1e-3, status);
      } else {
        hipLaunchKernelGGL((factor_kernel<false, T>), dim3(1), dim3(256), 0, stream(), (const T*)P, n, mj,
                           (const double*)Gc, parts_c, Xb, V, (T*)nullptr, (int64_t)0, Tp,
                           (double*)(base + L.Dblk) + p * 256, (const double*)nullptr, 1e-9, status);
      }
      if (mj > 16)
        hipLaunchKernelGGL((formv_kernel<false, T>), dim3((unsigned)((mj - 16 + 255) / 256)), dim3(256), 0, stream(),
                           (const T*)P, n, mj, (const double*)Xb, V, (T*)nullptr, (int64_t)0);
      if (nc > 0) {
        T* C = Af + j * n + j + 16;
        const int chunks = (int)((mj + WRC - 1) / WRC);
        hipLaunchKernelGGL((wpass_kernel<T>), dim3((unsigned)((nc + 16 * w_cpl<T>() - 1) / (16 * w_cpl<T>())), chunks), dim3(256), 0, stream(),
                           (const T*)C, n, mj, nc, (const T*)V, Wpart);
        hipLaunchKernelGGL((wreduce_kernel<T>), dim3((unsigned)((nc + 63) / 64)), dim3(256), 0, stream(),
                           (const T*)Wpart, chunks, nc, (const double*)Tp, 1, Wt);
        // the update also leaves the partial Grams of the row panel (first 16 rows of the updated block)
        const dim3 grid((unsigned)((nc + 63) / 64), (unsigned)((mj + U_RR - 1) / U_RR));
        hipLaunchKernelGGL((update_kernel<2, T>), grid, dim3(256), 0, stream(), C, n, mj, nc, (const T*)V,
                           (const T*)Wt, nc, Gr, (int64_t)0);
        parts_r = (int)grid.x;
      }
    }
    // ---- row panel: rows j .. j + 15, columns j + 16 ..
    if (nc > 0) {
      T* P = Af + j * n + j + 16;
      T* V = (T*)(base + L.Vr) + vr_offset(n, p);
      double* Tp = (double*)(base + L.Tr) + p * 256;
      if (QR2) {
        const int qparts = (int)((nc + 255) / 256);
        hipLaunchKernelGGL(chol_kernel, dim3(1), dim3(256), 0, stream(), (const double*)Gr, parts_r, R1, R1inv, status);
        hipLaunchKernelGGL((scaleq_kernel<true, T>), dim3(qparts), dim3(256), 0, stream(), P, n, nc,
                           (const double*)R1inv, Gq);
        hipLaunchKernelGGL((factor_kernel<true, T>), dim3(1), dim3(256), 0, stream(), (const T*)P, n, nc,
                           (const double*)Gq, qparts, Xb, V, Vt, nc, Tp, (double*)(base + L.Eblk) + p * 256,
                           (const double*)R1, 1e-3, status);
      } else {
        hipLaunchKernelGGL((factor_kernel<true, T>), dim3(1), dim3(256), 0, stream(), (const T*)P, n, nc,
                           (const double*)Gr, parts_r, Xb, V, Vt, nc, Tp, (double*)(base + L.Eblk) + p * 256,
                           (const double*)nullptr, 1e-9, status);
      }
      if (nc > 16)
        hipLaunchKernelGGL((formv_kernel<true, T>), dim3((unsigned)((nc - 16 + 255) / 256)), dim3(256), 0, stream(),
                           (const T*)P, n, nc, (const double*)Xb, V, Vt, nc);
      if (mr > 0) {
        T* C = Af + (j + 16) * n + j + 16;
        if constexpr (sizeof(T) == 4) {
          if (g_row_fused) {
            hipLaunchKernelGGL(rowupdate_kernel, dim3((unsigned)((mr + 15) / 16)), dim3(256), 0, stream(), C, n, mr, nc,
                               (const float*)Vt, nc, (const double*)Tp);
            parts_c = (int)((mr + 255) / 256);
            hipLaunchKernelGGL((gram_kernel<false, T>), dim3(parts_c), dim3(256), 0, stream(), (const T*)C, n, mr, Gc);
            TNH_LAUNCH_CHECK();
            continue;
          }
        }
        const int ych = (int)((nc + Y_COLS - 1) / Y_COLS);
        T* Z = (T*)(base + L.Zr);
        hipLaunchKernelGGL((ypass_kernel<T>), dim3(ych, (unsigned)((mr + YR - 1) / YR)), dim3(256), 0, stream(),
                           (const T*)C, n, mr, nc, (const T*)Vt, nc, Wpart);
        hipLaunchKernelGGL((yreduce_kernel<T>), dim3((unsigned)((mr + 63) / 64)), dim3(256), 0, stream(),
                           (const T*)Wpart, ych, mr, (const double*)Tp, Z);
        // ... and this update the partial Grams of the next column panel (first 16 columns of the updated block)
        const dim3 grid((unsigned)((nc + 63) / 64), (unsigned)((mr + U_RR - 1) / U_RR));
        hipLaunchKernelGGL((update_kernel<1, T>), grid, dim3(256), 0, stream(), C, n, mr, nc, (const T*)Z,
                           (const T*)Vt, nc, Gc, (int64_t)0);
        parts_c = (int)grid.y;
      }
    }
    TNH_LAUNCH_CHECK();
  }
  return TNH_OK;
}


// ---- fast stage 1: tnh_svd_band_fast.inc.  Panels 0 .. *p_next - 1 are reduced here; the caller finishes with
// stage1<T>(..., *p_next, *parts_out).

template <typename T, int CW>
static void launch_colupd(const ColUpdArgs<T>& a, unsigned grid) {
  hipLaunchKernelGGL((colupd_rowpass_kernel<T, CW, true>), dim3(grid), dim3(256), 0, stream(), a);
}
template <typename T, int CW>
static void launch_rowupd(const RowUpdArgs<T>& a, unsigned grid, bool upd, bool pass) {
  if (upd && pass) hipLaunchKernelGGL((rowupd_colpass_kernel<T, CW, true, true, true>), dim3(grid), dim3(256), 0, stream(), a);
  else if (pass) hipLaunchKernelGGL((rowupd_colpass_kernel<T, CW, false, true, true>), dim3(grid), dim3(256), 0, stream(), a);
  else hipLaunchKernelGGL((rowupd_colpass_kernel<T, CW, true, false, false>), dim3(grid), dim3(256), 0, stream(), a);
}
// 64-row steps per workgroup: as many workgroups as g_fast_wgs allows, at most 4 steps
static int fast_iters(int64_t rows, int ntc) {
  const int64_t units = (rows + 63) / 64;
  int64_t it = (units * ntc + g_fast_wgs - 1) / g_fast_wgs;
  if (it < 1) it = 1;
  if (it > 4) it = 4;
  return (int)it;
}

template <typename T>
static int stage1_fast(const Layout& L, char* base, int64_t m, int64_t n, int64_t* p_next, int* parts_out) {
  constexpr bool F64 = sizeof(T) == 8;
  // column tiles (registers: a lane's accumulators, prefetch and operand tiles scale with CW x sizeof(T)): f32 64, and
  // 128 for colupd_rowpass on wide blocks (half the Y partials); f64 half of that
  constexpr int CWA_BIG = F64 ? 32 : 128, CWA = F64 ? 32 : 64, CWB = F64 ? 32 : 64;     // (f64 at 64 columns spills: 54.1 against 52.7 ms)
  T* Af = (T*)(base + L.Af);
  double* Gc = (double*)(base + L.Gpart);
  double* Gr = (double*)(base + L.Gpart2);
  double* Gq = (double*)(base + L.Gq);
  double* R1 = (double*)(base + L.R1);
  T* Part = (T*)(base + L.Wpart);
  T* Wx = (T*)(base + L.Wt);
  T* Zx = (T*)(base + L.Zr);
  T* Vl = (T*)(base + L.Vl);
  T* Vr = (T*)(base + L.Vr);
  double* Xl = (double*)(base + L.Xl);
  double* Xr = (double*)(base + L.Xr);
  double* Tl = (double*)(base + L.Tl);
  double* Tr = (double*)(base + L.Tr);
  int* status = (int*)(base + L.status);
  T* Ffc = (T*)(base + L.Ff);
  T* Ffr = Ffc + 512;
  const double rel = F64 ? 1e-3 : 1e-9;              // f64: the factor sees Q1, whose Gram matrix is I + O(eps64 cond^2)
  const double cond = F64 ? 0.0 : g_fast_cond;       // ... and the passes run on Q1: nothing to guard
  int64_t ps = 0;
  while (ps < L.np && n - 16 * ps > g_fast_switch) ++ps;
  *p_next = 0;
  if (ps < 2) return TNH_OK;
  int parts_c = (int)((m + 255) / 256), parts_r = 0;
  hipLaunchKernelGGL((gram_kernel<false, T>), dim3(parts_c), dim3(256), 0, stream(), (const T*)Af, n, m, Gc);
  for (int64_t p = 0; p <= ps; ++p) {
    const int64_t j = 16 * p;
    const int64_t mj = m - j, nc = n - j - 16, mr = m - j - 16;
    const bool last = p == ps;           // only the pending row update of panel ps - 1
    if (nc <= 0) break;
    const bool bigA = g_fast_cw ? g_fast_cw == 128 : nc >= 2048;
    const int cw = bigA ? CWA_BIG : CWA;                                   // column tile of colupd_rowpass
    const int ntc = (int)((nc + cw - 1) / cw);
    const double* gram_c = Gc;
    int parts_cf = parts_c;
    if (F64 && !last) {
      // first Cholesky-QR pass of column panel p: Q1 = P R1^-1 in place, its partial Grams
      if constexpr (F64) {
        const int qparts = (int)((mj + 255) / 256);
        hipLaunchKernelGGL((cholscale_kernel<false>), dim3(qparts), dim3(256), 0, stream(), (double*)(Af + j * n + j), n, mj,
                           (const double*)Gc, parts_c, R1, Gq, status);
        gram_c = Gq;
        parts_cf = qparts;
      }
    }
    // ---- row panel p - 1's update of everything right of column panel p + the raw pass and the factor of that panel
    int nrt;
    {
      RowUpdArgs<T> a;
      a.C = Af + j * n + j + 16;
      a.ldc = n;
      a.rows = mj;
      a.nc = nc;
      a.Zx = Zx + j * 16;
      a.zx_ld = 16;
      a.Pr = p > 0 ? Af + (j - 16) * n + j + 16 : Af;
      a.pr_ld = n;
      a.Pcol = Af + j * n + j;
      a.Wpart = Part;
      a.ntc = (int)((nc + CWB - 1) / CWB);
      a.iters = fast_iters(mj, a.ntc);
      nrt = (int)(((mj + 63) / 64 + a.iters - 1) / a.iters);
      a.fa = FactorArgs<T>{Af + j * n + j, n, mj, gram_c, parts_cf, Xl + p * 256, Vl + vl_offset(m, p), Tl + p * 256,
                           (double*)(base + L.Dblk) + p * 256, F64 ? (const double*)R1 : (const double*)nullptr, rel, cond,
                           status, Ffc};
      const unsigned grid = (unsigned)(nrt * a.ntc) + (last ? 0u : 1u);
      launch_rowupd<T, CWB>(a, grid, p > 0, !last);
    }
    if (last) break;
    // ---- the 16 x 16 algebra per trailing column: row panel p finished, Wx, its partial Grams
    parts_r = (int)((nc + 63) / 64);
    hipLaunchKernelGGL((colreduce_kernel<T>), dim3((unsigned)parts_r), dim3(256), 0, stream(), (const T*)Part, nrt, nc,
                       Af + j * n + j + 16, n, (const T*)(Vl + vl_offset(m, p)), (const T*)Ffc, Wx, n, Gr);
    const double* gram_r = Gr;
    int parts_rf = parts_r;
    if constexpr (F64) {
      const int qparts = (int)((nc + 255) / 256);
      hipLaunchKernelGGL((cholscale_kernel<true>), dim3(qparts), dim3(256), 0, stream(), (double*)(Af + j * n + j + 16), n, nc,
                         (const double*)Gr, parts_r, R1, Gq, status);
      gram_r = Gq;
      parts_rf = qparts;
    }
    // ---- column panel p's update + the raw pass and the factor of row panel p
    {
      ColUpdArgs<T> a;
      a.C = Af + (j + 16) * n + j + 16;
      a.ldc = n;
      a.rows = mr;
      a.nc = nc;
      a.Pcol = Af + (j + 16) * n + j;
      a.Wx = Wx;
      a.wx_pitch = n;
      a.Pr = Af + j * n + j + 16;
      a.Ypart = Part;
      a.ntc = ntc;
      a.iters = fast_iters(mr, ntc);
      const int nrt2 = (int)(((mr + 63) / 64 + a.iters - 1) / a.iters);
      a.fa = FactorArgs<T>{Af + j * n + j + 16, n, nc, gram_r, parts_rf, Xr + p * 256, Vr + vr_offset(n, p), Tr + p * 256,
                           (double*)(base + L.Eblk) + p * 256, F64 ? (const double*)R1 : (const double*)nullptr, rel, cond,
                           status, Ffr};
      const unsigned grid = (unsigned)(nrt2 * ntc) + 1u;
      if (bigA) launch_colupd<T, CWA_BIG>(a, grid);
      else launch_colupd<T, CWA>(a, grid);
    }
    // ---- the same algebra per trailing row: column panel p + 1 finished, Zx, its partial Grams
    parts_c = (int)((mr + 63) / 64);
    hipLaunchKernelGGL((rowreduce_kernel<T>), dim3((unsigned)parts_c), dim3(256), 0, stream(), (const T*)Part, ntc, mr,
                       Af + (j + 16) * n + j + 16, n, (const T*)(Vr + vr_offset(n, p)), (const T*)Ffr,
                       Zx + (j + 16) * 16, Gc);
    TNH_LAUNCH_CHECK();
  }
  {
    FormVAllArgs<T> a{Af, m, n, ps, Xl, Xr, Vl, Vr};
    hipLaunchKernelGGL((formv_all_kernel<T>), dim3((unsigned)((m - 16 + 255) / 256), (unsigned)ps, 2), dim3(256), 0, stream(), a);
  }
  TNH_LAUNCH_CHECK();
  *p_next = ps;
  *parts_out = parts_c;
  return TNH_OK;
}

template <typename T>
static int values(const Layout& L, char* base, int64_t n, T* S_out) {
  constexpr bool F64 = sizeof(T) == 8;
  const unsigned nb17 = (unsigned)((n * 17 + 255) / 256);
  hipLaunchKernelGGL(band_kernel, dim3(nb17), dim3(256), 0, stream(), (const double*)(base + L.Dblk),
                     (const double*)(base + L.Eblk), n, (double*)(base + L.Bd));
  hipLaunchKernelGGL(tband_kernel, dim3(nb17), dim3(256), 0, stream(), (const double*)(base + L.Bd), n,
                     (double*)(base + L.Tb));
  hipLaunchKernelGGL(trot_kernel, dim3((unsigned)(((n + 32) * TP + 255) / 256)), dim3(256), 0, stream(),
                     (const double*)(base + L.Tb), n, (double*)(base + L.Trot));
  hipLaunchKernelGGL(smax_kernel, dim3(1), dim3(1024), 0, stream(), (const double*)(base + L.Tb), n,
                     (double*)(base + L.scal));
  int64_t ng = (int64_t)g_grid_mult * n;
  if (ng > 65536) ng = 65536 > n ? 65536 : n;
  if (ng > L.nshift) ng = L.nshift;
  hipLaunchKernelGGL(grid_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, stream(),
                     (const double*)(base + L.scal), ng, (double*)(base + L.shifts));
  TNH_LAUNCH_CHECK();
  int rc = launch_counts(L, base, n, ng, g_lane, false, 1.2e-7);
  if (rc) return rc;
  hipLaunchKernelGGL(bracket_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream(),
                     (const double*)(base + L.scal), (const int*)(base + L.counts), ng, n, (double*)(base + L.lo),
                     (double*)(base + L.hi));
  TNH_LAUNCH_CHECK();
  // at least 20 (f64 input: g_bits64) bits per value in total: log2(ng) from the grid, log2(P + 1) per section round
  const double want = F64 ? (double)g_bits64 : 20.0;
  const double per_round = log2((double)g_sect_p + 1.0);
  int rounds = g_sect_rounds;
  {
    double bits = log2((double)ng) + rounds * per_round;
    while (bits < want && rounds < 16) {
      ++rounds;
      bits += per_round;
    }
  }
  double bits = log2((double)ng);
  for (int r = 0; r < rounds; ++r) {
    bits += per_round;
    // f64 input: the flag threshold follows the resolution the round is after (never finer than 2^-44)
    const double tau = F64 ? fmax(exp2(-bits), exp2(-44.0)) : 1.2e-7;
    rc = section_round(L, base, n, 0, n, g_sect_p, r, g_lane, F64 ? fmin(tau, 1.2e-7) : tau);
    if (rc) return rc;
  }
  hipLaunchKernelGGL((values_out_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream(),
                     (const double*)(base + L.lo), (const double*)(base + L.hi), n, S_out);
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

// One Newton-Schulz step  Y = (1.5 I - 0.5 X X^T) X  on k row vectors of length n (f64): the polar factor's first
// iterate -- orthonormality error e -> ~e^2, the smallest change that achieves it, two GEMMs on the f64 matrix cores.
static int newton_schulz(double* X, double* Y, double* G, int64_t k, int64_t n) {
  int rc = tnh_gemm(TNH_F64, TNH_F64, 0, 1, k, k, n, X, n, X, n, G, k, 1, 0, 0, 0);
  if (rc) return rc;
  hipLaunchKernelGGL(ns_coeff_kernel, dim3((unsigned)((k * k + 255) / 256)), dim3(256), 0, stream(), G, k);
  TNH_LAUNCH_CHECK();
  return tnh_gemm(TNH_F64, TNH_F64, 0, 0, k, n, k, G, k, X, n, Y, n, 1, 0, 0, 0);
}

template <typename T>
static int vectors(const Layout& L, char* base, int64_t m, int64_t n, int64_t k, T* U, T* Vh, T* S_kept) {
  constexpr bool F64 = sizeof(T) == 8;
  // kept values: brackets down to 2^-32 (f64 input: 2^-44) sigma_max so that inverse iteration separates close neighbours
  int rc;
  const int refine_rounds = g_refine_rounds + (F64 ? 3 : 0);
  double bits = F64 ? (double)g_bits64 : 20.0;
  for (int r = 0; r < refine_rounds; ++r) {
    bits += log2((double)g_refine_p + 1.0);
    const double tau = F64 ? fmin(fmax(exp2(-bits), exp2(-44.0)), 1.2e-7) : 1.2e-7;
    rc = section_round(L, base, n, n - k, k, g_refine_p, r, g_lane && g_refine_p > 15, tau);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(vshift_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, stream(),
                     (const double*)(base + L.lo), (const double*)(base + L.hi), (const double*)(base + L.scal), n, k,
                     F64 ? g_cluster_tol64 : 0.0, F64 ? 1e-5 : 1e-6, (double*)(base + L.shifts), (int*)(base + L.status));
  const unsigned blocks = (unsigned)((k + 3) / 4);
  if (g_dpp) {
    hipLaunchKernelGGL((ldl_kernel<true, true>), dim3(blocks), dim3(64), 0, stream(), (const double*)(base + L.Trot), n,
                       (const double*)(base + L.shifts), k, (const double*)(base + L.scal), 1.2e-7, (int*)nullptr,
                       (int*)nullptr, (double*)(base + L.Lc), (double*)(base + L.Dd));
    hipLaunchKernelGGL((solve_kernel<true>), dim3(blocks), dim3(64), 0, stream(), (const double*)(base + L.Lc),
                       (const double*)(base + L.Dd), n, k, 3, (double*)(base + L.X));
  } else {
    hipLaunchKernelGGL((ldl_kernel<true, false>), dim3(blocks), dim3(64), 0, stream(), (const double*)(base + L.Trot), n,
                       (const double*)(base + L.shifts), k, (const double*)(base + L.scal), 1.2e-7, (int*)nullptr,
                       (int*)nullptr, (double*)(base + L.Lc), (double*)(base + L.Dd));
    hipLaunchKernelGGL((solve_kernel<false>), dim3(blocks), dim3(64), 0, stream(), (const double*)(base + L.Lc),
                       (const double*)(base + L.Dd), n, k, 3, (double*)(base + L.X));
  }
  hipLaunchKernelGGL(cluster_mgs_kernel, dim3(1), dim3(1024), 0, stream(), (double*)(base + L.X),
                     (const double*)(base + L.shifts), (const double*)(base + L.scal), n, k,
                     F64 ? g_cluster_tol64 : g_cluster_tol, (int*)(base + L.status));
  TNH_LAUNCH_CHECK();
  double* Xcur = (double*)(base + L.X);
  double* ray = F64 ? (double*)(base + L.ray) : nullptr;
  double* Ub = nullptr;
  if (F64 && g_ns64) {
    // neighbours closer than ~1e-6 sigma_max come out of the inverse iteration orthogonal to eps64 / gap only
    rc = newton_schulz(Xcur, (double*)(base + L.X2), (double*)(base + L.Gns), k, n);
    if (rc) return rc;
    Xcur = (double*)(base + L.X2);
    Ub = (double*)(base + L.X);          // free again: the left vectors on the band, k x n
  }
  T* Uu = U;                           // m x k, transformed in place
  T* Vv = (T*)(base + L.Vv);           // n x k
  hipLaunchKernelGGL((uv_init_kernel<T>), dim3((unsigned)k), dim3(256), 0, stream(), (const double*)(base + L.Bd),
                     (const double*)Xcur, (const double*)(base + L.shifts), (const double*)(base + L.scal), m, n,
                     k, Uu, Vv, ray, Ub, F64 ? 1e-6 : 5e-6, F64 ? 1e-10 : 1e-6, F64 ? 1e-5 : 1e-6,
                     (int*)(base + L.status));
  TNH_LAUNCH_CHECK();
  if (F64 && g_ns64) {
    // u = B v / |B v| inherits eps64 (s_1 / s)^2 of non-orthogonality from v's error along its neighbours: same cure
    rc = newton_schulz(Ub, (double*)(base + L.X3), (double*)(base + L.Gns), k, n);
    if (rc) return rc;
    hipLaunchKernelGGL((ub_out_kernel<T>), dim3((unsigned)k), dim3(256), 0, stream(), (const double*)(base + L.X3), m, n,
                       k, Uu);
    TNH_LAUNCH_CHECK();
  }
  if (S_kept)     // the caller's S holds the coarse values of every bracket; the kept ones are known better now
    hipLaunchKernelGGL((values_kept_kernel<T>), dim3((unsigned)((k + 255) / 256)), dim3(256), 0, stream(),
                       (const double*)(base + L.lo), (const double*)(base + L.hi), (const double*)ray, n, k, S_kept);
  T* Wpart = (T*)(base + L.Wpart);
  T* Wt = (T*)(base + L.Wt);
  constexpr int WRC = w_rc<T>();
  const size_t lds_u = (size_t)m * BT_COLS * sizeof(T), lds_v = (size_t)n * BT_COLS * sizeof(T);
  if (g_bt_fused && lds_u <= 150 * 1024) {
    // one launch per side: every workgroup carries BT_COLS columns through all the reflectors
    static bool attr_done = false;
    if (!attr_done) {
      TNH_HIP(hipFuncSetAttribute((const void*)backtransform_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  150 * 1024));
      attr_done = true;
    }
    // the two sides are independent: ONE launch, blockIdx.y picks the side (k / 2 workgroups each: together they
    // fill the chip)
    BtArgs<T> a;
    a.side[0] = BtSide<T>{Uu, k, m, (const T*)(base + L.Vl), (const double*)(base + L.Tl), L.np, (int64_t)0, m, 0};
    a.side[1] = BtSide<T>{Vv, k, n, (const T*)(base + L.Vr), (const double*)(base + L.Tr), L.np - 1, (int64_t)16, n - 16, 0};
    hipLaunchKernelGGL((backtransform_kernel<T>), dim3((unsigned)(k / BT_COLS), 2), dim3(BT_THREADS),
                       lds_u > lds_v ? lds_u : lds_v, stream(), a);
    hipLaunchKernelGGL((transpose_out_kernel<T>), dim3((unsigned)((n + 31) / 32), (unsigned)((k + 31) / 32)), dim3(256), 0,
                       stream(), (const T*)Vv, n, k, Vh);
    TNH_LAUNCH_CHECK();
    return TNH_OK;
  } else {
  // U = Q_L [U_b; 0]: column-panel reflectors, last to first
  for (int64_t p = L.np - 1; p >= 0; --p) {
    const int64_t j = 16 * p, mj = m - j;
    const T* V = (const T*)(base + L.Vl) + vl_offset(m, p);
    T* C = Uu + j * k;
    const int chunks = (int)((mj + WRC - 1) / WRC);
    hipLaunchKernelGGL((wpass_kernel<T>), dim3((unsigned)((k + 16 * w_cpl<T>() - 1) / (16 * w_cpl<T>())), chunks), dim3(256), 0, stream(), (const T*)C, k,
                       mj, k, V, Wpart);
    hipLaunchKernelGGL((wreduce_kernel<T>), dim3((unsigned)((k + 63) / 64)), dim3(256), 0, stream(), (const T*)Wpart,
                       chunks, k, (const double*)(base + L.Tl) + p * 256, 0, Wt);
    hipLaunchKernelGGL((update_kernel<0, T>), dim3((unsigned)((k + 63) / 64), (unsigned)((mj + U_RR - 1) / U_RR)), dim3(256), 0,
                       stream(), C, k, mj, k, V, (const T*)Wt, k, (double*)nullptr, (int64_t)0);
  }
  TNH_LAUNCH_CHECK();
  // V = Q_R V_b: row-panel reflectors, last to first
  for (int64_t p = L.np - 2; p >= 0; --p) {
    const int64_t j = 16 * (p + 1), nj = n - j;
    const T* V = (const T*)(base + L.Vr) + vr_offset(n, p);
    T* C = Vv + j * k;
    const int chunks = (int)((nj + WRC - 1) / WRC);
    hipLaunchKernelGGL((wpass_kernel<T>), dim3((unsigned)((k + 16 * w_cpl<T>() - 1) / (16 * w_cpl<T>())), chunks), dim3(256), 0, stream(), (const T*)C, k,
                       nj, k, V, Wpart);
    hipLaunchKernelGGL((wreduce_kernel<T>), dim3((unsigned)((k + 63) / 64)), dim3(256), 0, stream(), (const T*)Wpart,
                       chunks, k, (const double*)(base + L.Tr) + p * 256, 0, Wt);
    hipLaunchKernelGGL((update_kernel<0, T>), dim3((unsigned)((k + 63) / 64), (unsigned)((nj + U_RR - 1) / U_RR)), dim3(256), 0,
                       stream(), C, k, nj, k, V, (const T*)Wt, k, (double*)nullptr, (int64_t)0);
  }
  }
  hipLaunchKernelGGL((transpose_out_kernel<T>), dim3((unsigned)((n + 31) / 32), (unsigned)((k + 31) / 32)), dim3(256), 0,
                     stream(), (const T*)Vv, n, k, Vh);
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

// ------------------------------------------------------------------------------------------------ QR on the panels
// Thin Householder QR of an f32 matrix with m >= n, n % 16 == 0 (K9 fast path, round 3): the column-panel half of
// stage 1 -- Gram partials, one-workgroup factor, V rows, rank-16 update -- and Q = H_0 ... H_(np-1) [I; 0] by the
// column-owning back-transformation.  Same reflector sign rule as LAPACK's geqrf (S_jj = -sign of the pivot
// candidate), so R matches np.linalg.qr.  16 launches per 16 columns instead of K9's ~30 per column.
template <typename T>
__global__ __launch_bounds__(256) void qr_out_kernel(const T* __restrict__ Af, const double* __restrict__ Dblk,
                                                     int64_t m, int64_t n, T* __restrict__ R, T* __restrict__ Q) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n * n) {                       // R (n x n): diagonal blocks from the factor kernel, the rest from Af
    const int64_t i = e / n, c = e % n;
    const int64_t p = i / 16;
    T v = T(0);
    if (c >= 16 * (p + 1)) v = Af[i * n + c];
    else if (c >= i) v = (T)Dblk[p * 256 + (i - 16 * p) * 16 + (c - 16 * p)];
    R[e] = v;
  }
  if (e < m * n) Q[e] = (e / n == e % n) ? T(1) : T(0);       // thin identity, transformed afterwards
}

// The leading panels of the Householder QR by the fused kernels (see qr_lookahead_kernel).  Panels 0 .. *p_next - 1 are
// done here (their V formed, R rows in place); the caller's loop continues at *p_next with *parts partial Grams in Gpart.
template <typename T>
static int qr_panels_fast(const Layout& L, char* base, int64_t m, int64_t n, int64_t* p_next, int* parts_out) {
  constexpr bool F64 = sizeof(T) == 8;
  constexpr int CWB = F64 ? 32 : 64;
  T* Af = (T*)(base + L.Af);
  double* Gc = (double*)(base + L.Gpart);
  double* Gr = (double*)(base + L.Gpart2);
  double* Gq = (double*)(base + L.Gq);
  double* R1 = (double*)(base + L.R1);
  T* Part = (T*)(base + L.Wpart);
  T* Wx = (T*)(base + L.Wt);
  T* Vl = (T*)(base + L.Vl);
  double* Xl = (double*)(base + L.Xl);
  double* Tl = (double*)(base + L.Tl);
  int* status = (int*)(base + L.status);
  T* Ffc = (T*)(base + L.Ff);
  const double rel = F64 ? 1e-3 : 1e-9;
  const double cond = F64 ? 0.0 : g_fast_cond;
  // fast while the panel has more than `switch` rows and a trailing block of at least 16 columns
  int64_t ps = 0;
  while (ps < L.np && m - 16 * ps > g_fast_switch && n - 16 * ps - 16 >= 16) ++ps;
  *p_next = 0;
  if (ps < 2) return TNH_OK;
  int parts = (int)((m + 255) / 256);
  hipLaunchKernelGGL((gram_kernel<false, T>), dim3(parts), dim3(256), 0, stream(), (const T*)Af, n, m, Gc);
  for (int64_t p = 0; p <= ps; ++p) {
    const int64_t j = 16 * p, mj = m - j, nc = n - j - 16;
    const bool last = p == ps;
    if (p > 0) {
      // panel p's 16 columns (rows j ..) take panel p - 1's update now; its partial Grams
      parts = (int)((mj + 63) / 64);
      hipLaunchKernelGGL((qr_lookahead_kernel<T>), dim3((unsigned)parts), dim3(256), 0, stream(), Af + j * n + j,
                         (const T*)(Af + j * n + j - 16), n, mj, (const T*)Wx, n, Gc);
    }
    if (last) {
      // hand-over: the rest of the block takes the pending update, the caller's loop goes on with panel ps
      if (nc > 0) {
        RowUpdArgs<T> a{};
        a.C = Af + j * n + j + 16;
        a.ldc = n; a.rows = mj; a.nc = nc;
        a.Zx = Af + j * n + j - 16; a.zx_ld = n;
        a.Pr = Wx + 16; a.pr_ld = n;
        a.Pcol = Af; a.Wpart = Part;
        a.ntc = (int)((nc + CWB - 1) / CWB);
        a.iters = fast_iters(mj, a.ntc);
        const int nrt = (int)(((mj + 63) / 64 + a.iters - 1) / a.iters);
        launch_rowupd<T, CWB>(a, (unsigned)(nrt * a.ntc), true, false);
      }
      break;
    }
    const double* gram = Gc;
    int gparts = parts;
    if constexpr (F64) {
      const int qparts = (int)((mj + 255) / 256);
      hipLaunchKernelGGL((cholscale_kernel<false>), dim3(qparts), dim3(256), 0, stream(), (double*)(Af + j * n + j), n, mj,
                         (const double*)Gc, parts, R1, Gq, status);
      gram = Gq;
      gparts = qparts;
    }
    int nrt;
    {
      RowUpdArgs<T> a{};
      a.C = Af + j * n + j + 16;
      a.ldc = n; a.rows = mj; a.nc = nc;
      a.Zx = p > 0 ? Af + j * n + j - 16 : Af; a.zx_ld = n;      // panel p - 1's rows j .. (below its top block)
      a.Pr = Wx + 16; a.pr_ld = n;                               // its Wx beyond the 16 columns the look-ahead took
      a.Pcol = Af + j * n + j;
      a.Wpart = Part;
      a.ntc = (int)((nc + CWB - 1) / CWB);
      a.iters = fast_iters(mj, a.ntc);
      nrt = (int)(((mj + 63) / 64 + a.iters - 1) / a.iters);
      a.fa = FactorArgs<T>{Af + j * n + j, n, mj, gram, gparts, Xl + p * 256, Vl + vl_offset(m, p), Tl + p * 256,
                           (double*)(base + L.Dblk) + p * 256, F64 ? (const double*)R1 : (const double*)nullptr, rel, cond,
                           status, Ffc};
      launch_rowupd<T, CWB>(a, (unsigned)(nrt * a.ntc) + 1u, p > 0, true);
    }
    hipLaunchKernelGGL((colreduce_kernel<T>), dim3((unsigned)((nc + 63) / 64)), dim3(256), 0, stream(), (const T*)Part, nrt, nc,
                       Af + j * n + j + 16, n, (const T*)(Vl + vl_offset(m, p)), (const T*)Ffc, Wx, n, Gr);
    TNH_LAUNCH_CHECK();
  }
  {
    FormVAllArgs<T> a{Af, m, n, ps, Xl, Xl, Vl, Vl};
    hipLaunchKernelGGL((formv_all_kernel<T>), dim3((unsigned)((m - 16 + 255) / 256), (unsigned)ps, 1), dim3(256), 0, stream(), a);
  }
  TNH_LAUNCH_CHECK();
  *p_next = ps;
  *parts_out = parts;
  return TNH_OK;
}


// T = float (round 3) or double (round 4: two Cholesky-QR passes per panel, as in the f64 band reduction)
template <typename T>
static int qr_panels(int64_t m, int64_t n, const T* A, T* Q, T* R, char* base, int* status_host) {
  constexpr bool QR2 = sizeof(T) == 8;
  constexpr int WRC = w_rc<T>();
  const Layout L = make_layout(m, n, 4, (int)sizeof(T));
  T* Af = (T*)(base + L.Af);
  double* Gc = (double*)(base + L.Gpart);
  double* Gq = (double*)(base + L.Gq);
  double* R1 = (double*)(base + L.R1);
  double* R1inv = (double*)(base + L.R1inv);
  double* Xb = (double*)(base + L.Xbuf);
  T* Wpart = (T*)(base + L.Wpart);
  T* Wt = (T*)(base + L.Wt);
  int* status = (int*)(base + L.status);
  const int64_t np = L.np;
  // (very tall inputs -- 65536 x 256 -- gain nothing: 2.69 against 2.59 ms; profiles/r06_qr_fast_probe.jsonl)
  bool fast = g_fast && !capturing() && (!QR2 || g_fast64) && m <= 32 * n;
  int64_t p0 = 0;
  int parts = 0;
 again:
  TNH_HIP(hipMemcpyAsync(Af, A, (size_t)m * n * sizeof(T), hipMemcpyDeviceToDevice, stream()));
  TNH_HIP(hipMemsetAsync(status, 0, 64, stream()));
  p0 = 0;
  parts = (int)((m + 255) / 256);
  if (fast) {
    // round 6: the leading panels by the fused kernels of tnh_svd_band_fast.inc -- three launches per panel (look-ahead,
    // previous panel's update + raw pass + factor, reduce) instead of five (f64: four instead of seven)
    const int rc = qr_panels_fast<T>(L, base, m, n, &p0, &parts);
    if (rc) return rc;
  }
  if (p0 == 0) hipLaunchKernelGGL((gram_kernel<false, T>), dim3(parts), dim3(256), 0, stream(), (const T*)Af, n, m, Gc);
  for (int64_t p = p0; p < np; ++p) {
    const int64_t j = 16 * p, mj = m - j, nc = n - j - 16;
    T* P = Af + j * n + j;
    T* V = (T*)(base + L.Vl) + vl_offset(m, p);
    double* Tp = (double*)(base + L.Tl) + p * 256;
    if (QR2) {
      const int qparts = (int)((mj + 255) / 256);
      hipLaunchKernelGGL(chol_kernel, dim3(1), dim3(256), 0, stream(), (const double*)Gc, parts, R1, R1inv, status);
      hipLaunchKernelGGL((scaleq_kernel<false, T>), dim3(qparts), dim3(256), 0, stream(), P, n, mj, (const double*)R1inv, Gq);
      hipLaunchKernelGGL((factor_kernel<false, T>), dim3(1), dim3(256), 0, stream(), (const T*)P, n, mj, (const double*)Gq,
                         qparts, Xb, V, (T*)nullptr, (int64_t)0, Tp, (double*)(base + L.Dblk) + p * 256,
                         (const double*)R1, 1e-3, status);
    } else {
      hipLaunchKernelGGL((factor_kernel<false, T>), dim3(1), dim3(256), 0, stream(), (const T*)P, n, mj, (const double*)Gc,
                         parts, Xb, V, (T*)nullptr, (int64_t)0, Tp, (double*)(base + L.Dblk) + p * 256,
                         (const double*)nullptr, 1e-9, status);
    }
    if (mj > 16)
      hipLaunchKernelGGL((formv_kernel<false, T>), dim3((unsigned)((mj - 16 + 255) / 256)), dim3(256), 0, stream(),
                         (const T*)P, n, mj, (const double*)Xb, V, (T*)nullptr, (int64_t)0);
    if (nc > 0) {
      T* C = Af + j * n + j + 16;
      const int chunks = (int)((mj + WRC - 1) / WRC);
      hipLaunchKernelGGL((wpass_kernel<T>), dim3((unsigned)((nc + 16 * w_cpl<T>() - 1) / (16 * w_cpl<T>())), chunks), dim3(256), 0, stream(), (const T*)C,
                         n, mj, nc, (const T*)V, Wpart);
      hipLaunchKernelGGL((wreduce_kernel<T>), dim3((unsigned)((nc + 63) / 64)), dim3(256), 0, stream(), (const T*)Wpart,
                         chunks, nc, (const double*)Tp, 1, Wt);
      // the update leaves the partial Grams of the next panel: first 16 columns of the block, rows 16 .. (the first
      // 16 rows are R)
      const dim3 grid((unsigned)((nc + 63) / 64), (unsigned)((mj + U_RR - 1) / U_RR));
      hipLaunchKernelGGL((update_kernel<1, T>), grid, dim3(256), 0, stream(), C, n, mj, nc, (const T*)V,
                         (const T*)Wt, nc, Gc, (int64_t)16);
      parts = (int)grid.y;
    }
    TNH_LAUNCH_CHECK();
  }
  const int64_t ne = m * n;
  hipLaunchKernelGGL((qr_out_kernel<T>), dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, stream(), (const T*)Af,
                     (const double*)(base + L.Dblk), m, n, R, Q);
  TNH_LAUNCH_CHECK();
  // ---- Q = H_0 ... H_(np-1) [I; 0]
  const size_t lds = (size_t)m * BT_COLS * sizeof(T);
  if (lds <= 150 * 1024) {
    static bool attr_done = false;
    if (!attr_done) {
      TNH_HIP(hipFuncSetAttribute((const void*)backtransform_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  150 * 1024));
      attr_done = true;
    }
    BtArgs<T> a;
    a.side[0] = BtSide<T>{Q, n, m, (const T*)(base + L.Vl), (const double*)(base + L.Tl), np, (int64_t)0, m, 1};
    a.side[1] = a.side[0];
    hipLaunchKernelGGL((backtransform_kernel<T>), dim3((unsigned)(n / BT_COLS), 1), dim3(BT_THREADS), lds, stream(), a);
  } else {
    // very tall inputs: the columns do not fit in LDS -- per panel W = V^T Q, W <- T W, Q -= V W (last panel first)
    for (int64_t p = np - 1; p >= 0; --p) {
      const int64_t j = 16 * p, mj = m - j;
      const T* V = (const T*)(base + L.Vl) + vl_offset(m, p);
      T* C = Q + j * n + j;               // columns before j are untouched by panel p (identity start)
      const int64_t kc = n - j;
      const int chunks = (int)((mj + WRC - 1) / WRC);
      hipLaunchKernelGGL((wpass_kernel<T>), dim3((unsigned)((kc + 16 * w_cpl<T>() - 1) / (16 * w_cpl<T>())), chunks), dim3(256), 0, stream(), (const T*)C,
                         n, mj, kc, V, Wpart);
      hipLaunchKernelGGL((wreduce_kernel<T>), dim3((unsigned)((kc + 63) / 64)), dim3(256), 0, stream(), (const T*)Wpart,
                         chunks, kc, (const double*)(base + L.Tl) + p * 256, 0, Wt);
      hipLaunchKernelGGL((update_kernel<0, T>), dim3((unsigned)((kc + 63) / 64), (unsigned)((mj + U_RR - 1) / U_RR)),
                         dim3(256), 0, stream(), C, n, mj, kc, V, (const T*)Wt, kc, (double*)nullptr, (int64_t)0);
    }
  }
  TNH_LAUNCH_CHECK();
  TNH_HIP(hipMemcpyAsync(status_host, status, sizeof(int), hipMemcpyDeviceToHost, stream()));
  TNH_HIP(hipStreamSynchronize(stream()));
  if (fast && (*status_host & (int)ST_FASTCOND)) {      // an ill-conditioned panel under the raw-panel passes: once more, the old loop
    fast = false;
    goto again;
  }
  *status_host &= ~(int)ST_FASTCOND;
  return TNH_OK;
}

}  // namespace svdb

// entry points for tnh_qr.hip
bool qr_panel16_supported(int dtype, int64_t m, int64_t n) {
  const char* e = getenv("TNH_QR_PANEL16");
  if (e && e[0] == '0') return false;
  return (dtype == TNH_F32 || dtype == TNH_F64) && m >= n && n >= 64 && (n % 16) == 0;
}
size_t qr_panel16_work_bytes(int dtype, int64_t m, int64_t n) {
  return svdb::make_layout(m, n, 4, dtype == TNH_F64 ? 8 : 4).total + 256;
}
int qr_panel16(int dtype, int64_t m, int64_t n, const void* A, void* Q, void* R, void* work, int* status_host) {
  svdb::read_env();
  char* base = (char*)(((uintptr_t)work + 255) & ~(uintptr_t)255);
  if (dtype == TNH_F64)
    return svdb::qr_panels<double>(m, n, (const double*)A, (double*)Q, (double*)R, base, status_host);
  return svdb::qr_panels<float>(m, n, (const float*)A, (float*)Q, (float*)R, base, status_host);
}
}  // namespace tnh

using namespace tnh;
using namespace tnh::svdb;

extern "C" {

int tnh_svd_band_supported(int dtype, int64_t m, int64_t n, int64_t k) {
  if (dtype != TNH_F32 && dtype != TNH_F64) return 0;
  if (m < n) return 0;                       // the caller passes the tall orientation
  if (n < 256 || (n % 16) != 0) return 0;
  if (k < 0 || k > n) return 0;
  if (k > 0 && (k % 4) != 0) return 0;       // four-element columns in the back-transformation
  return 1;
}

int tnh_svd_band_work_bytes(int dtype, int64_t m, int64_t n, int64_t kcap, size_t* nbytes) {
  TNH_REQUIRE(nbytes != nullptr, "null nbytes");
  TNH_REQUIRE((dtype == TNH_F32 || dtype == TNH_F64) && m >= n && n >= 32 && n % 16 == 0 && kcap >= 0 && kcap <= n,
              "tnh_svd_band: unsupported dtype %d / shape %lld x %lld", dtype, (long long)m, (long long)n);
  *nbytes = make_layout(m, n, kcap > 4 ? kcap : 4, dtype == TNH_F64 ? 8 : 4).total + 256;
  return TNH_OK;
}

int tnh_svd_band_layout(int dtype, int64_t m, int64_t n, int64_t kcap, int64_t* offsets, int count) {
  TNH_REQUIRE(offsets != nullptr && count >= 12, "tnh_svd_band_layout: need room for 12 offsets");
  const Layout L = make_layout(m, n, kcap > 4 ? kcap : 4, dtype == TNH_F64 ? 8 : 4);
  const size_t o[16] = {L.Af, L.Vl, L.Vr, L.Tl, L.Tr, L.Dblk, L.Eblk, L.Bd, L.Tb, L.lo, L.hi, L.X, L.shifts, L.status, L.X2,
                        L.Gns};
  for (int i = 0; i < count && i < 16; ++i) offsets[i] = (int64_t)o[i];
  return TNH_OK;
}

int tnh_svd_band_factor(int dtype, int64_t m, int64_t n, const void* A, void* S, void* work, int64_t kcap,
                        int* status_out) {
  TNH_NEED_INIT();
  TNH_REQUIRE(A && S && work, "null pointer");
  TNH_REQUIRE(tnh_svd_band_supported(dtype, m, n, 0), "tnh_svd_band_factor: unsupported dtype %d / shape %lld x %lld",
              dtype, (long long)m, (long long)n);
  TNH_REQUIRE(!(status_out && capturing()), "tnh_svd_band_factor: a status read-back synchronises the stream (graph capture)");
  read_env();
  char* base = (char*)(((uintptr_t)work + 255) & ~(uintptr_t)255);
  const int esz = dtype == TNH_F64 ? 8 : 4;
  const Layout L = make_layout(m, n, kcap > 4 ? kcap : 4, esz);
  // f32: the fast stage 1 first (raw-panel passes, four launches per pair of panels).  It cannot know beforehand
  // whether every panel is well enough conditioned for it, so the whole factor stage is enqueued speculatively and
  // the status word read at the end (one stream synchronisation, also when the caller did not ask for the status);
  // ST_FASTCOND repeats the stage with the loop of rounds 3-5.  Under graph capture nothing can be read back, the
  // speculation cannot be checked: the accurate loop runs directly.
  bool fast = g_fast && !capturing() && (esz == 4 || g_fast64);
  g_last_fast = 0;
  // Inputs whose panels are too ill-conditioned for the fast stage (numerically rank-deficient blocks: two-site DMRG
  // splits, zero-padded tensors) come in runs of the same shape; a shape that reported ST_FASTCOND f times in a row
  // goes straight to the accurate loop for the next min(2^f, 64) calls instead of paying both stages every time.
  static std::map<std::pair<int64_t, int64_t>, std::pair<int, int>> backoff;      // (m, n) -> (failures in a row, calls to skip)
  const std::pair<int64_t, int64_t> key(m, n);
  if (fast) {
    auto it = backoff.find(key);
    if (it != backoff.end() && it->second.second > 0) {
      --it->second.second;
      fast = false;
      g_last_fast = 3;
    }
  }
  for (int attempt = 0; attempt < 2; ++attempt) {
    TNH_HIP(hipMemcpyAsync(base + L.Af, A, (size_t)m * n * esz, hipMemcpyDeviceToDevice, stream()));
    TNH_HIP(hipMemsetAsync(base + L.status, 0, 64, stream()));
    TNH_HIP(hipMemsetAsync(base + L.Eblk, 0, (size_t)L.np * 256 * 8, stream()));
    int rc;
    if (fast) {
      int64_t p_next = 0;
      int parts = 0;
      rc = esz == 8 ? stage1_fast<double>(L, base, m, n, &p_next, &parts) : stage1_fast<float>(L, base, m, n, &p_next, &parts);
      if (rc) return rc;
      rc = esz == 8 ? stage1<double>(L, base, m, n, p_next, parts) : stage1<float>(L, base, m, n, p_next, parts);
      g_last_fast = p_next > 0 ? 1 : 0;
    } else {
      rc = esz == 8 ? stage1<double>(L, base, m, n) : stage1<float>(L, base, m, n);
    }
    if (rc) return rc;
    rc = esz == 8 ? values<double>(L, base, n, (double*)S) : values<float>(L, base, n, (float*)S);
    if (rc) return rc;
    if (!status_out && !fast) break;
    int st = 0;
    TNH_HIP(hipMemcpyAsync(&st, base + L.status, sizeof(int), hipMemcpyDeviceToHost, stream()));
    TNH_HIP(hipStreamSynchronize(stream()));
    if (fast && (st & (int)ST_FASTCOND)) {      // an ill-conditioned panel: once more, accurately
      fast = false;
      g_last_fast = 2;
      auto& e = backoff[key];
      e.first = e.first < 6 ? e.first + 1 : 6;
      e.second = 1 << e.first;
      continue;
    }
    if (fast) backoff.erase(key);
    if (status_out) *status_out = st & ~(int)ST_FASTCOND;
    break;
  }
  return TNH_OK;
}

// 0: the last tnh_svd_band_factor ran the loop of rounds 3-5; 1: the fast stage 1; 2: the fast stage 1 reported an
// ill-conditioned panel and the stage was repeated with the accurate loop; 3: the accurate loop directly, because the
// shape's previous calls kept reporting (back-off).
int tnh_svd_band_last_stage1(void) { return g_last_fast; }

int tnh_svd_band_vectors(int dtype, int64_t m, int64_t n, void* work, int64_t kcap, int64_t k, void* U, void* Vh,
                         void* S_kept, int* status_out) {
  TNH_NEED_INIT();
  TNH_REQUIRE(work && U && Vh, "null pointer");
  TNH_REQUIRE(k > 0 && k <= kcap && tnh_svd_band_supported(dtype, m, n, k),
              "tnh_svd_band_vectors: unsupported k = %lld (kcap %lld) for %lld x %lld", (long long)k, (long long)kcap,
              (long long)m, (long long)n);
  TNH_REQUIRE(!(status_out && capturing()), "tnh_svd_band_vectors: a status read-back synchronises the stream (graph capture)");
  read_env();
  char* base = (char*)(((uintptr_t)work + 255) & ~(uintptr_t)255);
  const int esz = dtype == TNH_F64 ? 8 : 4;
  const Layout L = make_layout(m, n, kcap > 4 ? kcap : 4, esz);
  int rc = esz == 8 ? vectors<double>(L, base, m, n, k, (double*)U, (double*)Vh, (double*)S_kept)
                    : vectors<float>(L, base, m, n, k, (float*)U, (float*)Vh, (float*)S_kept);
  if (rc) return rc;
  if (status_out) {
    int st = 0;
    TNH_HIP(hipMemcpyAsync(&st, base + L.status, sizeof(int), hipMemcpyDeviceToHost, stream()));
    TNH_HIP(hipStreamSynchronize(stream()));
    *status_out = st;
  }
  return TNH_OK;
}

}  // extern "C"
