// K9: thin Householder QR (f32 / f64), the device side of AbstractBackend.qr / rq
// (abstract_backend.py:139-153; oracle decompositions.py:77-124 = np.linalg.qr, LAPACK
// geqrf + orgqr).  Same algorithm as LAPACK so that R comes out with LAPACK's signs:
//
//   for each panel of 32 columns:
//     qr_panel_kernel   unblocked Householder (larfg / larf) on the m_j x 32 panel by ONE
//                       workgroup of 1024 threads (32 columns x 32 row lanes; a row of the
//                       panel is one 128-B line; v_r reaches the other columns of its row
//                       by a wave shuffle).  Two passes over the panel per column: the
//                       w = v^T P pass, and the rank-1 update pass fused with the norm of
//                       the next column.  Then S = V^T V (one pass) and the 32 x 32 T of
//                       the compact WY form I - V T V^T (larft) by recurrence in LDS.
//     trailing update   C <- (I - V T^T V^T) C as three GEMMs on the matrix pipe (larfb):
//                       W = V^T C, W <- T^T W, C <- C - V W   (tnh_gemm_ex, alpha/beta).
//   Q (m x k) is then built like orgqr: Q = I_thin, panels applied last to first.
//
// Bound: the panel passes stream m_j x 32 elements through one CU (L2-resident), the
// updates are skinny (K = 32) f32 / f64 MFMA GEMMs, HBM-bound at 2 * m_j * n_t elements each.
#include <stdlib.h>
#include "tnh_types.h"

namespace tnh {

constexpr int QB = 32;  // panel width
static bool g_qr_tall = true;  // multi-workgroup panels for m_j > 512 (TNH_QR_TALL=0 disables: A/B)

template <typename T>
__device__ __forceinline__ double block_sum_1024(double v, double* red) {
  // red: >= 16 doubles of LDS.  All 1024 threads call.
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += red[i];
  return s;
}

// A: full matrix (row-major, lda); panel = rows j0.., columns j0 .. j0+jb-1.
// Outputs: panel overwritten with R (upper) / V (strictly lower), tau[j0 ..], Tm (32x32,
// row-major, zero outside the upper triangle), Vp ((m-j0) x 32 clean V: unit diagonal, zeros above).
template <typename T>
__global__ __launch_bounds__(1024) void qr_panel_kernel(T* __restrict__ A, int64_t lda, int64_t m, int64_t j0,
                                                        int jb, T* __restrict__ tau, T* __restrict__ Tm,
                                                        T* __restrict__ Vp) {
  __shared__ double red[16];
  __shared__ double colred[32][33];
  __shared__ double w_s[32];
  __shared__ double S_s[32][33];
  __shared__ double T_s[32][33];
  __shared__ double tau_s[32];
  __shared__ double sc_tau, sc_scale, sc_beta;
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5, lane = tid & 63;
  const int64_t mj = m - j0;
  T* P = A + j0 * lda + j0;  // panel origin
  const bool col_ok = tx < jb;

  // norm^2 of column 0 below the diagonal
  double part = 0.0;
  for (int64_t li = 1 + tid; li < mj; li += 1024) {
    const double x = (double)P[li * lda];
    part += x * x;
  }
  double xnorm2 = block_sum_1024<T>(part, red);

  for (int c = 0; c < jb; ++c) {
    // ---- larfg on column c (thread 0), rows c .. mj-1
    if (tid == 0) {
      const double alpha = (c < mj) ? (double)P[(int64_t)c * lda + c] : 0.0;
      double t = 0.0, scale = 0.0, beta = alpha;
      if (xnorm2 > 0.0 && c < mj) {
        beta = -copysign(sqrt(alpha * alpha + xnorm2), alpha);
        t = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
      }
      sc_tau = t; sc_scale = scale; sc_beta = beta;
      tau_s[c] = t;
    }
    __syncthreads();
    const double t_c = sc_tau, scale = sc_scale;
    // ---- pass W: v = x * scale stored in place; w_j = P[c][j] + sum_r v_r P[r][j]  (j > c)
    double wp = 0.0;
    for (int64_t li = c + 1 + ty; li < mj; li += 32) {
      T a = col_ok ? P[li * lda + tx] : (T)0;
      const T x = __shfl(a, (lane & 32) | c, 64);
      const T v = (T)((double)x * scale);
      if (tx == c) P[li * lda + c] = v;
      else if (tx > c) wp += (double)v * (double)a;
    }
    colred[ty][tx] = wp;
    __syncthreads();
    if (tid < 32) {
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < 32; ++i) s += colred[i][tid];
      if (tid > c && tid < jb && c < mj) s += (double)P[(int64_t)c * lda + tid];
      w_s[tid] = s;
    }
    __syncthreads();
    // ---- pass U: P[r][j] -= tau v_r w_j (j > c); pivot row: v = 1; next column's norm
    const double tw = (tx > c && col_ok) ? t_c * w_s[tx] : 0.0;
    double n2 = 0.0;
    if (ty == 0 && c < mj) {
      if (tx > c && col_ok) P[(int64_t)c * lda + tx] = (T)((double)P[(int64_t)c * lda + tx] - tw);
      if (tx == c) P[(int64_t)c * lda + c] = (T)sc_beta;
    }
    for (int64_t li = c + 1 + ty; li < mj; li += 32) {
      T a = col_ok ? P[li * lda + tx] : (T)0;
      const T v = __shfl(a, (lane & 32) | c, 64);
      if (tx > c && col_ok) {
        a = (T)((double)a - (double)v * tw);
        P[li * lda + tx] = a;
        if (tx == c + 1 && li > c + 1) n2 += (double)a * (double)a;
      }
    }
    xnorm2 = block_sum_1024<T>(n2, red);
  }

  // ---- clean V panel and S = V^T V
  double sacc[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) sacc[k] = 0.0;
  for (int64_t li = ty; li < mj; li += 32) {
    T v = (T)0;
    if (col_ok) {
      if (li > tx) v = P[li * lda + tx];
      else if (li == tx) v = (T)1;
    }
    Vp[li * QB + tx] = v;
#pragma unroll
    for (int k = 0; k < 32; ++k) sacc[k] += (double)v * (double)__shfl(v, (lane & 32) | k, 64);
  }
  // reduce sacc over the 32 row lanes, one k at a time
  for (int k = 0; k < 32; ++k) {
    __syncthreads();
    colred[ty][tx] = sacc[k];
    __syncthreads();
    if (tid < 32) {
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < 32; ++i) s += colred[i][tid];
      S_s[tid][k] = s;   // S[tx][k]
    }
  }
  __syncthreads();
  // ---- T (larft, forward / columnwise): T[c][c] = tau_c, T[0:c, c] = -tau_c T[0:c, 0:c] S[0:c, c]
  if (tid < 32)
#pragma unroll
    for (int k = 0; k < 32; ++k) T_s[tid][k] = 0.0;
  __syncthreads();
  for (int c = 0; c < jb; ++c) {
    if (tid < c) {
      double s = 0.0;
      for (int k = tid; k < c; ++k) s += T_s[tid][k] * S_s[k][c];
      T_s[tid][c] = -tau_s[c] * s;
    } else if (tid == c) {
      T_s[c][c] = tau_s[c];
    }
    __syncthreads();
  }
  if (tid < 32) {
    if (tid < jb) tau[j0 + tid] = (T)tau_s[tid];
  }
  Tm[ty * QB + tx] = (T)((ty < jb && tx < jb) ? T_s[ty][tx] : 0.0);
}

// ---------------------------------------------------------------------------
// Tall panels (m_j > 512): the same Householder recurrence spread over G workgroups, ONE
// launch per column (a dependent kernel boundary costs ~1.5 us on the device, less than any
// grid-wide barrier, and cannot deadlock).  The trick that makes one pass per column enough:
// the dot products  d_j = sum_{r>c} P[r][c] P[r][j]  (j >= c) give both the norm of the
// sub-column (j = c) and, scaled, every w_j = v^T P[:, j]; and the launch that applies
// reflector c accumulates the d_j of column c+1 from the rows it has just updated.
//   qr_coldot_kernel   d_j partials of column 0                     (once per panel)
//   qr_col_kernel      reduce partials -> (tau, scale, w) -> update own rows -> next partials
//   qr_vts_kernel      clean V panel + partials of S = V^T V        (once per panel)
//   qr_tmat_kernel     T from S and tau (larft)                     (once per panel, 1 WG)
// Workgroup g owns rows [g * rpw, (g+1) * rpw) of the panel; 256 threads = 8 row lanes x 32 columns.
// ---------------------------------------------------------------------------
constexpr int QR_MAX_WG = 256;

template <typename T>
__device__ __forceinline__ void qr_wg_reduce32(double v, double (*colred)[33], double* out_partial) {
  // sum v over the 8 row lanes for each of the 32 columns; thread tid < 32 writes out_partial[tid]
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  __syncthreads();
  colred[ty][tx] = v;
  __syncthreads();
  if (threadIdx.x < 32) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += colred[i][threadIdx.x];
    out_partial[threadIdx.x] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void qr_coldot_kernel(const T* __restrict__ P, int64_t lda, int64_t mj, int jb,
                                                        int64_t rpw, double* __restrict__ part_out,
                                                        double* __restrict__ row_out) {
  __shared__ double colred[8][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, lane = threadIdx.x & 63;
  const int64_t r0 = (int64_t)blockIdx.x * rpw;
  int64_t r1 = r0 + rpw;
  if (r1 > mj) r1 = mj;
  if (blockIdx.x == 0 && ty == 0) row_out[tx] = (tx < jb) ? (double)P[tx] : 0.0;   // pivot row 0
  double acc = 0.0;
  for (int64_t li = r0 + ty; li < r1; li += 8) {
    const T a = (tx < jb) ? P[li * lda + tx] : (T)0;
    const T x = __shfl(a, (lane & 32) | 0, 64);
    if (li > 0) acc += (double)x * (double)a;
  }
  qr_wg_reduce32<T>(acc, colred, part_out + (int64_t)blockIdx.x * 32);
}

template <typename T>
__global__ __launch_bounds__(256) void qr_col_kernel(T* __restrict__ P, int64_t lda, int64_t mj, int c, int jb,
                                                     int64_t rpw, int nwg, const double* __restrict__ part_in,
                                                     double* __restrict__ part_out, double* __restrict__ tau_out,
                                                     const double* __restrict__ row_in, double* __restrict__ row_out) {
  __shared__ double colred[8][33];
  __shared__ double d_s[32];
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5, lane = tid & 63;
  // 1. reduce the partial dots of column c: 8 row lanes x 32 columns, each lane sums every 8th
  //    workgroup's partial, then the lanes are combined (fixed order: deterministic)
  {
    // all partials of this lane requested before the first add (nwg <= QR_MAX_WG = 256: <= 32 per lane; clamped
    // index, no branch around a load): a load / add loop with a runtime bound takes one L2 round trip per partial
    double pv[QR_MAX_WG / 8];
#pragma unroll
    for (int q = 0; q < QR_MAX_WG / 8; ++q) {
      const int g = ty + 8 * q;
      pv[q] = part_in[(int64_t)(g < nwg ? g : nwg - 1) * 32 + tx];
    }
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < QR_MAX_WG / 8; ++q) s += (ty + 8 * q < nwg) ? pv[q] : 0.0;
    colred[ty][tx] = s;
    __syncthreads();
    if (tid < 32) {
      double r = 0.0;
#pragma unroll
      for (int i = 0; i < 8; ++i) r += colred[i][tid];
      d_s[tid] = r;
    }
  }
  __syncthreads();
  // 2. larfg: every thread derives the same scalars.  The pivot row is read from row_in (written by
  //    the previous launch) -- its owner rewrites P[c][:] below while other workgroups still need it.
  const double alpha = row_in[c];
  const double xnorm2 = d_s[c];
  double t_c = 0.0, scale = 0.0, beta = alpha;
  if (xnorm2 > 0.0) {
    beta = -copysign(sqrt(alpha * alpha + xnorm2), alpha);
    t_c = (beta - alpha) / beta;
    scale = 1.0 / (alpha - beta);
  }
  const bool col_ok = tx < jb;
  double tw = 0.0;   // tau * w_j for this thread's column (j > c)
  const double prow = row_in[tx];
  if (tx > c && col_ok) tw = t_c * (prow + scale * d_s[tx]);
  // 3. update own rows; accumulate the dots of column c+1 over rows > c+1
  const int64_t r0 = (int64_t)blockIdx.x * rpw;
  int64_t r1 = r0 + rpw;
  if (r1 > mj) r1 = mj;
  if (c >= r0 && c < r1 && ty == 0) {   // pivot row (v = 1)
    if (tx > c && col_ok) P[(int64_t)c * lda + tx] = (T)(prow - tw);
    if (tx == c) P[(int64_t)c * lda + c] = (T)beta;
    if (tx == 0) tau_out[c] = t_c;
  }
  double acc = 0.0;
  int64_t start = r0 + ty;
  if (start <= c) start += ((c - start) / 8 + 1) * 8;   // first row > c in this lane
  for (int64_t li = start; li < r1; li += 8) {
    T a = col_ok ? P[li * lda + tx] : (T)0;
    const double x = (double)__shfl(a, (lane & 32) | c, 64);
    const double v = x * scale;
    if (tx == c) {
      a = (T)v;
      P[li * lda + c] = a;
    } else if (tx > c && col_ok) {
      a = (T)((double)a - v * tw);
      P[li * lda + tx] = a;
    }
    if (c + 1 < 32) {
      const T y = __shfl(a, (lane & 32) | (c + 1), 64);
      if (li > c + 1 && tx > c) acc += (double)y * (double)a;
      if (li == c + 1) row_out[tx] = col_ok ? (double)a : 0.0;   // next launch's pivot row
    }
  }
  qr_wg_reduce32<T>(acc, colred, part_out + (int64_t)blockIdx.x * 32);
}

// clean V panel (unit diagonal, zeros above) and the partials of S = V^T V:
// s_part[g][k][tx] = sum over the WG's rows of v[r][k] v[r][tx]
template <typename T>
__global__ __launch_bounds__(256) void qr_vts_kernel(const T* __restrict__ P, int64_t lda, int64_t mj, int jb,
                                                     int64_t rpw, T* __restrict__ Vp, double* __restrict__ s_part) {
  __shared__ double colred[8][33];
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5, lane = tid & 63;
  const int64_t r0 = (int64_t)blockIdx.x * rpw;
  int64_t r1 = r0 + rpw;
  if (r1 > mj) r1 = mj;
  double sacc[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) sacc[k] = 0.0;
  for (int64_t li = r0 + ty; li < r1; li += 8) {
    T v = (T)0;
    if (tx < jb) {
      if (li > tx) v = P[li * lda + tx];
      else if (li == tx) v = (T)1;
    }
    Vp[li * QB + tx] = v;
#pragma unroll
    for (int k = 0; k < 32; ++k) sacc[k] += (double)v * (double)__shfl(v, (lane & 32) | k, 64);
  }
  for (int k = 0; k < 32; ++k)
    qr_wg_reduce32<T>(sacc[k], colred, s_part + ((int64_t)blockIdx.x * 32 + k) * 32);
}

template <typename T>
__global__ __launch_bounds__(1024) void qr_tmat_kernel(const double* __restrict__ s_part, int nwg,
                                                       const double* __restrict__ tau_d, int jb, int64_t j0,
                                                       T* __restrict__ tau, T* __restrict__ Tm) {
  __shared__ double S_s[32][33];
  __shared__ double T_s[32][33];
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  double s = 0.0;
  // 16 partials in flight at a time (clamped index, no branch around a load; same summation order)
  for (int g0 = 0; g0 < nwg; g0 += 16) {
    double pv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int g = g0 + q;
      pv[q] = s_part[((int64_t)(g < nwg ? g : nwg - 1) * 32 + ty) * 32 + tx];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) s += (g0 + q < nwg) ? pv[q] : 0.0;
  }
  S_s[ty][tx] = s;     // S[k = ty][tx]
  T_s[ty][tx] = 0.0;
  __syncthreads();
  for (int c = 0; c < jb; ++c) {
    if (tid < c) {
      double a = 0.0;
      for (int k = tid; k < c; ++k) a += T_s[tid][k] * S_s[k][c];
      T_s[tid][c] = -tau_d[c] * a;
    } else if (tid == c) {
      T_s[c][c] = tau_d[c];
    }
    __syncthreads();
  }
  if (tid < jb) tau[j0 + tid] = (T)tau_d[tid];
  Tm[ty * QB + tx] = (T)((ty < jb && tx < jb) ? T_s[ty][tx] : 0.0);
}

// scratch (doubles): 2 x [QR_MAX_WG][32] ping-pong dot partials, [QR_MAX_WG][32][32] S partials, [32] tau,
// 2 x [32] ping-pong pivot rows
constexpr size_t QR_TALL_SCRATCH = ((size_t)2 * QR_MAX_WG * 32 + (size_t)QR_MAX_WG * 1024 + 32 + 64) * sizeof(double);

template <typename T>
static int qr_panel_tall(T* Af, int64_t lda, int64_t m, int64_t j0, int jb, T* tau, T* Tm, T* Vp, double* scratch) {
  const int64_t mj = m - j0;
  int nwg = (int)((mj + 63) / 64);
  if (nwg > QR_MAX_WG) nwg = QR_MAX_WG;
  const int64_t rpw = ((mj + nwg - 1) / nwg + 7) / 8 * 8;
  nwg = (int)((mj + rpw - 1) / rpw);
  double* part[2] = {scratch, scratch + (size_t)QR_MAX_WG * 32};
  double* s_part = scratch + (size_t)2 * QR_MAX_WG * 32;
  double* tau_d = s_part + (size_t)QR_MAX_WG * 1024;
  double* row[2] = {tau_d + 32, tau_d + 64};
  T* P = Af + j0 * lda + j0;
  hipLaunchKernelGGL((qr_coldot_kernel<T>), dim3(nwg), dim3(256), 0, stream(), (const T*)P, lda, mj, jb, rpw, part[0],
                     row[0]);
  for (int c = 0; c < jb; ++c)
    hipLaunchKernelGGL((qr_col_kernel<T>), dim3(nwg), dim3(256), 0, stream(), P, lda, mj, c, jb, rpw, nwg,
                       (const double*)part[c & 1], part[(c + 1) & 1], tau_d, (const double*)row[c & 1],
                       row[(c + 1) & 1]);
  hipLaunchKernelGGL((qr_vts_kernel<T>), dim3(nwg), dim3(256), 0, stream(), (const T*)P, lda, mj, jb, rpw, Vp, s_part);
  hipLaunchKernelGGL((qr_tmat_kernel<T>), dim3(1), dim3(1024), 0, stream(), (const double*)s_part, nwg,
                     (const double*)tau_d, jb, j0, tau, Tm);
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

// Vp ((m-j0) x 32) <- clean V of the panel stored in the factored matrix.
template <typename T>
__global__ __launch_bounds__(256) void qr_extract_v_kernel(const T* __restrict__ A, int64_t lda, int64_t m, int64_t j0,
                                                           int jb, T* __restrict__ Vp) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t li = e >> 5;
  const int tx = (int)(e & 31);
  if (li >= m - j0) return;
  T v = (T)0;
  if (tx < jb) {
    if (li > tx) v = A[(j0 + li) * lda + j0 + tx];
    else if (li == tx) v = (T)1;
  }
  Vp[li * QB + tx] = v;
}

// R (k x n) = upper triangle of the factored matrix; Q (m x k) = thin identity.
template <typename T>
__global__ __launch_bounds__(256) void qr_triu_kernel(T* __restrict__ R, const T* __restrict__ A, int64_t lda, int64_t k,
                                                      int64_t n) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= k * n) return;
  const int64_t i = e / n, j = e % n;
  R[e] = (j >= i) ? A[i * lda + j] : (T)0;
}
template <typename T>
__global__ __launch_bounds__(256) void qr_eye_kernel(T* __restrict__ Q, int64_t m, int64_t k) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= m * k) return;
  Q[e] = (e / k == e % k) ? (T)1 : (T)0;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// W (32 x nt) = V^T C with V = Vp (mj x 32) and C (mj x nt, ldc): a product with a tiny M and a
// long K.  One GEMM would run nt / 128 workgroups through the whole of K, so K is cut into up to
// QR_SPLITS slices computed as ONE strided-batched GEMM (slice b -> Wpart[b]) and summed by the
// K4 reduction kernel (fixed order: deterministic).
constexpr int QR_SPLITS = 32;
static int qr_vt_times_c(int dt, const void* Vp, const void* C, int64_t ldc, int64_t mj, int64_t nt, void* W,
                         void* Wpart) {
  const int esz = dtype_size(dt);
  int64_t splits = mj / 256;
  if (splits > QR_SPLITS) splits = QR_SPLITS;
  if (splits <= 1)
    return tnh_gemm_ex(dt, dt, 1, 0, QB, nt, mj, Vp, QB, C, ldc, W, nt, 1, 0, 0, 0, 1.0, 0.0);
  const int64_t kc = mj / splits, rem = mj - kc * splits;
  int rc = tnh_gemm_ex(dt, dt, 1, 0, QB, nt, kc, Vp, QB, C, ldc, Wpart, nt, splits, kc * QB, kc * ldc, QB * nt, 1.0,
                       0.0);
  if (rc) return rc;
  int64_t parts = splits;
  if (rem > 0) {
    rc = tnh_gemm_ex(dt, dt, 1, 0, QB, nt, rem, (const char*)Vp + (size_t)kc * splits * QB * esz, QB,
                     (const char*)C + (size_t)kc * splits * ldc * esz, ldc, (char*)Wpart + (size_t)splits * QB * nt * esz,
                     nt, 1, 0, 0, 0, 1.0, 0.0);
    if (rc) return rc;
    ++parts;
  }
  return tnh_sum_mid(W, Wpart, 1, parts, QB * nt, dt);
}

struct QrWork {
  char* Af; char* Vp; char* Tm; char* W; char* W2; char* tau; char* tall; char* Wpart;
  size_t total;
};
static QrWork qr_layout(char* base, int esz, int64_t m, int64_t n) {
  const int64_t k = m < n ? m : n;
  const int64_t np = (k + QB - 1) / QB;
  const int64_t wide = n > k ? n : k;
  QrWork w;
  size_t off = 0;
  w.Af = base + off;  off += align256((size_t)m * n * esz);
  w.Vp = base + off;  off += align256((size_t)m * QB * esz);
  w.Tm = base + off;  off += align256((size_t)np * QB * QB * esz);
  w.W = base + off;   off += align256((size_t)QB * wide * esz);
  w.W2 = base + off;  off += align256((size_t)QB * wide * esz);
  w.tau = base + off; off += align256((size_t)(k + QB) * esz);
  w.tall = base + off; off += align256(QR_TALL_SCRATCH);
  w.Wpart = base + off; off += align256((size_t)(QR_SPLITS + 1) * QB * wide * esz);
  w.total = off;
  return w;
}

template <typename T>
static int qr_run(int dt, int64_t m, int64_t n, const T* A, T* Q, T* R, char* work) {
  const int64_t k = m < n ? m : n;
  const QrWork w = qr_layout(work, sizeof(T), m, n);
  T* Af = (T*)w.Af;
  T* Vp = (T*)w.Vp;
  T* W = (T*)w.W;
  T* W2 = (T*)w.W2;
  TNH_HIP(hipMemcpyAsync(Af, A, (size_t)m * n * sizeof(T), hipMemcpyDeviceToDevice, stream()));
  // ---- factor (geqrf)
  for (int64_t j0 = 0; j0 < k; j0 += QB) {
    const int jb = (int)((k - j0 < QB) ? (k - j0) : QB);
    const int64_t mj = m - j0;
    T* Tm = (T*)w.Tm + (j0 / QB) * QB * QB;
    if (mj > 512 && g_qr_tall) {
      const int rc = qr_panel_tall<T>(Af, n, m, j0, jb, (T*)w.tau, Tm, Vp, (double*)w.tall);
      if (rc) return rc;
    } else {
      hipLaunchKernelGGL((qr_panel_kernel<T>), dim3(1), dim3(1024), 0, stream(), Af, n, m, j0, jb, (T*)w.tau, Tm, Vp);
      TNH_LAUNCH_CHECK();
    }
    const int64_t nt = n - j0 - jb;
    if (nt > 0) {
      T* C = Af + j0 * n + j0 + jb;
      int rc = qr_vt_times_c(dt, Vp, C, n, mj, nt, W, w.Wpart);                                   // W = V^T C
      if (rc) return rc;
      rc = tnh_gemm_ex(dt, dt, 1, 0, QB, nt, QB, Tm, QB, W, nt, W2, nt, 1, 0, 0, 0, 1.0, 0.0);    // W2 = T^T W
      if (rc) return rc;
      rc = tnh_gemm_ex(dt, dt, 0, 0, mj, nt, QB, Vp, QB, W2, nt, C, n, 1, 0, 0, 0, -1.0, 1.0);    // C -= V W2
      if (rc) return rc;
    }
  }
  // ---- R and Q (orgqr)
  {
    const int64_t ne = k * n;
    hipLaunchKernelGGL((qr_triu_kernel<T>), dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, stream(), R, Af, n, k, n);
    const int64_t nq = m * k;
    hipLaunchKernelGGL((qr_eye_kernel<T>), dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream(), Q, m, k);
    TNH_LAUNCH_CHECK();
  }
  const int64_t last = ((k - 1) / QB) * QB;
  for (int64_t j0 = last; j0 >= 0; j0 -= QB) {
    const int jb = (int)((k - j0 < QB) ? (k - j0) : QB);
    const int64_t mj = m - j0;
    const int64_t nc = k - j0;  // columns j0 .. k-1 of Q are touched by this and later panels
    T* Tm = (T*)w.Tm + (j0 / QB) * QB * QB;
    const int64_t ne = mj * QB;
    hipLaunchKernelGGL((qr_extract_v_kernel<T>), dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, stream(), Af, n, m,
                       j0, jb, Vp);
    TNH_LAUNCH_CHECK();
    T* C = Q + j0 * k + j0;
    int rc = qr_vt_times_c(dt, Vp, C, k, mj, nc, W, w.Wpart);                                   // W = V^T C
    if (rc) return rc;
    rc = tnh_gemm_ex(dt, dt, 0, 0, QB, nc, QB, Tm, QB, W, nc, W2, nc, 1, 0, 0, 0, 1.0, 0.0);     // W2 = T W
    if (rc) return rc;
    rc = tnh_gemm_ex(dt, dt, 0, 0, mj, nc, QB, Vp, QB, W2, nc, C, k, 1, 0, 0, 0, -1.0, 1.0);     // C -= V W2
    if (rc) return rc;
  }
  return TNH_OK;
}

}  // namespace tnh

using namespace tnh;

extern "C" {

int tnh_qr_work_bytes(int dtype, int64_t m, int64_t n, size_t* nbytes) {
  TNH_REQUIRE(nbytes != nullptr, "null nbytes");
  TNH_REQUIRE(dtype == TNH_F32 || dtype == TNH_F64, "tnh_qr supports f32 / f64 (got dtype %d)", dtype);
  TNH_REQUIRE(m >= 0 && n >= 0, "negative extent");
  size_t need = qr_layout(nullptr, dtype_size(dtype), m, n).total + 256;
  if (qr_panel16_supported(dtype, m, n)) {
    const size_t fast = qr_panel16_work_bytes(dtype, m, n);
    if (fast > need) need = fast;
  }
  *nbytes = need;
  return TNH_OK;
}

int tnh_qr(int dtype, int64_t m, int64_t n, const void* A, void* Q, void* R, void* work) {
  TNH_NEED_INIT();
  {
    const char* env = getenv("TNH_QR_TALL");
    g_qr_tall = !(env && env[0] == '0');
  }
  TNH_REQUIRE(dtype == TNH_F32 || dtype == TNH_F64, "tnh_qr supports f32 / f64 (got dtype %d)", dtype);
  TNH_REQUIRE(m >= 0 && n >= 0, "negative extent");
  if (m == 0 || n == 0) return TNH_OK;
  TNH_REQUIRE(A && Q && R && work, "null pointer");
  if (qr_panel16_supported(dtype, m, n) && !capturing()) {
    // 16-wide Cholesky-QR panels (one small kernel per panel instead of a launch per column); a numerically
    // rank-deficient panel is reported and the matrix goes through the column-by-column path below.  The report is a
    // status word read back by the host (one stream synchronisation per call: tnh.h says so), which is illegal while
    // the stream is being captured into a hipGraph -- a captured QR takes the column path
    int st = 0;
    const int rc = qr_panel16(dtype, m, n, A, Q, R, work, &st);
    if (rc != TNH_OK) return rc;
    if (st == 0) return TNH_OK;
  }
  char* base = (char*)(((uintptr_t)work + 255) & ~(uintptr_t)255);
  if (dtype == TNH_F32) return qr_run<float>(dtype, m, n, (const float*)A, (float*)Q, (float*)R, base);
  return qr_run<double>(dtype, m, n, (const double*)A, (double*)Q, (double*)R, base);
}

}  // extern "C"
