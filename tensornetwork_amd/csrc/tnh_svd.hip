// K7: thin SVD by one-sided (Hestenes) Jacobi, f32 and f64.
//
// The row-major m x n input is orthogonalised ROW-wise on the smaller side:
//   p = min(m, n) rows of length q = max(m, n)   (X = A if m <= n, else A^T)
//   R X = diag(s) W,  R (p x p) = product of the plane rotations, W row-orthonormal
//   => X = R^T diag(s) W.
// Rotations within one round-robin round touch disjoint row pairs, so a round is
// one launch of p/2 workgroups; a sweep is p-1 rounds.  Each workgroup streams
// its two rows twice (Gram entries with f64 accumulation, then the rotation) and
// the matching two rows of R.  All min(m,n) singular values come out (the
// reference API returns the discarded ones too, decompositions.py:65); vectors
// are emitted for the leading k only.
//
// This phase is not HBM-roofline work: a sweep moves ~(p-1) * 2 * (p*q + p*p)
// elements through L2/Infinity-Cache; see DESIGN.md for the measured rates.
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <numeric>
#include <vector>
#include "tnh_types.h"

namespace tnh {

// Work buffer layout (byte offsets, 256-B aligned pieces).
struct SvdLayout {
  int64_t p, q;
  bool transposed;  // X = A^T
  bool block;       // f32 block-Jacobi speed path (tnh_svd_block.hip): X, R zero padded
  int64_t ldx, ldr; // row pitch of X (>= q) and R (>= p); rows allocated: ldr
  size_t off_X, off_R, off_norm, off_perm, off_flag, off_scratch, total;
};

// tnh_svd_block.hip
template <typename T>
int svd_block_pad_copy(T* Xp, int64_t P, int64_t Q, const T* A, int64_t m, int64_t n, bool trans);
size_t svd_block_scratch_bytes(int esz, int64_t P, int64_t Q);
int svd_block_schedule_pairs(int nb, int groups, int32_t* out, int* rounds_out);
template <typename T>
int svd_block_sweeps(T* X, T* R, int64_t P, int64_t Q, char* scratch, int* flag, double tol,
                     int max_sweeps, int* sweeps_out, bool* converged_out);

static bool block_path_enabled() {
  const char* e = getenv("TNH_SVD_BLOCK");
  return !(e && e[0] == '0');
}

static SvdLayout svd_layout(int dtype, int64_t m, int64_t n) {
  SvdLayout L;
  L.transposed = (m > n);
  L.p = L.transposed ? n : m;
  L.q = L.transposed ? m : n;
  const size_t esz = (size_t)dtype_size(dtype);
  auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
  L.block = ((dtype == TNH_F32 || dtype == TNH_F64) && L.p > 64 && block_path_enabled());
  L.ldx = L.block ? ((L.q + 127) / 128) * 128 : L.q;
  L.ldr = L.block ? ((L.p + 127) / 128) * 128 : L.p;
  size_t off = 0;
  L.off_X = off; off += al((size_t)L.ldr * L.ldx * esz);
  L.off_R = off; off += al((size_t)L.ldr * L.ldr * esz);
  L.off_norm = off; off += al((size_t)L.p * sizeof(double));
  L.off_perm = off; off += al((size_t)L.p * sizeof(int32_t));
  L.off_flag = off; off += 256;
  L.off_scratch = off; off += L.block ? al(svd_block_scratch_bytes((int)esz, L.ldr, L.ldx)) : 0;
  L.total = off;
  return L;
}

// Circle-method round robin over P (even) players: pair `i` of round `r`.
__device__ __forceinline__ void rr_pair(int64_t P, int64_t r, int64_t i, int64_t& a, int64_t& b) {
  const int64_t Q = P - 1;
  if (i == 0) {
    a = Q;
    b = r % Q;
  } else {
    a = (r + i) % Q;
    b = (r - i + Q) % Q;
  }
  if (a > b) {
    const int64_t t = a;
    a = b;
    b = t;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void jacobi_round_kernel(T* __restrict__ X, T* __restrict__ R,
                                                           int64_t p, int64_t P, int64_t q,
                                                           int64_t round, double tol,
                                                           int* __restrict__ flag) {
  int64_t a, b;
  rr_pair(P, round, blockIdx.x, a, b);
  if (b >= p) return;  // bye (odd p)
  T* xa = X + a * q;
  T* xb = X + b * q;
  const int tid = threadIdx.x;
  double aa = 0.0, bb = 0.0, ab = 0.0;
  for (int64_t j = tid; j < q; j += 256) {
    const double va = (double)xa[j], vb = (double)xb[j];
    aa += va * va;
    bb += vb * vb;
    ab += va * vb;
  }
  __shared__ double red[4][3];
  aa = wave_sum(aa);
  bb = wave_sum(bb);
  ab = wave_sum(ab);
  if ((tid & 63) == 0) {
    red[tid >> 6][0] = aa;
    red[tid >> 6][1] = bb;
    red[tid >> 6][2] = ab;
  }
  __syncthreads();
  aa = red[0][0] + red[1][0] + red[2][0] + red[3][0];
  bb = red[0][1] + red[1][1] + red[2][1] + red[3][1];
  ab = red[0][2] + red[1][2] + red[2][2] + red[3][2];
  if (!(fabs(ab) > tol * sqrt(aa * bb))) return;  // already orthogonal (or a zero row)
  if (tid == 0) *flag = 1;
  const double zeta = (bb - aa) / (2.0 * ab);
  const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double cd = 1.0 / sqrt(1.0 + t * t);
  const T c = (T)cd, s = (T)(cd * t);
  for (int64_t j = tid; j < q; j += 256) {
    const T va = xa[j], vb = xb[j];
    xa[j] = c * va - s * vb;
    xb[j] = s * va + c * vb;
  }
  T* ra = R + a * p;
  T* rb = R + b * p;
  for (int64_t j = tid; j < p; j += 256) {
    const T va = ra[j], vb = rb[j];
    ra[j] = c * va - s * vb;
    rb[j] = s * va + c * vb;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void row_norm_kernel(const T* __restrict__ X, int64_t p, int64_t q,
                                                       int64_t ld, double* __restrict__ norms) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p) return;
  const T* x = X + row * ld;
  double acc = 0.0;
  for (int64_t j = lane; j < q; j += 64) {
    const double v = (double)x[j];
    acc += v * v;
  }
  acc = wave_sum(acc);
  if (lane == 0) norms[row] = sqrt(acc);
}

template <typename T>
__global__ __launch_bounds__(256) void emit_s_kernel(T* __restrict__ S, const double* __restrict__ norms,
                                                     const int32_t* __restrict__ perm, int64_t p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < p) S[i] = (T)norms[perm[i]];
}

// out (k x len)  [TRANS=false]: out[i][j] = src[perm[i]][j] * scale_i
// out (len x k)  [TRANS=true ]: out[j][i] = src[perm[i]][j] * scale_i
// scale_i = 1/norm[perm[i]] when `normalize`, else 1; zero rows stay zero.
template <typename T, bool TRANS>
__global__ __launch_bounds__(256) void emit_rows_kernel(T* __restrict__ out, const T* __restrict__ src,
                                                        const int32_t* __restrict__ perm,
                                                        const double* __restrict__ norms, int64_t k,
                                                        int64_t len, int64_t src_ld, int normalize) {
  const int64_t total = k * len;
  const int64_t step = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += step) {
    int64_t i, j;
    if (TRANS) { j = e / k; i = e - j * k; }
    else { i = e / len; j = e - i * len; }
    const int32_t r = perm[i];
    double v = (double)src[(int64_t)r * src_ld + j];
    if (normalize) {
      const double nr = norms[r];
      v = (nr > 0.0) ? v / nr : 0.0;
    }
    out[e] = (T)v;
  }
}

template <typename T>
static int svd_finish(const SvdLayout& L, int64_t m, int64_t n, void* S, char* work, bool converged,
                      int max_sweeps);

// Speed path (f32 / f64): zero-padded copy, block-Jacobi sweeps (tnh_svd_block.hip).
template <typename T>
static int svd_factor_block(const SvdLayout& L, int dtype, int64_t m, int64_t n, const void* A, void* S,
                            char* work, int* sweeps_out, bool accumulate = true) {
  T* X = (T*)(work + L.off_X);
  T* R = accumulate ? (T*)(work + L.off_R) : nullptr;
  int* flag = (int*)(work + L.off_flag);
  const int64_t P = L.ldr, Q = L.ldx;
  int rc = svd_block_pad_copy<T>(X, P, Q, (const T*)A, m, n, L.transposed);
  if (rc) return rc;
  if (accumulate) {
    rc = tnh_eye(R, P, P, dtype);
    if (rc) return rc;
  }
  const double eps = (sizeof(T) == 4) ? 5.9604644775390625e-08 : 1.1102230246251565e-16;
  const double tol = eps * sqrt((double)L.q);
  // graded inputs (singular values over many decades: every DMRG split) need far more sweeps than Gaussian ones
  // (f64 2^-i, n = 256: 36); 40 was not enough for f64 DMRG at D = 256
  const int max_sweeps = 120;
  int sweeps = 0;
  bool converged = false;
  rc = svd_block_sweeps<T>(X, R, P, Q, work + L.off_scratch, flag, tol, max_sweeps, &sweeps, &converged);
  if (rc) return rc;
  if (sweeps_out) *sweeps_out = sweeps;
  return svd_finish<T>(L, m, n, S, work, converged, max_sweeps);
}

template <typename T>
static int svd_factor_t(const SvdLayout& L, int dtype, int64_t m, int64_t n, const void* A, void* S,
                        char* work, int* sweeps_out) {
  const int64_t p = L.p, q = L.q;
  T* X = (T*)(work + L.off_X);
  T* R = (T*)(work + L.off_R);
  int* flag = (int*)(work + L.off_flag);
  int rc;
  if (L.block) return svd_factor_block<T>(L, dtype, m, n, A, S, work, sweeps_out);
  if (L.transposed) {
    const int64_t shape[2] = {m, n};
    const int32_t pm[2] = {1, 0};
    rc = tnh_permute(X, A, 2, shape, pm, (int)sizeof(T));
  } else {
    rc = tnh_d2d(X, A, (size_t)(p * q) * sizeof(T));
  }
  if (rc) return rc;
  rc = tnh_eye(R, p, p, dtype);
  if (rc) return rc;

  const double eps = (sizeof(T) == 4) ? 5.9604644775390625e-08 : 1.1102230246251565e-16;
  const double tol = eps * sqrt((double)q);
  const int64_t P = (p + 1) & ~int64_t(1);
  const int max_sweeps = 40;
  int sweeps = 0;
  bool converged = (p < 2);
  while (!converged && sweeps < max_sweeps) {
    TNH_HIP(hipMemsetAsync(flag, 0, sizeof(int), stream()));
    for (int64_t r = 0; r < P - 1; ++r) {
      hipLaunchKernelGGL((jacobi_round_kernel<T>), dim3((unsigned)(P / 2)), dim3(256), 0, stream(), X, R, p,
                         P, q, r, tol, flag);
    }
    TNH_LAUNCH_CHECK();
    int h = 0;
    TNH_HIP(hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, stream()));
    TNH_HIP(hipStreamSynchronize(stream()));
    ++sweeps;
    converged = (h == 0);
  }
  if (sweeps_out) *sweeps_out = sweeps;

  return svd_finish<T>(L, m, n, S, work, converged, max_sweeps);
}

// Row norms -> singular values, descending order, S.
template <typename T>
static int svd_finish(const SvdLayout& L, int64_t m, int64_t n, void* S, char* work, bool converged,
                      int max_sweeps) {
  const int64_t p = L.p, q = L.q;
  T* X = (T*)(work + L.off_X);
  double* norms = (double*)(work + L.off_norm);
  int32_t* perm = (int32_t*)(work + L.off_perm);
  hipLaunchKernelGGL((row_norm_kernel<T>), dim3((unsigned)((p + 3) / 4)), dim3(256), 0, stream(), X, p, q,
                     L.ldx, norms);
  TNH_LAUNCH_CHECK();
  std::vector<double> hn((size_t)p);
  TNH_HIP(hipMemcpyAsync(hn.data(), norms, (size_t)p * sizeof(double), hipMemcpyDeviceToHost, stream()));
  TNH_HIP(hipStreamSynchronize(stream()));
  std::vector<int32_t> hp((size_t)p);
  std::iota(hp.begin(), hp.end(), 0);
  std::stable_sort(hp.begin(), hp.end(), [&](int32_t x, int32_t y) { return hn[x] > hn[y]; });
  TNH_HIP(hipMemcpyAsync(perm, hp.data(), (size_t)p * sizeof(int32_t), hipMemcpyHostToDevice, stream()));
  TNH_HIP(hipStreamSynchronize(stream()));
  if (p > 0) {
    hipLaunchKernelGGL((emit_s_kernel<T>), dim3((unsigned)((p + 255) / 256)), dim3(256), 0, stream(), (T*)S,
                       norms, perm, p);
    TNH_LAUNCH_CHECK();
  }
  if (!converged) {
    set_error("Jacobi SVD did not converge in %d sweeps (%lld x %lld)", max_sweeps, (long long)m,
              (long long)n);
    return TNH_ERR_NO_CONVERGE;
  }
  return TNH_OK;
}

// Replace zero rows of the row-orthonormal factor W (k x q, or its transpose when `trans`,
// i.e. stored q x k) by unit vectors orthogonal to all other rows, so that the emitted factor
// is orthonormal even for rank-deficient input (LAPACK's behaviour, which the reference relies
// on for u / vh of e.g. a zero tensor).  Rare path, done ON THE DEVICE with the K9 kernels:
// M (q x k) = [non-zero vectors | Gaussian columns]; the thin Householder QR of M leaves, in its
// last columns, orthonormal vectors orthogonal to every non-zero vector; they are copied into the
// zero slots.  (The non-zero vectors themselves are not touched.)
template <typename T>
static int complete_basis(T* dW, int64_t k, int64_t q, bool trans, const std::vector<int>& zero_rows) {
  if (zero_rows.empty()) return TNH_OK;
  const int dt = sizeof(T) == 4 ? TNH_F32 : TNH_F64;
  if ((int64_t)zero_rows.size() == k) {
    // everything is zero: W = leading rows of the identity
    if (trans) return tnh_eye(dW, q, k, dt);
    return tnh_eye(dW, k, q, dt);
  }
  std::vector<char> is_zero((size_t)k, 0);
  for (int z : zero_rows) is_zero[(size_t)z] = 1;
  const int64_t nz = (int64_t)zero_rows.size(), knz = k - nz;
  size_t qr_bytes = 0;
  int rc = tnh_qr_work_bytes(dt, q, k, &qr_bytes);
  if (rc) return rc;
  void *Mt = nullptr, *M = nullptr, *Q = nullptr, *Rr = nullptr, *work = nullptr;
  auto cleanup = [&]() {
    for (void* ptr : {Mt, M, Q, Rr, work})
      if (ptr) tnh_free(ptr);
  };
#define TNH_CB(call)          \
  do {                        \
    rc = (call);              \
    if (rc) {                 \
      cleanup();              \
      return rc;              \
    }                         \
  } while (0)
  TNH_CB(tnh_malloc(&Mt, (size_t)k * q * sizeof(T)));
  TNH_CB(tnh_malloc(&M, (size_t)k * q * sizeof(T)));
  TNH_CB(tnh_malloc(&Q, (size_t)k * q * sizeof(T)));
  TNH_CB(tnh_malloc(&Rr, (size_t)k * k * sizeof(T)));
  TNH_CB(tnh_malloc(&work, qr_bytes));
  // Mt (k x q, rows are vectors): the non-zero vectors first, then Gaussian rows
  int64_t row = 0;
  for (int64_t i = 0; i < k; ++i) {
    if (is_zero[(size_t)i]) continue;
    T* dst = (T*)Mt + row * q;
    if (!trans) {
      TNH_CB(tnh_d2d(dst, dW + i * q, (size_t)q * sizeof(T)));
    } else {
      const int64_t shape[1] = {q}, stride[1] = {k};
      TNH_CB(tnh_strided_copy(dst, dW, 1, shape, stride, i, (int)sizeof(T)));
    }
    ++row;
  }
  TNH_CB(tnh_random((T*)Mt + knz * q, nz * q, dt, 0x5eedull + (uint64_t)k * 131 + (uint64_t)q, 1, 0.0, 1.0));
  {
    const int64_t shape[2] = {k, q};
    const int32_t pm[2] = {1, 0};
    TNH_CB(tnh_permute(M, Mt, 2, shape, pm, (int)sizeof(T)));
  }
  TNH_CB(tnh_qr(dt, q, k, M, Q, Rr, work));
  // column knz + j of Q -> zero slot zero_rows[j]
  for (int64_t j = 0; j < nz; ++j) {
    const int z = zero_rows[(size_t)j];
    const int64_t shape[1] = {q}, stride[1] = {k};
    if (!trans) {
      TNH_CB(tnh_strided_copy(dW + (int64_t)z * q, Q, 1, shape, stride, knz + j, (int)sizeof(T)));
    } else {
      // W stored q x k: column z <- column knz + j of Q (both with stride k): via a contiguous row of Mt
      TNH_CB(tnh_strided_copy(Mt, Q, 1, shape, stride, knz + j, (int)sizeof(T)));
      TNH_CB(tnh_strided_scatter(dW, Mt, 1, shape, stride, z, (int)sizeof(T)));
    }
  }
  TNH_HIP(hipStreamSynchronize(stream()));   // scratch is recycled in stream order, but be explicit
  cleanup();
#undef TNH_CB
  return TNH_OK;
}

template <typename T>
static int svd_vectors_t(const SvdLayout& L, int64_t m, int64_t n, char* work, int64_t k, void* U, void* Vh) {
  if (k == 0) return TNH_OK;
  const int64_t p = L.p, q = L.q;
  const T* X = (const T*)(work + L.off_X);
  const T* R = (const T*)(work + L.off_R);
  const double* norms = (const double*)(work + L.off_norm);
  const int32_t* perm = (const int32_t*)(work + L.off_perm);
  auto grid = [&](int64_t total) {
    int64_t b = (total + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 16;
    return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
  };
  // which of the leading k rows are exactly zero?
  std::vector<double> hn((size_t)p);
  std::vector<int32_t> hp((size_t)p);
  TNH_HIP(hipMemcpyAsync(hn.data(), norms, (size_t)p * sizeof(double), hipMemcpyDeviceToHost, stream()));
  TNH_HIP(hipMemcpyAsync(hp.data(), perm, (size_t)p * sizeof(int32_t), hipMemcpyDeviceToHost, stream()));
  TNH_HIP(hipStreamSynchronize(stream()));
  std::vector<int> zero_rows;
  for (int64_t i = 0; i < k; ++i)
    if (!(hn[(size_t)hp[(size_t)i]] > 0.0)) zero_rows.push_back((int)i);

  if (!L.transposed) {
    // A = R^T diag(s) W :  U[j][i] = R[perm i][j] (m x k),  Vh[i][:] = X[perm i][:] / s_i (k x n)
    hipLaunchKernelGGL((emit_rows_kernel<T, true>), dim3(grid(k * p)), dim3(256), 0, stream(), (T*)U, R, perm,
                       norms, k, p, L.ldr, 0);
    hipLaunchKernelGGL((emit_rows_kernel<T, false>), dim3(grid(k * q)), dim3(256), 0, stream(), (T*)Vh, X, perm,
                       norms, k, q, L.ldx, 1);
    TNH_LAUNCH_CHECK();
    return complete_basis<T>((T*)Vh, k, q, false, zero_rows);
  }
  // A^T = R^T diag(s) W  =>  A = W^T diag(s) R :
  //   U[j][i] = X[perm i][j] / s_i (m x k),  Vh[i][:] = R[perm i][:] (k x n)
  hipLaunchKernelGGL((emit_rows_kernel<T, true>), dim3(grid(k * q)), dim3(256), 0, stream(), (T*)U, X, perm,
                     norms, k, q, L.ldx, 1);
  hipLaunchKernelGGL((emit_rows_kernel<T, false>), dim3(grid(k * p)), dim3(256), 0, stream(), (T*)Vh, R, perm,
                     norms, k, p, L.ldr, 0);
  TNH_LAUNCH_CHECK();
  (void)m; (void)n;
  return complete_basis<T>((T*)U, k, q, true, zero_rows);
}

// ---------------------------------------------------------------------------
// complex64 / complex128: the same one-sided Jacobi with unitary plane rotations (no block path
// yet).  For rows x_a, x_b with g = <x_a, x_b> = sum conj(x_a) x_b = |g| e^{i phi}:  x_b is first
// turned by e^{-i phi} (the Gram entry becomes real), then the real rotation (c, s) of the real
// algorithm is applied; the accumulated factor R receives the same unitary, so X_0 = R^H X.
// ---------------------------------------------------------------------------
template <typename Z, typename Rl>
__global__ __launch_bounds__(256) void jacobi_round_cplx_kernel(Z* __restrict__ X, Z* __restrict__ R, int64_t p,
                                                                int64_t P, int64_t q, int64_t round, double tol,
                                                                int* __restrict__ flag) {
  int64_t a, b;
  rr_pair(P, round, blockIdx.x, a, b);
  if (b >= p) return;  // bye (odd p)
  Z* xa = X + a * q;
  Z* xb = X + b * q;
  const int tid = threadIdx.x;
  double aa = 0.0, bb = 0.0, gr = 0.0, gi = 0.0;
  for (int64_t j = tid; j < q; j += 256) {
    const double ar = (double)xa[j].re, ai = (double)xa[j].im, br = (double)xb[j].re, bi = (double)xb[j].im;
    aa += ar * ar + ai * ai;
    bb += br * br + bi * bi;
    gr += ar * br + ai * bi;   // conj(a) * b
    gi += ar * bi - ai * br;
  }
  __shared__ double red[4][4];
  aa = wave_sum(aa); bb = wave_sum(bb); gr = wave_sum(gr); gi = wave_sum(gi);
  if ((tid & 63) == 0) {
    red[tid >> 6][0] = aa; red[tid >> 6][1] = bb; red[tid >> 6][2] = gr; red[tid >> 6][3] = gi;
  }
  __syncthreads();
  aa = red[0][0] + red[1][0] + red[2][0] + red[3][0];
  bb = red[0][1] + red[1][1] + red[2][1] + red[3][1];
  gr = red[0][2] + red[1][2] + red[2][2] + red[3][2];
  gi = red[0][3] + red[1][3] + red[2][3] + red[3][3];
  const double ab = sqrt(gr * gr + gi * gi);
  if (!(ab > tol * sqrt(aa * bb))) return;  // already orthogonal (or a zero row)
  if (tid == 0) *flag = 1;
  const double pr = gr / ab, pi = -gi / ab;           // e^{-i phi}
  const double zeta = (bb - aa) / (2.0 * ab);
  const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
  auto rotate = [&](Z* ra, Z* rb, int64_t len) {
    for (int64_t j = tid; j < len; j += 256) {
      const double ar = (double)ra[j].re, ai = (double)ra[j].im;
      const double br0 = (double)rb[j].re, bi0 = (double)rb[j].im;
      const double br = br0 * pr - bi0 * pi, bi = br0 * pi + bi0 * pr;   // e^{-i phi} b
      ra[j].re = (Rl)(c * ar - sn * br);
      ra[j].im = (Rl)(c * ai - sn * bi);
      rb[j].re = (Rl)(sn * ar + c * br);
      rb[j].im = (Rl)(sn * ai + c * bi);
    }
  };
  rotate(xa, xb, q);
  rotate(R + a * p, R + b * p, p);
}

template <typename Z>
__global__ __launch_bounds__(256) void row_norm_cplx_kernel(const Z* __restrict__ X, int64_t p, int64_t q,
                                                            double* __restrict__ norms) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p) return;
  const Z* x = X + row * q;
  double acc = 0.0;
  for (int64_t j = lane; j < q; j += 64) {
    const double re = (double)x[j].re, im = (double)x[j].im;
    acc += re * re + im * im;
  }
  acc = wave_sum(acc);
  if (lane == 0) norms[row] = sqrt(acc);
}

// complex variant of emit_rows_kernel with an optional conjugation
template <typename Z, typename Rl, bool TRANS>
__global__ __launch_bounds__(256) void emit_rows_cplx_kernel(Z* __restrict__ out, const Z* __restrict__ src,
                                                             const int32_t* __restrict__ perm,
                                                             const double* __restrict__ norms, int64_t k,
                                                             int64_t len, int normalize, int conj) {
  const int64_t total = k * len;
  const int64_t step = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += step) {
    int64_t i, j;
    if (TRANS) { j = e / k; i = e - j * k; }
    else { i = e / len; j = e - i * len; }
    const int32_t r = perm[i];
    double re = (double)src[(int64_t)r * len + j].re, im = (double)src[(int64_t)r * len + j].im;
    if (conj) im = -im;
    if (normalize) {
      const double nr = norms[r];
      re = (nr > 0.0) ? re / nr : 0.0;
      im = (nr > 0.0) ? im / nr : 0.0;
    }
    out[e].re = (Rl)re;
    out[e].im = (Rl)im;
  }
}

template <typename Z, typename Rl>
static int svd_factor_cplx(const SvdLayout& L, int dtype, int64_t m, int64_t n, const void* A, void* S, char* work,
                           int* sweeps_out) {
  const int64_t p = L.p, q = L.q;
  Z* X = (Z*)(work + L.off_X);
  Z* R = (Z*)(work + L.off_R);
  int* flag = (int*)(work + L.off_flag);
  int rc;
  if (L.transposed) {
    const int64_t shape[2] = {m, n};
    const int32_t pm[2] = {1, 0};
    rc = tnh_permute(X, A, 2, shape, pm, (int)sizeof(Z));   // plain transpose (no conjugation)
  } else {
    rc = tnh_d2d(X, A, (size_t)(p * q) * sizeof(Z));
  }
  if (rc) return rc;
  rc = tnh_eye(R, p, p, dtype);
  if (rc) return rc;
  const double eps = (sizeof(Rl) == 4) ? 5.9604644775390625e-08 : 1.1102230246251565e-16;
  const double tol = eps * sqrt((double)q);
  const int64_t P = (p + 1) & ~int64_t(1);
  const int max_sweeps = 40;
  int sweeps = 0;
  bool converged = (p < 2);
  while (!converged && sweeps < max_sweeps) {
    TNH_HIP(hipMemsetAsync(flag, 0, sizeof(int), stream()));
    for (int64_t r = 0; r < P - 1; ++r)
      hipLaunchKernelGGL((jacobi_round_cplx_kernel<Z, Rl>), dim3((unsigned)(P / 2)), dim3(256), 0, stream(), X, R, p,
                         P, q, r, tol, flag);
    TNH_LAUNCH_CHECK();
    int h = 0;
    TNH_HIP(hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, stream()));
    TNH_HIP(hipStreamSynchronize(stream()));
    ++sweeps;
    converged = (h == 0);
  }
  if (sweeps_out) *sweeps_out = sweeps;
  // row norms -> singular values (REAL dtype of the same precision), descending
  double* norms = (double*)(work + L.off_norm);
  int32_t* perm = (int32_t*)(work + L.off_perm);
  hipLaunchKernelGGL((row_norm_cplx_kernel<Z>), dim3((unsigned)((p + 3) / 4)), dim3(256), 0, stream(), X, p, q, norms);
  TNH_LAUNCH_CHECK();
  std::vector<double> hn((size_t)p);
  TNH_HIP(hipMemcpyAsync(hn.data(), norms, (size_t)p * sizeof(double), hipMemcpyDeviceToHost, stream()));
  TNH_HIP(hipStreamSynchronize(stream()));
  std::vector<int32_t> hp((size_t)p);
  std::iota(hp.begin(), hp.end(), 0);
  std::stable_sort(hp.begin(), hp.end(), [&](int32_t x, int32_t y) { return hn[x] > hn[y]; });
  TNH_HIP(hipMemcpyAsync(perm, hp.data(), (size_t)p * sizeof(int32_t), hipMemcpyHostToDevice, stream()));
  TNH_HIP(hipStreamSynchronize(stream()));
  hipLaunchKernelGGL((emit_s_kernel<Rl>), dim3((unsigned)((p + 255) / 256)), dim3(256), 0, stream(), (Rl*)S, norms,
                     perm, p);
  TNH_LAUNCH_CHECK();
  if (!converged) {
    set_error("Jacobi SVD did not converge in %d sweeps (%lld x %lld)", max_sweeps, (long long)m, (long long)n);
    return TNH_ERR_NO_CONVERGE;
  }
  return TNH_OK;
}

template <typename Z, typename Rl>
static int svd_vectors_cplx(const SvdLayout& L, int dtype, int64_t m, int64_t n, char* work, int64_t k, void* U,
                            void* Vh) {
  if (k == 0) return TNH_OK;
  const int64_t p = L.p, q = L.q;
  const Z* X = (const Z*)(work + L.off_X);
  const Z* R = (const Z*)(work + L.off_R);
  const double* norms = (const double*)(work + L.off_norm);
  const int32_t* perm = (const int32_t*)(work + L.off_perm);
  auto grid = [&](int64_t total) {
    int64_t b = (total + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 16;
    return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
  };
  std::vector<double> hn((size_t)p);
  std::vector<int32_t> hp((size_t)p);
  TNH_HIP(hipMemcpyAsync(hn.data(), norms, (size_t)p * sizeof(double), hipMemcpyDeviceToHost, stream()));
  TNH_HIP(hipMemcpyAsync(hp.data(), perm, (size_t)p * sizeof(int32_t), hipMemcpyDeviceToHost, stream()));
  TNH_HIP(hipStreamSynchronize(stream()));
  int64_t zeros = 0;
  for (int64_t i = 0; i < k; ++i)
    if (!(hn[(size_t)hp[(size_t)i]] > 0.0)) ++zeros;
  if (zeros == k) {   // zero matrix: LAPACK's answer, leading rows / columns of the identity
    int rc = tnh_eye(U, m, k, dtype);
    if (rc) return rc;
    return tnh_eye(Vh, k, n, dtype);
  }
  if (zeros > 0) {
    set_error("complex SVD of an exactly rank-deficient matrix: basis completion is not implemented yet");
    return TNH_ERR_UNSUPPORTED;
  }
  if (!L.transposed) {
    // X_0 = A = R^H diag(s) W :  U[j][i] = conj(R[perm i][j]) (m x k),  Vh[i][:] = X[perm i][:] / s_i
    hipLaunchKernelGGL((emit_rows_cplx_kernel<Z, Rl, true>), dim3(grid(k * p)), dim3(256), 0, stream(), (Z*)U, R, perm,
                       norms, k, p, 0, 1);
    hipLaunchKernelGGL((emit_rows_cplx_kernel<Z, Rl, false>), dim3(grid(k * q)), dim3(256), 0, stream(), (Z*)Vh, X, perm,
                       norms, k, q, 1, 0);
  } else {
    // X_0 = A^T = R^H diag(s) W  =>  A = W^T diag(s) conj(R):
    //   U[j][i] = X[perm i][j] / s_i (m x k),  Vh[i][:] = conj(R[perm i][:]) (k x n)
    hipLaunchKernelGGL((emit_rows_cplx_kernel<Z, Rl, true>), dim3(grid(k * q)), dim3(256), 0, stream(), (Z*)U, X, perm,
                       norms, k, q, 1, 0);
    hipLaunchKernelGGL((emit_rows_cplx_kernel<Z, Rl, false>), dim3(grid(k * p)), dim3(256), 0, stream(), (Z*)Vh, R, perm,
                       norms, k, p, 0, 1);
  }
  TNH_LAUNCH_CHECK();
  return TNH_OK;
}

// Top-k mode (f32 / f64 block path): the sweeps ran WITHOUT accumulating the rotations (half the
// update work); the orthogonalised side gives its vectors directly (rows of X / s), the other side
// comes from A by one GEMM and a column / row scaling:
//   m <= n:  Vh_k = X_k / s_k,   U_k  = A Vh_k^T diag(1 / s_k)
//   m >  n:  U_k  = (X_k / s_k)^T,   Vh_k = diag(1 / s_k) U_k^T A
// Accurate to eps * s_1 / s_k, i.e. for the LEADING singular triplets only -- the caller checks
// s_k / s_1 before choosing this mode (HipBackend.svd).
template <typename T>
static int svd_vectors_topk_t(const SvdLayout& L, int dtype, int64_t m, int64_t n, const void* A, char* work,
                              const void* S, int64_t k, void* U, void* Vh) {
  if (k == 0) return TNH_OK;
  const int64_t q = L.q;
  const T* X = (const T*)(work + L.off_X);
  const double* norms = (const double*)(work + L.off_norm);
  const int32_t* perm = (const int32_t*)(work + L.off_perm);
  auto grid = [&](int64_t total) {
    int64_t b = (total + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 16;
    return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
  };
  int rc;
  if (!L.transposed) {
    hipLaunchKernelGGL((emit_rows_kernel<T, false>), dim3(grid(k * q)), dim3(256), 0, stream(), (T*)Vh, X, perm,
                       norms, k, q, L.ldx, 1);                                      // Vh (k x n)
    TNH_LAUNCH_CHECK();
    rc = tnh_gemm_ex(dtype, dtype, 0, 1, m, k, n, A, n, Vh, n, U, k, 1, 0, 0, 0, 1.0, 0.0);   // U = A Vh^T
    if (rc) return rc;
    const int64_t shape[2] = {m, k}, sa[2] = {k, 1}, sb[2] = {0, 1};
    return tnh_binary(TNH_OP_DIV, U, U, S, 2, shape, sa, sb, dtype);                 // columns / s
  }
  hipLaunchKernelGGL((emit_rows_kernel<T, true>), dim3(grid(k * q)), dim3(256), 0, stream(), (T*)U, X, perm, norms,
                     k, q, L.ldx, 1);                                               // U (m x k)
  TNH_LAUNCH_CHECK();
  rc = tnh_gemm_ex(dtype, dtype, 1, 0, k, n, m, U, k, A, n, Vh, n, 1, 0, 0, 0, 1.0, 0.0);     // Vh = U^T A
  if (rc) return rc;
  const int64_t shape[2] = {k, n}, sa[2] = {n, 1}, sb[2] = {1, 0};
  return tnh_binary(TNH_OP_DIV, Vh, Vh, S, 2, shape, sa, sb, dtype);                 // rows / s
}

}  // namespace tnh

using namespace tnh;

extern "C" {

int tnh_svd_work_bytes(int dtype, int64_t m, int64_t n, size_t* nbytes) {
  TNH_REQUIRE(nbytes != nullptr, "null pointer");
  TNH_REQUIRE(dtype == TNH_F32 || dtype == TNH_F64 || dtype == TNH_C64 || dtype == TNH_C128,
              "SVD supports f32 / f64 / complex64 / complex128 (got dtype %d)", dtype);
  TNH_REQUIRE(m >= 0 && n >= 0, "negative size");
  *nbytes = svd_layout(dtype, m, n).total;
  return TNH_OK;
}

int tnh_svd_factor(int dtype, int64_t m, int64_t n, const void* A, void* S, void* work, int* sweeps_out) {
  TNH_NEED_INIT();
  TNH_REQUIRE(dtype == TNH_F32 || dtype == TNH_F64 || dtype == TNH_C64 || dtype == TNH_C128,
              "SVD supports f32 / f64 / complex64 / complex128 (got dtype %d)", dtype);
  TNH_REQUIRE(m >= 0 && n >= 0, "negative size");
  if (sweeps_out) *sweeps_out = 0;
  if (m == 0 || n == 0) return TNH_OK;
  TNH_REQUIRE(A && S && work, "null pointer");
  TNH_REQUIRE(std::min(m, n) < (int64_t(1) << 31), "matrix too large");
  const SvdLayout L = svd_layout(dtype, m, n);
  if (dtype == TNH_C64) return svd_factor_cplx<cf32, float>(L, dtype, m, n, A, S, (char*)work, sweeps_out);
  if (dtype == TNH_C128) return svd_factor_cplx<cf64, double>(L, dtype, m, n, A, S, (char*)work, sweeps_out);
  if (dtype == TNH_F32) return svd_factor_t<float>(L, dtype, m, n, A, S, (char*)work, sweeps_out);
  return svd_factor_t<double>(L, dtype, m, n, A, S, (char*)work, sweeps_out);
}

int tnh_svd_vectors(int dtype, int64_t m, int64_t n, void* work, int64_t k, void* U, void* Vh) {
  TNH_NEED_INIT();
  TNH_REQUIRE(dtype == TNH_F32 || dtype == TNH_F64 || dtype == TNH_C64 || dtype == TNH_C128,
              "SVD supports f32 / f64 / complex64 / complex128 (got dtype %d)", dtype);
  TNH_REQUIRE(k >= 0 && k <= std::min(m, n), "k out of range");
  if (k == 0) return TNH_OK;
  TNH_REQUIRE(U && Vh && work, "null pointer");
  const SvdLayout L = svd_layout(dtype, m, n);
  if (dtype == TNH_C64) return svd_vectors_cplx<cf32, float>(L, dtype, m, n, (char*)work, k, U, Vh);
  if (dtype == TNH_C128) return svd_vectors_cplx<cf64, double>(L, dtype, m, n, (char*)work, k, U, Vh);
  if (dtype == TNH_F32) return svd_vectors_t<float>(L, m, n, (char*)work, k, U, Vh);
  return svd_vectors_t<double>(L, m, n, (char*)work, k, U, Vh);
}

int tnh_svd_block_schedule(int nb, int groups, int32_t* pairs_out, int* rounds_out) {
  TNH_REQUIRE(pairs_out && rounds_out, "tnh_svd_block_schedule: null output");
  TNH_REQUIRE(nb >= 2 && nb % 2 == 0, "tnh_svd_block_schedule: nb must be even (got %d)", nb);
  TNH_REQUIRE(groups == 1 || ((groups == 2 || groups == 4) && nb % (2 * groups) == 0 && nb >= 4 * groups),
              "tnh_svd_block_schedule: %d groups need nb %% %d == 0", groups, 2 * groups);
  return tnh::svd_block_schedule_pairs(nb, groups, pairs_out, rounds_out);
}

int tnh_svd_factor_topk(int dtype, int64_t m, int64_t n, const void* A, void* S, void* work, int* sweeps_out,
                        int* mode_out) {
  TNH_NEED_INIT();
  TNH_REQUIRE(mode_out != nullptr, "null pointer");
  *mode_out = 0;
  const bool real = (dtype == TNH_F32 || dtype == TNH_F64);
  if (!real || m <= 0 || n <= 0) return tnh_svd_factor(dtype, m, n, A, S, work, sweeps_out);
  TNH_REQUIRE(A && S && work, "null pointer");
  const SvdLayout L = svd_layout(dtype, m, n);
  if (!L.block) return tnh_svd_factor(dtype, m, n, A, S, work, sweeps_out);   // small path: always accumulates
  if (sweeps_out) *sweeps_out = 0;
  *mode_out = 1;
  if (dtype == TNH_F32) return svd_factor_block<float>(L, dtype, m, n, A, S, (char*)work, sweeps_out, false);
  return svd_factor_block<double>(L, dtype, m, n, A, S, (char*)work, sweeps_out, false);
}

int tnh_svd_vectors_topk(int dtype, int64_t m, int64_t n, const void* A, void* work, const void* S, int64_t k,
                         void* U, void* Vh) {
  TNH_NEED_INIT();
  TNH_REQUIRE(dtype == TNH_F32 || dtype == TNH_F64, "top-k SVD vectors support f32 / f64 (got dtype %d)", dtype);
  TNH_REQUIRE(k >= 0 && k <= std::min(m, n), "k out of range");
  if (k == 0) return TNH_OK;
  TNH_REQUIRE(A && U && Vh && work && S, "null pointer");
  const SvdLayout L = svd_layout(dtype, m, n);
  TNH_REQUIRE(L.block, "top-k vectors need the block path (min(m, n) > 64)");
  if (dtype == TNH_F32) return svd_vectors_topk_t<float>(L, dtype, m, n, A, (char*)work, S, k, U, Vh);
  return svd_vectors_topk_t<double>(L, dtype, m, n, A, (char*)work, S, k, U, Vh);
}

int tnh_svd(int dtype, int64_t m, int64_t n, const void* A, void* U, void* S, void* Vh, int64_t k,
            void* work, int* sweeps_out) {
  int rc = tnh_svd_factor(dtype, m, n, A, S, work, sweeps_out);
  if (rc) return rc;
  return tnh_svd_vectors(dtype, m, n, work, k, U, Vh);
}

}  // extern "C"
