"""NumPy model of the round-3 SVD ("band + spectrum slicing"), the algorithm behind tnh_svd_band.hip.

Host-only design check (no GPU, no reference code): every step below is what one kernel of
tensornetwork_amd/csrc/tnh_svd_band.hip does, written with the same formulas so that the GPU tests can
compare stage by stage.

  stage 1   A (m x n, f32, m >= n, n % B == 0)  ->  Q_L^T A Q_R = band (upper triangular, B super-diagonals)
            by alternating column panels (QR) and row panels (LQ) of width B.  A panel is factored by
            Cholesky-QR (Gram in f64) followed by the Householder reconstruction of Ballard et al. (LU of
            Q1 - S), so that ONE small kernel per panel replaces B dependent column steps; the block
            reflector is I - V T V^T with T^-1 = striu(V^T V) + diag(V^T V) / 2 (exactly orthogonal for any V).
  stage 2   T = Bd^T Bd (symmetric, half bandwidth B, f64); all n singular values by spectrum slicing:
            Sturm counts nu(sigma) = #negative pivots of LDL^T(T - sigma^2 I) (no pivoting), a multi-section
            over n evenly spaced shifts followed by bisection per value, in sigma space.
  stage 3   k right vectors by inverse iteration with the same LDL^T (shift just off the eigenvalue),
            left vectors u = Bd v / s, both in f64 on the band; then the back-transformation through the
            stage-1 reflectors.
"""
import numpy as np


# ----------------------------------------------------------------------------- stage 1
def panel_factor(G, Ptop):
  """What the one-workgroup `sb_factor_kernel` computes for a panel P (rows x B), from G = P^T P (f64) and the
  top B x B block of P.  Returns X (V_below = P_below X), Vtop (unit lower), T (upper), Rh (new top block)
  and ok (Cholesky went through)."""
  B = G.shape[0]
  G = G.astype(np.float64)
  R = np.zeros((B, B))
  ok = True
  gmax = max(np.max(np.diag(G)), 1e-300)
  for j in range(B):           # Cholesky, upper: G = R^T R
    d = G[j, j] - R[:j, j] @ R[:j, j]
    if not d > 1e-12 * gmax:
      ok = False
      d = 1e-12 * gmax
    R[j, j] = np.sqrt(d)
    R[j, j + 1:] = (G[j, j + 1:] - R[:j, j] @ R[:j, j + 1:]) / R[j, j]
  Rinv = np.linalg.inv(R)       # kernel: back substitution on the 16 x 16 triangle
  Q1top = Ptop.astype(np.float64) @ Rinv
  # modified LU of Q1 - [S; 0]: S_jj = -sign(pivot candidate) so that |pivot| >= 1
  W = Q1top.copy()
  S = np.zeros(B)
  L = np.eye(B)
  U = np.zeros((B, B))
  for j in range(B):
    S[j] = -1.0 if W[j, j] >= 0 else 1.0
    W[j, j] -= S[j]
    U[j, j:] = W[j, j:]
    L[j + 1:, j] = W[j + 1:, j] / W[j, j]
    W[j + 1:, j + 1:] -= np.outer(L[j + 1:, j], W[j, j + 1:])
  Uinv = np.linalg.inv(U)
  X = Rinv @ Uinv               # rows below the top block: V = (P R^-1) U^-1
  Vtop = L
  # Gram of V from Q1^T Q1 = I:  V^T V = U^-T (2 I - Q1top^T S - S Q1top) U^-1
  N = Uinv.T @ (2 * np.eye(B) - Q1top.T * S[None, :] - S[:, None] * Q1top) @ Uinv
  Tinv = np.triu(N, 1) + np.diag(np.diag(N)) / 2
  T = np.linalg.inv(Tinv)
  Rh = S[:, None] * R           # H^T P = [S R; 0]
  return X, Vtop, T, Rh, ok


def to_band(A, B=16, dtype=np.float32):
  """Stage 1.  Returns (Af, Tl, Tr, ok): Af holds the band in its upper band and the reflector vectors below /
  right of it (LAPACK gebrd style), Tl / Tr the T factors of the column / row panels."""
  A = np.array(A, dtype=dtype)
  m, n = A.shape
  assert m >= n and n % B == 0
  Tl, Tr = [], []
  ok_all = True
  for j in range(0, n, B):
    # ---- column panel: rows j.., columns j .. j+B
    P = A[j:, j:j + B]
    G = P.astype(np.float64).T @ P.astype(np.float64)
    X, Vtop, T, Rh, ok = panel_factor(G, P[:B])
    ok_all &= ok
    V = np.empty((m - j, B), dtype=dtype)
    V[:B] = Vtop.astype(dtype)
    V[B:] = (P[B:].astype(np.float64) @ X).astype(dtype)
    Tl.append(T.astype(dtype))
    if j + B < n:
      C = A[j:, j + B:]
      Wt = T.astype(dtype).T @ (V.T @ C)            # (B x nc)
      C -= V @ Wt
    A[j:j + B, j:j + B] = np.triu(Rh).astype(dtype)
    A[j:j + B, j:j + B] += np.tril(V[:B], -1)       # V's strict lower triangle in the top block
    A[j + B:, j:j + B] = V[B:]
    # ---- row panel: rows j .. j+B, columns j+B..
    if j + B < n:
      Pr = A[j:j + B, j + B:].T                      # (nc x B): the LQ of the row panel is the QR of its transpose
      G = Pr.astype(np.float64).T @ Pr.astype(np.float64)
      X, Vtop, T, Rh, ok = panel_factor(G, Pr[:B])
      ok_all &= ok
      nc = n - j - B
      V = np.empty((nc, B), dtype=dtype)
      V[:B] = Vtop.astype(dtype)
      V[B:] = (Pr[B:].astype(np.float64) @ X).astype(dtype)
      Tr.append(T.astype(dtype))
      if j + B < m:
        C = A[j + B:, j + B:]
        Y = (C @ V) @ T.astype(dtype)                # (rows x B)
        C -= Y @ V.T
      blk = np.triu(Rh).T + np.triu(V[:B].T, 1)      # L (lower) + V^T's strict upper part
      A[j:j + B, j + B:j + 2 * B] = blk.astype(dtype)
      A[j:j + B, j + 2 * B:] = V[B:].T
  return A, Tl, Tr, ok_all


def band_of(Af, B=16):
  """Bd[i, d] = band entry (i, i + d), d = 0..B, f64 (rows of the square n x n part)."""
  n = Af.shape[1]
  Bd = np.zeros((n, B + 1))
  for d in range(B + 1):
    Bd[:n - d, d] = np.diagonal(Af[:n, :n], d)
  # inside the band the strictly lower part of diagonal blocks / strictly upper part of the super-diagonal
  # blocks hold reflector entries, not band entries
  for i in range(n):
    for d in range(B + 1):
      c = i + d
      if c >= n:
        continue
      bi, bc = i // B, c // B
      if bc == bi + 1 and (c % B) > (i % B):
        Bd[i, d] = 0.0
  return Bd


def band_dense(Bd):
  n, b1 = Bd.shape
  M = np.zeros((n, n))
  for d in range(b1):
    M[np.arange(n - d), np.arange(n - d) + d] = Bd[:n - d, d]
  return M


def gram_band(Bd):
  """T = Bd^T Bd in symmetric band storage Tb[i, d] = T[i, i + d], d = 0..B (f64)."""
  n, b1 = Bd.shape
  Bw = b1 - 1
  Tb = np.zeros((n, b1))
  for i in range(n):
    for d in range(b1):
      c = i + d
      if c >= n:
        continue
      # sum_r Bd[r][i] * Bd[r][c], r from max(0, c - Bw) .. i
      s = 0.0
      for r in range(max(0, c - Bw), i + 1):
        s += Bd[r, i - r] * Bd[r, c - r]
      Tb[i, d] = s
  return Tb


# ----------------------------------------------------------------------------- stage 2
def sturm_counts(Tb, shifts2):
  """nu(sigma^2) for many shifts at once: negative pivots of the un-pivoted LDL^T of T - sigma^2 I.
  Window form (what a lane group holds in registers): W[q] (Bw+1 x Bw+1, symmetric) = active rows."""
  n, b1 = Tb.shape
  Bw = b1 - 1
  ns = len(shifts2)
  W = np.zeros((ns, b1, b1))
  # window rows/cols = matrix indices j .. j+Bw
  def row_entries(i):       # full symmetric row i restricted to columns i .. i+Bw
    return Tb[i]
  # initial window: indices 0..Bw
  for a in range(b1):
    for c in range(a, b1):
      if a < n and c < n:
        v = Tb[a, c - a]
        W[:, a, c] = v
        W[:, c, a] = v
  for a in range(b1):
    W[:, a, a] -= shifts2
  count = np.zeros(ns, dtype=np.int64)
  tiny = 1e-300
  for j in range(n):
    d = W[:, 0, 0].copy()
    d = np.where(np.abs(d) < tiny, -tiny, d)
    count += (d < 0)
    l = W[:, 1:, 0] / d[:, None]
    W[:, 1:, 1:] -= l[:, :, None] * W[:, None, 1:, 0]
    # slide
    W[:, :-1, :-1] = W[:, 1:, 1:]
    W[:, -1, :] = 0
    W[:, :, -1] = 0
    i = j + b1                # new index entering the window at position Bw
    if i < n:
      for a in range(1, b1):  # T[i - a... ] entries: T[i][i - (Bw - p)]
        pass
      for p in range(b1):     # window position p <-> matrix index j + 1 + p
        r = j + 1 + p
        dd = i - r
        if 0 <= dd <= Bw:
          v = Tb[r, dd]
          W[:, p, Bw] = v
          W[:, Bw, p] = v
      W[:, Bw, Bw] -= shifts2
    else:
      W[:, Bw, Bw] = 1.0      # padding rows: positive pivots, no coupling
  return count


def singular_values(Tb, smax, extra_steps=10):
  """All n singular values: multi-section over n evenly spaced sigma, then bisection."""
  n = Tb.shape[0]
  grid = smax * (np.arange(1, n + 1) / n)             # sigma_1 .. sigma_n (upper ends)
  below = sturm_counts(Tb, grid ** 2)                 # number of values < grid[i]
  # value number q (ascending, 0-based) lies in (grid[i-1], grid[i]] with i = first index where below > q
  q = np.arange(n)
  idx = np.searchsorted(below, q, side="right")
  idx = np.minimum(idx, n - 1)
  lo = np.where(idx > 0, grid[np.maximum(idx - 1, 0)], 0.0)
  hi = grid[idx]
  for _ in range(extra_steps):
    mid = 0.5 * (lo + hi)
    c = sturm_counts(Tb, mid ** 2)
    right = c <= q          # fewer than q+1 values below mid: value q is above mid
    lo = np.where(right, mid, lo)
    hi = np.where(right, hi, mid)
  return (0.5 * (lo + hi))[::-1]                     # descending


# ----------------------------------------------------------------------------- stage 3
def ldl_band(Tb, shift2):
  """Un-pivoted LDL^T of T - shift2 I; returns (D (n), L (n x Bw): L[i, p] = L[i + 1 + p, i])."""
  n, b1 = Tb.shape
  Bw = b1 - 1
  M = np.zeros((n + Bw, b1))          # working band rows i: M[i, d] = entry (i, i + d) (upper = lower by symmetry)
  M[:n] = Tb
  M[:n, 0] -= shift2
  D = np.zeros(n)
  L = np.zeros((n, Bw))
  for j in range(n):
    d = M[j, 0]
    if abs(d) < 1e-300:
      d = -1e-300
    D[j] = d
    col = M[j, 1:].copy()             # entries (j, j+1 .. j+Bw) = (j+1.., j)
    l = col / d
    L[j] = l
    for p in range(Bw):               # row j+1+p, columns j+1+p .. j+Bw
      if j + 1 + p < n:
        M[j + 1 + p, 0:Bw - p] -= l[p] * col[p:]
  return D, L


def ldl_solve(D, L, rhs):
  n, Bw = L.shape
  x = rhs.copy()
  for j in range(n):                  # L y = b
    hi = min(Bw, n - 1 - j)
    x[j + 1:j + 1 + hi] -= L[j, :hi] * x[j]
  x /= D
  for j in range(n - 1, -1, -1):      # L^T z = y
    hi = min(Bw, n - 1 - j)
    x[j] -= L[j, :hi] @ x[j + 1:j + 1 + hi]
  return x


def top_vectors(Tb, Bd, s, k, iters=3, seed=0):
  """k leading right vectors of the band by inverse iteration, left vectors u = Bd v / s."""
  n = Tb.shape[0]
  rng = np.random.default_rng(seed)
  V = np.zeros((n, k))
  smax = s[0]
  for i in range(k):
    lam = s[i] ** 2
    shift = lam + (1e-10 * smax ** 2) * (1 if i % 2 else -1)     # just off the eigenvalue
    D, L = ldl_band(Tb, shift)
    x = rng.standard_normal(n)
    for _ in range(iters):
      x = ldl_solve(D, L, x)
      x /= np.linalg.norm(x)
    V[:, i] = x
  Bm = band_dense(Bd)
  U = (Bm @ V) / s[None, :k]
  return U, V


def back_transform(Af, Tl, Tr, Ub, Vb, B=16):
  """U = Q_L [Ub; 0], V = Q_R Vb through the stage-1 block reflectors (reverse order)."""
  m, n = Af.shape
  k = Ub.shape[1]
  U = np.zeros((m, k))
  U[:n] = Ub
  for p in range(len(Tl) - 1, -1, -1):
    j = p * B
    V = np.tril(Af[j:, j:j + B].astype(np.float64), -1)
    V[:B] += np.eye(B)
    V[:B] = np.tril(V[:B])
    T = Tl[p].astype(np.float64)
    U[j:] -= V @ (T @ (V.T @ U[j:]))
  Vv = Vb.copy()
  for p in range(len(Tr) - 1, -1, -1):
    j = p * B
    Vt = np.triu(Af[j:j + B, j + B:].astype(np.float64), 1)      # B x nc, unit diagonal at (i, i)
    Vt[:, :B] = np.triu(Vt[:, :B], 1) + np.eye(B)
    Vr = Vt.T
    T = Tr[p].astype(np.float64)
    Vv[j + B:] -= Vr @ (T @ (Vr.T @ Vv[j + B:]))
  return U, Vv


def svd_band(A, k, B=16):
  A = np.asarray(A)
  Af, Tl, Tr, ok = to_band(A, B)
  Bd = band_of(Af, B)
  Tb = gram_band(Bd)
  smax = np.sqrt(np.max(np.sum(np.abs(band_dense(Tb) + band_dense(Tb).T - np.diag(Tb[:, 0])), axis=1)))  # Gershgorin
  s = singular_values(Tb, smax * (1 + 1e-12), extra_steps=12 + 10)
  Ub, Vb = top_vectors(Tb, Bd, s, k)
  U, V = back_transform(Af, Tl, Tr, Ub, Vb, B)
  return U, s, V.T, ok


if __name__ == "__main__":
  import sys
  import time
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
  kind = sys.argv[2] if len(sys.argv) > 2 else "gauss"
  rng = np.random.default_rng(1)
  if kind == "gauss":
    A = rng.standard_normal((n, n)).astype(np.float32)
  else:
    qu, _ = np.linalg.qr(rng.standard_normal((n, n)))
    qv, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = ((qu * 2.0 ** (-np.arange(n) / 32.0)) @ qv.T).astype(np.float32)
  k = n // 16
  t0 = time.time()
  U, s, Vh, ok = svd_band(A, k)
  print("model seconds", time.time() - t0, "ok", ok)
  sr = np.linalg.svd(A.astype(np.float64), compute_uv=False)
  print("max |s - s_ref| / s0 =", np.max(np.abs(s - sr)) / sr[0])
  print("orth U", np.max(np.abs(U.T @ U - np.eye(k))), "orth V", np.max(np.abs(Vh @ Vh.T - np.eye(k))))
  Ak = (U * s[:k]) @ Vh
  ur, srr, vr = np.linalg.svd(A.astype(np.float64))
  best = (ur[:, :k] * srr[:k]) @ vr[:k]
  print("|A_k - best_k|_F / s0 =", np.linalg.norm(Ak - best) / sr[0])
  print("residual |A v - s u| / s0 =", np.max(np.linalg.norm(A.astype(np.float64) @ Vh.T - U * s[:k], axis=0)) / sr[0])
