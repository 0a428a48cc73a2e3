"""Krylov solvers on backend tensors: ``eigsh_lanczos``, ``eigsh``, ``eigs``, ``gmres``.

The reference's NumPy backend implements Lanczos by hand
(``backends/numpy/numpy_backend.py:415-534``) and wraps SciPy/ARPACK for ``eigsh``
(168-214) and ``gmres`` (300-412).  Here every vector stays a backend tensor (in HBM
for ``HipBackend``): the operator ``A`` is the caller's contraction, inner products,
norms and axpys are backend ops, and only the Krylov coefficients -- a handful of
scalars per iteration -- come back to the host.  The small projected matrices
(``num_krylov_vecs`` x ``num_krylov_vecs``) are diagonalised with ``backend.eigh``.

All functions take the backend object first; ``HipBackend`` binds them as methods, and
the CPU test-suite runs the same code on the oracle backend.
"""
import numpy as np


def _flat(be, x):
  return be.reshape(x, (-1,))


def _vdot(be, a, b):
  """<a, b> with a conjugated, as a host scalar (one 1 x 1 GEMM + read-back)."""
  return be.item(be.tensordot(be.conj(_flat(be, a)), _flat(be, b), 1))


def _norm(be, x):
  return float(np.real(be.item(be.norm(x))))


def _axpy(be, y, alpha, x):
  """y + alpha * x."""
  return be.addition(y, be.multiply(x, alpha))


def _tridiag(diag, off):
  k = len(diag)
  t = np.zeros((k, k), dtype=np.result_type(np.float64, *[np.asarray(d).dtype for d in diag]))
  for i, d in enumerate(diag):
    t[i, i] = d
  for i, o in enumerate(off[:k - 1]):
    t[i, i + 1] = o
    t[i + 1, i] = np.conj(o)
  return t


def _small_eigh(be, matrix):
  """Eigen-decomposition of a small host matrix of Krylov coefficients THROUGH the backend
  (device Jacobi kernels for HipBackend); returns host arrays.  Real symmetric, or complex
  Hermitian when ``matrix`` is complex."""
  if np.iscomplexobj(matrix):
    w, u = be.eigh(be.convert_to_tensor(np.ascontiguousarray(matrix, dtype=np.complex128)))
    return np.asarray(w).real.astype(np.float64), np.asarray(u, dtype=np.complex128)
  w, u = be.eigh(be.convert_to_tensor(np.ascontiguousarray(np.real(matrix), dtype=np.float64)))
  return np.asarray(w, dtype=np.float64), np.asarray(u, dtype=np.float64)


def _is_complex(x):
  if hasattr(x, "is_complex"):
    return bool(x.is_complex)
  try:
    return np.dtype(x.dtype).kind == "c"
  except TypeError:
    return False


def _coef(c, cplx):
  return complex(c) if cplx else float(np.real(c))


def _combine(be, vectors, coefs, cplx):
  """sum_i coefs[i] * vectors[i] (coefficients with |c| == 0 skipped)."""
  out = None
  for v, c in zip(vectors, coefs):
    c = _coef(c, cplx)
    if c == 0 and out is not None:
      continue
    out = be.multiply(v, c) if out is None else _axpy(be, out, c, v)
  return out


def eigsh_lanczos(be, A, args=None, initial_state=None, shape=None, dtype=None, num_krylov_vecs=20,
                  numeig=1, tol=1e-8, delta=1e-8, ndiag=20, reorthogonalize=False):
  """Lowest eigenpairs of a Hermitian operator by the Lanczos iteration.

  Same algorithm, arguments, stopping rules and error behaviour as
  ``numpy_backend.py:415-534``: iterate until ``num_krylov_vecs`` vectors, until a Krylov
  vector's norm drops below ``delta``, or until the ``numeig`` lowest Ritz values move by
  less than ``tol`` between two diagonalisations (every ``ndiag`` iterations)."""
  if args is None:
    args = []
  if num_krylov_vecs < numeig:
    raise ValueError('`num_krylov_vecs` >= `numeig` required!')
  if numeig > 1 and not reorthogonalize:
    raise ValueError(
        "Got numeig = {} > 1 and `reorthogonalize = False`. "
        "Use `reorthogonalize=True` for `numeig > 1`".format(numeig))
  if initial_state is None:
    if (shape is None) or (dtype is None):
      raise ValueError("if no `initial_state` is passed, then `shape` and"
                       "`dtype` have to be provided")
    initial_state = be.randn(shape, dtype)
  if not be.is_tensor(initial_state):
    raise TypeError("Expected a backend tensor. Got {}".format(type(initial_state)))

  vector_n = be.divide(initial_state, _norm(be, initial_state))
  norms, diags, krylov = [], [], []
  first, eigvalsold = True, None
  for it in range(num_krylov_vecs):
    nrm = _norm(be, vector_n)
    if abs(nrm) < delta:
      break
    norms.append(nrm)
    vector_n = be.divide(vector_n, nrm)
    if reorthogonalize:
      for v in krylov:
        vector_n = _axpy(be, vector_n, -_vdot(be, v, vector_n), v)
    krylov.append(vector_n)
    a_vec = A(vector_n, *args)
    diags.append(_vdot(be, vector_n, a_vec))
    if it > 0 and it % ndiag == 0 and len(diags) >= numeig:
      eigvals, _ = _small_eigh(be, _tridiag(diags, norms[1:]))
      if not first and np.linalg.norm(eigvals[:numeig] - eigvalsold[:numeig]) < tol:
        break
      first = False
      eigvalsold = eigvals[:numeig]
    a_vec = _axpy(be, a_vec, -diags[-1], krylov[-1])
    if it > 0:
      a_vec = _axpy(be, a_vec, -norms[-1], krylov[-2])
    vector_n = a_vec

  eigvals, u = _small_eigh(be, _tridiag(diags, norms[1:]))
  eigenvectors = []
  for n2 in range(min(numeig, len(eigvals))):
    # the Ritz coefficients are complex for a complex Hermitian operator (the reference multiplies by
    # u[n1, n2] directly, numpy_backend.py:527-531); float() would drop the imaginary part
    cplx = bool(np.iscomplexobj(u))
    state = be.multiply(krylov[0], _coef(u[0, n2], cplx))
    for n1 in range(1, len(krylov)):
      state = _axpy(be, state, _coef(u[n1, n2], cplx), krylov[n1])
    eigenvectors.append(be.divide(state, _norm(be, state)))
  return eigvals[:numeig], eigenvectors


def eigsh_lanczos_deferred(be, A, args=None, initial_state=None, shape=None, dtype=None, num_krylov_vecs=20,
                           numeig=1, tol=1e-8, delta=1e-8, ndiag=20, reorthogonalize=False):
  """`eigsh_lanczos` with the Krylov coefficients kept ON THE DEVICE between convergence checks.

  The reference's loop reads two scalars back per iteration (the norm of the new vector and the diagonal
  element): on a GPU that is two pipeline drains per matvec.  Here the recurrence divides / scales by the
  0-d device tensors themselves, so nothing is read back until a convergence check (every `ndiag`
  iterations) or the end, where all pending coefficients come back in one burst.  The `delta` test
  (norm of a Krylov vector below `delta` = invariant subspace) is applied at those points too: vectors
  built past such a breakdown are garbage, but they are cut off before anything uses them, so the
  returned eigenpairs are the ones `eigsh_lanczos` returns.  Same arguments, errors and results.
  Opt-in (see `FiniteDMRG.deferred_lanczos`) until it has been measured on the GPU."""
  if args is None:
    args = []
  if num_krylov_vecs < numeig:
    raise ValueError('`num_krylov_vecs` >= `numeig` required!')
  if numeig > 1 and not reorthogonalize:
    raise ValueError(
        "Got numeig = {} > 1 and `reorthogonalize = False`. "
        "Use `reorthogonalize=True` for `numeig > 1`".format(numeig))
  if initial_state is None:
    if (shape is None) or (dtype is None):
      raise ValueError("if no `initial_state` is passed, then `shape` and"
                       "`dtype` have to be provided")
    initial_state = be.randn(shape, dtype)
  if not be.is_tensor(initial_state):
    raise TypeError("Expected a backend tensor. Got {}".format(type(initial_state)))

  dev_dot = lambda a, b: be.tensordot(be.conj(_flat(be, a)), _flat(be, b), 1)      # 0-d device tensor
  vector_n = be.divide(initial_state, be.norm(initial_state))
  norms_dev, diags_dev, krylov = [], [], []       # device scalars of iterations not yet read back
  norms, diags = [], []                           # host copies (same meaning as in eigsh_lanczos)
  first, eigvalsold = True, None

  def flush():
    """Read the pending coefficients back; returns the number of usable Krylov vectors."""
    for n_t, d_t in zip(norms_dev, diags_dev):
      norms.append(float(np.real(be.item(n_t))))
      diags.append(be.item(d_t))
    del norms_dev[:], diags_dev[:]
    for j, nrm in enumerate(norms):
      if abs(nrm) < delta:
        return j
    return len(norms)

  usable = None
  for it in range(num_krylov_vecs):
    nrm_t = be.norm(vector_n)
    vector_n = be.divide(vector_n, nrm_t)
    if reorthogonalize:
      for v in krylov:
        vector_n = be.subtraction(vector_n, be.multiply(v, dev_dot(v, vector_n)))
    krylov.append(vector_n)
    a_vec = A(vector_n, *args)
    diag_t = dev_dot(vector_n, a_vec)
    norms_dev.append(nrm_t)
    diags_dev.append(diag_t)
    if it > 0 and it % ndiag == 0 and it + 1 >= numeig:
      usable = flush()
      if usable < len(norms):
        break
      eigvals, _ = _small_eigh(be, _tridiag(diags, norms[1:]))
      if not first and np.linalg.norm(eigvals[:numeig] - eigvalsold[:numeig]) < tol:
        break
      first = False
      eigvalsold = eigvals[:numeig]
    a_vec = be.subtraction(a_vec, be.multiply(krylov[-1], diag_t))
    if it > 0:
      a_vec = be.subtraction(a_vec, be.multiply(krylov[-2], nrm_t))
    vector_n = a_vec

  usable = flush()
  norms, diags, krylov = norms[:usable], diags[:usable], krylov[:usable]
  eigvals, u = _small_eigh(be, _tridiag(diags, norms[1:]))
  eigenvectors = []
  for n2 in range(min(numeig, len(eigvals))):
    # the Ritz coefficients are complex for a complex Hermitian operator (the reference multiplies by
    # u[n1, n2] directly, numpy_backend.py:527-531); float() would drop the imaginary part
    cplx = bool(np.iscomplexobj(u))
    state = be.multiply(krylov[0], _coef(u[0, n2], cplx))
    for n1 in range(1, len(krylov)):
      state = _axpy(be, state, _coef(u[n1, n2], cplx), krylov[n1])
    eigenvectors.append(be.divide(state, _norm(be, state)))
  return eigvals[:numeig], eigenvectors


def eigsh(be, A, args=None, initial_state=None, shape=None, dtype=None, num_krylov_vecs=50, numeig=6,
          tol=1e-8, which='LA', maxiter=None):
  """``numeig`` extremal eigenpairs of a Hermitian operator (interface of
  ``numpy_backend.py:168-214``, which wraps ``scipy.sparse.linalg.eigsh``).

  Thick-restart Lanczos with full re-orthogonalisation: a cycle grows the orthonormal
  basis to ``num_krylov_vecs`` vectors, Rayleigh-Ritz on the projected matrix picks the
  wanted Ritz pairs (``which`` in 'LA', 'SA', 'LM'), converged when every residual
  ``||A y - theta y|| <= tol * max(|theta|, 1)``; otherwise the basis is compressed to the
  wanted Ritz vectors plus the residual direction and the cycle repeats."""
  if args is None:
    args = []
  if which in ('SI', 'LI'):
    raise ValueError(f'which = {which} is currently not supported.')
  if which not in ('LA', 'SA', 'LM'):
    raise ValueError(f"which = {which} is not supported by the hip backend (use 'LA', 'SA' or 'LM').")
  if numeig + 1 >= num_krylov_vecs:
    raise ValueError('`num_krylov_vecs` > `numeig + 1` required!')
  if initial_state is None:
    if (shape is None) or (dtype is None):
      raise ValueError("if no `initial_state` is passed, then `shape` and"
                       "`dtype` have to be provided")
    initial_state = be.randn(shape, dtype)
  if not be.is_tensor(initial_state):
    raise TypeError("Expected a backend tensor. Got {}".format(type(initial_state)))
  size = int(np.prod(be.shape_tuple(initial_state)))
  ncv = min(num_krylov_vecs, size)
  numeig = min(numeig, size)
  if maxiter is None:
    maxiter = 10 * size

  def wanted(theta):
    if which == 'LA':
      return np.argsort(-theta)[:numeig]
    if which == 'SA':
      return np.argsort(theta)[:numeig]
    return np.argsort(-np.abs(theta))[:numeig]

  cplx = _is_complex(initial_state)
  basis = [be.divide(initial_state, _norm(be, initial_state))]
  images = []                               # A applied to each basis vector
  # projected matrix V^H A V (its symmetric / Hermitian part)
  h = np.zeros((ncv, ncv), dtype=np.complex128 if cplx else np.float64)
  theta_sel, ritz = None, None
  for _ in range(max(1, maxiter)):
    # grow the basis to ncv vectors (or to an invariant subspace)
    while True:
      j = len(images)
      w = A(basis[j], *args)
      images.append(w)
      for i in range(j + 1):
        hij = _coef(_vdot(be, basis[i], w), cplx)
        h[i, j] = hij
        h[j, i] = np.conj(hij)
      h[j, j] = h[j, j].real
      if len(basis) == ncv:
        break
      for _pass in range(2):                # classical Gram-Schmidt twice == full re-orthogonalisation
        for v in basis:
          w = _axpy(be, w, -_vdot(be, v, w), v)
      nrm = _norm(be, w)
      if nrm < 1e-14 * max(1.0, abs(h[j, j])):
        break
      basis.append(be.divide(w, nrm))
    m = len(images)
    theta, u = _small_eigh(be, h[:m, :m])
    sel = wanted(theta)
    theta_sel = theta[sel]
    ritz, ritz_img, resid = [], [], []
    for col in sel:
      y = be.multiply(basis[0], _coef(u[0, col], cplx))
      ay = be.multiply(images[0], _coef(u[0, col], cplx))
      for i in range(1, m):
        y = _axpy(be, y, _coef(u[i, col], cplx), basis[i])
        ay = _axpy(be, ay, _coef(u[i, col], cplx), images[i])
      ritz.append(y)
      ritz_img.append(ay)
      resid.append(_norm(be, _axpy(be, ay, -float(theta[col]), y)))
    if all(r <= tol * max(abs(t), 1.0) for r, t in zip(resid, theta_sel)) or len(basis) < ncv or m >= size:
      break
    # thick restart: keep the wanted Ritz vectors, continue from the largest residual direction
    k = len(ritz)
    worst = int(np.argmax(resid))
    nxt = _axpy(be, ritz_img[worst], -float(theta_sel[worst]), ritz[worst])
    for _pass in range(2):
      for v in ritz:
        nxt = _axpy(be, nxt, -_vdot(be, v, nxt), v)
    nrm = _norm(be, nxt)
    h[:, :] = 0.0
    for i in range(k):
      h[i, i] = theta_sel[i]
    basis, images = list(ritz), list(ritz_img)
    if nrm < 1e-300:
      break
    basis.append(be.divide(nxt, nrm))
  order = np.argsort(theta_sel) if which == 'SA' else np.argsort(-theta_sel if which == 'LA' else -np.abs(theta_sel))
  vecs = [be.divide(ritz[i], _norm(be, ritz[i])) for i in order]
  return theta_sel[order], vecs


def eigs(be, A, args=None, initial_state=None, shape=None, dtype=None, num_krylov_vecs=50, numeig=6,
         tol=1e-8, which='LR', maxiter=None):
  """``numeig`` eigenpairs of a general (non-Hermitian) operator: interface, argument checks and
  return convention of ``numpy_backend.py:216-283`` (which wraps ``scipy.sparse.linalg.eigs``, i.e.
  ARPACK's implicitly restarted Arnoldi).

  Here: Arnoldi with a Krylov-Schur (thick) restart.  The orthonormal basis lives on the device; one
  cycle extends it to ``num_krylov_vecs`` vectors with twice-iterated Gram-Schmidt, the projected
  ``ncv x ncv`` coefficient matrix is diagonalised on the host (it is control data -- ARPACK does the
  same on the host in the reference), the residual of Ritz pair i is ``|h[m+1,m]| |y_i[m]|``, and a
  restart compresses the basis onto an orthonormal basis of the wanted Ritz vectors' span
  (conjugate pairs kept together so a real operator keeps a real basis) followed by the Arnoldi
  residual vector.  Eigenvalues and vectors come back complex, as from SciPy; a run that hits
  ``maxiter`` restarts returns its current Ritz pairs."""
  if args is None:
    args = []
  if which in ('SI', 'LI'):
    raise ValueError(f'which = {which} is currently not supported.')
  if which not in ('LM', 'SM', 'LR', 'SR'):
    raise ValueError(f"which = {which} is not supported (use 'LM', 'SM', 'LR' or 'SR').")
  if numeig + 1 >= num_krylov_vecs:
    raise ValueError('`num_krylov_vecs` > `numeig + 1` required!')
  if initial_state is None:
    if (shape is None) or (dtype is None):
      raise ValueError("if no `initial_state` is passed, then `shape` and"
                       "`dtype` have to be provided")
    initial_state = be.randn(shape, dtype)
  if not be.is_tensor(initial_state):
    raise TypeError("Expected a backend tensor. Got {}".format(type(initial_state)))
  size = int(np.prod(be.shape_tuple(initial_state)))
  ncv = min(int(num_krylov_vecs), size)
  numeig = min(int(numeig), size)
  if maxiter is None:
    maxiter = 10 * size
  cplx = _is_complex(initial_state)
  eps = float(be.eps(initial_state.dtype)) if hasattr(be, "eps") else 2.2e-16
  tol = eps if tol is None or tol <= 0 else float(tol)

  def ranked(theta):
    key = {'LM': -np.abs(theta), 'SM': np.abs(theta), 'LR': -theta.real, 'SR': theta.real}[which]
    return np.argsort(key, kind="stable")

  hmat = np.zeros((ncv + 1, ncv), dtype=np.complex128 if cplx else np.float64)
  basis = [be.divide(initial_state, _norm(be, initial_state))]
  kept = 0                                  # leading columns of hmat already filled by the last restart
  theta = ymat = None
  m = 0
  for _ in range(max(1, int(maxiter))):
    j, invariant = kept, False
    while j < ncv:
      w = A(basis[j], *args)
      scale = _norm(be, w)
      for _pass in range(2):
        for i in range(j + 1):
          hij = _coef(_vdot(be, basis[i], w), cplx)
          hmat[i, j] += hij
          w = _axpy(be, w, -hij, basis[i])
      nrm = _norm(be, w)
      hmat[j + 1, j] = nrm
      j += 1
      if nrm <= 100 * eps * max(scale, 1e-300):
        invariant = True
        break
      basis.append(be.divide(w, nrm))
    m = j
    theta, ymat = np.linalg.eig(hmat[:m, :m])
    order = ranked(theta)
    sel = order[:numeig]
    resid = abs(hmat[m, m - 1]) * np.abs(ymat[m - 1, :])
    floor = eps ** (2.0 / 3.0)
    done = resid[sel] <= tol * np.maximum(np.abs(theta[sel]), floor)
    if invariant or m >= size or done.all() or m <= numeig + 1:
      break
    # ---- Krylov-Schur restart --------------------------------------------------------------
    nkeep = min(m - 2, numeig + min(int(done.sum()), (m - numeig) // 2))
    keep = list(order[:nkeep])
    if cplx:
      span = ymat[:, keep]
    else:
      cols, seen = [], set()
      for idx in keep:
        if idx in seen:
          continue
        seen.add(idx)
        if theta[idx].imag == 0:
          cols.append(ymat[:, idx].real)
        else:
          partner = int(np.argmin(np.abs(theta - np.conj(theta[idx])) + 1e300 * (np.arange(m) == idx)))
          seen.add(partner)
          cols.append(ymat[:, idx].real)
          cols.append(ymat[:, idx].imag)
      span = np.stack(cols, axis=1)
      if span.shape[1] > m - 1:
        span = span[:, :m - 1]
    z, _ = np.linalg.qr(span)
    nk = z.shape[1]
    small = z.conj().T @ hmat[:m, :m] @ z
    coupling = hmat[m, m - 1] * z[m - 1, :]
    new_basis = [_combine(be, basis[:m], z[:, i], cplx) for i in range(nk)]
    new_basis.append(basis[m])
    hmat[:, :] = 0
    hmat[:nk, :nk] = small
    hmat[nk, :nk] = coupling
    basis, kept = new_basis, nk

  order = ranked(theta)[:numeig]
  eta = np.asarray(theta[order], dtype=np.complex128)
  vectors = []
  for col in order:
    y = ymat[:, col]
    if cplx:
      vec = _combine(be, basis[:m], y, True)
    else:
      re = _combine(be, basis[:m], y.real, False)
      im = _combine(be, basis[:m], y.imag, False)
      vec = be.addition(re, be.multiply(im, 1j))
    vectors.append(be.divide(vec, _norm(be, vec)))
  return eta, vectors


def gmres(be, A_mv, b, A_args, A_kwargs, x0, tol, atol, num_krylov_vectors, maxiter, M=None):
  """Restarted GMRES(m) (interface of ``numpy_backend.py:300-412`` / ``abstract_backend.py:478-631``;
  the reference wraps ``scipy.sparse.linalg.gmres``).  Arnoldi with modified Gram-Schmidt on backend
  tensors, Givens rotations on the host for the (m+1) x m Hessenberg least-squares problem.
  Stops when ``norm(residual) <= max(tol * norm(b), atol)``.  Returns ``(x, info)`` with info 0 on
  convergence, else the number of restarts performed."""
  if M is not None:
    raise NotImplementedError("M is not supported by the hip backend.")
  bshape = be.shape_tuple(b)
  b_norm = _norm(be, b)
  target = max(tol * b_norm, atol)
  x = be.reshape(x0, bshape)
  m = int(num_krylov_vectors)
  maxiter = 1 if maxiter is None else int(maxiter)
  if b_norm == 0.0:
    return be.multiply(b, 0.0), 0
  for restart in range(maxiter):
    r = be.subtraction(b, A_mv(x, *A_args, **A_kwargs))
    beta = _norm(be, r)
    if beta <= target:
      return x, 0
    basis = [be.divide(r, beta)]
    hess = np.zeros((m + 1, m), dtype=np.complex128)
    cs, sn = np.zeros(m, dtype=np.complex128), np.zeros(m, dtype=np.complex128)
    g = np.zeros(m + 1, dtype=np.complex128)
    g[0] = beta
    k_used, converged = 0, False
    for j in range(m):
      w = A_mv(basis[j], *A_args, **A_kwargs)
      for i in range(j + 1):
        hij = _vdot(be, basis[i], w)
        hess[i, j] = hij
        w = _axpy(be, w, -hij, basis[i])
      hn = _norm(be, w)
      hess[j + 1, j] = hn
      for i in range(j):                      # previous rotations on the new column
        t = cs[i] * hess[i, j] + sn[i] * hess[i + 1, j]
        hess[i + 1, j] = -np.conj(sn[i]) * hess[i, j] + cs[i] * hess[i + 1, j]
        hess[i, j] = t
      denom = np.sqrt(abs(hess[j, j]) ** 2 + hn ** 2)
      if denom == 0.0:
        k_used = j
        break
      cs[j] = abs(hess[j, j]) / denom if hess[j, j] != 0 else 0.0
      phase = hess[j, j] / abs(hess[j, j]) if hess[j, j] != 0 else 1.0
      sn[j] = phase * hn / denom
      hess[j, j] = cs[j] * hess[j, j] + sn[j] * hn
      hess[j + 1, j] = 0.0
      g[j + 1] = -np.conj(sn[j]) * g[j]
      g[j] = cs[j] * g[j]
      k_used = j + 1
      if abs(g[j + 1]) <= target:
        converged = True
        break
      if hn <= 1e-300:
        break
      basis.append(be.divide(w, hn))
    # back substitution on the k_used x k_used triangle, then the update x += V y
    y = np.zeros(k_used, dtype=np.complex128)
    for i in range(k_used - 1, -1, -1):
      acc = g[i] - np.dot(hess[i, i + 1:k_used], y[i + 1:])
      y[i] = acc / hess[i, i]
    for i in range(k_used):
      coef = y[i] if abs(y[i].imag) > 0 else float(y[i].real)
      x = _axpy(be, x, coef, basis[i])
    if converged:
      return x, 0
  r = be.subtraction(b, A_mv(x, *A_args, **A_kwargs))
  return x, (0 if _norm(be, r) <= target else maxiter)
