"""Finite MPS, nearest-neighbour MPOs and two-/one-site DMRG on backend tensors.

The heaviest real consumer of the hot path: every step is ``ncon`` (pairwise
GEMMs), ``qr`` / ``rq`` (gauge moves), ``svd`` with truncation (two-site splits) and
``eigsh_lanczos`` on vectors that never leave the device.  Behaviour follows the reference's
``matrixproductstates`` package: tensor conventions of ``mpo.py:129-220`` (MPO tensors are
(left bond, right bond, physical out, physical in)) and ``base_mps.py`` (MPS tensors are
(left bond, physical, right bond)), environments and effective-Hamiltonian contractions of
``dmrg.py:90-112``, gauge moves of ``base_mps.py:139-226``, local updates of
``dmrg.py:184-343``, sweeps of ``dmrg.py:345-559``.  Only backend methods are called, so the
same code runs on ``HipBackend`` (GPU suite) and on the oracle backend (CPU suite).
"""
from typing import List, Optional, Sequence

import numpy as np

from tensornetwork_amd.ncon import ncon


# ------------------------------------------------------------------------ MPOs
def xxz_mpo(backend, Jz: Sequence[float], Jxy: Sequence[float], Bz: Sequence[float], dtype=np.float64):
  """Tensors of ``mpo.FiniteXXZ`` (mpo.py:129-220) as a plain list."""
  from tensornetwork_amd import mpo as _mpo  # pylint: disable=import-outside-toplevel
  return _mpo.FiniteXXZ(Jz, Jxy, Bz, dtype, backend=backend).tensors


def tfi_mpo(backend, Jx: Sequence[float], Bz: Sequence[float], dtype=np.float64):
  """Tensors of ``mpo.FiniteTFI`` (mpo.py:223-288) as a plain list."""
  from tensornetwork_amd import mpo as _mpo  # pylint: disable=import-outside-toplevel
  return _mpo.FiniteTFI(Jx, Bz, dtype, backend=backend).tensors


def mpo_to_dense(mpo_host: Sequence[np.ndarray]) -> np.ndarray:
  """Dense Hamiltonian of a (small) MPO given as host arrays -- test / oracle helper."""
  acc = mpo_host[0][0]                       # (right bond, out, in)
  acc = np.transpose(acc, (1, 2, 0))         # (out, in, bond)
  for w in mpo_host[1:]:
    acc = np.einsum("abk,klcd->acbdl", acc, w)
    s = acc.shape
    acc = acc.reshape(s[0] * s[1], s[2] * s[3], s[4])
  return acc[:, :, 0]


# ------------------------------------------------------------------------- MPS
class FiniteMPS:
  """Open-boundary MPS with one orthogonality centre (finite_mps.py:26-121, base_mps.py:29-226)."""

  def __init__(self, tensors: List, backend, center_position: Optional[int] = None, canonicalize: bool = True):
    self.backend = backend
    self.tensors = [backend.convert_to_tensor(t) if not backend.is_tensor(t) else t for t in tensors]
    if self.tensors[0].shape[0] != 1 or self.tensors[-1].shape[2] != 1:
      raise ValueError("FiniteMPS needs boundary bond dimensions 1")
    if center_position is None:
      center_position = 0
      canonicalize = True
    if center_position < 0 or center_position >= len(self.tensors):
      raise ValueError("center_position = {} not between 0 <= center_position < N = {}"
                       .format(center_position, len(self.tensors)))
    self.center_position = center_position
    if canonicalize:
      self.canonicalize()

  @classmethod
  def random(cls, d: Sequence[int], D: Sequence[int], dtype, backend, seed: Optional[int] = None):
    """Random MPS (finite_mps.py:88-121): bond dimensions ``D`` (len N-1), tensors from the
    backend's ``randn`` (NumPy's global stream, like the reference)."""
    if len(D) != len(d) - 1:
      raise ValueError("len(D) = {} is different from len(d) - 1 = {}".format(len(D), len(d) - 1))
    if seed is not None:
      np.random.seed(seed)
    bonds = [1] + list(D) + [1]
    tensors = [backend.randn((bonds[n], d[n], bonds[n + 1]), dtype=dtype) for n in range(len(d))]
    return cls(tensors, backend, center_position=0, canonicalize=True)

  def __len__(self):
    return len(self.tensors)

  @property
  def dtype(self):
    return self.tensors[0].dtype

  @property
  def bond_dimensions(self):
    return [self.tensors[0].shape[0]] + [t.shape[2] for t in self.tensors]

  def bond_dimension(self, bond: int):
    """Dimension of bond ``bond`` (bond n sits left of site n; base_mps.py:238-245)."""
    if bond > len(self):
      raise IndexError(f"bond {bond} out of bounds for an MPS of length {len(self)}")
    if bond < len(self):
      return self.tensors[bond].shape[0]
    return self.tensors[-1].shape[2]

  @property
  def physical_dimensions(self):
    return [t.shape[1] for t in self.tensors]

  def get_tensor(self, site: int):
    """The tensor at ``site``; at the last site a connector matrix (InfiniteMPS) is absorbed from
    the right (base_mps.py:676-700)."""
    if site >= len(self):
      raise IndexError('index `site` = {} is out of range for len(mps)= {}'.format(site, len(self)))
    if site < 0:
      raise ValueError('index `site` has to be larger than 0 (found `site`={}).'.format(site))
    connector = getattr(self, "connector_matrix", None)
    if site == len(self) - 1 and connector is not None:
      return ncon([self.tensors[site], connector], [[-1, -2, 1], [1, -3]], backend=self.backend)
    return self.tensors[site]

  def save(self, path: str):
    raise NotImplementedError()           # finite_mps.py:328-329: neither does the reference

  # ------------------------------------------------------- transfer operators
  def left_transfer_operator(self, A, l, Abar):
    """l[a, a'] -> sum A[a, s, b] l[a, a'] Abar[a', s, b']  (base_mps.py:112-123)."""
    return ncon([A, l, Abar], [[1, 2, -1], [1, 3], [3, 2, -2]], backend=self.backend)

  def right_transfer_operator(self, B, r, Bbar):
    """r[b, b'] -> sum B[a, s, b] r[b, b'] Bbar[a', s, b']  (base_mps.py:125-136)."""
    return ncon([B, r, Bbar], [[-1, 2, 1], [1, 3], [-2, 2, 3]], backend=self.backend)

  def apply_transfer_operator(self, site: int, direction, matrix):
    """Left (1, 'l', 'left') or right (-1, 'r', 'right') action of the transfer operator of
    ``site`` on ``matrix`` (base_mps.py:262-285)."""
    t = self.tensors[site]
    if direction in (1, 'l', 'left'):
      return self.left_transfer_operator(t, matrix, self.backend.conj(t))
    if direction in (-1, 'r', 'right'):
      return self.right_transfer_operator(t, matrix, self.backend.conj(t))
    raise ValueError(f'unknown value {direction} for direction')

  def left_envs(self, sites: Sequence[int]):
    """{site: left reduced density matrix on the bond left of ``site``} (finite_mps.py:167-243).
    Identities up to and including the centre (the tensors there are left isometries), one
    transfer-operator application per site beyond it."""
    be = self.backend
    sites = np.array(sites, dtype=np.int64)
    if len(sites) == 0:
      return {}
    if not np.all(sites <= len(self)):
      raise ValueError('all elements of `sites` have to be <= N = {}'.format(len(self)))
    if not np.all(sites >= 0):
      raise ValueError('all elements of `sites` have to be positive')
    center = self.center_position
    wanted = set(int(x) for x in sites)
    envs = {}
    for site in wanted:
      if site <= center:
        envs[site] = be.eye(self.bond_dimension(site), dtype=self.dtype)
    last = max(wanted)
    if last > center:
      t = self.tensors[center]
      env = ncon([t, be.conj(t)], [[1, 2, -1], [1, 2, -2]], backend=be)
      for site in range(center + 1, last + 1):
        if site in wanted:
          envs[site] = env
        if site < last:
          env = self.apply_transfer_operator(site, 'left', env)
    return envs

  def right_envs(self, sites: Sequence[int]):
    """{site: right reduced density matrix on the bond right of ``site``} (finite_mps.py:245-326)."""
    be = self.backend
    sites = np.array(sites, dtype=np.int64)
    if len(sites) == 0:
      return {}
    if not np.all(sites < len(self)):
      raise ValueError('all elements of `sites` have to be < N = {}'.format(len(self)))
    if not np.all(sites >= -1):
      raise ValueError('all elements of `sites` have to be >= -1')
    center = self.center_position
    wanted = set(int(x) for x in sites)
    envs = {}
    for site in wanted:
      if site >= center:
        envs[site] = be.eye(self.bond_dimension(site + 1), dtype=self.dtype)
    first = min(wanted)
    if first < center:
      t = self.tensors[center]
      env = ncon([t, be.conj(t)], [[-1, 1, 2], [-2, 1, 2]], backend=be)
      for site in range(center - 1, first - 1, -1):
        if site in wanted:
          envs[site] = env
        if site > first:
          env = self.apply_transfer_operator(site, 'right', env)
    return envs

  def _norm(self, t):
    return float(np.real(self.backend.item(self.backend.norm(t))))

  def position(self, site: int, normalize: bool = True, D: Optional[int] = None,
               max_truncation_err: Optional[float] = None, norms_out: Optional[List[float]] = None):
    """Move the orthogonality centre to ``site`` with QR / RQ steps, or truncating SVDs when
    ``D`` / ``max_truncation_err`` ask for it (base_mps.py:139-226).  Returns the norm of the
    centre tensor before the last normalisation."""
    be = self.backend
    if site >= len(self.tensors) or site < 0:
      raise ValueError('site = {} not between values 0 < site < N = {}'.format(site, len(self)))
    if max_truncation_err is not None and max_truncation_err >= 1.0:
      raise ValueError("max_truncation_err should be 0 <= max_truncation_er < 1")
    z = None
    if site == self.center_position:
      z = self._norm(self.tensors[site])
      if normalize:
        self.tensors[site] = be.divide(self.tensors[site], z)
        if norms_out is not None:
          norms_out.append(z)       # a one-site (or already centred) state still reports its norm
      return z
    if site > self.center_position:
      for n in range(self.center_position, site):
        truncate = (D is not None and D < self.tensors[n].shape[2]) or max_truncation_err is not None
        if not truncate:
          iso, rest = be.qr(self.tensors[n], 2)
        else:
          iso, s, vh, _ = be.svd(self.tensors[n], 2, D, max_truncation_err)
          rest = be.broadcast_left_multiplication(s, vh)
        self.tensors[n] = iso
        self.tensors[n + 1] = ncon([rest, self.tensors[n + 1]], [[-1, 1], [1, -2, -3]], backend=be)
        z = self._norm(self.tensors[n + 1])
        if norms_out is not None:
          norms_out.append(z)
        if normalize:
          self.tensors[n + 1] = be.divide(self.tensors[n + 1], z)
    else:
      for n in range(self.center_position, site, -1):
        truncate = (D is not None and D < self.tensors[n].shape[0]) or max_truncation_err is not None
        if not truncate:
          rest, iso = be.rq(self.tensors[n], 1)
        else:
          u, s, iso, _ = be.svd(self.tensors[n], 1, D, max_truncation_err)
          rest = be.broadcast_right_multiplication(u, s)
        self.tensors[n] = iso
        self.tensors[n - 1] = ncon([self.tensors[n - 1], rest], [[-1, -2, 1], [1, -3]], backend=be)
        z = self._norm(self.tensors[n - 1])
        if norms_out is not None:
          norms_out.append(z)
        if normalize:
          self.tensors[n - 1] = be.divide(self.tensors[n - 1], z)
    self.center_position = site
    return z

  def canonicalize(self, normalize: bool = True):
    """Bring the state into canonical form around ``center_position`` (finite_mps.py:123-146):
    sweep the centre to the right end, back to the left end, then to where it was.  Every step is
    normalised and the norms are multiplied up on the host in float64 -- the norm of a random
    32-site state does not fit float32, and an un-normalised sweep would overflow inside the
    tensors.  Returns the norm of the state; with ``normalize=False`` it is put back on the centre."""
    be = self.backend
    pos = self.center_position
    norms: List[float] = []
    self.center_position = 0
    self.position(len(self.tensors) - 1, normalize=True, norms_out=norms)
    self.position(0, normalize=True, norms_out=norms)
    self.position(pos, normalize=True, norms_out=norms)
    z = self._norm(self.tensors[pos])
    self.tensors[pos] = be.divide(self.tensors[pos], z)
    total = z
    for x in norms:
      total *= x
    if not normalize:
      self.tensors[pos] = be.multiply(self.tensors[pos], total)
    return total

  def check_orthonormality(self, which: str, site: int):
    """|| A^dagger A - 1 || (left) or || B B^dagger - 1 || (right) (base_mps.py:616-650)."""
    be = self.backend
    t = self.get_tensor(site)
    if which in ("l", "left"):
      g = ncon([t, be.conj(t)], [[1, 2, -1], [1, 2, -2]], backend=be)
    elif which in ("r", "right"):
      g = ncon([t, be.conj(t)], [[-1, 1, 2], [-2, 1, 2]], backend=be)
    else:
      raise ValueError("which = {} is not recognized.".format(which))
    return self._norm(be.subtraction(g, be.eye(g.shape[0], dtype=self.dtype)))

  def check_canonical(self):
    """L2 norm of the per-site isometry deviations around the centre (finite_mps.py:148-165)."""
    total = 0.0
    for site in range(len(self.tensors)):
      if site < self.center_position:
        total += self.check_orthonormality('l', site) ** 2
      elif site > self.center_position:
        total += self.check_orthonormality('r', site) ** 2
    return float(np.sqrt(total))

  def apply_one_site_gate(self, gate, site: int):
    """tensors[site][a, s, b] <- sum_t gate[s, t] tensors[site][a, t, b] (base_mps.py:597-614).
    Generally breaks the canonical form; `position` restores it."""
    if len(gate.shape) != 2:
      raise ValueError('rank of gate is {} but has to be 2'.format(len(gate.shape)))
    if site < 0 or site >= len(self):
      raise ValueError('site = {} is not between 0 <= site < N={}'.format(site, len(self)))
    self.tensors[site] = ncon([gate, self.tensors[site]], [[-2, 1], [-1, 1, -3]], backend=self.backend)

  def apply_two_site_gate(self, gate, site1: int, site2: int, max_singular_values: Optional[int] = None,
                          max_truncation_err: Optional[float] = None, relative: bool = False):
    """Apply a (d, d, d, d) gate [out1, out2, in1, in2] to neighbouring sites and split the result
    with a truncated SVD (base_mps.py:481-596).  The centre must sit on one of the two sites and
    stays where it was (the singular values are absorbed into that site).  Returns the discarded
    singular values."""
    be = self.backend
    if site2 != site1 + 1:
      raise ValueError("site2 ={} != site1={}. Only nearest neighbor gates are currently supported"
                       .format(site2, site1))
    if self.center_position not in (site1, site2):
      raise ValueError("center_position is not on site1 or site2")
    theta = ncon([self.tensors[site1], self.tensors[site2], gate],
                 [[-1, 1, 2], [2, 3, -4], [-2, -3, 1, 3]], backend=be)
    u, s, vh, trunc = be.svd(theta, 2, max_singular_values, max_truncation_err, relative=relative)
    if self.center_position == site1:
      self.tensors[site1] = be.broadcast_right_multiplication(u, s)
      self.tensors[site2] = vh
    else:
      self.tensors[site1] = u
      self.tensors[site2] = be.broadcast_left_multiplication(s, vh)
    return trunc

  def measure_local_operator(self, ops: Sequence, sites: Sequence[int]):
    """<psi| op_n |psi> for each (op, site) (base_mps.py:287-320): the site tensor sandwiched
    between its left and right reduced density matrices; the gauge is left untouched."""
    be = self.backend
    if len(ops) != len(sites):
      raise ValueError('measure_1site_ops: len(ops) has to be len(sites)!')
    rs = self.right_envs(sites)
    ls = self.left_envs(sites)
    out = []
    for op, site in zip(ops, sites):
      site = int(site)
      t = self.tensors[site]
      val = ncon([ls[site], t, op, be.conj(t), rs[site]],
                 [[1, 2], [1, 3, 5], [4, 3], [2, 4, 6], [5, 6]], backend=be)
      out.append(be.item(val))
    return out

  def measure_two_body_correlator(self, op1, op2, site1: int, sites2: Sequence[int]):
    """<op1[site1] op2[s]> for every s in ``sites2``, returned in ascending order of s
    (base_mps.py:322-479; for s == site1 the product op1 @ op2 is measured).  One environment
    sweep to each side of ``site1``: the op1-dressed environment is pushed outwards with
    transfer operators and closed with op2 and the far-side density matrix at each wanted site."""
    be = self.backend
    N = len(self)
    if site1 < 0:
      raise ValueError("Site site1 out of range: {} not between 0 <= site < N = {}.".format(site1, N))
    sites2 = np.array(sites2, dtype=np.int64)
    left_sites = sorted(set(int(x) for x in sites2[sites2 < site1]))
    right_sites = sorted(set(int(x) for x in sites2[sites2 > site1]))
    rs = self.right_envs([site1] + right_sites)
    ls = self.left_envs(left_sites + [site1])
    t1 = self.tensors[site1]
    out = []
    if left_sites:
      # env[a, a']: everything right of the bond, with op1 inserted at site1
      env = ncon([t1, op1, be.conj(t1), rs[site1]], [[-1, 1, 2], [3, 1], [-2, 3, 4], [2, 4]], backend=be)
      vals = {}
      for n in range(site1 - 1, left_sites[0] - 1, -1):
        t = self.tensors[n]
        if n in left_sites:
          vals[n] = ncon([ls[n], t, op2, be.conj(t), env],
                         [[1, 2], [1, 3, 5], [4, 3], [2, 4, 6], [5, 6]], backend=be)
        if n > left_sites[0]:
          env = self.apply_transfer_operator(n, 'right', env)
      out.extend(vals[n] for n in left_sites)
    if site1 in sites2:
      op12 = ncon([op1, op2], [[-1, 1], [1, -2]], backend=be)
      out.append(ncon([ls[site1], t1, op12, be.conj(t1), rs[site1]],
                      [[1, 2], [1, 3, 5], [4, 3], [2, 4, 6], [5, 6]], backend=be))
    if right_sites:
      env = ncon([ls[site1], t1, op1, be.conj(t1)], [[1, 2], [1, 3, -1], [4, 3], [2, 4, -2]], backend=be)
      for n in range(site1 + 1, right_sites[-1] + 1):
        t = self.tensors[n]
        if n in right_sites:
          out.append(ncon([env, t, op2, be.conj(t), rs[n]],
                          [[1, 2], [1, 3, 5], [4, 3], [2, 4, 6], [5, 6]], backend=be))
        if n < right_sites[-1]:
          env = self.apply_transfer_operator(n, 'left', env)
    return [be.item(o) for o in out]


class InfiniteMPS(FiniteMPS):
  """Translation-invariant MPS given by one unit cell (infinite_mps.py:26-307).  ``tensors`` is the
  unit cell, ``connector_matrix`` (optional) sits between consecutive cells; ``canonicalize``
  brings the state into Schmidt-canonical form from the dominant eigenvectors of the unit-cell
  transfer matrix (``backend.eigs`` on device vectors)."""

  def __init__(self, tensors: List, backend, center_position: Optional[int] = None, connector_matrix=None):
    # pylint: disable=super-init-not-called
    self.backend = backend
    self.tensors = [backend.convert_to_tensor(t) if not backend.is_tensor(t) else t for t in tensors]
    if center_position is not None and (center_position < 0 or center_position >= len(self.tensors)):
      raise ValueError("`center_position = {}` is different from `None` and not between "
                       "0 <= center_position < {}".format(center_position, len(self.tensors)))
    self.center_position = center_position
    self.connector_matrix = connector_matrix

  @classmethod
  def random(cls, d: Sequence[int], D: Sequence[int], dtype, backend, seed: Optional[int] = None):
    """Random unit cell with bond dimensions ``D`` (len(d) + 1, periodic: D[0] == D[-1]); centre at 0
    (infinite_mps.py:64-92)."""
    if len(D) != len(d) + 1:
      raise ValueError('len(D) = {} is different from len(d) + 1= {}'.format(len(D), len(d) + 1))
    if D[-1] != D[0]:
      raise ValueError('D[0]={} != D[-1]={}.'.format(D[0], D[-1]))
    if seed is not None:
      np.random.seed(seed)
    return cls([backend.randn((D[n], d[n], D[n + 1]), dtype=dtype) for n in range(len(d))], backend,
               center_position=0)

  def left_envs(self, sites):
    raise NotImplementedError()

  def right_envs(self, sites):
    raise NotImplementedError()

  def position(self, site: int, normalize: bool = True, D: Optional[int] = None,
               max_truncation_err: Optional[float] = None, norms_out: Optional[List[float]] = None):
    if self.center_position is None:
      raise ValueError("BaseMPS.center_position is `None`, cannot shift `center_position`."
                       "Reset `center_position` manually or use `canonicalize`")
    return super().position(site, normalize, D, max_truncation_err, norms_out)

  def unit_cell_transfer_operator(self, direction, matrix):
    """``matrix`` pushed through the whole unit cell, site by site (infinite_mps.py:94-101)."""
    sites = range(len(self))
    if direction in (-1, 'r', 'right'):
      sites = reversed(sites)
    for site in sites:
      matrix = self.apply_transfer_operator(site, direction, matrix)
    return matrix

  def transfer_matrix_eigs(self, direction, initial_state=None, precision: float = 1e-10,
                           num_krylov_vecs: int = 30, maxiter: Optional[int] = None):
    """Dominant eigenvalue (host scalar) and eigenvector (D x D tensor) of the unit-cell transfer
    matrix (infinite_mps.py:103-166).  For a real state both are real: the Arnoldi basis of a real
    operator stays real and the dominant eigenvalue of a transfer matrix is real, so the complex
    result ``eigs`` returns by convention is narrowed back to the state's dtype."""
    be = self.backend
    D = self.bond_dimensions[0]

    def mv(vector):
      return be.reshape(self.unit_cell_transfer_operator(direction, be.reshape(vector, (D, D))), (D * D,))

    if initial_state is None:
      initial_state = be.randn((D * D,), dtype=self.dtype)
    else:
      initial_state = be.reshape(initial_state, (D * D,))
    if D == 1:
      initial_state = be.divide(initial_state, self._norm(initial_state))
      result = mv(initial_state)
      return self._norm(result), be.reshape(result, (D, D))
    ncv = min(num_krylov_vecs, D * D)
    eta, vecs = be.eigs(A=mv, initial_state=initial_state, num_krylov_vecs=ncv, numeig=min(1, ncv - 2),
                        tol=precision, which='LR', maxiter=maxiter, dtype=self.dtype)
    vec, val = vecs[0], eta[0]
    if np.dtype(self.dtype).kind != "c":
      vec, val = be.real(vec), float(np.real(val))
    return val, be.reshape(vec, (D, D))

  def _psd_factors(self, mat, cutoff):
    """For a (numerically) positive matrix ``mat``: normalise to unit trace, hermitise,
    diagonalise; returns (u, sqrt(w), sqrt(1/w)) with eigenvalues at or below ``cutoff`` -- and their
    inverses -- set to zero (the reference's pseudo-inverse, infinite_mps.py:229-241)."""
    be = self.backend
    mat = be.divide(mat, be.trace(mat))
    mat = be.divide(be.addition(mat, be.transpose(be.conj(mat), (1, 0))), 2.0)
    w, u = be.eigh(mat)
    w = be.divide(w, be.norm(w))
    mask = w <= cutoff
    w = be.index_update(w, mask, 0.0)
    winv = be.index_update(be.divide(1.0, w), mask, 0.0)
    return u, be.sqrt(w), be.sqrt(winv)

  def canonicalize(self, left_initial_state=None, right_initial_state=None, precision: float = 1e-10,  # pylint: disable=arguments-differ
                   truncation_threshold: float = 1e-15, D: Optional[int] = None, num_krylov_vecs: int = 50,
                   maxiter: Optional[int] = 1000, pseudo_inverse_cutoff: Optional[float] = None):
    """Schmidt-canonical form (infinite_mps.py:178-307).  With ``l = X^dagger X`` and ``r = Y Y^dagger`` the
    dominant left / right eigenvectors of the unit-cell transfer matrix, the SVD ``X Y = U lam V``
    gives the Schmidt values ``lam`` on the cell boundary; the gauge ``lam V Y^-1`` is absorbed into the
    first tensor and ``X^-1 U lam`` into the last, a QR sweep makes the cell left-orthonormal, and the
    connector becomes ``lam^-1``.  Returns the norm of ``lam``."""
    be = self.backend
    if pseudo_inverse_cutoff is None:
      pseudo_inverse_cutoff = be.eps(self.dtype)
    if self.center_position is None:
      self.center_position = 0
    self.position(0)
    eta, l = self.transfer_matrix_eigs('left', left_initial_state, precision, num_krylov_vecs, maxiter)
    self.tensors[0] = be.divide(self.tensors[0], float(np.sqrt(abs(eta))))
    u, sw, siw = self._psd_factors(l, pseudo_inverse_cutoff)
    sqrtl = be.transpose(be.broadcast_right_multiplication(u, sw), (1, 0))        # sqrt(w) u^T
    inv_sqrtl = be.broadcast_right_multiplication(be.conj(u), siw)                # conj(u) / sqrt(w)
    _, r = self.transfer_matrix_eigs('right', right_initial_state, precision, num_krylov_vecs, maxiter)
    u, sw, siw = self._psd_factors(r, pseudo_inverse_cutoff)
    sqrtr = be.broadcast_right_multiplication(u, sw)                              # u sqrt(w)
    inv_sqrtr = be.transpose(be.broadcast_right_multiplication(be.conj(u), siw), (1, 0))
    U, singvals, V, _ = be.svd(be.tensordot(sqrtl, sqrtr, 1), 1, D, truncation_threshold, relative=True)
    lam = be.diagflat(singvals)
    self.tensors[0] = ncon([lam, V, inv_sqrtr, self.tensors[0]],
                           [[-1, 1], [1, 2], [2, 3], [3, -2, -3]], backend=be)
    self.tensors[-1] = ncon([self.get_tensor(len(self) - 1), inv_sqrtl, U, lam],
                            [[-1, -2, 1], [1, 2], [2, 3], [3, -3]], backend=be)
    self.connector_matrix = None               # absorbed just above
    self.position(len(self) - 1)
    lam_norm = self._norm(singvals)
    self.center_position = len(self) - 1
    self.connector_matrix = be.inv(be.divide(lam, lam_norm))
    return lam_norm


# ------------------------------------------------------------------------ DMRG
class FiniteDMRG:
  """Ground-state search for a finite MPO (dmrg.py:23-604).

  Left / right environments are rank-3 tensors (MPO bond, ket bond, bra bond); the effective
  Hamiltonian is applied by ``ncon`` (dmrg.py:90-112) inside ``backend.eigsh_lanczos``."""

  def __init__(self, mps: FiniteMPS, mpo: List):
    if len(mps) != len(mpo):
      raise ValueError('len(mps) = {} is different from len(mpo) = {}'.format(len(mps), len(mpo)))
    self.mps = mps
    self.mpo = list(mpo)
    self.backend = mps.backend
    be = self.backend
    if self.mpo[0].dtype != mps.dtype:
      raise TypeError('mps.dtype = {} is different from mpo.dtype = {}'.format(mps.dtype, self.mpo[0].dtype))
    self.left_envs = {0: be.ones((self.mpo[0].shape[0], 1, 1), dtype=mps.dtype)}
    self.right_envs = {len(mps) - 1: be.ones((self.mpo[-1].shape[1], 1, 1), dtype=mps.dtype)}
    self.last_truncation = None

  # contractions (index strings of dmrg.py:90-112)
  def single_site_matvec(self, t, L, w, R):
    return ncon([L, t, w, R], [[3, 1, -1], [1, 2, 4], [3, 5, -2, 2], [5, 4, -3]], backend=self.backend)

  def two_site_matvec(self, theta, L, wl, wr, R):
    return ncon([L, theta, wl, wr, R],
                [[3, 1, -1], [1, 2, 5, 6], [3, 4, -2, 2], [4, 7, -3, 5], [7, 6, -4]], backend=self.backend)

  def add_left_layer(self, L, t, w):
    return ncon([L, t, w, self.backend.conj(t)], [[2, 1, 5], [1, 3, -2], [2, -1, 4, 3], [5, 4, -3]],
                backend=self.backend)

  def add_right_layer(self, R, t, w):
    return ncon([R, t, w, self.backend.conj(t)], [[2, 1, 5], [-2, 3, 1], [-1, 2, 4, 3], [-3, 4, 5]],
                backend=self.backend)

  def compute_right_envs(self):
    for n in range(len(self.mps) - 1, self.mps.center_position, -1):
      self.right_envs[n - 1] = self.add_right_layer(self.right_envs[n], self.mps.tensors[n], self.mpo[n])

  def compute_left_envs(self):
    for n in range(self.mps.center_position):
      self.left_envs[n + 1] = self.add_left_layer(self.left_envs[n], self.mps.tensors[n], self.mpo[n])

  def position(self, site: int):
    """Move the MPS centre to ``site`` keeping the environments consistent (dmrg.py:114-157)."""
    old = self.mps.center_position
    if site >= len(self.mps) or site < 0:
      raise IndexError("site {} is out of range".format(site))
    if site == old:
      return
    self.mps.position(site)
    if site > old:
      for n in range(old, site):
        self.left_envs[n + 1] = self.add_left_layer(self.left_envs[n], self.mps.tensors[n], self.mpo[n])
    else:
      for n in range(old, site, -1):
        self.right_envs[n - 1] = self.add_right_layer(self.right_envs[n], self.mps.tensors[n], self.mpo[n])

  def _lanczos(self, matvec, args, init, num_krylov_vecs, tol, delta, ndiag):
    be = self.backend
    if getattr(self, "deferred_lanczos", False):
      # opt-in: coefficients stay on the device between convergence checks (krylov.eigsh_lanczos_deferred)
      from tensornetwork_amd import krylov  # pylint: disable=import-outside-toplevel
      energies, states = krylov.eigsh_lanczos_deferred(be, matvec, args, init, None, None, num_krylov_vecs, 1, tol,
                                                       delta, ndiag, False)
    else:
      energies, states = be.eigsh_lanczos(A=matvec, args=args, initial_state=init, num_krylov_vecs=num_krylov_vecs,
                                          numeig=1, tol=tol, delta=delta, ndiag=ndiag, reorthogonalize=False)
    state = states[0]
    return energies[0], be.divide(state, float(np.real(be.item(be.norm(state)))))

  def _optimize_2s_local(self, max_bond_dim, sweep_dir, num_krylov_vecs=10, tol=1e-5, delta=1e-6, ndiag=10):
    """dmrg.py:251-343."""
    be, mps = self.backend, self.mps
    site = mps.center_position
    left = site if sweep_dir == "right" else site - 1
    theta = ncon([mps.tensors[left], mps.tensors[left + 1]], [[-1, -2, 1], [1, -3, -4]], backend=be)
    energy, ground = self._lanczos(self.two_site_matvec,
                                   [self.left_envs[left], self.mpo[left], self.mpo[left + 1],
                                    self.right_envs[left + 1]], theta, num_krylov_vecs, tol, delta, ndiag)
    u, s, vh, trunc = be.svd(ground, 2, max_bond_dim, None)
    self.last_truncation = trunc
    if sweep_dir == "right":
      mps.tensors[left] = u
      mps.tensors[left + 1] = be.broadcast_left_multiplication(s, vh)
      mps.center_position = left + 1
      self.left_envs[left + 1] = self.add_left_layer(self.left_envs[left], u, self.mpo[left])
    else:
      mps.tensors[left + 1] = vh
      mps.tensors[left] = be.broadcast_right_multiplication(u, s)
      mps.center_position = left
      self.right_envs[left] = self.add_right_layer(self.right_envs[left + 1], vh, self.mpo[left + 1])
    return energy

  def _optimize_1s_local(self, sweep_dir, num_krylov_vecs=10, tol=1e-5, delta=1e-6, ndiag=10):
    """dmrg.py:184-249."""
    be, mps = self.backend, self.mps
    site = mps.center_position
    energy, ground = self._lanczos(self.single_site_matvec,
                                   [self.left_envs[site], self.mpo[site], self.right_envs[site]],
                                   mps.tensors[site], num_krylov_vecs, tol, delta, ndiag)
    if sweep_dir == "right":
      q, r = be.qr(ground, 2)
      mps.tensors[site] = q
      if site < len(mps) - 1:
        mps.center_position += 1
        mps.tensors[site + 1] = ncon([r, mps.tensors[site + 1]], [[-1, 1], [1, -2, -3]], backend=be)
        self.left_envs[site + 1] = self.add_left_layer(self.left_envs[site], q, self.mpo[site])
    else:
      r, q = be.rq(ground, 1)
      mps.tensors[site] = q
      if site > 0:
        mps.center_position -= 1
        mps.tensors[site - 1] = ncon([mps.tensors[site - 1], r], [[-1, -2, 1], [1, -3]], backend=be)
        self.right_envs[site - 1] = self.add_right_layer(self.right_envs[site], q, self.mpo[site])
    return energy

  def _prepare(self):
    self.position(0)
    self.compute_right_envs()

  def run_two_site(self, max_bond_dim: int, num_sweeps: int = 4, precision: float = 1e-6,
                   num_krylov_vecs: int = 10, tol: float = 1e-5, delta: float = 1e-6, ndiag: int = 10):
    """Two-site DMRG sweeps (dmrg.py:445-559): right then left over all bonds, until the energy
    changes by less than ``precision`` between sweeps or ``num_sweeps`` is reached.  Returns the
    last local energy (the total energy: the local problem is the full projected Hamiltonian)."""
    if num_sweeps == 0:
      return self.compute_energy()
    n = len(self.mps)
    self._prepare()
    energy, previous = None, 1e100
    for _ in range(num_sweeps):
      for _site in range(n - 2):
        energy = self._optimize_2s_local(max_bond_dim, "right", num_krylov_vecs, tol, delta, ndiag)
      # last bond (n-2, n-1): optimise while turning around
      energy = self._optimize_2s_local(max_bond_dim, "right", num_krylov_vecs, tol, delta, ndiag)
      for _site in range(n - 2):
        energy = self._optimize_2s_local(max_bond_dim, "left", num_krylov_vecs, tol, delta, ndiag)
      self.position(0)
      if abs(float(np.real(energy)) - previous) < precision:
        break
      previous = float(np.real(energy))
    return energy

  def run_one_site(self, num_sweeps: int = 4, precision: float = 1e-6, num_krylov_vecs: int = 10,
                   tol: float = 1e-5, delta: float = 1e-6, ndiag: int = 10):
    """Single-site DMRG sweeps (dmrg.py:345-443); bond dimensions stay fixed."""
    if num_sweeps == 0:
      return self.compute_energy()
    n = len(self.mps)
    self._prepare()
    energy, previous = None, 1e100
    for _ in range(num_sweeps):
      for _site in range(n - 1):
        energy = self._optimize_1s_local("right", num_krylov_vecs, tol, delta, ndiag)
      for _site in range(n - 1):
        energy = self._optimize_1s_local("left", num_krylov_vecs, tol, delta, ndiag)
      if abs(float(np.real(energy)) - previous) < precision:
        break
      previous = float(np.real(energy))
    return energy

  def compute_energy(self):
    """<psi|H|psi> from the environments (dmrg.py:561-569)."""
    be = self.backend
    self._prepare()
    t = self.mps.tensors[0]
    val = ncon([self.left_envs[0], t, self.mpo[0], self.right_envs[0], be.conj(t)],
               [[3, 1, 6], [1, 2, 4], [3, 5, 7, 2], [5, 4, 8], [6, 7, 8]], backend=be)
    return be.item(val)
