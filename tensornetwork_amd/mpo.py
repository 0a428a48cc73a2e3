"""Matrix product operators: the containers and model Hamiltonians of the reference's
``matrixproductstates/mpo.py`` (BaseMPO :25-74, InfiniteMPO :77-102, FiniteMPO :105-126,
FiniteXXZ :129-220, FiniteTFI :223-288, FiniteFreeFermion2D :291-387).

MPO tensors have index order (left bond, right bond, physical out, physical in).  A model is
written down here as a sparse operator-valued matrix ``{(row, col): 2x2 operator}`` per site and
densified once on the host; the tensors then live on the backend (HBM for ``HipBackend``) and are
consumed by ``FiniteDMRG`` through ``ncon``.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


def _resolve_backend(backend):
  from tensornetwork_amd.ncon import _resolve_backend as _rb  # pylint: disable=import-outside-toplevel
  return _rb(backend)


def _densify(rows: int, cols: int, entries: Dict[Tuple[int, int], np.ndarray], dtype) -> np.ndarray:
  w = np.zeros((rows, cols, 2, 2), dtype=dtype)
  for (r, c), op in entries.items():
    w[r, c] = op
  return w


class BaseMPO:
  """A list of rank-4 tensors on one backend, all of one dtype (mpo.py:25-74)."""

  def __init__(self, tensors: Sequence, backend=None, name: Optional[str] = None):
    self.backend = _resolve_backend(backend)
    be = self.backend
    self.tensors = [t if be.is_tensor(t) else be.convert_to_tensor(t) for t in tensors]
    if self.tensors and not all(t.dtype == self.tensors[0].dtype for t in self.tensors):
      raise TypeError('not all dtypes in BaseMPO.tensors are the same')
    self.name = name

  def __iter__(self):
    return iter(self.tensors)

  def __len__(self) -> int:
    return len(self.tensors)

  def __getitem__(self, site):
    return self.tensors[site]

  @property
  def dtype(self):
    if not all(t.dtype == self.tensors[0].dtype for t in self.tensors):
      raise TypeError('not all dtypes in BaseMPO.tensors are the same')
    return self.tensors[0].dtype

  @property
  def bond_dimensions(self) -> List[int]:
    """Ancillary dimensions, length N + 1."""
    return [self.tensors[0].shape[0]] + [t.shape[1] for t in self.tensors]


class InfiniteMPO(BaseMPO):
  """Unit cell of a translation-invariant MPO: outer ancillary dimensions must match (mpo.py:77-102)."""

  def __init__(self, tensors: Sequence, backend=None, name: Optional[str] = None):
    super().__init__(tensors, backend, name)
    if self.bond_dimensions[0] != self.bond_dimensions[-1]:
      raise ValueError('left and right MPO ancillary dimension have to match')

  def roll(self, num_sites: int) -> None:
    """Cyclically shift the unit cell by ``num_sites`` sites."""
    self.tensors = self.tensors[num_sites:] + self.tensors[:num_sites]


class FiniteMPO(BaseMPO):
  """Open-boundary MPO: outer ancillary dimensions are 1 (mpo.py:105-126)."""

  def __init__(self, tensors: Sequence, backend=None, name: Optional[str] = None):
    super().__init__(tensors, backend, name)
    if self.bond_dimensions[0] != 1 or self.bond_dimensions[-1] != 1:
      raise ValueError('left and right MPO ancillary dimensions have to be 1')


def _chain(bulk_entries, dim: int, n_sites: int, dtype) -> List[np.ndarray]:
  """Lower-triangular MPO chain from a per-site bulk matrix: the first site keeps the last row, the
  last site the first column."""
  out = []
  for site in range(n_sites):
    w = _densify(dim, dim, bulk_entries(site), dtype)
    if site == 0:
      w = w[dim - 1:dim]
    if site == n_sites - 1:
      w = w[:, 0:1]
    out.append(np.ascontiguousarray(w))
  return out


class FiniteXXZ(FiniteMPO):
  """H = sum_n Jz[n] Sz Sz + Jxy[n]/2 (S+ S- + S- S+) + Bz[n] Sz on an open chain (mpo.py:129-220).
  Basis index 0 = spin down (Sz = -1/2).  Ancillary dimension 5: identity strings on (0,0) and (4,4),
  (S-, S+, Sz) open a bond term in row 4 and are closed by (S+, S-, Sz) in column 0."""

  def __init__(self, Jz, Jxy, Bz, dtype, backend=None, name: str = 'XXZ_MPO'):
    self.Jz, self.Jxy, self.Bz = (np.asarray(x).astype(dtype) for x in (Jz, Jxy, Bz))
    n = len(self.Bz)
    sz = np.diag([-0.5, 0.5]).astype(dtype)
    sp = np.array([[0, 0], [1, 0]]).astype(dtype)
    sm = sp.T.copy()
    one = np.eye(2, dtype=dtype)

    def bulk(site):
      e = {(0, 0): one, (1, 0): sp, (2, 0): sm, (3, 0): sz, (4, 0): self.Bz[site] * sz, (4, 4): one}
      if site < n - 1:
        e.update({(4, 1): self.Jxy[site] / 2.0 * sm, (4, 2): self.Jxy[site] / 2.0 * sp, (4, 3): self.Jz[site] * sz})
      return e

    super().__init__(_chain(bulk, 5, n, dtype), backend, name)


class FiniteTFI(FiniteMPO):
  """H = sum_n Jx[n] X X + Bz[n] Z with Z = diag(-1, 1) on an open chain (mpo.py:223-288)."""

  def __init__(self, Jx, Bz, dtype, backend=None, name: str = 'TFI_MPO'):
    self.Jx, self.Bz = np.asarray(Jx).astype(dtype), np.asarray(Bz).astype(dtype)
    n = len(self.Bz)
    sx = np.array([[0, 1], [1, 0]]).astype(dtype)
    sz = np.diag([-1, 1]).astype(dtype)
    one = np.eye(2, dtype=dtype)

    def bulk(site):
      e = {(0, 0): one, (1, 0): sx, (2, 0): self.Bz[site] * sz, (2, 2): one}
      if site < n - 1:
        e[(2, 1)] = self.Jx[site] * sx
      return e

    super().__init__(_chain(bulk, 3, n, dtype), backend, name)


class FiniteFreeFermion2D(FiniteMPO):
  """Spinless free fermions on an N1 x N2 grid, H = sum t c^dag c + h.c. + v n, as an MPO snaked
  along the vertical direction (mpo.py:291-387).  Jordan-Wigner: a hop over distance r along the
  snake carries r - 1 parity operators, so the MPO keeps 2 N1 delay lines (N1 for c^dag ... c and N1
  for c ... c^dag) whose entries shift by one channel per site through sigma_z: channel 1 closes a
  vertical (distance 1) hop, channel N1 a horizontal (distance N1) hop."""

  def __init__(self, t1: float, t2: float, v: float, N1: int, N2: int, dtype, backend=None,
               name: str = '2DTFI_MPO'):
    self.t1, self.t2, self.v, self.N1, self.N2 = t1, t2, v, N1, N2
    one = np.eye(2, dtype=dtype)
    c = np.array([[0, 1], [0, 0]]).astype(dtype)
    cdag = c.T.conj()
    num = np.diag([0, 1]).astype(dtype)
    par = np.diag([1, -1]).astype(dtype)
    last = 2 * N1 + 1
    n_sites = N1 * N2

    def opening(hop1, hop2):
      # row `last`: start the delay lines (later keys win when N1 == 1 makes channels coincide)
      e = {(last, 0): v * num}
      for col, op in ((1, hop1 * cdag), (N1, hop2 * cdag), (N1 + 1, hop1 * c), (2 * N1, hop2 * c)):
        e[(last, col)] = op
      e[(last, last)] = one
      return e

    def bulk(site):
      if site == 0:
        return opening(t1, t2)
      hop1 = 0 if (site + 1) % N1 == 0 else t1           # no vertical bond across a column end
      hop2 = t2 if site < N1 * (N2 - 1) else 0           # no horizontal bond out of the last column
      e = {(0, 0): one, (1, 0): c, (N1 + 1, 0): cdag}
      for ch in list(range(2, N1 + 1)) + list(range(N1 + 2, 2 * N1 + 1)):
        e[(ch, ch - 1)] = par
      if site < n_sites - 1:
        e.update(opening(hop1, hop2))
      else:
        e[(last, 0)] = v * num
      return e

    super().__init__(_chain(bulk, last + 1, n_sites, dtype), backend, name)
