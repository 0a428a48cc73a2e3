"""NumPy semantics of the backend's methods (the reference hands every NumPy dtype to every backend,
tests/testing_utils.py:12-20; its NumPy backend is np.* verbatim, numpy_backend.py): dtype promotion between tensors
and with Python scalars (NEP 50), sum / trace over random axes, slices, transposes, abs / sign / conj, norm, mixed-dtype
tensordot -- HipBackend's host code on the emulated C ABI (tests/emu_tnh.py) against NumPy, 12 dtypes."""
import numpy as np

from emu_tnh import emulated_backend


def test_backend_methods_follow_numpy_dtype_and_value_semantics():
  fails = []
  DT = [np.float32, np.float64, np.complex64, np.complex128, np.int32, np.int64, np.bool_, np.uint8, np.int8, np.int16,
        np.uint16, np.uint32]
  rng = np.random.default_rng(0)

  def rand(shape, dt):
    dt = np.dtype(dt)
    if dt.kind == "b":
      return rng.integers(0, 2, size=shape).astype(dt)
    if dt.kind in "iu":
      info = np.iinfo(dt)
      return rng.integers(max(info.min, -100), min(info.max, 100), size=shape).astype(dt)
    x = rng.standard_normal(shape)
    if dt.kind == "c":
      x = x + 1j * rng.standard_normal(shape)
    return x.astype(dt)

  def close(got, ref, what):
    got = np.asarray(got)
    ref = np.asarray(ref)
    ok = got.shape == ref.shape
    if ok:
      if ref.dtype.kind in "fc":
        ok = got.dtype.kind == ref.dtype.kind or (got.dtype.kind == "f" and ref.dtype.kind == "f")
        tol = 1e-4 if ref.dtype in (np.float32, np.complex64, np.float16) else 1e-10
        ok = ok and np.allclose(got, ref, rtol=tol, atol=tol, equal_nan=True)
      else:
        ok = got.dtype == ref.dtype and np.array_equal(got, ref)
    if not ok:
      fails.append((what, got.dtype, ref.dtype, got.shape, ref.shape))

  with emulated_backend() as be:
    with np.errstate(all="ignore"):
      for dt in DT:
        for trial in range(6):
          nd = int(rng.integers(1, 5))
          shape = tuple(int(rng.integers(1, 5)) for _ in range(nd))
          x = rand(shape, dt); dx = be.convert_to_tensor(x)
          # sum over random axes
          axes = tuple(sorted(rng.choice(nd, int(rng.integers(1, nd + 1)), replace=False).tolist()))
          for kd in (False, True):
            try:
              close(be.sum(dx, axis=axes, keepdims=kd), np.sum(x, axis=axes, keepdims=kd), f"sum {dt} {shape} {axes} {kd}")
            except Exception as e:
              fails.append(("sum exc", dt))
          try:
            close(be.sum(dx), np.sum(x), f"sum all {dt} {shape}")
          except Exception as e:
            fails.append(("sumall exc", dt))
          # transpose / reshape / conj / abs / sign
          perm = tuple(rng.permutation(nd).tolist())
          close(be.transpose(dx, perm), np.transpose(x, perm), f"transpose {dt} {shape} {perm}")
          close(be.reshape(dx, (-1,)), x.reshape(-1), f"reshape {dt}")
          close(be.conj(dx), np.conj(x), f"conj {dt}")
          if np.dtype(dt).kind != "b":
            try:
              close(be.abs(dx), np.abs(x), f"abs {dt}")
            except Exception as e:
              fails.append(("abs exc", dt))
            if np.dtype(dt).kind in "fi":
              close(be.sign(dx), np.sign(x), f"sign {dt}")
          # trace
          if nd >= 2:
            a1, a2 = rng.choice(nd, 2, replace=False).tolist()
            off = int(rng.integers(-1, 2))
            try:
              close(be.trace(dx, offset=off, axis1=a1, axis2=a2), np.trace(x, offset=off, axis1=a1, axis2=a2), f"trace {dt} {shape} {a1} {a2} {off}")
            except Exception as e:
              fails.append(("trace exc", dt))
          # slice
          start = tuple(int(rng.integers(0, s)) for s in shape)
          size = tuple(int(rng.integers(1, s - st + 1)) for s, st in zip(shape, start))
          close(be.slice(dx, start, size), x[tuple(slice(a, a + b) for a, b in zip(start, size))], f"slice {dt}")
          # norm
          if np.dtype(dt).kind in "fc":
            close(be.norm(dx), np.linalg.norm(x), f"norm {dt}")
        # binary ops with every other dtype
        for dt2 in DT:
          x = rand((3, 4), dt); y = rand((4,), dt2)
          dx, dy = be.convert_to_tensor(x), be.convert_to_tensor(y)
          for name, fn, rf in (("add", be.addition, np.add), ("sub", be.subtraction, np.subtract), ("mul", be.multiply, np.multiply),
                               ("div", be.divide, np.divide)):
            if name == "sub" and (np.dtype(dt).kind == "b" and np.dtype(dt2).kind == "b"):
              continue
            if name == "div" and (x.dtype.kind in "iub" and y.dtype.kind in "iub") and np.any(y == 0):
              y = np.where(y == 0, 1, y).astype(dt2); dy = be.convert_to_tensor(y)
            try:
              ref = rf(x, y)
              if name == "div" and ref.dtype.kind == "f" and not np.all(np.isfinite(ref)):
                continue
              close(fn(dx, dy), ref, f"{name} {np.dtype(dt)} {np.dtype(dt2)}")
            except Exception as e:
              fails.append((name + " exc", dt, dt2))
          # tensordot / outer mixed dtypes
          a = rand((3, 4), dt); b = rand((4, 5), dt2)
          try:
            close(be.tensordot(be.convert_to_tensor(a), be.convert_to_tensor(b), 1), np.tensordot(a, b, 1), f"tensordot {np.dtype(dt)} {np.dtype(dt2)}")
          except Exception as e:
            fails.append(("tensordot exc", dt, dt2))
        # scalar ops
        x = rand((5,), dt); dx = be.convert_to_tensor(x)
        for sc in (3, 2.5, True, 1 + 2j):
          for name, fn, rf in (("mul", be.multiply, np.multiply), ("add", be.addition, np.add), ("div", be.divide, np.divide)):
            try:
              close(fn(dx, sc), rf(x, sc), f"{name} scalar {np.dtype(dt)} {sc!r}")
            except Exception as e:
              fails.append((name + " scalar exc", dt, sc))
  assert not fails, fails[:20]


def test_methods_and_error_paths_match_the_numpy_backend():
  """Values, result dtypes and exception TYPES of the backend's methods against the oracle backend (the restatement
  of numpy_backend.py) -- batched matmul and its rank check (numpy_backend.py:609-612), the broadcast multiplications
  and their ValueErrors (560-575), diagonal / diagflat / trace with offsets and axes (847-888, 684-707), sums,
  slices, reshapes, transposes, elementwise math, eye / ones / zeros, index_update (548-552), tensordot axis errors."""
  from oracle.numpy_oracle import OracleBackend  # pylint: disable=import-outside-toplevel
  rng = np.random.default_rng(1)
  ob = OracleBackend()
  fails = []

  def close(got, ref, what):
    got, ref = np.asarray(got), np.asarray(ref)
    ok = got.shape == ref.shape and got.dtype == ref.dtype
    if ok:
      tol = 1e-4 if ref.dtype in (np.float32, np.complex64) else 1e-10
      ok = np.allclose(got, ref, rtol=tol, atol=tol, equal_nan=True) if ref.dtype.kind in "fc" else np.array_equal(got, ref)
    if not ok:
      fails.append((what, str(got.dtype), str(ref.dtype), got.shape, ref.shape))

  with emulated_backend() as be_hip:
    def both(what, fn, ref_fn=None):
      """fn(backend, convert) on the oracle (or ref_fn on NumPy) and on the emulated backend: same values, or the same
      exception type."""
      try:
        r, rexc = (ref_fn() if ref_fn is not None else fn(ob, lambda x: x)), None
      except Exception as e:  # pylint: disable=broad-except
        r, rexc = None, e
      try:
        g, gexc = fn(be_hip, be_hip.convert_to_tensor), None
      except Exception as e:  # pylint: disable=broad-except
        g, gexc = None, e
      if rexc is not None or gexc is not None:
        if type(rexc) is not type(gexc):
          fails.append((what, "exceptions differ", repr(rexc)[:80], repr(gexc)[:80]))
        return
      if isinstance(r, (tuple, list)):
        for i, (a, b) in enumerate(zip(g, r)):
          close(a, b, f"{what}[{i}]")
      else:
        close(g, r, what)

    with np.errstate(all="ignore"):
      for dt in (np.float32, np.float64, np.complex64, np.int64, np.int32):
        x = (rng.standard_normal((2, 3, 4, 5)) * 3).astype(dt)
        m = (rng.standard_normal((4, 4)) * 3).astype(dt)
        v = (rng.standard_normal((5,)) * 3).astype(dt)
        w = (rng.standard_normal((2,)) * 3).astype(dt)
        xt = np.swapaxes(x, -1, -2).copy()
        both(f"matmul batch {dt}", lambda be, c: be.matmul(c(x), c(xt)))
        both(f"matmul rank1 {dt}", lambda be, c: be.matmul(c(v), c(v)))
        both(f"matmul mismatch {dt}", lambda be, c: be.matmul(c(x), c(x)))
        both(f"brm {dt}", lambda be, c: be.broadcast_right_multiplication(c(x), c(v)))
        both(f"blm {dt}", lambda be, c: be.broadcast_left_multiplication(c(w), c(x)))
        both(f"brm bad {dt}", lambda be, c: be.broadcast_right_multiplication(c(x), c(m)))
        both(f"blm bad {dt}", lambda be, c: be.broadcast_left_multiplication(c(m), c(x)))
        both(f"outer {dt}", lambda be, c: be.outer_product(c(v), c(m)))
        for off in (-1, 0, 1, 7):
          both(f"diagonal {dt} {off}", lambda be, c: be.diagonal(c(x), offset=off))
          both(f"diagonal axes {dt} {off}", lambda be, c: be.diagonal(c(x), offset=off, axis1=0, axis2=2))
          both(f"trace {dt} {off}", lambda be, c: be.trace(c(x), offset=off))
          both(f"diagflat {dt} {off}", lambda be, c: be.diagflat(c(v), k=off))
        both(f"diagonal same axes {dt}", lambda be, c: be.diagonal(c(x), axis1=1, axis2=1))
        both(f"diagonal rank1 {dt}", lambda be, c: be.diagonal(c(v)))
        both(f"trace rank1 {dt}", lambda be, c: be.trace(c(v)))
        both(f"diagflat rank2 {dt}", lambda be, c: be.diagflat(c(m)))
        both(f"sum tuple neg {dt}", lambda be, c: be.sum(c(x), axis=(-1, 0)))
        both(f"slice bad {dt}", lambda be, c: be.slice(c(x), (0, 0), (1, 1)))
        both(f"slice too big {dt}", lambda be, c: be.slice(c(x), (0, 0, 0, 3), (1, 1, 1, 9)))
        both(f"reshape -1 {dt}", lambda be, c: be.reshape(c(x), (-1, 5)))
        both(f"reshape bad {dt}", lambda be, c: be.reshape(c(x), (7, 7)))
        both(f"transpose none {dt}", lambda be, c: be.transpose(c(x)))
        both(f"transpose bad {dt}", lambda be, c: be.transpose(c(x), (0, 1)))
        both(f"power {dt}", lambda be, c: be.power(c(np.abs(m) + 1), c(np.ones_like(m) * 2)),
             ref_fn=lambda: np.power(np.abs(m) + 1, np.ones_like(m) * 2))
        both(f"power scalar {dt}", lambda be, c: be.power(c(np.abs(m) + 1), 2), ref_fn=lambda: np.power(np.abs(m) + 1, 2))
        both(f"divide tensors {dt}", lambda be, c: be.divide(c(x), c(np.abs(v) + 1)))
        both(f"shape_tensor {dt}", lambda be, c: be.shape_tensor(c(x)))
        both(f"shape_tuple {dt}", lambda be, c: be.shape_tuple(c(x)))
        both(f"shape_prod {dt}", lambda be, c: be.shape_prod(c(x)))
        both(f"sqrt {dt}", lambda be, c: be.sqrt(c(np.abs(m))))
        both(f"exp {dt}", lambda be, c: be.exp(c(m / 4)))
        both(f"log {dt}", lambda be, c: be.log(c(np.abs(m) + 1)))
        both(f"sin cos {dt}", lambda be, c: (be.sin(c(m)), be.cos(c(m))))
        both(f"norm {dt}", lambda be, c: be.norm(c(x)))
        both(f"eye {dt}", lambda be, c: be.eye(3, dtype=dt, M=5))
        both(f"ones zeros {dt}", lambda be, c: (be.ones((2, 3), dtype=dt), be.zeros((2, 3), dtype=dt)))
        both(f"sign {dt}", lambda be, c: be.sign(c(m)))
        both(f"tensordot bad axes {dt}", lambda be, c: be.tensordot(c(x), c(x), [[0], [1]]))
        x2 = np.moveaxis(x, (0, 1), (2, 3)).copy()
        both(f"tensordot int axes {dt}", lambda be, c: be.tensordot(c(x), c(x2), 2))
        if np.dtype(dt).kind == "f":
          both(f"index_update {dt}", lambda be, c: be.index_update(c(m), c(m) > 0, 7))
          both(f"expm {dt}", lambda be, c: be.expm(c(m / 8)))
          both(f"inv {dt}", lambda be, c: be.inv(c(m + 5 * np.eye(4, dtype=dt))))
        if np.dtype(dt).kind == "i":      # integer comparisons: int32 0 / 1 mask (the device form of a bool array)
          mask = np.asarray(be_hip.convert_to_tensor(m) > 0)
          np.testing.assert_array_equal(mask != 0, m > 0)
  assert not fails, fails[:20]
