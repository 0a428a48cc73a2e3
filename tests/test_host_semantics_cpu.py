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
