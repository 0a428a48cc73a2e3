"""`jit` with the reference's signature (backends/decorators.py:26-98): resolves the backend -- fixed, or taken
from one of the call's arguments -- and hands `fun` to `backend.jit`.  `HipBackend.jit` returns `fun`
unchanged: launches are asynchronous already, and launch-bound sequences are recorded explicitly with
`HipBackend.capture` (hipGraph) where the caller knows that nothing inside reads back to the host."""
import functools
from typing import Callable, Iterable, Optional, Union


def _resolve_backend(backend):
  from tensornetwork_amd.ncon import _resolve_backend as _rb  # pylint: disable=import-outside-toplevel
  return _rb(backend)


def jit(fun: Callable, backend=None, backend_argnum: Optional[int] = None,
        static_argnums: Union[int, Iterable[int]] = (), device=None, xla_backend: Optional[str] = None) -> Callable:
  if isinstance(static_argnums, int):
    static_argnums = (static_argnums,)
  if backend_argnum is not None:
    if backend is not None:
      raise ValueError("backend must be None if backend_argnum is specified.")
    static = tuple(static_argnums) + (backend_argnum,)

    @functools.wraps(fun)
    def by_argument(*args, **kwargs):
      try:
        be = _resolve_backend(args[backend_argnum])
        if not hasattr(be, "jit"):
          raise ValueError("not a backend")
      except (ValueError, TypeError) as err:
        raise ValueError(f"backend_argnum={backend_argnum} was specifiedbut the corresponding argument "
                         f"{args[backend_argnum]}did not specify a backend.") from err
      return be.jit(fun, static_argnums=static, device=device, backend=xla_backend)(*args, **kwargs)

    return by_argument

  be = _resolve_backend(backend)

  @functools.wraps(fun)
  def fixed(*args, **kwargs):
    return be.jit(fun, static_argnums=tuple(static_argnums), device=device, backend=xla_backend)(*args, **kwargs)

  return fixed
