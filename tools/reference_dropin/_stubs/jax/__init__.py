"""Inert stand-in for jax (not installed): only lets the reference's test MODULES import.

Several reference test files do ``import jax`` / ``from jax import config`` at module level and
build dtype tables from ``jax.numpy`` before any test runs.  Tests parametrised for the "jax"
backend are never selected by the drop-in harness (``backend`` fixture = "hip"); anything that
really calls into jax fails with this module's AttributeError / RuntimeError and is reported as
"needs jax" in the triage table."""
import numpy as _np

from jax import numpy  # noqa: F401  pylint: disable=import-self


class _Config:
  def update(self, *args, **kwargs):  # pylint: disable=unused-argument
    return None


config = _Config()


class _Namespace:  # pylint: disable=too-few-public-methods
  """attribute bag: ``JaxBackend.__init__`` (jax_backend.py:42-51) touches jax.scipy and
  jax.lax.Precision.DEFAULT when tests/ncon_interface_test.py builds its fixture list at import."""


scipy = _Namespace()
lax = _Namespace()
lax.Precision = _Namespace()
lax.Precision.DEFAULT = None


def jit(fun, *args, **kwargs):  # pylint: disable=unused-argument
  raise RuntimeError("jax is not installed (inert stub)")


def __getattr__(name):
  raise AttributeError(f"jax stub has no attribute {name!r} (jax is not installed)")
