"""Inert stand-in for tensorflow (not installed): dtype names for the reference tests' module-level
tables and ``tf.compat.v1.enable_v2_behavior``.  Anything else raises (reported as "needs tensorflow")."""


class _DType:  # pylint: disable=too-few-public-methods
  def __init__(self, name):
    self.name = name

  def __repr__(self):
    return f"tf.{self.name}"


for _n in ("float16", "float32", "float64", "complex64", "complex128", "int8", "int16", "int32", "int64",
           "uint8", "uint16", "uint32", "uint64", "bool"):
  globals()[_n] = _DType(_n)


class _V1:  # pylint: disable=too-few-public-methods
  @staticmethod
  def enable_v2_behavior():
    return None


class _Compat:  # pylint: disable=too-few-public-methods
  v1 = _V1()


compat = _Compat()


class Tensor:  # pylint: disable=too-few-public-methods
  pass


def __getattr__(name):
  raise AttributeError(f"tensorflow stub has no attribute {name!r} (tensorflow is not installed)")
