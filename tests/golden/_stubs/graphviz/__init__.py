"""Inert stand-in for graphviz (only used by tensornetwork.visualization)."""
class Graph:  # pylint: disable=too-few-public-methods
  def __init__(self, *args, **kwargs):
    raise RuntimeError("graphviz is not installed")
