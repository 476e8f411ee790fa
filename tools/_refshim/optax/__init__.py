"""`optax` stand-in (dev-only): the two names the reference uses."""
from typing import Any, Callable, NamedTuple


class MaskedNode(NamedTuple):
  """Empty container: a pytree node with no children."""


class GradientTransformation(NamedTuple):
  init: Callable[..., Any]
  update: Callable[..., Any]
