"""`jax.lax` stand-in: control flow as Python loops (dev-only)."""
import enum

import numpy as _np

from _shimcore import asarray, tree_map, wrap, ShimArray


class Precision(enum.Enum):
  DEFAULT = 0
  HIGH = 1
  HIGHEST = 2


def _promote(x):
  if isinstance(x, (bool, int, float, _np.generic)):
    return asarray(x)
  return x


def while_loop(cond_fn, body_fn, init):
  state = tree_map(_promote, init)
  while bool(cond_fn(state)):
    state = tree_map(_promote, body_fn(state))
  return state


def cond(pred, true_fn, false_fn, *operands, operand=_np._NoValue):
  if operand is not _np._NoValue:
    operands = (operand,)
  return true_fn(*operands) if bool(pred) else false_fn(*operands)


def psum(x, axis_name):
  del axis_name
  return x


def axis_index(axis_name):
  del axis_name
  return 0


def all_gather(x, axis_name):
  return tree_map(lambda a: wrap(_np.asarray(a)[None]), x)


def with_sharding_constraint(x, spec):
  del spec
  return x
