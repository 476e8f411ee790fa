"""Dev-only NumPy stand-in for the subset of `jax` the reference imports.

See tools/_refshim/_shimcore.py for purpose and fidelity rules.  Not product
code; used only by tools/gen_golden.py in the build container.
"""
import types as _types

from _shimcore import (ShimArray, asarray as _asarray, tree_flatten,
                       tree_unflatten, tree_map, tree_all, wrap as _wrap)
import numpy as _np

from . import numpy  # noqa: F401  (jax.numpy)
from . import lax  # noqa: F401
from . import experimental  # noqa: F401

Array = ShimArray

tree = _types.SimpleNamespace(
    map=tree_map, flatten=tree_flatten, unflatten=tree_unflatten)
tree_util = _types.SimpleNamespace(
    tree_all=tree_all, tree_map=tree_map, tree_flatten=tree_flatten,
    tree_unflatten=tree_unflatten)


class _PartitionSpec(tuple):

  def __new__(cls, *args):
    return super().__new__(cls, args)


sharding = _types.SimpleNamespace(PartitionSpec=_PartitionSpec)


def _stack_tree(items):
  first = items[0]
  leaves0, treedef = tree_flatten(first)
  cols = [treedef.flatten_up_to(it) for it in items]
  stacked = [
      _wrap(_np.stack([_np.asarray(c[i]) for c in cols]))
      for i in range(len(leaves0))
  ]
  return treedef.unflatten(stacked)


def vmap(fn):
  """Python-loop vmap over the leading axis of every non-None argument."""

  def mapped(*args, **kwargs):
    n = None
    for a in list(args) + list(kwargs.values()):
      if a is not None:
        n = len(a)
        break
    outs = []
    for i in range(n):
      a_i = [None if a is None else a[i] for a in args]
      k_i = {k: (None if v is None else v[i]) for k, v in kwargs.items()}
      outs.append(fn(*a_i, **k_i))
    return _stack_tree(outs)

  return mapped


def jit(fn, **_):
  return fn
