"""Tiny pytree helpers (tuple / list / dict / NamedTuple / None / registered
dataclasses) — the subset of jax.tree the reference's optimizer surface uses
(jax.tree.flatten / flatten_up_to / unflatten / map, DS:3639-3656)."""
from __future__ import annotations

import dataclasses
from typing import Any, Callable, List, Tuple


class _LeafMarker:

  def __repr__(self):
    return "*"


LEAF = _LeafMarker()
_REGISTERED = {}


def register_dataclass(cls, dynamic_fields, static_fields=()):
  """Marks a dataclass as a pytree node whose children are `dynamic_fields`."""
  _REGISTERED[cls] = (tuple(dynamic_fields), tuple(static_fields))
  return cls


def _node(x):
  """Returns (children, rebuild) for containers, None for leaves."""
  if x is None:
    return [], lambda ch: None
  if isinstance(x, tuple) and hasattr(x, "_fields"):  # NamedTuple
    t = type(x)
    return list(x), lambda ch: t(*ch)
  if isinstance(x, tuple):
    return list(x), lambda ch: tuple(ch)
  if isinstance(x, list):
    return list(x), lambda ch: list(ch)
  if isinstance(x, dict):
    keys = sorted(x)
    return [x[k] for k in keys], lambda ch: dict(zip(keys, ch))
  reg = _REGISTERED.get(type(x))
  if reg is not None:
    dyn, static = reg
    t = type(x)
    fixed = {k: getattr(x, k) for k in static}

    def rebuild(ch):
      kw = dict(zip(dyn, ch))
      kw.update(fixed)
      return t(**kw)

    return [getattr(x, k) for k in dyn], rebuild
  return None


# The recursive walkers are module-level functions on purpose: a nested `def rec` that calls
# itself forms a reference cycle (function -> closure cell -> function) that also holds the
# accumulated leaf list, i.e. every tensor of the flattened optimizer state, until the cyclic
# garbage collector runs — gigabytes of device memory per update() pinned behind the GC.
def _flatten_up_to(sk, t, out):
  if sk is LEAF:
    out.append(t)
    return
  a = _node(sk)[0]
  nt = _node(t)
  if nt is None or len(nt[0]) != len(a):
    raise ValueError(f"tree structure mismatch: {sk!r} vs {t!r}")
  for s, c in zip(a, nt[0]):
    _flatten_up_to(s, c, out)


def _unflatten(sk, it):
  if sk is LEAF:
    return next(it)
  ch, rebuild = _node(sk)
  return rebuild([_unflatten(c, it) for c in ch])


def _flatten(t, leaves, is_leaf):
  if is_leaf is not None and is_leaf(t):
    leaves.append(t)
    return LEAF
  nd = _node(t)
  if nd is None:
    leaves.append(t)
    return LEAF
  ch, rebuild = nd
  return rebuild([_flatten(c, leaves, is_leaf) for c in ch])


class TreeDef:
  """Structure of a pytree: the tree itself with every leaf replaced by LEAF."""

  def __init__(self, skeleton):
    self.skeleton = skeleton

  def flatten_up_to(self, tree) -> List[Any]:
    """Flattens `tree` only as deep as this structure goes (DS:3640-3641)."""
    out = []
    _flatten_up_to(self.skeleton, tree, out)
    return out

  def unflatten(self, leaves):
    return _unflatten(self.skeleton, iter(leaves))


def tree_flatten(tree, is_leaf: Callable[[Any], bool] = None) -> Tuple[List[Any], TreeDef]:
  leaves = []
  skeleton = _flatten(tree, leaves, is_leaf)
  return leaves, TreeDef(skeleton)


def tree_unflatten(treedef: TreeDef, leaves):
  return treedef.unflatten(list(leaves))


def tree_map(f, tree, *rest, is_leaf=None):
  leaves, treedef = tree_flatten(tree, is_leaf=is_leaf)
  others = [treedef.flatten_up_to(r) for r in rest]
  return treedef.unflatten([f(*xs) for xs in zip(leaves, *others)])


def tree_leaves(tree):
  return tree_flatten(tree)[0]
