"""Core of the dev-only NumPy stand-in for jax/flax/chex/optax.

PURPOSE: lets ``tools/gen_golden.py`` import the *reference's own source* from
``/root/reference`` in the build container (where jax is not installed) so that
its control flow (retry loops, padding masks, bookkeeping) produces golden
vectors for ``tests/golden``.  This is test tooling written for this repo; it
is not part of the product, never travels to the GPU box as a dependency of the
tests, and contains no reference code.

Fidelity rule (SURVEY.md §8c "shim fidelity pitfall"): JAX with x64 disabled
never produces float64/int64.  NumPy does (int32/float32 -> float64, stacking
Python ints -> int64).  ``ShimArray`` demotes the result of every ufunc and
every ``__array_function__`` call back to float32/int32/complex64, which for
+,-,*,/ and sqrt is bit-identical to having computed in float32 directly
(double rounding from binary64 to binary32 is innocuous for these ops).
"""
from __future__ import annotations

import numpy as np

_DEMOTE = {
    np.dtype(np.float64): np.dtype(np.float32),
    np.dtype(np.int64): np.dtype(np.int32),
    np.dtype(np.uint64): np.dtype(np.uint32),
    np.dtype(np.complex128): np.dtype(np.complex64),
}


def canon_dtype(dt):
  if dt is None:
    return None
  dt = np.dtype(dt)
  return _DEMOTE.get(dt, dt)


class _At:
  """``x.at[idx].set(v)`` functional update."""

  def __init__(self, arr):
    self._arr = arr

  def __getitem__(self, idx):
    return _AtIdx(self._arr, idx)


class _AtIdx:

  def __init__(self, arr, idx):
    self._arr, self._idx = arr, idx

  def set(self, v):
    out = np.array(self._arr, copy=True).view(ShimArray)
    np.ndarray.__setitem__(out, self._idx, _unwrap(v))
    return out

  def add(self, v):
    out = np.array(self._arr, copy=True).view(ShimArray)
    out[self._idx] = _unwrap(out[self._idx]) + _unwrap(v)
    return out


def _unwrap(x):
  if isinstance(x, ShimArray):
    return x.view(np.ndarray)
  if isinstance(x, (list, tuple)):
    return type(x)(_unwrap(v) for v in x)
  if isinstance(x, dict):
    return {k: _unwrap(v) for k, v in x.items()}
  return x


def wrap(x):
  """ndarray / numpy scalar / python scalar -> demoted ShimArray."""
  if isinstance(x, ShimArray):
    dt = canon_dtype(x.dtype)
    return x if dt == x.dtype else x.astype(dt)
  if isinstance(x, (np.ndarray, np.generic)):
    a = np.asarray(x)
    dt = canon_dtype(a.dtype)
    if dt != a.dtype:
      a = a.astype(dt)
    return a.view(ShimArray)
  if isinstance(x, tuple):
    return tuple(wrap(v) for v in x)
  if isinstance(x, list):
    return [wrap(v) for v in x]
  return x


class ShimArray(np.ndarray):
  """ndarray that never holds 64-bit results (mimics jax_enable_x64=False)."""

  __array_priority__ = 100.0

  def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
    ins = tuple(_unwrap(i) for i in inputs)
    if out is not None:
      outs = tuple(_unwrap(o) for o in out)
      # In-place forms (a *= b): compute out-of-place, demote, copy in.
      res = getattr(ufunc, method)(*ins, **kwargs)
      if isinstance(res, tuple):
        for o, r in zip(outs, res):
          o[...] = r
        return tuple(wrap(o) for o in out)
      outs[0][...] = res
      return out[0]
    res = getattr(ufunc, method)(*ins, **kwargs)
    return wrap(res)

  def __array_function__(self, func, types, args, kwargs):
    res = func(*_unwrap(args), **_unwrap(kwargs))
    return wrap(res)

  @property
  def at(self):
    return _At(self)

  def astype(self, dtype, *a, **k):
    return np.ndarray.astype(self.view(np.ndarray), canon_dtype(dtype), *a,
                             **k).view(ShimArray)

  def __getitem__(self, idx):
    r = np.ndarray.__getitem__(self.view(np.ndarray), _unwrap(idx))
    return wrap(r)

  def __bool__(self):
    return bool(self.view(np.ndarray))

  def __int__(self):
    return int(self.view(np.ndarray))

  def __float__(self):
    return float(self.view(np.ndarray))

  def __index__(self):
    return int(self.view(np.ndarray))

  def __hash__(self):
    return id(self)

  def __iter__(self):
    for i in range(self.shape[0]):
      yield self[i]

  def dot(self, other, precision=None):
    del precision
    return wrap(np.dot(_unwrap(self), _unwrap(other)))

  @property
  def T(self):
    return self.view(np.ndarray).T.view(ShimArray)

  # JAX arrays are immutable: ``x op= y`` rebinds, it never mutates shared
  # storage (mat_power's ``i //= 2`` at DS:673 would otherwise destroy ``p``).
  def __iadd__(self, o):
    return self + o

  def __isub__(self, o):
    return self - o

  def __imul__(self, o):
    return self * o

  def __itruediv__(self, o):
    return self / o

  def __ifloordiv__(self, o):
    return self // o

  def __imod__(self, o):
    return self % o

  def __ipow__(self, o):
    return self ** o

  def __iand__(self, o):
    return self & o

  def __ior__(self, o):
    return self | o

  def __pow__(self, other):
    return wrap(np.power(_unwrap(self), _unwrap(other)))

  def __rpow__(self, other):
    return wrap(np.power(_unwrap(other), _unwrap(self)))


def asarray(x, dtype=None):
  dt = canon_dtype(dtype)
  if isinstance(x, (bool, np.bool_)) and dt is None:
    return np.asarray(x, dtype=np.bool_).view(ShimArray)
  if isinstance(x, int) and dt is None:
    return np.asarray(x, dtype=np.int32).view(ShimArray)
  if isinstance(x, float) and dt is None:
    return np.asarray(x, dtype=np.float32).view(ShimArray)
  a = np.array(_unwrap(x), dtype=dt)
  return wrap(a)


# --------------------------------------------------------------------------
# pytrees
# --------------------------------------------------------------------------
class _Leaf:
  pass


LEAF = _Leaf()


def _is_namedtuple(x):
  return isinstance(x, tuple) and hasattr(x, "_fields")


def _is_struct(x):
  return hasattr(type(x), "_shim_struct_fields")


def _children(x):
  """Returns (children, rebuild) or None if x is a leaf."""
  if x is None:
    return [], lambda ch: None
  if _is_namedtuple(x):
    t = type(x)
    return list(x), lambda ch: t(*ch)
  if isinstance(x, tuple):
    return list(x), lambda ch: tuple(ch)
  if isinstance(x, list):
    return list(x), lambda ch: list(ch)
  if isinstance(x, dict):
    keys = sorted(x.keys())
    return [x[k] for k in keys], lambda ch: dict(zip(keys, ch))
  if _is_struct(x):
    t = type(x)
    dyn, static = t._shim_struct_fields
    svals = {k: getattr(x, k) for k in static}

    def rebuild(ch):
      kw = dict(zip(dyn, ch))
      kw.update(svals)
      return t(**kw)

    return [getattr(x, k) for k in dyn], rebuild
  return None


class TreeDef:

  def __init__(self, skeleton):
    self.skeleton = skeleton  # nested structure with LEAF markers

  def flatten_up_to(self, tree):
    out = []

    def rec(sk, t):
      if sk is LEAF:
        out.append(t)
        return
      ch_sk = _children(sk)[0]
      ch_t = _children(t)
      assert ch_t is not None, (sk, t)
      ch_t = ch_t[0]
      assert len(ch_sk) == len(ch_t), (sk, t)
      for a, b in zip(ch_sk, ch_t):
        rec(a, b)

    rec(self.skeleton, tree)
    return out

  def unflatten(self, leaves):
    it = iter(leaves)

    def rec(sk):
      if sk is LEAF:
        return next(it)
      ch, rebuild = _children(sk)
      return rebuild([rec(c) for c in ch])

    return rec(self.skeleton)


def tree_flatten(tree, is_leaf=None):
  leaves = []

  def rec(t):
    if is_leaf is not None and is_leaf(t):
      leaves.append(t)
      return LEAF
    c = _children(t)
    if c is None:
      leaves.append(t)
      return LEAF
    ch, rebuild = c
    return rebuild([rec(x) for x in ch])

  sk = rec(tree)
  return leaves, TreeDef(sk)


def tree_unflatten(treedef, leaves):
  return treedef.unflatten(list(leaves))


def tree_map(f, tree, *rest, is_leaf=None):
  leaves, treedef = tree_flatten(tree, is_leaf=is_leaf)
  others = [treedef.flatten_up_to(r) for r in rest]
  return treedef.unflatten([f(*xs) for xs in zip(leaves, *others)])


def tree_all(tree):
  return all(bool(x) for x in tree_flatten(tree)[0])
