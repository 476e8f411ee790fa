"""Shape / block bookkeeping of the Shampoo hot path (bit-exact integer logic).

Mirrors, by name and behaviour, the reference's helpers in
precondition/distributed_shampoo.py: merge_small_dims DS:1293-1321,
pad_square_matrix DS:1324-1350, pad_vector DS:1353-1369, BlockPartitioner
DS:1387-1437, Preconditioner DS:1508-1708, batch/unbatch DS:1827-1846,
_precond_dim / _should_compress DS:520-537.  Pure Python ints + torch views:
partitioning a tensor never copies (blocks are strided views that the HIP
statistics kernel reads in place).
"""
from __future__ import annotations

import itertools
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from .state import PreconditionerType


def _precond_dim(compression_rank: int, dim: int) -> int:
  """Stored width of a (possibly rank-compressed) preconditioner (DS:520-532)."""
  if not compression_rank:
    return dim
  packed = abs(compression_rank) + 2
  return dim if packed >= dim else packed


def _should_compress(compression_rank: int, dim) -> bool:
  """DS:535-537."""
  return compression_rank != 0 and abs(compression_rank) + 2 < dim


def merge_small_dims(shape_to_merge: Sequence[int], max_dim: int) -> List[int]:
  """Greedy left-to-right merge of neighbouring dims while the product stays
  <= max_dim, e.g. [1,2,512,1,2048,1,3,4] -> [1024,2048,12] for max_dim=1024."""
  dims = [int(d) for d in shape_to_merge]
  if dims and all(d == 1 for d in dims):
    return [1]
  merged, running = [], 1
  for d in dims:
    if running * d <= max_dim:
      running *= d
      continue
    if running > 1:
      merged.append(running)
    running = d
  if running > 1:
    merged.append(running)
  return merged


def pad_square_matrix(mat: torch.Tensor, max_size: int) -> torch.Tensor:
  """[[M, 0], [0, I]] of side max_size (DS:1324-1350), same errors."""
  rows, cols = mat.shape
  if rows != cols:
    raise ValueError("Must have rows == cols, instead got "
                     f"rows={rows}, cols={cols}")
  if cols > max_size:
    raise ValueError("Must have cols <= max_size. Instead got "
                     f"cols={cols}, max_size={max_size}.")
  if rows == max_size:
    return mat
  out = torch.zeros((max_size, max_size), dtype=mat.dtype, device=mat.device)
  out[:rows, :rows] = mat
  idx = torch.arange(rows, max_size, device=mat.device)
  out[idx, idx] = 1
  return out


def pad_vector(vec: torch.Tensor, max_size: int) -> torch.Tensor:
  """[V, 0] (DS:1353-1369)."""
  size = vec.shape[0]
  assert size <= max_size
  if size == max_size:
    return vec
  out = torch.zeros((max_size,), dtype=vec.dtype, device=vec.device)
  out[:size] = vec
  return out


def batch(x: Sequence[torch.Tensor], num_devices: int) -> torch.Tensor:
  """Stacks a list into [num_devices, len/num_devices, ...]: rank r owns the
  contiguous chunk r*b .. (r+1)*b-1 (DS:1827-1831)."""
  n = len(x)
  b = int(n / num_devices)
  return torch.stack([torch.stack(list(x[i:i + b])) for i in range(0, n, b)])


def unbatch(batched_values: torch.Tensor) -> List[torch.Tensor]:
  """Inverse of batch(): rank-major, then position (DS:1834-1846)."""
  b1, b2 = batched_values.shape[0], batched_values.shape[1]
  out = []
  for r in range(b1):
    for j in range(b2):
      out.append(batched_values[r, j])
  return out


def owner_of_statistic(index: int, num_padded: int, num_devices: int) -> int:
  """Rank that computes statistic `index` under the reference's batch() order."""
  return index // (num_padded // num_devices)


class BlockPartitioner:
  """Splits a tensor into blocks of at most block_size per dim; the last block
  of a dim is ragged.  Accepts a tensor or a bare shape."""

  def __init__(self, param, block_size: int):
    shape = tuple(param.shape) if hasattr(param, "shape") else tuple(param)
    self._shape = shape
    self._splits = []        # (axis, cut points) for every split axis
    self._split_sizes = []   # per axis: int32 array of block extents
    for axis, d in enumerate(shape):
      d = int(d)
      if 0 < block_size < d:
        nsplit = (d - 1) // block_size  # d-1: never produce an empty tail block
        cuts = (np.arange(nsplit, dtype=np.int32) + 1) * block_size
        sizes = np.full(nsplit + 1, block_size, dtype=np.int32)
        sizes[-1] = d - cuts[-1]
        self._splits.append((axis, cuts))
        self._split_sizes.append(sizes)
      else:
        self._split_sizes.append(np.array([d], dtype=np.int32))

  def split_sizes(self):
    return self._split_sizes

  def num_blocks(self) -> int:
    return int(np.prod([len(s) for s in self._split_sizes], dtype=np.int64)) \
        if self._split_sizes else 1

  def _ranges(self):
    if getattr(self, "_ranges_cache", None) is not None:
      return self._ranges_cache
    per_axis = []
    for sizes in self._split_sizes:
      offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
      per_axis.append([(int(o), int(s)) for o, s in zip(offs, sizes)])
    self._ranges_cache = per_axis
    return per_axis

  def partition(self, tensor: torch.Tensor) -> List[torch.Tensor]:
    """Blocks in row-major block order (first axis slowest), as strided views."""
    assert tuple(tensor.shape) == self._shape, (tensor.shape, self._shape)
    if not self._shape:
      return [tensor]
    out = []
    for combo in itertools.product(*self._ranges()):
      view = tensor
      for axis, (off, size) in enumerate(combo):
        if size != self._shape[axis]:
          view = view.narrow(axis, off, size)
      out.append(view)
    return out

  def merge_partitions(self, partitions: Sequence[torch.Tensor]) -> torch.Tensor:
    """Inverse of partition()."""
    if not self._shape:
      assert len(partitions) == 1
      return partitions[0]
    assert len(partitions) == self.num_blocks()
    ref = partitions[0]
    out = torch.empty(self._shape, dtype=ref.dtype, device=ref.device)
    for combo, part in zip(itertools.product(*self._ranges()), partitions):
      view = out
      for axis, (off, size) in enumerate(combo):
        view = view.narrow(axis, off, size)
      view.copy_(part)
    return out


class Preconditioner:
  """Per-parameter bookkeeping: which axes get a factor, their shapes, the root
  exponent, and the statistic index order (block-major, axis-minor)."""

  def __init__(self, param, block_size, merge_small_dims_block_size,
               best_effort_shape_interpretation,
               preconditioner_type=PreconditionerType.ALL, compression_rank=0):
    self._original_shape = tuple(param.shape)
    self._transformed_shape = tuple(param.shape)
    if best_effort_shape_interpretation:
      self._transformed_shape = tuple(
          merge_small_dims(self._original_shape, merge_small_dims_block_size))
    self._partitioner = BlockPartitioner(self._transformed_shape, block_size)
    self._preconditioner_type = preconditioner_type
    self._compression_rank = compression_rank

  # -- structure ---------------------------------------------------------------
  def should_precondition_dims(self) -> List[bool]:
    rank = len(self._partitioner.split_sizes())
    if self._preconditioner_type == PreconditionerType.ALL or rank <= 1:
      return [True] * rank
    if self._preconditioner_type == PreconditionerType.INPUT:
      return [True] * (rank - 1) + [False]
    if self._preconditioner_type == PreconditionerType.OUTPUT:
      return [False] * (rank - 1) + [True]
    raise ValueError(self._preconditioner_type)

  def _preconditioner_shape(self, dim):
    dim = int(dim)
    if self._compression_rank:
      return [dim, _precond_dim(self._compression_rank, dim)]
    return [dim, dim]

  def shapes_for_preconditioners(self) -> List[List[int]]:
    split_sizes = self._partitioner.split_sizes()
    rank = len(split_sizes)
    shapes = []
    for block in itertools.product(*split_sizes):
      if self._preconditioner_type == PreconditionerType.ALL or rank <= 1:
        dims = block
      elif self._preconditioner_type == PreconditionerType.INPUT:
        dims = block[:-1]
      else:
        dims = block[-1:]
      shapes.extend(self._preconditioner_shape(d) for d in dims)
    return shapes

  def contraction_lengths(self) -> List[int]:
    """Per statistic (same order as shapes_for_preconditioners): the number of columns k of the block
    matricised along that axis -- one update adds a Gram matrix of rank <= k to the statistic."""
    split_sizes = self._partitioner.split_sizes()
    rank = len(split_sizes)
    out = []
    for block in itertools.product(*split_sizes):
      total = 1
      for d in block:
        total *= int(d)
      if self._preconditioner_type == PreconditionerType.ALL or rank <= 1:
        dims = block
      elif self._preconditioner_type == PreconditionerType.INPUT:
        dims = block[:-1]
      else:
        dims = block[-1:]
      out.extend(total // max(int(d), 1) for d in dims)
    return out

  def exponent_for_preconditioner(self) -> int:
    """p of M^{-1/p}: twice the number of preconditioned dims (DS:1639-1643)."""
    return 2 * sum(self.should_precondition_dims())

  def partitioned_blocks(self, tensor: torch.Tensor) -> List[torch.Tensor]:
    return self._partitioner.partition(tensor.reshape(self._transformed_shape))

  # -- statistics ---------------------------------------------------------------
  def statistics_update_items(self, stats, grad, new_stats):
    """(block, axis, stat_in, stat_out) tuples in statistic-index order, for one
    grouped launch of the HIP Gram kernel over a whole parameter tree."""
    dims = [i for i, p in enumerate(self.should_precondition_dims()) if p]
    items, index = [], 0
    for g in self.partitioned_blocks(grad):
      for axis in dims:
        items.append((g, axis, stats[index], new_stats[index]))
        index += 1
    return items

  def updated_statistics_from_grad(self, stats, grad, w1, w2, to_float=None,
                                   from_float=None, precision=None,
                                   frequent_directions=False):
    """DS:1540-1591 (dense branch): new_stats[i] = w1*stats[i] + w2*Gram_i."""
    from . import kernels  # HIP path; raises without the library / a GPU
    del precision
    to_float = to_float or (lambda x: x)
    from_float = from_float or (lambda x: x)
    if frequent_directions:
      # DS:1585-1588: blocks whose axis is compressed take frequent_directions_update
      # (a square factor R with R R^T = Gram), the others the dense weighted update.
      from . import low_rank
      dims = [i for i, p in enumerate(self.should_precondition_dims()) if p]
      new_stats, index = [], 0
      for g in self.partitioned_blocks(grad):
        for axis in dims:
          old = to_float(stats[index]).contiguous()
          if _should_compress(self._compression_rank, g.shape[axis]):
            new = low_rank.frequent_directions_update(old, g, axis, w1, w2)
          else:
            new = kernels.gram_weighted_update(old, g, axis, w1, w2)
          new_stats.append(from_float(new))
          index += 1
      return new_stats
    olds = [to_float(s).contiguous() for s in stats]
    news = [torch.empty_like(s) for s in olds]
    kernels.stats_update_grouped(self.statistics_update_items(olds, grad, news),
                                 w1, w2)
    return [from_float(s) for s in news]

  # -- application (the reference's preconditioned_grad, DS:1645-1708) ------------
  def _preconds_for_grad(self, preconditioners, rank, start, end):
    sel = list(preconditioners[start:end])
    if self._preconditioner_type == PreconditionerType.INPUT:
      sel = sel + [None]
    elif self._preconditioner_type == PreconditionerType.OUTPUT:
      sel = [None] * (rank - 1) + sel
    assert len(sel) == rank
    return sel

  def preconditioned_grad(self, grad, preconditioners, tensordot_fn=None,
                          matmul_fn=None):
    """Per block: contract every preconditioned axis with its factor, keeping
    the axes in their original cyclic order."""
    if tensordot_fn is None:
      from . import kernels
      tensordot_fn = kernels.tensordot_axis0
    if matmul_fn is None:
      from . import kernels
      matmul_fn = kernels.matmul
    self._matmul_fn = matmul_fn
    should = self.should_precondition_dims()
    num = sum(should)
    out_blocks = []
    for i, g in enumerate(self.partitioned_blocks(grad)):
      pcs = self._preconds_for_grad(preconditioners, len(should), i * num,
                                    (i + 1) * num)
      out_blocks.append(self._precondition_block(g, should, pcs, tensordot_fn))
    merged = self._partitioner.merge_partitions(out_blocks)
    return merged.reshape(self._original_shape)

  def _precondition_block(self, g, should_precondition_dim, preconditioners,
                          tensordot_fn):
    for j, should in enumerate(should_precondition_dim):
      rank = g.dim()
      if not should:
        g = g.permute(*range(1, rank), 0)
        continue
      pc = preconditioners[j]
      dim, application_dim = pc.shape
      if application_dim != dim:
        # low rank + constant (DS:1689-1705): g' = const * (g - B E^T) + (B * eigs) E^T
        # with B = tensordot(g, E, [[0],[0]]); result has the contracted axis last.
        r = abs(self._compression_rank)
        eigvecs = pc[:, :r].contiguous()
        eigvals, const, skip = pc[:r, -2], pc[0, -1], pc[-1, -2].bool()
        basis = tensordot_fn(g, eigvecs)                      # [..., r]
        b2 = basis.reshape(-1, r)
        lowrank = self._matmul_fn(b2, eigvecs, transb=True).reshape(
            tuple(basis.shape[:-1]) + (dim,))
        g = g.permute(*range(1, rank), 0)
        scaled = self._matmul_fn((b2 * eigvals).contiguous(), eigvecs, transb=True
                                 ).reshape(lowrank.shape)
        new_g = const * (g - lowrank) + scaled
        g = torch.where(skip, g, new_g)
        continue
      # tensordot(g, P, axes=[[0],[0]]): leading axis contracted, result last.
      g = tensordot_fn(g, pc)
    return g
