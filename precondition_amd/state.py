"""State containers and enums of the optimizer surface, with the reference's
names and field order (precondition/distributed_shampoo.py: TrainingMetrics
DS:338-363, ParameterStats DS:367-375, ShampooState DS:488-490, GraftingType
DS:499-506, PreconditionerType DS:509-517; optax.GradientTransformation /
optax.MaskedNode; quantization_utils.QuantizedValue :26-113 in its float32
pass-through mode)."""
from __future__ import annotations

import dataclasses
import enum
from typing import Any, Callable, List, NamedTuple, Optional, Union

import torch

from . import pytree


class GradientTransformation(NamedTuple):
  """Same shape as optax.GradientTransformation: (init, update)."""
  init: Callable[..., Any]
  update: Callable[..., Any]


class MaskedNode(NamedTuple):
  """Empty pytree node (optax.MaskedNode)."""


class GraftingType(enum.IntEnum):
  NONE = 0
  SGD = 1
  ADAGRAD = 2
  RMSPROP = 3
  RMSPROP_NORMALIZED = 4
  SQRT_N = 5
  ADAGRAD_NORMALIZED = 6


class PreconditionerType(enum.IntEnum):
  ALL = 1     # a factor for every dim
  INPUT = 2   # every dim but the last
  OUTPUT = 3  # only the last dim


def _zero():
  return torch.zeros((), dtype=torch.float32)


@dataclasses.dataclass(frozen=True)
class LOBPCGDiagnostics:
  """DS:149-194; filled by the top-k deflated root (deflation.py)."""
  lobpcg_iters: Any = dataclasses.field(default_factory=_zero)
  max_consistency_error: Any = dataclasses.field(default_factory=_zero)
  avg_consistency_error: Any = dataclasses.field(default_factory=_zero)
  avg_orthogonality_error: Any = dataclasses.field(default_factory=_zero)
  max_eigenvalue: Any = dataclasses.field(default_factory=_zero)
  min_eigenvalue: Any = dataclasses.field(default_factory=_zero)
  num_topk_eigenvectors: Any = dataclasses.field(default_factory=_zero)

  def replace(self, **kw):
    return dataclasses.replace(self, **kw)


@dataclasses.dataclass(frozen=True)
class InversePthRootDiagnostics:
  """DS:109-146; the reference (and this build) fill it only in the deflated branch."""
  max_diag_error: Any = dataclasses.field(default_factory=_zero)
  avg_diag_error: Any = dataclasses.field(default_factory=_zero)
  max_off_diag_error: Any = dataclasses.field(default_factory=_zero)
  avg_off_diag_error: Any = dataclasses.field(default_factory=_zero)
  p: Any = dataclasses.field(default_factory=_zero)

  def replace(self, **kw):
    return dataclasses.replace(self, **kw)


_FD_FIELDS = ("size_max_size", "size_rank", "size_padding_start", "rho", "tail", "eig_sparsity",
              "eig_max", "eig_min", "new_grad_abs_max", "new_grad_sparsity",
              "new_grad_col_sparsity", "ggt_eig_max", "ggt_intrinsic_dimension",
              "max_ortho_err", "num_neg_eigs", "num_zero_initial_eigs", "num_unsafe_norms",
              "num_has_padding", "square_frob", "heuristic_frob", "entrywise_err",
              "total_frob")


@dataclasses.dataclass(frozen=True)
class FDDiagnostics:
  """Diagnostics of a Frequent-Directions update (DS:197-335), same fields in the same
  order; filled by low_rank._fd_update_root when generate_fd_metrics is set."""
  size_max_size: Any = dataclasses.field(default_factory=_zero)
  size_rank: Any = dataclasses.field(default_factory=_zero)
  size_padding_start: Any = dataclasses.field(default_factory=_zero)
  rho: Any = dataclasses.field(default_factory=_zero)
  tail: Any = dataclasses.field(default_factory=_zero)
  eig_sparsity: Any = dataclasses.field(default_factory=_zero)
  eig_max: Any = dataclasses.field(default_factory=_zero)
  eig_min: Any = dataclasses.field(default_factory=_zero)
  new_grad_abs_max: Any = dataclasses.field(default_factory=_zero)
  new_grad_sparsity: Any = dataclasses.field(default_factory=_zero)
  new_grad_col_sparsity: Any = dataclasses.field(default_factory=_zero)
  ggt_eig_max: Any = dataclasses.field(default_factory=_zero)
  ggt_intrinsic_dimension: Any = dataclasses.field(default_factory=_zero)
  max_ortho_err: Any = dataclasses.field(default_factory=_zero)
  num_neg_eigs: Any = dataclasses.field(default_factory=_zero)
  num_zero_initial_eigs: Any = dataclasses.field(default_factory=_zero)
  num_unsafe_norms: Any = dataclasses.field(default_factory=_zero)
  num_has_padding: Any = dataclasses.field(default_factory=_zero)
  square_frob: Any = dataclasses.field(default_factory=_zero)
  heuristic_frob: Any = dataclasses.field(default_factory=_zero)
  entrywise_err: Any = dataclasses.field(default_factory=_zero)
  total_frob: Any = dataclasses.field(default_factory=_zero)

  def replace(self, **kw):
    return dataclasses.replace(self, **kw)


@dataclasses.dataclass(frozen=True)
class TrainingMetrics:
  """Per-statistic diagnostics kept inside the optimizer state (DS:338-363)."""
  inverse_pth_root_errors: Any = dataclasses.field(default_factory=_zero)
  inverse_pth_root_iters: Any = dataclasses.field(default_factory=_zero)
  final_error_ratio: Any = dataclasses.field(default_factory=_zero)
  max_eigen_value: Any = dataclasses.field(default_factory=_zero)
  total_retries: Any = dataclasses.field(default_factory=_zero)
  lobpcg_diagnostics: LOBPCGDiagnostics = dataclasses.field(
      default_factory=LOBPCGDiagnostics)
  inverse_pth_root_diagnostics: InversePthRootDiagnostics = dataclasses.field(
      default_factory=InversePthRootDiagnostics)
  conditioned_inverse_pth_root_diagnostics: InversePthRootDiagnostics = (
      dataclasses.field(default_factory=InversePthRootDiagnostics))
  fd: Any = dataclasses.field(default_factory=MaskedNode)

  def replace(self, **kw):
    return dataclasses.replace(self, **kw)


pytree.register_dataclass(LOBPCGDiagnostics,
                          [f.name for f in dataclasses.fields(LOBPCGDiagnostics)])
pytree.register_dataclass(
    InversePthRootDiagnostics,
    [f.name for f in dataclasses.fields(InversePthRootDiagnostics)])
pytree.register_dataclass(FDDiagnostics, list(_FD_FIELDS))
pytree.register_dataclass(TrainingMetrics,
                          [f.name for f in dataclasses.fields(TrainingMetrics)])


@dataclasses.dataclass(frozen=True)
class QuantizedValue:
  """quantization_utils.QuantizedValue (QU:26-113): float32 / bfloat16 pass-through, or
  int8 / int16 codes with one float32 bucket size per column (max over axis 0) and,
  with extract_diagonal, the float32 diagonal of a square matrix kept aside.  The
  integer modes run on the HIP kernels (ps_quantize_f32 / ps_dequantize_f32)."""
  quantized: Any
  diagonal: Any
  bucket_size: Any
  quantized_dtype: Any
  extract_diagonal: bool
  shape: Any

  @classmethod
  def from_float_value(cls, fvalue, quantized_dtype, extract_diagonal=False):
    if isinstance(fvalue, list) and not fvalue:
      return cls([], [], [], quantized_dtype, extract_diagonal, [])
    quantized, diagonal, bucket_size = cls.quantize(fvalue, quantized_dtype, extract_diagonal)
    return cls(quantized, diagonal, bucket_size, quantized_dtype, extract_diagonal,
               list(quantized.shape))

  @classmethod
  def quantize(cls, fvalue, quantized_dtype, extract_diagonal=False):
    """Returns (quantized, diagonal, bucket_size) (QU:45-95)."""
    if quantized_dtype == torch.float32:
      return fvalue, [], []
    if quantized_dtype == torch.bfloat16:
      return fvalue.to(torch.bfloat16), [], []
    if quantized_dtype not in (torch.int8, torch.int16):
      raise ValueError(f"Quantized dtype {quantized_dtype} not supported.")
    from . import kernels
    return kernels.quantize_grouped([fvalue], quantized_dtype, extract_diagonal)[0]

  def to_float(self):
    """QU:97-113."""
    if isinstance(self.quantized, list) and not self.quantized:
      return self.quantized
    if self.quantized_dtype == torch.float32:
      return self.quantized
    if self.quantized_dtype == torch.bfloat16:
      return self.quantized.to(torch.float32)
    from . import kernels
    diag = self.diagonal if self.extract_diagonal else []
    return kernels.dequantize_grouped([(self.quantized, diag, self.bucket_size)])[0]

  def replace(self, **kw):
    return dataclasses.replace(self, **kw)


pytree.register_dataclass(QuantizedValue, ["quantized", "diagonal", "bucket_size"],
                          ["quantized_dtype", "extract_diagonal", "shape"])


class ParameterStats(NamedTuple):
  """State associated to each parameter (DS:367-375, same field order)."""
  diagonal_statistics: QuantizedValue
  statistics: Optional[List[Any]]
  preconditioners: List[Any]
  diagonal_momentum: QuantizedValue
  momentum: QuantizedValue
  avg_grad: Union[Any, MaskedNode]
  training_metrics: Union[TrainingMetrics, MaskedNode]


class ShampooState(NamedTuple):
  count: Any
  stats: Any


# ---- sharded optimizer state (the reference's pjit mode, DS:376-399, 483-500) ----------
@dataclasses.dataclass(frozen=True)
class GlobalShardedParameterStats:
  """One stacked array per kind for ALL statistics of the model (DS:381-385).  The leading
  axis is the one the reference shards with pjit; here a rank keeps its own contiguous
  chunk of `statistics` (rows rank*b .. (rank+1)*b-1 of the padded list, the batch() order
  of DS:1827) and the whole of `preconditioners` and `exponents`."""
  statistics: Any       # [b, max_size, max_size]
  preconditioners: Any  # [N_padded, max_size, precond_dim(max_size)]
  exponents: Any        # [N_padded] int32

  def replace(self, **kw):
    return dataclasses.replace(self, **kw)


@dataclasses.dataclass(frozen=True)
class LocalShardedParameterStats:
  """Per-parameter state of the sharded mode (DS:390-399): everything that mirrors the
  parameter, plus where its statistics sit in the global arrays."""
  diagonal_statistics: Any
  diagonal_momentum: Any
  momentum: Any
  avg_grad: Any
  training_metrics: Any
  index_start: int = 0     # static: index into the global statistics list
  sizes: Any = ()          # static: sizes of this parameter's statistics

  def replace(self, **kw):
    return dataclasses.replace(self, **kw)


class ShardedShampooStats(NamedTuple):
  """DS:483-486."""
  global_stats: Any
  local_stats: Any


class InitFnState(NamedTuple):
  """DS:493-496: what init() returns in the sharded mode."""
  init_fn: Any
  pspec_fn: Any
  shape_and_dtype_fn: Any


pytree.register_dataclass(GlobalShardedParameterStats,
                          ["statistics", "preconditioners", "exponents"])
pytree.register_dataclass(LocalShardedParameterStats,
                          ["diagonal_statistics", "diagonal_momentum", "momentum", "avg_grad",
                           "training_metrics"], ["index_start", "sizes"])


def default_training_metrics(generate_fd_metrics: bool = False) -> TrainingMetrics:
  """DS:429-436."""
  if generate_fd_metrics:
    return TrainingMetrics(fd=FDDiagnostics())
  return TrainingMetrics()


def init_training_metrics(num_statistics: int, generate_training_metrics: bool,
                          generate_fd_metrics: bool = False, device=None):
  """DS:439-451: every leaf repeated `num_statistics` times, or MaskedNode."""
  if not generate_training_metrics:
    return MaskedNode()
  return pytree.tree_map(
      lambda x: torch.zeros((num_statistics,), dtype=torch.float32, device=device),
      default_training_metrics(generate_fd_metrics))
