"""precondition_amd — MI355X-native preconditioner-compute path of Distributed Shampoo.

Drop-in for the hot path of google-research/precondition's
``precondition/distributed_shampoo.py``: statistics accumulation and the batched
matrix inverse p-th root, behind the reference's GradientTransformation /
ShampooState surface.  All arithmetic runs in hand-written HIP kernels for
gfx950 (libprecondition_amd.so, C-ABI in include/ps_api.h).
"""
__version__ = "0.1.0"

from .blocking import (BlockPartitioner, Preconditioner, batch, merge_small_dims,
                       pad_square_matrix, pad_vector, unbatch)
from .distributed_shampoo import (_pth_root_difference, distributed_shampoo,
                                  preconditioning_compute_steps_schedule)
from .state import (GradientTransformation, GraftingType, MaskedNode,
                    ParameterStats, PreconditionerType, QuantizedValue,
                    ShampooState, TrainingMetrics)


def __getattr__(name):
  # HIP-backed helpers (import lazily so that bookkeeping works without torch.cuda)
  if name in ("matrix_inverse_pth_root", "matrix_inverse_pth_root_batched",
              "power_iteration", "mat_power", "gram_weighted_update"):
    from . import kernels
    return getattr(kernels, name)
  if name in ("frequent_directions_update", "_fd_update_root", "_low_rank_root",
              "_fd_low_rank_pack", "_fd_low_rank_unpack", "_low_rank_pack",
              "_low_rank_unpack"):  # config-5 branch, same names as the reference module
    from . import low_rank
    return getattr(low_rank, name)
  raise AttributeError(name)
