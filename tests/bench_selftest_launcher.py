"""TESTS ONLY.  Runs bench.py's control flow (rank set-up, the two-phase async all-gather step,
barriers, max-over-ranks timing, the ViT-B strong-scaling leg, the JSON line) on CPU tensors
over gloo: the numerical kernels of precondition_amd.kernels are replaced HERE, from the test
side, by tests/cpu_backend (the oracle), and every size is shrunk.  bench.py itself has no such
hook.  Launched by tests/test_distributed_gloo.py through torch.distributed.run exactly as the
driver launches bench.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from oracle import shampoo_oracle as orc  # noqa: E402
from precondition_amd import kernels as K  # noqa: E402
from tests import cpu_backend  # noqa: E402


def _roots(matrices, ps, padding_starts=None, out=None, max_ev=None, symmetry="verify",
           eigh=False, **kw):
  del max_ev, symmetry
  kw.pop("num_iters", None)
  return cpu_backend.matrix_inverse_pth_root_batched(matrices, ps, padding_starts, out=out,
                                                     eigh=eigh, **kw)


def _power(matrices, padding_starts=None, **kw):
  lam = [orc.power_iteration(m.numpy(), padding_start=None if padding_starts is None
                             else int(padding_starts[i]))[1] for i, m in enumerate(matrices)]
  return torch.tensor(lam, dtype=torch.float32), torch.full((len(lam),), 100, dtype=torch.int32)


if __name__ == "__main__":
  K.matrix_inverse_pth_root_batched = _roots
  K.power_iteration_batched = _power
  K.stats_update_grouped = cpu_backend.stats_update_grouped
  bench.WORKLOADS.update({"cfg2_256x512_p4": (8, 16, 64, 4, 1234),
                          "headline_64x1024_p4": (4, 32, 128, 4, 1024),
                          "eigh_cfg3_64x2048_p2": (2, 16, 32, 2, 2048)})
  bench.SELFTEST = True
  bench.main()
