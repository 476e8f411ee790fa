import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# Tests that A/B developer switches of the kernels (PS_NEWTON_PIPE, PS_EIGH_CJ, ...) do it through
# the environment; the library reads it in ONE function and only under PS_DEV_ENV=1
# (csrc/options.hip).  Every public mode is an argument (ps_options) and is tested as one.
os.environ.setdefault("PS_DEV_ENV", "1")


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
  return GOLDEN


@pytest.fixture(scope="session")
def device():
  import torch
  if not torch.cuda.is_available():
    pytest.skip("no GPU")
  return torch.device("cuda:0")


_GROUP = {}


def single_rank_group(backend):
  """A world-size-1 process group (the reference's int16 second-moment mode needs a
  batch axis, DS:2051-2054).  gloo on CPU; on the GPU box "nccl" is RCCL."""
  import tempfile
  import torch.distributed as dist
  if "g" not in _GROUP:
    if not dist.is_initialized():
      store = dist.FileStore(os.path.join(tempfile.mkdtemp(), "store"), 1)
      dist.init_process_group(backend, store=store, rank=0, world_size=1)
    _GROUP["g"] = dist.group.WORLD
  return _GROUP["g"]
