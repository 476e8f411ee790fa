"""The C-ABI shared library loads and exports every symbol include/ps_api.h
declares (no compute calls here: this runs without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from precondition_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
  hdr = open(os.path.join(ROOT, "include", "ps_api.h")).read()
  hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
  return sorted(set(re.findall(r"\b(ps_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
  syms = declared_symbols()
  assert len(syms) >= 18
  raw = ctypes.CDLL(_lib.LIB_PATH)
  for s in syms:
    assert hasattr(raw, s), f"{s} declared in ps_api.h but not exported"
  assert sorted(_lib.exported_symbols()) == syms
  assert _lib.lib().ps_version() >= 100
  assert _lib.lib().ps_error_string(-2) == b"workspace too small"


def test_power_iteration_v0_is_numpy_randomstate_1729():
  for n in (1, 7, 624, 1024, 4096):
    out = np.zeros(n, np.float32)
    assert _lib.lib().ps_power_iteration_v0(n, out.ctypes.data) == 0
    ref = np.random.RandomState(1729).uniform(-1.0, 1.0, n).astype(np.float32)
    assert np.array_equal(out, ref)


def test_workspace_queries_and_argument_checks_on_host():
  L = _lib.lib()
  n = np.array([512, 768, 197], np.int32)
  p = np.array([4, 2, 4], np.int32)
  nbytes = L.ps_newton_root_workspace_bytes(3, n.ctypes.data, p.ctypes.data, None)
  # 10 padded square buffers per block dominate
  expect = 10 * 4 * (512 ** 2 + 768 ** 2 + 256 ** 2)
  assert expect <= nbytes <= expect * 1.05 + (1 << 20)
  bad_p = np.array([4, 0, 4], np.int32)
  assert L.ps_newton_root_workspace_bytes(3, n.ctypes.data, bad_p.ctypes.data, None) == 0
  assert L.ps_mat_power_workspace_bytes(64, 4) > 0
  assert L.ps_gemm_f32(None, 0, 0, None, None, None, 4, 4, 4, 4, 4, 4, 1, 0, 0, 0) == -1


def test_no_cpu_fallback():
  from precondition_amd import kernels
  with pytest.raises(_lib.PsError, match="no CPU path"):
    kernels.matrix_inverse_pth_root_batched([torch.eye(4)], [4])
  with pytest.raises(_lib.PsError, match="no CPU path"):
    kernels.gram_weighted_update(torch.eye(4), torch.ones(4, 3), 0, 1.0, 1.0)


def test_options_defaults_validation_and_precision_mapping():
  """ps_options: defaults from ps_options_init, the dict front end, and the reference's
  `precision` kwarg (DS:599, 708, 1883) mapped onto the product arithmetic."""
  from precondition_amd import kernels
  o, keep = _lib.make_options(None)
  assert o.struct_size == ctypes.sizeof(_lib.PsOptions) and not keep
  assert (o.products, o.accumulation, o.averaged_steps, o.execution) == (0, 0, -1, 0)
  # the marker a full-size struct must carry (a zero-initialised one is refused with PS_EINVAL)
  assert o.reserved[0] == 0x5053 and "PS_OPTIONS_MAGIC 0x5053" in open(
      os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "ps_api.h")).read()
  assert _lib.lib().ps_newton_averaged_steps() == 4
  o, keep = _lib.make_options({"products": "bf16x6", "execution": "persistent",
                               "iters_hint": [8, 14, 0], "fast_max_iters": 9,
                               "power_iteration": "streaming", "pi_timeout_ms": 0})
  assert (o.products, o.execution, o.power_iteration, o.pi_timeout_ms) == (1, 1, 1, 0)
  assert o.fast_max_iters == 9 and o.iters_hint == keep[0].ctypes.data and keep[0].dtype == np.float32
  with pytest.raises(ValueError):
    _lib.make_options({"products": "fp8"})
  with pytest.raises(ValueError):
    _lib.make_options({"no_such_option": 1})

  class P:  # stands for jax.lax.Precision members
    def __init__(self, name):
      self.name = name
  assert kernels.products_for_precision(None) == "f32"
  assert kernels.products_for_precision(P("HIGHEST")) == "f32"
  assert kernels.products_for_precision(P("HIGH")) == "bf16x6"
  assert kernels.products_for_precision("default") == "bf16x3"
  with pytest.raises(ValueError):
    kernels.products_for_precision("int4")
  from precondition_amd.distributed_shampoo import distributed_shampoo
  distributed_shampoo(0.1, 32, precision=P("HIGH"), tensordot_precision=None)
  with pytest.raises(ValueError):
    distributed_shampoo(0.1, 32, precision="nonsense")


def test_environment_is_read_in_one_function_only():
  """Modes are arguments (ps_options); getenv survives in csrc/options.hip alone, behind
  PS_DEV_ENV."""
  csrc = os.path.join(ROOT, "precondition_amd", "csrc")
  offenders = []
  for name in sorted(os.listdir(csrc)):
    if name.endswith((".hip", ".h")) and name != "options.hip":
      if "getenv" in open(os.path.join(csrc, name)).read():
        offenders.append(name)
  assert not offenders, offenders
  src = open(os.path.join(csrc, "options.hip")).read()
  assert src.count("void ps_dev_env_overrides(") == 1 and 'getenv("PS_DEV_ENV")' in src
  body = src[src.index("void ps_dev_env_overrides("):src.index("}  // namespace\n")]
  assert src.count("getenv(") == body.count("getenv(")
