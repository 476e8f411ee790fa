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
