"""CPU tests of the tridiagonal divide-and-conquer logic (precondition_amd/csrc/dc_core.h) that the
HIP eigensolver csrc/eigh_td.hip.h runs on the device: tests/dc_host.cpp drives the SAME scalar
routines (QL leaves, deflation, secular equation, Loewner weights) on the host, with float32
eigenvector matrices as on the GPU, and is checked here against NumPy / SciPy float64."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
from scipy.linalg import hessenberg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dc(tmp_path_factory):
  out = tmp_path_factory.mktemp("dc") / "dc_host.so"
  subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC",
                         os.path.join(ROOT, "tests", "dc_host.cpp"), "-o", str(out)])
  lib = ctypes.CDLL(str(out))

  def run(d, e, eps=1e-8):
    n = len(d)
    z = np.zeros((n, n), np.float32); ev = np.zeros(n); st = np.zeros(8, np.int32)
    d = np.ascontiguousarray(d, np.float64); e = np.ascontiguousarray(e, np.float64)
    rc = lib.dc_host_eigh(ctypes.c_int(n), d.ctypes.data_as(ctypes.c_void_p), e.ctypes.data_as(ctypes.c_void_p),
                          z.ctypes.data_as(ctypes.c_void_p), ev.ctypes.data_as(ctypes.c_void_p),
                          ctypes.c_double(eps), st.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return ev, z.astype(np.float64), st
  return run


def _check(run, d, e, eps=1e-8, ev_tol=1e-6):
  n = len(d)
  ev, z, st = run(d, e, eps)
  t = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
  ref = np.linalg.eigvalsh(t)
  nrm = max(np.abs(ref).max(), 1e-300)
  assert np.all(np.diff(ev) >= 0)                                   # ascending
  assert np.abs(ev - ref).max() <= ev_tol * nrm
  assert np.abs(z.T @ z - np.eye(n)).max() < 2e-6                   # float32 eigenvector matrices
  assert np.abs(t @ z - z * ev).max() <= 2e-6 * nrm
  assert st[5] == 0 and st[3] < 60                                  # QL converged, secular iterations bounded
  return st


@pytest.mark.parametrize("n", [33, 64, 100, 129, 161, 257, 512, 1000])
def test_random_tridiagonal(dc, n):
  rng = np.random.default_rng(n)
  _check(dc, rng.standard_normal(n), rng.standard_normal(n - 1))


def test_structured_spectra(dc):
  n = 512
  rng = np.random.default_rng(1)
  _check(dc, 2 * np.ones(n), -np.ones(n - 1))                                      # Toeplitz: close poles, rotations
  _check(dc, np.abs(np.arange(n) - n // 2).astype(float), np.ones(n - 1))          # Wilkinson: pairs of eigenvalues
  st = _check(dc, np.ones(n), 1e-9 * rng.standard_normal(n - 1))                   # everything deflates
  assert st[0] == 0
  _check(dc, rng.standard_normal(n), np.zeros(n - 1))                              # diagonal matrix
  g = np.logspace(0, -8, n)
  _check(dc, g, 0.3 * np.sqrt(g[:-1] * g[1:]))                                     # graded over eight decades


@pytest.mark.parametrize("n,k", [(256, 1024), (512, 100)])
def test_tridiagonal_of_statistics(dc, n, k):
  """Wishart and rank-deficient + ridge statistics (float32-rounded, as the GPU reduction sees them)."""
  rng = np.random.default_rng(n + k)
  g = rng.standard_normal((n, k)); a = g @ g.T
  a = a + 1e-6 * np.linalg.eigvalsh(a).max() * np.eye(n)
  a = a.astype(np.float32).astype(np.float64)
  h = hessenberg(a)
  for eps in (1e-8, 2.0 ** -24):
    st = _check(dc, np.diag(h).copy(), np.diag(h, 1).copy(), eps)
  if k < n:
    assert st[0] < 0.5 * st[1]   # the cluster at the ridge deflates
