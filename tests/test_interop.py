"""Frequent-Directions optimizer states across the two implementations (precondition_amd/interop.py).

tests/golden/fd_resume.npz: the reference's own source (tools/gen_golden.py gen_fd_resume) runs 3 FD updates, its
COMPLETE state is saved (statistics slots = its triangular factors R), then 3 more updates.  Here that state is
loaded into this build's ShampooState leaf for leaf, converted (R -> R R^T), and the run continues: the following
updates and the final sketches match the reference's.  CPU: host logic over the test backend; GPU: the HIP kernels."""
import json
import os

import numpy as np
import pytest
import torch

import precondition_amd as pa
from precondition_amd import interop, pytree
from precondition_amd.state import ParameterStats, ShampooState
from tests import cpu_backend
from tests.test_optimizer_host_logic import np_float, packed_matches

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cases():
  with open(os.path.join(GOLD, "fd_resume_index.json")) as f:
    return json.load(f)


def _load_mid_state(opt, params, z, name, device):
  """init() gives the structure; every leaf is then replaced by the reference's value at the interruption."""
  st = opt.init(params)
  flat, treedef = pytree.tree_flatten(st.stats, is_leaf=lambda x: isinstance(x, ParameterStats))
  t = lambda a: torch.tensor(np.asarray(a), device=device)
  out = []
  for i, s in enumerate(flat):
    stats = [t(z[f"{name}__mid_stat{i}_{j}"]) for j in range(len(s.statistics))]
    pres = [t(z[f"{name}__mid_precond{i}_{j}"]) for j in range(len(s.preconditioners))]
    for a, b in zip(stats + pres, list(s.statistics) + list(s.preconditioners)):
      assert a.shape == b.shape                                       # same layout, slot for slot
    diag = s.diagonal_statistics
    key = f"{name}__mid_diag_stats{i}"
    if key in z.files:
      diag = diag.replace(quantized=t(z[key]))
    mom = s.momentum.replace(quantized=t(z[f"{name}__mid_momentum{i}"]))
    dmom = s.diagonal_momentum.replace(quantized=t(z[f"{name}__mid_diag_momentum{i}"]))
    avg = s.avg_grad
    if f"{name}__mid_avg_grad{i}" in z.files:
      avg = t(z[f"{name}__mid_avg_grad{i}"])
    out.append(ParameterStats(diag, stats, pres, dmom, mom, avg, s.training_metrics))
  count = torch.tensor(int(z[f"{name}__count_mid"]), dtype=torch.int32)
  return ShampooState(count=count, stats=treedef.unflatten(out))


def _continue_from_reference_state(case, device, backend):
  name, n = case["name"], case["n_params"]
  z = np.load(os.path.join(GOLD, "fd_resume.npz"))
  kw = dict(case["kwargs"])
  block_size = kw.pop("block_size")
  rank = kw["compression_rank"]
  opt = pa.distributed_shampoo(case["lr"], block_size, _backend_for_testing=backend, **kw)
  params = tuple(torch.tensor(z[f"{name}__param{i}"], device=device) for i in range(n))
  ref_state = _load_mid_state(opt, params, z, name, device)
  # the reference's slots are triangular factors (LAPACK's signs: negative diagonal entries occur)
  r0 = np_float(pytree.tree_flatten(ref_state.stats, is_leaf=lambda x: isinstance(x, ParameterStats))[0][0].statistics[0])
  assert np.allclose(r0, np.tril(r0)) and (np.diag(r0) < 0).any()
  st = interop.import_reference_state(ref_state, rank)
  g0 = np_float(pytree.tree_flatten(st.stats, is_leaf=lambda x: isinstance(x, ParameterStats))[0][0].statistics[0])
  assert np.allclose(g0, r0 @ r0.T, rtol=1e-5, atol=1e-5 * np.abs(g0).max())
  worst = 0.0
  for t_ in range(case["first"], case["first"] + case["later"]):
    grads = tuple(torch.tensor(z[f"{name}__grad{i}_t{t_}"], device=device) for i in range(n))
    upd, st = opt.update(grads, st, params)
    for i in range(n):
      ref = z[f"{name}__upd{i}_t{t_}"]
      got = upd[i].cpu().numpy()
      worst = max(worst, np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))
  assert int(st.count) == int(z[f"{name}__count_end"])
  assert worst < 2e-3, worst
  flat = pytree.tree_flatten(st.stats, is_leaf=lambda x: isinstance(x, ParameterStats))[0]
  for i, s in enumerate(flat):
    for j, x in enumerate(s.preconditioners):
      ref = z[f"{name}__end_precond{i}_{j}"]
      got = np_float(x)
      if ref.shape[0] != ref.shape[1]:
        assert packed_matches(got, ref, rank, tol=5e-3), (name, i, j)
      else:
        assert np.linalg.norm(got - ref) <= 3e-2 * np.linalg.norm(ref), (name, i, j)
  # and back: exported factors reproduce the Gram matrices (the reference consumes only F F^T)
  back = interop.export_reference_state(st, rank)
  for s_b, s_o in zip(pytree.tree_flatten(back.stats, is_leaf=lambda x: isinstance(x, ParameterStats))[0], flat):
    for f_, g_ in zip(s_b.statistics, s_o.statistics):
      f_, g_ = np_float(f_), np_float(g_)
      assert np.allclose(f_ @ f_.T, g_, rtol=1e-4, atol=1e-4 * max(np.abs(g_).max(), 1e-30))


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_reference_fd_state_continues_here_host_logic(case):
  _continue_from_reference_state(case, torch.device("cpu"), cpu_backend)


@pytest.mark.gpu
@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_reference_fd_state_continues_here_hip(case, device):
  _continue_from_reference_state(case, device, None)
