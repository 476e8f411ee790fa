"""Host logic of the optimizer surface (tree handling, step gating, grafting,
momentum, failure select) against end-to-end goldens produced by the reference.
The numerical kernels are replaced by tests/cpu_backend.py here; the same goldens
are checked through the HIP kernels in tests/test_gpu_parity.py."""
import json
import os

import numpy as np
import pytest
import torch

import precondition_amd as pa
from precondition_amd import pytree
from tests import cpu_backend


def _index(golden_dir, name="e2e_index.json"):
  with open(os.path.join(golden_dir, name)) as f:
    return json.load(f)


def np_float(x):
  """Tensor or QuantizedValue -> float32 ndarray (int codes through the oracle's to_float,
  so that the CPU tests never touch the HIP library)."""
  from precondition_amd.state import QuantizedValue
  if isinstance(x, QuantizedValue):
    if x.quantized_dtype in (torch.int8, torch.int16):
      from oracle import quantization_oracle as qorc
      codes = x.quantized.cpu().numpy()
      return qorc.to_float(codes, x.diagonal.cpu().numpy() if x.extract_diagonal else [],
                           x.bucket_size.cpu().numpy(), codes.dtype, x.extract_diagonal)
    return x.quantized.float().cpu().numpy()
  return x.cpu().numpy()


def run_e2e_case(c, z, device, backend, skip_params=(), group=None, extra_kwargs=None, updates_out=None):
  name, n = c["name"], c["n_params"]
  kw = dict(c["kwargs"])
  kw.update(extra_kwargs or {})
  if "graft_type" in kw:
    kw["graft_type"] = pa.GraftingType(kw["graft_type"])
  if "precondtioner_type" in kw:
    kw["precondtioner_type"] = pa.PreconditionerType(kw["precondtioner_type"])
  block_size = kw.pop("block_size")
  if c.get("batch_axis"):
    assert group is not None
  opt = pa.distributed_shampoo(c["lr"], block_size,
                               batch_axis_name=group if c.get("batch_axis") else None,
                               _backend_for_testing=backend, **kw)
  params = tuple(torch.tensor(z[f"{name}__param{i}"], device=device) for i in range(n))
  st = opt.init(params)
  worst = 0.0
  for t in range(c["steps"]):
    grads = tuple(torch.tensor(z[f"{name}__grad{i}_t{t}"], device=device) for i in range(n))
    upd, st = opt.update(grads, st, params)
    if updates_out is not None:   # (copies: a donated state's updates live in reused buffers)
      updates_out.append([u.clone() for u in upd])
    for i in range(n):
      if i in skip_params:
        continue
      ref = z[f"{name}__upd{i}_t{t}"]
      got = upd[i].cpu().numpy()
      assert got.dtype == np.float32 and got.shape == ref.shape
      dist = np.linalg.norm(got - ref)
      # configurations that reach LAPACK (eigh / low-rank / FD roots) carry the same run over a float64-
      # internal LAPACK: twice the distance between the two goldens -- the reference's own float32
      # arithmetic uncertainty (bottom eigenpairs of rank-deficient statistics are rounding noise in
      # ssyevd) -- is not held against the build
      alt = f"{name}__upd{i}_t{t}_f64lapack"
      if alt in z.files:
        dist = max(0.0, dist - 2.0 * np.linalg.norm(ref - z[alt]))
      err = dist / max(np.linalg.norm(ref), 1e-30)
      worst = max(worst, err)
  assert int(st.count) == c["count"]
  return st, worst


@pytest.mark.parametrize("case", _index(os.path.join(os.path.dirname(__file__), "golden")),
                         ids=lambda c: c["name"])
def test_e2e_host_logic_vs_reference_golden(case, golden_dir):
  z = np.load(os.path.join(golden_dir, "e2e.npz"))
  st, worst = run_e2e_case(case, z, torch.device("cpu"), cpu_backend)
  # statistics here are sums of a few rank-1 terms + 1e-6 I (cond ~1e5): roots
  # move by ~1e-4 with the rounding order of the Gram update
  assert worst < 1e-3, worst
  name = case["name"]
  check_final_state(case, z, st)


def stat_matches(mine, ref, quantized=False):
  """Dense mode: equal.  FD mode: the reference keeps a triangular factor R of the
  Gram matrix in the slot (DS:1497-1505), this build keeps the Gram itself.
  int16 mode: equal up to a couple of code steps (max|column| / 32767)."""
  if np.allclose(mine, ref, rtol=1e-5, atol=1e-6):
    return True
  if quantized:
    return bool(np.abs(mine - ref).max() <= 1e-4 * np.abs(ref).max())
  return np.allclose(mine, ref @ ref.T, rtol=1e-4, atol=1e-5 * max(np.abs(mine).max(), 1e-30))


def packed_matches(mine, ref, rank, tol=2e-3, alt=None):
  """Rank-compressed preconditioner [d, r+2] (DS:555-592): scalars by value;
  eigenvectors only through what the application uses, sum_i w_i v_i v_i^T.
  `alt`: the same quantity from the reference over a float64-internal LAPACK (the goldens are float32
  LAPACK): twice its distance from `ref`, the reference's own arithmetic uncertainty, is added to the
  tolerance of the scalars."""
  r = abs(rank)
  ok = True
  slices = [(slice(0, r), -2), (slice(-r, None), -1)]
  if ref[1, -1] > 1e-20:  # const = tail^(-1/p) is noise when the tail is a rounding zero
    slices.append((slice(0, 2), -1))
  for sl in slices:
    a, b = mine[sl], ref[sl]
    slack = 0.0 if alt is None else 2.0 * float(np.abs(b - alt[sl]).max())
    ok &= bool(np.allclose(a, b, rtol=tol, atol=tol * max(np.abs(b).max(), 1e-30) + slack))
  ok &= bool(mine[-1, -2] == ref[-1, -2])
  if rank > 0:
    # (rank < 0 keeps the SMALLEST eigenpairs, which for few-sample statistics sit
    # in the ridge-dominated, numerically degenerate null space: any basis of it is
    # as good as another, so only the scalars are comparable there.)
    wa, wb = mine[:r, -2] - mine[0, -1], ref[:r, -2] - ref[0, -1]
    pa = (mine[:, :r] * wa) @ mine[:, :r].T
    pb = (ref[:, :r] * wb) @ ref[:, :r].T
    ok &= bool(np.linalg.norm(pa - pb) <= 5e-2 * max(np.linalg.norm(pb), 1e-30))
  return ok


def check_final_state(case, z, st, skip_params=()):
  name = case["name"]
  rank = case["kwargs"].get("compression_rank", 0)
  for i in range(case["n_params"]):
    if i in skip_params:
      continue
    s = st.stats[i]
    quantized = f"{name}__stat{i}_0_codes" in z.files
    for j, x in enumerate(s.statistics):
      if quantized:
        assert x.quantized.dtype == torch.int16 and x.extract_diagonal
      assert stat_matches(np_float(x), z[f"{name}__stat{i}_{j}"], quantized), (name, i, j)
    for j, x in enumerate(s.preconditioners):
      ref = z[f"{name}__precond{i}_{j}"]
      got = np_float(x)
      assert got.shape == ref.shape, (name, i, j, got.shape, ref.shape)
      akey = f"{name}__precond{i}_{j}_f64lapack"
      alt = z[akey] if akey in z.files else None
      if ref.shape[0] != ref.shape[1]:
        assert packed_matches(got, ref, rank, alt=alt), (name, i, j)
      else:
        # few-sample statistics are ill conditioned (cond ~ 1e6): roots move by ~1e-2
        slack = 0.0 if alt is None else 2.0 * np.linalg.norm(ref - alt)
        assert np.linalg.norm(got - ref) <= 3e-2 * np.linalg.norm(ref) + slack, (name, i, j)


def check_momentum(case, z, st, skip_params=(), tol=2e-3):
  """Momentum buffers: int8 codes + per-column bucket sizes where the reference
  quantizes (rank > 1 parameters, DS:2047-2049), float32 elsewhere."""
  name = case["name"]
  for i in range(case["n_params"]):
    if i in skip_params:
      continue
    m = st.stats[i].momentum
    ref = z[f"{name}__momentum{i}"]
    got = np_float(m)
    if f"{name}__momentum{i}_codes" in z.files:
      assert m.quantized.dtype == torch.int8 and not m.extract_diagonal
      assert tuple(m.bucket_size.shape) == z[f"{name}__momentum{i}_bucket"].shape
      # within a few code steps of the reference's dequantized momentum: a rounding-level
      # difference in an update flips a code now and then, and a flipped code is carried
      # into the next step's momentum (x beta1), where it can flip again
      step = z[f"{name}__momentum{i}_bucket"][None, ...]
      assert np.all(np.abs(got - ref) <= 4.0 * step + 1e-3 * np.abs(ref)), (name, i)
      assert np.mean(np.abs(got - ref) > 0.75 * step) < 0.25, (name, i)
    else:
      assert m.quantized.dtype == torch.float32
      assert np.linalg.norm(got - ref) <= tol * max(np.linalg.norm(ref), 1e-30), (name, i)


@pytest.mark.parametrize(
    "case", _index(os.path.join(os.path.dirname(__file__), "golden"), "e2e_quant_index.json"),
    ids=lambda c: c["name"])
def test_e2e_quantized_state_host_logic_vs_reference_golden(case, golden_dir):
  """best_effort_memory_usage_reduction: int8 momentum; int16 statistics
  and preconditioners with a batch axis.  Kernels replaced by the oracle here."""
  from tests.conftest import single_rank_group
  z = np.load(os.path.join(golden_dir, "e2e_quant.npz"))
  group = single_rank_group("gloo") if case.get("batch_axis") else None
  st, worst = run_e2e_case(case, z, torch.device("cpu"), cpu_backend, group=group)
  assert worst < 2e-3, worst
  check_final_state(case, z, st)
  check_momentum(case, z, st)


def test_step0_known_answer_from_reference_test():
  """DST:121-209/253-258: first update of the default config ends in -0.57."""
  opt = pa.distributed_shampoo(0.1, 32, batch_axis_name=None,
                               preconditioning_compute_steps=2,
                               _backend_for_testing=cpu_backend)
  params = (torch.tensor([[1., 3.], [2., 4.]]), torch.tensor([[3., 4.], [3., 4.]]))
  grads = (torch.tensor([[3., 4.], [5., 6.]]), torch.tensor([[1., 3.], [2., 1.]]))
  st = opt.init(params)
  upd, st = opt.update(grads, st, params)
  assert abs(float(upd[0][0, 0]) - (-0.57)) < 1e-4  # -0.1 * (3 + 0.9 * 3)
  assert all(torch.isfinite(u).all() for u in upd)


def test_constructor_validation_matches_reference():
  with pytest.raises(ValueError, match="reset_preconditioner=True requries frequent_directions"):
    pa.distributed_shampoo(0.1, 32, reset_preconditioner=True)
  with pytest.raises(ValueError, match="frequent_directions=True requires compression_rank > 0"):
    pa.distributed_shampoo(0.1, 32, frequent_directions=True)
  with pytest.raises(ValueError, match="average_grad requested but frequent_directions is False"):
    pa.distributed_shampoo(0.1, 32, average_grad=True)
  with pytest.raises(ValueError, match="to equal != preconditioning_compute_steps"):
    pa.distributed_shampoo(0.1, 32, frequent_directions=True, compression_rank=2,
                           statistics_compute_steps=2, preconditioning_compute_steps=3)


def test_state_layout():
  opt = pa.distributed_shampoo(0.1, 32)
  params = {"w": torch.zeros(100, 70), "b": torch.zeros(24), "big": torch.zeros(5000, 3)}
  st = opt.init(params)
  assert pa.ShampooState._fields == ("count", "stats")
  assert pa.ParameterStats._fields == ("diagonal_statistics", "statistics", "preconditioners",
                                       "diagonal_momentum", "momentum", "avg_grad",
                                       "training_metrics")
  assert st.count.dtype == torch.int32 and st.count.ndim == 0
  w = st.stats["w"]
  # [100, 70] does not merge (7000 > 4096); blocks [32,32,32,4] x [32,32,6]
  shapes = [tuple(s.shape) for s in w.statistics]
  assert len(shapes) == 24 and shapes[:6] == [(32, 32), (32, 32), (32, 32), (32, 32),
                                               (32, 32), (6, 6)]
  assert shapes[-2:] == [(4, 4), (6, 6)]
  assert torch.equal(w.statistics[0], 1e-6 * torch.eye(32))
  assert torch.equal(w.preconditioners[5], torch.eye(6))
  assert [tuple(s.shape) for s in st.stats["b"].statistics] == [(24, 24)]
  assert st.stats["big"].statistics == []  # dim 5000 > skip_preconditioning_dim_size_gt
  assert tuple(w.training_metrics.inverse_pth_root_errors.shape) == (24,)
  # the state is a plain pytree of tensors => torch.save-able
  leaves = pytree.tree_leaves(st)
  assert all(isinstance(x, torch.Tensor) for x in leaves)


def test_pytree_roundtrip():
  tree = {"a": [torch.ones(1), (torch.zeros(2), None)], "b": pa.MaskedNode()}
  leaves, td = pytree.tree_flatten(tree)
  assert len(leaves) == 2
  back = pytree.tree_unflatten(td, leaves)
  assert back["a"][1][1] is None and isinstance(back["b"], pa.MaskedNode)
  doubled = pytree.tree_map(lambda x: x * 2, tree)
  assert float(doubled["a"][0]) == 2.0


def test_schedule():
  lr = lambda t: 0.1 * (0.5 ** (t / 100))
  assert pa.preconditioning_compute_steps_schedule(lr, 10, 100, 0) == 10
  assert pa.preconditioning_compute_steps_schedule(lr, 10, 100, 100) == 60


@pytest.mark.parametrize(
    "case", _index(os.path.join(os.path.dirname(__file__), "golden"), "e2e_more_index.json"),
    ids=lambda c: c["name"])
def test_e2e_more_options_host_logic_vs_reference_golden(case, golden_dir):
  """Grafting variants (Adagrad-normalised, RMSProp + clipping, none, sqrt-n), skip rules,
  OUTPUT-only preconditioners with exponent_override, beta2 = 1, decoupled learning rate
  off, absolute epsilon: trajectories produced by the reference's own source."""
  z = np.load(os.path.join(golden_dir, "e2e_more.npz"))
  st, worst = run_e2e_case(case, z, torch.device("cpu"), cpu_backend)
  assert worst < 1e-3, worst
  check_final_state(case, z, st)
  check_momentum(case, z, st)


# ---------------------------------------------------------------------------
# pjit mode (shard_optimizer_states=True, DS:2162-2583) against goldens generated from the
# reference's own sharded_init_fn / sharded_update_fn (tools/gen_golden.py gen_e2e_sharded)
def run_sharded_case(c, z, device, backend, group=None):
  """Replays a pjit-mode golden: returns (final state, worst rel-Fro error of an update)."""
  name, n = c["name"], c["n_params"]
  kw = dict(c["kwargs"])
  if "graft_type" in kw:
    kw["graft_type"] = pa.GraftingType(kw["graft_type"])
  block_size = kw.pop("block_size")
  opt = pa.distributed_shampoo(c["lr"], block_size, batch_axis_name=group,
                               shard_optimizer_states=True,
                               num_devices_for_pjit=c["num_devices"],
                               _backend_for_testing=backend, **kw)
  params = tuple(torch.tensor(z[f"{name}__param{i}"], device=device) for i in range(n))
  fns = opt.init(params)
  assert isinstance(fns, pa.state.InitFnState)
  st = fns.init_fn(params)
  worst = 0.0
  for t in range(c["steps"]):
    grads = tuple(torch.tensor(z[f"{name}__grad{i}_t{t}"], device=device) for i in range(n))
    upd, st = opt.update(grads, st, params)
    for i in range(n):
      ref = z[f"{name}__upd{i}_t{t}"]
      got = upd[i].cpu().numpy()
      assert got.dtype == np.float32 and got.shape == ref.shape
      worst = max(worst, np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))
  assert int(st.count) == c["count"]
  return st, worst


def check_sharded_final_state(c, z, st, rank=0, world=1, tol_p=2e-3):
  """GlobalShardedParameterStats / LocalShardedParameterStats against the reference's: layout
  (stack sizes, index_start, sizes, exponents incl. the p = 1 padding rows) exactly, statistics
  and preconditioners by value; a rank holds rows [rank * b, (rank + 1) * b) of the statistics."""
  name = c["name"]
  gs = st.stats.global_stats
  ref_s, ref_p = z[f"{name}__global_statistics"], z[f"{name}__global_preconditioners"]
  ref_e = z[f"{name}__global_exponents"]
  b = ref_s.shape[0] // world
  assert tuple(gs.statistics.shape) == (b,) + ref_s.shape[1:]
  assert tuple(gs.preconditioners.shape) == ref_p.shape
  assert np.array_equal(gs.exponents.cpu().numpy(), ref_e)
  mine_s = gs.statistics.cpu().numpy()
  assert np.allclose(mine_s, ref_s[rank * b:(rank + 1) * b], rtol=1e-5, atol=1e-6)
  mine_p = gs.preconditioners.cpu().numpy()
  # roots of few-sample statistics (cond ~1e5) move by ~1e-4 with the rounding order of the
  # Gram update, as in the plain-mode goldens
  assert np.linalg.norm(mine_p - ref_p) <= tol_p * np.linalg.norm(ref_p)
  from precondition_amd import pytree
  locs = pytree.tree_flatten(
      st.stats.local_stats,
      is_leaf=lambda x: isinstance(x, pa.state.LocalShardedParameterStats))[0]
  for i, loc in enumerate(locs):
    assert int(loc.index_start) == int(z[f"{name}__index_start{i}"])
    assert [int(s) for s in loc.sizes] == [int(s) for s in z[f"{name}__sizes{i}"]]
    key = f"{name}__momentum{i}_codes"
    got, ref = np_float(loc.momentum), z[f"{name}__momentum{i}"]
    if key in z.files:   # int8 momentum (best_effort_memory_usage_reduction, DS:2047-2049)
      assert loc.momentum.quantized.dtype == torch.int8 and not loc.momentum.extract_diagonal
      step = z[f"{name}__momentum{i}_bucket"][None, ...]
      assert np.all(np.abs(got - ref) <= 4.0 * step + 1e-3 * np.abs(ref)), (name, i)
      assert np.mean(np.abs(got - ref) > 0.75 * step) < 0.25, (name, i)
    else:
      assert loc.momentum.quantized.dtype == torch.float32
      # norm-wise, as check_momentum does: roots of few-sample statistics move by ~1e-4 with the
      # rounding order of the products, and the momentum accumulates preconditioned gradients
      assert np.linalg.norm(got - ref) <= max(tol_p, 2e-3) * max(np.linalg.norm(ref), 1e-30), (name, i)
      dm, dref = np_float(loc.diagonal_momentum), z[f"{name}__diag_momentum{i}"]
      assert np.linalg.norm(dm - dref) <= max(tol_p, 2e-3) * max(np.linalg.norm(dref), 1e-30), (name, i)
    key = f"{name}__diag_stats{i}"
    if key in z.files:
      assert np.allclose(np_float(loc.diagonal_statistics), z[key], rtol=1e-5, atol=1e-7)


def _sharded_index(golden_dir, ndev):
  return [c for c in _index(golden_dir, "e2e_sharded_index.json") if c["num_devices"] == ndev]


@pytest.mark.parametrize(
    "case", _sharded_index(os.path.join(os.path.dirname(__file__), "golden"), 1),
    ids=lambda c: c["name"])
def test_e2e_sharded_state_host_logic_vs_reference_golden(case, golden_dir):
  z = np.load(os.path.join(golden_dir, "e2e_sharded.npz"))
  st, worst = run_sharded_case(case, z, torch.device("cpu"), cpu_backend)
  assert worst < 1e-3, worst
  check_sharded_final_state(case, z, st)
