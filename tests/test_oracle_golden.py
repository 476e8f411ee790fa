"""The CPU oracle against the golden vectors produced by the reference's own
source (tools/gen_golden.py).  Bit-exact on the generating machine; elsewhere
OpenBLAS may pick other kernels, so tolerances scale with conditioning."""
import json
import os

import numpy as np
import pytest

from oracle import shampoo_oracle as orc


def _load(golden_dir, name):
  return np.load(os.path.join(golden_dir, name))


def root_tolerance(name):
  """rel-Fro tolerance by input conditioning (fp32 roots move by ~cond * eps)."""
  if any(k in name for k in ("cond1e6", "cond1e7", "rank_deficient")):
    return 3e-2
  if any(k in name for k in ("cond1e4", "cond1e5", "square_wishart", "rank1")):
    return 3e-3
  return 5e-5


def check_root_case(case, z, root_fn):
  """Shared by the oracle test (CPU) and the HIP parity test (GPU)."""
  name = case["name"]
  probe = z["probe"]
  if case["full"]:
    a = z[name + "__a"]
  else:
    g = np.random.default_rng(1024).standard_normal((1024, 4096)).astype(np.float32)
    a = (g @ g.T).astype(np.float32)
  h, m = root_fn(a, case["p"], ridge_epsilon=case["ridge"],
                 relative_matrix_epsilon=case["rel"],
                 padding_start=case["padding_start"])
  gold_m = z[name + "__metrics"]
  got_m = np.array([m["inverse_pth_root_errors"], m["inverse_pth_root_iters"],
                    m["final_error_ratio"], m["max_eigen_value"], m["total_retries"]],
                   np.float32)
  tol = root_tolerance(name)
  # iteration / retry counts: exact (the 1e-6 threshold can flip by one step on
  # ill-conditioned inputs when products round differently)
  slack = 1 if tol > 1e-3 else 0
  assert abs(got_m[1] - gold_m[1]) <= slack, (name, got_m, gold_m)
  assert got_m[4] == gold_m[4], (name, got_m, gold_m)
  if np.isnan(gold_m[0]):
    assert np.isnan(got_m[0]), name
  else:
    assert got_m[0] <= max(4 * gold_m[0], 2e-6), (name, got_m, gold_m)
  if np.isnan(gold_m[3]):
    assert np.isnan(got_m[3]), name
  else:
    assert np.isclose(got_m[3], gold_m[3], rtol=2e-5, atol=1e-30), (name, got_m, gold_m)
  if case["full"]:
    ref = z[name + "__root"]
    if not np.all(np.isfinite(ref)):
      assert np.array_equal(np.isfinite(h), np.isfinite(ref)), name
      return
    nrm = np.linalg.norm(ref)
    if nrm == 0:
      assert not h.any(), name
    else:
      assert np.linalg.norm(h - ref) / nrm <= tol, (name, np.linalg.norm(h - ref) / nrm)
    ps = case["padding_start"]
    if ps is not None:  # padding rows/cols exactly zero (DST:367-398)
      assert not h[ps:, :].any() and not h[:, ps:].any(), name
  else:
    n = h.shape[0]
    pr = (h @ probe[:n]).astype(np.float32)
    ref = z[name + "__root_probe"]
    assert np.linalg.norm(pr - ref) / np.linalg.norm(ref) <= tol, name
    assert np.isclose(np.linalg.norm(h.astype(np.float64)), float(z[name + "__root_fro"]),
                      rtol=tol)


def newton_cases(golden_dir):
  with open(os.path.join(golden_dir, "newton_root_index.json")) as f:
    return json.load(f)


GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case", newton_cases(GOLD), ids=lambda c: c["name"])
def test_newton_root_oracle_vs_reference_golden(case, golden_dir):
  assert case["oracle_bitexact_at_gen"]
  z = _load(golden_dir, "newton_root.npz")
  check_root_case(case, z, orc.matrix_inverse_pth_root)


def test_power_iteration_and_mat_power_oracle_vs_golden(golden_dir):
  z = _load(golden_dir, "power_iter_matpower.npz")
  for nm in ("wishart128", "spec16_1e4", "wishart200", "diag_sep_n8"):
    a = z[f"pi_{nm}__a"]
    v, s, it = orc.power_iteration(a)
    assert np.isclose(s, z[f"pi_{nm}__s"], rtol=1e-5)
    assert abs(it - int(z[f"pi_{nm}__iters"])) <= 2 or it == 100
    assert min(np.linalg.norm(v - z[f"pi_{nm}__v"]), np.linalg.norm(v + z[f"pi_{nm}__v"])) < 2e-3
  a = z["pi_padded40in64__a"]
  v, s, _ = orc.power_iteration(a, padding_start=40)
  assert np.isclose(s, z["pi_padded40in64__s"], rtol=1e-5)
  assert not v[40:].any()
  for p in range(1, 9):
    r = orc.mat_power(z[f"mp_p{p}__m"], p)
    assert np.allclose(r, z[f"mp_p{p}__r"], rtol=1e-5, atol=1e-7)


def test_eigh_root_oracle_vs_golden(golden_dir):
  """The goldens come from the reference's source over FLOAT32 LAPACK (ssyevd: what jax's CPU path runs with
  x64 off, DS:35-38); each also carries the float64-internal result (`*_f64lapack`, NumPy's eigh: the
  accuracy yardstick) and, in the index, the reference's own root error against the float64 closed form.
  Another CPU's OpenBLAS rounds ssyevd differently: two float32 runs of an ill-conditioned block agree to
  the reference's own error (3.8e-2 at cond 1e6!), not to 1e-4 -- the bar scales with it."""
  z = _load(golden_dir, "eigh_root.npz")
  with open(os.path.join(golden_dir, "eigh_root_index.json")) as f:
    idx = json.load(f)
  assert any(c["root_error_vs_f64"] > 1e-3 for c in idx)          # ill-conditioned cases are in the set
  for c in idx:
    a = z[c["name"] + "__a"]
    ref = z[c["name"] + "__root"]
    assert ref.dtype == np.float32 and z[c["name"] + "__root_f64lapack"].dtype == np.float32
    assert c["lapack"].startswith("ssyevd")
    nrm = np.linalg.norm(ref)
    for lp, sfx, tol in (("f32", "", 4 * c["root_error_vs_f64"] + 1e-5), ("f64", "_f64lapack", 1e-4)):
      h, m = orc.matrix_inverse_pth_root_eigh(a, c["p"], padding_start=c["padding_start"], lapack=lp)
      assert h.dtype == np.float32
      if nrm == 0:
        assert not h.any()
        continue
      assert np.linalg.norm(h - z[c["name"] + "__root" + sfx]) / nrm < tol, (c["name"], lp)
      assert np.isclose(m["inverse_pth_root_errors"], float(z[c["name"] + "__err" + sfx]),
                        rtol=0.5, atol=1e-5)
    if nrm > 0:   # the index's accuracy figures are what the arrays say
      truth = orc.eigh_root_float64(a, c["p"], padding_start=c["padding_start"])
      tn = np.linalg.norm(truth)
      assert np.isclose(np.linalg.norm(ref - truth) / tn, c["root_error_vs_f64"], rtol=1e-2, atol=1e-9)
      # float32 LAPACK is what the reference runs: on ill-conditioned blocks it is decades from the
      # float64-internal result that rounds 1-5 mistook for it
      if c["root_error_vs_f64"] > 1e-4:
        assert c["root_error_vs_f64"] > 30 * c["root_error_vs_f64_f64lapack"], c["name"]


def test_float32_lapack_routines_are_single_precision():
  """oracle/lapack32.py: eigh32 / svd32 / qr_r32 are ssyevd / sgesdd / sgeqrf (float32 in, float32 out,
  float32 INSIDE), not NumPy's float64-internal routines rounded at the end."""
  from oracle import lapack32 as lp
  rng = np.random.default_rng(0)
  n = 96
  q, _ = np.linalg.qr(rng.standard_normal((n, n)))
  a = (q * 1e5 ** (-np.arange(n) / (n - 1))) @ q.T
  a = ((a + a.T) / 2).astype(np.float32)
  w64 = np.linalg.eigvalsh(a.astype(np.float64))
  w, v = lp.eigh32(a)
  wy, vy = lp.eigh64(a)
  assert w.dtype == np.float32 and v.dtype == np.float32 and np.all(np.diff(w) >= 0)
  assert np.abs(v.T @ v - np.eye(n)).max() < 1e-5
  e32, e64 = np.abs(w - w64).max(), np.abs(wy - w64).max()
  assert e64 < 1e-7 and e32 > 3 * e64, (e32, e64)                 # single precision shows
  assert e32 < 1e-5
  x = rng.standard_normal((40, 70)).astype(np.float32)
  u, s, vt = lp.svd32(x)
  assert u.shape == (40, 40) and s.shape == (40,) and vt.shape == (40, 70) and u.dtype == np.float32
  assert np.abs((u * s) @ vt - x).max() < 2e-5 and np.all(np.diff(s) <= 0)
  r = lp.qr_r32(np.ascontiguousarray(x.T))                        # [40, 40] upper triangular
  assert r.shape == (40, 40) and r.dtype == np.float32 and not np.tril(r, -1).any()
  assert np.abs(r.T @ r - x @ x.T).max() < 1e-3 * np.abs(x @ x.T).max()
  # LAPACK's sign convention (the same Householder vectors as dgeqrf): signs equal NumPy's R
  assert np.array_equal(np.sign(np.diag(r)), np.sign(np.diag(lp.qr_r64(np.ascontiguousarray(x.T)))))


def test_gram_update_oracle_vs_golden(golden_dir):
  z = _load(golden_dir, "gram_update.npz")
  keys = [k for k in z.files if k.endswith("__new")]
  assert len(keys) >= 20
  for k in keys:
    base = k[:-len("__new")]
    nm, ax, w = base.split("__")
    axis, w1 = int(ax[2:]), float(w[1:])
    w2 = 1.0 if w1 == 1.0 else {0.999: 0.001, 0.9: 0.1}[w1]
    r = orc.gram_weighted_update(z[base + "__old"], z[nm + "__g"], axis, w1, w2)
    assert np.allclose(r, z[k], rtol=2e-6, atol=1e-6), k


def test_oracle_closed_form_fp64():
  """Accuracy grading the reference lacks: vs V (L + eps)^(-1/p) V^T in float64."""
  g = np.random.default_rng(5).standard_normal((96, 384)).astype(np.float32)
  a = (g @ g.T).astype(np.float32)
  for p in (2, 4, 8):
    h, m = orc.matrix_inverse_pth_root(a, p)
    w, v = np.linalg.eigh(a.astype(np.float64))
    ref = (v * (w + 1e-6 * w.max()) ** (-1.0 / p)) @ v.T
    assert np.linalg.norm(h - ref) / np.linalg.norm(ref) < 1e-5
    assert m["total_retries"] == 1.0


def test_multi_replica_emulation_is_order_preserving():
  """pad-to-multiple / batch / gather / unbatch (DS:2841-2879) keeps list order
  for every world size, and padding entries are all-padding blocks."""
  rng = np.random.default_rng(0)
  stats = []
  for n in (8, 12, 8, 5, 16, 8, 12):
    g = rng.standard_normal((n, 4 * n)).astype(np.float32)
    stats.append((g @ g.T).astype(np.float32))
  exps = [4, 2, 4, 4, 2, 4, 2]
  prev = [np.eye(s.shape[0], dtype=np.float32) for s in stats]
  base, _, _ = orc.compute_preconditioners_reference_order(stats, exps, prev, 1)
  for world in (2, 4, 8):
    got, metrics, owners = orc.compute_preconditioners_reference_order(stats, exps, prev, world)
    assert len(metrics) % world == 0
    for a, b in zip(base, got):
      assert np.allclose(a, b, rtol=1e-5, atol=1e-7)
    assert owners == sorted(owners)
    for m in metrics[len(stats):]:
      assert m["inverse_pth_root_errors"] == 0.0


def test_quantization_oracle_vs_reference_golden(golden_dir):
  """quantization_utils.py:45-113 — elementwise IEEE arithmetic, so bit-exact anywhere."""
  import json
  from oracle import quantization_oracle as qorc
  z = np.load(os.path.join(golden_dir, "quantization.npz"))
  with open(os.path.join(golden_dir, "quantization_index.json")) as f:
    index = json.load(f)
  assert len(index) >= 15
  for c in index:
    name = c["name"]
    assert c["oracle_bitexact_at_gen"]
    dt = np.int8 if c["bits"] == 8 else np.int16
    x = z[f"{name}__x"]
    q, d, b = qorc.quantize(x, dt, c["extract"])
    assert q.dtype == dt and np.array_equal(q, z[f"{name}__codes"]), name
    assert np.array_equal(np.asarray(b, np.float32).view(np.uint32),
                          z[f"{name}__bucket"].view(np.uint32)), name
    if c["extract"]:
      assert np.array_equal(d.view(np.uint32), z[f"{name}__diag"].view(np.uint32)), name
    if f"{name}__float" in z.files:
      f32 = qorc.to_float(q, d, b, dt, c["extract"])
      assert np.array_equal(f32.view(np.uint32), z[f"{name}__float"].view(np.uint32)), name
  # the most negative code is never produced (QU:56-63)
  assert int(z["half_ties_i8__codes"].min()) >= -127
  # round half to even
  t = z["half_ties_i8__x"][1:, 0]
  assert np.array_equal(z["half_ties_i8__codes"][1:, 0], np.round(t).astype(np.int8))


def test_quantization_oracle_errors():
  from oracle import quantization_oracle as qorc
  import pytest
  with pytest.raises(ValueError):
    qorc.quantize(np.zeros((3, 4, 5), np.float32), np.int16, True)   # QU:67-69
  with pytest.raises(ValueError):
    qorc.quantize(np.zeros((3, 3), np.float32), np.int32, False)     # QU:64


# ---------------------------------------------------------------------------
# Frequent-Directions sketch update (config 5, DS:1123-1290): the oracle's fd_update_root
# against the reference's own packed sketches, small chains and the d = 1024 / rank 8
# full-size chain (inputs rebuilt from the seed, tools/gen_golden.py fd_big_grad).
def fd_big_grad(ps, rank, t, rng):
  g = rng.standard_normal((ps, 3 * ps)).astype(np.float32) * np.float32(1.0 + 0.3 * t)
  g[:rank + 2] *= np.linspace(6.0, 2.0, rank + 2)[:, None].astype(np.float32)
  return g


def test_oracle_fd_update_root_vs_reference_golden():
  from tests.test_optimizer_host_logic import packed_matches
  z = np.load(os.path.join(GOLD, "low_rank.npz"))
  with open(os.path.join(GOLD, "low_rank_index.json")) as f:
    idx = [c for c in json.load(f) if c["kind"] == "fd_chain"]
  assert idx
  for c in idx:
    nm, r = c["name"], c["rank"]
    for t in range(c["steps"]):
      got = orc.fd_update_root(z[f"fd_{nm}__factor{t}"], c["p"], r, ridge_epsilon=c["ridge"],
                               error_tolerance=0.0, relative_matrix_epsilon=c["rel"],
                               decay=c["decay"], padding_start=c["padding_start"],
                               prev=z[f"fd_{nm}__prev{t}"])
      assert packed_matches(got, z[f"fd_{nm}__new{t}"], r, tol=1e-4), (nm, t)


def test_oracle_fd_update_root_full_size_chain_vs_reference_golden():
  from tests.test_optimizer_host_logic import packed_matches
  z = np.load(os.path.join(GOLD, "low_rank_big.npz"))
  with open(os.path.join(GOLD, "low_rank_big_index.json")) as f:
    c = [c for c in json.load(f) if c["name"] == "d1024_r8"][0]
  rng = np.random.default_rng(c["seed"])
  d, r = c["d"], c["rank"]
  prev = np.zeros((d, r + 2), np.float32)
  for t in range(c["steps"]):
    g = fd_big_grad(d, r, t, rng)
    gram = (g.astype(np.float64) @ g.astype(np.float64).T)
    w, v = np.linalg.eigh(gram)   # any factor R with R R^T = Gram is equivalent (DS:1179-1193)
    fac = (v * np.sqrt(np.maximum(w, 0))).astype(np.float32)
    got = orc.fd_update_root(fac, c["p"], r, ridge_epsilon=c["ridge"], error_tolerance=0.0,
                             relative_matrix_epsilon=c["rel"], decay=c["decay"],
                             padding_start=d, prev=prev)
    ref = z[f"fd_d1024_r8__new{t}"]
    assert packed_matches(got, ref, r, tol=1e-3), t
    prev = ref
