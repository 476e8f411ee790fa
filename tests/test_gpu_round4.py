"""Round-4 GPU tests (through the C-ABI):

* modes are arguments (ps_options): two executions / three product arithmetics selected per call
  in ONE process, bit-identical where the library promises it; a malformed struct is PS_EINVAL;
* the iteration-count hint: blocks the previous recompute marked well conditioned skip the averaged
  M updates and the segmented accumulation, and stay within 1e-6 of the careful result;
* segmented accumulation of the Newton products (DS:845-846): on the ill-conditioned p = 4 blocks of
  a ViT-B-like state the root is at least as close to the float64 root as the oracle's own float32
  evaluation (NumPy/OpenBLAS), which one fmaf chain per output element is not;
* comm.sharded_inverse_pth_roots repairs the eigenvalues of a power iteration that ran into an
  expired resident wait (the pi_first path has no in-call recovery of its own).
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import shampoo_oracle as orc

pytestmark = pytest.mark.gpu


def K():
  from precondition_amd import kernels
  return kernels


def L():
  from precondition_amd import _lib
  return _lib.lib()


def wishart(n, k, seed):
  g = np.random.default_rng(seed).standard_normal((n, k)).astype(np.float32)
  return (g @ g.T).astype(np.float32)


def vitb_like(n, m, seed=0):
  """The statistics of a ViT-B block after a few steps: beta2-averaged Gram of a rank-m gradient
  + the 1e-6 initial diagonal (bench.py VitBWorkload); cond ~ 4e3 ... 7e3 with the relative ridge."""
  rng = np.random.default_rng(seed)
  s = (1e-6 * 0.999 ** 5) * np.eye(n, dtype=np.float32)
  g = (rng.standard_normal((m, n)) * 0.02).astype(np.float32)
  for _ in range(5):
    s = (np.float32(0.999) * s + np.float32(0.001) * (g.T @ g)).astype(np.float32)
  return ((s + s.T) / 2).astype(np.float32)


def f64_root(a, p, ridge):
  w, v = np.linalg.eigh(a.astype(np.float64))
  return (v * (np.maximum(w, 0) + ridge) ** (-1.0 / p)) @ v.T


def rel(a, b):
  return float(np.linalg.norm(a - b) / np.linalg.norm(b))


# ---------------------------------------------------------------------------
def test_two_executions_selected_per_call_are_bit_identical(device):
  """ps_options.execution: the staged and the persistent execution in one process, interleaved,
  with and without hints -- no environment variable involved (VERDICT: 'two callers in one
  process cannot pick different modes')."""
  arrs = [wishart(300, 900, 1), vitb_like(384, 288, 2), wishart(128, 512, 3), wishart(130, 140, 4)]
  ps = [4, 4, 2, 4]
  mats = [torch.tensor(a, device=device) for a in arrs]
  out = {}
  for tag, opts in (("staged", {"execution": "staged"}), ("persistent", {"execution": "persistent"}),
                    ("staged2", {"execution": "staged"})):
    r, m = K().matrix_inverse_pth_root_batched(mats, ps, options=opts)
    torch.cuda.synchronize()
    out[tag] = ([x.clone() for x in r], m.clone())
  for a, b in (("staged", "persistent"), ("staged", "staged2")):
    assert torch.equal(out[a][1], out[b][1]), (a, b)
    for x, y in zip(out[a][0], out[b][0]):
      assert torch.equal(x, y), (a, b)
  hint = out["staged"][1][:, 1].cpu().numpy()
  r1, m1 = K().matrix_inverse_pth_root_batched(mats, ps, options={"execution": "staged", "iters_hint": hint})
  r2, m2 = K().matrix_inverse_pth_root_batched(mats, ps, options={"execution": "persistent", "iters_hint": hint})
  assert torch.equal(m1, m2)
  for x, y in zip(r1, r2):
    assert torch.equal(x, y)


def test_malformed_options_struct_is_einval(device):
  from precondition_amd import _lib
  a = torch.tensor(wishart(64, 128, 0), device=device)
  with pytest.raises(_lib.PsError, match="-1"):
    K().matrix_inverse_pth_root_batched([a], [4], options={"products": 7})
  with pytest.raises(ValueError):
    K().matrix_inverse_pth_root_batched([a, a], [4, 4], options={"iters_hint": [8.0]})


@pytest.mark.parametrize("n,k", [(512, 2048), (1000, 3000)])
def test_product_arithmetic_selected_per_call(n, k, device):
  """ps_options.products = what the factory's `precision` selects (DS:599, 708): f32 (HIGHEST),
  bf16x6 (HIGH), bf16x3 (DEFAULT) on the same inputs in one process.  The split modes finish in
  exact float32, so all three meet north_star's 1e-4 against the oracle on well-conditioned
  blocks; iteration counts may differ by one."""
  arrs = [wishart(n, k, 10 + i) for i in range(3)]
  mats = [torch.tensor(a, device=device) for a in arrs]
  h_ref, m_ref = orc.matrix_inverse_pth_root(arrs[0], 4)
  for mode, bar in (("f32", 5e-5), ("bf16x6", 5e-5), ("bf16x3", 1e-4)):
    r, m = K().matrix_inverse_pth_root_batched(mats, [4, 4, 4], options={"products": mode})
    m = m.cpu().numpy()
    assert rel(r[0].cpu().numpy(), h_ref) < bar, mode
    assert abs(m[0, 1] - m_ref["inverse_pth_root_iters"]) <= (0 if mode == "f32" else 1), (mode, m[0])
    assert (m[:, 0] < 1e-5).all(), (mode, m[:, 0])


def test_hint_marks_well_conditioned_blocks_fast(device):
  """A block whose previous iteration count is <= fast_max_iters (8) takes 0 averaged M updates
  (PS_M_AVG_STEPS) and plain chains; one without a hint, or with a larger count, takes the careful
  path (4 averaged steps).  On well-conditioned blocks both are within 1e-6 of each other and of
  the oracle; the hint is per block."""
  arrs = [wishart(512, 2048, 40 + i) for i in range(4)]
  mats = [torch.tensor(a, device=device) for a in arrs]
  r0, m0 = K().matrix_inverse_pth_root_batched(mats, [4] * 4)
  m0 = m0.cpu().numpy()
  assert (m0[:, 1] == 8).all() and (m0[:, 7] == 4).all(), m0
  hint = m0[:, 1].copy()
  hint[1] = 0.0            # no hint for block 1
  hint[2] = 13.0           # a slow block last time
  r1, m1 = K().matrix_inverse_pth_root_batched(mats, [4] * 4, options={"iters_hint": hint})
  m1 = m1.cpu().numpy()
  assert list(m1[:, 7]) == [0.0, 4.0, 4.0, 0.0], m1[:, 7]
  assert (m1[:, 1] == 8).all()
  for i in range(4):
    a, b = r0[i].cpu().numpy(), r1[i].cpu().numpy()
    assert rel(b, a) < 2e-6, i
    if i in (1, 2):
      assert np.array_equal(a, b)      # same path as the unhinted call
  h_ref, _ = orc.matrix_inverse_pth_root(arrs[0], 4)
  assert rel(r1[0].cpu().numpy(), h_ref) < 5e-6


@pytest.mark.parametrize("n,m", [(768, 576), (1024, 768)])
def test_segmented_accumulation_at_least_as_accurate_as_reference_arithmetic(n, m, device):
  """The north_star bar is 1e-4 against the reference; on the cond ~5e3, p = 4 blocks of the ViT-B
  state two float32 evaluations of the same iteration differ by more than that, so the comparison
  is made against the float64 root: the build (default options: averaged M updates + chains of 128
  for the M-side products) must not be further from it than the oracle's NumPy/OpenBLAS float32
  evaluation.  With plain chains (accumulation='chain', rounds 1-3) it is 1.3-1.6x further."""
  errs = {"seg": [], "chain": [], "oracle": []}
  for seed in range(3):
    a = vitb_like(n, m, seed)
    t = torch.tensor(a, device=device)
    h_o, m_o = orc.matrix_inverse_pth_root(a, 4)
    ridge = 1e-6 * m_o["max_eigen_value"]
    h64 = f64_root(a, 4, ridge)
    errs["oracle"].append(rel(h_o, h64))
    for name, opts in (("seg", None), ("chain", {"accumulation": "chain"})):
      r, met = K().matrix_inverse_pth_root_batched([t], [4], options=opts)
      met = met.cpu().numpy()
      assert met[0, 1] == m_o["inverse_pth_root_iters"] and met[0, 4] == m_o["total_retries"], (name, met[0], m_o)
      errs[name].append(rel(r[0].cpu().numpy(), h64))
  seg, chain, oracle = (float(np.mean(errs[k])) for k in ("seg", "chain", "oracle"))
  print(f"n={n}: error vs float64  build(seg) {seg:.3e}  build(chain) {chain:.3e}  oracle {oracle:.3e}")
  assert seg <= 1.0 * oracle, (seg, oracle)
  assert max(e / o for e, o in zip(errs["seg"], errs["oracle"])) <= 1.1
  assert seg < chain


def test_sharded_roots_repair_eigenvalues_after_an_expired_power_iteration(device):
  """comm.sharded_inverse_pth_roots' pi_first path runs the standalone power iteration, which has
  no in-call recovery: with every resident wait forced to expire (pi_timeout_ms = 0) its
  eigenvalues are NaN.  The function reads the expiry count before it and after the first root
  call and, when it moved, recomputes the eigenvalues on the streaming kernels and roots the
  phase again: nothing NaN, roots match the oracle."""
  import torch.distributed as dist
  from tests.conftest import single_rank_group
  from precondition_amd import comm
  L().ps_power_iteration_reset_health()
  arrs = [wishart(256, 1024, 70 + i) for i in range(12)]
  mats = [torch.tensor(a, device=device) for a in arrs]
  group = single_rank_group("nccl")
  roots, met = comm.sharded_inverse_pth_roots(
      mats, [4] * 12, group=group, ownership="lpt", overlap_min_bytes=0, pi_first=True,
      options={"pi_timeout_ms": 0})
  torch.cuda.synchronize()
  met = met.cpu().numpy()
  e = C.c_uint()
  L().ps_power_iteration_health(C.addressof(e), None, None)
  assert e.value > 0
  assert np.isfinite(met[:, :5]).all(), met
  for i in (0, 11):
    h_ref, m_ref = orc.matrix_inverse_pth_root(arrs[i], 4)
    assert rel(roots[i].cpu().numpy(), h_ref) < 5e-5
    assert np.isclose(met[i, 3], m_ref["max_eigen_value"], rtol=2e-5)
  L().ps_power_iteration_reset_health()


def test_stacked_batch_form_bit_identical_to_list_form(device):
  """matrix_inverse_pth_root_batched(xs[b, n, n]) -- the reference's own vmap signature
  (DS:2742-2744) -- against the list form: same roots and metrics, also on a strided view."""
  arrs = np.stack([wishart(200, 600, 90 + i) for i in range(5)])
  xs = torch.tensor(arrs, device=device)
  r_list, m_list = K().matrix_inverse_pth_root_batched(list(xs.unbind(0)), [4, 2, 4, 8, 1])
  r_st, m_st = K().matrix_inverse_pth_root_batched(xs, [4, 2, 4, 8, 1])
  assert isinstance(r_st, torch.Tensor) and tuple(r_st.shape) == (5, 200, 200)
  assert torch.equal(m_list, m_st)
  for i in range(5):
    assert torch.equal(r_list[i], r_st[i])
  big = torch.zeros((5, 256, 256), device=device)
  big[:, :200, :200] = xs
  out = torch.empty((5, 200, 200), device=device)
  r_v, m_v = K().matrix_inverse_pth_root_batched(big[:, :200, :200], [4, 2, 4, 8, 1], out=out)
  assert r_v is out and torch.equal(m_v, m_st) and torch.equal(out, r_st)


# ---------------------------------------------------------------------------
# Fused filter step of the FD branch (ps_fd_cy_step_f32) and its fragment-major operands
def _frag_index_a(n, k):
  r = torch.arange(n)[:, None]
  c = torch.arange(k)[None, :]
  return ((((r // 64) * (k // 16) + c // 16) * 2 + (r % 64) // 32) * 512 + (32 * ((c % 16) // 8) + r % 32) * 8 + c % 8)


def _frag_index_y(n, b):
  r = torch.arange(n)[:, None]          # row of the iterate = k of the next product
  c = torch.arange(b)[None, :]
  return ((r // 16) * (b // 32) + c // 32) * 512 + (32 * ((r % 16) // 8) + c % 32) * 8 + r % 8


def test_fragment_major_conversion_is_a_permutation_of_the_plain_one(device):
  x = torch.randn((256, 192), device=device) * 3.0
  hi, lo = K().to_bf16(x, split=True)
  f = K().to_bf16(x, split=True, tiled="frag")
  idx = _frag_index_a(256, 192).to(device)
  assert torch.equal(f.hi[idx.reshape(-1)].view(256, 192), hi)
  assert torch.equal(f.lo[idx.reshape(-1)].view(256, 192), lo)
  assert sorted(idx.reshape(-1).tolist()) == list(range(256 * 192))
  with pytest.raises(ValueError):
    K().to_bf16(torch.randn((100, 64), device=device), split=True, tiled="frag")


@pytest.mark.parametrize("bsz,n,b", [(3, 256, 96), (2, 384, 64), (1, 128, 32), (8, 1024, 96), (16, 1024, 96)])
def test_fused_filter_step_matches_product_plus_recurrence(bsz, n, b, device):
  """(at most 128 blocks of 64 rows run on 32-row workgroups: the last case is the 64-row form)
  ps_fd_cy_step_f32 against its two-launch form (gemm_bf16_grouped on the same hi/lo operands, then
  ps_fd_filter_step_f32): same arithmetic per element, another order of the k sum -> agreement to
  float32 rounding of the sum; the bf16 planes it writes are exactly the split of ITS y_next; a factor
  whose degree is below the step keeps its iterate bit for bit."""
  gen = torch.Generator(device=device).manual_seed(11)
  cs = []
  for j in range(bsz):
    g = torch.randn((n, n // 2), generator=gen, device=device)
    cs.append((g @ g.T) / n)
  y = torch.randn((bsz, n, b), generator=gen, device=device)
  y_prev = torch.randn((bsz, n, b), generator=gen, device=device)
  params = torch.tensor([[0.4, 0.5, 0.3, 12.0]] * bsz, device=device)
  params[-1, 3] = 2.0 if bsz > 1 else 12.0          # last factor: degree 2 < step 3 (bsz > 1)
  step = 3
  # reference: plain layouts
  c_t = [K().to_bf16(c, split=True, tiled=True) for c in cs]
  yt_hi, yt_lo = K().to_bf16(y.view(bsz * n, b), split=True, transpose=True)
  z = torch.empty_like(y)
  K().gemm_bf16_grouped([(c_t[j], (yt_hi[:, j * n:(j + 1) * n], yt_lo[:, j * n:(j + 1) * n]), z[j])
                         for j in range(bsz)])
  want = torch.empty_like(y)
  K().fd_filter_step(z, y, y_prev, want, params, step, want_bf16=False)
  # fused: fragment-major layouts
  c_f = [K().to_bf16(c, split=True, tiled="frag") for c in cs]
  idx = _frag_index_y(n, b).to(device).reshape(-1)
  y_hi, y_lo = K().to_bf16(y.view(bsz * n, b), split=True)
  pl_hi = torch.empty((bsz * n * b,), dtype=torch.bfloat16, device=device)
  pl_lo = torch.empty_like(pl_hi)
  for j in range(bsz):
    pl_hi[j * n * b + idx] = y_hi[j * n:(j + 1) * n].reshape(-1)
    pl_lo[j * n * b + idx] = y_lo[j * n:(j + 1) * n].reshape(-1)
  got = torch.empty_like(y)
  nt = (torch.zeros_like(pl_hi), torch.zeros_like(pl_lo))
  K().fd_cy_step(c_f, (pl_hi, pl_lo), y, y_prev, got, nt, params, step)
  torch.cuda.synchronize()
  scale = float(want.abs().max())
  assert float((got - want).abs().max()) <= 2e-6 * scale, float((got - want).abs().max()) / scale
  if bsz > 1:
    assert torch.equal(got[-1], y[-1])
  g_hi, g_lo = K().to_bf16(got.view(bsz * n, b), split=True)
  for j in range(bsz - 1 if bsz > 1 else 1):
    assert torch.equal(nt[0][j * n * b + idx].view(n, b), g_hi[j * n:(j + 1) * n])
    assert torch.equal(nt[1][j * n * b + idx].view(n, b), g_lo[j * n:(j + 1) * n])
  # the first-step kernel writes the same planes (ldt = 0)
  t_hi, t_lo = K().fd_filter_step(z, y, None, want, params, 1, frag=True)
  w_hi, w_lo = K().to_bf16(want.view(bsz * n, b), split=True)
  for j in range(bsz):
    assert torch.equal(t_hi[j * n * b + idx].view(n, b), w_hi[j * n:(j + 1) * n])
    assert torch.equal(t_lo[j * n * b + idx].view(n, b), w_lo[j * n:(j + 1) * n])
  # shapes the fused step does not take are refused, not mangled
  from precondition_amd import _lib
  with pytest.raises((_lib.PsError, ValueError)):
    K().fd_cy_step(c_f, (pl_hi, pl_lo), y, y_prev, got, (pl_hi, pl_lo), params, step)


def test_filter_round_fragment_major_equals_tile_blocked_round(device):
  """ps_fd_filter_round_f32 with fragment-major covariances (one launch per step) against the same
  round on tile-blocked ones (product + reduce + recurrence launches): 12 steps, three factors with
  different degrees."""
  bsz, n, b = 3, 512, 96
  gen = torch.Generator(device=device).manual_seed(5)
  cs = []
  for j in range(bsz):
    g = torch.randn((n, n // 4), generator=gen, device=device)
    cs.append((g @ g.T) / n)
  x = torch.linalg.qr(torch.randn((bsz, n, b), generator=gen, device=device))[0].contiguous()
  z = torch.stack([cs[j] @ x[j] for j in range(bsz)]).contiguous()
  top = [float(torch.linalg.eigvalsh(c)[-1]) for c in cs]
  params = torch.tensor([[0.05 * t, 0.05 * t, 0.05 / 0.95, d] for t, d in zip(top, (12.0, 7.0, 3.0))], device=device)
  outs = []
  for layout in (True, "frag"):
    c16 = [K().to_bf16(c, split=True, tiled=layout) for c in cs]
    bufs = [x.clone(), torch.empty_like(x), torch.empty_like(x)]
    outs.append(K().fd_filter_round(c16, z.clone(), bufs, params, 12).clone())
  torch.cuda.synchronize()
  for j in range(bsz):
    a, b_ = outs[0][j], outs[1][j]
    assert float((a - b_).norm() / a.norm()) < 5e-6, (j, float((a - b_).norm() / a.norm()))


@pytest.mark.parametrize("b", [96, 64, 33, 5])
def test_chol_rinv_against_float64_cholesky(b, device):
  """ps_chol_rinv_batched_f32 (the CholeskyQR step of the FD branch's orthonormalisation): R^-1 of
  G = R^T R against numpy's float64 factorisation on well and badly conditioned Gram matrices, and
  the dropped directions of a rank-deficient one (zero row and column; the kept block still
  whitens G)."""
  rng = np.random.default_rng(b)
  mats = []
  for cond in (1.0, 1e3, 1e6):
    x = rng.standard_normal((4 * b, b)) * np.logspace(0, -np.log10(cond) / 2, b)[None, :]
    mats.append((x.T @ x).astype(np.float32))
  g = torch.tensor(np.stack(mats), device=device)
  out = K().chol_rinv_batched(g, 1e-10).cpu().numpy().astype(np.float64)
  for m, o in zip(mats, out):
    m64 = 0.5 * (m.astype(np.float64) + m.astype(np.float64).T)
    want = np.linalg.inv(np.linalg.cholesky(m64)).T          # R^-1, upper triangular
    assert np.allclose(o, np.triu(o)), "R^-1 must be upper triangular"
    assert np.abs(o - want).max() <= 2e-6 * np.abs(want).max()
    assert np.abs(o.T @ m64 @ o - np.eye(b)).max() < 5e-5 * np.sqrt(np.linalg.cond(m64))
  if b >= 33:
    x = rng.standard_normal((4 * b, b))
    x[:, 7] = 0.0                      # a zero column: its pivot is exactly 0
    x[:, 20] = x[:, 3]                 # a duplicate: its pivot is at rounding level
    m = (x.T @ x).astype(np.float32)
    o = K().chol_rinv_batched(torch.tensor(m[None], device=device), 1e-6).cpu().numpy()[0].astype(np.float64)
    for d in (7, 20):
      assert not o[d].any() and not o[:, d].any(), d
    keep = [i for i in range(b) if i not in (7, 20)]
    w = o.T @ m.astype(np.float64) @ o
    assert np.abs(w[np.ix_(keep, keep)] - np.eye(len(keep))).max() < 1e-4


@pytest.mark.parametrize("bsz,n", [(3, 256), (2, 200), (1, 65)])
def test_fd_cov_update_one_pass_equals_torch_form(bsz, n, device):
  """ps_fd_cov_update_f32: c <- 0.5 ((decay c + g) + (decay c + g)^T) in one pass, against the three
  torch passes it replaces (low_rank._fd_update_root_group); exactly symmetric result, also on
  inputs that are not symmetric."""
  gen = torch.Generator(device=device).manual_seed(n)
  c = torch.randn((bsz, n, n), generator=gen, device=device)
  grams = [torch.randn((n, n), generator=gen, device=device) for _ in range(bsz)]
  want = torch.stack([0.999 * c[j] + grams[j] for j in range(bsz)])
  want = (want + want.transpose(1, 2)) * 0.5
  got = K().fd_cov_update(c.clone(), grams, 0.999)
  assert torch.equal(got, got.transpose(1, 2))
  assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())


@pytest.mark.parametrize("d,k", [(4096, 4096), (4224, 1056), (1024, 2048), (2048, 512), (200, 64)])
def test_symmetric_bf16_product_equals_full_product(d, k, device):
  """gemm_bf16_grouped(symmetric=True) -- the Gram matrix of an FD gradient block -- against the full
  product on the same hi/lo operands: the same three products per element, k summed in pieces whose
  boundaries differ (the K-split tail round at 4096 / 4224, every tile below 512 tiles) -> float32
  rounding of the sum; bitwise symmetric; two products in one call."""
  gen = torch.Generator(device=device).manual_seed(d)
  g = torch.randn((d, k), generator=gen, device=device)
  hi, lo = K().to_bf16(g, split=True)
  full = torch.empty((d, d), device=device)
  K().gemm_bf16_grouped([((hi, lo), (hi, lo), full)])
  sym = torch.full((d, d), float("nan"), device=device)
  K().gemm_bf16_grouped([((hi, lo), (hi, lo), sym)], symmetric=True)
  torch.cuda.synchronize()
  assert torch.equal(sym, sym.t())
  iu = torch.triu(torch.ones((d, d), dtype=torch.bool, device=device))
  diff = (sym - full).abs()[iu]
  # tiles of the K-split tail sum their k pieces in another order: float32 rounding of the sum
  assert float(diff.max()) <= 2e-6 * float(full.abs().max()), float(diff.max())
  ref = g.double() @ g.double().t()
  assert float((sym.double() - ref).norm() / ref.norm()) < 2e-5
  if d <= 2048:   # two symmetric products in ONE grouped call (each has its own partial tiles)
    g2 = torch.randn((d, k), generator=gen, device=device)
    hi2, lo2 = K().to_bf16(g2, split=True)
    s1, s2 = torch.empty_like(sym), torch.empty_like(sym)
    K().gemm_bf16_grouped([((hi, lo), (hi, lo), s1), ((hi2, lo2), (hi2, lo2), s2)], symmetric=True)
    alone = torch.empty_like(sym)
    K().gemm_bf16_grouped([((hi2, lo2), (hi2, lo2), alone)], symmetric=True)
    assert torch.equal(s1, sym) and torch.equal(s2, alone)


@pytest.mark.parametrize("bsz,n,b", [(2, 512, 96), (1, 256, 64), (8, 1024, 96), (16, 1024, 64)])
def test_six_product_cx_has_float32_accuracy(bsz, n, b, device):
  """ps_fd_cx6_f32 (C x of the Rayleigh-Ritz step on three bf16 planes per operand) against float64:
  as close as the float32 MFMA product it replaces, and the hi / lo planes of the three-plane
  conversion are those of the two-plane one."""
  gen = torch.Generator(device=device).manual_seed(7 * n)
  cs = []
  for j in range(bsz):
    g = torch.randn((n, n // 2), generator=gen, device=device)
    cs.append((g @ g.T) / n)
  x = torch.randn((bsz, n, b), generator=gen, device=device)
  c3 = [K().to_bf16(c, split=True, tiled="frag3") for c in cs]
  c2 = K().to_bf16(cs[0], split=True, tiled="frag")
  assert torch.equal(c3[0].hi, c2.hi) and torch.equal(c3[0].lo, c2.lo)
  rec = c3[0].hi.float() + c3[0].lo.float() + c3[0].lo2.float()
  idx = _frag_index_a(n, n).to(device).reshape(-1)
  assert float((rec[idx].view(n, n) - cs[0]).abs().max()) <= 2.0 ** -22 * float(cs[0].abs().max())
  z = torch.empty_like(x)
  K().fd_cx6(c3, x, z)
  z32 = torch.stack([cs[j] @ x[j] for j in range(bsz)])
  ref = torch.stack([cs[j].double() @ x[j].double() for j in range(bsz)])
  e6 = float((z.double() - ref).norm() / ref.norm())
  e32 = float((z32.double() - ref).norm() / ref.norm())
  print(f"n={n}: six-product {e6:.2e}  float32 {e32:.2e}")
  assert e6 <= 3e-7 and e6 <= 3.0 * e32 + 1e-7


@pytest.mark.parametrize("bsz,n,b", [(3, 512, 96), (1, 256, 64)])
def test_library_round_equals_python_issued_round(bsz, n, b, device):
  """ps_fd_round_f32 (orthonormalisation + Rayleigh-Ritz + control in one call) against the same steps
  issued one by one from Python (subspace._Planned.orthonormalize / rayleigh_ritz + fd_round_control):
  same kernels for the products and solvers; the glue (polish, symmetrisation, flip, residual norms)
  has its own kernels, so Ritz values agree to float32 rounding and the control decisions exactly."""
  from precondition_amd import subspace
  gen = torch.Generator(device=device).manual_seed(3 * n)
  cs = []
  for j in range(bsz):
    g = torch.randn((n, n // 4), generator=gen, device=device)
    cs.append(((g @ g.T) / n).contiguous())
  x0 = torch.randn((bsz, n, b), generator=gen, device=device)
  k = b - 31
  outs = []
  for lib_round in (False, True):
    x = x0.clone(); z = torch.empty_like(x); tmp = torch.empty_like(x)
    c16 = [K().to_bf16(c, split=True, tiled="frag3") for c in cs]
    pl = subspace._Planned(cs, x, z, tmp, c16x6=c16)
    if lib_round:
      theta, res, params, conv, summ = pl.round_call(k, 1e-5, 12)
    else:
      pl.orthonormalize()
      theta, res = pl.rayleigh_ritz()
      params, conv, summ = K().fd_round_control(theta, res, k, n, 1e-5, 12)
    torch.cuda.synchronize()
    outs.append([t.clone() for t in (theta, res, params, conv, summ, x, z)])
  a, c_ = outs
  top = float(a[0].abs().max())
  assert float((a[0] - c_[0]).abs().max()) <= 2e-6 * top                     # Ritz values
  assert float((a[1] - c_[1]).abs().max()) <= 1e-5 * top                     # residual norms
  assert torch.equal(a[3], c_[3]) and torch.equal(a[4][:3], c_[4][:3])       # converged flags, degrees
  assert float((a[2] - c_[2]).abs().max()) <= 1e-5 * max(float(a[2].abs().max()), 1e-30)
  # the Ritz vectors span the same space: compare the projectors on the leading k
  for j in range(bsz):
    pa, pb = a[5][j][:, :k], c_[5][j][:, :k]
    assert float((pa @ (pa.T @ pb) - pb).norm() / pb.norm()) < 1e-3


def test_eigh_solver_option_two_sided_reproduces_lapack_on_rank_deficient_input(device):
  """ps_options.eigh_solver: on a rank-deficient + ridge statistic (a noise cluster of eigenvalues around
  the ridge, where max(e, ridge)^(-1/p) has its kink).  The two-sided solver reproduces a FLOAT64-INTERNAL
  LAPACK result (oracle lapack="f64", the accuracy yardstick) to 1e-4; the one-sided solver is within 4 x
  of that yardstick's distance from the float64 root of a + ridge I; the reference's own arithmetic (float32
  ssyevd, oracle default) is an order of magnitude further out, and the default solver ("auto", the fast
  path kept) is at or below it."""
  rng = np.random.default_rng(11)
  n, p = 200, 2
  g = rng.standard_normal((n, n // 4))
  a = (g @ g.T).astype(np.float32); a = (a + a.T) / 2
  a64 = a.astype(np.float64)
  ridge = 1e-6 * np.linalg.eigvalsh(a64).max()
  w64, v64 = np.linalg.eigh(a64 + ridge * np.eye(n))
  truth = (v64 * np.maximum(w64, ridge) ** (-1.0 / p)) @ v64.T
  h_o, _ = orc.matrix_inverse_pth_root_eigh(a, p, padding_start=n, lapack="f64")
  h_s, _ = orc.matrix_inverse_pth_root_eigh(a, p, padding_start=n)          # float32 ssyevd
  t = torch.tensor(a, device=device)
  r0, _ = K().matrix_inverse_pth_root_batched([t], [p], [n], eigh=True)
  r1, _ = K().matrix_inverse_pth_root_batched([t], [p], [n], eigh=True, options={"eigh_solver": "one_sided"})
  r2, _ = K().matrix_inverse_pth_root_batched([t], [p], [n], eigh=True, options={"eigh_solver": "two_sided"})
  e_o, e_s = rel(h_o, truth), rel(h_s, truth)
  e0, e1, e2 = (rel(r[0].cpu().numpy(), truth) for r in (r0, r1, r2))
  print(f"vs float64: LAPACK f64-internal {e_o:.2e} ssyevd {e_s:.2e}  auto {e0:.2e} one-sided {e1:.2e}  two-sided {e2:.2e};"
        f"  two-sided vs f64-internal LAPACK {rel(r2[0].cpu().numpy(), h_o):.2e}")
  assert rel(r2[0].cpu().numpy(), h_o) < 1e-4
  assert e2 <= 1.1 * e_o and e1 <= 4.0 * e_o
  assert e_s > 3 * e_o and e0 <= 1.25 * e_s
  with pytest.raises(Exception):
    K().matrix_inverse_pth_root_batched([t], [p], [n], eigh=True, options={"eigh_solver": 7})


def test_fd_entry_points_refuse_what_they_do_not_support(device):
  """Shapes and argument combinations outside the fused FD kernels' domain are refused with a status
  (PS_EUNSUPPORTED / PS_EINVAL), never computed wrongly."""
  from precondition_amd import _lib
  n, b = 256, 96
  c = torch.randn((n, n), device=device)
  c = (c @ c.T).contiguous()
  y = torch.randn((1, n, b), device=device)
  params = torch.tensor([[0.1, 0.2, 0.3, 4.0]], device=device)
  frag = K().to_bf16(c, split=True, tiled="frag")
  planes = K().fd_filter_step(y, y, None, torch.empty_like(y), params, 1, frag=True)
  # a tile-blocked covariance is not a fragment-major one
  with pytest.raises(ValueError):
    K().fd_cy_step([K().to_bf16(c, split=True, tiled=True)], planes, y, y, torch.empty_like(y), None, params, 2)
  # step 1 has no product: the fused step starts at 2
  with pytest.raises(_lib.PsError):
    K().fd_cy_step([frag], planes, y, y, torch.empty_like(y), None, params, 1)
  # n must be a multiple of 128 for the fused step; the conversion itself needs multiples of 64
  c2 = torch.randn((192, 192), device=device)
  f2 = K().to_bf16(c2, split=True, tiled="frag")
  y2 = torch.randn((1, 192, 32), device=device)
  p2 = K().fd_filter_step(y2, y2, None, torch.empty_like(y2), params, 1, frag=True)
  with pytest.raises(_lib.PsError, match="-3"):
    K().fd_cy_step([f2], p2, y2, y2, torch.empty_like(y2), None, params, 2)
  with pytest.raises(ValueError):
    K().to_bf16(torch.randn((100, 128), device=device), split=True, tiled="frag3")
  # the six-product C x needs the third plane
  with pytest.raises(ValueError):
    K().fd_cx6([frag], y, torch.empty_like(y))
  # a symmetric product of two different operands is a caller error
  hi, lo = K().to_bf16(c, split=True)
  hi2, lo2 = K().to_bf16(c + 1.0, split=True)
  with pytest.raises(ValueError):
    K().gemm_bf16_grouped([((hi, lo), (hi2, lo2), torch.empty((n, n), device=device))], symmetric=True)
  # covariance update: shapes must match
  with pytest.raises(ValueError):
    K().fd_cov_update(torch.zeros((2, n, n), device=device), [c], 0.9)
