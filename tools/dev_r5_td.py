"""Dev (round 5): stage-by-stage check of the tridiagonalisation + divide-and-conquer eigensolver
(csrc/eigh_td.hip.h) against NumPy float64.

  stage 1 (PS_EIGH_TD_STAGE=1): the output vectors are Q of D = Q T Q^T and the values diag(T):
      Q orthogonal, Q^T D Q tridiagonal with that diagonal;
  full: eigenvalues, residual |A Z - Z L|, orthogonality, on random / Wishart / rank-deficient /
      graded inputs and odd sizes.
"""
import os
os.environ.setdefault("PS_DEV_ENV", "1")
import sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import kernels as K

dev = torch.device("cuda:0")


def make(n, kind, seed):
  rng = np.random.default_rng(seed)
  if kind == "randsym":
    g = rng.standard_normal((n, n)); a = (g + g.T) / 2
  elif kind == "wishart":
    g = rng.standard_normal((n, 2 * n)); a = g @ g.T
  elif kind == "lowrank":
    g = rng.standard_normal((n, n // 4)); a = g @ g.T
    a = a + 1e-6 * np.linalg.eigvalsh(a).max() * np.eye(n)
  else:
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    a = (q * 10.0 ** rng.uniform(-4, 2, n)) @ q.T
  return ((a + a.T) / 2).astype(np.float32)


def run(mats):
  ts = [torch.tensor(a, device=dev) for a in mats]
  e, v = K.eigh_batched(ts, options={"eigh_solver": "tridiagonal"})   # plain eigh keeps the fast path's result only on request
  torch.cuda.synchronize()
  return [x.cpu().numpy().astype(np.float64) for x in e], [x.cpu().numpy().astype(np.float64) for x in v]


def stage1():
  os.environ["PS_EIGH_TD_STAGE"] = "1"
  for n, kind in ((129, "randsym"), (200, "wishart"), (256, "randsym"), (300, "graded"), (512, "wishart"),
                  (1000, "randsym")):
    a = make(n, kind, n)
    # eigh_batched sorts by eigenvalue (= diag(T) here): undo nothing, the check is permutation invariant
    e, v = run([a])
    q = v[0]; d = e[0]
    a64 = a.astype(np.float64)
    t = q.T @ a64 @ q
    nrm = np.abs(a64).max() * n ** 0.5
    orth = np.abs(q.T @ q - np.eye(n)).max()
    # the columns were sorted by diag(T): sort destroys tridiagonal structure; compare invariants instead
    ev_t = np.linalg.eigvalsh((t + t.T) / 2)
    ev_a = np.linalg.eigvalsh(a64)
    print(f"stage1 n={n:5d} {kind:8s} orth={orth:.2e} diag_err={np.abs(np.diag(t) - d).max() / nrm:.2e} "
          f"spectrum_err={np.abs(ev_t - ev_a).max() / np.abs(ev_a).max():.2e} finite={np.isfinite(q).all()}", flush=True)
  os.environ.pop("PS_EIGH_TD_STAGE")


def full():
  for n, kind in ((129, "randsym"), (200, "wishart"), (256, "randsym"), (300, "graded"), (512, "wishart"),
                  (520, "lowrank"), (1000, "randsym"), (1024, "wishart"), (2048, "lowrank")):
    a = make(n, kind, n)
    e, v = run([a])
    z = v[0]; lam = e[0]
    a64 = a.astype(np.float64)
    ref = np.linalg.eigvalsh(a64)
    nrm = np.abs(ref).max()
    orth = np.abs(z.T @ z - np.eye(n)).max()
    res = np.abs(a64 @ z - z * lam).max() / nrm
    print(f"full   n={n:5d} {kind:8s} ev_err={np.abs(lam - ref).max() / nrm:.2e} orth={orth:.2e} res={res:.2e}", flush=True)
  # a mixed batch
  mats = [make(n, k, 7 * n) for n, k in ((300, "wishart"), (640, "randsym"), (130, "graded"), (1024, "lowrank"))]
  e, v = run(mats)
  for a, lam, z in zip(mats, e, v):
    a64 = a.astype(np.float64); ref = np.linalg.eigvalsh(a64); nrm = np.abs(ref).max(); n = a.shape[0]
    print(f"mixed  n={n:5d} ev_err={np.abs(lam - ref).max() / nrm:.2e} orth={np.abs(z.T @ z - np.eye(n)).max():.2e} "
          f"res={np.abs(a64 @ z - z * lam).max() / nrm:.2e}", flush=True)


def roots():
  for n, kind, p in ((169, "graded", 4), (512, "graded", 2), (260, "lowrank", 2), (1024, "graded", 4),
                     (2048, "lowrank", 2), (1000, "wishart", 2), (2048, "wishart", 2)):
    a = make(n, kind, n + p) if kind != "lowrank" else None
    if a is None:
      rng = np.random.default_rng(n + p); g = rng.standard_normal((n, n // 4)); a = (g @ g.T).astype(np.float32)
    a64 = a.astype(np.float64)
    ridge = 1e-6 * np.linalg.eigvalsh(a64).max()
    w, v = np.linalg.eigh(a64 + ridge * np.eye(n))
    f = lambda e: np.maximum(e, ridge) ** (-1.0 / p)
    truth = (v * f(w)) @ v.T
    d32 = (a + np.float32(ridge) * np.eye(n, dtype=np.float32)).astype(np.float32)
    wl, vl = np.linalg.eigh(d32)
    lap = (vl.astype(np.float64) * f(wl.astype(np.float64))) @ vl.T.astype(np.float64)
    tn = np.linalg.norm(truth)
    line = f"root   n={n:5d} {kind:8s} p={p} lapack32={np.linalg.norm(lap - truth) / tn:.2e}"
    for name, td in (("td", "1"), ("jacobi", "0")):
      os.environ["PS_EIGH_TD"] = td
      r, m = K.matrix_inverse_pth_root_batched([torch.tensor(a, device=dev)], [p], [n], eigh=True)
      got = r[0].cpu().numpy().astype(np.float64)
      line += f" {name}={np.linalg.norm(got - truth) / tn:.2e} (err metric {m[0, 0].item():.2e})"
    print(line, flush=True)
  os.environ.pop("PS_EIGH_TD")


def bench():
  for nb, n in ((64, 2048), (64, 1024), (256, 512)):
    gen = torch.Generator(device=dev).manual_seed(n)
    stats = torch.zeros((nb, n, n), device=dev)
    for b0 in range(0, nb, 8):
      g = torch.randn((8, n, 2 * n), generator=gen, device=dev)
      K.stats_update_grouped([(g[i], 0, stats[b0 + i], stats[b0 + i]) for i in range(8)], 0.0, 1.0)
    torch.cuda.synchronize()
    out = torch.empty_like(stats)
    for td in ("1", "0"):
      os.environ["PS_EIGH_TD"] = td
      for rep in range(3):
        t0 = time.perf_counter()
        _, m = K.matrix_inverse_pth_root_batched(list(stats.unbind(0)), [2] * nb, [n] * nb, eigh=True, out=list(out.unbind(0)))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
      print(f"bench eigh {nb}x{n} td={td}: {dt * 1e3:.1f} ms, err max {m[:, 0].max().item():.2e}", flush=True)
  os.environ.pop("PS_EIGH_TD")




def streams():
  """Stream groups of the reduction: 64 x 2048^2 and 64 x 1024^2, best of 5."""
  for nb, n in ((64, 2048), (64, 1024), (256, 512)):
    gen = torch.Generator(device=dev).manual_seed(n)
    stats = torch.zeros((nb, n, n), device=dev)
    for b0 in range(0, nb, 8):
      g = torch.randn((8, n, 2 * n), generator=gen, device=dev)
      K.stats_update_grouped([(g[i], 0, stats[b0 + i], stats[b0 + i]) for i in range(8)], 0.0, 1.0)
    torch.cuda.synchronize()
    out = torch.empty_like(stats)
    for sg in os.environ.get("SG_LIST", "1,2,4,6,8").split(","):
      os.environ["PS_EIGH_TD_STREAMS"] = sg
      best = 1e9
      for rep in range(6):
        t0 = time.perf_counter()
        _, m = K.matrix_inverse_pth_root_batched(list(stats.unbind(0)), [2] * nb, [n] * nb, eigh=True, out=list(out.unbind(0)))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if rep: best = min(best, dt)
      print(f"streams {nb}x{n} groups={sg}: {best * 1e3:.1f} ms, err max {m[:, 0].max().item():.2e}", flush=True)
  os.environ.pop("PS_EIGH_TD_STREAMS")


def special():
  """Structured inputs that stress deflation, tau = 0 reflectors and the secular solver."""
  rng = np.random.default_rng(5)
  def cases(n):
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    tri = np.diag(rng.standard_normal(n)) + np.diag(rng.standard_normal(n - 1), 1); tri = tri + np.triu(tri, 1).T
    wilk = np.diag(np.abs(np.arange(n) - n // 2).astype(np.float64)) + np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1)
    blk = np.zeros((n, n)); h = n // 3
    for lo, hi in ((0, h), (h, 2 * h), (2 * h, n)):
      g = rng.standard_normal((hi - lo, hi - lo)); blk[lo:hi, lo:hi] = g + g.T
    arrow = np.diag(np.linspace(1, 2, n)); arrow[0, :] = 0.1; arrow[:, 0] = 0.1; arrow[0, 0] = 3
    x = rng.standard_normal((n, 1))
    yield "identity", np.eye(n)
    yield "zero", np.zeros((n, n))
    yield "2I+1e-6noise", 2 * np.eye(n) + 1e-6 * (lambda g: g + g.T)(rng.standard_normal((n, n)))
    yield "diag_repeats", np.diag(np.repeat([1.0, 2.0, 3.0, 5.0], (n + 3) // 4)[:n])
    yield "rotated_repeats", (q * np.repeat([1.0, 2.0, 3.0, 5.0], (n + 3) // 4)[:n]) @ q.T
    yield "rank1", x @ x.T
    yield "rank1+I", x @ x.T + np.eye(n)
    yield "tridiagonal", tri
    yield "wilkinson", wilk
    yield "blockdiag", blk
    yield "arrow", arrow
    yield "neg_definite", -(lambda g: g @ g.T)(rng.standard_normal((n, 2 * n)))
    yield "tiny_scale", 1e-20 * (lambda g: g + g.T)(rng.standard_normal((n, n)))
    yield "huge_scale", 1e18 * (lambda g: g + g.T)(rng.standard_normal((n, n)))
    yield "ones", np.ones((n, n))
  for n in (130, 257, 600):
    names, mats = zip(*cases(n))
    mats = [((m + m.T) / 2).astype(np.float32) for m in mats]
    e, v = run(list(mats))
    for name, a, lam, z in zip(names, mats, e, v):
      a64 = a.astype(np.float64); ref = np.linalg.eigvalsh(a64); nrm = max(np.abs(ref).max(), 1e-300)
      ok = np.isfinite(z).all() and np.isfinite(lam).all()
      print(f"special n={n:4d} {name:16s} ev_err={np.abs(lam - ref).max() / nrm:.2e} "
            f"orth={np.abs(z.T @ z - np.eye(n)).max():.2e} res={np.abs(a64 @ z - z * lam).max() / nrm:.2e} finite={ok}", flush=True)


def debug129():
  """stage 1 gives Q (T = Q^T A Q), stage 2 gives Z_T: which eigenpairs of T are off?"""
  for n, kind in ((129, "randsym"), (130, "randsym"), (129, "wishart"), (161, "randsym")):
    a = make(n, kind, n)
    a64 = a.astype(np.float64)
    os.environ["PS_EIGH_TD_STAGE"] = "1"
    ts = [torch.tensor(a, device=dev)]
    import ctypes
    e1, v1 = K.eigh_batched(ts); torch.cuda.synchronize()
    q = v1[0].cpu().numpy().astype(np.float64)
    t = q.T @ a64 @ q   # columns sorted by diag: a permuted tridiagonal matrix
    os.environ["PS_EIGH_TD_STAGE"] = "2"; os.environ["PS_EIGH_CJ_REFINE"] = "0"   # raw D&C values
    e2, v2 = K.eigh_batched(ts); torch.cuda.synchronize()
    os.environ.pop("PS_EIGH_TD_STAGE"); os.environ.pop("PS_EIGH_CJ_REFINE")
    lam = e2[0].cpu().numpy().astype(np.float64)
    ref = np.linalg.eigvalsh(a64)
    bad = np.nonzero(np.abs(lam - ref) > 1e-5 * np.abs(ref).max())[0]
    print(f"debug n={n} {kind}: ev_err={np.abs(lam - ref).max() / np.abs(ref).max():.2e} bad indices {bad[:20]} of {n}", flush=True)
    zt = v2[0].cpu().numpy().astype(np.float64)
    print("   orth of Z_T", np.abs(zt.T @ zt - np.eye(n)).max(), flush=True)


if __name__ == "__main__":
  what = sys.argv[1:] or ["stage1", "full", "roots", "bench"]
  for w in what:
    globals()[w]()
