"""Dev-only NumPy prototype of the round-3 eigensolver of precondition_amd/csrc/eigh.hip:
one-sided (Hestenes) block Jacobi on the Cholesky factor L of D = A + ridge I.

  D = L L^T (float64 Cholesky of the float32 input, L rounded to float32)
  G <- L; sweeps over round-robin pairs of 64-wide block columns (I, J):
      P = [G_I G_J]^T [G_I G_J]   (128 x 128 Gram matrix, float32 MFMA on the GPU)
      Q = eigenvectors of P        (small symmetric eigenproblem; `inner` Jacobi sweeps or exact)
      [G_I G_J] <- [G_I G_J] Q
  converged when every scaled off-diagonal Gram entry |g_i.g_j| / (|g_i||g_j|) <= tol;
  eigenvectors of D = normalised columns of G, eigenvalues = squared column norms.
No eigenvector accumulation and no left-side update: 6 n^3 flops per sweep instead of 12 n^3.
Parameter study: sweeps to converge and root accuracy vs float64 / LAPACK float32 on the
bench's Wishart input, graded spectra and rank-deficient + ridge inputs."""
import sys, time
import numpy as np
F = np.float32


def round_robin(m):
  idx = list(range(m)); rounds = []
  for r in range(m - 1):
    rounds.append([(min(idx[k], idx[m - 1 - k]), max(idx[k], idx[m - 1 - k])) for k in range(m // 2)])
    idx = [idx[0]] + [idx[-1]] + idx[1:-1]
  return rounds


def pivot_q(P, inner, order_desc=True):
  """Orthogonal Q that diagonalises the Gram matrix P (exact LAPACK f64 when inner <= 0,
  else `inner` cyclic two-sided Jacobi sweeps in float32)."""
  if inner <= 0:
    w, q = np.linalg.eigh(P.astype(np.float64))
    if order_desc:
      q = q[:, ::-1]
    return q.astype(F)
  m = P.shape[0]; S = P.astype(F).copy(); Q = np.eye(m, dtype=F)
  rr = round_robin(m)
  for _ in range(inner):
    for pairs in rr:
      p = np.array([a for a, b in pairs]); q = np.array([b for a, b in pairs])
      app = S[p, p]; aqq = S[q, q]; apq = S[p, q]
      with np.errstate(all='ignore'):
        tau = (aqq - app) / (F(2) * apq)
        t = np.sign(tau) / (np.abs(tau) + np.sqrt(F(1) + tau * tau))
        t = np.where(tau == 0, F(1), t)
      small = np.abs(apq) <= F(1e-7) * np.sqrt(np.abs(app * aqq))
      t = np.where(small, F(0), t).astype(F)
      c = (F(1) / np.sqrt(F(1) + t * t)).astype(F); s = (t * c).astype(F)
      Sp = S[p, :].copy(); Sq = S[q, :].copy()
      S[p, :] = c[:, None] * Sp - s[:, None] * Sq; S[q, :] = s[:, None] * Sp + c[:, None] * Sq
      Sp = S[:, p].copy(); Sq = S[:, q].copy()
      S[:, p] = c[None, :] * Sp - s[None, :] * Sq; S[:, q] = s[None, :] * Sp + c[None, :] * Sq
      Qp = Q[:, p].copy(); Qq = Q[:, q].copy()
      Q[:, p] = c[None, :] * Qp - s[None, :] * Qq; Q[:, q] = s[None, :] * Qp + c[None, :] * Qq
  return Q


def onesided(D, b=64, inner=0, max_sweeps=24, tol=2e-6, verbose=True, transpose=False, sort=False, order='desc', cross_only=False):
  n = D.shape[0]
  L = np.linalg.cholesky(D.astype(np.float64)).astype(F)
  G = (L.T if transpose else L).copy()
  if sort:
    o = np.argsort(-np.sum(G.astype(np.float64) ** 2, axis=0)); G = G[:, o].copy()
  nb = n // b; rr = round_robin(nb)
  hist = []
  for sw in range(max_sweeps):
    worst = 0.0; rotated = 0
    for pairs in rr:
      for (I, J) in pairs:
        idx = np.r_[I * b:(I + 1) * b, J * b:(J + 1) * b]
        X = G[:, idx]
        P = (X.T @ X).astype(F)
        d = np.sqrt(np.maximum(np.diag(P), F(1e-37)))
        S = np.abs(P) / np.outer(d, d); np.fill_diagonal(S, 0)
        w = float(S.max()); worst = max(worst, w)
        if w <= tol:
          continue
        rotated += 1
        Q = pivot_q(P, inner, order_desc=(order == 'desc'))
        if order == 'rand':
          Q = Q[:, np.random.default_rng(sw * 1000 + I * 37 + J).permutation(Q.shape[1])]
        elif order == 'normsort' and inner > 0:
          Q = Q[:, np.argsort(-np.sum((X @ Q).astype(np.float64) ** 2, axis=0))]
        G[:, idx] = (X @ Q).astype(F)
    hist.append(worst)
    if verbose:
      print('sweep', sw, 'max scaled off %.3e' % worst, 'pairs rotated', rotated, flush=True)
    if worst <= tol:
      break
  e = np.sum(G.astype(np.float64) ** 2, axis=0)
  U = (G / np.sqrt(e)[None, :]).astype(F)
  return e, U, hist


def make(kind, n, seed=0):
  rng = np.random.default_rng(seed)
  if kind == 'wishart2':   # the cfg3 input: G [n, 2n]
    g = rng.standard_normal((n, 2 * n)).astype(F); A = (g @ g.T).astype(F)
  elif kind == 'wishart4':
    g = rng.standard_normal((n, 4 * n)).astype(F); A = (g @ g.T).astype(F)
  elif kind == 'square':
    g = rng.standard_normal((n, n)).astype(F); A = (g @ g.T).astype(F)
  elif kind == 'rankdef':
    g = rng.standard_normal((n, n // 4)).astype(F); A = (g @ g.T).astype(F)
  elif kind.startswith('graded'):
    cond = float(kind[6:])
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    w = cond ** (-np.arange(n) / (n - 1.0))
    A = ((q * w) @ q.T); A = (0.5 * (A + A.T)).astype(F)
  lam = np.linalg.eigvalsh(A.astype(np.float64)).max()
  D = A + F(1e-6 * lam) * np.eye(n, dtype=F)
  return D.astype(F), F(1e-6 * lam)


if __name__ == '__main__':
  kind = sys.argv[1]; n = int(sys.argv[2]); b = int(sys.argv[3]); inner = int(sys.argv[4])
  transpose = len(sys.argv) > 5 and sys.argv[5] == 'T'
  D, eps = make(kind, n)
  t = time.time()
  order = [a[6:] for a in sys.argv[5:] if a.startswith('order=')]
  e, U, h = onesided(D, b, inner, transpose='T' in sys.argv[5:], sort='S' in sys.argv[5:], order=order[0] if order else 'desc')
  print('time %.1f s, sweeps %d' % (time.time() - t, len(h)))
  p = 2
  w32, U32 = np.linalg.eigh(D)
  w64, U64 = np.linalg.eigh(D.astype(np.float64))
  f = lambda x, y: np.linalg.norm(x - y) / np.linalg.norm(y)
  root = lambda w, u: (u * np.maximum(w, eps) ** (-1.0 / p)) @ u.T
  ref64 = root(w64, U64)
  val = root(e.astype(F), U)
  print('root vs f64: this %.3e   lapack32 %.3e' % (f(val, ref64), f(root(w32, U32), ref64)))
  print('orth %.2e   resid max|U^T D U - diag(e)| %.2e (lapack32 %.2e)' % (
      np.abs(U.T @ U - np.eye(n)).max(), np.abs(U.T @ D @ U - np.diag(e.astype(F))).max(),
      np.abs(U32.T @ D @ U32 - np.diag(w32)).max()))
