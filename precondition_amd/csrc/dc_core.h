// dc_core.h — scalar logic of the symmetric tridiagonal divide-and-conquer eigensolver
// (Cuppen's method with Gu-Eisenstat vectors; the algorithm class of LAPACK's ssyevd / sstedc, which
// is what the reference's jnp.linalg.eigh runs on its CPU path, DS:1007).
//
// Everything here is plain C++ that compiles for the host (g++, CPU tests: tests/test_dc_host.py
// drives it through tests/dc_host.cpp) and for the device (hipcc, csrc/eigh_td.hip.h calls the same
// functions from its kernels, one thread / one wavefront per unit of work).  No reference code is
// restated here: the reference has no eigensolver of its own (jnp.linalg.eigh is third-party).
//
// Arithmetic: float64 for the tridiagonal problem (eigenvalues, secular equation, Loewner weights,
// eigenvector entries); the eigenvector matrices themselves are float32 (MFMA products).  The
// deflation tolerance is a parameter: with float32 input (the tridiagonal matrix carries the
// rounding of a float32 Householder reduction) it is set at float32 level, as LAPACK's single
// precision solver does, which is also what keeps clusters (rank-deficient statistics + ridge) cheap.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define DC_HD __host__ __device__ __forceinline__
#else
#define DC_HD static inline
#endif

namespace psdc {

constexpr double DC_EPS64 = 2.220446049250313e-16;
constexpr int DC_LEAF = 32;  // largest leaf handed to the QL iteration

// ---- partition tree ---------------------------------------------------------------------------
// A node covers rows/columns [r0, r0 + m) of the tridiagonal matrix; n1 = 0 marks a leaf, else its
// children are [r0, r0 + n1) and [r0 + n1, r0 + m).  height: leaves 0, parents 1 + max(children).
struct DcNode {
  int r0, m, n1, height;
};

// ---- leaf: implicit-shift QL with eigenvector accumulation (the classic tql2 recurrence) ----------
// d[0..n): diagonal in, eigenvalues out (unsorted); e[0..n): e[i] couples i and i+1, e[n-1] is
// ignored.  The caller owns rows row0, row0 + rstep, ... of Z (z[r * ldz + c], identity on entry).
// On the device one wavefront runs this with d / e in LDS shared by its lanes: every lane executes
// the same scalar recurrence (uniform control flow), only `writer` lanes store to d / e, and each
// lane rotates its own rows of Z.  Returns the number of eigenvalues that hit the iteration cap.
template <typename DT, typename ZT>
DC_HD int dc_tql2(int n, DT* d, DT* e, ZT* z, int ldz, int row0, int rstep, bool writer) {
  int bad = 0;
  if (n <= 1) return 0;
  if (writer) e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  for (int l = 0; l < n; ++l) {
    const double t = fabs((double)d[l]) + fabs((double)e[l]);
    tst1 = t > tst1 ? t : tst1;
    int m = l;
    while (m < n - 1) {
      if (fabs((double)e[m]) <= DC_EPS64 * tst1) break;
      ++m;
    }
    if (m > l) {
      int iter = 0;
      for (;;) {
        ++iter;
        double g = d[l];
        const double el = e[l];
        double p = ((double)d[l + 1] - g) / (2.0 * el);
        double r = hypot(p, 1.0);
        if (p < 0) r = -r;
        const double dl0 = el / (p + r);
        const double dl1 = el * (p + r);
        const double h = g - dl0;
        if (writer) {
          d[l] = dl0;
          d[l + 1] = dl1;
          for (int i = l + 2; i < n; ++i) d[i] = (double)d[i] - h;
        }
        f += h;
        p = d[m];
        double c = 1.0, c2 = 1.0, c3 = 1.0, s = 0.0, s2 = 0.0;
        const double el1 = e[l + 1];
        for (int i = m - 1; i >= l; --i) {
          c3 = c2;
          c2 = c;
          s2 = s;
          const double ei = e[i];
          g = c * ei;
          const double hh = c * p;
          r = hypot(p, ei);
          const double enew = s * r;
          s = ei / r;
          c = p / r;
          const double di = d[i];
          p = c * di - s * g;
          const double dnew = hh + s * (c * g + s * di);
          if (writer) {
            e[i + 1] = enew;
            d[i + 1] = dnew;
          }
          for (int k = row0; k < n; k += rstep) {
            const double zk1 = z[k * ldz + i + 1], zk0 = z[k * ldz + i];
            z[k * ldz + i + 1] = (ZT)(s * zk0 + c * zk1);
            z[k * ldz + i] = (ZT)(c * zk0 - s * zk1);
          }
        }
        p = -s * s2 * c3 * el1 * (double)e[l] / dl1;
        const double enl = s * p, dnl = c * p;
        if (writer) {
          e[l] = enl;
          d[l] = dnl;
        }
        if (!(fabs(enl) > DC_EPS64 * tst1)) break;
        if (iter >= 60) { ++bad; break; }
      }
    }
    if (writer) {
      d[l] = (double)d[l] + f;
      e[l] = 0.0;
    }
  }
  return bad;
}

// ---- deflation (one thread per merge node) ----------------------------------------------------------
// In:  m, the merged ascending order perm[0..m) (indices into d / z), d[m], z[m] (z normalised to
//      unit length, already multiplied by 1/sqrt(2) and the sign of the coupling), rho > 0, tol.
// Out: K non-deflated entries: dl[k] (ascending), w[k], col[k] (index of the column of the current
//      eigenvector matrix);  ndefl deflated entries: dfl_val[t], dfl_col[t] (in order of deflation);
//      nrot plane rotations rot_a[t], rot_b[t], rot_c[t], rot_s[t] acting on columns (a, b) of the
//      eigenvector matrix: (q_a, q_b) <- (c q_a + s q_b, c q_b - s q_a), in this order.
// d and z are updated in place (rotated pairs).  Returns K.
struct DcDeflateOut {
  int K, ndefl, nrot;
};

DC_HD DcDeflateOut dc_deflate(int m, const int* perm, double* d, double* z, double rho, double tol,
                              double* dl, double* w, int* col, double* dfl_val, int* dfl_col,
                              int* rot_a, int* rot_b, double* rot_c, double* rot_s) {
  DcDeflateOut o;
  o.K = 0; o.ndefl = 0; o.nrot = 0;
  int pj = -1;
  for (int jj = 0; jj < m; ++jj) {
    const int nj = perm[jj];
    if (rho * fabs(z[nj]) <= tol) {  // negligible coupling: (d, e_nj) is an eigenpair already
      dfl_val[o.ndefl] = d[nj];
      dfl_col[o.ndefl] = nj;
      ++o.ndefl;
      continue;
    }
    if (pj < 0) { pj = nj; continue; }
    double s = z[pj], c = z[nj];
    const double tau = hypot(c, s);
    const double t = d[nj] - d[pj];
    c /= tau;
    s = -s / tau;
    if (fabs(t * c * s) <= tol) {  // two close poles: rotate the coupling of pj into nj
      z[nj] = tau;
      z[pj] = 0.0;
      rot_a[o.nrot] = pj; rot_b[o.nrot] = nj; rot_c[o.nrot] = c; rot_s[o.nrot] = s;
      ++o.nrot;
      const double dp = d[pj] * c * c + d[nj] * s * s;
      d[nj] = d[pj] * s * s + d[nj] * c * c;
      d[pj] = dp;
      dfl_val[o.ndefl] = dp;
      dfl_col[o.ndefl] = pj;
      ++o.ndefl;
      pj = nj;
    } else {
      dl[o.K] = d[pj]; w[o.K] = z[pj]; col[o.K] = pj;
      ++o.K;
      pj = nj;
    }
  }
  if (pj >= 0) {
    dl[o.K] = d[pj]; w[o.K] = z[pj]; col[o.K] = pj;
    ++o.K;
  }
  return o;
}

// delta(i, j) = dl[i] - lambda_j with lambda_j = dl[org_j] + mu_j, evaluated the same way wherever it
// is used (secular iteration, Loewner weights, eigenvector entries): that consistency, not the
// accuracy of mu itself, is what makes the computed vectors orthogonal (Gu & Eisenstat).
DC_HD double dc_delta(const double* dl, int i, int org, double mu) { return (dl[i] - dl[org]) - mu; }

// ---- secular equation: root j of 1/rho + sum_i w_i^2 / (dl_i - lambda) = 0 in (dl_j, dl_{j+1}) ---------
// (the last root lies in (dl_{K-1}, dl_{K-1} + rho |w|^2]).  Returns the origin pole in *org and the
// offset in *mu.  Rational ("middle way") iteration safeguarded by the bracket; float64.
// Return value: iterations used (> DC_SEC_MAXIT: not converged to the function tolerance; the bracket
// result is still returned).
constexpr int DC_SEC_MAXIT = 100;

DC_HD void dc_secular_eval(int K, int j, const double* dl, const double* w, int org, double x,
                           double rhoinv, double& f, double& psi, double& dpsi, double& phi,
                           double& dphi, double& asum) {
  psi = 0.0; dpsi = 0.0; phi = 0.0; dphi = 0.0; asum = rhoinv;
  const double dorg = dl[org];
  // one loop of uniform length (the lanes of a wavefront solve for different j)
  for (int i = 0; i < K; ++i) {
    const double wi = w[i];
    const double t = wi / ((dl[i] - dorg) - x);
    const double wt = wi * t, tt = t * t;
    const bool left = i <= j;
    psi += left ? wt : 0.0;
    dpsi += left ? tt : 0.0;
    phi += left ? 0.0 : wt;
    dphi += left ? 0.0 : tt;
    asum += fabs(wt);
  }
  f = rhoinv + psi + phi;
}

DC_HD int dc_secular_root(int K, int j, const double* dl, const double* w, double rho, int* org_out,
                          double* mu_out) {
  const double rhoinv = 1.0 / rho;
  double f, psi, dpsi, phi, dphi, asum;
  int org;
  double lo, hi;     // bracket of mu (root strictly inside, poles at the ends are excluded)
  double p1, p2;     // the two poles next to the root, in shifted coordinates (p2 unused for the last)
  const bool last = (j == K - 1);
  double x;
  if (!last) {
    const double gap = dl[j + 1] - dl[j];
    // sign of f at the midpoint decides which pole is closer to the root
    dc_secular_eval(K, j, dl, w, j, 0.5 * gap, rhoinv, f, psi, dpsi, phi, dphi, asum);
    if (f >= 0.0) {  // root in the left half: origin dl[j]
      org = j; lo = 0.0; hi = 0.5 * gap; p1 = 0.0; p2 = gap;
      x = 0.5 * gap;
    } else {
      org = j + 1; lo = -0.5 * gap; hi = 0.0; p1 = -gap; p2 = 0.0;
      x = -0.5 * gap;
      dc_secular_eval(K, j, dl, w, org, x, rhoinv, f, psi, dpsi, phi, dphi, asum);
    }
  } else {
    double ww = 0.0;
    for (int i = 0; i < K; ++i) ww += w[i] * w[i];
    org = j; lo = 0.0; hi = rho * ww; p1 = 0.0; p2 = 2.0 * hi + 1.0;  // no pole on the right
    if (!(hi > 0.0)) { *org_out = org; *mu_out = 0.0; return 0; }
    x = 0.5 * hi;
    dc_secular_eval(K, j, dl, w, org, x, rhoinv, f, psi, dpsi, phi, dphi, asum);
    // f(hi) >= 0 always; keep hi as the upper end
  }
  const double sk = 4.0 + sqrt((double)K);
  int it = 0;
  for (; it < DC_SEC_MAXIT; ++it) {
    if (fabs(f) <= DC_EPS64 * sk * asum) break;
    if (f < 0.0) lo = x; else hi = x;
    if (!(hi - lo > 2.0 * DC_EPS64 * fmax(fabs(lo), fabs(hi)))) break;
    // middle way: psi ~ s + a / (p1 - y), phi ~ r + b / (p2 - y), matched in value and slope at x
    const double D1 = p1 - x, D2 = p2 - x;  // D1 < 0 < D2
    const double a = dpsi * D1 * D1, b = last ? 0.0 : dphi * D2 * D2;
    const double c = last ? (f - dpsi * D1) : (f - dpsi * D1 - dphi * D2);
    // c (D1 - eta)(D2 - eta) + a (D2 - eta) + b (D1 - eta) = 0
    double eta;
    bool ok = false;
    if (last) {
      // c + a / (D1 - eta) = 0
      if (c > 0.0) { eta = D1 + a / c; ok = true; }
    } else {
      const double A = c, B = -(c * (D1 + D2) + a + b), C = D1 * D2 * f;
      if (A == 0.0) {
        if (B != 0.0) { eta = -C / B; ok = true; }
      } else {
        double disc = B * B - 4.0 * A * C;
        if (disc < 0.0) disc = 0.0;
        const double sq = sqrt(disc);
        const double q = -0.5 * (B + (B >= 0.0 ? sq : -sq));
        const double e1 = q / A, e2 = (q != 0.0) ? C / q : e1;
        // the root that keeps x + eta between the poles (and inside the bracket)
        const bool in1 = (x + e1 > lo) && (x + e1 < hi);
        const bool in2 = (x + e2 > lo) && (x + e2 < hi);
        if (in1 && in2) { eta = fabs(e1) < fabs(e2) ? e1 : e2; ok = true; }
        else if (in1) { eta = e1; ok = true; }
        else if (in2) { eta = e2; ok = true; }
      }
    }
    double xn = ok ? x + eta : 0.5 * (lo + hi);
    if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
    // the model converges in ~4 steps (15 at worst on the test spectra); past 12 steps every fourth
    // one is a plain bisection, which bounds the iteration count whatever the model does
    if (it >= 12 && (it & 3) == 3) xn = 0.5 * (lo + hi);
    x = xn;
    dc_secular_eval(K, j, dl, w, org, x, rhoinv, f, psi, dpsi, phi, dphi, asum);
  }
  *org_out = org;
  *mu_out = x;
  return it;
}

// ---- Loewner weights: the z for which the computed lambdas are the exact eigenvalues ----------------
// zhat_i = sign(w_i) sqrt( prod_j (lambda_j - dl_i) / prod_{j != i} (dl_j - dl_i) )   (overall 1/rho
// dropped: the vectors are normalised afterwards).
DC_HD double dc_zhat(int K, int i, const double* dl, const double* w, const int* org,
                     const double* mu) {
  double p = -dc_delta(dl, i, org[i], mu[i]);  // lambda_i - dl_i > 0
  for (int j = 0; j < K; ++j) {
    if (j == i) continue;
    p *= dc_delta(dl, i, org[j], mu[j]) / (dl[i] - dl[j]);
  }
  const double r = sqrt(fabs(p));
  return w[i] < 0.0 ? -r : r;
}

// 1 / |x_j| for x_j(i) = zhat_i / (dl_i - lambda_j)
DC_HD double dc_vec_rnorm(int K, int j, const double* dl, const double* zhat, int org, double mu) {
  double ss = 0.0;
  for (int i = 0; i < K; ++i) {
    double dlt = dc_delta(dl, i, org, mu);
    if (dlt == 0.0) dlt = 1e-300;
    const double x = zhat[i] / dlt;
    ss += x * x;
  }
  return 1.0 / sqrt(ss);
}

}  // namespace psdc
