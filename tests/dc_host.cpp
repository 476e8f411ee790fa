// dc_host.cpp — CPU driver of precondition_amd/csrc/dc_core.h (test infrastructure only).
//
// Runs the tridiagonal divide-and-conquer eigensolver on the host with exactly the scalar routines
// the HIP kernels of csrc/eigh_td.hip.h call (dc_tql2, dc_deflate, dc_secular_root, dc_zhat,
// dc_vec_rnorm) and the same data flow: float64 tridiagonal problem, float32 eigenvector matrices,
// merges as products [Q1 S_top; Q2 S_bot].  tests/test_dc_host.py compiles this file with g++ and
// checks it against numpy / scipy on the CPU; nothing in the product path uses it.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../precondition_amd/csrc/dc_core.h"
#include "../precondition_amd/csrc/dc_plan.h"

using namespace psdc;

namespace {

struct Work {
  int n;
  std::vector<double> d, e, D, Dn;
  std::vector<float> Q, Qn, S;
};

void merge_node(Work& wk, const DcNode& nd, double eps_defl, int* stats) {
  const int n = wk.n, r0 = nd.r0, m = nd.m, n1 = nd.n1;
  const double beta = wk.e[r0 + n1 - 1];
  const double rho = 2.0 * fabs(beta);
  const double sgn = beta < 0 ? -1.0 : 1.0;
  std::vector<double> d(m), z(m);
  const double rs2 = 1.0 / sqrt(2.0);
  for (int i = 0; i < m; ++i) {
    d[i] = wk.D[r0 + i];
    z[i] = i < n1 ? (double)wk.Q[(size_t)(r0 + n1 - 1) * n + r0 + i] * rs2
                  : sgn * (double)wk.Q[(size_t)(r0 + n1) * n + r0 + i] * rs2;
  }
  std::vector<int> perm(m);
  {
    int a = 0, b = n1, k = 0;
    while (a < n1 && b < m) perm[k++] = d[b] < d[a] ? b++ : a++;
    while (a < n1) perm[k++] = a++;
    while (b < m) perm[k++] = b++;
  }
  double dmax = 0, zmax = 0;
  for (int i = 0; i < m; ++i) { dmax = std::max(dmax, fabs(d[i])); zmax = std::max(zmax, fabs(z[i])); }
  const double tol = 8.0 * eps_defl * std::max(dmax, zmax);
  std::vector<double> dl(m), w(m), dfl_val(m), rc(m), rsn(m), mu(m), zh(m);
  std::vector<int> col(m), dfl_col(m), ra(m), rb(m), org(m);
  DcDeflateOut o;
  if (rho * zmax <= tol) {
    o.K = 0; o.nrot = 0; o.ndefl = m;
    for (int i = 0; i < m; ++i) { dfl_val[i] = d[perm[i]]; dfl_col[i] = perm[i]; }
  } else {
    o = dc_deflate(m, perm.data(), d.data(), z.data(), rho, tol, dl.data(), w.data(), col.data(),
                   dfl_val.data(), dfl_col.data(), ra.data(), rb.data(), rc.data(), rsn.data());
  }
  const int K = o.K;
  stats[0] += K; stats[1] += m; stats[2] += o.nrot;
  for (int j = 0; j < K; ++j) {
    const int it = dc_secular_root(K, j, dl.data(), w.data(), rho, &org[j], &mu[j]);
    stats[3] = std::max(stats[3], it);
    stats[4] += it;
  }
  for (int i = 0; i < K; ++i) zh[i] = dc_zhat(K, i, dl.data(), w.data(), org.data(), mu.data());
  // order of the deflated values (insertion sort: they arrive almost sorted)
  std::vector<int> dord(o.ndefl);
  for (int t = 0; t < o.ndefl; ++t) dord[t] = t;
  for (int t = 1; t < o.ndefl; ++t) {
    const int v = dord[t];
    int u = t - 1;
    while (u >= 0 && dfl_val[dord[u]] > dfl_val[v]) { dord[u + 1] = dord[u]; --u; }
    dord[u + 1] = v;
  }
  // merged order of roots and deflated values
  std::vector<int> pos_root(K), pos_defl(o.ndefl);
  {
    int a = 0, b = 0, k = 0;
    while (a < K || b < o.ndefl) {
      bool take_root;
      if (a >= K) take_root = false;
      else if (b >= o.ndefl) take_root = true;
      else take_root = (dl[org[a]] + mu[a]) <= dfl_val[dord[b]];
      if (take_root) { wk.Dn[r0 + k] = dl[org[a]] + mu[a]; pos_root[a++] = k++; }
      else { wk.Dn[r0 + k] = dfl_val[dord[b]]; pos_defl[dord[b++]] = k++; }
    }
  }
  // S
  float* S = wk.S.data();
  for (int i = 0; i < m; ++i) memset(S + (size_t)(r0 + i) * n + r0, 0, sizeof(float) * m);
  for (int j = 0; j < K; ++j) {
    const double rn = dc_vec_rnorm(K, j, dl.data(), zh.data(), org[j], mu[j]);
    for (int i = 0; i < K; ++i) {
      double dlt = dc_delta(dl.data(), i, org[j], mu[j]);
      if (dlt == 0.0) dlt = 1e-300;
      S[(size_t)(r0 + col[i]) * n + r0 + pos_root[j]] = (float)(zh[i] / dlt * rn);
    }
  }
  for (int t = 0; t < o.ndefl; ++t) S[(size_t)(r0 + dfl_col[t]) * n + r0 + pos_defl[t]] = 1.f;
  for (int t = o.nrot - 1; t >= 0; --t) {
    float* Sa = S + (size_t)(r0 + ra[t]) * n + r0;
    float* Sb = S + (size_t)(r0 + rb[t]) * n + r0;
    const double c = rc[t], s = rsn[t];
    for (int k = 0; k < m; ++k) {
      const double a = Sa[k], b = Sb[k];
      Sa[k] = (float)(c * a - s * b);
      Sb[k] = (float)(s * a + c * b);
    }
  }
  // Qn[block] = blockdiag(Q1, Q2) S   (float32 accumulation, as on the MFMA)
  for (int r = 0; r < m; ++r) {
    const int k0 = r < n1 ? 0 : n1, k1 = r < n1 ? n1 : m;
    float* out = wk.Qn.data() + (size_t)(r0 + r) * n + r0;
    for (int c = 0; c < m; ++c) out[c] = 0.f;
    for (int k = k0; k < k1; ++k) {
      const float q = wk.Q[(size_t)(r0 + r) * n + r0 + k];
      const float* Sr = S + (size_t)(r0 + k) * n + r0;
      for (int c = 0; c < m; ++c) out[c] += q * Sr[c];
    }
  }
}

}  // namespace

// d[n], e[n-1] -> evals[n] ascending, Z[n*n] row-major with eigenvectors in columns.
// stats[8]: sum K, sum m, rotations, max secular iterations, total secular iterations, tql2 failures
extern "C" int dc_host_eigh(int n, const double* d_in, const double* e_in, float* Z, double* evals,
                            double eps_defl, int* stats) {
  for (int i = 0; i < 8; ++i) stats[i] = 0;
  Work wk;
  wk.n = n;
  wk.d.assign(d_in, d_in + n);
  wk.e.assign(n, 0.0);
  for (int i = 0; i + 1 < n; ++i) wk.e[i] = e_in[i];
  double scale = 0;
  for (int i = 0; i < n; ++i) scale = std::max(scale, std::max(fabs(wk.d[i]), fabs(wk.e[i])));
  if (scale == 0) scale = 1;
  for (int i = 0; i < n; ++i) { wk.d[i] /= scale; wk.e[i] /= scale; }
  std::vector<DcNode> nodes;
  int height = 0;
  dc_make_plan(n, nodes, &height);
  for (const DcNode& nd : nodes)
    if (nd.n1 > 0) {
      const int k = nd.r0 + nd.n1 - 1;
      const double b = fabs(wk.e[k]);
      wk.d[k] -= b;
      wk.d[k + 1] -= b;
    }
  wk.D.assign(n, 0.0);
  wk.Dn.assign(n, 0.0);
  wk.Q.assign((size_t)n * n, 0.f);
  wk.Qn.assign((size_t)n * n, 0.f);
  wk.S.assign((size_t)n * n, 0.f);
  for (const DcNode& nd : nodes) {
    if (nd.n1 != 0) continue;
    const int m = nd.m, r0 = nd.r0;
    std::vector<double> dd(wk.d.begin() + r0, wk.d.begin() + r0 + m), ee(m, 0.0), zz((size_t)m * m, 0.0);
    for (int i = 0; i + 1 < m; ++i) ee[i] = wk.e[r0 + i];
    for (int i = 0; i < m; ++i) zz[(size_t)i * m + i] = 1.0;
    stats[5] += dc_tql2(m, dd.data(), ee.data(), zz.data(), m, 0, 1, true);
    std::vector<int> ord(m);
    for (int i = 0; i < m; ++i) ord[i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return dd[a] < dd[b]; });
    for (int c = 0; c < m; ++c) {
      wk.D[r0 + c] = dd[ord[c]];
      for (int r = 0; r < m; ++r) wk.Q[(size_t)(r0 + r) * n + r0 + c] = (float)zz[(size_t)r * m + ord[c]];
    }
  }
  for (int h = 1; h <= height; ++h) {
    // nodes below this height that are not merged at this level keep their blocks
    for (const DcNode& nd : nodes) {
      if (nd.height == h) merge_node(wk, nd, eps_defl, stats);
    }
    for (const DcNode& nd : nodes) {
      if (nd.height != h) continue;
      for (int r = 0; r < nd.m; ++r) {
        memcpy(wk.Q.data() + (size_t)(nd.r0 + r) * n + nd.r0, wk.Qn.data() + (size_t)(nd.r0 + r) * n + nd.r0,
               sizeof(float) * nd.m);
        wk.D[nd.r0 + r] = wk.Dn[nd.r0 + r];
      }
    }
  }
  for (int i = 0; i < n; ++i) evals[i] = wk.D[i] * scale;
  memcpy(Z, wk.Q.data(), sizeof(float) * (size_t)n * n);
  return 0;
}
