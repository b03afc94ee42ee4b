// omc_cpu_ref.cpp -- TEST / BASELINE INFRASTRUCTURE, not the product.  A compiled, single-threaded restatement of the node relaxation that
// oracle/omc_oracle.py:sdp_relaxation states in numpy (the same consensus ADMM the HIP engine runs: column prox, two spectral projections,
// weighted projection on the node's linear rows by NNQP, Fenchel / Lagrangian certificate every `check_every` iterations).  It exists so that
// bench.py's cpu_baseline can time the algorithm the way the reference pins its solver: ONE thread per node (MSK_IPAR_NUM_THREADS = 1,
// OMC.jl:1486), and -- with OpenMP over independent nodes -- on all host cores.  Only tests/, bench.py's cpu_baseline leg and
// __graft_entry__ may load it (oracle/Makefile builds oracle/libomc_cpu_ref.so).  Row-major (numpy C order) everywhere.
//
// What it follows, line by line: oracle/omc_oracle.py:621-810 (loop), :475-515 (_prox_columns), :290-322 (nnqp), :546-571 (dual_bound_from),
// :360-376 (Instance.f_value).  Not restated: Anderson acceleration (off by default), warm starts, the adaptive-penalty option (off by default).
// The rows of the node (OMC.jl:1558-1685) are built by the Python oracle (build_rows) and handed over as dense coefficient arrays.
// The symmetric eigensolver is the classical Householder tridiagonalisation + implicit QL iteration (EISPACK tred2 / tql2 as restated in the
// public-domain JAMA package), written out here because the image has no LAPACK headers.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {
typedef std::vector<double> vec;

// ---- symmetric eigendecomposition: V (n x n, row-major) holds the matrix on entry, the eigenvectors (columns) on exit; d ascending ----
void tred2(int n, double* V, double* d, double* e) {
  for (int j = 0; j < n; ++j) d[j] = V[(size_t)(n - 1) * n + j];
  for (int i = n - 1; i > 0; --i) {
    double scale = 0.0, h = 0.0;
    for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
    if (scale == 0.0) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; ++j) { d[j] = V[(size_t)(i - 1) * n + j]; V[(size_t)i * n + j] = 0.0; V[(size_t)j * n + i] = 0.0; }
    } else {
      for (int k = 0; k < i; ++k) { d[k] /= scale; h += d[k] * d[k]; }
      double f = d[i - 1], g = std::sqrt(h);
      if (f > 0) g = -g;
      e[i] = scale * g; h -= f * g; d[i - 1] = f - g;
      for (int j = 0; j < i; ++j) e[j] = 0.0;
      for (int j = 0; j < i; ++j) {
        f = d[j]; V[(size_t)j * n + i] = f; g = e[j] + V[(size_t)j * n + j] * f;
        for (int k = j + 1; k <= i - 1; ++k) { g += V[(size_t)k * n + j] * d[k]; e[k] += V[(size_t)k * n + j] * f; }
        e[j] = g;
      }
      f = 0.0;
      for (int j = 0; j < i; ++j) { e[j] /= h; f += e[j] * d[j]; }
      const double hh = f / (h + h);
      for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
      for (int j = 0; j < i; ++j) {
        f = d[j]; g = e[j];
        for (int k = j; k <= i - 1; ++k) V[(size_t)k * n + j] -= (f * e[k] + g * d[k]);
        d[j] = V[(size_t)(i - 1) * n + j]; V[(size_t)i * n + j] = 0.0;
      }
    }
    d[i] = h;
  }
  for (int i = 0; i < n - 1; ++i) {
    V[(size_t)(n - 1) * n + i] = V[(size_t)i * n + i]; V[(size_t)i * n + i] = 1.0;
    const double h = d[i + 1];
    if (h != 0.0) {
      for (int k = 0; k <= i; ++k) d[k] = V[(size_t)k * n + i + 1] / h;
      for (int j = 0; j <= i; ++j) {
        double g = 0.0;
        for (int k = 0; k <= i; ++k) g += V[(size_t)k * n + i + 1] * V[(size_t)k * n + j];
        for (int k = 0; k <= i; ++k) V[(size_t)k * n + j] -= g * d[k];
      }
    }
    for (int k = 0; k <= i; ++k) V[(size_t)k * n + i + 1] = 0.0;
  }
  for (int j = 0; j < n; ++j) { d[j] = V[(size_t)(n - 1) * n + j]; V[(size_t)(n - 1) * n + j] = 0.0; }
  V[(size_t)(n - 1) * n + n - 1] = 1.0; e[0] = 0.0;
}
void tql2(int n, double* V, double* d, double* e) {
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  const double eps = std::ldexp(1.0, -52);
  for (int l = 0; l < n; ++l) {
    tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
    int m = l;
    while (m < n) { if (std::fabs(e[m]) <= eps * tst1) break; ++m; }
    if (m > l) {
      int iter = 0;
      do {
        ++iter;
        double g = d[l], p = (d[l + 1] - g) / (2.0 * e[l]), r = std::hypot(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r); d[l + 1] = e[l] * (p + r);
        const double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; ++i) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1.0, c2 = c, c3 = c, s = 0.0, s2 = 0.0;
        const double el1 = e[l + 1];
        for (int i = m - 1; i >= l; --i) {
          c3 = c2; c2 = c; s2 = s;
          g = c * e[i]; h = c * p; r = std::hypot(p, e[i]);
          e[i + 1] = s * r; s = e[i] / r; c = p / r; p = c * d[i] - s * g; d[i + 1] = h + s * (c * g + s * d[i]);
          for (int k = 0; k < n; ++k) {
            h = V[(size_t)k * n + i + 1];
            V[(size_t)k * n + i + 1] = s * V[(size_t)k * n + i] + c * h;
            V[(size_t)k * n + i] = c * V[(size_t)k * n + i] - s * h;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1; e[l] = s * p; d[l] = c * p;
      } while (std::fabs(e[l]) > eps * tst1 && iter < 200);
    }
    d[l] += f; e[l] = 0.0;
  }
  for (int i = 0; i < n - 1; ++i) {      // ascending order
    int kk = i; double p = d[i];
    for (int j = i + 1; j < n; ++j) if (d[j] < p) { kk = j; p = d[j]; }
    if (kk != i) { d[kk] = d[i]; d[i] = p; for (int j = 0; j < n; ++j) std::swap(V[(size_t)j * n + i], V[(size_t)j * n + kk]); }
  }
}
// M symmetrised on entry; w ascending, V columns
void eigh(int n, const double* M, double* w, double* V, double* work) {
  if (n == 0) return;
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[(size_t)i * n + j] = 0.5 * (M[(size_t)i * n + j] + M[(size_t)j * n + i]);
  tred2(n, V, w, work); tql2(n, V, w, work);
}

struct Node {
  int R, r;
  const double *AY, *AU, *b, *xs, *Q;
  const int* kind;      // 0 trace, 1 cut, 2 other (box / bound rows)
};
struct Inst {
  int n, m, k; double gamma, sumA2;
  const double* A; const uint8_t* mask;
  std::vector<std::vector<int>> cols; std::vector<vec> a;
  vec N;
};

// oracle/omc_oracle.py:290-322
void nnqp(int R, const vec& G, const vec& c, vec& lam) {
  lam.assign(R, 0.0);
  if (R == 0) return;
  std::vector<char> P(R, 0);
  double scale = 1e-300;
  for (int i = 0; i < R; ++i) scale = std::max(scale, std::fabs(c[i]));
  vec w(R), s(R), Gp, wp, Vp, wk, rhs;
  std::vector<int> idx;
  for (int outer = 0; outer < 10 * R + 10; ++outer) {
    int t = -1; double best = -1e300;
    for (int i = 0; i < R; ++i) {
      if (P[i]) continue;
      double v = c[i];
      for (int j = 0; j < R; ++j) v -= G[(size_t)i * R + j] * lam[j];
      if (v > best) { best = v; t = i; }
    }
    if (t < 0 || best <= 1e-13 * scale) break;
    P[t] = 1;
    for (int inner = 0; inner < 10 * R + 10; ++inner) {
      idx.clear();
      for (int i = 0; i < R; ++i) if (P[i]) idx.push_back(i);
      const int np = (int)idx.size();
      if (np == 0) break;
      Gp.assign((size_t)np * np, 0.0); wp.assign(np, 0.0); Vp.assign((size_t)np * np, 0.0); wk.assign(np, 0.0); rhs.assign(np, 0.0);
      for (int a = 0; a < np; ++a) for (int q = 0; q < np; ++q) Gp[(size_t)a * np + q] = G[(size_t)idx[a] * R + idx[q]];
      eigh(np, Gp.data(), wp.data(), Vp.data(), wk.data());      // minimum-norm least squares (numpy lstsq, rcond = eps * np)
      double wmax = 0.0;
      for (int a = 0; a < np; ++a) wmax = std::max(wmax, std::fabs(wp[a]));
      const double cut = std::ldexp(1.0, -52) * np * wmax;
      for (int a = 0; a < np; ++a) { double v = 0.0; for (int q = 0; q < np; ++q) v += Vp[(size_t)q * np + a] * c[idx[q]]; rhs[a] = (std::fabs(wp[a]) > cut) ? v / wp[a] : 0.0; }
      std::fill(s.begin(), s.end(), 0.0);
      double smin = 1e300;
      for (int q = 0; q < np; ++q) { double v = 0.0; for (int a = 0; a < np; ++a) v += Vp[(size_t)q * np + a] * rhs[a]; s[idx[q]] = v; smin = std::min(smin, v); }
      if (smin > 0) { lam = s; break; }
      double al = 1e300;
      for (int q = 0; q < np; ++q) { const int i = idx[q]; if (s[i] <= 0) al = std::min(al, lam[i] / (lam[i] - s[i])); }
      for (int i = 0; i < R; ++i) lam[i] += al * (s[i] - lam[i]);
      bool dropped = false; int amin = -1; double lmin = 1e300;
      for (int q = 0; q < np; ++q) {
        const int i = idx[q];
        if (s[i] > 0) continue;
        if (lam[i] <= 1e-18 * scale) { P[i] = 0; dropped = true; }
        if (lam[i] < lmin) { lmin = lam[i]; amin = i; }
      }
      if (!dropped && amin >= 0) P[amin] = 0;
      for (int i = 0; i < R; ++i) if (!P[i]) lam[i] = 0.0;
    }
  }
}

bool chol_solve(int c, vec& B, vec& x) {      // in place: B -> L, x -> B^-1 x ; false when not positive definite
  for (int j = 0; j < c; ++j) {
    double d = B[(size_t)j * c + j];
    for (int q = 0; q < j; ++q) d -= B[(size_t)j * c + q] * B[(size_t)j * c + q];
    if (!(d > 0)) return false;
    d = std::sqrt(d); B[(size_t)j * c + j] = d;
    for (int i = j + 1; i < c; ++i) {
      double v = B[(size_t)i * c + j];
      for (int q = 0; q < j; ++q) v -= B[(size_t)i * c + q] * B[(size_t)j * c + q];
      B[(size_t)i * c + j] = v / d;
    }
  }
  for (int i = 0; i < c; ++i) { double v = x[i]; for (int q = 0; q < i; ++q) v -= B[(size_t)i * c + q] * x[q]; x[i] = v / B[(size_t)i * c + i]; }
  for (int i = c - 1; i >= 0; --i) { double v = x[i]; for (int q = i + 1; q < c; ++q) v -= B[(size_t)q * c + i] * x[q]; x[i] = v / B[(size_t)i * c + i]; }
  return true;
}

// f(Y) and Lam (n x m, zero off the support): oracle/omc_oracle.py:360-376
double f_value(const Inst& I, const vec& Y, vec& Lam) {
  const int n = I.n, m = I.m;
  std::fill(Lam.begin(), Lam.end(), 0.0);
  double v = 0.0; vec B, x;
  for (int j = 0; j < m; ++j) {
    const auto& o = I.cols[j]; const int c = (int)o.size();
    if (!c) continue;
    B.assign((size_t)c * c, 0.0); x = I.a[j];
    for (int p = 0; p < c; ++p) for (int q = 0; q < c; ++q) B[(size_t)p * c + q] = ((p == q) ? 1.0 : 0.0) + I.gamma * Y[(size_t)o[p] * n + o[q]];
    if (!chol_solve(c, B, x)) return INFINITY;
    for (int p = 0; p < c; ++p) { v += 0.5 * I.a[j][p] * x[p]; Lam[(size_t)o[p] * m + j] = x[p]; }
  }
  return v;
}

struct Out { double objective, lb, rp, rd, rho; int iters, status; };

// oracle/omc_oracle.py:621-810 (cold start, no acceleration)
void relax(const Inst& I, const Node& nd, const double* P, Out& out, double* Yout) {
  const int n = I.n, m = I.m, k = I.k, R = nd.R, r = nd.r, nk = r + k;
  const double g = I.gamma;
  const double eps_gap = P[0], eps_feas = P[1]; const int max_iters = (int)P[2], check_every = (int)P[3];
  const double rho_scale = P[4], rff = P[5], rx = P[6]; const int stall_checks = (int)P[7];
  const int bump = (int)P[8]; const double bump_factor = P[9]; const int bump_window = (int)P[10], bump_after = (int)P[11], bump_max = (int)P[12];
  const double bump_ratio = P[13]; const int es_after = (int)P[14]; const double es_factor = P[15];
  double rho = rho_scale * 0.5 * g * I.sumA2 / (m * (1.0 + g * k / n) * (1.0 + g * k / n));
  if (!(rho > 0)) rho = 1.0;
  const size_t nn = (size_t)n * n;
  vec wY1(nn); for (size_t e = 0; e < nn; ++e) wY1[e] = rff * I.N[e] + 2.0;
  vec G1((size_t)R * R, 0.0);
  for (int a = 0; a < R; ++a) for (int q = a; q < R; ++q) {
    double v = 0.0;
    for (size_t e = 0; e < nn; ++e) v += nd.AY[(size_t)a * nn + e] * nd.AY[(size_t)q * nn + e] / wY1[e];
    for (int e = 0; e < n * k; ++e) v += 0.5 * nd.AU[(size_t)a * n * k + e] * nd.AU[(size_t)q * n * k + e];
    G1[(size_t)a * R + q] = v; G1[(size_t)q * R + a] = v;
  }
  vec Y(nn, 0.0), Yp, D1(nn, 0.0), D3(nn, 0.0), Vt((size_t)r * k, 0.0), D3V((size_t)r * k, 0.0), D3T((size_t)k * k, 0.0);
  for (int i = 0; i < n; ++i) Y[(size_t)i * n + i] = (double)k / n;
  Yp = Y;
  std::vector<vec> alpha(m); vec sval(m, -1.0);
  for (int j = 0; j < m; ++j) alpha[j].assign(I.cols[j].size(), 0.0);
  vec LL(nn), W1(nn), M(nn), w(n), V(nn), wk(std::max(n, nk) + 1), M3((size_t)nk * nk), w3(nk), V3((size_t)nk * nk), P3((size_t)nk * nk), Q3((size_t)nk * nk, 0.0);
  vec tY(nn), tV((size_t)r * k), tU((size_t)n * k), cvec(R), mu, lam(R, 0.0), Yn(nn), Vn((size_t)r * k), QdSQ(nn), tmp((size_t)n * std::max(r, 1)), dS((size_t)r * r);
  vec Lam((size_t)n * m), B, Qc, bq, qa;
  int status = 1;      // OMC_SLOW_PROGRESS
  double obj = INFINITY, lb = -INFINITY, rp = INFINITY, rd = INFINITY;
  int stall = 0; double obj_prev = INFINITY, lb_prev = -INFINITY;
  int n_bumps = 0, last_bump = 0, slow_votes = 0; double gap_prev = 1e300, gap_rate = 1.0;
  int it = 0;
  for (it = 1; it <= max_iters; ++it) {
    const double rho_f = rho * rff, cp = g * g / (2.0 * rho_f);
    // ---- (F) columns: oracle/omc_oracle.py:475-515 ----
    std::fill(LL.begin(), LL.end(), 0.0);
    for (int j = 0; j < m; ++j) {
      const auto& o = I.cols[j]; const int c = (int)o.size();
      if (!c) continue;
      B.assign((size_t)c * c, 0.0); Qc.assign((size_t)c * c, 0.0); bq.assign(c, 0.0); qa.assign(c, 0.0);
      vec& al = alpha[j];
      for (int p = 0; p < c; ++p) for (int q = 0; q < c; ++q) {
        const size_t e = (size_t)o[p] * n + o[q];
        B[(size_t)p * c + q] = ((p == q) ? 1.0 : 0.0) + g * ((2.0 * Y[e] - Yp[e]) - (g / (2.0 * rho_f)) * al[p] * al[q]);
      }
      eigh(c, B.data(), bq.data(), Qc.data(), wk.data());
      for (int t = 0; t < c; ++t) { double v = 0.0; for (int p = 0; p < c; ++p) v += Qc[(size_t)p * c + t] * I.a[j][p]; qa[t] = v; }
      auto phi = [&](double s_, double& dph) { double ph = -s_; dph = -1.0; for (int t = 0; t < c; ++t) { const double d = bq[t] + cp * s_; ph += qa[t] * qa[t] / (d * d); dph -= 2.0 * cp * qa[t] * qa[t] / (d * d * d); } return ph; };
      double lo = std::max(0.0, -bq[0] / cp) * (1.0 + 1e-12), hi = std::max(2.0 * lo + 1.0, 1.0), dph;
      for (int q = 0; q < 200 && phi(hi, dph) > 0; ++q) hi *= 2.0;
      double s = (sval[j] >= 0.0) ? sval[j] : hi;
      if (!(s > lo && s < hi)) s = hi;
      for (int q = 0; q < 100; ++q) {
        const double ph = phi(s, dph);
        if (ph > 0) lo = s; else hi = s;
        double sn = s - ph / dph;
        if (!(sn > lo && sn < hi)) sn = 0.5 * (lo + hi);
        const bool done = std::fabs(sn - s) <= 1e-15 * std::max(1.0, std::fabs(s));
        s = sn;
        if (done) break;
      }
      sval[j] = s;
      for (int p = 0; p < c; ++p) { double v = 0.0; for (int t = 0; t < c; ++t) v += Qc[(size_t)p * c + t] * qa[t] / (bq[t] + cp * s); al[p] = v; }
      for (int p = 0; p < c; ++p) for (int q = 0; q < c; ++q) LL[(size_t)o[p] * n + o[q]] += al[p] * al[q];
    }
    // ---- (C1) clip of Y - D1 to [0, 1] ----
    for (size_t e = 0; e < nn; ++e) M[e] = Y[e] - D1[e];
    eigh(n, M.data(), w.data(), V.data(), wk.data());
    std::fill(W1.begin(), W1.end(), 0.0);
    for (int t = 0; t < n; ++t) {
      const double lt = std::min(std::max(w[t], 0.0), 1.0);
      if (lt == 0.0) continue;
      for (int i = 0; i < n; ++i) { const double vi = lt * V[(size_t)i * n + t]; for (int j = 0; j < n; ++j) W1[(size_t)i * n + j] += vi * V[(size_t)j * n + t]; }
    }
    // ---- (C3) small cone [Q'(Y - D3)Q  Vt - D3V ; .  I - D3T] ----
    for (int i = 0; i < n; ++i) for (int a = 0; a < r; ++a) { double v = 0.0; for (int j = 0; j < n; ++j) v += (Y[(size_t)i * n + j] - D3[(size_t)i * n + j]) * nd.Q[(size_t)j * r + a]; tmp[(size_t)i * r + a] = v; }
    for (int a = 0; a < r; ++a) for (int q = 0; q < r; ++q) { double v = 0.0; for (int i = 0; i < n; ++i) v += nd.Q[(size_t)i * r + a] * tmp[(size_t)i * r + q]; M3[(size_t)a * nk + q] = v; }
    for (int a = 0; a < r; ++a) for (int t = 0; t < k; ++t) { const double v = Vt[(size_t)a * k + t] - D3V[(size_t)a * k + t]; M3[(size_t)a * nk + r + t] = v; M3[(size_t)(r + t) * nk + a] = v; }
    for (int s_ = 0; s_ < k; ++s_) for (int t = 0; t < k; ++t) M3[(size_t)(r + s_) * nk + r + t] = ((s_ == t) ? 1.0 : 0.0) - D3T[(size_t)s_ * k + t];
    eigh(nk, M3.data(), w3.data(), V3.data(), wk.data());
    std::fill(P3.begin(), P3.end(), 0.0);
    for (int t = 0; t < nk; ++t) { if (w3[t] <= 0) continue; for (int i = 0; i < nk; ++i) { const double vi = w3[t] * V3[(size_t)i * nk + t]; for (int j = 0; j < nk; ++j) P3[(size_t)i * nk + j] += vi * V3[(size_t)j * nk + t]; } }
    for (int i = 0; i < nk; ++i) for (int j = 0; j < nk; ++j) Q3[(size_t)i * nk + j] = P3[(size_t)i * nk + j] - 0.5 * (M3[(size_t)i * nk + j] + M3[(size_t)j * nk + i]);
    for (int a = 0; a < r; ++a) for (int q = 0; q < r; ++q) dS[(size_t)a * r + q] = Q3[(size_t)a * nk + q];
    for (int i = 0; i < n; ++i) for (int q = 0; q < r; ++q) { double v = 0.0; for (int a = 0; a < r; ++a) v += nd.Q[(size_t)i * r + a] * dS[(size_t)a * r + q]; tmp[(size_t)i * r + q] = v; }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double v = 0.0; for (int q = 0; q < r; ++q) v += tmp[(size_t)i * r + q] * nd.Q[(size_t)j * r + q]; QdSQ[(size_t)i * n + j] = v; }
    // ---- global step ----
    for (size_t e = 0; e < nn; ++e)
      tY[e] = (rho_f * (I.N[e] * Y[e]) + 0.5 * g * LL[e] + rho * (rx * W1[e] + (1.0 - rx) * Y[e] + D1[e]) + rho * (Y[e] + (1.0 - rx) * D3[e] + rx * QdSQ[e])) / (rho * wY1[e]);
    for (int a = 0; a < r; ++a) for (int t = 0; t < k; ++t) tV[(size_t)a * k + t] = rx * P3[(size_t)a * nk + r + t] + (1.0 - rx) * Vt[(size_t)a * k + t] + D3V[(size_t)a * k + t];
    for (int i = 0; i < n; ++i) for (int t = 0; t < k; ++t) { double v = 0.0; for (int a = 0; a < r; ++a) v += nd.Q[(size_t)i * r + a] * tV[(size_t)a * k + t]; tU[(size_t)i * k + t] = v; }
    for (int a = 0; a < R; ++a) {
      double v = -nd.b[a];
      for (size_t e = 0; e < nn; ++e) v += nd.AY[(size_t)a * nn + e] * tY[e];
      for (int e = 0; e < n * k; ++e) v += nd.AU[(size_t)a * n * k + e] * tU[e];
      cvec[a] = v;
    }
    nnqp(R, G1, cvec, mu);
    for (int a = 0; a < R; ++a) lam[a] = rho * mu[a];
    Yn = tY;
    for (int a = 0; a < R; ++a) { if (mu[a] == 0.0) continue; for (size_t e = 0; e < nn; ++e) Yn[e] -= mu[a] * nd.AY[(size_t)a * nn + e] / wY1[e]; }
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) { const double v = 0.5 * (Yn[(size_t)i * n + j] + Yn[(size_t)j * n + i]); Yn[(size_t)i * n + j] = v; Yn[(size_t)j * n + i] = v; }
    {
      vec cu((size_t)n * k, 0.0);
      for (int a = 0; a < R; ++a) { if (mu[a] == 0.0) continue; for (int e = 0; e < n * k; ++e) cu[e] += 0.5 * mu[a] * nd.AU[(size_t)a * n * k + e]; }
      for (int a = 0; a < r; ++a) for (int t = 0; t < k; ++t) { double v = 0.0; for (int i = 0; i < n; ++i) v += nd.Q[(size_t)i * r + a] * cu[(size_t)i * k + t]; Vn[(size_t)a * k + t] = tV[(size_t)a * k + t] - v; }
    }
    double r1 = 0.0, r2 = 0.0, r3 = 0.0, r4 = 0.0, d1 = 0.0, d2 = 0.0;
    for (size_t e = 0; e < nn; ++e) {
      const double w3y = (Y[e] - D3[e]) + QdSQ[e] - Yn[e];
      r1 += (W1[e] - Yn[e]) * (W1[e] - Yn[e]); r2 += w3y * w3y; d1 += (Yn[e] - Y[e]) * (Yn[e] - Y[e]);
      D1[e] = D1[e] + rx * W1[e] + (1.0 - rx) * Y[e] - Yn[e];
      D3[e] = (1.0 - rx) * D3[e] + Y[e] + rx * QdSQ[e] - Yn[e];
    }
    for (int a = 0; a < r; ++a) for (int t = 0; t < k; ++t) {
      const size_t e = (size_t)a * k + t; const double w3v = P3[(size_t)a * nk + r + t];
      r3 += (w3v - Vn[e]) * (w3v - Vn[e]); d2 += (Vn[e] - Vt[e]) * (Vn[e] - Vt[e]);
      D3V[e] = D3V[e] + rx * w3v + (1.0 - rx) * Vt[e] - Vn[e];
    }
    for (int s_ = 0; s_ < k; ++s_) for (int t = 0; t < k; ++t) {
      const double w3t = P3[(size_t)(r + s_) * nk + r + t] - ((s_ == t) ? 1.0 : 0.0);
      r4 += w3t * w3t; D3T[(size_t)s_ * k + t] += rx * w3t;
    }
    rp = std::sqrt(r1 + r2 + 2.0 * r3 + r4); rd = std::sqrt(d1 + 2.0 * d2);
    Yp = Y; Y = Yn; Vt = Vn;
    if (it % check_every != 0 && it != max_iters) continue;
    // ---- certificate: oracle/omc_oracle.py:546-571 ----
    obj = f_value(I, Y, Lam);
    double lb_new;
    {
      double c0 = 0.0;
      for (size_t e = 0; e < (size_t)n * m; ++e) c0 += I.A[e] * Lam[e] - 0.5 * Lam[e] * Lam[e];
      for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) { double v = 0.0; for (int c = 0; c < m; ++c) v += Lam[(size_t)i * m + c] * Lam[(size_t)j * m + c]; M[(size_t)i * n + j] = -0.5 * g * v; M[(size_t)j * n + i] = -0.5 * g * v; }
      vec cU((size_t)n * k, 0.0); double cst = 0.0;
      for (int a = 0; a < R; ++a) {
        if (nd.kind[a] == 0 || lam[a] == 0.0) continue;
        if (nd.kind[a] == 1) { const double* x = nd.xs + (size_t)a * n; for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) M[(size_t)i * n + j] += lam[a] * x[i] * x[j]; }
        for (int e = 0; e < n * k; ++e) cU[e] += lam[a] * nd.AU[(size_t)a * n * k + e];
        cst -= lam[a] * nd.b[a];
      }
      for (size_t e = 0; e < nn; ++e) M[e] -= rho * QdSQ[e];
      double nrm = 0.0;
      for (int t = 0; t < k; ++t) {
        double s2 = 0.0;
        for (int a = 0; a < r; ++a) { double v = 0.0; for (int i = 0; i < n; ++i) v += nd.Q[(size_t)i * r + a] * cU[(size_t)i * k + t]; v -= 2.0 * rho * Q3[(size_t)a * nk + r + t]; s2 += v * v; }
        nrm += std::sqrt(s2);
      }
      for (int t = 0; t < k; ++t) cst -= rho * Q3[(size_t)(r + t) * nk + r + t];
      eigh(n, M.data(), w.data(), V.data(), wk.data());
      double ev = 0.0;
      for (int t = 0; t < k && t < n; ++t) ev += std::min(w[t], 0.0);
      lb_new = c0 + ev - nrm + cst;
    }
    lb = std::max(lb, lb_new);
    const double sc = std::max(1.0, std::fabs(obj));
    if (std::fabs(obj - lb) <= eps_gap * sc && rp <= eps_feas * std::sqrt((double)(n + k))) { status = 0; break; }
    if (lb > 0.5 * I.sumA2 * (1.0 + 1e-9) + 1e-9) { status = 3; break; }
    if (std::fabs(obj - obj_prev) <= 1e-7 * sc && lb_new <= lb_prev + 1e-7 * sc) ++stall; else stall = 0;
    obj_prev = obj; lb_prev = lb;
    if (stall >= stall_checks) { if (std::fabs(obj - lb) <= eps_gap * sc && rp <= 10.0 * eps_feas * std::sqrt((double)(n + k))) status = 0; break; }
    if (it >= max_iters) break;
    if (es_factor > 0.0) {
      const double target = eps_gap * sc, gnow = obj - lb;
      const double q = (gap_prev < 1e299 && gap_prev > 0.0 && gnow > 0.0) ? 0.5 * gap_rate + 0.5 * std::min(gnow / gap_prev, 2.0) : 1.0;
      gap_prev = gnow; gap_rate = q;
      const double left = (max_iters - it) / (double)check_every;
      const double need = (q < 1.0) ? std::log(std::max(gnow, target) / target) / -std::log(q) : 1e300;
      const bool hopeless = it >= es_after && gnow > target && need > es_factor * left;
      slow_votes = hopeless ? slow_votes + 1 : 0;
      if (slow_votes >= 8) break;
    }
    if (bump && it >= bump_after && n_bumps < bump_max && it - last_bump >= check_every * bump_window && rp > bump_ratio * rd) {
      rho *= bump_factor;
      for (auto* d : {&D1, &D3, &D3V, &D3T}) for (double& v : *d) v /= bump_factor;
      ++n_bumps; last_bump = it; slow_votes = 0; gap_rate = 1.0;
    }
  }
  if (it > max_iters) it = max_iters;
  obj = f_value(I, Y, Lam);
  out.objective = obj; out.lb = lb; out.rp = rp; out.rd = rd; out.rho = rho; out.iters = it; out.status = status;
  if (Yout) std::memcpy(Yout, Y.data(), sizeof(double) * nn);
}

void make_inst(Inst& I, int n, int m, int k, const double* A, const uint8_t* mask, double gamma) {
  I.n = n; I.m = m; I.k = k; I.gamma = gamma; I.A = A; I.mask = mask; I.sumA2 = 0.0;
  I.cols.assign(m, {}); I.a.assign(m, {});
  for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) if (mask[(size_t)i * m + j]) { I.cols[j].push_back(i); I.a[j].push_back(A[(size_t)i * m + j]); I.sumA2 += A[(size_t)i * m + j] * A[(size_t)i * m + j]; }
  I.N.assign((size_t)n * n, 0.0);
  for (int j = 0; j < m; ++j) for (int p : I.cols[j]) for (int q : I.cols[j]) I.N[(size_t)p * n + q] += 1.0;
}
}  // namespace

extern "C" {
// B nodes of one instance; node b: R[b] rows, r[b] basis vectors, dense row coefficients AY[b] (R x n*n), AU[b] (R x n*k), rhs b[b], kind[b],
// xs[b] (R x n), Q[b] (n x r).  params: 16 doubles (see relax()).  out: 7 doubles per node (objective, dual bound, rp, rd, rho, iters, status).
// threads <= 1: the nodes one after the other on the calling thread; otherwise OpenMP over the nodes (each node stays single-threaded).
int omc_cpu_relax_nodes(int n, int m, int k, const double* A, const uint8_t* mask, double gamma, int B, const int* R, const int* r,
                        const double* const* AY, const double* const* AU, const double* const* b, const int* const* kind,
                        const double* const* xs, const double* const* Q, const double* params, int threads, double* out, double* Y0) {
  Inst I; make_inst(I, n, m, k, A, mask, gamma);
#ifdef _OPENMP
  if (threads > 1) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1) if (threads > 1)
#endif
  for (int q = 0; q < B; ++q) {
    Node nd{R[q], r[q], AY[q], AU[q], b[q], xs[q], Q[q], kind[q]};
    Out o{};
    relax(I, nd, params, o, (q == 0) ? Y0 : nullptr);
    double* d = out + (size_t)7 * q;
    d[0] = o.objective; d[1] = o.lb; d[2] = o.rp; d[3] = o.rd; d[4] = o.rho; d[5] = o.iters; d[6] = o.status;
  }
  return 0;
}
int omc_cpu_ref_openmp(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 0;
#endif
}
}
