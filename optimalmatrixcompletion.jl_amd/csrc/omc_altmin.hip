// omc_altmin.hip -- alternating minimisation (OMC.jl:1979-2279), rank 1, one workgroup per problem.
//
// Both JuMP models of the reference minimise  1/2 sum_Omega ((UV)_ij - A_ij)^2 + 1/(2 gamma) sum_all (UV)_ij^2
// (OMC.jl:2193-2207, 2213-2227).  For k = 1:
//   V-step (model_V, unconstrained): v_j = sum_{i in O_j} u_i A_ij / (sum_{i in O_j} u_i^2 + u'u/gamma)   -- masked LS,
//          one coalesced pass over the CSC copy of (A, indices);
//   U-step (model_U): min 1/2 sum_i h_i u_i^2 - g_i u_i  s.t.  box rows, per-cut bounds lo <= x'u <= hi (OMC.jl:2047-2093),
//          ||u|| <= 1 (OMC.jl:2164-2171);  h_i = sum_{j in O_i} v_j^2 + v'v/gamma, g_i = sum_{j in O_i} A_ij v_j (CSR pass).
//          Solved exactly by duality: multiplier theta of the ball by safeguarded bisection, rows by the same active-set
//          NNQP as the relaxation (Gram matrix C diag(1/(h+theta)) C').
// The whole loop (<= max_iters iterations, convergence rules of OMC.jl:2234-2245 incl. quirk Q3) runs inside one launch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "omc_device.h"
#include "omc_wave.h"
#include "omc_altmin.h"

__device__ __forceinline__ double am_row_entry(const AltminWS& w, int b, int r, int i) {
  const int kind = w.rkind[(size_t)b * w.Rmax + r];
  const double cf = w.rcoef[(size_t)b * w.Rmax + r];
  if (kind == ROW_BOX) return (w.rbi[(size_t)b * w.Rmax + r] == i) ? cf : 0.0;
  return cf * w.cutx[((size_t)b * w.Lmax + w.rcut[(size_t)b * w.Rmax + r]) * w.n + i];
}

#define AM_MAX_DOUBLINGS 64      // bracket search of the ball multiplier (theta <= 2^64 max|g|: beyond that model_U is infeasible)
#define AM_FEAS_TOL 1e-6         // largest row violation / excess of ||u||^2 - 1 accepted from a U-step (the oracle uses the same)
__device__ __forceinline__ double block_max(double v, double* red) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, WAVE));
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (l == 0) red[wv] = v;
  __syncthreads();
  double s = red[0];
  for (int i = 1; i < nw; ++i) s = fmax(s, red[i]);
  return s;
}

__global__ void __launch_bounds__(256) k_altmin(AltminWS w) {
  extern __shared__ double sm_lds[];
  // problems that do not fit the LDS (config 5: 1000 x 1000) keep the same state in a per-problem global slab (L2 / HBM): same code, flat addresses
  double* sm = w.scratch ? w.scratch + (size_t)blockIdx.x * w.scratch_stride : sm_lds;
  __shared__ double red[32];
  __shared__ double s_Gp[NNQP_PMAX * (NNQP_PMAX + 1) / 2];
  __shared__ double s_sv[NNQP_PMAX], s_tmp[NNQP_PMAX];
  __shared__ int s_pl[NNQP_PMAX];
  __shared__ int s_stop, s_ov;
  const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  const int n = w.n, m = w.m, R = w.R[b];
  double* u = sm;            // n
  double* v = u + n;         // m
  double* h = v + m;         // n
  double* g = h + n;         // n
  double* u0 = g + n;        // n  unconstrained-in-rows point for the current theta
  double* cvec = u0 + n;     // Rmax
  double* mu = cvec + w.Rmax;  // Rmax
  double* G = w.G + (size_t)b * w.Rmax * w.Rmax;
  double* objs = w.objectives + (size_t)b * w.max_iters;
  for (int i = tid; i < n; i += T) u[i] = w.U0[(size_t)b * n + i];
  for (int r = tid; r < w.Rmax; r += T) mu[r] = 0.0;
  for (int t = tid; t < w.max_iters; t += T) objs[t] = __longlong_as_double(0x7ff8000000000000LL);  // NaN padding
  __syncthreads();
  double objective_current = 1e10;   // OMC.jl:2012
  int counter = 0, converged = 0, failed = 0;
  while (counter < w.max_iters) {
    ++counter;
    // ---- V-step ------------------------------------------------------------------------------------------
    double uu = 0.0;
    for (int i = tid; i < n; i += T) uu += u[i] * u[i];
    uu = block_sum(uu, red) / w.gamma;
    for (int j = tid; j < m; j += T) {
      double num = 0.0, den = uu;
      for (int p = w.col_ptr[j]; p < w.col_ptr[j + 1]; ++p) { double ui = u[w.col_idx[p]]; num += ui * w.col_val[p]; den += ui * ui; }
      v[j] = (den > 0.0) ? num / den : 0.0;
    }
    __syncthreads();
    // ---- U-step quantities -------------------------------------------------------------------------------
    double vv = 0.0;
    for (int j = tid; j < m; j += T) vv += v[j] * v[j];
    vv = block_sum(vv, red) / w.gamma;
    for (int i = tid; i < n; i += T) {
      double hh = vv, gg = 0.0;
      for (int p = w.row_ptr[i]; p < w.row_ptr[i + 1]; ++p) { double vj = v[w.row_idx[p]]; hh += vj * vj; gg += vj * w.row_val[p]; }
      h[i] = hh; g[i] = gg;
    }
    __syncthreads();
    // ---- QP: theta by bisection, rows by NNQP ----------------------------------------------------------------
    // solve(theta): minimiser over the rows for a fixed ball multiplier; returns ||u||^2 (same value in every thread)
    auto solve = [&](double theta) -> double {
      for (int i = tid; i < n; i += T) u0[i] = g[i] / (h[i] + theta);
      __syncthreads();
      // one thread per Gram entry / per row (R is small: serial dot products of length n beat R^2 block reductions)
      for (int e = tid; e < R * R; e += T) {
        int r1 = e / R, r2 = e - r1 * R;
        if (r2 < r1) continue;
        double acc = 0.0;
        for (int i = 0; i < n; ++i) { double a1 = am_row_entry(w, b, r1, i); if (a1 != 0.0) acc += a1 * am_row_entry(w, b, r2, i) / (h[i] + theta); }
        G[(size_t)r1 * w.Rmax + r2] = acc; G[(size_t)r2 * w.Rmax + r1] = acc;
      }
      for (int r = tid; r < R; r += T) {
        double acc = 0.0;
        for (int i = 0; i < n; ++i) acc += am_row_entry(w, b, r, i) * u0[i];
        cvec[r] = acc - w.rrhs[(size_t)b * w.Rmax + r];
      }
      __syncthreads();
      if (R > 0 && tid < 64) { const int ov = wave_nnqp(G, w.Rmax, cvec, mu, R, s_Gp, s_sv, s_tmp, s_pl, tid); if (tid == 0) s_ov = ov; }
      __syncthreads();
      double nn = 0.0;
      for (int i = tid; i < n; i += T) {
        double corr = 0.0;
        for (int r = 0; r < R; ++r) { double mv = mu[r]; if (mv != 0.0) corr += mv * am_row_entry(w, b, r, i); }
        double ui = u0[i] - corr / (h[i] + theta);
        u[i] = ui; nn += ui * ui;
      }
      return block_sum(nn, red);
    };
    if (tid == 0) s_ov = 0;
    __syncthreads();
    if (solve(0.0) > 1.0) {                       // ball active (OMC.jl:2164-2171)
      double lo_t = 0.0, hi_t = 1.0;
      for (int i = 0; i < n; ++i) hi_t = fmax(hi_t, fabs(g[i]));
      // an infeasible model_U (contradictory cut bounds, or rows that leave no point inside the unit ball) never enters the ball:
      // the bracket search is capped and the feasibility test below reports the failure
      for (int dbl = 0; dbl < AM_MAX_DOUBLINGS && solve(hi_t) > 1.0; ++dbl) hi_t *= 2.0;
      for (int it = 0; it < 200; ++it) {
        const double theta = 0.5 * (lo_t + hi_t);
        if (solve(theta) > 1.0) lo_t = theta; else hi_t = theta;
        if (hi_t - lo_t <= 1e-15 * fmax(1.0, hi_t)) break;
      }
      solve(hi_t);
    }
    // ---- did model_U have a solution?  (OMC.jl:2231, 2263-2265: a failed solve ends the loop with converged = false) ---------
    {
      double viol = 0.0, nn = 0.0;
      for (int r = tid; r < R; r += T) {
        double acc = 0.0;
        for (int i = 0; i < n; ++i) acc += am_row_entry(w, b, r, i) * u[i];
        viol = fmax(viol, acc - w.rrhs[(size_t)b * w.Rmax + r]);
      }
      for (int i = tid; i < n; i += T) nn += u[i] * u[i];
      nn = block_sum(nn, red);
      viol = block_max(viol, red);
      const bool bad = !(viol <= AM_FEAS_TOL) || !(nn <= 1.0 + AM_FEAS_TOL) || s_ov;
      __syncthreads();
      if (bad) { failed = 1; break; }
    }
    // ---- objective of model_U (OMC.jl:2232) ------------------------------------------------------------------
    double q = 0.0;
    for (int i = tid; i < n; i += T) q += 0.5 * h[i] * u[i] * u[i] - g[i] * u[i];
    const double objective_new = block_sum(q, red) + 0.5 * w.sumA2;
    if (tid == 0) {
      objs[counter - 1] = objective_new;
      int conv = 0;
      const double diff = fabs((objective_new - objective_current) / objective_current);
      if (diff < w.eps) conv = 1;                                                  // OMC.jl:2235
      else if (counter > 5) {                                                      // OMC.jl:2237-2243 (quirk Q3)
        conv = 1;
        for (int i2 = 0; i2 < 5; ++i2) if (!(objs[counter - 1 - i2] > objs[counter - 6])) conv = 0;
      }
      s_stop = conv;
    }
    __syncthreads();
    converged = s_stop;
    __syncthreads();
    if (converged) break;
    objective_current = objective_new;
  }
  for (int i = tid; i < n; i += T) w.U[(size_t)b * n + i] = u[i];
  for (int j = tid; j < m; j += T) w.V[(size_t)b * m + j] = v[j];
  if (tid == 0) { w.converged[b] = failed ? 0 : converged; w.n_iters[b] = counter; }
  {   // evaluate_objective(U V) from the factors (OMC.jl:920, 925-927): 1/2 sum_Omega (u_i v_j - A_ij)^2 + ||u||^2 ||v||^2 / (2 gamma)
    __syncthreads();
    double fit = 0.0, uu2 = 0.0, vv2 = 0.0;
    for (int i = tid; i < n; i += T) {
      const double ui = u[i]; uu2 += ui * ui;
      for (int p = w.row_ptr[i]; p < w.row_ptr[i + 1]; ++p) { const double d = ui * v[w.row_idx[p]] - w.row_val[p]; fit += d * d; }
    }
    for (int j = tid; j < m; j += T) vv2 += v[j] * v[j];
    fit = block_sum(fit, red); uu2 = block_sum(uu2, red); vv2 = block_sum(vv2, red);
    if (tid == 0) w.mobj[b] = 0.5 * fit + uu2 * vv2 / (2.0 * w.gamma);
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Rank k > 1 (k <= 4).  model_U has k^2 quadratic constraints sum_i (w_c' u_i)^2 <= r_c: the k unit balls (OMC.jl:2164-2171) and,
// per pair of columns, ||U_j1 +- U_j2||^2 <= 2 (OMC.jl:2029-2045).  Their multipliers theta_c >= 0 shift every row Hessian by
// the same k x k matrix S(theta) = 2 sum_c theta_c w_c w_c'; for fixed theta the problem is the row-separable QP with linear
// rows of the rank-1 case (active-set NNQP on the Gram matrix C H(theta)^-1 C').  Outer: projected Newton on the concave dual
// (gradient q_c = constraint values, Jacobian by forward differences, symmetrised, Levenberg-Marquardt damping, ascent test on
// the dual value) -- exactly the oracle's _ustep_dual_newton.  V-step: one k x k normal-equation solve per column.
// ---------------------------------------------------------------------------------------------------------------------
#define AK_KMAX 4
#define AK_QMAX 16

__device__ __forceinline__ double ak_row_x(const AltminWS& w, int b, int r, int i) {
  const int kind = w.rkind[(size_t)b * w.Rmax + r];
  const double cf = w.rcoef[(size_t)b * w.Rmax + r];
  if (kind == ROW_BOX) return (w.rbi[(size_t)b * w.Rmax + r] == i) ? cf : 0.0;
  return cf * w.cutx[((size_t)b * w.Lmax + w.rcut[(size_t)b * w.Rmax + r]) * w.n + i];
}
// in-place Gauss-Jordan inverse of a k x k symmetric positive definite matrix held in registers; false if a pivot is not positive
__device__ __forceinline__ bool ak_inv(double (&a)[AK_KMAX][AK_KMAX], int k) {
  bool ok = true;
#pragma unroll
  for (int p = 0; p < AK_KMAX; ++p) {
    if (p >= k) continue;
    const double piv = a[p][p];
    if (!(piv > 0.0)) ok = false;
    const double d = 1.0 / piv;
    a[p][p] = 1.0;
#pragma unroll
    for (int c = 0; c < AK_KMAX; ++c) if (c < k) a[p][c] *= d;
#pragma unroll
    for (int r = 0; r < AK_KMAX; ++r) {
      if (r >= k || r == p) continue;
      const double f = a[r][p];
      a[r][p] = 0.0;
#pragma unroll
      for (int c = 0; c < AK_KMAX; ++c) if (c < k) a[r][c] -= f * a[p][c];
    }
  }
  return ok;
}

__global__ void __launch_bounds__(256) k_altmin_k(AltminWS w) {
  extern __shared__ double sm_lds[];
  double* sm = w.scratch ? w.scratch + (size_t)blockIdx.x * w.scratch_stride : sm_lds;
  __shared__ double red[32];
  __shared__ double s_Gp[NNQP_PMAX * (NNQP_PMAX + 1) / 2];
  __shared__ double s_sv[NNQP_PMAX], s_tmp[NNQP_PMAX];
  __shared__ int s_pl[NNQP_PMAX];
  __shared__ double s_th[AK_QMAX], s_th2[AK_QMAX], s_q[AK_QMAX], s_qt[AK_QMAX], s_W[AK_QMAX * AK_KMAX], s_rad[AK_QMAX];
  __shared__ double s_J[AK_QMAX * AK_QMAX], s_P[AK_QMAX * AK_QMAX], s_L[AK_QMAX * AK_QMAX], s_step[AK_QMAX];
  __shared__ int s_act[AK_QMAX], s_nact, s_flag, s_stop, s_ov;
  const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  const int n = w.n, m = w.m, k = w.k, kk = k * k, mq = k * k, R = w.R[b];
  double* u = sm;                 // n*k, row i at u + i*k  (accepted iterate)
  double* ut = u + n * k;         // candidate of the last evaluation
  double* u0 = ut + n * k;
  double* g = u0 + n * k;
  double* H = g + n * k;          // n*k*k
  double* Hinv = H + n * kk;
  double* v = Hinv + n * kk;      // k*m, column j at v + j*k
  double* cvec = v + k * m;
  double* mu = cvec + w.Rmax;
  double* G = w.G + (size_t)b * w.Rmax * w.Rmax;
  double* objs = w.objectives + (size_t)b * w.max_iters;
  for (int e = tid; e < n * k; e += T) { const int i = e % n, a = e / n; u[i * k + a] = w.U0[(size_t)b * n * k + e]; }   // input is column-major n x k
  for (int r = tid; r < w.Rmax; r += T) mu[r] = 0.0;
  for (int t = tid; t < w.max_iters; t += T) objs[t] = __longlong_as_double(0x7ff8000000000000LL);
  if (tid == 0) {   // quadratic constraint vectors: balls, then (+, -) per pair  (quadratic_constraint_vectors in the oracle)
    int c = 0;
    for (int j = 0; j < k; ++j) { for (int a = 0; a < k; ++a) s_W[c * AK_KMAX + a] = (a == j) ? 1.0 : 0.0; s_rad[c] = 1.0; ++c; }
    for (int j1 = 0; j1 < k - 1; ++j1)
      for (int j2 = j1 + 1; j2 < k; ++j2)
        for (int sg = 0; sg < 2; ++sg) {
          for (int a = 0; a < k; ++a) s_W[c * AK_KMAX + a] = (a == j1) ? 1.0 : (a == j2) ? (sg ? -1.0 : 1.0) : 0.0;
          s_rad[c] = 2.0; ++c;
        }
  }
  __syncthreads();
  // ---- one evaluation of the inner problem for multipliers th: candidate ut, constraint values qout, primal quadratic value
  auto evaluate = [&](const double* th, double* qout) -> double {
    double S[AK_KMAX][AK_KMAX];
#pragma unroll
    for (int a = 0; a < AK_KMAX; ++a)
#pragma unroll
      for (int c2 = 0; c2 < AK_KMAX; ++c2) S[a][c2] = 0.0;
    for (int c = 0; c < mq; ++c) {
      const double t2 = 2.0 * th[c];
#pragma unroll
      for (int a = 0; a < AK_KMAX; ++a)
#pragma unroll
        for (int c2 = 0; c2 < AK_KMAX; ++c2) if (a < k && c2 < k) S[a][c2] += t2 * s_W[c * AK_KMAX + a] * s_W[c * AK_KMAX + c2];
    }
    for (int i = tid; i < n; i += T) {
      double a_[AK_KMAX][AK_KMAX];
#pragma unroll
      for (int a = 0; a < AK_KMAX; ++a)
#pragma unroll
        for (int c2 = 0; c2 < AK_KMAX; ++c2) a_[a][c2] = (a < k && c2 < k) ? H[i * kk + a * k + c2] + S[a][c2] : ((a == c2) ? 1.0 : 0.0);
      ak_inv(a_, k);
#pragma unroll
      for (int a = 0; a < AK_KMAX; ++a) {
        if (a >= k) continue;
        double acc = 0.0;
#pragma unroll
        for (int c2 = 0; c2 < AK_KMAX; ++c2) if (c2 < k) { Hinv[i * kk + a * k + c2] = a_[a][c2]; acc += a_[a][c2] * g[i * k + c2]; }
        u0[i * k + a] = acc;
      }
    }
    __syncthreads();
    for (int e = tid; e < R * R; e += T) {
      const int r1 = e / R, r2 = e - r1 * R;
      if (r2 < r1) continue;
      const int j1 = w.rbj[(size_t)b * w.Rmax + r1], j2 = w.rbj[(size_t)b * w.Rmax + r2];
      double acc = 0.0;
      for (int i = 0; i < n; ++i) { const double x1 = ak_row_x(w, b, r1, i); if (x1 != 0.0) acc += x1 * ak_row_x(w, b, r2, i) * Hinv[i * kk + j1 * k + j2]; }
      G[(size_t)r1 * w.Rmax + r2] = acc; G[(size_t)r2 * w.Rmax + r1] = acc;
    }
    for (int r = tid; r < R; r += T) {
      const int j = w.rbj[(size_t)b * w.Rmax + r];
      double acc = 0.0;
      for (int i = 0; i < n; ++i) acc += ak_row_x(w, b, r, i) * u0[i * k + j];
      cvec[r] = acc - w.rrhs[(size_t)b * w.Rmax + r];
    }
    __syncthreads();
    if (R > 0 && tid < 64) { const int ov = wave_nnqp(G, w.Rmax, cvec, mu, R, s_Gp, s_sv, s_tmp, s_pl, tid); if (tid == 0 && ov) s_ov = 1; }
    __syncthreads();
    double pv = 0.0;
    double qa[AK_QMAX];
#pragma unroll
    for (int c = 0; c < AK_QMAX; ++c) qa[c] = 0.0;
    for (int i = tid; i < n; i += T) {
      double ui[AK_KMAX];
#pragma unroll
      for (int a = 0; a < AK_KMAX; ++a) ui[a] = (a < k) ? u0[i * k + a] : 0.0;
      for (int r = 0; r < R; ++r) {
        const double mv = mu[r];
        if (mv == 0.0) continue;
        const double x = ak_row_x(w, b, r, i);
        if (x == 0.0) continue;
        const int j = w.rbj[(size_t)b * w.Rmax + r];
#pragma unroll
        for (int a = 0; a < AK_KMAX; ++a) if (a < k) ui[a] -= mv * x * Hinv[i * kk + a * k + j];
      }
#pragma unroll
      for (int a = 0; a < AK_KMAX; ++a) {
        if (a >= k) continue;
        ut[i * k + a] = ui[a];
        double hu = 0.0;
#pragma unroll
        for (int c2 = 0; c2 < AK_KMAX; ++c2) if (c2 < k) hu += H[i * kk + a * k + c2] * ui[c2];
        pv += ui[a] * (0.5 * hu - g[i * k + a]);
      }
#pragma unroll
      for (int c = 0; c < AK_QMAX; ++c) {
        if (c >= mq) continue;
        double d = 0.0;
#pragma unroll
        for (int a = 0; a < AK_KMAX; ++a) if (a < k) d += s_W[c * AK_KMAX + a] * ui[a];
        qa[c] += d * d;
      }
    }
#pragma unroll
    for (int c = 0; c < AK_QMAX; ++c) {
      if (c >= mq) continue;
      const double qs = block_sum(qa[c], red);
      if (tid == 0) qout[c] = qs - s_rad[c];
    }
    pv = block_sum(pv, red);
    __syncthreads();
    return pv;
  };
  auto kkt_res = [&](const double* th, const double* q) { double r_ = 0.0; for (int c = 0; c < mq; ++c) r_ = fmax(r_, fabs(fmin(th[c], -q[c]))); return r_; };
  auto dotq = [&](const double* th, const double* q) { double r_ = 0.0; for (int c = 0; c < mq; ++c) r_ += th[c] * q[c]; return r_; };

  double objective_current = 1e10;   // OMC.jl:2012
  int counter = 0, converged = 0, failed = 0;
  while (counter < w.max_iters) {
    ++counter;
    // ---- V-step (model_V, OMC.jl:2192-2209): (sum_{i in O_j} u_i u_i' + U'U / gamma) v_j = sum_{i in O_j} A_ij u_i ---------
    double utu[AK_KMAX][AK_KMAX];
#pragma unroll
    for (int a = 0; a < AK_KMAX; ++a)
#pragma unroll
      for (int c2 = 0; c2 < AK_KMAX; ++c2) {
        double acc = 0.0;
        if (a < k && c2 < k) { for (int i = tid; i < n; i += T) acc += u[i * k + a] * u[i * k + c2]; acc = block_sum(acc, red) / w.gamma; }
        utu[a][c2] = acc;
      }
    for (int j = tid; j < m; j += T) {
      double a_[AK_KMAX][AK_KMAX], rhs[AK_KMAX];
#pragma unroll
      for (int a = 0; a < AK_KMAX; ++a) { rhs[a] = 0.0;
#pragma unroll
        for (int c2 = 0; c2 < AK_KMAX; ++c2) a_[a][c2] = (a < k && c2 < k) ? utu[a][c2] : ((a == c2) ? 1.0 : 0.0); }
      for (int p = w.col_ptr[j]; p < w.col_ptr[j + 1]; ++p) {
        const int i = w.col_idx[p]; const double av = w.col_val[p];
#pragma unroll
        for (int a = 0; a < AK_KMAX; ++a) {
          if (a >= k) continue;
          const double ua = u[i * k + a];
          rhs[a] += av * ua;
#pragma unroll
          for (int c2 = 0; c2 < AK_KMAX; ++c2) if (c2 < k) a_[a][c2] += ua * u[i * k + c2];
        }
      }
      const bool ok = ak_inv(a_, k);
#pragma unroll
      for (int a = 0; a < AK_KMAX; ++a) {
        if (a >= k) continue;
        double acc = 0.0;
#pragma unroll
        for (int c2 = 0; c2 < AK_KMAX; ++c2) if (c2 < k) acc += a_[a][c2] * rhs[c2];
        v[j * k + a] = ok ? acc : 0.0;
      }
    }
    __syncthreads();
    // ---- U-step quantities: H_i = sum_{j in O_i} v_j v_j' + V V' / gamma, g_i = sum_{j in O_i} A_ij v_j --------------------------
    double vvt[AK_KMAX][AK_KMAX];
#pragma unroll
    for (int a = 0; a < AK_KMAX; ++a)
#pragma unroll
      for (int c2 = 0; c2 < AK_KMAX; ++c2) {
        double acc = 0.0;
        if (a < k && c2 < k) { for (int j = tid; j < m; j += T) acc += v[j * k + a] * v[j * k + c2]; acc = block_sum(acc, red) / w.gamma; }
        vvt[a][c2] = acc;
      }
    for (int i = tid; i < n; i += T) {
      double hh[AK_KMAX][AK_KMAX], gg[AK_KMAX];
#pragma unroll
      for (int a = 0; a < AK_KMAX; ++a) { gg[a] = 0.0;
#pragma unroll
        for (int c2 = 0; c2 < AK_KMAX; ++c2) hh[a][c2] = vvt[a][c2]; }
      for (int p = w.row_ptr[i]; p < w.row_ptr[i + 1]; ++p) {
        const int j = w.row_idx[p]; const double av = w.row_val[p];
#pragma unroll
        for (int a = 0; a < AK_KMAX; ++a) {
          if (a >= k) continue;
          const double va = v[j * k + a];
          gg[a] += av * va;
#pragma unroll
          for (int c2 = 0; c2 < AK_KMAX; ++c2) if (c2 < k) hh[a][c2] += va * v[j * k + c2];
        }
      }
#pragma unroll
      for (int a = 0; a < AK_KMAX; ++a) {
        if (a >= k) continue;
        g[i * k + a] = gg[a];
#pragma unroll
        for (int c2 = 0; c2 < AK_KMAX; ++c2) if (c2 < k) H[i * kk + a * k + c2] = hh[a][c2];
      }
    }
    if (tid == 0) for (int c = 0; c < mq; ++c) s_th[c] = 0.0;
    __syncthreads();
    // ---- U-step: projected Newton on the dual ------------------------------------------------------------------------------
    if (tid == 0) s_ov = 0;
    __syncthreads();
    double pv = evaluate(s_th, s_q);
    for (int e = tid; e < n * k; e += T) u[e] = ut[e];
    __syncthreads();
    double dv = pv + dotq(s_th, s_q), res = kkt_res(s_th, s_q), lm = 1e-10;
    for (int nit = 0; nit < 60 && res > 1e-12; ++nit) {
      if (tid == 0) { int na = 0; for (int c = 0; c < mq; ++c) if (s_th[c] > 0.0 || s_q[c] > 0.0) s_act[na++] = c; s_nact = na; }
      __syncthreads();
      const int na = s_nact;
      for (int a = 0; a < na; ++a) {
        const int c = s_act[a];
        const double dl = 1e-7 * fmax(1.0, s_th[c]);
        if (tid == 0) { for (int c2 = 0; c2 < mq; ++c2) s_th2[c2] = s_th[c2]; s_th2[c] += dl; }
        __syncthreads();
        evaluate(s_th2, s_qt);
        if (tid == 0) for (int a2 = 0; a2 < na; ++a2) s_J[a2 * AK_QMAX + a] = (s_qt[s_act[a2]] - s_q[s_act[a2]]) / dl;
        __syncthreads();
      }
      double tr = 0.0;
      if (tid == 0) for (int a = 0; a < na; ++a) for (int a2 = 0; a2 < na; ++a2) s_P[a * AK_QMAX + a2] = -0.5 * (s_J[a * AK_QMAX + a2] + s_J[a2 * AK_QMAX + a]);
      __syncthreads();
      for (int a = 0; a < na; ++a) tr += s_P[a * AK_QMAX + a];
      const double scale = fmax(tr, 1e-300) / na;
      bool accepted = false;
      for (int tr_ = 0; tr_ < 40; ++tr_) {
        if (tid == 0) {   // (P + lm scale I) step = q_act by Cholesky; candidate multipliers
          int ok = 1;
          for (int c = 0; c < na && ok; ++c) {
            double dg = s_P[c * AK_QMAX + c] + lm * scale;
            for (int q2 = 0; q2 < c; ++q2) dg -= s_L[c * AK_QMAX + q2] * s_L[c * AK_QMAX + q2];
            if (!(dg > 0.0)) { ok = 0; break; }
            dg = sqrt(dg); s_L[c * AK_QMAX + c] = dg;
            for (int r2 = c + 1; r2 < na; ++r2) {
              double vv = s_P[r2 * AK_QMAX + c];
              for (int q2 = 0; q2 < c; ++q2) vv -= s_L[r2 * AK_QMAX + q2] * s_L[c * AK_QMAX + q2];
              s_L[r2 * AK_QMAX + c] = vv / dg;
            }
          }
          if (ok) {
            for (int i = 0; i < na; ++i) { double vv = s_q[s_act[i]]; for (int q2 = 0; q2 < i; ++q2) vv -= s_L[i * AK_QMAX + q2] * s_step[q2]; s_step[i] = vv / s_L[i * AK_QMAX + i]; }
            for (int i = na - 1; i >= 0; --i) { double vv = s_step[i]; for (int q2 = i + 1; q2 < na; ++q2) vv -= s_L[q2 * AK_QMAX + i] * s_step[q2]; s_step[i] = vv / s_L[i * AK_QMAX + i]; }
            for (int c = 0; c < mq; ++c) s_th2[c] = s_th[c];
            for (int i = 0; i < na; ++i) s_th2[s_act[i]] = fmax(s_th[s_act[i]] + s_step[i], 0.0);
          }
          s_flag = ok;
        }
        __syncthreads();
        if (!s_flag) { lm *= 10.0; __syncthreads(); continue; }
        const double pvn = evaluate(s_th2, s_qt);
        const double dn = pvn + dotq(s_th2, s_qt), rn = kkt_res(s_th2, s_qt);
        double lin = 0.0;
        for (int c = 0; c < mq; ++c) lin += s_q[c] * (s_th2[c] - s_th[c]);
        const bool ok2 = (dn >= dv + 1e-4 * lin - 1e-14 * fmax(1.0, fabs(dv))) && (dn > dv || rn < res);
        __syncthreads();
        if (ok2) {
          if (tid == 0) for (int c = 0; c < mq; ++c) { s_th[c] = s_th2[c]; s_q[c] = s_qt[c]; }
          for (int e = tid; e < n * k; e += T) u[e] = ut[e];
          __syncthreads();
          res = rn; dv = dn; pv = pvn; lm = fmax(lm * 0.1, 1e-12); accepted = true;
          break;
        }
        lm *= 10.0;
      }
      if (!accepted) break;
    }
    // ---- did model_U have a solution?  (OMC.jl:2231, 2263-2265: a failed solve ends the loop with converged = false) ---------
    {
      double viol = 0.0;
      for (int r = tid; r < R; r += T) {
        const int j = w.rbj[(size_t)b * w.Rmax + r];
        double acc = 0.0;
        for (int i = 0; i < n; ++i) acc += ak_row_x(w, b, r, i) * u[i * k + j];
        viol = fmax(viol, acc - w.rrhs[(size_t)b * w.Rmax + r]);
      }
      for (int c = 0; c < mq; ++c) viol = fmax(viol, s_q[c]);
      viol = block_max(viol, red);
      // s_ov is sticky over the evaluations of this U-step: a rejected trial may have set it, so it is only trusted together with a violation
      const bool bad = !(viol <= AM_FEAS_TOL) || (s_ov && viol > 1e-9);
      __syncthreads();
      if (bad) { failed = 1; break; }
    }
    // ---- objective of model_U (OMC.jl:2232) and the convergence rules (2234-2245, quirk Q3) ---------------------------------
    const double objective_new = pv + 0.5 * w.sumA2;
    if (tid == 0) {
      objs[counter - 1] = objective_new;
      int conv = 0;
      const double diff = fabs((objective_new - objective_current) / objective_current);
      if (diff < w.eps) conv = 1;
      else if (counter > 5) {
        conv = 1;
        for (int i2 = 0; i2 < 5; ++i2) if (!(objs[counter - 1 - i2] > objs[counter - 6])) conv = 0;
      }
      s_stop = conv;
    }
    __syncthreads();
    converged = s_stop;
    __syncthreads();
    if (converged) break;
    objective_current = objective_new;
  }
  for (int e = tid; e < n * k; e += T) { const int i = e % n, a = e / n; w.U[(size_t)b * n * k + e] = u[i * k + a]; }
  for (int e = tid; e < k * m; e += T) w.V[(size_t)b * k * m + e] = v[e];      // k x m column-major = v[j*k + a]
  if (tid == 0) { w.converged[b] = failed ? 0 : converged; w.n_iters[b] = counter; }
  {   // evaluate_objective(U V) from the factors: masked fit over the CSR rows + tr((U'U)(V V')) / (2 gamma)
    __syncthreads();
    double fit = 0.0;
    for (int i = tid; i < n; i += T) {
      for (int p = w.row_ptr[i]; p < w.row_ptr[i + 1]; ++p) {
        const int j = w.row_idx[p];
        double x = 0.0;
        for (int a = 0; a < k; ++a) x += u[i * k + a] * v[j * k + a];
        const double d = x - w.row_val[p]; fit += d * d;
      }
    }
    fit = block_sum(fit, red);
    double reg = 0.0;
    for (int a = 0; a < k; ++a)
      for (int c2 = 0; c2 < k; ++c2) {
        double s1 = 0.0, s2 = 0.0;
        for (int i = tid; i < n; i += T) s1 += u[i * k + a] * u[i * k + c2];
        for (int j = tid; j < m; j += T) s2 += v[j * k + a] * v[j * k + c2];
        s1 = block_sum(s1, red); s2 = block_sum(s2, red);
        reg += s1 * s2;
      }
    if (tid == 0) w.mobj[b] = 0.5 * fit + reg / (2.0 * w.gamma);
  }
}

extern "C" void omc_launch_altmin_k(const void* ws, size_t lds_bytes, hipStream_t s) {
  const AltminWS* w = (const AltminWS*)ws;
  hipLaunchKernelGGL(k_altmin_k, dim3(w->B), dim3(256), lds_bytes, s, *w);
}

extern "C" void omc_launch_altmin(const void* ws, size_t lds_bytes, hipStream_t s) {
  const AltminWS* w = (const AltminWS*)ws;
  hipLaunchKernelGGL(k_altmin, dim3(w->B), dim3(256), lds_bytes, s, *w);
}
extern "C" int omc_altmin_set_lds(void) {
  // k_altmin_k holds ~20 KB of static LDS (Newton scratch, NNQP): its dynamic budget is 128 KB
  if (hipFuncSetAttribute((const void*)k_altmin_k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) (void)hipGetLastError();
  return (int)hipFuncSetAttribute((const void*)k_altmin, hipFuncAttributeMaxDynamicSharedMemorySize, OMC_MAX_DYN_LDS - 8 * 1024);   // ~18 KB static (NNQP scratch)
}
