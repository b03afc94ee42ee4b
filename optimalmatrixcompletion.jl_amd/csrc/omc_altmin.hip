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

__global__ void __launch_bounds__(256) k_altmin(AltminWS w) {
  extern __shared__ double sm[];
  __shared__ double red[32];
  __shared__ double s_Gp[NNQP_PMAX * (NNQP_PMAX + 1) / 2];
  __shared__ double s_sv[NNQP_PMAX], s_tmp[NNQP_PMAX];
  __shared__ int s_pl[NNQP_PMAX];
  __shared__ int s_stop;
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
  int counter = 0, converged = 0;
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
      if (R > 0 && tid < 64) wave_nnqp(G, w.Rmax, cvec, mu, R, s_Gp, s_sv, s_tmp, s_pl, tid);
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
    if (solve(0.0) > 1.0) {                       // ball active (OMC.jl:2164-2171)
      double lo_t = 0.0, hi_t = 1.0;
      for (int i = 0; i < n; ++i) hi_t = fmax(hi_t, fabs(g[i]));
      while (solve(hi_t) > 1.0) hi_t *= 2.0;
      for (int it = 0; it < 200; ++it) {
        const double theta = 0.5 * (lo_t + hi_t);
        if (solve(theta) > 1.0) lo_t = theta; else hi_t = theta;
        if (hi_t - lo_t <= 1e-15 * fmax(1.0, hi_t)) break;
      }
      solve(hi_t);
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
  if (tid == 0) { w.converged[b] = converged; w.n_iters[b] = counter; }
}

extern "C" void omc_launch_altmin(const void* ws, size_t lds_bytes, hipStream_t s) {
  const AltminWS* w = (const AltminWS*)ws;
  hipLaunchKernelGGL(k_altmin, dim3(w->B), dim3(256), lds_bytes, s, *w);
}
extern "C" int omc_altmin_set_lds(void) {
  return (int)hipFuncSetAttribute((const void*)k_altmin, hipFuncAttributeMaxDynamicSharedMemorySize, OMC_MAX_DYN_LDS);
}
