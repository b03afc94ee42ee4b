// omc_device.hip -- CDNA4 (gfx950) kernels of the node-relaxation engine.
//
// Hot path = per-B&B-node relaxation of OptimalMatrixCompletion.jl (OMC.jl:1431-1943) restated as
//     min f(Y) = 1/2 sum_j a_j'(I + gamma Y[O_j,O_j])^-1 a_j   over  {[Y U;U' I]>=0, Y<=I, trY<=k, rows}
// and solved by a consensus ADMM.  Only v = X'U enters the rows, so with Q an orthonormal
// basis of the row functionals the order-(n+k) cone is replaced by 0 <= Y <= I plus the order-(r+k) cone
// [Q'YQ Vt; Vt' I] >= 0, Vt = Q'U.  One ADMM iteration is four kernel classes, each batched over B nodes:
//   k_colprox : one WAVE per (node, column)  -- secular-equation prox of one column block (Cholesky + Newton)
//   k_cone    : one WORKGROUP per node       -- LDS-resident one-sided Jacobi eigensolver + spectral clip to [0,1]
//   k_small   : one WORKGROUP per node       -- the small cone (order r+k), same Jacobi routine
//   k_global  : one WORKGROUP per node       -- consensus average, exact projection on the linear rows (NNQP),
//                                               dual updates, residuals
// plus k_check* (certificate: exact f(Y), Lagrangian dual bound) every `check_every` iterations.
// Wavefront = 64 everywhere.  No CUDA compatibility paths.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include "omc_device.h"

#include "omc_wave.h"

// slot handled by workgroup / wave group i of a per-iteration launch

// optional phase stamps (diagnostic builds only: -DOMC_STAMPS): block 0 accumulates s_memtime deltas per phase
#ifdef OMC_STAMPS
#define STAMP(slot) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) { long long t_ = __builtin_amdgcn_s_memtime(); w.stamps[slot] += (double)(t_ - t_prev_); t_prev_ = t_; } } while (0)
#define STAMP_BEGIN() long long t_prev_ = __builtin_amdgcn_s_memtime()
// per-slot diagnostics: diag[cls * B + b] += v  (w.stamps + 32)
#define DIAG_T0() long long t_diag0_ = __builtin_amdgcn_s_memtime()
#define DIAG_ADD(cls, b, v) atomicAdd(&w.stamps[32 + (size_t)(cls) * w.B + (b)], (double)(v))
#define DIAG_CYC(cls, b) DIAG_ADD(cls, b, __builtin_amdgcn_s_memtime() - t_diag0_)
#define WSTAMP_BEGIN() long long tw_prev_ = __builtin_amdgcn_s_memtime()
#define WSTAMP_BEGIN2() tw_prev_ = __builtin_amdgcn_s_memtime()
#define WSTAMP(slot) do { if (b == 0 && j == 0 && lane == 0) { long long t_ = __builtin_amdgcn_s_memtime(); w.stamps[slot] += (double)(t_ - tw_prev_); tw_prev_ = t_; } } while (0)
#else
#define WSTAMP_BEGIN() do {} while (0)
#define WSTAMP_BEGIN2() do {} while (0)
#define WSTAMP(slot) do {} while (0)
#define STAMP(slot) do {} while (0)
#define STAMP_BEGIN() do {} while (0)
#define DIAG_T0() do {} while (0)
#define DIAG_ADD(cls, b, v) do {} while (0)
#define DIAG_CYC(cls, b) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------------------
// k_setup: initial iterate + Gram matrix (for rho = 1) of the linear rows in the metric of the consensus weights
//   rows of node b:  <CY_r, Y> + <CU_r, U> <= rhs_r ;  CY_r = I (trace) | x x' (cut) | 0 ;  CU_r = x (x) coef_r
//   (bound / cut rows) or a single entry (box rows).   G1_rr' = <CY_r, CY_r'>_{1/wY1} + <CU_r, CU_r'> / 2
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double rowU_entry(const OmcWS& w, int nb, int r, int i, int j) {
  // coefficient of U[i][j] in row r of node nb (descriptor index, not the slot)
  const int kind = w.rkind[(size_t)nb * w.Rmax + r];
  if (kind == ROW_TRACE) return 0.0;
  if (kind == ROW_BOX) {
    return (w.rbi[(size_t)nb * w.Rmax + r] == i && w.rbj[(size_t)nb * w.Rmax + r] == j)
               ? w.rcoef[((size_t)nb * w.Rmax + r) * w.k]
               : 0.0;
  }
  const int l = w.rcut[(size_t)nb * w.Rmax + r];
  return w.cutx[((size_t)nb * w.Lmax + l) * w.n + i] * w.rcoef[((size_t)nb * w.Rmax + r) * w.k + j];
}

__global__ void k_setup(OmcWS w) {
  if (!w.init[blockIdx.x]) return;                 // only slots that received a new node
  const int nb = w.node_of[blockIdx.x];
  const int b = blockIdx.x, n = w.n, k = w.k, tid = threadIdx.x, T = blockDim.x, rm = w.rmax;
  __shared__ double red[32];
  double* Y = w.Y + (size_t)b * n * n;
  double* Yp = w.Yp + (size_t)b * n * n;
  const double d0 = (double)k / (double)n;
  // warm start: the parent's final state (pool entry lf) instead of the cold start.  The child keeps its own base penalty, so the parent's scaled
  // duals are rescaled by rho_parent / rho_child; Yp = Y; U = Q_child Q_child' U_parent (the parent's rows are a prefix of the child's, so this
  // is the parent's Vt padded with a zero for the new direction); small-cone duals and row multipliers start from zero; the tracked block of
  // the cone follows (it is the eigen-space of Y - D1, which the rescaling only changes when the parent's penalty had been bumped)
  const int lf = w.load_from ? w.load_from[nb] : -1;
  const bool warm = lf >= 0;
  const int NP = w.np16;
  const int rc = w.rr[nb];
  const double* Qc = w.Qb + (size_t)nb * n * rm;
  if (warm) {
    const double fw = w.pscal[(size_t)lf * 4] / w.rho_node[nb];
    double fr2 = 0.0, tr1 = 0.0;
    for (int e = tid; e < n * n; e += T) {
      const int i = e % n, j = e / n;
      const double y = w.pY[(size_t)lf * n * n + e], d1 = w.pD1[(size_t)lf * n * n + e] * fw;
      Y[e] = y; Yp[e] = y; if (w.Yx) w.Yx[(size_t)b * n * n + e] = y;
      w.D1[(size_t)b * n * n + e] = d1; w.D3[(size_t)b * n * n + e] = w.pD3[(size_t)lf * n * n + e] * fw; w.E3[(size_t)b * n * n + e] = 0.0;
      const double mv = y - d1;
      w.Mbuf[(size_t)b * NP * NP + (size_t)j * NP + i] = mv; fr2 += mv * mv; if (i == j) tr1 += mv;
    }
    for (int e = tid; e < NP * NP; e += T) {
      const int i = e % NP, j = e / NP;
      if (i >= n || j >= n) w.Mbuf[(size_t)b * NP * NP + e] = 0.0;
      w.Vrow[(size_t)b * NP * NP + e] = 0.0;
    }
    for (int e = tid; e < NP * 16; e += T) w.Xs[(size_t)b * NP * 16 + e] = w.pXs[(size_t)lf * NP * 16 + e];
    if (tid < 16) w.sub_theta[(size_t)b * 16 + tid] = w.ptheta[(size_t)lf * 16 + tid];
    fr2 = block_sum(fr2, red); tr1 = block_sum(tr1, red);
    if (tid == 0) {
      w.fro2[b] = fr2; w.trM[b] = tr1; w.sub_on[b] = (w.sub_enable && w.pscal[(size_t)lf * 4 + 1] != 0.0) ? 1 : 0; w.sub_wait[b] = 0; w.sub_nfail[b] = 0; w.cone_done[b] = 0; w.v3valid[b] = 0; w.sub_onC[b] = 0; w.confirm[b] = 1; w.lb_est[b] = -1e300; w.vvalid[b] = 0; if (w.vvalidC) w.vvalidC[b] = 0; if (w.ws_first) w.ws_first[b] = w.sub_on[b] ? 0 : 1;
    }
    // Vt = Q_child' U_parent, U = Q_child Vt
    for (int e = tid; e < rm * k; e += T) {
      const int a = e % rm, j = e / rm;
      double v = 0.0;
      if (a < rc) for (int i = 0; i < n; ++i) v += Qc[(size_t)a * n + i] * w.pU[(size_t)lf * n * k + (size_t)j * n + i];
      w.Vt[(size_t)b * rm * k + e] = v; w.D3V[(size_t)b * rm * k + e] = 0.0;
      w.W3V[(size_t)b * rm * k + e] = 0.0; w.Q3V[(size_t)b * rm * k + e] = 0.0;
    }
    __syncthreads();
    __threadfence_block();
    for (int e = tid; e < n * k; e += T) {
      const int i = e % n, j = e / n;
      double v = 0.0;
      for (int a = 0; a < rc; ++a) v += Qc[(size_t)a * n + i] * w.Vt[(size_t)b * rm * k + (size_t)j * rm + a];
      w.U[(size_t)b * n * k + e] = v;
    }
    for (int e = tid; e < w.nnz; e += T) w.alpha[(size_t)b * w.nnz + e] = w.palpha[(size_t)lf * w.nnz + e];
    for (int e = tid; e < w.m; e += T) w.sval[(size_t)b * w.m + e] = w.psval[(size_t)lf * w.m + e];
  } else {
  for (int e = tid; e < n * n; e += T) {
    int i = e % n, j = e / n;
    double v = (i == j) ? d0 : 0.0;
    Y[e] = v; Yp[e] = v; if (w.Yx) w.Yx[(size_t)b * n * n + e] = v;
    w.D1[(size_t)b * n * n + e] = 0.0; w.D3[(size_t)b * n * n + e] = 0.0; w.E3[(size_t)b * n * n + e] = 0.0;
  }
  {
    for (int e = tid; e < NP * NP; e += T) {
      int i = e % NP, j = e / NP;
      w.Mbuf[(size_t)b * NP * NP + e] = (i == j && i < n) ? d0 : 0.0;
      w.Vrow[(size_t)b * NP * NP + e] = 0.0;
    }
    if (tid == 0) {
      w.fro2[b] = d0 * d0 * n; w.trM[b] = d0 * n; w.sub_on[b] = 0; w.sub_wait[b] = 0; w.sub_nfail[b] = 0; w.cone_done[b] = 0; w.v3valid[b] = 0; w.sub_onC[b] = 0; w.confirm[b] = 1; w.lb_est[b] = -1e300; w.vvalid[b] = 0; if (w.vvalidC) w.vvalidC[b] = 0; if (w.ws_first) w.ws_first[b] = w.sub_on[b] ? 0 : 1;
    }
  }
  for (int e = tid; e < n * k; e += T) w.U[(size_t)b * n * k + e] = 0.0;
  for (int e = tid; e < rm * k; e += T) {
    w.Vt[(size_t)b * rm * k + e] = 0.0; w.D3V[(size_t)b * rm * k + e] = 0.0;
    w.W3V[(size_t)b * rm * k + e] = 0.0; w.Q3V[(size_t)b * rm * k + e] = 0.0;
  }
  for (int e = tid; e < w.nnz; e += T) w.alpha[(size_t)b * w.nnz + e] = 0.0;
  for (int e = tid; e < w.m; e += T) w.sval[(size_t)b * w.m + e] = -1.0;
  }
  for (int e = tid; e < k * k; e += T) {
    w.D3T[(size_t)b * k * k + e] = 0.0; w.Q3T[(size_t)b * k * k + e] = 0.0;
    w.W3T[(size_t)b * k * k + e] = ((e % k) == (e / k)) ? 1.0 : 0.0;
  }
  const int R = w.R[nb];
  for (int e = tid; e < w.Rmax; e += T) w.lam[(size_t)b * w.Rmax + e] = 0.0;
  if (tid == 0) {
    w.init[b] = 0; w.rho_b[b] = w.rho_node[nb];
    if (w.accel) { w.aa_valid[b] = 0; w.aa_hist[b] = 0; w.aa_head[b] = 0; w.aa_pending[b] = 0; w.aa_nacc[b] = 0; w.aa_nrej[b] = 0; }
    w.done[b] = 0; w.rowov[b] = 0; w.gap_prev[b] = 1e300; w.gap_rate[b] = 1.0; w.slow_votes[b] = 0; w.status[b] = OMC_ST_SLOW; w.iters[b] = 0; w.stall[b] = 0; w.nbump[b] = 0; w.lastbump[b] = 0; w.bfac[b] = 1.0;
    w.obj[b] = 1e300; w.objout[b] = 1e300; w.objprev[b] = 1e300; w.lbprev[b] = -1e300; w.lb[b] = -1e300; w.rp[b] = 1e300; w.rd[b] = 1e300;
  }
  // Gram matrix for rho = 1
  double* G = w.G + (size_t)b * w.Rmax * w.Rmax;
  for (int r = 0; r < R; ++r) {
    const int kr = w.rkind[(size_t)nb * w.Rmax + r];
    for (int s = r; s < R; ++s) {
      const int ks = w.rkind[(size_t)nb * w.Rmax + s];
      double acc = 0.0;
      const bool yr = (kr == ROW_TRACE || kr == ROW_CUT), ys = (ks == ROW_TRACE || ks == ROW_CUT);
      if (yr && ys) {
        const double* xr = (kr == ROW_CUT) ? w.cutx + ((size_t)nb * w.Lmax + w.rcut[(size_t)nb * w.Rmax + r]) * n : nullptr;
        const double* xs = (ks == ROW_CUT) ? w.cutx + ((size_t)nb * w.Lmax + w.rcut[(size_t)nb * w.Rmax + s]) * n : nullptr;
        if (!xr && !xs) {
          for (int i = tid; i < n; i += T) acc += 1.0 / w.wY1[(size_t)i * n + i];
        } else if (!xr || !xs) {
          const double* x = xr ? xr : xs;
          for (int i = tid; i < n; i += T) acc += x[i] * x[i] / w.wY1[(size_t)i * n + i];
        } else {
          for (int e = tid; e < n * n; e += T) {
            int i = e % n, j = e / n;
            acc += xr[i] * xr[j] * xs[i] * xs[j] / w.wY1[e];
          }
        }
      }
      if (kr != ROW_TRACE && ks != ROW_TRACE) {
        for (int e = tid; e < n * k; e += T) {
          int i = e % n, j = e / n;
          double a = rowU_entry(w, nb, r, i, j);
          if (a != 0.0) acc += a * rowU_entry(w, nb, s, i, j) * 0.5;
        }
      }
      double tot = block_sum(acc, red);
      if (tid == 0) { G[(size_t)r * w.Rmax + s] = tot; G[(size_t)s * w.Rmax + r] = tot; }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// k_colprox: one wave per (node b, column j).
//   mode 0 (prox):   B = I + gamma*((2Y - Yp)[O,O] - gamma/(2 rho_f) a_old a_old'),  cp = gamma^2/(2 rho_f)
//                    find s >= 0:  || (B + cp s I)^-1 a ||^2 = s   (phi convex decreasing -> Newton from the left)
//                    alpha = (B + cp s I)^-1 a
//   mode 1 (exact):  B = I + gamma*Y[O,O], alpha = B^-1 a ;  obj += 1/2 a'alpha ; c0 += a'alpha - 1/2||alpha||^2
// The c x c matrix lives in LDS (c <= CP_LDS_C) or in a per-wave global scratch slab.
// ---------------------------------------------------------------------------------------------------------
template <class PT>
__device__ __forceinline__ void colprox_body(const OmcWS& w, int mode, int b, int j, int off, int c, int lane, PT base) {
  const int n = w.n;
  const size_t tri = (size_t)c * (c + 1) / 2;
  auto Bm = base;
  auto Lm = Bm + tri;
  auto va = Lm + tri;
  auto vy = va + c;
  auto vz = vy + c;
  auto vo = vz + c;
  const int* idx = w.col_idx + off;
  const double g = w.gamma;
  const double* Y = w.Y + (size_t)b * n * n;
  const double* Yp = w.Yp + (size_t)b * n * n;
  double* alpha = ((mode == 0) ? w.alpha : w.alphaX) + (size_t)b * w.nnz + off;
  for (int p = lane; p < c; p += WAVE) {
    va[p] = w.col_val[off + p];
    vo[p] = (mode == 0) ? alpha[p] : 0.0;
  }
  WAVE_SYNC();
  const double rho_f = w.rho_b[b] * w.rho_f_ratio;
  const double coef = (mode == 0) ? g / (2.0 * rho_f) : 0.0;
  for (int e = lane; e < c * c; e += WAVE) {
    int p = e / c, q = e - p * c;
    if (q > p) continue;
    size_t a = (size_t)idx[q] * n + idx[p];
    double yv = (mode == 0) ? (w.Yx ? w.Yx[(size_t)b * n * n + a] : (2.0 * Y[a] - Yp[a])) : Y[a];
    double v = g * (yv - coef * vo[p] * vo[q]);
    if (p == q) v += 1.0;
    Bm[TRI(p, q)] = v;
  }
  WAVE_SYNC();
  double s = 0.0;
  const bool regpath = (c <= WAVE);        // vectors in registers (lane r owns entry r)
  auto dinv = vz;                          // reuse: reciprocals of the Cholesky diagonal
  const double a_reg = (lane < c) ? va[lane] : 0.0;
  auto solve2 = [&](double& y_out, double& z_out, bool want_z) {
    if (regpath) {
      for (int p = lane; p < c; p += WAVE) dinv[p] = 1.0 / Lm[TRI(p, p)];
      WAVE_SYNC();
      y_out = wave_chol_solve_reg(Lm, dinv, c, a_reg, lane);
      z_out = want_z ? wave_chol_solve_reg(Lm, dinv, c, y_out, lane) : 0.0;
    } else {
      wave_chol_solve(Lm, c, va, vy, lane);
      if (want_z) wave_chol_solve(Lm, c, vy, vz, lane);
    }
  };
  if (mode == 0) {
    const double cp = g * g / (2.0 * rho_f);
    double sprev = w.sval[(size_t)b * w.m + j];
    s = (sprev > 0.0) ? sprev : 0.0;
    double lo = 0.0, hi = -1.0;  // hi < 0: unknown
    bool lo_valid = false;       // lo_valid: Cholesky succeeded at lo and phi(lo) >= 0
    double yr = 0.0, zr = 0.0;
    bool have_final = false;
    int nfact_ = 0; (void)nfact_;
    for (int it = 0; it < 60; ++it) {
      ++nfact_;
      for (int e = lane; e < c * c; e += WAVE) {
        int p = e / c, q = e - p * c;
        if (q <= p) Lm[TRI(p, q)] = Bm[TRI(p, q)] + ((p == q) ? cp * s : 0.0);
      }
      WAVE_SYNC();
      bool ok = wave_cholesky(Lm, c, lane);
      if (!ok) {  // s below the PD range: move right
        lo = s; lo_valid = false;
        s = (hi > 0.0) ? 0.5 * (s + hi) : (2.0 * s + 1.0);
        continue;
      }
      solve2(yr, zr, true);
      double yy = 0.0, yz = 0.0;
      if (regpath) { yy = yr * yr; yz = yr * zr; }
      else for (int p = lane; p < c; p += WAVE) { yy += vy[p] * vy[p]; yz += vy[p] * vz[p]; }
      yy = wave_sum(yy); yz = wave_sum(yz);
      const double ph = yy - s, dph = -2.0 * cp * yz - 1.0;
      if (ph >= 0.0) { lo = s; lo_valid = true; } else { hi = s; }
      double sn = s - ph / dph;
      if (!(sn > lo) && !lo_valid) sn = 0.5 * (lo + s);
      if (sn < lo) sn = lo;
      if (hi > 0.0 && sn > hi) sn = 0.5 * (lo + hi);
      // alpha(s) is Lipschitz in s with constant << 1 here: once the Newton step is below 1e-13 |s| the current solve IS the answer
      if (fabs(sn - s) <= 1e-13 * fmax(1.0, fabs(s))) { have_final = true; break; }
      s = sn;
    }
    if (!have_final) {
      ++nfact_;
      for (int e = lane; e < c * c; e += WAVE) {
        int p = e / c, q = e - p * c;
        if (q <= p) Lm[TRI(p, q)] = Bm[TRI(p, q)] + ((p == q) ? cp * s : 0.0);
      }
      WAVE_SYNC();
      wave_cholesky(Lm, c, lane);
      solve2(yr, zr, false);
    }
    if (lane == 0) w.sval[(size_t)b * w.m + j] = s;
    if (lane == 0) DIAG_ADD(1, b, nfact_);
    double* lamD = w.lamD + ((size_t)b * w.m + j) * n;      // dense copy (zeros off the support) for the output-stationary Lambda Lambda'
    if (regpath) { if (lane < c) { alpha[lane] = yr; lamD[idx[lane]] = yr; } }
    else for (int p = lane; p < c; p += WAVE) { alpha[p] = vy[p]; lamD[idx[p]] = vy[p]; }
  } else {
    for (int e = lane; e < c * c; e += WAVE) {
      int p = e / c, q = e - p * c;
      if (q <= p) Lm[TRI(p, q)] = Bm[TRI(p, q)];
    }
    WAVE_SYNC();
    bool ok = wave_cholesky(Lm, c, lane);
    if (!ok) {  // Y not PSD enough on this block: report +inf objective contribution
      if (lane == 0) { w.objcol[(size_t)b * w.m + j] = 1e300; w.c0col[(size_t)b * w.m + j] = 0.0; }
      return;
    }
    double yr = 0.0, zr = 0.0;
    solve2(yr, zr, false);
    double aa = 0.0, al2 = 0.0;
    double* lamX = w.lamDX ? w.lamDX + ((size_t)b * w.m + j) * n : nullptr;
    if (regpath) { aa = a_reg * yr; al2 = yr * yr; if (lane < c) { alpha[lane] = yr; if (lamX) lamX[idx[lane]] = yr; } }
    else for (int p = lane; p < c; p += WAVE) { aa += va[p] * vy[p]; al2 += vy[p] * vy[p]; alpha[p] = vy[p]; if (lamX) lamX[idx[p]] = vy[p]; }
    aa = wave_sum(aa); al2 = wave_sum(al2);
    if (lane == 0) { w.objcol[(size_t)b * w.m + j] = 0.5 * aa; w.c0col[(size_t)b * w.m + j] = aa - 0.5 * al2; }   // summed in a fixed order by k_check_build
  }
}


// Fast path for c <= 64 (every lane holds one vector entry in a register).  Root-free Cholesky with G = 64/c lanes per row
// (omc_wave.h), Halley steps on the secular equation (phi'' = 6 cp^2 z'z comes for free with the Newton data) and a
// second-order Taylor finish  alpha(s + d) = y - cp d z + cp^2 d^2 w  once the step is small (relative third-order term
// < 1e-15): one factorization per call late in the ADMM run, two early, instead of the three of Newton + confirmation.
template <class PT>
__device__ __forceinline__ void colprox_reg(const OmcWS& w, int mode, int b, int j, int off, int c, int lane, PT base) {
  const int n = w.n;
  const int tri = (c * (c + 1)) >> 1;
  // dense instances (columns with more than 40 observed rows): B is not kept beside L -- a second factorization gathers it again from
  // L2 -- which halves the LDS per wave and doubles the waves per CU
  const bool keepB = w.cp_keepB != 0;
  auto Bm = base;
  auto Lm = keepB ? Bm + tri : base;
  auto vo = Lm + tri;
  auto pinv = vo + c;
  auto sidx = (int*)(pinv + c);
  const double gm = w.gamma;
  const double* Y = w.Y + (size_t)b * n * n;
  const double* Yp = w.Yp + (size_t)b * n * n;
  const double* Yx = w.Yx ? w.Yx + (size_t)b * n * n : nullptr;
  double* alpha = ((mode == 0) ? w.alpha : w.alphaX) + (size_t)b * w.nnz + off;
  WSTAMP_BEGIN();
  const double a_reg = (lane < c) ? w.col_val[off + lane] : 0.0;
  if (lane < c) { vo[lane] = (mode == 0) ? alpha[lane] : 0.0; sidx[lane] = w.col_idx[off + lane]; }
  WAVE_SYNC();
  const double rho_f = w.rho_b[b] * w.rho_f_ratio;
  const double coef = (mode == 0) ? gm / (2.0 * rho_f) : 0.0;
  auto gather = [&](decltype(Lm) dst, double diag_shift) {
    // gather of the packed lower triangle: flat index e = r (r + 1) / 2 + q over all 64 lanes, four entries per trip with every
    // global load issued before the first use (unguarded, clamped addresses), so that the L2 latencies overlap
    for (int e0 = lane; e0 < tri; e0 += 4 * WAVE) {
      int rr[4], qq[4]; bool vv[4]; double y1[4], y2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * WAVE;
        vv[u] = e < tri;
        const int ec = vv[u] ? e : 0;
        int ri = (int)((__builtin_sqrtf(8.0f * (float)ec + 1.0f) - 1.0f) * 0.5f);
        while (tri_i(ri, 0) > ec) --ri;
        while (tri_i(ri + 1, 0) <= ec) ++ri;
        rr[u] = ri; qq[u] = ec - tri_i(ri, 0);
        const size_t a = (size_t)sidx[qq[u]] * n + sidx[ri];
        y1[u] = (mode == 0 && Yx) ? Yx[a] : Y[a];
        y2[u] = (mode == 0 && !Yx) ? Yp[a] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (vv[u]) {
          const double yv = (mode == 0 && !Yx) ? (2.0 * y1[u] - y2[u]) : y1[u];
          double v = gm * (yv - coef * vo[rr[u]] * vo[qq[u]]);
          if (rr[u] == qq[u]) v += 1.0 + diag_shift;
          dst[e0 + u * WAVE] = v;
        }
      }
    }
  };
  gather((mode == 0 && keepB) ? Bm : Lm, 0.0);
  const int my = (lane < c) ? sidx[lane] : 0;
  WSTAMP(21);
  if (mode == 0) {
    const double cp = gm * gm / (2.0 * rho_f);
    const double sprev = w.sval[(size_t)b * w.m + j];
    double s = (sprev > 0.0) ? sprev : 0.0;
    double lo = 0.0, hi = -1.0;  // hi < 0: unknown
    bool lo_valid = false;       // the factorization succeeded at lo and phi(lo) >= 0
    double yr = 0.0;
    bool fin = false;
    bool first_ = true;      // without the copy of B: L still holds the gathered B for the first factorization
    int nfact_ = 0; (void)nfact_;
    for (int it = 0; it < 60; ++it) {
      ++nfact_;
      WAVE_SYNC();
      { if (keepB) { for (int e = lane; e < tri; e += WAVE) Lm[e] = Bm[e]; WAVE_SYNC(); if (lane < c) Lm[tri_i(lane, lane)] += cp * s; } else if (first_) { WAVE_SYNC(); if (lane < c) Lm[tri_i(lane, lane)] += cp * s; } else gather(Lm, cp * s); first_ = false; }
      WSTAMP(22);
      const bool ok_ = wave_ldl(Lm, pinv, c, lane);
      WSTAMP(23);
      if (!ok_) {  // s below the positive definite range: move right
        lo = s; lo_valid = false;
        s = (hi > 0.0) ? 0.5 * (s + hi) : (2.0 * s + 1.0);
        continue;
      }
      yr = wave_ldl_solve_reg(Lm, pinv, c, a_reg, lane);
      const double zr = wave_ldl_solve_reg(Lm, pinv, c, yr, lane);
      WSTAMP(24);
      const double yy = wave_sum(yr * yr), yz = wave_sum(yr * zr), zz = wave_sum(zr * zr);
      const double ph = yy - s, dph = -2.0 * cp * yz - 1.0, ddph = 6.0 * cp * cp * zz;
      if (ph >= 0.0) { lo = s; lo_valid = true; } else { hi = s; }
      const double den = 2.0 * dph * dph - ph * ddph;
      double sn = s + ((den > dph * dph) ? (-2.0 * ph * dph / den) : (-ph / dph));
      bool guarded = false;
      if (!(sn > lo) && !lo_valid) { sn = 0.5 * (lo + s); guarded = true; }
      if (sn < lo) { sn = lo; guarded = true; }
      if (hi > 0.0 && sn > hi) { sn = 0.5 * (lo + hi); guarded = true; }
      const double d = sn - s;
      if (fabs(d) <= 1e-13 * fmax(1.0, fabs(s))) { fin = true; break; }      // the current solve is the answer
      if (!guarded && yy > 0.0 && cp * fabs(d) * sqrt(zz / yy) < 1e-5) {
        const double wr = wave_ldl_solve_reg(Lm, pinv, c, zr, lane);
        yr = yr - cp * d * (zr - cp * d * wr);
        s = sn; fin = true;
        WSTAMP(25);
#ifdef OMC_STAMPS
        {  // accuracy audit of the Taylor finish: re-factor at the accepted s and compare (diag 6 = max relative error of
           // alpha, diag 7 = max |phi(s)| / s)
          WAVE_SYNC();
          { if (keepB) { for (int e = lane; e < tri; e += WAVE) Lm[e] = Bm[e]; WAVE_SYNC(); if (lane < c) Lm[tri_i(lane, lane)] += cp * s; } else if (first_) { WAVE_SYNC(); if (lane < c) Lm[tri_i(lane, lane)] += cp * s; } else gather(Lm, cp * s); first_ = false; }
          wave_ldl(Lm, pinv, c, lane);
          const double ye = wave_ldl_solve_reg(Lm, pinv, c, a_reg, lane);
          double e1 = fabs(ye - yr), e2 = fabs(ye);
          for (int o = 32; o > 0; o >>= 1) { e1 = fmax(e1, __shfl_xor(e1, o, WAVE)); e2 = fmax(e2, __shfl_xor(e2, o, WAVE)); }
          const double phe = fabs(wave_sum(ye * ye) - s) / fmax(s, 1e-300);
          if (lane == 0) {
            atomicMax((unsigned long long*)&w.stamps[32 + (size_t)6 * w.B + b], (unsigned long long)__double_as_longlong(e1 / fmax(e2, 1e-300)));
            atomicMax((unsigned long long*)&w.stamps[32 + (size_t)7 * w.B + b], (unsigned long long)__double_as_longlong(phe));
          }
        }
#endif
        WSTAMP_BEGIN2();
        break;
      }
      s = sn;
      WSTAMP(26);
    }
    if (!fin) {
      ++nfact_;
      WAVE_SYNC();
      { if (keepB) { for (int e = lane; e < tri; e += WAVE) Lm[e] = Bm[e]; WAVE_SYNC(); if (lane < c) Lm[tri_i(lane, lane)] += cp * s; } else if (first_) { WAVE_SYNC(); if (lane < c) Lm[tri_i(lane, lane)] += cp * s; } else gather(Lm, cp * s); first_ = false; }
      wave_ldl(Lm, pinv, c, lane);
      yr = wave_ldl_solve_reg(Lm, pinv, c, a_reg, lane);
    }
    if (lane == 0) w.sval[(size_t)b * w.m + j] = s;
    if (lane == 0) DIAG_ADD(1, b, nfact_);
    double* lamD = w.lamD + ((size_t)b * w.m + j) * n;      // dense copy (zeros off the support) for the output-stationary Lambda Lambda'
    if (lane < c) { alpha[lane] = yr; lamD[my] = yr; }
    WSTAMP(27);
  } else {
    if (!wave_ldl(Lm, pinv, c, lane)) {  // Y not PSD enough on this block: report +inf objective contribution
      if (lane == 0) { w.objcol[(size_t)b * w.m + j] = 1e300; w.c0col[(size_t)b * w.m + j] = 0.0; }
      return;
    }
    const double yr = wave_ldl_solve_reg(Lm, pinv, c, a_reg, lane);
    if (lane < c) { alpha[lane] = yr; if (w.lamDX) w.lamDX[((size_t)b * w.m + j) * n + my] = yr; }
    const double aa = wave_sum(a_reg * yr), al2 = wave_sum(yr * yr);
    if (lane == 0) { w.objcol[(size_t)b * w.m + j] = 0.5 * aa; w.c0col[(size_t)b * w.m + j] = aa - 0.5 * al2; }   // summed in a fixed order by k_check_build
  }
}

__global__ void __launch_bounds__(256) k_colprox(OmcWS w, int mode) {
  extern __shared__ double smem[];
  const int wave_in_blk = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  const int gw = blockIdx.x * wpb + wave_in_blk;  // global wave id
  const int ncol = (mode == 0 && w.cp_pair) ? w.cp_nsolo : w.m;      // mode 0 with k_colprox_pair: the columns it leaves (more than 32 rows, unpaired last column)
  const int bl = gw / ncol, jj = gw - bl * ncol;
  if (bl >= w.nB) return;
  const int j = (mode == 0 && w.cp_pair) ? w.cp_solo[jj] : jj;
  const int b = slot_of(w, bl);
  if (w.done[b]) return;
  const int off = w.col_ptr[j], c = w.col_ptr[j + 1] - off;
  if (c == 0) { if (mode == 1 && lane == 0) { w.objcol[(size_t)b * w.m + j] = 0.0; w.c0col[(size_t)b * w.m + j] = 0.0; } return; }
  DIAG_T0();
  // two inlined copies so that the LDS copy compiles to ds_read/ds_write (not flat) instructions
  if (c <= w.cp_lds_c) colprox_reg(w, mode, b, j, off, c, lane, smem + (size_t)wave_in_blk * w.cp_lds_doubles);   // cp_lds_c <= 64
  else colprox_body(w, mode, b, j, off, c, lane, w.cp_scratch + ((size_t)b * w.m + j) * w.cp_scratch_stride);
  if (lane == 0 && mode == 0) DIAG_CYC(0, b);
}

// ---------------------------------------------------------------------------------------------------------
// k_cone: symmetric eigendecomposition by one-sided (Hestenes) Jacobi on the SHIFTED matrix M' = M + sigma I,
// sigma = 1.5 ||M||_F, so that M' is positive definite with condition <= 5.  Columns g_t of G = M' V are then
// orthogonal with ||g_t|| = lambda_t + sigma and v_t = g_t / ||g_t||: no eigenvector accumulation, one N x N
// array, LDS-resident for N <= ~136.  Parallel (round-robin) ordering: Np/2 disjoint column pairs per step, each
// handled by `lpp` lanes that split the rows, reduce the three inner products with wave shuffles and apply the
// rotation from registers.
//   mode CONE_CLIP01: M = Y - D1 (n x n)         -> W1 = V clip(lambda,0,1) V'   (the cone block 0 <= Y <= I)
//   mode CONE_EVALS : M = Mchk (n x n)           -> sum of min(lam_i,0) over the k smallest -> evsum[b]
//   mode CONE_SEP   : M = U U' - Y               -> two smallest eigenpairs -> lmin[2b..], bx
// ---------------------------------------------------------------------------------------------------------
#define JROWS 20  // rows cached in registers per lane
#define WS_JROWS 32  // same for the warm-started kernel's run-time-bound variant (n <= 16 * 32 = 512)

__device__ __forceinline__ void rr_pair(int step, int t, int Np, int& p, int& q) {
  // round-robin tournament on Np (even) players: step in [0,Np-1), t in [0,Np/2)
  const int M1 = Np - 1;
  if (t == 0) { p = step; q = Np - 1; }
  else {
    p = step + t; if (p >= M1) p -= M1;          // step < M1, t < M1: one conditional subtraction replaces the modulo
    q = step - t; if (q < 0) q += M1;
  }
  if (p > q) { int tmp = p; p = q; q = tmp; }
}

// One-sided Jacobi with cached squared column norms (only the cross product needs a reduction), DPP reductions,
// rotations applied from registers, one barrier per step.  nrm2: Np doubles (LDS).  Returns the sweeps done.
template <int LPP>
__device__ __forceinline__ int jacobi_sweeps_t(double* Gm, double* nrm2, int Nr, int Np, int ld, double tau, int max_sweeps) {
  const int tid = threadIdx.x, T = blockDim.x;
  const int ngroups = T / LPP, grp = tid / LPP, lg = tid % LPP;
  const int npairs = Np >> 1;
  const int wv = tid >> 6, lane = tid & 63, nw = T >> 6;
  const double tau2 = tau * tau;
  int sweeps = 0;
  for (; sweeps < max_sweeps; ++sweeps) {
    // refresh the squared norms (analytic updates drift)
    for (int t = wv; t < Np; t += nw) {
      double a = 0.0;
      if (t < Nr) for (int r = lane; r < Nr; r += WAVE) { double x = Gm[(size_t)t * ld + r]; a += x * x; }
      a = wave_sum(a);
      if (lane == 0) nrm2[t] = a;
    }
    __syncthreads();
    int rotated = 0;
    for (int step = 0; step < Np - 1; ++step) {
      for (int pr = grp; pr < npairs; pr += ngroups) {
        int p, q;
        rr_pair(step, pr, Np, p, q);
        double* gp = Gm + (size_t)p * ld;
        double* gq = Gm + (size_t)q * ld;
        double cp_[JROWS], cq_[JROWS];
        double gm = 0.0;
#pragma unroll
        for (int i = 0; i < JROWS; ++i) {
          const int r = lg + i * LPP;
          double x = 0.0, y = 0.0;
          if (r < Nr) { x = gp[r]; y = gq[r]; }
          cp_[i] = x; cq_[i] = y;
          gm += x * y;
        }
        gm = group_sum_dpp<LPP>(gm);
        const double a = nrm2[p], bb = nrm2[q];
        if (gm * gm > tau2 * a * bb && a > 0.0 && bb > 0.0) {
          // tangent of the rotation angle: only its accuracy relative to 1 matters (c^2 + s^2 = 1 holds by construction)
          const double zeta = (bb - a) / (2.0 * gm);
          const double az = fabs(zeta);
          double tt = 1.0 / (az + sqrt(1.0 + az * az));
          tt = (zeta >= 0.0) ? tt : -tt;
          const double cs = rsqrt(1.0 + tt * tt), sn = cs * tt;
#pragma unroll
          for (int i = 0; i < JROWS; ++i) {
            const int r = lg + i * LPP;
            if (r < Nr) { gp[r] = cs * cp_[i] - sn * cq_[i]; gq[r] = sn * cp_[i] + cs * cq_[i]; }
          }
          if (lg == 0) { nrm2[p] = a - tt * gm; nrm2[q] = bb + tt * gm; }
          rotated = 1;
        }
      }
      __syncthreads();
    }
    if (!__syncthreads_or(rotated)) { ++sweeps; break; }
  }
  return sweeps;
}

// The same sweep for tiny matrices (Np <= 16), entirely inside wave 0: 8 pair groups of 8 lanes, each lane owns rows lg and lg + 8,
// wave barriers instead of workgroup barriers (a 12 x 12 small-cone matrix took 117 us with __syncthreads per step, ~10 us so).
__device__ __forceinline__ int jacobi_sweeps_wave16(double* Gm, double* nrm2, int Nr, int Np, int ld, double tau, int max_sweeps) {
  const int lane = threadIdx.x & 63, grp = lane >> 3, lg = lane & 7;
  const int npairs = Np >> 1;
  const double tau2 = tau * tau;
  int sweeps = 0;
  for (; sweeps < max_sweeps; ++sweeps) {
    if (lane < Np) {
      double a = 0.0;
      if (lane < Nr) for (int r = 0; r < Nr; ++r) { const double x = Gm[(size_t)lane * ld + r]; a += x * x; }
      nrm2[lane] = a;
    }
    WAVE_SYNC();
    int rotated = 0;
    for (int step = 0; step < Np - 1; ++step) {
      if (grp < npairs) {
        int p, q;
        rr_pair(step, grp, Np, p, q);
        double* gp = Gm + (size_t)p * ld;
        double* gq = Gm + (size_t)q * ld;
        const bool r0 = lg < Nr, r1 = lg + 8 < Nr;
        const double x0 = r0 ? gp[lg] : 0.0, x1 = r1 ? gp[lg + 8] : 0.0, y0 = r0 ? gq[lg] : 0.0, y1 = r1 ? gq[lg + 8] : 0.0;
        const double gm = group_sum_dpp<8>(x0 * y0 + x1 * y1);
        const double a = nrm2[p], bb = nrm2[q];
        if (gm * gm > tau2 * a * bb && a > 0.0 && bb > 0.0) {
          const double zeta = (bb - a) / (2.0 * gm);
          const double az = fabs(zeta);
          double tt = 1.0 / (az + sqrt(1.0 + az * az));
          tt = (zeta >= 0.0) ? tt : -tt;
          const double cs = rsqrt(1.0 + tt * tt), sn = cs * tt;
          if (r0) { gp[lg] = cs * x0 - sn * y0; gq[lg] = sn * x0 + cs * y0; }
          if (r1) { gp[lg + 8] = cs * x1 - sn * y1; gq[lg + 8] = sn * x1 + cs * y1; }
          if (lg == 0) { nrm2[p] = a - tt * gm; nrm2[q] = bb + tt * gm; }
          rotated = 1;
        }
      }
      WAVE_SYNC();
    }
    if (!__any(rotated)) { ++sweeps; break; }
  }
  return sweeps;
}

// Variant for the Rayleigh-Ritz matrix of k_cone_sub (order 16, nearly diagonal): tangent in fp32 (its error only slows the last
// digits; cosine / sine in fp64 keep the rotation orthogonal) and the sweep loop ends as soon as every relative cross product of a
// sweep was below sqrt(tau) -- the rotations of that sweep leave them below tau (quadratic convergence).
__device__ __forceinline__ int jacobi16_fast(double* Gm, double* nrm2, int ld, double tau, int max_sweeps) {
  const int lane = threadIdx.x & 63, grp = lane >> 3, lg = lane & 7;
  const double tau2 = tau * tau;
  int sweeps = 0;
  for (; sweeps < max_sweeps; ++sweeps) {
    if (lane < 16) {
      double a = 0.0;
#pragma unroll
      for (int r = 0; r < 16; ++r) { const double x = Gm[(size_t)lane * ld + r]; a += x * x; }
      nrm2[lane] = a;
    }
    WAVE_SYNC();
    int big = 0;
    for (int step = 0; step < 15; ++step) {
      int p, q;
      rr_pair(step, grp, 16, p, q);
      double* gp = Gm + (size_t)p * ld;
      double* gq = Gm + (size_t)q * ld;
      const double x0 = gp[lg], x1 = gp[lg + 8], y0 = gq[lg], y1 = gq[lg + 8];
      const double gm = group_sum_dpp<8>(x0 * y0 + x1 * y1);
      const double a = nrm2[p], bb = nrm2[q];
      const double g2 = gm * gm, ab = a * bb;
      if (g2 > tau2 * ab && ab > 0.0) {
        const float df = (float)(bb - a), gf = (float)gm;
        const float rtf = __builtin_sqrtf(df * df + 4.0f * gf * gf);
        const float den = (df >= 0.0f) ? (df + rtf) : (df - rtf);
        const double tt = (den != 0.0f) ? (double)((2.0f * gf) / den) : 0.0;
        const double cs = rsqrt(1.0 + tt * tt), sn = cs * tt;
        gp[lg] = cs * x0 - sn * y0; gq[lg] = sn * x0 + cs * y0;
        gp[lg + 8] = cs * x1 - sn * y1; gq[lg + 8] = sn * x1 + cs * y1;
        if (lg == 0) { nrm2[p] = a - tt * gm; nrm2[q] = bb + tt * gm; }
        if (g2 > tau * ab) big = 1;
      }
      WAVE_SYNC();
    }
    if (!__any(big)) { ++sweeps; break; }
  }
  return sweeps;
}

// generic fall-back (any N, rows not cached): used when a lane would own more than JROWS rows
__device__ __forceinline__ int jacobi_onesided(double* Gm, int Nr, int Np, int ld, int lpp, double tau, int max_sweeps, int* s_cnt) {
  const int tid = threadIdx.x, T = blockDim.x;
  const int ngroups = T / lpp, grp = tid / lpp, lg = tid % lpp;
  const int npairs = Np >> 1;
  int sweeps = 0;
  for (; sweeps < max_sweeps; ++sweeps) {
    if (tid == 0) *s_cnt = 0;
    __syncthreads();
    for (int step = 0; step < Np - 1; ++step) {
      for (int pr = grp; pr < npairs; pr += ngroups) {
        int p, q;
        rr_pair(step, pr, Np, p, q);
        double* gp = Gm + (size_t)p * ld;
        double* gq = Gm + (size_t)q * ld;
        double a = 0.0, bb = 0.0, gm = 0.0;
        for (int r = lg; r < Nr; r += lpp) { double x = gp[r], y = gq[r]; a += x * x; bb += y * y; gm += x * y; }
        a = group_sum(a, lpp); bb = group_sum(bb, lpp); gm = group_sum(gm, lpp);
        if (fabs(gm) > tau * sqrt(a * bb) && a > 0.0 && bb > 0.0) {
          const double zeta = (bb - a) / (2.0 * gm);
          const double tt = ((zeta >= 0.0) ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
          for (int r = lg; r < Nr; r += lpp) { double x = gp[r], y = gq[r]; gp[r] = cs * x - sn * y; gq[r] = sn * x + cs * y; }
          if (lg == 0) atomicAdd(s_cnt, 1);
        }
      }
      __syncthreads();
    }
    int cnt = *s_cnt;
    __syncthreads();
    if (cnt == 0) { ++sweeps; break; }
  }
  return sweeps;
}

__device__ __forceinline__ double cone_M_entry(const OmcWS& w, int b, int mode, int i, int j) {
  const int n = w.n, k = w.k;
  if (mode == CONE_CLIP01) {
    return w.Y[(size_t)b * n * n + (size_t)j * n + i] - w.D1[(size_t)b * n * n + (size_t)j * n + i];
  } else if (mode == CONE_EVALS) {
    return w.Mchk[(size_t)b * n * n + (size_t)j * n + i];
  } else if (mode == CONE_TOPK) {
    return w.Y[(size_t)b * n * n + (size_t)j * n + i];
  } else if (mode == CONE_BIG) {
    return w.Mbuf[(size_t)b * w.np16 * w.np16 + (size_t)j * w.np16 + i];
  } else {  // CONE_SEP: U U' - Y
    double s = 0.0;
    for (int t = 0; t < k; ++t) s += w.U[(size_t)b * n * k + (size_t)t * n + i] * w.U[(size_t)b * n * k + (size_t)t * n + j];
    return s - w.Y[(size_t)b * n * n + (size_t)j * n + i];
  }
}

// Rebuild  out = base*M + sum_{s<nsel} wgt[s] * g_sel[s] g_sel[s]'   in 4x4 register tiles over the lower triangle.
// `entry(i,j)` returns the original M (only evaluated when base != 0).
template <class EntryF, class StoreF>
__device__ __forceinline__ void spectral_rebuild(const double* Gm, int ld, int N, const int* sel, const double* wgt, int nsel, double base,
                                 EntryF entry, StoreF store) {
  const int tid = threadIdx.x, T = blockDim.x;
  const int nt = (N + 3) >> 2;
  const int ntiles = nt * (nt + 1) / 2;
  for (int tile = tid; tile < ntiles; tile += T) {
    int ti = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > tile) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
    const int tj = tile - ti * (ti + 1) / 2;
    const int i0 = ti * 4, j0 = tj * 4;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][c] = 0.0;
    for (int s = 0; s < nsel; ++s) {
      const double* gt = Gm + (size_t)sel[s] * ld;
      const double ws = wgt[s];
      double gi[4], gj[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) { gi[a] = (i0 + a < N) ? gt[i0 + a] : 0.0; gj[a] = (j0 + a < N) ? gt[j0 + a] * ws : 0.0; }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] += gi[a] * gj[c];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = i0 + a, j = j0 + c;
        if (i < N && j < N && i >= j) {
          double Mv = entry(i, j);
          store(i, j, base * Mv + acc[a][c], Mv);
        }
      }
  }
}

// common front end: load the symmetrised matrix into Gm (Np x ld), shift, Jacobi, squared column norms -> ev
template <class EntryF>
__device__ __forceinline__ double eig_frontend(double* Gm, double* ev, int N, int Np, int ld, EntryF entry, double* red, int* s_cnt, int* sweeps_out, double tau_in = 1e-14) {
  const int tid = threadIdx.x, T = blockDim.x;
  double fro = 0.0;
  for (int e = tid; e < Np * Np; e += T) {
    int i = e % Np, j = e / Np;
    double v = 0.0;
    if (i < N && j < N) { v = 0.5 * (entry(i, j) + entry(j, i)); fro += v * v; }
    Gm[(size_t)j * ld + i] = v;
  }
  fro = sqrt(block_sum(fro, red));
  const double sigma = 1.5 * fro + 1e-300;
  for (int i = tid; i < N; i += T) Gm[(size_t)i * ld + i] += sigma;
  __syncthreads();
  int lpp = 64;
  while (lpp > 1 && lpp * (Np >> 1) > T) lpp >>= 1;
  if (lpp > 16) lpp = 16;
  const double tau = tau_in;
  int sweeps = 0;
  if (Np <= 16) {
    if (tid < WAVE) sweeps = jacobi_sweeps_wave16(Gm, ev, N, Np, ld, tau, 30);
    __syncthreads();
  } else if (lpp >= 4 && (N + lpp - 1) / lpp <= JROWS) {
    // ev doubles as the norm cache during the sweeps (recomputed below)
    if (lpp == 16) sweeps = jacobi_sweeps_t<16>(Gm, ev, N, Np, ld, tau, 30);
    else if (lpp == 8) sweeps = jacobi_sweeps_t<8>(Gm, ev, N, Np, ld, tau, 30);
    else sweeps = jacobi_sweeps_t<4>(Gm, ev, N, Np, ld, tau, 30);
  } else {
    sweeps = jacobi_onesided(Gm, N, Np, ld, lpp, tau, 30, s_cnt);
  }
  if (sweeps_out && tid == 0) *sweeps_out += sweeps;
  const int wv = tid >> 6, lane = tid & 63, nw = T >> 6;
  for (int t = wv; t < N; t += nw) {
    double a = 0.0;
    for (int r = lane; r < N; r += WAVE) { double x = Gm[(size_t)t * ld + r]; a += x * x; }
    a = wave_sum(a);
    if (lane == 0) ev[t] = a;
  }
  __syncthreads();
  return sigma;
}

template <bool USE_LDS>
__global__ void __launch_bounds__(512) k_cone(OmcWS w, int mode) {
  extern __shared__ double smem[];
  __shared__ double red[32];
  __shared__ int s_cnt;
  __shared__ int s_nsel;
  __shared__ double s_base;
  const int b = slot_of(w, blockIdx.x), tid = threadIdx.x, T = blockDim.x;
  if (w.done[b] && mode != CONE_SEP && mode != CONE_TOPK) return;
  if ((mode == CONE_SEP || mode == CONE_TOPK) && !w.fin[b]) return;
  if (mode == CONE_SEP && w.sep_done && w.sep_done[b]) return;      // k_cone_sub<2> has delivered the separation vector
  const int n = w.n, k = w.k;
  const int N = n;
  const int Np = (N + 1) & ~1, ld = Np | 1;
  // with USE_LDS the pointers stay in the LDS address space (ds_read/ds_write, not flat)
  auto Gm = [&]() { if constexpr (USE_LDS) return (double*)smem; else return w.cone_scratch + (size_t)b * w.cone_scratch_stride; }();
  auto ev = Gm + (size_t)Np * ld;
  auto wgt = ev + Np;
  int* sel = (int*)(wgt + Np);
  auto entry = [&](int i, int j) { return cone_M_entry(w, b, mode, i, j); };
  // eigenvalues only (certificate): their error is second order in the residual cross products, 1e-9 is ample
  const double sigma = eig_frontend(Gm, ev, N, Np, ld, entry, red, &s_cnt, w.sweeps + b, (mode == CONE_EVALS) ? 1e-9 : 1e-14);
  if (mode == CONE_EVALS) {
    if (tid == 0) {
      double best[8]; int kk = (k < 8) ? k : 8;
      for (int i = 0; i < kk; ++i) best[i] = 1e300;
      for (int t = 0; t < N; ++t) {
        double lamv = sqrt(ev[t]) - sigma;
        for (int i = 0; i < kk; ++i) if (lamv < best[i]) { double tmp = best[i]; best[i] = lamv; lamv = tmp; }
      }
      double s = 0.0;
      for (int i = 0; i < kk; ++i) s += fmin(best[i], 0.0);
      w.evsum[b] = s;
    }
    return;
  }
  if (mode == CONE_TOPK) {
    // svd(Y).U[:, 1:k] of the symmetric PSD Y (OMC.jl:873): eigenvectors of the k largest eigenvalues, canonical sign
    __shared__ int s_top[8];
    __shared__ double s_sg[8];
    if (tid == 0) {
      for (int j = 0; j < k; ++j) {
        int bi = -1; double bl = -1e300;
        for (int t = 0; t < N; ++t) {
          bool used = false;
          for (int q = 0; q < j; ++q) if (s_top[q] == t) used = true;
          double lamv = sqrt(ev[t]) - sigma;
          if (!used && lamv > bl) { bl = lamv; bi = t; }
        }
        s_top[j] = bi;
        const double* gt = Gm + (size_t)bi * ld;
        int a1 = 0;
        for (int r = 1; r < N; ++r) if (fabs(gt[r]) > fabs(gt[a1])) a1 = r;
        s_sg[j] = ((gt[a1] >= 0.0) ? 1.0 : -1.0) / sqrt(ev[bi]);
      }
    }
    __syncthreads();
    for (int e = tid; e < N * k; e += T) {
      int r = e % N, j = e / N;
      w.U[(size_t)b * n * k + e] = s_sg[j] * Gm[(size_t)s_top[j] * ld + r];
    }
    return;
  }
  if (mode == CONE_SEP) {
    __shared__ int s_i1, s_i2;
    __shared__ double s_sg1, s_sg2, s_w1, s_w2;
    if (tid == 0) {
      int i1 = 0, i2 = -1; double l1 = 1e300, l2 = 1e300;
      for (int t = 0; t < N; ++t) {
        double lamv = sqrt(ev[t]) - sigma;
        if (lamv < l1) { l2 = l1; i2 = i1; l1 = lamv; i1 = t; }
        else if (lamv < l2) { l2 = lamv; i2 = t; }
      }
      if (i2 < 0) { i2 = i1; l2 = l1; }
      s_i1 = i1; s_i2 = i2;
      w.lmin[2 * b] = l1; w.lmin[2 * b + 1] = l2;
      const double* g1 = Gm + (size_t)i1 * ld; const double* g2 = Gm + (size_t)i2 * ld;
      int a1 = 0, a2 = 0;
      for (int r = 1; r < N; ++r) { if (fabs(g1[r]) > fabs(g1[a1])) a1 = r; if (fabs(g2[r]) > fabs(g2[a2])) a2 = r; }
      s_sg1 = ((g1[a1] >= 0.0) ? 1.0 : -1.0) / sqrt(ev[i1]);
      s_sg2 = ((g2[a2] >= 0.0) ? 1.0 : -1.0) / sqrt(ev[i2]);
      if (w.breakpoints == 2 && l2 < -1e-10) {       // OMC.jl:2471-2473
        double nn = sqrt(l1 * l1 + l2 * l2);
        s_w1 = fabs(l1) / nn; s_w2 = fabs(l2) / nn;
      } else { s_w1 = 1.0; s_w2 = 0.0; }
    }
    __syncthreads();
    for (int r = tid; r < N; r += T)
      w.bx[(size_t)b * n + r] = s_w1 * s_sg1 * Gm[(size_t)s_i1 * ld + r] + s_w2 * s_sg2 * Gm[(size_t)s_i2 * ld + r];
    return;
  }
  // CONE_CLIP01 / CONE_BIG: W1 = V clip(lambda, 0, hi) V'.  Either rebuild the defect (lambda<0 or >hi) on top of M, or the kept part.
  const double hi = w.clip_hi;
  if (tid == 0) {
    int ndef = 0, nkeep = 0;
    for (int t = 0; t < N; ++t) {
      double lamv = sqrt(ev[t]) - sigma;
      if (lamv < 0.0 || lamv > hi) ++ndef;
      if (lamv > 0.0) ++nkeep;
    }
    int c = 0;
    if (ndef <= nkeep) {
      for (int t = 0; t < N; ++t) {
        double nu2 = ev[t], lamv = sqrt(nu2) - sigma;
        if (lamv < 0.0) { sel[c] = t; wgt[c] = -lamv / nu2; ++c; }
        else if (lamv > hi) { sel[c] = t; wgt[c] = -(lamv - hi) / nu2; ++c; }
      }
      s_base = 1.0;
    } else {
      for (int t = 0; t < N; ++t) {
        double nu2 = ev[t], lamv = sqrt(nu2) - sigma;
        if (lamv > 0.0) { sel[c] = t; wgt[c] = fmin(lamv, hi) / nu2; ++c; }
      }
      s_base = 0.0;
    }
    s_nsel = c;
  }
  __syncthreads();
  double* Wout = w.W1 + (size_t)b * n * n;
  auto entry2 = [&](int i, int j) { return 0.5 * (cone_M_entry(w, b, mode, i, j) + cone_M_entry(w, b, mode, j, i)); };
  auto store = [&](int i, int j, double v, double) { Wout[(size_t)j * n + i] = v; Wout[(size_t)i * n + j] = v; };
  spectral_rebuild(Gm, ld, N, sel, wgt, s_nsel, s_base, entry2, store);
}

// ---------------------------------------------------------------------------------------------------------
// k_cone_ws: the hot kernel.  Spectral clip of M = Y - D1 to [0,1] with a WARM-STARTED one-sided Jacobi:
//   G = (M + sigma I) V_prev   by v_mfma_f64_16x16x4_f64 (operands straight from L2, result tile -> LDS),
//   V_prev = eigenvectors of the previous ADMM iteration, so the columns of G are already nearly orthogonal and
//   one or two sweeps restore orthogonality (quadratic convergence; a sweep whose largest relative cross product
//   is < 1e-7 is the last one).  Rows are padded to LPP*rpl so the register-cached row loops are wave-uniform.
// Inputs written by k_global / k_setup: Mbuf (NP16 x NP16, zero padded), fro2, Vrow (row-major V), vvalid.
// ---------------------------------------------------------------------------------------------------------
typedef double double4v __attribute__((ext_vector_type(4)));
#define SUBP 16   // tracked subspace dimension of k_cone_sub
#define SUBG 2    // default of w.sub_guard: Ritz values that must stay negative (4 until round 3: nodes whose Y - D1 keeps 13-14 positive eigenvalues -- ~1 % of the slot-iterations at depth 11 -- then sent EVERY launch through the full kernel)

template <int LPP, bool USE_LDS, int RPL2, int TPB = 512>   // RPL2 = rows per lane / 2 as a compile-time constant (0: run-time bound)
__global__ void __launch_bounds__(TPB) k_cone_ws(OmcWS w) {
  extern __shared__ double smem[];
  __shared__ int s_nsel, s_nkeep, s_top[SUBP];
  __shared__ double s_base;
  const int b = slot_of(w, blockIdx.x), tid = threadIdx.x, T = blockDim.x;
  if (w.done[b]) return;
  if (!w.ws_mode && w.sub_enable) {
    if (w.ws_phase == 1) { if (!w.ws_first[b]) return; }          // the slots known to need the full kernel, beside k_cone_sub
    else if (w.cone_done[b]) return;                              // k_cone_sub (or phase 1) has already written W1 for this iteration
  }
  if (w.ws_mode && w.cert_enable && !w.confirm[b]) return;       // the estimate of k_cone_sub<1> is enough for this check
  const int n = w.n, N = n, NP = w.np16;
  const int Np = (N + 1) & ~1;
  // lane lg of a pair group owns the CONTIGUOUS rows [lg*rpl, (lg+1)*rpl): 16-byte LDS reads, rpl even, ld even
  const int rpl = (((N + LPP - 1) / LPP) + 1) & ~1, Nrp = rpl * LPP;
  // leading dimension = 16 (mod 32) doubles and rows interleaved over the lanes of a pair group (lane lg owns the 16-byte chunks
  // lg, lg + LPP, ...): a group then reads 128 contiguous bytes = 32 banks, and the groups of neighbouring pairs (columns p, p + 1)
  // start 32 banks apart -- the contiguous-rows layout with ld = Nrp + 2 lost 36 % of the LDS cycles to bank conflicts (PMC)
  const int ld = w.ws_ld;
  auto Gm = [&]() { if constexpr (USE_LDS) return (double*)smem; else return w.cone_scratch + (size_t)b * w.cone_scratch_stride; }();
  auto ev = Gm + (size_t)Np * ld;
  auto wgt = ev + Np;
  int* sel = (int*)(wgt + Np);
  const int evals_only = w.ws_mode;   // 1: certificate matrix (k_check_build), return sum of min(lambda_i, 0) over the k smallest
  const double* Mb = (evals_only ? w.MbufC : w.Mbuf) + (size_t)b * NP * NP;
  double* Vr = (evals_only ? w.VrowC : w.Vrow) + (size_t)b * NP * NP;
  int* vvalid = evals_only ? w.vvalidC : w.vvalid;
  const double sigma = 1.5 * sqrt(evals_only ? w.fro2c[b] : w.fro2[b]) + 1e-300;
  const int wv = tid >> 6, lane = tid & 63, nw = T >> 6;
  STAMP_BEGIN();
  DIAG_T0();
  // ---- 1. G = (M + sigma I) V_prev  (or M + sigma I on the first call) ------------------------------------
  STAMP(5);
  for (int e = tid; e < Np * ld; e += T) Gm[e] = 0.0;
  __syncthreads();
  STAMP(6);
  if (!vvalid[b]) {
    for (int e = tid; e < N * N; e += T) {
      int i = e % N, j = e / N;
      Gm[(size_t)j * ld + i] = Mb[(size_t)j * NP + i] + ((i == j) ? sigma : 0.0);
    }
  } else {
    const int nt = NP >> 4;
    const int li = lane & 15, lk = lane >> 4;
    // software pipeline of depth PF k-steps with NO conditionals inside the K loop (a guarded load makes hipcc drain
    // vmcnt before every MFMA): K runs over the whole zero-padded NP (a multiple of 16 = 4*PF), the prefetch index wraps
    constexpr int PF = 4;
    for (int tile = wv; tile < nt * nt; tile += nw) {
      const int ti = tile % nt, tj = tile / nt;
      const int i0 = ti << 4, j0 = tj << 4;
      double4v acc = {0.0, 0.0, 0.0, 0.0};
      const int ia = i0 + li, jb = j0 + li;
      const double* Ma = Mb + ia;
      const double* Vb = Vr + jb;
      double aq[PF], bq[PF];
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int kk = 4 * u + lk;
        aq[u] = Ma[(size_t)kk * NP];
        bq[u] = Vb[(size_t)kk * NP];
      }
      for (int k0 = 0; k0 < NP; k0 += 4 * PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          // the shift is added when the operand is CONSUMED: touching the prefetched value earlier would stall on it
          const double a = aq[u] + ((ia == k0 + 4 * u + lk) ? sigma : 0.0), bv = bq[u];
          int kn = k0 + 4 * (PF + u) + lk;
          kn = (kn < NP) ? kn : kn - NP;                 // wrapped prefetch of the last chunk is never used
          aq[u] = Ma[(size_t)kn * NP];
          bq[u] = Vb[(size_t)kn * NP];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc, 0, 0, 0);
        }
      }
      if (jb < N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = i0 + lk + 4 * r; if (row < N) Gm[(size_t)jb * ld + row] = acc[r]; }
      }
    }
  }
  __syncthreads();
  STAMP(0);
  // ---- 2. sweeps ---------------------------------------------------------------------------------------------
  {
    const int ngroups = T / LPP, grp = tid / LPP, lg = tid % LPP;
    const int npairs = Np >> 1;
    // rotation threshold 1e-10 (relative cross product): the projection error it leaves, ~1e-10 ||M||, is far below the
    // ADMM tolerances; 1e-14 would force a second rotating sweep that changes nothing the solver can see
    const double tau = (w.jacobi_tau > 0.0) ? w.jacobi_tau : 1e-10, tau2 = tau * tau;
    int sweeps = 0;
    for (; sweeps < w.max_sweeps; ++sweeps) {
      for (int t = wv; t < Np; t += nw) {
        double a = 0.0;
        for (int r = lane; r < Nrp; r += WAVE) { double x = Gm[(size_t)t * ld + r]; a += x * x; }
        a = wave_sum(a);
        if (lane == 0) ev[t] = a;
      }
      __syncthreads();
      int big = 0;
      for (int step = 0; step < Np - 1; ++step) {
        for (int pr = grp; pr < npairs; pr += ngroups) {
          int p, q;
          rr_pair(step, pr, Np, p, q);
          typedef double double2v __attribute__((ext_vector_type(2)));
          auto gp = (double2v*)(Gm + (size_t)p * ld) + lg;
          auto gq = (double2v*)(Gm + (size_t)q * ld) + lg;
          const int rpl2 = rpl >> 1;
          constexpr int R2 = RPL2 ? RPL2 : WS_JROWS / 2;
          double2v cp_[R2], cq_[R2];
          double gm0 = 0.0, gm1 = 0.0;
#pragma unroll
          for (int i = 0; i < R2; ++i) {
            if (RPL2 || i < rpl2) { cp_[i] = gp[i * LPP]; cq_[i] = gq[i * LPP]; gm0 += cp_[i].x * cq_[i].x; gm1 += cp_[i].y * cq_[i].y; }
          }
          double gm = (LPP == 64) ? wave_sum(gm0 + gm1) : group_sum_dpp<(LPP == 64) ? 16 : LPP>(gm0 + gm1);
          const double a = ev[p], bb = ev[q];
          const double g2 = gm * gm, ab = a * bb;
          if (g2 > tau2 * ab && ab > 0.0) {
            // tangent in fp32 (its error only slows the last digits of convergence: the next visit of the pair sees a
            // cross product reduced by ~1e-7 instead of 1e-16), cosine/sine in fp64 so the rotation stays orthogonal
            const float df = (float)(bb - a), gf = (float)gm;
            const float rtf = __builtin_sqrtf(df * df + 4.0f * gf * gf);
            const float den = (df >= 0.0f) ? (df + rtf) : (df - rtf);
            const double tt = (den != 0.0f) ? (double)((2.0f * gf) / den) : 0.0;
            const double cs = rsqrt(1.0 + tt * tt), sn = cs * tt;
#pragma unroll
            for (int i = 0; i < R2; ++i) {
              if (RPL2 || i < rpl2) {
                double2v np_, nq_;
                np_.x = cs * cp_[i].x - sn * cq_[i].x; np_.y = cs * cp_[i].y - sn * cq_[i].y;
                nq_.x = sn * cp_[i].x + cs * cq_[i].x; nq_.y = sn * cp_[i].y + cs * cq_[i].y;
                gp[i * LPP] = np_; gq[i * LPP] = nq_;
              }
            }
            if (lg == 0) { ev[p] = a - tt * gm; ev[q] = bb + tt * gm; }
#ifdef OMC_STAMPS
            if (lg == 0 && b == 0 && !evals_only) atomicAdd(&w.stamps[32 + (size_t)6 * w.B + (sweeps < 8 ? sweeps : 7)], 1.0);   // rotations per sweep index (slot 0)
#endif
            if (g2 > tau * ab) big = 1;   // relative cross product above sqrt(tau): one more sweep needed
          }
        }
        __syncthreads();
      }
      if (!__syncthreads_or(big)) { ++sweeps; break; }   // all cross products were < 1e-7 relative: now < 1e-14
    }
    if (tid == 0) w.sweeps[b] += sweeps;
#ifdef OMC_STAMPS
    if (tid == 0 && b == 0 && !evals_only) atomicAdd(&w.stamps[32 + (size_t)7 * w.B + (sweeps < 8 ? sweeps : 7)], 1.0);   // histogram of sweeps per call (slot 0)
#endif
  }
  STAMP(1);
  // ---- 3. squared norms, eigenvectors for the next call ---------------------------------------------------------
  for (int t = wv; t < N; t += nw) {
    double a = 0.0;
    for (int r = lane; r < Nrp; r += WAVE) { double x = Gm[(size_t)t * ld + r]; a += x * x; }
    a = wave_sum(a);
    if (lane == 0) ev[t] = a;
  }
  __syncthreads();
  for (int e = tid; e < N * N; e += T) {
    int t = e % N, kk = e / N;     // consecutive threads -> consecutive t -> contiguous row-major store
    Vr[(size_t)kk * NP + t] = Gm[(size_t)t * ld + kk] * rsqrt(ev[t]);
  }
  if (tid == 0) vvalid[b] = 1;
  STAMP(2);
  if (evals_only) {
    if (w.cert_enable && N >= 3 * SUBP) {     // seed the certificate block with the SUBP most negative eigenpairs (largest of -Mchk)
      double* lam_s = (double*)(sel + Np + (Np & 1));
      for (int t = tid; t < N; t += T) lam_s[t] = sqrt(ev[t]) - sigma;
      __syncthreads();
      if (tid == 0) {
        for (int j = 0; j < SUBP; ++j) {
          int bi = -1; double bl = 1e300;
          for (int t = 0; t < N; ++t) {
            bool used = false;
            for (int q = 0; q < j; ++q) if (s_top[q] == t) used = true;
            if (!used && lam_s[t] < bl) { bl = lam_s[t]; bi = t; }
          }
          s_top[j] = bi;
          w.sub_thetaC[(size_t)b * SUBP + j] = -bl;
        }
        w.sub_onC[b] = 1;
      }
      __syncthreads();
      double* Xg = w.XsC + (size_t)b * NP * SUBP;
      for (int e = tid; e < SUBP * NP; e += T) {
        const int j = e / NP, r = e - j * NP;
        Xg[e] = (r < N) ? Gm[(size_t)s_top[j] * ld + r] * rsqrt(ev[s_top[j]]) : 0.0;
      }
    }
    if (tid == 0) {
      const int k = w.k;
      double best[8]; const int kk2 = (k < 8) ? k : 8;
      for (int i = 0; i < kk2; ++i) best[i] = 1e300;
      for (int t = 0; t < N; ++t) {
        double lamv = sqrt(ev[t]) - sigma;
        for (int i = 0; i < kk2; ++i) if (lamv < best[i]) { const double tmp = best[i]; best[i] = lamv; lamv = tmp; }
      }
      double s2 = 0.0;
      for (int i = 0; i < kk2; ++i) s2 += fmin(best[i], 0.0);
      w.evsum[b] = s2;
    }
    return;
  }
  // ---- 4. clip: rebuild either the defect or the kept part --------------------------------------------------------
  // eigenvalues in parallel (the square roots), then one lane compacts the selected columns (integer work only)
  double* lamv_s = (double*)(sel + Np + (Np & 1));
  for (int t = tid; t < N; t += T) lamv_s[t] = sqrt(ev[t]) - sigma;
  __syncthreads();
  const double hi = w.clip_hi;
  if (tid == 0) {
    int ndef = 0, nkeep = 0;
    for (int t = 0; t < N; ++t) {
      const double lamv = lamv_s[t];
      if (lamv < 0.0 || lamv > hi) ++ndef;
      if (lamv > 0.0) ++nkeep;
    }
    s_nkeep = nkeep;
    if (w.sub_debug == 3) { const int mn = (ndef < nkeep) ? ndef : nkeep; atomicAdd(&w.stamps[32 + ((ndef < 31) ? ndef : 31)], 1.0); atomicAdd(&w.stamps[64 + ((mn < 31) ? mn : 31)], 1.0); }   // diagnostics
    int c = 0;
    if (ndef <= nkeep) {
      for (int t = 0; t < N; ++t) {
        const double lamv = lamv_s[t];
        if (lamv < 0.0) { sel[c] = t; wgt[c] = -lamv; ++c; }
        else if (lamv > hi) { sel[c] = t; wgt[c] = -(lamv - hi); ++c; }
      }
      s_base = 1.0;
    } else {
      for (int t = 0; t < N; ++t) {
        const double lamv = lamv_s[t];
        if (lamv > 0.0) { sel[c] = t; wgt[c] = fmin(lamv, hi); ++c; }
      }
      s_base = 0.0;
    }
    s_nsel = c;
  }
  __syncthreads();
  // seed the tracked subspace (k_cone_sub) with the SUBP dominant eigenvectors when few eigenvalues are positive
  if (w.sub_enable) {
    const bool seed = s_nkeep <= SUBP - w.sub_guard && N >= 3 * SUBP;
    if (w.sub_debug == 3 && tid == 0) atomicAdd(&w.stamps[(s_nkeep < 31) ? s_nkeep : 31], 1.0);      // diagnostics: positive eigenvalues seen by the full kernel
    if (seed) {
      // the SUBP largest eigenvalues by rank (ties: the smaller index first), one thread per eigenvalue -- the serial selection this
      // replaces (16 passes over N values on one thread) took a third of a warm call
      for (int t = tid; t < N; t += T) {
        const double lt = lamv_s[t];
        int rk = 0;
        for (int u = 0; u < N; ++u) { const double lu = lamv_s[u]; rk += (lu > lt || (lu == lt && u < t)) ? 1 : 0; }
        if (rk < SUBP) { s_top[rk] = t; w.sub_theta[(size_t)b * SUBP + rk] = lt; }
      }
      __syncthreads();
      double* Xg = w.Xs + (size_t)b * NP * SUBP;
      for (int e = tid; e < SUBP * NP; e += T) {
        const int j = e / NP, r = e - j * NP;
        Xg[e] = (r < N) ? Gm[(size_t)s_top[j] * ld + r] * rsqrt(ev[s_top[j]]) : 0.0;
      }
      if (tid == 0) atomicAdd(&w.sub_stat[8 * b + 3], 1);
    }
    if (tid == 0) w.sub_on[b] = seed ? 1 : 0;
  }
  for (int c2 = tid; c2 < s_nsel; c2 += T) wgt[c2] /= ev[sel[c2]];      // weight / nu^2
  __syncthreads();
  double* Wout = w.W1 + (size_t)b * n * n;
  auto entry2 = [&](int i, int j) { return Mb[(size_t)j * NP + i]; };
  auto store = [&](int i, int j, double v, double) { Wout[(size_t)j * n + i] = v; Wout[(size_t)i * n + j] = v; };
  STAMP(3);
  spectral_rebuild(Gm, ld, N, sel, wgt, s_nsel, s_base, entry2, store);
  STAMP(4);
  if (tid == 0) { DIAG_CYC(2, b); DIAG_ADD(3, b, 1); if (w.ws_phase == 1) w.cone_done[b] = 1; }
}

// ---------------------------------------------------------------------------------------------------------
// k_cone_sub: the cone block without an eigendecomposition.  clip(M, 0, 1) = sum over the eigenpairs with lambda > 0 of
// min(lambda, 1) v v', and once the ADMM iterate has settled M = Y - D1 has only a handful of positive eigenvalues (config 2:
// 2 of 100; the other 98 carry the scaled dual of Y >= 0).  The slot therefore follows the dominant invariant subspace with a
// block X of SUBP = 16 orthonormal vectors, warm-started from the previous ADMM iteration:
//     `chunk` steps of  X <- orth((M + s I) X)   (v_mfma_f64_16x16x4_f64 for M X and the Gram matrix, Cholesky-QR),
//     Rayleigh-Ritz on the 16 x 16 projection (in-wave Jacobi), residuals || M x - theta x || of every Ritz pair with theta >= 0;
//     accepted when they are below sub_tol ||M||_F and at least SUBG Ritz values are negative (so that no positive eigenvalue
//     hides outside the block); otherwise more steps, and after sub_qmax steps the slot falls back to the full kernel.
// The shift s centres the untracked spectrum [lo, theta_min]; lo comes from the first two moments of M (trace and Frobenius
// norm, written by k_global) by Samuelson's inequality, so it is a rigorous bound, not a guess.
// One 256-thread workgroup per slot, ~38 KB of LDS: four workgroups per CU.  The full kernel (k_cone_ws) seeds X from its
// eigenvectors whenever it finds at most SUBP - SUBG positive eigenvalues.
// ---------------------------------------------------------------------------------------------------------

// MODE 0: the cone block (above).  MODE 1: the certificate -- the k most negative eigenvalues of the Lagrangian matrix Mchk (k_check_build)
// are the k largest of -Mchk, followed by the same machinery on its own block XsC.  Ritz values are lower bounds of those, so the
// result is an ESTIMATE of the dual bound (optimistic by the square of the residual): k_check_final uses it for its decisions and every
// node that is about to finish gets the rigorous eigendecomposition (k_cone_ws, ws_mode = 1) before anything is reported (w.confirm).
template <int MODE>
__global__ void __launch_bounds__(256) k_cone_sub(OmcWS w) {
  extern __shared__ double smem[];
  __shared__ int s_flag, s_nsel, s_cond;
  __shared__ double s_shift, s_cs[SUBP];
  const int b = (MODE == 0) ? slot_of(w, blockIdx.x) : (int)blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  if (MODE == 2) {
    if (!w.fin[b] || !w.sub_on[b]) return;      // separation of a harvested slot, seeded by the block that followed Y - D1
  } else {
    if (w.done[b]) return;
    if (MODE == 0) {
      if (w.ws_first) {      // decided before this iteration (k_global): the full kernel runs this slot beside us and may re-seed sub_on / Xs meanwhile
        if (w.ws_first[b]) { if (threadIdx.x == 0 && w.sub_wait[b] > 0) w.sub_wait[b] -= 1; return; }
      } else {
        if (!w.sub_on[b]) return;
        if (w.sub_wait[b] > 0) { if (threadIdx.x == 0) w.sub_wait[b] -= 1; return; }     // backing off after a failed call
      }
    } else {
      if (!w.sub_onC[b]) { if (threadIdx.x == 0) w.confirm[b] = 1; return; }           // no block yet: the full kernel evaluates (and seeds)
    }
  }
  const int n = w.n, NP = w.np16, nt = NP >> 4, LD = NP + 2;
  const int wv = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  double* Xa = smem;                     // SUBP x LD   X (column j at Xa + j*LD)
  // orders beyond 512 (config 5: n = 1000): the two SUBP x LD blocks no longer fit the LDS together -- Z lives in a per-slot global slab (L2) there
  const bool zglob = NP > 512;
  double* Za = zglob ? w.sub_zscratch + (size_t)b * SUBP * LD : Xa + (size_t)SUBP * LD;   // SUBP x LD   Z = M X, then Y = Z + s X
  double* Cs = zglob ? Xa + (size_t)SUBP * LD : Za + (size_t)SUBP * LD;   // 4 partial 16 x 16 products
  double* Hs = Cs + 4 * 256;             // 16 x 17: Gram matrix / Cholesky factor / H / Ritz rotation
  double* Gj = Hs + 16 * 17;             // 16 x 17: Jacobi work
  double* th = Gj + 16 * 17;             // 16 Ritz values
  double* evj = th + 16;                 // 16 squared norms
  double* red = evj + 16;                // 32
  double* wgt = red + 32;                // 16
  int* sel = (int*)(wgt + 16);           // 16
  const double* Mb = ((MODE == 1) ? w.MbufC : w.Mbuf) + (size_t)b * NP * NP;      // MODE 2: k_sep_prepare has put Y - U U' there
  double* Xg = ((MODE == 1) ? w.XsC : w.Xs) + (size_t)b * NP * SUBP;
  double* thg = ((MODE == 1) ? w.sub_thetaC : w.sub_theta) + (size_t)b * SUBP;
  constexpr double sgn = (MODE == 1) ? -1.0 : 1.0;          // MODE 1 works on -Mchk
  const double fro2 = (MODE == 1) ? w.fro2c[b] : w.fro2[b], nF = sqrt(fro2), trM = (MODE == 1) ? -w.trMc[b] : w.trM[b];
  // inexact projections are harmless while the ADMM iterate itself still moves (errors proportional to the step are summable):
  // the residual target follows the last dual residual ||Y_new - Y_old|| down to sub_tol
  const double tol_eff = (MODE == 2) ? 1e-11 : (MODE == 1) ? 1e-9 : ((w.sub_adapt > 0.0) ? fmin(1e-6, fmax(w.sub_tol, w.sub_adapt * w.rd[b] / fmax(nF, 1e-300))) : w.sub_tol);
  for (int e = tid; e < SUBP * NP; e += T) { const int j = e / NP, r = e - j * NP; Xa[(size_t)j * LD + r] = Xg[e]; }
  if (tid < SUBP) th[tid] = thg[tid];
  __syncthreads();

  // Z (+ s X) = M X for the row tiles of this wave (up to three, computed together: independent accumulators and twelve loads in
  // flight per lane); operands: M from L2 (symmetric: column k of M is row k), X from LDS
  auto mul_MX = [&](double shift) {
    for (int tb = 0; tb < nt; tb += 12) {      // 12 row tiles per pass (NP <= 192: one pass)
      const int b0 = tb + wv;
      if (b0 >= nt) break;
      const int t0 = b0, t1 = (b0 + 4 < nt) ? b0 + 4 : b0, t2 = (b0 + 8 < nt) ? b0 + 8 : b0;     // invalid tiles alias tile t0 (results dropped)
      const bool v1 = b0 + 4 < nt, v2 = b0 + 8 < nt;
      double4v acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0, acc2 = acc0;
      const double* M0 = Mb + (t0 << 4) + li;
      const double* M1 = Mb + (t1 << 4) + li;
      const double* M2 = Mb + (t2 << 4) + li;
      const double* Xb = Xa + (size_t)li * LD;
      double a0[4], a1[4], a2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const size_t o = (size_t)(4 * u + lk) * NP; a0[u] = M0[o]; a1[u] = M1[o]; a2[u] = M2[o]; }
      for (int k0 = 0; k0 < NP; k0 += 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double x0 = sgn * a0[u], x1 = sgn * a1[u], x2 = sgn * a2[u], bv = Xb[k0 + 4 * u + lk];
          int kn = k0 + 16 + 4 * u + lk;
          kn = (kn < NP) ? kn : kn - NP;               // wrapped prefetch of the last chunk is never used
          const size_t o = (size_t)kn * NP;
          a0[u] = M0[o]; a1[u] = M1[o]; a2[u] = M2[o];
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, bv, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, bv, acc1, 0, 0, 0);
          if (nt > 8) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, bv, acc2, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t a = (size_t)li * LD + lk + 4 * r;
        Za[a + (t0 << 4)] = acc0[r] + shift * Xa[a + (t0 << 4)];
        if (v1) Za[a + (t1 << 4)] = acc1[r] + shift * Xa[a + (t1 << 4)];
        if (v2) Za[a + (t2 << 4)] = acc2[r] + shift * Xa[a + (t2 << 4)];
      }
    }
  };
  // Hs = P' Q (16 x 16) for two SUBP x LD blocks: the rows are split over the four waves, partial tiles summed in a fixed order
  auto gram = [&](const double* P, const double* Q) {
    double4v acc = {0.0, 0.0, 0.0, 0.0};
    const double* Pa = P + (size_t)li * LD;
    const double* Qa = Q + (size_t)li * LD;
    for (int u = wv; u < (NP >> 2); u += 4) {
      const int row = 4 * u + lk;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pa[row], Qa[row], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Cs[wv * 256 + (lk + 4 * r) * 16 + li] = acc[r];
    __syncthreads();
    Hs[(tid >> 4) * 17 + (tid & 15)] = (Cs[tid] + Cs[256 + tid]) + (Cs[512 + tid] + Cs[768 + tid]);
    __syncthreads();
  };

  int steps = 0, ok = 0, fail = 0, nrr = 0;
  const int qmax = (MODE == 2) ? 4 * w.sub_qmax : w.sub_qmax, chunk = w.sub_chunk;      // MODE 2: the alternative is a cold eigendecomposition
  const bool prof = (w.sub_debug == 2) && b == 0;
  long long tprev = prof ? __builtin_amdgcn_s_memtime() : 0;
#define SUBSTAMP(slot) do { if (prof) { __syncthreads(); if (tid == 0) { const long long t_ = __builtin_amdgcn_s_memtime(); w.stamps[slot] += (double)(t_ - tprev); tprev = t_; } } } while (0)
  SUBSTAMP(0);
  bool skip_steps = (MODE == 2);      // MODE 2: the block belongs to another matrix -- Rayleigh-Ritz first, its Ritz values then set the shift
  while (true) {
    // ---- `chunk` power steps with the shift that centres the untracked part of the spectrum --------------------------------
    if (wv == 0 && !skip_steps) {
      const double tj = (lane < SUBP) ? th[lane] : 0.0;
      const double st = wave_sum(tj), st2 = wave_sum(tj * tj);
      const double nr = (double)(n - SUBP);
      const double mu = (trM - st) / nr;
      const double var = fmax((fro2 - st2) / nr - mu * mu, 0.0);
      // The untracked spectrum lies in [lo, hi].  hi = the smallest Ritz value on the upper side (above the mean mu of the untracked
      // eigenvalues).  lo: Samuelson's inequality (every untracked eigenvalue is >= mu - sigma sqrt(nr - 1)) is rigorous but several
      // times too low when the bottom is a tight cluster (the scaled dual of Y >= 0), so mu - 3 sigma is used when it is the larger
      // one; a block vector that the iteration has pulled to the BOTTOM of the spectrum (Ritz value below mu: the shift was too small)
      // is itself the best estimate of lambda_min and takes precedence.  A wrong estimate costs steps, never accuracy (residual test).
      double hi = (lane < SUBP && tj > mu) ? tj : 1e300, lo_t = (lane < SUBP && !(tj > mu)) ? tj : 1e300;
      for (int o = 32; o > 0; o >>= 1) { hi = fmin(hi, __shfl_xor(hi, o, WAVE)); lo_t = fmin(lo_t, __shfl_xor(lo_t, o, WAVE)); }
      if (hi > 1e299) hi = mu;
      double lo = fmax(mu - sqrt(var * (nr - 1.0)), mu - 3.0 * sqrt(var));
      if (lo_t < 1e299) lo = fmin(lo, lo_t - 0.05 * (hi - lo_t));
      double tmax = (lane < SUBP) ? tj : -1e300;
      for (int o = 32; o > 0; o >>= 1) tmax = fmax(tmax, __shfl_xor(tmax, o, WAVE));
      // column j is (close to) the Ritz vector of theta_j and grows by theta_j + s per unorthogonalised step: dividing it out keeps the
      // block near orthonormal between the orthonormalisations of a chunk
      const double sh = -0.5 * (lo + hi);
      if (lane < SUBP) s_cs[lane] = 1.0 / fmax(fabs(tj + sh), 1e-3 * fmax(fabs(tmax + sh), 1e-300));
      if (lane == 0) s_shift = sh;
    }
    __syncthreads();
    const double shift = s_shift;
    SUBSTAMP(1);
    for (int c = 0; c < chunk && !fail && !skip_steps; ++c) {
      mul_MX(shift);
      __syncthreads();
      SUBSTAMP(2);
      if (MODE != 2 && w.sub_lazy && c + 1 < chunk) {
        // inside a chunk the block is only rescaled: the span after `chunk` steps is the same, and the condition of the block grows by
        // at most (largest / smallest shifted Ritz value)^chunk, which the orthonormalisation at the end of the chunk absorbs (it is
        // repeated once when its pivots say the block had become ill-conditioned)
        for (int e = tid; e < SUBP * NP; e += T) { const int j = e / NP, r = e - j * NP; Xa[(size_t)j * LD + r] = s_cs[j] * Za[(size_t)j * LD + r]; }
        __syncthreads();
        SUBSTAMP(5);
        ++steps;
        continue;
      }
      for (int pass = 0; pass < 2; ++pass) {
        gram(Za, Za);
        SUBSTAMP(3);
        if (wv == 0) {
          // Cholesky of the Gram matrix in REGISTERS: lane i < 16 holds row i, pivots and multipliers travel by v_readlane (no LDS
          // round trips, no barrier); then column j of L^-1 by forward substitution in lane j, the rows of L again by v_readlane (an
          // LDS copy of L made the compiler hoist all 136 broadcast loads: 430 VGPRs, one workgroup per CU).  ~1.5 k instructions, one wave.
          double c[SUBP], dinv[SUBP];
          const int i = (lane < SUBP) ? lane : SUBP - 1;
#pragma unroll
          for (int q = 0; q < SUBP; ++q) c[q] = Hs[i * 17 + q];
          int bad = 0;
          double dmin = 1e300, dmax = 0.0;
#pragma unroll
          for (int j = 0; j < SUBP; ++j) {
            const double d = readlane_d(c[j], j);
            if (!(d > 1e-280)) bad = 1;
            dmin = fmin(dmin, d); dmax = fmax(dmax, d);
            const double inv = rsqrt(fmax(d, 1e-280));
            dinv[j] = inv;                                // 1 / L[j][j]
            const double lj = c[j] * inv;                 // L[i][j] for i >= j (lane j: sqrt(d))
#pragma unroll
            for (int q = j + 1; q < SUBP; ++q) c[q] = fma(-lj, readlane_d(lj, q), c[q]);
            c[j] = lj;
          }
          // L^-1: lane j solves L z = e_j; L[r][q] lives in lane r and arrives by v_readlane; the result goes to Hs as Linv[i][j]
          double z[SUBP];
#pragma unroll
          for (int r = 0; r < SUBP; ++r) {
            double v = (r == i) ? 1.0 : 0.0;
#pragma unroll
            for (int q = 0; q < SUBP; ++q) if (q < r) v = fma(-readlane_d(c[q], r), z[q], v);
            z[r] = v * dinv[r];
          }
          if (lane < SUBP) {
#pragma unroll
            for (int r = 0; r < SUBP; ++r) Hs[r * 17 + lane] = (r >= lane) ? z[r] : 0.0;       // Linv[r][j = lane]
          }
          if (lane == 0) { s_flag = bad; s_cond = (pass == 0 && w.sub_lazy && dmin < 1e-6 * dmax) ? 1 : 0; }   // pivot ratio ~ cond(block)^2
        }
        __syncthreads();
        SUBSTAMP(4);
        if (s_flag) { fail = 3; break; }
        const bool again = s_cond != 0;
        {   // X = Y L^-T by MFMA: X[row][j] = sum_q Y[row][q] Linv[j][q]  (a row tile of X depends on the same rows of Y only)
          double* Xo = again ? Za : Xa;            // second pass: the result goes back into Za (a wave reads its tile completely before it writes)
          for (int ti = wv; ti < nt; ti += 4) {
            const int i0 = ti << 4;
            double4v a1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 4; ++u)
              a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Za[(size_t)(4 * u + lk) * LD + i0 + li], Hs[li * 17 + 4 * u + lk], a1, 0, 0, 0);
            WAVE_SYNC();
#pragma unroll
            for (int r = 0; r < 4; ++r) Xo[(size_t)li * LD + i0 + lk + 4 * r] = a1[r];
          }
        }
        __syncthreads();
        SUBSTAMP(5);
        if (!again) break;
      }
      if (fail) break;
      ++steps;
    }
    if (fail) break;
    skip_steps = false;
    // ---- Rayleigh-Ritz ------------------------------------------------------------------------------------------------------
    mul_MX(0.0);
    __syncthreads();
    gram(Xa, Za);
    SUBSTAMP(6);
    {   // symmetrised H + sigma I -> Gj (sigma = 1.5 ||H||_F makes it positive definite), one-sided Jacobi inside wave 0
      const int i = tid >> 4, j = tid & 15;
      const double hv = 0.5 * (Hs[i * 17 + j] + Hs[j * 17 + i]);
      const double f2 = block_sum(hv * hv, red);
      if (tid == 0) s_shift = 1.5 * sqrt(f2) + 1e-300;
      __syncthreads();
      Gj[j * 17 + i] = hv + ((i == j) ? s_shift : 0.0);
      __syncthreads();
      if (wv == 0) {
        jacobi16_fast(Gj, evj, 17, 1e-13, 12);
        if (lane < 16) { double a = 0.0;
#pragma unroll
          for (int r = 0; r < 16; ++r) { const double x = Gj[lane * 17 + r]; a += x * x; }
          evj[lane] = a; }
      }
      __syncthreads();
    }
    const double sigma = s_shift;
    if (tid < SUBP) th[tid] = sqrt(evj[tid]) - sigma;
    __syncthreads();
    SUBSTAMP(7);
    Hs[(tid >> 4) * 17 + (tid & 15)] = Gj[(tid & 15) * 17 + (tid >> 4)] * rsqrt(evj[tid & 15]);    // Wr[k][t] = g_t[k] / nu_t
    __syncthreads();
    {   // X <- X Wr, Z <- Z Wr, in place: a row tile depends on its own rows only, and a wave reads its tile completely before it writes
      for (int ti = wv; ti < nt; ti += 4) {
        const int i0 = ti << 4;
        double4v a1 = {0.0, 0.0, 0.0, 0.0}, a2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double bw = Hs[(4 * u + lk) * 17 + li];
          a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Xa[(size_t)(4 * u + lk) * LD + i0 + li], bw, a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Za[(size_t)(4 * u + lk) * LD + i0 + li], bw, a2, 0, 0, 0);
        }
        WAVE_SYNC();
#pragma unroll
        for (int r = 0; r < 4; ++r) { const size_t a = (size_t)li * LD + i0 + lk + 4 * r; Xa[a] = a1[r]; Za[a] = a2[r]; }
      }
      __syncthreads();
    }
    SUBSTAMP(8);
    // residuals of the Ritz pairs: wave wv handles columns 4 wv .. 4 wv + 3
    for (int c = 0; c < 4; ++c) {
      const int t = 4 * wv + c;
      const double tv = th[t];
      double a = 0.0;
      for (int r = lane; r < NP; r += WAVE) { const double d = Za[(size_t)t * LD + r] - tv * Xa[(size_t)t * LD + r]; a += d * d; }
      a = wave_sum(a);
      if (lane == 0) evj[t] = sqrt(a);
    }
    __syncthreads();
    if (MODE >= 1) {
      if (tid == 0) {      // only the largest Ritz values have to be accurate: MODE 1 the k most negative eigenvalues of Mchk, MODE 2 two
        int bad = 0; double used[8]; const int kk = (MODE == 2) ? 2 : ((w.k < 8) ? w.k : 8);
        const int kreq = (MODE == 2 && w.breakpoints != 2) ? 1 : kk;      // smallest_1_eigvec uses the first pair only; the second eigenvalue is then informative
        double ssum = 0.0;
        for (int q = 0; q < kk; ++q) {
          int bi = -1; double bl = -1e300;
          for (int t = 0; t < SUBP; ++t) {
            bool u_ = false;
            for (int q2 = 0; q2 < q; ++q2) if (used[q2] == (double)t) u_ = true;
            if (!u_ && th[t] > bl) { bl = th[t]; bi = t; }
          }
          used[q] = (double)bi;
          if (q < kreq && !(evj[bi] <= tol_eff * fmax(nF, 1e-300))) bad = 1;
          ssum += fmax(bl, 0.0);
          if (MODE == 2) sel[q] = bi;
        }
        s_flag = bad ? 0 : 1;
        s_shift = -ssum;         // estimate of sum_{i <= k} min(eig_i(Mchk), 0)
      }
    } else if (tid == 0) {
      // a Ritz pair (theta, x) contaminated by an untracked eigenvector (eigenvalue <= gneg < 0) by an angle phi has residual
      // ~ (theta - gneg) phi and perturbs W1 by ~ theta phi: pairs with a small theta may carry a proportionally larger residual
      int nk = 0, bad = 0; double gneg = -1e300;
      for (int t = 0; t < SUBP; ++t) { if (th[t] > 0.0) ++nk; else gneg = fmax(gneg, th[t]); }
      if (gneg < -1e299) gneg = 0.0;
      for (int t = 0; t < SUBP; ++t) {
        if (th[t] > -1e-9 * nF) {      // (nearly) positive Ritz values must be accurate
          const double scale = (th[t] > 0.0) ? 1.0 + fmin(-gneg / th[t], 1e6) : 1.0;
          if (!(evj[t] <= tol_eff * nF * scale)) bad = 1;
        }
      }
      s_flag = (nk > SUBP - w.sub_guard) ? 2 : (bad ? 0 : 1);
    }
    __syncthreads();
    SUBSTAMP(9);
    ++nrr;
    if (s_flag == 2) { fail = 1; break; }
    if (s_flag == 1) { ok = 1; break; }
    if (steps >= qmax) {
      if (w.sub_debug == 1 && tid < SUBP) { w.stamps[tid] = th[tid]; w.stamps[16 + tid] = evj[tid] / nF; }     // diagnostics: the Ritz values / relative residuals that did not make it
      fail = 2; break;
    }
  }
  if (MODE == 2) {
    if (!ok) return;                      // sep_done stays 0: the cold kernel computes the separation vector
    // lambda_min(U U' - Y) = -theta_1; canonical sign (largest-magnitude entry positive) and the smallest_2 mix of OMC.jl:2471-2476
    const int i1 = sel[0], i2 = sel[1];
    __shared__ double s_sg[2], s_wt[2];
    // A small residual shows that (theta_1, x_1) is AN eigenpair of Y - U U', not that it is the extreme one: the block was seeded from
    // another matrix (Y - D1).  Rigorous guard: every eigenvalue outside the block is at most rem = sqrt(||M||_F^2 - sum theta^2) in
    // magnitude, so the pair is the largest one only if rem < theta_1 (and theta_2 likewise for smallest_2_eigvec).  A result that would
    // declare the node master-feasible (lambda_min >= -1e-6, OMC.jl:1274-1276: the node is fathomed on it) is never taken from the
    // block either.  In both cases sep_done stays 0 and the cold eigendecomposition (CONE_SEP) decides.
    __shared__ int s_acc;
    if (tid == 0) {
      double s2 = 0.0;
      for (int t = 0; t < SUBP; ++t) s2 += th[t] * th[t];
      const double rem = sqrt(fmax(fro2 - s2, 0.0)) + 1e-12 * nF;
      bool acc = (-th[i1] < -1e-6) && rem < th[i1];
      if (w.breakpoints == 2) acc = acc && rem < fmax(th[i2], 1e-10);
      s_acc = acc ? 1 : 0;
    }
    __syncthreads();
    if (!s_acc) return;
    if (tid == 0) {
      const double l1 = -th[i1], l2 = -th[i2];
      w.lmin[2 * b] = l1; w.lmin[2 * b + 1] = l2;
      for (int q = 0; q < 2; ++q) {
        const double* g = Xa + (size_t)sel[q] * LD;
        int a1 = 0;
        for (int r = 1; r < n; ++r) if (fabs(g[r]) > fabs(g[a1])) a1 = r;
        s_sg[q] = (g[a1] >= 0.0) ? 1.0 : -1.0;
      }
      if (w.breakpoints == 2 && l2 < -1e-10) { const double nn = sqrt(l1 * l1 + l2 * l2); s_wt[0] = fabs(l1) / nn; s_wt[1] = fabs(l2) / nn; }
      else { s_wt[0] = 1.0; s_wt[1] = 0.0; }
    }
    __syncthreads();
    for (int r = tid; r < n; r += T) w.bx[(size_t)b * n + r] = s_wt[0] * s_sg[0] * Xa[(size_t)i1 * LD + r] + s_wt[1] * s_sg[1] * Xa[(size_t)i2 * LD + r];
    if (tid == 0) w.sep_done[b] = 1;
    return;
  }
  if (MODE == 1) {
    if (!ok) { if (tid == 0) { w.confirm[b] = 1; w.sub_onC[b] = 0; } return; }      // the full kernel evaluates this slot and seeds a new block
    const double est = s_shift;
    for (int e = tid; e < SUBP * NP; e += T) { const int j = e / NP, r = e - j * NP; Xg[e] = Xa[(size_t)j * LD + r]; }
    if (tid < SUBP) thg[tid] = th[tid];
    if (tid == 0) { w.evsum[b] = est; w.confirm[b] = 0; }
    return;
  }
  if (tid == 0) {
    atomicAdd(&w.sub_stat[8 * b + 0], 1); atomicAdd(&w.sub_stat[8 * b + 1], steps); atomicAdd(&w.sub_stat[8 * b + 7], nrr);
    if (!ok) {
      atomicAdd(&w.sub_stat[8 * b + 2], 1); atomicAdd(&w.sub_stat[8 * b + 3 + fail], 1);   // 4: too many positive Ritz values, 5: step cap, 6: Cholesky
      const int nf = w.sub_nfail[b] + 1;          // exponential back-off: the spectrum still moves too fast for the tracked block
      w.sub_nfail[b] = nf; w.sub_wait[b] = (nf >= 7) ? 128 : (1 << nf);
    } else if (w.sub_nfail[b] > 0 && (w.iters[b] & 15) == 0) {
      w.sub_nfail[b] -= 1;                        // ... and forgets: a failure every few hundred calls (long solves: Shor mode runs thousands of iterations) must not cost 128 full decompositions each
    }
  }
  if (!ok) return;                       // cone_done stays 0: the full kernel projects this slot (and re-seeds X)
  for (int e = tid; e < SUBP * NP; e += T) { const int j = e / NP, r = e - j * NP; Xg[e] = Xa[(size_t)j * LD + r]; }
  if (tid < SUBP) thg[tid] = th[tid];
  if (tid == 0) {
    int c = 0;
    for (int t = 0; t < SUBP; ++t) if (th[t] > 0.0) { sel[c] = t; wgt[c] = fmin(th[t], w.clip_hi); ++c; }
    s_nsel = c;
  }
  __syncthreads();
  double* Wout = w.W1 + (size_t)b * n * n;
  auto entry0 = [&](int, int) { return 0.0; };
  auto store = [&](int i, int j, double v, double) { Wout[(size_t)j * n + i] = v; Wout[(size_t)i * n + j] = v; };
  SUBSTAMP(10);
  spectral_rebuild(Xa, LD, n, sel, wgt, s_nsel, 0.0, entry0, store);
  SUBSTAMP(11);
  if (prof && tid == 0) { w.stamps[12] += 1.0; w.stamps[13] += steps; w.stamps[14] += nrr; }
  if (tid == 0) w.cone_done[b] = 1;
}

// Mbuf <- Y - U U' (zero padded), fro2, trace for the harvested slots whose separation vector k_cone_sub<2> will compute
__global__ void __launch_bounds__(256) k_sep_prepare(OmcWS w) {
  __shared__ double red[32];
  const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  if (tid == 0 && w.fin[b]) w.sep_done[b] = 0;
  if (!w.fin[b] || !w.sub_on[b]) return;
  const int n = w.n, k = w.k, NP = w.np16;
  const double* Y = w.Y + (size_t)b * n * n;
  const double* U = w.U + (size_t)b * n * k;
  double* Mb = w.Mbuf + (size_t)b * NP * NP;
  double fr2 = 0.0, tr1 = 0.0;
  for (int e = tid; e < n * n; e += T) {
    const int i = e % n, j = e / n;
    double s2 = 0.0;
    for (int t = 0; t < k; ++t) s2 += U[(size_t)t * n + i] * U[(size_t)t * n + j];
    const double mv = 0.5 * (Y[e] + Y[(size_t)i * n + j]) - s2;
    Mb[(size_t)j * NP + i] = mv; fr2 += mv * mv; if (i == j) tr1 += mv;
  }
  fr2 = block_sum(fr2, red); tr1 = block_sum(tr1, red);
  if (tid == 0) { w.fro2[b] = fr2; w.trM[b] = tr1; }
}

// ---------------------------------------------------------------------------------------------------------
// C = L L' for a dense column-major n x m matrix L (zero off the support of the observed pattern) by v_mfma_f64_16x16x4_f64,
// one 16 x 16 tile of the lower triangle per wave and pass, operands straight from L2.  store(i, j, v) receives every entry of
// the tile with i >= j inside the matrix.  k_global (every order) and k_check_build (n > 144) use it: the per-entry walk over the CSR
// lists that it replaced did a fifth of the flops and cost 41 us of 108 per node at n = 100, 4.8 ms at n = 200.
// ---------------------------------------------------------------------------------------------------------
template <class StoreF>
__device__ __forceinline__ void mfma_LLt(const double* L, int n, int m, StoreF store) {
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6, li = lane & 15, lk = lane >> 4;
  const int nt = (n + 15) >> 4, ntile = nt * (nt + 1) / 2;
  const int m4 = m & ~3;
  for (int tile = wv; tile < ntile; tile += nw) {
    int ti = (int)((sqrtf(8.0f * (float)tile + 1.0f) - 1.0f) * 0.5f);
    while (ti * (ti + 1) / 2 > tile) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
    const int tj = tile - ti * (ti + 1) / 2;
    const int ia = (ti << 4) + li, jb = (tj << 4) + li;
    const double ma = (ia < n) ? 1.0 : 0.0, mb = (jb < n) ? 1.0 : 0.0;
    const double* La = L + ((ia < n) ? ia : n - 1);
    const double* Lb = L + ((jb < n) ? jb : n - 1);
    double4v acc = {0.0, 0.0, 0.0, 0.0};
    double aq[4], bq[4];
    if (m4 >= 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { aq[u] = La[(size_t)(4 * u + lk) * n]; bq[u] = Lb[(size_t)(4 * u + lk) * n]; }
    }
    int k0 = 0;
    for (; k0 + 16 <= m4; k0 += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double a = aq[u] * ma, bv = bq[u] * mb;
        int kn = k0 + 16 + 4 * u + lk;
        kn = (kn < m4) ? kn : kn - 16;              // re-reads a valid column at the end; the value is not used
        aq[u] = La[(size_t)kn * n]; bq[u] = Lb[(size_t)kn * n];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc, 0, 0, 0);
      }
    }
    for (; k0 < m; k0 += 4) {                       // remainder (< 16 columns), guarded
      const int kk = k0 + lk;
      const double a = (kk < m) ? La[(size_t)kk * n] * ma : 0.0, bv = (kk < m) ? Lb[(size_t)kk * n] * mb : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = (ti << 4) + lk + 4 * r, j = (tj << 4) + li;
      if (i < n && j < n && i >= j) store(i, j, acc[r]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// k_small: the small cone  [S Vt; Vt' T] >= 0,  S = Q'(Y - D3)Q (r x r),  Vt - D3V (r x k),  T = I - D3T.
//   mode SMALL_PROJ   : P3 = P_+(M3);  Q3 = P3 - M3 (>= 0: the multiplier direction);
//                       dS = Q3_11, E3 = Q dS Q' (n x n), W3V = P3_12, W3T = P3_22, Q3V = Q3_12, Q3T = Q3_22
//   mode SMALL_RECOVER: U = Y Q (Q'YQ)^+ Vt  (a U with U U' <= Y and Q'U = Vt)
// One workgroup per node; T1 = (Y - D3) Q is staged in dynamic LDS or global scratch.
// ---------------------------------------------------------------------------------------------------------
template <bool USE_LDS>
__global__ void __launch_bounds__(256) k_small(OmcWS w, int mode) {
  extern __shared__ double smem[];
  __shared__ double red[32];
  __shared__ int s_cnt, s_nsel;
  __shared__ double s_base;
  const int b = slot_of(w, blockIdx.x), tid = threadIdx.x, T = blockDim.x;
  if (w.done[b] && mode == SMALL_PROJ) return;
  if (mode == SMALL_RECOVER && !w.fin[b]) return;   // recovery runs once, when the slot's node is harvested
  const int nb = w.node_of[b];
  const int n = w.n, k = w.k, rm = w.rmax;
  const int r = w.rr[nb];
  const int N3 = (mode == SMALL_PROJ) ? r + k : r;
  const int Np = (N3 + 1) & ~1, ld = Np | 1;
  auto base = [&]() { if constexpr (USE_LDS) return (double*)smem; else return w.small_scratch + (size_t)b * w.small_scratch_stride; }();
  double* T1 = base;                         // n x r
  double* M3 = T1 + (size_t)n * rm;          // N3max x N3max  (column-major, ld3 = rm + k)
  const int ld3 = rm + k;
  double* Gm = M3 + (size_t)ld3 * ld3;       // Jacobi work
  const int Npm = (rm + k + 1) & ~1, ldm = Npm | 1;
  double* ev = Gm + (size_t)Npm * ldm;
  double* wgt = ev + Npm;
  int* sel = (int*)(wgt + Npm);
  double* Cc = (double*)(sel + Npm + (Npm & 1));  // r x k coefficients (recover)
  double* Qs = Cc + (size_t)rm * k + 2;          // n x 16: Q' padded with zero columns (LDS variant only, r <= 16)
  const double* Q = w.Qb + (size_t)nb * n * rm;
  const double* Y = w.Y + (size_t)b * n * n;
  const double* D3 = w.D3 + (size_t)b * n * n;
  const double* Vt = w.Vt + (size_t)b * rm * k;
  STAMP_BEGIN();
  DIAG_T0();
  // T1 = (Y - D3) Q   or  Y Q
  constexpr int T1_RS = 16;
  if (USE_LDS && r <= T1_RS && n <= T) {
    // Q' staged in LDS, padded to 16 columns (zero beyond column r)
    for (int e = tid; e < n * T1_RS; e += T) { const int j = e / T1_RS, a = e % T1_RS; Qs[e] = (a < r) ? Q[(size_t)a * n + j] : 0.0; }
    __syncthreads();
    {
      // (Y - D3) Q on the matrix cores: wave w owns the 16-row tiles w, w + 4, ..; A = 16 x 4 blocks of Y - D3 straight from L2 (a column
      // of the column-major matrix is contiguous over the 16 rows: 128-byte loads, every entry read once), B = the staged Q' from LDS
      const int wv_ = tid >> 6, lane_ = tid & 63, li = lane_ & 15, lk = lane_ >> 4, nt = (n + 15) >> 4, nw_ = T >> 6;
      const bool proj = (mode == SMALL_PROJ);
      for (int ti = wv_; ti < nt; ti += nw_) {
        const int ia = (ti << 4) + li; const bool va = ia < n; const int iac = va ? ia : 0;
        double4v acc = {0.0, 0.0, 0.0, 0.0};
        double ya[4], da[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int kk = 4 * u + lk, kc = (kk < n) ? kk : 0; ya[u] = Y[(size_t)kc * n + iac]; da[u] = proj ? D3[(size_t)kc * n + iac] : 0.0; }
        for (int k0 = 0; k0 < n; k0 += 16) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int kk = k0 + 4 * u + lk; const bool vk = kk < n;
            const double av = ya[u] - da[u], bv = Qs[(vk ? kk : 0) * T1_RS + li];
            const int kn = kk + 16, knc = (kn < n) ? kn : 0;          // prefetch of the next chunk (the value past the end is not used)
            ya[u] = Y[(size_t)knc * n + iac]; da[u] = proj ? D3[(size_t)knc * n + iac] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64((va && vk) ? av : 0.0, vk ? bv : 0.0, acc, 0, 0, 0);
          }
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int row = (ti << 4) + lk + 4 * r4;
          if (row < n && li < r) T1[(size_t)li * n + row] = acc[r4];
        }
      }
      __syncthreads();
    }
  } else {
    for (int e = tid; e < n * r; e += T) {
      int i = e % n, a = e / n;
      double acc = 0.0;
      if (mode == SMALL_PROJ) for (int j = 0; j < n; ++j) acc += (Y[(size_t)j * n + i] - D3[(size_t)j * n + i]) * Q[(size_t)a * n + j];
      else for (int j = 0; j < n; ++j) acc += Y[(size_t)j * n + i] * Q[(size_t)a * n + j];
      T1[(size_t)a * n + i] = acc;
    }
    __syncthreads();
  }
  STAMP(16);
  // M3
  for (int e = tid; e < N3 * N3; e += T) {
    int i = e % N3, j = e / N3;
    double v;
    if (i < r && j < r) { v = 0.0; for (int t = 0; t < n; ++t) v += Q[(size_t)i * n + t] * T1[(size_t)j * n + t]; }
    else if (i < r) v = Vt[(size_t)(j - r) * rm + i] - w.D3V[(size_t)b * rm * k + (size_t)(j - r) * rm + i];
    else if (j < r) v = Vt[(size_t)(i - r) * rm + j] - w.D3V[(size_t)b * rm * k + (size_t)(i - r) * rm + j];
    else v = ((i == j) ? 1.0 : 0.0) - w.D3T[(size_t)b * k * k + (size_t)(j - r) * k + (i - r)];
    M3[(size_t)j * ld3 + i] = v;
  }
  __syncthreads();
  STAMP(17);
  if (N3 == 0) return;
  auto entry = [&](int i, int j) { return M3[(size_t)j * ld3 + i]; };
  double sigma;
  if (mode == SMALL_PROJ && Np <= 16 && w.V3) {
    // order <= 16: one-sided Jacobi inside one wave on G = (M3 + sigma I) V_prev -- M3 moves little from one ADMM iteration to the next,
    // so the columns start almost orthogonal (two sweeps instead of the six of a cold start); the normalised columns are the eigenvectors
    // and are kept for the next iteration
    double* Vp = w.V3 + (size_t)b * 256;
    const bool warm = w.v3valid[b] != 0;
    double fro = 0.0;
    for (int e = tid; e < N3 * N3; e += T) { const int i = e % N3, j = e / N3; const double v = 0.5 * (entry(i, j) + entry(j, i)); fro += v * v; }
    fro = sqrt(block_sum(fro, red));
    sigma = 1.5 * fro + 1e-300;
    for (int e = tid; e < Np * Np; e += T) {
      const int i = e % Np, t = e / Np;
      double v = 0.0;
      if (i < N3 && t < N3) {
        if (warm) { for (int j = 0; j < N3; ++j) v += (0.5 * (entry(i, j) + entry(j, i)) + ((i == j) ? sigma : 0.0)) * Vp[t * 16 + j]; }
        else v = 0.5 * (entry(i, t) + entry(t, i)) + ((i == t) ? sigma : 0.0);
      }
      Gm[(size_t)t * ld + i] = v;
    }
    __syncthreads();
    if (tid < WAVE) jacobi_sweeps_wave16(Gm, ev, N3, Np, ld, 1e-14, 30);
    __syncthreads();
    for (int t = tid; t < N3; t += T) { double a = 0.0; for (int i = 0; i < N3; ++i) { const double x = Gm[(size_t)t * ld + i]; a += x * x; } ev[t] = a; }
    __syncthreads();
    for (int e = tid; e < N3 * N3; e += T) { const int i = e % N3, t = e / N3; Vp[t * 16 + i] = Gm[(size_t)t * ld + i] * rsqrt(ev[t]); }
    if (tid == 0) w.v3valid[b] = 1;
  } else {
    sigma = eig_frontend(Gm, ev, N3, Np, ld, entry, red, &s_cnt, nullptr);
  }
  if (mode == SMALL_RECOVER) {
    // pinv weights: S^+ = sum_{lam > tol} (1/lam) v v' ;  v = g/nu  ->  weight 1/(lam nu^2)
    if (tid == 0) {
      double lmax = 0.0;
      for (int t = 0; t < N3; ++t) lmax = fmax(lmax, sqrt(ev[t]) - sigma);
      const double tol = 1e-12 * fmax(1.0, lmax);
      int c = 0;
      for (int t = 0; t < N3; ++t) {
        double nu2 = ev[t], lamv = sqrt(nu2) - sigma;
        if (lamv > tol) { sel[c] = t; wgt[c] = 1.0 / (lamv * nu2); ++c; }
      }
      s_nsel = c;
    }
    __syncthreads();
    // Cc = S^+ Vt  (r x k)
    for (int e = tid; e < r * k; e += T) {
      int a = e % r, j = e / r;
      double acc = 0.0;
      for (int s = 0; s < s_nsel; ++s) {
        const double* gt = Gm + (size_t)sel[s] * ld;
        double dot = 0.0;
        for (int c2 = 0; c2 < r; ++c2) dot += gt[c2] * Vt[(size_t)j * rm + c2];
        acc += wgt[s] * gt[a] * dot;
      }
      Cc[(size_t)j * rm + a] = acc;
    }
    __syncthreads();
    for (int e = tid; e < n * k; e += T) {
      int i = e % n, j = e / n;
      double acc = 0.0;
      for (int a = 0; a < r; ++a) acc += T1[(size_t)a * n + i] * Cc[(size_t)j * rm + a];
      w.U[(size_t)b * n * k + e] = acc;
    }
    return;
  }
  STAMP(18);
  // SMALL_PROJ: Q3 = P3 - M3 = sum_{lam<0} |lam| v v'   (or P3 directly when fewer positive eigenvalues)
  if (tid == 0) {
    int npos = 0, nneg = 0;
    for (int t = 0; t < N3; ++t) { double lamv = sqrt(ev[t]) - sigma; if (lamv > 0.0) ++npos; else if (lamv < 0.0) ++nneg; }
    const bool use_neg = nneg <= npos;
    int c = 0;
    for (int t = 0; t < N3; ++t) {
      double nu2 = ev[t], lamv = sqrt(nu2) - sigma;
      if (use_neg ? (lamv < 0.0) : (lamv > 0.0)) { sel[c] = t; wgt[c] = fabs(lamv) / nu2; ++c; }
    }
    s_nsel = c; s_base = use_neg ? 1.0 : 0.0;   // base=1: acc is Q3; base=0: acc is P3
  }
  __syncthreads();
  const bool acc_is_Q3 = (s_base == 1.0);
  double* dS = w.dS + (size_t)b * rm * rm;
  auto store = [&](int i, int j, double v, double Mv) {
    // v = base*Mv + acc ; recover (P3, Q3) from acc
    double accv = v - s_base * Mv;
    double q3 = acc_is_Q3 ? accv : (accv - Mv);
    double p3 = acc_is_Q3 ? (Mv + accv) : accv;
    if (i < r && j < r) { dS[(size_t)j * rm + i] = q3; dS[(size_t)i * rm + j] = q3; }
    else if (i >= r && j < r) {  // (row i in T block, col j in S block): V' entry -> store as V[j][i-r]
      w.W3V[(size_t)b * rm * k + (size_t)(i - r) * rm + j] = p3;
      w.Q3V[(size_t)b * rm * k + (size_t)(i - r) * rm + j] = q3;
    } else if (i >= r && j >= r) {
      w.W3T[(size_t)b * k * k + (size_t)(j - r) * k + (i - r)] = p3; w.W3T[(size_t)b * k * k + (size_t)(i - r) * k + (j - r)] = p3;
      w.Q3T[(size_t)b * k * k + (size_t)(j - r) * k + (i - r)] = q3; w.Q3T[(size_t)b * k * k + (size_t)(i - r) * k + (j - r)] = q3;
    }
  };
  spectral_rebuild(Gm, ld, N3, sel, wgt, s_nsel, s_base, entry, store);
  __syncthreads();
  __threadfence_block();
  STAMP(19);
  // E3 = Q dS Q'  (n x n): T1 <- Q dS (n x r), then E3 = T1 Q'
  for (int e = tid; e < n * r; e += T) {
    int i = e % n, a = e / n;
    double acc = 0.0;
    for (int c2 = 0; c2 < r; ++c2) acc += Q[(size_t)c2 * n + i] * dS[(size_t)a * rm + c2];
    T1[(size_t)a * n + i] = acc;
  }
  __syncthreads();
  double* E3 = w.E3 + (size_t)b * n * n;
  if (USE_LDS && r <= 16 && n <= T) {
    // rank-r product T1 Q' on the matrix cores: 16 x 16 tiles, K = 16 (the staged Q' is zero beyond column r), operands from LDS; the
    // tile is written through its transposed position (E3 is symmetric), which makes the 16 lanes of a row store 128 contiguous bytes
    const int wv_ = tid >> 6, lane_ = tid & 63, li = lane_ & 15, lk = lane_ >> 4, nt = (n + 15) >> 4, nw_ = T >> 6;
    for (int tile = wv_; tile < nt * nt; tile += nw_) {
      const int ti = tile / nt, tj = tile - ti * nt;
      const int ia = (ti << 4) + li, jb = (tj << 4) + li;
      const bool va = ia < n, vb = jb < n;
      const int iac = va ? ia : 0, jbc = vb ? jb : 0;
      double4v acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k0 = 0; k0 < 16; k0 += 4) {
        const int a = k0 + lk; const bool vk = a < r;
        const double av = T1[(size_t)(vk ? a : 0) * n + iac], bv = Qs[jbc * 16 + a];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64((va && vk) ? av : 0.0, vb ? bv : 0.0, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int row = (ti << 4) + lk + 4 * r4, col = (tj << 4) + li;
        if (row < n && col < n) E3[(size_t)row * n + col] = acc[r4];
      }
    }
  } else {
    for (int e = tid; e < n * n; e += T) {
      int i = e % n, j = e / n;
      if (i < j) continue;
      double acc = 0.0;
      for (int a = 0; a < r; ++a) acc += T1[(size_t)a * n + i] * Q[(size_t)a * n + j];
      E3[(size_t)j * n + i] = acc; E3[(size_t)i * n + j] = acc;
    }
  }
  STAMP(20);
  if (tid == 0) DIAG_CYC(5, b);
}

// ---------------------------------------------------------------------------------------------------------
// k_global: consensus average + projection on the rows + dual updates (one workgroup per node)
//   tY = [rho_f N.Y + gamma/2 LL + rho (rx W1 + (1-rx) Y + D1) + rho (Y + (1-rx) D3 + rx E3)] / (rho wY1)
//   tV = rx W3V + (1-rx) Vt + D3V ;  tU = Q tV
//   mu = argmin 1/2 mu'G1 mu - c'mu, mu >= 0 (c = A t - b) ;  lam = rho mu
//   Yn = tY - A_Y' mu / wY1 ;  Vn = tV - Q'(A_U' mu)/2
// ---------------------------------------------------------------------------------------------------------
#define GL_XS 16   // cut vectors staged per pass of k_global (one MFMA column block)
template <bool USE_LDS>
__global__ void __launch_bounds__(512) k_global(OmcWS w) {
  extern __shared__ double smem[];
  __shared__ double red[32];
  __shared__ double s_Gp[NNQP_PMAX * (NNQP_PMAX + 1) / 2];
  __shared__ double s_sv[NNQP_PMAX], s_tmp[NNQP_PMAX];
  __shared__ int s_pl[NNQP_PMAX];
  const int b = slot_of(w, blockIdx.x), tid = threadIdx.x, T = blockDim.x;
  if (w.done[b]) return;
  const int nb = w.node_of[b];
  const int n = w.n, k = w.k, m = w.m, rm = w.rmax;
  const int R = w.R[nb], r = w.rr[nb];
  // The target is symmetric: only its lower triangle is kept, packed by rows (entry (i, j), i >= j, at i (i + 1) / 2 + j) -- 40 KB instead
  // of 80 KB at n = 100, which is what lets two workgroups share a CU.
  auto tY = [&]() { if constexpr (USE_LDS) return (double*)smem; else return w.glob_scratch + (size_t)b * w.glob_scratch_stride; }();
  auto TRIX = [](int i, int j) { return (i >= j) ? ((i * (i + 1)) >> 1) + j : ((j * (j + 1)) >> 1) + i; };
  double* tU = tY + (size_t)n * (n + 1) / 2;       // n*k   (tU, then the full-space U correction)
  double* tV = tU + (size_t)n * k;       // rm*k
  double* cvec = tV + (size_t)rm * k;    // Rmax
  double* mu = cvec + w.Rmax;            // Rmax
  double* lam = w.lam + (size_t)b * w.Rmax;
  double* Y = w.Y + (size_t)b * n * n;
  double* Yp = w.Yp + (size_t)b * n * n;
  double* U = w.U + (size_t)b * n * k;
  double* D1 = w.D1 + (size_t)b * n * n;
  double* D3 = w.D3 + (size_t)b * n * n;
  double* Vt = w.Vt + (size_t)b * rm * k;
  double* D3V = w.D3V + (size_t)b * rm * k;
  const double* W1 = w.W1 + (size_t)b * n * n;
  const double* E3 = w.E3 + (size_t)b * n * n;
  const double* W3V = w.W3V + (size_t)b * rm * k;
  const double* Q = w.Qb + (size_t)nb * n * rm;
  const double rho = w.rho_b[b], rho_f = rho * w.rho_f_ratio, rx = w.relax, g = w.gamma;
  STAMP_BEGIN();
  DIAG_T0();
  // 1. (the product gamma/2 * Lambda Lambda' is added to the target in step 2, by v_mfma_f64_16x16x4_f64 with the operands -- the dense
  //    column-major copy of Lambda written by k_colprox, zero where a row is not observed -- straight from L2)
  const double* lamD = w.lamD + (size_t)b * m * n;
  const int nt16 = (n + 15) >> 4;
  STAMP(8);
  // 2. cone + multiplicity part of the target (lower triangle), then the Lambda term, then the weights
  for (int e = tid; e < n * n; e += T) {
    const int i = e % n, j = e / n;
    if (i < j) continue;
    double y = Y[e];
    // Shor mode: the third copy of Y is the big cone's (over-relaxed projection + its scaled dual) instead of the column blocks
    const double first = w.shor ? rho * (rx * w.shP0[(size_t)b * w.shN * w.shN + (size_t)j * w.shN + i] + (1.0 - rx) * y + w.shD0[(size_t)b * w.shN * w.shN + (size_t)j * w.shN + i])
                                : rho_f * w.Ncnt[e] * y;
    tY[TRIX(i, j)] = first + rho * (rx * W1[e] + (1.0 - rx) * y + D1[e]) + rho * (y + (1.0 - rx) * D3[e] + rx * E3[e]);
  }
  for (int e = tid; e < r * k; e += T) {
    int a = e % r, j = e / r;
    tV[(size_t)j * rm + a] = rx * W3V[(size_t)j * rm + a] + (1.0 - rx) * Vt[(size_t)j * rm + a] + D3V[(size_t)j * rm + a];
  }
  for (int e = tid; e < R; e += T) mu[e] = lam[e] / rho;
  __syncthreads();
  if (!w.shor) mfma_LLt(lamD, n, m, [&](int i, int j, double v) { tY[TRIX(i, j)] += 0.5 * g * v; });      // one wave owns a tile: deterministic
  __syncthreads();
  for (int e = tid; e < n * n; e += T) {
    const int i = e % n, j = e / n;
    if (i >= j) tY[TRIX(i, j)] /= (rho * w.wY1[e]);
  }
  for (int e = tid; e < n * k; e += T) {
    int i = e % n, j = e / n;
    double acc = 0.0;
    for (int a = 0; a < r; ++a) acc += Q[(size_t)a * n + i] * tV[(size_t)j * rm + a];
    tU[e] = acc;
  }
  __syncthreads();
  STAMP(9);
  // 3. c = A t - b.  The quadratic forms x' tY x of the cut rows come from the matrix cores: the cut vectors of up to GL_XS rows are
  //    staged as the columns of X, wave w forms the 16-row tiles w, w + nw, .. of tY X and reduces x_i (tY X)_i over its rows; the
  //    waves' partial sums are added in a fixed order.
  const double* cutx = w.cutx + (size_t)nb * w.Lmax * n;
  double* xs = mu + w.Rmax;                              // GL_XS * n: staged cut vectors
  double* qrow = xs + (size_t)GL_XS * n;                 // Rmax: x' tY x of the cut rows
  __shared__ int s_crow[GL_XS]; __shared__ int s_nc, s_rnext; __shared__ double s_qp[8][GL_XS];   // 512 threads = 8 waves
  if (tid == 0) s_rnext = 0;
  for (;;) {
    __syncthreads();
    if (tid == 0) {
      int c = 0, rr = s_rnext;
      for (; rr < R && c < GL_XS; ++rr) if (w.rkind[(size_t)nb * w.Rmax + rr] == ROW_CUT) s_crow[c++] = rr;
      s_nc = c; s_rnext = rr;
    }
    __syncthreads();
    const int nc = s_nc, rnext = s_rnext;
    if (nc == 0) break;
    for (int e = tid; e < nc * n; e += T) { const int c = e / n, i = e - c * n; xs[e] = cutx[(size_t)w.rcut[(size_t)nb * w.Rmax + s_crow[c]] * n + i]; }
    __syncthreads();
    {
      const int wv_ = tid >> 6, lane_ = tid & 63, nw_ = T >> 6, li = lane_ & 15, lk = lane_ >> 4;
      const bool vc = li < nc;
      const double* xc = xs + (size_t)(vc ? li : 0) * n;
      double qsum = 0.0;
      for (int ti = wv_; ti < nt16; ti += nw_) {
        const int ia = (ti << 4) + li; const bool va = ia < n; const int ic = va ? ia : 0;
        double4v acc = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < n; k0 += 4) {
          const int kk = k0 + lk; const bool vk = kk < n; const int kc = vk ? kk : 0;
          const double a = tY[TRIX(ic, kc)], bv = xc[kc];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64((va && vk) ? a : 0.0, (vc && vk) ? bv : 0.0, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int i = (ti << 4) + lk + 4 * r4;
          const double xv = xc[(i < n) ? i : 0];
          qsum += (vc && i < n) ? xv * acc[r4] : 0.0;
        }
      }
      qsum += __shfl_xor(qsum, 16, WAVE);
      qsum += __shfl_xor(qsum, 32, WAVE);
      if (lk == 0) s_qp[wv_][li] = qsum;
    }
    __syncthreads();
    if (tid < nc) {
      double q = 0.0;
      for (int v = 0; v < (T >> 6); ++v) q += s_qp[v][tid];
      qrow[s_crow[tid]] = q;
    }
    if (rnext >= R) break;
  }
  __syncthreads();
  {   // one WAVE per row (wave-level reductions, no workgroup barrier per row)
    const int wv_ = tid >> 6, lane_ = tid & 63, nw_ = T >> 6;
    for (int rr = wv_; rr < R; rr += nw_) {
      const int kind = w.rkind[(size_t)nb * w.Rmax + rr];
      double acc = 0.0;
      if (kind == ROW_TRACE) {
        for (int i = lane_; i < n; i += WAVE) acc += tY[TRIX(i, i)];
      } else if (kind == ROW_BOX) {
        if (lane_ == 0) acc = w.rcoef[((size_t)nb * w.Rmax + rr) * k] * tU[(size_t)w.rbj[(size_t)nb * w.Rmax + rr] * n + w.rbi[(size_t)nb * w.Rmax + rr]];
      } else {
        const double* x = cutx + (size_t)w.rcut[(size_t)nb * w.Rmax + rr] * n;
        const double* cf = w.rcoef + ((size_t)nb * w.Rmax + rr) * k;
        if (kind == ROW_CUT && lane_ == 0) acc = qrow[rr];
        for (int j = 0; j < k; ++j) {
          const double cj = cf[j];
          if (cj != 0.0) for (int i = lane_; i < n; i += WAVE) acc += cj * x[i] * tU[(size_t)j * n + i];
        }
      }
      const double tot = wave_sum(acc);
      if (lane_ == 0) cvec[rr] = tot - w.rrhs[(size_t)nb * w.Rmax + rr];
    }
  }
  __syncthreads();
  STAMP(10);
  // 4. multipliers (mu = lam / rho)
  if (tid < WAVE) { const int ov = wave_nnqp(w.G + (size_t)b * w.Rmax * w.Rmax, w.Rmax, cvec, mu, R, s_Gp, s_sv, s_tmp, s_pl, tid); if (tid == 0) w.rowov[b] = ov; }
  __syncthreads();
  for (int e = tid; e < R; e += T) lam[e] = rho * mu[e];
  STAMP(11);
  // 5. U-space correction  corr = A_U' mu / 2  (n x k, stored over tU)  and Vn = tV - Q' corr
  for (int e = tid; e < n * k; e += T) {
    int i = e % n, j = e / n;
    double corr = 0.0;
    for (int rr = 0; rr < R; ++rr) { double mv = mu[rr]; if (mv != 0.0) corr += mv * rowU_entry(w, nb, rr, i, j); }
    tU[e] = 0.5 * corr;
  }
  __syncthreads();
  double rp2 = 0.0, rd2 = 0.0, fr2 = 0.0, tr1 = 0.0;
  const int NP = w.np16;
  double* Mb = w.Mbuf + (size_t)b * NP * NP;
  for (int e = tid; e < r * k; e += T) {
    int a = e % r, j = e / r;
    double qc = 0.0;
    for (int i = 0; i < n; ++i) qc += Q[(size_t)a * n + i] * tU[(size_t)j * n + i];
    const size_t ix = (size_t)j * rm + a;
    double vn = tV[ix] - qc, vold = Vt[ix], w3 = W3V[ix];
    D3V[ix] += rx * w3 + (1.0 - rx) * vold - vn;
    rp2 += 2.0 * (w3 - vn) * (w3 - vn); rd2 += 2.0 * (vn - vold) * (vn - vold);
    tV[ix] = vn;
  }
  for (int e = tid; e < k * k; e += T) {
    double zi = ((e % k) == (e / k)) ? 1.0 : 0.0;
    double w3t = w.W3T[(size_t)b * k * k + e];
    w.D3T[(size_t)b * k * k + e] += rx * (w3t - zi);
    rp2 += (w3t - zi) * (w3t - zi);
  }
  __syncthreads();
  for (int e = tid; e < r * k; e += T) { int a = e % r, j = e / r; Vt[(size_t)j * rm + a] = tV[(size_t)j * rm + a]; }
  for (int e = tid; e < n * k; e += T) {  // U = Q Vt (iterate in the subspace; a feasible U is recovered at the end)
    int i = e % n, j = e / n;
    double acc = 0.0;
    for (int a = 0; a < r; ++a) acc += Q[(size_t)a * n + i] * tV[(size_t)j * rm + a];
    U[e] = acc;
  }
  STAMP(12);
  // 6. Y: every thread owns one entry (all quantities are symmetric, so no mirrored stores: coalesced reads and writes);
  //    the active cut rows are staged once in LDS
  __shared__ int s_nact; __shared__ int s_act[NNQP_PMAX]; __shared__ double s_mu[NNQP_PMAX]; __shared__ double s_trace_mu;
  if (tid == 0) {
    int c2 = 0; double tm = 0.0;
    for (int rr = 0; rr < R; ++rr) {
      const double mv = mu[rr];
      if (mv == 0.0) continue;
      const int kind = w.rkind[(size_t)nb * w.Rmax + rr];
      if (kind == ROW_TRACE) tm += mv;
      else if (kind == ROW_CUT && c2 < NNQP_PMAX) { s_act[c2] = w.rcut[(size_t)nb * w.Rmax + rr]; s_mu[c2] = mv; ++c2; }
    }
    s_nact = c2; s_trace_mu = tm;
  }
  __syncthreads();
  const int nact = s_nact, nstg = (nact < GL_XS) ? nact : GL_XS;      // the first GL_XS active cut vectors are read from LDS
  for (int e = tid; e < nstg * n; e += T) { const int a = e / n, i = e - a * n; xs[e] = cutx[(size_t)s_act[a] * n + i]; }
  __syncthreads();
  for (int e = tid; e < n * n; e += T) {
    const int i = e % n, j = e / n;
    double corr = (i == j) ? s_trace_mu : 0.0;
    for (int a = 0; a < nstg; ++a) { const double* x = xs + (size_t)a * n; corr += s_mu[a] * (x[i] * x[j]); }             // (x_i x_j) first: exactly symmetric in (i, j)
    for (int a = nstg; a < nact; ++a) { const double* x = cutx + (size_t)s_act[a] * n; corr += s_mu[a] * (x[i] * x[j]); }
    const size_t a1 = (size_t)j * n + i;
    const double t = tY[TRIX(i, j)];
    const double yn = t - corr / w.wY1[a1];
    const double yold = Y[a1];
    const double w1 = W1[a1], e3 = E3[a1], d3 = D3[a1];
    const double d1n = D1[a1] + rx * w1 + (1.0 - rx) * yold - yn;
    const double w3y = yold - d3 + e3;                 // projection output of the small-cone block
    const double d3n = (1.0 - rx) * d3 + yold + rx * e3 - yn;
    D1[a1] = d1n; D3[a1] = d3n;
    const double mv = yn - d1n;                        // next input of the cone block
    Mb[(size_t)j * NP + i] = mv;
    fr2 += mv * mv;
    if (i == j) tr1 += mv;
    rp2 += (w1 - yn) * (w1 - yn) + (w3y - yn) * (w3y - yn);
    rd2 += (yn - yold) * (yn - yold);
    Yp[a1] = yold; Y[a1] = yn;
    if (w.Yx) w.Yx[(size_t)b * n * n + a1] = 2.0 * yn - yold;      // the matrix the column prox of the next iteration gathers
  }
  STAMP(13);
  rp2 = block_sum(rp2, red);
  rd2 = block_sum(rd2, red);
  fr2 = block_sum(fr2, red);
  tr1 = block_sum(tr1, red);
  if (tid == 0) { w.rp[b] = sqrt(rp2); w.rd[b] = sqrt(rd2); w.fro2[b] = fr2; w.trM[b] = tr1; w.cone_done[b] = 0; w.iters[b] += 1; if (w.ws_first) w.ws_first[b] = (!w.sub_on[b] || w.sub_wait[b] > 0) ? 1 : 0; DIAG_CYC(4, b); }
}

// ---------------------------------------------------------------------------------------------------------
// certificate kernels
//   k_check_build : Mchk = -gamma/2 Lx Lx' + sum_cut lam x x' - rho E3   (E3 = Q Q3_11 Q' from the last k_small)
//                   cpen = sum_j || Q'(sum_r lam_r CU_r)_j - 2 rho Q3V_j || ;  cst = -sum_{r != trace} lam_r rhs_r - rho tr(Q3T)
//   (k_cone CONE_EVALS then fills evsum)
//   k_check_final : lb = c0 + evsum - cpen + cst ; stop tests (gap, infeasibility, stall)
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) k_check_build(OmcWS w) {
  __shared__ double red[32];
  extern __shared__ double smem[];   // n*k doubles: cU
  const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  if (w.done[b]) return;
  const int nb = w.node_of[b];
  const int n = w.n, k = w.k, m = w.m, R = w.R[nb], rm = w.rmax, r = w.rr[nb];
  double* M = w.Mchk + (size_t)b * n * n;
  const double* E3 = w.E3 + (size_t)b * n * n;
  const double* lam = w.lam + (size_t)b * w.Rmax;
  const double* cutx = w.cutx + (size_t)nb * w.Lmax * n;
  const double* Q = w.Qb + (size_t)nb * n * rm;
  const double rho = w.rho_b[b], g = w.gamma;
  double* cU = w.chk_scratch + (size_t)b * n * k;
  {   // exact objective and Fenchel constant: per-column terms of k_colprox (mode 1) added in a fixed order (no atomics: run-to-run identical)
    double so = 0.0, sc = 0.0;
    for (int j = tid; j < m; j += T) { so += w.objcol[(size_t)b * m + j]; sc += w.c0col[(size_t)b * m + j]; }
    so = block_sum(so, red); sc = block_sum(sc, red);
    if (tid == 0) { w.obj[b] = so; w.c0[b] = sc; }
  }
  for (int e = tid; e < n * n; e += T) {
    int i = e % n, j = e / n;
    double v = -rho * E3[e];
    for (int rr = 0; rr < R; ++rr) {
      double lv = lam[rr];
      if (lv != 0.0 && w.rkind[(size_t)nb * w.Rmax + rr] == ROW_CUT) {
        const double* x = cutx + (size_t)w.rcut[(size_t)nb * w.Rmax + rr] * n;
        v += lv * x[i] * x[j];
      }
    }
    M[e] = v;
  }
  for (int e = tid; e < n * k; e += T) {
    int i = e % n, j = e / n;
    double cu = 0.0;
    for (int rr = 0; rr < R; ++rr) { double lv = lam[rr]; if (lv != 0.0) cu += lv * rowU_entry(w, nb, rr, i, j); }
    cU[e] = cu;
  }
  __syncthreads();
  const double* al = w.alphaX + (size_t)b * w.nnz;
  if (w.lamDX) {     // dense copy of the exact multipliers (k_colprox mode 1): one MFMA product instead of m scattered rank-one updates
    mfma_LLt(w.lamDX + (size_t)b * m * n, n, m, [&](int i, int j, double v) {
      const double t = 0.5 * g * v;
      M[(size_t)j * n + i] -= t;
      if (i != j) M[(size_t)i * n + j] -= t;
    });
    __syncthreads();
  } else {
    for (int j = 0; j < m; ++j) {
      const int off = w.col_ptr[j], c = w.col_ptr[j + 1] - off;
      for (int e = tid; e < c * c; e += T) {
        int p = e % c, q = e / c;
        M[(size_t)w.col_idx[off + q] * n + w.col_idx[off + p]] -= 0.5 * g * al[off + p] * al[off + q];
      }
      __syncthreads();
    }
  }
  double pen = 0.0;
  for (int j = 0; j < k; ++j) {
    double acc = 0.0;
    for (int a = tid; a < r; a += T) {
      double cv = -2.0 * rho * w.Q3V[(size_t)b * rm * k + (size_t)j * rm + a];
      for (int i = 0; i < n; ++i) cv += Q[(size_t)a * n + i] * cU[(size_t)j * n + i];
      acc += cv * cv;
    }
    pen += sqrt(block_sum(acc, red));
  }
  if (tid == 0) {
    double cst = 0.0;
    for (int rr = 0; rr < R; ++rr)
      if (w.rkind[(size_t)nb * w.Rmax + rr] != ROW_TRACE) cst -= lam[rr] * w.rrhs[(size_t)nb * w.Rmax + rr];
    for (int j = 0; j < k; ++j) cst -= rho * w.Q3T[(size_t)b * k * k + (size_t)j * k + j];
    w.cpen[b] = pen; w.cst[b] = cst;
  }
  if (w.MbufC) {   // zero-padded copy + Frobenius norm for the warm-started eigenvalue kernel
    const int NP = w.np16;
    double fr2 = 0.0, tr1 = 0.0;
    for (int e = tid; e < n * n; e += T) {
      const int i = e % n, j = e / n;
      const double mv = 0.5 * (M[e] + M[(size_t)i * n + j]);
      w.MbufC[(size_t)b * NP * NP + (size_t)j * NP + i] = mv; fr2 += mv * mv; if (i == j) tr1 += mv;
    }
    fr2 = block_sum(fr2, red);
    tr1 = block_sum(tr1, red);
    if (tid == 0) { w.fro2c[b] = fr2; w.trMc[b] = tr1; }
  }
}

// phase 2: the bound of this check is rigorous for every slot (no certificate estimator).  With the estimator (k_cone_sub<1>):
// phase 0 decides on the ESTIMATED bound for the slots that have one; a slot that would finish (or that has no estimate) is only
// flagged (w.confirm) and nothing is committed for it; k_cone_ws then evaluates the flagged slots rigorously and phase 1 repeats the
// logic for them with that bound.  w.lb (the bound that is reported) only ever receives rigorous values.
__global__ void k_check_final(OmcWS w, int last, int phase) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= w.B || w.done[b]) return;
  if (phase == 0 && w.confirm[b]) return;
  if (phase == 1 && !w.confirm[b]) return;
  const bool est = (phase == 0);
  const double lbv = (w.c0[b] + w.evsum[b] - w.cpen[b] + w.cst[b]) * w.inv_s2;      // Shor mode solves for scale * A: values are reported unscaled
  const double lb_rig = w.lb[b];
  const double lb_dec = fmax(est ? fmax(lb_rig, w.lb_est[b]) : lb_rig, lbv);      // the bound the decisions of this check use
  const double obj = w.obj[b] * w.inv_s2;
  const double Nk = (double)(w.n + w.k + (w.shor ? w.m : 0));
  const bool feas = w.rp[b] <= w.eps_feas * sqrt(Nk) && !w.rowov[b];   // rows deferred by the NNQP cap: not a feasible point
  int term = -1;
  int stall_new = w.stall[b], votes_new = w.slow_votes[b];
  double gnow = 0.0, q = 1.0;
  // two-sided: the primal value of an eps-feasible iterate can sit BELOW the certified bound (by ~ residual x ||multipliers||); such a
  // point is not within eps_gap of the optimum value, so the node keeps iterating until feasibility has pulled the value up
  if (fabs(obj - lb_dec) <= w.eps_gap * fmax(1.0, fabs(obj)) && feas) term = OMC_ST_OPTIMAL;
  // f(Y) <= f(0) = 1/2 ||A_Omega||^2 on the feasible set: a larger certified bound proves infeasibility
  else if (lb_dec > 0.5 * w.sumA2 * (1.0 + 1e-9) + 1e-9) term = OMC_ST_INFEASIBLE;
  else {
    // stationary primal value and no progress of the bound: give up with values (MOI.SLOW_PROGRESS)
    if (fabs(obj - w.objprev[b]) <= 1e-7 * fmax(1.0, fabs(obj)) && lbv <= w.lbprev[b] + 1e-7 * fmax(1.0, fabs(obj))) stall_new += 1;
    else stall_new = 0;
    if (stall_new >= w.stall_checks) {
      const bool okgap = fabs(obj - lb_dec) <= w.eps_gap * fmax(1.0, fabs(obj)) && w.rp[b] <= 10.0 * w.eps_feas * sqrt(Nk) && !w.rowov[b];
      term = okgap ? OMC_ST_OPTIMAL : OMC_ST_SLOW;
    } else if (last) term = last;
    else if (w.iters[b] >= w.max_iters) term = OMC_ST_SLOW;
    else if (w.early_stop_factor > 0.0) {
      // early SLOW_PROGRESS: the gap of a crawling node decays geometrically (measured: a constant factor per check).  With q the
      // exponential average of gap_now / gap_previous, closing the rest takes log(gap / target) / log(1 / q) checks; when that exceeds
      // early_stop_factor x the checks left before max_iters the node cannot be certified any more -- its values and its (valid) bound
      // are returned now instead of after the iteration cap.  Eight consecutive such predictions are required.
      const double target = w.eps_gap * fmax(1.0, fabs(obj));
      gnow = obj - lb_dec;
      const double gp = w.gap_prev[b];
      q = (gp < 1e299 && gp > 0.0 && gnow > 0.0) ? 0.5 * w.gap_rate[b] + 0.5 * fmin(gnow / gp, 2.0) : 1.0;
      const double left = (double)(w.max_iters - w.iters[b]) / (double)w.check_every;
      const double need = (q < 1.0) ? log(fmax(gnow, target) / target) / -log(q) : 1e300;
      const bool hopeless = w.iters[b] >= w.early_stop_after && gnow > target && need > w.early_stop_factor * left;   // an infeasible iterate understates the gap: no feasibility condition
      votes_new = hopeless ? votes_new + 1 : 0;
      if (votes_new >= 8) term = OMC_ST_SLOW;
    }
  }
  if (est && term >= 0) { w.confirm[b] = 1; return; }      // nothing committed: phase 1 decides on the rigorous bound
  // ---- commit ----------------------------------------------------------------------------------------------------------------
  if (!est) w.lb[b] = fmax(lb_rig, lbv);
  w.lb_est[b] = fmax(w.lb_est[b], lbv);
  w.objout[b] = obj;
  w.stall[b] = stall_new; w.objprev[b] = obj; w.lbprev[b] = lb_dec;
  if (w.early_stop_factor > 0.0 && term < 0) { w.gap_prev[b] = gnow; w.gap_rate[b] = q; w.slow_votes[b] = votes_new; }
  if (phase == 1) w.confirm[b] = 0;
  if (term >= 0) { w.done[b] = 1; w.status[b] = term; return; }
  // penalty bump: crawling nodes with active cuts show rp >> rd
  w.bfac[b] = 1.0;
  if (w.bump_max > 0 && w.iters[b] >= w.bump_after && w.nbump[b] < w.bump_max && w.iters[b] - w.lastbump[b] >= w.bump_gap &&
      w.rp[b] > w.bump_ratio * w.rd[b]) {
    w.bfac[b] = w.bump_factor; w.rho_b[b] *= w.bump_factor; w.nbump[b] += 1; w.lastbump[b] = w.iters[b];
    w.slow_votes[b] = 0; w.gap_rate[b] = 1.0;      // a new penalty changes the rate: the early-stop prediction starts over
  }
}

// scaled duals follow the penalty: D <- D / factor  (G1, wY1 and the multipliers lambda are penalty-free)
__global__ void __launch_bounds__(256) k_rho_rescale(OmcWS w) {
  const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  if (w.done[b]) return;
  const double f = w.bfac[b];
  if (f == 1.0) return;
  const double inv = 1.0 / f;
  const int n = w.n, k = w.k, rm = w.rmax;
  for (int e = tid; e < n * n; e += T) { w.D1[(size_t)b * n * n + e] *= inv; w.D3[(size_t)b * n * n + e] *= inv; }
  for (int e = tid; e < rm * k; e += T) w.D3V[(size_t)b * rm * k + e] *= inv;
  for (int e = tid; e < k * k; e += T) w.D3T[(size_t)b * k * k + e] *= inv;
  // the cone input buffer holds Y - D1: rebuild it with the rescaled dual
  const int NP = w.np16;
  double fr2 = 0.0, tr1 = 0.0;
  __shared__ double red[32];
  for (int e = tid; e < n * n; e += T) {
    const int i = e % n, j = e / n;
    const double mv = w.Y[(size_t)b * n * n + e] - w.D1[(size_t)b * n * n + e];
    w.Mbuf[(size_t)b * NP * NP + (size_t)j * NP + i] = mv; fr2 += mv * mv; if (i == j) tr1 += mv;
  }
  fr2 = block_sum(fr2, red);
  tr1 = block_sum(tr1, red);
  if (tid == 0) { w.fro2[b] = fr2; w.trM[b] = tr1; w.bfac[b] = 1.0; if (w.accel && w.aa_valid[b] != 2) w.aa_valid[b] = 0; }   // the map changed: restart the history
}

// ---------------------------------------------------------------------------------------------------------
// k_aa: Anderson acceleration (type II) of the ADMM fixed-point map T, one workgroup per slot, after every iteration.
//   z = (Y, Yp, D1, D3, Vt, D3V, D3T, alpha) is the state an iteration reads; the arrays hold g = T(z) when this runs.
//   Every iteration (from aa_start on): f = g - zin, push (f, g) on the ring.  Every aa_every-th iteration, with h >= 3
//   entries: gamma = argmin || f - dF gamma ||^2 + reg (normal equations on the h-1 residual differences, Cholesky),
//   z+ = g - dG gamma replaces the state.  The point is verified by the NEXT ordinary iteration: if its residual
//   ||T(z+) - z+|| exceeds aa_safeguard * ||f||, the state is put back to g and the history is dropped; otherwise the
//   iteration simply continues from T(z+) -- a verified point costs no extra evaluation of T.
//   Certificates (k_check_*) run before this kernel, i.e. on images of T only.
// ---------------------------------------------------------------------------------------------------------
struct AaSeg { double* p; int len; };
__global__ void __launch_bounds__(512) k_aa(OmcWS w) {
  __shared__ double red[32];
  __shared__ double s_H[AA_MAXMEM * AA_MAXMEM], s_rhs[AA_MAXMEM], s_gam[AA_MAXMEM];
  __shared__ int s_ok;
  const int b = slot_of(w, blockIdx.x), tid = threadIdx.x, T = blockDim.x;
  if (w.done[b]) return;
  const int it = w.iters[b];
  if (it < w.aa_start - 1) return;
  if (w.aa_valid[b] == 2) return;     // switched off for this node: its extrapolated points keep being rejected
  const int n = w.n, k = w.k, rm = w.rmax, dim = w.aa_dim, M1 = w.aa_mem + 1;
  AaSeg seg[8] = {{w.Y + (size_t)b * n * n, n * n}, {w.Yp + (size_t)b * n * n, n * n}, {w.D1 + (size_t)b * n * n, n * n},
                  {w.D3 + (size_t)b * n * n, n * n}, {w.Vt + (size_t)b * rm * k, rm * k}, {w.D3V + (size_t)b * rm * k, rm * k},
                  {w.D3T + (size_t)b * k * k, k * k}, {w.alpha + (size_t)b * w.nnz, w.nnz}};
  double* zin = w.aa_zin + (size_t)b * dim;
  double* Fh = w.aa_F + (size_t)b * M1 * dim;
  double* Gh = w.aa_G + (size_t)b * M1 * dim;
  auto rebuild_cone_input = [&]() {   // Mbuf = Y - D1 (zero padded), fro2
    __syncthreads();
    const int NP = w.np16;
    double fr2 = 0.0, tr1 = 0.0;
    for (int e = tid; e < n * n; e += T) {
      const int i = e % n, j = e / n;
      const double mv = seg[0].p[e] - seg[2].p[e];
      w.Mbuf[(size_t)b * NP * NP + (size_t)j * NP + i] = mv; fr2 += mv * mv; if (i == j) tr1 += mv;
    }
    fr2 = block_sum(fr2, red);
    tr1 = block_sum(tr1, red);
    if (tid == 0) { w.fro2[b] = fr2; w.trM[b] = tr1; }
  };
  if (!w.aa_valid[b]) {   // first use, new node or a penalty bump: zin = state, empty history
    int o = 0;
    for (int sgi = 0; sgi < 8; ++sgi) { for (int e = tid; e < seg[sgi].len; e += T) zin[o + e] = seg[sgi].p[e]; o += seg[sgi].len; }
    __syncthreads();
    if (tid == 0) { w.aa_valid[b] = 1; w.aa_hist[b] = 0; w.aa_head[b] = 0; w.aa_pending[b] = 0; }
    return;
  }
  int hist = w.aa_hist[b], head = w.aa_head[b];
  const int pending = w.aa_pending[b];
  const int latest = (head + hist - 1 + M1) % M1;            // meaningful when hist > 0
  const int slot = (hist < M1) ? (head + hist) % M1 : head;   // where this iteration's (f, g) goes
  // ---- pass 1: f = g - zin, tentative push, ||f|| ----------------------------------------------------------------
  double acc = 0.0;
  {
    int o = 0;
    double* Fn = Fh + (size_t)slot * dim; double* Gn = Gh + (size_t)slot * dim;
    for (int sgi = 0; sgi < 8; ++sgi) {
      for (int e = tid; e < seg[sgi].len; e += T) {
        const double g = seg[sgi].p[e], f = g - zin[o + e];
        Fn[o + e] = f; Gn[o + e] = g; acc += f * f;
      }
      o += seg[sgi].len;
    }
  }
  const double fn = sqrt(block_sum(acc, red));
  if (pending && !(fn <= w.aa_safeguard * w.aa_fn[b])) {
    // reject: back to the image of the last plain step (ring entry `latest`), forget the history
    const double* Gl = Gh + (size_t)latest * dim;
    int o = 0;
    for (int sgi = 0; sgi < 8; ++sgi) {
      for (int e = tid; e < seg[sgi].len; e += T) { const double v = Gl[o + e]; seg[sgi].p[e] = v; zin[o + e] = v; }
      o += seg[sgi].len;
    }
    rebuild_cone_input();
    if (tid == 0) {
      w.aa_hist[b] = 0; w.aa_head[b] = 0; w.aa_pending[b] = 0; w.aa_nrej[b] += 1;
      // a node that rejects its points (non-smooth map: active sets keep changing) pays for the history without using it: stop there
      if (w.aa_nrej[b] >= 4 && w.aa_nrej[b] > w.aa_nacc[b]) w.aa_valid[b] = 2;
    }
    return;
  }
  if (pending && tid == 0) w.aa_nacc[b] += 1;
  if (hist < M1) ++hist; else head = (head + 1) % M1;
  const int h = hist;                                          // entries, chronological: idx(i) = (head + i) % M1
  bool extrapolate = (it % w.aa_every == 0) && h >= 3;
  if (extrapolate) {
    // ---- Gram matrix of the residual differences and right-hand side: one pass over the ring ---------------------
    const int nd = h - 1;
    double hacc[AA_MAXMEM * (AA_MAXMEM + 1) / 2], racc[AA_MAXMEM];
#pragma unroll
    for (int i = 0; i < AA_MAXMEM * (AA_MAXMEM + 1) / 2; ++i) hacc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < AA_MAXMEM; ++i) racc[i] = 0.0;
    const double* Fl = Fh + (size_t)((head + h - 1) % M1) * dim;
    for (int e = tid; e < dim; e += T) {
      double d[AA_MAXMEM];
      double prev = Fh[(size_t)(head % M1) * dim + e];
#pragma unroll
      for (int i = 0; i < AA_MAXMEM; ++i) {
        if (i < nd) { const double cur = Fh[(size_t)((head + i + 1) % M1) * dim + e]; d[i] = cur - prev; prev = cur; } else d[i] = 0.0;
      }
      const double fe = Fl[e];
#pragma unroll
      for (int i = 0; i < AA_MAXMEM; ++i) {
        racc[i] += d[i] * fe;
#pragma unroll
        for (int j = 0; j <= i; ++j) hacc[i * (i + 1) / 2 + j] += d[i] * d[j];
      }
    }
#pragma unroll
    for (int i = 0; i < AA_MAXMEM; ++i) {
      const double rv = block_sum(racc[i], red);
      if (tid == 0) s_rhs[i] = rv;
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        const double hv = block_sum(hacc[i * (i + 1) / 2 + j], red);
        if (tid == 0) { s_H[i * AA_MAXMEM + j] = hv; s_H[j * AA_MAXMEM + i] = hv; }
      }
    }
    __syncthreads();
    if (tid == 0) {
      double tr = 0.0;
      for (int i = 0; i < nd; ++i) tr += s_H[i * AA_MAXMEM + i];
      const double ridge = w.aa_reg * tr / nd;
      int ok = (tr > 0.0) && (tr < 1e300);
      for (int i = 0; i < nd; ++i) s_H[i * AA_MAXMEM + i] += ridge;
      // Cholesky H = L L', then two triangular solves
      for (int c = 0; c < nd && ok; ++c) {
        double dg = s_H[c * AA_MAXMEM + c];
        for (int q = 0; q < c; ++q) dg -= s_H[c * AA_MAXMEM + q] * s_H[c * AA_MAXMEM + q];
        if (!(dg > 0.0)) { ok = 0; break; }
        dg = sqrt(dg); s_H[c * AA_MAXMEM + c] = dg;
        for (int r2 = c + 1; r2 < nd; ++r2) {
          double v = s_H[r2 * AA_MAXMEM + c];
          for (int q = 0; q < c; ++q) v -= s_H[r2 * AA_MAXMEM + q] * s_H[c * AA_MAXMEM + q];
          s_H[r2 * AA_MAXMEM + c] = v / dg;
        }
      }
      if (ok) {
        for (int i = 0; i < nd; ++i) { double v = s_rhs[i]; for (int q = 0; q < i; ++q) v -= s_H[i * AA_MAXMEM + q] * s_gam[q]; s_gam[i] = v / s_H[i * AA_MAXMEM + i]; }
        for (int i = nd - 1; i >= 0; --i) { double v = s_gam[i]; for (int q = i + 1; q < nd; ++q) v -= s_H[q * AA_MAXMEM + i] * s_gam[q]; s_gam[i] = v / s_H[i * AA_MAXMEM + i]; }
        for (int i = 0; i < nd; ++i) if (!(fabs(s_gam[i]) < 1e300)) ok = 0;
      }
      s_ok = ok;
    }
    __syncthreads();
    extrapolate = s_ok != 0;
    if (extrapolate) {
      // ---- z+ = g - sum_i gamma_i (G_{i+1} - G_i):  coefficients of the ring entries ------------------------------
      double cg[AA_MAXMEM + 1];
#pragma unroll
      for (int i = 0; i <= AA_MAXMEM; ++i) {
        double c = 0.0;
        if (i < h) { if (i >= 1) c -= s_gam[i - 1]; if (i < nd) c += s_gam[i]; }
        if (i == h - 1) c += 1.0;
        cg[i] = c;
      }
      int o = 0;
      for (int sgi = 0; sgi < 8; ++sgi) {
        for (int e = tid; e < seg[sgi].len; e += T) {
          double v = 0.0;
#pragma unroll
          for (int i = 0; i <= AA_MAXMEM; ++i) if (i < h) v += cg[i] * Gh[(size_t)((head + i) % M1) * dim + o + e];
          seg[sgi].p[e] = v; zin[o + e] = v;
        }
        o += seg[sgi].len;
      }
      rebuild_cone_input();
    }
  }
  if (!extrapolate) {   // plain step: the next input is g
    int o = 0;
    for (int sgi = 0; sgi < 8; ++sgi) { for (int e = tid; e < seg[sgi].len; e += T) zin[o + e] = seg[sgi].p[e]; o += seg[sgi].len; }
  }
  __syncthreads();
  if (tid == 0) { w.aa_hist[b] = hist; w.aa_head[b] = head; w.aa_pending[b] = extrapolate ? 1 : 0; if (extrapolate) w.aa_fn[b] = fn; }
}

// copy the results of the slots flagged `fin` to the per-node output arrays (continuous batching: a slot is re-used)
__global__ void __launch_bounds__(256) k_harvest(OmcWS w) {
  const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  if (!w.fin[b]) return;
  const int nb = w.node_of[b];
  const int n = w.n, k = w.k;
  for (int e = tid; e < n * n; e += T) w.oY[(size_t)nb * n * n + e] = w.Y[(size_t)b * n * n + e];
  for (int e = tid; e < n * k; e += T) w.oU[(size_t)nb * n * k + e] = w.U[(size_t)b * n * k + e];
  for (int e = tid; e < w.nnz; e += T) w.oalphaX[(size_t)nb * w.nnz + e] = w.alphaX[(size_t)b * w.nnz + e];
  for (int e = tid; e < n; e += T) w.obx[(size_t)nb * n + e] = w.bx[(size_t)b * n + e];
  if (tid == 0) {
    w.oobj[nb] = w.objout[b]; w.olb[nb] = w.lb[b]; w.ostatus[nb] = w.status[b]; w.oiters[nb] = w.iters[b];
    w.olmin[2 * nb] = w.lmin[2 * b]; w.olmin[2 * nb + 1] = w.lmin[2 * b + 1];
    w.orho[nb] = w.rho_b[b];
  }
}

// warm-start pool: the final state of the slots flagged fin whose node asked for it (runs BEFORE the recovery of a feasible U overwrites the iterate U = Q Vt)
__global__ void __launch_bounds__(256) k_state_save(OmcWS w) {
  const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  if (!w.fin[b]) return;
  const int nb = w.node_of[b];
  const int sv = w.save_to ? w.save_to[nb] : -1;
  if (sv < 0) return;
  const int n = w.n, k = w.k, NP = w.np16;
  for (int e = tid; e < n * n; e += T) {
    w.pY[(size_t)sv * n * n + e] = w.Y[(size_t)b * n * n + e]; w.pD1[(size_t)sv * n * n + e] = w.D1[(size_t)b * n * n + e]; w.pD3[(size_t)sv * n * n + e] = w.D3[(size_t)b * n * n + e];
  }
  for (int e = tid; e < n * k; e += T) w.pU[(size_t)sv * n * k + e] = w.U[(size_t)b * n * k + e];
  for (int e = tid; e < w.nnz; e += T) w.palpha[(size_t)sv * w.nnz + e] = w.alpha[(size_t)b * w.nnz + e];
  for (int e = tid; e < w.m; e += T) w.psval[(size_t)sv * w.m + e] = w.sval[(size_t)b * w.m + e];
  for (int e = tid; e < NP * 16; e += T) w.pXs[(size_t)sv * NP * 16 + e] = w.Xs[(size_t)b * NP * 16 + e];
  if (tid < 16) w.ptheta[(size_t)sv * 16 + tid] = w.sub_theta[(size_t)b * 16 + tid];
  if (tid == 0) { w.pscal[(size_t)sv * 4] = w.rho_b[b]; w.pscal[(size_t)sv * 4 + 1] = w.sub_on[b] ? 1.0 : 0.0; w.pscal[(size_t)sv * 4 + 2] = 0.0; w.pscal[(size_t)sv * 4 + 3] = 0.0; }
}

__global__ void k_zero_check(OmcWS w) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= w.B || w.done[b]) return;
  w.obj[b] = 0.0; w.c0[b] = 0.0;
}

// ---------------------------------------------------------------------------------------------------------
// outputs: X = gamma * Y * Lx (n x m), Theta = gamma * Lx' X (m x m)
// ---------------------------------------------------------------------------------------------------------
__global__ void k_make_X(OmcWS w, double* X) {
  const int b = blockIdx.y, n = w.n, m = w.m;
  const double* Y = w.oY + (size_t)b * n * n;
  const double* al = w.oalphaX + (size_t)b * w.nnz;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n * m; e += gridDim.x * blockDim.x) {
    int i = e % n, j = e / n;
    const int off = w.col_ptr[j], c = w.col_ptr[j + 1] - off;
    double acc = 0.0;
    for (int p = 0; p < c; ++p) acc += Y[(size_t)w.col_idx[off + p] * n + i] * al[off + p];
    X[(size_t)b * n * m + e] = w.gamma * acc;
  }
}
__global__ void k_make_Theta(OmcWS w, const double* X, double* Th) {
  const int b = blockIdx.y, n = w.n, m = w.m;
  const double* al = w.oalphaX + (size_t)b * w.nnz;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < m * m; e += gridDim.x * blockDim.x) {
    int i = e % m, j = e / m;  // Theta[i][j] = gamma * sum_p Lx[p,i] X[p,j]
    const int off = w.col_ptr[i], c = w.col_ptr[i + 1] - off;
    double acc = 0.0;
    for (int p = 0; p < c; ++p) acc += al[off + p] * X[(size_t)b * n * m + (size_t)j * n + w.col_idx[off + p]];
    Th[(size_t)b * m * m + e] = w.gamma * acc;
  }
}

// ---------------------------------------------------------------------------------------------------------
// evaluate_objective (OMC.jl:2352-2358): one workgroup per matrix, coalesced column-major sweep
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_eval_objective(int n, int m, double gamma, const double* A, const uint8_t* mask,
                                                       const double* X, double* out) {
  __shared__ double red[32];
  const int b = blockIdx.x;
  const double* Xb = X + (size_t)b * n * m;
  double fit = 0.0, reg = 0.0;
  for (int e = threadIdx.x; e < n * m; e += blockDim.x) {
    double x = Xb[e];
    reg += x * x;
    if (mask[e]) { double d = x - A[e]; fit += d * d; }
  }
  fit = block_sum(fit, red);
  reg = block_sum(reg, red);
  if (threadIdx.x == 0) out[b] = 0.5 * fit + reg / (2.0 * gamma);
}

// ---------------------------------------------------------------------------------------------------------
// host-callable launchers
// ---------------------------------------------------------------------------------------------------------
template <int LPP>
static void launch_ws_lds(const OmcWS* w, int rpl2, size_t lds_bytes, hipStream_t s) {
  switch (rpl2) {   // straight-line (branch-free) row loops for the common sizes, run-time bound otherwise
    case 4: hipLaunchKernelGGL((k_cone_ws<LPP, true, 4>), dim3(w->nB), dim3(512), lds_bytes, s, *w); break;
    case 5: hipLaunchKernelGGL((k_cone_ws<LPP, true, 5>), dim3(w->nB), dim3(512), lds_bytes, s, *w); break;
    case 6: hipLaunchKernelGGL((k_cone_ws<LPP, true, 6>), dim3(w->nB), dim3(512), lds_bytes, s, *w); break;
    case 7: hipLaunchKernelGGL((k_cone_ws<LPP, true, 7>), dim3(w->nB), dim3(512), lds_bytes, s, *w); break;
    case 8: hipLaunchKernelGGL((k_cone_ws<LPP, true, 8>), dim3(w->nB), dim3(512), lds_bytes, s, *w); break;
    default: hipLaunchKernelGGL((k_cone_ws<LPP, true, 0>), dim3(w->nB), dim3(512), lds_bytes, s, *w); break;
  }
}
// Y[b] = X[b] X[b]' (both triangles) for B matrices X of size n x m (column-major), on the matrix cores: the Gram matrix whose k dominant eigenvectors
// are svd(X).U[:, 1:k] (OMC.jl:524, 564, 921: the rank-k rounding of an incumbent)
__global__ void __launch_bounds__(256) k_gram_XXt(OmcWS w, const double* X) {
  const int b = blockIdx.x, n = w.n;
  double* Y = w.Y + (size_t)b * n * n;
  mfma_LLt(X + (size_t)b * n * w.m, n, w.m, [&](int i, int j, double v) { Y[(size_t)j * n + i] = v; Y[(size_t)i * n + j] = v; });
}
extern "C" {
void omc_launch_gram_XXt(const OmcWS* w, const double* X, int B, hipStream_t s) { hipLaunchKernelGGL(k_gram_XXt, dim3(B), dim3(256), 0, s, *w, X); }
void omc_launch_setup(const OmcWS* w, hipStream_t s) { hipLaunchKernelGGL(k_setup, dim3(w->B), dim3(256), 0, s, *w); }
void omc_launch_colprox(const OmcWS* w, int mode, hipStream_t s) {
  const int wpb = 4;
  if (mode == 0 && w->cp_pair) {
    omc_launch_colprox_sweep(w, s);      // omc_colprox.hip: k_colprox_pair (two columns per wave) and k_colprox_wide (one column, up to 64 rows)
    if (w->cp_nsolo == 0) return;
  }
  const int waves = w->nB * ((mode == 0 && w->cp_pair) ? w->cp_nsolo : w->m);
  const int blocks = (waves + wpb - 1) / wpb;
  hipLaunchKernelGGL(k_colprox, dim3(blocks), dim3(wpb * 64), (size_t)wpb * w->cp_lds_doubles * sizeof(double), s, *w, mode);
}
void omc_launch_cone(const OmcWS* w, int mode, int use_lds, size_t lds_bytes, hipStream_t s) {
  if (use_lds) hipLaunchKernelGGL(k_cone<true>, dim3(w->nB), dim3(512), lds_bytes, s, *w, mode);
  else hipLaunchKernelGGL(k_cone<false>, dim3(w->nB), dim3(512), 0, s, *w, mode);
}
void omc_launch_cone_ws(const OmcWS* w, int lpp, int use_lds, size_t lds_bytes, hipStream_t s) {
  const int rpl2 = ((((w->n + lpp - 1) / lpp) + 1) & ~1) >> 1;
  if (use_lds) {
    if (lpp == 16) launch_ws_lds<16>(w, rpl2, lds_bytes, s);
    else if (lpp == 8) launch_ws_lds<8>(w, rpl2, lds_bytes, s);
    else launch_ws_lds<4>(w, rpl2, lds_bytes, s);
  } else {
    // L2-resident G (n > 144): 1024 threads = 64 pair groups halve the passes per step (the kernel is bound by L2 latency there)
    if (lpp == 64) {      // orders 513 .. 1024: a whole wave per column pair (16 rows per lane), G in global scratch
      if (rpl2 == 8) hipLaunchKernelGGL((k_cone_ws<64, false, 8, 1024>), dim3(w->nB), dim3(1024), 0, s, *w);
      else hipLaunchKernelGGL((k_cone_ws<64, false, 0, 1024>), dim3(w->nB), dim3(1024), 0, s, *w);
    } else if (rpl2 == 7 && !w->cone_512) hipLaunchKernelGGL((k_cone_ws<16, false, 7, 1024>), dim3(w->nB), dim3(1024), 0, s, *w);
    else if (rpl2 == 8 && !w->cone_512) hipLaunchKernelGGL((k_cone_ws<16, false, 8, 1024>), dim3(w->nB), dim3(1024), 0, s, *w);
    else hipLaunchKernelGGL((k_cone_ws<16, false, 0>), dim3(w->nB), dim3(512), 0, s, *w);
  }
}
size_t omc_cone_sub_lds(int np16) { return ((size_t)(np16 > 512 ? 1 : 2) * SUBP * (np16 + 2) + 4 * 256 + 2 * 16 * 17 + 16 + 16 + 32 + 16 + 8) * sizeof(double); }
void omc_launch_cone_sub(const OmcWS* w, hipStream_t s) {
  hipLaunchKernelGGL(k_cone_sub<0>, dim3(w->nB), dim3(256), omc_cone_sub_lds(w->np16), s, *w);
}
void omc_launch_sep_sub(const OmcWS* w, hipStream_t s) {
  hipLaunchKernelGGL(k_sep_prepare, dim3(w->B), dim3(256), 0, s, *w);
  hipLaunchKernelGGL(k_cone_sub<2>, dim3(w->B), dim3(256), omc_cone_sub_lds(w->np16), s, *w);
}
void omc_launch_cert_sub(const OmcWS* w, hipStream_t s) {
  hipLaunchKernelGGL(k_cone_sub<1>, dim3(w->B), dim3(256), omc_cone_sub_lds(w->np16), s, *w);
}
void omc_launch_small(const OmcWS* w, int mode, int use_lds, size_t lds_bytes, hipStream_t s) {
  if (use_lds) hipLaunchKernelGGL(k_small<true>, dim3(w->nB), dim3(256), lds_bytes, s, *w, mode);
  else hipLaunchKernelGGL(k_small<false>, dim3(w->nB), dim3(256), 0, s, *w, mode);
}
void omc_launch_global(const OmcWS* w, int use_lds, size_t lds_bytes, hipStream_t s) {
  if (use_lds) hipLaunchKernelGGL(k_global<true>, dim3(w->nB), dim3(512), lds_bytes, s, *w);
  else hipLaunchKernelGGL(k_global<false>, dim3(w->nB), dim3(512), 0, s, *w);
}
void omc_launch_check_zero(const OmcWS* w, hipStream_t s) { hipLaunchKernelGGL(k_zero_check, dim3((w->B + 63) / 64), dim3(64), 0, s, *w); }
void omc_launch_check_build(const OmcWS* w, hipStream_t s) { hipLaunchKernelGGL(k_check_build, dim3(w->B), dim3(512), 0, s, *w); }
void omc_launch_check_final(const OmcWS* w, int last, int phase, hipStream_t s) {
  hipLaunchKernelGGL(k_check_final, dim3((w->B + 63) / 64), dim3(64), 0, s, *w, last, phase);
}
void omc_launch_rho_rescale(const OmcWS* w, hipStream_t s) { hipLaunchKernelGGL(k_rho_rescale, dim3(w->B), dim3(256), 0, s, *w); }
void omc_launch_harvest(const OmcWS* w, hipStream_t s) { hipLaunchKernelGGL(k_harvest, dim3(w->B), dim3(256), 0, s, *w); }
void omc_launch_state_save(const OmcWS* w, hipStream_t s) { hipLaunchKernelGGL(k_state_save, dim3(w->B), dim3(256), 0, s, *w); }
void omc_launch_aa(const OmcWS* w, hipStream_t s) { hipLaunchKernelGGL(k_aa, dim3(w->nB), dim3(512), 0, s, *w); }
void omc_launch_make_X(const OmcWS* w, double* X, hipStream_t s) { hipLaunchKernelGGL(k_make_X, dim3(64, w->Btot), dim3(256), 0, s, *w, X); }
void omc_launch_make_Theta(const OmcWS* w, const double* X, double* Th, hipStream_t s) {
  hipLaunchKernelGGL(k_make_Theta, dim3(64, w->Btot), dim3(256), 0, s, *w, X, Th);
}
void omc_launch_eval_objective(int B, int n, int m, double gamma, const double* A, const uint8_t* mask, const double* X,
                               double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_eval_objective, dim3(B), dim3(256), 0, s, n, m, gamma, A, mask, X, out);
}
int omc_set_max_lds(void) {
  hipError_t e1 = hipFuncSetAttribute((const void*)k_cone<true>, hipFuncAttributeMaxDynamicSharedMemorySize, OMC_MAX_DYN_LDS);
  hipError_t e2 = hipFuncSetAttribute((const void*)k_global<true>, hipFuncAttributeMaxDynamicSharedMemorySize, OMC_MAX_DYN_LDS - 8 * 1024);   // ~19 KB of static LDS (NNQP scratch)
  hipError_t e3 = hipFuncSetAttribute((const void*)k_colprox, hipFuncAttributeMaxDynamicSharedMemorySize, OMC_MAX_DYN_LDS);
  hipError_t e4 = hipFuncSetAttribute((const void*)k_small<true>, hipFuncAttributeMaxDynamicSharedMemorySize, OMC_MAX_DYN_LDS);
  (void)hipFuncSetAttribute((const void*)k_cone_sub<0>, hipFuncAttributeMaxDynamicSharedMemorySize, OMC_MAX_DYN_LDS);
  (void)hipFuncSetAttribute((const void*)k_cone_sub<1>, hipFuncAttributeMaxDynamicSharedMemorySize, OMC_MAX_DYN_LDS);
  (void)hipFuncSetAttribute((const void*)k_cone_sub<2>, hipFuncAttributeMaxDynamicSharedMemorySize, OMC_MAX_DYN_LDS);
#define WS_ATTR(L, R) (void)hipFuncSetAttribute((const void*)k_cone_ws<L, true, R>, hipFuncAttributeMaxDynamicSharedMemorySize, OMC_MAX_DYN_LDS)
#define WS_ATTR_ALL(L) WS_ATTR(L, 0); WS_ATTR(L, 4); WS_ATTR(L, 5); WS_ATTR(L, 6); WS_ATTR(L, 7); WS_ATTR(L, 8)
  WS_ATTR_ALL(4); WS_ATTR_ALL(8); WS_ATTR_ALL(16);
  if (e1 != hipSuccess) return 1000 + (int)e1;
  if (e2 != hipSuccess) return 2000 + (int)e2;
  if (e3 != hipSuccess) return 3000 + (int)e3;
  if (e4 != hipSuccess) return 4000 + (int)e4;
  return 0;
}
}
