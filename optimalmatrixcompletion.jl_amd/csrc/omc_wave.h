// omc_wave.h -- wave / workgroup level building blocks shared by the kernels (wavefront = 64)
#ifndef OMC_WAVE_H
#define OMC_WAVE_H
#include <hip/hip_runtime.h>
#include "omc_device.h"

#define WAVE 64

// slot of the i-th workgroup of a launch over the compact list of live slots
__device__ __forceinline__ int slot_of(const OmcWS& w, int i) { return w.slot_list ? w.slot_list[w.b0 + i] : w.b0 + i; }

// ---------------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------------
// ---- cross-lane sums on the VALU (DPP), no LDS traffic: lanes are grouped 4 / 8 / 16 wide inside a row of 16 ----
__device__ __forceinline__ double dpp_move(double v, const int ctrl_sel) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  int lo2, hi2;
  switch (ctrl_sel) {
    case 0: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); break;   // quad_perm [1,0,3,2]
    case 1: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); break;   // quad_perm [2,3,0,1]
    case 2: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, true); break; // row_half_mirror
    default: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, true); break; // row_mirror
  }
  return __hiloint2double(hi2, lo2);
}
template <int LPP>
__device__ __forceinline__ double group_sum_dpp(double v) {
  v += dpp_move(v, 0);
  v += dpp_move(v, 1);
  if (LPP >= 8) v += dpp_move(v, 2);
  if (LPP >= 16) v += dpp_move(v, 3);
  return v;
}

__device__ __forceinline__ double readlane_d(double v, int l) {   // l must be wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// full-wave sum without LDS traffic: DPP inside each row of 16 lanes, then four scalar reads
__device__ __forceinline__ double wave_sum(double v) {
  v = group_sum_dpp<16>(v);
  return readlane_d(v, 0) + readlane_d(v, 16) + readlane_d(v, 32) + readlane_d(v, 48);
}
__device__ __forceinline__ double group_sum(double v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
// block-wide sum, result valid in every thread; red must hold >= 32 doubles
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}

#define TRI(r, q) ((size_t)(r) * ((r) + 1) / 2 + (q))
#define WAVE_SYNC()                                      \
  do {                                                   \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                     \
  } while (0)

// in-place Cholesky of a symmetric matrix stored as packed lower triangle (row r holds q = 0..r)
__device__ __forceinline__ bool wave_cholesky(double* Lm, int c, int lane) {
  for (int kk = 0; kk < c; ++kk) {
    double piv = Lm[TRI(kk, kk)];
    if (!(piv > 0.0)) return false;
    double d = sqrt(piv);
    WAVE_SYNC();
    for (int r = kk + lane; r < c; r += WAVE) {
      double v = (r == kk) ? d : Lm[TRI(r, kk)] / d;
      Lm[TRI(r, kk)] = v;
    }
    WAVE_SYNC();
    const int t = c - kk - 1;
    const int tot = t * t;
    for (int e = lane; e < tot; e += WAVE) {
      int rr = e / t, qq = e - rr * t;
      if (qq <= rr) {
        int r = kk + 1 + rr, q = kk + 1 + qq;
        Lm[TRI(r, q)] -= Lm[TRI(r, kk)] * Lm[TRI(q, kk)];
      }
    }
    WAVE_SYNC();
  }
  return true;
}

// solve L L' y = rhs with the vector held in registers (lane r owns entry r, c <= 64): column-oriented substitution,
// one scalar broadcast (v_readlane) per step instead of a cross-lane reduction per row.  dinv[q] = 1 / L[q][q].
template <class LP, class DP>
__device__ __forceinline__ double wave_chol_solve_reg(LP Lm, DP dinv, int c, double rhs, int lane) {
  double y = (lane < c) ? rhs : 0.0;
  for (int q = 0; q < c; ++q) {
    const double zq = readlane_d(y, q) * dinv[q];
    if (lane == q) y = zq;
    else if (lane > q && lane < c) y -= Lm[TRI(lane, q)] * zq;
  }
  for (int q = c - 1; q >= 0; --q) {
    const double xq = readlane_d(y, q) * dinv[q];
    if (lane == q) y = xq;
    else if (lane < q) y -= Lm[TRI(q, lane)] * xq;
  }
  return y;
}


// ---- root-free Cholesky (B = L D L', unit lower L) for c <= 64, packed lower triangle --------------------------------
// Lane layout: G = 64 / c lanes per row; lane (r, g) owns the entries (r, q) with q = g (mod G).  Step kk needs ONE wave
// sync: the update  L(r,q) -= L(r,kk) L(q,kk) / piv  reads only column kk, which no lane writes during the step, so neither
// a square root nor a scaled copy of the column is needed.  On exit Lm(r,q) = L(r,q) d_q below the diagonal and pinv[q] = 1/d_q.
__device__ __forceinline__ int tri_i(int r, int q) { return (int)(__umul24(r, r + 1) >> 1) + q; }
// 1 / p for p > 0 in the normal range: v_rcp_f64 + two Newton steps (no scaling / fix-up sequence of the IEEE division)
__device__ __forceinline__ double fast_rcp(double p) {
  double x = __builtin_amdgcn_rcp(p);
  x = fma(fma(-p, x, 1.0), x, x);
  x = fma(fma(-p, x, 1.0), x, x);
  return x;
}
// Register-panel version: lane r owns row r.  Panels of 8 columns are factorised entirely in registers (each lane holds its 8
// panel entries; pivots and pivot-column entries of other rows arrive by v_readlane, so no LDS traffic and no barrier), then every
// trailing column q gets its rank-8 update  L(r,q) -= sum_j L(r,kb+j) L~(q,kb+j)  with the eight L~(q,.) broadcast by v_readlane:
// one LDS read-modify-write per (row, column, panel) instead of one per (row, column, pivot).  No cross-lane LDS communication at
// all, hence a single wave sync at the end.  On exit Lm(r,q) = L(r,q) d_q below the diagonal, Lm(q,q) = d_q, pinv[q] = 1/d_q.
template <class PT>
__device__ __forceinline__ bool wave_ldl(PT Lm, PT pinv, int c, int lane) {
  const bool mine = lane < c;
  const int rb = tri_i(mine ? lane : 0, 0);
  WAVE_SYNC();
  for (int kb = 0; kb < c; kb += 8) {
    const int nb = (c - kb < 8) ? (c - kb) : 8;
    double P[8], pis[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool v = mine && (kb + j <= lane);
      const double x = Lm[rb + (v ? kb + j : 0)];      // clamped address, never a guarded load
      P[j] = v ? x : 0.0;
      pis[j] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < nb) {                                    // wave-uniform
        const double piv = readlane_d(P[j], kb + j);
        if (!(piv > 1e-290)) return false;             // not positive definite (also keeps fast_rcp in its normal range)
        const double pi_ = fast_rcp(piv);
        pis[j] = pi_;
        const double lr = P[j] * pi_;                  // unit-lower entry L(r, kb+j) for the rows below the pivot
#pragma unroll
        for (int j2 = j + 1; j2 < 8; ++j2) {
          if (j2 < nb) {
            const double lq = readlane_d(P[j], kb + j2);
            if (lane >= kb + j2) P[j2] = fma(-lr, lq, P[j2]);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < nb) {
        if (mine && lane >= kb + j) Lm[rb + kb + j] = P[j];
        if (lane == j) pinv[kb + j] = pis[j];
      }
    }
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = P[j] * pis[j];
    for (int q = kb + nb; q < c; ++q) {                // wave-uniform loop over the trailing columns
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = fma(a[j], readlane_d(P[j], q), acc);
      if (mine && lane >= q) Lm[rb + q] -= acc;
    }
  }
  WAVE_SYNC();
  return true;
}
// solve L D L' x = rhs with the vector in registers (lane = entry); column-oriented substitutions, one v_readlane per step,
// operands of four steps fetched ahead of their use
template <class PT>
__device__ __forceinline__ double wave_ldl_solve_reg(PT Lm, PT pinv, int c, double rhs, int lane) {
  double y = (lane < c) ? rhs : 0.0;
  const int lb = tri_i(lane, 0);
  const bool act = lane < c;
  int q = 0;
  for (; q + 4 <= c; q += 4) {
    const double p0 = pinv[q], p1 = pinv[q + 1], p2 = pinv[q + 2], p3 = pinv[q + 3];
    const double l0 = (act && lane > q) ? Lm[lb + q] : 0.0, l1 = (act && lane > q + 1) ? Lm[lb + q + 1] : 0.0;
    const double l2 = (act && lane > q + 2) ? Lm[lb + q + 2] : 0.0, l3 = (act && lane > q + 3) ? Lm[lb + q + 3] : 0.0;
    y = fma(-l0, readlane_d(y, q) * p0, y);
    y = fma(-l1, readlane_d(y, q + 1) * p1, y);
    y = fma(-l2, readlane_d(y, q + 2) * p2, y);
    y = fma(-l3, readlane_d(y, q + 3) * p3, y);
  }
  for (; q < c; ++q) {
    const double t = readlane_d(y, q) * pinv[q];
    if (act && lane > q) y -= Lm[lb + q] * t;
  }
  q = c - 1;
  for (; q >= 3; q -= 4) {
    const double p0 = pinv[q], p1 = pinv[q - 1], p2 = pinv[q - 2], p3 = pinv[q - 3];
    const double l0 = (lane < q) ? Lm[tri_i(q, 0) + lane] : 0.0, l1 = (lane < q - 1) ? Lm[tri_i(q - 1, 0) + lane] : 0.0;
    const double l2 = (lane < q - 2) ? Lm[tri_i(q - 2, 0) + lane] : 0.0, l3 = (lane < q - 3) ? Lm[tri_i(q - 3, 0) + lane] : 0.0;
    double xq = readlane_d(y, q) * p0;
    y = (lane == q) ? xq : fma(-l0, xq, y);
    xq = readlane_d(y, q - 1) * p1;
    y = (lane == q - 1) ? xq : fma(-l1, xq, y);
    xq = readlane_d(y, q - 2) * p2;
    y = (lane == q - 2) ? xq : fma(-l2, xq, y);
    xq = readlane_d(y, q - 3) * p3;
    y = (lane == q - 3) ? xq : fma(-l3, xq, y);
  }
  for (; q >= 0; --q) {
    const double xq = readlane_d(y, q) * pinv[q];
    if (lane == q) y = xq;
    else if (lane < q) y -= Lm[tri_i(q, 0) + lane] * xq;
  }
  return y;
}

// solve L L' y = rhs (packed L); y, rhs length c
__device__ __forceinline__ void wave_chol_solve(const double* Lm, int c, const double* rhs, double* y, int lane) {
  for (int r = 0; r < c; ++r) {
    double p = 0.0;
    for (int q = lane; q < r; q += WAVE) p += Lm[TRI(r, q)] * y[q];
    p = wave_sum(p);
    if (lane == 0) y[r] = (rhs[r] - p) / Lm[TRI(r, r)];
    WAVE_SYNC();
  }
  for (int r = c - 1; r >= 0; --r) {
    double p = 0.0;
    for (int q = r + 1 + lane; q < c; q += WAVE) p += Lm[TRI(q, r)] * y[q];
    p = wave_sum(p);
    if (lane == 0) y[r] = (y[r] - p) / Lm[TRI(r, r)];
    WAVE_SYNC();
  }
}

// ---------------------------------------------------------------------------------------------------------
// NNQP on one wave:  min 1/2 l'Gl - c'l, l >= 0  (Lawson-Hanson active set in QP form, warm-started from the
// previous multipliers).  G is R x R (ld = Rmax) in global memory; the passive-set system (<= NNQP_PMAX) is
// solved by Cholesky in LDS with a tiny ridge (parallel rows make G singular).
// ---------------------------------------------------------------------------------------------------------
// Returns 1 (wave-uniform) when the passive set hit NNQP_PMAX with a violated row still outside it -- the result is then NOT the
// exact projection and the caller must not certify / accept the point (relaxation: no OPTIMAL status; altmin: converged = false).
__device__ __forceinline__ int wave_nnqp(const double* G, int ldG, const double* cvec, double* lam, int R, double* Gp, double* sv,
                          double* tmp, int* plist, int lane) {
  // plist: passive indices (LDS, NNQP_PMAX ints); Gp: PMAX x (PMAX+1); sv,tmp: PMAX doubles
  int np = 0;
  int overflow = 0;
  // warm start: passive = {lam > 0}
  for (int r = 0; r < R; ++r) {  // uniform over the wave (lam in LDS/global, same for all lanes)
    if (lam[r] > 0.0 && np < NNQP_PMAX) { if (lane == 0) plist[np] = r; ++np; }
    else if (lam[r] != 0.0) { if (lam[r] > 0.0) overflow = 1; if (lane == 0) lam[r] = 0.0; }
  }
  WAVE_SYNC();
  double cmax = 0.0;
  for (int r = lane; r < R; r += WAVE) cmax = fmax(cmax, fabs(cvec[r]));
  for (int o = 32; o > 0; o >>= 1) cmax = fmax(cmax, __shfl_xor(cmax, o, WAVE));
  const double tol = 1e-13 * fmax(cmax, 1e-300);
  bool need_inner = (np > 0);
  for (int outer = 0; outer < 3 * R + 10; ++outer) {
    if (!need_inner) {
      // w = c - G lam over non-passive rows; pick the max
      double best = -1e300; int bi = -1;
      for (int r = lane; r < R; r += WAVE) {
        bool inP = false;
        for (int a = 0; a < np; ++a) if (plist[a] == r) inP = true;
        if (inP) continue;
        double wv = cvec[r];
        for (int a = 0; a < np; ++a) wv -= G[(size_t)r * ldG + plist[a]] * lam[plist[a]];
        if (wv > best) { best = wv; bi = r; }
      }
      for (int o = 32; o > 0; o >>= 1) {
        double ob = __shfl_xor(best, o, WAVE); int oi = __shfl_xor(bi, o, WAVE);
        if (ob > best || (ob == best && oi >= 0 && (bi < 0 || oi < bi))) { best = ob; bi = oi; }
      }
      if (bi < 0 || best <= tol) { overflow = 0; break; }      // every row satisfied: whatever was dropped at the warm start is not needed
      if (np >= NNQP_PMAX) { overflow = 1; break; }
      if (lane == 0) plist[np] = bi;
      ++np;
      WAVE_SYNC();
    }
    need_inner = false;
    // inner loop
    for (int inner = 0; inner < 3 * NNQP_PMAX + 10; ++inner) {
      double dmax = 0.0;
      for (int e = lane; e < np * np; e += WAVE) {
        int a = e / np, c2 = e - a * np;
        if (c2 <= a) Gp[TRI(a, c2)] = G[(size_t)plist[a] * ldG + plist[c2]];
      }
      for (int a = 0; a < np; ++a) dmax = fmax(dmax, G[(size_t)plist[a] * ldG + plist[a]]);
      WAVE_SYNC();
      for (int a = lane; a < np; a += WAVE) { Gp[TRI(a, a)] += 1e-14 * dmax; tmp[a] = cvec[plist[a]]; }
      WAVE_SYNC();
      bool ok = wave_cholesky(Gp, np, lane);
      if (!ok) {  // numerically singular: drop the newest index
        --np; WAVE_SYNC();
        if (np == 0) break;
        continue;
      }
      wave_chol_solve(Gp, np, tmp, sv, lane);
      // all positive?
      double mins = 1e300;
      for (int a = lane; a < np; a += WAVE) mins = fmin(mins, sv[a]);
      for (int o = 32; o > 0; o >>= 1) mins = fmin(mins, __shfl_xor(mins, o, WAVE));
      if (mins > 0.0) {
        for (int a = lane; a < np; a += WAVE) lam[plist[a]] = sv[a];
        WAVE_SYNC();
        break;
      }
      // step toward s until the first multiplier hits zero
      double al = 1e300;
      for (int a = lane; a < np; a += WAVE) {
        double lv = lam[plist[a]];
        if (sv[a] <= 0.0) al = fmin(al, lv / (lv - sv[a]));
      }
      for (int o = 32; o > 0; o >>= 1) al = fmin(al, __shfl_xor(al, o, WAVE));
      if (!(al >= 0.0)) al = 0.0;
      for (int a = lane; a < np; a += WAVE) {
        double lv = lam[plist[a]];
        lam[plist[a]] = lv + al * (sv[a] - lv);
      }
      WAVE_SYNC();
      // remove zeros (serial compaction, uniform)
      int nn = 0; bool removed = false; double minl = 1e300; int mini = -1;
      for (int a = 0; a < np; ++a) {
        double lv = lam[plist[a]];
        if (sv[a] <= 0.0 && lv < minl) { minl = lv; mini = a; }
      }
      for (int a = 0; a < np; ++a) {
        int r = plist[a];
        double lv = lam[r];
        bool drop = (sv[a] <= 0.0) && (lv <= 1e-18 * fmax(cmax, 1e-300) || a == mini);
        WAVE_SYNC();
        if (drop) { if (lane == 0) lam[r] = 0.0; removed = true; }
        else { if (lane == 0) { plist[nn] = r; } ++nn; }
        WAVE_SYNC();
      }
      // sv must be compacted consistently: recomputed next inner iteration, so nothing to do
      np = nn;
      (void)removed;
      if (np == 0) break;
    }
  }
  return overflow;
}

#endif
