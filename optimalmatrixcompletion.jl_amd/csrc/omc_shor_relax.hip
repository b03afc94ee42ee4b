// omc_shor_relax.hip -- CDNA4 (gfx950) kernels of the Shor-mode node relaxation (rank 1).
//
// Reference program: matrix_completion_SDP_relaxation with add_Shor_valid_inequalities = true (OMC.jl = /root/reference/src/
// OptimalMatrixCompletion.jl: variables 1503-1525, rotated cones 1757-1762, Theta_jj = sum_i W_ij 1763-1767, one order-5 PSD block per
// minor 1768-1779, objective 1838-1846).  Formulation and splitting: oracle/omc_oracle_shor.py (header) and DESIGN.md section 3.7.
//
// One ADMM iteration in Shor mode (B = slots, all kernels skip finished slots):
//   big cone  P0 = P_+(G - D0), G = [Y X; X' Theta]      base kernels k_cone_ws / k_cone through a view (order n + m)   [stream b]
//   clip, small cone                                     base kernels                                                    [streams main, c]
//   k_shor_minor_pre   one LANE per minor: gather the order-5 block, Jacobi in registers, P_+, over-relaxed target        [stream c]
//   k_shor_vkeys       V1, V2, V3 = averages of their copies (CSR key -> members)                                         [stream c]
//   k_global           rows, Y, small-cone duals (base kernel; Shor flag: third copy of Y = the big cone's)               [main]
//   k_shor_cols        one WORKGROUP per (slot, column): paraboloid of the column, X / W / Theta_jj with the column coupling, the duals
//                      of the big cone and its next input, residual partial sums                                          [main]
//   k_shor_minor_post  duals of the order-5 blocks, their residuals                                                       [main]
//   k_shor_reduce      per-slot sums in a fixed order (run-to-run identical)                                              [main]
// Memory layout: matrices column-major fp64; per-minor arrays are SoA [entry][minor] so that a wave touches 64 consecutive doubles per entry.
// HBM-bound integer/fp64 streaming work: no MFMA here (the order-(n+m) eigen-kernel of the base engine carries the matrix-core work).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "omc_shor_relax.h"
#include "omc_wave.h"

#define SH_T 256

// packed lower triangle of the order-5 block, index of (r, c), r >= c:  r (r + 1) / 2 + c
//   0:(0,0) | 1:(1,0) 2:(1,1) | 3:(2,0) 4:(2,1) 5:(2,2) | 6:(3,0) 7:(3,1) 8:(3,2) 9:(3,3) | 10:(4,0) 11:(4,1) 12:(4,2) 13:(4,3) 14:(4,4)
// rows: 0 = the constant 1, 1 = (i1,j1), 2 = (i1,j2), 3 = (i2,j1), 4 = (i2,j2)   (OMC.jl:1772-1776)
__device__ __forceinline__ void gather15(const ShorGroupDev& G, int q, int n, const double* X, const double* W, const double* V1,
                                         const double* V2, const double* V3, double* M) {
  const int nq = G.nq;
  const int i1 = G.mi[q], i2 = G.mi[nq + q], j1 = G.mi[2 * nq + q], j2 = G.mi[3 * nq + q];
  const int e1 = j1 * n + i1, e2 = j2 * n + i1, e3 = j1 * n + i2, e4 = j2 * n + i2;
  M[0] = 1.0;
  M[1] = X[e1]; M[3] = X[e2]; M[6] = X[e3]; M[10] = X[e4];
  M[2] = W[e1]; M[5] = W[e2]; M[9] = W[e3]; M[14] = W[e4];
  M[4] = V1[G.kid[q]];               // (2,1): V1[i1,(j1,j2)]
  M[13] = V1[G.kid[nq + q]];         // (4,3): V1[i2,(j1,j2)]
  M[7] = V2[G.kid[2 * nq + q]];      // (3,1): V2[(i1,i2),j1]
  M[12] = V2[G.kid[3 * nq + q]];     // (4,2): V2[(i1,i2),j2]
  const double v3 = V3[q];
  M[11] = v3; M[8] = v3;             // (4,1) and (3,2): V3[(i1,i2),(j1,j2)] twice -- the shared entry that encodes the vanishing minor
}

// P = P_+(A) for a symmetric order-5 matrix given by its packed lower triangle: cyclic Jacobi entirely in registers (every index is a
// compile-time constant after unrolling), then the positive part rebuilt from the eigenpairs.
__device__ __forceinline__ void psd5(const double* in, double* P) {
  double a[5][5], v[5][5];
#pragma unroll
  for (int r = 0; r < 5; ++r)
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      a[r][c] = in[(r >= c) ? (r * (r + 1) / 2 + c) : (c * (c + 1) / 2 + r)];
      v[r][c] = (r == c) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 16; ++sweep) {
    double off = 0.0, dg = 0.0;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      dg += a[r][r] * a[r][r];
#pragma unroll
      for (int c = 0; c < r; ++c) off += a[r][c] * a[r][c];
    }
    if (off <= 1e-32 * dg || off == 0.0) break;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int q = p + 1; q < 5; ++q) {
        const double apq = a[p][q];
        if (fabs(apq) > 1e-300) {
          const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
          const double at = fabs(theta);
          double t = 1.0 / (at + sqrt(at * at + 1.0));
          t = (theta >= 0.0) ? t : -t;
          const double c = rsqrt(t * t + 1.0), s = t * c;
          a[p][p] -= t * apq; a[q][q] += t * apq; a[p][q] = 0.0; a[q][p] = 0.0;
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            if (k != p && k != q) {
              const double akp = a[k][p], akq = a[k][q];
              const double np_ = c * akp - s * akq, nq_ = s * akp + c * akq;
              a[k][p] = np_; a[p][k] = np_; a[k][q] = nq_; a[q][k] = nq_;
            }
            const double vkp = v[k][p], vkq = v[k][q];
            v[k][p] = c * vkp - s * vkq; v[k][q] = s * vkp + c * vkq;
          }
        }
      }
  }
  double lp[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) lp[i] = fmax(a[i][i], 0.0);
#pragma unroll
  for (int r = 0; r < 5; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < 5; ++i) acc += lp[i] * v[r][i] * v[c][i];
      P[r * (r + 1) / 2 + c] = acc;
    }
}

__device__ __forceinline__ double fro_weight(int e) { return (e == 0 || e == 2 || e == 5 || e == 9 || e == 14) ? 1.0 : 2.0; }

// ---------------------------------------------------------------------------------------------------------------------------------
// initial Shor state of the slots that received a new node (runs BEFORE the base k_setup, which clears the init flag)
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SH_T) k_shor_setup(ShWS w, double y0) {
  const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  if (!w.init[b]) return;
  const int n = w.n, m = w.m, N = w.N, NP = w.NPb;
  const size_t nm = (size_t)n * m;
  for (size_t e = tid; e < nm; e += T) { w.X[b * nm + e] = 0.0; w.W[b * nm + e] = 0.0; w.D5x[b * nm + e] = 0.0; w.P5x[b * nm + e] = 0.0; }
  for (size_t e = tid; e < (size_t)m * m; e += T) w.Th[(size_t)b * m * m + e] = 0.0;
  for (int e = tid; e < m; e += T) { w.D5t[(size_t)b * m + e] = 0.0; w.nu5[(size_t)b * m + e] = 0.0; }
  for (int e = tid; e < w.nv1max; e += T) w.V1[(size_t)b * w.nv1max + e] = 0.0;
  for (int e = tid; e < w.nv2max; e += T) w.V2[(size_t)b * w.nv2max + e] = 0.0;
  for (int e = tid; e < w.nqmax; e += T) w.V3[(size_t)b * w.nqmax + e] = 0.0;
  for (size_t e = tid; e < (size_t)15 * w.nqmax; e += T) { w.Tq[(size_t)b * 15 * w.nqmax + e] = 0.0; w.Pq[(size_t)b * 15 * w.nqmax + e] = 0.0; w.Nq[(size_t)b * 15 * w.nqmax + e] = 0.0; }
  for (size_t e = tid; e < (size_t)N * N; e += T) { w.D0[(size_t)b * N * N + e] = 0.0; w.P0[(size_t)b * N * N + e] = 0.0; }
  for (size_t e = tid; e < (size_t)NP * NP; e += T) {
    const int i = (int)(e % NP), j = (int)(e / NP);
    w.MbufB[(size_t)b * NP * NP + e] = (i == j && i < n) ? y0 : 0.0;
    w.VrowB[(size_t)b * NP * NP + e] = 0.0;
  }
  if (tid == 0) {
    w.fro2B[b] = y0 * y0 * n; w.trB[b] = y0 * n; w.vvalidB[b] = 0;
    if (w.sub_onB) { w.sub_onB[b] = 0; w.cone_doneB[b] = 0; w.sub_waitB[b] = 0; w.sub_nfailB[b] = 0; }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// order-5 blocks: projection.  Tq holds the scaled dual D_q on entry and the over-relaxed target rx P + (1 - rx) M + D on exit.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SH_T) k_shor_minor_pre(ShWS w) {
  const int b = blockIdx.y;
  if (w.done[b]) return;
  const int nb = w.node_of[b];
  const ShorGroupDev G = w.groups[w.node_group[nb]];
  const int q = blockIdx.x * SH_T + threadIdx.x;
  if (q >= G.nq) return;
  const int n = w.n; const size_t nm = (size_t)n * w.m;
  double M[15], In[15], P[15];
  gather15(G, q, n, w.X + b * nm, w.W + b * nm, w.V1 + (size_t)b * w.nv1max, w.V2 + (size_t)b * w.nv2max, w.V3 + (size_t)b * w.nqmax, M);
  double* Tq = w.Tq + (size_t)b * 15 * w.nqmax + q;
  double* Pq = w.Pq + (size_t)b * 15 * w.nqmax + q;
  double* Nq = w.Nq + (size_t)b * 15 * w.nqmax + q;
  double D[15];
#pragma unroll
  for (int e = 0; e < 15; ++e) { D[e] = Tq[(size_t)e * w.nqmax]; In[e] = M[e] - D[e]; }
  psd5(In, P);
  const double rx = w.rx;
#pragma unroll
  for (int e = 0; e < 15; ++e) {
    Tq[(size_t)e * w.nqmax] = rx * P[e] + (1.0 - rx) * M[e] + D[e];
    Pq[(size_t)e * w.nqmax] = P[e];
    Nq[(size_t)e * w.nqmax] = P[e] - In[e];        // >= 0: the multiplier direction (certificate)
  }
}

// V1, V2 = average of the copies over the blocks that share the key; V3 = average of its two positions
__global__ void __launch_bounds__(SH_T) k_shor_vkeys(ShWS w) {
  const int b = blockIdx.y;
  if (w.done[b]) return;
  const int nb = w.node_of[b];
  const ShorGroupDev G = w.groups[w.node_group[nb]];
  const int t = blockIdx.x * SH_T + threadIdx.x;
  const double* Tq = w.Tq + (size_t)b * 15 * w.nqmax;
  if (t < G.nv1) {
    double s = 0.0; const int p0 = G.v1ptr[t], p1 = G.v1ptr[t + 1];
    for (int p = p0; p < p1; ++p) { const int ent = G.v1ent[p]; s += Tq[(size_t)((ent & 1) ? 13 : 4) * w.nqmax + (ent >> 1)]; }
    w.V1[(size_t)b * w.nv1max + t] = s / (double)(p1 - p0);
  }
  if (t < G.nv2) {
    double s = 0.0; const int p0 = G.v2ptr[t], p1 = G.v2ptr[t + 1];
    for (int p = p0; p < p1; ++p) { const int ent = G.v2ent[p]; s += Tq[(size_t)((ent & 1) ? 12 : 7) * w.nqmax + (ent >> 1)]; }
    w.V2[(size_t)b * w.nv2max + t] = s / (double)(p1 - p0);
  }
  if (t < G.nq) w.V3[(size_t)b * w.nqmax + t] = 0.5 * (Tq[(size_t)11 * w.nqmax + t] + Tq[(size_t)8 * w.nqmax + t]);
}

// duals of the order-5 blocks D = T - M(new point) and their residual || P - M(new point) ||_F^2
__global__ void __launch_bounds__(SH_T) k_shor_minor_post(ShWS w) {
  __shared__ double red[32];
  const int b = blockIdx.y;
  if (w.done[b]) return;
  const int nb = w.node_of[b];
  const ShorGroupDev G = w.groups[w.node_group[nb]];
  const int q = blockIdx.x * SH_T + threadIdx.x;
  const int n = w.n; const size_t nm = (size_t)n * w.m;
  double r2 = 0.0;
  if (q < G.nq) {
    double M[15];
    gather15(G, q, n, w.X + b * nm, w.W + b * nm, w.V1 + (size_t)b * w.nv1max, w.V2 + (size_t)b * w.nv2max, w.V3 + (size_t)b * w.nqmax, M);
    double* Tq = w.Tq + (size_t)b * 15 * w.nqmax + q;
    const double* Pq = w.Pq + (size_t)b * 15 * w.nqmax + q;
#pragma unroll
    for (int e = 0; e < 15; ++e) {
      const double t = Tq[(size_t)e * w.nqmax], p = Pq[(size_t)e * w.nqmax];
      Tq[(size_t)e * w.nqmax] = t - M[e];
      r2 += fro_weight(e) * (p - M[e]) * (p - M[e]);
    }
  }
  r2 = block_sum(r2, red);
  if (threadIdx.x == 0) w.minpart[(size_t)b * w.nmb + blockIdx.x] = r2;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_shor_cols: one workgroup per (slot, column j).  Everything of the global step that is not Y / U:
//   paraboloid block of the column:  (x_S, t) -> projection on {t >= ||x_S||^2},  t = Theta_jj - sum_{i in C_j} W_ij
//   X[:, j]   = (2 T0x + r5 T5x + 2 r4 sum_members T[0,p] - cX / rho) / (2 + 2 r4 cnt + r5 [S] + qX / rho)
//   Theta_jj, W[C_j, j]: weighted averages shifted by the objective, coupled through the paraboloid copy of t (types 0 / 1) or through
//             the equality Theta_jj = sum_C W (type 2): one scalar per column in closed form
//   Theta[:, j] off the diagonal = the big cone's target (nothing else holds a copy)
//   D0 = T0 - G_new on column n + j (and its transposed positions in the X block, and column j of the Y block for j < n), next input
//   of the big cone G_new - D0, paraboloid duals, residual partial sums (fixed order: no atomics)
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SH_T) k_shor_cols(ShWS w) {
  __shared__ double red[32];
  __shared__ double s_nu;
  const int b = blockIdx.y, j = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  if (w.done[b]) return;
  const int nb = w.node_of[b];
  const ShorGroupDev G = w.groups[w.node_group[nb]];
  const int n = w.n, m = w.m, N = w.N, NP = w.NPb;
  const size_t nm = (size_t)n * m;
  const double rho = w.rho_b[b], rx = w.rx, r4 = G.r4, r5 = w.r5;
  double* X = w.X + b * nm + (size_t)j * n;
  double* W = w.W + b * nm + (size_t)j * n;
  double* Th = w.Th + (size_t)b * m * m + (size_t)j * m;
  double* D5x = w.D5x + b * nm + (size_t)j * n;
  double* P5x = w.P5x + b * nm + (size_t)j * n;
  const uint8_t* ecl = G.eclass + (size_t)j * n;
  const uint8_t* msk = w.mask + (size_t)j * n;
  const double* Ah = w.Ah + (size_t)j * n;
  const int* cptr = G.cptr + (size_t)j * n;
  const double* P0 = w.P0 + (size_t)b * N * N;
  double* D0 = w.D0 + (size_t)b * N * N;
  double* Mb = w.MbufB + (size_t)b * NP * NP;
  const double* Tq = w.Tq + (size_t)b * 15 * w.nqmax;
  const int ct = G.ctype[j];
  const int cj = n + j;                                   // column of the big matrix
  // ---- 1. paraboloid ---------------------------------------------------------------------------------------------------------
  double s = 0.0, swc = 0.0;
  for (int i = tid; i < n; i += T) {
    const int cl = ecl[i];
    if (cl == 1) { const double x = X[i] - D5x[i]; s += x * x; }
    else if (cl == 2) swc += W[i];
  }
  s = block_sum(s, red);
  swc = block_sum(swc, red);
  const double th_old = Th[j];
  const double tcur = th_old - swc;
  const double d5t = w.D5t[(size_t)b * m + j];
  double nu = 0.0, p5t = 0.0, t5t = 0.0;
  if (ct != 2) {
    const double that = tcur - d5t;
    if (tid == 0) {
      double v = 0.0;
      if (that < s) {
        // f(nu) = s / (1 + 2 nu)^2 - that - nu is convex and decreasing: Newton from the left never overshoots
        v = fmax(0.0, -that);
        for (int it = 0; it < 100; ++it) {
          const double d = 1.0 + 2.0 * v, f = s / (d * d) - that - v, fp = -4.0 * s / (d * d * d) - 1.0;
          const double vn = v - f / fp;
          if (!(vn > v) || vn - v <= 1e-16 * fmax(1.0, vn)) { v = fmax(vn, v); break; }
          v = vn;
        }
      }
      s_nu = v;
    }
    __syncthreads();
    nu = s_nu;
    p5t = that + nu;
    t5t = rx * p5t + (1.0 - rx) * tcur + d5t;
  }
  // ---- 2. X column, W on C: per-entry targets; column sums for the coupling ------------------------------------------------------
  const double cT = 1.0 / (2.0 * w.gamma) + ((ct == 1) ? 0.5 : 0.0);
  double sum_wbar = 0.0, sum_iw = 0.0;
  double fro2 = 0.0, rp2 = 0.0, rd2 = 0.0, tr1 = 0.0;
  const double inv12nu = 1.0 / (1.0 + 2.0 * nu);
  for (int i = tid; i < n; i += T) {
    const int cl = ecl[i];
    const double xo = X[i];
    const size_t a0 = (size_t)cj * N + i;                  // entry (i, n + j) of the big matrix
    const double p0 = P0[a0];
    const double t0 = rx * p0 + (1.0 - rx) * xo + D0[a0];
    double tx = 2.0 * t0, wx = 2.0, t5x = 0.0, p5x = 0.0;
    if (cl == 1 && ct != 2) {
      p5x = (xo - D5x[i]) * inv12nu;
      t5x = rx * p5x + (1.0 - rx) * xo + D5x[i];
      tx += r5 * t5x; wx += r5;
    }
    double tw = 0.0; int cnt = 0;
    if (cl == 2) {
      const int c0 = cptr[i], c1 = cptr[i + 1];
      cnt = c1 - c0;
      double sx = 0.0;
      for (int c = c0; c < c1; ++c) {
        const int ent = G.cent[c], q = ent >> 2, p = ent & 3;
        const int ex = (p == 0) ? 1 : (p == 1) ? 3 : (p == 2) ? 6 : 10;
        const int ew = (p == 0) ? 2 : (p == 1) ? 5 : (p == 2) ? 9 : 14;
        sx += Tq[(size_t)ex * w.nqmax + q];
        tw += Tq[(size_t)ew * w.nqmax + q];
      }
      tx += 2.0 * r4 * sx; wx += 2.0 * r4 * cnt;
    }
    const bool ob = msk[i] != 0;
    const double cX = ob ? -Ah[i] : 0.0;
    const double qX = (ob && cl == 1 && ct == 0) ? 1.0 : 0.0;
    const double xn = (tx - cX / rho) / (wx + qX / rho);
    // duals and the next cone input of the X block (both symmetric positions)
    const double d0n = t0 - xn;
    D0[a0] = d0n; D0[(size_t)i * N + cj] = d0n;
    const double mv = xn - d0n;
    Mb[(size_t)cj * NP + i] = mv; Mb[(size_t)i * NP + cj] = mv;
    fro2 += 2.0 * mv * mv;
    rp2 += 2.0 * (p0 - xn) * (p0 - xn);
    rd2 += 2.0 * (xn - xo) * (xn - xo);
    if (cl == 1 && ct != 2) { D5x[i] = t5x - xn; P5x[i] = p5x; rp2 += (p5x - xn) * (p5x - xn); }
    X[i] = xn;
    if (cl == 2) {
      const double ww = r4 * cnt;
      const double cW = (ob ? 0.5 : 0.0) - ((ct == 1) ? 0.5 : 0.0);
      const double wbar = tw * r4 / ww - cW / (rho * ww);
      sum_wbar += wbar; sum_iw += 1.0 / ww;
      P5x[i] = wbar;                                       // parked until the coupling scalar is known (P5x is unused on C)
    }
  }
  sum_wbar = block_sum(sum_wbar, red);
  sum_iw = block_sum(sum_iw, red);
  // ---- 3. Theta_jj and the coupling ------------------------------------------------------------------------------------------
  const size_t ajj = (size_t)cj * N + cj;
  const double p0jj = P0[ajj];
  const double t0jj = rx * p0jj + (1.0 - rx) * th_old + D0[ajj];
  const double thbar = t0jj - cT / rho;
  const double base = thbar - sum_wbar, siw = 1.0 + sum_iw;
  const double cpl = (ct != 2) ? r5 * (base - t5t) / (1.0 + r5 * siw) : base / siw;
  const double thn = thbar - cpl;
  double swn = 0.0;
  for (int i = tid; i < n; i += T) {
    if (ecl[i] == 2) {
      const int cnt = cptr[i + 1] - cptr[i];
      const double wn = P5x[i] + cpl / (r4 * cnt);
      const double wo = W[i];
      rd2 += (wn - wo) * (wn - wo);
      W[i] = wn; swn += wn;
      P5x[i] = 0.0;
    }
  }
  swn = block_sum(swn, red);
  // ---- 4. Theta column (off-diagonal: the big cone's own target; D0 stays exactly symmetric, so no transposed reads) -------------
  for (int jp = tid; jp < m; jp += T) {
    const size_t a1 = (size_t)cj * N + n + jp;
    const double p0 = P0[a1], tho = Th[jp];
    const double t0 = (jp == j) ? t0jj : rx * p0 + (1.0 - rx) * tho + D0[a1];
    const double tn = (jp == j) ? thn : t0;
    const double d0n = t0 - tn;
    D0[a1] = d0n;
    const double mv = tn - d0n;
    Mb[(size_t)cj * NP + n + jp] = mv;
    fro2 += mv * mv;
    if (jp == j) tr1 += mv;
    rp2 += (p0 - tn) * (p0 - tn);
    rd2 += (tn - tho) * (tn - tho);
    Th[jp] = tn;
  }
  // ---- 5. Y block, column j (j < n): dual of the big cone's copy of Y (Y was updated by k_global; Yp holds the previous iterate) ----
  if (j < n) {
    const double* Yn = w.Y + (size_t)b * n * n + (size_t)j * n;
    const double* Yo = w.Yp + (size_t)b * n * n + (size_t)j * n;
    for (int i = tid; i < n; i += T) {
      const size_t a2 = (size_t)j * N + i;
      const double p0 = P0[a2], yn = Yn[i];
      const double t0 = rx * p0 + (1.0 - rx) * Yo[i] + D0[a2];
      const double d0n = t0 - yn;
      D0[a2] = d0n;
      const double mv = yn - d0n;
      Mb[(size_t)j * NP + i] = mv;
      fro2 += mv * mv;
      if (i == j) tr1 += mv;
      rp2 += (p0 - yn) * (p0 - yn);
    }
  }
  fro2 = block_sum(fro2, red);
  rp2 = block_sum(rp2, red);
  rd2 = block_sum(rd2, red);
  tr1 = block_sum(tr1, red);
  if (tid == 0) {
    if (ct != 2) {
      const double tnew = thn - swn;
      w.D5t[(size_t)b * m + j] = t5t - tnew;
      w.nu5[(size_t)b * m + j] = nu;
      rp2 += (p5t - tnew) * (p5t - tnew);
    }
    double* cp = w.colpart + ((size_t)b * m + j) * 4;
    cp[0] = fro2; cp[1] = rp2; cp[2] = rd2; cp[3] = tr1;
  }
}

// per-slot sums in a fixed order: Frobenius norm of the next big-cone input, primal / dual residuals (added to the base kernel's)
__global__ void __launch_bounds__(SH_T) k_shor_reduce(ShWS w) {
  __shared__ double red[32];
  const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  if (w.done[b]) return;
  double f = 0.0, p = 0.0, d = 0.0, tr = 0.0;
  for (int j = tid; j < w.m; j += T) { const double* cp = w.colpart + ((size_t)b * w.m + j) * 4; f += cp[0]; p += cp[1]; d += cp[2]; tr += cp[3]; }
  const int nb = w.node_of[b];
  const int nblk = (w.groups[w.node_group[nb]].nq + SH_T - 1) / SH_T;
  for (int t = tid; t < nblk; t += T) p += w.minpart[(size_t)b * w.nmb + t];
  f = block_sum(f, red); p = block_sum(p, red); d = block_sum(d, red); tr = block_sum(tr, red);
  if (tid == 0) {
    w.fro2B[b] = f; w.trB[b] = tr;
    if (w.cone_doneB) w.cone_doneB[b] = 0;
    const double rp0 = w.rp[b], rd0 = w.rd[b];
    w.rp[b] = sqrt(rp0 * rp0 + p);
    w.rd[b] = sqrt(rd0 * rd0 + d);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// certificate (every check_every iterations): inputs of the base kernel k_check_build
//   objcol[j]: column j of the primal value;  c0col[j]: column j of the constants of the Lagrangian bound;
//   lamDX[:, j] = phi_j sqrt(2 / (gamma mu_j)), so that the base kernel's -gamma/2 Lx Lx' is -sum_j phi_j phi_j' / mu_j.
// The multipliers (oracle/omc_oracle_shor.py: shor_dual_bound): Gamma_q = rho r4 (P - input) on the order-5 blocks, made to cancel exactly in
// V1, V2, V3 (each block shifted by the Frobenius norm of its correction, so it stays PSD); zeta_j on the paraboloid (raised to what
// the W-stationarity on C needs); mu_j = cT_j - zeta_j; phi from the X-stationarity.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SH_T) k_shor_chk_keys(ShWS w) {
  const int b = blockIdx.y;
  if (w.done[b]) return;
  const int nb = w.node_of[b];
  const ShorGroupDev G = w.groups[w.node_group[nb]];
  const int t = blockIdx.x * SH_T + threadIdx.x;
  const double* Nq = w.Nq + (size_t)b * 15 * w.nqmax;
  const double sc = w.rho_b[b] * G.r4;
  if (t < G.nv1) {
    double s = 0.0; const int p0 = G.v1ptr[t], p1 = G.v1ptr[t + 1];
    for (int p = p0; p < p1; ++p) { const int ent = G.v1ent[p]; s += Nq[(size_t)((ent & 1) ? 13 : 4) * w.nqmax + (ent >> 1)]; }
    w.e1[(size_t)b * w.nv1max + t] = sc * s / (double)(p1 - p0);
  }
  if (t < G.nv2) {
    double s = 0.0; const int p0 = G.v2ptr[t], p1 = G.v2ptr[t + 1];
    for (int p = p0; p < p1; ++p) { const int ent = G.v2ent[p]; s += Nq[(size_t)((ent & 1) ? 12 : 7) * w.nqmax + (ent >> 1)]; }
    w.e2[(size_t)b * w.nv2max + t] = sc * s / (double)(p1 - p0);
  }
}

// per minor: Gamma' = Gamma - E + ||E||_F I ; kept in place in Nq: entry 0 = Gamma'[0,0], entries 1,3,6,10 = Gamma'[0,p], entries 2,5,9,14 = Gamma'[p,p]
__global__ void __launch_bounds__(SH_T) k_shor_chk_minor(ShWS w) {
  __shared__ double red[32];
  const int b = blockIdx.y;
  if (w.done[b]) return;
  const int nb = w.node_of[b];
  const ShorGroupDev G = w.groups[w.node_group[nb]];
  const int q = blockIdx.x * SH_T + threadIdx.x;
  double g00 = 0.0;
  if (q < G.nq) {
    const int nq = G.nq;
    double* Nq = w.Nq + (size_t)b * 15 * w.nqmax + q;
    const double sc = w.rho_b[b] * G.r4;
    const double e12 = w.e1[(size_t)b * w.nv1max + G.kid[q]], e34 = w.e1[(size_t)b * w.nv1max + G.kid[nq + q]];
    const double e13 = w.e2[(size_t)b * w.nv2max + G.kid[2 * nq + q]], e24 = w.e2[(size_t)b * w.nv2max + G.kid[3 * nq + q]];
    const double e3 = 0.5 * sc * (Nq[(size_t)11 * w.nqmax] + Nq[(size_t)8 * w.nqmax]);
    const double shift = sqrt(2.0 * (e12 * e12 + e34 * e34 + e13 * e13 + e24 * e24 + 2.0 * e3 * e3));
    g00 = sc * Nq[0] + shift;
    Nq[0] = g00;
    Nq[(size_t)1 * w.nqmax] *= sc; Nq[(size_t)3 * w.nqmax] *= sc; Nq[(size_t)6 * w.nqmax] *= sc; Nq[(size_t)10 * w.nqmax] *= sc;
    Nq[(size_t)2 * w.nqmax] = sc * Nq[(size_t)2 * w.nqmax] + shift; Nq[(size_t)5 * w.nqmax] = sc * Nq[(size_t)5 * w.nqmax] + shift;
    Nq[(size_t)9 * w.nqmax] = sc * Nq[(size_t)9 * w.nqmax] + shift; Nq[(size_t)14 * w.nqmax] = sc * Nq[(size_t)14 * w.nqmax] + shift;
  }
  g00 = block_sum(g00, red);
  if (threadIdx.x == 0) w.minpart2[(size_t)b * w.nmb + blockIdx.x] = g00;
}

__global__ void __launch_bounds__(SH_T) k_shor_chk_cols(ShWS w) {
  __shared__ double red[32];
  __shared__ double s_red2[8];
  const int b = blockIdx.y, j = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  if (w.done[b]) return;
  const int nb = w.node_of[b];
  const ShorGroupDev G = w.groups[w.node_group[nb]];
  const int n = w.n, m = w.m;
  const size_t nm = (size_t)n * m;
  const double rho = w.rho_b[b];
  const double* X = w.X + b * nm + (size_t)j * n;
  const double* W = w.W + b * nm + (size_t)j * n;
  const double* P5x = w.P5x + b * nm + (size_t)j * n;
  const uint8_t* ecl = G.eclass + (size_t)j * n;
  const uint8_t* msk = w.mask + (size_t)j * n;
  const double* Ah = w.Ah + (size_t)j * n;
  const int* cptr = G.cptr + (size_t)j * n;
  const double* Nq = w.Nq + (size_t)b * 15 * w.nqmax;
  const int ct = G.ctype[j];
  const double cT = 1.0 / (2.0 * w.gamma) + ((ct == 1) ? 0.5 : 0.0);
  // pass 1: zmin = max over C of (sum_q Gamma'[p,p] - cW), primal column value, sum of xi^2
  double zmin = -1e300, objc = 0.0, sxi2 = 0.0, cst = 0.0;
  for (int i = tid; i < n; i += T) {
    const int cl = ecl[i];
    const bool ob = msk[i] != 0;
    const double x = X[i];
    const double a = ob ? Ah[i] : 0.0;
    const double qX = (ob && cl == 1 && ct == 0) ? 1.0 : 0.0;
    objc += 0.5 * a * a - a * x + 0.5 * qX * x * x;
    cst += 0.5 * a * a - 0.5 * qX * x * x;
    if (cl == 2) {
      const double cW = (ob ? 0.5 : 0.0) - ((ct == 1) ? 0.5 : 0.0);
      objc += cW * W[i];
      double gd = 0.0;
      for (int c = cptr[i]; c < cptr[i + 1]; ++c) {
        const int ent = G.cent[c], q = ent >> 2, p = ent & 3;
        gd += Nq[(size_t)((p == 0) ? 2 : (p == 1) ? 5 : (p == 2) ? 9 : 14) * w.nqmax + q];
      }
      zmin = fmax(zmin, gd - cW);
    } else if (cl == 1 && ct != 2) {
      sxi2 += P5x[i] * P5x[i];
    }
  }
  // block max of zmin
  {
    double v = zmin;
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, WAVE));
    __syncthreads();
    if ((tid & 63) == 0) s_red2[tid >> 6] = v;
    __syncthreads();
    v = s_red2[0];
    for (int q = 1; q < (T >> 6); ++q) v = fmax(v, s_red2[q]);
    zmin = v;
  }
  objc = block_sum(objc, red);
  sxi2 = block_sum(sxi2, red);
  cst = block_sum(cst, red);
  const double zadm = fmax(rho * w.r5 * w.nu5[(size_t)b * m + j], 0.0);
  const double zeta = (ct == 2) ? zmin : fmax(zadm, zmin);
  double mu = cT - zeta;
  // pass 2: phi and the dense multiplier column
  double* Lx = w.lamDX + (size_t)b * nm + (size_t)j * n;
  const bool badmu = !(mu > 0.0);
  const double scl = badmu ? 0.0 : sqrt(2.0 / (w.gamma * mu));
  double anyphi = 0.0;
  for (int i = tid; i < n; i += T) {
    const int cl = ecl[i];
    const bool ob = msk[i] != 0;
    const double x = X[i];
    const double cX = ob ? -Ah[i] : 0.0;
    const double qX = (ob && cl == 1 && ct == 0) ? 1.0 : 0.0;
    double phi = 0.5 * (cX + qX * x);
    if (cl == 1 && ct != 2) phi += zeta * P5x[i];
    if (cl == 2) {
      double g0 = 0.0;
      for (int c = cptr[i]; c < cptr[i + 1]; ++c) {
        const int ent = G.cent[c], q = ent >> 2, p = ent & 3;
        g0 += Nq[(size_t)((p == 0) ? 1 : (p == 1) ? 3 : (p == 2) ? 6 : 10) * w.nqmax + q];
      }
      phi -= g0;
    }
    anyphi += fabs(phi);
    Lx[i] = phi * scl;
  }
  anyphi = block_sum(anyphi, red);
  if (tid == 0) {
    double c0 = cst - ((ct != 2) ? zeta * sxi2 : 0.0);
    if (badmu && anyphi > 0.0) c0 = -1e300;               // no valid multiplier for this column at this check: the bound is -inf
    if (j == 0) {                                          // constants of the order-5 blocks, added in a fixed order
      const int nblk = (G.nq + SH_T - 1) / SH_T;
      double g = 0.0;
      for (int t = 0; t < nblk; ++t) g += w.minpart2[(size_t)b * w.nmb + t];
      c0 -= g;
    }
    w.c0col[(size_t)b * m + j] = c0;
    w.objcol[(size_t)b * m + j] = objc + cT * w.Th[(size_t)b * m * m + (size_t)j * m + j];
  }
}

// penalty bump (k_check_final decided bfac): scaled duals follow the penalty, the big cone's input is rebuilt.  Runs BEFORE the base
// k_rho_rescale (which resets bfac).  Tq holds the duals of the order-5 blocks at this point (k_shor_minor_post).
__global__ void __launch_bounds__(SH_T) k_shor_rescale(ShWS w) {
  __shared__ double red[32];
  const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  if (w.done[b]) return;
  const double f = w.bfac[b];
  if (f == 1.0) return;
  const double inv = 1.0 / f;
  const int n = w.n, m = w.m, N = w.N, NP = w.NPb;
  const size_t nm = (size_t)n * m;
  for (size_t e = tid; e < (size_t)15 * w.nqmax; e += T) w.Tq[(size_t)b * 15 * w.nqmax + e] *= inv;
  for (size_t e = tid; e < nm; e += T) w.D5x[b * nm + e] *= inv;
  for (int e = tid; e < m; e += T) w.D5t[(size_t)b * m + e] *= inv;
  double fr2 = 0.0, tr1 = 0.0;
  for (size_t e = tid; e < (size_t)N * N; e += T) {
    const int i = (int)(e % N), jj = (int)(e / N);
    const double d = w.D0[(size_t)b * N * N + e] * inv;
    w.D0[(size_t)b * N * N + e] = d;
    double g;
    if (i < n && jj < n) g = w.Y[(size_t)b * n * n + (size_t)jj * n + i];
    else if (i < n) g = w.X[b * nm + (size_t)(jj - n) * n + i];
    else if (jj < n) g = w.X[b * nm + (size_t)(i - n) * n + jj];
    else g = w.Th[(size_t)b * m * m + (size_t)(jj - n) * m + (i - n)];
    const double mv = g - d;
    w.MbufB[(size_t)b * NP * NP + (size_t)jj * NP + i] = mv;
    fr2 += mv * mv;
    if (i == jj) tr1 += mv;
  }
  fr2 = block_sum(fr2, red);
  tr1 = block_sum(tr1, red);
  if (tid == 0) { w.fro2B[b] = fr2; w.trB[b] = tr1; }
}

// results of the slots flagged fin: X, Theta unscaled; the full W of the reference's program (X^2 on the SOC entries, the slack of
// Theta_jj = sum_i W_ij on the cheapest entry of the column outside the minors)
__global__ void __launch_bounds__(SH_T) k_shor_harvest(ShWS w) {
  __shared__ double red[32];
  const int b = blockIdx.y, j = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  if (!w.fin[b]) return;
  const int nb = w.node_of[b];
  const ShorGroupDev G = w.groups[w.node_group[nb]];
  const int n = w.n, m = w.m;
  const size_t nm = (size_t)n * m;
  const double isc = 1.0 / w.sc, is2 = isc * isc;
  const uint8_t* ecl = G.eclass + (size_t)j * n;
  double sw = 0.0;
  for (int i = tid; i < n; i += T) {
    const double x = w.X[b * nm + (size_t)j * n + i] * isc;
    w.oX[(size_t)nb * nm + (size_t)j * n + i] = x;
    const int cl = ecl[i];
    const double wv = (cl == 2) ? w.W[b * nm + (size_t)j * n + i] * is2 : (cl == 1) ? x * x : 0.0;
    w.oW[(size_t)nb * nm + (size_t)j * n + i] = wv;
    sw += wv;
  }
  for (int jp = tid; jp < m; jp += T) w.oTh[(size_t)nb * m * m + (size_t)j * m + jp] = w.Th[(size_t)b * m * m + (size_t)j * m + jp] * is2;
  if (w.oV) {      // the lifted products of the blocks, unscaled, in the node's minor order
    const double* V1 = w.V1 + (size_t)b * w.nv1max; const double* V2 = w.V2 + (size_t)b * w.nv2max; const double* V3 = w.V3 + (size_t)b * w.nqmax;
    for (int q = j * T + tid; q < G.nq; q += m * T) {
      double* o = w.oV + ((size_t)nb * w.nqmax + q) * 5;
      o[0] = V1[G.kid[q]] * is2; o[1] = V1[G.kid[G.nq + q]] * is2; o[2] = V2[G.kid[2 * G.nq + q]] * is2; o[3] = V2[G.kid[3 * G.nq + q]] * is2; o[4] = V3[q] * is2;
    }
  }
  sw = block_sum(sw, red);
  if (tid == 0) {
    const int sr = G.slackrow[j];
    if (sr >= 0) w.oW[(size_t)nb * nm + (size_t)j * n + sr] += w.Th[(size_t)b * m * m + (size_t)j * m + j] * is2 - sw;
  }
}

extern "C" {
void omc_shor_launch_setup(const ShWS* w, hipStream_t s) {
  hipLaunchKernelGGL(k_shor_setup, dim3(w->S), dim3(SH_T), 0, s, *w, (double)w->k / (double)w->n);      // Y0 = (k / n) I
}
void omc_shor_launch_minor_pre(const ShWS* w, hipStream_t s) {
  if (w->nqmax > 0) hipLaunchKernelGGL(k_shor_minor_pre, dim3(w->nmb, w->S), dim3(SH_T), 0, s, *w);
}
void omc_shor_launch_vkeys(const ShWS* w, hipStream_t s) {
  const int mx = w->nv1max > w->nv2max ? (w->nv1max > w->nqmax ? w->nv1max : w->nqmax) : (w->nv2max > w->nqmax ? w->nv2max : w->nqmax);
  if (w->nqmax > 0) hipLaunchKernelGGL(k_shor_vkeys, dim3((mx + SH_T - 1) / SH_T, w->S), dim3(SH_T), 0, s, *w);
}
void omc_shor_launch_cols(const ShWS* w, hipStream_t s) { hipLaunchKernelGGL(k_shor_cols, dim3(w->m, w->S), dim3(SH_T), 0, s, *w); }
void omc_shor_launch_minor_post(const ShWS* w, hipStream_t s) {
  if (w->nqmax > 0) hipLaunchKernelGGL(k_shor_minor_post, dim3(w->nmb, w->S), dim3(SH_T), 0, s, *w);
}
void omc_shor_launch_reduce(const ShWS* w, hipStream_t s) { hipLaunchKernelGGL(k_shor_reduce, dim3(w->S), dim3(SH_T), 0, s, *w); }
void omc_shor_launch_check(const ShWS* w, hipStream_t s) {
  if (w->nqmax > 0) {
    const int mx = w->nv1max > w->nv2max ? w->nv1max : w->nv2max;
    hipLaunchKernelGGL(k_shor_chk_keys, dim3((mx + SH_T - 1) / SH_T, w->S), dim3(SH_T), 0, s, *w);
    hipLaunchKernelGGL(k_shor_chk_minor, dim3(w->nmb, w->S), dim3(SH_T), 0, s, *w);
  }
  hipLaunchKernelGGL(k_shor_chk_cols, dim3(w->m, w->S), dim3(SH_T), 0, s, *w);
}
void omc_shor_launch_rescale(const ShWS* w, hipStream_t s) { hipLaunchKernelGGL(k_shor_rescale, dim3(w->S), dim3(SH_T), 0, s, *w); }
void omc_shor_launch_harvest(const ShWS* w, hipStream_t s) { hipLaunchKernelGGL(k_shor_harvest, dim3(w->m, w->S), dim3(SH_T), 0, s, *w); }
}
