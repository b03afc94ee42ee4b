// omc_colprox.hip -- the column prox of the ADMM iteration (block F of DESIGN.md section 3.1) for columns with at most 64 observed rows:
// k_colprox_pair (two columns per wave) and k_colprox_wide (one column per wave).  A translation unit of its own because the fully unrolled
// sweep (64 steps x 64 entries for the wide form) needs -mllvm -pragma-unroll-threshold above the compiler's default: below it the outer loop is
// not unrolled, the row array is indexed dynamically and lives in private memory (measured: 2120 scratch instructions in the wide kernel).
// The reference has no counterpart (Mosek solves the node's conic program, OMC.jl:1482-1500, 1857); oracle: _prox_columns (oracle/omc_oracle.py:475-515).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include "omc_device.h"
#include "omc_wave.h"

// ---------------------------------------------------------------------------------------------------------
// k_colprox_pair: the column prox (mode 0 of k_colprox) with TWO columns per wave.  A column of config 2 has ~20 observed rows: with one
// column per wave 20 of 64 lanes worked, and the factorisation spent most of its instructions on v_readlane broadcasts (SGPR round trips)
// and its solves on a dependent chain of one broadcast per row.  Here each half-wave (32 lanes) owns one column, lane l of a half owns
// matrix row l, and the whole row lives in REGISTERS.  Instead of a factorisation and triangular solves the half computes the inverse:
//   -(B + cp s I)^-1 by the symmetric sweep operator (the Gauss-Jordan form that keeps the matrix symmetric, so that row k is available as
//   column k): at step k every lane stages its entry of column k in LDS (one ds_write), reads the pivot and the column back as
//   half-uniform (broadcast) ds_reads and updates its row with one fma per entry -- no cross-lane VALU traffic, no SGPR round trip;
//   the pivots are those of L D L', so their sign is the positive-definiteness test the secular iteration needs;
//   every solve (y = A^-1 a, z = A^-1 y, w = A^-1 z) is then one staged vector and 32 fma per lane, with no dependent chain.
// (A first form with L D L' in registers and substitutions by DPP row_newbcast / v_permlane16_swap broadcasts took 813 us per launch at
// ~950 live slots against 620 us for this one and 1111 us for k_colprox; the launch is bound by the LDS return path of the broadcast reads.)
// The secular iteration of colprox_reg is carried per half (all of its scalars are half-uniform lane values); a half that has finished
// idles while the other one goes on.  Columns with more than 32 rows and unpaired last columns stay with k_colprox (w.cp_solo).
// ---------------------------------------------------------------------------------------------------------
template <int Q>
__device__ __forceinline__ double half_bcast(double v) {      // value of lane Q (0..31) of each half-wave, in every lane of that half
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + (Q & 15), 0xF, 0xF, false);      // row_newbcast: lane Q & 15 of every row of 16
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + (Q & 15), 0xF, 0xF, false);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);                  // [0]: even rows copied into the odd ones, [1]: the reverse
  auto c = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return (Q < 16) ? __hiloint2double(c[0], a[0]) : __hiloint2double(c[1], a[1]);
}
__device__ __forceinline__ double half_sum(double v) {        // sum over the 32 lanes of each half-wave, in every lane of that half
  v = group_sum_dpp<16>(v);
  const int lo = __double2loint(v), hi = __double2hiint(v);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto c = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(c[0], a[0]) + __hiloint2double(c[1], a[1]);
}
#define CPP_LDS_DOUBLES (64 + 64 + 32)      // per wave: staged column, previous alpha, row indices (64 ints)

// The body for LPC lanes per column: 32 (two columns per wave, k_colprox_pair) or 64 (one column per wave, up to 64 observed rows: k_colprox_wide --
// config 3 has ~40 rows per column).  j, off, c are per lane (uniform inside a column's lanes), cmax the largest c of the wave (SGPR).
template <int LPC>
__device__ __forceinline__ double lanes_sum(double v) { if constexpr (LPC == 32) return half_sum(v); else return wave_sum(v); }
#define RQ(q) (((q) < 32) ? R0[(q) & 31] : R1[((NC > 32) ? (q) : 0) & 31])
template <int LPC>
__device__ __forceinline__ void colprox_sweep(const OmcWS& w, const int b, const int lane, const int j, const int off, const int c, const int cmax,
                                              double* st, double* vo_s, int* sidx) {
  constexpr int NC = LPC;
  const int h = (LPC == 32) ? (lane >> 5) : 0, l = lane & (LPC - 1), hb = h * LPC;
  const bool act = l < c;
  const int n = w.n;
  const double gm = w.gamma;
  const double* Y = w.Y + (size_t)b * n * n;
  const double* Yp = w.Yp + (size_t)b * n * n;
  const double* Yx = w.Yx ? w.Yx + (size_t)b * n * n : nullptr;      // 2 Y - Yp, one load per entry
  double* alpha = w.alpha + (size_t)b * w.nnz + off;
  const double a_reg = act ? w.col_val[off + l] : 0.0;
  const int my = act ? w.col_idx[off + l] : 0;
  const double vo_r = act ? alpha[l] : 0.0;
  sidx[lane] = my; vo_s[lane] = vo_r;
  const double rho_f = w.rho_b[b] * w.rho_f_ratio;
  const double coef = gm / (2.0 * rho_f), cp = gm * gm / (2.0 * rho_f);
  // ---- B + shift I, B = I + gamma ((2 Y - Yp)[O, O] - coef a_old a_old'): the whole row l in registers (entry (l, q) and entry (q, l) read the
  // same element of Y: the matrix is symmetric bit for bit), every load of a batch issued before its first use.  The rare second inversion
  // (3 % of the columns since the series finish) gathers the matrix again from L2: a copy kept in registers (64 VGPRs) costs a wave per SIMD
  // the row in two arrays of 32: one array of 64 doubles is not split into registers by the compiler (it stays a 512-byte private-memory object)
  double R0[32], R1[(NC > 32) ? 32 : 1], sc;
  auto gather = [&](double shift) __attribute__((always_inline)) {
#pragma unroll
    for (int qb = 0; qb < NC; qb += 8) {
      if (qb < cmax) {
        double y1[8], y2[8], vq[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int iq = sidx[hb + qb + u];
          vq[u] = vo_s[hb + qb + u];
          const size_t a = (l > qb + u) ? (size_t)iq * n + my : (size_t)my * n + iq;
          y1[u] = Yx ? Yx[a] : Y[a]; y2[u] = Yx ? 0.0 : Yp[a];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int q = qb + u;
          const double v = gm * ((Yx ? y1[u] : 2.0 * y1[u] - y2[u]) - coef * vo_r * vq[u]);
          RQ(q) = (l == q) ? (act ? v + 1.0 : 1.0) + shift : ((act && q < c) ? v : 0.0);      // rows / columns beyond the column's size: identity
        }
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) RQ(qb + u) = 0.0;
      }
    }
  };
  // -(B + shift I)^-1 by the symmetric sweep operator, both halves at once.  Lane l holds row l as sc * RQ(q) (sc = 1 until the row has been
  // the pivot row, 1 / d_l afterwards: the pivot row is never rescaled entry by entry).  Step k: every lane stages its entry of column k
  // (= row k, by symmetry) in LDS, reads pivot and row back as half-uniform ds_reads and updates its row with one fma per entry.  The pivots
  // are those of the L D L' factorisation, so their sign is the positive-definiteness test.  Returns (per half) whether every pivot was positive.
  auto invert = [&]() __attribute__((always_inline)) -> bool {
    int cm = cmax;
    asm volatile("" : "+s"(cm));      // opaque copies: the guards (k < cmax) and lane masks (l == k) are evaluated where they stand -- hoisted out
    int lv = l;                       // of the secular loop they were spilled to VGPR lanes (v_writelane / v_readlane + s_nop per use)
    asm volatile("" : "+v"(lv));
    sc = 1.0;
    double isc = 1.0;
    bool good = true;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      if (k < cm) {                                              // wave-uniform
        const double ak = RQ(k) * sc;                             // a(l, k); lane k: the pivot
        st[lane] = ak;
        const double d = st[hb + k];
        good = good && (d > 1e-290);
        const double pinv = fast_rcp(d);
        const bool piv = (lv == k);
        const double tt = piv ? 0.0 : ak * pinv * isc;           // a(l, k) / d in units of this lane's scale; the pivot row itself is not touched
#pragma unroll
        for (int qb = 0; qb < NC; qb += 4) {
          if (qb < cm) {                                         // wave-uniform; rows beyond a column's size are identity rows
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int q = qb + u;
              if (q != k) RQ(q) = fma(-tt, st[hb + q], RQ(q));
            }
          }
        }
        RQ(k) = piv ? -1.0 : tt;                                  // column k: a(l, k) / d ; pivot: -1 / d = -1 * (new scale)
        sc = piv ? pinv : sc;
        isc = piv ? d : isc;
      }
    }
    return good;
  };
  // x = (B + shift I)^-1 v: the vector is staged in LDS and every lane takes the inner product with its row (four partial sums)
  auto apply = [&](double v) __attribute__((always_inline)) -> double {
    int cm = cmax;
    asm volatile("" : "+s"(cm));
    st[lane] = v;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int qb = 0; qb < NC; qb += 4) {
      if (qb < cm) {
        a0 = fma(RQ(qb), st[hb + qb], a0); a1 = fma(RQ(qb + 1), st[hb + qb + 1], a1);
        a2 = fma(RQ(qb + 2), st[hb + qb + 2], a2); a3 = fma(RQ(qb + 3), st[hb + qb + 3], a3);
      }
    }
    return -sc * ((a0 + a1) + (a2 + a3));
  };
  // ---- secular equation || (B + cp s I)^-1 a ||^2 = s: Halley steps with the Taylor finish of colprox_reg, per half.  ONE instance of the
  // factorisation and ONE of the solve in the instruction stream (the inlined code of both is ~2.5 k instructions): the solves of a pass
  // (y, z = A^-1 y and, for the Taylor finish, w = A^-1 z) run through a loop, and the closing factorisation of a column that has used up
  // its 60 steps is pass 60 of the same loop
  const double sprev = (j < w.m) ? w.sval[(size_t)b * w.m + j] : 0.0;
  double s = (sprev > 0.0) ? sprev : 0.0;
  double lo = 0.0, hi = -1.0;
  bool lo_valid = false, fin = (c == 0);
  double yout = 0.0;
  // One inversion per ADMM iteration in the common case.  After the Halley step d from (y, z) the answer at s + d is read off the Neumann
  // series  alpha(s + d) = sum_k (-cp d)^k A^-(k+1) a  instead of a second inversion: the vectors v_k = A^-k a are one cheap product with the
  // inverse each, the moments m_p = a' A^-p a = <v_i, v_j> (i + j = p) give ||alpha_K(s + d)||^2 as a polynomial in d, and the secular
  // equation of the TRUNCATED alpha_K is solved for d by scalar Newton steps (so the returned pair satisfies ||alpha||^2 = s to round-off);
  // accepted when the first term left out is below 2e-15 of the first kept.  K = 3 (colprox_reg's second-order finish) when cp |d| ||z|| / ||y|| < 1e-5,
  // else K = 6, which carries steps up to cp |d| ||A^-1|| ~ 5e-3 -- the size s moves per ADMM iteration until late in a solve; with K = 3
  // alone the kernel averaged more than two inversions per column pair.  A step the series cannot carry is taken as before (s <- s + d, invert).
  const int ser_max = (w.cp_series > 0) ? w.cp_series : 3;
  int npass = 0; double dr_first = -1.0; int why = 0;      // diagnostics (OMC_SUB_DEBUG=4)
  for (int it = 0; it <= 60; ++it) {
    if (!__any(!fin)) break;
    const bool last = (it >= w.cp_maxpass);
    gather(cp * s);
    const bool ok_ = invert();
    if (!fin) ++npass;
    double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double rhs = a_reg, yy = 0.0, yz = 0.0, zz = 0.0;
    bool want_ser = false; double dstep = 0.0;
    int K = 2;
    for (int sv = 0; sv < K; ++sv) {
      const double x = apply(rhs);
      rhs = x;
#pragma unroll
      for (int q = 0; q < 6; ++q) v[q] = (sv == q) ? x : v[q];
      if (sv == 0 && last) { if (!fin) { yout = x; fin = true; } break; }
      if (sv == 1) {
        yy = lanes_sum<LPC>(v[0] * v[0]); yz = lanes_sum<LPC>(v[0] * v[1]); zz = lanes_sum<LPC>(v[1] * v[1]);
        bool big = false;
        if (!fin) {
          if (!ok_) {                                              // s below the positive definite range: move right
            lo = s; lo_valid = false;
            s = (hi > 0.0) ? 0.5 * (s + hi) : (2.0 * s + 1.0);
          } else {
            const double ph = yy - s, dph = -2.0 * cp * yz - 1.0, ddph = 6.0 * cp * cp * zz;
            if (ph >= 0.0) { lo = s; lo_valid = true; } else { hi = s; }
            const double den = 2.0 * dph * dph - ph * ddph;
            double sn = s + ((den > dph * dph) ? (-2.0 * ph * dph / den) : (-ph / dph));
            bool guarded = false;
            if (!(sn > lo) && !lo_valid) { sn = 0.5 * (lo + s); guarded = true; }
            if (sn < lo) { sn = lo; guarded = true; }
            if (hi > 0.0 && sn > hi) { sn = 0.5 * (lo + hi); guarded = true; }
            const double d = sn - s;
            const double dr = (yy > 0.0) ? cp * fabs(d) * sqrt(zz / yy) : 1.0;
            if (dr_first < 0.0) { dr_first = dr; why = guarded ? 1 : 0; }
            if (fabs(d) <= 1e-13 * fmax(1.0, fabs(s))) { fin = true; yout = v[0]; }      // the current solve is the answer
            else if (!guarded && dr < ((ser_max >= 6) ? 5e-3 : 1e-5)) { want_ser = true; dstep = d; big = dr >= 1e-5; }
            else s = sn;
          }
        }
        if (__any(want_ser)) K = __any(want_ser && big) ? 6 : 3;
      }
    }
    if (K > 2) {
      // moments m_2 .. m_2K of the half's column (m_2 = yy, m_3 = yz, m_4 = zz)
      double mo[13];
      mo[2] = yy; mo[3] = yz; mo[4] = zz;
#pragma unroll
      for (int i = 2; i < 6; ++i) {
        if (i < K) { mo[2 * i + 1] = lanes_sum<LPC>(v[i - 1] * v[i]); mo[2 * i + 2] = lanes_sum<LPC>(v[i] * v[i]); }
        else { mo[2 * i + 1] = 0.0; mo[2 * i + 2] = 0.0; }
      }
      if (want_ser) {
        // G(t) = || sum_{k < K} (-t)^k v_{k+1} ||^2 = sum_p cnt(p) m_p (-t)^(p-2), t = cp d ; solve G(cp d) = s + d
        double cf[11];
#pragma unroll
        for (int p = 2; p <= 12; ++p) {
          const int cnt = (p <= K + 1) ? p - 1 : ((p <= 2 * K) ? 2 * K - p + 1 : 0);
          cf[p - 2] = (double)cnt * mo[p];
        }
        double d = dstep;
        bool conv = false;
        for (int nt = 0; nt < 8; ++nt) {
          const double t = -cp * d;
          double G = 0.0, dG = 0.0;
#pragma unroll
          for (int q = 10; q >= 0; --q) { dG = fma(dG, t, G); G = fma(G, t, cf[q]); }      // Horner: G and dG / dt
          const double F = G - s - d, dF = -cp * dG - 1.0;
          const double dn = d - F / dF;
          conv = fabs(dn - d) <= 1e-14 * fmax(fabs(s), fabs(dn));
          d = dn;
          if (conv) break;
        }
        const double t = cp * d;
        const double t2 = t * t;
        const double lastterm = ((K == 6) ? t2 * t2 * fabs(t) : t2) * sqrt(fmax((K == 6) ? mo[12] : mo[6], 0.0));      // |t|^(K-1) ||v_K||
        const double growth = sqrt(fmax((K == 6) ? mo[12] : mo[6], 0.0) / fmax((K == 6) ? mo[10] : mo[4], 1e-300));      // ||v_K|| / ||v_(K-1)||: the ratios grow towards ||A^-1||
        // the truncation error is the first term left out, ~ (last term kept) x |t| x growth
        if (conv && s + d > lo * (lo_valid ? 1.0 : 0.0) && lastterm * fabs(t) * growth <= 2e-15 * sqrt(yy) && fabs(t) * growth < 0.05) {
          double acc = 0.0;
#pragma unroll
          for (int q = 5; q >= 0; --q) acc = (q < K) ? fma(acc, -t, v[q]) : acc;
          yout = acc; s = s + d; fin = true;
        } else {
          s = s + dstep;                                           // the series cannot carry this step: invert again at the Halley point
        }
      }
    }
  }
  if (w.sub_debug == 4 && l == 0 && c > 0) {      // diagnostics: passes per column (stamps[0..7]), log10 of the first relative step (stamps[8..23]: bin 8 + min(15, -log10)), guarded first steps (24), failed first inversions (25)
    atomicAdd(&w.stamps[(npass < 7) ? npass : 7], 1.0);
    if (dr_first >= 0.0) { const int bin = (dr_first > 0.0) ? (int)fmin(15.0, fmax(0.0, -log10(dr_first))) : 15; atomicAdd(&w.stamps[8 + bin], 1.0); if (why) atomicAdd(&w.stamps[24], 1.0); }
    else atomicAdd(&w.stamps[25], 1.0);
  }
  if (act) {
    alpha[l] = yout;
    w.lamD[((size_t)b * w.m + j) * n + my] = yout;               // dense copy (zeros off the support) for the output-stationary Lambda Lambda'
    if (l == 0) w.sval[(size_t)b * w.m + j] = s;
  }
}


#undef RQ
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) k_colprox_pair(OmcWS w) {
  extern __shared__ double smem[];
  const int wave_in_blk = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, wpb = blockDim.x >> 6;      // scalar: every guard below is a wave-uniform branch
  const int gw = blockIdx.x * wpb + wave_in_blk;
  const int mp = (w.m + 1) >> 1;
  const int bl = gw / mp, pr = gw - bl * mp;
  if (bl >= w.nB) return;
  const int b = slot_of(w, bl);
  if (w.done[b]) return;
  const int j0 = 2 * pr, j1 = j0 + 1;
  const int off0 = w.col_ptr[j0], c0 = w.col_ptr[j0 + 1] - off0;
  const int off1 = (j1 < w.m) ? w.col_ptr[j1] : 0, c1 = (j1 < w.m) ? w.col_ptr[j1 + 1] - off1 : 0;
  if (c0 > 32 || c1 > 32 || j1 >= w.m) return;                  // k_colprox_wide / k_colprox run these columns (w.cp_wide, w.cp_solo)
  const int cmax = __builtin_amdgcn_readfirstlane((c0 > c1) ? c0 : c1);      // wave-uniform, in an SGPR
  if (cmax == 0) return;
  double* st = smem + (size_t)wave_in_blk * CPP_LDS_DOUBLES;    // 64: staged column
  double* vo_s = st + 64;                                       // 64: alpha of the previous iteration
  int* sidx = (int*)(vo_s + 64);                                // 64: row indices
  const int h = lane >> 5;
  colprox_sweep<32>(w, b, lane, j0 + h, h ? off1 : off0, h ? c1 : c0, cmax, st, vo_s, sidx);
}

// one column per wave for the columns k_colprox_pair leaves that hold at most 64 observed rows (w.cp_wide)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) k_colprox_wide(OmcWS w) {
  extern __shared__ double smem[];
  const int wave_in_blk = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  const int gw = blockIdx.x * wpb + wave_in_blk;
  const int bl = gw / w.cp_nwide, jj = gw - bl * w.cp_nwide;
  if (bl >= w.nB) return;
  const int b = slot_of(w, bl);
  if (w.done[b]) return;
  const int j = w.cp_wide[jj];
  const int off = w.col_ptr[j], c = w.col_ptr[j + 1] - off;
  const int cmax = __builtin_amdgcn_readfirstlane(c);
  if (cmax == 0 || cmax > 64) return;
  double* st = smem + (size_t)wave_in_blk * CPP_LDS_DOUBLES;
  double* vo_s = st + 64;
  int* sidx = (int*)(vo_s + 64);
  colprox_sweep<64>(w, b, lane, j, off, c, cmax, st, vo_s, sidx);
}


extern "C" void omc_launch_colprox_sweep(const OmcWS* w, hipStream_t s) {
  const int wpb = 4;
  const int wp = w->nB * ((w->m + 1) / 2);
  hipLaunchKernelGGL(k_colprox_pair, dim3((wp + wpb - 1) / wpb), dim3(wpb * 64), (size_t)wpb * CPP_LDS_DOUBLES * sizeof(double), s, *w);
  if (w->cp_nwide > 0) {
    const int ww = w->nB * w->cp_nwide;
    hipLaunchKernelGGL(k_colprox_wide, dim3((ww + wpb - 1) / wpb), dim3(wpb * 64), (size_t)wpb * CPP_LDS_DOUBLES * sizeof(double), s, *w);
  }
}
