// omc_device.h -- workspace descriptor shared by the kernels (omc_device.hip) and the host API (omc_api.cpp)
#ifndef OMC_DEVICE_H
#define OMC_DEVICE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ROW_TRACE 0
#define ROW_BOX 1
#define ROW_BOUND 2
#define ROW_CUT 3

#define CONE_CLIP01 0
#define CONE_EVALS 2
#define CONE_SEP 3
#define CONE_TOPK 4
#define CONE_BIG 5     // Shor mode: PSD projection of the order-(n+m) matrix in Mbuf (view with n = n + m), output in W1
#define SMALL_PROJ 0
#define SMALL_RECOVER 1

#define OMC_ST_OPTIMAL 0
#define OMC_ST_SLOW 1
#define OMC_ST_TIME 2
#define OMC_ST_INFEASIBLE 3

#define NNQP_PMAX 64
#define OMC_MAX_DYN_LDS (144 * 1024)

struct OmcWS {
  // sizes
  // B = number of SLOTS (state arrays, grid size); Btot = number of nodes of the staged batch (descriptor and output arrays).
  // node_of[b] = node currently relaxed in slot b.
  int b0, nB;   // slot range [b0, b0 + nB) handled by a launch of the per-iteration kernels (the slots are split in groups that run on their own streams)
  int B, Btot, n, m, k, nnz, Rmax, Lmax, breakpoints, rmax, stall_checks, np16, max_sweeps, max_iters;
  int *node_of, *init, *fin;   // B
  const int* slot_list;        // compact list of the slots that hold a node (rebuilt by the host at every check): the per-iteration kernels launch over it
  int check_every, early_stop_after; double early_stop_factor;   // stop a node as SLOW_PROGRESS once its gap cannot close before max_iters (k_check_final)
  double *gap_prev, *gap_rate; int* slow_votes;                  // B each
  const double* rho_node;      // Btot: initial penalty of each node
  double *oY, *oU, *oalphaX, *obx, *oobj, *olb, *olmin, *orho; int *ostatus, *oiters;   // per-node outputs
  // parameters
  double gamma, rho, rho_f_ratio, relax, eps_gap, eps_feas, sumA2, jacobi_tau;   // rho: batch default (rho_b holds the per-node value)
  double* rho_b;          // B: ADMM penalty of node b (bumped on the device)
  double* bfac;           // B: rescale factor decided at the last check (1 = none)
  int *nbump, *lastbump;  // B
  int bump_max, bump_after, bump_gap; double bump_ratio, bump_factor;
  // instance (device, read-only)
  const int* col_ptr;     // m+1
  const int* col_idx;     // nnz: observed rows of each column, ascending
  const double* col_val;  // nnz: A at those
  const double* Ncnt;     // n*n: number of columns observing both rows
  const int* row_ptr;     // n+1: CSR of the observed entries (row -> observed columns)
  const int* row_idx;     // nnz
  double* lamD;           // B*m*n: dense column-major copy of Lambda (zero off the support)
  double* lamDX;          // same for the exact multipliers of the certificate (k_colprox mode 1); NULL for the small orders that keep the scattered form
  const double* wY1;      // n*n: consensus weight of Y entries for rho = 1: rho_f_ratio*Ncnt + 2
  // per-node state (node stride in comments)
  double *Y, *Yp;         // n*n
  double* U;              // n*k
  double *D1, *D3;        // n*n scaled duals of the two full-Y cone blocks
  double *W1, *E3;        // n*n: clip(Y-D1) ; Q dS Q'
  double* Qb;             // n*rmax: orthonormal basis of the row functionals of node b; rr[b] columns used
  int* rr;                // B
  double *Vt, *D3V, *W3V, *Q3V;  // rmax*k
  double *D3T, *W3T, *Q3T;       // k*k
  double* dS;             // rmax*rmax
  double *alpha, *alphaX; // nnz
  double* sval;           // m
  double *objcol, *c0col; // m: per-column terms of the exact objective / Fenchel constant (k_colprox mode 1 -> k_check_build)
  double* Mchk;           // n*n
  double *Mbuf, *Vrow;    // np16*np16: next cone input Y - D1 (zero padded) ; eigenvectors of the last projection, row-major
  double* fro2;           // B: ||Mbuf||_F^2
  int* vvalid;            // B: Vrow holds eigenvectors
  // the same triple for the certificate matrix Mchk (k_cone_ws with ws_mode = 1 returns the k smallest eigenvalues only)
  double *MbufC, *VrowC, *fro2c; int* vvalidC; int ws_mode;
  int ws_ld;              // leading dimension of G in k_cone_ws (chosen on the host: 16 mod 32 when it fits)
  // tracked top-16 subspace of the cone input (k_cone_sub): once Y has settled, clip(M, 0, 1) = sum over the FEW positive eigenpairs
  // (config 2: 2 of 100), so only the dominant invariant subspace is followed from one ADMM iteration to the next
  int sub_enable, sub_qmax, sub_chunk, sub_debug, sub_lazy; double sub_tol, sub_adapt;   // sub_lazy: orthonormalise once per chunk instead of after every power step
  double* sub_zscratch;   // B * 16 * (np16 + 2): Z block of k_cone_sub for orders beyond 512 (NULL below)
  double* Xs;             // B * np16 * 16: orthonormal Ritz basis (column-major, ld = np16, zero padded rows)
  double* sub_theta;      // B * 16: Ritz values of the last accepted call
  double* trM;            // B: trace of Mbuf (with fro2 it bounds the untracked part of the spectrum)
  double* V3; int* v3valid;  // B * 256, B: eigenvectors of the last small-cone projection (order <= 16), warm start of the next one (NULL: cold every time)
  int *sub_wait, *sub_nfail; // B: iterations left before the subspace is tried again after a failure ; failures so far (back-off)
  int *sub_on, *cone_done;   // B: slot follows the subspace ; this iteration's W1 has been written by k_cone_sub
  // the slots the full kernel must project this iteration are known before it starts (no block yet, or backing off): ws_first[b], written by
  // k_setup / k_global.  ws_phase 1 = those slots only (runs beside k_cone_sub on its own stream), 2 = what k_cone_sub then left (a failed
  // call: ~1 in 30 000), 0 = both in one launch after k_cone_sub (ws_first NULL: the Shor-mode views)
  int* ws_first; int ws_phase;
  int sub_guard;             // Ritz values of the tracked block that must stay negative (the block holds at most 16 - sub_guard positive ones)
  // certificate estimator (k_cone_sub<1>): block of the most negative eigenvectors of Mchk, its Ritz values (of -Mchk), trace of MbufC
  int* sep_done;          // B: the separation vector of this harvested slot came from the tracked block (k_cone_sub<2>); NULL = feature off
  int cert_enable; double *XsC, *sub_thetaC, *trMc, *lb_est; int *sub_onC, *confirm;
  // Shor mode (omc_shor_relax.hip): the big cone [Y X; X' Theta] is explicit, the column blocks are not used
  int shor;               // 1: k_global takes the big cone's copy of Y (rx P0 + (1 - rx) Y + D0, leading dimension shN) instead of the column blocks
  int shN;                // n + m
  const double *shP0, *shD0;   // B * shN * shN
  double clip_hi;         // upper clip of the cone kernels (1 for 0 <= Y <= I; 1e300 for the big cone's PSD projection)
  double inv_s2;          // 1 / scale^2: the Shor solve runs on scale * A, objective and bound are reported unscaled (1 otherwise)
  // warm start from a parent's final state (omc_state_pool_create / omc_relax_set_warm): per-node pool indices (-1: cold / not saved)
  const int *load_from, *save_to;      // Btot each, or NULL
  double *pY, *pD1, *pD3, *pU, *palpha, *psval, *pXs, *ptheta, *pscal;   // pool: strides n*n, n*n, n*n, n*k, nnz, m, np16*16, 16, 4 (rho, sub_on, -, -)
  int* sub_stat;          // B * 8: calls, power steps, failures (fall back to the full decomposition), seeds, failures by cause (too many positive Ritz values, step cap, Cholesky), Rayleigh-Ritz passes
  // rows
  int* R;                 // B
  int *rkind, *rcut, *rbi, *rbj;  // B*Rmax
  double* rcoef;          // B*Rmax*k
  double* rrhs;           // B*Rmax
  double* cutx;           // B*Lmax*n
  double* G;              // B*Rmax*Rmax
  double* lam;            // B*Rmax
  // scalars per node
  double *obj, *objout, *objprev, *lbprev, *lb, *c0, *evsum, *cpen, *cst, *rp, *rd, *lmin;  // B (lmin 2B)
  double* bx;             // B*n
  int *done, *status, *iters, *sweeps, *stall;
  int* rowov;             // B: 1 = the last row projection overflowed its passive set (NNQP_PMAX): the iterate may violate rows
  // scratch
  double* cp_scratch;  size_t cp_scratch_stride;   // per wave (B*m waves) when a column is too large for LDS
  int cp_lds_c; int cp_lds_doubles; int cp_keepB;   // cp_keepB = 0: dense columns, B is gathered again instead of kept in LDS
  // k_colprox_pair: two columns per wave (32 lanes each, matrix rows in registers) for the column pairs (2p, 2p + 1) whose columns hold at
  // most 32 observed rows each; the other columns (cp_solo, cp_nsolo of them) keep the one-column-per-wave kernel
  int cp_pair; int cp_nsolo; const int* cp_solo;
  int cp_nwide; const int* cp_wide;      // columns outside the pairs with at most 64 observed rows: k_colprox_wide (one column per wave, same algorithm)
  double* Yx;                // B*n*n: 2 Y - Yp, written by k_global (and k_setup) for the column gathers of k_colprox* (one load per entry instead of two); NULL with acceleration or Shor mode
  int cone_512;              // diagnostics (OMC_CONE_512): the 512-thread form of the L2-resident eigen-kernel at orders 193..256
  int cp_series;             // Neumann-series order of k_colprox_pair's finish (6; 3 = the second-order finish of colprox_reg)
  int cp_maxpass;            // diagnostics (OMC_CP_MAXPASS): cap on the secular passes of k_colprox_pair; 60 = the algorithm
  double* cone_scratch; size_t cone_scratch_stride; // per node when N is too large for LDS
  double* glob_scratch; size_t glob_scratch_stride;
  double* small_scratch; size_t small_scratch_stride;
  double* chk_scratch;   // B*n*k
  double* stamps;        // 32 doubles (diagnostic builds)
  // Anderson acceleration (k_aa): per slot a ring of (aa_mem + 1) residuals f = T(z) - z and images g = T(z) of the state
  // vector z = (Y, Yp, D1, D3, Vt, D3V, D3T, alpha), the input zin of the running iteration, and a few scalars
  int accel, aa_mem, aa_every, aa_start, aa_dim;
  double aa_reg, aa_safeguard;
  double *aa_F, *aa_G, *aa_zin;   // B*(aa_mem+1)*aa_dim, same, B*aa_dim
  double* aa_fn;                  // B: residual norm of the last plain step while a point awaits verification
  int *aa_hist, *aa_head, *aa_pending, *aa_valid, *aa_nacc, *aa_nrej;   // B each; aa_valid = 0 requests "copy the state into zin, forget history"
};
#define AA_MAXMEM 10

#ifdef __cplusplus
extern "C" {
#endif
void omc_launch_setup(const OmcWS* w, hipStream_t s);
void omc_launch_colprox(const OmcWS* w, int mode, hipStream_t s);
void omc_launch_cone(const OmcWS* w, int mode, int use_lds, size_t lds_bytes, hipStream_t s);
void omc_launch_global(const OmcWS* w, int use_lds, size_t lds_bytes, hipStream_t s);
void omc_launch_small(const OmcWS* w, int mode, int use_lds, size_t lds_bytes, hipStream_t s);
void omc_launch_cone_ws(const OmcWS* w, int lpp, int use_lds, size_t lds_bytes, hipStream_t s);
void omc_launch_cone_sub(const OmcWS* w, hipStream_t s);
size_t omc_cone_sub_lds(int np16);
void omc_launch_check_zero(const OmcWS* w, hipStream_t s);
void omc_launch_check_build(const OmcWS* w, hipStream_t s);
void omc_launch_check_final(const OmcWS* w, int last, int phase, hipStream_t s);
void omc_launch_cert_sub(const OmcWS* w, hipStream_t s);
void omc_launch_sep_sub(const OmcWS* w, hipStream_t s);
void omc_launch_rho_rescale(const OmcWS* w, hipStream_t s);
void omc_launch_harvest(const OmcWS* w, hipStream_t s);
void omc_launch_state_save(const OmcWS* w, hipStream_t s);
void omc_launch_aa(const OmcWS* w, hipStream_t s);
void omc_launch_make_X(const OmcWS* w, double* X, hipStream_t s);
void omc_launch_make_Theta(const OmcWS* w, const double* X, double* Th, hipStream_t s);
void omc_launch_eval_objective(int B, int n, int m, double gamma, const double* A, const uint8_t* mask, const double* X,
                               double* out, hipStream_t s);
int omc_set_max_lds(void);
void omc_launch_gram_XXt(const OmcWS* w, const double* X, int B, hipStream_t s);
void omc_launch_colprox_sweep(const OmcWS* w, hipStream_t s);      /* omc_colprox.hip */
#ifdef __cplusplus
}
#endif
#endif
