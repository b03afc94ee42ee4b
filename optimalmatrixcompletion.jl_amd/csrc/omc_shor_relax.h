// omc_shor_relax.h -- Shor mode of the node relaxation (matrix_completion_SDP_relaxation with add_Shor_valid_inequalities = true,
// rank 1: OMC.jl:1503-1525, 1755-1779, 1838-1846): workspace descriptor and launchers of omc_shor_relax.hip.
//
// The Shor-mode solve rides on the base engine (omc_device.hip: slots, rows, small cone, clip, certificate eigenvalues, harvest);
// what is added here: X, W (on the minor coordinates), Theta explicit, the order-(n+m) cone [Y X; X' Theta] >= 0 (projected by the
// base engine's eigen-kernels through a view of its workspace), one order-5 PSD block per minor (projected in registers, one lane
// per minor), one paraboloid per column (the rotated cones W >= X^2 and Theta_jj = sum_i W_ij off the minor coordinates), and the
// Lagrangian bound of that program.  See DESIGN.md section 3.7 and oracle/omc_oracle_shor.py (the checker mirrors this file).
#ifndef OMC_SHOR_RELAX_H
#define OMC_SHOR_RELAX_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "omc_device.h"

// index structure of one Shor list (shared by every node that carries the same list: the reference's static mode gives all nodes one list)
struct ShorGroupDev {
  int nq, nv1, nv2, pad;
  double r4;            // penalty of this list's order-5 blocks relative to rho (auto: 75 n m / (4 nq), clamped to [0.25, 40])
  const int* mi;        // 4 * nq: i1, i2, j1, j2 (0-based), SoA: mi[c * nq + q]
  const int* kid;       // 4 * nq: key ids of V1[i1,(j1,j2)], V1[i2,(j1,j2)], V2[(i1,i2),j1], V2[(i1,i2),j2], SoA
  const int* cptr;      // n*m + 1: CSR coordinate (column-major e = j*n + i) -> members
  const int* cent;      // 4 * nq: member = 4 * q + p, p = 0..3 the position of the coordinate in the block (OMC.jl:1772 order)
  const int* v1ptr;     // nv1 + 1
  const int* v1ent;     // 2 * nq: member = 2 * q + which (0: position (1,2), 1: position (3,4))
  const int* v2ptr;     // nv2 + 1
  const int* v2ent;     // 2 * nq: member = 2 * q + which (0: position (1,3), 1: position (2,4))
  const uint8_t* eclass;  // n*m: 0 = neither list (W >= 0 only), 1 = SOC list, 2 = in a minor
  const uint8_t* ctype;   // m: 0 free slack, 1 paid slack, 2 equality (oracle/omc_oracle_shor.py header)
  const int* slackrow;    // m: row that carries the slack of Theta_jj = sum_i W_ij in the returned W (-1: type 2)
};

struct ShWS {
  int n, m, k, N, NPb, S, Btot, nqmax, nv1max, nv2max, nmb;   // nmb = minor blocks (256 minors each) per slot
  double rx, r4, r5, gamma, sc;
  const ShorGroupDev* groups; const int* node_group;       // Btot
  const int *node_of, *done, *init, *fin;                  // slot arrays of the base workspace
  const double *rho_b, *bfac;
  const double* Ah;          // n*m: scale * A
  const uint8_t* mask;       // n*m
  double *X, *W, *Th, *V1, *V2, *V3;            // S * (n*m, n*m, m*m, nv1max, nv2max, nqmax)
  double *D0, *P0, *MbufB, *VrowB;              // S * N*N, S * N*N, S * NPb*NPb (x2)
  double *Tq, *Pq, *Nq;                         // S * 15 * nqmax, SoA [e][q]: over-relaxed target / dual ; projection ; projection - input
  double *D5x, *D5t, *nu5, *P5x;                // S * (n*m, m, m, n*m)
  double *colpart, *minpart, *minpart2;         // S * m * 4, S * nmb, S * nmb
  double *fro2B, *trB; int* vvalidB;            // S: squared Frobenius norm and trace of the big cone's next input ; eigenvectors valid
  int *sub_onB, *cone_doneB, *sub_waitB, *sub_nfailB;   // S each: tracked-subspace state of the big cone (k_cone_sub through the view); NULL = off
  const double *Y, *Yp; double *rp, *rd;        // base state
  double *e1, *e2;                              // S * nv1max, S * nv2max: certificate: V residual per key
  double *objcol, *c0col, *lamDX;               // base certificate inputs (S*m, S*m, S*m*n)
  double *oX, *oW, *oTh;                        // Btot * (n*m, n*m, m*m)
  double* oV;                                   // Btot * 5 * nqmax (NULL: not kept): V1a, V1b, V2a, V2b, V3 per minor
};

#ifdef __cplusplus
extern "C" {
#endif
void omc_shor_launch_setup(const ShWS* w, hipStream_t s);
void omc_shor_launch_minor_pre(const ShWS* w, hipStream_t s);
void omc_shor_launch_vkeys(const ShWS* w, hipStream_t s);
void omc_shor_launch_cols(const ShWS* w, hipStream_t s);
void omc_shor_launch_minor_post(const ShWS* w, hipStream_t s);
void omc_shor_launch_reduce(const ShWS* w, hipStream_t s);
void omc_shor_launch_check(const ShWS* w, hipStream_t s);
void omc_shor_launch_rescale(const ShWS* w, hipStream_t s);
void omc_shor_launch_harvest(const ShWS* w, hipStream_t s);
#ifdef __cplusplus
}
#endif
#endif
