// omc_altmin.h -- workspace descriptor of the alternating-minimisation kernel
#ifndef OMC_ALTMIN_H
#define OMC_ALTMIN_H
#include <hip/hip_runtime.h>
struct AltminWS {
  int B, n, m, k, Rmax, Lmax, max_iters;
  double gamma, eps, sumA2;
  const int *col_ptr, *col_idx; const double* col_val;   // CSC of observed entries
  const int *row_ptr, *row_idx; const double* row_val;   // CSR of observed entries
  const int* R;            // B
  const int *rkind, *rcut, *rbi;   // B*Rmax   (ROW_BOX uses rbi = row index)
  const int* rbj;          // B*Rmax   column of U the row acts on (rank > 1)
  const double *rcoef, *rrhs;      // B*Rmax   (coefficient on x'u or on u_i)
  const double* cutx;      // B*Lmax*n
  const double* U0;        // B*n
  double *U, *V;           // B*n, B*m
  double* objectives;      // B*max_iters
  int *converged, *n_iters;
  double* mobj;            // B: master objective (OMC.jl:2352-2358) of X = U V, evaluated on the device from the factors
  double* G;               // B*Rmax*Rmax scratch
  double* scratch; size_t scratch_stride;   // per-problem global slab that replaces the dynamic LDS when the problem does not fit it (NULL: LDS)
};

#ifdef __cplusplus
extern "C" {
#endif
void omc_launch_altmin(const void* ws, size_t lds_bytes, hipStream_t s);
void omc_launch_altmin_k(const void* ws, size_t lds_bytes, hipStream_t s);
int omc_altmin_set_lds(void);
#ifdef __cplusplus
}
#endif
#endif
