// omc_shor.h -- launchers of omc_shor.hip (Shor-minor enumeration / violated-minor selection)
#ifndef OMC_SHOR_H
#define OMC_SHOR_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SH_BOTH 0
#define SH_XOR 1
#define SH_NONE 2
#define SH_COMBO 0
#define SH_PRODUCT 1

#ifdef __cplusplus
extern "C" {
#endif
void omc_shor_launch_pair_counts(int n, int m, int W, const uint64_t* bits, long long npairs, int* cb, int* cx, int* cz, hipStream_t s);
void omc_shor_launch_seg_scan(int kind, int la, int lb, const int* cb, const int* cx, const int* cz, long long npairs, long long base,
                              long long* off, long long* total, hipStream_t s);
void omc_shor_launch_enum_tuples(int n, int m, int W, const uint64_t* bits, int kind, int la, int lb, const long long* off,
                                 long long npairs, long long* out, hipStream_t s);
void omc_shor_launch_enum_keys(int n, int m, int W, const uint64_t* bits, int kind, int la, int lb, const long long* off,
                               long long npairs, const double* X, int k, const uint64_t* existing, long long n_existing,
                               uint64_t* hi, uint64_t* lo, unsigned long long* n_excluded, hipStream_t s);
void omc_shor_launch_hist(long long N, const uint64_t* hi, const uint64_t* lo, uint64_t phi, uint64_t plo, int level,
                          unsigned long long* hist, hipStream_t s);
void omc_shor_launch_emit(long long N, const uint64_t* hi, const uint64_t* lo, uint64_t bhi, uint64_t blo, uint64_t* ohi, uint64_t* olo,
                          unsigned long long* counter, unsigned long long cap, hipStream_t s);
#ifdef __cplusplus
}
#endif
#endif
