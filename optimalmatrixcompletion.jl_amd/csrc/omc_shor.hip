// omc_shor.hip -- Shor-minor enumeration and violated-minor selection (integer / byte work, HBM-write bound)
//
//   generate_rank1_matrix_completion_Shor_constraints_indexes   OMC.jl:2545-2612   
//   generate_violated_Shor_minors                               OMC.jl:2614-2640   (row a11)
//
// A candidate is a 2 x 2 minor (i1 < i2, j1 < j2).  For a row pair the observed-column bitsets r1, r2 give three lists:
// both = r1 & r2, xor = r1 ^ r2, none = ~(r1 | r2).  Every class of the reference ("num_entries_present" p) is either
// the 2-combinations of one list or the product of two lists, pushed pair by pair in (i1, i2) lexicographic order:
//     p = 4 : COMBO(both)                      p = 3 : PRODUCT(both, xor)
//     p = 2 : PRODUCT(both, none) for ALL pairs, then COMBO(xor) for all pairs   (OMC.jl:2568-2583)
//     p = 1 : PRODUCT(xor, none)               p = 0 : COMBO(none)
// (PRODUCT: first list outer, second inner, the two columns stored sorted.)  A "segment" is one of those six passes.
//
// Kernels: per-pair popcounts -> per-segment counts + exclusive scan (one block, carries a running total) ->
// enumeration with one WAVE per pair (bit lists expanded into LDS by popcount ranks, flat index decoded per lane, so
// consecutive lanes write consecutive 32-byte tuples).  The same enumerator feeds the violated-minor scorer, which
// writes a 128-bit key (score bits, tuple rank) per candidate; the top n_minors are found by an MSD radix select over
// the keys (8-bit digits, LDS histograms), i.e. the reference's partialsort! on (score, tuple) tuples, rev = true.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "omc_shor.h"

#define SH_WAVE 64

__device__ __forceinline__ void pair_from_index(long long t, int n, int& i1, int& i2) {
  // t = i1 (2n - i1 - 1) / 2 + (i2 - i1 - 1), 0 <= i1 < i2 < n
  const double b = 2.0 * n - 1.0;
  long long a = (long long)floor((b - sqrt(b * b - 8.0 * (double)t)) * 0.5);
  if (a < 0) a = 0;
  if (a > n - 2) a = n - 2;
  while (a > 0 && a * (2LL * n - a - 1) / 2 > t) --a;
  while ((a + 1) * (2LL * n - (a + 1) - 1) / 2 <= t) ++a;
  i1 = (int)a;
  i2 = (int)(t - a * (2LL * n - a - 1) / 2) + i1 + 1;
}

__device__ __forceinline__ uint64_t list_word(int which, uint64_t a, uint64_t b, uint64_t valid) {
  return (which == SH_BOTH) ? (a & b) : (which == SH_XOR) ? (a ^ b) : (~(a | b) & valid);
}
__device__ __forceinline__ uint64_t valid_word(int w, int m) {
  const int rem = m - w * 64;
  return (rem >= 64) ? ~0ULL : ((1ULL << rem) - 1ULL);
}

// |both|, |xor|, |none| of every row pair
__global__ void __launch_bounds__(256) k_shor_pair_counts(int n, int m, int W, const uint64_t* bits, long long npairs, int* cb,
                                                          int* cx, int* cz) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= npairs) return;
  int i1, i2;
  pair_from_index(t, n, i1, i2);
  const uint64_t* r1 = bits + (size_t)i1 * W;
  const uint64_t* r2 = bits + (size_t)i2 * W;
  int b = 0, x = 0, z = 0;
  for (int w = 0; w < W; ++w) {
    const uint64_t a = r1[w], c = r2[w], v = valid_word(w, m);
    b += __popcll(a & c); x += __popcll(a ^ c); z += __popcll(~(a | c) & v);
  }
  cb[t] = b; cx[t] = x; cz[t] = z;
}

__device__ __forceinline__ long long seg_count(int kind, int la, int lb_, const int* cb, const int* cx, const int* cz, long long t) {
  const long long a = (la == SH_BOTH) ? cb[t] : (la == SH_XOR) ? cx[t] : cz[t];
  if (kind == SH_COMBO) return a * (a - 1) / 2;
  const long long b = (lb_ == SH_BOTH) ? cb[t] : (lb_ == SH_XOR) ? cx[t] : cz[t];
  return a * b;
}

// exclusive scan of one segment's per-pair counts; off[t] = base + sum_{t' < t}; total[0] = sum.  One block.
__global__ void __launch_bounds__(1024) k_shor_seg_scan(int kind, int la, int lb_, const int* cb, const int* cx, const int* cz,
                                                        long long npairs, long long base, long long* off, long long* total) {
  __shared__ long long sw[16];
  __shared__ long long s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (long long t0 = 0; t0 < npairs; t0 += 1024) {
    const long long t = t0 + tid;
    const long long c = (t < npairs) ? seg_count(kind, la, lb_, cb, cx, cz, t) : 0;
    long long v = c;   // inclusive scan inside the wave
    for (int o = 1; o < 64; o <<= 1) { long long u = __shfl_up(v, o, 64); if (lane >= o) v += u; }
    if (lane == 63) sw[wv] = v;
    __syncthreads();
    long long pre = 0;
    for (int i = 0; i < wv; ++i) pre += sw[i];
    long long chunk = 0;
    for (int i = 0; i < 16; ++i) chunk += sw[i];
    const long long carry = s_carry;
    if (t < npairs) off[t] = base + carry + pre + v - c;
    __syncthreads();
    if (tid == 0) s_carry = carry + chunk;
    __syncthreads();
  }
  if (tid == 0) total[0] = s_carry;
}

// ---- enumeration: one wave per row pair ---------------------------------------------------------------------------
struct TupleWriter {   // a10: 1-based (i1, i2, j1, j2) as Int64, the reference's NTuple{4, Int}
  long long* out;
  __device__ __forceinline__ void operator()(long long idx, int i1, int i2, int j1, int j2) const {
    long long* o = out + 4 * idx;
    o[0] = i1 + 1; o[1] = i2 + 1; o[2] = j1 + 1; o[3] = j2 + 1;
  }
};
struct KeyWriter {     // a11: (score bits, tuple rank); candidates found in `existing` become the lowest key (0, 0)
  const double* X; int k, n, m;
  const uint64_t* existing; long long n_existing;
  uint64_t *hi, *lo; unsigned long long* n_excluded;
  __device__ __forceinline__ void operator()(long long idx, int i1, int i2, int j1, int j2) const {
#pragma clang fp contract(off)   // products and the difference are rounded separately, as Julia / numpy do (HIP's __dmul_rn is a plain `*`)
    const uint64_t key = (((uint64_t)i1 * n + i2) * m + j1) * m + j2 + 1;   // > 0; lexicographic order of the tuple
    long long a = 0, b = n_existing;
    while (a < b) { const long long c = (a + b) >> 1; if (existing[c] < key) a = c + 1; else b = c; }
    if (a < n_existing && existing[a] == key) { hi[idx] = 0; lo[idx] = 0; atomicAdd(n_excluded, 1ULL); return; }
    const double* x11 = X + (size_t)k * (i1 + (size_t)n * j1);
    const double* x22 = X + (size_t)k * (i2 + (size_t)n * j2);
    const double* x12 = X + (size_t)k * (i1 + (size_t)n * j2);
    const double* x21 = X + (size_t)k * (i2 + (size_t)n * j1);
    double s = 0.0;   // sum(abs.(X[:,i1,j1] .* X[:,i2,j2] .- X[:,i1,j2] .* X[:,i2,j1])), no contraction, t ascending
    for (int t = 0; t < k; ++t) { const double p1 = x11[t] * x22[t], p2 = x12[t] * x21[t]; s = s + fabs(p1 - p2); }
    hi[idx] = (uint64_t)__double_as_longlong(s);   // s >= 0: the bit pattern orders like the value
    lo[idx] = key;
  }
};

template <class F>
__global__ void __launch_bounds__(256) k_shor_enum(int n, int m, int W, const uint64_t* bits, int kind, int la, int lb_,
                                                   const long long* off, long long npairs, F f) {
  extern __shared__ int s_lists[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  int* LA = s_lists + (size_t)wv * 2 * m;
  int* LB = LA + m;
  for (long long t = (long long)blockIdx.x * wpb + wv; t < npairs; t += (long long)gridDim.x * wpb) {
    int i1, i2;
    pair_from_index(t, n, i1, i2);
    const uint64_t* r1 = bits + (size_t)i1 * W;
    const uint64_t* r2 = bits + (size_t)i2 * W;
    int na = 0, nb = 0;
    for (int w = 0; w < W; ++w) {
      const uint64_t a = r1[w], c = r2[w], v = valid_word(w, m);
      const uint64_t wa = list_word(la, a, c, v);
      if ((wa >> lane) & 1ULL) LA[na + __popcll(wa & ((1ULL << lane) - 1ULL))] = w * 64 + lane;
      na += __popcll(wa);
      if (kind == SH_PRODUCT) {
        const uint64_t wb = list_word(lb_, a, c, v);
        if ((wb >> lane) & 1ULL) LB[nb + __popcll(wb & ((1ULL << lane) - 1ULL))] = w * 64 + lane;
        nb += __popcll(wb);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const long long o = off[t];
    if (kind == SH_COMBO) {
      const int cnt = na * (na - 1) / 2;
      const double bq = 2.0 * na - 1.0;
      for (int e = lane; e < cnt; e += SH_WAVE) {
        int p = (int)floor((bq - sqrt(bq * bq - 8.0 * (double)e)) * 0.5);
        if (p < 0) p = 0;
        while (p > 0 && p * (2 * na - p - 1) / 2 > e) --p;
        while ((p + 1) * (2 * na - (p + 1) - 1) / 2 <= e) ++p;
        const int q = e - p * (2 * na - p - 1) / 2 + p + 1;
        f(o + e, i1, i2, LA[p], LA[q]);
      }
    } else {
      const unsigned cnt = (unsigned)na * (unsigned)nb;   // m <= 8192: fits 32 bits
      for (unsigned e = lane; e < cnt; e += SH_WAVE) {
        const int p = (int)(e / (unsigned)nb), q = (int)(e - (unsigned)p * (unsigned)nb);
        const int ja = LA[p], jb = LB[q];
        f(o + e, i1, i2, min(ja, jb), max(ja, jb));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- radix select over 128-bit keys (hi, lo): histogram of the digit at `level` (0 = most significant byte) among the
//      keys whose higher digits equal the prefix ------------------------------------------------------------------------
__device__ __forceinline__ unsigned key_digit(uint64_t hi, uint64_t lo, int level) {
  return (level < 8) ? (unsigned)((hi >> (56 - 8 * level)) & 0xFF) : (unsigned)((lo >> (56 - 8 * (level - 8))) & 0xFF);
}
__device__ __forceinline__ bool key_matches(uint64_t hi, uint64_t lo, uint64_t phi, uint64_t plo, int level) {
  if (level == 0) return true;
  if (level <= 8) { const int sh = 64 - 8 * level; return (sh == 0) ? (hi == phi) : ((hi >> sh) == (phi >> sh)); }
  const int sh = 64 - 8 * (level - 8);
  return hi == phi && ((lo >> sh) == (plo >> sh));
}
__global__ void __launch_bounds__(256) k_sel_hist(long long N, const uint64_t* hi, const uint64_t* lo, uint64_t phi, uint64_t plo,
                                                  int level, unsigned long long* hist) {
  __shared__ unsigned int sh[256];
  sh[threadIdx.x] = 0;
  __syncthreads();
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < N; e += (long long)gridDim.x * blockDim.x) {
    const uint64_t h = hi[e], l = lo[e];
    if (key_matches(h, l, phi, plo, level)) atomicAdd(&sh[key_digit(h, l, level)], 1u);
  }
  __syncthreads();
  if (sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)sh[threadIdx.x]);
}
// append every key >= (bhi, blo)
__global__ void __launch_bounds__(256) k_sel_emit(long long N, const uint64_t* hi, const uint64_t* lo, uint64_t bhi, uint64_t blo,
                                                  uint64_t* ohi, uint64_t* olo, unsigned long long* counter, unsigned long long cap) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < N; e += (long long)gridDim.x * blockDim.x) {
    const uint64_t h = hi[e], l = lo[e];
    if (h > bhi || (h == bhi && l >= blo)) {
      const unsigned long long p = atomicAdd(counter, 1ULL);
      if (p < cap) { ohi[p] = h; olo[p] = l; }
    }
  }
}

extern "C" {
void omc_shor_launch_pair_counts(int n, int m, int W, const uint64_t* bits, long long npairs, int* cb, int* cx, int* cz, hipStream_t s) {
  const int blocks = (int)((npairs + 255) / 256);
  if (blocks > 0) hipLaunchKernelGGL(k_shor_pair_counts, dim3(blocks), dim3(256), 0, s, n, m, W, bits, npairs, cb, cx, cz);
}
void omc_shor_launch_seg_scan(int kind, int la, int lb, const int* cb, const int* cx, const int* cz, long long npairs, long long base,
                              long long* off, long long* total, hipStream_t s) {
  hipLaunchKernelGGL(k_shor_seg_scan, dim3(1), dim3(1024), 0, s, kind, la, lb, cb, cx, cz, npairs, base, off, total);
}
static void enum_geometry(int m, long long npairs, int& wpb, int& blocks, size_t& lds) {
  wpb = 4;
  while (wpb > 1 && (size_t)wpb * 2 * m * sizeof(int) > 48 * 1024) wpb >>= 1;
  lds = (size_t)wpb * 2 * m * sizeof(int);
  long long b = (npairs + wpb - 1) / wpb;
  const long long cap = 256LL * 64;   // >> 256 CUs; waves stride over the pairs
  blocks = (int)(b < cap ? b : cap);
}
void omc_shor_launch_enum_tuples(int n, int m, int W, const uint64_t* bits, int kind, int la, int lb, const long long* off,
                                 long long npairs, long long* out, hipStream_t s) {
  int wpb, blocks; size_t lds;
  enum_geometry(m, npairs, wpb, blocks, lds);
  if (blocks <= 0) return;
  TupleWriter f{out};
  hipLaunchKernelGGL(k_shor_enum<TupleWriter>, dim3(blocks), dim3(wpb * 64), lds, s, n, m, W, bits, kind, la, lb, off, npairs, f);
}
void omc_shor_launch_enum_keys(int n, int m, int W, const uint64_t* bits, int kind, int la, int lb, const long long* off,
                               long long npairs, const double* X, int k, const uint64_t* existing, long long n_existing,
                               uint64_t* hi, uint64_t* lo, unsigned long long* n_excluded, hipStream_t s) {
  int wpb, blocks; size_t lds;
  enum_geometry(m, npairs, wpb, blocks, lds);
  if (blocks <= 0) return;
  KeyWriter f{X, k, n, m, existing, n_existing, hi, lo, n_excluded};
  hipLaunchKernelGGL(k_shor_enum<KeyWriter>, dim3(blocks), dim3(wpb * 64), lds, s, n, m, W, bits, kind, la, lb, off, npairs, f);
}
void omc_shor_launch_hist(long long N, const uint64_t* hi, const uint64_t* lo, uint64_t phi, uint64_t plo, int level,
                          unsigned long long* hist, hipStream_t s) {
  long long b = (N + 255) / 256;
  const int blocks = (int)(b < 4096 ? b : 4096);
  if (blocks > 0) hipLaunchKernelGGL(k_sel_hist, dim3(blocks), dim3(256), 0, s, N, hi, lo, phi, plo, level, hist);
}
void omc_shor_launch_emit(long long N, const uint64_t* hi, const uint64_t* lo, uint64_t bhi, uint64_t blo, uint64_t* ohi, uint64_t* olo,
                          unsigned long long* counter, unsigned long long cap, hipStream_t s) {
  long long b = (N + 255) / 256;
  const int blocks = (int)(b < 4096 ? b : 4096);
  if (blocks > 0) hipLaunchKernelGGL(k_sel_emit, dim3(blocks), dim3(256), 0, s, N, hi, lo, bhi, blo, ohi, olo, counter, cap);
}
}
