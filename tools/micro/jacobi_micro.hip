// micro-benchmark of the one-sided Jacobi sweep step (timing only variants) -- build: hipcc --offload-arch=gfx950 -O3 jacobi_micro.hip -o jacobi_micro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double double2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double dpp_move(double v, const int ctrl_sel) {
  int lo = __double2loint(v), hi = __double2hiint(v); int lo2, hi2;
  switch (ctrl_sel) {
    case 0: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); break;
    case 1: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); break;
    case 2: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, true); break;
    default: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, true); break;
  }
  return __hiloint2double(hi2, lo2);
}
template <int LPP> __device__ __forceinline__ double gsum(double v) {
  v += dpp_move(v, 0); v += dpp_move(v, 1);
  if (LPP >= 8) v += dpp_move(v, 2);
  if (LPP >= 16) v += dpp_move(v, 3);
  return v;
}
__device__ __forceinline__ void rr_pair(int step, int t, int Np, int& p, int& q) {
  const int M1 = Np - 1;
  if (t == 0) { p = step; q = Np - 1; } else { p = step + t; if (p >= M1) p -= M1; q = step - t; if (q < 0) q += M1; }
  if (p > q) { int tmp = p; p = q; q = tmp; }
}
// VAR: 0 full, 1 no barrier, 2 no rotation branch (dot only), 3 no ev reads (constants)
template <int LPP, int R2, int VAR>
__global__ void __launch_bounds__(512) k_sweeps(const double* G0, double* out, int N, int nsweeps, double tau) {
  extern __shared__ double smem[];
  const int tid = threadIdx.x, T = blockDim.x;
  const int Np = (N + 1) & ~1, rpl = 2 * R2, Nrp = rpl * LPP, ld = Nrp + 2;
  double* Gm = smem; double* ev = Gm + (size_t)Np * ld;
  for (int e = tid; e < Np * ld; e += T) Gm[e] = 0.0;
  __syncthreads();
  for (int e = tid; e < N * N; e += T) { int i = e % N, j = e / N; Gm[(size_t)j * ld + i] = G0[(size_t)blockIdx.x * N * N + e]; }
  for (int t = tid; t < Np; t += T) ev[t] = 1.0;
  __syncthreads();
  const int ngroups = T / LPP, grp = tid / LPP, lg = tid % LPP, npairs = Np >> 1;
  const double tau2 = tau * tau;
  int big = 0;
  for (int sw = 0; sw < nsweeps; ++sw) {
    for (int step = 0; step < Np - 1; ++step) {
      for (int pr = grp; pr < npairs; pr += ngroups) {
        int p, q; rr_pair(step, pr, Np, p, q);
        auto gp = (double2v*)(Gm + (size_t)p * ld + lg * rpl);
        auto gq = (double2v*)(Gm + (size_t)q * ld + lg * rpl);
        double2v cp_[R2], cq_[R2]; double gm0 = 0.0, gm1 = 0.0;
#pragma unroll
        for (int i = 0; i < R2; ++i) { cp_[i] = gp[i]; cq_[i] = gq[i]; gm0 += cp_[i].x * cq_[i].x; gm1 += cp_[i].y * cq_[i].y; }
        double gm = gsum<LPP>(gm0 + gm1);
        const double a = (VAR == 3) ? 1.0 : ev[p], bb = (VAR == 3) ? 1.0 : ev[q];
        const double g2 = gm * gm, ab = a * bb;
        if (VAR != 2 && g2 > tau2 * ab && ab > 0.0) {
          const float df = (float)(bb - a), gf = (float)gm;
          const float rtf = __builtin_sqrtf(df * df + 4.0f * gf * gf);
          const float den = (df >= 0.0f) ? (df + rtf) : (df - rtf);
          const double tt = (den != 0.0f) ? (double)((2.0f * gf) / den) : 0.0;
          const double cs = rsqrt(1.0 + tt * tt), sn = cs * tt;
#pragma unroll
          for (int i = 0; i < R2; ++i) {
            double2v np_, nq_;
            np_.x = cs * cp_[i].x - sn * cq_[i].x; np_.y = cs * cp_[i].y - sn * cq_[i].y;
            nq_.x = sn * cp_[i].x + cs * cq_[i].x; nq_.y = sn * cp_[i].y + cs * cq_[i].y;
            gp[i] = np_; gq[i] = nq_;
          }
          if (lg == 0 && VAR != 3) { ev[p] = a - tt * gm; ev[q] = bb + tt * gm; }
          big = 1;
        }
        if (VAR == 2) big += (g2 > 1e300);
      }
      if (VAR != 1) __syncthreads();
    }
  }
  if (tid == 0) out[blockIdx.x] = Gm[5] + big;
}
template <int LPP, int R2, int VAR>
float run(const double* dG, double* dout, int B, int N, int nsweeps, double tau, int threads) {
  const int Np = (N + 1) & ~1, ld = 2 * R2 * LPP + 2;
  size_t lds = ((size_t)Np * ld + Np) * 8;
  hipFuncSetAttribute((const void*)k_sweeps<LPP, R2, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_sweeps<LPP, R2, VAR>), dim3(B), dim3(threads), lds, 0, dG, dout, N, nsweeps, tau);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_sweeps<LPP, R2, VAR>), dim3(B), dim3(threads), lds, 0, dG, dout, N, nsweeps, tau);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipError_t er = hipGetLastError(); if (er != hipSuccess) printf("err %s\n", hipGetErrorString(er));
  return ms / 5;
}
int main(int argc, char** argv) {
  const int N = 100, B = argc > 1 ? atoi(argv[1]) : 32, nsw = 4;
  std::vector<double> h((size_t)B * N * N);
  srand(1);
  for (auto& v : h) v = (rand() / (double)RAND_MAX - 0.5) * 1e-3;
  for (int b = 0; b < B; ++b) for (int i = 0; i < N; ++i) h[(size_t)b * N * N + i * N + i] += 1.0 + 0.01 * i;   // nearly orthogonal columns
  double *dG, *dout; hipMalloc(&dG, h.size() * 8); hipMalloc(&dout, B * 8);
  hipMemcpy(dG, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  const double steps = nsw * 99.0;
#define REP(name, L, R, V, thr, tau) { float ms = run<L, R, V>(dG, dout, B, N, nsw, tau, thr); printf("%-44s %8.3f ms  %7.3f us/step %8.0f cyc/step@2.4GHz\n", name, ms, ms * 1e3 / steps, ms * 1e3 / steps * 2400); }
  REP("LPP8 512thr full (rotations rare)", 8, 7, 0, 512, 1e-2);
  REP("LPP8 512thr full (rotations always)", 8, 7, 0, 512, 1e-30);
  REP("LPP8 512thr no barrier", 8, 7, 1, 512, 1e-2);
  REP("LPP8 512thr dot only", 8, 7, 2, 512, 1e-2);
  REP("LPP8 512thr no ev reads", 8, 7, 3, 512, 1e-2);
  REP("LPP4 256thr full (rare)", 4, 13, 0, 256, 1e-2);
  REP("LPP4 256thr full (always)", 4, 13, 0, 256, 1e-30);
  REP("LPP16 1024thr full (rare)", 16, 4, 0, 1024, 1e-2);
  REP("LPP16 1024thr full (always)", 16, 4, 0, 1024, 1e-30);
  REP("LPP16 512thr (2 rounds) rare", 16, 4, 0, 512, 1e-2);
  return 0;
}
