// fp64_peak.hip -- issue-rate microbenchmark for the two fp64 pipes of gfx950: v_mfma_f64_16x16x4_f64 and v_fma_f64.
// The local guide lists no fp64 peak and AMD's datasheet figure (78.6 TFLOP/s, vector = matrix) is not in this image, so the
// denominator of bench.py's roofline is measured here (SURVEY.md 8d asks for exactly this).  Prints one JSON line.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/fp64_peak.hip -o tools/micro/fp64_peak.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double double4v __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(double* out, int iters) {
  double4v acc[NACC];
#pragma unroll
  for (int q = 0; q < NACC; ++q) acc[q] = {0.0, 0.0, 0.0, 0.0};
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) k_fma(double* out, int iters) {
  double acc[NACC];
#pragma unroll
  for (int q = 0; q < NACC; ++q) acc[q] = 1e-3 * (q + threadIdx.x);
  const double a = 1.0 + 1e-12 * threadIdx.x, b = 1e-13;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = fma(acc[q], a, b);
  }
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < NACC; ++q) s += acc[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* out; hipMalloc(&out, sizeof(double) * 256 * cus * 8);
  const int iters = 4096;
  double best_mfma = 0, best_fma = 0; int wm = 0, wf = 0;
  for (int wpc = 1; wpc <= 2; ++wpc) {          // workgroups of 4 waves per CU: 1 or 2 waves per SIMD
    const int blocks = cus * wpc;
    double ms = time_ms([&] { hipLaunchKernelGGL((k_mfma<8>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
    double tf = (double)blocks * 4 * 8 * iters * 2048.0 / (ms * 1e-3) / 1e12;          // 16*16*4*2 flop per MFMA per wave
    if (tf > best_mfma) { best_mfma = tf; wm = wpc; }
    ms = time_ms([&] { hipLaunchKernelGGL((k_fma<16>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
    tf = (double)blocks * 256 * 16 * iters * 2.0 / (ms * 1e-3) / 1e12;
    if (tf > best_fma) { best_fma = tf; wf = wpc; }
  }
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"mfma_f64_16x16x4_tflops\": %.2f, \"mfma_waves_per_simd\": %d, \"fma_f64_tflops\": %.2f, \"fma_waves_per_simd\": %d, "
         "\"datasheet_tflops\": 78.6, \"method\": \"back-to-back issue, 8 (MFMA) / 16 (FMA) independent accumulators per wave, %d iterations, HIP events\"}\n",
         p.name, cus, p.clockRate / 1000, best_mfma, wm, best_fma, wf, iters);
  hipFree(out);
  return 0;
}
