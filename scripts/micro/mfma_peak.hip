// Raw v_mfma_f64_16x16x4_f64 issue-rate probe: NACC independent accumulators per wave, no memory traffic.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// 4x4x4 (4 blocks) variant and plain VALU fp64 FMA for comparison
template <int NACC>
__global__ void __launch_bounds__(256) k_mfma4(double* out, int iters, double a0, double b0) {
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) k_fma(double* out, int iters, double a0, double b0) {
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(a, acc[i], b);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int KIND>
void run2(int wgs_per_cu) {
  double* out; hipMalloc(&out, 256 * 8 * 1024 * 8);
  const int grid = 256 * wgs_per_cu, iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  if (KIND == 0) hipLaunchKernelGGL(k_mfma4<NACC>, dim3(grid), dim3(256), 0, 0, out, 10, 1.0, 1.0); else hipLaunchKernelGGL(k_fma<NACC>, dim3(grid), dim3(256), 0, 0, out, 10, 0.5, 1.0);
  hipEventRecord(e0, 0);
  if (KIND == 0) hipLaunchKernelGGL(k_mfma4<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1.0); else hipLaunchKernelGGL(k_fma<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 0.5, 1.0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double nm = (double)grid * 4 * iters * NACC;
  const double fl = KIND == 0 ? 512.0 : 128.0;
  printf("%s NACC=%2d wgs/CU=%d: %8.3f ms  %7.2f TF/s\n", KIND == 0 ? "mfma_f64_4x4x4" : "v_fma_f64     ", NACC, wgs_per_cu, ms, nm * fl / ms / 1e9);
  hipFree(out);
}
template <int NACC>
void run(int wgs_per_cu, int threads) {
  double* out; hipMalloc(&out, 256 * 8 * 1024 * 8);
  const int grid = 256 * wgs_per_cu, iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(threads), 0, 0, out, 10, 1.0, 1.0);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(threads), 0, 0, out, iters, 1.0, 1.0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double nm = (double)grid * (threads / 64) * iters * NACC;
  printf("NACC=%2d wgs/CU=%d waves/WG=%d: %8.3f ms  %7.2f TF/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz if 4 SIMDs busy)\n", NACC, wgs_per_cu, threads / 64, ms,
         nm * 2048.0 / ms / 1e9, ms * 1e-3 * 2.4e9 / (nm / (256.0 * 4)));
  hipFree(out);
}
int main() {
  run<1>(1, 256); run<2>(1, 256); run<4>(1, 256); run<8>(1, 256); run<16>(1, 256); run<25>(1, 256);
  run<4>(2, 256); run<8>(2, 256); run<16>(2, 256); run<8>(4, 256);
  run2<8, 0>(1); run2<8, 0>(2); run2<16, 0>(2); run2<8, 0>(4);
  run2<8, 1>(1); run2<8, 1>(2); run2<16, 1>(2); run2<16, 1>(4);
  return 0;
}
