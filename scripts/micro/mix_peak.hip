// Do v_mfma_f64_16x16x4 and v_fma_f64 overlap on a gfx950 SIMD?  One wave issues NM independent MFMAs and NF independent
// VALU FMAs per loop trip (no memory traffic).  If the two pipes are separate hardware the trip costs max(t_mfma, t_fma),
// if they share the fp64 multipliers it costs the sum.
// Build: hipcc --offload-arch=gfx950 -O3 mix_peak.hip -o mix_peak ; run: ./mix_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NM, int NF, int KIND>
__global__ void __launch_bounds__(256) k_mix(double* out, int iters, double a0, double b0) {
  d4 acc[NM > 0 ? NM : 1];
  double m4[NM > 0 ? NM : 1];
  double f[NF > 0 ? NF : 1];
#pragma unroll
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) { acc[i] = (d4){0.0, 0.0, 0.0, 0.0}; m4[i] = 0.0; }
#pragma unroll
  for (int i = 0; i < (NF > 0 ? NF : 1); ++i) f[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
    // interleave: one MFMA, then NF/NM FMAs
#pragma unroll
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) {
      if (NM > 0) {
        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        else m4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m4[i], 0, 0, 0);
      }
      constexpr int per = NF / (NM > 0 ? NM : 1);
#pragma unroll
      for (int j = 0; j < per; ++j) f[i * per + j] = __builtin_fma(a, f[i * per + j], b);
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + m4[i];
#pragma unroll
  for (int i = 0; i < (NF > 0 ? NF : 1); ++i) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NM, int NF, int KIND>
void run(int wgs_per_cu) {
  double* out; hipMalloc(&out, 256 * 8 * 1024 * 8);
  const int grid = 256 * wgs_per_cu, iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_mix<NM, NF, KIND>), dim3(grid), dim3(256), 0, 0, out, 10, 0.5, 1.0);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_mix<NM, NF, KIND>), dim3(grid), dim3(256), 0, 0, out, iters, 0.5, 1.0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)grid * 4;
  const double fl = waves * iters * (NM * (KIND == 0 ? 2048.0 : 512.0) + NF * 128.0);
  printf("%s NM=%2d NF=%2d wgs/CU=%d: %8.3f ms  %7.2f TF/s total (mfma part %6.2f, valu part %6.2f)\n", KIND == 0 ? "16x16x4" : "4x4x4  ", NM, NF,
         wgs_per_cu, ms, fl / ms / 1e9, waves * iters * NM * (KIND == 0 ? 2048.0 : 512.0) / ms / 1e9, waves * iters * NF * 128.0 / ms / 1e9);
  hipFree(out);
}
int main() {
  // 16x16x4: one MFMA = 2048 flop ~ 105 cycles; one FMA = 128 flop ~ 4-8 cycles
  run<8, 0, 0>(2); run<0, 64, 0>(2);
  run<8, 32, 0>(2); run<8, 64, 0>(2); run<8, 96, 0>(2); run<4, 64, 0>(2); run<4, 96, 0>(2);
  run<8, 64, 0>(1); run<4, 64, 0>(4);
  // 4x4x4: one MFMA = 512 flop ~ 16 cycles
  run<8, 0, 1>(2); run<8, 8, 1>(2); run<8, 16, 1>(2); run<8, 32, 1>(2);
  return 0;
}
