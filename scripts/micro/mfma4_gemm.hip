// LDS-fed inner loop of a fp64 GEMM core on v_mfma_f64_4x4x4_4b: what does the 4x4x4 form deliver once its operands come
// from LDS?  (Register-only it issues 71-75 TFLOP/s, v_mfma_f64_16x16x4 45-48: profiles/r1_mfma_f64_issue_rate.txt.)
// One workgroup = 4 waves (2 x 2); a wave owns a (16*RA) x (4*CB) tile: RA A-fragments (16 rows x 4 k: lane = 16k + 4q + i holds
// row 4q + i) times CB B-fragments (4 cols x 4 k, replicated over the four blocks q: lane = 16k + 4q + j holds col j), one MFMA per
// pair; the D lane 16i + 4q + j holds C[4q + i][j].  The K slab (16 columns, [row][k] with leading dimension 17 like the product
// core) sits in LDS and is swept `iters` times -- no global traffic in the timed loop.
// Build: hipcc --offload-arch=gfx950 -O3 mfma4_gemm.hip -o mfma4_gemm ; run: ./mfma4_gemm
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int RA, int CB, int OCC>
__global__ void __launch_bounds__(256, OCC) k_m4(double* out, int iters) {
  constexpr int TR = 2 * 16 * RA, TC = 2 * 4 * CB, KS = 16, LD = 17;
  __shared__ __attribute__((aligned(16))) double As[TR * LD];
  __shared__ __attribute__((aligned(16))) double Bs[TC * LD];
  const int tid = threadIdx.x;
  for (int i = tid; i < TR * LD; i += 256) As[i] = 1.0 + 1e-3 * (i % 7);
  for (int i = tid; i < TC * LD; i += 256) Bs[i] = 1.0 - 1e-3 * (i % 5);
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const int lk = lane >> 4, lq = (lane >> 2) & 3, li = lane & 3;
  const int r0 = (wave >> 1) * 16 * RA, c0 = (wave & 1) * 4 * CB;
  double acc[RA][CB];
#pragma unroll
  for (int r = 0; r < RA; ++r)
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[r][c] = 0.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < KS / 4; ++ks) {
      double a[RA], b[CB];
#pragma unroll
      for (int r = 0; r < RA; ++r) a[r] = As[(r0 + 16 * r + 4 * lq + li) * LD + ks * 4 + lk];
#pragma unroll
      for (int c = 0; c < CB; ++c) b[c] = Bs[(c0 + 4 * c + li) * LD + ks * 4 + lk];
#pragma unroll
      for (int r = 0; r < RA; ++r)
#pragma unroll
        for (int c = 0; c < CB; ++c) acc[r][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[r], b[c], acc[r][c], 0, 0, 0);
    }
  }
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < RA; ++r)
#pragma unroll
    for (int c = 0; c < CB; ++c) s += acc[r][c];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int RA, int CB, int OCC>
void run(int wgs_per_cu) {
  double* out; hipMalloc(&out, (size_t)256 * 8 * 256 * 8);
  const int grid = 256 * wgs_per_cu, iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_m4<RA, CB, OCC>), dim3(grid), dim3(256), 0, 0, out, 10);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_m4<RA, CB, OCC>), dim3(grid), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)grid * iters * 16.0 * (2 * 16 * RA) * (2 * 4 * CB) * 2.0;
  printf("4x4x4 LDS-fed core, wave tile %3d x %3d (%d A x %d B fragments, %d accumulators/lane), %d workgroup(s)/CU: %8.3f ms  %6.2f TFLOP/s\n", 16 * RA, 4 * CB, RA, CB,
         RA * CB, wgs_per_cu, ms, fl / ms / 1e9);
  hipFree(out);
}

int main() {
  run<2, 8, 2>(2); run<2, 8, 2>(3);
  run<4, 8, 1>(1); run<4, 8, 2>(2); run<4, 8, 3>(3);
  run<4, 4, 2>(2); run<4, 4, 4>(4);
  run<8, 8, 1>(1); run<8, 8, 2>(2);
  run<4, 16, 1>(1); run<4, 16, 2>(2);
  run<3, 12, 2>(2);
  return 0;
}
