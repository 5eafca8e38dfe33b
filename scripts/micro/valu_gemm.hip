// Inner loop of a register-tiled fp64 GEMM on the VECTOR ALU (v_fma_f64), operands read from LDS as broadcasts:
// how fast can a non-MFMA core go on gfx950?  (v_mfma_f64_16x16x4 tops out at 48 TFLOP/s, scripts/micro/mfma_peak.hip.)
// One workgroup = 4 waves, each wave a (8*MR) x (8*MC) tile (lanes in an 8 x 8 grid, MR x MC accumulators per lane); the K slab
// (16 columns of A and of B, k-major so that a lane's MR rows at one k are contiguous) sits in LDS and is swept `iters`
// times -- no global traffic in the timed loop, i.e. the ceiling of VALU + LDS operand fetch.
// Build: hipcc --offload-arch=gfx950 -O3 valu_gemm.hip -o valu_gemm ; run: ./valu_gemm
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MR, int MC, int OCC>
__global__ void __launch_bounds__(256, OCC) k_valu(double* out, int iters) {
  constexpr int TR = 2 * 8 * MR, TC = 2 * 8 * MC, KS = 16;      // workgroup tile TR x TC (2 x 2 waves)
  __shared__ __attribute__((aligned(16))) double As[KS][TR];
  __shared__ __attribute__((aligned(16))) double Bs[KS][TC];
  const int tid = threadIdx.x;
  for (int i = tid; i < KS * TR; i += 256) As[i / TR][i % TR] = 1.0 + 1e-3 * (i % 7);
  for (int i = tid; i < KS * TC; i += 256) Bs[i / TC][i % TC] = 1.0 - 1e-3 * (i % 5);
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63, lr = lane >> 3, lc = lane & 7;
  const int r0 = (wave >> 1) * 8 * MR + lr * MR, c0 = (wave & 1) * 8 * MC + lc * MC;
  double acc[MR][MC];
#pragma unroll
  for (int r = 0; r < MR; ++r)
#pragma unroll
    for (int c = 0; c < MC; ++c) acc[r][c] = 0.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll 4
    for (int k = 0; k < KS; ++k) {
      double a[MR], b[MC];
#pragma unroll
      for (int r = 0; r < MR; ++r) a[r] = As[k][r0 + r];
#pragma unroll
      for (int c = 0; c < MC; ++c) b[c] = Bs[k][c0 + c];
#pragma unroll
      for (int r = 0; r < MR; ++r)
#pragma unroll
        for (int c = 0; c < MC; ++c) acc[r][c] = __builtin_fma(a[r], b[c], acc[r][c]);
    }
  }
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < MR; ++r)
#pragma unroll
    for (int c = 0; c < MC; ++c) s += acc[r][c];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int MR, int MC, int OCC>
void run(int wgs_per_cu) {
  double* out; hipMalloc(&out, (size_t)256 * 8 * 256 * 8);
  const int grid = 256 * wgs_per_cu, iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_valu<MR, MC, OCC>), dim3(grid), dim3(256), 0, 0, out, 10);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_valu<MR, MC, OCC>), dim3(grid), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)grid * iters * 16.0 * (2 * 8 * MR) * (2 * 8 * MC) * 2.0;
  printf("VALU GEMM core %dx%d accumulators/lane (wave tile %dx%d), %d workgroup(s)/CU: %8.3f ms  %6.2f TFLOP/s\n", MR, MC, 8 * MR, 8 * MC, wgs_per_cu, ms,
         fl / ms / 1e9);
  hipFree(out);
}

int main() {
  run<8, 8, 1>(1); run<8, 8, 2>(2);
  run<8, 4, 2>(2); run<8, 4, 2>(3);
  run<4, 4, 2>(2); run<4, 4, 2>(4);
  run<6, 6, 2>(2);
  return 0;
}
