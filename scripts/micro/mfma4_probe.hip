// Operand/result lane layout probe of v_mfma_f64_4x4x4_4b_f64: one-hot A lane x one-hot B lane -> which D lanes light up.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k_probe(double* out) {
  const int lane = threadIdx.x;
  for (int t = 0; t < 64 * 64; ++t) {
    const double a = (lane == t / 64) ? 1.0 : 0.0, b = (lane == t % 64) ? 1.0 : 0.0;
    out[(size_t)t * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  }
}
int main() {
  double* d; hipMalloc(&d, 64 * 64 * 64 * 8);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d);
  std::vector<double> h(64 * 64 * 64);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  // for every (la, lb) list the D lanes that received 1
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d pairs with:", la);
    for (int lb = 0; lb < 64; ++lb)
      for (int ld = 0; ld < 64; ++ld)
        if (h[((size_t)la * 64 + lb) * 64 + ld] != 0.0) printf(" (B%d->D%d)", lb, ld);
    printf("\n");
  }
  return 0;
}
