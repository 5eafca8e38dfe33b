// General-size symmetric eigenvalue clip on the GPU: out = A + V diag(max(tol - lambda, 0)) V'   (reference: sqp_method.py:327-403,
// `Sqp.__regularize_hessian`: eigenvalues below `regularization_tol` of the (reduced) Hessian are lifted to it).
//
// Eigen-decomposition by one-sided (Hestenes) Jacobi on B = A + s I, s = 1.5 x the Gershgorin bound: B is positive definite with
// condition <= 5, so its singular vectors ARE the eigenvectors of A (no +-sigma mixing of an indefinite matrix) and
// lambda_i = |u_i| - s.  The n vectors u_i (columns of B V, stored contiguously) are rotated in pairs until mutually orthogonal;
// a round of the round-robin tournament holds n/2 disjoint pairs = n/2 workgroups, each one streaming its two u's and two v's
// once (coalesced), so the method is HBM/L2-bound vector work: 3 dot products and 4 axpys per pair, ~10 sweeps.
// This path is sequential inside the SQP loop of the reference (one matrix per call), hence a batch dimension but no further tuning.
#pragma once
#include "tmpc_common.h"

namespace tmpc {

__device__ __forceinline__ double eig_block_sum(double v, double* red) {      // 256 threads
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// shift[b] = 1.5 * max_i sum_j |A_ij| (+ tiny), U = A + shift I, V = I
__global__ void __launch_bounds__(256) k_eig_init(const double* A, double* U, double* V, double* shift, int n) {
  __shared__ double red[4];
  __shared__ double smax;
  const int b = blockIdx.x;
  const double* Ab = A + (size_t)b * n * n;
  double gmax = 0.0;
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (int j = threadIdx.x; j < n; j += 256) s += fabs(0.5 * (Ab[(size_t)i * n + j] + Ab[(size_t)j * n + i]));
    s = eig_block_sum(s, red);
    gmax = fmax(gmax, s);
  }
  if (threadIdx.x == 0) { smax = 1.5 * gmax + 1e-300; shift[b] = smax; }
  __syncthreads();
  const double sh = smax;
  for (size_t e = threadIdx.x; e < (size_t)n * n; e += 256) {
    const size_t i = e / n, j = e - i * n;
    U[(size_t)b * n * n + e] = 0.5 * (Ab[e] + Ab[j * n + i]) + (i == j ? sh : 0.0);
    V[(size_t)b * n * n + e] = (i == j) ? 1.0 : 0.0;
  }
}

// one round of the tournament: workgroup k rotates the pair (pa, pb) of round r (m = n rounded up to even players, player m-1 fixed)
__global__ void __launch_bounds__(256) k_eig_round(double* U, double* V, int n, int m, int r, double* offmax, double thr) {
  __shared__ double red[4];
  const int k = blockIdx.x, b = blockIdx.y;
  int pa, pb;
  if (k == 0) { pa = m - 1; pb = r % (m - 1); }
  else { pa = (r + k) % (m - 1); pb = (r - k + 2 * (m - 1)) % (m - 1); }
  if (pa >= n || pb >= n) return;
  double* up = U + ((size_t)b * n + pa) * n; double* uq = U + ((size_t)b * n + pb) * n;
  double* vp = V + ((size_t)b * n + pa) * n; double* vq = V + ((size_t)b * n + pb) * n;
  double al = 0.0, be = 0.0, ga = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) { const double x = up[j], y = uq[j]; al = fma(x, x, al); be = fma(y, y, be); ga = fma(x, y, ga); }
  al = eig_block_sum(al, red); be = eig_block_sum(be, red); ga = eig_block_sum(ga, red);
  const double den = sqrt(al) * sqrt(be);                        // (the product al * be itself underflows for tiny-norm inputs)
  const double rel = (den > 0.0) ? fabs(ga) / den : 0.0;
  if (threadIdx.x == 0) atomicMax((unsigned long long*)(offmax + b), (unsigned long long)__double_as_longlong(rel));     // rel >= 0: ordered as integers
  if (!(rel > thr)) return;                                      // thr ~ n eps: the rounding level of the n-long dot products
  const double zeta = (be - al) / (2.0 * ga);
  const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
  for (int j = threadIdx.x; j < n; j += 256) {
    const double x = up[j], y = uq[j];
    up[j] = c * x - s * y; uq[j] = s * x + c * y;
    const double vx = vp[j], vy = vq[j];
    vp[j] = c * vx - s * vy; vq[j] = s * vx + c * vy;
  }
}

// lambda_i = |u_i| - shift, lift_i = max(tol - lambda_i, 0); reg[b] = max_i lift_i   (grid: (n, nb))
__global__ void __launch_bounds__(256) k_eig_values(const double* U, const double* shift, double* evals, double* lift, double* reg, int n, double tol) {
  __shared__ double red[4];
  const int i = blockIdx.x, b = blockIdx.y;
  const double* u = U + ((size_t)b * n + i) * n;
  double s = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) s = fma(u[j], u[j], s);
  s = eig_block_sum(s, red);
  if (threadIdx.x == 0) {
    const double lam = sqrt(s) - shift[b];
    const double d = (lam < tol) ? tol - lam : 0.0;
    evals[(size_t)b * n + i] = lam; lift[(size_t)b * n + i] = d;
    atomicMax((unsigned long long*)(reg + b), (unsigned long long)__double_as_longlong(d));
  }
}

// out = sym(A) + sum_i lift_i v_i v_i'      (grid: (ceil(n*n/256), nb))
__global__ void __launch_bounds__(256) k_eig_apply(const double* A, const double* V, const double* lift, double* out, int n) {
  const int b = blockIdx.y;
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)n * n) return;
  const size_t i = e / n, j = e - i * n;
  const double* Ab = A + (size_t)b * n * n;
  double acc = 0.5 * (Ab[e] + Ab[j * n + i]);
  const double* Vb = V + (size_t)b * n * n;
  const double* lb = lift + (size_t)b * n;
  for (int q = 0; q < n; ++q) {
    const double d = lb[q];
    if (d != 0.0) acc = fma(d * Vb[(size_t)q * n + i], Vb[(size_t)q * n + j], acc);
  }
  out[(size_t)b * n * n + e] = acc;
}

}  // namespace tmpc
