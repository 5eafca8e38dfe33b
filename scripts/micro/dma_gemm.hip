// fp64 tile GEMM core fed by LDS-DMA (buffer_load_dwordx4 ... lds): does a deeper, register-free operand pipeline lift the batched
// symmetric update above the ~45 TFLOP/s of the register-staged core (tmpc_factor.h: global -> VGPR -> ds_write -> barrier, one
// 16-column slab of prefetch)?  Workload = k_cr_update's: nb blocks O (dp x dp), D -= O O' on the 15 lower 64 x 64 tiles.
//   * slab = 64 rows x 16 columns per operand, unpadded [row][8 granules of 16 B]; granule c of row r holds the K pair c ^ s(r),
//     s(r) = (r & 7) ^ 2*((r >> 3) & 1): the DMA writes lane-linear (8 lanes = one 128-byte row), the swizzle sits in the per-lane
//     GLOBAL address; fragment reads are ds_read_b128 (one K pair per lane), conflict-free for the A rows (8 consecutive) and the
//     B rows (stride 4)
//   * v_mfma_f64_4x4x4_4b consumes 4 k per instruction, one per lane group fk: instruction 1 takes the even k of the four pairs
//     (lane's .x), instruction 2 the odd ones (.y) -- A and B agree, so any assignment of k to the slots is a valid product
//   * DEPTH LDS buffers, slab t+DEPTH-1 is issued right after the barrier of step t, s_waitcnt vmcnt(N) keeps the younger slabs in flight
// Build: hipcc --offload-arch=gfx950 -O3 dma_gemm.hip -o dma_gemm ; run: ./dma_gemm [nblocks]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "../../tunempc_amd/csrc/tmpc_factor.h"

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;

template <int DEPTH, int OCC>
__global__ void __launch_bounds__(256, OCC) k_upd(const double* O, double* Dm, int dp, int nitems) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int i8 = blockIdx.x, per = gridDim.x >> 3;
  const int item = (i8 & 7) * per + (i8 >> 3);               // XCD-contiguous item runs
  if (item >= nitems) return;
  const int b = item / 15;
  int t = item % 15, tm = 0;
  while (t > tm) { t -= tm + 1; ++tm; }
  const int tn = t, m0 = tm * 64, n0 = tn * 64;
  const int M = dp - m0 < 64 ? dp - m0 : 64, N = dp - n0 < 64 ? dp - n0 : 64, K = dp;
  const double* A = O + (size_t)b * dp * dp + (size_t)m0 * dp;
  const double* B = O + (size_t)b * dp * dp + (size_t)n0 * dp;
  double* C = Dm + (size_t)b * dp * dp + (size_t)m0 * dp + n0;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wr0 = (wv >> 1) * 32, wc0 = (wv & 1) * 32;
  const int fk = lane >> 4, fq = (lane >> 2) & 3, fj = lane & 3;
  const bool wave_on = (wr0 < M) && (wc0 < N) && !(tm == tn && wc0 >= wr0 + 32);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(((unsigned)(M - 1) * (unsigned)dp + (unsigned)K) * 8u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(((unsigned)(N - 1) * (unsigned)dp + (unsigned)K) * 8u), 0x00020000);
  // DMA: wave wv moves rows 16 wv .. 16 wv + 15 of both operands, two instructions of 8 rows each
  unsigned vo[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = wv * 16 + 8 * h + (lane >> 3), c = lane & 7;
    const int sw = (row & 7) ^ (((row >> 3) & 1) << 1);
    vo[h] = (unsigned)(row * dp + 2 * (c ^ sw)) * 8u;
  }
  // fragment read offsets (doubles) inside a slab
  int oa[2], ob[4][2];
  {
    const int swa = ((4 * fq + fj) & 7) ^ ((fq >> 1) << 1);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) oa[hh] = (wr0 + 4 * fq + fj) * 16 + 2 * ((4 * hh + fk) ^ swa);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int swb = ((4 * fj + e) & 7) ^ ((fj >> 1) << 1);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) ob[e][hh] = (wc0 + 4 * fj + e) * 16 + 2 * ((4 * hh + fk) ^ swb);
    }
  }
  double acc[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[i][c] = 0.0;
  const int nks = K / 16;
#define ISSUE(S, Q)                                                                                       \
  {                                                                                                       \
    double* As_ = lds + (Q) * 2048;                                                                       \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                       \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(As_ + (wv * 16 + 8 * h) * 16), 16, vo[h], (S) * 128, 0, 0); \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(As_ + 1024 + (wv * 16 + 8 * h) * 16), 16, vo[h], (S) * 128, 0, 0); \
    }                                                                                                     \
  }
#define STEP(S, Q)                                                                                        \
  {                                                                                                       \
    if ((S) + DEPTH - 2 < nks - 1) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (DEPTH - 2)) : "memory"); }   \
    else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }                                             \
    __builtin_amdgcn_s_barrier();                                                                         \
    if ((S) + DEPTH - 1 < nks) ISSUE((S) + DEPTH - 1, ((Q) + DEPTH - 1) % DEPTH)                          \
    if (wave_on) {                                                                                        \
      const double* As_ = lds + (Q) * 2048;                                                               \
      const double* Bs_ = As_ + 1024;                                                                     \
      _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                                  \
        double2_t a[2];                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) a[i] = *(const double2_t*)(As_ + oa[hh] + i * 256); \
        _Pragma("unroll") for (int cb = 0; cb < 8; ++cb) {                                                \
          const double2_t bv = *(const double2_t*)(Bs_ + ob[cb & 3][hh] + (cb >> 2) * 256);               \
          _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                 \
            acc[i][cb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i][0], bv[0], acc[i][cb], 0, 0, 0);         \
            acc[i][cb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i][1], bv[1], acc[i][cb], 0, 0, 0);         \
          }                                                                                               \
        }                                                                                                 \
      }                                                                                                   \
    }                                                                                                     \
  }
  // prologue: slabs 0 .. DEPTH-2
#pragma unroll
  for (int s = 0; s < DEPTH - 1; ++s)
    if (s < nks) ISSUE(s, s)
  for (int s = 0; s < nks; s += DEPTH) {
#pragma unroll
    for (int q = 0; q < DEPTH; ++q)
      if (s + q < nks) STEP(s + q, q)
  }
  if (wave_on) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rbase = wr0 + i * 16;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int cbase = wc0 + m * 16;
        if (rbase < M && cbase < N) {
          double2_t* cp = (double2_t*)(C + (size_t)(rbase + 4 * fq + fk) * dp + cbase + 4 * fj);
          double2_t u0 = cp[0], u1 = cp[1];
          u0[0] -= acc[i][4 * m]; u0[1] -= acc[i][4 * m + 1]; u1[0] -= acc[i][4 * m + 2]; u1[1] -= acc[i][4 * m + 3];
          cp[0] = u0; cp[1] = u1;
        }
      }
    }
  }
}

template <int DEPTH, int OCC>
__global__ void __launch_bounds__(256, OCC) k_upd14(const double* O, double* Dm, int dp, int nitems) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int i8 = blockIdx.x, per = gridDim.x >> 3;
  const int item = (i8 & 7) * per + (i8 >> 3);               // XCD-contiguous item runs
  if (item >= nitems) return;
  const int b = item / 15;
  int t = item % 15, tm = 0;
  while (t > tm) { t -= tm + 1; ++tm; }
  const int tn = t, m0 = tm * 64, n0 = tn * 64;
  const int M = dp - m0 < 64 ? dp - m0 : 64, N = dp - n0 < 64 ? dp - n0 : 64, K = dp;
  const double* A = O + (size_t)b * dp * dp + (size_t)m0 * dp;
  const double* B = O + (size_t)b * dp * dp + (size_t)n0 * dp;
  double* C = Dm + (size_t)b * dp * dp + (size_t)m0 * dp + n0;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wr0 = 0, wc0 = ((wv + item) & 3) * 16;        // wave = 16-column strip; the strip order rotates with the item so that no SIMD always gets the short strip of a diagonal tile
  const int fk = lane >> 4, fq = (lane >> 2) & 3, fj = lane & 3;
  const bool wave_on = (wc0 < N);
  const int i0 = (tm == tn) ? (wc0 >> 4) : 0;               // diagonal tile: row fragments above the strip's first column are not needed
  const int i1 = (M + 15) >> 4;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(((unsigned)(M - 1) * (unsigned)dp + (unsigned)K) * 8u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(((unsigned)(N - 1) * (unsigned)dp + (unsigned)K) * 8u), 0x00020000);
  // DMA: wave wv moves rows 16 wv .. 16 wv + 15 of both operands, two instructions of 8 rows each
  unsigned vo[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = wv * 16 + 8 * h + (lane >> 3), c = lane & 7;
    const int sw = (row & 7) ^ (((row >> 3) & 1) << 1);
    vo[h] = (unsigned)(row * dp + 2 * (c ^ sw)) * 8u;
  }
  // fragment read offsets (doubles) inside a slab
  int oa[2], ob[4][2];
  {
    const int swa = ((4 * fq + fj) & 7) ^ ((fq >> 1) << 1);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) oa[hh] = (wr0 + 4 * fq + fj) * 16 + 2 * ((4 * hh + fk) ^ swa);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int swb = ((4 * fj + e) & 7) ^ ((fj >> 1) << 1);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) ob[e][hh] = (wc0 + 4 * fj + e) * 16 + 2 * ((4 * hh + fk) ^ swb);
    }
  }
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[i][c] = 0.0;
  const int nks = K / 16;
#define ISSUE(S, Q)                                                                                       \
  {                                                                                                       \
    double* As_ = lds + (Q) * 2048;                                                                       \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                       \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(As_ + (wv * 16 + 8 * h) * 16), 16, vo[h], (S) * 128, 0, 0); \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(As_ + 1024 + (wv * 16 + 8 * h) * 16), 16, vo[h], (S) * 128, 0, 0); \
    }                                                                                                     \
  }
#define STEP(S, Q)                                                                                        \
  {                                                                                                       \
    if ((S) + DEPTH - 2 < nks - 1) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (DEPTH - 2)) : "memory"); }   \
    else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }                                             \
    __builtin_amdgcn_s_barrier();                                                                         \
    if ((S) + DEPTH - 1 < nks) ISSUE((S) + DEPTH - 1, ((Q) + DEPTH - 1) % DEPTH)                          \
    if (wave_on) {                                                                                        \
      const double* As_ = lds + (Q) * 2048;                                                               \
      const double* Bs_ = As_ + 1024;                                                                     \
      _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                                  \
        double2_t a[4];                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) a[i] = *(const double2_t*)(As_ + oa[hh] + i * 256); \
        _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) {                                                \
          const double2_t bv = *(const double2_t*)(Bs_ + ob[cb][hh]);                                     \
          _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                 \
            if (i >= i0 && i < i1) {                                                                      \
            acc[i][cb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i][0], bv[0], acc[i][cb], 0, 0, 0);         \
            acc[i][cb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i][1], bv[1], acc[i][cb], 0, 0, 0);         \
            }                                                                                             \
          }                                                                                               \
        }                                                                                                 \
      }                                                                                                   \
    }                                                                                                     \
  }
  // prologue: slabs 0 .. DEPTH-2
#pragma unroll
  for (int s = 0; s < DEPTH - 1; ++s)
    if (s < nks) ISSUE(s, s)
  for (int s = 0; s < nks; s += DEPTH) {
#pragma unroll
    for (int q = 0; q < DEPTH; ++q)
      if (s + q < nks) STEP(s + q, q)
  }
  if (wave_on) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rbase = 16 * i;
      if (i >= i0 && rbase < M) {
        double2_t* cp = (double2_t*)(C + (size_t)(rbase + 4 * fq + fk) * dp + wc0 + 4 * fj);
        double2_t u0 = cp[0], u1 = cp[1];
        u0[0] -= acc[i][0]; u0[1] -= acc[i][1]; u1[0] -= acc[i][2]; u1[1] -= acc[i][3];
        cp[0] = u0; cp[1] = u1;
      }
    }
  }
}

// the register-staged core of the library on the same items
template <int NS>
__global__ void __launch_bounds__(256, NS == 1 ? 4 : 2) k_upd_old(const double* O, double* Dm, int dp, int nitems) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int i8 = blockIdx.x, per = gridDim.x >> 3;
  const int item = (i8 & 7) * per + (i8 >> 3);
  if (item >= nitems) return;
  const int b = item / 15;
  int t = item % 15, tm = 0;
  while (t > tm) { t -= tm + 1; ++tm; }
  const int tn = t, m0 = tm * 64, n0 = tn * 64;
  const int M = dp - m0 < 64 ? dp - m0 : 64, N = dp - n0 < 64 ? dp - n0 : 64;
  const double* Ob = O + (size_t)b * dp * dp;
  tmpc::wg_gemm_nt<true, 2, 2, 2, NS>(Dm + (size_t)b * dp * dp + (size_t)m0 * dp + n0, dp, Ob + (size_t)m0 * dp, dp, Ob + (size_t)n0 * dp, dp, M, N, dp, tmpc::GM_SUB, tm == tn, lds);
}
template <int NS>
static void run_old(const double* dO, double* dD, int dp, int nb) {
  const int nitems = nb * 15, grid = (nitems + 7) / 8 * 8;
  const size_t ldsb = (size_t)tmpc::GemmCfg<2, 2, 2, NS>::LDS_DOUBLES * 8;
  hipFuncSetAttribute((const void*)k_upd_old<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((k_upd_old<NS>), dim3(grid), dim3(256), ldsb, 0, dO, dD, dp, nitems);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 5;
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_upd_old<NS>), dim3(grid), dim3(256), ldsb, 0, dO, dD, dp, nitems);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double alg = (double)nb * dp * (double)dp * dp, exe = (double)nb * 55.0 * 32 * 32 * dp * 2.0;
  printf("register-staged core, NS = %d: %8.3f ms  %6.2f TFLOP/s algorithmic (%5.1f executed)\n", NS, ms, alg / ms / 1e9, exe / ms / 1e9);
}

template <int DEPTH, int OCC, int LAY = 0>
static void run(const double* dO, double* dD, int dp, int nb, const std::vector<double>& hO, const std::vector<double>& hD0) {
  const int nitems = nb * 15, grid = (nitems + 7) / 8 * 8;
  const size_t ldsb = (size_t)DEPTH * 2048 * 8;
  auto kern = LAY ? k_upd14<DEPTH, OCC> : k_upd<DEPTH, OCC>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipMemcpy(dD, hD0.data(), hD0.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), ldsb, 0, dO, dD, dp, nitems);
  hipDeviceSynchronize();
  // check block 0 and the last block (lower triangle)
  std::vector<double> h((size_t)dp * dp);
  double worst = 0.0;
  for (int b : {0, nb - 1}) {
    hipMemcpy(h.data(), dD + (size_t)b * dp * dp, h.size() * 8, hipMemcpyDeviceToHost);
    const double* Ob = hO.data() + (size_t)b * dp * dp;
    for (int i = 0; i < dp; i += 7)
      for (int j = 0; j <= i; j += 3) {
        double s = hD0[(size_t)b * dp * dp + (size_t)i * dp + j];
        for (int k = 0; k < dp; ++k) s -= Ob[(size_t)i * dp + k] * Ob[(size_t)j * dp + k];
        worst = fmax(worst, fabs(s - h[(size_t)i * dp + j]));
      }
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 5;
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), ldsb, 0, dO, dD, dp, nitems);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double alg = (double)nb * dp * (double)dp * dp;            // syrk, lower: n^2 k
  const double exe = (double)nb * 55.0 * 32 * 32 * dp * 2.0;     // 10 off-diagonal tiles x 4 waves + 5 diagonal tiles x 3 waves, 32 x 32 x dp each
  printf("%s depth %d, occupancy %d: %8.3f ms  %6.2f TFLOP/s algorithmic (%5.1f executed)  max abs err %.2e\n", LAY ? "64x16 strips" : "32x32 quads ", DEPTH, OCC, ms, alg / ms / 1e9, exe / ms / 1e9, worst);
}

int main(int argc, char** argv) {
  const int dp = 304, nb = argc > 1 ? atoi(argv[1]) : 4096;
  const size_t n = (size_t)nb * dp * dp;
  std::vector<double> hO(n), hD(n);
  unsigned long long s = 88172645463325252ull;
  for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hO[i] = (double)(s >> 11) / 9007199254740992.0 - 0.5; hD[i] = (double)(i % 97) * 0.01; }
  double *dO, *dD;
  hipMalloc(&dO, n * 8); hipMalloc(&dD, n * 8);
  hipMemcpy(dO, hO.data(), n * 8, hipMemcpyHostToDevice);
  run_old<1>(dO, dD, dp, nb);
  run_old<2>(dO, dD, dp, nb);
  run<2, 4>(dO, dD, dp, nb, hO, hD);
  run<3, 3>(dO, dD, dp, nb, hO, hD);
  run<4, 2>(dO, dD, dp, nb, hO, hD);
  run<3, 2>(dO, dD, dp, nb, hO, hD);
  run<2, 5>(dO, dD, dp, nb, hO, hD);
  run<2, 4, 1>(dO, dD, dp, nb, hO, hD);
  run<2, 5, 1>(dO, dD, dp, nb, hO, hD);
  run<3, 3, 1>(dO, dD, dp, nb, hO, hD);
  run<3, 4, 1>(dO, dD, dp, nb, hO, hD);
  run<3, 4>(dO, dD, dp, nb, hO, hD);
  return 0;
}
