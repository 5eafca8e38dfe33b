// Small blocks (dp = 16, i.e. nx <= 5 without stage-local multipliers): the whole cyclic-reduction factorisation of a problem, and a whole
// forward + backward substitution, as ONE kernel each with one workgroup per problem.
//
// "small configs (n = 5, p = 30, batch = 1) are pure latency".  With 16 x 16 blocks a level of tmpc_cr.h is a few
// microseconds of work, so the level-by-level launch sequence of the batched tile kernels (3 launches per level to factor, 3 to solve:
// ~1500 launches per solve at p = 30) costs the unicycle-shaped problem 11 ms of launch latency for 7 ms of kernels that are themselves
// mostly launch ramp (profiles/r3_small_config_kernel_stats.txt).  Here the levels are a loop inside the kernel with workgroup
// barriers in between; the sixteen waves of the workgroup take the eliminated nodes / update items of a level in turn.  The factor is stored
// exactly as the batched kernels store it (L in the lower triangle of D, inverted diagonal tile in Linv, O and fill blocks in their
// edge slots), so everything else -- chord steps, the border solve, the debug entries -- is unchanged, and a result computed on this path
// differs from the batched one only by rounding (same operations, 16 x 16 MFMA products instead of 64 x 64 tiles).
#pragma once
#include "tmpc_common.h"
#include "tmpc_factor.h"
#include "tmpc_cr.h"

namespace tmpc {

constexpr int CRS_MAXLEV = 24;           // levels of the schedule passed by value (p <= CRS_PMAX needs at most log2(p) + 4)
constexpr int CRS_PMAX = 160;            // right-hand sides of a whole problem live in LDS (vectors + scratch: 2 * p * 16 * 3 doubles = 123 KB at p = 160)
struct CrLevs { int n; int v[4 * CRS_MAXLEV]; };          // (eoff, nelim, uoff, nupd) per level

constexpr int CRS_NW = 16;               // waves per workgroup: a level of p = 30 (15 eliminated nodes) is one round; 1024 threads
constexpr int CRS_NT = 64 * CRS_NW;
constexpr int CRS_LD = 17;               // leading dimension of the 16 x 16 images in LDS
constexpr int crs_factor_lds_doubles() { return CRS_NW * (2 * 16 * CRS_LD + 8); }      // per wave: S, Si, two statistics
// (body: the CRS_NT threads of one workgroup factor problem b -- the kernel below and the persistent kernel of tmpc_persist.h)
__device__ __forceinline__ void cr_small_factor_body(const WS& w, const Dims& dm, const CrDev& cr, const CrLevs& lv, int b, double* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int dp = 16;
  const size_t bs = (size_t)dp * dp;
  double* S = lds + wv * (2 * 16 * CRS_LD + 8);
  double* Si = S + 16 * CRS_LD;
  double* stat = Si + 16 * CRS_LD;                          // [0] frozen pivots, [1] smallest pivot / reference of this wave's nodes
  if (lane == 0) { stat[0] = 0.0; stat[1] = 1.0; }
  for (int l = 0; l < lv.n; ++l) {
    const int eoff = lv.v[4 * l], nelim = lv.v[4 * l + 1], uoff = lv.v[4 * l + 2], nupd = lv.v[4 * l + 3];
    // ---- eliminated nodes: D_i = L_i L_i', L_i^-1, and O_x <- T[x,i] L_i^-T for the (up to) two neighbours; one wave per node
    for (int j = wv; j < nelim; j += CRS_NW) {
      const int* er = cr.elim + (size_t)(eoff + j) * CR_EW;
      const int node = er[CE_NODE];
      double* Dk = w.D + ((size_t)b * dm.p + node) * bs;
      double* Li = w.Linv + ((size_t)b * dm.p + node) * dm.nt * TB * TB;
      const double* dref = w.Ddiag + ((size_t)b * dm.p + node) * dp;
      for (int e = lane; e < 256; e += 64) { const int i = e >> 4, c = e & 15; S[i * CRS_LD + c] = Dk[e]; Si[i * CRS_LD + c] = 0.0; }
      wave_lds_sync();
      wave_potrf16<CRS_LD>(S, Si, dref, stat, lane);
      for (int e = lane; e < 256; e += 64) {
        const int i = e >> 4, c = e & 15;
        if (c <= i) Dk[e] = S[i * CRS_LD + c];
        Li[i * TB + c] = Si[i * CRS_LD + c];
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int slot = er[q ? CE_EB : CE_EA];
        if (slot < 0) continue;
        double* X = cr_edge(w, dm, b, slot);
        double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
        acc = mm16(X, dp, Si, 1, CRS_LD, acc, 1.0, lane);     // (X L^-T)[r][c] = sum_k X[r][k] Linv[c][k]
        store_d16(X, dp, acc, lane);
      }
    }
    __threadfence_block();
    __syncthreads();
    // ---- surviving neighbours D_s -= O_s O_s' (one or two eliminated neighbours) and the fill edges T[x,y] (=|-=) -O_x O_y'
    for (int it = wv; it < nupd + nelim; it += CRS_NW) {
      if (it < nupd) {
        const int* ur = cr.upd + (size_t)(uoff + it) * CR_UW;
        double* C = w.D + ((size_t)b * dm.p + ur[CU_NODE]) * bs;
        const double* O0 = cr_edge(w, dm, b, ur[CU_E0]);
        double4_t acc = load_d16(C, dp, lane);
        acc = mm16(O0, dp, O0, 1, dp, acc, -1.0, lane);
        if (ur[CU_E1] >= 0) { const double* O1 = cr_edge(w, dm, b, ur[CU_E1]); acc = mm16(O1, dp, O1, 1, dp, acc, -1.0, lane); }
        store_d16(C, dp, acc, lane);
      } else {
        const int* er = cr.elim + (size_t)(eoff + it - nupd) * CR_EW;
        if (er[CE_FILL] < 0) continue;
        const double* Ox = cr_edge(w, dm, b, er[CE_FX] ? er[CE_EB] : er[CE_EA]);
        const double* Oy = cr_edge(w, dm, b, er[CE_FX] ? er[CE_EA] : er[CE_EB]);
        double* C = cr_edge(w, dm, b, er[CE_FILL]);
        double4_t acc = er[CE_FACC] ? load_d16(C, dp, lane) : (double4_t){0.0, 0.0, 0.0, 0.0};
        acc = mm16(Ox, dp, Oy, 1, dp, acc, -1.0, lane);
        store_d16(C, dp, acc, lane);
      }
    }
    __threadfence_block();
    __syncthreads();
  }
  if (lane == 0) {
    const int nbad = (int)stat[0];
    if (nbad) atomicAdd(w.iprob + (size_t)b * IS + I_NSHIFT, nbad);
    if (w.prob && stat[1] < 1.0) atomic_min_pos(w.prob + (size_t)b * PS + P_MINPIV, stat[1]);
  }
}
__global__ void __launch_bounds__(CRS_NT) k_cr_small_factor(WS w, Dims dm, CrDev cr, CrLevs lv) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  cr_small_factor_body(w, dm, cr, lv, cr.alist[blockIdx.x], lds);
}

// Forward and backward substitution with that factor, right-hand sides of `pass` (cr_nc / cr_rhs of tmpc_cr.h), all levels in one kernel.
// The vectors of the whole problem sit in LDS ([p][16][nc], as in memory); every (node, row, right-hand side) entry of a level is one
// thread's 16-long dot product against a block read from L2.
constexpr int crs_solve_lds_doubles(int p) { return 2 * p * 16 * 3 + 16; }
__device__ __forceinline__ void cr_small_solve_body(const WS& w, const Dims& dm, const CrDev& cr, const CrLevs& lv, int b, double* lds, int pass) {
  const int nc = cr_nc(w, b, pass);
  if (nc == 0) return;
  const int tid = threadIdx.x;
  constexpr int dp = 16;
  const int p = dm.p, nz = p * dp * nc;
  double* z = lds;                                        // [p][16][nc]
  double* t = lds + (size_t)p * dp * 3;                   // scratch of the same shape
  double* R = cr_rhs(w, dm, b, 0, nc);
  for (int e = tid; e < nz; e += CRS_NT) z[e] = R[e];
  __syncthreads();
  // ---- forward, level by level: z_i <- L_i^-1 z_i for the eliminated nodes, then z_s -= O_s z_i for their surviving neighbours
  for (int l = 0; l < lv.n; ++l) {
    const int eoff = lv.v[4 * l], nelim = lv.v[4 * l + 1], uoff = lv.v[4 * l + 2], nupd = lv.v[4 * l + 3];
    for (int e = tid; e < nelim * dp * nc; e += CRS_NT) {
      const int j = e / (dp * nc), rq = e - j * (dp * nc), r = rq / nc, q = rq - r * nc;
      const int node = cr.elim[(size_t)(eoff + j) * CR_EW + CE_NODE];
      const double* Li = w.Linv + ((size_t)b * p + node) * dm.nt * TB * TB + r * TB;
      const double* zi = z + (size_t)node * dp * nc + q;
      // (all sixteen terms: the entries of L^-1 above the diagonal are exact zeros, and with a fixed trip count the sixteen loads of the
      // row leave together instead of one L2 round trip per term -- the loop bound r made this the longest phase of the kernel)
      double li[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) li[k] = Li[k];
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc = fma(li[k], zi[k * nc], acc);
      t[e] = acc;
    }
    __syncthreads();
    for (int e = tid; e < nelim * dp * nc; e += CRS_NT) {
      const int j = e / (dp * nc), rq = e - j * (dp * nc);
      const int node = cr.elim[(size_t)(eoff + j) * CR_EW + CE_NODE];
      z[(size_t)node * dp * nc + rq] = t[e];
    }
    __syncthreads();
    for (int e = tid; e < nupd * dp * nc; e += CRS_NT) {
      const int j = e / (dp * nc), rq = e - j * (dp * nc), r = rq / nc, q = rq - r * nc;
      const int* ur = cr.upd + (size_t)(uoff + j) * CR_UW;
      double acc = z[(size_t)ur[CU_NODE] * dp * nc + rq];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int slot = ur[s ? CU_E1 : CU_E0];
        if (slot < 0) break;
        const double* O = cr_edge(w, dm, b, slot) + r * dp;
        const double* zi = z + (size_t)ur[s ? CU_S1 : CU_S0] * dp * nc + q;
        double ov[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) ov[k] = O[k];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fma(-ov[k], zi[k * nc], acc);
      }
      z[(size_t)ur[CU_NODE] * dp * nc + rq] = acc;         // (an entry of a surviving node: nobody else reads or writes it in this phase)
    }
    __syncthreads();
  }
  // ---- backward, levels in reverse: z_i <- L_i^-T (z_i - O_a' z_a - O_b' z_b)
  for (int l = lv.n - 1; l >= 0; --l) {
    const int eoff = lv.v[4 * l], nelim = lv.v[4 * l + 1];
    for (int e = tid; e < nelim * dp * nc; e += CRS_NT) {
      const int j = e / (dp * nc), rq = e - j * (dp * nc), r = rq / nc, q = rq - r * nc;
      const int* er = cr.elim + (size_t)(eoff + j) * CR_EW;
      double acc = z[(size_t)er[CE_NODE] * dp * nc + rq];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int slot = er[s ? CE_EB : CE_EA];
        if (slot < 0) continue;
        const double* O = cr_edge(w, dm, b, slot) + r;     // column r of O_x
        const double* zx = z + (size_t)er[s ? CE_NB : CE_NA] * dp * nc + q;
        double ov[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) ov[k] = O[k * dp];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fma(-ov[k], zx[k * nc], acc);
      }
      t[e] = acc;
    }
    __syncthreads();
    for (int e = tid; e < nelim * dp * nc; e += CRS_NT) {
      const int j = e / (dp * nc), rq = e - j * (dp * nc), r = rq / nc, q = rq - r * nc;
      const int node = cr.elim[(size_t)(eoff + j) * CR_EW + CE_NODE];
      const double* Li = w.Linv + ((size_t)b * p + node) * dm.nt * TB * TB + r;      // column r of L^-1
      const double* tj = t + (size_t)j * dp * nc + q;
      double li[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) li[k] = Li[k * TB];       // (zeros for k < r, see the forward sweep)
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc = fma(li[k], tj[k * nc], acc);
      z[(size_t)node * dp * nc + rq] = acc;
    }
    __syncthreads();
  }
  for (int e = tid; e < nz; e += CRS_NT) R[e] = z[e];
}
__global__ void __launch_bounds__(CRS_NT) k_cr_small_solve(WS w, Dims dm, CrDev cr, CrLevs lv, int pass) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  cr_small_solve_body(w, dm, cr, lv, cr.alist[blockIdx.x], lds, pass);
}

}  // namespace tmpc
