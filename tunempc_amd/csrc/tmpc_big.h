// Per-stage kernels for stage blocks wider than one LDS tile: 32 < n = nx + m <= 64 (round 4; the reference accepts any size,
// preprocessing.py:157-185, convexifier.py:242).
//
// The kernels of tmpc_stage.h are written around n <= 32: a stage's matrices are 32 x 33 LDS tiles, the helpers of tmpc_small.h map 64 or
// 256 threads onto 32 x 32 outputs, six tiles fit a CU three times over.  At n = 64 a tile is 33 KB and six of them do not fit the 160 KB
// of LDS at all, so this is not a parameter change.  The generic form below keeps every matrix where it already lives -- in the global
// arrays of the workspace (X_r, S_r, S_r^-1, L^-1, Rd, T, dS, dX, ...: a few hundred kB per stage, L2-resident while a workgroup works on
// them) plus five n x n scratch matrices per stage -- and runs the same formulas with one thread per output entry, one 256-thread workgroup
// per stage.  Same inputs, same outputs, same partial sums as k_init_stage / k_stage_pre / k_stage_rhs / k_stage_dir / k_eigmin /
// k_final_stage, so the control kernels, the Schur assembly, the block factorisation (blocks wider than 320: the register-staged kernels of
// tmpc_cr.h) and k_update / k_gather / k_solve_border are unchanged.  Eigenvalues (scaling, step lengths, status) by a parallel-ordered
// cyclic Jacobi iteration on one LDS tile.  Every model: the terms J' diag(phi) J of the multipliers and T_k of Step 3 are added here, and the kernels of
// tmpc_phi.h / tmpc_t3.h that touch n x n matrices have a <true> form reading them from global memory; the tight mode has its stage matrices in a global
// scratch (tmpc_dd.h: sdd_slot) and kb_polish_step below.  Roughly 5 x the stage time per entry of the tuned n <= 32 kernels, which
// matters little at these sizes: the d x d blocks (d = nx (nx + 1) / 2 up to 2016) dominate.  TMPC_DEBUG_FLAG_GENERIC_STAGE runs this
// path at n <= 32 as well, which is how it is tested against the tuned one.
#pragma once
#include "tmpc_common.h"
#include "tmpc_small.h"

namespace tmpc {

constexpr int BLD = NB + 1;                 // leading dimension of its one LDS tile (Jacobi)

__device__ __forceinline__ void gsync() { __threadfence_block(); __syncthreads(); }
// C (M x N, ldc) {=, +=, -=} A B; element (i, k) of A at A[i * ars + k * acs], (k, j) of B at B[k * brs + j * bcs].  C must not alias A or B.
__device__ __forceinline__ void gmm(double* C, int ldc, const double* A, int ars, int acs, const double* B, int brs, int bcs, int M, int N, int K, int mode) {
  for (int e = threadIdx.x; e < M * N; e += 256) {
    const int i = e / N, j = e - i * N;
    const double* ap = A + (size_t)i * ars; const double* bp = B + (size_t)j * bcs;
    double acc = 0.0;
#pragma unroll 4
    for (int k = 0; k < K; ++k) acc = fma(ap[(size_t)k * acs], bp[(size_t)k * brs], acc);
    double* p = C + (size_t)i * ldc + j;
    *p = (mode == 0) ? acc : (mode == 1 ? *p + acc : *p - acc);
  }
  gsync();
}
__device__ __forceinline__ void gsym(double* A, int n) {       // in place (A + A') / 2
  for (int e = threadIdx.x; e < n * n; e += 256) {
    const int i = e / n, j = e - i * n;
    if (j < i) { const double v = 0.5 * (A[i * n + j] + A[j * n + i]); A[i * n + j] = v; A[j * n + i] = v; }
  }
  gsync();
}
// in-place lower Cholesky (row-major n x n in global memory); a non-positive pivot is replaced by a tiny one and counted (the problem then stops)
__device__ __forceinline__ int gchol(double* A, int n, double* sflag) {
  const int tid = threadIdx.x;
  if (tid == 0) sflag[0] = 0.0;
  gsync();
  for (int j = 0; j < n; ++j) {
    double piv = A[j * n + j];
    if (!(piv > 0.0)) { piv = 1e-30; if (tid == 0) sflag[0] += 1.0; }
    const double r = sqrt(piv);
    gsync();
    if (tid == 0) A[j * n + j] = r;
    for (int i = j + 1 + tid; i < n; i += 256) A[i * n + j] /= r;
    gsync();
    const int m = n - j - 1;
    for (int e = tid; e < m * m; e += 256) {
      const int i = j + 1 + e / m, k = j + 1 + e % m;
      if (k <= i) A[i * n + k] = fma(-A[i * n + j], A[k * n + j], A[i * n + k]);
    }
    gsync();
  }
  return (int)sflag[0];
}
// Li = L^-1 (lower triangular, column c by thread c), upper part zero
__device__ __forceinline__ void gtri_inv(double* Li, const double* L, int n) {
  const int c = threadIdx.x;
  if (c < n) {
    for (int i = 0; i < c; ++i) Li[i * n + c] = 0.0;
    for (int i = c; i < n; ++i) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = c; k < i; ++k) s = fma(-L[i * n + k], Li[k * n + c], s);
      Li[i * n + c] = s / L[i * n + i];
    }
  }
  gsync();
}
// M = coef * Hb + V' Pn V - E' Pk E  (tq: n x nx scratch)
__device__ __forceinline__ void gbuild_M(double* M, double* tq, const double* V, const double* Hb, const double* Pk, const double* Pn, double coef, int n, int nx) {
  gmm(tq, nx, V, 1, n, Pn, nx, 1, n, nx, nx, 0);                // V' Pn   (n x nx)
  gmm(M, n, tq, nx, 1, V, n, 1, n, n, nx, 0);                   // (V' Pn) V
  for (int e = threadIdx.x; e < n * n; e += 256) {
    const int i = e / n, j = e - i * n;
    double v = M[e] + (coef != 0.0 ? coef * Hb[e] : 0.0);
    if (i < nx && j < nx) v -= Pk[i * nx + j];
    M[e] = v;
  }
  gsync();
}
// M (n x n, global) += scale * sum_i coef[i] g_i g_i'   (the multiplier term of [G_k; C_k]; add_gtg of tmpc_stage.h)
__device__ __forceinline__ void gadd_gtg(double* M, const double* Gg, const double* coef, double scale, int nrow, int n) {
  for (int e = threadIdx.x; e < n * n; e += 256) {
    const int i = e / n, j = e - i * n;
    double acc = 0.0;
    for (int r = 0; r < nrow; ++r) acc = fma(scale * coef[r] * Gg[r * n + i], Gg[r * n + j], acc);
    M[e] += acc;
  }
  gsync();
}
// M (n x n, global) += scale * smat(theta): the regularisation T_k of Step 3 (add_smat_t3 of tmpc_stage.h)
__device__ __forceinline__ void gadd_smat(double* M, const double* th, double scale, int n) {
  for (int e = threadIdx.x; e < n * n; e += 256) {
    const int i = e / n, j = e - i * n;
    const int a = i < j ? i : j, b = i < j ? j : i;
    M[e] += scale * th[a * n - a * (a - 1) / 2 + (b - a)];
  }
  gsync();
}
// out (nx x nx) = V G V'   (tq: nx x n scratch)
__device__ __forceinline__ void gadj_V(double* out, double* tq, const double* V, const double* G, int n, int nx) {
  gmm(tq, n, V, n, 1, G, n, 1, nx, n, n, 0);
  gmm(out, nx, tq, n, 1, V, 1, n, nx, nx, n, 0);
}
__device__ __forceinline__ void gcopy_block(double* out, int ldo, const double* in, int ldi, int r, int c) {
  for (int e = threadIdx.x; e < r * c; e += 256) { const int i = e / c, j = e - i * c; out[(size_t)i * ldo + j] = in[(size_t)i * ldi + j]; }
}

// eigenvalues of the symmetric n x n matrix in the LDS tile A (ld BLD, n <= 64) by the parallel-ordered cyclic Jacobi method: every round rotates
// the n / 2 disjoint pairs of a round-robin tournament at once (columns, barrier, rows); the diagonal holds the eigenvalues at the end.
__device__ __forceinline__ void big_jacobi(double* A, int n, double* cs, double* red) {
  const int tid = threadIdx.x;
  const int m = (n + 1) & ~1, np = m >> 1;
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0, dg = 0.0;
    for (int e = tid; e < n * n; e += 256) { const int i = e / n, j = e - i * n; const double v = A[i * BLD + j]; if (i == j) dg = fma(v, v, dg); else off = fma(v, v, off); }
    off = wave_sum(off); dg = wave_sum(dg);
    __syncthreads();
    if ((tid & 63) == 0) { red[tid >> 6] = off; red[4 + (tid >> 6)] = dg; }
    __syncthreads();
    off = red[0] + red[1] + red[2] + red[3]; dg = red[4] + red[5] + red[6] + red[7];
    if (!(off > 1e-30 * dg)) break;                   // (uniform)
    for (int r = 0; r < m - 1; ++r) {
      if (tid < np) {
        int p_, q_;
        if (tid == 0) { p_ = m - 1; q_ = r; } else { p_ = (r + tid) % (m - 1); q_ = (r + m - 1 - tid) % (m - 1); }
        if (p_ > q_) { const int t_ = p_; p_ = q_; q_ = t_; }
        double c = 1.0, s = 0.0;
        if (q_ < n) {
          const double apq = A[p_ * BLD + q_], app = A[p_ * BLD + p_], aqq = A[q_ * BLD + q_];
          if (fabs(apq) > 1e-300) {
            const double th = (aqq - app) / (2.0 * apq);
            const double t = ((th >= 0.0) ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
            c = 1.0 / sqrt(t * t + 1.0); s = t * c;
          }
        }
        cs[4 * tid] = c; cs[4 * tid + 1] = s; cs[4 * tid + 2] = (double)p_; cs[4 * tid + 3] = (double)((q_ < n) ? q_ : -1);
      }
      __syncthreads();
      for (int e = tid; e < np * n; e += 256) {       // columns:  A <- A J
        const int pr = e / n, i = e - pr * n;
        const int q_ = (int)cs[4 * pr + 3];
        if (q_ < 0) continue;
        const int p_ = (int)cs[4 * pr + 2];
        const double c = cs[4 * pr], s = cs[4 * pr + 1];
        const double x = A[i * BLD + p_], y = A[i * BLD + q_];
        A[i * BLD + p_] = c * x - s * y; A[i * BLD + q_] = s * x + c * y;
      }
      __syncthreads();
      for (int e = tid; e < np * n; e += 256) {       // rows:  A <- J' A
        const int pr = e / n, j = e - pr * n;
        const int q_ = (int)cs[4 * pr + 3];
        if (q_ < 0) continue;
        const int p_ = (int)cs[4 * pr + 2];
        const double c = cs[4 * pr], s = cs[4 * pr + 1];
        const double x = A[p_ * BLD + j], y = A[q_ * BLD + j];
        A[p_ * BLD + j] = c * x - s * y; A[q_ * BLD + j] = s * x + c * y;
      }
      __syncthreads();
    }
  }
}
constexpr int BIG_EIG_LDS = NB * BLD + 4 * (NB / 2) + 16;        // tile + rotations + reduction scratch (doubles)
// smallest / largest eigenvalue (and the extreme absolute values, exact zeros excluded) of the symmetric matrix G (global, n x n, symmetrised on load)
__device__ __forceinline__ void big_eig_extremes(const double* G, int n, double* lds, double* lo, double* hi, double* amin, double* amax) {
  double* A = lds; double* cs = lds + NB * BLD; double* red = cs + 4 * (NB / 2);
  const int tid = threadIdx.x;
  __syncthreads();
  for (int e = tid; e < n * n; e += 256) { const int i = e / n, j = e - i * n; A[i * BLD + j] = 0.5 * (G[i * n + j] + G[j * n + i]); }
  __syncthreads();
  big_jacobi(A, n, cs, red);
  double l = 1e300, h = -1e300, an = 1e300, ax = 0.0;
  if (tid < n) { const double ev = A[tid * BLD + tid]; l = ev; h = ev; const double a = fabs(ev); if (a != 0.0) { an = a; ax = a; } }
  l = wave_min(l); h = wave_max(h); an = wave_min(an); ax = wave_max(ax);
  __syncthreads();
  if ((tid & 63) == 0) { red[tid >> 6] = l; red[4 + (tid >> 6)] = h; red[8 + (tid >> 6)] = an; red[12 + (tid >> 6)] = ax; }
  __syncthreads();
  *lo = fmin(fmin(red[0], red[1]), fmin(red[2], red[3])); *hi = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
  *amin = fmin(fmin(red[8], red[9]), fmin(red[10], red[11])); *amax = fmax(fmax(red[12], red[13]), fmax(red[14], red[15]));
  __syncthreads();
}

#define TMPC_BIG_PROLOGUE                                                        \
  const int sid = stage_id(w, dm);                                               \
  const int b = sid / dm.p, k = sid - b * dm.p;                                  \
  const int tid = threadIdx.x, n = dm.n, nx = dm.nx, nn = n * n, nxx = nx * nx;  \
  const size_t so = (size_t)sid * nn;                                            \
  double* scr = w.bscr + (size_t)sid * BIG_SCR * nn;                             \
  (void)k; (void)tid; (void)nxx; (void)so; (void)scr;

// ------------------------------------------------------------------ init: eigen-scan of H_k, V = [A B]   (k_init_stage)
__global__ void __launch_bounds__(256) kb_init_stage(WS w, Dims dm) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  TMPC_BIG_PROLOGUE
  const double* Hg = w.H + so;
  for (int e = tid; e < nn; e += 256) { const int i = e / n, j = e - i * n; w.Hb[so + e] = 0.5 * (Hg[i * n + j] + Hg[j * n + i]); }      // Hb := sym(H) for now (scaled by k_init_state)
  double lo, hi, amin, amax;
  big_eig_extremes(Hg, n, lds, &lo, &hi, &amin, &amax);
  if (tid == 0) { double* q = w.part + (size_t)sid * NPART; q[Q_MINEIG] = lo; q[Q_MAXEIG] = hi; q[Q_MINABS] = amin; q[Q_MAXABS] = amax; }
  const int mb = dm.mb;
  double* Vg = w.V + (size_t)sid * nx * n;
  const double* Ag = w.A + (size_t)sid * nxx; const double* Bg = w.Bm + (size_t)sid * nx * mb;
  for (int e = tid; e < nx * n; e += 256) { const int i = e / n, j = e - i * n; Vg[e] = (j < nx) ? Ag[i * nx + j] : Bg[i * mb + (j - nx)]; }
}

// ------------------------------------------------------------------ stage_pre   (k_stage_pre: residuals, factors of S_r and X_r, Kronecker factors, border pieces)
__global__ void __launch_bounds__(256) kb_stage_pre(WS w, Dims dm) {
  __shared__ double sflag[2];
  TMPC_BIG_PROLOGUE
  const int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] == PH_DONE) return;
  const double* pr = w.prob + (size_t)b * PS;
  const double alpha = pr[P_ALPHA], tau = pr[P_TAU];
  const int kn = (k + 1 == dm.p) ? 0 : k + 1;
  const double* V = w.V + (size_t)sid * nx * n; const double* Hb = w.Hb + so;
  double* sM = scr; double* t0 = scr + nn; double* t1 = scr + 2 * nn; double* sPhi = scr + 3 * nn; double* t3 = scr + 4 * nn;
  gbuild_M(sM, t0, V, Hb, w.P + (size_t)sid * nxx, w.P + (size_t)(b * dm.p + kn) * nxx, alpha, n, nx);
  if (dm.nr > 0) gadd_gtg(sM, w.G + (size_t)sid * dm.nr * n, w.phi + (size_t)sid * dm.nr, 1.0, stage_rows(w, dm, sid), n);      // + J' diag(phi) J
  if (dm.nT > 0) gadd_smat(sM, w.t3th + (size_t)sid * dm.nT, 1.0, n);                                                               // + T_k
  double rd2 = 0.0, s2 = 0.0, xs = 0.0, trx2 = 0.0, hby = 0.0, trpsi = 0.0, trphi2 = 0.0, hbphi = 0.0;
  int nbad = 0;
  for (int r = 0; r < 2; ++r) {
    const double* Sg = (r ? w.S2 : w.S1) + so; double* Rdg = (r ? w.Rd2 : w.Rd1) + so;
    for (int e = tid; e < nn; e += 256) {
      const int i = e / n, j = e - i * n;
      const double m = sM[e], sv = Sg[e], dg = (i == j) ? 1.0 : 0.0;
      const double rd = (r == 0 ? (m - dg) : (tau * dg - m)) - sv;
      Rdg[e] = rd; rd2 = fma(rd, rd, rd2); s2 = fma(sv, sv, s2);
    }
  }
  for (int e = tid; e < nn; e += 256) sPhi[e] = 0.0;
  gsync();
  double* kf = w.KF + (size_t)sid * 12 * nxx;
  for (int r = 0; r < 2; ++r) {
    const double* X = (r ? w.X2 : w.X1) + so; const double* S = (r ? w.S2 : w.S1) + so;
    double* Si = (r ? w.S2i : w.S1i) + so; double* Li = (r ? w.L2i : w.L1i) + so; double* LXi = (r ? w.LX2i : w.LX1i) + so;
    for (int e = tid; e < nn; e += 256) {
      const int i = e / n;
      const double x = X[e];
      xs = fma(x, S[e], xs);
      const double hx = Hb[e] * x;
      hby += r ? -hx : hx;
      if (r == 1 && e == i * n + i) trx2 += x;
      t0[e] = S[e]; t1[e] = x;
    }
    gsync();
    nbad += gchol(t0, n, sflag); gtri_inv(Li, t0, n);
    nbad += gchol(t1, n, sflag); gtri_inv(LXi, t1, n);
    gmm(Si, n, Li, 1, n, Li, n, 1, n, n, n, 0);                  // Li' Li
    gsym(Si, n);
    double* kfr = kf + (size_t)r * KF_PER_LMI * nxx;
    gmm(t0, n, V, n, 1, X, n, 1, nx, n, n, 0);                   // V X   (nx x n)
    gmm(kfr + KF_KX * nxx, nx, t0, n, 1, V, 1, n, nx, nx, n, 0); // V X V'
    for (int e = tid; e < nxx; e += 256) { const int a = e / nx, c = e - a * nx; kfr[KF_FX * nxx + e] = t0[c * n + a]; kfr[KF_XXX * nxx + e] = X[a * n + c]; }     // Fx = X[:nx,:] V' = ((V X)[:, :nx])'
    gsync();
    gmm(t0, n, V, n, 1, Si, n, 1, nx, n, n, 0);
    gmm(kfr + KF_KS * nxx, nx, t0, n, 1, V, 1, n, nx, nx, n, 0);
    for (int e = tid; e < nxx; e += 256) { const int a = e / nx, c = e - a * nx; kfr[KF_FS * nxx + e] = t0[c * n + a]; kfr[KF_SIXX * nxx + e] = Si[a * n + c]; }
    gsync();
    gmm(t0, n, X, n, 1, Hb, n, 1, n, n, n, 0);                   // Phi_r(Hb) = sym(X Hb Si)
    gmm(t1, n, t0, n, 1, Si, n, 1, n, n, n, 0);
    for (int e = tid; e < nn; e += 256) {
      const int i = e / n, j = e - i * n;
      const double phi = 0.5 * (t1[i * n + j] + t1[j * n + i]);
      sPhi[e] += phi;
      if (r == 1 && i == j) trphi2 += phi;
    }
    gsync();
    if (r == 1) {
      gmm(t0, n, X, n, 1, Si, n, 1, n, n, n, 0);                 // Psi = sym(X2 S2i)
      gsym(t0, n);
      for (int i = tid; i < n; i += 256) trpsi += t0[i * n + i];
      if (dm.nT > 0) { for (int e = tid; e < nn; e += 256) w.t3psi[so + e] = t0[e]; }
      gadj_V(w.adjV + ((size_t)sid * NADJ + ADJ_PSI) * nxx, t3, V, t0, n, nx);
      gcopy_block(w.adjE + ((size_t)sid * NADJ + ADJ_PSI) * nxx, nx, t0, n, nx, nx);
      gsync();
    }
  }
  for (int e = tid; e < nn; e += 256) hbphi = fma(Hb[e], sPhi[e], hbphi);
  if (dm.nT > 0) { for (int e = tid; e < nn; e += 256) w.t3phi[so + e] = sPhi[e]; }
  gadj_V(w.adjV + ((size_t)sid * NADJ + ADJ_PHI) * nxx, t3, V, sPhi, n, nx);
  gcopy_block(w.adjE + ((size_t)sid * NADJ + ADJ_PHI) * nxx, nx, sPhi, n, nx, nx);
  rd2 = block_sum<256>(rd2); s2 = block_sum<256>(s2); xs = block_sum<256>(xs); trx2 = block_sum<256>(trx2); hby = block_sum<256>(hby);
  trpsi = block_sum<256>(trpsi); trphi2 = block_sum<256>(trphi2); hbphi = block_sum<256>(hbphi);
  if (tid == 0) {
    double* q = w.part + (size_t)sid * NPART;
    q[Q_XS] = xs; q[Q_RD2] = rd2; q[Q_S2] = s2; q[Q_TRX2] = trx2; q[Q_HBY] = hby;
    q[Q_TRPSI] = trpsi; q[Q_TRPHI2] = trphi2; q[Q_HBPHI] = hbphi; q[Q_CHOLBAD] = (double)nbad;
  }
}

// ------------------------------------------------------------------ stage_rhs   (k_stage_rhs: T_r and the adjoint of G = T1 - T2)
__global__ void __launch_bounds__(256) kb_stage_rhs(WS w, Dims dm, int pass) {
  TMPC_BIG_PROLOGUE
  const int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return;
  const double* pr = w.prob + (size_t)b * PS;
  const double sig = (pass == 1) ? 0.0 : pr[P_SIGMU];
  const bool use_corr = (pass == 2 && phase == PH_MAIN);
  const double* V = w.V + (size_t)sid * nx * n; const double* Hb = w.Hb + so;
  double* t0 = scr; double* t1 = scr + nn; double* sG = scr + 2 * nn; double* t3 = scr + 3 * nn;
  double trt2 = 0.0, hbg = 0.0;
  for (int r = 0; r < 2; ++r) {
    const double* X = (r ? w.X2 : w.X1) + so; const double* Si = (r ? w.S2i : w.S1i) + so; const double* Rd = (r ? w.Rd2 : w.Rd1) + so;
    gmm(t0, n, X, n, 1, Rd, n, 1, n, n, n, 0);
    gmm(t1, n, t0, n, 1, Si, n, 1, n, n, n, 0);
    double* Tg = (r ? w.T2 : w.T1) + so; const double* cg = (r ? w.c2 : w.c1) + so;
    for (int e = tid; e < nn; e += 256) {
      const int i = e / n, j = e - i * n;
      double t = sig * Si[e] - 0.5 * (t1[i * n + j] + t1[j * n + i]);
      if (use_corr) t -= cg[e];
      Tg[e] = t;
      if (r == 0) sG[e] = t; else { sG[e] -= t; if (i == j) trt2 += t; }
    }
    gsync();
  }
  for (int e = tid; e < nn; e += 256) hbg = fma(Hb[e], sG[e], hbg);
  gadj_V(w.adjV + ((size_t)sid * NADJ + ADJ_G) * nxx, t3, V, sG, n, nx);
  gcopy_block(w.adjE + ((size_t)sid * NADJ + ADJ_G) * nxx, nx, sG, n, nx, nx);
  trt2 = block_sum<256>(trt2); hbg = block_sum<256>(hbg);
  if (tid == 0) { double* q = w.part + (size_t)sid * NPART; q[Q_TRT2] = trt2; q[Q_HBG] = hbg; }
}

// ------------------------------------------------------------------ stage_dir   (k_stage_dir: dS, dX, step-length matrices, corrector term)
__global__ void __launch_bounds__(256) kb_stage_dir(WS w, Dims dm, int pass) {
  TMPC_BIG_PROLOGUE
  const int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return;
  const double* pr = w.prob + (size_t)b * PS;
  const double dtau = pr[P_DTAU], dalpha = pr[P_DALPHA];
  const int kn = (k + 1 == dm.p) ? 0 : k + 1;
  const double* V = w.V + (size_t)sid * nx * n; const double* Hb = w.Hb + so;
  double* dM = scr; double* t0 = scr + nn; double* t1 = scr + 2 * nn; double* ldy = scr + 3 * nn;
  const double* dPk = w.dP + (size_t)sid * nxx;
  gbuild_M(dM, t0, V, Hb, dPk, w.dP + (size_t)(b * dm.p + kn) * nxx, dalpha, n, nx);
  if (dm.nr > 0) gadd_gtg(dM, w.G + (size_t)sid * dm.nr * n, w.dphi + (size_t)sid * dm.nr, 1.0, stage_rows(w, dm, sid), n);     // + J' diag(dphi) J
  if (dm.nT > 0) gadd_smat(dM, w.t3dth + (size_t)sid * dm.nT, 1.0, n);                                                             // + dT_k
  double dxs = 0.0, xds = 0.0, dxds = 0.0;
  for (int r = 0; r < 2; ++r) {
    const double* X = (r ? w.X2 : w.X1) + so; const double* S = (r ? w.S2 : w.S1) + so; const double* Si = (r ? w.S2i : w.S1i) + so;
    const double* Rd = (r ? w.Rd2 : w.Rd1) + so; const double* Tg = (r ? w.T2 : w.T1) + so;
    const double* Li = (r ? w.L2i : w.L1i) + so; const double* LXi = (r ? w.LX2i : w.LX1i) + so;
    double* dS = (r ? w.dS2 : w.dS1) + so; double* dX = (r ? w.dX2 : w.dX1) + so;
    for (int e = tid; e < nn; e += 256) {
      const int i = e / n, j = e - i * n;
      const double l = (r == 0) ? dM[e] : ((i == j ? dtau : 0.0) - dM[e]);
      ldy[e] = l; dS[e] = l + Rd[e];
    }
    gsync();
    gmm(t0, n, X, n, 1, ldy, n, 1, n, n, n, 0);
    gmm(t1, n, t0, n, 1, Si, n, 1, n, n, n, 0);
    for (int e = tid; e < nn; e += 256) {
      const int i = e / n, j = e - i * n;
      const double dx = Tg[e] - X[e] - 0.5 * (t1[i * n + j] + t1[j * n + i]);
      dX[e] = dx;
      const double ds = dS[e];
      dxs = fma(dx, S[e], dxs); xds = fma(X[e], ds, xds); dxds = fma(dx, ds, dxds);
    }
    gsync();
    double* Wd = w.Wm + ((size_t)sid * 4 + 2 * r) * nn; double* Wp = Wd + nn;
    gmm(t0, n, Li, n, 1, dS, n, 1, n, n, n, 0);                  // W_S = L^-1 dS L^-T
    gmm(Wd, n, t0, n, 1, Li, 1, n, n, n, n, 0);
    gsym(Wd, n);
    gmm(t0, n, LXi, n, 1, dX, n, 1, n, n, n, 0);                 // W_X = LX^-1 dX LX^-T
    gmm(Wp, n, t0, n, 1, LXi, 1, n, n, n, n, 0);
    gsym(Wp, n);
    if (pass == 1) {                                             // Mehrotra second-order term  sym(dX dS S^-1)
      double* cg = (r ? w.c2 : w.c1) + so;
      gmm(t0, n, dX, n, 1, dS, n, 1, n, n, n, 0);
      gmm(cg, n, t0, n, 1, Si, n, 1, n, n, n, 0);
      gsym(cg, n);
    }
  }
  double dh2 = 0.0, m2 = 0.0, dp2 = 0.0, p2 = 0.0;
  const double ra = dalpha / pr[P_ALPHA];
  for (int e = tid; e < nn; e += 256) {
    const int i = e / n, j = e - i * n;
    const double m = w.S1[so + e] + w.Rd1[so + e] + (i == j ? 1.0 : 0.0);
    const double dh = dM[e] - ra * m;
    dh2 = fma(dh, dh, dh2); m2 = fma(m, m, m2);
  }
  const double* Pk = w.P + (size_t)sid * nxx;
  for (int e = tid; e < nxx; e += 256) { dp2 = fma(dPk[e], dPk[e], dp2); p2 = fma(Pk[e], Pk[e], p2); }
  dxs = block_sum<256>(dxs); xds = block_sum<256>(xds); dxds = block_sum<256>(dxds); dp2 = block_sum<256>(dp2); p2 = block_sum<256>(p2);
  dh2 = block_sum<256>(dh2); m2 = block_sum<256>(m2);
  if (tid == 0) {
    double* q = w.part + (size_t)sid * NPART;
    q[Q_DXS] = dxs; q[Q_XDS] = xds; q[Q_DXDS] = dxds; q[Q_DP2] = dp2; q[Q_P2] = p2; q[Q_DH2] = dh2; q[Q_M2] = m2;
  }
}

// dalpha*Hb + calH(dP) of the stage into scratch slot 4, for k_phi_dir<true> (which runs before kb_stage_dir and needs it without the multiplier term)
__global__ void __launch_bounds__(256) kb_phi_dm(WS w, Dims dm, int pass) {
  TMPC_BIG_PROLOGUE
  const int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return;
  const double* pr = w.prob + (size_t)b * PS;
  const int kn = (k + 1 == dm.p) ? 0 : k + 1;
  gbuild_M(scr + 4 * nn, scr, w.V + (size_t)sid * nx * n, w.Hb + so, w.dP + (size_t)sid * nxx, w.dP + (size_t)(b * dm.p + kn) * nxx, pr[P_DALPHA], n, nx);
}

// ------------------------------------------------------------------ smallest eigenvalue of one step-length matrix   (k_eigmin; one workgroup per matrix)
__global__ void __launch_bounds__(256) kb_eigmin(WS w, Dims dm, int pass) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int mid = stage_id4(w, dm);
  const int b = (mid >> 2) / dm.p;
  const int phase = w.iprob[(size_t)b * IS + I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return;
  double lo, hi, amin, amax;
  big_eig_extremes(w.Wm + (size_t)mid * dm.n * dm.n, dm.n, lds, &lo, &hi, &amin, &amax);
  if (threadIdx.x == 0) w.eigmin[mid] = lo;
}

// ------------------------------------------------------------------ final: un-scale, supplement, status eigenvalues   (k_final_stage, plain model)
__global__ void __launch_bounds__(256) kb_final_stage(WS w, Dims dm) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  TMPC_BIG_PROLOGUE
  const int* ip = w.iprob + (size_t)b * IS;
  const double* pr = w.prob + (size_t)b * PS;
  const int kn = (k + 1 == dm.p) ? 0 : k + 1;
  const double sc = ip[I_EARLY] ? 0.0 : 1.0 / (pr[P_S] * pr[P_ALPHA]);      // dP = sP*P/(s_alpha*alpha), convexifier.py:406
  double* Po = w.Pout + (size_t)sid * nxx; double* Pon = w.Pout + (size_t)(b * dm.p + kn) * nxx;
  const double* Pn = w.P + (size_t)(b * dm.p + kn) * nxx;
  for (int e = tid; e < nxx; e += 256) Po[e] = sc * w.P[(size_t)sid * nxx + e];
  if (kn != k) for (int e = tid; e < nxx; e += 256) Pon[e] = sc * Pn[e];       // same values any writer would store
  double* sM = scr; double* t0 = scr + nn; double* sH = scr + 2 * nn;
  const double* Hg = w.H + so;
  for (int e = tid; e < nn; e += 256) { const int i = e / n, j = e - i * n; sH[e] = 0.5 * (Hg[i * n + j] + Hg[j * n + i]); }
  gsync();
  gbuild_M(sM, t0, w.V + (size_t)sid * nx * n, sH, Po, Pon, 0.0, n, nx);       // dH = V' Pst+ V - E' Pst E
  if (dm.nr > 0) {                          // [Fg; F] = sF*phi/(s_alpha*alpha) and their terms of the supplement (k_final_stage)
    const int nrow = stage_rows(w, dm, sid);
    if (tid < dm.nr) w.Fg[(size_t)sid * dm.nr + tid] = (tid < nrow) ? sc * w.phi[(size_t)sid * dm.nr + tid] : 0.0;
    gadd_gtg(sM, w.G + (size_t)sid * dm.nr * n, w.phi + (size_t)sid * dm.nr, sc, nrow, n);
  }
  if (dm.nT > 0) {                          // T_k = s_T theta / (s_alpha alpha) and its term of the supplement
    const double* th = w.t3th + (size_t)sid * dm.nT;
    gadd_smat(sM, th, sc, n);
    for (int e = tid; e < nn; e += 256) {
      const int i = e / n, j = e - i * n;
      const int a = i < j ? i : j, bq = i < j ? j : i;
      w.Tout[so + e] = sc * th[a * n - a * (a - 1) / 2 + (bq - a)];
    }
  }
  gsym(sM, n);
  for (int e = tid; e < nn; e += 256) { const double dh = sM[e]; w.dHc[so + e] = dh; w.Hc[so + e] = sH[e] + dh; }
  gsync();
  double lo, hi, amin, amax;
  big_eig_extremes(w.Hc + so, n, lds, &lo, &hi, &amin, &amax);
  if (tid == 0) { double* q = w.part + (size_t)sid * NPART; q[Q_MINEIG] = lo; q[Q_MAXEIG] = hi; }
}

// ------------------------------------------------------------------ tight mode, polish (k_polish_step of tmpc_dd.h): dM of the step, the norms of the step test, the dual iterate of the step
__global__ void __launch_bounds__(256) kb_polish_step(WS w, Dims dm) {
  TMPC_BIG_PROLOGUE
  if (w.iprob[(size_t)b * IS + I_PHASE] != PH_POLISH) return;
  const double* pr = w.prob + (size_t)b * PS;
  const int kn = (k + 1 == dm.p) ? 0 : k + 1;
  double* dM = scr; double* t0 = scr + nn; double* t1 = scr + 2 * nn;
  gbuild_M(dM, t0, w.V + (size_t)sid * nx * n, w.Hb + so, w.dP + (size_t)sid * nxx, w.dP + (size_t)(b * dm.p + kn) * nxx, pr[P_DALPHA], n, nx);
  if (dm.nr > 0) gadd_gtg(dM, w.G + (size_t)sid * dm.nr * n, w.dphi + (size_t)sid * dm.nr, 1.0, stage_rows(w, dm, sid), n);     // + J' diag(dphi) J (round 5: rows in the tight mode)
  const double ra = pr[P_DALPHA] / pr[P_ALPHA];
  double dh2 = 0.0, m2 = 0.0;
  for (int e = tid; e < nn; e += 256) { const double m = w.T1[so + e], dh = dM[e] - ra * m; dh2 = fma(dh, dh, dh2); m2 = fma(m, m, m2); }
  dh2 = block_sum<256>(dh2); m2 = block_sum<256>(m2);
  if (tid == 0) { double* q = w.part + (size_t)sid * NPART; q[Q_DH2] = dh2; q[Q_M2] = m2; }
  const double dtau = pr[P_DTAU];
  for (int r = 0; r < 2; ++r) {
    const double* X = (r ? w.X2 : w.X1) + so; const double* Z = (r ? w.S2i : w.S1i) + so;
    if (r == 1) { for (int e = tid; e < nn; e += 256) { const int i = e / n, j = e - i * n; dM[e] = ((i == j) ? dtau : 0.0) - dM[e]; } }      // dS2 = dtau I - dM
    gsync();
    gmm(t0, n, X, n, 1, dM, n, 1, n, n, n, 0);
    gmm(t1, n, t0, n, 1, Z, n, 1, n, n, n, 0);
    double* out = (r ? w.dX2 : w.dX1) + so;
    if (!w.iprob[(size_t)b * IS + I_CHORD]) for (int e = tid; e < nn; e += 256) { const int i = e / n, j = e - i * n; out[e] = X[e] - 0.5 * (t1[i * n + j] + t1[j * n + i]); }      // (chord step: k_polish_step)
    gsync();
  }
}

// ------------------------------------------------------------------ the other entry points of the boundary at 32 < n <= 64
// eigen-scan of arbitrary stage blocks (tmpc_eig_scan_host; k_eig_scan)
__global__ void __launch_bounds__(256) kb_eig_scan(const double* H, double* out, int n) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double lo, hi, amin, amax;
  big_eig_extremes(H + (size_t)blockIdx.x * n * n, n, lds, &lo, &hi, &amin, &amax);
  if (threadIdx.x == 0) { double* o = out + (size_t)blockIdx.x * 4; o[0] = lo; o[1] = hi; o[2] = amin; o[3] = amax; }
}
// dHc = sym(V' P+ V - E' P E [+ J' diag(w) J] [+ T]) for arbitrary P (tmpc_supplement_batch_host; the generic twin of k_supplement); scr: 3 n x n doubles per stage
__global__ void __launch_bounds__(256) kb_supplement(const double* A, const double* Bm, const double* P, double* dHc, Dims dm, double* scr_all, int nr, const double* J, const double* wts,
                                                     const double* T) {
  const int sid = blockIdx.x, tid = threadIdx.x;
  const int b = sid / dm.p, k = sid - b * dm.p;
  const int n = dm.n, nx = dm.nx, mb = dm.mb, nn = n * n, nxx = nx * nx;
  double* sV = scr_all + (size_t)sid * 3 * nn; double* sM = sV + nn; double* t0 = sM + nn;
  for (int e = tid; e < nx * n; e += 256) { const int i = e / n, j = e - i * n; sV[e] = (j < nx) ? A[(size_t)sid * nxx + i * nx + j] : Bm[(size_t)sid * nx * mb + i * mb + (j - nx)]; }
  gsync();
  const int kn = (k + 1 == dm.p) ? 0 : k + 1;
  gbuild_M(sM, t0, sV, sM, P + (size_t)sid * nxx, P + (size_t)(b * dm.p + kn) * nxx, 0.0, n, nx);
  if (J) gadd_gtg(sM, J + (size_t)sid * nr * n, wts + (size_t)sid * nr, 1.0, nr, n);      // + J' diag(w) J (padded rows carry weight 0)
  if (T) { for (int e = tid; e < nn; e += 256) sM[e] += T[(size_t)sid * nn + e]; gsync(); }  // + T_k (convexifier.py:202-203; round 5: Step 3 runs at n > 32, so does its supplement)
  for (int e = tid; e < nn; e += 256) { const int i = e / n, j = e - i * n; dHc[(size_t)sid * nn + e] = 0.5 * (sM[i * n + j] + sM[j * n + i]); }
}
// W = sym(Hc) / ts, yref = wref - Hc^-1 q by Cholesky and two substitutions (tmpc_tracking_reference_host; k_tracking_ref); scr: n x n doubles per stage
__global__ void __launch_bounds__(256) kb_tracking_ref(const double* Hc, const double* q, const double* wref, double inv_ts, double* W, double* yref, int* info, int n, double* scr_all) {
  __shared__ double sflag[2];
  __shared__ double v[NB];
  const size_t sid = blockIdx.x;
  const int tid = threadIdx.x, nn = n * n;
  double* Aw = scr_all + sid * nn;
  const double* Hg = Hc + sid * nn;
  for (int e = tid; e < nn; e += 256) { const int i = e / n, j = e - i * n; const double a = 0.5 * (Hg[i * n + j] + Hg[j * n + i]); Aw[e] = a; if (W) W[sid * nn + e] = a * inv_ts; }
  if (tid < n) v[tid] = q[sid * n + tid];
  gsync();
  const int nbad = gchol(Aw, n, sflag);
  for (int j = 0; j < n; ++j) {             // L z = q
    if (tid == j) v[j] /= Aw[j * n + j];
    __syncthreads();
    if (tid > j && tid < n) v[tid] -= Aw[tid * n + j] * v[j];
    __syncthreads();
  }
  for (int j = n - 1; j >= 0; --j) {        // L' x = z
    if (tid == j) v[j] /= Aw[j * n + j];
    __syncthreads();
    if (tid < j) v[tid] -= Aw[j * n + tid] * v[j];
    __syncthreads();
  }
  if (tid < n) yref[sid * n + tid] = wref[sid * n + tid] - v[tid];
  if (info && tid == 0) info[sid] = nbad;
}

}  // namespace tmpc
