// Per-stage kernels: one 64-lane wavefront per (problem b, stage k).  They evaluate the LMI blocks
// M_k = alpha*Hb_k + V_k' P_{k+1} V_k - E' P_k E  (reference: convexifier.py:325-357), their
// Cholesky factors / inverses, the HKM scaling pieces, residuals and step lengths.
#pragma once
#include "tmpc_common.h"
#include "tmpc_small.h"

namespace tmpc {

// index (b*p + k)*4 + which for k_eigmin: four workgroups per stage
__device__ __forceinline__ int stage_id4(const WS& w, const Dims& dm) {
  if (!w.alist) return blockIdx.x;
  const int sb = blockIdx.x >> 2, bi = sb / dm.p;
  return ((w.alist[bi] * dm.p + (sb - bi * dm.p)) << 2) | (blockIdx.x & 3);
}

#define TMPC_STAGE_PROLOGUE                                     \
  const int sid = stage_id(w, dm);                              \
  const int b = sid / dm.p;                                     \
  const int k = sid - b * dm.p;                                 \
  const int lane = threadIdx.x;                                 \
  const int n = dm.n, nx = dm.nx;                               \
  const int nn = n * n, nxx = nx * nx;                          \
  (void)k; (void)nn; (void)nxx;                                 \
  extern __shared__ __attribute__((aligned(16))) double sm[];

#ifdef TMPC_EIG_DEBUG
__device__ int g_eig_dbg[4]; __device__ double g_eig_dbgv[8];
#endif

// rows of [G_k; C_k] present at stage sid
__device__ __forceinline__ int stage_rows(const WS& w, const Dims& dm, size_t sid) {
  int nc = w.ncnt ? w.ncnt[sid] : 0;                    // the device entry cannot range-check the caller's counts: clamp to the room of the handle
  nc = nc < 0 ? 0 : (nc > dm.nr - dm.ng ? dm.nr - dm.ng : nc);
  return dm.ng + nc;
}

#include "tmpc_stage_impl.h"
namespace sm8 {
#include "tmpc_stage_impl.h"
}  // namespace sm8

// ------------------------------------------------------------------ init: eigen-scan of H_k, V = [A B]
// reference: convexifier.py:82 (pre-check) and :374-401 (autoScaling)
__global__ void __launch_bounds__(64) k_init_stage(WS w, Dims dm) {
  TMPC_STAGE_PROLOGUE
  double* sH = sm; double* cs = sm + MS;
  const double* Hg = w.H + (size_t)sid * nn;
  for (int e = lane; e < nn; e += 64) {
    int i, j; ediv(e, n, i, j);
    sH[i * LD + j] = 0.5 * (Hg[i * n + j] + Hg[j * n + i]);
  }
  wsync();
  // Hb := sym(H) for now (scaled later by k_init_state)
  s2g(w.Hb + (size_t)sid * nn, sH, n, n, n, lane);
  jacobi_eigvals(sH, n, cs, lane);
  double lo = 1e300, hi = -1e300, amin = 1e300, amax = 0.0;
  if (lane < n) {
    const double ev = sH[lane * LD + lane];
    lo = ev; hi = ev;
    const double a = fabs(ev);
    if (a != 0.0) { amin = a; amax = a; }       // exact zeros excluded (convexifier.py:387-388)
  }
  lo = wave_min(lo); hi = wave_max(hi); amin = wave_min(amin); amax = wave_max(amax);
  if (lane == 0) {
    double* q = w.part + (size_t)sid * NPART;
    q[Q_MINEIG] = lo; q[Q_MAXEIG] = hi; q[Q_MINABS] = amin; q[Q_MAXABS] = amax;
  }
  // V = [A B]
  const int mb = dm.mb;
  double* Vg = w.V + (size_t)sid * nx * n;
  const double* Ag = w.A + (size_t)sid * nxx;
  const double* Bg = w.Bm + (size_t)sid * nx * mb;
  for (int e = lane; e < nx * n; e += 64) {
    int i, j; ediv(e, n, i, j);
    Vg[e] = (j < nx) ? Ag[i * nx + j] : Bg[i * mb + (j - nx)];
  }
}

__global__ void __launch_bounds__(64) k_init_prob(WS w, Dims dm) {
  const int b = blockIdx.x, lane = threadIdx.x;
  double lo = 1e300, amin = 1e10 /* convexifier.py:383 */, amax = 0.0;
  for (int k = lane; k < dm.p; k += 64) {
    const double* q = w.part + (size_t)(b * dm.p + k) * NPART;
    lo = fmin(lo, q[Q_MINEIG]); amin = fmin(amin, q[Q_MINABS]); amax = fmax(amax, q[Q_MAXABS]);
  }
  lo = wave_min(lo); amin = wave_min(amin); amax = wave_max(amax);
  if (lane == 0) {
    double* pr = w.prob + (size_t)b * PS;
    int* ip = w.iprob + (size_t)b * IS;
    for (int i = 0; i < PS; ++i) pr[i] = 0.0;
    for (int i = 0; i < IS; ++i) ip[i] = 0;
    const double s = 1.0 / amin, sbeta = amax / amin;
    pr[P_S] = s; pr[P_SBETA] = sbeta; pr[P_MINEIG_H] = lo;
    // everything O(1) from the start: alpha*Hb has eigenvalues in [-1, 1], S2 = tau*I - alpha*Hb in [1, 3]
    pr[P_TAU] = 2.0; pr[P_ALPHA] = 1.0 / sbeta; pr[P_S0] = 1.0 / sbeta; pr[P_X0] = 1.0 / (double)(dm.p * dm.n);
    pr[P_MUT] = -1.0; pr[P_PREVSTEPN] = -1.0; pr[P_STEPN] = 1e300; pr[P_MINPIV] = 1.0;
    const int early = (lo > 0.0) ? 1 : 0;          // convexifier.py:83
    ip[I_EARLY] = early;
    ip[I_PHASE] = early ? PH_DONE : PH_MAIN;
    ip[I_IPMSTATUS] = early ? IPM_OPTIMAL : IPM_MAXITER;
    if (!early) { const int slot = atomicAdd(w.active, 1); w.alist[slot] = b; w.flist[slot] = b; }
    // the first factorisation with single-precision updates when the call has them on (later iterations: k_ctrl_d)
    if (!early && (dm.flags & 8) && w.O32) { ip[I_LOWP] = 1; atomicAdd(w.active + 4, 1); }
  }
}

__global__ void __launch_bounds__(64) k_init_state(WS w, Dims dm) {
  TMPC_STAGE_PROLOGUE
  const double* pr = w.prob + (size_t)b * PS;
  const double s = pr[P_S], tau = pr[P_TAU], alpha = pr[P_ALPHA];
  const double x0 = 1.0 / (double)(dm.p * n);
  double* Hb = w.Hb + (size_t)sid * nn;
  for (int e = lane; e < nn; e += 64) {
    int i, j; ediv(e, n, i, j);
    const double hb = s * Hb[e];
    Hb[e] = hb;
    const double dg = (i == j) ? 1.0 : 0.0;
    w.S1[(size_t)sid * nn + e] = dg;
    w.S2[(size_t)sid * nn + e] = tau * dg - alpha * hb;
    w.X1[(size_t)sid * nn + e] = x0 * dg;
    w.X2[(size_t)sid * nn + e] = x0 * dg;
  }
  for (int e = lane; e < nxx; e += 64) w.P[(size_t)sid * nxx + e] = 0.0;
}

// ------------------------------------------------------------------ smallest eigenvalue of a step-length matrix with n <= 8: ONE THREAD per matrix (round 5)
// The wave-per-matrix form (eigmin_body) spends ~20 k cycles of one wave on a 5 x 5 matrix: every Householder step is a handful of LDS round trips and
// barriers for a dozen flops.  At n <= 8 the whole matrix fits the registers of ONE lane (36 doubles), so a lane runs the textbook scalar algorithm with
// every index a literal: Householder tridiagonalisation of the lower triangle, then Laguerre's iteration on the characteristic polynomial of the
// tridiagonal matrix from the Gershgorin bound -- for a polynomial with real roots it rises monotonically to the smallest root, cubically for a simple
// root, and is exact in one step for an n-fold one.  No LDS, no barrier; 64 matrices per wave at once.  Always the exact eigenvalue (the pre-test of
// eigmin_body reports a bound that leads to the same clipped step and the same decisions, see there).
__device__ __forceinline__ double lane_min_eig8(const double* __restrict__ Wg, int n) {
  double a[36];                                             // lower triangle, (i, j) at i (i + 1) / 2 + j; rows / columns >= n: zero
#define TMPC_A8(i, j) a[(i) * ((i) + 1) / 2 + (j)]
#define TMPC_S8(r, c) (((c) <= (r)) ? TMPC_A8(r, c) : TMPC_A8(c, r))
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) TMPC_A8(i, j) = (i < n) ? 0.5 * (Wg[i * n + j] + Wg[j * n + i]) : 0.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double sigma = 0.0;
#pragma unroll
    for (int i = j + 2; i < 8; ++i) sigma = fma(TMPC_A8(i, j), TMPC_A8(i, j), sigma);
    if (sigma != 0.0) {
      const double x0 = TMPC_A8(j + 1, j);
      const double mu = sqrt(fma(x0, x0, sigma));
      const double v0 = (x0 <= 0.0) ? (x0 - mu) : (-sigma / (x0 + mu));
      const double beta = 2.0 * v0 * v0 / (sigma + v0 * v0);
      const double rv0 = 1.0 / v0;
      double v[8], pw[8];
#pragma unroll
      for (int i = j + 1; i < 8; ++i) v[i] = (i == j + 1) ? 1.0 : TMPC_A8(i, j) * rv0;
      double kk = 0.0;
#pragma unroll
      for (int r = j + 1; r < 8; ++r) {
        double acc = 0.0;
#pragma unroll
        for (int c = j + 1; c < 8; ++c) acc = fma(TMPC_S8(r, c), v[c], acc);
        pw[r] = beta * acc;
        kk = fma(pw[r], v[r], kk);
      }
      kk *= 0.5 * beta;
#pragma unroll
      for (int r = j + 1; r < 8; ++r) pw[r] = fma(-kk, v[r], pw[r]);          // w
#pragma unroll
      for (int r = j + 1; r < 8; ++r)
#pragma unroll
        for (int c = j + 1; c <= r; ++c) TMPC_A8(r, c) -= fma(v[r], pw[c], pw[r] * v[c]);
      TMPC_A8(j + 1, j) = mu;                               // |H x| = mu e_1 (the sign does not matter for the eigenvalues of the tridiagonal matrix)
    }
  }
  double dd[8], ee[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { dd[i] = TMPC_A8(i, i); ee[i] = (i < 7) ? TMPC_A8(i + 1, i) : 0.0; }
#undef TMPC_S8
#undef TMPC_A8
  if (n == 1) return dd[0];
  double scale = 0.0;
  bool bad = false;                                          // fmax / fmin drop NaN operands: test for it explicitly, so that a NaN step-length matrix is handed on
#pragma unroll
  for (int i = 0; i < 8; ++i) if (i < n) { bad |= !(dd[i] == dd[i]) || ((i + 1 < n) && !(ee[i] == ee[i])); scale = fmax(scale, fmax(fabs(dd[i]), (i + 1 < n) ? fabs(ee[i]) : 0.0)); }
  if (bad) return __builtin_nan("");                         // (the control body treats a non-finite step as a breakdown)
  if (!(scale > 0.0) || !(scale < 1e300)) return (scale == 0.0) ? 0.0 : scale;       // zero matrix; Inf is handed on
  const double rs = 1.0 / scale;
  double e2[8];
  double lo = 1e300;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    dd[i] *= rs; ee[i] *= rs; e2[i] = ee[i] * ee[i];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < n) lo = fmin(lo, dd[i] - ((i > 0) ? fabs(ee[i - 1]) : 0.0) - ((i + 1 < n) ? fabs(ee[i]) : 0.0));
  double x = lo - 1e-14;
  const double dn = (double)n;
  for (int it = 0; it < 64; ++it) {
    // q_i = det(T_i - x I) with first and second derivatives: q_i = (d_{i-1} - x) q_{i-1} - e_{i-2}^2 q_{i-2}
    double qm = 1.0, q = dd[0] - x, gm = 0.0, g = -1.0, hm = 0.0, h = 0.0;
#pragma unroll
    for (int i = 1; i < 8; ++i) {
      if (i < n) {
        const double t = dd[i] - x, c = e2[i - 1];
        const double qn = fma(t, q, -c * qm);
        const double gn = fma(t, g, -c * gm) - q;
        const double hn = fma(t, h, -c * hm) - 2.0 * g;
        qm = q; q = qn; gm = g; g = gn; hm = h; h = hn;
      }
    }
    if (!(q > 0.0)) break;                                  // det(T - x I) > 0 left of the smallest eigenvalue: at it (or past it by rounding) the iteration is done
    const double G = g / q, H = G * G - h / q;             // G = -sum 1 / (lambda_i - x), H = sum 1 / (lambda_i - x)^2
    double disc = (dn - 1.0) * (dn * H - G * G);
    if (!(disc > 0.0)) disc = 0.0;
    const double step = dn / (sqrt(disc) - G);
    if (!(step > 4e-16 * fmax(1.0, fabs(x)))) break;        // converged (or past the root by rounding: the step turns negative / NaN)
    x += step;
  }
  return x * scale;
}
// (b*p + k)*4 + which of the matrices [m0, m0 + count): thread i takes matrix m0 + i
__device__ __forceinline__ void eigmin_lane_body(const WS& w, const Dims& dm, int mid, int pass) {
  const int b = (mid >> 2) / dm.p;
  const int phase = w.iprob[(size_t)b * IS + I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return;
  w.eigmin[mid] = lane_min_eig8(w.Wm + (size_t)mid * dm.n * dm.n, dm.n);
}
__global__ void __launch_bounds__(64) k_eigmin_lane(WS w, Dims dm, int pass, int nmat) {      // n <= 8: nmat = 4 * (active problems) * p
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= nmat) return;
  const int sb = i >> 2, bi = sb / dm.p;
  const int mid = w.alist ? (((w.alist[bi] * dm.p + (sb - bi * dm.p)) << 2) | (i & 3)) : i;
  eigmin_lane_body(w, dm, mid, pass);
}

// ------------------------------------------------------------------ the per-iteration kernels: one workgroup per stage around the bodies of tmpc_stage_impl.h
template <int NT>
__global__ void __launch_bounds__(NT, NT == 256 ? 3 : 1) k_stage_pre(WS w, Dims dm) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  stage_pre_body<NT>(w, dm, stage_id(w, dm), threadIdx.x, sm);
}
template <int NT>
__global__ void __launch_bounds__(NT, NT == 256 ? 3 : 1) k_stage_rhs(WS w, Dims dm, int pass) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  stage_rhs_body<NT>(w, dm, stage_id(w, dm), threadIdx.x, sm, pass);
}
template <int NT>
__global__ void __launch_bounds__(NT, NT == 256 ? 3 : 1) k_stage_dir(WS w, Dims dm, int pass) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  stage_dir_body<NT>(w, dm, stage_id(w, dm), threadIdx.x, sm, pass);
}
// one single-wave block per step-length matrix (4 per stage): 8.9 KB of LDS each -> ~17 blocks (waves) resident per CU
__global__ void __launch_bounds__(64) k_eigmin(WS w, Dims dm, int pass, double chord_step) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  eigmin_body(w, dm, stage_id4(w, dm), threadIdx.x, sm, pass, chord_step);
}
template <int NT>
__global__ void __launch_bounds__(NT) k_update(WS w, Dims dm) {
  update_body<NT>(w, dm, stage_id(w, dm), threadIdx.x);
}

constexpr int FIN_SLOTS = 6;
// ------------------------------------------------------------------ final: un-scale, supplement, status eigenvalues
// reference: convexifier.py:403-440 (check_convergence) and :165-211 (convexHessianSuppl)
__global__ void __launch_bounds__(64) k_final_stage(WS w, Dims dm) {
  TMPC_STAGE_PROLOGUE
  const int* ip = w.iprob + (size_t)b * IS;
  const double* pr = w.prob + (size_t)b * PS;
  double* sV = sm; double* sM = sm + MS; double* t0 = sm + 2 * MS; double* t1 = sm + 3 * MS; double* sHb = sm + 4 * MS;
  double* cs = sm + 5 * MS;
  const int kn = (k + 1 == dm.p) ? 0 : k + 1;
  const double sc = ip[I_EARLY] ? 0.0 : 1.0 / (pr[P_S] * pr[P_ALPHA]);      // dP = sP*P/(s_alpha*alpha), convexifier.py:406
  // Pout = sc * P
  double* Po = w.Pout + (size_t)sid * nxx;
  for (int e = lane; e < nxx; e += 64) Po[e] = sc * w.P[(size_t)sid * nxx + e];
  double* Pon = w.Pout + (size_t)(b * dm.p + kn) * nxx;
  const double* Pn = w.P + (size_t)(b * dm.p + kn) * nxx;
  if (kn != k) { for (int e = lane; e < nxx; e += 64) Pon[e] = sc * Pn[e]; }   // same values any writer would store
  __threadfence_block();
  wsync();
  g2s(sV, w.V + (size_t)sid * nx * n, nx, n, n, lane);
  // H (unscaled, symmetrised as in k_init_stage)
  const double* Hg = w.H + (size_t)sid * nn;
  for (int e = lane; e < nn; e += 64) { int i, j; ediv(e, n, i, j); sHb[i * LD + j] = 0.5 * (Hg[i * n + j] + Hg[j * n + i]); }
  wsync();
  // dH = V' Pst+ V - E' Pst E  (coef 0: no Hb term), then symmetrise (mtools.symmetrize, convexifier.py:206)
  build_M(sM, sV, t0, t1, sHb, Po, Pon, 0.0, n, nx, lane);
  if (dm.nr > 0) {                          // [Fg; F] = sF*phi/(s_alpha*alpha) (convexifier.py:409-420) and their terms of the supplement (:196-201)
    const int nrow = stage_rows(w, dm, sid);
    if (lane < dm.nr) w.Fg[(size_t)sid * dm.nr + lane] = (lane < nrow) ? sc * w.phi[(size_t)sid * dm.nr + lane] : 0.0;
    add_gtg(sM, w.G + (size_t)sid * dm.nr * n, w.phi + (size_t)sid * dm.nr, sc, nrow, n, lane);
  }
  if (dm.nT > 0) {                          // T_k = s_T theta / (s_alpha alpha) (convexifier.py:422-423) and its term of the supplement (:202-203)
    add_smat_t3(sM, w.t3th + (size_t)sid * dm.nT, sc, n, lane);
    for (int e = lane; e < nn; e += 64) {
      int i, j; ediv(e, n, i, j);
      const int a = i < j ? i : j, bq = i < j ? j : i;
      w.Tout[(size_t)sid * nn + e] = sc * w.t3th[(size_t)sid * dm.nT + a * n - a * (a - 1) / 2 + (bq - a)];
    }
  }
  s_sym(sM, n, lane);
  double* dHg = w.dHc + (size_t)sid * nn;
  double* Hcg = w.Hc + (size_t)sid * nn;
  for (int e = lane; e < nn; e += 64) {
    int i, j; ediv(e, n, i, j);
    const double dh = sM[i * LD + j];
    dHg[e] = dh;
    const double hc = sHb[i * LD + j] + dh;
    Hcg[e] = hc;
    t0[i * LD + j] = hc;
  }
  wsync();
  // extreme eigenvalues of Hc_k for the status rule (convexifier.py:438-451): Householder tridiagonalisation + Sturm multisection at
  // both ends (a full Jacobi diagonalisation cost 3/4 of this kernel)
  const double lo = tridiag_min_eig(t0, n, cs, lane);
  const double hi = tridiag_max_after(cs, n, lane);
  if (lane == 0) {
    double* q = w.part + (size_t)sid * NPART;
    q[Q_MINEIG] = lo; q[Q_MAXEIG] = hi;
  }
}

__global__ void __launch_bounds__(64) k_final_prob(WS w, Dims dm) {
  const int b = blockIdx.x, lane = threadIdx.x;
  double lo = 1e300, hi = -1e300, mc = 0.0;
  for (int k = lane; k < dm.p; k += 64) {
    const double* q = w.part + (size_t)(b * dm.p + k) * NPART;
    lo = fmin(lo, q[Q_MINEIG]); hi = fmax(hi, q[Q_MAXEIG]);
    mc = fmax(mc, (q[Q_MINEIG] > 0.0) ? q[Q_MAXEIG] / q[Q_MINEIG] : 1e300);
  }
  lo = wave_min(lo); hi = wave_max(hi); mc = wave_max(mc);
  if (lane == 0) {
    double* pr = w.prob + (size_t)b * PS;
    int* ip = w.iprob + (size_t)b * IS;
    pr[P_MINEIG_HC] = lo; pr[P_MAXEIG_HC] = hi; pr[P_MAXCOND] = mc;
    // status rule of convexifier.py:442-451
    int st;
    if (lo > 0.0) st = (ip[I_IPMSTATUS] == IPM_OPTIMAL || ip[I_IPMSTATUS] == IPM_FAST_EXIT || ip[I_IPMSTATUS] == IPM_TIGHT_FALLBACK) ? ST_OPTIMAL : ST_FEASIBLE;
    else st = ST_INFEASIBLE;
    ip[I_STATUS] = st;
    if (ip[I_EARLY]) { pr[P_KAPPA] = 0.0; pr[P_BETA] = 0.0; pr[P_ALPHA_OUT] = 1.0; }
    else { pr[P_KAPPA] = pr[P_TAU]; pr[P_BETA] = pr[P_TAU] / pr[P_SBETA]; pr[P_ALPHA_OUT] = pr[P_ALPHA]; }
  }
}

}  // namespace tmpc
