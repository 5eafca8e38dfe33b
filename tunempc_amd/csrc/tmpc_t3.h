// Step 3 of the convexifier (reference: convexifier.py:137-147): the forced regularisation T_k.
//   T_k symmetric n x n with every entry > 0 (:269-273), term s_T*T_k in HcE_k (:352-353), rho*||T_k||_F in the objective (:284-285),
//   un-scaling T_k <- s_T T_k / (s_alpha alpha) (:422-423).
// In the scaled problem M_k = alpha*Hb_k + calH_k(P) + smat(theta_k), theta_k = the m = n(n+1)/2 entries (a <= b) of T_k:
//   * theta_e >= 0: a linear cone (slack = theta itself, dual z_e), like the constraint multipliers of tmpc_phi.h;
//   * the norm term: epigraph variable t_k with (t_k; w c o theta_k) in the second-order cone Q^{m+1}, w = rho*sbeta/s, c_e = 1 on
//     the diagonal and sqrt(2) off it (Frobenius norm of the symmetric matrix), cost 1 -- the scaled objective is tau + sum_k t_k.
//     The cone is treated natively (Jordan algebra, Nesterov-Todd scaling W = beta (2 v v' - J), closed forms of the CVXOPT
//     cone-programming documentation): its multiplier x_k in Q^{m+1} is state, its slack is rebuilt from (t, theta) and never
//     carries a residual.  (An arrow LMI of order m + 1 = 529 at n = 32 would need a dense 529 x 529 primal block per stage.)
// theta_k and t_k are stage-local: their Schur rows reach P_k, P_{k+1}, tau, alpha and themselves only, and every entry is a Kronecker-
// type Gram product <E_ab, X E_cd S^-1> of the n x n LMI blocks of stage k (computed on the fly, never stored as matrices).  Like
// the multipliers of tmpc_phi.h they ride in block k+1 of the block-cyclic-tridiagonal system behind P_{k+1} (rows oT = d .. d+m:
// theta, row d+m+1: t), so the block kernels of tmpc_cr.h run unchanged on blocks of size d + m + 1.
// With G / C rows in the same solve (convexifier.py:144) the multipliers of tmpc_phi.h sit at rows d .. d + nz, theta behind them (k_t3_cross: the block between the two).
#pragma once
#include "tmpc_common.h"
#include "tmpc_small.h"
#include "tmpc_stage.h"
#include "tmpc_schur.h"

namespace tmpc {

// ---------------------------------------------------------------- second-order cone helpers (one wave, vectors of length m1 = m + 1 in memory)
__device__ __forceinline__ double soc_dot1(const double* u, const double* v, int m1, int lane) {       // sum_{i>=1} u_i v_i
  double a = 0.0;
  for (int i = 1 + lane; i < m1; i += 64) a = fma(u[i], v[i], a);
  return wave_sum(a);
}
__device__ __forceinline__ double soc_det(const double* u, int m1, int lane) { return u[0] * u[0] - soc_dot1(u, u, m1, lane); }
// largest step th with u + th du in Q (u in int Q), returned as the "eigenvalue" -1/th of the step-length convention (0: unbounded)
__device__ __forceinline__ double soc_step_eig(const double* u, const double* du, int m1, int lane) {
  const double a = du[0] * du[0] - soc_dot1(du, du, m1, lane);
  const double bq = u[0] * du[0] - soc_dot1(u, du, m1, lane);
  const double c = soc_det(u, m1, lane);
  double th = 1e300;
  if (du[0] < 0.0) th = fmin(th, -u[0] / du[0]);
  if (fabs(a) < 1e-300) { if (bq < 0.0) th = fmin(th, -c / (2.0 * bq)); }
  else {
    const double disc = bq * bq - a * c;
    if (disc >= 0.0) {
      const double sq = sqrt(disc);
      const double r1 = (-bq - sq) / a, r2 = (-bq + sq) / a;
      if (r1 > 0.0) th = fmin(th, r1);
      if (r2 > 0.0) th = fmin(th, r2);
    }
  }
  return (th < 1e299) ? -1.0 / th : 0.0;
}

struct T3Ptr {       // per-stage views
  double *th, *z, *dth, *dz, *cth;            // [m]
  double *x, *dx, *cq, *g, *v, *lam;          // [m + 1]
  double *t, *dt, *beta;                      // [1]
};
__device__ __forceinline__ T3Ptr t3_at(const WS& w, const Dims& dm, size_t sid) {
  const size_t m = dm.nT, m1 = m + 1;
  T3Ptr q;
  q.th = w.t3th + sid * m; q.z = w.t3z + sid * m; q.dth = w.t3dth + sid * m; q.dz = w.t3dz + sid * m; q.cth = w.t3cth + sid * m;
  q.x = w.t3x + sid * m1; q.dx = w.t3dx + sid * m1; q.cq = w.t3cq + sid * m1; q.g = w.t3g + sid * m1; q.v = w.t3v + sid * m1; q.lam = w.t3lam + sid * m1;
  q.t = w.t3t + sid; q.dt = w.t3dt + sid; q.beta = w.t3beta + sid;
  return q;
}
__device__ __forceinline__ double t3_wr(const WS& w, const double* pr) { return w.rho * pr[P_SBETA] / pr[P_S]; }
// e -> (a, b), a <= b, row-major upper triangle of an n x n matrix
__device__ __forceinline__ void t3_ab(int e, int n, int* a, int* b) {
  int aa = 0, rem = e;
  while (rem >= n - aa) { rem -= n - aa; ++aa; }
  *a = aa; *b = aa + rem;
}

// ---------------------------------------------------------------- init (after k_init_state)
// theta, z and the norm cone start ON the central path of the norm term, like the Step 2 norms: z_e = x0/theta_e, x = x0 s^-1 with x_0 = 1
__global__ void __launch_bounds__(64) k_t3_init(WS w, Dims dm) {
  const size_t sid = blockIdx.x;
  const int lane = threadIdx.x, b = (int)(sid / dm.p), n = dm.n, m = dm.nT, m1 = m + 1;
  const double* pr = w.prob + (size_t)b * PS;
  const T3Ptr q = t3_at(w, dm, sid);
  const double x0 = 1.0 / (double)(dm.p * n), wr = t3_wr(w, pr);
  const double ph = fmin(1.0, x0 * (double)n / wr);            // sqrt(sum c_e^2) = n
  const double an = wr * ph * (double)n;                       // |s_1|
  const double t0 = 0.5 * (x0 + sqrt(x0 * x0 + 4.0 * an * an));
  const double det = t0 * t0 - an * an;
  for (int e = lane; e < m; e += 64) {
    int a, bb; t3_ab(e, n, &a, &bb);
    const double c = (a == bb) ? 1.0 : 1.4142135623730951;
    q.th[e] = ph; q.z[e] = x0 / ph; q.dth[e] = 0.0; q.dz[e] = 0.0; q.cth[e] = 0.0;
    q.x[1 + e] = -x0 * wr * c * ph / det; q.dx[1 + e] = 0.0; q.cq[1 + e] = 0.0;
  }
  if (lane == 0) { q.t[0] = t0; q.dt[0] = 0.0; q.x[0] = x0 * t0 / det; q.dx[0] = 0.0; q.cq[0] = 0.0; }
}

// ---------------------------------------------------------------- after k_stage_pre: scaling of the norm cone, its share of mu and of the residuals
__global__ void __launch_bounds__(64) k_t3_pre(WS w, Dims dm) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int sid = stage_id(w, dm), lane = threadIdx.x;
  const int b = sid / dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] == PH_DONE) return;
  const double* pr = w.prob + (size_t)b * PS;
  const int n = dm.n, nn = n * n, m = dm.nT, m1 = m + 1;
  const T3Ptr q = t3_at(w, dm, sid);
  const double wr = t3_wr(w, pr);
  double* s = sm;                         // [m1] slack of the cone
  for (int e = lane; e < m; e += 64) { int a, bb; t3_ab(e, n, &a, &bb); s[1 + e] = wr * ((a == bb) ? 1.0 : 1.4142135623730951) * q.th[e]; }
  if (lane == 0) s[0] = q.t[0];
  wsync();
  const double sdet = soc_det(s, m1, lane), xdet = soc_det(q.x, m1, lane);
  const double rs = 1.0 / sqrt(sdet), rx = 1.0 / sqrt(xdet);
  const double xs1 = soc_dot1(q.x, s, m1, lane);
  const double xs = q.x[0] * s[0] + xs1;
  const double gam = sqrt(0.5 * (1.0 + xs * rs * rx));
  const double wb0 = (s[0] * rs + q.x[0] * rx) / (2.0 * gam);
  const double nv = 1.0 / sqrt(2.0 * (wb0 + 1.0));
  const double beta = sqrt(sqrt(sdet / xdet));
  for (int i = 1 + lane; i < m1; i += 64) q.v[i] = nv * (s[i] * rs - q.x[i] * rx) / (2.0 * gam);
  if (lane == 0) { q.v[0] = nv * (wb0 + 1.0); q.beta[0] = beta; }
  wsync();
  // lambda = W x
  const double vx = q.v[0] * q.x[0] + soc_dot1(q.v, q.x, m1, lane);
  for (int i = 1 + lane; i < m1; i += 64) q.lam[i] = beta * (2.0 * q.v[i] * vx + q.x[i]);
  if (lane == 0) q.lam[0] = beta * (2.0 * q.v[0] * vx - q.x[0]);
  // complementarity and stationarity residuals: r_theta_e = -<E_e, Y> - z_e - w c_e x_e, r_t = 1 - x_0   (Y = X1 - X2)
  const double* X1 = w.X1 + (size_t)sid * nn; const double* X2 = w.X2 + (size_t)sid * nn;
  double tz = 0.0, r2 = 0.0;
  for (int e = lane; e < m; e += 64) {
    int a, bb; t3_ab(e, n, &a, &bb);
    const double we = (a == bb) ? 1.0 : 2.0, c = (a == bb) ? 1.0 : 1.4142135623730951;
    const double r = -we * (X1[a * n + bb] - X2[a * n + bb]) - q.z[e] - wr * c * q.x[1 + e];
    r2 = fma(r, r, r2);
    tz = fma(q.th[e], q.z[e], tz);
  }
  tz = wave_sum(tz); r2 = wave_sum(r2);
  if (lane == 0) {
    double* pq = w.part + (size_t)sid * NPART;
    pq[Q_XS] += tz + xs;
    const double rt3 = r2 + (1.0 - q.x[0]) * (1.0 - q.x[0]);
    if (dm.nr > 0) { pq[Q_RPHI2] += rt3; pq[Q_NCONE] += (double)(m + 1); }      // after k_phi_pre: the multipliers of G / C are there already
    else { pq[Q_RPHI2] = rt3; pq[Q_NCONE] = (double)(m + 1); }
  }
}

// HKM Gram entry <E_ab, L E_cd R'> for rectangular factors (rows a, b of L / R, columns c, d)
__device__ __forceinline__ double hk_rect(const double* __restrict__ L, const double* __restrict__ R, int ld, int a, int b, int c, int d_) {
  const double xac = L[a * ld + c], xad = L[a * ld + d_], xbc = L[b * ld + c], xbd = L[b * ld + d_];
  const double sac = R[a * ld + c], sad = R[a * ld + d_], sbc = R[b * ld + c], sbd = R[b * ld + d_];
  const double t = (xac * sbd + xad * sbc) + (xbc * sad + xbd * sac);
  return ((a == b) ? 0.5 : 1.0) * ((c == d_) ? 0.5 : 1.0) * t;
}

// ---------------------------------------------------------------- after k_schur (which writes identity in the padding): rows of theta_k, t_k in block k+1
// BIGN (32 < n <= 64): X_r, S_r^-1 are read where they lie in global memory and X_r V', S_r^-1 V' go to the scratch matrices of the generic per-stage
// kernels (WS::bscr, slots 0 .. 3); only the vectors and the index tables are in LDS
constexpr int T3_LD = 33;
template <bool BIGN>
__global__ void __launch_bounds__(256) k_t3_schur(WS w, Dims dm) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int sid = stage_id(w, dm), tid = threadIdx.x;
  const int b = sid / dm.p, k = sid - b * dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] == PH_DONE) return;
  const double* pr = w.prob + (size_t)b * PS;
  const int n = dm.n, nx = dm.nx, nn = n * n, d = dm.d, dp = dm.dp, m = dm.nT, m1 = m + 1, p = dm.p;
  const int oT = d + dm.nz;
  const int kn = (k + 1 == p) ? 0 : k + 1;
  const size_t bs = (size_t)dp * dp;
  const double wr = t3_wr(w, pr);
  const T3Ptr q = t3_at(w, dm, sid);
  // LDS: X_r, Si_r (n x n), XV_r = X_r V', SV_r = Si_r V' (n x nx), all with leading dimension 33; v, J v, w c, theta-pairs
  const double* Xr[2]; const double* Sr[2]; double* XV[2]; double* SV[2];
  const int msz = 32 * T3_LD;
  const int ldx = BIGN ? n : T3_LD, ldv = BIGN ? nx : T3_LD;
  double* vv = BIGN ? sm : sm + 9 * msz;     // [m1] v
  double* wc = vv + m1;                      // [m] w c_e
  short* ea = (short*)(wc + m1); short* eb = ea + m1;       // entry e -> (a, b), n-space
  short* ca = eb + m1; short* cb = ca + d + 1;               // column (cd) -> (c, d), nx-space
  const double* mV;
  if (BIGN) {
    double* scr = w.bscr + (size_t)sid * BIG_SCR * nn;
    Xr[0] = w.X1 + (size_t)sid * nn; Sr[0] = w.S1i + (size_t)sid * nn; Xr[1] = w.X2 + (size_t)sid * nn; Sr[1] = w.S2i + (size_t)sid * nn;
    XV[0] = scr; SV[0] = scr + nn; XV[1] = scr + 2 * nn; SV[1] = scr + 3 * nn;
    mV = w.V + (size_t)sid * nx * n;
  } else {
    double* X0 = sm; double* S0 = sm + msz; double* X1 = sm + 2 * msz; double* S1 = sm + 3 * msz;
    XV[0] = sm + 4 * msz; SV[0] = sm + 5 * msz; XV[1] = sm + 6 * msz; SV[1] = sm + 7 * msz;
    double* sV = sm + 8 * msz;                 // V (nx x n)
    for (int e = tid; e < nn; e += 256) {
      const int i = e / n, j = e - i * n;
      X0[i * T3_LD + j] = w.X1[(size_t)sid * nn + e]; S0[i * T3_LD + j] = w.S1i[(size_t)sid * nn + e];
      X1[i * T3_LD + j] = w.X2[(size_t)sid * nn + e]; S1[i * T3_LD + j] = w.S2i[(size_t)sid * nn + e];
    }
    for (int e = tid; e < nx * n; e += 256) { const int i = e / n, j = e - i * n; sV[i * T3_LD + j] = w.V[(size_t)sid * nx * n + e]; }
    Xr[0] = X0; Sr[0] = S0; Xr[1] = X1; Sr[1] = S1; mV = sV;
  }
  for (int e = tid; e < m1; e += 256) vv[e] = q.v[e];
  for (int e = tid; e < m; e += 256) { int a, bb; t3_ab(e, n, &a, &bb); ea[e] = (short)a; eb[e] = (short)bb; wc[e] = wr * ((a == bb) ? 1.0 : 1.4142135623730951); }
  for (int e = tid; e < d; e += 256) { int a, bb; t3_ab(e, nx, &a, &bb); ca[e] = (short)a; cb[e] = (short)bb; }
  __syncthreads();
  for (int e = tid; e < 2 * n * nx; e += 256) {          // X_r V', Si_r V'
    const int r = e / (n * nx), rem = e - r * n * nx, a = rem / nx, c = rem - a * nx;
    double sx = 0.0, ss = 0.0;
    for (int qq = 0; qq < n; ++qq) { sx = fma(Xr[r][a * ldx + qq], mV[c * ldx + qq], sx); ss = fma(Sr[r][a * ldx + qq], mV[c * ldx + qq], ss); }
    XV[r][a * ldv + c] = sx; SV[r][a * ldv + c] = ss;
  }
  if (BIGN) __threadfence_block();
  __syncthreads();
  const double beta = q.beta[0], ib2 = 1.0 / (beta * beta);
  double vsq = 0.0;
  for (int i = 0; i < m1; ++i) vsq = fma(vv[i], vv[i], vsq);            // v'v (every thread: m1 <= 529 LDS broadcasts)
  double* Dn = w.D + ((size_t)b * p + kn) * bs;
  double* ddn = w.Ddiag + ((size_t)b * p + kn) * dp;
  const bool tr = (w.cr_orient[k] != 0);                   // coupling block stored as T[P_k, P_{k+1}] (rows = block k)
  double* Cg = w.O + (size_t)sid * bs;
  const double* th = q.th; const double* zz = q.z;
  // W^-2[i][j] = (delta_ij + 4 (v'v) Jv_i Jv_j - 2 (Jv_i v_j + v_i Jv_j)) / beta^2,  Jv = (v_0, -v_1)
  const int ncol = oT + m1;                                // columns of the rows to fill (lower part incl. the diagonal)
  for (int idx = tid; idx < m1 * ncol; idx += 256) {
    const int i = idx / ncol, j = idx - i * ncol;          // row oT + i, column j
    if (j > oT + i) continue;                              // upper part of the diagonal block
    double val = 0.0, aval = 0.0;
    bool has_a = false;
    if (i < m) {
      const int a = ea[i], bb = eb[i];
      if (j < d) {                                         // P columns: b-coupling (P_{k+1}) in D, a-coupling (P_k) in the edge slot
        const int c = ca[j], dd_ = cb[j];
        val = hk_rect(XV[0], SV[0], ldv, a, bb, c, dd_) + hk_rect(XV[1], SV[1], ldv, a, bb, c, dd_);
        aval = -(hk_rect(Xr[0], Sr[0], ldx, a, bb, c, dd_) + hk_rect(Xr[1], Sr[1], ldx, a, bb, c, dd_));
        has_a = true;
      } else if (j >= oT) {                                // theta columns f <= e
        const int f = j - oT;
        const int c = ea[f], dd_ = eb[f];
        val = hk_rect(Xr[0], Sr[0], ldx, a, bb, c, dd_) + hk_rect(Xr[1], Sr[1], ldx, a, bb, c, dd_);
        const double jvi = -vv[1 + i], jvf = -vv[1 + f];
        val += wc[i] * wc[f] * ib2 * (((i == f) ? 1.0 : 0.0) + 4.0 * vsq * jvi * jvf - 2.0 * (jvi * vv[1 + f] + vv[1 + i] * jvf));
        if (i == f) val += zz[i] / th[i];
      }                                                    // (columns d .. oT-1: other stage-local variables, none in this version)
    } else {                                               // row of t
      if (j >= oT && j < oT + m) { const int f = j - oT; const double jvf = -vv[1 + f]; val = wc[f] * ib2 * (4.0 * vsq * vv[0] * jvf - 2.0 * (vv[0] * vv[1 + f] + vv[0] * jvf)); }
      else if (j == oT + m) val = ib2 * (1.0 + 4.0 * vsq * vv[0] * vv[0] - 4.0 * vv[0] * vv[0]);
    }
    if (p == 1 && has_a) { val += aval; has_a = false; }   // both couplings land in the one P block
    Dn[(size_t)(oT + i) * dp + j] = val;
    if (j == oT + i) ddn[oT + i] = val;
    if (has_a) { if (tr) Cg[(size_t)j * dp + oT + i] = aval; else Cg[(size_t)(oT + i) * dp + j] = aval; }
  }
}

// ---------------------------------------------------------------- right-hand sides: g = sig_mu s^-1 - x - corr (the target of dx without the ds term)
__global__ void __launch_bounds__(64) k_t3_rhs(WS w, Dims dm, int pass) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int sid = stage_id(w, dm), lane = threadIdx.x;
  const int b = sid / dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return;
  const double* pr = w.prob + (size_t)b * PS;
  const double sig = (pass == 1) ? 0.0 : pr[P_SIGMU];
  const bool use_corr = (pass == 2 && phase == PH_MAIN);
  const int n = dm.n, m = dm.nT, m1 = m + 1;
  const T3Ptr q = t3_at(w, dm, sid);
  const double wr = t3_wr(w, pr);
  double* s = sm;
  for (int e = lane; e < m; e += 64) { int a, bb; t3_ab(e, n, &a, &bb); s[1 + e] = wr * ((a == bb) ? 1.0 : 1.4142135623730951) * q.th[e]; }
  if (lane == 0) s[0] = q.t[0];
  wsync();
  const double sdet = soc_det(s, m1, lane);
  for (int i = lane; i < m1; i += 64) q.g[i] = sig * ((i == 0) ? s[0] : -s[i]) / sdet - q.x[i] - (use_corr ? q.cq[i] : 0.0);
}

// after k_gather (which zero-pads the tails): right-hand side and border-column entries of (theta_k, t_k) in the vectors of block k+1
__global__ void __launch_bounds__(64) k_t3_gather(WS w, Dims dm, int pass) {
  const int sid = stage_id(w, dm), lane = threadIdx.x;
  const int b = sid / dm.p, k = sid - b * dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return;
  const bool three = (pass == 1) || (phase != PH_MAIN && !ip[I_CHORD]);
  const double* pr = w.prob + (size_t)b * PS;
  const double sig = (pass == 1) ? 0.0 : pr[P_SIGMU];
  const bool use_corr = (pass == 2 && phase == PH_MAIN);
  const int n = dm.n, nn = n * n, m = dm.nT, oT = dm.d + dm.nz;
  const int kn = (k + 1 == dm.p) ? 0 : k + 1;
  const T3Ptr q = t3_at(w, dm, sid);
  const double wr = t3_wr(w, pr);
  const double* T1 = w.T1 + (size_t)sid * nn; const double* T2 = w.T2 + (size_t)sid * nn;
  const double* Psi = w.t3psi + (size_t)sid * nn; const double* Phi = w.t3phi + (size_t)sid * nn;
  const size_t v0 = ((size_t)b * dm.p + kn) * dm.dp + oT;
  for (int e = lane; e <= m; e += 64) {
    double r, ct = 0.0, ca = 0.0;
    if (e < m) {
      int a, bb; t3_ab(e, n, &a, &bb);
      const double we = (a == bb) ? 1.0 : 2.0, c = (a == bb) ? 1.0 : 1.4142135623730951;
      r = we * (T1[a * n + bb] - T2[a * n + bb]) + sig / q.th[e] - (use_corr ? q.cth[e] : 0.0) + wr * c * (q.g[1 + e] + q.x[1 + e]);
      ct = -we * Psi[a * n + bb]; ca = we * Phi[a * n + bb];
    } else r = (q.g[0] + q.x[0]) - 1.0;
    const size_t vi = v0 + e;
    if (three) {
      double* w3 = w.W3 + vi * 3; w3[0] = r; w3[1] = ct; w3[2] = ca;
      double* u = w.U + vi * 2; u[0] = ct; u[1] = ca;
    } else w.Z[vi] = r;
  }
}

// ---------------------------------------------------------------- after the block solve, before k_stage_dir: d theta, dz, dt, dx; Mehrotra terms in pass 1
__global__ void __launch_bounds__(64) k_t3_dir(WS w, Dims dm, int pass) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int sid = stage_id(w, dm), lane = threadIdx.x;
  const int b = sid / dm.p, k = sid - b * dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return;
  const double* pr = w.prob + (size_t)b * PS;
  const double dtau = pr[P_DTAU], dalpha = pr[P_DALPHA];
  const double sig = (pass == 1) ? 0.0 : pr[P_SIGMU];
  const bool use_corr = (pass == 2 && phase == PH_MAIN);
  const int n = dm.n, m = dm.nT, m1 = m + 1, oT = dm.d + dm.nz;
  const int kn = (k + 1 == dm.p) ? 0 : k + 1;
  const T3Ptr q = t3_at(w, dm, sid);
  const double wr = t3_wr(w, pr);
  double* ds = sm;                // [m1]
  double* t1 = sm + m1;           // [m1] scratch
  double* t2 = t1 + m1;           // [m1] scratch
  const size_t v0 = ((size_t)b * dm.p + kn) * dm.dp + oT;
  for (int e = lane; e <= m; e += 64) {
    const size_t vi = v0 + e;
    const double v = w.Z[vi] - w.TU[vi * 2] * dtau - w.TU[vi * 2 + 1] * dalpha;
    if (e < m) {
      int a, bb; t3_ab(e, n, &a, &bb);
      q.dth[e] = v;
      const double dz = sig / q.th[e] - q.z[e] - q.z[e] * v / q.th[e] - (use_corr ? q.cth[e] : 0.0);
      q.dz[e] = dz;
      if (pass == 1) q.cth[e] = dz * v / q.th[e];
      ds[1 + e] = wr * ((a == bb) ? 1.0 : 1.4142135623730951) * v;
    } else { q.dt[0] = v; ds[0] = v; }
  }
  wsync();
  // dx = g - W^-2 ds,  W^-2 u = (u + 4 (v'v) Jv (Jv'u) - 2 (Jv (v'u) + v (Jv'u))) / beta^2
  const double beta = q.beta[0], ib2 = 1.0 / (beta * beta);
  const double vv1 = soc_dot1(q.v, q.v, m1, lane), vsq = q.v[0] * q.v[0] + vv1;
  const double vu1 = soc_dot1(q.v, ds, m1, lane);
  const double vu = q.v[0] * ds[0] + vu1, jvu = q.v[0] * ds[0] - vu1;
  for (int i = lane; i < m1; i += 64) {
    const double jv = (i == 0) ? q.v[0] : -q.v[i];
    q.dx[i] = q.g[i] - ib2 * (ds[i] + 4.0 * vsq * jv * jvu - 2.0 * (jv * vu + q.v[i] * jvu));
  }
  wsync();
  if (pass == 1) {
    // Mehrotra term of the cone: corr = W^-1 (lambda \ ((W dx) o (W^-1 ds)))
    const double vdx1 = soc_dot1(q.v, q.dx, m1, lane), vdx = q.v[0] * q.dx[0] + vdx1;
    for (int i = lane; i < m1; i += 64) {
      t1[i] = beta * (2.0 * q.v[i] * vdx + ((i == 0) ? -q.dx[0] : q.dx[i]));                           // W dx
      const double jv = (i == 0) ? q.v[0] : -q.v[i];
      t2[i] = (2.0 * jv * jvu + ((i == 0) ? -ds[0] : ds[i])) / beta;                                   // W^-1 ds
    }
    wsync();
    const double p0 = t1[0] * t2[0] + soc_dot1(t1, t2, m1, lane);           // (W dx) o (W^-1 ds) = (p0; a0 b1 + b0 a1)
    const double a0 = t1[0], b0 = t2[0];
    wsync();
    for (int i = 1 + lane; i < m1; i += 64) t1[i] = a0 * t2[i] + b0 * t1[i];
    wsync();
    if (lane == 0) t1[0] = p0;
    wsync();
    // u = lambda \ r  (r = t1)
    const double* lam = q.lam;
    const double ldet = soc_det(lam, m1, lane);
    const double lr1 = soc_dot1(lam, t1, m1, lane);
    const double r0 = t1[0];
    wsync();
    for (int i = 1 + lane; i < m1; i += 64) t2[i] = (-r0 * lam[i] + (ldet * t1[i] + lr1 * lam[i]) / lam[0]) / ldet;
    if (lane == 0) t2[0] = (lam[0] * r0 - lr1) / ldet;
    wsync();
    // corr = W^-1 u
    const double vt1 = soc_dot1(q.v, t2, m1, lane), jvt = q.v[0] * t2[0] - vt1;
    for (int i = lane; i < m1; i += 64) {
      const double jv = (i == 0) ? q.v[0] : -q.v[i];
      q.cq[i] = (2.0 * jv * jvt + ((i == 0) ? -t2[0] : t2[i])) / beta;
    }
  }
}

// ---------------------------------------------------------------- after k_eigmin: step lengths and complementarity sums of theta and the norm cone
__global__ void __launch_bounds__(64) k_t3_steps(WS w, Dims dm, int pass) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int sid = stage_id(w, dm), lane = threadIdx.x;
  const int b = sid / dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return;
  const double* pr = w.prob + (size_t)b * PS;
  const int n = dm.n, m = dm.nT, m1 = m + 1;
  const T3Ptr q = t3_at(w, dm, sid);
  const double wr = t3_wr(w, pr);
  double* s = sm; double* ds = sm + m1;
  double ls = 0.0, lx = 0.0, dxs = 0.0, xds = 0.0, dxds = 0.0;
  for (int e = lane; e < m; e += 64) {
    int a, bb; t3_ab(e, n, &a, &bb);
    const double c = wr * ((a == bb) ? 1.0 : 1.4142135623730951);
    s[1 + e] = c * q.th[e]; ds[1 + e] = c * q.dth[e];
    ls = fmin(ls, q.dth[e] / q.th[e]); lx = fmin(lx, q.dz[e] / q.z[e]);      // "eigenvalues" of the 1 x 1 blocks: step = -1/lambda
    dxs += q.dz[e] * q.th[e]; xds += q.z[e] * q.dth[e]; dxds += q.dz[e] * q.dth[e];
  }
  if (lane == 0) { s[0] = q.t[0]; ds[0] = q.dt[0]; }
  wsync();
  ls = wave_min(ls); lx = wave_min(lx);
  dxs = wave_sum(dxs); xds = wave_sum(xds); dxds = wave_sum(dxds);
  ls = fmin(ls, soc_step_eig(s, ds, m1, lane));
  lx = fmin(lx, soc_step_eig(q.x, q.dx, m1, lane));
  dxs += q.dx[0] * s[0] + soc_dot1(q.dx, s, m1, lane);
  xds += q.x[0] * ds[0] + soc_dot1(q.x, ds, m1, lane);
  dxds += q.dx[0] * ds[0] + soc_dot1(q.dx, ds, m1, lane);
  if (lane == 0) {
    double* e = w.eigmin + (size_t)sid * 4;
    e[0] = fmin(e[0], ls); e[1] = fmin(e[1], lx);
    double* pq = w.part + (size_t)sid * NPART;
    pq[Q_DXS] += dxs; pq[Q_XDS] += xds; pq[Q_DXDS] += dxds;
  }
}

// with k_update
__global__ void __launch_bounds__(64) k_t3_update(WS w, Dims dm) {
  const int sid = stage_id(w, dm), lane = threadIdx.x;
  const int b = sid / dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] == PH_DONE) return;
  const double* pr = w.prob + (size_t)b * PS;
  const double ap = pr[P_AP], ad = pr[P_AD];
  if (ap == 0.0 && ad == 0.0) return;      // discarded direction
  const int m = dm.nT;
  const T3Ptr q = t3_at(w, dm, sid);
  for (int e = lane; e < m; e += 64) { q.th[e] += ad * q.dth[e]; q.z[e] += ap * q.dz[e]; }
  for (int i = lane; i <= m; i += 64) q.x[i] += ap * q.dx[i];
  if (lane == 0) q.t[0] += ad * q.dt[0];
}

// ---------------------------------------------------------------- Step 3 together with the multipliers of G / C (convexifier.py:144 passes both):
// the block between the entries theta of T_k and the multipliers phi_k,i (direction g_i g_i') in D_{k+1}:
//   <E_ab, Phi(g g')> = sum_r (w_a u_b + w_b u_a)   (a < b),   w_a u_a   (a = b),     w = X_r g, u = S_r^-1 g  (pvec of tmpc_phi.h)
// rows oT + e (after the multipliers, which sit at d .. d + nz), columns d + i; after k_aug_fill and k_t3_schur.
__global__ void __launch_bounds__(64) k_t3_cross(WS w, Dims dm) {
  const int sid = stage_id(w, dm), lane = threadIdx.x;
  const int b = sid / dm.p, k = sid - b * dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] == PH_DONE) return;
  const int n = dm.n, m = dm.nT, d = dm.d, dp = dm.dp, oT = d + dm.nz;
  const int kn = (k + 1 == dm.p) ? 0 : k + 1;
  double* Dn = w.D + ((size_t)b * dm.p + kn) * dp * dp;
  const int rows = stage_rows(w, dm, sid);
  const int pvl = 2 * n + 2 * dm.nx;
  for (int e = lane; e < m; e += 64) {
    int a, bb; t3_ab(e, n, &a, &bb);
    for (int i = 0; i < rows; ++i) {
      double v = 0.0;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const double* pv = w.pvec + (((size_t)sid * 2 + r) * dm.nr + i) * (size_t)pvl;      // w = pv[0..n), u = pv[n..2n)
        v += (a == bb) ? pv[a] * pv[n + a] : pv[a] * pv[n + bb] + pv[bb] * pv[n + a];
      }
      Dn[(size_t)(oT + e) * dp + d + i] = v;
    }
  }
}

}  // namespace tmpc
