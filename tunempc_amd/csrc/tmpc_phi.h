// Equality-constraint term of Step 1 (reference: convexifier.py:249-255, :346-347, :409-411).
//
// With G given, M_k = alpha*Hb_k + calH_k(P) + G_k' diag(phi_k) G_k with cost-free multipliers phi_k = s*Fg_k >= 0 (ng per
// stage; their slack is phi itself, dual z).  phi_k touches only the two cone blocks of stage k, so in the HKM Schur
// system its columns reach P_k, P_{k+1}, tau and alpha only.  They are eliminated stage by stage BEFORE the block
// factorisation (K_k = T_phiphi^-1, ng x ng):
//     D_k     -= a_k K_k a_k' + b_{k-1} K_{k-1} b_{k-1}'        a_k,i = -svec(W_i[:nx,:nx]),  b_k,i = svec(V_k W_i V_k'),
//     C_k     -= a_k K_k b_k'                                   W_i = sum_r sym(X_r g_i g_i' S_r^-1) = sum_r sym(w_ri u_ri')
//     u_tau, u_alpha, rhs  -= a_k K_k (c_tau | c_alpha | r_phi)_k + b_{k-1} K_{k-1} (...)_{k-1},   and the 2 x 2 border likewise,
// so that k_schur / k_factor / k_solve stay what they are; afterwards dphi_k = K_k (r_phi - T_phi,y dy).
// Every kernel here is one single-wave workgroup per stage and only runs when dm.ng > 0.
#pragma once
#include "tmpc_common.h"
#include "tmpc_small.h"
#include "tmpc_stage.h"

namespace tmpc {

__device__ __forceinline__ int pv_len(const Dims& dm) { return 2 * dm.n + 2 * dm.nx; }
__device__ __forceinline__ double* pv_at(double* pvec, const Dims& dm, size_t sid, int r, int i) {
  return pvec + ((sid * 2 + r) * dm.ng + i) * (size_t)pv_len(dm);
}
// psm offsets (doubles): K, c_tau, c_alpha, K c_tau, K c_alpha, r_phi, K r_phi
__device__ __forceinline__ double* psm_at(double* psm, const Dims& dm, size_t sid) { return psm + sid * (size_t)(dm.ng * dm.ng + 8 * dm.ng); }
#define PSM_K(q) (q)
#define PSM_CT(q, g) ((q) + (g) * (g))
#define PSM_CA(q, g) ((q) + (g) * (g) + (g))
#define PSM_KCT(q, g) ((q) + (g) * (g) + 2 * (g))
#define PSM_KCA(q, g) ((q) + (g) * (g) + 3 * (g))
#define PSM_RPHI(q, g) ((q) + (g) * (g) + 4 * (g))
#define PSM_KR(q, g) ((q) + (g) * (g) + 5 * (g))

// after k_init_state
__global__ void __launch_bounds__(64) k_phi_init(WS w, Dims dm) {
  const size_t sid = blockIdx.x;
  const int lane = threadIdx.x;
  if (lane < dm.ng) {
    w.phi[sid * dm.ng + lane] = 1.0;
    w.zph[sid * dm.ng + lane] = 1.0 / (double)(dm.p * dm.n);
    w.corrp[sid * dm.ng + lane] = 0.0;
  }
}

constexpr int PHI_SLOTS = 6;
// after k_stage_pre (needs S_r^-1), before k_ctrl_a: vectors w, u, V w, V u; K; c_tau, c_alpha; border / mu / pinf partials
__global__ void __launch_bounds__(64) k_phi_pre(WS w, Dims dm) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int sid = blockIdx.x, lane = threadIdx.x;
  const int b = sid / dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] == PH_DONE) return;
  const int n = dm.n, nx = dm.nx, nn = n * n, ng = dm.ng;
  double* sX = sm; double* sSi = sm + MS; double* sHb = sm + 2 * MS; double* sV = sm + 3 * MS;
  double* gl = sm + 4 * MS;                 // [ng][NMAX]
  double* wl = gl + NGM * NMAX;             // [2][ng][NMAX]
  double* ul = wl + 2 * NGM * NMAX;         // [2][ng][NMAX]
  double* sc = ul + 2 * NGM * NMAX;         // GXG[2][ng][ng], GSG[2][ng][ng], ct[ng], ca[ng]
  const double* Gg = w.G + (size_t)sid * ng * n;
  for (int e = lane; e < ng * n; e += 64) gl[(e / n) * NMAX + (e % n)] = Gg[e];
  g2s(sHb, w.Hb + (size_t)sid * nn, n, n, n, lane);
  g2s(sV, w.V + (size_t)sid * nx * n, nx, n, n, lane);
  double* GXG = sc; double* GSG = sc + 2 * NGM * NGM; double* ctl = GSG + 2 * NGM * NGM; double* cal = ctl + NGM;
  if (lane < NGM) { ctl[lane] = 0.0; cal[lane] = 0.0; }
  for (int r = 0; r < 2; ++r) {
    g2s(sX, (r ? w.X2 : w.X1) + (size_t)sid * nn, n, n, n, lane);
    g2s(sSi, (r ? w.S2i : w.S1i) + (size_t)sid * nn, n, n, n, lane);
    for (int i = 0; i < ng; ++i) {
      if (lane < n) {
        double a0 = 0.0, a1 = 0.0;
        for (int c = 0; c < n; ++c) { const double g = gl[i * NMAX + c]; a0 = fma(sX[lane * LD + c], g, a0); a1 = fma(sSi[lane * LD + c], g, a1); }
        wl[(r * NGM + i) * NMAX + lane] = a0; ul[(r * NGM + i) * NMAX + lane] = a1;
      }
    }
    wsync();
    for (int i = 0; i < ng; ++i) {
      double* pv = pv_at(w.pvec, dm, sid, r, i);
      const double* wi = wl + (r * NGM + i) * NMAX; const double* ui = ul + (r * NGM + i) * NMAX;
      if (lane < n) { pv[lane] = wi[lane]; pv[n + lane] = ui[lane]; }
      if (lane < nx) {
        double a0 = 0.0, a1 = 0.0;
        for (int c = 0; c < n; ++c) { a0 = fma(sV[lane * LD + c], wi[c], a0); a1 = fma(sV[lane * LD + c], ui[c], a1); }
        pv[2 * n + lane] = a0; pv[2 * n + nx + lane] = a1;
      }
      // c_alpha_i += w' Hb u ;  c_tau_i = -(w_2 . u_2)
      double hu = 0.0;
      if (lane < n) { for (int c = 0; c < n; ++c) hu = fma(sHb[lane * LD + c], ui[c], hu); hu *= wi[lane]; }
      hu = wave_sum(hu);
      double wu = (lane < n) ? wi[lane] * ui[lane] : 0.0;
      wu = wave_sum(wu);
      if (lane == 0) { cal[i] += hu; if (r == 1) ctl[i] = -wu; }
      for (int j = 0; j < ng; ++j) {
        double x = (lane < n) ? gl[i * NMAX + lane] * wl[(r * NGM + j) * NMAX + lane] : 0.0;
        double y = (lane < n) ? gl[i * NMAX + lane] * ul[(r * NGM + j) * NMAX + lane] : 0.0;
        x = wave_sum(x); y = wave_sum(y);
        if (lane == 0) { GXG[(r * NGM + i) * NGM + j] = x; GSG[(r * NGM + i) * NGM + j] = y; }
      }
    }
    wsync();
  }
  if (lane == 0) {
    double* q = psm_at(w.psm, dm, sid);
    const double* phi = w.phi + (size_t)sid * ng; const double* z = w.zph + (size_t)sid * ng;
    double T[NGM][NGM], Ki[NGM][NGM];
    for (int i = 0; i < ng; ++i)
      for (int j = 0; j < ng; ++j) {
        double t = 0.0;
        for (int r = 0; r < 2; ++r)
          t += 0.5 * (GXG[(r * NGM + i) * NGM + j] * GSG[(r * NGM + j) * NGM + i] + GXG[(r * NGM + j) * NGM + i] * GSG[(r * NGM + i) * NGM + j]);
        T[i][j] = t + ((i == j) ? z[i] / phi[i] : 0.0);
        Ki[i][j] = (i == j) ? 1.0 : 0.0;
      }
    for (int c = 0; c < ng; ++c) {           // Gauss-Jordan (T is symmetric positive definite)
      const double piv = 1.0 / T[c][c];
      for (int j = 0; j < ng; ++j) { T[c][j] *= piv; Ki[c][j] *= piv; }
      for (int i = 0; i < ng; ++i)
        if (i != c) { const double f = T[i][c]; for (int j = 0; j < ng; ++j) { T[i][j] -= f * T[c][j]; Ki[i][j] -= f * Ki[c][j]; } }
    }
    double xs = 0.0, rp2 = 0.0, ctkct = 0.0, ctkca = 0.0, cakca = 0.0;
    for (int i = 0; i < ng; ++i) {
      double kct = 0.0, kca = 0.0;
      for (int j = 0; j < ng; ++j) { PSM_K(q)[i * ng + j] = Ki[i][j]; kct += Ki[i][j] * ctl[j]; kca += Ki[i][j] * cal[j]; }
      PSM_CT(q, ng)[i] = ctl[i]; PSM_CA(q, ng)[i] = cal[i]; PSM_KCT(q, ng)[i] = kct; PSM_KCA(q, ng)[i] = kca;
      ctkct += ctl[i] * kct; ctkca += ctl[i] * kca; cakca += cal[i] * kca;
      xs += phi[i] * z[i];
      const double rphi = -(GXG[(0 * NGM + i) * NGM + i] - GXG[(1 * NGM + i) * NGM + i]) - z[i];     // stationarity residual of phi_i
      rp2 += rphi * rphi;
    }
    double* pq = w.part + (size_t)sid * NPART;
    pq[Q_XS] += xs; pq[Q_RPHI2] = rp2;
    pq[Q_TRPSI] -= ctkct;       // b_tt
    pq[Q_TRPHI2] += ctkca;      // b_ta = -trphi2
    pq[Q_HBPHI] -= cakca;       // b_aa (+ x0/s0)
  }
}

// svec coordinates of a_i (P_k rows) / b_i (P_{k+1} rows) of stage `sid` at entry (a, c), a <= c
__device__ __forceinline__ double phi_avec(const double* pvec, const Dims& dm, size_t sid, int i, int a, int c) {
  const int n = dm.n;
  double v = 0.0;
  for (int r = 0; r < 2; ++r) {
    const double* pv = pvec + ((sid * 2 + r) * dm.ng + i) * (size_t)(2 * dm.n + 2 * dm.nx);
    v += 0.5 * (pv[a] * pv[n + c] + pv[c] * pv[n + a]);
  }
  return -((a == c) ? v : 2.0 * v);
}
__device__ __forceinline__ double phi_bvec(const double* pvec, const Dims& dm, size_t sid, int i, int a, int c) {
  const int n = dm.n, nx = dm.nx;
  double v = 0.0;
  for (int r = 0; r < 2; ++r) {
    const double* pv = pvec + ((sid * 2 + r) * dm.ng + i) * (size_t)(2 * dm.n + 2 * dm.nx);
    v += 0.5 * (pv[2 * n + a] * pv[2 * n + nx + c] + pv[2 * n + c] * pv[2 * n + nx + a]);
  }
  return (a == c) ? v : 2.0 * v;
}

// after k_schur, before k_factor: rank-ng corrections of D_k (lower triangle + pivot reference) and of the coupling block
__global__ void __launch_bounds__(256) k_phi_schur(WS w, Dims dm) {
  const int sid = blockIdx.x;
  const int b = sid / dm.p, k = sid - b * dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] == PH_DONE) return;
  const int tid = threadIdx.x, nx = dm.nx, d = dm.d, dp = dm.dp, ng = dm.ng;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* av = sm; double* Kav = av + ng * d; double* bm = Kav + ng * d; double* Kbm = bm + ng * d; double* Kbv = Kbm + ng * d;
  const int km = (k == 0) ? dm.p - 1 : k - 1;
  const size_t sm1 = (size_t)b * dm.p + km;
  double* qk = psm_at(w.psm, dm, sid); double* qm = psm_at(w.psm, dm, sm1);
  // enumerate idx -> (a, c)
  for (int idx = tid; idx < d; idx += 256) {
    int a = 0, rem = idx;
    while (rem >= nx - a) { rem -= nx - a; ++a; }
    const int c = a + rem;
    double ai[NGM], bi[NGM], bmi[NGM];
    for (int i = 0; i < ng; ++i) { ai[i] = phi_avec(w.pvec, dm, sid, i, a, c); bi[i] = phi_bvec(w.pvec, dm, sid, i, a, c); bmi[i] = phi_bvec(w.pvec, dm, sm1, i, a, c); }
    for (int i = 0; i < ng; ++i) {
      double ka = 0.0, kb = 0.0, kbm = 0.0;
      for (int j = 0; j < ng; ++j) { ka += PSM_K(qk)[i * ng + j] * ai[j]; kb += PSM_K(qk)[i * ng + j] * bi[j]; kbm += PSM_K(qm)[i * ng + j] * bmi[j]; }
      av[i * d + idx] = ai[i]; Kav[i * d + idx] = ka; bm[i * d + idx] = bmi[i]; Kbm[i * d + idx] = kbm; Kbv[i * d + idx] = kb;
    }
  }
  __syncthreads();
  double* Dg = w.D + (size_t)sid * dp * dp;
  const bool corner = (k == dm.p - 1);
  double* Cg = corner ? (w.F + (size_t)(b * dm.p) * dp * dp) : (w.O + (size_t)sid * dp * dp);
  double* dd = w.Ddiag + (size_t)sid * dp;
  for (int e = tid; e < d * d; e += 256) {
    const int row = e / d, col = e - row * d;
    const size_t o = (size_t)row * dp + col;
    // coupling block C_k[x][y] -= sum_i a_i[x] (K b)_i[y]; stored entry (row, col) is C_k[row][col] for the corner, C_k[col][row] otherwise
    {
      const int x = corner ? row : col, y = corner ? col : row;
      double cv = 0.0;
      for (int i = 0; i < ng; ++i) cv = fma(av[i * d + x], Kbv[i * d + y], cv);
      Cg[o] -= cv;
    }
    if (col <= row) {
      double dv = 0.0;
      for (int i = 0; i < ng; ++i) { dv = fma(av[i * d + row], Kav[i * d + col], dv); dv = fma(bm[i * d + row], Kbm[i * d + col], dv); }
      const double nv = Dg[o] - dv;
      Dg[o] = nv;
      if (row == col) dd[row] = nv;
    }
  }
}

// after k_stage_rhs, before k_gather / k_solve: r_phi, K r_phi, and the eliminated part of the border right-hand sides
__global__ void __launch_bounds__(64) k_phi_rhs(WS w, Dims dm, int pass) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int sid = blockIdx.x, lane = threadIdx.x;
  const int b = sid / dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return;
  const double* pr = w.prob + (size_t)b * PS;
  const double sig = (pass == 1) ? 0.0 : pr[P_SIGMU];
  const bool use_corr = (pass == 2 && phase == PH_MAIN);
  const int n = dm.n, nn = n * n, ng = dm.ng;
  double* sG = sm;                         // T1 - T2
  const double* T1 = w.T1 + (size_t)sid * nn; const double* T2 = w.T2 + (size_t)sid * nn;
  for (int e = lane; e < nn; e += 64) { const int i = e / n, j = e - i * n; sG[i * LD + j] = T1[e] - T2[e]; }
  wsync();
  const double* Gg = w.G + (size_t)sid * ng * n;
  double* q = psm_at(w.psm, dm, sid);
  double rl[NGM];
  for (int i = 0; i < ng; ++i) {
    double t = 0.0;
    if (lane < n) { for (int c = 0; c < n; ++c) t = fma(sG[lane * LD + c], Gg[i * n + c], t); t *= Gg[i * n + lane]; }
    rl[i] = wave_sum(t);
  }
  if (lane == 0) {
    const double* phi = w.phi + (size_t)sid * ng; const double* cp = w.corrp + (size_t)sid * ng;
    double r[NGM];
    for (int i = 0; i < ng; ++i) { r[i] = rl[i] + sig / phi[i] - (use_corr ? cp[i] : 0.0); PSM_RPHI(q, ng)[i] = r[i]; }
    double ctkr = 0.0, cakr = 0.0;
    for (int i = 0; i < ng; ++i) {
      double kr = 0.0;
      for (int j = 0; j < ng; ++j) kr += PSM_K(q)[i * ng + j] * r[j];
      PSM_KR(q, ng)[i] = kr;
      ctkr += PSM_CT(q, ng)[i] * kr; cakr += PSM_CA(q, ng)[i] * kr;
    }
    double* pq = w.part + (size_t)sid * NPART;
    pq[Q_TRT2] -= ctkr;     // rhs_tau   = sum trT2 - 1
    pq[Q_HBG] -= cakr;      // rhs_alpha = sum <Hb, T1 - T2> + t0
  }
}

// after k_gather, before k_solve: eliminated part of the P-block right-hand side and of the two border columns
__global__ void __launch_bounds__(64) k_phi_gather(WS w, Dims dm, int pass) {
  const int sid = blockIdx.x, lane = threadIdx.x;
  const int b = sid / dm.p, k = sid - b * dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return;
  const bool three = (pass == 1) || (phase != PH_MAIN);
  const int nx = dm.nx, dp = dm.dp, ng = dm.ng;
  const int km = (k == 0) ? dm.p - 1 : k - 1;
  const size_t sm1 = (size_t)b * dm.p + km;
  double* qk = psm_at(w.psm, dm, sid); double* qm = psm_at(w.psm, dm, sm1);
  int e = 0;
  for (int a = 0; a < nx; ++a) {
    for (int c = a + lane; c < nx; c += 64) {
      const int idx = e + (c - a);
      double g = 0.0, ut = 0.0, ua = 0.0;
      for (int i = 0; i < ng; ++i) {
        const double ai = phi_avec(w.pvec, dm, sid, i, a, c), bi = phi_bvec(w.pvec, dm, sm1, i, a, c);
        g += ai * PSM_KR(qk, ng)[i] + bi * PSM_KR(qm, ng)[i];
        ut += ai * PSM_KCT(qk, ng)[i] + bi * PSM_KCT(qm, ng)[i];
        ua += ai * PSM_KCA(qk, ng)[i] + bi * PSM_KCA(qm, ng)[i];
      }
      if (three) {
        double* w3 = w.W3 + ((size_t)sid * dp + idx) * 3;
        w3[0] -= g; w3[1] -= ut; w3[2] -= ua;
        double* u = w.U + ((size_t)sid * dp + idx) * 2;
        u[0] -= ut; u[1] -= ua;
      } else {
        w.Z[(size_t)sid * dp + idx] -= g;
      }
    }
    e += nx - a;
  }
}

// after k_solve, before k_stage_dir: dphi = K (r_phi - T_phi,y dy), dz; Mehrotra second-order term in pass 1
__global__ void __launch_bounds__(64) k_phi_dir(WS w, Dims dm, int pass) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int sid = blockIdx.x, lane = threadIdx.x;
  const int b = sid / dm.p, k = sid - b * dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return;
  const double* pr = w.prob + (size_t)b * PS;
  const double dtau = pr[P_DTAU], dalpha = pr[P_DALPHA];
  const double sig = (pass == 1) ? 0.0 : pr[P_SIGMU];
  const bool use_corr = (pass == 2 && phase == PH_MAIN);
  const int n = dm.n, nx = dm.nx, nn = n * n, nxx = nx * nx, ng = dm.ng;
  double* sV = sm; double* sM = sm + MS; double* t0 = sm + 2 * MS; double* t1 = sm + 3 * MS; double* sHb = sm + 4 * MS;
  const int kn = (k + 1 == dm.p) ? 0 : k + 1;
  g2s(sV, w.V + (size_t)sid * nx * n, nx, n, n, lane);
  g2s(sHb, w.Hb + (size_t)sid * nn, n, n, n, lane);
  build_M(sM, sV, t0, t1, sHb, w.dP + (size_t)sid * nxx, w.dP + (size_t)(b * dm.p + kn) * nxx, dalpha, n, nx, lane);   // dalpha*Hb + calH(dP)
  double* q = psm_at(w.psm, dm, sid);
  double tl[NGM];
  for (int i = 0; i < ng; ++i) {
    double t = 0.0;
    for (int r = 0; r < 2; ++r) {
      const double* pv = pv_at(w.pvec, dm, sid, r, i);       // w = pv[0..n), u = pv[n..2n)
      double x = 0.0;
      if (lane < n) { for (int c = 0; c < n; ++c) x = fma(sM[lane * LD + c], pv[n + c], x); x *= pv[lane]; }
      t += wave_sum(x);
    }
    tl[i] = t + dtau * PSM_CT(q, ng)[i];
  }
  if (lane == 0) {
    const double* phi = w.phi + (size_t)sid * ng; const double* z = w.zph + (size_t)sid * ng;
    double* cp = w.corrp + (size_t)sid * ng;
    double* dph = w.dphi + (size_t)sid * ng; double* dzp = w.dzph + (size_t)sid * ng;
    for (int i = 0; i < ng; ++i) {
      double v = 0.0;
      for (int j = 0; j < ng; ++j) v += PSM_K(q)[i * ng + j] * (PSM_RPHI(q, ng)[j] - tl[j]);
      dph[i] = v;
    }
    for (int i = 0; i < ng; ++i) {
      const double dz = sig / phi[i] - z[i] - z[i] * dph[i] / phi[i] - (use_corr ? cp[i] : 0.0);
      dzp[i] = dz;
    }
    if (pass == 1) for (int i = 0; i < ng; ++i) cp[i] = dzp[i] * dph[i] / phi[i];
  }
}

// after k_stage_dir and k_eigmin, before k_ctrl_b / k_ctrl_c: the linear cone joins the step-length minima and the mu_aff sums
__global__ void __launch_bounds__(64) k_phi_steps(WS w, Dims dm, int pass) {
  const int sid = blockIdx.x * 64 + threadIdx.x;
  if (sid >= dm.B * dm.p) return;
  const int b = sid / dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return;
  const int ng = dm.ng;
  const double* phi = w.phi + (size_t)sid * ng; const double* z = w.zph + (size_t)sid * ng;
  const double* dph = w.dphi + (size_t)sid * ng; const double* dzp = w.dzph + (size_t)sid * ng;
  double ls = 1e300, lx = 1e300, dxs = 0.0, xds = 0.0, dxds = 0.0;
  for (int i = 0; i < ng; ++i) {
    ls = fmin(ls, dph[i] / phi[i]); lx = fmin(lx, dzp[i] / z[i]);       // "eigenvalues" of the 1 x 1 blocks: step = -1/lambda
    dxs += dzp[i] * phi[i]; xds += z[i] * dph[i]; dxds += dzp[i] * dph[i];
  }
  double* e = w.eigmin + (size_t)sid * 4;
  e[0] = fmin(e[0], ls); e[1] = fmin(e[1], lx);
  double* pq = w.part + (size_t)sid * NPART;
  pq[Q_DXS] += dxs; pq[Q_XDS] += xds; pq[Q_DXDS] += dxds;
}

// with k_update
__global__ void __launch_bounds__(64) k_phi_update(WS w, Dims dm) {
  const int sid = blockIdx.x * 64 + threadIdx.x;
  if (sid >= dm.B * dm.p) return;
  const int b = sid / dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] == PH_DONE) return;
  const double* pr = w.prob + (size_t)b * PS;
  const double ap = pr[P_AP], ad = pr[P_AD];
  for (int i = 0; i < dm.ng; ++i) {
    w.phi[(size_t)sid * dm.ng + i] += ad * w.dphi[(size_t)sid * dm.ng + i];
    w.zph[(size_t)sid * dm.ng + i] += ap * w.dzph[(size_t)sid * dm.ng + i];
  }
}

}  // namespace tmpc
