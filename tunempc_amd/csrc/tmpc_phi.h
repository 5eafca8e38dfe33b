// Stage-local multipliers of the convexifier SDP:
//   * the equality-constraint term of every step (reference: convexifier.py:249-255, :346-347, :409-411), and
//   * Step 2 (:116-131): the multipliers F_k >= 0 of the active-constraint Jacobians C_k (:258-266, :348-350, :415-420) and the
//     objective terms rho*||F_k||, rho*||Fg_k|| (:276-283).
//
// M_k = alpha*Hb_k + calH_k(P) + J_k' diag(phi_k) J_k with J_k = [G_k; C_k] and phi_k = s*[Fg_k; F_k] >= 0 (their slack is phi
// itself, dual z).  Each norm term is an epigraph variable t with the arrow LMI S = [[t, w v'], [w v, t I]] >> 0, v the
// multipliers under the norm and w = rho*sbeta/s (the scaled objective is tau + sum t); S is a function of (t, phi) alone, so it
// is rebuilt instead of iterated and never carries a residual; its primal block X is (m+1) x (m+1).
// The stage-local vector y_loc = (phi_k, t_k) (at most 64 entries: up to 31 + 31 rows and two epigraph variables) touches only the cone blocks of stage k, so in the HKM Schur
// system its rows reach P_k, P_{k+1}, tau and alpha only:
//     a_k,i = -svec(W_i[:nx,:nx]) (P_k),   b_k,i = svec(V_k W_i V_k') (P_{k+1}),   W_i = sum_r sym(X_r g_i g_i' S_r^-1) = sum_r sym(w_ri u_ri'),
//     c_tau, c_alpha (border), T_loc,loc (own block; zero coupling for the epigraph entries).
// y_loc of stage k rides in block k+1 of the block-cyclic-tridiagonal system, behind P_{k+1} (block size
// d + nz instead of d): rows d.. of D_{k+1} hold b_k and T_loc,loc, the same rows of the coupling block hold a_k (k_aug_fill), the
// tails of the right-hand side / border vectors hold r_loc, c_tau, c_alpha (k_aug_gather), and dy_loc is the tail of the block
// solution.  y_loc is then pivoted AFTER the two P blocks it couples to -- the order of the oracle's border solve -- and k_schur and
// the factorisation / substitution kernels of tmpc_cr.h run unchanged on the larger blocks.
// (The first design eliminated y_loc stage by stage BEFORE the factorisation, D_k -= a_k K_k a_k' + ... with K_k = T_loc,loc^-1.  It lost
// digits whenever a multiplier is active (z/phi << T_phiphi) and its direction g g' is nearly reachable by calH(dP): D - aKa' then
// cancels to ~ (z/phi)/T_phiphi of D; replayed in numpy the direction error reached 1e-6 .. 0.4 on such problems while the block
// form stays at rounding level.  Removed in round 2.)
// Every kernel here is one single-wave workgroup per stage and only runs when dm.nr > 0.
#pragma once
#include "tmpc_common.h"
#include "tmpc_small.h"
#include "tmpc_stage.h"

namespace tmpc {

struct PhiStage {
  int nrow;          // rows of [G_k; C_k] at this stage
  int na;            // norm terms (arrow blocks) at this stage
  int nz;            // nrow + na
  int a0[2], am[2];  // arrow block e covers rows a0[e] .. a0[e] + am[e] - 1; its epigraph variable is entry nrow + e
};
__device__ __forceinline__ PhiStage phi_stage(const WS& w, const Dims& dm, size_t sid) {
  PhiStage s;
  const int nc = stage_rows(w, dm, sid) - dm.ng;          // clamped to the room of the handle
  s.nrow = dm.ng + nc; s.na = 0;
  s.a0[0] = s.a0[1] = 0; s.am[0] = s.am[1] = 0;
  if (dm.constr) {
    if (dm.ng > 0) { s.a0[s.na] = 0; s.am[s.na] = dm.ng; ++s.na; }
    if (nc > 0) { s.a0[s.na] = dm.ng; s.am[s.na] = nc; ++s.na; }
  }
  s.nz = s.nrow + s.na;
  return s;
}

__device__ __forceinline__ int pv_len(const Dims& dm) { return 2 * dm.n + 2 * dm.nx; }
__device__ __forceinline__ double* pv_at(double* pvec, const Dims& dm, size_t sid, int r, int i) {
  return pvec + ((sid * 2 + r) * dm.nr + i) * (size_t)pv_len(dm);
}
// psm offsets (doubles, stride g = dm.nz): K, c_tau, c_alpha, K c_tau, K c_alpha, r_loc, K r_loc
__device__ __forceinline__ double* psm_at(double* psm, const Dims& dm, size_t sid) { return psm + sid * (size_t)(dm.nz * dm.nz + 6 * dm.nz); }
#define PSM_K(q) (q)
#define PSM_CT(q, g) ((q) + (g) * (g))
#define PSM_CA(q, g) ((q) + (g) * (g) + (g))
#define PSM_KCT(q, g) ((q) + (g) * (g) + 2 * (g))
#define PSM_KCA(q, g) ((q) + (g) * (g) + 3 * (g))
#define PSM_RPHI(q, g) ((q) + (g) * (g) + 4 * (g))
#define PSM_KR(q, g) ((q) + (g) * (g) + 5 * (g))

__device__ __forceinline__ double phi_wr(const WS& w, const double* pr) { return w.rho * pr[P_SBETA] / pr[P_S]; }

// LDS slot (stride LD) <- arrow(t, v, wr) of size (m+1)
__device__ __forceinline__ void arrow_s(double* S, double t, const double* v, double wr, int m, int lane) {
  const int ne = m + 1;
  for (int e = lane; e < ne * ne; e += 64) {
    const int i = e / ne, j = e - i * ne;
    double x = 0.0;
    if (i == j) x = t;
    else if (i == 0) x = wr * v[j - 1];
    else if (j == 0) x = wr * v[i - 1];
    S[i * LD + j] = x;
  }
  wsync();
}
// LDS slot (stride LD) <-> global (stride AEL)
__device__ __forceinline__ void a_g2s(double* S, const double* g, int ne, int lane) {
  for (int e = lane; e < ne * ne; e += 64) { const int i = e / ne, j = e - i * ne; S[i * LD + j] = g[i * AEL + j]; }
  wsync();
}
__device__ __forceinline__ void a_s2g(double* g, const double* S, int ne, int lane) {
  for (int e = lane; e < ne * ne; e += 64) { const int i = e / ne, j = e - i * ne; g[i * AEL + j] = S[i * LD + j]; }
  wsync();
}
__device__ __forceinline__ void a_s2g_sym(double* g, const double* S, int ne, int lane) {
  for (int e = lane; e < ne * ne; e += 64) { const int i = e / ne, j = e - i * ne; g[i * AEL + j] = 0.5 * (S[i * LD + j] + S[j * LD + i]); }
  wsync();
}
__device__ __forceinline__ double a_dot(const double* A, const double* B, int ne, int lane) {   // <A, B> of two LDS slots
  double v = 0.0;
  for (int e = lane; e < ne * ne; e += 64) { const int i = e / ne, j = e - i * ne; v = fma(A[i * LD + j], B[i * LD + j], v); }
  return wave_sum(v);
}

// after k_init_state
__global__ void __launch_bounds__(64) k_phi_init(WS w, Dims dm) {
  const size_t sid = blockIdx.x;
  const int lane = threadIdx.x;
  const int b = (int)(sid / dm.p);
  const PhiStage ps = phi_stage(w, dm, sid);
  const double x0 = 1.0 / (double)(dm.p * dm.n);
  if (lane < dm.nr) {
    // phi_i = min(1, 1/|g_i|^2): the term g_i' phi_i g_i is O(1) whatever the scaling of the Jacobian rows; z_i = x0/phi_i
    double ph = 1.0;
    if (lane < ps.nrow) {
      const double* g = w.G + (sid * dm.nr + lane) * dm.n;
      double g2 = 0.0;
      for (int c = 0; c < dm.n; ++c) g2 = fma(g[c], g[c], g2);
      ph = fmin(1.0, 1.0 / fmax(g2, 1e-300));
    }
    w.phi[sid * dm.nr + lane] = ph;
    w.zph[sid * dm.nr + lane] = (lane < ps.nrow) ? x0 / ph : 0.0;
    w.corrp[sid * dm.nr + lane] = 0.0;
    w.dphi[sid * dm.nr + lane] = 0.0; w.dzph[sid * dm.nr + lane] = 0.0;
  }
  __syncthreads();
  if (dm.constr) {
    // Norm terms start ON their central path: z_i = w/sqrt(m) (the gradient of w||phi|| at equal phi_i, so the stationarity residual
    // of phi_i vanishes), phi_i = x0/z_i (<= 1), t = w||phi|| + delta with x0 tr(S^-1) = 1 (the cost of t), X = x0 S^-1.  With phi = 1,
    // X = x0 I, S = O(1) each of the 2p terms carries a residual ~ 1 and the first Newton steps blow mu up by 1e3 (diverges at p = 64).
    const double wr = phi_wr(w, w.prob + (size_t)b * PS);
    for (int e = 0; e < ps.na; ++e) {
      const int m = ps.am[e], c0 = ps.a0[e];
      const double ph = fmin(1.0, x0 * sqrt((double)m) / wr);
      if (lane < m) { w.phi[sid * dm.nr + c0 + lane] = ph; w.zph[sid * dm.nr + c0 + lane] = x0 / ph; }
      const double a = wr * ph * sqrt((double)m);
      double lo = x0, hi = (m + 1.0) * x0;                 // eigenvalues of S: delta, delta + 2a, delta + a (m - 1 times)
      for (int it = 0; it < 60; ++it) {
        const double dl = 0.5 * (lo + hi);
        const double f = x0 * (1.0 / dl + 1.0 / (dl + 2.0 * a) + (m - 1.0) / (dl + a)) - 1.0;
        if (f > 0.0) lo = dl; else hi = dl;
      }
      const double t = a + 0.5 * (lo + hi);
      const double u = wr * ph, gam = t * t - m * u * u;
      if (lane == 0) { w.at[sid * 2 + e] = t; w.adt[sid * 2 + e] = 0.0; }
      for (int q = lane; q < AE; q += 64) {
        const int i = q / AEL, j = q % AEL;
        double v = 0.0;                                    // closed-form inverse of the arrow matrix
        if (i <= m && j <= m) {
          if (i == 0 && j == 0) v = t / gam;
          else if (i == 0 || j == 0) v = -u / gam;
          else v = ((i == j) ? 1.0 / t : 0.0) + u * u / (t * gam);
        }
        w.aX[(sid * 2 + e) * AE + q] = x0 * v;
        w.acor[(sid * 2 + e) * AE + q] = 0.0;
      }
    }
  }
}

// LDS of k_phi_pre (doubles)
constexpr int phi_pre_lds(bool bign, bool bigr) {
  return 4 * MS + (bigr ? NZM * (2 * NZM + 1) + 4 * NZM
                        : (bign ? 0 : NRS * NMAX) + 4 * NRS * (bign ? NBM : NMAX) + 4 * NRS * NRS + NZS * (2 * NZS + 1) + 4 * NZS);
}
// after k_stage_pre (needs S_r^-1), before k_ctrl_a: vectors w, u, V w, V u; K; c_tau, c_alpha; border / mu / pinf partials
// aug: the multipliers stay in the block system (k_aug_fill): T_loc,loc itself is stored instead of its inverse and the border
// scalars are left alone.
// BIGN (32 < n <= 64, with the generic per-stage kernels of tmpc_big.h): the n x n matrices and the Jacobian rows are read where they
// lie in global memory instead of LDS slots (every entry is used once per row of [G; C]); the vectors are 64 long; the arrow blocks
// keep the four LDS slots.  Same arithmetic in the same order: at n <= 32 both forms return the same bits.
// BIGR (handles with room for more than NRS = 32 rows per stage, up to 31 + 31): the per-row vectors stay in their place of pvec and the
// Gram products in WS::prs (global memory) -- only T_loc,loc (64 x 129) and the arrow slots are in LDS.
template <bool BIGN, bool BIGR>
__global__ void __launch_bounds__(64) k_phi_pre(WS w, Dims dm, int aug) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int sid = stage_id(w, dm), lane = threadIdx.x;
  const int b = sid / dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] == PH_DONE) return;
  const double* pr = w.prob + (size_t)b * PS;
  const PhiStage ps = phi_stage(w, dm, sid);
  const int n = dm.n, nx = dm.nx, nn = n * n, ng = ps.nrow, nz = ps.nz, nzs = dm.nz;
  double* sX = sm; double* sSi = sm + MS; double* sHb = sm + 2 * MS; double* sV = sm + 3 * MS;
  constexpr int NV = BIGN ? NBM : NMAX;      // stride of the per-row vectors
  constexpr int NZ_ = BIGR ? NZM : NZS;
  double* gl = sm + 4 * MS;                 // [nrow][NMAX]   (BIGN, BIGR: the rows are read from global memory)
  double* wl = gl + ((BIGN || BIGR) ? 0 : NRS * NMAX);  // [2][nrow][NV]
  double* ul = wl + (BIGR ? 0 : 2 * NRS * NV);          // [2][nrow][NV]
  double* GXGl = ul + (BIGR ? 0 : 2 * NRS * NV);        // [2][NRS][NRS]
  double* GSGl = GXGl + (BIGR ? 0 : 2 * NRS * NRS);     // [2][NRS][NRS]
  double* Tm = GSGl + (BIGR ? 0 : 2 * NRS * NRS);       // [NZ_][2*NZ_+1]  T_loc,loc | I  ->  I | K
  double* ctl = Tm + NZ_ * (2 * NZ_ + 1); double* cal = ctl + NZ_; double* rres = cal + NZ_; double* kv = rres + NZ_;
  constexpr int TL = 2 * NZ_ + 1;
  const int gs = BIGR ? dm.nr : NRS;        // stride of the Gram products
  double* GXG = BIGR ? w.prs + (size_t)sid * 4 * dm.nr * dm.nr : GXGl;
  double* GSG = BIGR ? GXG + 2 * dm.nr * dm.nr : GSGl;
  auto WV = [&](int r, int i) -> double* { return BIGR ? pv_at(w.pvec, dm, sid, r, i) : wl + (r * NRS + i) * NV; };        // w_ri = X_r g_i
  auto UV = [&](int r, int i) -> double* { return BIGR ? pv_at(w.pvec, dm, sid, r, i) + n : ul + (r * NRS + i) * NV; };    // u_ri = S_r^-1 g_i
  const double* Gg = w.G + (size_t)sid * dm.nr * n;
  // matrices: LDS slots (stride LD) or, BIGN, global memory (stride n)
  const int ldm = BIGN ? n : LD, ldg = (BIGN || BIGR) ? n : NMAX;
  const double* mHb = BIGN ? w.Hb + (size_t)sid * nn : sHb; const double* mV = BIGN ? w.V + (size_t)sid * nx * n : sV;
  const double* mG = (BIGN || BIGR) ? Gg : gl;
  if (!BIGN && !BIGR) { for (int e = lane; e < ng * n; e += 64) gl[(e / n) * NMAX + (e % n)] = Gg[e]; }
  if (!BIGN) {
    g2s(sHb, w.Hb + (size_t)sid * nn, n, n, n, lane);
    g2s(sV, w.V + (size_t)sid * nx * n, nx, n, n, lane);
  }
  if (lane < NZ_) { ctl[lane] = 0.0; cal[lane] = 0.0; rres[lane] = 0.0; }
  for (int r = 0; r < 2; ++r) {
    const double* mX = BIGN ? (r ? w.X2 : w.X1) + (size_t)sid * nn : sX; const double* mSi = BIGN ? (r ? w.S2i : w.S1i) + (size_t)sid * nn : sSi;
    if (!BIGN) {
      g2s(sX, (r ? w.X2 : w.X1) + (size_t)sid * nn, n, n, n, lane);
      g2s(sSi, (r ? w.S2i : w.S1i) + (size_t)sid * nn, n, n, n, lane);
    }
    for (int i = 0; i < ng; ++i) {
      if (lane < n) {
        double a0 = 0.0, a1 = 0.0;
        for (int c = 0; c < n; ++c) { const double g = mG[i * ldg + c]; a0 = fma(mX[lane * ldm + c], g, a0); a1 = fma(mSi[lane * ldm + c], g, a1); }
        WV(r, i)[lane] = a0; UV(r, i)[lane] = a1;
      }
    }
    if (BIGR) __threadfence_block();
    wsync();
    for (int i = 0; i < ng; ++i) {
      double* pv = pv_at(w.pvec, dm, sid, r, i);
      const double* wi = WV(r, i); const double* ui = UV(r, i);
      if (!BIGR && lane < n) { pv[lane] = wi[lane]; pv[n + lane] = ui[lane]; }
      if (lane < nx) {
        double a0 = 0.0, a1 = 0.0;
        for (int c = 0; c < n; ++c) { a0 = fma(mV[lane * ldm + c], wi[c], a0); a1 = fma(mV[lane * ldm + c], ui[c], a1); }
        pv[2 * n + lane] = a0; pv[2 * n + nx + lane] = a1;
      }
      // c_alpha_i += w' Hb u ;  c_tau_i = -(w_2 . u_2)
      double hu = 0.0;
      if (lane < n) { for (int c = 0; c < n; ++c) hu = fma(mHb[lane * ldm + c], ui[c], hu); hu *= wi[lane]; }
      hu = wave_sum(hu);
      double wu = (lane < n) ? wi[lane] * ui[lane] : 0.0;
      wu = wave_sum(wu);
      if (lane == 0) { cal[i] += hu; if (r == 1) ctl[i] = -wu; }
      if (BIGR) {        // one lane per column j (up to 62 of them): a serial dot product each instead of 2 ng wave reductions per row
        if (lane < ng) {
          const double* wj = WV(r, lane); const double* uj = UV(r, lane);
          double x = 0.0, y = 0.0;
          for (int c = 0; c < n; ++c) { const double g = mG[i * ldg + c]; x = fma(g, wj[c], x); y = fma(g, uj[c], y); }
          GXG[(r * gs + i) * gs + lane] = x; GSG[(r * gs + i) * gs + lane] = y;
        }
      } else
      for (int j = 0; j < ng; ++j) {
        double x = (lane < n) ? mG[i * ldg + lane] * WV(r, j)[lane] : 0.0;
        double y = (lane < n) ? mG[i * ldg + lane] * UV(r, j)[lane] : 0.0;
        x = wave_sum(x); y = wave_sum(y);
        if (lane == 0) { GXG[(r * gs + i) * gs + j] = x; GSG[(r * gs + i) * gs + j] = y; }
      }
    }
    if (BIGR) __threadfence_block();
    wsync();
  }
  const double* phi = w.phi + (size_t)sid * dm.nr; const double* z = w.zph + (size_t)sid * dm.nr;
  // T_loc,loc | I
  for (int e = lane; e < nz * 2 * nz; e += 64) {
    const int i = e / (2 * nz), j = e - i * 2 * nz;
    double t = 0.0;
    if (j >= nz) t = (j - nz == i) ? 1.0 : 0.0;
    else if (i < ng && j < ng) {
      for (int r = 0; r < 2; ++r)
        t += 0.5 * (GXG[(r * gs + i) * gs + j] * GSG[(r * gs + j) * gs + i] + GXG[(r * gs + j) * gs + i] * GSG[(r * gs + i) * gs + j]);
      if (i == j) t += z[i] / phi[i];
    }
    Tm[i * TL + j] = t;
  }
  double xs = 0.0;
  if (lane < ng) {
    xs = phi[lane] * z[lane];
    rres[lane] = -(GXG[(0 * gs + lane) * gs + lane] - GXG[(1 * gs + lane) * gs + lane]) - z[lane];     // stationarity residual of phi_i
  }
  xs = wave_sum(xs);
  wsync();
  double ncone = (double)ng;
  // norm terms: arrow blocks
  if (ps.na > 0) {
    const double wr = phi_wr(w, pr);
    double* aS = sX; double* aXs = sSi; double* aW = sHb; double* aLi = sV;
    for (int e = 0; e < ps.na; ++e) {
      const int m = ps.am[e], ne = m + 1, c0 = ps.a0[e], te = ng + e;
      const size_t ao = ((size_t)sid * 2 + e) * AE;
      arrow_s(aS, w.at[(size_t)sid * 2 + e], phi + c0, wr, m, lane);
      a_g2s(aXs, w.aX + ao, ne, lane);
      xs += a_dot(aXs, aS, ne, lane);
      // L_X^-1 (primal step length)
      for (int q = lane; q < ne * ne; q += 64) { const int i = q / ne, j = q - i * ne; aW[i * LD + j] = aXs[i * LD + j]; }
      wsync();
      chol_lower(aW, ne, lane);
      tri_inv_lower(aLi, aW, ne, lane);
      a_s2g(w.aLXi + ao, aLi, ne, lane);
      // L_S^-1, S^-1 = L^-T L^-1
      chol_lower(aS, ne, lane);
      tri_inv_lower(aLi, aS, ne, lane);
      a_s2g(w.aLi + ao, aLi, ne, lane);
      mm(aW, aLi, 1, LD, aLi, LD, 1, ne, ne, ne, 0, lane);          // aW = S^-1
      a_s2g(w.aSi + ao, aW, ne, lane);
      // Schur entries of the block:  <E_a, sym(X E_q S^-1)>, <E_q, sym(X S^-1)>, tr(X S^-1)
      const double trp = a_dot(aXs, aW, ne, lane);
      if (lane < m) {
        const int q = lane;
        double u0 = 0.0, u1 = 0.0;
        for (int r = 0; r < ne; ++r) { u0 = fma(aXs[r], aW[r * LD + q + 1], u0); u1 = fma(aXs[(q + 1) * LD + r], aW[r * LD], u1); }
        const double v = wr * (u0 + u1);
        Tm[(c0 + q) * TL + te] = v; Tm[te * TL + c0 + q] = v;
        rres[c0 + q] -= 2.0 * wr * aXs[q + 1];
      }
      if (lane == 0) { Tm[te * TL + te] = trp; }
      for (int qq = lane; qq < m * m; qq += 64) {
        const int a = qq / m, q = qq - a * m;
        const double v = wr * wr * (aXs[0] * aW[(q + 1) * LD + a + 1] + aXs[q + 1] * aW[a + 1] + aXs[(a + 1) * LD] * aW[(q + 1) * LD]
                                    + aXs[(a + 1) * LD + q + 1] * aW[0]);
        Tm[(c0 + a) * TL + c0 + q] += v;
      }
      double trx = (lane < ne) ? aXs[lane * LD + lane] : 0.0;
      trx = wave_sum(trx);
      if (lane == 0) rres[te] = 1.0 - trx;
      ncone += (double)ne;
      wsync();
    }
  }
  // Gauss-Jordan on [T | I] (T symmetric positive definite): one lane per column, the pivot column travels through kv
  if (aug) {
    for (int e = lane; e < nz * nz; e += 64) { const int i = e / nz, j = e - i * nz; Tm[i * TL + nz + j] = Tm[i * TL + j]; }
    wsync();
  }
  for (int c = 0; c < (aug ? 0 : nz); ++c) {
    if (lane < nz) kv[lane] = Tm[lane * TL + c];
    wsync();
    if (lane < 2 * nz) {
      const double pc = Tm[c * TL + lane] / kv[c];
      for (int i = 0; i < nz; ++i)
        if (i != c) Tm[i * TL + lane] = fma(-kv[i], pc, Tm[i * TL + lane]);
      Tm[c * TL + lane] = pc;
    }
    wsync();
  }
  double* q = psm_at(w.psm, dm, sid);
  for (int e = lane; e < nz * nz; e += 64) { const int i = e / nz, j = e - i * nz; PSM_K(q)[i * nzs + j] = Tm[i * TL + nz + j]; }
  double kct = 0.0, kca = 0.0, rr = 0.0;
  if (lane < nz) {
    for (int j = 0; j < nz; ++j) { kct = fma(Tm[lane * TL + nz + j], ctl[j], kct); kca = fma(Tm[lane * TL + nz + j], cal[j], kca); }
    PSM_CT(q, nzs)[lane] = ctl[lane]; PSM_CA(q, nzs)[lane] = cal[lane]; PSM_KCT(q, nzs)[lane] = kct; PSM_KCA(q, nzs)[lane] = kca;
    rr = rres[lane];
  }
  const double ctkct = wave_sum((lane < nz) ? ctl[lane] * kct : 0.0);
  const double ctkca = wave_sum((lane < nz) ? ctl[lane] * kca : 0.0);
  const double cakca = wave_sum((lane < nz) ? cal[lane] * kca : 0.0);
  const double rp2 = wave_sum(rr * rr);
  if (lane == 0) {
    double* pq = w.part + (size_t)sid * NPART;
    pq[Q_XS] += xs; pq[Q_RPHI2] = rp2; pq[Q_NCONE] = ncone;
    if (!aug) {
      pq[Q_TRPSI] -= ctkct;       // b_tt
      pq[Q_TRPHI2] += ctkca;      // b_ta = -trphi2
      pq[Q_HBPHI] -= cakca;       // b_aa (+ x0/s0)
    }
  }
}

// svec coordinates of a_i (P_k rows) / b_i (P_{k+1} rows) of stage `sid` at entry (a, c), a <= c
__device__ __forceinline__ double phi_avec(const double* pvec, const Dims& dm, size_t sid, int i, int a, int c) {
  const int n = dm.n;
  double v = 0.0;
  for (int r = 0; r < 2; ++r) {
    const double* pv = pvec + ((sid * 2 + r) * dm.nr + i) * (size_t)(2 * dm.n + 2 * dm.nx);
    v += 0.5 * (pv[a] * pv[n + c] + pv[c] * pv[n + a]);
  }
  return -((a == c) ? v : 2.0 * v);
}
__device__ __forceinline__ double phi_bvec(const double* pvec, const Dims& dm, size_t sid, int i, int a, int c) {
  const int n = dm.n, nx = dm.nx;
  double v = 0.0;
  for (int r = 0; r < 2; ++r) {
    const double* pv = pvec + ((sid * 2 + r) * dm.nr + i) * (size_t)(2 * dm.n + 2 * dm.nx);
    v += 0.5 * (pv[2 * n + a] * pv[2 * n + nx + c] + pv[2 * n + c] * pv[2 * n + nx + a]);
  }
  return (a == c) ? v : 2.0 * v;
}

// after k_stage_rhs, before k_gather / the substitutions: r_loc, K r_loc, and the eliminated part of the border right-hand sides
template <bool BIGN>
__global__ void __launch_bounds__(64) k_phi_rhs(WS w, Dims dm, int pass, int aug) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int sid = stage_id(w, dm), lane = threadIdx.x;
  const int b = sid / dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return;
  const double* pr = w.prob + (size_t)b * PS;
  const double sig = (pass == 1) ? 0.0 : pr[P_SIGMU];
  const bool use_corr = (pass == 2 && phase == PH_MAIN);
  const PhiStage ps = phi_stage(w, dm, sid);
  const int n = dm.n, nn = n * n, ng = ps.nrow, nz = ps.nz, nzs = dm.nz;
  double* sG = sm;                         // T1 - T2
  double* rl = sm + MS;                    // [NZM]
  const double* T1 = w.T1 + (size_t)sid * nn; const double* T2 = w.T2 + (size_t)sid * nn;
  if (!BIGN) {
    for (int e = lane; e < nn; e += 64) { const int i = e / n, j = e - i * n; sG[i * LD + j] = T1[e] - T2[e]; }
    wsync();
  }
  const double* Gg = w.G + (size_t)sid * dm.nr * n;
  double* q = psm_at(w.psm, dm, sid);
  const double* phi = w.phi + (size_t)sid * dm.nr; const double* cp = w.corrp + (size_t)sid * dm.nr;
  for (int i = 0; i < ng; ++i) {
    double t = 0.0;
    if (lane < n) {
      if (BIGN) { for (int c = 0; c < n; ++c) t = fma(T1[lane * n + c] - T2[lane * n + c], Gg[i * n + c], t); }
      else { for (int c = 0; c < n; ++c) t = fma(sG[lane * LD + c], Gg[i * n + c], t); }
      t *= Gg[i * n + lane];
    }
    t = wave_sum(t);
    if (lane == 0) rl[i] = t + sig / phi[i] - (use_corr ? cp[i] : 0.0);
  }
  wsync();
  if (ps.na > 0) {
    const double wr = phi_wr(w, pr);
    for (int e = 0; e < ps.na; ++e) {
      const int m = ps.am[e], ne = m + 1, c0 = ps.a0[e];
      const double* Si = w.aSi + ((size_t)sid * 2 + e) * AE; const double* co = w.acor + ((size_t)sid * 2 + e) * AE;
      // T_e = sig S^-1 - corr:  r_i += 2 w T_e[0][i+1],  r_t = tr T_e - 1
      if (lane < m) rl[c0 + lane] += 2.0 * wr * (sig * Si[lane + 1] - (use_corr ? co[lane + 1] : 0.0));
      double tr = (lane < ne) ? sig * Si[lane * AEL + lane] - (use_corr ? co[lane * AEL + lane] : 0.0) : 0.0;
      tr = wave_sum(tr);
      if (lane == 0) rl[ng + e] = tr - 1.0;
    }
    wsync();
  }
  double kr = 0.0;
  if (lane < nz) {
    for (int j = 0; j < nz; ++j) kr = fma(PSM_K(q)[lane * nzs + j], rl[j], kr);
    PSM_RPHI(q, nzs)[lane] = rl[lane]; PSM_KR(q, nzs)[lane] = kr;
  }
  const double ctkr = wave_sum((lane < nz) ? PSM_CT(q, nzs)[lane] * kr : 0.0);
  const double cakr = wave_sum((lane < nz) ? PSM_CA(q, nzs)[lane] * kr : 0.0);
  if (lane == 0 && !aug) {
    double* pq = w.part + (size_t)sid * NPART;
    pq[Q_TRT2] -= ctkr;     // rhs_tau   = sum trT2 - 1
    pq[Q_HBG] -= cakr;      // rhs_alpha = sum <Hb, T1 - T2> + t0
  }
}

// after the substitutions, before k_stage_dir: dy_loc = K (r_loc - T_loc,y dy), dz; arrow blocks: dS, dX, step-length eigenvalues;
// Mehrotra second-order terms in pass 1
// BIGN: dalpha*Hb + calH(dP) was left in scratch slot 4 of the stage by kb_phi_dm (tmpc_big.h)
template <bool BIGN>
__global__ void __launch_bounds__(64) k_phi_dir(WS w, Dims dm, int pass, int aug) {
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
  const PhiStage ps = phi_stage(w, dm, sid);
  const int n = dm.n, nx = dm.nx, nn = n * n, nxx = nx * nx, ng = ps.nrow, nz = ps.nz, nzs = dm.nz;
  double* sV = sm; double* sM = sm + MS; double* t0 = sm + 2 * MS; double* t1 = sm + 3 * MS; double* sHb = sm + 4 * MS;
  double* tl = sm + 5 * MS;                // [NZM] r_loc - T_loc,y dy
  double* dl = tl + NZM;                   // [NZM] dy_loc
  const int kn = (k + 1 == dm.p) ? 0 : k + 1;
  const int ldm = BIGN ? n : LD;
  const double* mM = BIGN ? w.bscr + ((size_t)sid * BIG_SCR + 4) * nn : sM;
  if (!BIGN) {
    g2s(sV, w.V + (size_t)sid * nx * n, nx, n, n, lane);
    g2s(sHb, w.Hb + (size_t)sid * nn, n, n, n, lane);
    build_M(sM, sV, t0, t1, sHb, w.dP + (size_t)sid * nxx, w.dP + (size_t)(b * dm.p + kn) * nxx, dalpha, n, nx, lane);   // dalpha*Hb + calH(dP)
  }
  double* q = psm_at(w.psm, dm, sid);
  for (int i = 0; i < ng; ++i) {
    double t = 0.0;
    for (int r = 0; r < 2; ++r) {
      const double* pv = pv_at(w.pvec, dm, sid, r, i);       // w = pv[0..n), u = pv[n..2n)
      double x = 0.0;
      if (lane < n) { for (int c = 0; c < n; ++c) x = fma(mM[lane * ldm + c], pv[n + c], x); x *= pv[lane]; }
      t += wave_sum(x);
    }
    if (lane == 0) tl[i] = PSM_RPHI(q, nzs)[i] - (t + dtau * PSM_CT(q, nzs)[i]);
  }
  if (lane < ps.na) tl[ng + lane] = PSM_RPHI(q, nzs)[ng + lane];          // the epigraph variables do not couple to (P, tau, alpha)
  wsync();
  const double* phi = w.phi + (size_t)sid * dm.nr; const double* z = w.zph + (size_t)sid * dm.nr;
  double* cp = w.corrp + (size_t)sid * dm.nr;
  double* dph = w.dphi + (size_t)sid * dm.nr; double* dzp = w.dzph + (size_t)sid * dm.nr;
  if (lane < nz) {
    double v = 0.0;
    if (aug) {        // the block solve already produced dy_loc: tail of the solution vector of block k+1
      const size_t vi = ((size_t)b * dm.p + kn) * dm.dp + dm.d + lane;
      v = w.Z[vi] - w.TU[vi * 2] * dtau - w.TU[vi * 2 + 1] * dalpha;
    } else {
      for (int j = 0; j < nz; ++j) v = fma(PSM_K(q)[lane * nzs + j], tl[j], v);
    }
    dl[lane] = v;
    if (lane < ng) {
      dph[lane] = v;
      const double dz = sig / phi[lane] - z[lane] - z[lane] * v / phi[lane] - (use_corr ? cp[lane] : 0.0);
      dzp[lane] = dz;
      if (pass == 1) cp[lane] = dz * v / phi[lane];
    }
  }
  wsync();
  double dxs = 0.0, xds = 0.0, dxds = 0.0, emd = 0.0, emp = 0.0;
  if (ps.na > 0) {
    const double wr = phi_wr(w, pr);
    double* aS = sV; double* aX = sM; double* aSi = sHb; double* aDS = sm + 5 * MS + 2 * NZM;   // aDS, aDX: two more slots
    double* aDX = aDS + MS;
    double* vv = aDX + MS;                 // tridiag scratch (>= 96 doubles)
    for (int e = 0; e < ps.na; ++e) {
      const int m = ps.am[e], ne = m + 1, c0 = ps.a0[e];
      const size_t ao = ((size_t)sid * 2 + e) * AE;
      const double dt = dl[ng + e];
      if (lane == 0) w.adt[(size_t)sid * 2 + e] = dt;
      arrow_s(aS, w.at[(size_t)sid * 2 + e], phi + c0, wr, m, lane);
      arrow_s(aDS, dt, dl + c0, wr, m, lane);
      a_g2s(aX, w.aX + ao, ne, lane);
      a_g2s(aSi, w.aSi + ao, ne, lane);
      // dX = sig S^-1 - X - sym(X dS S^-1) - corr
      mm(t0, aX, LD, 1, aDS, LD, 1, ne, ne, ne, 0, lane);
      mm(t1, t0, LD, 1, aSi, LD, 1, ne, ne, ne, 0, lane);
      for (int qq = lane; qq < ne * ne; qq += 64) {
        const int i = qq / ne, j = qq - i * ne;
        aDX[i * LD + j] = sig * aSi[i * LD + j] - aX[i * LD + j] - 0.5 * (t1[i * LD + j] + t1[j * LD + i]) - (use_corr ? w.acor[ao + i * AEL + j] : 0.0);
      }
      wsync();
      a_s2g(w.adX + ao, aDX, ne, lane);
      dxs += a_dot(aDX, aS, ne, lane); xds += a_dot(aX, aDS, ne, lane); dxds += a_dot(aDX, aDS, ne, lane);
      if (pass == 1) {      // sym(dX dS S^-1)
        mm(t0, aDX, LD, 1, aDS, LD, 1, ne, ne, ne, 0, lane);
        mm(t1, t0, LD, 1, aSi, LD, 1, ne, ne, ne, 0, lane);
        a_s2g_sym(w.acor + ao, t1, ne, lane);
      }
      // step-length matrices L^-1 dS L^-T, LX^-1 dX LX^-T
      a_g2s(aX, w.aLi + ao, ne, lane);
      mm(t0, aX, LD, 1, aDS, LD, 1, ne, ne, ne, 0, lane);
      mm(t1, t0, LD, 1, aX, 1, LD, ne, ne, ne, 0, lane);
      s_sym(t1, ne, lane);
      emd = fmin(emd, tridiag_min_eig(t1, ne, vv, lane));
      wsync();
      a_g2s(aX, w.aLXi + ao, ne, lane);
      mm(t0, aX, LD, 1, aDX, LD, 1, ne, ne, ne, 0, lane);
      mm(t1, t0, LD, 1, aX, 1, LD, ne, ne, ne, 0, lane);
      s_sym(t1, ne, lane);
      emp = fmin(emp, tridiag_min_eig(t1, ne, vv, lane));
      wsync();
    }
  }
  if (lane == 0) {
    double* as = w.asum + (size_t)sid * 5;
    as[0] = dxs; as[1] = xds; as[2] = dxds; as[3] = emd; as[4] = emp;
  }
}
constexpr int PHI_DIR_LDS = 5 * MS + 2 * NZM + 2 * MS + 128;

// after k_stage_dir and k_eigmin, before k_ctrl_b / k_ctrl_c: the linear cone and the arrow blocks join the step-length minima and
// the mu_aff sums
__global__ void __launch_bounds__(64) k_phi_steps(WS w, Dims dm, int pass) {
  const int sid = blockIdx.x * 64 + threadIdx.x;
  if (sid >= dm.B * dm.p) return;
  const int b = sid / dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE || phase == PH_POLISH || (pass == 1 && phase != PH_MAIN)) return;
  const int ng = stage_rows(w, dm, sid);
  const double* phi = w.phi + (size_t)sid * dm.nr; const double* z = w.zph + (size_t)sid * dm.nr;
  const double* dph = w.dphi + (size_t)sid * dm.nr; const double* dzp = w.dzph + (size_t)sid * dm.nr;
  const double* as = w.asum + (size_t)sid * 5;
  double ls = as[3], lx = as[4], dxs = as[0], xds = as[1], dxds = as[2];      // zero without arrow blocks
  for (int i = 0; i < ng; ++i) {
    ls = fmin(ls, dph[i] / phi[i]); lx = fmin(lx, dzp[i] / z[i]);       // "eigenvalues" of the 1 x 1 blocks: step = -1/lambda
    dxs += dzp[i] * phi[i]; xds += z[i] * dph[i]; dxds += dzp[i] * dph[i];
  }
  double* e = w.eigmin + (size_t)sid * 4;
  e[0] = fmin(e[0], ls); e[1] = fmin(e[1], lx);
  double* pq = w.part + (size_t)sid * NPART;
  pq[Q_DXS] += dxs; pq[Q_XDS] += xds; pq[Q_DXDS] += dxds;
}

// ---- augmented-block form: y_loc of stage k rides in block k+1 behind P_{k+1}, i.e. it is pivoted AFTER the two P blocks it
// couples to (the order of the oracle's border solve).  Rows d .. d+nz-1 of block k+1:
//   D_{k+1}[d+i][0..d) = b_k,i   D_{k+1}[d+i][d+j] = T_loc,loc[i][j]   coupling block [block k+1][block k]: row d+i = a_k,i
// after k_schur (which writes identity there), before the factorisation
__global__ void __launch_bounds__(64) k_aug_fill(WS w, Dims dm) {
  const int sid = stage_id(w, dm), lane = threadIdx.x;
  const int b = sid / dm.p, k = sid - b * dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] == PH_DONE) return;
  const PhiStage ps = phi_stage(w, dm, sid);
  const int nx = dm.nx, d = dm.d, dp = dm.dp, nzs = dm.nz, nz = ps.nz, ng = ps.nrow, p = dm.p;
  const int kn = (k + 1 == p) ? 0 : k + 1;
  const size_t bs = (size_t)dp * dp;
  double* Dn = w.D + ((size_t)b * p + kn) * bs;
  double* ddn = w.Ddiag + ((size_t)b * p + kn) * dp;
  const bool corner = (w.cr_orient[k] != 0);        // coupling block stored as T[P_k,P_{k+1}] (rows = block k) instead of T[P_{k+1},P_k]
  double* Cg = w.O + (size_t)sid * bs;
  const double* q = psm_at(w.psm, dm, sid);
  for (int i = 0; i < nz; ++i) {
    double* drow = Dn + (size_t)(d + i) * dp;
    // local block (lower triangle) and its pivot reference
    for (int j = lane; j <= i; j += 64) drow[d + j] = PSM_K(q)[i * nzs + j];
    if (lane == 0) ddn[d + i] = PSM_K(q)[i * nzs + i];
    if (i >= ng) continue;                   // epigraph variables do not couple to P
    int e = 0;
    for (int a = 0; a < nx; ++a) {
      for (int c = a + lane; c < nx; c += 64) {
        const int idx = e + (c - a);
        const double av = phi_avec(w.pvec, dm, sid, i, a, c), bv = phi_bvec(w.pvec, dm, sid, i, a, c);
        if (p == 1) { drow[idx] = av + bv; continue; }          // both couplings land in the one P block
        drow[idx] = bv;
        if (corner) Cg[(size_t)idx * dp + d + i] = av;          // stored [block k][block k+1]
        else Cg[(size_t)(d + i) * dp + idx] = av;               // stored [block k+1][block k]
      }
      e += nx - a;
    }
  }
}

// after k_gather (which zero-pads the tails): right-hand side and border-column entries of y_loc in the vectors of block k+1
__global__ void __launch_bounds__(64) k_aug_gather(WS w, Dims dm, int pass) {
  const int sid = stage_id(w, dm), lane = threadIdx.x;
  const int b = sid / dm.p, k = sid - b * dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return;
  const bool three = (pass == 1) || (phase != PH_MAIN && !ip[I_CHORD]);
  const PhiStage ps = phi_stage(w, dm, sid);
  const int kn = (k + 1 == dm.p) ? 0 : k + 1;
  const int nzs = dm.nz;
  const double* q = psm_at(w.psm, dm, sid);
  if (lane < ps.nz) {
    const size_t vi = ((size_t)b * dm.p + kn) * dm.dp + dm.d + lane;
    const double r = PSM_RPHI(q, nzs)[lane], ct = PSM_CT(q, nzs)[lane], ca = PSM_CA(q, nzs)[lane];
    if (three) {
      double* w3 = w.W3 + vi * 3; w3[0] = r; w3[1] = ct; w3[2] = ca;
      double* u = w.U + vi * 2; u[0] = ct; u[1] = ca;
    } else {
      w.Z[vi] = r;
    }
  }
}

// with k_update
__global__ void __launch_bounds__(64) k_phi_update(WS w, Dims dm) {
  const int sid = blockIdx.x * 64 + threadIdx.x;
  if (sid >= dm.B * dm.p) return;
  const int b = sid / dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] == PH_DONE || ip[I_PHASE] == PH_POLISH) return;      // (PH_POLISH: a member of the tight mode that waits for the polish while others still iterate -- this grid covers every problem)
  const double* pr = w.prob + (size_t)b * PS;
  const double ap = pr[P_AP], ad = pr[P_AD];
  if (ap == 0.0 && ad == 0.0) return;      // discarded direction
  const PhiStage ps = phi_stage(w, dm, sid);
  for (int i = 0; i < ps.nrow; ++i) {
    w.phi[(size_t)sid * dm.nr + i] += ad * w.dphi[(size_t)sid * dm.nr + i];
    w.zph[(size_t)sid * dm.nr + i] += ap * w.dzph[(size_t)sid * dm.nr + i];
  }
  for (int e = 0; e < ps.na; ++e) {
    w.at[(size_t)sid * 2 + e] += ad * w.adt[(size_t)sid * 2 + e];
    const int ne = ps.am[e] + 1;
    double* X = w.aX + ((size_t)sid * 2 + e) * AE; const double* dX = w.adX + ((size_t)sid * 2 + e) * AE;
    for (int i = 0; i < ne; ++i)
      for (int j = 0; j <= i; ++j) {
        const double v = 0.5 * ((X[i * AEL + j] + ap * dX[i * AEL + j]) + (X[j * AEL + i] + ap * dX[j * AEL + i]));
        X[i * AEL + j] = v; X[j * AEL + i] = v;
      }
  }
}

}  // namespace tmpc
