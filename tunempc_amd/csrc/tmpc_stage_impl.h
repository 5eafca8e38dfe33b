// Bodies of the per-iteration per-stage kernels (tmpc_stage.h), included once per instantiation of tmpc_small_impl.h:
//   namespace tmpc        called by the kernel wrappers k_stage_pre / k_stage_rhs / k_stage_dir / k_eigmin / k_update (one workgroup per stage)
//   namespace tmpc::sm8   called wave by wave from the persistent one-workgroup-per-problem kernel (tmpc_persist.h; n <= 8, NT = 64)
// sid = b * p + k; lane = index of the thread among the NT threads that share the stage; sm = their LDS slots.
// transposed store of the leading rows x cols block: g[j*ldg + i] = s[i][j]
template <int NT = 64>
__device__ __forceinline__ void s2g_T(double* g, const double* s, int rows, int cols, int ldg, int lane) {
  const int tot = rows * cols;
  for (int e = lane; e < tot; e += NT) {
    const int j = e / rows, i = e - j * rows;
    g[(size_t)j * ldg + i] = s[i * LD + j];
  }
  wsync();
}

// ------------------------------------------------------------------ M_k (or dM_k) into an LDS slot
// out = coef*Hb_k + V' Pn V - E' Pk E ; uses sV (already loaded), scratch slots t0,t1,sHb
template <int NT = 64>
__device__ __forceinline__ void build_M(double* out, const double* sV, double* t0, double* t1, const double* sHb,
                                        const double* Pk, const double* Pn, double coef, int n, int nx, int lane) {
  g2s<NT>(t0, Pn, nx, nx, nx, lane);
  mm<NT>(t1, sV, 1, LD, t0, LD, 1, n, nx, nx, 0, lane);        // V' Pn   (n x nx)
  mm<NT>(out, t1, LD, 1, sV, LD, 1, n, n, nx, 0, lane);        // (V' Pn) V
  g2s<NT>(t0, Pk, nx, nx, nx, lane);
  for (int e = lane; e < n * n; e += NT) {
    int i, j; ediv(e, n, i, j);
    double v = out[i * LD + j] + coef * sHb[i * LD + j];
    if (i < nx && j < nx) v -= t0[i * LD + j];
    out[i * LD + j] = v;
  }
  wsync();
}

// V G V' (nx x nx) into `out`, scratch t
template <int NT = 64>
__device__ __forceinline__ void adj_V(double* out, double* t, const double* sV, const double* G, int n, int nx, int lane) {
  mm<NT>(t, sV, LD, 1, G, LD, 1, nx, n, n, 0, lane);           // V G    (nx x n)
  mm<NT>(out, t, LD, 1, sV, 1, LD, nx, nx, n, 0, lane);        // (V G) V'
}

// out (n x n LDS slot) += scale * sum_i coef[i] g_i g_i'   (equality-constraint term, G rows and coefficients in global memory)
template <int NT = 64>
__device__ __forceinline__ void add_gtg(double* out, const double* Gg, const double* coef, double scale, int ng, int n, int lane) {
  for (int e = lane; e < n * n; e += NT) {
    int i, j; ediv(e, n, i, j);
    double acc = 0.0;
    for (int r = 0; r < ng; ++r) acc = fma(scale * coef[r] * Gg[r * n + i], Gg[r * n + j], acc);
    out[i * LD + j] += acc;
  }
  wsync();
}

// out (n x n LDS slot) += scale * smat(theta): the regularisation T_k of Step 3 (tmpc_t3.h), theta = its entries (a <= b), row-major upper triangle
template <int NT = 64>
__device__ __forceinline__ void add_smat_t3(double* out, const double* th, double scale, int n, int lane) {
  for (int e = lane; e < n * n; e += NT) {
    int i, j; ediv(e, n, i, j);
    const int a = i < j ? i : j, b = i < j ? j : i;
    out[i * LD + j] += scale * th[a * n - a * (a - 1) / 2 + (b - a)];
  }
  wsync();
}

constexpr int PRE_SLOTS = 6;     // 51 KB of LDS: three blocks per CU (the kernel is latency-bound: 268 / 155 ms per step at one / two blocks per CU).
                                 // Hb and the accumulated Phi live in registers (element e = lane + q NT of the n x n matrix), slots are re-used.
// ------------------------------------------------------------------ stage_pre
template <int NT>
__device__ __forceinline__ void stage_pre_body(const WS& w, const Dims& dm, int sid, int lane, double* sm) {
  const int b = sid / dm.p;
  const int k = sid - b * dm.p;
  const int n = dm.n, nx = dm.nx;
  const int nn = n * n, nxx = nx * nx;
  (void)k; (void)nn; (void)nxx;
  const int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] == PH_DONE) return;
  const double* pr = w.prob + (size_t)b * PS;
  const double alpha = pr[P_ALPHA], tau = pr[P_TAU];
  // slot 0: V.  First phase: Hb (1), M (2), scratch (3, 4).  Per LMI: X (1), S then S^-1 (2), scratch (3, 4), L^-1 then scratch (5).  End: Phi (1).
  double* sV = sm; double* sX = sm + MS; double* sS = sm + 2 * MS; double* t0 = sm + 3 * MS; double* t1 = sm + 4 * MS; double* sLi = sm + 5 * MS;
  double* sHb = sX;       // only until M is built; afterwards Hb is in registers (hbr) or re-read into a scratch slot
  double* sM = sS;        // consumed by the residuals before S is loaded
  double* sSi = sS;       // S^-1 = Li' Li takes the slot of the factor once L^-1 has been formed
  double* t2 = sLi;       // free once S^-1 has been formed
  double* sPhi = sX;      // the accumulated Phi(Hb) goes to LDS after the last use of X
  constexpr int EPT = NMAX * NMAX / NT;                     // elements of an n x n matrix per thread
  double hbr[EPT], phir[EPT];
  const int kn = (k + 1 == dm.p) ? 0 : k + 1;
  TMPC_T0()
  double ldA[EPT];                                          // loads in flight (g2r / r2s): one exposed memory latency per group of matrices
  g2r<NT>(ldA, w.V + (size_t)sid * nx * n, nx, n, n, lane);
  g2r<NT>(hbr, w.Hb + (size_t)sid * nn, n, n, n, lane);
  r2s<NT>(sV, ldA, nx, n, lane);
  r2s<NT>(sHb, hbr, n, n, lane);
  TMPC_T(8)
#pragma unroll
  for (int q = 0; q < EPT; ++q) phir[q] = 0.0;
  build_M<NT>(sM, sV, t0, t1, sHb, w.P + (size_t)sid * nxx, w.P + (size_t)(b * dm.p + kn) * nxx, alpha, n, nx, lane);
  if (dm.nr > 0) add_gtg<NT>(sM, w.G + (size_t)sid * dm.nr * n, w.phi + (size_t)sid * dm.nr, 1.0, stage_rows(w, dm, sid), n, lane);   // + G' diag(phi) G
  if (dm.nT > 0) add_smat_t3<NT>(sM, w.t3th + (size_t)sid * dm.nT, 1.0, n, lane);                                                       // + T_k
  TMPC_T(9)
  double rd2 = 0.0, s2 = 0.0, xs = 0.0, trx2 = 0.0, hby = 0.0, trpsi = 0.0, trphi2 = 0.0;
  int nbad = 0;
  double* kf = w.KF + (size_t)sid * 12 * nxx;
  // residuals of both slack blocks first (frees the M slot)
  for (int r = 0; r < 2; ++r) {
    const double* Sg = (r ? w.S2 : w.S1) + (size_t)sid * nn;
    double* Rdg = (r ? w.Rd2 : w.Rd1) + (size_t)sid * nn;
    for (int e = lane; e < nn; e += NT) {
      int i, j; ediv(e, n, i, j);
      const double m = sM[i * LD + j], sv = Sg[e];
      const double dg = (i == j) ? 1.0 : 0.0;
      const double rd = (r == 0 ? (m - dg) : (tau * dg - m)) - sv;
      Rdg[e] = rd;
      rd2 = fma(rd, rd, rd2); s2 = fma(sv, sv, s2);
    }
  }
  wsync();
  TMPC_T(10)
  for (int r = 0; r < 2; ++r) {
    const double* Xg = (r ? w.X2 : w.X1) + (size_t)sid * nn;
    const double* Sg = (r ? w.S2 : w.S1) + (size_t)sid * nn;
    {
      double ldB[EPT];
      g2r<NT>(ldA, Xg, n, n, n, lane);
      g2r<NT>(ldB, Sg, n, n, n, lane);
      r2s<NT>(sX, ldA, n, n, lane);
      r2s<NT>(sS, ldB, n, n, lane);
    }
    TMPC_T(8)
    for (int e = lane; e < nn; e += NT) { int i, j; ediv(e, n, i, j); xs = fma(sX[i * LD + j], sS[i * LD + j], xs); }
    {
      double hx = 0.0;
#pragma unroll
      for (int q = 0; q < EPT; ++q) { const int e = lane + q * NT; int i, j; ediv(e, n, i, j); if (e < nn) hx = fma(hbr[q], sX[i * LD + j], hx); }
      hx = block_sum<NT>(hx);
      if (r == 0) hby += hx; else { hby -= hx; trx2 = trace_s<NT>(sX, n, lane); }
    }
    // S_r = L L', X_r = Lx Lx' (for the primal step length) and both inverses, the two matrices side by side in one wave
    for (int e = lane; e < nn; e += NT) { int i, j; ediv(e, n, i, j); t0[i * LD + j] = sX[i * LD + j]; }
    wsync();
    TMPC_T(10)
    if (NT == 256 && n == 32 && NMAX >= 32) {
      nbad += chol_inv_pair32<NT>(sS, sLi, t0, t1, lane);
      TMPC_T(11)
    } else {
      nbad += chol_lower_pair_t<NT>(sS, t0, n, lane);
      TMPC_T(11)
      tri_inv_lower_pair_t<NT>(sLi, sS, t1, t0, n, lane);
    }
    TMPC_T(12)
    s2g<NT>((r ? w.L2i : w.L1i) + (size_t)sid * nn, sLi, n, n, n, lane);
    s2g<NT>((r ? w.LX2i : w.LX1i) + (size_t)sid * nn, t1, n, n, n, lane);
    TMPC_T(13)
    mm<NT>(sSi, sLi, 1, LD, sLi, LD, 1, n, n, n, 0, lane);                   // Li' Li
    TMPC_T(14)
    s_sym<NT>(sSi, n, lane);
    s2g<NT>((r ? w.S2i : w.S1i) + (size_t)sid * nn, sSi, n, n, n, lane);
    TMPC_T(13)
    // Kronecker factors of the HKM Schur blocks
    double* kfr = kf + (size_t)r * KF_PER_LMI * nxx;
    mm<NT>(t0, sV, LD, 1, sX, LD, 1, nx, n, n, 0, lane);                     // V X     (nx x n)
    mm<NT>(t1, t0, LD, 1, sV, 1, LD, nx, nx, n, 0, lane);                    // V X V'
    s2g<NT>(kfr + KF_KX * nxx, t1, nx, nx, nx, lane);
    s2g_T<NT>(kfr + KF_FX * nxx, t0, nx, nx, nx, lane);                      // Fx = X[:nx,:] V' = ((VX)[:, :nx])'
    s2g<NT>(kfr + KF_XXX * nxx, sX, nx, nx, nx, lane);
    mm<NT>(t0, sV, LD, 1, sSi, LD, 1, nx, n, n, 0, lane);                    // V Si
    mm<NT>(t1, t0, LD, 1, sV, 1, LD, nx, nx, n, 0, lane);
    s2g<NT>(kfr + KF_KS * nxx, t1, nx, nx, nx, lane);
    s2g_T<NT>(kfr + KF_FS * nxx, t0, nx, nx, nx, lane);
    s2g<NT>(kfr + KF_SIXX * nxx, sSi, nx, nx, nx, lane);
    // Phi_r(Hb) = sym(X Hb Si)
    r2s<NT>(t1, hbr, n, n, lane);
    mm<NT>(t0, sX, LD, 1, t1, LD, 1, n, n, n, 0, lane);
    mm<NT>(t1, t0, LD, 1, sSi, LD, 1, n, n, n, 0, lane);
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int e = lane + q * NT;
      if (e < nn) {
        int i, j; ediv(e, n, i, j);
        const double phi = 0.5 * (t1[i * LD + j] + t1[j * LD + i]);
        phir[q] += phi;
        if (r == 1 && i == j) trphi2 += phi;
      }
    }
    wsync();
    if (r == 1) {
      mm<NT>(t0, sX, LD, 1, sSi, LD, 1, n, n, n, 0, lane);                   // Psi = sym(X2 S2i)
      s_sym<NT>(t0, n, lane);
      trpsi = trace_s<NT>(t0, n, lane);
      if (dm.nT > 0) s2g<NT>(w.t3psi + (size_t)sid * nn, t0, n, n, n, lane);
      adj_V<NT>(t1, t2, sV, t0, n, nx, lane);
      s2g<NT>(w.adjV + ((size_t)sid * NADJ + ADJ_PSI) * nxx, t1, nx, nx, nx, lane);
      s2g<NT>(w.adjE + ((size_t)sid * NADJ + ADJ_PSI) * nxx, t0, nx, nx, nx, lane);
    }
  }
  TMPC_T(15)
  double hbphi = 0.0;
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int e = lane + q * NT;
    if (e < nn) { int i, j; ediv(e, n, i, j); sPhi[i * LD + j] = phir[q]; hbphi = fma(hbr[q], phir[q], hbphi); }
  }
  wsync();
  hbphi = block_sum<NT>(hbphi);
  if (dm.nT > 0) s2g<NT>(w.t3phi + (size_t)sid * nn, sPhi, n, n, n, lane);
  adj_V<NT>(t1, t2, sV, sPhi, n, nx, lane);
  s2g<NT>(w.adjV + ((size_t)sid * NADJ + ADJ_PHI) * nxx, t1, nx, nx, nx, lane);
  s2g<NT>(w.adjE + ((size_t)sid * NADJ + ADJ_PHI) * nxx, sPhi, nx, nx, nx, lane);
  rd2 = block_sum<NT>(rd2); s2 = block_sum<NT>(s2); xs = block_sum<NT>(xs); trphi2 = block_sum<NT>(trphi2);
  if (lane == 0) {
    double* q = w.part + (size_t)sid * NPART;
    q[Q_XS] = xs; q[Q_RD2] = rd2; q[Q_S2] = s2; q[Q_TRX2] = trx2; q[Q_HBY] = hby;
    q[Q_TRPSI] = trpsi; q[Q_TRPHI2] = trphi2; q[Q_HBPHI] = hbphi; q[Q_CHOLBAD] = (double)nbad;
  }
}

constexpr int RHS_SLOTS = 6;      // three blocks per CU (see PRE_SLOTS): G = T1 - T2 accumulates in registers, Hb is read elementwise from memory
// ------------------------------------------------------------------ stage_rhs: T_r and the adjoint of G = T1 - T2
// pass 1 = predictor (sigma*mu = 0, no corrector term; main-phase problems only)
// pass 2 = corrector (main phase) or pure centering step (centering phase)
template <int NT>
__device__ __forceinline__ void stage_rhs_body(const WS& w, const Dims& dm, int sid, int lane, double* sm, int pass) {
  const int b = sid / dm.p;
  const int k = sid - b * dm.p;
  const int n = dm.n, nx = dm.nx;
  const int nn = n * n, nxx = nx * nx;
  (void)k; (void)nn; (void)nxx;
  const int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return;
  const double* pr = w.prob + (size_t)b * PS;
  const double sig = (pass == 1) ? 0.0 : pr[P_SIGMU];
  const bool use_corr = (pass == 2 && phase == PH_MAIN);
  double* sV = sm; double* sX = sm + MS; double* sSi = sm + 2 * MS; double* sRd = sm + 3 * MS;
  double* t0 = sm + 4 * MS; double* t1 = sm + 5 * MS;
  double* sG = sRd;       // G goes to LDS after the last use of Rd
  constexpr int EPT = NMAX * NMAX / NT;                     // elements of an n x n matrix per thread (element e = lane + q NT)
  double gr[EPT];
#pragma unroll
  for (int q = 0; q < EPT; ++q) gr[q] = 0.0;
  g2s<NT>(sV, w.V + (size_t)sid * nx * n, nx, n, n, lane);
  double trt2 = 0.0;
  for (int r = 0; r < 2; ++r) {
    {                                        // the three operands in flight together (one exposed memory latency)
      double l0[EPT], l1[EPT], l2[EPT];
      g2r<NT>(l0, (r ? w.X2 : w.X1) + (size_t)sid * nn, n, n, n, lane);
      g2r<NT>(l1, (r ? w.S2i : w.S1i) + (size_t)sid * nn, n, n, n, lane);
      g2r<NT>(l2, (r ? w.Rd2 : w.Rd1) + (size_t)sid * nn, n, n, n, lane);
      r2s<NT>(sX, l0, n, n, lane); r2s<NT>(sSi, l1, n, n, lane); r2s<NT>(sRd, l2, n, n, lane);
    }
    mm<NT>(t0, sX, LD, 1, sRd, LD, 1, n, n, n, 0, lane);
    mm<NT>(t1, t0, LD, 1, sSi, LD, 1, n, n, n, 0, lane);
    double* Tg = (r ? w.T2 : w.T1) + (size_t)sid * nn;
    const double* cg = (r ? w.c2 : w.c1) + (size_t)sid * nn;
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int e = lane + q * NT;
      if (e < nn) {
        int i, j; ediv(e, n, i, j);
        double t = sig * sSi[i * LD + j] - 0.5 * (t1[i * LD + j] + t1[j * LD + i]);
        if (use_corr) t -= cg[e];
        Tg[e] = t;
        if (r == 0) gr[q] = t; else { gr[q] -= t; if (i == j) trt2 += t; }
      }
    }
    wsync();
  }
  double hbg = 0.0;
  {
    const double* Hbg = w.Hb + (size_t)sid * nn;
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int e = lane + q * NT;
      if (e < nn) { int i, j; ediv(e, n, i, j); sG[i * LD + j] = gr[q]; hbg = fma(Hbg[e], gr[q], hbg); }
    }
  }
  wsync();
  hbg = block_sum<NT>(hbg);
  adj_V<NT>(t1, t0, sV, sG, n, nx, lane);
  s2g<NT>(w.adjV + ((size_t)sid * NADJ + ADJ_G) * nxx, t1, nx, nx, nx, lane);
  s2g<NT>(w.adjE + ((size_t)sid * NADJ + ADJ_G) * nxx, sG, nx, nx, nx, lane);
  trt2 = block_sum<NT>(trt2);
  if (lane == 0) {
    double* q = w.part + (size_t)sid * NPART;
    q[Q_TRT2] = trt2; q[Q_HBG] = hbg;
  }
}

constexpr int DIR_SLOTS = 6;      // three blocks per CU (see PRE_SLOTS): dM in registers after it is built, V / Hb / L^-1 share slots with X, S^-1, dS
// ------------------------------------------------------------------ stage_dir: dS, dX, step-length eigenvalues, corrector term
template <int NT>
__device__ __forceinline__ void stage_dir_body(const WS& w, const Dims& dm, int sid, int lane, double* sm, int pass) {
  const int b = sid / dm.p;
  const int k = sid - b * dm.p;
  const int n = dm.n, nx = dm.nx;
  const int nn = n * n, nxx = nx * nx;
  (void)k; (void)nn; (void)nxx;
  const int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return;
  const double* pr = w.prob + (size_t)b * PS;
  const double dtau = pr[P_DTAU], dalpha = pr[P_DALPHA];
  double* sX = sm; double* sSi = sm + MS; double* sDS = sm + 2 * MS; double* sDX = sm + 3 * MS; double* t0 = sm + 4 * MS; double* t1 = sm + 5 * MS;
  double* sV = sX; double* sHb = sSi; double* sM = sDS;    // V, Hb and the slot of dM are only needed to build dM (kept in registers: dmr)
  double* sL = sX;        // X is dead once dX is known
  constexpr int EPT = NMAX * NMAX / NT;                     // elements of an n x n matrix per thread (element e = lane + q NT)
  double dmr[EPT];
  const int kn = (k + 1 == dm.p) ? 0 : k + 1;
  {
    double l0[EPT], l1[EPT];
    g2r<NT>(l0, w.V + (size_t)sid * nx * n, nx, n, n, lane);
    g2r<NT>(l1, w.Hb + (size_t)sid * nn, n, n, n, lane);
    r2s<NT>(sV, l0, nx, n, lane); r2s<NT>(sHb, l1, n, n, lane);
  }
  const double* dPk = w.dP + (size_t)sid * nxx;
  build_M<NT>(sM, sV, t0, t1, sHb, dPk, w.dP + (size_t)(b * dm.p + kn) * nxx, dalpha, n, nx, lane);   // dM
  if (dm.nr > 0) add_gtg<NT>(sM, w.G + (size_t)sid * dm.nr * n, w.dphi + (size_t)sid * dm.nr, 1.0, stage_rows(w, dm, sid), n, lane);   // + G' diag(dphi) G
  if (dm.nT > 0) add_smat_t3<NT>(sM, w.t3dth + (size_t)sid * dm.nT, 1.0, n, lane);                                                       // + dT_k
#pragma unroll
  for (int q = 0; q < EPT; ++q) { const int e = lane + q * NT; int i, j; ediv(e, n, i, j); dmr[q] = (e < nn) ? sM[i * LD + j] : 0.0; }
  wsync();
  double dxs = 0.0, xds = 0.0, dxds = 0.0;
  for (int r = 0; r < 2; ++r) {
    const double* Xg = (r ? w.X2 : w.X1) + (size_t)sid * nn;
    const double* Sg = (r ? w.S2 : w.S1) + (size_t)sid * nn;
    const double* Rdg = (r ? w.Rd2 : w.Rd1) + (size_t)sid * nn;
    const double* Tg = (r ? w.T2 : w.T1) + (size_t)sid * nn;
    double* dSg = (r ? w.dS2 : w.dS1) + (size_t)sid * nn;
    double* dXg = (r ? w.dX2 : w.dX1) + (size_t)sid * nn;
    double lL[EPT], lLX[EPT];                // L_r^-1 and LX_r^-1: needed further down, fetched with X and S^-1 (one exposed memory latency)
    {
      double l0[EPT], l1[EPT];
      g2r<NT>(l0, Xg, n, n, n, lane);
      g2r<NT>(l1, (r ? w.S2i : w.S1i) + (size_t)sid * nn, n, n, n, lane);
      g2r<NT>(lL, (r ? w.L2i : w.L1i) + (size_t)sid * nn, n, n, n, lane);
      g2r<NT>(lLX, (r ? w.LX2i : w.LX1i) + (size_t)sid * nn, n, n, n, lane);
      r2s<NT>(sX, l0, n, n, lane); r2s<NT>(sSi, l1, n, n, lane);
    }
    // sDX := Ldy = dS - Rd   (linear part of the slack direction)
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int e = lane + q * NT;
      if (e < nn) {
        int i, j; ediv(e, n, i, j);
        const double dm_ = dmr[q];
        const double ldy = (r == 0) ? dm_ : ((i == j ? dtau : 0.0) - dm_);
        sDX[i * LD + j] = ldy;
        const double ds = ldy + Rdg[e];
        sDS[i * LD + j] = ds;
        dSg[e] = ds;
      }
    }
    wsync();
    mm<NT>(t0, sX, LD, 1, sDX, LD, 1, n, n, n, 0, lane);
    mm<NT>(t1, t0, LD, 1, sSi, LD, 1, n, n, n, 0, lane);
    for (int e = lane; e < nn; e += NT) {
      int i, j; ediv(e, n, i, j);
      const double dx = Tg[e] - sX[i * LD + j] - 0.5 * (t1[i * LD + j] + t1[j * LD + i]);
      sDX[i * LD + j] = dx;
      dXg[e] = dx;
      const double ds = sDS[i * LD + j];
      dxs = fma(dx, Sg[e], dxs); xds = fma(sX[i * LD + j], ds, xds); dxds = fma(dx, ds, dxds);
    }
    wsync();
    // step-length matrices  W_S = L^-1 dS L^-T  and  W_X = LX^-1 dX LX^-T ; their smallest eigenvalues are
    // computed by k_eigmin (one wave per matrix, 16 waves per CU) -- slots 2r (dual) and 2r+1 (primal)
    r2s<NT>(sL, lL, n, n, lane);
    mm<NT>(t0, sL, LD, 1, sDS, LD, 1, n, n, n, 0, lane);
    mm<NT>(t1, t0, LD, 1, sL, 1, LD, n, n, n, 0, lane);
    s2g_sym<NT>(w.Wm + ((size_t)sid * 4 + 2 * r) * nn, t1, n, lane);
    r2s<NT>(sL, lLX, n, n, lane);
    mm<NT>(t0, sL, LD, 1, sDX, LD, 1, n, n, n, 0, lane);
    mm<NT>(t1, t0, LD, 1, sL, 1, LD, n, n, n, 0, lane);
    s2g_sym<NT>(w.Wm + ((size_t)sid * 4 + 2 * r + 1) * nn, t1, n, lane);
    if (pass == 1) {   // Mehrotra second-order term  sym(dX dS S^-1)
      mm<NT>(t0, sDX, LD, 1, sDS, LD, 1, n, n, n, 0, lane);
      mm<NT>(t1, t0, LD, 1, sSi, LD, 1, n, n, n, 0, lane);
      s2g_sym<NT>((r ? w.c2 : w.c1) + (size_t)sid * nn, t1, n, lane);
    }
  }
  // first-order relative change of the output Hc_k = M_k/(s*alpha):  dM_k - (dalpha/alpha) M_k,  M_k = S1 + Rd1 + I
  double dh2 = 0.0, m2 = 0.0;
  {
    const double ra = dalpha / pr[P_ALPHA];
    const double* S1g = w.S1 + (size_t)sid * nn; const double* R1g = w.Rd1 + (size_t)sid * nn;
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int e = lane + q * NT;
      if (e < nn) {
        int i, j; ediv(e, n, i, j);
        const double m = S1g[e] + R1g[e] + (i == j ? 1.0 : 0.0);
        const double dh = dmr[q] - ra * m;
        dh2 = fma(dh, dh, dh2); m2 = fma(m, m, m2);
      }
    }
    dh2 = block_sum<NT>(dh2); m2 = block_sum<NT>(m2);
  }
  double dp2 = 0.0, p2 = 0.0;
  const double* Pk = w.P + (size_t)sid * nxx;
  for (int e = lane; e < nxx; e += NT) { dp2 = fma(dPk[e], dPk[e], dp2); p2 = fma(Pk[e], Pk[e], p2); }
  dxs = block_sum<NT>(dxs); xds = block_sum<NT>(xds); dxds = block_sum<NT>(dxds); dp2 = block_sum<NT>(dp2); p2 = block_sum<NT>(p2);
  if (lane == 0) {
    double* q = w.part + (size_t)sid * NPART;
    q[Q_DXS] = dxs; q[Q_XDS] = xds; q[Q_DXDS] = dxds; q[Q_DP2] = dp2; q[Q_P2] = p2; q[Q_DH2] = dh2; q[Q_M2] = m2;
  }
}

// one single-wave block per matrix (4 per stage): 8.9 KB of LDS each -> ~17 blocks (waves) resident per CU
// The largest step theta* = -1 / lambda_min(W) that keeps a cone block positive definite is needed EXACTLY only when it is short: the
// control kernels clip every step at 1 (k_ctrl_b: min(1, theta*); k_ctrl_c: min(1, gamma theta*), gamma >= 0.9, in the main phase, min(1, 0.95 theta*) while
// centering) and ask one more thing, whether theta* >= chord_step (the chord decision).  So the wave first asks whether I + theta W is positive
// definite at those thresholds -- a Cholesky sweep that stops at the first bad pivot, a fifth of the tridiagonalisation + Sturm search --
// and reports -1 / theta (a bound that yields the same clipped step and the same decision as the exact value) when it is; the eigenvalue
// is computed only for the blocks that fail.  Per problem the minimum over stages is exact whenever any block is short (a failing block has
// lambda_min <= -1 / theta <= every reported bound).  Same iterates bit for bit; the raw step in the trace is the bound.  Pass 2 only: the
// affine step of pass 1 is short in some block of most problems, and asking first cost more than it saved there.  On the bench batch 54 % of
// the questions are answered by the sweep (no wrong answer in 4.05 M, checked against the eigenvalue with -DTMPC_EIG_DEBUG); pass 2 -4 %,
// bench line +0.8 % (profiles/r3_eig_pretest_ab.txt).  (Round 3.)
__device__ __forceinline__ void eigmin_body(const WS& w, const Dims& dm, int mid, int lane, double* sm, int pass, double chord_step) {      // mid = (b*p + k)*4 + which
  const int b = (mid >> 2) / dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return;
  const int n = dm.n;
  double* A = sm;
  double* cs = A + MS;
  const double* Wg = w.Wm + (size_t)mid * n * n;
  TMPC_TC0()
  g2s(A, Wg, n, n, n, lane);               // (four loads in flight per lane)
  TMPC_TC(5, 0)
  // thresholds, longest first: the chord decision (centering only), then "the clipped step is 1"
  double th[2]; int nth = 0;
  if (chord_step < 0.0) {}                                                      // TMPC_EIG_PRETEST=0: every eigenvalue
  else if (pass == 1) {}                                                        // affine step: min(1, theta*) -- usually short in some block; asking first cost more than it saved (pass 1 +0.7 %)
  else if (phase == PH_MAIN) th[nth++] = (1.0 / 0.9) * (1.0 + 1e-9);             // gamma in [0.9, 0.99] (the other cone may be the short one): gamma theta >= 1
  else {
    if (chord_step > 1.0 / TMPC_CENTER_DAMP) th[nth++] = chord_step * (1.0 + 1e-9);
    th[nth++] = (1.0 / TMPC_CENTER_DAMP) * (1.0 + 1e-9);
  }
  for (int t = 0; t < nth; ++t) {
    if (shifted_is_pd(A, th[t], n, lane)) {
#ifdef TMPC_EIG_DEBUG
      wsync(); g2s(A, Wg, n, n, n, lane);
      const double ex = tridiag_min_eig(A, n, cs, lane);
      if (lane == 0) { atomicAdd(&g_eig_dbg[0], 1); if (ex <= -1.0 / th[t]) { if (atomicAdd(&g_eig_dbg[1], 1) == 0) { g_eig_dbgv[0] = ex; g_eig_dbgv[1] = th[t]; g_eig_dbgv[2] = pass; g_eig_dbgv[3] = phase; } } }
#endif
      if (lane == 0) w.eigmin[mid] = -1.0 / th[t];
      return;
    }
    wsync();
    g2s(A, Wg, n, n, n, lane);             // the sweep destroyed its copy (L2-hot)
#ifdef TMPC_EIG_DEBUG
    { const double ex = tridiag_min_eig(A, n, cs, lane);
      if (lane == 0) { atomicAdd(&g_eig_dbg[2], 1); if (ex > -1.0 / th[t]) { if (atomicAdd(&g_eig_dbg[3], 1) == 0) { g_eig_dbgv[4] = ex; g_eig_dbgv[5] = th[t]; g_eig_dbgv[6] = pass; g_eig_dbgv[7] = phase; } } }
      wsync(); g2s(A, Wg, n, n, n, lane); }
#endif
  }
  const double lo = tridiag_min_eig(A, n, cs, lane);
  TMPC_TC(5, 3)
  if (lane == 0) w.eigmin[mid] = lo;
}

// ------------------------------------------------------------------ update: X += ap dX, S += ad dS, P += ad dP
template <int NT>
__device__ __forceinline__ void update_body(const WS& w, const Dims& dm, int sid, int lane) {
  const int b = sid / dm.p;
  const int n = dm.n, nx = dm.nx;
  const int nn = n * n, nxx = nx * nx;
  const int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] == PH_DONE) return;
  const double* pr = w.prob + (size_t)b * PS;
  const double ap = pr[P_AP], ad = pr[P_AD];
  if (ap == 0.0 && ad == 0.0) return;      // discarded direction (it may hold NaN: 0 * NaN would poison the iterate)
  const size_t o = (size_t)sid * nn;
  for (int e = lane; e < nn; e += NT) {
    int i, j; ediv(e, n, i, j);
    const int et = j * n + i;
    if (j <= i) {
      const double x1 = 0.5 * ((w.X1[o + e] + ap * w.dX1[o + e]) + (w.X1[o + et] + ap * w.dX1[o + et]));
      const double x2 = 0.5 * ((w.X2[o + e] + ap * w.dX2[o + e]) + (w.X2[o + et] + ap * w.dX2[o + et]));
      const double s1 = 0.5 * ((w.S1[o + e] + ad * w.dS1[o + e]) + (w.S1[o + et] + ad * w.dS1[o + et]));
      const double s2 = 0.5 * ((w.S2[o + e] + ad * w.dS2[o + e]) + (w.S2[o + et] + ad * w.dS2[o + et]));
      w.X1[o + e] = x1; w.X1[o + et] = x1; w.X2[o + e] = x2; w.X2[o + et] = x2;
      w.S1[o + e] = s1; w.S1[o + et] = s1; w.S2[o + e] = s2; w.S2[o + et] = s2;
    }
  }
  for (int e = lane; e < nxx; e += NT) w.P[(size_t)sid * nxx + e] += ad * w.dP[(size_t)sid * nxx + e];
}
