// CPU baseline, the models WITH ROWS  --  TEST / MEASUREMENT INFRASTRUCTURE ONLY (included by cpu_ipm.cpp inside its anonymous namespace).
//
// C++ restatement of oracle/convexify_oracle.py::sdp_step1 for every model of tunempc/convexifier.py beyond the plain one:
//   * Step 1 with the cost-free multipliers Fg_k >= 0 of the equality-constraint Jacobians G_k        (convexifier.py:249-255, :346-347)
//   * Step 2: multipliers F_k >= 0 of the ragged active-constraint Jacobians C_k (:258-266, :348-350), objective
//     beta + rho (||F_k|| + ||Fg_k||) (:276-283), every norm as an arrow LMI [[t, w v'], [w v, t I]] >> 0; or the beta-only reading (cost_free)
//   * Step 3: T_k symmetric with every entry > 0 (:269-273, :352-353), rho ||T_k||_F (:284-285) as a second-order cone.
// Same interior-point iteration as solve_problem() (same start point, step rules, centering phase, back-offs).  What differs from the
// numpy oracle is only WHERE the stage-local variables live in the linear algebra: the oracle carries them as dense border columns
// (2 + p ng + arrows of them: minutes at p = 64, hours at p = 200); here, as in the HIP library (tmpc_phi.h), the locals of stage k ride
// behind svec(P_{k+1}) inside block k+1 of the block-cyclic-tridiagonal Schur matrix -- their column touches P_k (coupling block C_k)
// and P_{k+1} (diagonal block) and nothing else -- so a problem with rows costs (1 + nz/d)^3 of a plain one.  Same Newton system, another
// elimination order: the two agree to rounding (tests/test_cpu_ipm.py).

// ---- second-order cone helpers (convexify_oracle.py: _soc_*; Jordan algebra of Q^{m+1})
static inline double soc_det(const vec& u) { double s = u[0] * u[0]; for (size_t i = 1; i < u.size(); ++i) s -= u[i] * u[i]; return s; }
static inline vec soc_inv(const vec& u) { const double dt = soc_det(u); vec r(u.size()); r[0] = u[0] / dt; for (size_t i = 1; i < u.size(); ++i) r[i] = -u[i] / dt; return r; }
static inline double vdot(const vec& a, const vec& b) { double s = 0; for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i]; return s; }
static inline double vdot1(const vec& a, const vec& b) { double s = 0; for (size_t i = 1; i < a.size(); ++i) s += a[i] * b[i]; return s; }
static inline vec soc_prod(const vec& u, const vec& v) { vec r(u.size()); r[0] = vdot(u, v); for (size_t i = 1; i < u.size(); ++i) r[i] = u[0] * v[i] + v[0] * u[i]; return r; }
static inline vec soc_div(const vec& lam, const vec& r) {      // solve lam o u = r
  const double dl = soc_det(lam), lr = vdot1(lam, r);
  vec u(lam.size());
  u[0] = (lam[0] * r[0] - lr) / dl;
  for (size_t i = 1; i < lam.size(); ++i) u[i] = (-r[0] * lam[i] + (dl * r[i] + lr * lam[i]) / lam[0]) / dl;
  return u;
}
static inline void soc_scaling(const vec& s, const vec& x, double& beta, vec& v) {      // NT scaling W = beta (2 v v' - J)
  const double ds_ = soc_det(s), dx_ = soc_det(x), rs = sqrt(ds_), rx = sqrt(dx_);
  const size_t m1 = s.size();
  vec sb(m1), xb(m1);
  for (size_t i = 0; i < m1; ++i) { sb[i] = s[i] / rs; xb[i] = x[i] / rx; }
  const double gam = sqrt((1.0 + vdot(xb, sb)) / 2.0);
  v.resize(m1);
  for (size_t i = 0; i < m1; ++i) v[i] = (sb[i] + (i == 0 ? xb[0] : -xb[i])) / (2.0 * gam);
  const double wb0 = v[0];
  v[0] += 1.0;
  const double sc = sqrt(2.0 * (wb0 + 1.0));
  for (size_t i = 0; i < m1; ++i) v[i] /= sc;
  beta = pow(ds_ / dx_, 0.25);
}
static inline vec soc_W(double beta, const vec& v, const vec& u, bool inverse) {
  const size_t m1 = u.size();
  vec r(m1);
  if (inverse) {
    double ju = v[0] * u[0]; for (size_t i = 1; i < m1; ++i) ju -= v[i] * u[i];
    for (size_t i = 0; i < m1; ++i) { const double jv = (i == 0 ? v[0] : -v[i]), Ju = (i == 0 ? u[0] : -u[i]); r[i] = (2.0 * jv * ju - Ju) / beta; }
  } else {
    const double vu = vdot(v, u);
    for (size_t i = 0; i < m1; ++i) { const double Ju = (i == 0 ? u[0] : -u[i]); r[i] = beta * (2.0 * v[i] * vu - Ju); }
  }
  return r;
}
static inline void soc_W2inv(double beta, const vec& v, vec& W2) {      // W^-2, dense (m+1) x (m+1)
  const size_t m1 = v.size();
  const double vv = vdot(v, v), b2 = beta * beta;
  W2.assign(m1 * m1, 0.0);
  for (size_t i = 0; i < m1; ++i) for (size_t j = 0; j < m1; ++j) {
    const double jvi = (i == 0 ? v[0] : -v[i]), jvj = (j == 0 ? v[0] : -v[j]);
    W2[i * m1 + j] = ((i == j ? 1.0 : 0.0) + 4.0 * vv * jvi * jvj - 2.0 * (jvi * v[j] + v[i] * jvj)) / b2;
  }
}
static inline double soc_max_step(const vec& u, const vec& du) {
  const double a = du[0] * du[0] - vdot1(du, du), b = u[0] * du[0] - vdot1(u, du), c = soc_det(u);
  double best = 1e300;
  if (du[0] < 0) best = std::min(best, -u[0] / du[0]);
  if (fabs(a) < 1e-300) { if (b < 0) best = std::min(best, -c / (2.0 * b)); }
  else {
    const double disc = b * b - a * c;
    if (disc >= 0) { const double sq = sqrt(disc); const double r1 = (-b - sq) / a, r2 = (-b + sq) / a; if (r1 > 0) best = std::min(best, r1); if (r2 > 0) best = std::min(best, r2); }
  }
  return best;
}

struct ConIn {
  int ng0 = 0, ncmax = 0;            // rows of G_k per stage; rows per stage of the padded C array (J = [G_k; C_k padded], [p][ng0 + ncmax][n])
  const double* J = nullptr;
  const int32_t* ncnt = nullptr;     // [p] active rows of C_k (may be 0: convexifier.py:261-266)
  double rho = 0.0;
  bool constr = false, cost_free = false, force = false;
};
struct ConOut { double* P = nullptr; double* FgF = nullptr; double* T = nullptr; double objective = 0.0; };

struct Arrow {
  int k = 0, slot = 0, m = 0; bool soc = false;
  std::vector<int> idx;
  double t = 0, dt = 0;
  vec X, S, Si, Tm, dX, dS, corr, Pe;      // arrow LMI, (m+1) x (m+1) row-major
  vec x, s, g, dx, ds, W2, lam, v, corrv;  // second-order cone, m + 1
  double beta = 1.0;
};

static inline double arrow_start(double a, int m, double mu) {      // convexify_oracle._arrow_start
  double lo = mu, hi = (m + 1.0) * mu;
  for (int i = 0; i < 60; ++i) {
    const double dl = 0.5 * (lo + hi);
    const double f = mu * (1.0 / dl + 1.0 / (dl + 2.0 * a) + (m - 1.0) / (dl + a)) - 1.0;
    if (f > 0.0) lo = dl; else hi = dl;
  }
  return a + 0.5 * (lo + hi);
}
// S = [[t, w v'], [w v, t I]]
static inline void arrow_fill(vec& S, double t, const double* v, double w, int m) {
  const int m1 = m + 1;
  S.assign((size_t)m1 * m1, 0.0);
  for (int i = 0; i < m1; ++i) S[(size_t)i * m1 + i] = t;
  for (int i = 0; i < m; ++i) { S[i + 1] = w * v[i]; S[(size_t)(i + 1) * m1] = w * v[i]; }
}
static bool spd_inv(const Small& sm, vec& out, const vec& S, vec& L, vec& Li) {      // out = S^-1 via Cholesky; L, Li left for the step lengths
  const int n = sm.n;
  L.resize((size_t)n * n); Li.resize((size_t)n * n); out.resize((size_t)n * n);
  if (!sm.chol(L.data(), S.data())) return false;
  sm.tri_inv(Li.data(), L.data());
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double v = 0; for (int r = 0; r < n; ++r) v += Li[r * n + i] * Li[r * n + j]; out[i * n + j] = v; }
  return true;
}


// ---------------------------------------------------------------- tight mode with rows (round 6): the stage-local rows of the augmented blocks in double-double
// As the HIP library (tmpc_dd.h: k_dd_aug_fill): the rows of a multiplier are double-double FUNCTIONS of the same (X_r, Z_r = S_r^-1) the P part of the blocks is
// made of -- w = X g, u = Z g, V w, V u, the Gram products -- so the augmented block is, to double-double accuracy, the Gram matrix it is in exact arithmetic and its
// pivots stay positive where a pivot of an active multiplier is what is left of ~1/mu after ~1/mu has been taken off.  (The numpy oracle carries these variables as
// dense border columns, forms its Schur complement in fp64 and solves it by LU: nothing there needs a positive pivot.)
static inline ddk::dd dd_dot(const std::vector<ddk::dd>& a, const double* b, int n) { ddk::dd s{0.0, 0.0}; for (int i = 0; i < n; ++i) s = ddk::add(s, ddk::muld(a[i], b[i])); return s; }
// closed-form inverse of the arrow matrix [[t, u'], [u, t I]] in dd, gamma = t^2 - |u|^2 (cancels to ~ mu t on an active norm); false outside the cone
static bool arrow_inv_dd(ddk::mat& Si, double t, const std::vector<ddk::dd>& u, int m) {
  using namespace ddk;
  const int m1 = m + 1;
  if (!(t > 0.0)) return false;
  dd gam = tp(t, t);
  for (int i = 0; i < m; ++i) gam = sub(gam, mul(u[i], u[i]));
  if (!(gam.h > 0.0)) return false;
  Si.assign((size_t)m1 * m1, dd{0.0, 0.0});
  const dd td = from(t);
  Si[0] = div(td, gam);
  for (int i = 0; i < m; ++i) { const dd v = neg(div(u[i], gam)); Si[i + 1] = v; Si[(size_t)(i + 1) * m1] = v; }
  for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) {
    dd v = div(mul(u[i], u[j]), gam);
    if (i == j) v = add(v, from(1.0));
    Si[(size_t)(i + 1) * m1 + j + 1] = div(v, td);
  }
  return true;
}

Result solve_problem_con(int p, int nx, int mb, const double* A, const double* B, const double* Hin, const ConIn& ci, double tol, int max_iter, int center_iter,
                         double center_tol, double* Hc_out, ConOut& co, bool tight = false, bool par = false) {
  const int n = nx + mb, nn = n * n, d = nx * (nx + 1) / 2, nxx = nx * nx;
  Small sm(n);
  Result res; res.kappa = 0; res.alpha = 1; res.status = 0; res.iters = 0; res.early = 0;
  const int nJin = ci.ng0 + ci.ncmax;
  vec Hs((size_t)p * nn);
  for (int k = 0; k < p; ++k) for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j)
    Hs[((size_t)k * n + i) * n + j] = 0.5 * (Hin[((size_t)k * n + i) * n + j] + Hin[((size_t)k * n + j) * n + i]);
  vec work((size_t)nn + 4 * n + 64), ev(n);
  double lo = 1e300, amin = 1e10, amax = 0.0;
  for (int k = 0; k < p; ++k) {
    sm.eigvals(&Hs[(size_t)k * nn], ev.data(), work.data());
    for (int i = 0; i < n; ++i) { lo = std::min(lo, ev[i]); const double a = fabs(ev[i]); if (a != 0.0) { amin = std::min(amin, a); amax = std::max(amax, a); } }
  }
  if (co.P) std::fill(co.P, co.P + (size_t)p * nxx, 0.0);
  if (co.FgF) std::fill(co.FgF, co.FgF + (size_t)p * nJin, 0.0);
  if (co.T) std::fill(co.T, co.T + (size_t)p * nn, 0.0);
  if (lo > 0.0) { memcpy(Hc_out, Hs.data(), sizeof(double) * p * nn); res.early = 1; return res; }      // convexifier.py:82-85
  const double s = 1.0 / amin, sbeta = amax / amin;
  Problem pr; pr.setup(p, nx, mb, A, B, Hs.data());
  pr.Hb.resize((size_t)p * nn);
  for (size_t e = 0; e < (size_t)p * nn; ++e) pr.Hb[e] = s * Hs[e];
  const vec& Hb = pr.Hb;
  double N = 2.0 * p * n + 1.0;
  double tau = 2.0, alpha = 1.0 / sbeta, s0 = alpha, x0 = 1.0 / (p * n);
  // ---- the stage-local variables (convexify_oracle.py:744-809)
  const bool constr = ci.constr, cost_free = ci.cost_free || (constr && ci.rho == 0.0), force = ci.force;
  std::vector<int> ncs(p, 0);
  int maxnc = 0;
  if (constr) for (int k = 0; k < p; ++k) { ncs[k] = ci.ncnt ? ci.ncnt[k] : ci.ncmax; maxnc = std::max(maxnc, ncs[k]); }
  const int ng0 = ci.ng0, nJ = ng0 + maxnc, nT = force ? n * (n + 1) / 2 : 0, ng = nJ + nT;
  const bool norms = (constr && !cost_free);
  const int slotG = (norms && ng0 > 0) ? 0 : -1, slotC = (norms && maxnc > 0) ? (slotG + 1) : -1, slotT = force ? (std::max(slotG, slotC) + 1) : -1;
  const int na = std::max(std::max(slotG, slotC), slotT) + 1, nz = ng + na;
  const int db = d + nz;
  pr.db = db;
  std::vector<int> ta, tb;
  for (int a = 0; a < n && force; ++a) for (int b = a; b < n; ++b) { ta.push_back(a); tb.push_back(b); }
  // direction matrices GG[k][i] (n x n): g g' per row, E_ab per entry of T_k (the same for every stage)
  vec GGr((size_t)p * nJ * nn, 0.0), GGt((size_t)nT * nn, 0.0), cw(std::max(ng, 1), 1.0);
  std::vector<char> mask((size_t)p * std::max(ng, 1), 0);
  for (int k = 0; k < p; ++k) {
    for (int i = 0; i < nJ; ++i) {
      const bool real = (i < ng0) || (i - ng0 < ncs[k]);
      if (!real) continue;
      const double* g = ci.J + ((size_t)k * nJin + (i < ng0 ? i : ng0 + (i - ng0))) * n;
      mask[(size_t)k * ng + i] = 1;
      double* M = &GGr[((size_t)k * nJ + i) * nn];
      for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) M[a * n + b] = g[a] * g[b];
    }
    for (int q = 0; q < nT; ++q) mask[(size_t)k * ng + nJ + q] = 1;
  }
  for (int q = 0; q < nT; ++q) { GGt[(size_t)q * nn + ta[q] * n + tb[q]] = 1.0; GGt[(size_t)q * nn + tb[q] * n + ta[q]] = 1.0; cw[nJ + q] = (ta[q] == tb[q]) ? 1.0 : sqrt(2.0); }
  auto gg = [&](int k, int i) -> const double* { return i < nJ ? &GGr[((size_t)k * nJ + i) * nn] : &GGt[(size_t)(i - nJ) * nn]; };
  vec phi((size_t)p * std::max(ng, 1), 1.0), z((size_t)p * std::max(ng, 1), 0.0);
  for (int k = 0; k < p; ++k) for (int i = 0; i < ng; ++i) if (mask[(size_t)k * ng + i]) {
    const double* M = gg(k, i); double f2 = 0; for (int e = 0; e < nn; ++e) f2 += M[e] * M[e];
    const double g2 = std::max(sqrt(f2), 1e-300);
    phi[(size_t)k * ng + i] = std::min(1.0, 1.0 / g2); z[(size_t)k * ng + i] = x0 / phi[(size_t)k * ng + i];
    N += 1.0;
  }
  const double wr = (norms || force) ? ci.rho * sbeta / s : 0.0;
  std::vector<Arrow> arrows;
  if (norms || force) {
    for (int k = 0; k < p; ++k) {
      struct Blk { int slot, i0, m; bool soc; };
      std::vector<Blk> blocks;
      if (norms && ng0) blocks.push_back({slotG, 0, ng0, false});
      if (norms && ncs[k]) blocks.push_back({slotC, ng0, ncs[k], false});
      if (nT) blocks.push_back({slotT, nJ, nT, true});
      for (const Blk& b : blocks) {
        Arrow a; a.k = k; a.slot = b.slot; a.m = b.m; a.soc = b.soc;
        double w2 = 0;
        for (int i = 0; i < b.m; ++i) { a.idx.push_back(b.i0 + i); w2 += cw[b.i0 + i] * cw[b.i0 + i]; }
        const double ph = std::min(1.0, x0 * sqrt(w2) / wr);
        for (int i : a.idx) { phi[(size_t)k * ng + i] = ph; z[(size_t)k * ng + i] = x0 / ph; }
        if (b.soc) {
          const double a_ = wr * ph * sqrt(w2);
          a.t = 0.5 * (x0 + sqrt(x0 * x0 + 4.0 * a_ * a_));
          a.s.resize(b.m + 1); a.s[0] = a.t;
          for (int i = 0; i < b.m; ++i) a.s[i + 1] = wr * cw[a.idx[i]] * phi[(size_t)k * ng + a.idx[i]];
          a.x = soc_inv(a.s); for (double& v : a.x) v *= x0;
          N += 1.0;
        } else {
          a.t = arrow_start(wr * ph * sqrt(w2), b.m, x0);
          vec vv(b.m); for (int i = 0; i < b.m; ++i) vv[i] = cw[a.idx[i]] * phi[(size_t)k * ng + a.idx[i]];
          arrow_fill(a.S, a.t, vv.data(), wr, b.m);
          Small sa(b.m + 1); vec L, Li;
          spd_inv(sa, a.X, a.S, L, Li);
          for (double& v : a.X) v *= x0;
          N += b.m + 1.0;
        }
        arrows.push_back(std::move(a));
      }
    }
  }
  vec P((size_t)p * nxx, 0.0), S1((size_t)p * nn, 0.0), S2((size_t)p * nn), X1((size_t)p * nn, 0.0), X2;
  for (int k = 0; k < p; ++k) for (int i = 0; i < n; ++i) { S1[((size_t)k * n + i) * n + i] = 1.0; X1[((size_t)k * n + i) * n + i] = x0; }
  for (int k = 0; k < p; ++k) for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j)
    S2[((size_t)k * n + i) * n + j] = (i == j ? tau : 0.0) - alpha * Hb[((size_t)k * n + i) * n + j];
  X2 = X1;
  const size_t PN = (size_t)p * nn, bs = (size_t)db * db, PG = (size_t)p * std::max(ng, 1);
  vec M(PN), Rd1(PN), Rd2(PN), Y(PN), rP((size_t)p * nxx), L1(PN), L2(PN), L1i(PN), L2i(PN), S1i(PN), S2i(PN), LX1i(PN), LX2i(PN);
  vec Psi(PN), PhiH(PN), Phi2(PN), T1(PN), T2(PN), G(PN), dM(PN), dS1(PN), dS2(PN), dX1(PN), dX2(PN), c1(PN), c2(PN), dP((size_t)p * nxx);
  vec adjb((size_t)p * nxx), U((size_t)2 * p * db), TU((size_t)2 * p * db), zs((size_t)p * db), t0(nn), t1(nn), t2(nn);
  vec Kx(nxx), Ks(nxx), Fx(nxx), Fs(nxx), tx((size_t)nx * n), Xxx(nxx), Sxx(nxx);
  vec W(PG * nn), Bpp((size_t)std::max(ng, 1) * std::max(ng, 1)), dphi(PG, 0.0), dz(PG, 0.0), corrp(PG, 0.0), rph(PG), wv(nxx), wsv(d), wsv2(d);
  pr.D.resize((size_t)p * bs); pr.Csub.resize((size_t)p * bs);
  double mu_t = -1.0, mu = 0, mu0 = 0, pinf = 0, dinf = 0, stepn = 1e300, prev_stepn = -1.0;
  int phase = 0, ncent = 0, njam = 0, nshiftrun = 0, nbackoff = 0, it = 0;
  const int MUT_BACKOFF_MAX = 10;
  enum { ST_MAXIT, ST_OPT, ST_INACC, ST_DIV } ipm = ST_MAXIT;
  auto dot = [&](const vec& a, const vec& b) { double v = 0; for (size_t e = 0; e < PN; ++e) v += a[e] * b[e]; return v; };
  auto dotn = [&](const double* a, const double* b) { double v = 0; for (int e = 0; e < nn; ++e) v += a[e] * b[e]; return v; };
  auto trace_sum = [&](const vec& a) { double v = 0; for (int k = 0; k < p; ++k) for (int i = 0; i < n; ++i) v += a[((size_t)k * n + i) * n + i]; return v; };
  auto max_step = [&](const vec& Li, const vec& dXv) {
    double lm = 1e300;
    for (int k = 0; k < p; ++k) {
      sm.mm(t0.data(), &Li[(size_t)k * nn], &dXv[(size_t)k * nn]);
      for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double v = 0; for (int r = 0; r < n; ++r) v += t0[i * n + r] * Li[(size_t)k * nn + j * n + r]; t1[i * n + j] = v; }
      lm = std::min(lm, sm.min_eig(t1.data(), work.data()));
    }
    return lm >= 0.0 ? 1e300 : -1.0 / lm;
  };
  auto max_step_small = [&](const vec& Xm, const vec& dXm, int m1) {      // largest theta with Xm + theta dXm >= 0 (one small matrix)
    Small sa(m1); vec L((size_t)m1 * m1), Li((size_t)m1 * m1), a0((size_t)m1 * m1), a1((size_t)m1 * m1), wk((size_t)m1 * m1 + 4 * m1 + 64);
    if (!sa.chol(L.data(), Xm.data())) return 0.0;
    sa.tri_inv(Li.data(), L.data());
    sa.mm(a0.data(), Li.data(), dXm.data());
    for (int i = 0; i < m1; ++i) for (int j = 0; j < m1; ++j) { double v = 0; for (int r = 0; r < m1; ++r) v += a0[i * m1 + r] * Li[j * m1 + r]; a1[i * m1 + j] = v; }
    const double lm = sa.min_eig(a1.data(), wk.data());
    return lm >= 0.0 ? 1e300 : -1.0 / lm;
  };
  // M += sum_i v[k][i] GG[k][i]
  auto add_gg = [&](vec& Mv, const vec& v) {
    for (int k = 0; k < p; ++k) for (int i = 0; i < ng; ++i) {
      const double c = v[(size_t)k * ng + i];
      if (c == 0.0 || !mask[(size_t)k * ng + i]) continue;
      double* Mk = &Mv[(size_t)k * nn];
      if (i < nJ) { const double* g = gg(k, i); for (int e = 0; e < nn; ++e) Mk[e] += c * g[e]; }
      else { const int q = i - nJ; Mk[ta[q] * n + tb[q]] += c; if (ta[q] != tb[q]) Mk[tb[q] * n + ta[q]] += c; }
    }
  };
  // ---- tight mode (convexify_oracle.sdp_step1(tight=True) with rows; Steps 1 / 2): block linear algebra in double-double below DD_SWITCH, dd dual-Newton polish
  bool dd_on = false, polish = false; int ndd = 0;
  DdSys dsys; dsys.par = par;
  std::vector<ddk::mat> qX1, qZ1, qX2, qZ2;
  auto to_dd = [&](std::vector<ddk::mat>& out, const vec& src) { out.resize(p); for (int k = 0; k < p; ++k) { out[k].resize(nn); for (int e = 0; e < nn; ++e) out[k][e] = ddk::from(src[(size_t)k * nn + e]); } };
  // the rows of the stage-local variables in the augmented dd blocks (after assemble_dd filled the P part): Xa / Za, Xb / Zb the dd matrices (X_r, S_r^-1) of the two
  // LMI blocks per stage, aX / aSi those of the arrow blocks (per arrow), zphi [p][ng] = z / phi (loop) or mu / phi^2 (polish)
  auto fill_locals_dd = [&](const std::vector<ddk::mat>& Xa, const std::vector<ddk::mat>& Za, const std::vector<ddk::mat>& Xb, const std::vector<ddk::mat>& Zb,
                            const std::vector<ddk::mat>& aX, const std::vector<ddk::mat>& aSi, const std::vector<ddk::dd>& zphi) {
    using namespace ddk;
    const size_t bsd = (size_t)db * db;
    auto accD = [&](int blk, int r, int c, dd v) { const size_t e = blk * bsd + (size_t)r * db + c; const dd t = add(dd{dsys.Dh[e], dsys.Dl[e]}, v); dsys.Dh[e] = t.h; dsys.Dl[e] = t.l; };
    auto setD = [&](int blk, int r, int c, dd v) { const size_t e = blk * bsd + (size_t)r * db + c; dsys.Dh[e] = v.h; dsys.Dl[e] = v.l; };
    auto accC = [&](int blk, int r, int c, dd v) { const size_t e = blk * bsd + (size_t)r * db + c; const dd t = add(dd{dsys.Ch[e], dsys.Cl[e]}, v); dsys.Ch[e] = t.h; dsys.Cl[e] = t.l; };
    std::vector<std::vector<dd>> wv((size_t)ng * 4, std::vector<dd>(n)), vw((size_t)ng * 4, std::vector<dd>(nx));      // per row: X_a g, Z_a g, X_b g, Z_b g and their images under V
    std::vector<dd> Bl((size_t)ng * ng);
    for (int k = 0; k < p; ++k) {
      const int kn = (k + 1) % p; const double* v = pr.Vk(k);
      for (int i = 0; i < nz; ++i) setD(kn, d + i, d + i, from(1.0));                       // padding rows / unused arrow slots
      for (int i = 0; i < nJ; ++i) {
        if (!mask[(size_t)k * ng + i]) continue;
        const double* g = ci.J + ((size_t)k * nJin + i) * n;
        const mat* Ms[4] = {&Xa[k], &Za[k], &Xb[k], &Zb[k]};
        for (int r = 0; r < 4; ++r) {
          for (int a = 0; a < n; ++a) { dd sacc{0.0, 0.0}; for (int q = 0; q < n; ++q) sacc = add(sacc, muld((*Ms[r])[(size_t)a * n + q], g[q])); wv[(size_t)i * 4 + r][a] = sacc; }
          for (int a = 0; a < nx; ++a) { dd sacc{0.0, 0.0}; for (int q = 0; q < n; ++q) sacc = add(sacc, muld(wv[(size_t)i * 4 + r][q], v[a * n + q])); vw[(size_t)i * 4 + r][a] = sacc; }
        }
        const int col = d + i;
        for (int e = 0; e < d; ++e) {
          const int a = pr.ia[e], c = pr.ib[e];
          dd cV{0.0, 0.0}, cE{0.0, 0.0};
          for (int r = 0; r < 4; r += 2) {
            const std::vector<dd>& Vw = vw[(size_t)i * 4 + r]; const std::vector<dd>& Vu = vw[(size_t)i * 4 + r + 1];
            const std::vector<dd>& w_ = wv[(size_t)i * 4 + r]; const std::vector<dd>& u_ = wv[(size_t)i * 4 + r + 1];
            if (a == c) { cV = add(cV, mul(Vw[a], Vu[a])); cE = add(cE, mul(w_[a], u_[a])); }
            else { cV = add(cV, add(mul(Vw[a], Vu[c]), mul(Vu[a], Vw[c]))); cE = add(cE, add(mul(w_[a], u_[c]), mul(u_[a], w_[c]))); }
          }
          accD(kn, e, col, cV); accD(kn, col, e, cV);
          accC(k, e, col, neg(cE));                                                          // C_k = T[block k, block k+1]: rows of block k
        }
      }
      // T_loc,loc: <g_a g_a', Phi(g_b g_b')> = sum_r (g_a' X_r g_b)(g_a' Z_r g_b), symmetrised, + z / phi on the diagonal
      for (int a = 0; a < nJ; ++a) for (int b = 0; b < nJ; ++b) {
        dd val{0.0, 0.0};
        if (mask[(size_t)k * ng + a] && mask[(size_t)k * ng + b]) {
          const double* ga = ci.J + ((size_t)k * nJin + a) * n;
          for (int r = 0; r < 4; r += 2) val = add(val, mul(dd_dot(wv[(size_t)b * 4 + r], ga, n), dd_dot(wv[(size_t)b * 4 + r + 1], ga, n)));
        }
        Bl[(size_t)a * ng + b] = val;
      }
      for (int a = 0; a < nJ; ++a) for (int b = 0; b < nJ; ++b) {
        if (!(mask[(size_t)k * ng + a] && mask[(size_t)k * ng + b])) continue;
        dd val = muld(add(Bl[(size_t)a * ng + b], Bl[(size_t)b * ng + a]), 0.5);
        if (a == b) val = add(val, zphi[(size_t)k * ng + a]);
        setD(kn, d + a, d + b, val);
      }
    }
    for (size_t e_ = 0; e_ < arrows.size(); ++e_) {
      const Arrow& a = arrows[e_];
      const int kn = (a.k + 1) % p, te = d + ng + a.slot, m1 = a.m + 1;
      const mat& X = aX[e_]; const mat& Si = aSi[e_];
      mat T; mm(T, X, Si, m1, m1, m1, false);
      auto Pe = [&](int i, int j) { return muld(add(T[(size_t)i * m1 + j], T[(size_t)j * m1 + i]), 0.5); };
      dd trP{0.0, 0.0}; for (int i = 0; i < m1; ++i) trP = add(trP, T[(size_t)i * m1 + i]);
      setD(kn, te, te, trP);
      for (int q = 0; q < a.m; ++q) {
        const double wq = wr * cw[a.idx[q]];
        const dd v = muld(Pe(0, q + 1), 2.0 * wq);
        setD(kn, d + a.idx[q], te, v); setD(kn, te, d + a.idx[q], v);
        for (int r = 0; r < a.m; ++r) {
          // Fq[0][r+1] of sym(X Eq Si), Eq = wq (e_0 e_{q+1}' + e_{q+1} e_0')
          dd f = add(add(mul(X[0], Si[(size_t)(q + 1) * m1 + r + 1]), mul(X[q + 1], Si[r + 1])), add(mul(X[(size_t)(r + 1) * m1], Si[(size_t)(q + 1) * m1]), mul(X[(size_t)(r + 1) * m1 + q + 1], Si[0])));
          f = muld(f, 0.5 * wq * 2.0 * wr * cw[a.idx[r]]);
          accD(kn, d + a.idx[r], d + a.idx[q], f);
        }
      }
    }
  };
  // arrow matrices (dd) of the current (t, phi): the slack S^-1 in closed form, and X as given (fp64 iterate) or mu S^-1
  auto arrow_mats = [&](const vec& phi_, const std::vector<double>& tt_, std::vector<ddk::mat>& aSi) -> bool {
    aSi.resize(arrows.size());
    for (size_t e_ = 0; e_ < arrows.size(); ++e_) {
      const Arrow& a = arrows[e_];
      std::vector<ddk::dd> u(a.m);
      for (int i = 0; i < a.m; ++i) u[i] = ddk::tp(wr * cw[a.idx[i]], phi_[(size_t)a.k * ng + a.idx[i]]);
      if (!arrow_inv_dd(aSi[e_], tt_[e_], u, a.m)) return false;
    }
    return true;
  };
  double dtau = 0, dalpha = 0, ds0 = 0, dx0 = 0;
  for (it = 0; it < max_iter + center_iter * (MUT_BACKOFF_MAX + 1) + 1; ++it) {
    pr.calH(M, P, alpha);
    add_gg(M, phi);
    double rd2 = 0, s2 = 0;
    for (int k = 0; k < p; ++k) for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
      const size_t e = ((size_t)k * n + i) * n + j; const double dg = (i == j) ? 1.0 : 0.0;
      Rd1[e] = (M[e] - dg) - S1[e]; Rd2[e] = (tau * dg - M[e]) - S2[e]; Y[e] = X1[e] - X2[e];
      rd2 += Rd1[e] * Rd1[e] + Rd2[e] * Rd2[e]; s2 += S1[e] * S1[e] + S2[e] * S2[e];
    }
    const double rd0 = (alpha - ALPHA_MIN) - s0;
    double comp = dot(X1, S1) + dot(X2, S2) + x0 * s0;
    for (size_t e = 0; e < PG && ng; ++e) comp += phi[e] * z[e];
    for (Arrow& a : arrows) {
      if (a.soc) { a.s.resize(a.m + 1); a.s[0] = a.t; for (int i = 0; i < a.m; ++i) a.s[i + 1] = wr * cw[a.idx[i]] * phi[(size_t)a.k * ng + a.idx[i]]; comp += vdot(a.x, a.s); }
      else { vec vv(a.m); for (int i = 0; i < a.m; ++i) vv[i] = cw[a.idx[i]] * phi[(size_t)a.k * ng + a.idx[i]]; arrow_fill(a.S, a.t, vv.data(), wr, a.m); comp += vdot(a.X, a.S); }
    }
    mu = comp / N;
    const double r_tau = 1.0 - trace_sum(X2), r_alpha = -dot(Hb, Y) - x0;
    pr.adj(rP, Y);
    double rp2 = 0;
    for (int k = 0; k < p; ++k) for (int e = 0; e < d; ++e) { const double v = -rP[(size_t)k * nxx + pr.ia[e] * nx + pr.ib[e]] * (pr.ia[e] == pr.ib[e] ? 1.0 : 2.0); rp2 += v * v; }
    double r_phi2 = 0;
    if (ng) {
      for (int k = 0; k < p; ++k) for (int i = 0; i < ng; ++i) rph[(size_t)k * ng + i] = mask[(size_t)k * ng + i] ? -dotn(gg(k, i), &Y[(size_t)k * nn]) - z[(size_t)k * ng + i] : 0.0;
      for (const Arrow& a : arrows) {
        if (a.soc) { for (int i = 0; i < a.m; ++i) rph[(size_t)a.k * ng + a.idx[i]] -= wr * cw[a.idx[i]] * a.x[i + 1]; r_phi2 += (1.0 - a.x[0]) * (1.0 - a.x[0]); }
        else { double tr = 0; for (int i = 0; i <= a.m; ++i) tr += a.X[(size_t)i * (a.m + 1) + i]; for (int i = 0; i < a.m; ++i) rph[(size_t)a.k * ng + a.idx[i]] -= 2.0 * wr * cw[a.idx[i]] * a.X[i + 1]; r_phi2 += (1.0 - tr) * (1.0 - tr); }
      }
      for (size_t e = 0; e < PG; ++e) if (mask[e]) r_phi2 += rph[e] * rph[e];
    }
    pinf = sqrt(r_tau * r_tau + r_alpha * r_alpha + rp2 + r_phi2) / 2.0;
    dinf = sqrt(rd2 + rd0 * rd0) / (1.0 + sqrt(s2));
    const double relgap = N * mu / std::max(1.0, fabs(tau));
    if (it == 0) mu0 = mu;
    if (!(mu > 0.0) || !std::isfinite(mu) || !std::isfinite(tau) || mu > 1e6 * mu0) { ipm = ST_DIV; break; }
    if (mu_t < 0.0 && relgap < 1e-2 && dinf < 1e-2) mu_t = exp2(rint(log2(tol * std::max(1.0, fabs(tau)))));
    if (phase == 0 && mu_t > 0.0 && mu <= 2.0 * mu_t && dinf < 1e-6 && (pinf < 1e-3 || nshiftrun >= 1)) phase = 1;
    if (phase == 0 && nshiftrun >= 2) {
      const int kb = std::max(0, (int)ceil(log2(mu / mu_t)));
      if (dinf < 1e-6 && nbackoff + kb <= MUT_BACKOFF_MAX) { mu_t = ldexp(mu_t, kb); nbackoff += kb; phase = 1; ncent = 0; prev_stepn = -1.0; nshiftrun = 0; }
      else { ipm = ST_INACC; break; }
    }
    if (phase == 0 && it >= max_iter) break;
    bool ok = true;
    for (int k = 0; k < p && ok; ++k) {
      const size_t o = (size_t)k * nn;
      ok = sm.chol(&L1[o], &S1[o]) && sm.chol(&L2[o], &S2[o]) && sm.chol(t0.data(), &X1[o]);
      if (!ok) break;
      sm.tri_inv(&LX1i[o], t0.data());
      ok = sm.chol(t0.data(), &X2[o]);
      if (!ok) break;
      sm.tri_inv(&LX2i[o], t0.data());
      sm.tri_inv(&L1i[o], &L1[o]); sm.tri_inv(&L2i[o], &L2[o]);
      for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
        double v1 = 0, v2 = 0;
        for (int r = 0; r < n; ++r) { v1 += L1i[o + r * n + i] * L1i[o + r * n + j]; v2 += L2i[o + r * n + i] * L2i[o + r * n + j]; }
        S1i[o + i * n + j] = v1; S2i[o + i * n + j] = v2;
      }
    }
    if (!ok) { ipm = ST_DIV; break; }
    // ---- Schur complement: the P part of every block (as solve_problem), leading dimension db
    std::fill(pr.D.begin(), pr.D.end(), 0.0); std::fill(pr.Csub.begin(), pr.Csub.end(), 0.0);
    for (int r = 0; r < 2; ++r) {
      const vec& X = r ? X2 : X1; const vec& Si = r ? S2i : S1i;
      for (int k = 0; k < p; ++k) {
        const size_t o = (size_t)k * nn; const double* v = pr.Vk(k); const int kn = (k + 1) % p;
        auto VZVt = [&](double* out, double* fz, const double* Z) {
          for (int a = 0; a < nx; ++a) for (int c = 0; c < n; ++c) { double sacc = 0; for (int q = 0; q < n; ++q) sacc += v[a * n + q] * Z[q * n + c]; tx[a * n + c] = sacc; }
          for (int a = 0; a < nx; ++a) for (int c = 0; c < nx; ++c) { double sacc = 0; for (int q = 0; q < n; ++q) sacc += tx[a * n + q] * v[c * n + q]; out[a * nx + c] = sacc; }
          for (int a = 0; a < nx; ++a) for (int c = 0; c < nx; ++c) { double sacc = 0; for (int q = 0; q < n; ++q) sacc += Z[a * n + q] * v[c * n + q]; fz[a * nx + c] = sacc; }
        };
        VZVt(Kx.data(), Fx.data(), &X[o]); VZVt(Ks.data(), Fs.data(), &Si[o]);
        for (int a = 0; a < nx; ++a) for (int c = 0; c < nx; ++c) { Xxx[a * nx + c] = X[o + a * n + c]; Sxx[a * nx + c] = Si[o + a * n + c]; }
        pr.add_T(&pr.D[k * bs], Xxx.data(), Sxx.data(), 1.0, false);
        pr.add_T(&pr.D[kn * bs], Kx.data(), Ks.data(), 1.0, false);
        pr.add_T(&pr.Csub[k * bs], Fx.data(), Fs.data(), -1.0, k < p - 1);
      }
    }
    double b_tt = 0, b_ta = 0, b_aa = 0;
    for (int k = 0; k < p; ++k) {
      const size_t o = (size_t)k * nn;
      sm.mm(t0.data(), &X2[o], &S2i[o]);
      for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Psi[o + i * n + j] = 0.5 * (t0[i * n + j] + t0[j * n + i]);
      sm.sym3(&PhiH[o], &X1[o], &Hb[o], &S1i[o], t0.data(), t1.data());
      sm.sym3(&Phi2[o], &X2[o], &Hb[o], &S2i[o], t0.data(), t1.data());
      for (int e = 0; e < nn; ++e) PhiH[o + e] += Phi2[o + e];
    }
    b_tt = trace_sum(Psi); b_ta = -trace_sum(Phi2); b_aa = dot(Hb, PhiH) + x0 / s0;
    std::fill(U.begin(), U.end(), 0.0);
    pr.adj(adjb, Psi);
    for (int k = 0; k < p; ++k) { pr.svec_grad(&U[(size_t)k * db], &adjb[(size_t)k * nxx]); for (int e = 0; e < d; ++e) U[(size_t)k * db + e] = -U[(size_t)k * db + e]; }
    pr.adj(adjb, PhiH);
    for (int k = 0; k < p; ++k) pr.svec_grad(&U[(size_t)(p + k) * db], &adjb[(size_t)k * nxx]);
    // ---- the stage-local rows: locals of stage k at rows d .. d+nz of block kn = k+1
    for (int k = 0; k < p; ++k) {
      const size_t o = (size_t)k * nn; const int kn = (k + 1) % p; const double* v = pr.Vk(k);
      double* Dn = &pr.D[kn * bs];
      for (int i = 0; i < nz; ++i) Dn[(size_t)(d + i) * db + d + i] = 1.0;                     // padding / unused arrow slots: identity rows
      for (int i = 0; i < ng; ++i) {
        double* Wi = &W[((size_t)k * ng + i) * nn];
        if (!mask[(size_t)k * ng + i]) { std::fill(Wi, Wi + nn, 0.0); continue; }
        const double* g = gg(k, i);
        sm.sym3(Wi, &X1[o], g, &S1i[o], t0.data(), t1.data());
        sm.sym3(t2.data(), &X2[o], g, &S2i[o], t0.data(), t1.data());
        for (int e = 0; e < nn; ++e) Wi[e] += t2[e];
        // column of the multiplier: -svec(W[:nx,:nx]) at P_k, svec(V W V') at P_{k+1}
        for (int a = 0; a < nx; ++a) for (int c = 0; c < n; ++c) { double sacc = 0; for (int q = 0; q < n; ++q) sacc += v[a * n + q] * Wi[q * n + c]; tx[a * n + c] = sacc; }
        for (int a = 0; a < nx; ++a) for (int c = 0; c < nx; ++c) { double sacc = 0; for (int q = 0; q < n; ++q) sacc += tx[a * n + q] * v[c * n + q]; wv[a * nx + c] = sacc; }
        pr.svec_grad(wsv.data(), wv.data());
        for (int a = 0; a < nx; ++a) for (int c = 0; c < nx; ++c) wv[a * nx + c] = Wi[a * n + c];
        pr.svec_grad(wsv2.data(), wv.data());
        const int col = d + i;
        for (int e = 0; e < d; ++e) {
          Dn[(size_t)col * db + e] += wsv[e]; Dn[(size_t)e * db + col] += wsv[e];                 // diagonal block k+1: (P_{k+1}, phi_ki), both triangles
          // coupling block C_k (rows block k, columns block k+1), entry (e, col) = -wsv2[e]: stored transposed for k < p-1 (block [k+1][k]), as is for k = p-1
          if (k < p - 1) pr.Csub[k * bs + (size_t)e * db + col] += -wsv2[e]; else pr.Csub[k * bs + (size_t)col * db + e] += -wsv2[e];
        }
        U[(size_t)kn * db + col] = -dotn(g, &Psi[o]);                                          // c_tau
        U[(size_t)(p + kn) * db + col] = dotn(g, &PhiH[o]);                                    // c_alpha
      }
      if (ng) {
        // Bpp[v][w] = <GG_v, W_w>, symmetrised, + diag(z / phi)
        for (int a = 0; a < ng; ++a) for (int b = 0; b < ng; ++b)
          Bpp[(size_t)a * ng + b] = (mask[(size_t)k * ng + a] && mask[(size_t)k * ng + b]) ? dotn(gg(k, a), &W[((size_t)k * ng + b) * nn]) : 0.0;
        for (int a = 0; a < ng; ++a) for (int b = 0; b < ng; ++b) {
          double val = 0.5 * (Bpp[(size_t)a * ng + b] + Bpp[(size_t)b * ng + a]);
          if (a == b) val += mask[(size_t)k * ng + a] ? z[(size_t)k * ng + a] / phi[(size_t)k * ng + a] : 1.0;
          Dn[(size_t)(d + b) * db + d + a] = val;
        }
      }
    }
    for (Arrow& a : arrows) {
      const int kn = (a.k + 1) % p, te = d + ng + a.slot, m1 = a.m + 1;
      double* Dn = &pr.D[kn * bs];
      auto at = [&](int r, int c) -> double& { return Dn[(size_t)c * db + r]; };
      if (a.soc) {
        soc_scaling(a.s, a.x, a.beta, a.v);
        a.lam = soc_W(a.beta, a.v, a.x, false);
        soc_W2inv(a.beta, a.v, a.W2);
        at(te, te) = a.W2[0];
        for (int q = 0; q < a.m; ++q) { const double wq = wr * cw[a.idx[q]]; at(d + a.idx[q], te) = wq * a.W2[q + 1]; at(te, d + a.idx[q]) = wq * a.W2[q + 1]; }
        for (int q = 0; q < a.m; ++q) for (int r = 0; r < a.m; ++r) at(d + a.idx[q], d + a.idx[r]) += wr * cw[a.idx[q]] * wr * cw[a.idx[r]] * a.W2[(size_t)(q + 1) * m1 + r + 1];
        continue;
      }
      Small sa(m1); vec L, Li;
      if (!spd_inv(sa, a.Si, a.S, L, Li)) { ok = false; break; }
      a.Pe.resize((size_t)m1 * m1); vec tmp((size_t)m1 * m1), tmp2((size_t)m1 * m1), Eq((size_t)m1 * m1), Fq((size_t)m1 * m1);
      sa.mm(tmp.data(), a.X.data(), a.Si.data());
      double trP = 0;
      for (int i = 0; i < m1; ++i) for (int j = 0; j < m1; ++j) a.Pe[i * m1 + j] = 0.5 * (tmp[i * m1 + j] + tmp[j * m1 + i]);
      for (int i = 0; i < m1; ++i) trP += a.Pe[i * m1 + i];
      at(te, te) = trP;
      for (int q = 0; q < a.m; ++q) {
        const double wq = wr * cw[a.idx[q]];
        at(d + a.idx[q], te) = 2.0 * wq * a.Pe[q + 1]; at(te, d + a.idx[q]) = 2.0 * wq * a.Pe[q + 1];
        std::fill(Eq.begin(), Eq.end(), 0.0); Eq[q + 1] = wq; Eq[(size_t)(q + 1) * m1] = wq;
        sa.sym3(Fq.data(), a.X.data(), Eq.data(), a.Si.data(), tmp.data(), tmp2.data());
        for (int r = 0; r < a.m; ++r) at(d + a.idx[r], d + a.idx[q]) += 2.0 * wr * cw[a.idx[r]] * Fq[r + 1];
      }
    }
    if (!ok) { ipm = ST_DIV; break; }
    if (tight && !dd_on && mu <= DD_SWITCH * std::max(1.0, fabs(tau))) dd_on = true;
    if (!dd_on) {
      if (!pr.factor()) { ipm = ST_INACC; break; }
      if (tight && pr.shift > 0.0) dd_on = true;
    }
    if (dd_on) {
      ++ndd;
      to_dd(qX1, X1); to_dd(qZ1, S1i); to_dd(qX2, X2); to_dd(qZ2, S2i);
      assemble_dd(dsys, pr, qX1, qZ1, qX2, qZ2);
      std::vector<ddk::mat> aXd(arrows.size()), aSid;
      std::vector<double> tts(arrows.size());
      for (size_t e_ = 0; e_ < arrows.size(); ++e_) { tts[e_] = arrows[e_].t; aXd[e_].resize(arrows[e_].X.size()); for (size_t q = 0; q < arrows[e_].X.size(); ++q) aXd[e_][q] = ddk::from(arrows[e_].X[q]); }
      std::vector<ddk::dd> zphi(PG, ddk::dd{0.0, 0.0});
      for (size_t e = 0; e < PG && ng; ++e) if (mask[e]) zphi[e] = ddk::div(ddk::from(z[e]), ddk::from(phi[e]));
      if (!arrow_mats(phi, tts, aSid)) { ipm = ST_INACC; break; }
      fill_locals_dd(qX1, qZ1, qX2, qZ2, aXd, aSid, zphi);
      if (!dsys.factor()) { ipm = ST_INACC; break; }
      pr.shift = 0.0;
    }
    auto bsolve = [&](double* R, int nrhs) { if (dd_on) dsys.solve(R, nrhs); else pr.solve(R, nrhs); };
    nshiftrun = pr.shift > 0.0 ? nshiftrun + 1 : 0;
    if (phase == 1 && pr.shift > 0.0 && nbackoff < MUT_BACKOFF_MAX) { mu_t *= 2.0; ++nbackoff; ncent = 0; prev_stepn = -1.0; nshiftrun = 0; }
    else if ((phase == 1 && pr.shift > 0.0) || (nshiftrun >= 2 && (mu_t < 0.0 || nbackoff >= MUT_BACKOFF_MAX))) { ipm = ST_INACC; break; }
    TU = U; bsolve(TU.data(), 2);
    double sb00 = b_tt, sb01 = b_ta, sb11 = b_aa;
    for (size_t e = 0; e < (size_t)p * db; ++e) { sb00 -= U[e] * TU[e]; sb01 -= U[e] * TU[(size_t)p * db + e]; sb11 -= U[(size_t)p * db + e] * TU[(size_t)p * db + e]; }
    auto direction = [&](double sig, bool corr, double corr0) {
      for (int k = 0; k < p; ++k) {
        const size_t o = (size_t)k * nn;
        sm.sym3(t2.data(), &X1[o], &Rd1[o], &S1i[o], t0.data(), t1.data());
        for (int e = 0; e < nn; ++e) T1[o + e] = sig * S1i[o + e] - t2[e] - (corr ? c1[o + e] : 0.0);
        sm.sym3(t2.data(), &X2[o], &Rd2[o], &S2i[o], t0.data(), t1.data());
        for (int e = 0; e < nn; ++e) { T2[o + e] = sig * S2i[o + e] - t2[e] - (corr ? c2[o + e] : 0.0); G[o + e] = T1[o + e] - T2[o + e]; }
      }
      const double t0s = sig / s0 - x0 * rd0 / s0 - corr0;
      const double rhs_tau = trace_sum(T2) - 1.0, rhs_alpha = dot(Hb, G) + t0s;
      pr.adj(adjb, G);
      std::fill(zs.begin(), zs.end(), 0.0);
      for (int k = 0; k < p; ++k) pr.svec_grad(&zs[(size_t)k * db], &adjb[(size_t)k * nxx]);
      for (int k = 0; k < p; ++k) for (int i = 0; i < ng; ++i) {
        const size_t e = (size_t)k * ng + i;
        rph[e] = mask[e] ? dotn(gg(k, i), &G[(size_t)k * nn]) + sig / phi[e] - (corr ? corrp[e] : 0.0) : 0.0;
      }
      for (Arrow& a : arrows) {
        const int kn = (a.k + 1) % p, m1 = a.m + 1;
        double rt;
        if (a.soc) {
          a.g = soc_inv(a.s);
          for (int i = 0; i < m1; ++i) a.g[i] = sig * a.g[i] - a.x[i] - (corr ? a.corrv[i] : 0.0);
          for (int i = 0; i < a.m; ++i) rph[(size_t)a.k * ng + a.idx[i]] += wr * cw[a.idx[i]] * (a.g[i + 1] + a.x[i + 1]);
          rt = (a.g[0] + a.x[0]) - 1.0;
        } else {
          a.Tm.resize((size_t)m1 * m1);
          for (int e = 0; e < m1 * m1; ++e) a.Tm[e] = sig * a.Si[e] - (corr ? a.corr[e] : 0.0);
          for (int i = 0; i < a.m; ++i) rph[(size_t)a.k * ng + a.idx[i]] += 2.0 * wr * cw[a.idx[i]] * a.Tm[i + 1];
          rt = -1.0; for (int i = 0; i < m1; ++i) rt += a.Tm[(size_t)i * m1 + i];
        }
        zs[(size_t)kn * db + d + ng + a.slot] = rt;
      }
      for (int k = 0; k < p; ++k) for (int i = 0; i < ng; ++i) zs[(size_t)((k + 1) % p) * db + d + i] = mask[(size_t)k * ng + i] ? rph[(size_t)k * ng + i] : 0.0;
      bsolve(zs.data(), 1);
      double u0 = 0, u1 = 0;
      for (size_t e = 0; e < (size_t)p * db; ++e) { u0 += U[e] * zs[e]; u1 += U[(size_t)p * db + e] * zs[e]; }
      const double rb0 = rhs_tau - u0, rb1 = rhs_alpha - u1, det = sb00 * sb11 - sb01 * sb01;
      dtau = (sb11 * rb0 - sb01 * rb1) / det; dalpha = (sb00 * rb1 - sb01 * rb0) / det;
      for (size_t e = 0; e < (size_t)p * db; ++e) zs[e] -= TU[e] * dtau + TU[(size_t)p * db + e] * dalpha;
      for (int k = 0; k < p; ++k) for (int e = 0; e < d; ++e) {
        const double v = zs[(size_t)k * db + e];
        dP[(size_t)k * nxx + pr.ia[e] * nx + pr.ib[e]] = v; dP[(size_t)k * nxx + pr.ib[e] * nx + pr.ia[e]] = v;
      }
      for (int k = 0; k < p; ++k) for (int i = 0; i < ng; ++i) dphi[(size_t)k * ng + i] = mask[(size_t)k * ng + i] ? zs[(size_t)((k + 1) % p) * db + d + i] : 0.0;
      pr.calH(dM, dP, dalpha);
      add_gg(dM, dphi);
      for (Arrow& a : arrows) {
        const int kn = (a.k + 1) % p, m1 = a.m + 1;
        a.dt = zs[(size_t)kn * db + d + ng + a.slot];
        if (a.soc) {
          a.ds.resize(m1); a.ds[0] = a.dt;
          for (int i = 0; i < a.m; ++i) a.ds[i + 1] = wr * cw[a.idx[i]] * dphi[(size_t)a.k * ng + a.idx[i]];
          a.dx.resize(m1);
          for (int i = 0; i < m1; ++i) { double v = 0; for (int j = 0; j < m1; ++j) v += a.W2[(size_t)i * m1 + j] * a.ds[j]; a.dx[i] = a.g[i] - v; }
        } else {
          vec vv(a.m); for (int i = 0; i < a.m; ++i) vv[i] = cw[a.idx[i]] * dphi[(size_t)a.k * ng + a.idx[i]];
          arrow_fill(a.dS, a.dt, vv.data(), wr, a.m);
          Small sa(m1); vec tmp((size_t)m1 * m1), tmp2((size_t)m1 * m1), sy((size_t)m1 * m1);
          sa.sym3(sy.data(), a.X.data(), a.dS.data(), a.Si.data(), tmp.data(), tmp2.data());
          a.dX.resize((size_t)m1 * m1);
          for (int e = 0; e < m1 * m1; ++e) a.dX[e] = sig * a.Si[e] - a.X[e] - sy[e] - (corr ? a.corr[e] : 0.0);
        }
      }
      for (int k = 0; k < p; ++k) {
        const size_t o = (size_t)k * nn;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
          const size_t e = o + i * n + j;
          dS1[e] = dM[e] + Rd1[e]; dS2[e] = (i == j ? dtau : 0.0) - dM[e] + Rd2[e];
        }
        sm.sym3(t2.data(), &X1[o], &dS1[o], &S1i[o], t0.data(), t1.data());
        for (int e = 0; e < nn; ++e) dX1[o + e] = sig * S1i[o + e] - X1[o + e] - t2[e] - (corr ? c1[o + e] : 0.0);
        sm.sym3(t2.data(), &X2[o], &dS2[o], &S2i[o], t0.data(), t1.data());
        for (int e = 0; e < nn; ++e) dX2[o + e] = sig * S2i[o + e] - X2[o + e] - t2[e] - (corr ? c2[o + e] : 0.0);
      }
      ds0 = dalpha + rd0;
      dx0 = sig / s0 - x0 - x0 * ds0 / s0 - corr0;
      for (size_t e = 0; e < PG && ng; ++e) dz[e] = mask[e] ? sig / phi[e] - z[e] - z[e] * dphi[e] / phi[e] - (corr ? corrp[e] : 0.0) : 0.0;
    };
    auto steps = [&](double& ap, double& ad) {
      ap = std::min(max_step(LX1i, dX1), max_step(LX2i, dX2));
      ad = std::min(max_step(L1i, dS1), max_step(L2i, dS2));
      if (dx0 < 0) ap = std::min(ap, -x0 / dx0);
      if (ds0 < 0) ad = std::min(ad, -s0 / ds0);
      for (size_t e = 0; e < PG && ng; ++e) {
        if (dz[e] < 0) ap = std::min(ap, -z[e] / dz[e]);
        if (dphi[e] < 0) ad = std::min(ad, -phi[e] / dphi[e]);
      }
      for (const Arrow& a : arrows) {
        if (a.soc) { ap = std::min(ap, soc_max_step(a.x, a.dx)); ad = std::min(ad, soc_max_step(a.s, a.ds)); }
        else { ap = std::min(ap, max_step_small(a.X, a.dX, a.m + 1)); ad = std::min(ad, max_step_small(a.S, a.dS, a.m + 1)); }
      }
    };
    double ap, ad;
    if (phase == 0) {
      direction(0.0, false, 0.0);
      steps(ap, ad);
      ap = std::min(1.0, ap); ad = std::min(1.0, ad);
      double xs_aff = (x0 + ap * dx0) * (s0 + ad * ds0);
      for (size_t e = 0; e < PN; ++e) xs_aff += (X1[e] + ap * dX1[e]) * (S1[e] + ad * dS1[e]) + (X2[e] + ap * dX2[e]) * (S2[e] + ad * dS2[e]);
      for (size_t e = 0; e < PG && ng; ++e) xs_aff += (z[e] + ap * dz[e]) * (phi[e] + ad * dphi[e]);
      for (const Arrow& a : arrows) {
        if (a.soc) for (int i = 0; i <= a.m; ++i) xs_aff += (a.x[i] + ap * a.dx[i]) * (a.s[i] + ad * a.ds[i]);
        else for (size_t e = 0; e < a.X.size(); ++e) xs_aff += (a.X[e] + ap * a.dX[e]) * (a.S[e] + ad * a.dS[e]);
      }
      const double rat = (xs_aff / N) / mu;
      double sigma = std::min(std::max(rat * rat, 1e-6), 1.0);
      double sig_mu = sigma * mu;
      if (mu_t > 0.0) sig_mu = std::max(sig_mu, mu_t);
      for (int k = 0; k < p; ++k) {
        const size_t o = (size_t)k * nn;
        sm.sym3(&c1[o], &dX1[o], &dS1[o], &S1i[o], t0.data(), t1.data());
        sm.sym3(&c2[o], &dX2[o], &dS2[o], &S2i[o], t0.data(), t1.data());
      }
      const double corr0 = dx0 * ds0 / s0;
      for (size_t e = 0; e < PG && ng; ++e) corrp[e] = dz[e] * dphi[e] / phi[e];
      for (Arrow& a : arrows) {
        if (a.soc) a.corrv = soc_W(a.beta, a.v, soc_div(a.lam, soc_prod(soc_W(a.beta, a.v, a.dx, false), soc_W(a.beta, a.v, a.ds, true))), true);
        else { const int m1 = a.m + 1; Small sa(m1); vec tmp((size_t)m1 * m1), tmp2((size_t)m1 * m1); a.corr.resize((size_t)m1 * m1); sa.sym3(a.corr.data(), a.dX.data(), a.dS.data(), a.Si.data(), tmp.data(), tmp2.data()); }
      }
      direction(sig_mu, true, corr0);
      steps(ap, ad);
      const double gam = 0.9 + 0.09 * std::min(std::min(ap, ad), 1.0);
      ap = std::min(1.0, gam * ap); ad = std::min(1.0, gam * ad);
    } else {
      ++ncent;
      direction(mu_t, false, 0.0);
      steps(ap, ad);
      ap = std::min(1.0, 0.95 * ap); ad = std::min(1.0, 0.95 * ad);
      double num = 0, den = 0;
      const double ra = dalpha / alpha;
      for (size_t e = 0; e < PN; ++e) { const double dh = dM[e] - ra * M[e]; num += dh * dh; den += M[e] * M[e]; }
      stepn = sqrt(num / den);
    }
    njam = (ap < 1e-6 && ad < 1e-6) ? njam + 1 : 0;
    if (njam >= 2) { ipm = ST_INACC; break; }
    for (int k = 0; k < p; ++k) for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
      const size_t e = ((size_t)k * n + i) * n + j, et = ((size_t)k * n + j) * n + i;
      const double x1 = 0.5 * ((X1[e] + ap * dX1[e]) + (X1[et] + ap * dX1[et])), x2 = 0.5 * ((X2[e] + ap * dX2[e]) + (X2[et] + ap * dX2[et]));
      const double s1 = 0.5 * ((S1[e] + ad * dS1[e]) + (S1[et] + ad * dS1[et])), s2v = 0.5 * ((S2[e] + ad * dS2[e]) + (S2[et] + ad * dS2[et]));
      X1[e] = X1[et] = x1; X2[e] = X2[et] = x2; S1[e] = S1[et] = s1; S2[e] = S2[et] = s2v;
    }
    x0 += ap * dx0; s0 += ad * ds0; tau += ad * dtau; alpha += ad * dalpha;
    for (size_t e = 0; e < (size_t)p * nxx; ++e) P[e] += ad * dP[e];
    for (size_t e = 0; e < PG && ng; ++e) { z[e] += ap * dz[e]; phi[e] += ad * dphi[e]; }
    for (Arrow& a : arrows) {
      a.t += ad * a.dt;
      if (a.soc) for (int i = 0; i <= a.m; ++i) a.x[i] += ap * a.dx[i];
      else { const int m1 = a.m + 1; for (int i = 0; i < m1; ++i) for (int j = 0; j <= i; ++j) { const double v = 0.5 * ((a.X[i * m1 + j] + ap * a.dX[i * m1 + j]) + (a.X[j * m1 + i] + ap * a.dX[j * m1 + i])); a.X[i * m1 + j] = a.X[j * m1 + i] = v; } }
    }
    if (phase == 1) {
      const bool full = (ap == 1.0 && ad == 1.0);
      const double est = (prev_stepn >= 0.0) ? stepn * pow(std::min(1.0, stepn / prev_stepn), 1.5) : stepn;
      if (tight && full && stepn < POLISH_ENTER) { polish = true; ++it; break; }
      if (full && (stepn < center_tol || est < 0.1 * center_tol)) { ipm = ST_OPT; ++it; break; }
      if (full && prev_stepn >= 0.0 && stepn > 0.5 * prev_stepn && stepn < 1e-6) { ipm = ST_OPT; ++it; break; }
      if (ncent >= center_iter) {
        if (nbackoff < MUT_BACKOFF_MAX) { mu_t *= 2.0; ++nbackoff; ncent = 0; prev_stepn = -1.0; continue; }
        ipm = ST_INACC; ++it; break;
      }
      prev_stepn = full ? stepn : -1.0;
    }
  }
  int npolish = 0;
  if (polish) {
    // dual-Newton polish in double-double (convexify_oracle._polish_dd with GG / arrows): y = (tau, alpha, P, phi, t); every stage quantity from the fp64 y
    using namespace ddk;
    ipm = ST_INACC;
    std::vector<mat> Mq(p), Z1q(p), Z2q(p), X1q(p), X2q(p);
    std::vector<double> tt(arrows.size());
    for (size_t e_ = 0; e_ < arrows.size(); ++e_) tt[e_] = arrows[e_].t;
    auto cones = [&](double tau_, double alpha_, const vec& P_, const vec& phi_) -> bool {
      if (!(alpha_ - ALPHA_MIN > 0.0)) return false;
      for (size_t e = 0; e < PG && ng; ++e) if (mask[e] && !(phi_[e] > 0.0)) return false;
      bool okc = true;
#pragma omp parallel for schedule(dynamic, 1) if (par)
      for (int k = 0; k < p; ++k) {
        const int kn = (k + 1) % p;
        mat Vd((size_t)nx * n), Pn((size_t)nx * nx), t, Mk, S((size_t)nn);
        for (int e = 0; e < nx * n; ++e) Vd[e] = from(pr.Vk(k)[e]);
        for (int e = 0; e < nxx; ++e) Pn[e] = from(P_[(size_t)kn * nxx + e]);
        mat Vt((size_t)n * nx); for (int a = 0; a < nx; ++a) for (int c = 0; c < n; ++c) Vt[(size_t)c * nx + a] = Vd[(size_t)a * n + c];
        mm(t, Vt, Pn, n, nx, nx, false); mm(Mk, t, Vd, n, nx, n, false);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
          dd v = add(Mk[(size_t)i * n + j], muld(from(Hb[(size_t)k * nn + i * n + j]), alpha_));
          if (i < nx && j < nx) v = sub(v, from(P_[(size_t)k * nxx + i * nx + j]));
          Mk[(size_t)i * n + j] = v;
        }
        for (int i2 = 0; i2 < nJ; ++i2) {
          if (!mask[(size_t)k * ng + i2]) continue;
          const double* g = ci.J + ((size_t)k * nJin + i2) * n; const double ph = phi_[(size_t)k * ng + i2];
          for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Mk[(size_t)i * n + j] = add(Mk[(size_t)i * n + j], muld(tp(g[i], g[j]), ph));
        }
        Mq[k] = Mk;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) S[(size_t)i * n + j] = (i == j) ? sub(Mk[(size_t)i * n + j], from(1.0)) : Mk[(size_t)i * n + j];
        if (!inv_spd(Z1q[k], S, n)) { okc = false; continue; }
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) S[(size_t)i * n + j] = (i == j) ? sub(from(tau_), Mk[(size_t)i * n + j]) : neg(Mk[(size_t)i * n + j]);
        if (!inv_spd(Z2q[k], S, n)) okc = false;
      }
      return okc;
    };
    auto adj_sv = [&](double* out, const std::vector<mat>& Gm, double sg) {      // svec of the adjoint, into the P rows of an augmented vector [p][db]
      std::vector<mat> Wm(p);
#pragma omp parallel for schedule(dynamic, 1) if (par)
      for (int k = 0; k < p; ++k) {
        mat Vd((size_t)nx * n), t;
        for (int e = 0; e < nx * n; ++e) Vd[e] = from(pr.Vk(k)[e]);
        mm(t, Vd, Gm[k], nx, n, n, false); mm(Wm[k], t, Vd, nx, n, nx, true);
      }
      for (int j = 0; j < p; ++j) {
        const int jm = (j + p - 1) % p;
        for (int e = 0; e < d; ++e) {
          const int a = pr.ia[e], c = pr.ib[e];
          out[(size_t)j * db + e] = sg * val(sub(Wm[jm][a * nx + c], Gm[j][(size_t)a * n + c])) * (a == c ? 1.0 : 2.0);
        }
      }
    };
    std::vector<mat> aSi;
    bool okp = cones(tau, alpha, P, phi) && arrow_mats(phi, tt, aSi);
    vec Pn_(P.size()), phin(PG, 1.0);
    std::vector<double> ttn(arrows.size());
    for (npolish = 1; okp && npolish <= POLISH_MAX; ++npolish) {
      const double mu_p = mu_t;
      s0 = alpha - ALPHA_MIN; x0 = mu_p / s0;
      std::vector<mat> Psi(p), Ph(p), Ph2(p), Yq(p);
      double b_tt = 0, b_ta = 0, b_aa = 0, g_tau = 1.0, g_alpha = -x0;
      for (int k = 0; k < p; ++k) {
        const size_t o = (size_t)k * nn;
        X1q[k].resize(nn); X2q[k].resize(nn); Yq[k].resize(nn);
        for (int e = 0; e < nn; ++e) { X1q[k][e] = muld(Z1q[k][e], mu_p); X2q[k][e] = muld(Z2q[k][e], mu_p); Yq[k][e] = sub(X1q[k][e], X2q[k][e]); }
        mat Hd(nn), t; for (int e = 0; e < nn; ++e) Hd[e] = from(Hb[o + e]);
        mm(Psi[k], X2q[k], Z2q[k], n, n, n, false);
        mm(t, X2q[k], Hd, n, n, n, false); mm(Ph2[k], t, Z2q[k], n, n, n, false);
        mm(t, X1q[k], Hd, n, n, n, false); mm(Ph[k], t, Z1q[k], n, n, n, false);
        for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) {
          auto sy = [&](mat& Gm) { const dd v = muld(add(Gm[(size_t)i * n + j], Gm[(size_t)j * n + i]), 0.5); Gm[(size_t)i * n + j] = v; Gm[(size_t)j * n + i] = v; };
          sy(Psi[k]); sy(Ph2[k]); sy(Ph[k]);
        }
        for (int e = 0; e < nn; ++e) Ph[k][e] = add(Ph[k][e], Ph2[k][e]);
        dd trp = from(0.0), trh = from(0.0), hp = from(0.0), trx = from(0.0), hy = from(0.0);
        for (int i = 0; i < n; ++i) { trp = add(trp, Psi[k][(size_t)i * n + i]); trh = add(trh, Ph2[k][(size_t)i * n + i]); trx = add(trx, X2q[k][(size_t)i * n + i]); }
        for (int e = 0; e < nn; ++e) { hp = add(hp, mul(Hd[e], Ph[k][e])); hy = add(hy, mul(Hd[e], Yq[k][e])); }
        b_tt += val(trp); b_ta -= val(trh); b_aa += val(hp); g_tau -= val(trx); g_alpha -= val(hy);
      }
      b_aa += x0 / s0;
      assemble_dd(dsys, pr, X1q, Z1q, X2q, Z2q);
      std::vector<mat> aXd(arrows.size());
      for (size_t e_ = 0; e_ < arrows.size(); ++e_) { aXd[e_].resize(aSi[e_].size()); for (size_t q = 0; q < aSi[e_].size(); ++q) aXd[e_][q] = muld(aSi[e_][q], mu_p); }
      std::vector<dd> zphi(PG, dd{0.0, 0.0});
      for (size_t e = 0; e < PG && ng; ++e) if (mask[e]) zphi[e] = div(div(from(mu_p), from(phi[e])), from(phi[e]));
      fill_locals_dd(X1q, Z1q, X2q, Z2q, aXd, aSi, zphi);
      if (!dsys.factor()) break;
      // border columns (tau, alpha) and the right-hand side = minus the gradient, both with their stage-local rows
      std::fill(U.begin(), U.end(), 0.0); std::fill(zs.begin(), zs.end(), 0.0);
      adj_sv(&U[0], Psi, -1.0); adj_sv(&U[(size_t)p * db], Ph, 1.0); adj_sv(zs.data(), Yq, 1.0);
      for (int k = 0; k < p; ++k) {
        const int kn = (k + 1) % p;
        for (int i = 0; i < nJ; ++i) {
          if (!mask[(size_t)k * ng + i]) continue;
          const double* g = ci.J + ((size_t)k * nJin + i) * n;
          dd cps{0.0, 0.0}, cph{0.0, 0.0}, gy{0.0, 0.0};
          for (int a = 0; a < n; ++a) for (int c = 0; c < n; ++c) { const dd gg_ = tp(g[a], g[c]); cps = add(cps, mul(gg_, Psi[k][(size_t)a * n + c])); cph = add(cph, mul(gg_, Ph[k][(size_t)a * n + c])); gy = add(gy, mul(gg_, Yq[k][(size_t)a * n + c])); }
          U[(size_t)kn * db + d + i] = -val(cps); U[(size_t)(p + kn) * db + d + i] = val(cph);
          // minus the gradient in phi: <g g', Y> + mu / phi (+ the arrow term below), evaluated in dd, then rounded
          zs[(size_t)kn * db + d + i] = val(add(gy, div(from(mu_p), from(phi[(size_t)k * ng + i]))));
        }
      }
      for (size_t e_ = 0; e_ < arrows.size(); ++e_) {
        const Arrow& a = arrows[e_];
        const int kn = (a.k + 1) % p, m1 = a.m + 1;
        dd trx{0.0, 0.0}; for (int i = 0; i < m1; ++i) trx = add(trx, aXd[e_][(size_t)i * m1 + i]);
        zs[(size_t)kn * db + d + ng + a.slot] = -(1.0 - val(trx));
        for (int q = 0; q < a.m; ++q) zs[(size_t)kn * db + d + a.idx[q]] += 2.0 * wr * cw[a.idx[q]] * val(aXd[e_][q + 1]);
      }
      TU = U; dsys.solve(TU.data(), 2); dsys.solve(zs.data(), 1);
      double sb00 = b_tt, sb01 = b_ta, sb11 = b_aa, u0 = 0, u1 = 0;
      for (size_t e = 0; e < (size_t)p * db; ++e) {
        sb00 -= U[e] * TU[e]; sb01 -= U[e] * TU[(size_t)p * db + e]; sb11 -= U[(size_t)p * db + e] * TU[(size_t)p * db + e];
        u0 += U[e] * zs[e]; u1 += U[(size_t)p * db + e] * zs[e];
      }
      const double rb0 = -g_tau - u0, rb1 = -g_alpha - u1, det = sb00 * sb11 - sb01 * sb01;
      dtau = (sb11 * rb0 - sb01 * rb1) / det; dalpha = (sb00 * rb1 - sb01 * rb0) / det;
      for (size_t e = 0; e < (size_t)p * db; ++e) zs[e] -= TU[e] * dtau + TU[(size_t)p * db + e] * dalpha;
      for (int k = 0; k < p; ++k) for (int e = 0; e < d; ++e) {
        const double v = zs[(size_t)k * db + e];
        dP[(size_t)k * nxx + pr.ia[e] * nx + pr.ib[e]] = v; dP[(size_t)k * nxx + pr.ib[e] * nx + pr.ia[e]] = v;
      }
      for (int k = 0; k < p; ++k) for (int i = 0; i < ng; ++i) dphi[(size_t)k * ng + i] = mask[(size_t)k * ng + i] ? zs[(size_t)((k + 1) % p) * db + d + i] : 0.0;
      pr.calH(dM, dP, dalpha);
      add_gg(dM, dphi);
      double num = 0, den = 0; const double ra = dalpha / alpha;
      for (int k = 0; k < p; ++k) for (int e = 0; e < nn; ++e) { const double mv = val(Mq[k][e]); const double dh = dM[(size_t)k * nn + e] - ra * mv; num += dh * dh; den += mv * mv; }
      stepn = sqrt(num / den);
      double th = 1.0;
      for (;;) {
        for (size_t e = 0; e < P.size(); ++e) Pn_[e] = P[e] + th * dP[e];
        for (size_t e = 0; e < PG && ng; ++e) phin[e] = mask[e] ? phi[e] + th * dphi[e] : 1.0;
        for (size_t e_ = 0; e_ < arrows.size(); ++e_) ttn[e_] = tt[e_] + th * zs[(size_t)((arrows[e_].k + 1) % p) * db + d + ng + arrows[e_].slot];
        if (cones(tau + th * dtau, alpha + th * dalpha, Pn_, phin) && arrow_mats(phin, ttn, aSi)) break;
        th *= 0.5;
        if (th < 1e-3) { okp = false; break; }
      }
      if (!okp) break;
      tau += th * dtau; alpha += th * dalpha; P = Pn_;
      for (size_t e = 0; e < PG && ng; ++e) if (mask[e]) phi[e] = phin[e];
      tt = ttn;
      if (th == 1.0 && stepn < center_tol) { ipm = ST_OPT; break; }
    }
    if (npolish > POLISH_MAX) npolish = POLISH_MAX;
    for (size_t e_ = 0; e_ < arrows.size(); ++e_) arrows[e_].t = tt[e_];
  }
  // un-scaling, supplement, status (convexifier.py:403-456)
  const double sc = 1.0 / (s * alpha);
  vec Pst((size_t)p * nxx), phs(PG, 0.0);
  for (size_t e = 0; e < Pst.size(); ++e) Pst[e] = sc * P[e];
  for (size_t e = 0; e < PG && ng; ++e) phs[e] = mask[e] ? sc * phi[e] : 0.0;
  std::fill(pr.Hb.begin(), pr.Hb.end(), 0.0);
  pr.calH(dM, Pst, 0.0);
  add_gg(dM, phs);
  double lo2 = 1e300;
  for (int k = 0; k < p; ++k) {
    const size_t o = (size_t)k * nn;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Hc_out[o + i * n + j] = Hs[o + i * n + j] + 0.5 * (dM[o + i * n + j] + dM[o + j * n + i]);
    lo2 = std::min(lo2, sm.min_eig(&Hc_out[o], work.data()));
  }
  if (co.P) memcpy(co.P, Pst.data(), sizeof(double) * Pst.size());
  if (co.FgF) for (int k = 0; k < p; ++k) {
    for (int i = 0; i < ng0; ++i) co.FgF[(size_t)k * nJin + i] = phs[(size_t)k * ng + i];
    for (int i = 0; i < ncs[k]; ++i) co.FgF[(size_t)k * nJin + ng0 + i] = phs[(size_t)k * ng + ng0 + i];
  }
  if (co.T && nT) for (int k = 0; k < p; ++k) for (int q = 0; q < nT; ++q) { co.T[(size_t)k * nn + ta[q] * n + tb[q]] = phs[(size_t)k * ng + nJ + q]; co.T[(size_t)k * nn + tb[q] * n + ta[q]] = phs[(size_t)k * ng + nJ + q]; }
  double tsum = 0; for (const Arrow& a : arrows) tsum += a.t;
  co.objective = tau / sbeta + tsum / sbeta;
  res.kappa = tau; res.alpha = alpha; res.iters = it; res.mu_t = mu_t; res.stepn = stepn; res.dd_iters = ndd; res.polish = npolish;
  res.status = (lo2 > 0.0) ? (ipm == ST_OPT ? 0 : 1) : 2;
  return res;
}
