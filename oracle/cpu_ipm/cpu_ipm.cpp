// CPU baseline of the convexify() hot path  --  TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and tests/).
//
// A C++/OpenMP restatement of the same structured interior-point algorithm as oracle/convexify_oracle.py (plain Step 1 model of
// tunempc/convexifier.py:213-308 with constr=False: scaling :374-401, LMIs :304-306, un-scaling :403-435, status :437-456),
// written for speed on host cores: the d x d blocks of the block-cyclic-tridiagonal Schur complement go through LAPACK/BLAS
// (dpotrf / dtrsm / dsyrk / dgemm of the OpenBLAS that ships inside the scipy wheel of this image), the n x n stage work is plain
// loops, and a batch is spread over OpenMP threads one problem per thread (BLAS single-threaded inside).  It is NOT PICOS+MOSEK
// (the reference stack cannot run here) and the product never links or calls it.
//
// Build: oracle/cpu_ipm/Makefile (g++ -O3 -fopenmp, linked against scipy.libs/libscipy_openblas*.so).
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

extern "C" {
// Fortran BLAS / LAPACK of scipy's bundled OpenBLAS (LP64, symbols carry the scipy_ prefix)
void scipy_dgemm_(const char*, const char*, const int*, const int*, const int*, const double*, const double*, const int*, const double*,
                  const int*, const double*, double*, const int*);
void scipy_dsyrk_(const char*, const char*, const int*, const int*, const double*, const double*, const int*, const double*, double*, const int*);
void scipy_dtrsm_(const char*, const char*, const char*, const char*, const int*, const int*, const double*, const double*, const int*, double*,
                  const int*);
void scipy_dpotrf_(const char*, const int*, double*, const int*, int*);
void scipy_dsyev_(const char*, const char*, const int*, double*, const int*, double*, double*, const int*, int*);
void scipy_openblas_set_num_threads(int);
int scipy_openblas_get_num_threads(void);
}

namespace {

constexpr double ALPHA_MIN = 1e-8;    // convexifier.py:245
typedef std::vector<double> vec;

// ---------------------------------------------------------------- small dense helpers (row-major n x n, n <= 32)
struct Small {
  int n;
  explicit Small(int n_) : n(n_) {}
  void mm(double* C, const double* A, const double* B) const {            // C = A B
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += A[i * n + k] * B[k * n + j]; C[i * n + j] = s; }
  }
  void sym3(double* out, const double* X, const double* Z, const double* S, double* t0, double* t1) const {   // out = sym(X Z S)
    mm(t0, X, Z); mm(t1, t0, S);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) out[i * n + j] = 0.5 * (t1[i * n + j] + t1[j * n + i]);
  }
  bool chol(double* L, const double* A) const {                           // lower Cholesky, false when not positive definite
    memset(L, 0, sizeof(double) * n * n);
    for (int j = 0; j < n; ++j) {
      double s = A[j * n + j];
      for (int k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
      if (!(s > 0.0)) return false;
      const double dj = sqrt(s);
      L[j * n + j] = dj;
      for (int i = j + 1; i < n; ++i) {
        double t = A[i * n + j];
        for (int k = 0; k < j; ++k) t -= L[i * n + k] * L[j * n + k];
        L[i * n + j] = t / dj;
      }
    }
    return true;
  }
  void tri_inv(double* Li, const double* L) const {                       // inverse of a lower-triangular matrix
    memset(Li, 0, sizeof(double) * n * n);
    for (int c = 0; c < n; ++c) {
      Li[c * n + c] = 1.0 / L[c * n + c];
      for (int i = c + 1; i < n; ++i) {
        double s = 0;
        for (int k = c; k < i; ++k) s += L[i * n + k] * Li[k * n + c];
        Li[i * n + c] = -s / L[i * n + i];
      }
    }
  }
  double min_eig(const double* W, double* work) const {                   // smallest eigenvalue of sym(W)
    double* a = work; double* ev = work + n * n; double* wk = ev + n;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) a[i * n + j] = 0.5 * (W[i * n + j] + W[j * n + i]);
    int info, lw = 3 * n + 32;
    scipy_dsyev_("N", "L", &n, a, &n, ev, wk, &lw, &info);
    return ev[0];
  }
  void eigvals(const double* W, double* ev, double* work) const {
    double* a = work; double* wk = work + n * n;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) a[i * n + j] = 0.5 * (W[i * n + j] + W[j * n + i]);
    int info, lw = 3 * n + 32;
    scipy_dsyev_("N", "L", &n, a, &n, ev, wk, &lw, &info);
  }
};

// ---------------------------------------------------------------- one tuning problem
struct Problem {
  int p, nx, mb, n, d;
  const double *A, *B, *H;        // inputs (row-major)
  vec V, Hb;                      // [p][nx][n], [p][n][n]
  std::vector<int> ia, ib;        // svec index -> (a, c), a <= c, row-major upper triangle

  // block system (column-major d x d blocks)
  vec D, Csub, Lkk, O, F, Ldense;
  double shift = 0.0;

  void setup(int p_, int nx_, int mb_, const double* A_, const double* B_, const double* H_) {
    p = p_; nx = nx_; mb = mb_; n = nx + mb; d = nx * (nx + 1) / 2; A = A_; B = B_; H = H_;
    V.assign((size_t)p * nx * n, 0.0);
    for (int k = 0; k < p; ++k)
      for (int i = 0; i < nx; ++i)
        for (int j = 0; j < n; ++j) V[((size_t)k * nx + i) * n + j] = (j < nx) ? A[((size_t)k * nx + i) * nx + j] : B[((size_t)k * nx + i) * mb + (j - nx)];
    ia.clear(); ib.clear();
    for (int a = 0; a < nx; ++a) for (int c = a; c < nx; ++c) { ia.push_back(a); ib.push_back(c); }
  }
  const double* Vk(int k) const { return &V[(size_t)k * nx * n]; }

  // out[k] = coef*Hb_k + V_k' P_{k+1} V_k - E' P_k E      (convexifier.py:339-343)
  void calH(vec& out, const vec& P, double coef) const {
    vec t((size_t)n * nx);
    for (int k = 0; k < p; ++k) {
      const int kn = (k + 1) % p;
      const double* v = Vk(k); const double* Pn = &P[(size_t)kn * nx * nx]; const double* Pk = &P[(size_t)k * nx * nx];
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < nx; ++j) { double s = 0; for (int a = 0; a < nx; ++a) s += v[a * n + i] * Pn[a * nx + j]; t[i * nx + j] = s; }   // V' Pn
      double* o = &out[(size_t)k * n * n];
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
          double s = 0; for (int a = 0; a < nx; ++a) s += t[i * nx + a] * v[a * n + j];
          s += coef * Hb[((size_t)k * n + i) * n + j];
          if (i < nx && j < nx) s -= Pk[i * nx + j];
          o[i * n + j] = s;
        }
    }
  }
  // adjoint: out_j = V_{j-1} G_{j-1} V_{j-1}' - G_j[:nx,:nx]
  void adj(vec& out, const vec& G) const {
    vec t((size_t)nx * n);
    for (int j = 0; j < p; ++j) {
      const int jm = (j + p - 1) % p;
      const double* v = Vk(jm); const double* g = &G[(size_t)jm * n * n]; const double* gj = &G[(size_t)j * n * n];
      for (int a = 0; a < nx; ++a)
        for (int c = 0; c < n; ++c) { double s = 0; for (int r = 0; r < n; ++r) s += v[a * n + r] * g[r * n + c]; t[a * n + c] = s; }
      double* o = &out[(size_t)j * nx * nx];
      for (int a = 0; a < nx; ++a)
        for (int c = 0; c < nx; ++c) { double s = 0; for (int r = 0; r < n; ++r) s += t[a * n + r] * v[c * n + r]; o[a * nx + c] = s - gj[a * n + c]; }
    }
  }
  // svec coordinates of the gradient: weight 1 on the diagonal, 2 off it
  void svec_grad(double* out, const double* G) const {
    for (int e = 0; e < d; ++e) out[e] = G[ia[e] * nx + ib[e]] * (ia[e] == ib[e] ? 1.0 : 2.0);
  }
  // T(L,R)[(ab),(cd)] = <E_ab, L E_cd R'>; accumulated with sign sg into the column-major block M; transposed store when tr
  void add_T(double* M, const double* L, const double* R, double sg, bool tr) const {
    for (int c = 0; c < d; ++c) {
      const int cc = ia[c], dd = ib[c];
      const double wc = (cc == dd) ? 0.5 : 1.0;
      for (int r = 0; r < d; ++r) {
        const int a = ia[r], b = ib[r];
        const double wr = (a == b) ? 0.5 : 1.0;
        const double t = (L[a * nx + cc] * R[b * nx + dd] + L[a * nx + dd] * R[b * nx + cc]) + (L[b * nx + cc] * R[a * nx + dd] + L[b * nx + dd] * R[a * nx + cc]);
        const double v = sg * wr * wc * t;
        if (tr) M[(size_t)r * d + c] += v; else M[(size_t)c * d + r] += v;      // column-major: entry (row r, col c) at c*d + r
      }
    }
  }

  // Cholesky of the block-cyclic-tridiagonal matrix: diagonal blocks D[k], Csub[k] = block [k+1][k] (k < p-1), Csub[p-1] = block [p-1][0]
  bool factor_once(double sh) {
    const int dd_ = d; const double one = 1.0, mone = -1.0, zero = 0.0; int info;
    const size_t bs = (size_t)d * d;
    if (p <= 2) {
      const int N = p * d;
      Ldense.assign((size_t)N * N, 0.0);
      for (int k = 0; k < p; ++k) {
        for (int c = 0; c < d; ++c) for (int r = 0; r < d; ++r) Ldense[(size_t)(k * d + c) * N + k * d + r] += D[k * bs + (size_t)c * d + r];
        const int kn = (k + 1) % p;
        if (kn == k) {      // p == 1: C + C' folds onto the diagonal (Csub[0] holds block [0][0] = C)
          for (int c = 0; c < d; ++c) for (int r = 0; r < d; ++r) Ldense[(size_t)c * N + r] += Csub[(size_t)c * d + r] + Csub[(size_t)r * d + c];
        } else if (k == 0) {   // p == 2: block [1][0] = C_0' + C_1
          for (int c = 0; c < d; ++c) for (int r = 0; r < d; ++r) {
            const double v = Csub[(size_t)c * d + r] + Csub[bs + (size_t)r * d + c];   // Csub[0] = [1][0]; Csub[1] = [1][0]' stored as block [p-1][0] = [1][0]
            Ldense[(size_t)c * N + d + r] += v;
          }
        }
      }
      if (p == 2) {   // Csub[1] is block [p-1][0] = [1][0] as well: both stored in the same orientation, undo the transpose above
        for (int c = 0; c < d; ++c) for (int r = 0; r < d; ++r) Ldense[(size_t)c * N + d + r] += Csub[bs + (size_t)c * d + r] - Csub[bs + (size_t)r * d + c];
      }
      if (sh != 0.0) for (int i = 0; i < N; ++i) Ldense[(size_t)i * N + i] *= (1.0 + sh);
      scipy_dpotrf_("L", &N, Ldense.data(), &N, &info);
      return info == 0;
    }
    Lkk = D;
    if (sh != 0.0) for (int k = 0; k < p; ++k) for (int i = 0; i < d; ++i) Lkk[k * bs + (size_t)i * d + i] += sh * D[k * bs + (size_t)i * d + i];
    O.assign((size_t)p * bs, 0.0); F.assign((size_t)p * bs, 0.0);
    vec Fpre(Csub.begin() + (size_t)(p - 1) * bs, Csub.begin() + (size_t)p * bs);      // block [p-1][0]
    for (int k = 0; k < p - 1; ++k) {
      double* Lk = &Lkk[k * bs];
      scipy_dpotrf_("L", &dd_, Lk, &dd_, &info);
      if (info != 0) return false;
      double* Ok = &O[k * bs];
      memcpy(Ok, &Csub[k * bs], bs * sizeof(double));
      if (k == p - 2) {
        for (size_t e = 0; e < bs; ++e) Ok[e] += Fpre[e];                              // the fill meets the sub-diagonal block
        scipy_dtrsm_("R", "L", "T", "N", &dd_, &dd_, &one, Lk, &dd_, Ok, &dd_);
        scipy_dsyrk_("L", "N", &dd_, &dd_, &mone, Ok, &dd_, &one, &Lkk[(size_t)(p - 1) * bs], &dd_);
      } else {
        double* Fk = &F[k * bs];
        memcpy(Fk, Fpre.data(), bs * sizeof(double));
        scipy_dtrsm_("R", "L", "T", "N", &dd_, &dd_, &one, Lk, &dd_, Ok, &dd_);
        scipy_dtrsm_("R", "L", "T", "N", &dd_, &dd_, &one, Lk, &dd_, Fk, &dd_);
        scipy_dsyrk_("L", "N", &dd_, &dd_, &mone, Ok, &dd_, &one, &Lkk[(size_t)(k + 1) * bs], &dd_);
        scipy_dsyrk_("L", "N", &dd_, &dd_, &mone, Fk, &dd_, &one, &Lkk[(size_t)(p - 1) * bs], &dd_);
        scipy_dgemm_("N", "T", &dd_, &dd_, &dd_, &mone, Fk, &dd_, Ok, &dd_, &zero, Fpre.data(), &dd_);
      }
    }
    scipy_dpotrf_("L", &dd_, &Lkk[(size_t)(p - 1) * bs], &dd_, &info);
    return info == 0;
  }
  bool factor() {
    double sh = 0.0;
    for (;;) {
      if (factor_once(sh)) { shift = sh; return true; }
      sh = (sh == 0.0) ? 1e-13 : sh * 100.0;
      if (sh > 1e-2) return false;
    }
  }
  // R [p][d] x nrhs, stored as nrhs column-major panels: R[(q*p + k)*d + i]
  void solve(double* R, int nrhs) const {
    const double one = 1.0, mone = -1.0; const int dd_ = d; const size_t bs = (size_t)d * d;
    for (int q = 0; q < nrhs; ++q) {
      double* z = R + (size_t)q * p * d;
      const int i1 = 1;
      if (p <= 2) {
        const int N = p * d;
        scipy_dtrsm_("L", "L", "N", "N", &N, &i1, &one, Ldense.data(), &N, z, &N);
        scipy_dtrsm_("L", "L", "T", "N", &N, &i1, &one, Ldense.data(), &N, z, &N);
        continue;
      }
      const double* beta1 = &one;
      for (int k = 0; k < p - 1; ++k) {
        scipy_dtrsm_("L", "L", "N", "N", &dd_, &i1, &one, &Lkk[k * bs], &dd_, z + (size_t)k * d, &dd_);
        scipy_dgemm_("N", "N", &dd_, &i1, &dd_, &mone, &O[k * bs], &dd_, z + (size_t)k * d, &dd_, beta1, z + (size_t)(k + 1) * d, &dd_);
        if (k < p - 2) scipy_dgemm_("N", "N", &dd_, &i1, &dd_, &mone, &F[k * bs], &dd_, z + (size_t)k * d, &dd_, beta1, z + (size_t)(p - 1) * d, &dd_);
      }
      scipy_dtrsm_("L", "L", "N", "N", &dd_, &i1, &one, &Lkk[(size_t)(p - 1) * bs], &dd_, z + (size_t)(p - 1) * d, &dd_);
      scipy_dtrsm_("L", "L", "T", "N", &dd_, &i1, &one, &Lkk[(size_t)(p - 1) * bs], &dd_, z + (size_t)(p - 1) * d, &dd_);
      for (int k = p - 2; k >= 0; --k) {
        scipy_dgemm_("T", "N", &dd_, &i1, &dd_, &mone, &O[k * bs], &dd_, z + (size_t)(k + 1) * d, &dd_, beta1, z + (size_t)k * d, &dd_);
        if (k < p - 2) scipy_dgemm_("T", "N", &dd_, &i1, &dd_, &mone, &F[k * bs], &dd_, z + (size_t)(p - 1) * d, &dd_, beta1, z + (size_t)k * d, &dd_);
        scipy_dtrsm_("L", "L", "T", "N", &dd_, &i1, &one, &Lkk[k * bs], &dd_, z + (size_t)k * d, &dd_);
      }
    }
  }
};

struct Result { double kappa, alpha; int status, iters, early; };

// status codes as include/tunempc_hip.h: 0 Optimal, 1 Feasible, 2 Infeasible
Result solve_problem(int p, int nx, int mb, const double* A, const double* B, const double* Hin, double tol, int max_iter, int center_iter,
                     double center_tol, double* Hc_out) {
  const int n = nx + mb, nn = n * n, d = nx * (nx + 1) / 2, nxx = nx * nx;
  Small sm(n);
  Result res; res.kappa = 0; res.alpha = 1; res.status = 0; res.iters = 0; res.early = 0;
  vec Hs((size_t)p * nn);
  for (int k = 0; k < p; ++k) for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j)
    Hs[((size_t)k * n + i) * n + j] = 0.5 * (Hin[((size_t)k * n + i) * n + j] + Hin[((size_t)k * n + j) * n + i]);
  vec work((size_t)nn + 4 * n + 64), ev(n);
  // pre-check (convexifier.py:82-85) and scaling (:374-401)
  double lo = 1e300, amin = 1e10, amax = 0.0;
  for (int k = 0; k < p; ++k) {
    sm.eigvals(&Hs[(size_t)k * nn], ev.data(), work.data());
    for (int i = 0; i < n; ++i) { lo = std::min(lo, ev[i]); const double a = fabs(ev[i]); if (a != 0.0) { amin = std::min(amin, a); amax = std::max(amax, a); } }
  }
  if (lo > 0.0) { memcpy(Hc_out, Hs.data(), sizeof(double) * p * nn); res.early = 1; return res; }
  const double s = 1.0 / amin, sbeta = amax / amin;
  Problem pr; pr.setup(p, nx, mb, A, B, Hs.data());
  pr.Hb.resize((size_t)p * nn);
  for (size_t e = 0; e < (size_t)p * nn; ++e) pr.Hb[e] = s * Hs[e];
  const vec& Hb = pr.Hb;
  const double N = 2.0 * p * n + 1.0;
  double tau = 2.0, alpha = 1.0 / sbeta, s0 = alpha, x0 = 1.0 / (p * n);
  vec P((size_t)p * nxx, 0.0), S1((size_t)p * nn, 0.0), S2((size_t)p * nn), X1((size_t)p * nn, 0.0), X2;
  for (int k = 0; k < p; ++k) for (int i = 0; i < n; ++i) { S1[((size_t)k * n + i) * n + i] = 1.0; X1[((size_t)k * n + i) * n + i] = x0; }
  for (int k = 0; k < p; ++k) for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j)
    S2[((size_t)k * n + i) * n + j] = (i == j ? tau : 0.0) - alpha * Hb[((size_t)k * n + i) * n + j];
  X2 = X1;
  const size_t PN = (size_t)p * nn, bs = (size_t)d * d;
  vec M(PN), Rd1(PN), Rd2(PN), Y(PN), rP((size_t)p * nxx), L1(PN), L2(PN), L1i(PN), L2i(PN), S1i(PN), S2i(PN), LX1i(PN), LX2i(PN);
  vec Psi(PN), PhiH(PN), Phi2(PN), T1(PN), T2(PN), G(PN), dM(PN), dS1(PN), dS2(PN), dX1(PN), dX2(PN), c1(PN), c2(PN), dP((size_t)p * nxx);
  vec adjb((size_t)p * nxx), U((size_t)2 * p * d), TU((size_t)2 * p * d), zs((size_t)p * d), t0(nn), t1(nn), t2(nn);
  vec Kx(nxx), Ks(nxx), Fx(nxx), Fs(nxx), tx((size_t)nx * n), Xxx(nxx), Sxx(nxx);
  pr.D.resize((size_t)p * bs); pr.Csub.resize((size_t)p * bs);
  double mu_t = -1.0, mu = 0, mu0 = 0, pinf = 0, dinf = 0, stepn = 1e300, prev_stepn = -1.0;
  int phase = 0, ncent = 0, njam = 0, nshiftrun = 0, nbackoff = 0, it = 0;
  const int MUT_BACKOFF_MAX = 10;          // as the oracle and the HIP path
  enum { ST_MAXIT, ST_OPT, ST_INACC, ST_DIV } ipm = ST_MAXIT;
  auto dot = [&](const vec& a, const vec& b) { double v = 0; for (size_t e = 0; e < PN; ++e) v += a[e] * b[e]; return v; };
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
  double dtau = 0, dalpha = 0, ds0 = 0, dx0 = 0;
  for (it = 0; it < max_iter + center_iter * (MUT_BACKOFF_MAX + 1) + 1; ++it) {
    pr.calH(M, P, alpha);
    double rd2 = 0, s2 = 0;
    for (int k = 0; k < p; ++k) for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
      const size_t e = ((size_t)k * n + i) * n + j; const double dg = (i == j) ? 1.0 : 0.0;
      Rd1[e] = (M[e] - dg) - S1[e]; Rd2[e] = (tau * dg - M[e]) - S2[e]; Y[e] = X1[e] - X2[e];
      rd2 += Rd1[e] * Rd1[e] + Rd2[e] * Rd2[e]; s2 += S1[e] * S1[e] + S2[e] * S2[e];
    }
    const double rd0 = (alpha - ALPHA_MIN) - s0;
    mu = (dot(X1, S1) + dot(X2, S2) + x0 * s0) / N;
    const double r_tau = 1.0 - trace_sum(X2), r_alpha = -dot(Hb, Y) - x0;
    pr.adj(rP, Y);
    double rp2 = 0;
    for (int k = 0; k < p; ++k) for (int e = 0; e < d; ++e) { const double v = -rP[(size_t)k * nxx + pr.ia[e] * nx + pr.ib[e]] * (pr.ia[e] == pr.ib[e] ? 1.0 : 2.0); rp2 += v * v; }
    pinf = sqrt(r_tau * r_tau + r_alpha * r_alpha + rp2) / 2.0;
    dinf = sqrt(rd2 + rd0 * rd0) / (1.0 + sqrt(s2));
    const double relgap = N * mu / std::max(1.0, fabs(tau));
    if (it == 0) mu0 = mu;
    if (!(mu > 0.0) || !std::isfinite(mu) || !std::isfinite(tau) || mu > 1e6 * mu0) { ipm = ST_DIV; break; }
    if (mu_t < 0.0 && relgap < 1e-2 && dinf < 1e-2) mu_t = exp2(rint(log2(tol * std::max(1.0, fabs(tau)))));
    if (phase == 0 && mu_t > 0.0 && mu <= 2.0 * mu_t && dinf < 1e-6 && (pinf < 1e-3 || nshiftrun >= 1)) phase = 1;      // (after a shifted factorisation pinf is noise: convexify_oracle.py)
    if (phase == 0 && nshiftrun >= 2) {      // the wall met on the way down: centre at the power of two above the current mu (convexify_oracle.py, k_ctrl_a)
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
    // ---- Schur complement blocks from the Kronecker factors
    std::fill(pr.D.begin(), pr.D.end(), 0.0); std::fill(pr.Csub.begin(), pr.Csub.end(), 0.0);
    for (int r = 0; r < 2; ++r) {
      const vec& X = r ? X2 : X1; const vec& Si = r ? S2i : S1i;
      for (int k = 0; k < p; ++k) {
        const size_t o = (size_t)k * nn; const double* v = pr.Vk(k); const int kn = (k + 1) % p;
        auto VZVt = [&](double* out, double* fz, const double* Z) {       // out = V Z V' (nx x nx); fz = Z[:nx,:] V' (nx x nx)
          for (int a = 0; a < nx; ++a) for (int c = 0; c < n; ++c) { double sacc = 0; for (int q = 0; q < n; ++q) sacc += v[a * n + q] * Z[q * n + c]; tx[a * n + c] = sacc; }
          for (int a = 0; a < nx; ++a) for (int c = 0; c < nx; ++c) { double sacc = 0; for (int q = 0; q < n; ++q) sacc += tx[a * n + q] * v[c * n + q]; out[a * nx + c] = sacc; }
          for (int a = 0; a < nx; ++a) for (int c = 0; c < nx; ++c) { double sacc = 0; for (int q = 0; q < n; ++q) sacc += Z[a * n + q] * v[c * n + q]; fz[a * nx + c] = sacc; }
        };
        VZVt(Kx.data(), Fx.data(), &X[o]); VZVt(Ks.data(), Fs.data(), &Si[o]);
        for (int a = 0; a < nx; ++a) for (int c = 0; c < nx; ++c) { Xxx[a * nx + c] = X[o + a * n + c]; Sxx[a * nx + c] = Si[o + a * n + c]; }
        pr.add_T(&pr.D[k * bs], Xxx.data(), Sxx.data(), 1.0, false);
        pr.add_T(&pr.D[kn * bs], Kx.data(), Ks.data(), 1.0, false);
        // C_k[(ab) in P_k, (cd) in P_{k+1}] = -T(Fx,Fs); stored as block [k+1][k] = C_k' (k < p-1) or block [p-1][0] = C_{p-1} (k = p-1)
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
    pr.adj(adjb, Psi);
    for (int k = 0; k < p; ++k) { pr.svec_grad(&U[(size_t)k * d], &adjb[(size_t)k * nxx]); for (int e = 0; e < d; ++e) U[(size_t)k * d + e] = -U[(size_t)k * d + e]; }
    pr.adj(adjb, PhiH);
    for (int k = 0; k < p; ++k) pr.svec_grad(&U[(size_t)(p + k) * d], &adjb[(size_t)k * nxx]);
    if (!pr.factor()) { ipm = ST_INACC; break; }
    nshiftrun = pr.shift > 0.0 ? nshiftrun + 1 : 0;
    if (phase == 1 && pr.shift > 0.0 && nbackoff < MUT_BACKOFF_MAX) {
      // hard target: aim one power of two earlier and take the step of the shifted factorisation towards it (convexify_oracle.py, k_ctrl_b)
      mu_t *= 2.0; ++nbackoff; ncent = 0; prev_stepn = -1.0; nshiftrun = 0;
    } else if ((phase == 1 && pr.shift > 0.0) || (nshiftrun >= 2 && (mu_t < 0.0 || nbackoff >= MUT_BACKOFF_MAX))) { ipm = ST_INACC; break; }
    TU = U; pr.solve(TU.data(), 2);
    double sb00 = b_tt, sb01 = b_ta, sb11 = b_aa;
    for (size_t e = 0; e < (size_t)p * d; ++e) { sb00 -= U[e] * TU[e]; sb01 -= U[e] * TU[(size_t)p * d + e]; sb11 -= U[(size_t)p * d + e] * TU[(size_t)p * d + e]; }
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
      for (int k = 0; k < p; ++k) pr.svec_grad(&zs[(size_t)k * d], &adjb[(size_t)k * nxx]);
      pr.solve(zs.data(), 1);
      double u0 = 0, u1 = 0;
      for (size_t e = 0; e < (size_t)p * d; ++e) { u0 += U[e] * zs[e]; u1 += U[(size_t)p * d + e] * zs[e]; }
      const double rb0 = rhs_tau - u0, rb1 = rhs_alpha - u1, det = sb00 * sb11 - sb01 * sb01;
      dtau = (sb11 * rb0 - sb01 * rb1) / det; dalpha = (sb00 * rb1 - sb01 * rb0) / det;
      for (int k = 0; k < p; ++k) for (int e = 0; e < d; ++e) {
        const double v = zs[(size_t)k * d + e] - TU[(size_t)k * d + e] * dtau - TU[(size_t)(p + k) * d + e] * dalpha;
        dP[(size_t)k * nxx + pr.ia[e] * nx + pr.ib[e]] = v; dP[(size_t)k * nxx + pr.ib[e] * nx + pr.ia[e]] = v;
      }
      pr.calH(dM, dP, dalpha);
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
    };
    auto steps = [&](double& ap, double& ad) {
      ap = std::min(max_step(LX1i, dX1), max_step(LX2i, dX2));
      ad = std::min(max_step(L1i, dS1), max_step(L2i, dS2));
      if (dx0 < 0) ap = std::min(ap, -x0 / dx0);
      if (ds0 < 0) ad = std::min(ad, -s0 / ds0);
    };
    double ap, ad;
    if (phase == 0) {
      direction(0.0, false, 0.0);
      steps(ap, ad);
      ap = std::min(1.0, ap); ad = std::min(1.0, ad);
      double xs_aff = (x0 + ap * dx0) * (s0 + ad * ds0);
      for (size_t e = 0; e < PN; ++e) xs_aff += (X1[e] + ap * dX1[e]) * (S1[e] + ad * dS1[e]) + (X2[e] + ap * dX2[e]) * (S2[e] + ad * dS2[e]);
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
    if (phase == 1) {
      const bool full = (ap == 1.0 && ad == 1.0);
      const double est = (prev_stepn >= 0.0) ? stepn * pow(std::min(1.0, stepn / prev_stepn), 1.5) : stepn;
      if (full && (stepn < center_tol || est < 0.1 * center_tol)) { ipm = ST_OPT; ++it; break; }
      if (full && prev_stepn >= 0.0 && stepn > 0.5 * prev_stepn && stepn < 1e-6) { ipm = ST_OPT; ++it; break; }
      if (ncent >= center_iter) {
        if (nbackoff < MUT_BACKOFF_MAX) { mu_t *= 2.0; ++nbackoff; ncent = 0; prev_stepn = -1.0; continue; }
        ipm = ST_INACC; ++it; break;
      }
      prev_stepn = full ? stepn : -1.0;
    }
  }
  // un-scaling, supplement, status (convexifier.py:403-456)
  const double sc = 1.0 / (s * alpha);
  vec Pst((size_t)p * nxx);
  for (size_t e = 0; e < Pst.size(); ++e) Pst[e] = sc * P[e];
  vec zero_hb = pr.Hb; std::fill(pr.Hb.begin(), pr.Hb.end(), 0.0);
  pr.calH(dM, Pst, 0.0);
  double lo2 = 1e300;
  for (int k = 0; k < p; ++k) {
    const size_t o = (size_t)k * nn;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Hc_out[o + i * n + j] = Hs[o + i * n + j] + 0.5 * (dM[o + i * n + j] + dM[o + j * n + i]);
    lo2 = std::min(lo2, sm.min_eig(&Hc_out[o], work.data()));
  }
  res.kappa = tau; res.alpha = alpha; res.iters = it;
  res.status = (lo2 > 0.0) ? (ipm == ST_OPT ? 0 : 1) : 2;
  return res;
}

}  // namespace

extern "C" {

// Batched Step 1 on host cores: problems are spread over `threads` OpenMP threads (BLAS single-threaded inside each problem).
// A [nb][p][nx][nx], B [nb][p][nx][mb], H [nb][p][n][n] -> Hc [nb][p][n][n], kappa [nb], status [nb] (0/1/2), iters [nb].
int cpu_ipm_convexify_batch(int nb, int p, int nx, int mb, const double* A, const double* B, const double* H, double tol, int threads,
                            double* Hc, double* kappa, int32_t* status, int32_t* iters) {
  if (nb < 0 || p < 1 || nx < 1 || mb < 0 || !A || !H || !Hc) return -1;
  const int n = nx + mb;
  if (tol <= 0.0) tol = 0x1p-25;
  const int blas_threads_before = scipy_openblas_get_num_threads();     // the process's scipy shares this OpenBLAS: put its setting back afterwards
  scipy_openblas_set_num_threads(1);
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
  for (int b = 0; b < nb; ++b) {
    const Result r = solve_problem(p, nx, mb, A + (size_t)b * p * nx * nx, B + (size_t)b * p * nx * mb, H + (size_t)b * p * n * n, tol, 50, 12, 1e-9,
                                   Hc + (size_t)b * p * n * n);
    if (kappa) kappa[b] = r.kappa;
    if (status) status[b] = r.status;
    if (iters) iters[b] = r.iters;
  }
  scipy_openblas_set_num_threads(blas_threads_before);
  return 0;
}

int cpu_ipm_max_threads(void) { return omp_get_max_threads(); }

}  // extern "C"
