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
void scipy_sgemm_(const char*, const char*, const int*, const int*, const int*, const float*, const float*, const int*, const float*, const int*, const float*, float*, const int*);
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

// ---------------------------------------------------------------- double-double ("dd") arithmetic for the tight-accuracy mode
// (oracle/ddnum.py is the numpy twin; algorithms: Dekker 1971, Knuth 4.2.2, Hida-Li-Bailey's QD).  A value is hi + lo, |lo| <= ulp(hi)/2.
// Block kernels work on split arrays (hi[], lo[]) with branch-free inner loops over contiguous entries, which gcc vectorises (AVX2 + FMA).
namespace ddk {
struct dd { double h, l; };
static inline dd qts(double a, double b) { const double s = a + b; return {s, b - (s - a)}; }
static inline dd ts(double a, double b) { const double s = a + b, v = s - a; return {s, (a - (s - v)) + (b - v)}; }
static inline dd tp(double a, double b) { const double p = a * b; return {p, fma(a, b, -p)}; }
static inline dd add(dd a, dd b) { dd s = ts(a.h, b.h); const dd t = ts(a.l, b.l); s.l += t.h; s = qts(s.h, s.l); s.l += t.l; return qts(s.h, s.l); }
static inline dd neg(dd a) { return {-a.h, -a.l}; }
static inline dd sub(dd a, dd b) { return add(a, neg(b)); }
static inline dd mul(dd a, dd b) { dd p = tp(a.h, b.h); p.l += a.h * b.l + a.l * b.h; return qts(p.h, p.l); }
static inline dd muld(dd a, double b) { dd p = tp(a.h, b); p.l += a.l * b; return qts(p.h, p.l); }
static inline dd from(double a) { return {a, 0.0}; }
static inline double val(dd a) { return a.h + a.l; }
static inline dd div(dd a, dd b) {
  const double q1 = a.h / b.h; dd r = sub(a, muld(b, q1));
  const double q2 = r.h / b.h; r = sub(r, muld(b, q2));
  const double q3 = r.h / b.h;
  return add(qts(q1, q2), from(q3));
}
static inline dd sqrt_(dd a) { const double x = 1.0 / sqrt(a.h), ax = a.h * x; const double err = sub(a, tp(ax, ax)).h; return qts(ax, err * x * 0.5); }

// c[0..n) -= a * b[0..n)   (dd scalar a, split dd vectors)
static inline void axpy_sub(double* __restrict ch, double* __restrict cl, double ah, double al, const double* __restrict bh, const double* __restrict bl, int n) {
#pragma omp simd
  for (int j = 0; j < n; ++j) {
    const double p = ah * bh[j];
    double e = fma(ah, bh[j], -p); e = fma(ah, bl[j], e); e = fma(al, bh[j], e);
    const double c0 = ch[j], s = c0 - p, v = s - c0;
    double t = (c0 - (s - v)) + (-p - v);
    t += cl[j] - e;
    const double h = s + t;
    cl[j] = t - (h - s); ch[j] = h;
  }
}
// c[0..n) *= a
static inline void scale(double* __restrict ch, double* __restrict cl, dd a, int n) {
  for (int j = 0; j < n; ++j) { const dd r = mul(dd{ch[j], cl[j]}, a); ch[j] = r.h; cl[j] = r.l; }
}
// lower Cholesky in place (row-major n x n, leading dimension ld, lower part used / written); false on a non-positive pivot
static bool potrf(double* Ah, double* Al, int n, int ld, std::vector<double>& tmp) {
  tmp.resize(2 * (size_t)n);
  double* colh = tmp.data(); double* coll = colh + n;
  for (int j = 0; j < n; ++j) {
    const dd piv{Ah[(size_t)j * ld + j], Al[(size_t)j * ld + j]};
    if (!(piv.h > 0.0)) return false;
    const dd r = sqrt_(piv);
    Ah[(size_t)j * ld + j] = r.h; Al[(size_t)j * ld + j] = r.l;
    const dd rinv = div(from(1.0), r);
    for (int i = j + 1; i < n; ++i) {
      const dd v = mul(dd{Ah[(size_t)i * ld + j], Al[(size_t)i * ld + j]}, rinv);
      Ah[(size_t)i * ld + j] = v.h; Al[(size_t)i * ld + j] = v.l; colh[i] = v.h; coll[i] = v.l;
    }
    for (int i = j + 1; i < n; ++i)      // A[i][j+1..i] -= l_ij * l_{j+1..i, j}
      axpy_sub(Ah + (size_t)i * ld + j + 1, Al + (size_t)i * ld + j + 1, colh[i], coll[i], colh + j + 1, coll + j + 1, i - j);
  }
  return true;
}
// Xt (n x m, row-major, rows = columns of X) <- solution of X L' = E given Et = E' in Xt:  Xt[j][:] = (Et[j][:] - sum_{k<j} L[j][k] Xt[k][:]) / L[j][j]
static void trsm_t(double* Xh, double* Xl, int m, const double* Lh, const double* Ll, int n, int ld, bool par) {
  const int nth = par ? omp_get_max_threads() : 1;
#pragma omp parallel for schedule(static) num_threads(nth) if (par)
  for (int c = 0; c < nth; ++c) {
    const int i0 = (int)((long)m * c / nth), i1 = (int)((long)m * (c + 1) / nth), len = i1 - i0;
    if (len <= 0) continue;
    for (int j = 0; j < n; ++j) {
      double* xh = Xh + (size_t)j * m + i0; double* xl = Xl + (size_t)j * m + i0;
      for (int k = 0; k < j; ++k) axpy_sub(xh, xl, Lh[(size_t)j * ld + k], Ll[(size_t)j * ld + k], Xh + (size_t)k * m + i0, Xl + (size_t)k * m + i0, len);
      scale(xh, xl, div(from(1.0), dd{Lh[(size_t)j * ld + j], Ll[(size_t)j * ld + j]}), len);
    }
  }
}
// C (n x n row-major) -= At' Bt  with At, Bt (k x n row-major); lower: only j <= i
static void gemm_tn_sub(double* Ch, double* Cl, const double* Ath, const double* Atl, const double* Bth, const double* Btl, int n, int kk, bool lower, bool par) {
#pragma omp parallel for schedule(dynamic, 4) if (par)
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < kk; ++k)
      axpy_sub(Ch + (size_t)i * n, Cl + (size_t)i * n, Ath[(size_t)k * n + i], Atl[(size_t)k * n + i], Bth + (size_t)k * n, Btl + (size_t)k * n, lower ? i + 1 : n);
}
// small dense dd matrices (stage level, n <= 32): plain loops
typedef std::vector<dd> mat;
static void mm(mat& C, const mat& A, const mat& B, int m, int k, int n, bool bt) {      // C (m x n) = A (m x k) B (k x n), or A B' with B (n x k) when bt
  C.assign((size_t)m * n, dd{0.0, 0.0});
  for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) {
    dd s{0.0, 0.0};
    for (int q = 0; q < k; ++q) s = add(s, mul(A[(size_t)i * k + q], bt ? B[(size_t)j * k + q] : B[(size_t)q * n + j]));
    C[(size_t)i * n + j] = s;
  }
}
static bool inv_spd(mat& Z, const mat& S, int n) {                       // Z = S^-1 by Cholesky and two substitutions per column
  mat L(S);
  for (int j = 0; j < n; ++j) {
    dd s = L[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) s = sub(s, mul(L[(size_t)j * n + k], L[(size_t)j * n + k]));
    if (!(s.h > 0.0)) return false;
    const dd r = sqrt_(s); L[(size_t)j * n + j] = r;
    for (int i = j + 1; i < n; ++i) {
      dd t = L[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) t = sub(t, mul(L[(size_t)i * n + k], L[(size_t)j * n + k]));
      L[(size_t)i * n + j] = div(t, r);
    }
  }
  Z.assign((size_t)n * n, dd{0.0, 0.0});
  std::vector<dd> y(n);
  for (int c = 0; c < n; ++c) {
    for (int i = 0; i < n; ++i) { dd t = from(i == c ? 1.0 : 0.0); for (int k = 0; k < i; ++k) t = sub(t, mul(L[(size_t)i * n + k], y[k])); y[i] = div(t, L[(size_t)i * n + i]); }
    for (int i = n - 1; i >= 0; --i) { dd t = y[i]; for (int k = i + 1; k < n; ++k) t = sub(t, mul(L[(size_t)k * n + i], y[k])); y[i] = div(t, L[(size_t)i * n + i]); }
    for (int i = 0; i < n; ++i) Z[(size_t)i * n + c] = y[i];
  }
  return true;
}
}  // namespace ddk

// ---------------------------------------------------------------- one tuning problem
struct Problem {
  int p, nx, mb, n, d;
  const double *A, *B, *H;        // inputs (row-major)
  vec V, Hb;                      // [p][nx][n], [p][n][n]
  std::vector<int> ia, ib;        // svec index -> (a, c), a <= c, row-major upper triangle

  // block system (column-major db x db blocks; db = d for the plain model, d + stage-local variables for the models with rows: cpu_ipm_con.h)
  int db = 0;
  vec D, Csub, Lkk, O, F, Ldense;
  double shift = 0.0;
  bool lowp = false;              // EXPERIMENT (tests/tools/fp32_update_probe.py): the Schur-complement updates D -= O O', F = -F O' with operands rounded to float32 and
  std::vector<float> fa, fb, fc;  // float32 accumulation (what an fp32-MFMA form of k_cr_update_dma would compute); Cholesky, triangular solves and substitutions stay fp64
  // C (d x d, col-major) {-=, =} -A B' with float32 products
  void upd32(double* C, const double* A, const double* B, int d, bool set) {
    const size_t n2 = (size_t)d * d;
    fa.resize(n2); fb.resize(n2); fc.resize(n2);
    for (size_t e = 0; e < n2; ++e) { fa[e] = (float)A[e]; fb[e] = (float)B[e]; }
    const float one = 1.0f, zero = 0.0f;
    scipy_sgemm_("N", "T", &d, &d, &d, &one, fa.data(), &d, fb.data(), &d, &zero, fc.data(), &d);
    if (set) for (size_t e = 0; e < n2; ++e) C[e] = -(double)fc[e];
    else for (size_t e = 0; e < n2; ++e) C[e] -= (double)fc[e];
  }

  void setup(int p_, int nx_, int mb_, const double* A_, const double* B_, const double* H_) {
    p = p_; nx = nx_; mb = mb_; n = nx + mb; d = nx * (nx + 1) / 2; db = d; A = A_; B = B_; H = H_;
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
        if (tr) M[(size_t)r * db + c] += v; else M[(size_t)c * db + r] += v;      // column-major: entry (row r, col c) at c*db + r
      }
    }
  }

  // Cholesky of the block-cyclic-tridiagonal matrix: diagonal blocks D[k], Csub[k] = block [k+1][k] (k < p-1), Csub[p-1] = block [p-1][0]
  bool factor_once(double sh) {
    const int dd_ = db, d = db; const double one = 1.0, mone = -1.0, zero = 0.0; int info;
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
        if (lowp) upd32(&Lkk[(size_t)(p - 1) * bs], Ok, Ok, d, false);
        else scipy_dsyrk_("L", "N", &dd_, &dd_, &mone, Ok, &dd_, &one, &Lkk[(size_t)(p - 1) * bs], &dd_);
      } else {
        double* Fk = &F[k * bs];
        memcpy(Fk, Fpre.data(), bs * sizeof(double));
        scipy_dtrsm_("R", "L", "T", "N", &dd_, &dd_, &one, Lk, &dd_, Ok, &dd_);
        scipy_dtrsm_("R", "L", "T", "N", &dd_, &dd_, &one, Lk, &dd_, Fk, &dd_);
        if (lowp) {
          upd32(&Lkk[(size_t)(k + 1) * bs], Ok, Ok, d, false); upd32(&Lkk[(size_t)(p - 1) * bs], Fk, Fk, d, false); upd32(Fpre.data(), Fk, Ok, d, true);
        } else {
        scipy_dsyrk_("L", "N", &dd_, &dd_, &mone, Ok, &dd_, &one, &Lkk[(size_t)(k + 1) * bs], &dd_);
        scipy_dsyrk_("L", "N", &dd_, &dd_, &mone, Fk, &dd_, &one, &Lkk[(size_t)(p - 1) * bs], &dd_);
        scipy_dgemm_("N", "T", &dd_, &dd_, &dd_, &mone, Fk, &dd_, Ok, &dd_, &zero, Fpre.data(), &dd_);
        }
      }
    }
    scipy_dpotrf_("L", &dd_, &Lkk[(size_t)(p - 1) * bs], &dd_, &info);
    if (lowp && lowp_solve) { for (double& v : O) v = (double)(float)v; for (double& v : F) v = (double)(float)v; }      // EXPERIMENT: the substitutions read the float32 copies of the O blocks
    return info == 0;
  }
  bool lowp_solve = false;
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
    const double one = 1.0, mone = -1.0; const int dd_ = db, d = db; const size_t bs = (size_t)d * d;
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

// ---------------------------------------------------------------- the block-cyclic-tridiagonal system in double-double (tight mode)
// Same elimination order as Problem::factor_once, no shift.  Row-major d x d blocks, split hi / lo arrays.  C_k = T[P_k, P_{k+1}] (rows P_k);
// the factors O_k = [k+1][k] L_k^-T and F_k = [p-1][k] L_k^-T are kept TRANSPOSED (Ot_k: rows = index of stage k), which makes every inner
// loop of the triangular solves, the updates and the forward substitution a contiguous dd axpy.
struct DdSys {
  int p = 0, d = 0; bool par = false;
  std::vector<double> Dh, Dl, Ch, Cl, Oh, Ol, Fh, Fl, Nh, Nl, tmp, fph, fpl;
  void init(int p_, int d_, bool par_) {
    p = p_; d = d_; par = par_;
    const size_t n = (size_t)p * d * d;
    Dh.assign(n, 0.0); Dl.assign(n, 0.0); Ch.assign(n, 0.0); Cl.assign(n, 0.0);
  }
  // M[(ab)][(cd)] += sg * T(L, R)[(ab),(cd)]   (Problem::add_T in dd; L, R nx x nx dd, row-major block M)
  static void add_T(double* Mh, double* Ml, const ddk::mat& L, const ddk::mat& R, double sg, int nx, int d, const std::vector<int>& ia, const std::vector<int>& ib, int ld = 0) {
    if (ld == 0) ld = d;      // (row stride of the block: d, or d + stage-local variables for the models with rows)
    using namespace ddk;
    for (int r = 0; r < d; ++r) {
      const int a = ia[r], b = ib[r];
      const double wr = (a == b) ? 0.5 : 1.0;
      for (int c = 0; c < d; ++c) {
        const int cc = ia[c], e = ib[c];
        const double wgt = sg * wr * ((cc == e) ? 0.5 : 1.0);
        dd t = add(add(mul(L[a * nx + cc], R[b * nx + e]), mul(L[a * nx + e], R[b * nx + cc])), add(mul(L[b * nx + cc], R[a * nx + e]), mul(L[b * nx + e], R[a * nx + cc])));
        t = muld(t, wgt);                                                          // (a power of two: exact)
        const dd v = add(dd{Mh[(size_t)r * ld + c], Ml[(size_t)r * ld + c]}, t);
        Mh[(size_t)r * ld + c] = v.h; Ml[(size_t)r * ld + c] = v.l;
      }
    }
  }
  bool factor() {
    const size_t bs = (size_t)d * d;
    if (p <= 2) {
      const int N = p * d;
      Nh.assign((size_t)N * N, 0.0); Nl.assign((size_t)N * N, 0.0);
      auto acc = [&](int r, int c, double h, double l) { const ddk::dd v = ddk::add(ddk::dd{Nh[(size_t)r * N + c], Nl[(size_t)r * N + c]}, ddk::dd{h, l}); Nh[(size_t)r * N + c] = v.h; Nl[(size_t)r * N + c] = v.l; };
      for (int k = 0; k < p; ++k) {
        const int kn = (k + 1) % p;
        for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) {
          acc(k * d + i, k * d + j, Dh[k * bs + (size_t)i * d + j], Dl[k * bs + (size_t)i * d + j]);
          const double ch = Ch[k * bs + (size_t)i * d + j], cl = Cl[k * bs + (size_t)i * d + j];
          if (kn == k) { acc(i, j, ch, cl); acc(j, i, ch, cl); }
          else { acc(k * d + i, kn * d + j, ch, cl); acc(kn * d + j, k * d + i, ch, cl); }
        }
      }
      return ddk::potrf(Nh.data(), Nl.data(), N, N, tmp);
    }
    Oh.assign((size_t)p * bs, 0.0); Ol.assign((size_t)p * bs, 0.0); Fh.assign((size_t)p * bs, 0.0); Fl.assign((size_t)p * bs, 0.0);
    fph.assign(bs, 0.0); fpl.assign(bs, 0.0);                                       // Fpre' : block [k][p-1]
    for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) { fph[(size_t)j * d + i] = Ch[(p - 1) * bs + (size_t)i * d + j]; fpl[(size_t)j * d + i] = Cl[(p - 1) * bs + (size_t)i * d + j]; }
    for (int k = 0; k < p - 1; ++k) {
      double* Lh = &Dh[k * bs]; double* Ll = &Dl[k * bs];
      if (!ddk::potrf(Lh, Ll, d, d, tmp)) return false;
      double* oh = &Oh[k * bs]; double* ol = &Ol[k * bs];
      memcpy(oh, &Ch[k * bs], bs * sizeof(double)); memcpy(ol, &Cl[k * bs], bs * sizeof(double));       // E' of O_k = C_k
      if (k == p - 2) {
        for (size_t e = 0; e < bs; ++e) { const ddk::dd v = ddk::add(ddk::dd{oh[e], ol[e]}, ddk::dd{fph[e], fpl[e]}); oh[e] = v.h; ol[e] = v.l; }
        ddk::trsm_t(oh, ol, d, Lh, Ll, d, d, par);
        ddk::gemm_tn_sub(&Dh[(size_t)(p - 1) * bs], &Dl[(size_t)(p - 1) * bs], oh, ol, oh, ol, d, d, true, par);
      } else {
        double* fh = &Fh[k * bs]; double* fl = &Fl[k * bs];
        memcpy(fh, fph.data(), bs * sizeof(double)); memcpy(fl, fpl.data(), bs * sizeof(double));
        ddk::trsm_t(oh, ol, d, Lh, Ll, d, d, par);
        ddk::trsm_t(fh, fl, d, Lh, Ll, d, d, par);
        ddk::gemm_tn_sub(&Dh[(size_t)(k + 1) * bs], &Dl[(size_t)(k + 1) * bs], oh, ol, oh, ol, d, d, true, par);
        ddk::gemm_tn_sub(&Dh[(size_t)(p - 1) * bs], &Dl[(size_t)(p - 1) * bs], fh, fl, fh, fl, d, d, true, par);
        std::fill(fph.begin(), fph.end(), 0.0); std::fill(fpl.begin(), fpl.end(), 0.0);
        ddk::gemm_tn_sub(fph.data(), fpl.data(), oh, ol, fh, fl, d, d, false, par);       // ([p-1][k+1])' = -O_k F_k'
      }
    }
    return ddk::potrf(&Dh[(size_t)(p - 1) * bs], &Dl[(size_t)(p - 1) * bs], d, d, tmp);
  }
  static void fwd(const double* Lh, const double* Ll, int n, int ld, ddk::dd* z) {
    for (int i = 0; i < n; ++i) { ddk::dd t = z[i]; for (int j = 0; j < i; ++j) t = ddk::sub(t, ddk::mul(ddk::dd{Lh[(size_t)i * ld + j], Ll[(size_t)i * ld + j]}, z[j])); z[i] = ddk::div(t, ddk::dd{Lh[(size_t)i * ld + i], Ll[(size_t)i * ld + i]}); }
  }
  static void bwd(const double* Lh, const double* Ll, int n, int ld, ddk::dd* z) {
    for (int i = n - 1; i >= 0; --i) { ddk::dd t = z[i]; for (int j = i + 1; j < n; ++j) t = ddk::sub(t, ddk::mul(ddk::dd{Lh[(size_t)j * ld + i], Ll[(size_t)j * ld + i]}, z[j])); z[i] = ddk::div(t, ddk::dd{Lh[(size_t)i * ld + i], Ll[(size_t)i * ld + i]}); }
  }
  // R: nrhs vectors [p][d] (fp64 in, fp64 out); substitutions in dd
  void solve(double* R, int nrhs) const {
    const size_t bs = (size_t)d * d;
    std::vector<ddk::dd> z((size_t)p * d);
    for (int q = 0; q < nrhs; ++q) {
      double* r = R + (size_t)q * p * d;
      for (size_t e = 0; e < z.size(); ++e) z[e] = ddk::from(r[e]);
      if (p <= 2) { const int N = p * d; fwd(Nh.data(), Nl.data(), N, N, z.data()); bwd(Nh.data(), Nl.data(), N, N, z.data()); }
      else {
        auto sub_t = [&](ddk::dd* y, const double* Th, const double* Tl, const ddk::dd* x) {      // y -= T' x  (T = Ot: y_i -= sum_q Ot[q][i] x_q)
          for (int qq = 0; qq < d; ++qq) for (int i = 0; i < d; ++i) y[i] = ddk::sub(y[i], ddk::mul(ddk::dd{Th[(size_t)qq * d + i], Tl[(size_t)qq * d + i]}, x[qq]));
        };
        auto sub_n = [&](ddk::dd* y, const double* Th, const double* Tl, const ddk::dd* x) {      // y -= T x   (y_q -= sum_i Ot[q][i] x_i)
          for (int qq = 0; qq < d; ++qq) { ddk::dd t = y[qq]; for (int i = 0; i < d; ++i) t = ddk::sub(t, ddk::mul(ddk::dd{Th[(size_t)qq * d + i], Tl[(size_t)qq * d + i]}, x[i])); y[qq] = t; }
        };
        for (int k = 0; k < p - 1; ++k) {
          fwd(&Dh[k * bs], &Dl[k * bs], d, d, &z[(size_t)k * d]);
          sub_t(&z[(size_t)(k + 1) * d], &Oh[k * bs], &Ol[k * bs], &z[(size_t)k * d]);
          if (k < p - 2) sub_t(&z[(size_t)(p - 1) * d], &Fh[k * bs], &Fl[k * bs], &z[(size_t)k * d]);
        }
        fwd(&Dh[(size_t)(p - 1) * bs], &Dl[(size_t)(p - 1) * bs], d, d, &z[(size_t)(p - 1) * d]);
        bwd(&Dh[(size_t)(p - 1) * bs], &Dl[(size_t)(p - 1) * bs], d, d, &z[(size_t)(p - 1) * d]);
        for (int k = p - 2; k >= 0; --k) {
          sub_n(&z[(size_t)k * d], &Oh[k * bs], &Ol[k * bs], &z[(size_t)(k + 1) * d]);
          if (k < p - 2) sub_n(&z[(size_t)k * d], &Fh[k * bs], &Fl[k * bs], &z[(size_t)(p - 1) * d]);
          bwd(&Dh[k * bs], &Dl[k * bs], d, d, &z[(size_t)k * d]);
        }
      }
      for (size_t e = 0; e < z.size(); ++e) r[e] = ddk::val(z[e]);
    }
  }
};

// blocks D_k, C_k in dd from the stage pairs (X_r, Z_r = S_r^-1), r = 1, 2, given as dd matrices (fp64 iterates: lo = 0)
static void assemble_dd(DdSys& sys, const Problem& pr, const std::vector<ddk::mat>& X1, const std::vector<ddk::mat>& Z1, const std::vector<ddk::mat>& X2,
                        const std::vector<ddk::mat>& Z2) {
  using namespace ddk;
  const int p = pr.p, nx = pr.nx, n = pr.n, d = pr.d, db = pr.db;      // db >= d: the P part sits in the leading d x d corner of every block
  sys.init(p, db, sys.par);
  const size_t bs = (size_t)db * db;
#pragma omp parallel for schedule(dynamic, 1) if (sys.par)
  for (int k = 0; k < p; ++k) {
    mat Vd((size_t)nx * n), Fx, Fs, Xe((size_t)nx * nx), Ze((size_t)nx * nx), Xr((size_t)nx * n), Zr((size_t)nx * n);
    for (int e = 0; e < nx * n; ++e) Vd[e] = from(pr.Vk(k)[e]);
    for (int r = 0; r < 2; ++r) {
      const mat& X = r ? X2[k] : X1[k]; const mat& Z = r ? Z2[k] : Z1[k];
      for (int a = 0; a < nx; ++a) for (int c = 0; c < n; ++c) { Xr[(size_t)a * n + c] = X[(size_t)a * n + c]; Zr[(size_t)a * n + c] = Z[(size_t)a * n + c]; }
      mm(Fx, Xr, Vd, nx, n, nx, true); mm(Fs, Zr, Vd, nx, n, nx, true);
      for (int a = 0; a < nx; ++a) for (int c = 0; c < nx; ++c) { Xe[a * nx + c] = X[(size_t)a * n + c]; Ze[a * nx + c] = Z[(size_t)a * n + c]; }
      DdSys::add_T(&sys.Dh[k * bs], &sys.Dl[k * bs], Xe, Ze, 1.0, nx, d, pr.ia, pr.ib, db);
      DdSys::add_T(&sys.Ch[k * bs], &sys.Cl[k * bs], Fx, Fs, -1.0, nx, d, pr.ia, pr.ib, db);
    }
  }
  // second sweep for the V-side terms (a different block than the loop index: no races)
  sys.Oh.assign((size_t)p * bs, 0.0); sys.Ol.assign((size_t)p * bs, 0.0);
#pragma omp parallel for schedule(dynamic, 1) if (sys.par)
  for (int k = 0; k < p; ++k) {
    mat Vd((size_t)nx * n), t, Kx, Ks;
    for (int e = 0; e < nx * n; ++e) Vd[e] = from(pr.Vk(k)[e]);
    for (int r = 0; r < 2; ++r) {
      const mat& X = r ? X2[k] : X1[k]; const mat& Z = r ? Z2[k] : Z1[k];
      mm(t, Vd, X, nx, n, n, false); mm(Kx, t, Vd, nx, n, nx, true);
      mm(t, Vd, Z, nx, n, n, false); mm(Ks, t, Vd, nx, n, nx, true);
      DdSys::add_T(&sys.Oh[k * bs], &sys.Ol[k * bs], Kx, Ks, 1.0, nx, d, pr.ia, pr.ib, db);
    }
  }
  for (int k = 0; k < p; ++k) {
    const int kn = (k + 1) % p;
    for (size_t e = 0; e < bs; ++e) { const dd v = add(dd{sys.Dh[kn * bs + e], sys.Dl[kn * bs + e]}, dd{sys.Oh[k * bs + e], sys.Ol[k * bs + e]}); sys.Dh[kn * bs + e] = v.h; sys.Dl[kn * bs + e] = v.l; }
  }
}

struct Result { double kappa, alpha; int status, iters, early; double mu_t = 0, stepn = 0; int dd_iters = 0, polish = 0, lowp = 0; };
constexpr double DD_SWITCH = 0x1p-23, POLISH_ENTER = 1e-4;      // as oracle/convexify_oracle.py
constexpr int POLISH_MAX = 6;

// status codes as include/tunempc_hip.h: 0 Optimal, 1 Feasible, 2 Infeasible
Result solve_problem(int p, int nx, int mb, const double* A, const double* B, const double* Hin, double tol, int max_iter, int center_iter,
                     double center_tol, double* Hc_out, bool tight = false, bool par = false) {
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
  bool dd_on = false, polish = false; int ndd = 0;      // tight mode (convexify_oracle.py: DD_SWITCH, _polish_dd)
  DdSys dsys; dsys.par = par;
  std::vector<ddk::mat> qX1, qZ1, qX2, qZ2;
  auto to_dd = [&](std::vector<ddk::mat>& out, const vec& src) { out.resize(p); for (int k = 0; k < p; ++k) { out[k].resize(nn); for (int e = 0; e < nn; ++e) out[k][e] = ddk::from(src[(size_t)k * nn + e]); } };
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
    if (tight && !dd_on && mu <= DD_SWITCH * std::max(1.0, fabs(tau))) dd_on = true;
    // ---- Schur complement blocks from the Kronecker factors
    std::fill(pr.D.begin(), pr.D.end(), 0.0); std::fill(pr.Csub.begin(), pr.Csub.end(), 0.0);
    for (int r = 0; r < 2 && !dd_on; ++r) {
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
    if (!dd_on) {
      { static const char* lps = getenv("CPU_IPM_LOWP_SWITCH"); static const double lpsw = lps ? atof(lps) : 0.0;      // EXPERIMENT hook (see Problem::lowp)
        static const bool lpsolve = getenv("CPU_IPM_LOWP_SOLVE") != nullptr; pr.lowp_solve = lpsolve;
        pr.lowp = lpsw > 0.0 && phase == 0 && mu > lpsw * std::max(1.0, fabs(tau)); if (pr.lowp) ++res.lowp; }
      if (!pr.factor()) { ipm = ST_INACC; break; }
      if (pr.lowp && pr.shift > 0.0) { pr.lowp = false; res.lowp += 1000; if (!pr.factor()) { ipm = ST_INACC; break; } }      // (an fp32 update that costs a pivot: once more in fp64)
      if (tight && pr.shift > 0.0) dd_on = true;
    }
    if (dd_on) {
      ++ndd;
      to_dd(qX1, X1); to_dd(qZ1, S1i); to_dd(qX2, X2); to_dd(qZ2, S2i);
      assemble_dd(dsys, pr, qX1, qZ1, qX2, qZ2);
      if (!dsys.factor()) { ipm = ST_INACC; break; }
      pr.shift = 0.0;
    }
    auto bsolve = [&](double* R, int nrhs) { if (dd_on) dsys.solve(R, nrhs); else pr.solve(R, nrhs); };
    nshiftrun = pr.shift > 0.0 ? nshiftrun + 1 : 0;
    if (phase == 1 && pr.shift > 0.0 && nbackoff < MUT_BACKOFF_MAX) {
      // hard target: aim one power of two earlier and take the step of the shifted factorisation towards it (convexify_oracle.py, k_ctrl_b)
      mu_t *= 2.0; ++nbackoff; ncent = 0; prev_stepn = -1.0; nshiftrun = 0;
    } else if ((phase == 1 && pr.shift > 0.0) || (nshiftrun >= 2 && (mu_t < 0.0 || nbackoff >= MUT_BACKOFF_MAX))) { ipm = ST_INACC; break; }
    TU = U; bsolve(TU.data(), 2);
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
      bsolve(zs.data(), 1);
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
      if (const char* ev_ = getenv("CPU_IPM_SIGMA_EXP")) {      // experiment hook (tests/tools/sigma_rule_probe.py): other centering rules, iteration counts only
        const double e_ = atof(ev_);
        if (e_ > 0.0) sigma = std::min(std::max(pow(rat, e_), 1e-6), 1.0);
        else { const double am_ = std::min(ap, ad); const double ee_ = std::max(1.0, 3.0 * am_ * am_); sigma = std::min(std::max(pow(rat, ee_), 1e-6), 1.0); }      // SDPT3's adaptive exponent
      }
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
      double gam = 0.9 + 0.09 * std::min(std::min(ap, ad), 1.0);
      if (const char* ev_ = getenv("CPU_IPM_GAMMA_MAX")) gam = 0.9 + (atof(ev_) - 0.9) * std::min(std::min(ap, ad), 1.0);
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
    // dual-Newton polish in double-double (convexify_oracle._polish_dd): every stage quantity from the fp64 y = (tau, alpha, P)
    using namespace ddk;
    ipm = ST_INACC;
    std::vector<mat> Mq(p), Z1q(p), Z2q(p), X1q(p), X2q(p), W1(p), W2(p), W3(p);
    auto cones = [&](double tau_, double alpha_, const vec& P_) -> bool {
      if (!(alpha_ - ALPHA_MIN > 0.0)) return false;
      bool okc = true;
#pragma omp parallel for schedule(dynamic, 1) if (par)
      for (int k = 0; k < p; ++k) {
        const int kn = (k + 1) % p;
        mat Vd((size_t)nx * n), Pn((size_t)nx * nx), t, M, S((size_t)nn);
        for (int e = 0; e < nx * n; ++e) Vd[e] = from(pr.Vk(k)[e]);
        for (int e = 0; e < nxx; ++e) Pn[e] = from(P_[(size_t)kn * nxx + e]);
        mat Vt((size_t)n * nx); for (int a = 0; a < nx; ++a) for (int c = 0; c < n; ++c) Vt[(size_t)c * nx + a] = Vd[(size_t)a * n + c];
        mm(t, Vt, Pn, n, nx, nx, false); mm(M, t, Vd, n, nx, n, false);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
          dd v = add(M[(size_t)i * n + j], muld(from(Hb[(size_t)k * nn + i * n + j]), alpha_));
          if (i < nx && j < nx) v = sub(v, from(P_[(size_t)k * nxx + i * nx + j]));
          M[(size_t)i * n + j] = v;
        }
        Mq[k] = M;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) S[(size_t)i * n + j] = (i == j) ? sub(M[(size_t)i * n + j], from(1.0)) : M[(size_t)i * n + j];
        if (!inv_spd(Z1q[k], S, n)) { okc = false; continue; }
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) S[(size_t)i * n + j] = (i == j) ? sub(from(tau_), M[(size_t)i * n + j]) : neg(M[(size_t)i * n + j]);
        if (!inv_spd(Z2q[k], S, n)) okc = false;
      }
      return okc;
    };
    // adjoint in dd, then svec (weights 1 / 2) rounded to fp64:  out[j] = svec(V_{j-1} G_{j-1} V_{j-1}' - G_j[:nx,:nx])
    auto adj_sv = [&](double* out, const std::vector<mat>& G, double sg) {
      std::vector<mat> W(p);
#pragma omp parallel for schedule(dynamic, 1) if (par)
      for (int k = 0; k < p; ++k) {
        mat Vd((size_t)nx * n), t;
        for (int e = 0; e < nx * n; ++e) Vd[e] = from(pr.Vk(k)[e]);
        mm(t, Vd, G[k], nx, n, n, false); mm(W[k], t, Vd, nx, n, nx, true);
      }
      for (int j = 0; j < p; ++j) {
        const int jm = (j + p - 1) % p;
        for (int e = 0; e < d; ++e) {
          const int a = pr.ia[e], c = pr.ib[e];
          const dd v = sub(W[jm][a * nx + c], G[j][(size_t)a * n + c]);
          out[(size_t)j * d + e] = sg * val(v) * (a == c ? 1.0 : 2.0);
        }
      }
    };
    bool okp = cones(tau, alpha, P);
    vec Pn_(P.size()), Mf(PN);
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
        for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) {       // symmetrise (exact up to rounding already)
          auto sy = [&](mat& G) { const dd v = muld(add(G[(size_t)i * n + j], G[(size_t)j * n + i]), 0.5); G[(size_t)i * n + j] = v; G[(size_t)j * n + i] = v; };
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
      if (!dsys.factor()) break;
      adj_sv(&U[0], Psi, -1.0); adj_sv(&U[(size_t)p * d], Ph, 1.0); adj_sv(zs.data(), Yq, 1.0);
      TU = U; dsys.solve(TU.data(), 2); dsys.solve(zs.data(), 1);
      double sb00 = b_tt, sb01 = b_ta, sb11 = b_aa, u0 = 0, u1 = 0;
      for (size_t e = 0; e < (size_t)p * d; ++e) {
        sb00 -= U[e] * TU[e]; sb01 -= U[e] * TU[(size_t)p * d + e]; sb11 -= U[(size_t)p * d + e] * TU[(size_t)p * d + e];
        u0 += U[e] * zs[e]; u1 += U[(size_t)p * d + e] * zs[e];
      }
      const double rb0 = -g_tau - u0, rb1 = -g_alpha - u1, det = sb00 * sb11 - sb01 * sb01;
      dtau = (sb11 * rb0 - sb01 * rb1) / det; dalpha = (sb00 * rb1 - sb01 * rb0) / det;
      for (int k = 0; k < p; ++k) for (int e = 0; e < d; ++e) {
        const double v = zs[(size_t)k * d + e] - TU[(size_t)k * d + e] * dtau - TU[(size_t)(p + k) * d + e] * dalpha;
        dP[(size_t)k * nxx + pr.ia[e] * nx + pr.ib[e]] = v; dP[(size_t)k * nxx + pr.ib[e] * nx + pr.ia[e]] = v;
      }
      pr.calH(dM, dP, dalpha);
      double num = 0, den = 0; const double ra = dalpha / alpha;
      for (int k = 0; k < p; ++k) for (int e = 0; e < nn; ++e) { const double mv = val(Mq[k][e]); const double dh = dM[(size_t)k * nn + e] - ra * mv; num += dh * dh; den += mv * mv; }
      stepn = sqrt(num / den);
      double th = 1.0;
      for (;;) {
        for (size_t e = 0; e < P.size(); ++e) Pn_[e] = P[e] + th * dP[e];
        if (cones(tau + th * dtau, alpha + th * dalpha, Pn_)) break;
        th *= 0.5;
        if (th < 1e-3) { okp = false; break; }
      }
      if (!okp) break;
      tau += th * dtau; alpha += th * dalpha; P = Pn_;
      if (th == 1.0 && stepn < center_tol) { ipm = ST_OPT; break; }
    }
    if (npolish > POLISH_MAX) npolish = POLISH_MAX;
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
  res.kappa = tau; res.alpha = alpha; res.iters = it; res.mu_t = mu_t; res.stepn = stepn; res.dd_iters = ndd; res.polish = npolish;
  res.status = (lo2 > 0.0) ? (ipm == ST_OPT ? 0 : 1) : 2;
  return res;
}

#include "cpu_ipm_con.h"

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

// The same with the tight-accuracy mode (tight != 0: block linear algebra in double-double below DD_SWITCH, dual-Newton polish at the end;
// tol down to 2^-41).  A batch smaller than the thread count runs its problems one after the other with the threads inside the block
// kernels.  info [nb][4]: mu_t, dd iterations, polish steps, last relative step.
int cpu_ipm_convexify_batch2(int nb, int p, int nx, int mb, const double* A, const double* B, const double* H, double tol, int threads, int tight,
                             double* Hc, double* kappa, int32_t* status, int32_t* iters, double* info) {
  if (nb < 0 || p < 1 || nx < 1 || mb < 0 || !A || !H || !Hc) return -1;
  const int n = nx + mb;
  if (tol <= 0.0) tol = 0x1p-25;
  const int blas_threads_before = scipy_openblas_get_num_threads();
  scipy_openblas_set_num_threads(1);
  if (threads < 1) threads = 1;
  const bool inner = tight && nb < threads;
  omp_set_max_active_levels(1);
  const int saved = omp_get_max_threads();
  omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) if (!inner)
  for (int b = 0; b < nb; ++b) {
    const Result r = solve_problem(p, nx, mb, A + (size_t)b * p * nx * nx, B + (size_t)b * p * nx * mb, H + (size_t)b * p * n * n, tol, 50, 12, 1e-9,
                                   Hc + (size_t)b * p * n * n, tight != 0, inner);
    if (kappa) kappa[b] = r.kappa;
    if (status) status[b] = r.status;
    if (iters) iters[b] = r.iters;
    if (info) { info[4 * b] = r.mu_t; info[4 * b + 1] = tight ? r.dd_iters : r.lowp; info[4 * b + 2] = r.polish; info[4 * b + 3] = r.stepn; }
  }
  omp_set_num_threads(saved);
  scipy_openblas_set_num_threads(blas_threads_before);
  return 0;
}

// The models with rows (cpu_ipm_con.h): Step 1 with G, Step 2 (either objective), Step 3, Step 3 with rows.
// J [nb][p][ng0 + ncmax][n]: rows of G_k, then the rows of C_k padded to ncmax; ncnt [nb][p] active rows of C_k (NULL: all ncmax).
// flags: 1 constr (Step 2: the rows of C_k take part, norm terms with weight rho), 2 cost_free (beta-only objective), 4 force (Step 3).
// Outputs: Hc [nb][p][n][n], P [nb][p][nx][nx], FgF [nb][p][ng0 + ncmax] (padding zero), T [nb][p][n][n] (may be NULL), kappa, objective, status, iters.
int cpu_ipm_convexify_con_batch(int nb, int p, int nx, int mb, int ng0, int ncmax, const double* A, const double* B, const double* H, const double* J,
                                const int32_t* ncnt, double rho, int flags, double tol, int threads, double* Hc, double* P, double* FgF, double* T,
                                double* kappa, double* objective, int32_t* status, int32_t* iters) {
  if (nb < 0 || p < 1 || nx < 1 || mb < 0 || ng0 < 0 || ncmax < 0 || !A || !H || !Hc || ((ng0 + ncmax) > 0 && !J)) return -1;
  const int n = nx + mb, nJ = ng0 + ncmax;
  if (tol <= 0.0) tol = 0x1p-25;
  const int blas_threads_before = scipy_openblas_get_num_threads();
  scipy_openblas_set_num_threads(1);
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
  for (int b = 0; b < nb; ++b) {
    ConIn ci; ci.ng0 = ng0; ci.ncmax = ncmax; ci.J = J ? J + (size_t)b * p * nJ * n : nullptr; ci.ncnt = ncnt ? ncnt + (size_t)b * p : nullptr;
    ci.rho = rho; ci.constr = (flags & 1) != 0; ci.cost_free = (flags & 2) != 0; ci.force = (flags & 4) != 0;
    ConOut co; co.P = P ? P + (size_t)b * p * nx * nx : nullptr; co.FgF = FgF ? FgF + (size_t)b * p * nJ : nullptr; co.T = T ? T + (size_t)b * p * n * n : nullptr;
    const Result r = solve_problem_con(p, nx, mb, A + (size_t)b * p * nx * nx, B + (size_t)b * p * nx * mb, H + (size_t)b * p * n * n, ci, tol, 50, 12, 1e-9,
                                       Hc + (size_t)b * p * n * n, co);
    if (kappa) kappa[b] = r.kappa;
    if (objective) objective[b] = co.objective;
    if (status) status[b] = r.status;
    if (iters) iters[b] = r.iters;
  }
  scipy_openblas_set_num_threads(blas_threads_before);
  return 0;
}

// The same with the tight-accuracy mode for Step 1 with G and Step 2 (tight != 0; Step 3 in this mode: the numpy oracle only): double-double block linear algebra with the
// stage-local rows in double-double below DD_SWITCH, dd dual-Newton polish in (tau, alpha, P, phi, t).  info [nb][4]: mu_t, dd iterations, polish steps, last relative step.
int cpu_ipm_convexify_con_batch2(int nb, int p, int nx, int mb, int ng0, int ncmax, const double* A, const double* B, const double* H, const double* J,
                                 const int32_t* ncnt, double rho, int flags, double tol, int threads, int tight, double* Hc, double* P, double* FgF, double* T,
                                 double* kappa, double* objective, int32_t* status, int32_t* iters, double* info) {
  if (nb < 0 || p < 1 || nx < 1 || mb < 0 || ng0 < 0 || ncmax < 0 || !A || !H || !Hc || ((ng0 + ncmax) > 0 && !J)) return -1;
  if (tight && (flags & 4)) return -2;      // Step 3 in the tight mode: not in this port
  const int n = nx + mb, nJ = ng0 + ncmax;
  if (tol <= 0.0) tol = 0x1p-25;
  const int blas_threads_before = scipy_openblas_get_num_threads();
  scipy_openblas_set_num_threads(1);
  if (threads < 1) threads = 1;
  const bool inner = tight && nb < threads;
  omp_set_max_active_levels(1);
  const int saved = omp_get_max_threads();
  omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) if (!inner)
  for (int b = 0; b < nb; ++b) {
    ConIn ci; ci.ng0 = ng0; ci.ncmax = ncmax; ci.J = J ? J + (size_t)b * p * nJ * n : nullptr; ci.ncnt = ncnt ? ncnt + (size_t)b * p : nullptr;
    ci.rho = rho; ci.constr = (flags & 1) != 0; ci.cost_free = (flags & 2) != 0; ci.force = (flags & 4) != 0;
    ConOut co; co.P = P ? P + (size_t)b * p * nx * nx : nullptr; co.FgF = FgF ? FgF + (size_t)b * p * nJ : nullptr; co.T = T ? T + (size_t)b * p * n * n : nullptr;
    const Result r = solve_problem_con(p, nx, mb, A + (size_t)b * p * nx * nx, B + (size_t)b * p * nx * mb, H + (size_t)b * p * n * n, ci, tol, 50, 12, 1e-9,
                                       Hc + (size_t)b * p * n * n, co, tight != 0, inner);
    if (kappa) kappa[b] = r.kappa;
    if (objective) objective[b] = co.objective;
    if (status) status[b] = r.status;
    if (iters) iters[b] = r.iters;
    if (info) { info[4 * b] = r.mu_t; info[4 * b + 1] = r.dd_iters; info[4 * b + 2] = r.polish; info[4 * b + 3] = r.stepn; }
  }
  omp_set_num_threads(saved);
  scipy_openblas_set_num_threads(blas_threads_before);
  return 0;
}

int cpu_ipm_max_threads(void) { return omp_get_max_threads(); }

}  // extern "C"
