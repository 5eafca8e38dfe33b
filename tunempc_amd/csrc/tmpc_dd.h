// Tight-accuracy mode: the block linear algebra of the interior-point iteration in DOUBLE-DOUBLE, and a dd dual-Newton polish.
//
// Why.  The HKM Schur matrix B = L'(X (x) S^-1)L of the convexifier SDP (convexifier.py:213-308) has eigenvalues ~1/mu on the
// active x active part of the cone blocks and ~mu on directions inside the optimal face (the SDP minimises beta alone: its minimiser
// is a face), i.e. cond(B) ~ 1/mu^2.  Assembled and factored in fp64 it loses the small end at mu ~ sqrt(eps) = 1e-8, which is why
// the default solve stops at mu_t = 2^-25 kappa (a certified gap of N * 3e-8 on kappa).  Measured on the CPU restatement
// (tests/tools/tight_probe.py, 40-digit arithmetic in exactly these pieces): with the Kronecker-factor images, the assembly, the
// factorisation and the substitutions carried in extended precision -- and everything else left in fp64 -- the same iteration
// follows the path to mu ~ 1e-12.  Double-double (hi + lo, eps ~ 1e-32) is enough for that and needs no hardware beyond the fp64 FMA.
//
// What.  After the default solve (unchanged), problems that ended Optimal are restarted from their centred point with the target
// tight_tol * kappa: the same predictor-corrector loop, the same fp64 stage kernels, but k_schur / cr_factor / cr_solve replaced by
//   k_dd_images     V X V', V S^-1 V', X_E V', S^-1_E V' of both cone blocks in dd (their fp64 rounding alone, eps / mu, is the wall)
//   k_dd_schur      D_k, C_k in dd from those factors (same storage and orientation as k_schur)
//   k_dd_potrf / k_dd_trsm / k_dd_update     the cyclic-reduction block Cholesky of tmpc_cr.h (same schedule) on dd planes
//   k_dd_fwd_diag / k_dd_fwd_off / k_dd_bwd  the substitutions, right-hand sides carried in dd between the levels
// and the centering phase hands over (first full step below POLISH_ENTER) to Newton's method on the DUAL barrier problem in
// y = (tau, alpha, P) with every stage quantity in dd (k_dd_polish_pre): the primal-dual iteration keeps X and S^-1 as fp64
// matrices whose large part buries the small one under an absolute rounding error, so its centred point is reproducible to
// ~eps/mu only (1e-7 at mu = 2e-12); two or three polish steps reproduce it to 1e-12 (the CPU restatement used by the tests does the same).
//
// The dd kernels run on the vector ALU (the matrix cores have no extended format; emulating dd products on fp64 MFMA by
// Ozaki slicing costs as many issue slots as the VALU form because fp64 MFMA and VALU share them on this part): ~12 VALU
// operations per dd multiply-add, 64 x 64 output tiles, 4 x 4 register micro-tiles, K slabs of 16 through LDS.  Round 5: also the models with rows (Step 1 with G, Step 2) -- k_dd_aug_fill, k_dd_solve_border, k_polish_phi, k_polish_arrows below; not Step 3.
#pragma once
#include "tmpc_common.h"
#include "tmpc_small.h"
#include "tmpc_cr.h"

// The error-free transformations below (two_sum, two_prod) are exact only if every operation is rounded on its own: the AMDGPU back end fuses a
// multiply into a following add even when the product has other uses, which turns s = sh + p into fma(ah, bh, sh) and the "error" of the
// sum into garbage of size eps |p| (measured: non-positive pivots at mu ~ 1e-10 where the CPU restatement factors at 1e-12).
#pragma clang fp contract(off)

namespace tmpc {

constexpr int DD_SCR_MATS = 9;            // dd matrices per stage of the global scratch (32 < n <= 64; k_dd_polish_pre needs nine)
constexpr double POLISH_ENTER = 1e-4;    // as the CPU restatement used by the tests
constexpr double TIGHT_CHORD_STEP = 3.0;   // chord threshold of the tight phase (run_chunk): the centering phase re-uses a double-double factorisation once a full step could have been 3 x longer
constexpr int POLISH_MAX = 10;     // steps of the polish, chord steps included (round 5; 6 Newton steps before)

// ------------------------------------------------------------------ dd scalar arithmetic (Dekker / Knuth / QD)
struct ddv { double h, l; };
__device__ __forceinline__ ddv dd_qts(double a, double b) { const double s = a + b; return {s, b - (s - a)}; }
__device__ __forceinline__ ddv dd_ts(double a, double b) { const double s = a + b, v = s - a; return {s, (a - (s - v)) + (b - v)}; }
__device__ __forceinline__ ddv dd_tp(double a, double b) { const double p = a * b; return {p, fma(a, b, -p)}; }
__device__ __forceinline__ ddv dd_add(ddv a, ddv b) { ddv s = dd_ts(a.h, b.h); const ddv t = dd_ts(a.l, b.l); s.l += t.h; s = dd_qts(s.h, s.l); s.l += t.l; return dd_qts(s.h, s.l); }
__device__ __forceinline__ ddv dd_neg(ddv a) { return {-a.h, -a.l}; }
__device__ __forceinline__ ddv dd_sub(ddv a, ddv b) { return dd_add(a, dd_neg(b)); }
__device__ __forceinline__ ddv dd_mul(ddv a, ddv b) { ddv p = dd_tp(a.h, b.h); p.l += fma(a.h, b.l, a.l * b.h); return dd_qts(p.h, p.l); }
__device__ __forceinline__ ddv dd_muld(ddv a, double b) { ddv p = dd_tp(a.h, b); p.l = fma(a.l, b, p.l); return dd_qts(p.h, p.l); }
__device__ __forceinline__ ddv dd_from(double a) { return {a, 0.0}; }
__device__ __forceinline__ double dd_val(ddv a) { return a.h + a.l; }
__device__ __forceinline__ ddv dd_div(ddv a, ddv b) {
  const double q1 = a.h / b.h; ddv r = dd_sub(a, dd_muld(b, q1));
  const double q2 = r.h / b.h; r = dd_sub(r, dd_muld(b, q2));
  const double q3 = r.h / b.h;
  return dd_add(dd_qts(q1, q2), dd_from(q3));
}
__device__ __forceinline__ ddv dd_sqrt(ddv a) { const double x = 1.0 / sqrt(a.h), ax = a.h * x; const double err = dd_sub(a, dd_tp(ax, ax)).h; return dd_qts(ax, err * x * 0.5); }
// (sh, sl) += a * b with a deferred renormalisation: the two_sum of the high words is exact, the low word just collects (12 operations)
__device__ __forceinline__ void dd_fma_acc(double& sh, double& sl, double ah, double al, double bh, double bl) {
  const double p = ah * bh;
  double e = fma(ah, bh, -p); e = fma(ah, bl, e); e = fma(al, bh, e);
  const double s = sh + p, v = s - sh;
  const double t = (sh - (s - v)) + (p - v);
  sl += t + e; sh = s;
}
__device__ __forceinline__ double* dd_edge_lo(const WS& w, const Dims& dm, int b, int slot) {
  const size_t bs = (size_t)dm.dp * dm.dp;
  return (slot < dm.p) ? w.Ol + ((size_t)b * dm.p + slot) * bs : w.Fl + ((size_t)b * dm.p + (slot - dm.p)) * bs;
}

// ------------------------------------------------------------------ tile GEMM  C (M x N, N <= 64) <op> A (M x K) B (N x K)'
// 256 threads; thread (ty, tx) = (tid >> 4, tid & 15) owns rows 4 ty .. + 3, columns 4 tx .. + 3 of the 64-row tile.  K slabs of 16 sit in
// LDS transposed ([k][row], so a thread's four rows are two 16-byte reads).  TA: A is given transposed (element (m, k) at A[k * lda + m]).
// M, N, K need not be multiples of anything: loads outside read as zero, stores are masked.  In place (C aliasing A) is safe when the
// call has one N tile and K == its width: all slabs of a row tile are read before its C is written, and other row tiles touch other rows.
constexpr int DG_K = 16, DG_LD = 68, DG_SLAB = DG_K * DG_LD;
constexpr int DD_GEMM_LDS = 4 * DG_SLAB;          // doubles: A hi / lo, B hi / lo
template <bool TA>
__device__ __forceinline__ void wg_gemm_dd(double* Ch, double* Cl, int ldc, const double* Ah, const double* Al, int lda, const double* Bh, const double* Bl, int ldb,
                                           int M, int N, int K, int mode, double* lds) {
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  double* sAh = lds; double* sAl = lds + DG_SLAB; double* sBh = lds + 2 * DG_SLAB; double* sBl = lds + 3 * DG_SLAB;
  for (int m0 = 0; m0 < M; m0 += 64) {
    double ah_[4][4], al_[4][4];                     // accumulators [row][col]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { ah_[i][j] = 0.0; al_[i][j] = 0.0; }
    for (int k0 = 0; k0 < K; k0 += DG_K) {
      __syncthreads();                               // previous slab consumed
      if (TA) {
        const int k = tid >> 4, mq = (tid & 15) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool ok = (k0 + k < K) && (m0 + mq + q < M);
          sAh[k * DG_LD + mq + q] = ok ? Ah[(size_t)(k0 + k) * lda + m0 + mq + q] : 0.0;
          sAl[k * DG_LD + mq + q] = ok ? Al[(size_t)(k0 + k) * lda + m0 + mq + q] : 0.0;
        }
      } else {
        const int r = tid >> 2, kq = (tid & 3) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool ok = (m0 + r < M) && (k0 + kq + q < K);
          sAh[(kq + q) * DG_LD + r] = ok ? Ah[(size_t)(m0 + r) * lda + k0 + kq + q] : 0.0;
          sAl[(kq + q) * DG_LD + r] = ok ? Al[(size_t)(m0 + r) * lda + k0 + kq + q] : 0.0;
        }
      }
      {
        const int r = tid >> 2, kq = (tid & 3) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool ok = (r < N) && (k0 + kq + q < K);
          sBh[(kq + q) * DG_LD + r] = ok ? Bh[(size_t)r * ldb + k0 + kq + q] : 0.0;
          sBl[(kq + q) * DG_LD + r] = ok ? Bl[(size_t)r * ldb + k0 + kq + q] : 0.0;
        }
      }
      __syncthreads();
      if (m0 + 4 * ty < M && 4 * tx < N) {
#pragma unroll 4
        for (int k = 0; k < DG_K; ++k) {
          double xh[4], xl[4], yh[4], yl[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) { xh[i] = sAh[k * DG_LD + 4 * ty + i]; xl[i] = sAl[k * DG_LD + 4 * ty + i]; yh[i] = sBh[k * DG_LD + 4 * tx + i]; yl[i] = sBl[k * DG_LD + 4 * tx + i]; }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) dd_fma_acc(ah_[i][j], al_[i][j], xh[i], xl[i], yh[j], yl[j]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + 4 * ty + i;
      if (row >= M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = 4 * tx + j;
        if (col >= N) continue;
        const ddv acc = dd_qts(ah_[i][j], al_[i][j]);
        const size_t e = (size_t)row * ldc + col;
        ddv v;
        if (mode == GM_SUB) v = dd_sub(ddv{Ch[e], Cl[e]}, acc);
        else if (mode == GM_NEG) v = dd_neg(acc);
        else v = acc;
        Ch[e] = v.h; Cl[e] = v.l;
      }
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------ 64 x 64 tile Cholesky + inverse in dd
// The tile lives in LDS (hi / lo, leading dimension 65).  Right-looking column steps; then the inverse of the factor, column c by thread c
// (forward substitution), parked in the strict upper triangle of the same image (L^-1[i][c] at [c][i]; its diagonal in dinv).  Returns the
// number of non-positive pivots (0 = fine; the caller stops the problem otherwise: no shift in this mode).
constexpr int DD_POTRF_LDS = 2 * 64 * 65 + 128 + 8;
__device__ __forceinline__ int wg_potrf_inv_dd(double* Th, double* Tl, int ldt, double* Tih, double* Til, int nb, double* lds) {
  const int tid = threadIdx.x;
  double* Sh = lds; double* Sl = lds + 64 * 65; double* dih = Sl + 64 * 65; double* dil = dih + 64; double* flag = dil + 64;
  __syncthreads();
  for (int e = tid; e < nb * nb; e += 256) { const int i = e / nb, j = e - i * nb; Sh[i * 65 + j] = Th[(size_t)i * ldt + j]; Sl[i * 65 + j] = Tl[(size_t)i * ldt + j]; }
  if (tid == 0) flag[0] = 0.0;
  __syncthreads();
  for (int j = 0; j < nb; ++j) {
    ddv piv{Sh[j * 65 + j], Sl[j * 65 + j]};
    if (!(piv.h > 0.0)) { piv = ddv{1.0, 0.0}; if (tid == 0) flag[0] += 1.0; }      // (uniform: every thread reads the same pivot)
    const ddv r = dd_sqrt(piv);
    const ddv rinv = dd_div(dd_from(1.0), r);
    __syncthreads();
    if (tid == 0) { Sh[j * 65 + j] = r.h; Sl[j * 65 + j] = r.l; dih[j] = rinv.h; dil[j] = rinv.l; }
    if (tid > j && tid < nb) { const ddv v = dd_mul(ddv{Sh[tid * 65 + j], Sl[tid * 65 + j]}, rinv); Sh[tid * 65 + j] = v.h; Sl[tid * 65 + j] = v.l; }
    __syncthreads();
    for (int i = j + 1 + (tid >> 2); i < nb; i += 64) {
      const ddv li{Sh[i * 65 + j], Sl[i * 65 + j]};
      for (int k = j + 1 + (tid & 3); k <= i; k += 4) {
        const ddv v = dd_sub(ddv{Sh[i * 65 + k], Sl[i * 65 + k]}, dd_mul(li, ddv{Sh[k * 65 + j], Sl[k * 65 + j]}));
        Sh[i * 65 + k] = v.h; Sl[i * 65 + k] = v.l;
      }
    }
    __syncthreads();
  }
  if (tid < nb) {                     // column c of L^-1:  x_i = -(sum_{k=c}^{i-1} L[i][k] x_k) / L[i][i],  x_c = 1 / L[c][c]
    const int c = tid;
    for (int i = c + 1; i < nb; ++i) {
      ddv s = dd_mul(ddv{Sh[i * 65 + c], Sl[i * 65 + c]}, ddv{dih[c], dil[c]});
      for (int k = c + 1; k < i; ++k) s = dd_add(s, dd_mul(ddv{Sh[i * 65 + k], Sl[i * 65 + k]}, ddv{Sh[c * 65 + k], Sl[c * 65 + k]}));
      const ddv x = dd_neg(dd_mul(s, ddv{dih[i], dil[i]}));
      Sh[c * 65 + i] = x.h; Sl[c * 65 + i] = x.l;
    }
  }
  __syncthreads();
  for (int e = tid; e < nb * nb; e += 256) {
    const int i = e / nb, j = e - i * nb;
    if (j <= i) { Th[(size_t)i * ldt + j] = Sh[i * 65 + j]; Tl[(size_t)i * ldt + j] = Sl[i * 65 + j]; }
    double vh = 0.0, vl = 0.0;
    if (j == i) { vh = dih[i]; vl = dil[i]; } else if (j < i) { vh = Sh[j * 65 + i]; vl = Sl[j * 65 + i]; }
    Tih[i * TB + j] = vh; Til[i * TB + j] = vl;
  }
  __syncthreads();
  return (int)flag[0];
}

// left-looking blocked Cholesky of one dp x dp diagonal block (wg_block_column of tmpc_factor.h in dd); inverted diagonal tiles into Li
__device__ __forceinline__ int wg_block_potrf_dd(double* Dh, double* Dl, double* Lih, double* Lil, int dp, double* lds) {
  int nbad = 0, jt = 0;
  for (int j0 = 0; j0 < dp; j0 += TB, ++jt) {
    const int nb = (dp - j0 < TB) ? dp - j0 : TB;
    if (j0 > 0)
      wg_gemm_dd<false>(Dh + (size_t)j0 * dp + j0, Dl + (size_t)j0 * dp + j0, dp, Dh + (size_t)j0 * dp, Dl + (size_t)j0 * dp, dp, Dh + (size_t)j0 * dp, Dl + (size_t)j0 * dp, dp,
                        dp - j0, nb, j0, GM_SUB, lds);
    double* Tih = Lih + (size_t)jt * TB * TB; double* Til = Lil + (size_t)jt * TB * TB;
    nbad += wg_potrf_inv_dd(Dh + (size_t)j0 * dp + j0, Dl + (size_t)j0 * dp + j0, dp, Tih, Til, nb, lds);
    if (dp - j0 - nb > 0) {
      double* Xh = Dh + (size_t)(j0 + nb) * dp + j0; double* Xl = Dl + (size_t)(j0 + nb) * dp + j0;
      wg_gemm_dd<false>(Xh, Xl, dp, Xh, Xl, dp, Tih, Til, TB, dp - j0 - nb, nb, nb, GM_SET, lds);
    }
  }
  return nbad;
}

// ------------------------------------------------------------------ factorisation kernels (schedule and storage of tmpc_cr.h)
constexpr int DD_FACT_LDS = (DD_POTRF_LDS > DD_GEMM_LDS) ? DD_POTRF_LDS : DD_GEMM_LDS;

__global__ void __launch_bounds__(256) k_dd_prep(WS w, Dims dm, CrDev cr, int prep) {
  const int b = cr.alist[blockIdx.x];
  const int dp = dm.dp, tid = threadIdx.x;
  const size_t o = (size_t)b * dm.p * dp * dp;
  if (prep == 1) {
    for (int e = tid; e < dp * dp; e += 256) {
      const int i = e / dp, j = e - i * dp;
      const size_t et = (size_t)j * dp + i;
      ddv v = dd_add(ddv{w.D[o + e], w.Dl[o + e]}, dd_add(ddv{w.O[o + e], w.Ol[o + e]}, ddv{w.O[o + et], w.Ol[o + et]}));
      if (j > i) v = ddv{0.0, 0.0};                      // (the assembly wrote the lower triangle of D only)
      w.D[o + e] = v.h; w.Dl[o + e] = v.l;
    }
  } else {
    const size_t o1 = o + (size_t)dp * dp;
    for (int e = tid; e < dp * dp; e += 256) { const ddv v = dd_add(ddv{w.O[o + e], w.Ol[o + e]}, ddv{w.O[o1 + e], w.Ol[o1 + e]}); w.O[o + e] = v.h; w.Ol[o + e] = v.l; }
  }
}

__global__ void __launch_bounds__(256, 2) k_dd_potrf(WS w, Dims dm, CrDev cr, int eoff, int nelim, int count) {
  const int it = cr_item(count * nelim);
  if (it < 0) return;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int b = cr.alist[it / nelim];
  const int* er = cr.elim + (size_t)(eoff + it % nelim) * CR_EW;
  const int node = er[CE_NODE], dp = dm.dp;
  const size_t bs = (size_t)dp * dp, no = ((size_t)b * dm.p + node);
  const int nbad = wg_block_potrf_dd(w.D + no * bs, w.Dl + no * bs, w.Linv + no * dm.nt * TB * TB, w.Linvl + no * dm.nt * TB * TB, dp, lds);
  if (threadIdx.x == 0 && nbad) atomicAdd(w.iprob + (size_t)b * IS + I_NSHIFT, nbad);
}

// O_x <- T[x, i] L_i^-T for the two neighbours of an eliminated node, one workgroup per 64-row strip
__global__ void __launch_bounds__(256, 2) k_dd_trsm(WS w, Dims dm, CrDev cr, int eoff, int nelim, int count) {
  const int nstrip = (dm.dp + 63) / 64, per = 2 * nstrip;
  const int it = cr_item(count * nelim * per);
  if (it < 0) return;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int g = it / per, r = it - g * per;
  const int b = cr.alist[g / nelim];
  const int* er = cr.elim + (size_t)(eoff + g % nelim) * CR_EW;
  const int which = r / nstrip, strip = r - which * nstrip;
  const int slot = which ? er[CE_EB] : er[CE_EA];
  if (slot < 0) return;
  const int node = er[CE_NODE], dp = dm.dp;
  const size_t bs = (size_t)dp * dp, no = ((size_t)b * dm.p + node);
  const double* Dh = w.D + no * bs; const double* Dl = w.Dl + no * bs;
  const double* Lih = w.Linv + no * dm.nt * TB * TB; const double* Lil = w.Linvl + no * dm.nt * TB * TB;
  const int r0 = strip * 64, rows = (dp - r0 < 64) ? dp - r0 : 64;
  double* Xh = cr_edge(w, dm, b, slot) + (size_t)r0 * dp; double* Xl = dd_edge_lo(w, dm, b, slot) + (size_t)r0 * dp;
  int jt = 0;
  for (int j0 = 0; j0 < dp; j0 += TB, ++jt) {
    const int nb = (dp - j0 < TB) ? dp - j0 : TB;
    if (j0 > 0) wg_gemm_dd<false>(Xh + j0, Xl + j0, dp, Xh, Xl, dp, Dh + (size_t)j0 * dp, Dl + (size_t)j0 * dp, dp, rows, nb, j0, GM_SUB, lds);
    wg_gemm_dd<false>(Xh + j0, Xl + j0, dp, Xh + j0, Xl + j0, dp, Lih + (size_t)jt * TB * TB, Lil + (size_t)jt * TB * TB, TB, rows, nb, nb, GM_SET, lds);
  }
}

// per surviving node: lower tiles of D_s -= O_s O_s' (one or two eliminated neighbours); per eliminated node: the fill edge (-)= O_x O_y'
__global__ void __launch_bounds__(256, 2) k_dd_update(WS w, Dims dm, CrDev cr, int eoff, int nelim, int uoff, int nupd, int count) {
  const int dp = dm.dp, nm = (dp + 63) / 64;
  const int ntl = nm * (nm + 1) / 2, ntf = nm * nm;
  const int per = nupd * ntl + nelim * ntf;
  const int it = cr_item(count * per);
  if (it < 0) return;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int b = cr.alist[it / per];
  int r = it % per;
  const size_t bs = (size_t)dp * dp;
  if (r < nupd * ntl) {
    const int* ur = cr.upd + (size_t)(uoff + r / ntl) * CR_UW;
    int t = r % ntl, tm = 0;
    while (t > tm) { t -= tm + 1; ++tm; }
    const int tn = t, m0 = tm * 64, n0 = tn * 64;
    const int M = (dp - m0 < 64) ? dp - m0 : 64, N = (dp - n0 < 64) ? dp - n0 : 64;
    const size_t co = ((size_t)b * dm.p + ur[CU_NODE]) * bs + (size_t)m0 * dp + n0;
    for (int q = 0; q < 2; ++q) {
      const int slot = ur[q ? CU_E1 : CU_E0];
      if (slot < 0) break;
      const double* Oh = cr_edge(w, dm, b, slot); const double* Ol = dd_edge_lo(w, dm, b, slot);
      wg_gemm_dd<false>(w.D + co, w.Dl + co, dp, Oh + (size_t)m0 * dp, Ol + (size_t)m0 * dp, dp, Oh + (size_t)n0 * dp, Ol + (size_t)n0 * dp, dp, M, N, dp, GM_SUB, lds);
    }
  } else {
    r -= nupd * ntl;
    const int* er = cr.elim + (size_t)(eoff + r / ntf) * CR_EW;
    if (er[CE_FILL] < 0) return;
    const int t = r % ntf, tm = t / nm, tn = t - tm * nm;
    const int m0 = tm * 64, n0 = tn * 64;
    const int M = (dp - m0 < 64) ? dp - m0 : 64, N = (dp - n0 < 64) ? dp - n0 : 64;
    const int sx = er[CE_FX] ? er[CE_EB] : er[CE_EA], sy = er[CE_FX] ? er[CE_EA] : er[CE_EB];
    const double* Oxh = cr_edge(w, dm, b, sx); const double* Oxl = dd_edge_lo(w, dm, b, sx);
    const double* Oyh = cr_edge(w, dm, b, sy); const double* Oyl = dd_edge_lo(w, dm, b, sy);
    const size_t co = (size_t)m0 * dp + n0;
    wg_gemm_dd<false>(cr_edge(w, dm, b, er[CE_FILL]) + co, dd_edge_lo(w, dm, b, er[CE_FILL]) + co, dp, Oxh + (size_t)m0 * dp, Oxl + (size_t)m0 * dp, dp,
                      Oyh + (size_t)n0 * dp, Oyl + (size_t)n0 * dp, dp, M, N, dp, er[CE_FACC] ? GM_SUB : GM_NEG, lds);
  }
}

// ------------------------------------------------------------------ substitutions: dd right-hand sides [NCD][xld] in LDS
constexpr int NCD = 3;                          // right-hand sides per sweep
// Y[q][i] (+)= sgn * sum_c Mop[i][c] X[q][c],  Mop = M (rows x cols, ldm) or M'.  Slabs of 64 rows x 16 columns of Mop through LDS; thread
// (r, kq) = (tid & 63, tid >> 6) multiplies row r with the four columns 4 kq .. + 3 of every slab, the four partial sums meet in LDS.
constexpr int DD_GEMV_LDS = 2 * DG_SLAB + 2 * 4 * NCD * 64;
template <bool TRANS>
__device__ __forceinline__ void wg_gemv_dd(double* Yh, double* Yl, int yld, const double* Xh, const double* Xl, int xld, const double* Mh, const double* Ml, int ldm,
                                           int rows, int cols, bool accumulate, double sgn, double* As, int nc) {
  const int tid = threadIdx.x, r = tid & 63, kq = tid >> 6;
  double* sh = As; double* sl = As + DG_SLAB; double* redh = As + 2 * DG_SLAB; double* redl = redh + 4 * NCD * 64;
  for (int m0 = 0; m0 < rows; m0 += 64) {
    double ah_[NCD], al_[NCD];
#pragma unroll
    for (int q = 0; q < NCD; ++q) { ah_[q] = 0.0; al_[q] = 0.0; }
    for (int k0 = 0; k0 < cols; k0 += DG_K) {
      __syncthreads();
      if (TRANS) {                    // Mop[i][c] = M[c][i]: a slab row (fixed c) is contiguous in i
        const int k = tid >> 4, mq = (tid & 15) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool ok = (k0 + k < cols) && (m0 + mq + q < rows);
          sh[k * DG_LD + mq + q] = ok ? Mh[(size_t)(k0 + k) * ldm + m0 + mq + q] : 0.0;
          sl[k * DG_LD + mq + q] = ok ? Ml[(size_t)(k0 + k) * ldm + m0 + mq + q] : 0.0;
        }
      } else {
        const int rr = tid >> 2, kk = (tid & 3) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool ok = (m0 + rr < rows) && (k0 + kk + q < cols);
          sh[(kk + q) * DG_LD + rr] = ok ? Mh[(size_t)(m0 + rr) * ldm + k0 + kk + q] : 0.0;
          sl[(kk + q) * DG_LD + rr] = ok ? Ml[(size_t)(m0 + rr) * ldm + k0 + kk + q] : 0.0;
        }
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int k = 4 * kq + kk;
        if (k0 + k < cols) {
          const double mh = sh[k * DG_LD + r], ml = sl[k * DG_LD + r];
#pragma unroll
          for (int q = 0; q < NCD; ++q) if (q < nc) dd_fma_acc(ah_[q], al_[q], mh, ml, Xh[q * xld + k0 + k], Xl[q * xld + k0 + k]);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NCD; ++q) { const ddv v = dd_qts(ah_[q], al_[q]); redh[(kq * NCD + q) * 64 + r] = v.h; redl[(kq * NCD + q) * 64 + r] = v.l; }
    __syncthreads();
    if (kq < nc && m0 + r < rows) {            // wave kq joins the partial sums of right-hand side kq
      const int q = kq;
      ddv s = ddv{redh[q * 64 + r], redl[q * 64 + r]};
#pragma unroll
      for (int g = 1; g < 4; ++g) s = dd_add(s, ddv{redh[(g * NCD + q) * 64 + r], redl[(g * NCD + q) * 64 + r]});
      s = dd_muld(s, sgn);                      // (+-1: exact)
      const int i = m0 + r;
      const ddv y = accumulate ? dd_add(ddv{Yh[q * yld + i], Yl[q * yld + i]}, s) : s;
      Yh[q * yld + i] = y.h; Yl[q * yld + i] = y.l;
    }
  }
  __syncthreads();
}

constexpr int dd_solve_lds_doubles(int dp) { return 2 * 2 * NCD * (dp + 4) + 2 * NCD * (TB + 4) + DD_GEMV_LDS; }
// right-hand sides of a node: fp64 high words in W3 / Z (cr_rhs), low words in W3l / Zl
__device__ __forceinline__ double* dd_rhs_lo(const WS& w, const Dims& dm, int b, int node, int nc) {
  return ((nc == 3) ? w.W3l : w.Zl) + ((size_t)b * dm.p + node) * dm.dp * nc;
}
__device__ __forceinline__ void ddvec_g2s(double* zh, double* zl, int xld, const double* Rh, const double* Rl, int dp, int nc) {
  for (int e = threadIdx.x; e < dp * nc; e += 256) { const int i = e / nc, q = e - i * nc; zh[q * xld + i] = Rh[e]; zl[q * xld + i] = Rl[e]; }
}
__device__ __forceinline__ void ddvec_s2g(double* Rh, double* Rl, const double* zh, const double* zl, int xld, int dp, int nc) {
  for (int e = threadIdx.x; e < dp * nc; e += 256) { const int i = e / nc, q = e - i * nc; const ddv v = dd_qts(zh[q * xld + i], zl[q * xld + i]); Rh[e] = v.h; Rl[e] = v.l; }
}
// z <- L^-1 z and z <- L^-T z with the inverted diagonal tiles (blk_fwd / blk_bwd of tmpc_factor.h)
__device__ __forceinline__ void blk_fwd_dd(double* zh, double* zl, int xld, double* th, double* tl, int tld, const double* Dh, const double* Dl, const double* Lih, const double* Lil,
                                           int dp, double* As, int nc) {
  int jt = 0;
  for (int j0 = 0; j0 < dp; j0 += TB, ++jt) {
    const int nb = (dp - j0 < TB) ? dp - j0 : TB;
    wg_gemv_dd<false>(th, tl, tld, zh + j0, zl + j0, xld, Lih + (size_t)jt * TB * TB, Lil + (size_t)jt * TB * TB, TB, nb, nb, false, 1.0, As, nc);
    for (int e = threadIdx.x; e < nb * nc; e += 256) { const int q = e / nb, i = e - q * nb; zh[q * xld + j0 + i] = th[q * tld + i]; zl[q * xld + j0 + i] = tl[q * tld + i]; }
    __syncthreads();
    const int rem = dp - j0 - nb;
    if (rem > 0) wg_gemv_dd<false>(zh + j0 + nb, zl + j0 + nb, xld, zh + j0, zl + j0, xld, Dh + (size_t)(j0 + nb) * dp + j0, Dl + (size_t)(j0 + nb) * dp + j0, dp, rem, nb, true, -1.0, As, nc);
  }
}
__device__ __forceinline__ void blk_bwd_dd(double* zh, double* zl, int xld, double* th, double* tl, int tld, const double* Dh, const double* Dl, const double* Lih, const double* Lil,
                                           int dp, double* As, int nc) {
  const int nt = (dp + TB - 1) / TB;
  for (int jt = nt - 1; jt >= 0; --jt) {
    const int j0 = jt * TB, nb = (dp - j0 < TB) ? dp - j0 : TB;
    wg_gemv_dd<true>(th, tl, tld, zh + j0, zl + j0, xld, Lih + (size_t)jt * TB * TB, Lil + (size_t)jt * TB * TB, TB, nb, nb, false, 1.0, As, nc);
    for (int e = threadIdx.x; e < nb * nc; e += 256) { const int q = e / nb, i = e - q * nb; zh[q * xld + j0 + i] = th[q * tld + i]; zl[q * xld + j0 + i] = tl[q * tld + i]; }
    __syncthreads();
    if (j0 > 0) wg_gemv_dd<true>(zh, zl, xld, zh + j0, zl + j0, xld, Dh + (size_t)j0 * dp, Dl + (size_t)j0 * dp, dp, j0, nb, true, -1.0, As, nc);
  }
}
// number of right-hand sides of a problem in the tight loop / polish: pass 1 (main phase) and every centering / polish solve with a new factorisation carry three
__device__ __forceinline__ int dd_nc(const WS& w, int b, int pass) {
  const int phase = w.iprob[(size_t)b * IS + I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return 0;
  return (pass == 1 || (phase != PH_MAIN && !w.iprob[(size_t)b * IS + I_CHORD])) ? 3 : 1;      // chord step (round 5): the right-hand side alone, as cr_nc
}
#define TMPC_DD_SOLVE_LDS                                                                                     \
  extern __shared__ __attribute__((aligned(16))) double lds[];                                                \
  const int dp = dm.dp, xld = dp + 4, tld = TB + 4;                                                            \
  double* zch = lds; double* zcl = zch + NCD * xld; double* znh = zcl + NCD * xld; double* znl = znh + NCD * xld; \
  double* th = znl + NCD * xld; double* tl = th + NCD * tld; double* As = tl + NCD * tld;

__global__ void __launch_bounds__(256) k_dd_fwd_diag(WS w, Dims dm, CrDev cr, int eoff, int nelim, int count, int pass) {
  const int it = cr_item(count * nelim);
  if (it < 0) return;
  TMPC_DD_SOLVE_LDS
  const int b = cr.alist[it / nelim];
  const int nc = dd_nc(w, b, pass);
  if (nc == 0) return;
  const int* er = cr.elim + (size_t)(eoff + it % nelim) * CR_EW;
  const int node = er[CE_NODE];
  const size_t no = (size_t)b * dm.p + node;
  double* Rh = cr_rhs(w, dm, b, node, nc); double* Rl = dd_rhs_lo(w, dm, b, node, nc);
  ddvec_g2s(zch, zcl, xld, Rh, Rl, dp, nc);
  __syncthreads();
  blk_fwd_dd(zch, zcl, xld, th, tl, tld, w.D + no * dp * dp, w.Dl + no * dp * dp, w.Linv + no * dm.nt * TB * TB, w.Linvl + no * dm.nt * TB * TB, dp, As, nc);
  ddvec_s2g(Rh, Rl, zch, zcl, xld, dp, nc);
  (void)znh; (void)znl;
}
__global__ void __launch_bounds__(256) k_dd_fwd_off(WS w, Dims dm, CrDev cr, int uoff, int nupd, int count, int pass) {
  const int it = cr_item(count * nupd);
  if (it < 0) return;
  TMPC_DD_SOLVE_LDS
  const int b = cr.alist[it / nupd];
  const int nc = dd_nc(w, b, pass);
  if (nc == 0) return;
  const int* ur = cr.upd + (size_t)(uoff + it % nupd) * CR_UW;
  double* Rh = cr_rhs(w, dm, b, ur[CU_NODE], nc); double* Rl = dd_rhs_lo(w, dm, b, ur[CU_NODE], nc);
  ddvec_g2s(zch, zcl, xld, Rh, Rl, dp, nc);
  for (int q = 0; q < 2; ++q) {
    const int slot = ur[q ? CU_E1 : CU_E0];
    if (slot < 0) break;
    const int src = ur[q ? CU_S1 : CU_S0];
    __syncthreads();
    ddvec_g2s(znh, znl, xld, cr_rhs(w, dm, b, src, nc), dd_rhs_lo(w, dm, b, src, nc), dp, nc);
    __syncthreads();
    wg_gemv_dd<false>(zch, zcl, xld, znh, znl, xld, cr_edge(w, dm, b, slot), dd_edge_lo(w, dm, b, slot), dp, dp, dp, true, -1.0, As, nc);      // z_s -= O_s z_i
  }
  __syncthreads();
  ddvec_s2g(Rh, Rl, zch, zcl, xld, dp, nc);
  (void)th; (void)tl;
}
__global__ void __launch_bounds__(256) k_dd_bwd(WS w, Dims dm, CrDev cr, int eoff, int nelim, int count, int pass) {
  const int it = cr_item(count * nelim);
  if (it < 0) return;
  TMPC_DD_SOLVE_LDS
  const int b = cr.alist[it / nelim];
  const int nc = dd_nc(w, b, pass);
  if (nc == 0) return;
  const int* er = cr.elim + (size_t)(eoff + it % nelim) * CR_EW;
  const int node = er[CE_NODE];
  const size_t no = (size_t)b * dm.p + node;
  double* Rh = cr_rhs(w, dm, b, node, nc); double* Rl = dd_rhs_lo(w, dm, b, node, nc);
  ddvec_g2s(zch, zcl, xld, Rh, Rl, dp, nc);
  for (int q = 0; q < 2; ++q) {
    const int slot = er[q ? CE_EB : CE_EA];
    if (slot < 0) continue;
    const int nbr = er[q ? CE_NB : CE_NA];
    __syncthreads();
    ddvec_g2s(znh, znl, xld, cr_rhs(w, dm, b, nbr, nc), dd_rhs_lo(w, dm, b, nbr, nc), dp, nc);
    __syncthreads();
    wg_gemv_dd<true>(zch, zcl, xld, znh, znl, xld, cr_edge(w, dm, b, slot), dd_edge_lo(w, dm, b, slot), dp, dp, dp, true, -1.0, As, nc);       // z_i -= O_x' z_x
  }
  __syncthreads();
  blk_bwd_dd(zch, zcl, xld, th, tl, tld, w.D + no * dp * dp, w.Dl + no * dp * dp, w.Linv + no * dm.nt * TB * TB, w.Linvl + no * dm.nt * TB * TB, dp, As, nc);
  ddvec_s2g(Rh, Rl, zch, zcl, xld, dp, nc);
}

// ------------------------------------------------------------------ small dd matrices (stage level), 256 threads
// a matrix is a pair of images (hi at p, lo at p + ms), leading dimension ld: LDS slots (ld = LD, ms = MS) at n <= 32; for 32 < n <= 64 the same
// routines run on a per-stage scratch in global memory (ld = n, ms = n * n; WS::ddscr) -- every barrier below is therefore a fence + barrier
struct sdd {
  double* p; int ld, ms;
  __device__ __forceinline__ double& h(int i, int j) const { return p[i * ld + j]; }
  __device__ __forceinline__ double& l(int i, int j) const { return p[ms + i * ld + j]; }
  __device__ __forceinline__ ddv get(int i, int j) const { return ddv{p[i * ld + j], p[ms + i * ld + j]}; }
  __device__ __forceinline__ void set(int i, int j, ddv v) const { p[i * ld + j] = v.h; p[ms + i * ld + j] = v.l; }
};
__device__ __forceinline__ void dsync() { __threadfence_block(); __syncthreads(); }
// slot q of the stage's dd matrices: LDS (sm) or, for n > NMAX, the global scratch
__device__ __forceinline__ sdd sdd_slot(const WS& w, const Dims& dm, size_t sid, double* sm, int q) {
  if (dm.n > NMAX) { const int nn = dm.n * dm.n; return sdd{w.ddscr + (sid * DD_SCR_MATS + q) * 2 * (size_t)nn, dm.n, nn}; }
  return sdd{sm + 2 * q * MS, LD, MS};
}
// C (m x n) = A (m x k) op(B): tb ? B (n x k)' : B (k x n); ta: A given as (k x m)'
__device__ __forceinline__ void sdd_mm(sdd C, sdd A, bool ta, sdd B, bool tb, int m, int k, int n) {
  for (int e = threadIdx.x; e < m * n; e += 256) {
    const int i = e / n, j = e - i * n;
    double sh = 0.0, sl = 0.0;
    for (int q = 0; q < k; ++q) {
      const ddv a = ta ? A.get(q, i) : A.get(i, q), bb = tb ? B.get(j, q) : B.get(q, j);
      dd_fma_acc(sh, sl, a.h, a.l, bb.h, bb.l);
    }
    C.set(i, j, dd_qts(sh, sl));
  }
  dsync();
}
// in-place lower Cholesky of the n x n dd matrix S, then Z = S^-1 = L^-T L^-1 (W: scratch for L^-1).  Returns non-positive pivots (uniform).
__device__ __forceinline__ int sdd_inv_spd(sdd Z, sdd S, sdd W, int n, double* flag) {
  const int tid = threadIdx.x;
  if (tid == 0) flag[0] = 0.0;
  dsync();
  for (int j = 0; j < n; ++j) {
    ddv piv = S.get(j, j);
    if (!(piv.h > 0.0)) { piv = ddv{1.0, 0.0}; if (tid == 0) flag[0] += 1.0; }
    const ddv r = dd_sqrt(piv), rinv = dd_div(dd_from(1.0), r);
    dsync();
    if (tid == 0) S.set(j, j, r);
    if (tid > j && tid < n) S.set(tid, j, dd_mul(S.get(tid, j), rinv));
    dsync();
    for (int e = tid; e < (n - j - 1) * (n - j - 1); e += 256) {
      const int i = j + 1 + e / (n - j - 1), k = j + 1 + e % (n - j - 1);
      if (k <= i) S.set(i, k, dd_sub(S.get(i, k), dd_mul(S.get(i, j), S.get(k, j))));
    }
    dsync();
  }
  for (int e = tid; e < n * n; e += 256) W.set(e / n, e % n, ddv{0.0, 0.0});
  dsync();
  if (tid < n) {                     // column c of L^-1 (forward substitution)
    const int c = tid;
    for (int i = c; i < n; ++i) {
      ddv s = dd_from(i == c ? 1.0 : 0.0);
      for (int k = c; k < i; ++k) s = dd_sub(s, dd_mul(S.get(i, k), W.get(k, c)));
      W.set(i, c, dd_div(s, S.get(i, i)));
    }
  }
  dsync();
  sdd_mm(Z, W, true, W, false, n, n, n);          // L^-T L^-1
  for (int e = tid; e < n * n; e += 256) {        // exact symmetry
    const int i = e / n, j = e % n;
    if (j < i) { const ddv v = dd_muld(dd_add(Z.get(i, j), Z.get(j, i)), 0.5); S.set(i, j, v); }
  }
  dsync();
  for (int e = tid; e < n * n; e += 256) { const int i = e / n, j = e % n; if (j < i) { const ddv v = S.get(i, j); Z.set(i, j, v); Z.set(j, i, v); } }
  dsync();
  return (int)flag[0];
}
__device__ __forceinline__ void sdd_store(double* gh, double* gl, sdd A, int r, int c, int ldg) {       // A[:r, :c] -> global planes
  for (int e = threadIdx.x; e < r * c; e += 256) { const int i = e / c, j = e - i * c; gh[(size_t)i * ldg + j] = A.h(i, j); gl[(size_t)i * ldg + j] = A.l(i, j); }
}
__device__ __forceinline__ void sdd_load64(sdd A, const double* g, int r, int c, int ldg) {              // fp64 global -> dd (lo = 0)
  for (int e = threadIdx.x; e < r * c; e += 256) { const int i = e / c, j = e - i * c; A.h(i, j) = g[(size_t)i * ldg + j]; A.l(i, j) = 0.0; }
}

// Kronecker-factor images of one cone block from the dd pair (X, Z): KX = V X V', KS = V Z V', FX = X_E V', FS = Z_E V', XXX, SIXX (t0, t1: scratch)
__device__ __forceinline__ void dd_images(const WS& w, size_t sid, int r, sdd V, sdd X, sdd Z, sdd t0, sdd t1, int n, int nx) {
  const int nxx = nx * nx;
  double* kh = w.KF + (sid * 12 + (size_t)r * KF_PER_LMI) * nxx; double* kl = w.KFl + (sid * 12 + (size_t)r * KF_PER_LMI) * nxx;
  sdd_mm(t0, V, false, X, false, nx, n, n); sdd_mm(t1, t0, false, V, true, nx, n, nx);
  sdd_store(kh + KF_KX * nxx, kl + KF_KX * nxx, t1, nx, nx, nx);
  sdd_mm(t1, X, false, V, true, nx, n, nx);                                        // X[:nx, :] V'
  sdd_store(kh + KF_FX * nxx, kl + KF_FX * nxx, t1, nx, nx, nx);
  sdd_store(kh + KF_XXX * nxx, kl + KF_XXX * nxx, X, nx, nx, nx);
  dsync();
  sdd_mm(t0, V, false, Z, false, nx, n, n); sdd_mm(t1, t0, false, V, true, nx, n, nx);
  sdd_store(kh + KF_KS * nxx, kl + KF_KS * nxx, t1, nx, nx, nx);
  sdd_mm(t1, Z, false, V, true, nx, n, nx);
  sdd_store(kh + KF_FS * nxx, kl + KF_FS * nxx, t1, nx, nx, nx);
  sdd_store(kh + KF_SIXX * nxx, kl + KF_SIXX * nxx, Z, nx, nx, nx);
  dsync();
}

// tight loop: the images from the fp64 iterates X_r, S_r^-1 (k_stage_pre wrote them), one workgroup per stage of the active problems
constexpr int DD_IMG_SLOTS = 10;                 // five dd matrices
__global__ void __launch_bounds__(256) k_dd_images(WS w, Dims dm) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int sid = stage_id(w, dm);
  const int b = sid / dm.p;
  if (w.iprob[(size_t)b * IS + I_PHASE] == PH_DONE) return;
  const int n = dm.n, nx = dm.nx, nn = n * n;
  const sdd V = sdd_slot(w, dm, sid, sm, 0), X = sdd_slot(w, dm, sid, sm, 1), Z = sdd_slot(w, dm, sid, sm, 2), t0 = sdd_slot(w, dm, sid, sm, 3), t1 = sdd_slot(w, dm, sid, sm, 4);
  sdd_load64(V, w.V + (size_t)sid * nx * n, nx, n, n);
  for (int r = 0; r < 2; ++r) {
    dsync();
    sdd_load64(X, (r ? w.X2 : w.X1) + (size_t)sid * nn, n, n, n);
    sdd_load64(Z, (r ? w.S2i : w.S1i) + (size_t)sid * nn, n, n, n);
    dsync();
    dd_images(w, (size_t)sid, r, V, X, Z, t0, t1, n, nx);
  }
}

// ------------------------------------------------------------------ assembly of D_k and the coupling block in dd (k_schur of tmpc_schur.h)
// One workgroup per stage and part; a thread owns a stored column and walks the rows.  PART 0: D_k (lower triangle), PART 1: coupling block.
// GF (nx > 35: the eight dd factor matrices of PART 0 do not fit the LDS): the factors are read where k_dd_images / k_dd_polish_pre left them in global memory
template <int PART> constexpr int ddsch_mats() { return PART == 0 ? 8 : 4; }
template <int PART, bool GF = false>
__global__ void __launch_bounds__(256) k_dd_schur(WS w, Dims dm) {
  constexpr int NM = ddsch_mats<PART>();
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int sid = stage_id(w, dm);
  const int b = sid / dm.p, k = sid - b * dm.p;
  if (w.iprob[(size_t)b * IS + I_PHASE] == PH_DONE) return;
  const int tid = threadIdx.x, nx = dm.nx, nxx = nx * nx, d = dm.d, dp = dm.dp;
  double* mh = sm; double* ml = sm + (GF ? 0 : (size_t)NM * nxx);                 // [NM][nx][nx] hi, lo
  unsigned* pair = (unsigned*)(ml + (GF ? 0 : (size_t)NM * nxx));
  const bool corner = (w.cr_orient[k] != 0);
  const int km = (k == 0) ? dm.p - 1 : k - 1;
  const size_t ok = (size_t)sid * 12 * nxx, om = (size_t)(b * dm.p + km) * 12 * nxx;
  size_t gsrc[NM];                       // GF: where matrix q lives in KF / KFl
  const bool gtr = (PART == 1) && !corner;       // GF: the F matrices are used transposed
#pragma unroll
  for (int q = 0; q < NM; ++q) {
    const int lmi = (PART == 1) ? q / 2 : q / 4, slot = (PART == 1) ? KF_FX + (q - 2 * lmi) : q - 4 * lmi;
    gsrc[q] = ((slot == KF_KX || slot == KF_KS) ? om : ok) + (size_t)(lmi * KF_PER_LMI + slot) * nxx;
  }
  if (!GF)
  for (int e = tid; e < 12 * nxx; e += 256) {
    const int m = e / nxx, r = e - m * nxx;
    const int lmi = m / KF_PER_LMI, slot = m - lmi * KF_PER_LMI;
    const bool fmat = (slot == KF_FX || slot == KF_FS);
    if (fmat != (PART == 1)) continue;
    const size_t src = ((slot == KF_KX || slot == KF_KS) ? om : ok) + e;
    int i = r / nx, j = r - i * nx;
    if (fmat && !corner) { const int t_ = i; i = j; j = t_; }
    const int q = fmat ? 2 * lmi + (slot - KF_FX) : 4 * lmi + slot;       // D: XXX, SIXX, KX, KS per LMI; C: FX, FS per LMI
    mh[(size_t)q * nxx + i * nx + j] = w.KF[src]; ml[(size_t)q * nxx + i * nx + j] = w.KFl[src];
  }
  if (tid < nx) { int e = tid * nx - (tid * (tid - 1)) / 2; for (int c = tid; c < nx; ++c) pair[e++] = (unsigned)tid | ((unsigned)c << 16); }
  __syncthreads();
  double* Gh = (PART == 0 ? w.D : w.O) + (size_t)sid * dp * dp; double* Gl = (PART == 0 ? w.Dl : w.Ol) + (size_t)sid * dp * dp;
  auto at = [&](int q, int i, int j) {
    if (GF) { const size_t g = gsrc[q] + (gtr ? j * nx + i : i * nx + j); return ddv{w.KF[g], w.KFl[g]}; }
    return ddv{mh[(size_t)q * nxx + i * nx + j], ml[(size_t)q * nxx + i * nx + j]};
  };
  auto hkm = [&](int qx, int qs, int a, int bb, int c, int e_) {          // T(Lx, Ls)[(ab),(ce)] without the weights
    return dd_add(dd_add(dd_mul(at(qx, a, c), at(qs, bb, e_)), dd_mul(at(qx, a, e_), at(qs, bb, c))),
                  dd_add(dd_mul(at(qx, bb, c), at(qs, a, e_)), dd_mul(at(qx, bb, e_), at(qs, a, c))));
  };
  for (int col = tid; col < dp; col += 256) {
    const bool cin = col < d;
    const unsigned pc_ = cin ? pair[col] : 0u;
    const int c = (int)(pc_ & 0xffffu), e_ = (int)(pc_ >> 16);
    const double wc = (c == e_) ? 0.5 : 1.0;
    for (int row = (PART == 0 ? col : 0); row < dp; ++row) {
      const size_t g = (size_t)row * dp + col;
      if (row >= d || !cin) { Gh[g] = (PART == 0 && row == col) ? 1.0 : 0.0; Gl[g] = 0.0; continue; }
      const unsigned pr_ = pair[row];
      const int a = (int)(pr_ & 0xffffu), bb = (int)(pr_ >> 16);
      const double wgt = ((a == bb) ? 0.5 : 1.0) * wc;
      ddv v;
      if (PART == 0) v = dd_add(dd_add(hkm(0, 1, a, bb, c, e_), hkm(2, 3, a, bb, c, e_)), dd_add(hkm(4, 5, a, bb, c, e_), hkm(6, 7, a, bb, c, e_)));
      else v = dd_neg(dd_add(hkm(0, 1, a, bb, c, e_), hkm(2, 3, a, bb, c, e_)));
      v = dd_muld(v, wgt);                                               // (a power of two: exact)
      Gh[g] = v.h; Gl[g] = v.l;
    }
  }
}

// ------------------------------------------------------------------ rows of the stage-local multipliers in dd (k_phi_pre + k_aug_fill of tmpc_phi.h)
// Tight mode with rows of G (round 5).  The augmented blocks are factored by a Cholesky without any safeguard, and the pivot of a multiplier row is what is left of
// T_ii ~ 1/mu after b_i' D^-1 b_i has been taken off: z_i / phi_i ~ mu for an active row -- a difference 1e-21 of the terms.  With the rows formed in fp64 (relative error
// 1e-16) its sign is noise: measured, 91 non-positive pivots at the first polish step of a member whose oracle run (a pivoted LU of the border system) is uneventful.  Here
// every entry of the rows is a dd function of the same (X_r, S_r^-1) the dd blocks are made of: w = X g, u = S^-1 g, V w, V u,
//   a_i = -svec(sum_r sym(w u')[:nx,:nx]) (P_k),  b_i = svec(sum_r sym(V w (V u)')) (P_{k+1}),  T_ij = sum_r ((g_i'X g_j)(g_j'S^-1 g_i) + (i <-> j)) / 2 + delta_ij z_i / phi_i.
// polish = 0: the loop (X_r, S_r^-1, z, phi the fp64 iterates); 1: the polish (S_r^-1 in dd from k_dd_polish_pre, X_r = mu S_r^-1, z = mu / phi).
// The border entries c_tau, c_alpha and the vectors k_phi_dir needs stay with k_phi_pre (fp64, like the border columns of tau and alpha).  One workgroup per stage, n <= 32.
static_assert(AEL + 1 == LD && AEL == NMAX, "arrow blocks share the 32 x 33 LDS slots of the stage kernels");
constexpr int ARW_LD = AEL + 1;                                 // leading dimension of an arrow block in LDS
constexpr int ARW_DOUBLES = 4 * AEL * ARW_LD + 8;              // S^-1 and X of one arrow block, hi and lo
__host__ __device__ constexpr int dd_aug_lds_doubles(int nr, int n, int nx) { return 4 * nr * (2 * n + 2 * nx) + ARW_DOUBLES + 16; }       // rows of [G; C]: 125 KB at 31 rows, n = 32
// Closed-form inverse of the arrow matrix S = [[t, u'], [u, t I]], u_i = wr phi_i (m entries), in dd: gam = t^2 - |u|^2 cancels to ~mu t on an active norm term
// (k_phi_init has the fp64 form).  S^-1 = [[t, -u'], [-u, (gam / t) I + u u' / t]] / gam.  256 threads; Sih / Sil: ne x ne with leading dimension ARW_LD.
// Returns gam (uniform); the caller checks t > 0 and gam > 0.  sc: 8 doubles of LDS.  (Sih / Sil may be global memory -- the scratch of k_dd_polish_pre at n > 32: fences with the barriers.)
__device__ __forceinline__ ddv arrow_inv_dd(double* Sih, double* Sil, double t, const double* phi, double wr, int m, double* sc) {
  const int tid = threadIdx.x, ne = m + 1;
  __threadfence_block(); __syncthreads();
  if (tid == 0) {
    ddv usq = ddv{0.0, 0.0};
    for (int i = 0; i < m; ++i) { const ddv u = dd_tp(wr, phi[i]); usq = dd_add(usq, dd_mul(u, u)); }
    const ddv gam = dd_sub(dd_tp(t, t), usq);
    const ddv gi = dd_div(dd_from(1.0), gam), ti = dd_div(dd_from(1.0), dd_from(t));
    sc[0] = gam.h; sc[1] = gam.l; sc[2] = gi.h; sc[3] = gi.l; sc[4] = ti.h; sc[5] = ti.l;
  }
  __threadfence_block(); __syncthreads();
  const ddv gi = ddv{sc[2], sc[3]}, ti = ddv{sc[4], sc[5]};
  for (int e = tid; e < ne * ne; e += 256) {
    const int i = e / ne, j = e - i * ne;
    ddv v;
    if (i == 0 && j == 0) v = dd_muld(gi, t);
    else if (i == 0 || j == 0) v = dd_neg(dd_mul(dd_tp(wr, phi[(i ? i : j) - 1]), gi));
    else {
      v = dd_mul(dd_mul(dd_tp(wr, phi[i - 1]), dd_tp(wr, phi[j - 1])), dd_mul(ti, gi));
      if (i == j) v = dd_add(v, ti);
    }
    Sih[i * ARW_LD + j] = v.h; Sil[i * ARW_LD + j] = v.l;
  }
  __threadfence_block(); __syncthreads();
  return ddv{sc[0], sc[1]};
}
__global__ void __launch_bounds__(256) k_dd_aug_fill(WS w, Dims dm, int polish) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int sid = stage_id(w, dm), tid = threadIdx.x;
  const int b = sid / dm.p, k = sid - b * dm.p;
  const int phase = w.iprob[(size_t)b * IS + I_PHASE];
  if (polish ? phase != PH_POLISH : (phase == PH_DONE || phase == PH_POLISH)) return;
  const int ng = stage_rows(w, dm, sid);
  if (ng < 1) return;
  const double* pr = w.prob + (size_t)b * PS;
  const double mu = pr[P_MUT];
  const int n = dm.n, nx = dm.nx, nn = n * n, d = dm.d, dp = dm.dp, p = dm.p, vl = 2 * n + 2 * nx;
  double* vh = sm; double* vlo = sm + 2 * ng * vl;                    // [2][ng][w | u | V w | V u], hi and lo
  const double* Gg = w.G + (size_t)sid * dm.nr * n;
  const double* Vg = w.V + (size_t)sid * nx * n;
  auto mat = [&](int r, int which, int a, int c) -> ddv {             // which 0: X_r, 1: S_r^-1
    if (polish) {
      const double* zd = w.Zdd + ((size_t)sid * 2 + r) * 2 * nn;
      const ddv z = ddv{zd[a * n + c], zd[nn + a * n + c]};
      return which ? z : dd_muld(z, mu);
    }
    return ddv{(which ? (r ? w.S2i : w.S1i) : (r ? w.X2 : w.X1))[(size_t)sid * nn + a * n + c], 0.0};
  };
  for (int e = tid; e < 4 * ng * n; e += 256) {                       // w = X g, u = S^-1 g
    const int a = e % n, i = (e / n) % ng, which = (e / (n * ng)) & 1, r = e / (2 * n * ng);
    double sh = 0.0, sl = 0.0;
    for (int c = 0; c < n; ++c) { const ddv m = mat(r, which, a, c); dd_fma_acc(sh, sl, m.h, m.l, Gg[i * n + c], 0.0); }
    const ddv v = dd_qts(sh, sl);
    vh[(r * ng + i) * vl + which * n + a] = v.h; vlo[(r * ng + i) * vl + which * n + a] = v.l;
  }
  __syncthreads();
  for (int e = tid; e < 4 * ng * nx; e += 256) {                      // V w, V u
    const int a = e % nx, i = (e / nx) % ng, which = (e / (nx * ng)) & 1, r = e / (2 * nx * ng);
    const int o = (r * ng + i) * vl + which * n;
    double sh = 0.0, sl = 0.0;
    for (int c = 0; c < n; ++c) dd_fma_acc(sh, sl, vh[o + c], vlo[o + c], Vg[a * n + c], 0.0);
    const ddv v = dd_qts(sh, sl);
    vh[(r * ng + i) * vl + 2 * n + which * nx + a] = v.h; vlo[(r * ng + i) * vl + 2 * n + which * nx + a] = v.l;
  }
  __syncthreads();
  const int kn = (k + 1 == p) ? 0 : k + 1;
  const size_t bs = (size_t)dp * dp;
  double* Dh = w.D + ((size_t)b * p + kn) * bs; double* Dl = w.Dl + ((size_t)b * p + kn) * bs;
  double* ddn = w.Ddiag + ((size_t)b * p + kn) * dp;
  const bool corner = (w.cr_orient[k] != 0);
  double* Ch = w.O + (size_t)sid * bs; double* Cl = w.Ol + (size_t)sid * bs;
  const double* phi = w.phi + (size_t)sid * dm.nr; const double* z = w.zph + (size_t)sid * dm.nr;
  auto gram = [&](int r, int which, int i, int j) {                   // g_i' (X_r | S_r^-1) g_j
    const int o = (r * ng + j) * vl + which * n;
    double sh = 0.0, sl = 0.0;
    for (int c = 0; c < n; ++c) dd_fma_acc(sh, sl, vh[o + c], vlo[o + c], Gg[i * n + c], 0.0);
    return dd_qts(sh, sl);
  };
  for (int e = tid; e < ng * ng; e += 256) {                          // T_loc,loc (lower triangle) and its pivot reference
    const int i = e / ng, j = e - i * ng;
    if (j > i) continue;
    ddv t = ddv{0.0, 0.0};
    for (int r = 0; r < 2; ++r)
      t = dd_add(t, dd_muld(dd_add(dd_mul(gram(r, 0, i, j), gram(r, 1, j, i)), dd_mul(gram(r, 0, j, i), gram(r, 1, i, j))), 0.5));
    if (i == j) t = dd_add(t, polish ? dd_div(dd_from(mu), dd_mul(dd_from(phi[i]), dd_from(phi[i]))) : dd_div(dd_from(z[i]), dd_from(phi[i])));
    Dh[(size_t)(d + i) * dp + d + j] = t.h; Dl[(size_t)(d + i) * dp + d + j] = t.l;
    if (i == j) ddn[d + i] = t.h;
  }
  for (int e = tid; e < ng * d; e += 256) {                           // the rows a_i (coupling block) and b_i (D_{k+1})
    const int i = e / d, idx = e - i * d;
    int a = 0, rem = idx;
    while (rem >= nx - a) { rem -= nx - a; ++a; }
    const int c = a + rem;
    ddv av = ddv{0.0, 0.0}, bv = ddv{0.0, 0.0};
    for (int r = 0; r < 2; ++r) {
      const int o = (r * ng + i) * vl;
      auto V_ = [&](int off, int q) { return ddv{vh[o + off + q], vlo[o + off + q]}; };
      av = dd_add(av, dd_muld(dd_add(dd_mul(V_(0, a), V_(n, c)), dd_mul(V_(0, c), V_(n, a))), 0.5));
      bv = dd_add(bv, dd_muld(dd_add(dd_mul(V_(2 * n, a), V_(2 * n + nx, c)), dd_mul(V_(2 * n, c), V_(2 * n + nx, a))), 0.5));
    }
    if (a != c) { av = dd_muld(av, 2.0); bv = dd_muld(bv, 2.0); }
    av = dd_neg(av);
    const size_t ro = (size_t)(d + i) * dp + idx;
    if (p == 1) { const ddv s_ = dd_add(av, bv); Dh[ro] = s_.h; Dl[ro] = s_.l; continue; }       // both couplings land in the one P block
    Dh[ro] = bv.h; Dl[ro] = bv.l;
    const size_t co = corner ? (size_t)idx * dp + d + i : ro;                                 // stored [block k][block k+1] or [block k+1][block k]
    Ch[co] = av.h; Cl[co] = av.l;
  }
  // Step 2: the norm terms.  Arrow block e covers rows a0 .. a0 + m - 1; its epigraph variable is local variable ng + e (no coupling to P, tau, alpha).  Entries of
  // T_loc,loc as k_phi_pre forms them in fp64, here from the dd closed-form inverse:  <E_q, sym(X E_t S^-1)> etc.
  const PhiStage ps = phi_stage(w, dm, sid);
  if (ps.na > 0) {
    const double wr = phi_wr(w, pr);
    double* Sih = sm + 4 * ng * vl; double* Sil = Sih + AEL * ARW_LD; double* Xh = Sil + AEL * ARW_LD; double* Xl = Xh + AEL * ARW_LD; double* sc = Xl + AEL * ARW_LD;
    for (int e_ = 0; e_ < ps.na; ++e_) {
      const int m = ps.am[e_], ne = m + 1, c0 = ps.a0[e_], te = ng + e_;
      arrow_inv_dd(Sih, Sil, w.at[(size_t)sid * 2 + e_], phi + c0, wr, m, sc);
      const double* aXg = w.aX + ((size_t)sid * 2 + e_) * AE;
      for (int e = tid; e < ne * ne; e += 256) {
        const int i = e / ne, j = e - i * ne;
        const ddv x = polish ? dd_muld(ddv{Sih[i * ARW_LD + j], Sil[i * ARW_LD + j]}, mu) : ddv{aXg[i * AEL + j], 0.0};
        Xh[i * ARW_LD + j] = x.h; Xl[i * ARW_LD + j] = x.l;
      }
      __syncthreads();
      auto SI = [&](int i, int j) { return ddv{Sih[i * ARW_LD + j], Sil[i * ARW_LD + j]}; };
      auto XX = [&](int i, int j) { return ddv{Xh[i * ARW_LD + j], Xl[i * ARW_LD + j]}; };
      for (int e = tid; e < ng * ng; e += 256) {                        // phi-phi part: same thread as the Gram part above (read-modify-write of its own entry)
        const int i = e / ng, j = e - i * ng;
        if (j > i || i < c0 || i >= c0 + m || j < c0) continue;
        const int a = i - c0, q = j - c0;
        ddv v = dd_add(dd_add(dd_mul(XX(0, 0), SI(q + 1, a + 1)), dd_mul(XX(0, q + 1), SI(0, a + 1))), dd_add(dd_mul(XX(a + 1, 0), SI(q + 1, 0)), dd_mul(XX(a + 1, q + 1), SI(0, 0))));
        v = dd_mul(v, dd_tp(wr, wr));
        const size_t g = (size_t)(d + i) * dp + d + j;
        const ddv t = dd_add(ddv{Dh[g], Dl[g]}, v);
        Dh[g] = t.h; Dl[g] = t.l;
        if (i == j) ddn[d + i] = t.h;
      }
      for (int q = tid; q <= m; q += 256) {                             // row of the epigraph variable: couplings to its phi_q, and tr(X S^-1) on the diagonal
        ddv v = ddv{0.0, 0.0};
        if (q < m) {
          for (int r = 0; r < ne; ++r) v = dd_add(v, dd_add(dd_mul(XX(0, r), SI(r, q + 1)), dd_mul(XX(q + 1, r), SI(r, 0))));
          v = dd_muld(v, wr);
          Dh[(size_t)(d + te) * dp + d + c0 + q] = v.h; Dl[(size_t)(d + te) * dp + d + c0 + q] = v.l;
        } else {
          for (int i = 0; i < ne; ++i) for (int r = 0; r < ne; ++r) v = dd_add(v, dd_mul(XX(i, r), SI(r, i)));
          Dh[(size_t)(d + te) * dp + d + te] = v.h; Dl[(size_t)(d + te) * dp + d + te] = v.l; ddn[d + te] = v.h;
        }
      }
      for (int j = tid; j < te; j += 256)                                // (the rest of the row: no coupling to the other block's rows or to another epigraph variable)
        if (j < c0 || j >= c0 + m) { Dh[(size_t)(d + te) * dp + d + j] = 0.0; Dl[(size_t)(d + te) * dp + d + j] = 0.0; }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------ polish: every stage quantity in dd from the fp64 y = (tau, alpha, P)
// Writes, per stage: the dd Kronecker factors (X_r := mu S_r^-1); the adjoint pieces V G V' and G_EE in dd of G = X1 - X2 (gradient),
// Psi = X2 S2^-1 and Phi(Hb) (border columns); the partial sums of the border / gradient scalars; fp64 roundings of S_r, X_r (outputs,
// dual export) and of M (T1: the step norm).  Q_CHOLBAD counts non-positive pivots (the iterate left the cone).
constexpr int DD_POL_SLOTS = 18;                 // nine dd matrices
__global__ void __launch_bounds__(256) k_dd_polish_pre(WS w, Dims dm, int final_sweep) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ double flag[2];
  const int sid = stage_id(w, dm);
  const int b = sid / dm.p, k = sid - b * dm.p;
  // final_sweep: the problems whose polish ended Optimal, once more at their final iterate (cone check, X_r and S_r for the outputs)
  if (final_sweep ? (w.iprob[(size_t)b * IS + I_PHASE] != PH_DONE || w.iprob[(size_t)b * IS + I_NPOLISH] == 0 || w.iprob[(size_t)b * IS + I_IPMSTATUS] != IPM_OPTIMAL)
                  : (w.iprob[(size_t)b * IS + I_PHASE] != PH_POLISH)) return;
  const double* pr = w.prob + (size_t)b * PS;
  const double tau = pr[P_TAU], alpha = pr[P_ALPHA], mu = pr[P_MUT];
  const int tid = threadIdx.x, n = dm.n, nx = dm.nx, nn = n * n, nxx = nx * nx;
  const int kn = (k + 1 == dm.p) ? 0 : k + 1;
  const sdd V = sdd_slot(w, dm, sid, sm, 0), Hd = sdd_slot(w, dm, sid, sm, 1), M = sdd_slot(w, dm, sid, sm, 2), Z1 = sdd_slot(w, dm, sid, sm, 3), Z2 = sdd_slot(w, dm, sid, sm, 4),
            X = sdd_slot(w, dm, sid, sm, 5), t0 = sdd_slot(w, dm, sid, sm, 6), t1 = sdd_slot(w, dm, sid, sm, 7), t2 = sdd_slot(w, dm, sid, sm, 8);
  sdd_load64(V, w.V + (size_t)sid * nx * n, nx, n, n);
  sdd_load64(Hd, w.Hb + (size_t)sid * nn, n, n, n);
  sdd_load64(t0, w.P + (size_t)(b * dm.p + kn) * nxx, nx, nx, nx);
  dsync();
  sdd_mm(t1, V, true, t0, false, n, nx, nx);                 // V' P_{k+1}
  sdd_mm(M, t1, false, V, false, n, nx, n);                  // (V' P+) V
  const double* Pk = w.P + (size_t)sid * nxx;
  // rows of G_k (round 5; Step 1 with the cost-free multipliers phi of convexifier.py:249-255): M += sum_i phi_i g_i g_i', products exact, sums in dd
  const int ngs = dm.nr > 0 ? stage_rows(w, dm, sid) : 0;
  const double* Gg = w.G + (size_t)sid * dm.nr * n; const double* ph = w.phi + (size_t)sid * dm.nr;
  for (int e = tid; e < nn; e += 256) {
    const int i = e / n, j = e - i * n;
    ddv v = dd_add(M.get(i, j), dd_muld(Hd.get(i, j), alpha));
    if (i < nx && j < nx) v = dd_sub(v, dd_from(Pk[i * nx + j]));
    for (int r_ = 0; r_ < ngs; ++r_) v = dd_add(v, dd_muld(dd_tp(Gg[r_ * n + i], Gg[r_ * n + j]), ph[r_]));
    M.set(i, j, v);
  }
  dsync();
  for (int e = tid; e < nn; e += 256) {                     // exact symmetry, fp64 rounding for the step norm
    const int i = e / n, j = e - i * n;
    if (j < i) { const ddv v = dd_muld(dd_add(M.get(i, j), M.get(j, i)), 0.5); t0.set(i, j, v); }
  }
  dsync();
  for (int e = tid; e < nn; e += 256) { const int i = e / n, j = e - i * n; if (j < i) { const ddv v = t0.get(i, j); M.set(i, j, v); M.set(j, i, v); } }
  dsync();
  for (int e = tid; e < nn; e += 256) { const int i = e / n, j = e - i * n; w.T1[(size_t)sid * nn + e] = dd_val(M.get(i, j)); }
  int nbad = (alpha - ALPHA_MIN > 0.0) ? 0 : 1;
  for (int r_ = 0; r_ < ngs; ++r_) nbad += (ph[r_] > 0.0) ? 0 : 1;
  // S1 = M - I, S2 = tau I - M, their inverses
  for (int e = tid; e < nn; e += 256) { const int i = e / n, j = e - i * n; const ddv m = M.get(i, j); t0.set(i, j, (i == j) ? dd_sub(m, dd_from(1.0)) : m); w.S1[(size_t)sid * nn + e] = dd_val(t0.get(i, j)); }
  dsync();
  nbad += sdd_inv_spd(Z1, t0, t1, n, flag);
  for (int e = tid; e < nn; e += 256) { const int i = e / n, j = e - i * n; const ddv m = M.get(i, j); t0.set(i, j, (i == j) ? dd_sub(dd_from(tau), m) : dd_neg(m)); w.S2[(size_t)sid * nn + e] = dd_val(t0.get(i, j)); }
  dsync();
  nbad += sdd_inv_spd(Z2, t0, t1, n, flag);
  // per cone block: X = mu Z, images; the adjoint pieces accumulate G = X1 - X2 (t2) and Phi (M is free now: M <- Phi)
  double trx2 = 0.0, hby = 0.0, trpsi = 0.0, trphi2 = 0.0, hbphi = 0.0;
  for (int r = 0; r < 2; ++r) {
    sdd Z = r ? Z2 : Z1;
    for (int e = tid; e < nn; e += 256) {
      const int i = e / n, j = e - i * n;
      const ddv x = dd_muld(Z.get(i, j), mu);
      X.set(i, j, x);
      (r ? w.X2 : w.X1)[(size_t)sid * nn + e] = final_sweep ? (r ? w.dX2 : w.dX1)[(size_t)sid * nn + e] : dd_val(x);      // (final sweep: the dual iterate of the last step, see k_polish_step)
      (r ? w.S2i : w.S1i)[(size_t)sid * nn + e] = dd_val(Z.get(i, j));
      if (w.Zdd) { double* zd = w.Zdd + ((size_t)sid * 2 + r) * 2 * nn; zd[e] = Z.h(i, j); zd[nn + e] = Z.l(i, j); }
      t2.set(i, j, r ? dd_sub(t2.get(i, j), x) : x);
    }
    dsync();
    dd_images(w, (size_t)sid, r, V, X, Z, t0, t1, n, nx);
    sdd_mm(t0, X, false, Hd, false, n, n, n); sdd_mm(t1, t0, false, Z, false, n, n, n);      // X Hb Z (symmetric up to rounding: Z, Hb symmetric, X = mu Z)
    for (int e = tid; e < nn; e += 256) {
      const int i = e / n, j = e - i * n;
      const ddv ph = dd_muld(dd_add(t1.get(i, j), t1.get(j, i)), 0.5);
      if (r == 1 && i == j) trphi2 += dd_val(ph);
      M.set(i, j, r ? dd_add(M.get(i, j), ph) : ph);
    }
    dsync();
    if (r == 1) {
      sdd_mm(t0, X, false, Z, false, n, n, n);                                              // Psi = X2 Z2
      for (int e = tid; e < nn; e += 256) { const int i = e / n, j = e - i * n; if (i == j) { trpsi += dd_val(t0.get(i, j)); trx2 += dd_val(X.get(i, j)); } }
      sdd_mm(t1, V, false, t0, false, nx, n, n); sdd_mm(X, t1, false, V, true, nx, n, nx);   // V Psi V'   (X is free)
      sdd_store(w.adjV + ((size_t)sid * NADJ + ADJ_PSI) * nxx, w.adjVl + ((size_t)sid * NADJ + ADJ_PSI) * nxx, X, nx, nx, nx);
      sdd_store(w.adjE + ((size_t)sid * NADJ + ADJ_PSI) * nxx, w.adjEl + ((size_t)sid * NADJ + ADJ_PSI) * nxx, t0, nx, nx, nx);
      dsync();
    }
  }
  // Step 2: the norm terms.  X_e := mu S_e^-1 from the dd closed form (cone check: t > 0, gam > 0); tr(X_e) - 1 is minus the gradient in t_e, 2 wr X_e[0][i+1] joins the
  // one in phi_i -- both in dd, then rounded.
  const PhiStage pst = phi_stage(w, dm, sid);
  __shared__ double arw_gi[4];                        // 1 / gam of the two arrow blocks (dd)
  if (pst.na > 0) {
    const double wr = phi_wr(w, pr);
    __shared__ double sc[8];
    double* Sih = t0.p; double* Sil = t0.p + t0.ms;                  // (t0 is free here: an LDS slot of 32 x 33 = AEL x ARW_LD doubles per plane, or n x n > that of the global scratch at n > 32)
    for (int e_ = 0; e_ < pst.na; ++e_) {
      const int m = pst.am[e_], ne = m + 1, c0 = pst.a0[e_];
      const double te = w.at[(size_t)sid * 2 + e_];
      const ddv gam = arrow_inv_dd(Sih, Sil, te, ph + c0, wr, m, sc);
      nbad += (te > 0.0 && gam.h > 0.0) ? 0 : 1;
      double* aXg = w.aX + ((size_t)sid * 2 + e_) * AE; const double* adXg = w.adX + ((size_t)sid * 2 + e_) * AE;
      for (int e = tid; e < ne * ne; e += 256) {
        const int i = e / ne, j = e - i * ne;
        aXg[i * AEL + j] = final_sweep ? adXg[i * AEL + j] : dd_val(dd_muld(ddv{Sih[i * ARW_LD + j], Sil[i * ARW_LD + j]}, mu));      // (final sweep: the dual iterate of the last Newton step, k_polish_arrows)
      }
      if (tid == 0) {
        ddv tr = ddv{0.0, 0.0};
        for (int i = 0; i < ne; ++i) tr = dd_add(tr, ddv{Sih[i * ARW_LD + i], Sil[i * ARW_LD + i]});
        PSM_RPHI(psm_at(w.psm, dm, sid), dm.nz)[pst.nrow + e_] = dd_val(dd_sub(dd_muld(tr, mu), dd_from(1.0)));
        arw_gi[2 * e_] = sc[2]; arw_gi[2 * e_ + 1] = sc[3];
      }
      __syncthreads();
    }
  }
  if (ngs > 0) {
    // minus the gradient of the dual barrier in phi_i: g_i' (X1 - X2) g_i + mu / phi_i -- the quadratic form in dd (its terms are O(1), the sum ~ mu / phi_i), THEN rounded:
    // the tail of the right-hand side of block k+1 (k_aug_gather).  z_i := mu / phi_i is not an iterate any more (the diagonal z / phi of T_loc,loc: k_phi_pre).
    for (int e = tid; e < ngs * n; e += 256) {
      const int i = e / n, a = e - i * n;
      double sh = 0.0, sl = 0.0;
      for (int c = 0; c < n; ++c) { const ddv y = t2.get(a, c); dd_fma_acc(sh, sl, y.h, y.l, Gg[i * n + c], 0.0); }
      t0.set(i, a, dd_muld(dd_qts(sh, sl), Gg[i * n + a]));
    }
    dsync();
    if (tid < ngs) {
      ddv sq = ddv{0.0, 0.0};
      for (int a = 0; a < n; ++a) sq = dd_add(sq, t0.get(tid, a));
      const ddv zz = dd_div(dd_from(mu), dd_from(ph[tid]));
      for (int e_ = 0; e_ < pst.na; ++e_)
        if (tid >= pst.a0[e_] && tid < pst.a0[e_] + pst.am[e_]) {        // + 2 wr X_e[0][i+1],  X_e[0][i+1] = -mu wr phi_i / gam
          const double wr = phi_wr(w, pr);
          sq = dd_sub(sq, dd_muld(dd_mul(dd_tp(wr, ph[tid]), ddv{arw_gi[2 * e_], arw_gi[2 * e_ + 1]}), 2.0 * wr * mu));
        }
      PSM_RPHI(psm_at(w.psm, dm, sid), dm.nz)[tid] = dd_val(dd_add(sq, zz));
      // (final sweep: the multiplier of the dual iterate that goes with the last Newton step, k_polish_phi)
      w.zph[(size_t)sid * dm.nr + tid] = final_sweep ? w.dzph[(size_t)sid * dm.nr + tid] : dd_val(zz);
    }
    dsync();
  }
  for (int e = tid; e < nn; e += 256) { const int i = e / n, j = e - i * n; hby += dd_val(dd_mul(Hd.get(i, j), t2.get(i, j))); hbphi += dd_val(dd_mul(Hd.get(i, j), M.get(i, j))); }
  sdd_mm(t0, V, false, t2, false, nx, n, n); sdd_mm(t1, t0, false, V, true, nx, n, nx);       // V G V'
  sdd_store(w.adjV + ((size_t)sid * NADJ + ADJ_G) * nxx, w.adjVl + ((size_t)sid * NADJ + ADJ_G) * nxx, t1, nx, nx, nx);
  sdd_store(w.adjE + ((size_t)sid * NADJ + ADJ_G) * nxx, w.adjEl + ((size_t)sid * NADJ + ADJ_G) * nxx, t2, nx, nx, nx);
  dsync();
  sdd_mm(t0, V, false, M, false, nx, n, n); sdd_mm(t1, t0, false, V, true, nx, n, nx);        // V Phi V'
  sdd_store(w.adjV + ((size_t)sid * NADJ + ADJ_PHI) * nxx, w.adjVl + ((size_t)sid * NADJ + ADJ_PHI) * nxx, t1, nx, nx, nx);
  sdd_store(w.adjE + ((size_t)sid * NADJ + ADJ_PHI) * nxx, w.adjEl + ((size_t)sid * NADJ + ADJ_PHI) * nxx, M, nx, nx, nx);
  trx2 = block_sum<256>(trx2); hby = block_sum<256>(hby); trpsi = block_sum<256>(trpsi); trphi2 = block_sum<256>(trphi2); hbphi = block_sum<256>(hbphi);
  if (tid == 0) {
    double* q = w.part + (size_t)sid * NPART;
    q[Q_TRT2] = trx2; q[Q_HBG] = hby; q[Q_TRPSI] = trpsi; q[Q_TRPHI2] = trphi2; q[Q_HBPHI] = hbphi; q[Q_CHOLBAD] = (double)nbad; q[Q_TRX2] = trx2; q[Q_HBY] = hby;
  }
}

// right-hand sides [rhs | u_tau | u_alpha] of the polish step from the dd adjoint pieces: the subtraction adjV[j-1] - adjE[j] in dd, THEN rounded
__global__ void __launch_bounds__(64) k_dd_gather(WS w, Dims dm) {
  const int sid = stage_id(w, dm);
  const int b = sid / dm.p, k = sid - b * dm.p;
  if (w.iprob[(size_t)b * IS + I_PHASE] != PH_POLISH) return;
  const bool chord = w.iprob[(size_t)b * IS + I_CHORD] != 0;
  const int lane = threadIdx.x, nx = dm.nx, nxx = nx * nx, dp = dm.dp;
  const int km = (k == 0) ? dm.p - 1 : k - 1;
  const size_t ov = (size_t)(b * dm.p + km) * NADJ * nxx, oe = (size_t)sid * NADJ * nxx;
  int e = 0;
  for (int a = 0; a < nx; ++a) {
    for (int c = a + lane; c < nx; c += 64) {
      const int idx = e + (c - a), o = a * nx + c;
      const double wgt = (a == c) ? 1.0 : 2.0;
      auto piece = [&](int s) { return dd_val(dd_sub(ddv{w.adjV[ov + s * nxx + o], w.adjVl[ov + s * nxx + o]}, ddv{w.adjE[oe + s * nxx + o], w.adjEl[oe + s * nxx + o]})); };
      if (chord) { w.Z[(size_t)sid * dp + idx] = wgt * piece(ADJ_G); continue; }      // chord step: the gradient alone; border columns, T^-1 U and the 2 x 2 border complement of the last factorisation stay
      const double g = wgt * piece(ADJ_G), ut = -wgt * piece(ADJ_PSI), ua = wgt * piece(ADJ_PHI);
      double* w3 = w.W3 + ((size_t)sid * dp + idx) * 3; w3[0] = g; w3[1] = ut; w3[2] = ua;
      double* u = w.U + ((size_t)sid * dp + idx) * 2; u[0] = ut; u[1] = ua;
    }
    e += nx - a;
  }
  for (int i = dm.d + lane; i < dp; i += 64) {
    if (chord) { w.Z[(size_t)sid * dp + i] = 0.0; continue; }
    double* w3 = w.W3 + ((size_t)sid * dp + i) * 3; w3[0] = 0.0; w3[1] = 0.0; w3[2] = 0.0;
    double* u = w.U + ((size_t)sid * dp + i) * 2; u[0] = 0.0; u[1] = 0.0;
  }
}

// ------------------------------------------------------------------ control of the tight phase
// restart of the problems that ended Optimal: the same loop towards mu_t = tight_tol * kappa, block linear algebra in dd.  One workgroup per problem;
// the result of the default solve is kept (Pdef, P_TAU_DEF, P_ALPHA_DEF, P_MUT1) for k_tight_fallback.
__global__ void __launch_bounds__(64) k_tight_restart(WS w, Dims dm, Opts o) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int* ip = w.iprob + (size_t)b * IS;
  double* pr = w.prob + (size_t)b * PS;
  if (ip[I_PHASE] != PH_DONE || ip[I_EARLY] || ip[I_IPMSTATUS] != IPM_OPTIMAL) return;
  const double mut = exp2(rint(log2(o.tight_tol * fmax(1.0, fabs(pr[P_TAU])))));
  if (!(mut < pr[P_MUT])) return;                 // nothing tighter asked for
  const size_t np_ = (size_t)dm.p * dm.nx * dm.nx;
  for (size_t e = lane; e < np_; e += 64) w.Pdef[(size_t)b * np_ + e] = w.P[(size_t)b * np_ + e];
  for (size_t e = lane; e < (size_t)dm.p * dm.nr; e += 64) w.phidef[(size_t)b * dm.p * dm.nr + e] = w.phi[(size_t)b * dm.p * dm.nr + e];
  if (dm.constr) for (int e = lane; e < dm.p * 2; e += 64) w.atdef[(size_t)b * dm.p * 2 + e] = w.at[(size_t)b * dm.p * 2 + e];
  if (lane != 0) return;
  pr[P_TAU_DEF] = pr[P_TAU]; pr[P_ALPHA_DEF] = pr[P_ALPHA];
  pr[P_MUT1] = pr[P_MUT]; pr[P_MUT] = mut; pr[P_PREVSTEPN] = -1.0;
  ip[I_PHASE] = PH_MAIN; ip[I_IPMSTATUS] = IPM_MAXITER; ip[I_NCENT] = 0; ip[I_CHORD] = 0; ip[I_REG] = 0; ip[I_JAM] = 0; ip[I_SHIFTRUN] = 0; ip[I_BOSTEP] = 0;
  ip[I_DD] = 1;
  const int slot = atomicAdd(w.active, 1); w.alist[slot] = b; w.flist[slot] = b;
}
// A member whose tight phase did not end Optimal (a non-positive dd pivot, a polish that left the cone, the iteration cap: fp64 stage arithmetic has its own
// limits, most visibly on hard targets below 2^-33) gets the result of its default solve back: the mode never returns less than the default does.
// info[6] (mu_target) then shows the default's target; info[10] = 4 (IPM_TIGHT_FALLBACK) and I_DD = 2 mark the member (until round 4 only info[6] told).
__global__ void __launch_bounds__(64) k_tight_fallback(WS w, Dims dm) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int* ip = w.iprob + (size_t)b * IS;
  double* pr = w.prob + (size_t)b * PS;
  if (ip[I_DD] != 1 || (ip[I_PHASE] == PH_DONE && ip[I_IPMSTATUS] == IPM_OPTIMAL)) return;
  const size_t np_ = (size_t)dm.p * dm.nx * dm.nx;
  for (size_t e = lane; e < np_; e += 64) w.P[(size_t)b * np_ + e] = w.Pdef[(size_t)b * np_ + e];
  for (size_t e = lane; e < (size_t)dm.p * dm.nr; e += 64) w.phi[(size_t)b * dm.p * dm.nr + e] = w.phidef[(size_t)b * dm.p * dm.nr + e];
  if (dm.constr) for (int e = lane; e < dm.p * 2; e += 64) w.at[(size_t)b * dm.p * 2 + e] = w.atdef[(size_t)b * dm.p * 2 + e];
  if (lane != 0) return;
  pr[P_TAU] = pr[P_TAU_DEF]; pr[P_ALPHA] = pr[P_ALPHA_DEF]; pr[P_MUT] = pr[P_MUT1]; pr[P_MU] = pr[P_MUT1];
  ip[I_PHASE] = PH_DONE; ip[I_IPMSTATUS] = IPM_TIGHT_FALLBACK; ip[I_DD] = 2;      // (info[10] = 4: a caller who asked for the tight gap can tell that this member has the default one)
}

// polish, after k_dd_polish_pre: scalars of the border system and of the gradient (the slots k_solve_border reads), cone check
__global__ void __launch_bounds__(64) k_polish_ctrl_a(WS w, Dims dm) {
  const int b = prob_id(w), lane = threadIdx.x;
  int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] != PH_POLISH) return;
  double* pr = w.prob + (size_t)b * PS;
  const int p = dm.p;
  const double trpsi = psum(w.part, b, p, Q_TRPSI, lane), trphi2 = psum(w.part, b, p, Q_TRPHI2, lane), hbphi = psum(w.part, b, p, Q_HBPHI, lane);
  const double nbad = psum(w.part, b, p, Q_CHOLBAD, lane);
  if (nbad > 0.0) {
    // the last step left the cone (not seen after the centering phase; the CPU restatement would halve the step): back to the iterate before it
    const int nxx = dm.nx * dm.nx;
    if (ip[I_NPOLISH] > 0) for (int e = lane; e < p * nxx; e += 64) w.P[(size_t)b * p * nxx + e] = w.Pprev[(size_t)b * p * nxx + e];
    if (ip[I_NPOLISH] > 0) for (int e = lane; e < p * dm.nr; e += 64) w.phi[(size_t)b * p * dm.nr + e] = w.corrp[(size_t)b * p * dm.nr + e];
    if (ip[I_NPOLISH] > 0 && dm.constr) for (int e = lane; e < p * 2; e += 64) if ((e & 1) < phi_stage(w, dm, (size_t)b * p + (e >> 1)).na) w.at[(size_t)b * p * 2 + e] = w.acor[((size_t)b * p * 2 + e) * AE];
    if (lane == 0) {
      if (ip[I_NPOLISH] > 0) { pr[P_TAU] = pr[P_TAU_PREV]; pr[P_ALPHA] = pr[P_ALPHA_PREV]; }
      ip[I_PHASE] = PH_DONE; ip[I_IPMSTATUS] = IPM_INACCURATE;
    }
    return;
  }
  if (lane != 0) return;
  const double s0 = pr[P_ALPHA] - ALPHA_MIN, x0 = pr[P_MUT] / s0;
  pr[P_S0] = s0; pr[P_X0] = x0; pr[P_RD0] = 0.0; pr[P_CORR0] = 0.0; pr[P_SIGMU] = pr[P_MUT]; pr[P_MU] = pr[P_MUT];
  pr[P_BTT] = trpsi; pr[P_BTA] = -trphi2; pr[P_BAA] = hbphi + x0 / s0;
  // (I_CHORD: set by k_polish_ctrl_b for the next step -- 0 at the first step of the polish, k_ctrl_d)
}

// polish: the 2 x 2 border system (tau, alpha) and dP = z - T^-1 U db in double-double (k_solve_border of tmpc_factor.h does it in fp64).  The block solutions
// z, T^-1 U arrive as dd numbers (W3 / W3l, Z / Zl); with rows of multipliers in the blocks the border columns reach 1e12 and U' T^-1 U cancels against B to ten digits
// and more: in fp64 the step (dtau, dalpha) -- and with it the exported dual iterate X + dX, which takes the step at full weight -- was good to 1e-3 only on some
// members (measured: dual residuals 1e-3 instead of 1e-12; the iterate itself does not care, Newton's fixed point is the gradient's).  One workgroup per problem.
// The low words of dtau, dalpha and of the border complement (kept for the chord steps) live in the part[] slots of stage 0 that the polish does not use.
__device__ __forceinline__ ddv wg_reduce_dd(double sh, double sl, double* red) {       // red: 512 doubles
  const int tid = threadIdx.x;
  __syncthreads();
  red[tid] = sh; red[256 + tid] = sl;
  __syncthreads();
  ddv s = ddv{0.0, 0.0};
  if (tid == 0) { for (int i = 0; i < 256; ++i) s = dd_add(s, dd_ts(red[i], red[256 + i])); red[0] = s.h; red[256] = s.l; }
  __syncthreads();
  s = ddv{red[0], red[256]};
  return s;
}
__global__ void __launch_bounds__(256) k_dd_solve_border(WS w, Dims dm, const int* alist) {
  __shared__ double red[512];
  const int b = alist[blockIdx.x], tid = threadIdx.x;
  const int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] != PH_POLISH) return;
  double* pr = w.prob + (size_t)b * PS;
  const int p = dm.p, dp = dm.dp, nx = dm.nx, d = dm.d;
  const size_t vl = (size_t)p * dp;
  const double* W3 = w.W3 + (size_t)b * vl * 3; const double* W3l = w.W3l + (size_t)b * vl * 3;
  double* Z = w.Z + (size_t)b * vl; double* Zl = w.Zl + (size_t)b * vl;
  double* TU = w.TU + (size_t)b * vl * 2; double* TUl = w.TUl + (size_t)b * vl * 2;
  const double* U = w.U + (size_t)b * vl * 2;
  double* lo = w.part + (size_t)b * p * NPART;            // stage 0: Q_MINX, Q_MINS, Q_DXS = low words of Sb00, Sb01, Sb11; Q_XDS, Q_DXDS = of dtau, dalpha
  if (!ip[I_CHORD]) {
    double h00 = 0.0, l00 = 0.0, h01 = 0.0, l01 = 0.0, h11 = 0.0, l11 = 0.0;
    for (size_t e = tid; e < vl; e += 256) {
      const ddv t0 = ddv{W3[e * 3 + 1], W3l[e * 3 + 1]}, t1 = ddv{W3[e * 3 + 2], W3l[e * 3 + 2]};
      TU[e * 2] = t0.h; TUl[e * 2] = t0.l; TU[e * 2 + 1] = t1.h; TUl[e * 2 + 1] = t1.l;
      Z[e] = W3[e * 3]; Zl[e] = W3l[e * 3];
      dd_fma_acc(h00, l00, t0.h, t0.l, U[e * 2], 0.0); dd_fma_acc(h01, l01, t1.h, t1.l, U[e * 2], 0.0); dd_fma_acc(h11, l11, t1.h, t1.l, U[e * 2 + 1], 0.0);
    }
    const ddv s00 = wg_reduce_dd(h00, l00, red), s01 = wg_reduce_dd(h01, l01, red), s11 = wg_reduce_dd(h11, l11, red);
    if (tid == 0) {
      const ddv a = dd_sub(dd_from(pr[P_BTT]), s00), bb = dd_sub(dd_from(pr[P_BTA]), s01), c = dd_sub(dd_from(pr[P_BAA]), s11);
      pr[P_SB00] = a.h; pr[P_SB01] = bb.h; pr[P_SB11] = c.h; lo[Q_MINX] = a.l; lo[Q_MINS] = bb.l; lo[Q_DXS] = c.l;
    }
  }
  __syncthreads();
  double uh0 = 0.0, ul0 = 0.0, uh1 = 0.0, ul1 = 0.0;
  for (size_t e = tid; e < vl; e += 256) { dd_fma_acc(uh0, ul0, Z[e], Zl[e], U[e * 2], 0.0); dd_fma_acc(uh1, ul1, Z[e], Zl[e], U[e * 2 + 1], 0.0); }
  const ddv u0 = wg_reduce_dd(uh0, ul0, red), u1 = wg_reduce_dd(uh1, ul1, red);
  double trt2 = 0.0, hbg = 0.0;
  for (int k = tid; k < p; k += 256) { const double* q = w.part + (size_t)(b * p + k) * NPART; trt2 += q[Q_TRT2]; hbg += q[Q_HBG]; }
  const ddv st = wg_reduce_dd(trt2, 0.0, red), sh_ = wg_reduce_dd(hbg, 0.0, red);
  const double t0 = pr[P_SIGMU] / pr[P_S0] - pr[P_X0] * pr[P_RD0] / pr[P_S0] - pr[P_CORR0];
  const ddv rb0 = dd_sub(dd_sub(st, dd_from(1.0)), u0), rb1 = dd_sub(dd_add(sh_, dd_from(t0)), u1);
  const ddv a = ddv{pr[P_SB00], lo[Q_MINX]}, bb = ddv{pr[P_SB01], lo[Q_MINS]}, c = ddv{pr[P_SB11], lo[Q_DXS]};
  const ddv det = dd_sub(dd_mul(a, c), dd_mul(bb, bb));
  const ddv dtau = dd_div(dd_sub(dd_mul(c, rb0), dd_mul(bb, rb1)), det), dalpha = dd_div(dd_sub(dd_mul(a, rb1), dd_mul(bb, rb0)), det);
  __syncthreads();
  if (tid == 0) { pr[P_DTAU] = dtau.h; pr[P_DALPHA] = dalpha.h; lo[Q_XDS] = dtau.l; lo[Q_DXDS] = dalpha.l; }
  double* dPg = w.dP + (size_t)b * p * nx * nx;
  for (int e = tid; e < p * d; e += 256) {
    const int k = e / d, idx = e - k * d;
    int a2 = 0, rem = idx;
    while (rem >= nx - a2) { rem -= nx - a2; ++a2; }
    const int c2 = a2 + rem;
    const size_t vi = (size_t)k * dp + idx;
    const double v = dd_val(dd_sub(dd_sub(ddv{Z[vi], Zl[vi]}, dd_mul(ddv{TU[vi * 2], TUl[vi * 2]}, dtau)), dd_mul(ddv{TU[vi * 2 + 1], TUl[vi * 2 + 1]}, dalpha)));
    dPg[(size_t)k * nx * nx + a2 * nx + c2] = v;
    dPg[(size_t)k * nx * nx + c2 * nx + a2] = v;
  }
}

// polish with rows of G, after k_solve_border: dphi = the tail of the block solution (as k_phi_dir does in the loop), and the multiplier z + dz = mu / phi - mu dphi / phi^2
// of the dual iterate that goes with a Newton step (exported by tmpc_get_dual_con_host; a chord step keeps the one of the last Newton step, see k_polish_ctrl_b)
__global__ void __launch_bounds__(64) k_polish_phi(WS w, Dims dm) {
  const int sid = stage_id(w, dm), lane = threadIdx.x;
  const int b = sid / dm.p, k = sid - b * dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] != PH_POLISH) return;
  const double* pr = w.prob + (size_t)b * PS;
  const int kn = (k + 1 == dm.p) ? 0 : k + 1;
  if (lane < stage_rows(w, dm, sid)) {
    const size_t vi = ((size_t)b * dm.p + kn) * dm.dp + dm.d + lane, gi = (size_t)sid * dm.nr + lane;
    const double* lo = w.part + (size_t)b * dm.p * NPART;      // low words of dtau, dalpha (k_dd_solve_border)
    const ddv dtau = ddv{pr[P_DTAU], lo[Q_XDS]}, dalpha = ddv{pr[P_DALPHA], lo[Q_DXDS]};
    const double v = dd_val(dd_sub(dd_sub(ddv{w.Z[vi], w.Zl[vi]}, dd_mul(ddv{w.TU[vi * 2], w.TUl[vi * 2]}, dtau)), dd_mul(ddv{w.TU[vi * 2 + 1], w.TUl[vi * 2 + 1]}, dalpha)));
    w.dphi[gi] = v;
    const double phi = w.phi[gi], mu = pr[P_MUT];
    if (!ip[I_CHORD]) w.dzph[gi] = mu / phi - mu * v / (phi * phi);
  }
  const PhiStage ps = phi_stage(w, dm, sid);
  if (lane < ps.na) {                                     // epigraph variables: local variables nrow + e
    const double* lo = w.part + (size_t)b * dm.p * NPART;
    const ddv dtau = ddv{pr[P_DTAU], lo[Q_XDS]}, dalpha = ddv{pr[P_DALPHA], lo[Q_DXDS]};
    const size_t vi = ((size_t)b * dm.p + kn) * dm.dp + dm.d + ps.nrow + lane;
    w.adt[(size_t)sid * 2 + lane] = dd_val(dd_sub(dd_sub(ddv{w.Z[vi], w.Zl[vi]}, dd_mul(ddv{w.TU[vi * 2], w.TUl[vi * 2]}, dtau)), dd_mul(ddv{w.TU[vi * 2 + 1], w.TUl[vi * 2 + 1]}, dalpha)));
  }
}

// polish on the Step 2 model, after k_polish_phi: the primal block of every norm term that goes with a Newton step, X_e + dX_e = mu S_e^-1 - mu sym(S_e^-1 dS_e S_e^-1)
// with dS_e = arrow(dt_e, wr dphi) -- it satisfies tr(X_e) = 1 and the stationarity rows of its phi_i to rounding, as X_r + dX_r does for the LMI blocks
// (k_polish_step); S_e^-1 from the dd closed form.  Exported by tmpc_get_dual_con_host after the final sweep.
__global__ void __launch_bounds__(256) k_polish_arrows(WS w, Dims dm) {
  __shared__ double Sih[AEL * ARW_LD], Sil[AEL * ARW_LD], sc[8], dsv[AEL];
  const int sid = stage_id(w, dm), tid = threadIdx.x;
  const int b = sid / dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] != PH_POLISH || ip[I_CHORD]) return;
  const double* pr = w.prob + (size_t)b * PS;
  const PhiStage ps = phi_stage(w, dm, sid);
  const double wr = phi_wr(w, pr), mu = pr[P_MUT];
  const double* phi = w.phi + (size_t)sid * dm.nr; const double* dph = w.dphi + (size_t)sid * dm.nr;
  for (int e_ = 0; e_ < ps.na; ++e_) {
    const int m = ps.am[e_], ne = m + 1, c0 = ps.a0[e_];
    arrow_inv_dd(Sih, Sil, w.at[(size_t)sid * 2 + e_], phi + c0, wr, m, sc);
    if (tid < ne) dsv[tid] = tid ? wr * dph[c0 + tid - 1] : w.adt[(size_t)sid * 2 + e_];      // dS: diagonal dsv[0], first row / column dsv[1..m]
    __syncthreads();
    auto SI = [&](int i, int j) { return ddv{Sih[i * ARW_LD + j], Sil[i * ARW_LD + j]}; };
    double* out = w.adX + ((size_t)sid * 2 + e_) * AE;
    for (int e = tid; e < ne * ne; e += 256) {
      const int i = e / ne, j = e - i * ne;
      if (j > i) continue;
      // (S^-1 dS S^-1)[i][j] = dt sum_a Si[i][a] Si[a][j] + sum_{q >= 1} ds_q (Si[i][0] Si[q][j] + Si[i][q] Si[0][j])   (symmetric)
      ddv acc = ddv{0.0, 0.0}, s2 = ddv{0.0, 0.0};
      for (int a = 0; a < ne; ++a) s2 = dd_add(s2, dd_mul(SI(i, a), SI(a, j)));
      acc = dd_muld(s2, dsv[0]);
      for (int q = 1; q < ne; ++q) acc = dd_add(acc, dd_muld(dd_add(dd_mul(SI(i, 0), SI(q, j)), dd_mul(SI(i, q), SI(0, j))), dsv[q]));
      const double v = dd_val(dd_muld(dd_sub(SI(i, j), acc), mu));
      out[i * AEL + j] = v; out[j * AEL + i] = v;
    }
    __syncthreads();
  }
}

// polish, after k_solve_border: dM of the step and the norms of the step test (M itself: T1, written by k_dd_polish_pre)
__global__ void __launch_bounds__(256) k_polish_step(WS w, Dims dm) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int sid = stage_id(w, dm);
  const int b = sid / dm.p, k = sid - b * dm.p;
  if (w.iprob[(size_t)b * IS + I_PHASE] != PH_POLISH) return;
  const double* pr = w.prob + (size_t)b * PS;
  const int lane = threadIdx.x, n = dm.n, nx = dm.nx, nn = n * n, nxx = nx * nx;
  const int kn = (k + 1 == dm.p) ? 0 : k + 1;
  double* sV = sm; double* sM = sm + MS; double* t0 = sm + 2 * MS; double* t1 = sm + 3 * MS; double* sHb = sm + 4 * MS;
  g2s<256>(sV, w.V + (size_t)sid * nx * n, nx, n, n, lane);
  g2s<256>(sHb, w.Hb + (size_t)sid * nn, n, n, n, lane);
  wsync();
  build_M<256>(sM, sV, t0, t1, sHb, w.dP + (size_t)sid * nxx, w.dP + (size_t)(b * dm.p + kn) * nxx, pr[P_DALPHA], n, nx, lane);
  if (dm.nr > 0) {                       // + sum_i dphi_i g_i g_i'
    const int ngs = stage_rows(w, dm, sid);
    const double* Gg = w.G + (size_t)sid * dm.nr * n; const double* dph = w.dphi + (size_t)sid * dm.nr;
    for (int e = lane; e < nn; e += 256) {
      int i, j; ediv(e, n, i, j);
      double v = sM[i * LD + j];
      for (int r_ = 0; r_ < ngs; ++r_) v = fma(dph[r_] * Gg[r_ * n + i], Gg[r_ * n + j], v);
      sM[i * LD + j] = v;
    }
    wsync();
  }
  const double ra = pr[P_DALPHA] / pr[P_ALPHA];
  double dh2 = 0.0, m2 = 0.0;
  for (int e = lane; e < nn; e += 256) {
    int i, j; ediv(e, n, i, j);
    const double m = w.T1[(size_t)sid * nn + e], dh = sM[i * LD + j] - ra * m;
    dh2 = fma(dh, dh, dh2); m2 = fma(m, m, m2);
  }
  dh2 = block_sum<256>(dh2); m2 = block_sum<256>(m2);
  if (lane == 0) { double* q = w.part + (size_t)sid * NPART; q[Q_DH2] = dh2; q[Q_M2] = m2; }
  // The dual iterate that goes with this step: X_r + dX_r = X_r - sym(X_r dS_r S_r^-1) (X_r = mu S_r^-1: the full primal-dual step).  It satisfies
  // the linear dual equations EXACTLY whatever y is, while X = mu S(y)^-1 at an fp64 y misses them by eps / mu in the active directions
  // (1e-5 at mu = 3e-11): this is the X that tmpc_get_dual_host exports with the last step (dX1 / dX2 hold it until the final sweep).
  const double dtau = pr[P_DTAU];
  const bool chord = w.iprob[(size_t)b * IS + I_CHORD] != 0;
  double* sX = sV; double* sZ = sHb;                 // V and Hb are not needed any more
  for (int r = 0; r < 2; ++r) {
    wsync();
    g2s<256>(sX, (r ? w.X2 : w.X1) + (size_t)sid * nn, n, n, n, lane);
    g2s<256>(sZ, (r ? w.S2i : w.S1i) + (size_t)sid * nn, n, n, n, lane);
    if (r == 1) for (int e = lane; e < nn; e += 256) { int i, j; ediv(e, n, i, j); sM[i * LD + j] = ((i == j) ? dtau : 0.0) - sM[i * LD + j]; }      // dS2 = dtau I - dM
    wsync();
    mm<256>(t0, sX, LD, 1, sM, LD, 1, n, n, n, 0, lane);
    mm<256>(t1, t0, LD, 1, sZ, LD, 1, n, n, n, 0, lane);
    double* out = (r ? w.dX2 : w.dX1) + (size_t)sid * nn;
    if (!chord) for (int e = lane; e < nn; e += 256) { int i, j; ediv(e, n, i, j); out[e] = sX[i * LD + j] - 0.5 * (t1[i * LD + j] + t1[j * LD + i]); }      // (a chord step keeps the dual iterate of the last Newton step: k_polish_ctrl_b)
    if (r == 0) { wsync(); }
  }
}

// polish, end of a step: the (full) step is taken; done when it was smaller than center_tol.  Rebuilds the list of problems still polishing.
// Chord steps (round 5): the Hessian of the dual barrier at the next iterate differs from the factored one by about the size of the step just taken (1e-5 ... 1e-9
// here), so Newton with the OLD double-double factorisation contracts by that factor -- the second polish step, whose only job is to confirm convergence, and most
// third ones cost a substitution instead of a factorisation.  A chord step that contracts by less than 1/4 sends the problem back to a new factorisation (fac / nfac).
__global__ void __launch_bounds__(64) k_polish_ctrl_b(WS w, Dims dm, Opts o, const int* list, int count, int* next, int* nnext, int* fac, int* nfac) {
  if ((int)blockIdx.x >= count) return;
  const int b = list[blockIdx.x], lane = threadIdx.x;
  int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] != PH_POLISH) return;
  double* pr = w.prob + (size_t)b * PS;
  const int p = dm.p, nxx = dm.nx * dm.nx;
  const double dh2 = psum(w.part, b, p, Q_DH2, lane), m2 = psum(w.part, b, p, Q_M2, lane);
  const double stepn = sqrt(dh2 / m2);
  const bool fin = (fabs(pr[P_DTAU]) < 1e300) && (fabs(pr[P_DALPHA]) < 1e300) && (stepn == stepn) && (stepn < 1e300);
  if (fin) for (int e = lane; e < p * nxx; e += 64) { const size_t g = (size_t)b * p * nxx + e; w.Pprev[g] = w.P[g]; w.P[g] += w.dP[g]; }
  if (fin) for (int e = lane; e < p * dm.nr; e += 64) { const size_t g = (size_t)b * p * dm.nr + e; w.corrp[g] = w.phi[g]; w.phi[g] += w.dphi[g]; }      // (corrp: free in the polish -- phi before the step)
  if (fin && dm.constr) for (int e = lane; e < p * 2; e += 64) {      // (acor[0] of the block: free in the polish -- t_e before the step)
    const size_t g = (size_t)b * p * 2 + e;
    if ((e & 1) < phi_stage(w, dm, (size_t)b * p + (e >> 1)).na) { w.acor[g * AE] = w.at[g]; w.at[g] += w.adt[g]; }
  }
  if (lane != 0) return;
  if (!fin) { ip[I_PHASE] = PH_DONE; ip[I_IPMSTATUS] = IPM_INACCURATE; return; }
  pr[P_TAU_PREV] = pr[P_TAU]; pr[P_ALPHA_PREV] = pr[P_ALPHA];
  // x0 + dx0: the scalar of the dual iterate that goes with this step (k_polish_step).  Only a NEWTON step yields a dual iterate that satisfies the linear dual
  // equations exactly; a chord step misses them by (H - H_0) dy, so the exported dual iterate stays the one of the last Newton step -- weak duality pairs ANY dual
  // feasible point with the final primal one (round 5; with the chord step's iterate the certified gap of tests/test_gpu_tight.py widened from 4.4e-9 to 5.6e-9)
  if (!ip[I_CHORD]) pr[P_DX0] = pr[P_X0] - pr[P_X0] * pr[P_DALPHA] / pr[P_S0];
  pr[P_TAU] += pr[P_DTAU]; pr[P_ALPHA] += pr[P_DALPHA];
  pr[P_STEPN] = stepn; pr[P_AP] = 1.0; pr[P_AD] = 1.0;
  ip[I_NPOLISH] += 1; ip[I_ITERS] += 1;
  if (w.trace && ip[I_ITERS] >= 1 && ip[I_ITERS] <= TRACE_LEN) {
    double* t = w.trace + ((size_t)b * TRACE_LEN + (ip[I_ITERS] - 1)) * TRACE_W;
    t[0] = (double)ip[I_ITERS]; t[1] = (double)PH_POLISH; t[2] = pr[P_MUT]; t[3] = pr[P_TAU]; t[4] = 0.0; t[5] = 0.0; t[6] = 1.0; t[7] = 1.0; t[8] = stepn; t[9] = 0.0;
  }
  const bool was_chord = ip[I_CHORD] != 0;
  const double prev = pr[P_PREVSTEPN];
  pr[P_PREVSTEPN] = stepn;
  if (stepn < o.center_tol) { ip[I_PHASE] = PH_DONE; ip[I_IPMSTATUS] = IPM_OPTIMAL; ip[I_CHORD] = 0; }       // (k_dd_polish_pre of the final sweep checks the cone at the new point)
  else if (ip[I_NPOLISH] >= POLISH_MAX) { ip[I_PHASE] = PH_DONE; ip[I_IPMSTATUS] = IPM_INACCURATE; ip[I_CHORD] = 0; }
  else {
    const bool chord_next = (o.chord_step > 0.0) && (!was_chord || (prev > 0.0 && stepn <= 0.25 * prev));
    ip[I_CHORD] = chord_next ? 1 : 0;
    if (chord_next) ip[I_NCHORD] += 1;
    const int slot = atomicAdd(nnext, 1); next[slot] = b;
    if (!chord_next) { const int fs = atomicAdd(nfac, 1); fac[fs] = b; }
  }
}

// after the final sweep of k_dd_polish_pre: a last step that left the cone is undone (status Feasible)
__global__ void __launch_bounds__(64) k_polish_final(WS w, Dims dm) {
  const int b = prob_id(w), lane = threadIdx.x;
  int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] != PH_DONE || ip[I_NPOLISH] == 0 || ip[I_IPMSTATUS] != IPM_OPTIMAL) return;
  double* pr = w.prob + (size_t)b * PS;
  const int p = dm.p, nxx = dm.nx * dm.nx;
  const double nbad = psum(w.part, b, p, Q_CHOLBAD, lane);
  if (!(nbad > 0.0)) { if (lane == 0) { pr[P_X0] = pr[P_DX0]; pr[P_S0] = pr[P_ALPHA] - ALPHA_MIN; pr[P_MU] = pr[P_MUT]; } return; }
  for (int e = lane; e < p * nxx; e += 64) w.P[(size_t)b * p * nxx + e] = w.Pprev[(size_t)b * p * nxx + e];
  for (int e = lane; e < p * dm.nr; e += 64) w.phi[(size_t)b * p * dm.nr + e] = w.corrp[(size_t)b * p * dm.nr + e];
  if (dm.constr) for (int e = lane; e < p * 2; e += 64) if ((e & 1) < phi_stage(w, dm, (size_t)b * p + (e >> 1)).na) w.at[(size_t)b * p * 2 + e] = w.acor[((size_t)b * p * 2 + e) * AE];
  if (lane == 0) { pr[P_TAU] = pr[P_TAU_PREV]; pr[P_ALPHA] = pr[P_ALPHA_PREV]; ip[I_IPMSTATUS] = IPM_INACCURATE; }
}

}  // namespace tmpc

#pragma clang fp contract(fast)
