// Schur-complement assembly and per-problem control kernels.
//
// The HKM Schur matrix of the convexifier SDP is block-cyclic-tridiagonal in the stage index because
// stage k touches only P_k and P_{k+1} (reference: convexifier.py:335-336).  Every d x d block is a
// sum of symmetric Kronecker products of nx x nx matrices, so it is assembled in O(d^2) from the
// Kronecker factors written by k_stage_pre (no PICOS-style symbolic model is ever materialised;
// reference: convexifier.py:213-357).
#pragma once
#include "tmpc_common.h"
#include "tmpc_small.h"

namespace tmpc {

#ifndef TMPC_POLISH_ENTER
#define TMPC_POLISH_ENTER 1e-4
#endif
constexpr double POLISH_ENTER_GPU = TMPC_POLISH_ENTER;      // tight phase: a full centering step this small hands the problem to the polish (tmpc_dd.h)
constexpr int MUT_BACKOFF_MAX = 10;      // mu_t back-offs per problem (2^10: relative gap <= ~1e-4 * default at the worst)

// HKM block entry 0.5*(T(Lx,Ls)+T(Ls,Lx))[(ab),(cd)] with T(L,R)[(ab),(cd)] = <E_ab, L E_cd R'>; the two T's
// consist of the same four products, so the entry is T(Lx,Ls)[(ab),(cd)] = w_ab w_cd hkm_t(...), w = 1/2 on the diagonal pairs.
// One workgroup per (problem, stage): D_k, and the coupling block C_k = T[P_k, P_{k+1}]
// (stored transposed as the sub-diagonal block O_k = T[P_{k+1},P_k], or, for k = p-1, untransposed as the
// cyclic corner F_0 = T[P_{p-1}, P_0]).
// An entry needs the elements (a,c), (a,d), (b,c), (b,d) of twelve nx x nx Kronecker factors and a handful of flops: the kernel is
// bound by instruction issue and LDS bytes, then by the HBM writes (1.08 MB per stage).  Layout and walk are chosen for that:
//   * the factors sit interleaved in LDS, one record of 12 doubles per (i, j): XXX, SIXX, KX, KS of both LMIs, then FX, FS of both;
//     a record comes in by ds_read_b128 (the first form read 48 separate doubles from 12 matrix images and spent two thirds of its
//     issue slots on integer address work: profiles/r2_final_pmc.txt, 96 % issue-bound);
//   * a thread owns one stored COLUMN (its pair (cd) and weights stay in registers) and the workgroup walks the stored rows
//     together: the pair (ab) of the row is wave-uniform, the records (a, c), (a, d) stay in registers for the 24 - a rows that share a,
//     only (b, c) and (b, d) are fetched per entry (half the LDS bytes of an entry-by-entry walk), and every store is 64 consecutive
//     doubles of one row.  320 threads cover the 304 columns of the bench shape in one pass.
//   * two launches, one per block (PART 0: D_k, records of the eight D factors; PART 1: C_k, records of the four F factors): the LDS
//     image of one part is 46 KB / 28 KB instead of 65 KB, three / five workgroups per CU instead of two.  Record strides of 80 and
//     48 bytes keep the 16-byte reads of 16 different columns on different banks.
constexpr int SCH_NT = 320;
template <int PART> constexpr int sch_rec() { return PART == 0 ? 10 : 6; }      // doubles per record (8 / 4 used)
__device__ __forceinline__ double hkm_t(double xac, double xad, double xbc, double xbd, double sac, double sad, double sbc, double sbd) {
  return (xac * sbd + xad * sbc) + (xbc * sad + xbd * sac);
}
// GF: the interleaved factor records live in a per-stage global scratch (w.sscr) instead of LDS -- nx > 43 at n > 32, where 10 nx^2 doubles exceed the LDS; same code,
// separate instantiation (the LDS form is untouched)
// (body: NTS threads assemble the block(s) of stage sid; NTS = SCH_NT: the kernel below, NTS = 64: one wave of the persistent kernel of tmpc_persist.h)
template <int PART, bool GF, int NTS>
__device__ __forceinline__ void schur_body(const WS& w, const Dims& dm, int sid, int tid, double* sm) {
  constexpr int REC = sch_rec<PART>();
  const int b = sid / dm.p, k = sid - b * dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] == PH_DONE) return;
  const int nx = dm.nx, nxx = nx * nx, d = dm.d, dp = dm.dp;
  // relative lift of the diagonal after an iteration whose factorisation froze pivots (k_ctrl_c): 1e-12
  const int reg = ip[I_REG];
  const double regf = (reg <= 0) ? 1.0 : 1.0 + 1e-13 * ((reg == 1) ? 10.0 : (reg == 2) ? 100.0 : 1000.0);
  double* mats = GF ? w.sscr + (size_t)sid * 10 * nxx : sm;                  // [nx][nx][REC]
  unsigned* pair = (unsigned*)(GF ? sm : sm + (size_t)nxx * REC);            // [d]: a | b << 16 of the packed index (ab), a <= b
  // orientation of the stored coupling block (edge slot k of tmpc_cr.h): its columns belong to the stage that the cyclic
  // reduction eliminates first.  `corner` = stored as C_k = T[P_k,P_{k+1}], otherwise transposed, O_k = T[P_{k+1},P_k]:
  // stored entry (row, col) = C_k[col][row], i.e. the same expression on the TRANSPOSED factors F', so those go into LDS
  // transposed and every block uses the pair of the stored row as (ab) and the pair of the stored column as (cd).
  const bool corner = (w.cr_orient[k] != 0);
  const int km = (k == 0) ? dm.p - 1 : k - 1;
  const double* kfk = w.KF + (size_t)sid * 12 * nxx;
  const double* kfm = w.KF + (size_t)(b * dm.p + km) * 12 * nxx;
  for (int e = tid; e < 12 * nxx; e += NTS) {
    const int m = e / nxx, r = e - m * nxx;
    const int lmi = m / KF_PER_LMI, slot = m - lmi * KF_PER_LMI;
    const bool fmat = (slot == KF_FX || slot == KF_FS);
    if (fmat != (PART == 1)) continue;
    const double v = (slot == KF_KX || slot == KF_KS) ? kfm[e] : kfk[e];
    int i = r / nx, j = r - i * nx;
    if (fmat && !corner) { const int t_ = i; i = j; j = t_; }
    const int q = fmat ? 2 * lmi + (slot - KF_FX) : 4 * lmi + slot;
    mats[(size_t)(i * nx + j) * REC + q] = v;
  }
  if (tid < nx) {
    int e = tid * nx - (tid * (tid - 1)) / 2;
    for (int c = tid; c < nx; ++c) pair[e++] = (unsigned)tid | ((unsigned)c << 16);
  }
  if (NTS == 64) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // one wave: its own LDS writes, in order
  else __syncthreads();
  double* Dg = w.D + (size_t)sid * dp * dp;
  double* Cg = w.O + (size_t)sid * dp * dp;
  double* dd = w.Ddiag + (size_t)sid * dp;
  // D_k: only the lower triangle is ever read (the row strips of the block Cholesky stop at the diagonal tile, whose lower half is
  // used), so only col <= row is computed and written.  Rows and columns beyond d: identity / zero padding.
  for (int col = tid; col < dp; col += NTS) {
    const bool cin = col < d;
    const unsigned pc_ = cin ? pair[col] : 0u;
    const int c = (int)(pc_ & 0xffffu), d_ = (int)(pc_ >> 16);
    const double wc = (c == d_) ? 0.5 : 1.0;
    const double2_t* rc = (const double2_t*)(mats + (size_t)c * REC);           // + i * nx * REC: record (i, c)
    const double2_t* rd = (const double2_t*)(mats + (size_t)d_ * REC);
    constexpr int NQ = PART == 0 ? 4 : 2;                   // double2 per record
    double2_t ac[NQ], ad[NQ];
    int a_prev = -1;
    const int row0 = (PART == 0) ? (col / 64) * 64 : 0;     // D: rows above the wave's first column have nothing to write
    for (int row = row0; row < dp; ++row) {
      const size_t e = (size_t)row * dp + col;
      if (row >= d || !cin) {
        if (PART == 0) { if (col <= row) Dg[e] = (row == col) ? 1.0 : 0.0; if (row == col) dd[row] = 1.0; }
        else Cg[e] = 0.0;
        continue;
      }
      const unsigned pr_ = (unsigned)__builtin_amdgcn_readfirstlane((int)pair[row]);
      const int a = (int)(pr_ & 0xffffu), bb = (int)(pr_ >> 16);
      const int ro = nx * (REC / 2);                        // double2 per row of records
      if (a != a_prev) {
        a_prev = a;
#pragma unroll
        for (int q = 0; q < NQ; ++q) { ac[q] = rc[a * ro + q]; ad[q] = rd[a * ro + q]; }
      }
      const double wgt = ((a == bb) ? 0.5 : 1.0) * wc;
      if (PART == 1) {                                      // coupling block: -(T(FX_0, FS_0) + T(FX_1, FS_1))
        const double2_t bc0 = rc[bb * ro], bd0 = rd[bb * ro], bc1 = rc[bb * ro + 1], bd1 = rd[bb * ro + 1];
        double cv = 0.0;
        cv -= wgt * hkm_t(ac[0][0], ad[0][0], bc0[0], bd0[0], ac[0][1], ad[0][1], bc0[1], bd0[1]);
        cv -= wgt * hkm_t(ac[1][0], ad[1][0], bc1[0], bd1[0], ac[1][1], ad[1][1], bc1[1], bd1[1]);
        Cg[e] = cv;
      } else if (col <= row) {
        double dv = 0.0;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const double2_t bc0 = rc[bb * ro + 2 * r], bd0 = rd[bb * ro + 2 * r];                  // XXX, SIXX
          const double2_t bc1 = rc[bb * ro + 2 * r + 1], bd1 = rd[bb * ro + 2 * r + 1];          // KX, KS
          dv += wgt * hkm_t(ac[2 * r][0], ad[2 * r][0], bc0[0], bd0[0], ac[2 * r][1], ad[2 * r][1], bc0[1], bd0[1]);
          dv += wgt * hkm_t(ac[2 * r + 1][0], ad[2 * r + 1][0], bc1[0], bd1[0], ac[2 * r + 1][1], ad[2 * r + 1][1], bc1[1], bd1[1]);
        }
        if (row == col) { dv *= regf; dd[row] = dv; }
        Dg[e] = dv;
      }
    }
  }
}

template <int PART, bool GF = false>
__global__ void __launch_bounds__(SCH_NT) k_schur(WS w, Dims dm) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  schur_body<PART, GF, SCH_NT>(w, dm, stage_id(w, dm), threadIdx.x, sm);
}

// Gather the adjoint pieces into svec right-hand sides:  v_j[(ab)] = w_ab * (adjV[j-1] - adjE[j])[a][b]
// which: 0 -> Z (pass-2 rhs), 1 -> W3 (pass-1: rhs | u_tau | u_alpha) and U
__device__ __forceinline__ void gather_body(const WS& w, const Dims& dm, int sid, int lane, int pass) {
  const int b = sid / dm.p, k = sid - b * dm.p;
  const int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE) return;
  // main phase: pass 1 solves [predictor rhs | u_tau | u_alpha], pass 2 the corrector rhs alone;
  // centering phase: no predictor, so pass 1 is skipped and pass 2 solves [rhs | u_tau | u_alpha] in one sweep
  if (pass == 1 && phase != PH_MAIN) return;
  const bool three = (pass == 1) || (phase != PH_MAIN && !ip[I_CHORD]);     // chord step: the border columns of the last factorisation stay
  const int nx = dm.nx, nxx = nx * nx, dp = dm.dp;
  const int km = (k == 0) ? dm.p - 1 : k - 1;
  const double* av = w.adjV + (size_t)(b * dm.p + km) * NADJ * nxx;
  const double* ae = w.adjE + (size_t)sid * NADJ * nxx;
  // enumerate (a<=b) pairs: e -> (a,b)
  int e = 0;
  for (int a = 0; a < nx; ++a) {
    for (int c = a + lane; c < nx; c += 64) {
      const int idx = e + (c - a);
      const double wgt = (a == c) ? 1.0 : 2.0;
      const int o = a * nx + c;
      const double g = wgt * (av[ADJ_G * nxx + o] - ae[ADJ_G * nxx + o]);
      if (three) {
        const double ut = -wgt * (av[ADJ_PSI * nxx + o] - ae[ADJ_PSI * nxx + o]);
        const double ua = wgt * (av[ADJ_PHI * nxx + o] - ae[ADJ_PHI * nxx + o]);
        double* w3 = w.W3 + ((size_t)sid * dp + idx) * 3;
        w3[0] = g; w3[1] = ut; w3[2] = ua;
        double* u = w.U + ((size_t)sid * dp + idx) * 2;
        u[0] = ut; u[1] = ua;
      } else {
        w.Z[(size_t)sid * dp + idx] = g;
      }
    }
    e += nx - a;
  }
  // zero padding
  for (int i = dm.d + lane; i < dp; i += 64) {
    if (three) {
      double* w3 = w.W3 + ((size_t)sid * dp + i) * 3; w3[0] = 0.0; w3[1] = 0.0; w3[2] = 0.0;
      double* u = w.U + ((size_t)sid * dp + i) * 2; u[0] = 0.0; u[1] = 0.0;
    } else w.Z[(size_t)sid * dp + i] = 0.0;
  }
}

__global__ void __launch_bounds__(64) k_gather(WS w, Dims dm, int pass) { gather_body(w, dm, stage_id(w, dm), threadIdx.x, pass); }

// ------------------------------------------------------------------ per-problem control (64 lanes per problem)
// (bodies: one wave works on problem b -- called by the kernels k_ctrl_a .. k_ctrl_d below and by the persistent kernel of tmpc_persist.h)
__device__ __forceinline__ double psum(const double* part, int b, int p, int idx, int lane) {
  double acc = 0.0;
  for (int k = lane; k < p; k += 64) acc += part[(size_t)(b * p + k) * NPART + idx];
  return wave_sum(acc);
}
// min over stages of the smallest step-length eigenvalue; which = 0: dual (S1,S2), 1: primal (X1,X2)
__device__ __forceinline__ double emin(const double* eigmin, int b, int p, int which, int lane) {
  double acc = 1e300;
  for (int k = lane; k < p; k += 64) {
    const double* e = eigmin + (size_t)(b * p + k) * 4;
    acc = fmin(acc, fmin(e[which], e[2 + which]));
  }
  return wave_min(acc);
}

// after k_stage_pre: mu, residual norms, phase logic (the same control flow as the CPU restatement used by the tests)
__device__ __forceinline__ void ctrl_a_body(const WS& w, const Dims& dm, const Opts& o, int b, int lane) {
  int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] == PH_DONE) return;
  double* pr = w.prob + (size_t)b * PS;
  const int p = dm.p, nx = dm.nx, nxx = nx * nx;
  const double xs = psum(w.part, b, p, Q_XS, lane), rd2 = psum(w.part, b, p, Q_RD2, lane);
  const double s2 = psum(w.part, b, p, Q_S2, lane), trx2 = psum(w.part, b, p, Q_TRX2, lane);
  const double hby = psum(w.part, b, p, Q_HBY, lane), trpsi = psum(w.part, b, p, Q_TRPSI, lane);
  const double trphi2 = psum(w.part, b, p, Q_TRPHI2, lane), hbphi = psum(w.part, b, p, Q_HBPHI, lane);
  const double nbad = psum(w.part, b, p, Q_CHOLBAD, lane);
  // dual residual of the P block:  r_P[j] = -(V_{j-1} Y_{j-1} V_{j-1}' - E Y_j E'),  Y = X1 - X2
  double rp2 = 0.0;
  for (int e = lane; e < p * nxx; e += 64) {
    const int j = e / nxx, r = e - j * nxx;
    const int a = r / nx, c = r - a * nx;
    if (c < a) continue;
    const int jm = (j == 0) ? p - 1 : j - 1;
    const double* kfm = w.KF + (size_t)(b * p + jm) * 12 * nxx;
    const double* kfj = w.KF + (size_t)(b * p + j) * 12 * nxx;
    const double v = (kfm[KF_KX * nxx + r] - kfm[(KF_PER_LMI + KF_KX) * nxx + r]) -
                     (kfj[KF_XXX * nxx + r] - kfj[(KF_PER_LMI + KF_XXX) * nxx + r]);
    const double wv = (a == c) ? v : 2.0 * v;
    rp2 = fma(wv, wv, rp2);
  }
  rp2 = wave_sum(rp2);
  double ncone = 0.0;
  if (dm.nr > 0 || dm.nT > 0) { rp2 += psum(w.part, b, p, Q_RPHI2, lane); ncone = psum(w.part, b, p, Q_NCONE, lane); }   // stage-local multipliers: stationarity residual, cone dimension
  if (lane != 0) return;
  const double N = 2.0 * p * dm.n + 1.0 + ncone;     // cone dimension
  const double tau = pr[P_TAU], alpha = pr[P_ALPHA], s0 = pr[P_S0], x0 = pr[P_X0];
  const double rd0 = (alpha - ALPHA_MIN) - s0;
  const double mu = (xs + x0 * s0) / N;
  const double r_tau = 1.0 - trx2, r_alpha = -hby - x0;
  const double pinf = sqrt(r_tau * r_tau + r_alpha * r_alpha + rp2) / 2.0;
  const double dinf = sqrt(rd2 + rd0 * rd0) / (1.0 + sqrt(s2));
  const double relgap = N * mu / fmax(1.0, fabs(tau));
  pr[P_MU] = mu; pr[P_RD0] = rd0; pr[P_PINF] = pinf; pr[P_DINF] = dinf; pr[P_RELGAP] = relgap; pr[P_SXS] = xs;
  pr[P_BTT] = trpsi; pr[P_BTA] = -trphi2; pr[P_BAA] = hbphi + x0 / s0;
  if (nbad > 0.0) ip[I_CHOLBAD] += (int)nbad;
  if (ip[I_ITERS] == 0) pr[P_MU0] = mu;
  // breakdown or divergence (dual unbounded = Step 1 infeasible): stop this problem with its last iterate
  if (!(mu > 0.0) || !(mu < 1e300) || nbad > 0.0 || !(s0 > 0.0) || !(x0 > 0.0) || mu > 1e6 * pr[P_MU0] || !(fabs(tau) < 1e300)) {
    ip[I_PHASE] = PH_DONE; ip[I_IPMSTATUS] = IPM_MAXITER;
    return;
  }
  double mut = pr[P_MUT];
  if (mut < 0.0 && relgap < 1e-2 && dinf < 1e-2) {
    mut = exp2(rint(log2(o.tol * fmax(1.0, fabs(tau)))));
    pr[P_MUT] = mut;
  }
  int phase = ip[I_PHASE];
  // (a full Newton step removes the linear residuals, so the centering phase may start with pinf well above the final
  // accuracy; waiting for pinf < 1e-6 cost the slowest problems of a batch three extra factorisations)
  // (with a lifted diagonal the directions are inexact and pinf may sit at 1e-2 ... 1e-1 for good while mu has long arrived at mu_t -- a
  // cond-1e6 member of the sweep spent its 50 main-phase iterations there: the centering phase, whose contraction test drops the lift and
  // backs mu_t off, is the way out, and its full Newton steps remove the linear residual once the factorisation is exact again)
  if (phase == PH_MAIN && mut > 0.0 && mu <= 2.0 * mut && dinf < 1e-6 && (pinf < 1e-3 || ip[I_REG] >= 1)) {
    phase = PH_CENTER;
    ip[I_PHASE] = phase;
  }
  if (phase == PH_MAIN && ip[I_SHIFTRUN] >= 2) {
    // the wall met on the way down (frozen pivots in the last two factorisations before mu reached 2 mu_t; k_ctrl_c let the steps pass):
    // the path cannot be followed below the current mu -- centre at the power of two above it, if the back-off budget covers that
    // (the CPU restatement used by the tests does the same).  Rounds 1-2 stopped here with an inaccurate point.
    const int kb = (mut > 0.0 && mu > mut) ? (int)ceil(log2(mu / mut)) : 0;
    if (mut > 0.0 && dinf < 1e-6 && ip[I_BACKOFF] + kb <= MUT_BACKOFF_MAX) {        // (pinf is noise after two safeguarded factorisations)
      mut = ldexp(mut, kb); pr[P_MUT] = mut; ip[I_BACKOFF] += kb;
      phase = PH_CENTER; ip[I_PHASE] = phase; ip[I_SHIFTRUN] = 0; ip[I_NCENT] = 0; pr[P_PREVSTEPN] = -1.0;
    } else {
      ip[I_PHASE] = PH_DONE; ip[I_IPMSTATUS] = IPM_INACCURATE;
      return;
    }
  }
  if (phase == PH_MAIN && ip[I_ITERS] >= o.max_iter) {
    ip[I_PHASE] = PH_DONE; ip[I_IPMSTATUS] = IPM_MAXITER;
    return;
  }
  ip[I_ITERS] += 1;
  ip[I_SHIFT0] = ip[I_NSHIFT];
  ip[I_NLOWP] += ip[I_LOWP];      // (this iteration's Schur-complement updates run in single precision: decided by k_ctrl_d of the previous iteration / k_init_prob)
  // centering budget: a chord step (factorisation re-used, a fifth of the cost, linear convergence) counts a quarter -- spending the
  // budget of Newton steps on chord steps would trigger the mu_t back-off below, i.e. change the answer, on slowly contracting members
  if (phase == PH_CENTER) { ip[I_NCENT] += (ip[I_CHORD] && (ip[I_NCHORD] & 3)) ? 0 : 1; pr[P_SIGMU] = mut; pr[P_CORR0] = 0.0; }
  else { pr[P_SIGMU] = 0.0; pr[P_CORR0] = 0.0; }
}

__global__ void __launch_bounds__(64) k_ctrl_a(WS w, Dims dm, Opts o) { ctrl_a_body(w, dm, o, prob_id(w), threadIdx.x); }

// After the factorisation of a centering iteration, before its right-hand side: hard target.  Frozen pivots while centering mean the Schur matrix
// is numerically singular AT THIS ITERATE (cond ~ (tau/mu)^2 passes 1/eps before the default mu_t when cond(H) >~ 1e3).  Aim for the
// central-path point one power of two earlier and TAKE the step this factorisation gives towards it (a frozen pivot leaves its component
// of the direction at zero: Newton restricted to the subspace that can still be resolved); the step moves the iterate back up the path, where the
// matrix is definite again.  The problem then ends Optimal at the gap N mu_t it reports in info[6].  (Rounds 1-2 doubled mu_t and REPEATED
// the iteration from the same iterate -- the matrix belongs to the iterate, not to the target: ten back-offs in a row met the same singular
// matrix, and the cond(H) = 1e5 members of scripts/robustness_sweep.py ended Feasible at 1024 mu_t instead of Optimal at 2-32 mu_t.)
// (Runs at the top of k_ctrl_b -- after the factorisation and the predictor pass that centering problems skip, before the right-hand side of
// pass 2 -- so it costs no launch of its own.)
__device__ __forceinline__ void ctrl_backoff_before_rhs(const WS& w, int b, int reg_max) {
  int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] != PH_CENTER || ip[I_CHORD]) return;
  if (ip[I_NSHIFT] == ip[I_SHIFT0] || ip[I_REG] < reg_max || ip[I_BACKOFF] >= MUT_BACKOFF_MAX) return;   // (frozen pivots first lift the diagonal, 1e-12 ... 1e-10, and repeat: k_ctrl_c)
  double* pr = w.prob + (size_t)b * PS;
  pr[P_MUT] *= 2.0; pr[P_SIGMU] = pr[P_MUT];
  ip[I_BACKOFF] += 1; ip[I_NCENT] = 0; pr[P_PREVSTEPN] = -1.0; ip[I_BOSTEP] = 1;
  ip[I_REG] = 0;                                  // the next factorisation tries without the lift (see k_ctrl_d)
}

// step lengths from the per-stage extreme eigenvalues + the scalar (alpha) block
__device__ __forceinline__ void raw_steps(const double* pr, double minx, double mins, double dx0, double ds0,
                                          double* ap, double* ad) {
  double a_p = (minx >= 0.0) ? 1e300 : -1.0 / minx;
  double a_d = (mins >= 0.0) ? 1e300 : -1.0 / mins;
  if (dx0 < 0.0) a_p = fmin(a_p, -pr[P_X0] / dx0);
  if (ds0 < 0.0) a_d = fmin(a_d, -pr[P_S0] / ds0);
  *ap = a_p; *ad = a_d;
}

// after the predictor direction (pass 1): Mehrotra centring parameter
__device__ __forceinline__ void ctrl_b_body(const WS& w, const Dims& dm, int reg_max, int b, int lane) {
  int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] != PH_MAIN) {
    if (lane == 0) ctrl_backoff_before_rhs(w, b, reg_max);        // centering: hard-target back-off, see above
    return;
  }
  double* pr = w.prob + (size_t)b * PS;
  const int p = dm.p;
  const double minx = emin(w.eigmin, b, p, 1, lane), mins = emin(w.eigmin, b, p, 0, lane);
  const double dxs = psum(w.part, b, p, Q_DXS, lane), xds = psum(w.part, b, p, Q_XDS, lane);
  const double dxds = psum(w.part, b, p, Q_DXDS, lane);
  const double ncone = (dm.nr > 0 || dm.nT > 0) ? psum(w.part, b, p, Q_NCONE, lane) : 0.0;
  if (lane != 0) return;
  const double N = 2.0 * p * dm.n + 1.0 + ncone;     // cone dimension
  const double s0 = pr[P_S0], x0 = pr[P_X0], mu = pr[P_MU];
  const double ds0 = pr[P_DALPHA] + pr[P_RD0];
  const double dx0 = -x0 - x0 * ds0 / s0;
  double ap, ad;
  raw_steps(pr, minx, mins, dx0, ds0, &ap, &ad);
  ap = fmin(1.0, ap); ad = fmin(1.0, ad);
  const double mu_aff = (pr[P_SXS] + ap * dxs + ad * xds + ap * ad * dxds + (x0 + ap * dx0) * (s0 + ad * ds0)) / N;
  const double rat = mu_aff / mu;
  double sigma = fmin(fmax(rat * rat, 1e-6), 1.0);     // exponent 2: ~10 % fewer iterations than Mehrotra's 3 here
  if (!(sigma == sigma)) sigma = 1.0;
  double sig_mu = sigma * mu;
  if (pr[P_MUT] > 0.0) sig_mu = fmax(sig_mu, pr[P_MUT]);
  pr[P_SIGMU] = sig_mu;
  pr[P_CORR0] = dx0 * ds0 / s0;
}

__global__ void __launch_bounds__(64) k_ctrl_b(WS w, Dims dm, int reg_max) { ctrl_b_body(w, dm, reg_max, prob_id(w), threadIdx.x); }

// after the final direction (pass 2): step lengths, scalar updates
__device__ __forceinline__ void ctrl_c_body(const WS& w, const Dims& dm, int reg_max, int b, int lane) {      // reg_max: REG_MAX, or 0 under TMPC_DEBUG_FLAG_NO_LIFT (unit test of the back-off-and-step route)
  int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE) return;
  double* pr = w.prob + (size_t)b * PS;
  const int p = dm.p;
  const double minx = emin(w.eigmin, b, p, 1, lane), mins = emin(w.eigmin, b, p, 0, lane);
  const double dh2 = psum(w.part, b, p, Q_DH2, lane), m2 = psum(w.part, b, p, Q_M2, lane);
  if (lane != 0) return;
  const double s0 = pr[P_S0], x0 = pr[P_X0];
  const double dtau = pr[P_DTAU], dalpha = pr[P_DALPHA];
  const double ds0 = dalpha + pr[P_RD0];
  const double dx0 = pr[P_SIGMU] / s0 - x0 - x0 * ds0 / s0 - pr[P_CORR0];
  double ap, ad;
  raw_steps(pr, minx, mins, dx0, ds0, &ap, &ad);
  pr[P_RAWSTEP] = fmin(ap, ad);               // how many times longer the step could be: 1 / (size of the step in the local norm of the cone)
  if (phase == PH_MAIN) {
    const double mn = fmin(ap, ad);
    const double gam = 0.9 + 0.09 * fmin(mn, 1.0);
    ap = fmin(1.0, gam * ap); ad = fmin(1.0, gam * ad);
  } else {
    ap = fmin(1.0, TMPC_CENTER_DAMP * ap); ad = fmin(1.0, TMPC_CENTER_DAMP * ad);
  }
  bool retry = false;
  // Numerical breakdown: frozen pivots in this iteration's Schur factorisation during the centering phase (cond(T) grows
  // like (tau/mu)^2 and reaches 1/eps near the default mu_t when H is badly scaled) or in two main-phase iterations in
  // a row, or two iterations in a row whose step lengths collapse.  The direction is then worthless: keep the last iterate (it is a strictly feasible point
  // close to the central path at ~2 mu_t), report it as inaccurate, and do not hold the rest of the batch hostage until
  // the iteration cap.
  {
    const bool jam = (ap < 1e-6 && ad < 1e-6);
    const bool froze = ip[I_NSHIFT] != ip[I_SHIFT0];
    ip[I_JAM] = jam ? ip[I_JAM] + 1 : 0;
    ip[I_SHIFTRUN] = froze ? ip[I_SHIFTRUN] + 1 : 0;
    // a direction that is not finite (a cascade of frozen pivots can overflow the fill row) is a breakdown as well
    const bool nonfin = !(fabs(dtau) < 1e300) || !(fabs(dalpha) < 1e300) || !(ap == ap) || !(ad == ad) || !(dh2 == dh2);
    const bool bostep = ip[I_BOSTEP] != 0;       // ctrl_backoff_before_rhs (k_ctrl_b) backed mu_t off before this direction was computed
    ip[I_BOSTEP] = 0;
    // two main-phase iterations in a row with frozen pivots and back-offs left: let the step pass, k_ctrl_a starts centering where the iterate stands
    const bool wall_next = (phase == PH_MAIN && !nonfin && ip[I_JAM] < 2 && pr[P_MUT] > 0.0 && ip[I_BACKOFF] < MUT_BACKOFF_MAX);
    if (ip[I_DD] && (froze || nonfin)) {
      // tight phase (tmpc_dd.h): no lift, no back-off -- a non-positive pivot of the dd factorisation ends the problem with its last iterate
      ap = 0.0; ad = 0.0;
      ip[I_PHASE] = PH_DONE; ip[I_IPMSTATUS] = IPM_INACCURATE;
    } else if (bostep && !nonfin) {
      // ctrl_backoff_before_rhs backed mu_t off before this direction was computed: the step is TAKEN (it moves the iterate back up the path, where the
      // matrix is definite again).  Tested BEFORE the lift rule: the back-off has reset I_REG to 0, so the lift rule below would fire on the very pivots
      // that triggered the back-off, discard the direction and repeat the iteration from the same iterate (round 3 did exactly that: this branch was
      // unreachable and every back-off of this kind cost up to three idle lift retries -- ADVICE r3).
      ip[I_SHIFTRUN] = 0;
    } else if ((froze || nonfin) && ip[I_LOWP]) {
      // a pivot froze under single-precision updates (not seen on the benchmark distribution at the default switch): the same iteration once more, in fp64 --
      // no lift, no trace of the attempt in the iterate
      ip[I_LOWPOFF] = 1; ip[I_SHIFTRUN] = 0; ip[I_JAM] = 0;
      ap = 0.0; ad = 0.0; retry = true;
    } else if ((froze || nonfin) && ip[I_REG] < reg_max) {
      // first answer to frozen pivots: discard this direction, lift the Schur diagonal by 1e-12 relative
      // from now on and repeat the iteration from the same iterate -- the matrix sits within ~1e-13 (diagonally scaled) of
      // singular near mu_t, and e.g. the stage-local elimination of an active multiplier (tmpc_phi.h) can use that margin up.
      // If pivots freeze again the lift grows to 1e-11, then 1e-10 (round 3; the oracle's Cholesky-with-shift escalates the same way):
      // the block factorisation multiplies by explicitly inverted diagonal tiles, which costs ~sqrt(cond D_k) eps of accuracy against a
      // substitution, so a matrix with scaled lambda_min = 1.3e-13 that LAPACK still factors freezes pivots here even at 1e-12
      // (scripts/case93_factor_probe.py: a fuzz member of the benchmark distribution whose first centering iterate follows a 32-fold
      // drop of mu; it ended Feasible after ten idle back-offs).  Larger lifts damp the weakest eigen-direction and centering turns
      // linear -- k_ctrl_d then backs mu_t off and drops the lift.
      ip[I_REG] += 1; ip[I_SHIFTRUN] = 0; ip[I_JAM] = 0;
      ap = 0.0; ad = 0.0; retry = true;
    } else if (phase == PH_CENTER && (froze || nonfin) && ip[I_BACKOFF] < MUT_BACKOFF_MAX) {
      // hard target (cond(H) >~ 1e3: cond of the Schur matrix ~ (tau/mu)^2 passes 1/eps before the default mu_t): aim for the
      // central-path point one power of two earlier instead of giving up -- the problem then ends Optimal at the gap
      // N * mu_t it reports in info[6] (the default is tol * kappa), not Feasible at an uncontrolled one
      pr[P_MUT] *= 2.0; ip[I_BACKOFF] += 1; ip[I_SHIFTRUN] = 0; ip[I_JAM] = 0; ip[I_NCENT] = 0; pr[P_PREVSTEPN] = -1.0;
      ap = 0.0; ad = 0.0; retry = true;
    } else if ((phase == PH_CENTER && froze) || (ip[I_SHIFTRUN] >= 2 && !wall_next) || ip[I_JAM] >= 2 || nonfin) {
      ap = 0.0; ad = 0.0;
      ip[I_PHASE] = PH_DONE; ip[I_IPMSTATUS] = IPM_INACCURATE;
    }
  }
  const bool stopped = (ip[I_PHASE] == PH_DONE) || retry;
  // relative first-order change of the output Hc (the quantity the parity gate measures) in this step
  if (!stopped) pr[P_STEPN] = sqrt(dh2 / m2);
  pr[P_AP] = ap; pr[P_AD] = ad;
  if (w.trace && ip[I_ITERS] >= 1 && ip[I_ITERS] <= TRACE_LEN) {
    double* t = w.trace + ((size_t)b * TRACE_LEN + (ip[I_ITERS] - 1)) * TRACE_W;
    t[0] = (double)ip[I_ITERS]; t[1] = (double)phase + 0.25 * ip[I_CHORD];    // x.25: a chord step (factorisation re-used)
    t[2] = pr[P_MU]; t[3] = pr[P_TAU]; t[4] = pr[P_PINF]; t[5] = pr[P_DINF];
    t[6] = ap; t[7] = (phase == PH_CENTER) ? pr[P_RAWSTEP] : ad; t[8] = pr[P_STEPN];     // centering: ap = ad = 1 on full steps; slot 7 holds the raw step length instead
    t[9] = (double)(ip[I_NSHIFT] + ip[I_CHOLBAD]);
  }
  if (!stopped) {        // (0 * NaN would poison the kept iterate)
    pr[P_X0] = x0 + ap * dx0; pr[P_S0] = s0 + ad * ds0;
    pr[P_TAU] += ad * dtau; pr[P_ALPHA] += ad * dalpha;
  }
}

__global__ void __launch_bounds__(64) k_ctrl_c(WS w, Dims dm, int reg_max) { ctrl_c_body(w, dm, reg_max, prob_id(w), threadIdx.x); }

// after k_update: termination of the centering phase (one THREAD per problem); lists: re-enter the problem into the active / to-factor lists of the next
// iteration (the launch-sequence path; the persistent kernel keeps no lists)
__device__ __forceinline__ void ctrl_d_body(const WS& w, const Dims& dm, const Opts& o, int b, bool lists) {
  int* ip = w.iprob + (size_t)b * IS;
  if (ip[I_PHASE] == PH_DONE || ip[I_PHASE] == PH_POLISH) return;
  double* pr = w.prob + (size_t)b * PS;
  if (ip[I_DD]) ip[I_NDD] += 1;
  if (o.tight && ip[I_PHASE] == PH_CENTER && pr[P_AP] == 1.0 && pr[P_AD] == 1.0 && pr[P_STEPN] < POLISH_ENTER_GPU) {
    // tight phase: the first full centering step this small hands the problem to the dd dual-Newton polish (tmpc_dd.h: POLISH_ENTER)
    ip[I_PHASE] = PH_POLISH; ip[I_CHORD] = 0; pr[P_PREVSTEPN] = -1.0;
    const int ps = atomicAdd(w.active + 2, 1); w.plist[ps] = b;
    return;
  }
  if (ip[I_PHASE] == PH_CENTER) {
    const bool full = (pr[P_AP] == 1.0 && pr[P_AD] == 1.0);
    const bool was_chord = ip[I_CHORD] != 0;
    const double stepn = pr[P_STEPN], prev = pr[P_PREVSTEPN];
    const double rr = (prev >= 0.0) ? fmin(1.0, stepn / prev) : 1.0;                 // contraction of the last two full steps
    // extrapolated next step: between linear (r) and quadratic (r^2) convergence for Newton steps; a chord step (frozen
    // factorisation) converges linearly, its remaining error is ~ stepn * r / (1 - r) <= stepn for r <= 1/2
    // extrapolated next step: between linear (r) and quadratic (r^2) convergence for Newton steps; a chord step (frozen factorisation) is
    // never extrapolated: it ends the phase only by its own size.  (Round 3 tried the contraction of two consecutive chord steps,
    // remaining error ~ stepn r / (1 - r): one cheap iteration fewer in half of the problems, but the error it left reached 1.1e-8 on a
    // fuzz member whose last chord step contracted the step norm 200-fold and the output only 2-fold: reverted.)
    const double est = was_chord ? stepn : stepn * rr * sqrt(rr);
    bool chord_next = false, full_reset = false;
    if (full && (o.fast_exit || stepn < o.center_tol || (!was_chord && est < 0.1 * o.center_tol))) {
      ip[I_PHASE] = PH_DONE;
      ip[I_IPMSTATUS] = (stepn < o.center_tol || (!was_chord && est < 0.1 * o.center_tol)) ? IPM_OPTIMAL : IPM_FAST_EXIT;      // (a fast exit is told apart in info[10]: ADVICE r3)
    }
    // (rounding floor: the steps stopped contracting below 1e-6.  Not with a lifted diagonal: Newton is damped in the weakest directions then
    // and small steps say nothing about the distance to the centred point -- a fuzz member stopped 1e-4 away with steps of 1e-7; the next
    // branch backs mu_t off and drops the lift instead.)
    else if (!was_chord && full && prev >= 0.0 && stepn > 0.5 * prev && stepn < 1e-6 && ip[I_REG] == 0) { ip[I_PHASE] = PH_DONE; ip[I_IPMSTATUS] = IPM_OPTIMAL; }
    // with a lifted Schur diagonal Newton is inexact in the weakest direction: once the steps stop contracting -- or the centering
    // budget is spent -- there is nothing more to gain at this mu_t: back off to the next power of two (hard target), or stop
    else if (ip[I_NCENT] >= o.center_iter || (ip[I_REG] > 0 && full && prev >= 0.0 && stepn > 0.5 * prev)) {
      // (the lift of the diagonal goes with the target it was needed for: with it Newton is inexact and converges linearly at ANY mu_t -- the
      // Schur matrix of an ill-conditioned H has eigenvalues below 1e-12 relative all along the path -- while the plain factorisation may well
      // succeed one power of two up; if it does not, the first frozen pivot brings the lift back at the price of one factorisation)
      if (ip[I_BACKOFF] < MUT_BACKOFF_MAX) { pr[P_MUT] *= 2.0; ip[I_BACKOFF] += 1; ip[I_NCENT] = 0; ip[I_REG] = 0; full_reset = true; }
      else { ip[I_PHASE] = PH_DONE; ip[I_IPMSTATUS] = IPM_INACCURATE; }
    }
    else if (o.chord_step > 0.0 && full && ip[I_REG] == 0 && !ip[I_LOWP]) {      // (never on a factorisation with single-precision updates: its fp64 O blocks were not written, and the decision below has not replaced I_LOWP yet -- unreachable by the numbers, the switch sits far above 2 mu_t, but cheap to rule out)
      // Chord steps: the last step moved the iterate by less than 1/chord_step in the local norm, so the Schur matrix at the new
      // iterate differs from the factored one by about that much and Newton with the OLD factorisation still contracts by that
      // factor per step -- at a fifth of the cost.  A chord step that contracts by less than 1/4 goes back to a fresh factorisation.
      if (!was_chord) chord_next = pr[P_RAWSTEP] >= o.chord_step;
      else chord_next = (prev < 0.0) || (stepn <= 0.25 * prev);
    }
    ip[I_CHORD] = (ip[I_PHASE] != PH_DONE && chord_next) ? 1 : 0;
    if (ip[I_CHORD]) ip[I_NCHORD] += 1;
    pr[P_PREVSTEPN] = (full && !full_reset) ? stepn : -1.0;
  }
  // Early main-phase iterations: the Schur-complement updates of the NEXT factorisation in single precision (tmpc_gemm_dma.h: wg_tile_dma_f32) while the barrier
  // parameter it will meet -- predicted from this iteration's step, (1 - a) mu + a sigma mu with a = min(ap, ad) -- is above lowp_switch * kappa.  Decided here,
  // a whole iteration ahead, because the host sizes its launches from the counters this kernel leaves (w.active[4]: such problems among those to factor): it
  // starts the float32 kernel, the fp64 one, or both.  Not once a pivot froze under them (k_ctrl_c repeats that iteration in fp64), not with a lifted diagonal,
  // never while centering (the switch sits two to three orders of magnitude above mu_t).
  {
    int lowp = 0;
    if (o.lowp_switch > 0.0 && w.O32 != nullptr && !o.tight && ip[I_PHASE] == PH_MAIN && !ip[I_LOWPOFF] && ip[I_REG] == 0) {
      const double a = fmin(pr[P_AP], pr[P_AD]);
      const double mu_next = (1.0 - a) * pr[P_MU] + a * pr[P_SIGMU];
      lowp = mu_next > o.lowp_switch * fmax(1.0, fabs(pr[P_TAU]));
    }
    ip[I_LOWP] = lowp;
  }
  if (lists && ip[I_PHASE] != PH_DONE) {
    const int slot = atomicAdd(w.active, 1); w.alist[slot] = b;
    if (!ip[I_CHORD]) { const int fs = atomicAdd(w.active + 1, 1); w.flist[fs] = b; if (ip[I_LOWP]) atomicAdd(w.active + 4, 1); }
  }
}
__global__ void __launch_bounds__(64) k_ctrl_d(WS w, Dims dm, Opts o) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= dm.B) return;
  ctrl_d_body(w, dm, o, b, true);
}

}  // namespace tmpc
