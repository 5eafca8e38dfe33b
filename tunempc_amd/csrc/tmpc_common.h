// Shared definitions for the MI355X (gfx950) convexifier kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tmpc {

// optional in-kernel cycle split (scripts/cycle_prof.py, scripts/gemm2_prof.py, scripts/stage_prof.py; -DTMPC_CYCLE_PROF builds only)
#ifdef TMPC_CYCLE_PROF
__device__ unsigned long long g_prof[64];      // [class][8]: class set by the host before a launch (cr_factor: 1 potrf, 2 trsm, 3 update; 0: everything else)
__device__ int g_prof_cls;
#define TMPC_T(i) { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); g_prof[(g_prof_cls & 7) * 8 + (i)] += t_ - tprev_; tprev_ = t_; } }
#define TMPC_T0() unsigned long long tprev_ = __builtin_readcyclecounter();
#define TMPC_TC(c, i) { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); g_prof[(c) * 8 + (i)] += t_ - tprevc_; tprevc_ = t_; } }
#define TMPC_TC0() unsigned long long tprevc_ = __builtin_readcyclecounter();
#else
#define TMPC_TC(c, i)
#define TMPC_TC0()
#define TMPC_T(i)
#define TMPC_T0()
#endif



constexpr double ALPHA_MIN = 1e-8;     // reference: convexifier.py:245  (alpha > 1e-8)
constexpr int TRACE_LEN = 80, TRACE_W = 10;
constexpr int TB = 64;                 // tile size of the d x d block factorisation (potrf / trsm granularity)
#ifndef TMPC_CENTER_DAMP
#define TMPC_CENTER_DAMP 0.95      // damping of the raw step length in the centering phase (k_ctrl_c; k_eigmin's pre-test threshold)
#endif
constexpr int REG_MAX = 3;             // levels of the relative lift of the Schur diagonal after frozen pivots: 1e-12, 1e-11, 1e-10 (k_schur, k_ctrl_c)
constexpr int EIG_MAX_SWEEPS = 40;     // tmpc_eig_clip_host: Jacobi sweeps before TMPC_E_NOCONV

// ---- per-problem double scalars (prob[b*PS + idx])
enum {
  P_TAU = 0, P_ALPHA, P_S0, P_X0, P_MU, P_MUT, P_SIGMU, P_AP, P_AD, P_S, P_SBETA, P_PINF, P_DINF, P_RELGAP,
  P_STEPN, P_PREVSTEPN, P_DTAU, P_DALPHA, P_DS0, P_DX0, P_RD0, P_CORR0, P_MINEIG_H, P_BTT, P_BTA, P_BAA,
  P_SB00, P_SB01, P_SB11, P_RHS_TAU, P_RHS_ALPHA, P_SXS, P_MINEIG_HC, P_MAXCOND, P_KAPPA, P_BETA, P_ALPHA_OUT,
  P_MAXEIG_HC, P_MU0, P_MINPIV, P_RAWSTEP, P_TAU_PREV, P_ALPHA_PREV, P_MUT1, P_TAU_DEF, P_ALPHA_DEF, PS = 48
};
// ---- per-problem int scalars (iprob[b*IS + idx])
enum { I_PHASE = 0, I_ITERS, I_NCENT, I_IPMSTATUS, I_EARLY, I_NSHIFT, I_STATUS, I_BOSTEP, I_CHOLBAD, I_SHIFT0, I_JAM, I_SHIFTRUN, I_REG, I_CHORD, I_NCHORD, I_BACKOFF,
       I_DD, I_NDD, I_NPOLISH, I_LOWP, I_LOWPOFF, I_NLOWP, IS = 24 };   // I_LOWP: 1 = this iteration's Schur-complement updates run in float32 (k_cr_update_dma, Opts::lowp_switch); I_LOWPOFF: never again for this problem (a pivot froze under them); I_NLOWP: such iterations   // I_DD: 1 = tight phase (block linear algebra in double-double, tmpc_dd.h); I_NDD: such iterations; I_NPOLISH: polish steps   // I_BACKOFF: times mu_t was doubled for this problem (hard targets: the Schur matrix is numerically singular at the default mu_t)
//   // I_BOSTEP: 1 = k_ctrl_b (ctrl_backoff_before_rhs) backed mu_t off in this centering iteration (frozen pivots): the step of that factorisation is TAKEN (k_ctrl_c)
//   // I_REG: regularisation level of the Schur diagonal (0: none), raised after an iteration with frozen pivots
//   // I_CHORD: 1 = this centering iteration re-uses the factorisation (and border columns) of the previous one; I_NCHORD: such iterations so far
//   // I_SHIFT0: I_NSHIFT at the start of the iteration; I_JAM: consecutive iterations with collapsed step lengths; I_SHIFTRUN: consecutive iterations with frozen pivots
// phases
enum { PH_MAIN = 0, PH_CENTER = 1, PH_DONE = 2, PH_POLISH = 3 };     // PH_POLISH: waits for / runs the dd dual-Newton polish of the tight mode (out of the active list)
// ipm status
enum { IPM_OPTIMAL = 0, IPM_INACCURATE = 1, IPM_MAXITER = 2, IPM_FAST_EXIT = 3, IPM_TIGHT_FALLBACK = 4 };     // IPM_TIGHT_FALLBACK: the tight phase of this member failed, the result of its default solve was put back (status Optimal at the DEFAULT gap: info[6])
//     // IPM_FAST_EXIT: stopped by TMPC_FLAG_FAST_EXIT after the first full centering step (status Optimal, info[10] = 3)
// reference status strings (convexifier.py:442-451)
enum { ST_OPTIMAL = 0, ST_FEASIBLE = 1, ST_INFEASIBLE = 2 };

// ---- per-stage partial scalars (part[(b*p+k)*NPART + idx])
enum {
  Q_XS = 0, Q_RD2, Q_S2, Q_TRX2, Q_HBY, Q_TRPSI, Q_TRPHI2, Q_HBPHI,   // stage_pre
  Q_TRT2, Q_HBG,                                                       // stage_rhs
  Q_MINX, Q_MINS, Q_DXS, Q_XDS, Q_DXDS, Q_DP2, Q_P2,                   // stage_dir
  Q_MINEIG, Q_MINABS, Q_MAXABS, Q_MAXEIG, Q_CHOLBAD, Q_DH2, Q_M2,
  Q_RPHI2, Q_NCONE,                                                    // phi_pre (stage-local multipliers): residual, cone dimension
  NPART = 26
};

// Kronecker-factor slots per stage (KF[((b*p+k)*12 + slot) * nx*nx])
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));
enum { KF_XXX = 0, KF_SIXX, KF_KX, KF_KS, KF_FX, KF_FS, KF_PER_LMI = 6 };
// adjoint slots per stage: 0: G = T1-T2 (rhs), 1: Psi (tau column), 2: PhiH (alpha column)
enum { ADJ_G = 0, ADJ_PSI, ADJ_PHI, NADJ = 3 };

struct Dims {
  int B;        // problems in this chunk
  int p;        // period (stages per problem)
  int nx, mb;   // state / input(+slack) dims
  int n;        // nx + mb
  int d;        // nx(nx+1)/2
  int dp;       // d padded to a multiple of 16
  int nt;       // ceil(dp / TB)
  int flags;    // bit0: debug - replace MFMA by scalar FMAs; bit3 (DF_LOWP, tmpc_api.hip): single-precision updates are on for this call (k_init_prob: the first iteration)
  int ng;       // rows of the equality-constraint Jacobian G_k per stage (0: none), <= NGM
  int nr;       // row stride of the stage-local multipliers: ng + (max rows of the active-constraint Jacobians C_k, Step 2), <= NRM
  int nz;       // stride of the stage-local variable vector: nr (+ 2 epigraph variables of the norm terms in Step 2)
  int constr;   // 1: Step 2 model (convexifier.py:116-131): multipliers of C_k and the rho-norm terms
  int nT;       // Step 3 (convexifier.py:137-147): n(n+1)/2 entries of the regularisation T_k per stage (0: none); + 1 epigraph variable (tmpc_t3.h)
};
constexpr int NGM = 31;  // max ng
constexpr int NCM = 31;  // max rows of C_k
constexpr int NRM = NGM + NCM, NZM = NRM + 2;      // (nz <= 64: one lane per stage-local variable)
constexpr int NRS = 32, NZS = NRS + 2;             // up to NRS rows per stage the multiplier kernels keep their per-row vectors and Gram products in LDS (tmpc_phi.h)
constexpr int NAM = 31;  // max rows under one norm term; its arrow LMI is (NAM+1) x (NAM+1): one 32 x 33 LDS slot
constexpr int AEL = NAM + 1, AE = AEL * AEL;

struct Opts {
  double tol;          // complementarity tolerance: mu_target = tol * kappa (relative gap on kappa = (2pn+1)*tol)
  double center_tol;   // relative Newton step that ends the centering phase
  int max_iter;
  int center_iter;
  double chord_step;   // centering: once a full Newton step could have been this many times longer before leaving the cone (i.e. the
                       // iterate moved by < 1/chord_step in the local norm), the next steps re-use the factorisation; 0 = never
  int tight;           // 1: this loop is the tight phase (tmpc_dd.h): no chord steps, no lifts / back-offs, centering hands over to the polish
  double tight_tol;    // its complementarity tolerance (mu_target = tight_tol * kappa)
  double lowp_switch;  // > 0: the Schur-complement updates of the block factorisation (k_cr_update_dma) run on float32 copies of the O blocks with float32
                       // accumulation in the main-phase iterations with mu > lowp_switch * max(1, |tau|) (plain model, blocks of 32 ... 320); 0: never
  int fast_exit;       // TMPC_FLAG_FAST_EXIT: stop after the FIRST full centering step (feasible, kappa within the gap N mu_t of optimal, but
                       // not the converged central-path point: not reproducible to 1e-8 between implementations)
};

// Device workspace (all pointers device memory, fp64 row-major)
struct WS {
  // inputs (chunk views)
  const double* A; const double* Bm; const double* H;
  // scaled data
  double* Hb;      // [B,p,n,n]   s * H
  double* V;       // [B,p,nx,n]  [A B]
  // iterate
  double* P;       // [B,p,nx,nx] (scaled Pbar)
  double* X1; double* X2; double* S1; double* S2;       // [B,p,n,n]
  // per-iteration stage data
  double* S1i; double* S2i; double* L1i; double* L2i; double* LX1i; double* LX2i;
  double* Rd1; double* Rd2; double* T1; double* T2;
  double* dS1; double* dS2; double* dX1; double* dX2; double* c1; double* c2;
  double* dP;      // [B,p,nx,nx]
  double* Wm;      // [B,p,4,n,n]  step-length matrices (S1, X1, S2, X2 order: 2r dual, 2r+1 primal)
  double* eigmin;  // [B,p,4]      their smallest eigenvalues
  double* KF;      // [B,p,12,nx,nx]
  double* adjV;    // [B,p,3,nx,nx]
  double* adjE;    // [B,p,3,nx,nx]
  double* part;    // [B,p,NPART]
  double* prob;    // [B,PS]
  int* iprob;      // [B,IS]
  // Schur system
  double* D;       // [B,p,dp,dp]   diagonal blocks -> Cholesky factors (diag tiles hold L_jj, inverse in Linv)
  double* O;       // [B,p,dp,dp]   edge slots 0..p-1: coupling block of stages k, k+1 (orientation cr_orient[k]) -> O factors (tmpc_cr.h)
  double* F;       // [B,p,dp,dp]   edge slots p..2p-1: fill blocks of the cyclic reduction
  float* O32;      // [B,2p,dp,ld32] float32 copies of the O factors of both slot ranges (ld32 = dp rounded up to 32, zero padding), written by k_cr_trsm_dma for
                   //                the problems whose updates run in single precision (I_LOWP); nullptr: the handle has none
  double* Linv;    // [B,p,nt,TB,TB] inverses of the diagonal tiles
  double* Ddiag;   // [B,p,dp]      assembled diagonal of D (pivot reference for Cholesky-with-shift)
  double* W3;      // [B,p,dp,3]    pass-1 right-hand sides [rhs | u_tau | u_alpha] -> solutions
  double* U;       // [B,p,dp,2]    border columns (tau, alpha)
  double* TU;      // [B,p,dp,2]    T^-1 U
  double* Z;       // [B,p,dp]      rhs / solution
  int* active;     // [1] number of problems still iterating
  int* flist;      // [B] the problems that need a new factorisation this iteration (count in active[1]); the others take a chord step
  int* alist;      // [B] their indices, compacted (written by k_init_prob / k_ctrl_d).  The per-iteration kernels are launched over
                   // `active` problems and map blockIdx through this list; nullptr = identity (init / final kernels, all problems)
  const int* cr_orient;  // [p] storage orientation of the coupling block of stage k (tmpc_cr.h): 0: T[P_{k+1},P_k], 1: T[P_k,P_{k+1}]
  double* trace;   // [B][TRACE_LEN][TRACE_W] per-iteration diagnostics (it, phase, mu, tau, pinf, dinf, ap, ad, stepn, shifts)
  // outputs
  double* Hc;      // [B,p,n,n]
  double* dHc;     // [B,p,n,n]
  double* Pout;    // [B,p,nx,nx]
  // equality-constraint term of Step 1 (convexifier.py:249-255, :346-347): M_k += G_k' diag(phi_k) G_k, phi_k = s*Fg_k >= 0
  // Step 2 (convexifier.py:258-266, :348-350): the rows of C_k follow those of G_k, phi_k = s*[Fg_k; F_k] >= 0.
  const double* G; // [B,p,nr,n] input: rows of G_k, then rows of C_k, zero padding
  const int* ncnt; // [B,p] rows of C_k per stage (nullptr: none)
  double rho;      // weight of the norm terms (convexifier.py:276-283); in the scaled problem w = rho*sbeta/s
  double* phi; double* zph; double* dphi; double* dzph; double* corrp;   // [B,p,nr] multipliers (slack = phi itself), their duals, directions, Mehrotra term
  double* pvec;    // [B,p,2,nr,2n+2nx]  per cone block r and row i: w = X_r g, u = S_r^-1 g, V w, V u
  double* psm;     // [B,p,nz*nz+6*nz]   K = T_zz^-1, c_tau, c_alpha, K c_tau, K c_alpha, r_z, K r_z
  double* prs;     // [B,p,4,nr,nr]      handles with more than NRS rows per stage: the Gram products g_i' X_r g_j, g_i' S_r^-1 g_j of k_phi_pre (else null: LDS)
  double* Fg;      // [B,p,nr] output: phi / (s*alpha)
  // norm terms t >= ||w v|| as arrow LMIs S = [[t, w v'], [w v, t I]] (always feasible: S is rebuilt from t and phi), up to two per stage
  double* at; double* adt;              // [B,p,2]   epigraph variables and their directions
  double* aX; double* adX; double* acor; // [B,p,2,AE] primal blocks, directions, Mehrotra term
  double* aSi; double* aLi; double* aLXi; // [B,p,2,AE] S^-1, L_S^-1, L_X^-1
  double* asum;    // [B,p,5]   <dX,S>, <X,dS>, <dX,dS>, min eig dual, min eig primal of the arrow blocks (joined in k_phi_steps)
  // Step 3 (tmpc_t3.h): entries theta of T_k (linear cone, dual z) and the norm cone (t; w c o theta) with multiplier x
  double* t3th; double* t3z; double* t3dth; double* t3dz; double* t3cth;          // [B,p,nT]
  double* t3x; double* t3dx; double* t3cq; double* t3g; double* t3v; double* t3lam;   // [B,p,nT+1]
  double* t3t; double* t3dt; double* t3beta;                                      // [B,p]
  double* t3psi; double* t3phi;   // [B,p,n,n] full Psi = sym(X2 S2^-1) and Phi(Hb) of the stage (border entries of the theta rows)
  double* Tout;    // [B,p,n,n] output T_k
  // tight mode (tmpc_dd.h): low words of the double-double planes (high words = D, O, F, Linv, KF, adjV, adjE, W3, Z); null without it
  double* Dl; double* Ol; double* Fl; double* Linvl; double* KFl; double* adjVl; double* adjEl; double* W3l; double* Zl;
  double* ddscr;   // [B,p,DD_SCR_MATS,2,n,n] 32 < n <= 64: the dd stage matrices that the tuned form keeps in LDS (tmpc_dd.h: sdd_slot); else null
  double* Pprev;   // [B,p,nx,nx] iterate before the last polish step
  double* Pdef;    // [B,p,nx,nx] result of the default solve (restored when the tight phase of a member fails)
  double* phidef;  // [B,p,nr] multipliers of the default solve (tight mode with rows of G)
  double* atdef;   // [B,p,2] epigraph variables of the default solve (tight mode on the Step 2 model)
  double* TUl;     // [B,p,dp,2] low words of T^-1 U (the border system of the polish is solved in double-double: k_dd_solve_border)
  double* Zdd;     // [B,p,2,2,n,n] S_r^-1 of the polish in double-double (hi plane, lo plane per cone block): the rows of the multipliers are formed from it (k_dd_aug_fill)
  double* sscr;    // [B,p,10,nx,nx] factor records of k_schur when they do not fit the LDS (nx > 43); null otherwise
  double* bscr;    // [B,p,5,n,n] scratch of the generic per-stage kernels (tmpc_big.h: 32 < n <= 64); null otherwise
  int* plist;      // [B] problems handed to the polish (count in active[2])
  int* pnext;      // [B] second index buffer of the polish steps (plist itself must survive until the final sweep)
};

// (problem, stage) of this workgroup for kernels with one workgroup per stage: blockIdx.x = (index in the active list) * p + k
__device__ __forceinline__ int stage_id(const WS& w, const Dims& dm) {
  if (!w.alist) return blockIdx.x;
  const int bi = blockIdx.x / dm.p;
  return w.alist[bi] * dm.p + (blockIdx.x - bi * dm.p);
}
// problem of this workgroup for kernels with one workgroup per problem
__device__ __forceinline__ int prob_id(const WS& w) { return w.alist ? w.alist[blockIdx.x] : (int)blockIdx.x; }

}  // namespace tmpc
