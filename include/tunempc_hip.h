/*
 * tunempc_hip.h -- C ABI of the MI355X (gfx950) convexifier.
 *
 * Drop-in boundary for the hot path of TuneMPC, `tunempc/convexifier.py` in the reference:
 * the per-problem SDP that turns the p indefinite stage Hessians H_k of a solved periodic OCP into
 * positive-definite tracking-cost matrices  Hc_k = H_k + dHc_k.  The reference has no native boundary
 * (everything is Python calling PICOS -> CVXOPT/MOSEK); the entry points below are what a ctypes
 * binding of that path binds instead:
 *
 *   reference interface                                    replaced by
 *   ------------------------------------------------------------------------------------------------
 *   convexifier.convexify(A,B,Q,R,N,G,C,opts)               tmpc_convexify_batch_host / _device
 *       (convexifier.py:36-163; Step 1: :98-114)              (B independent problems per call)
 *   convexifier.autoScaling            (:374-401)           inside (k_init_*), reported via info[]
 *   convexifier.setUpModelPicos+solveSDP (:213-308,:359-372) inside (structured primal-dual IPM)
 *   convexifier.check_convergence      (:403-456)           status[] (0 Optimal,1 Feasible,2 Infeasible)
 *   convexifier.convexHessianSuppl     (:165-211)           tmpc_supplement_batch_* and dHc output
 *   opts = {'rho','solver','force'}    (:36)                tmpc_set_options (tol, iteration caps)
 *   Tuner.convexify                    (tuner.py:134-160)   Python side: tunempc_amd.tuner
 *   Pmpc tracking reference W, yref    (pmpc.py:594-609,961-974) tmpc_tracking_reference_host (consumer of Hc, q)
 *   Pocp.get_sensitivities post-processing (pocp.py:322-361) tmpc_pack_sensitivities_host (producer of H, C_As, q)
 *
 * Conventions: plain pointers + sizes, fp64, C (row-major) contiguous arrays:
 *   A  [B][p][nx][nx]      B  [B][p][nx][mb]      H  [B][p][n][n]   (n = nx + mb, H = [[Q,N],[N',R]])
 *   Hc, dHc [B][p][n][n]   P  [B][p][nx][nx]  (P = the reference's un-scaled dP_k, convexifier.py:406)
 * The caller owns every buffer; the library never frees or keeps caller memory.  `_device` variants
 * take device pointers and enqueue on `stream` (a hipStream_t passed as void*); they synchronise the
 * stream internally once per interior-point iteration (a 4-byte "problems still active" read-back).
 * nb = 0 is a no-op that returns TMPC_OK (an empty shard of a batch split over several GPUs).
 * Unit-test and diagnostic entry points are declared in tunempc_hip_debug.h, not here.
 * All functions return 0 on success or a negative TMPC_E_* code; per-problem solver outcomes are in
 * status[] so that one infeasible member does not abort a batch.
 */
#ifndef TUNEMPC_HIP_H
#define TUNEMPC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMPC_OK 0
#define TMPC_E_ARG (-1)        /* bad argument (null pointer, non-positive size)            */
#define TMPC_E_UNSUPPORTED (-2) /* shape outside what the handle / entry supports: nx+mb > 96 (> 64 with rows or Step 3), more than TMPC_MAX_ROWS rows, Schur blocks wider than 3168, p < 1; tight mode on a handle with rows */
#define TMPC_E_NOMEM (-3)      /* hipMalloc failed                                          */
#define TMPC_E_HIP (-4)        /* HIP runtime error (see tmpc_last_error)                   */
#define TMPC_E_NODEVICE (-5)   /* no gfx950 device visible                                  */
#define TMPC_E_NOCONV (-6)     /* tmpc_eig_clip_host: Jacobi sweeps exhausted (outputs hold the last iterate) */

#define TMPC_STATUS_OPTIMAL 0    /* convexifier.py:444 'Optimal'    */
#define TMPC_STATUS_FEASIBLE 1   /* convexifier.py:446 'Feasible'   */
#define TMPC_STATUS_INFEASIBLE 2 /* convexifier.py:451 'Infeasible' */

#define TMPC_FLAG_NO_MFMA 1      /* debug: scalar-FMA GEMM fragments on the register-staged kernels instead of the matrix cores (product path: v_mfma_f64_4x4x4 tiles fed by LDS-DMA) */
#define TMPC_FLAG_PROFILE 2      /* record hipEvent timings per phase (tmpc_get_profile)           */
#define TMPC_FLAG_FAST_EXIT 4    /* stop every problem after its FIRST full centering step instead of converging to the central-path point
                                    at mu_target: Hc is feasible (positive definite, cond <= kappa), kappa within the same gap N mu_target of
                                    optimal, the linear residuals are gone -- but the point is within ~1e-2 (relative) of the centred one, not
                                    at it, so two implementations agree on it to ~1e-3 only, not to 1e-8.  ~2 factorisations fewer per problem.
                                    Members stopped this way carry info[10] = 3.  Off by default: the default answer is the reproducible one. */

#define TMPC_INFO_STRIDE 16      /* doubles per problem in info[] (layout below)                   */
/* info[b*16 + i]: 0 s (=1/min|eig H|), 1 sbeta, 2 min eig H, 3 min eig Hc, 4 max cond Hc, 5 mu,
 *                 6 mu_target, 7 pinf, 8 dinf, 9 relgap, 10 ipm status (0 opt,1 inaccurate,2 maxiter,3 = stopped by TMPC_FLAG_FAST_EXIT: status Optimal, not the converged point,
 *                    4 = tight mode only: the continuation of this member failed and the result of its DEFAULT solve was returned -- status Optimal at the default gap, info[6]),
 *                 11 #shifted pivots, 12 centering iterations, 13 early-exit flag (convexifier.py:83-85),
 *                 14 last centering step norm, 15 smallest Cholesky pivot of the Schur factorisations relative to the assembled diagonal (1 if never below 1e-8; <= 1e-15 = frozen)                                          */

typedef struct tmpc_handle tmpc_handle;

/* Number of HIP devices visible (0 if none / HIP not initialisable).  Never touches a device. */
int tmpc_device_count(void);

/* Device workspace needed for `chunk` problems of shape (p, nx, mb), in bytes (0 if unsupported).
 * Supported: p >= 1, nx + mb <= 64 for every model (tuned per-stage kernels up to 32, generic ones above) and <= 96 for the plain model (no G / C rows, no Step 3),
 * rows of G_k / C_k up to TMPC_MAX_ROWS each, and
 * Schur blocks -- nx(nx+1)/2, plus the rows, plus (Step 3) the (nx+mb)(nx+mb+1)/2 + 1 entries of T_k and its epigraph variable -- of at most 3168 (round 4: 2384). */
uint64_t tmpc_workspace_bytes(int chunk, int p, int nx, int mb);

/* Create a handle on the current HIP device with workspace for `chunk` problems per launch wave.
 * Larger batches are processed in chunks.  chunk <= 0 selects a default that fits free HBM. */
int tmpc_create(tmpc_handle** out, int chunk, int p, int nx, int mb);
#define TMPC_MAX_ROWS 31         /* rows of G_k, and rows of C_k, per stage (each)                     */
#define TMPC_ARROW_LD 32         /* leading dimension of the arrow blocks tmpc_get_dual_con_host exports */
/* The same with room for `ng` equality-constraint rows per stage (0 <= ng <= TMPC_MAX_ROWS), for tmpc_convexify_eq_batch_host.
 * Such a handle also serves every call that takes no G. */
uint64_t tmpc_workspace_bytes_eq(int chunk, int p, int nx, int mb, int ng);
int tmpc_create_eq(tmpc_handle** out, int chunk, int p, int nx, int mb, int ng);
/* The same with room for up to `nc` active-constraint rows per stage as well (0 <= nc <= TMPC_MAX_ROWS),
 * for tmpc_convexify_step2_batch_host. */
uint64_t tmpc_workspace_bytes_con(int chunk, int p, int nx, int mb, int ng, int nc);
int tmpc_create_con(tmpc_handle** out, int chunk, int p, int nx, int mb, int ng, int nc);
int tmpc_destroy(tmpc_handle* h);
/* Problems processed per launch wave (the chunk the workspace was sized for). */
int tmpc_get_chunk(tmpc_handle* h);

/* Solver options: tol = complementarity tolerance mu_target/kappa per unit cone dimension, default 2^-25
 * (the relative duality gap on kappa, the max condition number, is then (2*p*n+1)*tol);
 * center_tol = relative Newton step ending the final centering phase, default 1e-9;
 * max_iter / center_iter = iteration caps (defaults 50 / 12): max_iter bounds the main phase, center_iter the centering iterations
 * per barrier target (a chord step, which re-uses the factorisation at a fifth of the cost, counts a quarter: up to 4 x center_iter cheap iterations) -- a hard target may visit up to 11 targets (ten back-offs by powers of two, reported in info[6]), so a problem ends
 * after at most max_iter + 11 * center_iter + 2 iterations; flags = TMPC_FLAG_*.  Values <= 0 keep
 * the current setting (flags is always applied; bits other than the TMPC_FLAG_* above -- and the debug bit of
 * tunempc_hip_debug.h -- are rejected with TMPC_E_ARG). */
int tmpc_set_options(tmpc_handle* h, double tol, double center_tol, int max_iter, int center_iter, int flags);

/* Performance knobs of a handle (none of them changes WHAT is computed beyond rounding; defaults are the measured optimum on MI355X):
 *   TMPC_TUNE_CHORD_STEP   value > 0: the centering phase re-uses a factorisation once a full Newton step could have been `value` times longer
 *                          before leaving the cone (default 10); 0: every centering step re-factors
 *   TMPC_TUNE_SMALL_BLOCKS 1 (default): blocks of one 16 x 16 tile (the reference's own examples, nx <= 5) are factored / solved by ONE kernel per
 *                          problem; 0: the batched launch sequence per elimination level
 *   TMPC_TUNE_EIG_PRETEST  1 (default): the step-length kernel first asks whether the step at the clipping threshold stays in the cone; 0: every
 *                          eigenvalue is computed
 *   TMPC_TUNE_FUSE_FWD     1 (default): the forward substitution of the predictor pass rides inside the factorisation; 0: separate sweep
 *   TMPC_TUNE_GRAPH        1 (default): problems whose Schur blocks are a single tile (nx <= 10: launch-bound) replay the launch sequence of an iteration as a
 *                          captured hipGraph (one submission instead of ~35; the handle's own streams only); 0: plain launches
 *   TMPC_TUNE_LOWP_SWITCH  value >= 0 (default TMPC_LOWP_SWITCH_DEFAULT): in the main-phase iterations of a problem with mu > value * max(1, kappa) the Schur-complement updates of
 *                          the block factorisation (k_cr_update_dma, a third of a solve) run on float32 copies of their operands with float32 accumulation (fp32 MFMA:
 *                          twice the fp64 matrix rate); Cholesky, triangular solves, substitutions and every later iteration stay fp64.  Same iteration counts, the
 *                          converged point moves by 1e-11 ... 2e-10 (profiles/r6_fp32_*.txt).  Steps 1 and 2 (not Step 3), stage blocks up to 32 x 32, Schur blocks of 80 ... 320; a pivot that freezes under
 *                          them repeats the iteration in fp64 and turns them off for that problem.  0: never (rounds 1-5, bit for bit)
 *   TMPC_TUNE_PERSISTENT   plain-model problems with single-tile Schur blocks and n = nx + mb <= 8 (the reference's own examples) can run their whole
 *                          interior-point loop as ONE launch, one workgroup per problem (tmpc_persist.h).  1 (default): where that is faster -- period
 *                          p <= 8, or at least 96 problems of the call on the chip at once; 0: never (the launch sequence); 2: whenever the shape allows it
 * (Rounds 1-3 read these from environment variables once per process.) */
#define TMPC_TUNE_CHORD_STEP 1
#define TMPC_TUNE_SMALL_BLOCKS 2
#define TMPC_TUNE_EIG_PRETEST 3
#define TMPC_TUNE_FUSE_FWD 4
#define TMPC_TUNE_GRAPH 5
/* (key 6 belonged to two measured-and-dropped experiments -- rounds 5 and 6, profiles/r5_fused_elim.txt, profiles/r6_update_stream_*.txt -- and is not reused) */
#define TMPC_TUNE_PERSISTENT 7
#define TMPC_TUNE_LOWP_SWITCH 8
#define TMPC_LOWP_SWITCH_DEFAULT 1e-5
int tmpc_set_tuning(tmpc_handle* h, int key, double value);
/* The general constructor: ng / nc rows of G_k / C_k (0: none), step3 != 0: room for T_k, lanes = concurrent half-waves on their own streams
 * (0: automatic -- two for problems whose blocks are a single 64 x 64 tile and chunk >= 2, one otherwise; at most 4). */
int tmpc_create_ex(tmpc_handle** out, int chunk, int p, int nx, int mb, int ng, int nc, int step3, int lanes);

/* Tight-accuracy mode (opt-in; Step 1 and Step 2 handles with nx <= 51 and nx + mb <= 64; with rows of G / C -- round 5 -- while
 * rows * (2 (nx + mb) + 2 nx) <= 4040; no Step 3).  The reference hands its SDP to MOSEK / CVXOPT, which stop at a relative gap of
 * ~1e-8 (convexifier.py:363); the default solve above stops at tol = 2^-25, a certified gap of (2*p*n+1)*3e-8 on kappa, because the HKM
 * Schur matrix (condition ~1/mu^2) cannot be factored in fp64 below mu ~ 1e-8.  With enable != 0 every problem that ended Optimal is
 * continued from its centred point towards mu_target = tight_tol * kappa (default 2^-37 ~ 7.3e-12: gap (2*p*n+1)*7.3e-12; accepted range
 * [2^-42, 1), every member of the test families converges down to 2^-37, most down to 2^-41) with the Kronecker-factor images, the
 * assembly, the block Cholesky and the substitutions in double-double arithmetic, and finished by Newton steps on the dual barrier
 * problem in which every stage quantity is double-double (the returned point is then reproducible to ~1e-12 instead of ~eps/mu).
 * Outputs as before; info[6] = the mu_target reached, iters includes the extra iterations.  A member whose continuation fails (a non-positive
 * double-double pivot, a polish step that leaves the cone, the iteration cap: most visibly hard targets below 2^-33) gets the result of its default
 * solve back, status Optimal, info[10] = 4 and info[6] = the default's mu_target: the mode never returns less than the default does, and says so.  Costs one more workspace of about the size of
 * the block storage (allocated at the first enable) and ~10 double-double factorisations per problem (vector ALU, no matrix cores).
 * With rows (Step 1 with G; the Step 2 model with either objective) the multipliers and the epigraph variables of the norm terms ride in the augmented blocks
 * as in the default solve, their rows formed in double-double, and join the polish as variables; tmpc_get_dual_host / tmpc_get_dual_con_host export the dual
 * iterate that goes with the last Newton step of the polish (LMI blocks, multipliers z, primal blocks of the norm terms): it certifies the gap to ~N * tight_tol.
 * enable == 0 switches back to the default (the workspace stays).  TMPC_E_UNSUPPORTED for Step 3 handles and for rows beyond the limits above. */
int tmpc_set_tight(tmpc_handle* h, int enable, double tight_tol);

/* Step 1 of convexifier.convexify for `nb` independent problems.  Any output pointer may be NULL.
 * status/iters are int32 [nb]; alpha/beta/kappa are double [nb]; info is double [nb][16]. */
int tmpc_convexify_batch_host(tmpc_handle* h, int nb, const double* A, const double* B, const double* H,
                              double* Hc, double* dHc, double* P, double* alpha, double* beta, double* kappa,
                              int32_t* status, int32_t* iters, double* info);
int tmpc_convexify_batch_device(tmpc_handle* h, int nb, const double* dA, const double* dB, const double* dH,
                                double* dHc_out, double* ddHc_out, double* dP_out, double* d_alpha, double* d_beta,
                                double* d_kappa, int32_t* d_status, int32_t* d_iters, double* d_info, void* stream);

/* Step 1 with the equality-constraint term (convexifier.py:249-255 multipliers Fg_k >= 0, :346-347 term G_k' diag(Fg_k) G_k
 * in M_k, :409-411 un-scaling): G [nb][p][ng][n] with ng of tmpc_create_eq; Fg [nb][p][ng] out.  dHc includes the G term
 * (convexifier.py:196-197).  The multipliers of stage k ride inside block k+1 of the block factorisation
 * (tunempc_amd/csrc/tmpc_phi.h). */
int tmpc_convexify_eq_batch_host(tmpc_handle* h, int nb, const double* A, const double* B, const double* H, const double* G,
                                 double* Hc, double* dHc, double* P, double* Fg, double* alpha, double* beta, double* kappa,
                                 int32_t* status, int32_t* iters, double* info);

/* The model of Step 2 (convexifier.py:116-131: setUpModelPicos with constr=True, :258-266 multipliers F_k >= 0 of the active
 * constraints, :276-283 objective beta + sum rho*||F_k|| + sum rho*||Fg_k||, :348-350 term C_k' diag(F_k) C_k, :415-420 un-scaling).
 * J [nb][p][ng+nc][n]: per stage the ng rows of G_k, then the rows of C_k, zero padding up to nc (ng, nc of tmpc_create_con);
 * ncnt int32 [nb][p]: rows of C_k actually present (0 for a stage whose C_k is None).  FgF [nb][p][ng+nc] out: Fg_k, then F_k,
 * zeros in the padding.  dHc includes both constraint terms.  The caller decides when to take this step (after Step 1 came
 * back Infeasible, as convexify() does).
 * rho = 0 selects the BETA-ONLY objective: the reference assembles the norm terms with `picos.sum(obj, abs(rho*F[i]))`
 * (convexifier.py:276-283); whether PICOS 1.2.0 adds or drops that second argument cannot be checked here (the package is not installed,
 * SURVEY.md 7.0).  If it drops it, the solver sees min beta with cost-free multipliers F_k, Fg_k >= 0 -- the rows of C_k then act exactly
 * like rows of G_k -- which is what rho = 0 solves (no norm cones at all, not a zero-weight limit of them).  rho > 0 is the paper's
 * objective (eq. 20a) and the default of the Python mirror. */
int tmpc_convexify_step2_batch_host(tmpc_handle* h, int nb, const double* A, const double* B, const double* H, const double* J,
                                    const int32_t* ncnt, double rho, double* Hc, double* dHc, double* P, double* FgF, double* alpha,
                                    double* beta, double* kappa, int32_t* status, int32_t* iters, double* info);

/* The model of Step 3 (convexifier.py:137-147: setUpModelPicos with force=True, :269-273 T_k symmetric with every entry > 0, :284-285
 * objective + sum rho*||T_k||_F, :352-353 term s_T*T_k, :422-423 un-scaling), here for the plain model (no G / C rows in the same solve):
 * handle from tmpc_create_step3; T [nb][p][n][n] out; dHc includes T_k (convexifier.py:202-203).  The caller decides when to take
 * this step (after the earlier steps came back Infeasible and with the 'force' option, as convexify() does).  The norm term is a
 * second-order cone handled natively (tunempc_amd/csrc/tmpc_t3.h); the blocks of the factorisation grow to d + n(n+1)/2 + 1. */
uint64_t tmpc_workspace_bytes_step3(int chunk, int p, int nx, int mb);
int tmpc_create_step3(tmpc_handle** out, int chunk, int p, int nx, int mb);
int tmpc_convexify_step3_batch_host(tmpc_handle* h, int nb, const double* A, const double* B, const double* H, double rho,
                                    double* Hc, double* dHc, double* P, double* T, double* alpha, double* beta, double* kappa,
                                    int32_t* status, int32_t* iters, double* info);

/* Step 3 with the multipliers of G and C in the same solve (convexifier.py:144: setUpModelPicos(..., constr = constraint_contribution,
 * force = True)): handle from tmpc_create_step3_con (ng, nc as in tmpc_create_con); J, ncnt, FgF as in tmpc_convexify_step2_batch_host;
 * ncnt == NULL: no C rows -- J holds the ng rows of G only, cost-free multipliers as in tmpc_convexify_eq_batch_host (the reference's
 * constr = False).  T [nb][p][n][n] out; dHc includes the constraint terms and T_k. */
uint64_t tmpc_workspace_bytes_step3_con(int chunk, int p, int nx, int mb, int ng, int nc);
int tmpc_create_step3_con(tmpc_handle** out, int chunk, int p, int nx, int mb, int ng, int nc);
int tmpc_convexify_step3_con_batch_host(tmpc_handle* h, int nb, const double* A, const double* B, const double* H, const double* J,
                                        const int32_t* ncnt, double rho, double* Hc, double* dHc, double* P, double* FgF, double* T,
                                        double* alpha, double* beta, double* kappa, int32_t* status, int32_t* iters, double* info);
/* Device-resident forms of the two Step 3 entries (device pointers in and out; `stream`: the caller's HIP stream, NULL = default -- the call returns
 * with every result written, as tmpc_convexify_batch_device). */
int tmpc_convexify_step3_batch_device(tmpc_handle* h, int nb, const double* dA, const double* dB, const double* dH, double rho, double* Hc, double* dHc,
                                      double* P, double* T, double* alpha, double* beta, double* kappa, int32_t* status, int32_t* iters, double* info, void* stream);
int tmpc_convexify_step3_con_batch_device(tmpc_handle* h, int nb, const double* dA, const double* dB, const double* dH, const double* dJ, const int32_t* d_ncnt,
                                          double rho, double* Hc, double* dHc, double* P, double* FgF, double* T, double* alpha, double* beta, double* kappa,
                                          int32_t* status, int32_t* iters, double* info, void* stream);

/* Device-resident form of the two entries above (inputs and outputs in HBM, work queued on `stream`): d_ncnt == NULL is Step 1
 * with G (dJ = G [nb][p][ng][n], FgF [nb][p][ng]); otherwise the Step 2 model (dJ [nb][p][ng+nc][n], d_ncnt [nb][p] with
 * 0 <= ncnt <= nc -- not checked here --, FgF [nb][p][ng+nc]). */
int tmpc_convexify_con_batch_device(tmpc_handle* h, int nb, const double* dA, const double* dB, const double* dH, const double* dJ,
                                    const int32_t* d_ncnt, double rho, double* dHc_out, double* ddHc_out, double* dP_out, double* dFgF,
                                    double* d_alpha, double* d_beta, double* d_kappa, int32_t* d_status, int32_t* d_iters,
                                    double* d_info, void* stream);

/* convexHessianSuppl (convexifier.py:165-211) alone: dHc_k = sym(V_k' P_{k+1} V_k - E' P_k E). */
int tmpc_supplement_batch_host(tmpc_handle* h, int nb, const double* A, const double* B, const double* P, double* dHc);
/* The same with the constraint and regularisation terms of convexifier.py:196-204:
 *   dHc_k = sym(V_k' P_{k+1} V_k - E' P_k E + J_k' diag(w_k) J_k + T_k),
 * J [nb][p][nr][n]: rows of G_k (weights Fg_k) followed by rows of C_k (weights F_k), zero-weight padding up to nr rows
 * per stage (ragged C_k, per-stage None = all weights zero); wts [nb][p][nr]; T [nb][p][n][n].  J/wts and T may be NULL. */
int tmpc_supplement_terms_batch_host(tmpc_handle* h, int nb, const double* A, const double* B, const double* P, int nr,
                                     const double* J, const double* wts, const double* T, double* dHc);

/* Producer side: the array post-processing of Pocp.get_sensitivities (pocp.py:322-361) that turns the raw NLP sensitivities into
 * the inputs of convexify(), for nb problems at once (p, n = nx + mb of the handle):
 *   C [nb][p][nh][n] path-constraint Jacobians and mu [nb][p][nh] their multipliers (both NULL: no path constraints):
 *     C_As [nb][p][ncmax][n] = the rows with |mu| > thr in their original order, zero-padded (pocp.py:322-340; threshold :73),
 *     ncnt [nb][p] their number (a value > ncmax means the padding was too small), idx [nb][p][nh] their row indices (-1 padding,
 *     may be NULL), q [nb][p][n] = -mu' C (pocp.py:357-361; zeros without constraints; may be NULL);
 *   Hbig [nb][p*n][p*n] Lagrangian Hessian of the whole NLP (may be NULL): Hst [nb][p][n][n] its diagonal stage blocks (:350-355).
 * C_As / ncnt are the J / ncnt inputs of tmpc_convexify_step2_batch_host. */
int tmpc_pack_sensitivities_host(tmpc_handle* h, int nb, int nh, const double* C, const double* mu, const double* Hbig, double thr, int ncmax,
                                 double* C_As, int32_t* ncnt, int32_t* idx, double* q, double* Hst);

/* Consumer side of the tuned matrices, the tracking-MPC reference update (pmpc.py:961-974; set-up :594-609):
 *   W_k = sym(Hc_k) / ts,   yref_k = wref_k - (Hc_k/ts)^-1 q_k / ts = wref_k - Hc_k^-1 q_k      for nstage independent stages.
 * Hc [nstage][n][n] (n = nx + mb of the handle), q, wref, yref [nstage][n], W [nstage][n][n]; info[k] = number of
 * non-positive Cholesky pivots of Hc_k (0 for every matrix convexify() reports Optimal/Feasible).  W, info may be NULL. */
int tmpc_tracking_reference_host(tmpc_handle* h, int nstage, const double* Hc, const double* q, const double* wref,
                                 double ts, double* W, double* yref, int32_t* info);

/* Stage-block eigen scan (pre-check convexifier.py:82, autoScaling :374-401, status check :438-440):
 * out[b*p+k][0..3] = min eig, max eig, min |eig| (zeros excluded), max |eig| of sym(H[b][k]). */
int tmpc_eig_scan_host(tmpc_handle* h, int nb, const double* H, double* out);

/* Eigenvalue clip of general-size symmetric matrices (reference: tunempc/sqp_method.py:327-403, `Sqp.__regularize_hessian`: the
 * eigenvalues of the (reduced) Hessian below `regularization_tol` are lifted to it, H += evec diag(evmod - eva) evec^-1):
 *   out[b] = sym(A[b]) + V diag(max(tol - lambda_i, 0)) V',   A, out [nb][n][n], any n >= 1 (no handle, current device).
 * evals [nb][n] (optional): the eigenvalues lambda_i of A[b] (unordered); reg [nb] (optional): the largest lift max_i(tol - lambda_i, 0)
 * (the reference's `self.__reg`); sweeps [nb] (optional): Jacobi sweeps taken.  Returns TMPC_E_NOCONV when 40 sweeps did not
 * orthogonalise the vectors to the rounding level (~2 sqrt(n) eps); the outputs then hold the last iterate. */
int tmpc_eig_clip_host(int nb, int n, const double* A, double tol, double* out, double* evals, double* reg, int32_t* sweeps);

/* Accumulated hipEvent timings since the last call (ms) when TMPC_FLAG_PROFILE is set, 16 doubles:
 * out[0] stage_pre+ctrl, [1] schur assembly, [2] block factorisation (all kernels of tmpc_cr.h's factor phase), [3] predictor
 * pass, [4] corrector pass + update, [5] number of factorisation phases (= IPM iterations of the chunks), [6] total ms of the
 * convexify calls, [7] IPM iterations (max over chunk, summed over chunks), [8] problem-factorisations (sum over the phases of
 * the problems still iterating), [9] / [10] / [11] ms inside k_cr_potrf / k_cr_trsm / k_cr_update (fp64), [12] lanes of the handle, [13] problem-factorisations
 * whose Schur-complement updates ran in single precision (TMPC_TUNE_LOWP_SWITCH), [14] ms inside k_cr_update_dma_f32.
 * Counted with or without the flag: [15] problems solved through the one-launch kernel of TMPC_TUNE_PERSISTENT (k_ipm_small) since the last call. */
int tmpc_get_profile(tmpc_handle* h, double* out16);

/* Optimality certificate of the LAST wave solved (nb <= chunk, plain Step 1 model): the DUAL iterate of the interior-point method,
 * i.e. the multipliers of the 2p LMIs of convexifier.py:304-306 in the scaled problem
 *     min tau  s.t.  S1_k = M_k - I >= 0,  S2_k = tau I - M_k >= 0,  alpha - 1e-8 >= 0,   M_k = alpha s H_k + calH_k(Pbar):
 * X1, X2 [nb][p][n][n] (>= 0) and scal [nb][4] = (x0, tau, alpha, mu_target).  For a dual-feasible triple (sum_k tr X2_k = 1,
 * sum_k <s H_k, X1_k - X2_k> + x0 = 0, calH*(X1 - X2) = 0) weak duality gives  sum_k tr X1_k + 1e-8 x0  <=  kappa* (the optimal
 * max condition number), so together with the primal point (P, alpha, kappa outputs: cond(Hc_k) <= kappa) a caller can bound the
 * optimality gap of kappa without trusting this solver (tests/test_gpu_parity.py::test_dual_certificate does it in numpy).
 * Any pointer may be NULL.  Early-exit members (already convex) hold no meaningful dual. */
int tmpc_get_dual_host(tmpc_handle* h, int nb, double* X1, double* X2, double* scal);
/* The dual side of the stage-local multipliers of the LAST wave solved by a handle with G / C rows (Step 1 with G, Step 2 model), scaled problem, for the
 * same solver-independent certificate (tests/test_gpu_parity.py::test_dual_certificate_with_multipliers):
 *   phi [nb][p][nr]  the multipliers s*[Fg_k; F_k] (nr = ng + nc of the handle; entries beyond a stage's row count are padding),
 *   z   [nb][p][nr]  their duals (phi_i >= 0  <->  z_i >= 0),
 *   aX  [nb][p][2][TMPC_ARROW_LD][TMPC_ARROW_LD], at [nb][p][2]   Step 2 with rho > 0 only (else pass NULL): the primal blocks X_e of the arrow LMIs of the (up to) two norm terms
 *                    per stage (leading (m_e + 1) x (m_e + 1) part valid; term 0 = the rows of G if ng > 0, then the rows of C_k) and the epigraph variables t_e.
 * X1, X2, x0, tau, alpha, mu_target come from tmpc_get_dual_host.  Any pointer may be NULL. */
int tmpc_get_dual_con_host(tmpc_handle* h, int nb, double* phi, double* z, double* aX, double* at);

/* Per-iteration diagnostics of the LAST chunk solved: out[nb][80][10] = (iteration, phase, mu, tau, pinf, dinf,
 * primal step, dual step, relative output change of the step, cumulative shifted pivots); nb <= chunk. */
int tmpc_get_trace(tmpc_handle* h, int nb, double* out);

const char* tmpc_last_error(void);
const char* tmpc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TUNEMPC_HIP_H */
