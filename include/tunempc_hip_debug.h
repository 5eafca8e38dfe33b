/*
 * tunempc_hip_debug.h -- unit-test and diagnostic entry points of libtunempc_hip.so.
 *
 * NOT part of the drop-in boundary (that is tunempc_hip.h): the tests call the building blocks of the kernels through the
 * same shared library, and the scripts under scripts/ read back raw workspace arrays.  Nothing here has a counterpart in the
 * reference.
 */
#ifndef TUNEMPC_HIP_DEBUG_H
#define TUNEMPC_HIP_DEBUG_H

#include "tunempc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Debug bit for tmpc_set_options(flags): the solve stops after ASSEMBLING the Schur system of the first iteration, which then sits
 * unfactored in the workspace for tmpc_debug_get_array (tests/tools/step3_asm_check.py).  The outputs of such a call are NOT a
 * solution -- never set it in product code. */
#define TMPC_DEBUG_FLAG_STOP_ASSEMBLED 8
/* Debug bit: no relative lift of the Schur diagonal after frozen pivots (REG_MAX = 0): frozen pivots in the centering phase then go straight to the
 * route 'back mu_t off and take the step of this factorisation' (ctrl_backoff_before_rhs / k_ctrl_c), which the lifts make rare in practice
 * (tests/test_gpu_hard_targets.py::test_backoff_step_is_taken).  Never set it in product code. */
#define TMPC_DEBUG_FLAG_NO_LIFT 16
/* Debug bit: the register-staged factorisation kernels (k_cr_potrf / k_cr_trsm / k_cr_update: the path of blocks wider than 320) for every block
 * size, instead of the LDS-DMA kernels -- keeps that path under test at small shapes. */
#define TMPC_DEBUG_FLAG_NO_DMA 32
/* Debug bit: the generic per-stage kernels of stage blocks wider than 32 (tmpc_big.h; with them the <true> forms of the multiplier and Step 3 kernels) also at n <= 32, so that they can be
 * compared with the tuned kernels on the same inputs. */
#define TMPC_DEBUG_FLAG_GENERIC_STAGE 64

/* C (M x N) <op> A (M x K) * B (N x K)' with the fp64 MFMA tile GEMMs of the factorisation; mode 0: C -= AB', 1: C = AB', 2: C = -AB'.
 * mode + 0: the register-staged core (tmpc_factor.h, one workgroup walks all tiles); + 16: the LDS-DMA tile core (tmpc_gemm_dma.h, one
 * workgroup per 64 x 64 tile); + 32: the LDS-DMA core with K split into two operand pairs of K/2 as one stream.  lower != 0: only the
 * part on and below the diagonal (register-staged: 64 x 64 tiles and the waves that reach the diagonal; LDS-DMA: 16 x 16 blocks).
 * All dims multiples of 16 (K of 32 for + 32). */
int tmpc_debug_gemm_nt(tmpc_handle* h, double* C, const double* A, const double* B, int M, int N, int K, int mode, int lower);

/* Factor + solve one SPD block-cyclic-tridiagonal system with the cyclic-reduction kernels (tmpc_cr.h):
 * D [p][d][d] diagonal blocks, Ccpl [p][d][d] with Ccpl[k] = T[block k, block k+1 mod p], rhs/x [p][d]. */
int tmpc_debug_block_solve(tmpc_handle* h, int p, int d, const double* D, const double* Ccpl, const double* rhs, double* x, int32_t* nshift);

/* The elimination schedule for period p (host only, no device needed): out = [nlev, prep, nelim, nupd, nlev x (eoff, nelim,
 * uoff, nupd), nelim x 8 ints (node, na, nb, ea, eb, fill, fx, facc), nupd x 8 ints (node, e0, src0, e1, src1, 0, 0, 0),
 * p x orientation].  Returns the number of ints; call with out = NULL to size the buffer. */
int tmpc_debug_cr_schedule(int p, int32_t* out, int cap);

/* Scaled stage-local multipliers of the LAST chunk solved (row stride nr of that call): phi, their duals z and the last
 * directions, each [nb][p][nr]; any pointer may be NULL. */
int tmpc_debug_get_multipliers(tmpc_handle* h, int nb, int nr, double* phi, double* z, double* dphi, double* dz);

/* Raw workspace read-back for diagnostics: which = 0 psm, 1 pvec, 2 Ddiag, 3 D, 4 part, 5 O (edge slots 0..p-1), 6 F (fill slots);
 * `count` doubles from `offset` (no bounds check beyond the pointer being allocated). */
int tmpc_debug_get_array(tmpc_handle* h, int which, uint64_t offset, uint64_t count, double* out);

/* Smallest eigenvalue of nmat symmetric n x n matrices (Householder tridiagonalisation + Sturm multisection). */
int tmpc_debug_min_eig(tmpc_handle* h, int nmat, int n, const double* W, double* out);
/* The same for n <= 8 by the one-thread-per-matrix routine the small shapes use (lane_min_eig8: Householder + Laguerre in registers; round 5). */
int tmpc_debug_min_eig_lane(tmpc_handle* h, int nmat, int n, const double* W, double* out);

/* Isolated timing of the block factorisation and of one single-right-hand-side solve on nb copies of one random SPD system:
 * ms_out2[0] = factorisation, [1] = solve (averages over `reps`). */
int tmpc_debug_factor_bench(tmpc_handle* h, int nb, int p, int d, int reps, double* ms_out2);

#ifdef __cplusplus
}
#endif
#endif /* TUNEMPC_HIP_DEBUG_H */
