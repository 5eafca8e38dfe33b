// Small problems (single-tile Schur blocks, n <= 8: the reference's own examples -- LQR, unicycle, CSTR, evaporation; BASELINE configs[1] / [2]):
// the WHOLE interior-point loop of a problem as ONE launch, one 16-wave workgroup per problem (round 5).
//
// The launch-sequence path (run_chunk in tmpc_api.hip) spends ~22 dependent launches, a list compaction and a host read-back per iteration on
// work that is a few microseconds per kernel at these sizes: 4.4 ms per unicycle-shaped solve against 5.7 ms on ONE CPU core
// (BENCH_r04.json:small_configs).  Here every phase of an iteration is a loop of the sixteen waves over the stages of the problem -- the
// bodies of the per-stage kernels in their one-wave form on 8 x 9 LDS slots (namespace sm8: tmpc_small_impl.h, tmpc_stage_impl.h), the
// assembly body of k_schur on one wave, the one-workgroup factorisation and substitution of tmpc_cr_small.h, the border solve and the four
// control bodies (tmpc_schur.h) -- separated by workgroup barriers.  State stays where the launch-sequence path keeps it (global memory,
// L2-resident: a p = 30, n = 5 problem is ~0.4 MB), so both paths run the SAME code on the same data layout and differ by rounding only
// (sums of the 4-wave kernels are taken by one wave here).  No list, no host round trip, no launch between the first iteration and the last.
//
// What it buys and what it cannot (profiles/r5_persist_*.txt): the sixteen waves take the stages sixteen at a time, so a batch that fills the chip
// gains 1.7 - 1.9 x (BASELINE configs[2]: 446 k -> 850 - 920 k stage-conv/s) and short periods gain 1.05 - 1.2 x at any batch, while ONE long problem is slower than on the launch
// sequence, which spreads its stages over the CUs -- run_chunk chooses (TMPC_TUNE_PERSISTENT).  The cost of an iteration is a chain of dependent LDS / L2 round trips on
// matrices of a few dozen entries (0.2 ms at p = 4, 0.43 ms at p = 30), not launches: the phases below are where it goes.
#pragma once
#include "tmpc_common.h"
#include "tmpc_cr_small.h"
#include "tmpc_factor.h"
#include "tmpc_schur.h"
#include "tmpc_stage.h"

namespace tmpc {

constexpr int PK_NMAX = sm8::NMAX;                         // stage blocks up to 8 x 8
constexpr int PK_WAVE_DOUBLES = 6 * sm8::MS + 16;          // per wave: the six slots of the stage bodies (>= the step-length scratch MS + 160 and the records of schur_body)
static_assert(PK_WAVE_DOUBLES >= sm8::MS + 160, "step-length scratch");
static_assert(PK_WAVE_DOUBLES >= 25 * 10 + 16, "records of schur_body<0> at nx = 5");
constexpr int pk_lds_doubles(int p) {
  int v = CRS_NW * PK_WAVE_DOUBLES;
  if (crs_factor_lds_doubles() > v) v = crs_factor_lds_doubles();
  if (crs_solve_lds_doubles(p) > v) v = crs_solve_lds_doubles(p);
  return v;
}

// cap: iterations per problem (run_chunk's bound); chord_pre: unused since the step-length eigenvalues are computed one per THREAD (eigmin_lane_body: always the exact value)
__global__ void __launch_bounds__(CRS_NT) k_ipm_small(WS w, Dims dm, Opts o, CrDev cr, CrLevs lv, int prep, int reg_max, double chord_pre, int cap) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int b = w.alist ? w.alist[blockIdx.x] : (int)blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  double* smw = lds + wv * PK_WAVE_DOUBLES;
  const int p = dm.p, s0 = b * p;
  volatile int* ip = w.iprob + (size_t)b * IS;
  // (-DTMPC_CYCLE_PROF: thread 0 of workgroup 0 adds the cycles of every phase to g_prof[48 ..]: scripts/persist_prof.py)
#ifdef TMPC_CYCLE_PROF
  unsigned long long tpk_ = __builtin_readcyclecounter();
#define PK_T(i) { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); g_prof[48 + (i)] += t_ - tpk_; tpk_ = t_; } }
#else
#define PK_T(i) __syncthreads();
#endif
  for (int it = 0; it < cap; ++it) {
    if (ip[I_PHASE] == PH_DONE) break;
    for (int k = wv; k < p; k += CRS_NW) sm8::stage_pre_body<64>(w, dm, s0 + k, lane, smw);
    PK_T(0)
    if (wv == 0) ctrl_a_body(w, dm, o, b, lane);
    PK_T(1)
    const int phase = ip[I_PHASE];
    if (phase == PH_DONE) break;
    if (!ip[I_CHORD]) {                                    // a new factorisation (chord steps keep the last one)
      for (int k = wv; k < p; k += CRS_NW) {
        schur_body<0, false, 64>(w, dm, s0 + k, lane, smw);
        schur_body<1, false, 64>(w, dm, s0 + k, lane, smw);
      }
      PK_T(2)
      if (prep) { cr_prep_body<CRS_NT>(w, dm, b, prep); __syncthreads(); }
      cr_small_factor_body(w, dm, cr, lv, b, lds);
      PK_T(3)
    }
    for (int pass = 1; pass <= 2; ++pass) {
      if (pass == 2 || phase == PH_MAIN) {                 // (centering: no predictor pass)
        for (int k = wv; k < p; k += CRS_NW) sm8::stage_rhs_body<64>(w, dm, s0 + k, lane, smw, pass);
        PK_T(4)
        for (int k = wv; k < p; k += CRS_NW) gather_body(w, dm, s0 + k, lane, pass);
        PK_T(5)
        cr_small_solve_body(w, dm, cr, lv, b, lds, pass);
        PK_T(6)
        solve_border_body<CRS_NT>(w, dm, b, pass, lds);
        PK_T(7)
        for (int k = wv; k < p; k += CRS_NW) sm8::stage_dir_body<64>(w, dm, s0 + k, lane, smw, pass);
        PK_T(8)
        for (int m = tid; m < 4 * p; m += CRS_NT) eigmin_lane_body(w, dm, 4 * s0 + m, pass);      // one thread per step-length matrix
        PK_T(9)
      }
      if (wv == 0) { if (pass == 1) ctrl_b_body(w, dm, reg_max, b, lane); else ctrl_c_body(w, dm, reg_max, b, lane); }
      PK_T(10)
    }
    for (int k = wv; k < p; k += CRS_NW) sm8::update_body<64>(w, dm, s0 + k, lane);
    // (update_body leaves at once when the phase is PH_DONE and ctrl_d_body may set it: every wave must be through its stages first -- on the launch
    // sequence k_update and k_ctrl_d are separate launches.  Without the barrier a wave that reaches its later stages after thread 0 is done skips their
    // last update: ADVICE r5, tests/test_gpu_parity.py::test_persistent_kernel_last_update_of_every_stage)
    __syncthreads();
    if (tid == 0) ctrl_d_body(w, dm, o, b, false);
    PK_T(11)
  }
#undef PK_T
}

}  // namespace tmpc
