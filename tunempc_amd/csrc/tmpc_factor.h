// Building blocks of the block factorisation of the HKM Schur matrix (tmpc_cr.h holds the kernels): the register-staged fp64 MFMA
// tile GEMM C (+)= A B' (v_mfma_f64_4x4x4_4b, 64 x 64 output tile per workgroup step, each wave a 32 x 32 sub-tile = 2 x 8 fragments of
// 16 x 4, K staged global -> VGPR -> LDS in double-buffered 32-column slabs; it carries the block Cholesky of k_cr_potrf, blocks wider
// than 320 and the scalar-FMA debug path -- the batched solves and updates run on the LDS-DMA core of tmpc_gemm_dma.h), the 64 x 64
// tile Cholesky with its inverse, the left-looking blocked Cholesky of a d x d block, and the skinny MFMA GEMM of the substitutions.
#pragma once
#include "tmpc_common.h"

namespace tmpc {


#ifndef TMPC_LDP
#define TMPC_LDP 65
#endif
constexpr int FACT_LDS_DOUBLES = 2 * 64 * TMPC_LDP + 72 + 4 * 16 * 17 + 8;   // the tile Cholesky (tile + inverse + pivot refs + one 16 x 17 scratch per wave); the <2,2,2> GEMM slabs (4*2*64*17) alias the front

enum { GM_SUB = 0, GM_SET = 1, GM_NEG = 2 };   // C -= A B',  C = A B',  C = -A B'
constexpr int GM_NOTRI = -(1 << 30);       // tmpc_gemm_dma.h: `tri` of a tile that does not touch the diagonal of a symmetric update

// C (M x N, ldc) <op> A (M x K, lda) * B (N x K, ldb)'   — all dims multiples of 16, K >= 16.
// Workgroup tile 64 x 64 (each wave a 32 x 32 quadrant = 2 x 2 MFMA tiles).  The K slabs (GK = 32 columns) of ALL
// tiles of the call form one software-pipelined stream: while the 32 MFMAs of slab s run, the global loads of
// slab s+1 -- possibly the first slab of the NEXT tile -- are in flight and land in the other LDS buffer; the C
// fragment of a read-modify-write tile is prefetched at the tile's first slab.  One barrier per slab, and only
// one exposed memory latency per call instead of two per tile.
// lower: skip 64x64 tiles strictly above the block diagonal (SYRK-style update of a symmetric block).
// In-place use (C aliasing A with one N-tile and K == its width) is safe: a tile's A slabs are all in LDS before
// its C fragment is stored, and the next tile reads other rows.
constexpr int SLD = 17, SUBD = 64 * SLD, SLABD = 2 * SUBD;   // sub-slab leading dim / doubles per sub-slab / per (operand) slab
#ifdef TMPC_NT
#define TMPC_LD(p) __builtin_nontemporal_load(p)
#else
#define TMPC_LD(p) (*(p))
#endif
typedef const double __attribute__((address_space(1)))* gcptr;
typedef const double2_t __attribute__((address_space(1)))* gcptr2;
// 32-byte piece (4 doubles, 32-byte aligned) as two 16-byte loads
#define TMPC_LD4(dst, off, p, ok)                                                       \
  {                                                                                     \
    const double2_t u0_ = (ok) ? ((gcptr2)(p))[0] : (double2_t){0.0, 0.0};              \
    const double2_t u1_ = (ok) ? ((gcptr2)(p))[1] : (double2_t){0.0, 0.0};              \
    dst[(off) + 0] = u0_[0]; dst[(off) + 1] = u0_[1]; dst[(off) + 2] = u1_[0]; dst[(off) + 3] = u1_[1]; \
  }
// the same piece of a float32 matrix (one 16-byte load), widened to double
#define TMPC_LD4F(dst, off, p, ok)                                                      \
  {                                                                                     \
    const float4_t u_ = (ok) ? *(const float4_t __attribute__((address_space(1)))*)(p) : (float4_t){0.f, 0.f, 0.f, 0.f}; \
    dst[(off) + 0] = (double)u_[0]; dst[(off) + 1] = (double)u_[1]; dst[(off) + 2] = (double)u_[2]; dst[(off) + 3] = (double)u_[3]; \
  }
typedef double __attribute__((address_space(1)))* gptr;
// WM x WN waves per workgroup (blockDim.x = 64 WM WN), FA fragments of 16 rows per wave: a wave owns (16 FA) x 32 of C, the
// workgroup tile is TM x TN = (16 FA WM) x (32 WN).  Two shapes are used: <2,2,2> = 64 x 64 with 256 threads (two workgroups per
// CU; inside the block Cholesky and the triangular solves) and <2,4,4> = 128 x 128 with 512 threads (one workgroup per CU, the
// same eight waves; half the global -> LDS traffic, LDS stores and barriers per flop) for the batched symmetric updates.
// Waves whose part of the tile lies outside M x N -- or, on a diagonal tile of a lower-only update, entirely above the
// diagonal -- skip their MFMAs and stores (they still help to load), so 304 = 128 + 128 + 48 costs like 128 + 128 + 64.
// NS: 16-column sub-slabs per K slab (slab = 16 NS columns of K): 2 in the block Cholesky; 1 halves the LDS of a workgroup
// (35 KB for the 64 x 64 shape) so that three workgroups share a CU.
template <int WM, int WN, int FA, int NS = 2>
struct GemmCfg {
  static constexpr int TM = 16 * FA * WM, TN = 32 * WN, NTH = 64 * WM * WN, RP = NTH / 4;      // RP: slab rows covered by one pass of the loader
  static constexpr int ARP = (TM + RP - 1) / RP, BRP = (TN + RP - 1) / RP;                       // loader passes over the A / B slab
  static constexpr int ASUB = TM * 17, BSUB = TN * 17, BUFD = NS * ASUB + NS * BSUB;             // doubles: 16-column sub-slab of A / B, one (A,B) buffer
  static constexpr int LDS_DOUBLES = 2 * BUFD;
};

template <bool USE_MFMA, int WM = 2, int WN = 2, int FA = 2, int NS = 2>
__device__ __forceinline__ void wg_gemm_nt(double* C, int ldc, const double* A, int lda,
                                           const double* B, int ldb, int M, int N, int K, int mode, bool lower,
                                           double* lds) {
  typedef GemmCfg<WM, WN, FA, NS> G;
  constexpr int GKT = 16 * NS;                          // K slab of this instantiation
  constexpr int TM = G::TM, TN = G::TN, RP = G::RP, ARP = G::ARP, BRP = G::BRP, ASUB = G::ASUB, BSUB = G::BSUB, BUFD = G::BUFD;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wr = wv / WN, wc = wv % WN;
  // slab loader: row lrow (+ RP per pass); each thread moves two 32-byte pieces per pass and operand, k = lk..lk+3 and
  // 16+lk..16+lk+3, into two 16-column sub-slabs of leading dimension 17 (conflict-free for the fragment reads below)
  const int lrow = tid >> 2, lk = (tid & 3) * 4;
  const int nks = (K + GKT - 1) / GKT;
  // v_mfma_f64_4x4x4_4b: four independent 4 x 4 x 4 products per instruction.  Operand lanes: A lane 16k + 4q + i holds A_q[i][k],
  // B lane 16k + 4q + j holds B_q[k][j]; result lane 16i + 4q + j holds D_q[i][j] (profiles/r1_mfma_f64_4x4x4_lane_layout.txt).
  // The four blocks q are four row groups of one 16-row A fragment and share one 4-column B fragment (replicated: the four
  // lanes read the same LDS word), so one instruction is a 16 x 4 x 4 product.  A wave's (16 FA) x 32 part = FA A fragments x
  // 8 B fragments = 8 FA instructions per 4 columns of K from FA + 8 LDS reads; LDS-fed this form issues 70-76 TFLOP/s where
  // v_mfma_f64_16x16x4 stops at 45-48 (profiles/r2_mfma_f64_4x4x4_lds_fed_core.txt).
  // Fragment cb covers the columns 16*(cb>>2) + 4*j + (cb&3): lane j then owns 4 adjacent columns of 4 fragments, i.e. 32
  // contiguous bytes of C per (fragment row, half), two 16-byte accesses like the operand loads.
  const int fk = lane >> 4, fq = (lane >> 2) & 3, fj = lane & 3;     // as operand lane: k, block, row/col in block; as result lane: row fk, block fq, col fj
  const int wr0 = wr * 16 * FA, wc0 = wc * 32;                        // this wave's corner inside the tile
  // the C fragment of a read-modify-write tile is prefetched at the tile's first slab in the small shape; the large shape (64
  // accumulator registers per lane already) reads it in the epilogue instead -- once per 10 slabs of 256 MFMAs
  constexpr bool CPRE = (FA <= 2) && (NS == 2);       // (the 16-column-slab shape keeps its registers for a fourth workgroup per CU)
  double acc[FA][8], cpre[CPRE ? FA : 1][8];
#pragma unroll
  for (int i = 0; i < FA; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc[i][j] = 0.0; if (CPRE) cpre[i][j] = 0.0; }
  double ra[ARP * 8], rb[BRP * 8];
  int m0 = 0, n0 = 0, ks = 0;            // current slab
  // Slab loads go through raw buffer resources: the address is an SGPR base + a per-thread 32-bit offset that never changes
  // + a scalar slab offset, and rows beyond M (N) fall outside the resource and read as zero -- no 64-bit address arithmetic, no
  // predicates, no zero-fill moves.  (On this part the fp64 MFMA and the vector ALU share the issue slots of a SIMD: every
  // VALU instruction of the loader is taken from the other wave's MFMA stream, profiles/r2_mfma_valu_mix_issue_rate.txt.)
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(((unsigned)(M - 1) * (unsigned)lda + (unsigned)K) * 8u), 0x00020000);
  const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(((unsigned)(N - 1) * (unsigned)ldb + (unsigned)K) * 8u), 0x00020000);
  const unsigned avo = (unsigned)(lrow * lda + lk) * 8u, bvo = (unsigned)(lrow * ldb + lk) * 8u;     // per-thread byte offsets (pass 0)
  typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
#define TMPC_BLD4(dst, off, rsrc, vo, so)                                                                   \
  {                                                                                                         \
    const double2_t u0_ = __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (vo), (so), 0));          \
    const double2_t u1_ = __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (vo) + 16u, (so), 0));    \
    dst[(off) + 0] = u0_[0]; dst[(off) + 1] = u0_[1]; dst[(off) + 2] = u1_[0]; dst[(off) + 3] = u1_[1];    \
  }
#define TMPC_SLAB_LOAD(M0, N0, KN)                                                                          \
  {                                                                                                         \
    const unsigned aso = (unsigned)((M0) * lda + (KN) - lk) * 8u, bso = (unsigned)((N0) * ldb + (KN) - lk) * 8u;   /* wave-uniform */ \
    _Pragma("unroll") for (int rp = 0; rp < ARP; ++rp) {                                                    \
      if (RP * ARP == TM || lrow + rp * RP < TM) {                                                          \
        _Pragma("unroll") for (int h2 = 0; h2 < NS; ++h2)                                                   \
          TMPC_BLD4(ra, rp * 8 + h2 * 4, arsrc, avo + (unsigned)(rp * RP * lda + h2 * 16) * 8u, aso)        \
      }                                                                                                     \
    }                                                                                                       \
    _Pragma("unroll") for (int rp = 0; rp < BRP; ++rp) {                                                    \
      if (RP * BRP == TN || lrow + rp * RP < TN) {                                                          \
        _Pragma("unroll") for (int h2 = 0; h2 < NS; ++h2)                                                   \
          TMPC_BLD4(rb, rp * 8 + h2 * 4, brsrc, bvo + (unsigned)(rp * RP * ldb + h2 * 16) * 8u, bso)        \
      }                                                                                                     \
    }                                                                                                       \
  }
#define TMPC_SLAB_STORE(BUF)                                                                                \
  {                                                                                                         \
    double* An_ = lds + (BUF) * BUFD;                                                                       \
    double* Bn_ = An_ + NS * ASUB;                                                                           \
    _Pragma("unroll") for (int rp = 0; rp < ARP; ++rp) {                                                    \
      const int row = lrow + rp * RP;                                                                       \
      if (RP * ARP == TM || row < TM) {                                                                     \
        _Pragma("unroll") for (int h2 = 0; h2 < NS; ++h2)                                                   \
          _Pragma("unroll") for (int q = 0; q < 4; ++q) An_[h2 * ASUB + row * SLD + lk + q] = ra[rp * 8 + h2 * 4 + q]; \
      }                                                                                                     \
    }                                                                                                       \
    _Pragma("unroll") for (int rp = 0; rp < BRP; ++rp) {                                                    \
      const int row = lrow + rp * RP;                                                                       \
      if (RP * BRP == TN || row < TN) {                                                                     \
        _Pragma("unroll") for (int h2 = 0; h2 < NS; ++h2)                                                   \
          _Pragma("unroll") for (int q = 0; q < 4; ++q) Bn_[h2 * BSUB + row * SLD + lk + q] = rb[rp * 8 + h2 * 4 + q]; \
      }                                                                                                     \
    }                                                                                                       \
  }
  // ---- prologue: first slab -> LDS buffer 0
  TMPC_SLAB_LOAD(0, 0, lk)
  __syncthreads();                         // LDS free (previous user)
  TMPC_SLAB_STORE(0)
  __syncthreads();
  int buf = 0;
  TMPC_T0()
  while (m0 < M) {
    // ---- next slab of the stream
    int nm0 = m0, nn0 = n0, nks_ = ks + 1;
    if (nks_ == nks) {
      nks_ = 0; nn0 = n0 + TN;
      if (nn0 >= N || (lower && nn0 > nm0)) { nn0 = 0; nm0 = m0 + TM; }
    }
    const bool more = nm0 < M;
    if (more) TMPC_SLAB_LOAD(nm0, nn0, nks_ * GKT + lk)
    TMPC_T(0)
    // does this wave own anything of the current tile?  (outside M x N, or above the diagonal of a lower-only diagonal tile)
    const bool wave_on = (m0 + wr0 < M) && (n0 + wc0 < N) && !(lower && n0 == m0 && wc0 >= wr0 + 16 * FA);
    if (CPRE && ks == 0 && mode == GM_SUB && wave_on) {       // prefetch the C fragment of this tile
#pragma unroll
      for (int i = 0; i < FA; ++i) {
        const int rbase = m0 + wr0 + i * 16;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int cbase = n0 + wc0 + m * 16;
          if (rbase < M && cbase < N) {
            gcptr2 cp2 = (gcptr2)(C + (size_t)(rbase + 4 * fq + fk) * ldc + cbase + 4 * fj);
            const double2_t u0 = cp2[0], u1 = cp2[1];
            cpre[CPRE ? i : 0][4 * m + 0] = u0[0]; cpre[CPRE ? i : 0][4 * m + 1] = u0[1]; cpre[CPRE ? i : 0][4 * m + 2] = u1[0]; cpre[CPRE ? i : 0][4 * m + 3] = u1[1];
          }
        }
      }
    }
    TMPC_T(1)
    // ---- compute current slab
    if (wave_on) {
      const double* As = lds + buf * BUFD;
      const double* Bs = As + NS * ASUB;
      const int krem = K - ks * GKT;
      if (USE_MFMA) {
#define TMPC_MFMA_STEP(kk)                                                                              \
  {                                                                                                     \
    const int so = ((kk) & 3) * 4 + fk;                                                                 \
    const double* Ak = As + ((kk) >> 2) * ASUB + so;                                                    \
    const double* Bk = Bs + ((kk) >> 2) * BSUB + so;                                                    \
    double av[FA];                                                                                      \
    _Pragma("unroll") for (int i = 0; i < FA; ++i) av[i] = Ak[(wr0 + 16 * i + 4 * fq + fj) * SLD];       \
    _Pragma("unroll") for (int cb = 0; cb < 8; ++cb) {                                                  \
      const double bv = Bk[(wc0 + 16 * (cb >> 2) + 4 * fj + (cb & 3)) * SLD];                           \
      _Pragma("unroll") for (int i = 0; i < FA; ++i)                                                    \
        acc[i][cb] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[i], bv, acc[i][cb], 0, 0, 0);                \
    }                                                                                                   \
  }
        // (the large shape is unrolled by two only: a full unroll makes the compiler fetch all eight steps' fragments up front and spill)
        if (NS == 1 || krem >= GKT) {
          if (FA <= 2) {
#pragma unroll
            for (int kk = 0; kk < GKT / 4; ++kk) TMPC_MFMA_STEP(kk)
          } else {
#pragma unroll 2
            for (int kk = 0; kk < GKT / 4; ++kk) TMPC_MFMA_STEP(kk)
          }
        } else {                 // half slab (K = 16 mod 32)
          if (FA <= 2) {
#pragma unroll
            for (int kk = 0; kk < GKT / 8; ++kk) TMPC_MFMA_STEP(kk)
          } else {
#pragma unroll 2
            for (int kk = 0; kk < GKT / 8; ++kk) TMPC_MFMA_STEP(kk)
          }
        }
#undef TMPC_MFMA_STEP
      } else {   // debug path: same fragment ownership, scalar FMAs
        const int kmax = (krem >= GKT) ? GKT : krem;
        for (int kk = 0; kk < kmax; ++kk) {
#pragma unroll
          for (int i = 0; i < FA; ++i)
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
              const int row = wr0 + i * 16 + 4 * fq + fk, col = wc0 + 16 * (cb >> 2) + 4 * fj + (cb & 3);
              acc[i][cb] = fma(As[(kk >> 4) * ASUB + row * SLD + (kk & 15)], Bs[(kk >> 4) * BSUB + col * SLD + (kk & 15)], acc[i][cb]);
            }
        }
      }
    }
    TMPC_T(2)
    // ---- stage the next slab into the other buffer
    if (more) TMPC_SLAB_STORE(buf ^ 1)
    TMPC_T(3)
    // ---- tile finished: store its C fragment (row 4*fq + fk of each 16-row fragment, columns 16m + 4*fj .. + 3)
    if (ks == nks - 1 && wave_on) {
#pragma unroll
      for (int i = 0; i < FA; ++i) {
        const int rbase = m0 + wr0 + i * 16;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int cbase = n0 + wc0 + m * 16;
          if (rbase < M && cbase < N) {
            typedef double2_t __attribute__((address_space(1)))* gptr2;
            gptr2 cp2 = (gptr2)(C + (size_t)(rbase + 4 * fq + fk) * ldc + cbase + 4 * fj);
            double c0[4] = {0.0, 0.0, 0.0, 0.0};
            if (mode == GM_SUB) {
              if (CPRE) { c0[0] = cpre[CPRE ? i : 0][4 * m]; c0[1] = cpre[CPRE ? i : 0][4 * m + 1]; c0[2] = cpre[CPRE ? i : 0][4 * m + 2]; c0[3] = cpre[CPRE ? i : 0][4 * m + 3]; }
              else { const double2_t u0 = ((gcptr2)cp2)[0], u1 = ((gcptr2)cp2)[1]; c0[0] = u0[0]; c0[1] = u0[1]; c0[2] = u1[0]; c0[3] = u1[1]; }
            }
            double v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const double t = acc[i][4 * m + e];
              v[e] = (mode == GM_SUB) ? c0[e] - t : ((mode == GM_NEG) ? -t : t);
            }
            cp2[0] = (double2_t){v[0], v[1]}; cp2[1] = (double2_t){v[2], v[3]};
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][4 * m + e] = 0.0;
        }
      }
    }
    TMPC_T(4)
    __syncthreads();
    TMPC_T(5)
    buf ^= 1;
    m0 = nm0; n0 = nn0; ks = nks_;
  }
#undef TMPC_SLAB_LOAD
#undef TMPC_SLAB_STORE
#undef TMPC_BLD4
}

// ---------------------------------------------------------------- 64 x 64 tile Cholesky + inverse, blocked by 16
// The tile is factored in four 16-column steps: one wave factors and inverts the 16 x 16 diagonal block in LDS (16 short
// column steps without workgroup barriers), the rows below are multiplied by that inverse and the trailing blocks updated with
// v_mfma_f64_16x16x4 on LDS operands (one 16 x 16 block per wave); the inverse of the whole tile is then assembled block row by
// block row from the diagonal inverses.  ~20 workgroup barriers per tile instead of the ~200 of a column-by-column sweep
// (the tile factorisations were 20 % of the factorisation phase, all of it barrier latency).
constexpr int LDP = TMPC_LDP;                            // leading dimension of the tile images in LDS (65; -DTMPC_LDP=66 / 68: the conflict experiments of round 5, profiles/r5_potrf_ldp.txt)
// (wave_lds_sync, mm16, load_d16, store_d16: tmpc_small.h)

// One wave: Cholesky (lower, in place) of the 16 x 16 block at S and its inverse into Si (same position; upper part zero).
// dr: the 16 pivot references; stat[0] counts frozen pivots, stat[1] tracks the smallest pivot / reference (Cholesky-with-shift,
// same rule as before: a pivot below 1e-15 of its reference is frozen at 1e20 * reference).
// The block lives in registers, row r in lane r (the other three lane groups mirror it): column step j takes the pivot and
// the column entries of the other rows by v_readlane (wave-uniform operands of the rank-1 update), no LDS round trip inside
// the 16 dependent steps; the inverse is then solved column by column (lane c owns column c, 16 registers) against the factor
// read back from LDS through wave-uniform addresses, all loops unrolled.  (The LDS version spent 39 k cycles per block in
// divergent dependent loops -- half of the whole k_cr_potrf kernel: profiles/r2v_cycle_prof_potrf.txt.)
// 1/sqrt(x), x > 0: v_rsq_f64 (~2^-26) + two Newton steps (full double precision)
__device__ __forceinline__ double potrf_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * (1.5 - 0.5 * x * r * r);
  r = r * (1.5 - 0.5 * x * r * r);
  return r;
}
// (wave_bcast: tmpc_small.h)
// value of lane J of every 16-lane row (DPP row_newbcast, one v_mov_b64_dpp): the source lane is J + 16 * (lane >> 4), which no v_readlane can express
template <int J>
__device__ __forceinline__ double row_bcast16(double v) { return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + J, 0xF, 0xF, false); }
// acc += (x of lane K of this 16-lane row) * y in ONE instruction: v_fmac_f64 with a DPP source (64-bit ALU DPP takes row_newbcast only).
// (s_nop 1: a VGPR written by the VALU needs two wait states before a DPP read; the hazard recogniser does not look inside inline asm.)
template <int K>
__device__ __forceinline__ void fmac_bcast16(double& acc, double x, double y) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(K));
}
// Round 3.  The serial 16 x 16 steps were 63 % of the tile Cholesky, which is all but ~30 us of a k_cr_potrf_dma workgroup
// (profiles/r3_cycle_prof_before.txt), and they are bound by the NUMBER of fp64 VALU instructions the one wave issues (every one of them behind the
// MFMAs of the workgroup it shares the SIMD with) at least as much as by the dependent chain: the first rewrite of this round only removed
// the dependent forward substitution of the inverse and changed nothing.  So:
//  * the inverse is carried along the elimination: M starts as I and every column step applies to M the row operations it applies to A
//    (rows i > j: M_i -= (l_ij / l_jj) M_j; row r is scaled by 1 / l_rr at the end), so M ends as L^-1 -- no second sweep, no LDS round trip;
//  * row r of A sits in the lanes r, r + 16, r + 32, r + 48 (mirrored, as before); lane (r, g) holds M[r][4q + g], q = 0..3, so the four lane
//    groups share the work on M;
//  * every rank-1 update is ONE instruction, v_fmac_f64 with a DPP row broadcast as its first source (the value of row k of the same lane
//    group), instead of two v_readlane and an FMA: ~480 instead of ~1100 VALU instructions per 16 x 16 block.
struct Potrf16State { double a[16]; double m[4]; double nbad, minr, myref, thr8; int r, g; };
template <int J, int K>
__device__ __forceinline__ void potrf16_arow(Potrf16State& st, double nlj, double lj) {        // a_rk -= l_rJ l_kJ, k = K .. 15
  if constexpr (K < 16) { fmac_bcast16<K>(st.a[K], nlj, lj); potrf16_arow<J, K + 1>(st, nlj, lj); }
}
template <int J, int Q>
__device__ __forceinline__ void potrf16_mrow(Potrf16State& st, double fr) {                    // M_rc -= (l_rJ / l_JJ) M_Jc, c = 4 q + g <= J (row J is still unscaled)
  if constexpr (Q <= J / 4) { fmac_bcast16<J>(st.m[Q], st.m[Q], fr); potrf16_mrow<J, Q + 1>(st, fr); }
}
// One column step.  The wave issues every instruction of it behind the MFMAs of the workgroup it shares the SIMD with, so the step is
// written for instruction count (the compiler's version of the straightforward code was 58 VALU instructions per step, this one is ~27):
//  * one comparison against a pre-scaled threshold (1e-8 x reference) guards a wave-uniform slow path that holds everything rare: the
//    smallest-pivot statistics and the frozen-pivot rule (pivot <= 1e-15 x reference or <= 0: frozen at 1e20 x reference, as in round 1);
//  * 1/sqrt = v_rsq_f64 (~2^-26) + ONE third-order step r (1 + h/2 + 3 h^2/8), h = 1 - x r^2 (error ~ h^3: full double precision);
//  * no select for the diagonal entry (a_JJ / sqrt(a_JJ) IS sqrt(a_JJ)), none for 1 / l_rr (read back once at the end).
template <int J>
__device__ __forceinline__ void potrf16_step(Potrf16State& st) {
  const double piv = row_bcast16<J>(st.a[J]);
  const double th8 = row_bcast16<J>(st.thr8);
  double rinv;
  {
    const double r0 = __builtin_amdgcn_rsq(piv);
    const double t = piv * r0, h = fma(-t, r0, 1.0);
    const double c = fma(0.375, h, 0.5), hc = h * c;
    rinv = fma(r0, hc, r0);
  }
  double lj = st.a[J] * rinv;                                             // l_rJ (rows above the diagonal carry unused values); row J: sqrt(pivot)
  if (__builtin_amdgcn_ballot_w64(!(piv > th8)) != 0) {                   // wave-uniform (the four lane groups mirror the block), rare
    const double ref = row_bcast16<J>(st.myref);
    if (piv < 1e-8 * ref) st.minr = fmin(st.minr, fmax(piv, 0.0) / ref);
    if (!(piv > 1e-15 * ref) || !(piv > 0.0)) {
      const double sub = (ref > 0.0 ? ref : 1.0) * 1e20;
      st.nbad += 1.0;
      rinv = potrf_rsqrt(sub);
      lj = (st.r == J) ? sub * rinv : st.a[J] * rinv;
    }
  }
  st.a[J] = lj;
  const double nlj = -lj;
  const double fr = (st.r > J) ? nlj * rinv : 0.0;
  potrf16_arow<J, J + 1>(st, nlj, lj);
  potrf16_mrow<J, 0>(st, fr);
  if constexpr (J + 1 < 16) potrf16_step<J + 1>(st);
}
template <int LDQ = LDP>                                     // leading dimension of S and Si (tmpc_cr_small.h packs them tighter)
__device__ __forceinline__ void wave_potrf16(double* S, double* Si, const double* dr, double* stat, int lane) {
  Potrf16State st;
  st.r = lane & 15; st.g = lane >> 4;
#pragma unroll
  for (int k = 0; k < 16; ++k) st.a[k] = S[st.r * LDQ + k];
#pragma unroll
  for (int q = 0; q < 4; ++q) st.m[q] = (4 * q + st.g == st.r) ? 1.0 : 0.0;
  st.myref = fabs(dr[st.r]); st.thr8 = 1e-8 * st.myref;
  st.nbad = 0.0; st.minr = 1.0;
  __builtin_amdgcn_s_setprio(3);          // the one wave every other wave of the workgroup waits for: ahead of the co-resident workgroup's waves at the issue port
  potrf16_step<0>(st);
  __builtin_amdgcn_s_setprio(0);
  if (lane < 16) {
#pragma unroll
    for (int k = 0; k < 16; ++k) if (k <= st.r) S[st.r * LDQ + k] = st.a[k];
  }
  wave_lds_sync();
  double myrinv;                                                           // 1 / l_rr: v_rcp_f64 + two Newton steps
  {
    const double d = S[st.r * LDQ + st.r];
    double x = __builtin_amdgcn_rcp(d);
    x = fma(fma(-d, x, 1.0), x, x);
    x = fma(fma(-d, x, 1.0), x, x);
    myrinv = x;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) { const int c = 4 * q + st.g; Si[st.r * LDQ + c] = (c <= st.r) ? st.m[q] * myrinv : 0.0; }      // row r of L^-1 = row r of M / l_rr
  if (lane == 0) { stat[0] += st.nbad; stat[1] = fmin(stat[1], st.minr); }
  wave_lds_sync();
}

// Cholesky of the nb x nb diagonal tile at T (ld = ldt; nb a multiple of 16, <= 64) + its inverse into Ti (nb x nb, ld = TB).
// dref: assembled diagonal entries (pivot reference).  Returns the number of frozen pivots (thread-uniform).
// stat (optional, LDS): [0] += frozen pivots, [1] = min(itself, smallest pivot ratio) instead of *minr
// preloaded: the tile already sits in the LDS image S (k_cr_potrf_dma: the sweep of its block row leaves it there instead of sending it through memory)
__device__ __noinline__ int wg_potrf_inv(double* T, int ldt, double* Ti, const double* dref, int nb, double* lds, double* minr, double* stat = nullptr, int sw = 0, bool preloaded = false) {
  // sw: the wave that runs the serial 16 x 16 steps
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  TMPC_TC0()
  double* S = lds;                         // 64 x 65 (aliases the GEMM slabs, never live at the same time)
  double* Si = S + 64 * LDP;               // 64 x 65
  double* dr = Si + 64 * LDP;              // 64 pivot references, [65] = shift counter, [66] = smallest pivot / reference
  double* tmp = dr + 72 + wv * (16 * 17);  // one 16 x 17 scratch block per wave
  {                                        // thread -> row tid >> 2, 16 consecutive columns: eight 16-byte loads in flight, then the LDS stores
    const int i = tid >> 2, j0 = (tid & 3) * 16;
    double2_t v[8];
    const bool on = i < nb && j0 < nb;
    if (on && !preloaded) {
      gcptr2 src = (gcptr2)(T + (size_t)i * ldt + j0);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = src[q];
#pragma unroll
      for (int q = 0; q < 8; ++q) { S[i * LDP + j0 + 2 * q] = v[q][0]; S[i * LDP + j0 + 2 * q + 1] = v[q][1]; }
    }
    if (on) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { Si[i * LDP + j0 + 2 * q] = 0.0; Si[i * LDP + j0 + 2 * q + 1] = 0.0; }
    }
  }
  if (tid < nb) dr[tid] = dref[tid];
  if (tid == 0) { dr[65] = 0.0; dr[66] = 1.0; }
  __syncthreads();
  const int nbk = nb >> 4;
  TMPC_TC(7, 0)
  for (int jb = 0; jb < nbk; ++jb) {
    const int o = 16 * jb;
    if (wv == sw) wave_potrf16(S + o * LDP + o, Si + o * LDP + o, dr + o, dr + 65, lane);
    __syncthreads();
    TMPC_TC(7, 1)
    {                                      // rows below: P <- P L11^-T, one 16-row block per wave
      const int bi = jb + 1 + wv;
      if (bi < nbk) {
        double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
        acc = mm16(S + 16 * bi * LDP + o, LDP, Si + o * LDP + o, 1, LDP, acc, 1.0, lane);     // B[k][c] = Linv11[c][k]
        store_d16(S + 16 * bi * LDP + o, LDP, acc, lane);
      }
    }
    __syncthreads();
    {                                      // trailing blocks (bi, bj), jb < bj <= bi:  S -= P_bi P_bj'
      int q = 0;
      for (int bi = jb + 1; bi < nbk; ++bi)
        for (int bj = jb + 1; bj <= bi; ++bj, ++q)
          if ((q & 3) == wv) {
            double* Cp = S + 16 * bi * LDP + 16 * bj;
            double4_t acc = load_d16(Cp, LDP, lane);
            acc = mm16(S + 16 * bi * LDP + o, LDP, S + 16 * bj * LDP + o, 1, LDP, acc, -1.0, lane);     // B[k][c] = P_bj[c][k]
            store_d16(Cp, LDP, acc, lane);
          }
    }
    __syncthreads();
    TMPC_TC(7, 2)
  }
  // inverse of the tile, block row by block row:  Linv[bi][bj] = -Linv[bi][bi] * sum_{k=bj}^{bi-1} L[bi][k] Linv[k][bj]
  for (int bi = 1; bi < nbk; ++bi) {
    const int bj = wv;
    if (bj < bi) {
      double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
      for (int k = bj; k < bi; ++k) acc = mm16(S + 16 * bi * LDP + 16 * k, LDP, Si + 16 * k * LDP + 16 * bj, LDP, 1, acc, 1.0, lane);
      store_d16(tmp, 17, acc, lane);
      wave_lds_sync();
      double4_t r = (double4_t){0.0, 0.0, 0.0, 0.0};
      r = mm16(Si + 16 * bi * LDP + 16 * bi, LDP, tmp, 17, 1, r, -1.0, lane);
      store_d16(Si + 16 * bi * LDP + 16 * bj, LDP, r, lane);
    }
    __syncthreads();
  }
  TMPC_TC(7, 3)
  {
    const int i = tid >> 2, j0 = (tid & 3) * 16;
    if (i < nb && j0 < nb) {
      typedef double2_t __attribute__((address_space(1)))* gptr2;
      gptr2 dt = (gptr2)(T + (size_t)i * ldt + j0);
      gptr2 di = (gptr2)(Ti + i * TB + j0);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = j0 + 2 * q;
        di[q] = (double2_t){Si[i * LDP + j], Si[i * LDP + j + 1]};
        if (j + 1 <= i) dt[q] = (double2_t){S[i * LDP + j], S[i * LDP + j + 1]};
        else if (j <= i) T[(size_t)i * ldt + j] = S[i * LDP + j];
      }
    }
  }
  __syncthreads();
  TMPC_TC(7, 4)
  const int nb_bad = (int)dr[65];
  if (stat) { if (tid == 0) { stat[0] += dr[65]; stat[1] = fmin(stat[1], dr[66]); } }
  else *minr = fmin(*minr, dr[66]);
  __syncthreads();
  return nb_bad;
}

// Left-looking blocked Cholesky of the block column headed by Dk, applied also to the rows of R1 (and R2):
//   Dk = L L' ;  R1 <- R1 L^-T ;  R2 <- R2 L^-T        (R1/R2 may be null)
template <bool USE_MFMA>
__device__ __forceinline__ int wg_block_column(double* Dk, double* R1, double* R2, double* Linv_k, const double* dref,
                                               int dp, double* lds, double* minr) {
  int nbad = 0;
  int jt = 0;
  TMPC_TC0()
  for (int j0 = 0; j0 < dp; j0 += TB, ++jt) {
    const int nb = (dp - j0 < TB) ? dp - j0 : TB;
    if (j0 > 0) {
      wg_gemm_nt<USE_MFMA>(Dk + (size_t)j0 * dp + j0, dp, Dk + (size_t)j0 * dp, dp, Dk + (size_t)j0 * dp, dp, dp - j0, nb, j0, GM_SUB, false, lds);
      if (R1) wg_gemm_nt<USE_MFMA>(R1 + j0, dp, R1, dp, Dk + (size_t)j0 * dp, dp, dp, nb, j0, GM_SUB, false, lds);
      if (R2) wg_gemm_nt<USE_MFMA>(R2 + j0, dp, R2, dp, Dk + (size_t)j0 * dp, dp, dp, nb, j0, GM_SUB, false, lds);
    }
    TMPC_TC(6, 0)
    double* Ti = Linv_k + (size_t)jt * TB * TB;
    nbad += wg_potrf_inv(Dk + (size_t)j0 * dp + j0, dp, Ti, dref + j0, nb, lds, minr);
    TMPC_TC(6, 1)
    // panel below / beside the diagonal tile:  X <- X * Ti'   (in place, K = nb)
    if (dp - j0 - nb > 0)
      wg_gemm_nt<USE_MFMA>(Dk + (size_t)(j0 + nb) * dp + j0, dp, Dk + (size_t)(j0 + nb) * dp + j0, dp, Ti, TB, dp - j0 - nb, nb, nb, GM_SET, false, lds);
    if (R1) wg_gemm_nt<USE_MFMA>(R1 + j0, dp, R1 + j0, dp, Ti, TB, dp, nb, nb, GM_SET, false, lds);
    if (R2) wg_gemm_nt<USE_MFMA>(R2 + j0, dp, R2 + j0, dp, Ti, TB, dp, nb, nb, GM_SET, false, lds);
    TMPC_TC(6, 2)
  }
  return nbad;
}

// ------------------------------------------------------------------ triangular solves with the block factor
// Right-hand sides: [p][dp][NC] in global memory (NC interleaved), solved in place.  Inside the kernel the
// active vectors live in LDS as [NCP][xld] (row q = right-hand side q).  All matrix-vector work runs through one
// primitive, a skinny GEMM on the fp64 matrix cores:  Y[q][i] (+)= sgn * sum_c Mop[i][c] X[q][c], Mop = M or M',
// with the 64 x 16 slabs of M staged through LDS from coalesced global loads (the factors are streamed from
// HBM exactly once per product).
constexpr int NCP = 4;      // rows of the LDS vectors (>= max NC = 3)
constexpr int GKV = 16, GLDV = GKV + 1;   // K slab of the skinny GEMM

// LOWER: M is a lower-triangular tile (an inverted diagonal tile): the 4-element pieces above the diagonal are zeros in memory and are
// not fetched (3/8 of the tile: the substitutions are bound by the bytes they stream)
// MT = float: M is a float32 copy of the matrix (the O blocks of an iteration whose updates run in single precision, tmpc_cr.h: half the bytes of a sweep that is
// bound by them); its entries are widened on the way into LDS, products and sums stay fp64.
template <bool TRANS, bool LOWER = false, typename MT = double>
__device__ __forceinline__ void wg_gemv16(double* Y, int yld, const double* X, int xld, const MT* M, int ldm,
                                          int rows, int cols, bool accumulate, double sgn, double* As, int nc) {
  constexpr bool F32 = sizeof(MT) == 4;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  const int nks = cols / GKV;
  for (int m0 = 0; m0 < rows; m0 += 64) {
    double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
    double ra[4];
    // slab loader
    const int lrow = TRANS ? (tid >> 4) : (tid >> 2);            // TRANS: k index (0..15) ; else row (0..63)
    const int lcol = TRANS ? ((tid & 15) * 4) : ((tid & 3) * 4); // TRANS: first i ; else first k
    const bool ok = TRANS ? (m0 + lcol < rows) : (m0 + lrow < rows);
    const MT* src = TRANS ? (M + (size_t)lrow * ldm + m0 + lcol) : (M + (size_t)(m0 + lrow) * ldm + lcol);
    const size_t kstep = TRANS ? (size_t)GKV * ldm : (size_t)GKV;
    // four slabs of loads in flight per thread (ra..rd, slab ks mod 4): the solve streams the factors once and is bound by
    // bytes in flight (512 workgroups x 8 KB with a single prefetch: 4.4 TB/s; two in flight: 5.2)
    double rb[4], rc[4], rd[4];
#define TMPC_GEMV_OK(KS) (ok && (!LOWER || (TRANS ? (m0 + lcol <= GKV * (KS) + lrow) : (GKV * (KS) + lcol <= m0 + lrow))))
#define TMPC_GEMV_LD(RG, P, OK) { if (F32) TMPC_LD4F(RG, 0, P, OK) else TMPC_LD4(RG, 0, P, OK) }
    TMPC_GEMV_LD(ra, src, TMPC_GEMV_OK(0))
    if (nks > 1) { const MT* s1 = src + kstep; TMPC_GEMV_LD(rb, s1, TMPC_GEMV_OK(1)) }
    if (nks > 2) { const MT* s1 = src + 2 * kstep; TMPC_GEMV_LD(rc, s1, TMPC_GEMV_OK(2)) }
    if (nks > 3) { const MT* s1 = src + 3 * kstep; TMPC_GEMV_LD(rd, s1, TMPC_GEMV_OK(3)) }
#define TMPC_GEMV_SLAB(RG, KS)                                                                                  \
    {                                                                                                             \
      __syncthreads();                                                                                            \
      if (TRANS) {                                                                                                \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) As[(lcol + q) * GLDV + lrow] = RG[q];                     \
      } else {                                                                                                    \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) As[lrow * GLDV + lcol + q] = RG[q];                       \
      }                                                                                                           \
      __syncthreads();                                                                                            \
      if ((KS) + 4 < nks) {                                                                                       \
        const MT* s2 = src + (size_t)((KS) + 4) * kstep;                                                          \
        TMPC_GEMV_LD(RG, s2, TMPC_GEMV_OK((KS) + 4))                                                             \
      }                                                                                                           \
      if (m0 + 16 * wv < rows) {                                                                                  \
        const int k0 = (KS) * GKV;                                                                                \
        _Pragma("unroll") for (int kk = 0; kk < GKV / 4; ++kk) {                                                 \
          const double a = As[(16 * wv + fr) * GLDV + kk * 4 + fk];                                               \
          const double bq = X[(fr & (NCP - 1)) * xld + k0 + kk * 4 + fk];                                         \
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq, acc, 0, 0, 0);                                        \
        }                                                                                                         \
      }                                                                                                           \
    }
    for (int ks = 0; ks < nks; ks += 4) {
      TMPC_GEMV_SLAB(ra, ks)
      if (ks + 1 < nks) TMPC_GEMV_SLAB(rb, ks + 1)
      if (ks + 2 < nks) TMPC_GEMV_SLAB(rc, ks + 2)
      if (ks + 3 < nks) TMPC_GEMV_SLAB(rd, ks + 3)
    }
#undef TMPC_GEMV_SLAB
#undef TMPC_GEMV_LD
#undef TMPC_GEMV_OK
    if (m0 + 16 * wv < rows && fr < nc) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = m0 + 16 * wv + fk + 4 * r;
        if (i < rows) {
          double* yp = Y + fr * yld + i;
          *yp = (accumulate ? *yp : 0.0) + sgn * acc[r];
        }
      }
    }
  }
  __syncthreads();
}

// z <- L_k^-1 z   (z: LDS [NCP][xld]), using the inverted diagonal tiles
__device__ __forceinline__ void blk_fwd(double* z, int xld, double* tmp, int tld, const double* Dk, const double* Lik, int dp,
                                        double* As, int nc) {
  int jt = 0;
  for (int j0 = 0; j0 < dp; j0 += TB, ++jt) {
    const int nb = (dp - j0 < TB) ? dp - j0 : TB;
    const double* Ti = Lik + (size_t)jt * TB * TB;
    wg_gemv16<false, true>(tmp, tld, z + j0, xld, Ti, TB, nb, nb, false, 1.0, As, nc);
    for (int e = threadIdx.x; e < nb * nc; e += 256) { const int q = e / nb, i = e - q * nb; z[q * xld + j0 + i] = tmp[q * tld + i]; }
    __syncthreads();
    const int rem = dp - j0 - nb;
    if (rem > 0) wg_gemv16<false>(z + j0 + nb, xld, z + j0, xld, Dk + (size_t)(j0 + nb) * dp + j0, dp, rem, nb, true, -1.0, As, nc);
  }
}
// z <- L_k^-T z
__device__ __forceinline__ void blk_bwd(double* z, int xld, double* tmp, int tld, const double* Dk, const double* Lik, int dp,
                                        double* As, int nc) {
  const int nt = (dp + TB - 1) / TB;
  for (int jt = nt - 1; jt >= 0; --jt) {
    const int j0 = jt * TB;
    const int nb = (dp - j0 < TB) ? dp - j0 : TB;
    const double* Ti = Lik + (size_t)jt * TB * TB;
    wg_gemv16<true, true>(tmp, tld, z + j0, xld, Ti, TB, nb, nb, false, 1.0, As, nc);
    for (int e = threadIdx.x; e < nb * nc; e += 256) { const int q = e / nb, i = e - q * nb; z[q * xld + j0 + i] = tmp[q * tld + i]; }
    __syncthreads();
    if (j0 > 0) wg_gemv16<true>(z, xld, z + j0, xld, Dk + (size_t)j0 * dp, dp, j0, nb, true, -1.0, As, nc);
  }
}

__device__ __forceinline__ void vec_g2s(double* z, int xld, const double* R, int dp, int nc) {
  for (int e = threadIdx.x; e < dp * nc; e += 256) { const int i = e / nc, q = e - i * nc; z[q * xld + i] = R[e]; }
}
__device__ __forceinline__ void vec_s2g(double* R, const double* z, int xld, int dp, int nc) {
  for (int e = threadIdx.x; e < dp * nc; e += 256) { const int i = e / nc, q = e - i * nc; R[e] = z[q * xld + i]; }
}

template <int NTH = 256>
__device__ __forceinline__ double wg_reduce_sum(double v, double* red) {      // red: NTH / 64 doubles of LDS
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if (lane == 0) red[wv] = v;
  __syncthreads();
  double r = red[0];
#pragma unroll
  for (int q = 1; q < NTH / 64; ++q) r += red[q];
  __syncthreads();
  return r;
}

// After the block solves (tmpc_cr.h) of pass 1 ([rhs | u_tau | u_alpha]) or pass 2 (rhs; all three while centering): the 2 x 2
// border system of (tau, alpha) and dP.  One workgroup per active problem.
// (body: the NTH threads of one workgroup; red: NTH / 64 doubles of LDS.  The sums are taken in a different order for NTH != 256.)
template <int NTH>
__device__ __forceinline__ void solve_border_body(const WS& w, const Dims& dm, int b, int pass, double* red) {
  const int* ip = w.iprob + (size_t)b * IS;
  const int phase = ip[I_PHASE];
  if (phase == PH_DONE) return;
  double* pr = w.prob + (size_t)b * PS;
  const int p = dm.p, dp = dm.dp, tid = threadIdx.x, nx = dm.nx;
  const size_t vl = (size_t)p * dp;
  double* W3 = w.W3 + (size_t)b * vl * 3;
  double* Z = w.Z + (size_t)b * vl;
  double* TU = w.TU + (size_t)b * vl * 2;
  const double* U = w.U + (size_t)b * vl * 2;
  if (pass == 1 && phase != PH_MAIN) return;           // centering: everything happens in pass 2
  const bool three = (pass == 1) || (phase != PH_MAIN && !ip[I_CHORD]);    // chord step: TU and the 2 x 2 border Schur complement of the last factorisation stay
  if (three) {
    double s00 = 0.0, s01 = 0.0, s11 = 0.0;
    for (size_t e = tid; e < vl; e += NTH) {
      const double t0 = W3[e * 3 + 1], t1 = W3[e * 3 + 2];
      TU[e * 2] = t0; TU[e * 2 + 1] = t1;
      Z[e] = W3[e * 3];
      s00 = fma(U[e * 2], t0, s00); s01 = fma(U[e * 2], t1, s01); s11 = fma(U[e * 2 + 1], t1, s11);
    }
    s00 = wg_reduce_sum<NTH>(s00, red); s01 = wg_reduce_sum<NTH>(s01, red); s11 = wg_reduce_sum<NTH>(s11, red);
    if (tid == 0) { pr[P_SB00] = pr[P_BTT] - s00; pr[P_SB01] = pr[P_BTA] - s01; pr[P_SB11] = pr[P_BAA] - s11; }
  }
  __syncthreads();
  // border:  rb = [rhs_tau, rhs_alpha] - U' z ;  db = Sb^-1 rb ;  dp = z - TU db
  double u0 = 0.0, u1 = 0.0;
  for (size_t e = tid; e < vl; e += NTH) { const double z = Z[e]; u0 = fma(U[e * 2], z, u0); u1 = fma(U[e * 2 + 1], z, u1); }
  u0 = wg_reduce_sum<NTH>(u0, red); u1 = wg_reduce_sum<NTH>(u1, red);
  double trt2 = 0.0, hbg = 0.0;
  for (int k = tid; k < p; k += NTH) { const double* q = w.part + (size_t)(b * p + k) * NPART; trt2 += q[Q_TRT2]; hbg += q[Q_HBG]; }
  trt2 = wg_reduce_sum<NTH>(trt2, red); hbg = wg_reduce_sum<NTH>(hbg, red);
  const double sig = (pass == 1) ? 0.0 : pr[P_SIGMU];
  const double corr0 = (pass == 1) ? 0.0 : pr[P_CORR0];
  const double s0 = pr[P_S0], x0 = pr[P_X0], rd0 = pr[P_RD0];
  const double t0 = sig / s0 - x0 * rd0 / s0 - corr0;
  const double rb0 = (trt2 - 1.0) - u0, rb1 = (hbg + t0) - u1;
  const double a = pr[P_SB00], bb = pr[P_SB01], c = pr[P_SB11];
  const double det = a * c - bb * bb;
  const double dtau = (c * rb0 - bb * rb1) / det, dalpha = (a * rb1 - bb * rb0) / det;
  __syncthreads();
  if (tid == 0) { pr[P_DTAU] = dtau; pr[P_DALPHA] = dalpha; }
  // dP_k = smat(z - TU db)
  double* dPg = w.dP + (size_t)b * p * nx * nx;
  const int d = dm.d;
  for (int e = tid; e < p * d; e += NTH) {
    const int k = e / d, idx = e - k * d;
    // idx -> (a,c), a <= c  (row-major upper triangle)
    int a2 = 0, rem = idx;
    while (rem >= nx - a2) { rem -= nx - a2; ++a2; }
    const int c2 = a2 + rem;
    const size_t vi = (size_t)k * dp + idx;
    const double v = Z[vi] - TU[vi * 2] * dtau - TU[vi * 2 + 1] * dalpha;
    dPg[(size_t)k * nx * nx + a2 * nx + c2] = v;
    dPg[(size_t)k * nx * nx + c2 * nx + a2] = v;
  }
}

__global__ void __launch_bounds__(256) k_solve_border(WS w, Dims dm, const int* alist, int pass) {
  __shared__ double red[8];
  solve_border_body<256>(w, dm, alist[blockIdx.x], pass, red);
}

}  // namespace tmpc
