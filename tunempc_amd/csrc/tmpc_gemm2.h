// Register-tile GEMM core of the block factorisation (second generation).
//
// The first-generation core (tmpc_factor.h: 64 x 64 workgroup tiles, two workgroups per CU) moves 8 flop per operand
// byte; with 512 problems in flight nothing survives in the 4 MB L2 of an XCD, so every operand panel comes from
// HBM and the kernel is bandwidth-bound at ~36 % of the fp64 matrix peak.  This core trades occupancy for reuse:
//   * ONE workgroup (4 wavefronts) per CU; every wave owns 5 x 5 MFMA fragments (16 x 16) = 100 accumulator
//     doubles per lane, so the workgroup tile is 160 x 160 (wave grid 2 x 2) or 320 x 80 (4 x 1, for the 64-wide
//     panel products) -- 20 resp. 16 flop per operand byte.
//   * fragments are dealt to the waves cyclically (wave (wr, wc) owns fragment rows wr, wr + WR, ... and columns
//     wc, wc + WC, ...), so ragged edges (d = 300 -> 19 fragments) and triangular outputs/operands are skipped at
//     fragment granularity with balanced work per wave.
//   * K slabs of 16 columns go global -> LDS by LDS-DMA (global_load_lds_dwordx4), double-buffered, no staging
//     registers.  The LDS image is row-major [row][16] with the 16-byte pair p of row r stored at slot
//     p ^ ((r >> 1) & 7); the DMA writes lane-linear, so the permutation is applied to the per-lane SOURCE address
//     and again on the operand reads (conflict-free ds_reads, full 128-byte lines per 8 lanes on the global side).
//   * the matrix instruction is v_mfma_f64_4x4x4_4b_f64 (four independent 4 x 4 x 4 blocks), NOT the 16 x 16 x 4 form:
//     on MI355X the 16 x 16 x 4 instruction tops out at 36 (one wave per SIMD) to 48 TFLOP/s (two or more), the
//     4 x 4 x 4 form reaches 71-75 TFLOP/s of the 78.6 peak already with one wave per SIMD (scripts/micro/mfma_peak.hip).
//     A 16 x 16 result fragment is four such instructions per 4 k (one per group of four columns) sharing the A register.
#pragma once
#include "tmpc_common.h"

namespace tmpc {

// TRI_CLOW: lower triangle of a square C only (tiles above the diagonal skipped, fragments above it not stored);
// TRI_BLOW: B lower-triangular (B[n][k] = 0 for k > n: trailing k-slabs of a tile column skipped);
// TRI_CDIAG / TRI_COFF: the two halves of TRI_CLOW for square tiles -- diagonal tiles only, with the compile-time
// fragment staircase s <= q (60 % of the MFMAs), resp. the tiles strictly below the diagonal (full).
enum { TRI_NONE = 0, TRI_CLOW = 1, TRI_BLOW = 2, TRI_CDIAG = 3, TRI_COFF = 4 };
constexpr int G2_LDS_DOUBLES = 3 * 384 * 16;  // three buffers of at most (TM + TN) = 384 rows x 16 doubles: 144 KB

typedef double g2_d4 __attribute__((ext_vector_type(4)));
typedef double g2_d2 __attribute__((ext_vector_type(2)));
typedef unsigned int g2_u4 __attribute__((ext_vector_type(4)));
typedef const g2_d2 __attribute__((address_space(1)))* g2_cptr2;
typedef g2_d2 __attribute__((address_space(1)))* g2_ptr2;

// C (M x N, ldc) <op> A (M x K, lda) * B (N x K, ldb)'  -- M, N, K multiples of 16, K >= 16.
// MODE: 0  C -= A B',  1  C = A B',  2  C = -A B'.   TRI: TRI_*.
// In-place use (C aliasing A, one N tile, K == N) is safe: all A slabs of a tile are in LDS before its epilogue.
template <int WR, int WC, int FR, int FC, int MODE, int TRI, int DBG = 0>   // DBG (timing experiments only): 1 no DMA in the loop, 2 no LDS operand fetch in the loop
__device__ __forceinline__ void wg_gemm2(double* C, int ldc, const double* A, int lda, const double* B, int ldb,
                                         int M, int N, int K, double* lds) {
  constexpr int TM = WR * FR * 16, TN = WC * FC * 16, GA = TM / 8, NG = (TM + TN) / 8, NGW = (NG + 3) / 4;
  constexpr int BUF = (TM + TN) * 16;                       // doubles per LDS buffer
  static_assert(3 * BUF <= G2_LDS_DOUBLES, "LDS budget");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wv / WC, wc = wv - wr * WC;
  const int fr = lane & 15, fk = lane >> 4;
  const int nks = K >> 4, mf = M >> 4, nf = N >> 4;
  // v_mfma_f64_4x4x4_4b: A lane = 16 k + 4 blk + i, B lane = 16 k + 4 blk + j, D lane = 16 i + 4 blk + j (probed,
  // scripts/micro/mfma4_probe.hip).  Block blk takes rows 4 blk .. 4 blk + 3 of a 16-row A fragment (so the A register
  // is row = lane & 15, k = lane >> 4) against four B rows shared by all blocks; B register c of a 16-column fragment
  // holds columns {c, 4 + c, 8 + c, 12 + c} so that a lane ends up with four CONTIGUOUS columns 4 (lane & 3) .. + 3 of
  // row 4 ((lane >> 2) & 3) + (lane >> 4).
  //
  // LDS image of a slab: row-major [row][16 doubles] (A rows, then B rows), the 16-byte pair p of row r stored at
  // slot p ^ ((r >> 1) & 7).  The DMA writes lane-linear -- 8 adjacent lanes = the 128 contiguous bytes of one row, so
  // every global request is a full line -- and the permutation is applied to the per-lane SOURCE address and again
  // on the operand reads: the 16 rows x 4 k of an A register and the 4 rows x 4 k of a B register hit distinct banks.
  const int bj = lane & 3;
  int offa[4], offb[4][2];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int pr = 2 * kk + (fk >> 1);
    offa[kk] = (wr * 16 + fr) * 16 + (((pr ^ (fr >> 1)) & 7) << 1) + (fk & 1);                 // + q * WR * 256
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)   // ch = c >> 1: B row 4 bj + c, swizzle term (row >> 1) & 7 = (2 bj + ch) & 7
      offb[kk][ch] = TM * 16 + (wc * 16 + 4 * bj + 2 * ch) * 16 + (((pr ^ (2 * bj + ch)) & 7) << 1) + (fk & 1);   // + s * WC * 256 + (c & 1) * 16
  }
  // LDS-DMA source offset (bytes) of this lane inside an 8-row group: row lane >> 3, pair (lane & 7) ^ swizzle.  The
  // swizzle term alternates with the group parity, and a wave only ever handles groups of one parity (g = wv + 4 t,
  // the A region has an even number of groups).
  static_assert(GA % 2 == 0, "group parity must be a wave constant");
  const int gpp = (lane & 7) ^ ((4 * (wv & 1) + (lane >> 4)) & 7);
  const unsigned loa = (unsigned)((lane >> 3) * lda + 2 * gpp) * 8u, lob = (unsigned)((lane >> 3) * ldb + 2 * gpp) * 8u;
  const unsigned dlo = (unsigned)((4 * ((lane >> 2) & 3) + (lane >> 4)) * ldc + 4 * bj) * 8u;   // byte offset of this lane inside a result fragment: row 4 blk + i, columns 4 j .. 4 j + 3
  const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)((unsigned)M * (unsigned)ldc * 8u), 0x00020000);

  g2_d4 acc[FR][FC];

  // slabs of tile column n0 (TRI_BLOW: rows n of B are zero beyond k = n)
#define G2_NKS(n0_) ((TRI == TRI_BLOW) ? min(nks, (min((n0_) + TN, N)) >> 4) : nks)
  // LDS-DMA piece t (0 .. NGW-1) of this wave for slab (m_, n_, k_) into buffer b_: 8-row group g = wv + 4 t of the
  // A rows then the B rows.  Every wave issues exactly NGW pieces per slab (groups past the matrix edge re-read the last
  // valid group into LDS rows nobody stores from), which keeps the counted s_waitcnt vmcnt(NGW) of the pipeline exact.
  static_assert(NG % 4 == 0, "equal number of DMA pieces per wave");
#define G2_PIECE(t, m_, n_, k_, b_)                                                                                \
  {                                                                                                               \
    const int g = wv + 4 * (t);                                                                                   \
    const bool isa = g < GA;                                                                                      \
    const int row = isa ? min((m_) + g * 8, M - 8) : min((n_) + (g - GA) * 8, N - 8);                             \
    const char* ub = (const char*)((isa ? A : B) + (size_t)row * (isa ? lda : ldb) + (k_) * 16);                  \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub + (isa ? loa : lob)),     \
                                     (__attribute__((address_space(3))) void*)(lds + (b_) * BUF + g * 128), 16, 0, 0); \
  }
#define G2_ISSUE(m_, n_, k_, b_) { _Pragma("unroll") for (int t = 0; t < NGW; ++t) G2_PIECE(t, m_, n_, k_, b_) }
  // fragment activity of this wave in tile (m_, n_)
#define G2_TILE_MASK(m_, n_)                                                                                       \
  {                                                                                                               \
    unsigned t_ = 0;                                                                                              \
    _Pragma("unroll") for (int q = 0; q < FR; ++q) _Pragma("unroll") for (int s = 0; s < FC; ++s) {               \
      const int fi = ((m_) >> 4) + wr + q * WR, fj = ((n_) >> 4) + wc + s * WC;                                   \
      bool on = (fi < mf) && (fj < nf);                                                                           \
      if (TRI == TRI_CLOW || TRI == TRI_CDIAG || TRI == TRI_COFF) on = on && (fj <= fi);                          \
      t_ |= on ? (1u << (q * FC + s)) : 0u;                                                                       \
    }                                                                                                             \
    amt = __builtin_amdgcn_readfirstlane(t_);                                                                     \
  }

  // next position of the flattened (tile, k-slab) stream
#define G2_ADV(m_, n_, k_, mo_, no_, ko_)                                                                          \
  {                                                                                                               \
    mo_ = (m_); no_ = (n_); ko_ = (k_) + 1;                                                                       \
    if (ko_ == G2_NKS(n_)) {                                                                                      \
      ko_ = 0; no_ = (n_) + TN;                                                                                   \
      if (TRI == TRI_CDIAG) { mo_ = (m_) + TM; no_ = mo_; }                                                       \
      else if (no_ >= N || ((TRI == TRI_CLOW || TRI == TRI_COFF) && no_ >= (m_) + (TRI == TRI_COFF ? 0 : TM))) { no_ = 0; mo_ = (m_) + TM; } \
    }                                                                                                             \
  }
  // operand registers of one 4-k step (A: FR fragments, B: FC fragments x 4 column groups) from LDS buffer bs_
#define G2_LOAD(bs_, kk, A_, B_)                                                                                   \
  if (!(DBG & 2) || !inloop) {                                                                                    \
    _Pragma("unroll") for (int q = 0; q < FR; ++q) {                                                              \
      const double a_ = (bs_)[offa[kk] + q * WR * 256];                                                           \
      A_[q] = (MODE == 1) ? a_ : -a_;           /* C -= A B' and C = -A B' accumulate (-A) B' onto C resp. 0 */   \
    }                                                                                                             \
    _Pragma("unroll") for (int s = 0; s < FC; ++s) _Pragma("unroll") for (int c = 0; c < 4; ++c)                  \
      B_[s][c] = (bs_)[offb[kk][c >> 1] + s * WC * 256 + (c & 1) * 16];                                           \
  }
  static_assert(FC <= FR, "the B registers of the next step are fetched along the FR fragment rows");
  // One 4-k step: FR x FC x 4 MFMAs.  Every fragment is computed unconditionally (fragments outside M x N multiply stale LDS rows and
  // are never stored: guarding single MFMAs makes hipcc shuttle the accumulators between AGPRs, VGPRs and scratch);
  // the only variant is the compile-time staircase s <= q of a diagonal tile of a symmetric update.
  // After fragment row q the wave fetches a fifth of the next step's operands from LDS and issues DMA piece T0 + q of
  // slab (pm, pn, pk) -> buffer pb: with one wave per SIMD every instruction that blocks at issue (a full LDS or VMEM
  // queue) idles the matrix pipe, so LDS reads and DMA are metered out between the MFMAs instead of issued in bursts.
#define G2_STEP(A_, B_, NA_, NB_, nbs_, nkk, ld_, T0, pv, pm, pn, pk, pb)                                           \
  _Pragma("unroll") for (int q = 0; q < FR; ++q) {                                                                \
    _Pragma("unroll") for (int s = 0; s < FC; ++s)                                                                \
      if (TRI != TRI_CDIAG || s <= q) { _Pragma("unroll") for (int c = 0; c < 4; ++c)                             \
        acc[q][s][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(A_[q], B_[s][c], acc[q][s][c], 0, 0, 0); }              \
    if ((ld_) && !(DBG & 2)) {            /* a fifth of the NEXT step's operand fetch rides on this fragment row */ \
      const double a_ = (nbs_)[offa[nkk] + q * WR * 256];                                                         \
      NA_[q] = (MODE == 1) ? a_ : -a_;    /* C -= A B' and C = -A B' accumulate (-A) B' onto C resp. 0 */         \
      if (q < FC) { _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                 \
        NB_[q][c] = (nbs_)[offb[nkk][c >> 1] + q * WC * 256 + (c & 1) * 16]; }                                    \
    }                                                                                                             \
    if (!(DBG & 1) && (T0) + q < NGW && (pv)) G2_PIECE((T0) + q, pm, pn, pk, pb)                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
  }
  // counted wait + workgroup barrier: at most n_ VMEM operations of this wave may still be in flight
#define G2_BARRIER(n_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(n_) : "memory")

  // Software pipeline over the flattened (tile, k-slab) stream, THREE LDS buffers (slab t lives in buffer t % 3):
  //   iteration s:  steps 0-2 = {fetch operands of step kk + 1, MFMAs of step kk}; the DMA pieces 5.. of slab s + 2 ride
  //                 on the MFMAs of steps 0 and 1;
  //                 barrier: slab s + 1 complete (counted vmcnt: the NGW pieces of slab s + 2 stay in flight) and
  //                 every wave has fetched its last operands of slab s, so buffer s % 3 is free;
  //                 step 3 = MFMAs of kk = 3 with pieces 0-4 of slab s + 3 (-> buffer s % 3) and the operand fetch
  //                 of step 0 of slab s + 1 riding on them;  epilogue if the tile is finished.
  // A slab's DMA therefore has about two slab times to land.  Operand registers are double-buffered by hand (set 0 /
  // set 1) and the steps fenced with sched_barrier: left alone the scheduler hoists every LDS read of a slab above
  // its first MFMA and spills.
  static_assert((TRI != TRI_CDIAG && TRI != TRI_COFF) || (TM == TN && FR == FC), "square tiles");
  static_assert(NGW <= 3 * FR && NGW >= FR, "DMA pieces are spread over three MFMA steps");
  int m0 = (TRI == TRI_COFF) ? TM : 0, n0 = 0, ks = 0;
  int m1, n1, k1, m2, n2, k2, m3, n3, k3;
  if (m0 >= M) return;
  unsigned amt;
  double a0[FR], b0[FC][4], a1[FR], b1[FC][4];
  G2_TILE_MASK(m0, n0)
#pragma unroll
  for (int q = 0; q < FR; ++q) {               // accumulators of the first tile: C (MODE 0) or 0
    __builtin_amdgcn_sched_barrier(0);         // one fragment row of loads in flight at a time (registers)
#pragma unroll
    for (int s = 0; s < FC; ++s) {
      if (MODE == 0) {
        const unsigned org = (unsigned)((((m0 >> 4) + wr + q * WR) * 16) * ldc + ((n0 >> 4) + wc + s * WC) * 16) * 8u;
        const unsigned so = (amt & (1u << (q * FC + s))) ? org : 0x80000000u;
        const g2_d2 c0 = __builtin_bit_cast(g2_d2, __builtin_amdgcn_raw_buffer_load_b128(crs, dlo, so, 0));
        const g2_d2 c1 = __builtin_bit_cast(g2_d2, __builtin_amdgcn_raw_buffer_load_b128(crs, dlo + 16u, so, 0));
        acc[q][s] = (g2_d4){c0[0], c0[1], c1[0], c1[1]};
      } else acc[q][s] = (g2_d4){0.0, 0.0, 0.0, 0.0};
    }
  }
  G2_ADV(m0, n0, ks, m1, n1, k1)
  G2_ADV(m1, n1, k1, m2, n2, k2)
  __syncthreads();                             // LDS free (previous user)
  G2_ISSUE(m0, n0, 0, 0)
  if (m1 < M) G2_ISSUE(m1, n1, k1, 1)
  if (m2 < M) { _Pragma("unroll") for (int t = 0; t < FR; ++t) G2_PIECE(t, m2, n2, k2, 2) }   // pieces 0 .. FR-1; the rest rides on steps 0-1
  if (m2 < M) { G2_BARRIER(NGW + FR); } else if (m1 < M) { G2_BARRIER(NGW); } else { G2_BARRIER(0); }   // slab 0 landed
  bool inloop = false;
  G2_LOAD(lds, 0, a0, b0)
  if (DBG & 2) { G2_LOAD(lds, 1, a1, b1) }
  inloop = true;
  int bi0 = 0, bi1 = 1, bi2 = 2;               // buffers of slabs s, s + 1, s + 2
  bool drain = false;                          // epilogue stores in flight: the next wait must be a full one
  TMPC_T0()
  while (m0 < M) {
    G2_ADV(m2, n2, k2, m3, n3, k3)
    const double* bs = lds + bi0 * BUF;
    const bool v1 = m1 < M, v2 = m2 < M, v3 = v2 && (m3 < M);
    __builtin_amdgcn_sched_barrier(0);
    G2_STEP(a0, b0, a1, b1, bs, 1, true, FR, v2, m2, n2, k2, bi2)
    G2_STEP(a1, b1, a0, b0, bs, 2, true, 2 * FR, v2, m2, n2, k2, bi2)
    G2_STEP(a0, b0, a1, b1, bs, 3, true, 3 * FR, false, 0, 0, 0, 0)
    TMPC_T(8)
    if (v2 && !drain) { G2_BARRIER(NGW); } else { G2_BARRIER(0); }
    drain = false;
    TMPC_T(9)
    const bool tile_end = (k1 == 0) || !v1;
    const double* bn = lds + bi1 * BUF;
    G2_STEP(a1, b1, a0, b0, bn, 0, !tile_end, 0, v3, m3, n3, k3, bi0)   // (at a tile end the fetch waits until the epilogue has released its registers)
    TMPC_T(11)
    // ---- tile finished: read-modify-write its C fragments (32 contiguous bytes per lane and fragment)
    if (tile_end) {
      const int fi0 = (m0 >> 4) + wr, fj0 = (n0 >> 4) + wc;
      const unsigned amc = amt;
      if (v1) G2_TILE_MASK(m1, n1)
      const int fi1 = (m1 >> 4) + wr, fj1 = (n1 >> 4) + wc;
      // Branch-free: C goes through a raw buffer resource; a fragment that must not be stored (outside M x N, above the
      // diagonal) gets an out-of-range scalar offset, so its loads return 0 and its stores are dropped by the bounds
      // check.  (Wave-uniform `if`s around the accumulators make hipcc spill them.)  For C -= A B' the accumulators
      // carry C from the start of the tile (loaded here for the NEXT tile, straight into the registers the stores
      // have just released) and the A operand is negated on its way out of LDS, so the epilogue is store-only.
#pragma unroll
      for (int q = 0; q < FR; ++q) {
#pragma unroll
        for (int s = 0; s < FC; ++s) {
          const unsigned org = (unsigned)(((fi0 + q * WR) * 16) * ldc + (fj0 + s * WC) * 16) * 8u;   // wave-uniform fragment origin (bytes)
          const unsigned so = (amc & (1u << (q * FC + s))) ? org : 0x80000000u;
          const g2_d4 v = acc[q][s];
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(g2_u4, (g2_d2){v[0], v[1]}), crs, dlo, so, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(g2_u4, (g2_d2){v[2], v[3]}), crs, dlo + 16u, so, 0);
        }
#pragma unroll
        for (int s = 0; s < FC; ++s) {
          if (MODE == 0) {
            const unsigned org = (unsigned)(((fi1 + q * WR) * 16) * ldc + (fj1 + s * WC) * 16) * 8u;
            const unsigned so = (v1 && (amt & (1u << (q * FC + s)))) ? org : 0x80000000u;
            const g2_d2 c0 = __builtin_bit_cast(g2_d2, __builtin_amdgcn_raw_buffer_load_b128(crs, dlo, so, 0));
            const g2_d2 c1 = __builtin_bit_cast(g2_d2, __builtin_amdgcn_raw_buffer_load_b128(crs, dlo + 16u, so, 0));
            acc[q][s] = (g2_d4){c0[0], c0[1], c1[0], c1[1]};
          } else acc[q][s] = (g2_d4){0.0, 0.0, 0.0, 0.0};
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (v1) G2_LOAD(bn, 0, a0, b0)
      drain = true;                            // loads and stores complete out of order with each other: no counted wait across them
      TMPC_T(12)
    }
    { const int t_ = bi0; bi0 = bi1; bi1 = bi2; bi2 = t_; }
    m0 = m1; n0 = n1; ks = k1;
    m1 = m2; n1 = n2; k1 = k2;
    m2 = m3; n2 = n3; k2 = k3;
  }
  G2_BARRIER(0);                               // stores issued, LDS reads of the last slab done before the caller reuses the buffers
#undef G2_BARRIER
#undef G2_PIECE
#undef G2_STEP
#undef G2_LOAD
#undef G2_ADV
#undef G2_ISSUE
#undef G2_NKS
#undef G2_TILE_MASK
}

}  // namespace tmpc
