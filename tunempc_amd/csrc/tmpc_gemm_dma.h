// fp64 tile GEMM fed by LDS-DMA: C (M x N <= 64 x 64) {-=, = -} sum_q A_q B_q'   (K columns each, up to two operand pairs as one stream)
//
// The K slabs (64 rows x 16 columns per operand) go global -> LDS by buffer_load_dwordx4 ... lds: no staging registers, no
// ds_write, and the slab after next is in flight while the current one is multiplied (DEPTH buffers, s_waitcnt vmcnt(N) with
// N > 0, raw s_barrier).  The DMA writes lane-linear -- 8 lanes = one 128-byte row -- so the LDS image is unpadded
// [row][8 granules of 16 B]; bank conflicts are avoided by a swizzle that lives in the per-lane GLOBAL address: granule c of row
// r holds the K pair c ^ s(r), s(r) = (r & 7) ^ 2 ((r >> 3) & 1).  Fragment reads are ds_read_b128, one K pair per lane,
// conflict-free for the A fragments (8 consecutive rows per LDS cycle) and the B fragments (4 rows, stride 4).
// v_mfma_f64_4x4x4_4b consumes 4 k per instruction, one per 16-lane group: instruction 1 takes the even k of the group's
// pairs (the lanes' .x), instruction 2 the odd ones (.y) -- A and B agree on the assignment, so the product is the same.
// Each wave owns a 16-column strip of the tile and all (up to four) 16-row fragments of it: 4 + 4 fragment reads per 32
// MFMAs, and on tiles that cross the diagonal of a symmetric update, or end at the edge of the block, a wave skips exactly
// the 16 x 16 blocks that are not needed (executed / algorithmic flops of the lower triangle of a 304-block: 1.05 instead of
// 1.22 with 32 x 32 wave quadrants).  The strip of wave w is (w + rot) & 3 so that the short strips of diagonal tiles do
// not always land on the same SIMD.   Measured against the register-staged core: scripts/micro/dma_gemm.hip,
// profiles/r2v_pmc_gemm_core_micro.txt.
#pragma once
#include "tmpc_common.h"
#include "tmpc_factor.h"

namespace tmpc {

typedef __attribute__((address_space(3))) void* lds_vptr;
constexpr int DMA_SLAB = 2048;                 // doubles per LDS buffer: A slab (64 x 16) + B slab
template <int DEPTH>
constexpr int dma_lds_doubles() { return DEPTH * DMA_SLAB + 512; }       // + two mini slabs for fused right-hand sides

// swizzle of row r
__device__ __forceinline__ int dma_sw(int r) { return (r & 7) ^ (((r >> 3) & 1) << 1); }

template <int DEPTH, bool ZFUSE = false>
__device__ __forceinline__ void wg_tile_dma(double* C, int ldc, const double* A0, const double* B0, const double* A1, const double* B1, int ld,
                                            int M, int N, int K, int mode, int tri, int rot, double* lds, int ldb = 0,
                                            const double* z0 = nullptr, const double* z1 = nullptr, double* yz = nullptr, int znc = 0) {
  // z0 / z1 (optional, [K][znc], znc <= 4): right-hand sides that ride along as one more B fragment of the wave that owns the first strip,
  //   yz[M][znc] -= A0 z0 (+ A1 z1)   -- the forward substitution step of the node this tile row belongs to, fused into its update
  //   (the slab of z for the next step is fetched during the MFMAs of the current one and parked next to the operand buffers).
  if (ldb == 0) ldb = ld;                                   // leading dimension of the B operands (default: as A)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wc0 = ((wv + rot) & 3) * 16;
  const int fk = lane >> 4, fq = (lane >> 2) & 3, fj = lane & 3;
  const bool wave_on = wc0 < N;
  int i0 = 0;
  if (tri != GM_NOTRI) { i0 = (tri + wc0) >> 4; if (i0 < 0) i0 = 0; }        // row fragments above the strip's first column are not needed
  const int i1 = (M + 15) >> 4;
  // (Round 6 measured a form with the fragment range [i0, i1) as a LITERAL of ten copies of the multiply block, chosen by a scalar branch per step -- the tiles
  // that are not "full", 14 of the 40 of a node at d = 300, run their MFMA pairs behind run-time tests.  Bit-identical, 2.5 % SLOWER for this fp64 tile
  // (62.9 against 61.3 ms per update phase, profiles/r6_ab_literal_ranges.txt): 124 VGPRs instead of 100 and ten times the code.  The float32 tile below
  // needs it: there the run-time form compiles to a compare, a branch and a block of register copies per MFMA.)
  const int frange = wave_on ? (i0 * 8 + i1) : -1;
  const unsigned abytes = ((unsigned)(M - 1) * (unsigned)ld + (unsigned)K) * 8u, bbytes = ((unsigned)(N - 1) * (unsigned)ldb + (unsigned)K) * 8u;
  const __amdgpu_buffer_rsrc_t ra0 = __builtin_amdgcn_make_buffer_rsrc((void*)A0, 0, (int)abytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb0 = __builtin_amdgcn_make_buffer_rsrc((void*)B0, 0, (int)bbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc((void*)(A1 ? A1 : A0), 0, (int)abytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb1 = __builtin_amdgcn_make_buffer_rsrc((void*)(B1 ? B1 : B0), 0, (int)bbytes, 0x00020000);
  // DMA: wave wv moves rows 16 wv .. 16 wv + 15 of both operands, two instructions of 8 rows each
  unsigned vo[2], vob[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = wv * 16 + 8 * h + (lane >> 3), c = lane & 7;
    vo[h] = (unsigned)(row * ld + 2 * (c ^ dma_sw(row))) * 8u;
    vob[h] = (unsigned)(row * ldb + 2 * (c ^ dma_sw(row))) * 8u;
  }
  // fragment read offsets (doubles) inside a slab, one per half hh of the 16-column slab (K pairs 4 hh + fk)
  int oa[2], ob[4][2];
  {
    const int swa = dma_sw(4 * fq + fj);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) oa[hh] = (4 * fq + fj) * 16 + 2 * ((4 * hh + fk) ^ swa);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int swb = dma_sw(wc0 + 4 * fj + e);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) ob[e][hh] = (wc0 + 4 * fj + e) * 16 + 2 * ((4 * hh + fk) ^ swb);
    }
  }
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[i][c] = 0.0;
  const bool zf = ZFUSE && (z0 != nullptr), zf0 = zf && wc0 == 0;   // (block-uniform / wave-uniform: the wave of the first strip, which always has work)
  double* Zs = lds + DEPTH * DMA_SLAB;                      // [2][256]: rows 4 q of a mini slab hold right-hand side q
  double accz[4] = {0.0, 0.0, 0.0, 0.0}, zreg = 0.0;
  int obz[2], zpos = 0;
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) obz[hh] = (4 * fj) * 16 + 2 * ((4 * hh + fk) ^ dma_sw(4 * fj));
  const bool zld = zf && tid < 16 * znc;                    // this thread moves one entry of every z slab
  if (zld) { const int k = tid / znc, q = tid - k * znc; zpos = (4 * q) * 16 + 2 * ((k >> 1) ^ dma_sw(4 * q)) + (k & 1); }
  const int nks = K >> 4, nst = A1 ? 2 * nks : nks;
#define TMPC_DMA_ISSUE(S, Q)                                                                              \
  {                                                                                                       \
    double* As_ = lds + (Q) * DMA_SLAB;                                                                   \
    const bool second_ = (S) >= nks;                                                                      \
    const int so_ = ((S) - (second_ ? nks : 0)) * 128;                                                    \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                       \
      if (second_) {                                                                                      \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra1, (lds_vptr)(As_ + (wv * 16 + 8 * h) * 16), 16, vo[h], so_, 0, 0);        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb1, (lds_vptr)(As_ + 1024 + (wv * 16 + 8 * h) * 16), 16, vob[h], so_, 0, 0); \
      } else {                                                                                            \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra0, (lds_vptr)(As_ + (wv * 16 + 8 * h) * 16), 16, vo[h], so_, 0, 0);        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb0, (lds_vptr)(As_ + 1024 + (wv * 16 + 8 * h) * 16), 16, vob[h], so_, 0, 0); \
      }                                                                                                   \
    }                                                                                                     \
  }
#define TMPC_DMA_MMA_RT(PRED)                                                                             \
  _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                                      \
    double2_t a[4];                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) a[i] = *(const double2_t*)(As_ + oa[hh] + i * 256);     \
    _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) {                                                    \
      const double2_t bv = *(const double2_t*)(Bs_ + ob[cb][hh]);                                         \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
        if (!(PRED) || (i >= i0 && i < i1)) {                                                             \
          acc[i][cb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i][0], bv[0], acc[i][cb], 0, 0, 0);           \
          acc[i][cb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i][1], bv[1], acc[i][cb], 0, 0, 0);           \
        }                                                                                                 \
      }                                                                                                   \
    }                                                                                                     \
    if (zf0) {                                                                                            \
      const double2_t bz = *(const double2_t*)(Zc_ + obz[hh]);                                            \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
        if (i < i1) {                                                                                     \
          accz[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i][0], bz[0], accz[i], 0, 0, 0);                 \
          accz[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i][1], bz[1], accz[i], 0, 0, 0);                 \
        }                                                                                                 \
      }                                                                                                   \
    }                                                                                                     \
  }
#define TMPC_DMA_MMA_DISPATCH()                                                                           \
  if (frange == 4) TMPC_DMA_MMA_RT(false)                                                                 \
  else if (frange >= 0) TMPC_DMA_MMA_RT(true)
#define TMPC_DMA_ZSRC(S) (((S) >= nks ? z1 : z0) + (size_t)((S) - ((S) >= nks ? nks : 0)) * 16 * znc + tid)
#define TMPC_DMA_STEP(S, Q)                                                                               \
  {                                                                                                       \
    if ((S) + DEPTH - 2 < nst - 1) { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * (DEPTH - 2)) : "memory"); } \
    else { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }                                  \
    __builtin_amdgcn_s_barrier();                                                                         \
    if ((S) + DEPTH - 1 < nst) TMPC_DMA_ISSUE((S) + DEPTH - 1, ((Q) + DEPTH - 1) % DEPTH)                 \
    if (zld && (S) + 1 < nst) zreg = *TMPC_DMA_ZSRC((S) + 1);                                             \
    const double* As_ = lds + (Q) * DMA_SLAB;                                                             \
    const double* Bs_ = As_ + 1024;                                                                       \
    const double* Zc_ = Zs + ((S) & 1) * 256;                                                             \
    TMPC_DMA_MMA_DISPATCH()                                                                               \
    if (zld && (S) + 1 < nst) Zs[(((S) + 1) & 1) * 256 + zpos] = zreg;                                    \
  }
  __syncthreads();                                          // LDS free (previous user)
  if (zld) Zs[zpos] = *TMPC_DMA_ZSRC(0);
#pragma unroll
  for (int s = 0; s < DEPTH - 1; ++s)
    if (s < nst) TMPC_DMA_ISSUE(s, s)
  for (int s = 0; s < nst; s += DEPTH) {
#pragma unroll
    for (int q = 0; q < DEPTH; ++q)
      if (s + q < nst) TMPC_DMA_STEP(s + q, q)
  }
#undef TMPC_DMA_STEP
#undef TMPC_DMA_ZSRC
#undef TMPC_DMA_MMA_DISPATCH
#undef TMPC_DMA_MMA_RT
#undef TMPC_DMA_ISSUE
  if (zf0 && fj < znc) {                                     // yz -= (rows of A) z: lane holds row 16 i + 4 fq + fk, right-hand side fj
    double yv[4];                                            // (all loads before the first store, as for C below)
#pragma unroll
    for (int i = 0; i < 4; ++i) yv[i] = (i < i1) ? yz[(size_t)(16 * i + 4 * fq + fk) * znc + fj] : 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < i1) yz[(size_t)(16 * i + 4 * fq + fk) * znc + fj] = yv[i] - accz[i];
  }
  if (wave_on) {
    typedef double2_t __attribute__((address_space(1)))* gptr2;
    // C -= ...: all (up to four) fragments of the tile are fetched before the first one is stored.  Written fragment by fragment the
    // compiler must keep every load behind the previous fragment's store (they may alias): four exposed round trips per tile instead of one.
    double2_t cu[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      cu[i][0] = (double2_t){0.0, 0.0}; cu[i][1] = cu[i][0];
      if (mode == GM_SUB && i >= i0 && i < i1) {
        gcptr2 cq = (gcptr2)(C + (size_t)(16 * i + 4 * fq + fk) * ldc + wc0 + 4 * fj);
        cu[i][0] = cq[0]; cu[i][1] = cq[1];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i >= i0 && i < i1) {
        gptr2 cp = (gptr2)(C + (size_t)(16 * i + 4 * fq + fk) * ldc + wc0 + 4 * fj);
        double2_t u0 = cu[i][0], u1 = cu[i][1];
        if (mode == GM_SET) { u0[0] = acc[i][0]; u0[1] = acc[i][1]; u1[0] = acc[i][2]; u1[1] = acc[i][3]; }
        else { u0[0] -= acc[i][0]; u0[1] -= acc[i][1]; u1[0] -= acc[i][2]; u1[1] -= acc[i][3]; }
        cp[0] = u0; cp[1] = u1;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// The same tile in SINGLE precision (round 6): C (fp64, M x N <= 64 x 64) {-=, = -} sum_q A_q B_q' with A, B float32 copies of the operands
// (row stride ld32, a multiple of 32 with zero padding) and float32 accumulation on v_mfma_f32_16x16x4f32 -- twice the matrix rate of the fp64
// form and half the operand bytes.  Used by k_cr_update_dma for the Schur-complement updates of the EARLY main-phase iterations only (per
// problem, while mu / kappa > Opts::lowp_switch): there the interior-point direction tolerates a 1e-7 perturbation of the factor -- same
// iteration counts, the converged central-path point moves by 1e-11 ... 2e-10 (tests/tools/fp32_update_probe.py, profiles/r6_fp32_*.txt).
// Layout: a slab is 64 rows x 32 floats = the same 128-byte rows as the fp64 slab of 16 doubles, so the DMA pattern, the LDS image and the
// granule swizzle are those of wg_tile_dma; a lane reads one 16-byte granule (four consecutive k) of row lane & 15 per fragment and feeds
// element e of it to MFMA e -- A and B agree on which k a slot means, the sum over k is the same.  Accumulator of fragment i: C rows
// 16 i + 4 (lane >> 4) + r, column wc0 + (lane & 15).
template <int DEPTH>
__device__ __forceinline__ void wg_tile_dma_f32(double* C, int ldc, const float* A0, const float* B0, const float* A1, const float* B1, int ld32,
                                                int M, int N, int K32, int mode, int tri, int rot, double* lds, int ablate = 0) {
  // ablate (-DTMPC_ABLATE builds only, scripts/gpu_r6_ablate.sh): 1 no C load / store, 2 no MFMA, 3 no slab DMA, 4 no barrier -- what bounds the tile
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc0 = ((wv + rot) & 3) * 16;
  const int fk = lane >> 4, fr = lane & 15;
  const bool wave_on = wc0 < N;
  int i0 = 0;
  if (tri != GM_NOTRI) { i0 = (tri + wc0) >> 4; if (i0 < 0) i0 = 0; }
  const int i1 = (M + 15) >> 4;
  const unsigned abytes = ((unsigned)(M - 1) * (unsigned)ld32 + (unsigned)K32) * 4u, bbytes = ((unsigned)(N - 1) * (unsigned)ld32 + (unsigned)K32) * 4u;
  const __amdgpu_buffer_rsrc_t ra0 = __builtin_amdgcn_make_buffer_rsrc((void*)A0, 0, (int)abytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb0 = __builtin_amdgcn_make_buffer_rsrc((void*)B0, 0, (int)bbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc((void*)(A1 ? A1 : A0), 0, (int)abytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb1 = __builtin_amdgcn_make_buffer_rsrc((void*)(B1 ? B1 : B0), 0, (int)bbytes, 0x00020000);
  unsigned vo[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = wv * 16 + 8 * h + (lane >> 3), c = lane & 7;
    vo[h] = (unsigned)(row * ld32 * 4 + 16 * (c ^ dma_sw(row)));
  }
  // fragment read offsets in doubles (16-byte granule g of row r at r * 16 + 2 g), one per 16-k half hh of the slab
  int oa[2], ob[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    oa[hh] = fr * 16 + 2 * ((4 * hh + fk) ^ dma_sw(fr));
    ob[hh] = (wc0 + fr) * 16 + 2 * ((4 * hh + fk) ^ dma_sw(wc0 + fr));
  }
  float4_t acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
  const int nks = K32 >> 5, nst = A1 ? 2 * nks : nks;
  const int frange = wave_on ? (i0 * 8 + i1) : -1;      // literal fragment ranges, as in wg_tile_dma
#define TMPC_DMAF_ISSUE(S, Q)                                                                             \
  {                                                                                                       \
    double* As_ = lds + (Q) * DMA_SLAB;                                                                   \
    const bool second_ = (S) >= nks;                                                                      \
    const int so_ = ((S) - (second_ ? nks : 0)) * 128;                                                    \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                       \
      if (second_) {                                                                                      \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra1, (lds_vptr)(As_ + (wv * 16 + 8 * h) * 16), 16, vo[h], so_, 0, 0);        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb1, (lds_vptr)(As_ + 1024 + (wv * 16 + 8 * h) * 16), 16, vo[h], so_, 0, 0); \
      } else {                                                                                            \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra0, (lds_vptr)(As_ + (wv * 16 + 8 * h) * 16), 16, vo[h], so_, 0, 0);        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb0, (lds_vptr)(As_ + 1024 + (wv * 16 + 8 * h) * 16), 16, vo[h], so_, 0, 0); \
      }                                                                                                   \
    }                                                                                                     \
  }
#define TMPC_DMAF_MMA(I0, I1)                                                                             \
  _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                                      \
    float4_t a[4];                                                                                        \
    _Pragma("unroll") for (int i = (I0); i < (I1); ++i) a[i] = *(const float4_t*)(As_ + oa[hh] + i * 256); \
    const float4_t bv = *(const float4_t*)(Bs_ + ob[hh]);                                                 \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                       \
      _Pragma("unroll") for (int i = (I0); i < (I1); ++i) {                                               \
        if (ablate != 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], bv[e], acc[i], 0, 0, 0);  \
      }                                                                                                   \
    }                                                                                                     \
  }
#define TMPC_DMAF_MMA_DISPATCH()                                                                          \
  switch (frange) {                                                                                       \
    case 0 * 8 + 4: TMPC_DMAF_MMA(0, 4) break;  case 1 * 8 + 4: TMPC_DMAF_MMA(1, 4) break;                \
    case 2 * 8 + 4: TMPC_DMAF_MMA(2, 4) break;  case 3 * 8 + 4: TMPC_DMAF_MMA(3, 4) break;                \
    case 0 * 8 + 3: TMPC_DMAF_MMA(0, 3) break;  case 1 * 8 + 3: TMPC_DMAF_MMA(1, 3) break;                \
    case 2 * 8 + 3: TMPC_DMAF_MMA(2, 3) break;                                                            \
    case 0 * 8 + 2: TMPC_DMAF_MMA(0, 2) break;  case 1 * 8 + 2: TMPC_DMAF_MMA(1, 2) break;                \
    case 0 * 8 + 1: TMPC_DMAF_MMA(0, 1) break;                                                            \
    default: break;                                                                                       \
  }
#define TMPC_DMAF_STEP(S, Q)                                                                              \
  {                                                                                                       \
    if ((S) + DEPTH - 2 < nst - 1) { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * (DEPTH - 2)) : "memory"); } \
    else { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }                                  \
    if (ablate != 4) __builtin_amdgcn_s_barrier();                                                        \
    if ((S) + DEPTH - 1 < nst && ablate != 3) TMPC_DMAF_ISSUE((S) + DEPTH - 1, ((Q) + DEPTH - 1) % DEPTH) \
    const double* As_ = lds + (Q) * DMA_SLAB;                                                             \
    const double* Bs_ = As_ + 1024;                                                                       \
    TMPC_DMAF_MMA_DISPATCH()                                                                              \
  }
  __syncthreads();                                          // LDS free (previous user)
#pragma unroll
  for (int s = 0; s < DEPTH - 1; ++s)
    if (s < nst) TMPC_DMAF_ISSUE(s, s)
  for (int s = 0; s < nst; s += DEPTH) {
#pragma unroll
    for (int q = 0; q < DEPTH; ++q)
      if (s + q < nst) TMPC_DMAF_STEP(s + q, q)
  }
#undef TMPC_DMAF_STEP
#undef TMPC_DMAF_MMA_DISPATCH
#undef TMPC_DMAF_MMA
#undef TMPC_DMAF_ISSUE
  if (ablate == 1) { if (acc[0][0] == 123.456f) C[0] = 1.0; return; }
  if (wave_on) {
    // all fragments are fetched before the first one is stored (see wg_tile_dma)
    double cu[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        cu[i][r] = 0.0;
        if (mode == GM_SUB && i >= i0 && i < i1) cu[i][r] = C[(size_t)(16 * i + 4 * fk + r) * ldc + wc0 + fr];
      }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i >= i0 && i < i1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) C[(size_t)(16 * i + 4 * fk + r) * ldc + wc0 + fr] = (mode == GM_SET) ? (double)acc[i][r] : cu[i][r] - (double)acc[i][r];
      }
  }
}

// Round 6 measured a cross-tile form of this core (a workgroup walks T tiles as ONE slab stream: the first slabs of a tile in flight during the epilogue of
// the previous one; bit-identical results): 5 - 8 % SLOWER per factorisation in three forms (profiles/r6_update_stream_*.txt, commit 5761969) -- with four
// workgroups per CU the other three already cover a tile's fill and epilogue, and a workgroup that owns T tiles gives up the hardware's tile-by-tile dispatch.

}  // namespace tmpc
