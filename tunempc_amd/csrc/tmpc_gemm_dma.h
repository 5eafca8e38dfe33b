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

template <int DEPTH>
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
  const bool zf = (z0 != nullptr), zf0 = zf && wc0 == 0;   // (block-uniform / wave-uniform: the wave of the first strip, which always has work)
  double* Zs = lds + DEPTH * DMA_SLAB;                      // [2][256]: rows 4 q of a mini slab hold right-hand side q
  double accz[4] = {0.0, 0.0, 0.0, 0.0}, zreg = 0.0;
  int obz[2], zpos = 0;
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) obz[hh] = (4 * fj) * 16 + 2 * ((4 * hh + fk) ^ dma_sw(4 * fj));
  const bool zld = zf && tid < 16 * znc;                    // this thread moves one entry of every z slab
  if (zld) { const int k = tid / znc, q = tid - k * znc; zpos = (4 * q) * 16 + 2 * ((k >> 1) ^ dma_sw(4 * q)) + (k & 1); }
  const int nks = K >> 4, nst = A1 ? 2 * nks : nks;
  const bool full = wave_on && i0 == 0 && i1 == 4;
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
#define TMPC_DMA_MMA(PRED)                                                                                \
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
    if (full) TMPC_DMA_MMA(false)                                                                         \
    else if (wave_on) TMPC_DMA_MMA(true)                                                                  \
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
#undef TMPC_DMA_MMA
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
// The same tile GEMM as ONE K stream over SEVERAL output tiles (round 6; VERDICT r5 item 2a: "cross-tile software pipelining").
// wg_tile_dma pays, per 64 x 64 tile, a workgroup barrier, the full latency of its first slab with nothing else of this workgroup in flight,
// and a load - subtract - store epilogue during which no slab of this workgroup is on its way.  Here a workgroup walks `ntiles` consecutive items
// and the slab stream never stops: while the last slabs of tile t are multiplied and its result is loaded, subtracted and stored, the first
// slabs of tile t + 1 are already travelling into the ring of LDS buffers.  Arithmetic per tile is exactly wg_tile_dma's (same slabs, same
// order, same accumulators): results are bit-identical.
// Registers bound the form: the multiply loop of wg_tile_dma takes 100 VGPRs and 86 SGPRs of the 128 / ~100 that four workgroups per CU leave, so
// the descriptions of the tiles (ten pointers and seven integers each, all workgroup-uniform) are computed ONCE up front into a small table in
// LDS; the loop keeps in scalar registers only the operand pair the slab issue is working on and re-reads the table at pair / tile boundaries.
struct DmaTile {                     // one output tile; every field is workgroup-uniform
  double* C; const double *A0, *B0, *A1, *B1;      // A1 / B1: the second operand pair of the stream (or nullptr)
  const double *z0, *z1; double* yz;               // fused right-hand sides (see wg_tile_dma), or nullptr
  int M, N, mode, tri, rot;
  int nst;                                         // slabs of the stream (K / 16, twice with a second pair); 0: no tile (an item without work)
};
constexpr int DMAS_TD = 12;          // doubles per table entry: C, yz, z0, z1, A0, B0, A1, B1 (pointers), then the ints M, N, mode, tri, rot, pairs, has-z, -
template <int DEPTH>
constexpr int dmas_lds_doubles(int ntiles) { return dma_lds_doubles<DEPTH>() + DMAS_TD * ntiles; }

__device__ __forceinline__ const double* lds_uptr(const double* slot) {          // a pointer stored in LDS, as a wave-uniform (scalar) value
  const unsigned long long v = *(const unsigned long long*)slot;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const double*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int lds_uint(const double* slot, int i) { return __builtin_amdgcn_readfirstlane(((const int*)slot)[i]); }

template <int DEPTH, class TileOf>
__device__ __forceinline__ void wg_tiles_dma_stream(TileOf&& tile_of, int ntiles, int ld, int K, double* lds, int znc = 3) {
  static_assert(DEPTH == 2, "ring of two slab buffers (three were measured slower: three workgroups per CU)");
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fk = lane >> 4, fq = (lane >> 2) & 3, fj = lane & 3;
  const int nks = K >> 4;
  double* Zs = lds + DEPTH * DMA_SLAB;
  double* desc = Zs + 512;
  int* nvalid_slot = (int*)(Zs + 511);                      // (the mini slabs of the right-hand sides end at 256 + 143)
  __syncthreads();                                          // LDS free (previous user)
  if (wv == 0) {                                            // the table of the tiles with work, in item order
    int nv = 0;
    for (int t = 0; t < ntiles; ++t) {
      const DmaTile T = tile_of(t);
      if (T.nst == 0) continue;
      if (lane == 0) {
        double* e = desc + DMAS_TD * nv;
        const double** ep = (const double**)e;
        ep[0] = T.C; ep[1] = T.yz; ep[2] = T.z0; ep[3] = T.z1; ep[4] = T.A0; ep[5] = T.B0; ep[6] = T.A1; ep[7] = T.B1;
        int* ei = (int*)(e + 8);
        ei[0] = T.M; ei[1] = T.N; ei[2] = T.mode; ei[3] = T.tri; ei[4] = T.rot; ei[5] = T.A1 ? 2 : 1; ei[6] = T.z0 ? 1 : 0; ei[7] = 0;
      }
      ++nv;
    }
    if (lane == 0) *nvalid_slot = nv;
  }
  __syncthreads();
  const int nvalid = __builtin_amdgcn_readfirstlane(*nvalid_slot);
  if (nvalid == 0) return;
  unsigned vo[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = wv * 16 + 8 * h + (lane >> 3), c = lane & 7;
    vo[h] = (unsigned)(row * ld + 2 * (c ^ dma_sw(row))) * 8u;
  }
  int oa[2];
  {
    const int swa = dma_sw(4 * fq + fj);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) oa[hh] = (4 * fq + fj) * 16 + 2 * ((4 * hh + fk) ^ swa);
  }
  int obz[2], zpos = 0;
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) obz[hh] = (4 * fj) * 16 + 2 * ((4 * hh + fk) ^ dma_sw(4 * fj));
  const bool zthr = tid < 16 * znc;                         // the threads that move one entry of a z slab
  if (zthr) { const int k = tid / znc, q = tid - k * znc; zpos = (4 * q) * 16 + 2 * ((k >> 1) ^ dma_sw(4 * q)) + (k & 1); }
  // ---- issue cursor: tile, pair, slab; the pair's operands in scalar registers
  int it = 0, ip = 0, is = 0, inp = lds_uint(desc + 8, 5), iM = lds_uint(desc + 8, 0), iN = lds_uint(desc + 8, 1);
  const double *iA = lds_uptr(desc + 4), *iB = lds_uptr(desc + 5);
  int gi = 0, g = 0;                                        // slabs issued / multiplied so far (ring positions gi % DEPTH, g % DEPTH)
  auto issue_one = [&]() {
    if (it >= nvalid) return;
    double* As_ = lds + (gi % DEPTH) * DMA_SLAB;
    const int so = is * 128;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)iA, 0, (int)(((unsigned)(iM - 1) * (unsigned)ld + (unsigned)K) * 8u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)iB, 0, (int)(((unsigned)(iN - 1) * (unsigned)ld + (unsigned)K) * 8u), 0x00020000);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_vptr)(As_ + (wv * 16 + 8 * h) * 16), 16, vo[h], so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_vptr)(As_ + 1024 + (wv * 16 + 8 * h) * 16), 16, vo[h], so, 0, 0);
    }
    ++gi;
    if (++is >= nks) {
      is = 0;
      if (++ip >= inp) {
        ip = 0; ++it;
        if (it < nvalid) { const double* e = desc + DMAS_TD * it + 8; inp = lds_uint(e, 5); iM = lds_uint(e, 0); iN = lds_uint(e, 1); }
      }
      if (it < nvalid) { const double* e = desc + DMAS_TD * it + 4 + 2 * ip; iA = lds_uptr(e); iB = lds_uptr(e + 1); }
    }
  };
  if (lds_uint(desc + 8, 6) && zthr) Zs[zpos] = ((const double* const*)desc)[2][tid];      // (ring position 0)
#pragma unroll
  for (int s = 0; s < DEPTH - 1; ++s) issue_one();
  int skipwait = 0;                                         // steps after an epilogue whose slab had landed before it (the epilogue waits for everything)
  for (int tc = 0; tc < nvalid; ++tc) {
    const double* ed = desc + DMAS_TD * tc;
    const int cM = lds_uint(ed + 8, 0), cN = lds_uint(ed + 8, 1), cmode = lds_uint(ed + 8, 2), ctri = lds_uint(ed + 8, 3), crot = lds_uint(ed + 8, 4);
    const int nst = lds_uint(ed + 8, 5) * nks;
    const bool zf = lds_uint(ed + 8, 6) != 0;
    const bool znt = (tc + 1 < nvalid) && lds_uint(ed + DMAS_TD + 8, 6) != 0;      // the next tile carries right-hand sides
    // ---- per-tile geometry
    const int wc0 = ((wv + crot) & 3) * 16;
    const bool wave_on = wc0 < cN;
    int i0 = 0;
    if (ctri != GM_NOTRI) { i0 = (ctri + wc0) >> 4; if (i0 < 0) i0 = 0; }
    const int i1 = (cM + 15) >> 4;
    const bool full = wave_on && i0 == 0 && i1 == 4;
    int ob[4][2];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int swb = dma_sw(wc0 + 4 * fj + e);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) ob[e][hh] = (wc0 + 4 * fj + e) * 16 + 2 * ((4 * hh + fk) ^ swb);
    }
    const bool zf0 = zf && wc0 == 0;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[i][c] = 0.0;
    double accz[4] = {0.0, 0.0, 0.0, 0.0};
    // One step of the stream with the ring position Q as a compile-time constant of the copy (with a run-time buffer address every fragment read costs
    // a vector add, and fp64 MFMA and the vector ALU share their issue slots: measured 8 % of the kernel).  A tile starts at whatever position the
    // previous one ended, so the tile loop exists twice: ring origin 0 and 1.
#define TMPC_DMAS_MMA(PRED, Q)                                                                            \
  {                                                                                                       \
    const double* As_ = lds + (Q) * DMA_SLAB;                                                             \
    const double* Bs_ = As_ + 1024;                                                                       \
    const double* Zc_ = Zs + (Q) * 256;                                                                   \
    _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                                    \
      double2_t a[4];                                                                                     \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) a[i] = *(const double2_t*)(As_ + oa[hh] + i * 256);   \
      _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) {                                                  \
        const double2_t bv = *(const double2_t*)(Bs_ + ob[cb][hh]);                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                   \
          if (!(PRED) || (i >= i0 && i < i1)) {                                                           \
            acc[i][cb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i][0], bv[0], acc[i][cb], 0, 0, 0);         \
            acc[i][cb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i][1], bv[1], acc[i][cb], 0, 0, 0);         \
          }                                                                                               \
        }                                                                                                 \
      }                                                                                                   \
      if (zf0) {                                                                                          \
        const double2_t bz = *(const double2_t*)(Zc_ + obz[hh]);                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                   \
          if (i < i1) {                                                                                   \
            accz[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i][0], bz[0], accz[i], 0, 0, 0);               \
            accz[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i][1], bz[1], accz[i], 0, 0, 0);               \
          }                                                                                               \
        }                                                                                                 \
      }                                                                                                   \
    }                                                                                                     \
  }
#define TMPC_DMAS_STEP(Q)                                                                                 \
  {                                                                                                       \
    if (skipwait > 0) { --skipwait; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }                  \
    else { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }                                  \
    __builtin_amdgcn_s_barrier();                                                                         \
    issue_one();                                                                                          \
    /* z slab of the NEXT step (this tile's next slab, or the first slab of the next tile): the pointer is read from the table by the threads that use it */ \
    double zreg = 0.0; bool znext = false;                                                                \
    if (zthr) {                                                                                           \
      if (sc + 1 < nst) {                                                                                 \
        if (zf) { const int s1 = sc + 1, sec = s1 >= nks; zreg = ((const double* const*)ed)[2 + sec][(size_t)(s1 - (sec ? nks : 0)) * 16 * znc + tid]; znext = true; } \
      } else if (znt) { zreg = ((const double* const*)(ed + DMAS_TD))[2][tid]; znext = true; }           \
    }                                                                                                     \
    if (full) TMPC_DMAS_MMA(false, Q) else if (wave_on) TMPC_DMAS_MMA(true, Q)                            \
    if (znext) Zs[(1 - (Q)) * 256 + zpos] = zreg;                                                         \
    ++sc; ++g;                                                                                            \
  }
    if ((g & 1) == 0) { for (int sc = 0; sc < nst;) { TMPC_DMAS_STEP(0) if (sc < nst) TMPC_DMAS_STEP(1) } }
    else { for (int sc = 0; sc < nst;) { TMPC_DMAS_STEP(1) if (sc < nst) TMPC_DMAS_STEP(0) } }
#undef TMPC_DMAS_STEP
#undef TMPC_DMAS_MMA
    // ---- epilogue of the tile; the slabs of the next tile are in flight.  The loads below are the youngest entries of the in-order counter, so once
    // they have arrived (the explicit wait covers waves without a fragment) every slab issued so far has landed: the next DEPTH - 1 steps need no
    // wait of their own, and the stores drain behind the first products of the next tile.
    if (zf0 && fj < znc) {
      double* yz = ((double* const*)ed)[1];
      double yv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) yv[i] = (i < i1) ? yz[(size_t)(16 * i + 4 * fq + fk) * znc + fj] : 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < i1) yz[(size_t)(16 * i + 4 * fq + fk) * znc + fj] = yv[i] - accz[i];
    }
    if (wave_on) {
      typedef double2_t __attribute__((address_space(1)))* gptr2;
      double* C = ((double* const*)ed)[0];
      double2_t cu[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        cu[i][0] = (double2_t){0.0, 0.0}; cu[i][1] = cu[i][0];
        if (cmode == GM_SUB && i >= i0 && i < i1) {
          gcptr2 cq = (gcptr2)(C + (size_t)(16 * i + 4 * fq + fk) * ld + wc0 + 4 * fj);
          cu[i][0] = cq[0]; cu[i][1] = cq[1];
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i >= i0 && i < i1) {
          gptr2 cp = (gptr2)(C + (size_t)(16 * i + 4 * fq + fk) * ld + wc0 + 4 * fj);
          double2_t u0 = cu[i][0], u1 = cu[i][1];
          if (cmode == GM_SET) { u0[0] = acc[i][0]; u0[1] = acc[i][1]; u1[0] = acc[i][2]; u1[1] = acc[i][3]; }
          else { u0[0] -= acc[i][0]; u0[1] -= acc[i][1]; u1[0] -= acc[i][2]; u1[1] -= acc[i][3]; }
          cp[0] = u0; cp[1] = u1;
        }
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    skipwait = DEPTH - 1;
  }
}

}  // namespace tmpc
