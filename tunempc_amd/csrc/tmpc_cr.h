// Block-cyclic-tridiagonal Cholesky by CYCLIC REDUCTION, as batched tile kernels.
//
// The HKM Schur matrix couples stage k only to k-1 and k+1 (mod p): its block graph is a cycle.  Eliminating a node of
// a cycle always costs the same (Cholesky of its diagonal block, two triangular solves, two symmetric updates and one
// fill block between its two neighbours: 6 1/3 d^3), whichever node goes first -- the sequential order pays the fill as
// a travelling corner row, cyclic reduction pays it as the edge of the next, half as long cycle.  So the odd-even order
// costs no extra flops but exposes p/2, p/4, ... independent eliminations per level, and every phase of a level is a
// batched tile kernel over (active problem, node, tile):
//
//   k_cr_potrf_dma    one workgroup per eliminated node      D_i = L_i L_i'  (block rows on the sweep of the triangular solve, 64 x 64 tile Cholesky + inverted diagonal tiles)
//   k_cr_trsm_dma     per node, neighbour, 64-row strip      O_x <- T[x,i] L_i^-T        (x = the two neighbours a, b of i)
//   k_cr_update_dma   per surviving node: lower tiles        D_s -= O_s O_s' (one or two eliminated neighbours, one K stream)
//                     per eliminated node: all tiles         T[x,y] (=|-=) -O_x O_y'     (the fill edge, x,y = a,b)
// (k_cr_potrf / k_cr_trsm / k_cr_update: the same phases on the register-staged GEMM core of tmpc_factor.h -- blocks wider than 320, debug flag.)
//
// Grids are sized by the number of problems still iterating (compacted list), so the tail of a lockstep batch and
// small batches of long-period problems keep the chip busy; workgroups working on the same node are placed on one XCD
// (block b runs on XCD b % 8) so that the operand panels they share are served by that XCD's L2.
//
// Storage (all dp x dp, row-major): D[node] -> L (diagonal tiles hold L_jj, inverses in Linv); edge slots 0..p-1 live
// in w.O (slot k = edge between stage k and k+1 mod p), fill slots p..2p-1 in w.F.  Every edge is stored with its
// COLUMNS belonging to the endpoint that is eliminated first (schedule: cr_build), which turns every product into the
// same NT form C (+)= A B'.
#pragma once
#include <climits>
#include <vector>

#include "tmpc_common.h"
#include "tmpc_factor.h"
#include "tmpc_gemm_dma.h"

namespace tmpc {

// ------------------------------------------------------------------ schedule (host)
constexpr int CR_EW = 8;   // ints per elimination record
constexpr int CR_UW = 8;   // ints per update record
// elimination record: node, na, nb (neighbours, -1: none), ea, eb (slots of T[na,node], T[nb,node]; -1: none),
//                     fill slot (-1: none), fx (0: fill rows = na / cols = nb, 1: rows = nb / cols = na), facc (1: the edge exists already)
enum { CE_NODE = 0, CE_NA, CE_NB, CE_EA, CE_EB, CE_FILL, CE_FX, CE_FACC };
// update record of a surviving node: node, e0, src0, e1, src1 (slots whose rows belong to `node`, and the eliminated node each came from; -1: none)
enum { CU_NODE = 0, CU_E0, CU_S0, CU_E1, CU_S1 };

struct CrLevel { int eoff, nelim, uoff, nupd; };
struct CrSched {
  int p = 0;
  int prep = 0;                   // 1: p == 1, fold the self-loop D_0 += E_0 + E_0';  2: p == 2, merge the double edge E_0 += E_1
  std::vector<int> orient;        // [p] 0: slot k holds T[k+1,k] (columns = stage k), 1: T[k,k+1] (columns = stage k+1)
  std::vector<int> elim, upd;     // flat records
  std::vector<CrLevel> lev;
};

static inline CrSched cr_build(int p) {
  CrSched s; s.p = p; s.orient.assign(p, 0);
  auto push_elim = [&](int node, int na, int nb, int ea, int eb, int fill, int fx, int facc) {
    const int r[CR_EW] = {node, na, nb, ea, eb, fill, fx, facc};
    s.elim.insert(s.elim.end(), r, r + CR_EW);
  };
  auto push_upd = [&](int node, int e0, int s0, int e1, int s1) {
    const int r[CR_UW] = {node, e0, s0, e1, s1, 0, 0, 0};
    s.upd.insert(s.upd.end(), r, r + CR_UW);
  };
  if (p == 1) {
    s.prep = 1;
    s.lev.push_back({0, 1, 0, 0});
    push_elim(0, -1, -1, -1, -1, -1, 0, 0);
    return s;
  }
  // pass 1: the level at which every node is eliminated
  std::vector<int> lvl(p, INT_MAX);
  auto positions = [](int n) {
    std::vector<int> pos;
    if (n <= 4) pos.push_back(1);                        // short cycles: one node at a time (two fills would meet in one block)
    else for (int i = 1; i < n; i += 2) pos.push_back(i);
    return pos;
  };
  {
    std::vector<int> cyc(p);
    for (int i = 0; i < p; ++i) cyc[i] = i;
    int L = 0;
    while ((int)cyc.size() > 1) {
      const int n = (int)cyc.size();
      std::vector<char> gone(n, 0);
      for (int i : positions(n)) { lvl[cyc[i]] = L; gone[i] = 1; }
      std::vector<int> nxt;
      for (int i = 0; i < n; ++i) if (!gone[i]) nxt.push_back(cyc[i]);
      cyc.swap(nxt); ++L;
    }
    lvl[cyc[0]] = L;
  }
  for (int k = 0; k < p; ++k) {
    const int kn = (k + 1) % p;
    s.orient[k] = (lvl[k] < lvl[kn]) ? 0 : 1;            // columns = the endpoint eliminated first (never equal: neighbours do not share a level)
  }
  if (p == 2) { s.prep = 2; s.orient[0] = 1; s.orient[1] = 0; }     // both edges as T[0,1] (node 1 goes first), merged into slot 0
  // pass 2: records
  std::vector<int> cyc(p), es(p);
  for (int i = 0; i < p; ++i) { cyc[i] = i; es[i] = i; }  // es[i] = slot of the edge between cyc[i] and cyc[i+1 mod n]
  int nfill = 0;
  while ((int)cyc.size() > 1) {
    const int n = (int)cyc.size();
    CrLevel lv; lv.eoff = (int)s.elim.size() / CR_EW; lv.uoff = (int)s.upd.size() / CR_UW;
    const std::vector<int> pos = positions(n);
    std::vector<char> gone(n, 0);
    std::vector<int> fillof(n, -1);                       // fill slot created by the node at position i
    for (int i : pos) {
      gone[i] = 1;
      const int c = cyc[i], a = cyc[i - 1], b = cyc[(i + 1) % n];
      if (n == 2) { push_elim(c, a, -1, es[0], -1, -1, 0, 0); continue; }
      int fill, facc = 0;
      if (n == 3) { fill = es[2]; facc = 1; }            // the neighbours are adjacent already: accumulate into their edge
      else fill = p + nfill++;
      fillof[i] = fill;
      const int fx = (lvl[b] < lvl[a]) ? 0 : 1;          // columns of the fill = the neighbour eliminated first
      push_elim(c, a, b, es[i - 1], es[i], fill, fx, facc);
    }
    // surviving nodes next to an eliminated one
    if (n == 2) push_upd(cyc[0], es[0], cyc[1], -1, -1);
    for (int i = 0; i < n && n > 2; ++i) {
      if (gone[i]) continue;
      const int il = (i + n - 1) % n, ir = (i + 1) % n;
      int e0 = -1, s0 = -1, e1 = -1, s1 = -1;
      if (gone[il]) { e0 = es[il]; s0 = cyc[il]; }       // edge between cyc[il] and cyc[i]
      if (gone[ir]) { if (e0 < 0) { e0 = es[i]; s0 = cyc[ir]; } else { e1 = es[i]; s1 = cyc[ir]; } }
      if (e0 >= 0) push_upd(cyc[i], e0, s0, e1, s1);
    }
    // next cycle
    std::vector<int> ncyc, nes;
    for (int i = 0; i < n; ++i) {
      if (gone[i]) continue;
      ncyc.push_back(cyc[i]);
      const int ir = (i + 1) % n;
      nes.push_back(gone[ir] ? fillof[ir] : es[i]);
    }
    if (n == 2) { nes.assign(1, -1); }
    if (n == 3) { nes.assign(2, es[2]); }
    lv.nelim = (int)s.elim.size() / CR_EW - lv.eoff; lv.nupd = (int)s.upd.size() / CR_UW - lv.uoff;
    s.lev.push_back(lv);
    cyc.swap(ncyc); es.swap(nes);
  }
  CrLevel lv; lv.eoff = (int)s.elim.size() / CR_EW; lv.nelim = 1; lv.uoff = (int)s.upd.size() / CR_UW; lv.nupd = 0;
  push_elim(cyc[0], -1, -1, -1, -1, -1, 0, 0);
  s.lev.push_back(lv);
  return s;
}

// ------------------------------------------------------------------ device helpers
struct CrDev {               // device copy of the schedule
  const int* elim; const int* upd; const int* orient;
  const int* alist;          // [count] problems to factor / solve (compacted)
};

__device__ __forceinline__ double* cr_edge(const WS& w, const Dims& dm, int b, int slot) {
  const size_t bs = (size_t)dm.dp * dm.dp;
  return (slot < dm.p) ? w.O + ((size_t)b * dm.p + slot) * bs : w.F + ((size_t)b * dm.p + (slot - dm.p)) * bs;
}
#ifdef TMPC_ABLATE
#define TMPC_ABL(dm) (((dm).flags >> 24) & 7)
#else
#define TMPC_ABL(dm) 0
#endif
// float32 copy of edge slot `slot` (same slot numbering as cr_edge), row stride cr_ld32
__device__ __forceinline__ int cr_ld32(const Dims& dm) { return (dm.dp + 31) & ~31; }
__device__ __forceinline__ float* cr_edge32(const WS& w, const Dims& dm, int b, int slot) {
  return w.O32 + ((size_t)b * 2 * dm.p + slot) * (size_t)dm.dp * cr_ld32(dm);
}
__device__ __forceinline__ bool cr_lowp(const WS& w, int b) { return w.O32 != nullptr && w.iprob[(size_t)b * IS + I_LOWP] != 0; }
// XCD-aware work-item id: block i runs on XCD i % 8, so items are dealt to the XCDs in contiguous runs (the tiles of one
// node -- consecutive items -- share their operands through one L2).  gridDim.x is a multiple of 8; -1: no item.
__device__ __forceinline__ int cr_item(int nitems) {
  const int i = blockIdx.x, per = gridDim.x >> 3;
  const int v = (i & 7) * per + (i >> 3);
  return v < nitems ? v : -1;
}
__device__ __forceinline__ void atomic_min_pos(double* addr, double v) {      // v, *addr >= 0: order-preserving as unsigned integers
  atomicMin((unsigned long long*)addr, (unsigned long long)__double_as_longlong(v));
}

// p == 1 / p == 2 preparation (one workgroup per active problem)
template <int NTH>
__device__ __forceinline__ void cr_prep_body(const WS& w, const Dims& dm, int b, int prep) {
  const int dp = dm.dp, tid = threadIdx.x;
  double* D = w.D + (size_t)b * dm.p * dp * dp;
  if (prep == 1) {            // P_{k+1} = P_k: the coupling block folds onto the diagonal  D += C + C'
    const double* E = cr_edge(w, dm, b, 0);
    for (int e = tid; e < dp * dp; e += NTH) { const int i = e / dp, j = e - i * dp; D[e] += E[e] + E[(size_t)j * dp + i]; }
  } else {                    // two edges between the same pair of nodes
    double* E0 = cr_edge(w, dm, b, 0); const double* E1 = cr_edge(w, dm, b, 1);
    for (int e = tid; e < dp * dp; e += NTH) E0[e] += E1[e];
  }
}
__global__ void __launch_bounds__(256) k_cr_prep(WS w, Dims dm, CrDev cr, int prep) { cr_prep_body<256>(w, dm, cr.alist[blockIdx.x], prep); }

constexpr int UPD_DMA_DEPTH = 2;                       // LDS buffers of the LDS-DMA tile GEMM in the batched kernels
// ---- phase 1: Cholesky of the diagonal blocks of this level's eliminated nodes (left-looking on the register-staged core: blocks wider
// than 320 and the no-MFMA debug flag; the product path is k_cr_potrf_dma below)
template <bool USE_MFMA>
__global__ void __launch_bounds__(256, 2) k_cr_potrf(WS w, Dims dm, CrDev cr, int eoff, int nelim, int count) {
  const int it = cr_item(count * nelim);
  if (it < 0) return;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int b = cr.alist[it / nelim];
  const int* er = cr.elim + (size_t)(eoff + it % nelim) * CR_EW;
  const int node = er[CE_NODE], dp = dm.dp;
  const size_t bs = (size_t)dp * dp;
  double* Dk = w.D + ((size_t)b * dm.p + node) * bs;
  double* Li = w.Linv + ((size_t)b * dm.p + node) * dm.nt * TB * TB;
  const double* dref = w.Ddiag + ((size_t)b * dm.p + node) * dp;
  double minr = 1.0;
  const int nbad = wg_block_column<USE_MFMA>(Dk, nullptr, nullptr, Li, dref, dp, lds, &minr);
  if (threadIdx.x == 0) {
    if (nbad) atomicAdd(w.iprob + (size_t)b * IS + I_NSHIFT, nbad);
    if (w.prob && minr < 1.0) atomic_min_pos(w.prob + (size_t)b * PS + P_MINPIV, minr);
  }
}

// ---- phase 2: O_x <- T[x,i] L_i^-T for the (up to) two neighbours, `rs` rows per workgroup
// (left-looking strips on the register-staged core: the path for blocks wider than 320 and for the no-MFMA debug flag)
template <bool USE_MFMA, int NS>
__global__ void __launch_bounds__(256, 2) k_cr_trsm(WS w, Dims dm, CrDev cr, int eoff, int nelim, int count, int rs) {
  const int nstrip = (dm.dp + rs - 1) / rs;
  const int per = 2 * nstrip;
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
  const size_t bs = (size_t)dp * dp;
  const double* Dk = w.D + ((size_t)b * dm.p + node) * bs;
  const double* Li = w.Linv + ((size_t)b * dm.p + node) * dm.nt * TB * TB;
  const int r0 = strip * rs;
  const int rows = (dp - r0 < rs) ? dp - r0 : rs;
  double* X = cr_edge(w, dm, b, slot) + (size_t)r0 * dp;
  int jt = 0;
  for (int j0 = 0; j0 < dp; j0 += TB, ++jt) {
    const int nb = (dp - j0 < TB) ? dp - j0 : TB;
    if (j0 > 0) wg_gemm_nt<USE_MFMA, 2, 2, 2, NS>(X + j0, dp, X, dp, Dk + (size_t)j0 * dp, dp, rows, nb, j0, GM_SUB, false, lds);
    wg_gemm_nt<USE_MFMA, 2, 2, 2, NS>(X + j0, dp, X + j0, dp, Li + (size_t)jt * TB * TB, TB, rows, nb, nb, GM_SET, false, lds);
  }
}

constexpr int TRR_NT = 5;                                  // column tiles whose partial sums fit the registers of k_cr_trsm_dma (dp <= 320)
// wait until at most n (even, wave-uniform) vector-memory operations of this wave are outstanding, and for LDS: the count is a run-time
// value in k_cr_trsm_dma (stores of X and loads of E sit between the slab DMAs), the instruction takes an immediate
__device__ __forceinline__ void vm_wait_le(int n) {
#define TMPC_VMW(K) case K / 2: asm volatile("s_waitcnt vmcnt(" #K ") lgkmcnt(0)" ::: "memory"); break;
  switch (n >> 1) {
    TMPC_VMW(0) TMPC_VMW(2) TMPC_VMW(4) TMPC_VMW(6) TMPC_VMW(8) TMPC_VMW(10) TMPC_VMW(12) TMPC_VMW(14) TMPC_VMW(16) TMPC_VMW(18) TMPC_VMW(20) TMPC_VMW(22)
    default: asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory"); break;
  }
#undef TMPC_VMW
}
// ---- phase 2, register-resident form on the LDS-DMA core: one workgroup owns a 64-row strip of an edge block and walks its column
// tiles once.  X_i = (E_i - sum_{k<i} X_k L_ik') L_ii^-T is evaluated right-looking with the partial sums of ALL column tiles in
// accumulator registers: as soon as X_i is known it is parked in LDS as the A operand and pushed into the accumulators of the
// tiles j > i, while the 64 x 16 slabs of L_ji stream in by buffer_load ... lds.  Every element of E is read once and every element
// of X written once; only L (shared by the 2 x 5 strips of a node, served by L2) is re-read -- the left-looking kernel above
// re-reads its own X strips from memory (3 x the traffic: profiles/r2e_pmc_*.txt).
// Work split over the four waves, chosen so that every step costs every wave the same:
//   * full 64-wide tiles j (all but the last): wave s owns the 16-column strip s of the tile, all four 16-row fragments ("C" layout);
//   * the product with the lower-triangular tile inverse and the last tile (48 wide at d = 300): wave s owns the 16-row fragment s
//     and every 16-column strip ("R" layout) -- by columns the triangle would give the waves 1 : 2 : 3 : 4 slabs and the narrow last
//     tile would leave one wave idle (in-kernel cycle split of the by-columns form: 58 % of wave 0's time outside the MFMA section).
// One run-time loop over the slab steps with a branch per accumulator set: the unrolled form (one copy of the step per tile pair) was
// 97 KB of code, more than the instruction cache.
#ifndef TMPC_TRD_DEPTH
#define TMPC_TRD_DEPTH 2
#endif
constexpr int TRD_DEPTH = TMPC_TRD_DEPTH;                        // B buffers, each one step = two 16-column slabs
constexpr int trd_lds_doubles() { return 4 * 1024 + TRD_DEPTH * 2048; }       // X_i / T_i as the A operand (four 64 x 16 slabs) + the B steps in flight
// POTRF: the same sweep as one block row of the Cholesky factorisation of the block itself (k_cr_potrf_dma): X = the row strip r of D,
// nt = r + 1 tiles, the last one the diagonal tile, whose updates take X_i itself as the second operand (from LDS: no slab stream), and
// which leaves as D_rr - sum_i X_i X_i' for the tile Cholesky instead of being multiplied by an inverse.
template <bool POTRF>
__device__ __forceinline__ void trd_strip(double* X, const double* Dk, const double* Li, int rows, int nt, int dp, int it, double* lds, float* X32 = nullptr, int ld32 = 0) {
  // X32 (wave-uniform, or nullptr): every finished strip X_i is ALSO stored as float32 (row stride ld32) -- the operand copy of the single-precision updates
  constexpr int FR = 4, RS = 16 * FR, ASL = RS * 16, DP = TRD_DEPTH;          // fragments and rows per strip, doubles per A slab
  int tid = threadIdx.x;
  if (POTRF) asm volatile("" : "+v"(tid));                  // per-lane constants are rebuilt for every block row instead of living (spilled) across the tile Cholesky
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: the slab walk below branches on scalars
  const int ws = (wv + it) & 3, wc0 = ws * 16;              // this wave's column strip ("C") or row fragment ("R")
  const int fk = lane >> 4, fq = (lane >> 2) & 3, fj = lane & 3;
  const int i1 = (rows + 15) >> 4;                          // 16-row fragments of the strip
  const int nlast = nt - 1, nbl = (dp - 64 * nlast < 64) ? dp - 64 * nlast : 64, ncl = nbl >> 4;      // the last tile: index, width, 16-column strips
  const int ntd = POTRF ? nt - 1 : nt;                      // tiles with a triangular product and a slab stream
  const bool rw = ws < i1;                                  // this wave has a row fragment
  double* At = lds;
  double* Bs = lds + 4 * ASL;
  unsigned voT[2], voL[2];                                  // DMA: wave wv moves rows 16 wv .. + 15 of a B slab, two pieces of 8 rows
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = wv * 16 + 8 * h + (lane >> 3), c = lane & 7;
    voT[h] = (unsigned)(row * TB + 2 * (c ^ dma_sw(row))) * 8u;
    voL[h] = (unsigned)(row * dp + 2 * (c ^ dma_sw(row))) * 8u;
  }
  int oa[2], ob[4][2];                                      // fragment offsets of A rows 4 fq + fj (+ 256 per fragment) and B rows 4 fj + e (+ 256 per strip)
  {
    const int swa = dma_sw(4 * fq + fj);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) oa[hh] = (4 * fq + fj) * 16 + 2 * ((4 * hh + fk) ^ swa);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int swb = dma_sw(4 * fj + e);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) ob[e][hh] = (4 * fj + e) * 16 + 2 * ((4 * hh + fk) ^ swb);
    }
  }
  // Four register sets of 16 doubles hold (partial sums) - E of the column tiles: acc1 .. acc3 the full tiles 1 .. 3 (C layout), acc0
  // tile 0 and, once X_0 is out, the last tile (R layout).  E is loaded negated and the products X_k L_jk' add up on it, so when the
  // block row of tile i - 1 is done set i holds -T_i: it goes to LDS negated as the A operand, is cleared and collects T_i L_ii^-T
  // (R layout) -- X_i, stored and parked for the updates.  E_1 .. E_3 are read at the start, E_last after X_0: no global load sits
  // on the path between two tiles.
  double acc0[4][4], acc1[4][4], acc2[4][4], acc3[4][4];
  int vmtot = 0, vmk0 = 0, vmk1 = 0, vmk2 = 0;             // vector-memory operations issued so far; the count right after the DMA of step s (slot s % DP)
#define TRD_NB(T) ((dp - 64 * (T) < 64) ? dp - 64 * (T) : 64)
  // C layout, lane <-> memory: rows 16 i + 4 fq + fk of the four fragments, columns wc0 + 4 fj .. + 3 of the tile
#define TRD_LOAD_C(I, SET)                                                                                  \
  {                                                                                                         \
    const bool on_ = wc0 < TRD_NB(I);                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
      double2_t u0 = (double2_t){0.0, 0.0}, u1 = u0;                                                        \
      if (on_ && i < i1) {                                                                                  \
        gcptr2 cp = (gcptr2)(X + (size_t)(16 * i + 4 * fq + fk) * dp + 64 * (I) + wc0 + 4 * fj);            \
        u0 = cp[0]; u1 = cp[1];                                                                             \
      }                                                                                                     \
      SET[i][0] = -u0[0]; SET[i][1] = -u0[1]; SET[i][2] = -u1[0]; SET[i][3] = -u1[1];                       \
    }                                                                                                       \
    if (on_) vmtot += 2 * i1;                                                                               \
  }
  // R layout: rows wc0 + 4 fq + fk, columns 16 c + 4 fj .. + 3 of the strips c < NC of the tile
#define TRD_LOAD_R(I, NC, SET)                                                                              \
  {                                                                                                         \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                          \
      double2_t u0 = (double2_t){0.0, 0.0}, u1 = u0;                                                        \
      if (rw && c < (NC)) {                                                                                 \
        gcptr2 cp = (gcptr2)(X + (size_t)(wc0 + 4 * fq + fk) * dp + 64 * (I) + 16 * c + 4 * fj);            \
        u0 = cp[0]; u1 = cp[1];                                                                             \
      }                                                                                                     \
      SET[c][0] = -u0[0]; SET[c][1] = -u0[1]; SET[c][2] = -u1[0]; SET[c][3] = -u1[1];                       \
    }                                                                                                       \
    if (rw) vmtot += 2 * (NC);                                                                              \
  }
#define TRD_STORE_R(I, NC, SET, SG)                                                                         \
  if (rw) {                                                                                                 \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                          \
      if (c < (NC)) {                                                                                       \
        typedef double2_t __attribute__((address_space(1)))* gptr2;                                         \
        gptr2 cp = (gptr2)(X + (size_t)(wc0 + 4 * fq + fk) * dp + 64 * (I) + 16 * c + 4 * fj);              \
        if (X32) {      /* an iteration in single precision reads ONLY the float32 copy of this block (updates, substitutions): the fp64 store is skipped */ \
          typedef float4_t __attribute__((address_space(1)))* gptr4;                                        \
          *(gptr4)(X32 + (size_t)(wc0 + 4 * fq + fk) * ld32 + 64 * (I) + 16 * c + 4 * fj) =                 \
              (float4_t){(float)(SG SET[c][0]), (float)(SG SET[c][1]), (float)(SG SET[c][2]), (float)(SG SET[c][3])}; \
        } else {                                                                                            \
          cp[0] = (double2_t){SG SET[c][0], SG SET[c][1]}; cp[1] = (double2_t){SG SET[c][2], SG SET[c][3]}; \
        }                                                                                                   \
      }                                                                                                     \
    }                                                                                                       \
    vmtot += (X32 ? 1 : 2) * (NC);                                                                          \
  }
  // (+-) set -> A operand.  C: slab = this wave's strip, K pairs 2 fj and 2 fj + 1 of every row; R: this wave's rows of the slabs c < NC
#define TRD_PARK_C(SET)      /* negated */                                                                  \
  {                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
      const int row = 16 * i + 4 * fq + fk, sw_ = dma_sw(row);                                              \
      double* q = At + ws * ASL + row * 16;                                                                \
      *(double2_t*)(q + 2 * ((2 * fj) ^ sw_)) = (double2_t){-SET[i][0], -SET[i][1]};                        \
      *(double2_t*)(q + 2 * ((2 * fj + 1) ^ sw_)) = (double2_t){-SET[i][2], -SET[i][3]};                    \
    }                                                                                                       \
  }
#define TRD_PARK_R(NC, SET, SG)                                                                             \
  if (rw) {                                                                                                 \
    const int row = wc0 + 4 * fq + fk, sw_ = dma_sw(row);                                                   \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                          \
      if (c < (NC)) {                                                                                       \
        double* q = At + c * ASL + row * 16;                                                               \
        *(double2_t*)(q + 2 * ((2 * fj) ^ sw_)) = (double2_t){SG SET[c][0], SG SET[c][1]};                  \
        *(double2_t*)(q + 2 * ((2 * fj + 1) ^ sw_)) = (double2_t){SG SET[c][2], SG SET[c][3]};              \
      }                                                                                                     \
    }                                                                                                       \
  }
#define TRD_ZERO(SET)                                                                                       \
  {                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                            \
      _Pragma("unroll") for (int c = 0; c < 4; ++c) SET[i][c] = 0.0;                                        \
  }
  // B stream: for every column tile i the slabs of Linv_i (K = nb_i), then those of L_ji, j > i (K = 64).  (ii, ij, is) = next slab to issue.
  int ii = 0, ij = 0, is = 0, nissued = 0, ndone = 0;
#define TRD_ISSUE()          /* one step = the slabs 2 is and 2 is + 1 (if the K range has it) of the pair (ii, ij) */ \
  {                                                                                                         \
    if (ii < ntd) {                                                                                         \
      double* dst_ = Bs + (nissued % DP) * 2048 + wv * 256;                                                 \
      const int nsl_ = (ij == ii) ? (TRD_NB(ii) >> 4) : 4;                                                  \
      const bool two_ = 2 * is + 1 < nsl_;                                                                  \
      if (ij == ii) {                                                                                       \
        const int nb_ = TRD_NB(ii);                                                                         \
        const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)(Li + (size_t)ii * TB * TB), 0, (int)(((unsigned)(nb_ - 1) * TB + (unsigned)nb_) * 8u), 0x00020000); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_vptr)(dst_), 16, voT[0], is * 256, 0, 0);        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_vptr)(dst_ + 128), 16, voT[1], is * 256, 0, 0);  \
        if (two_) {                                                                                         \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_vptr)(dst_ + 1024), 16, voT[0], is * 256 + 128, 0, 0);       \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_vptr)(dst_ + 1024 + 128), 16, voT[1], is * 256 + 128, 0, 0); \
        }                                                                                                   \
      } else {                                                                                              \
        const int nb_ = TRD_NB(ij);                                                                         \
        const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)(Dk + (size_t)(64 * ij) * dp + 64 * ii), 0, (int)(((unsigned)(nb_ - 1) * (unsigned)dp + 64u) * 8u), 0x00020000); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_vptr)(dst_), 16, voL[0], is * 256, 0, 0);        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_vptr)(dst_ + 128), 16, voL[1], is * 256, 0, 0);  \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_vptr)(dst_ + 1024), 16, voL[0], is * 256 + 128, 0, 0);         \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_vptr)(dst_ + 1024 + 128), 16, voL[1], is * 256 + 128, 0, 0);   \
      }                                                                                                     \
      vmtot += (ij == ii && !two_) ? 2 : 4;                                                                 \
      { const int k_ = nissued % DP; if (k_ == 0) vmk0 = vmtot; else if (k_ == 1) vmk1 = vmtot; else vmk2 = vmtot; }  \
      ++nissued;                                                                                            \
      if (++is == ((nsl_ + 1) >> 1)) { is = 0; if (++ij >= ntd) { ++ii; ij = ii; } }                        \
    }                                                                                                       \
  }
  // one 16-deep slab into TGT.  C: the four fragments x this wave's strip; R: this wave's fragment x the strips C0 <= c < NC.
  // No run-time condition inside (conditional updates of single accumulators made the register allocator keep copies of the sets and
  // spill; a reload from scratch in turn forces vmcnt(0), i.e. waits for the slab DMA just issued): the fragments beyond the rows of a
  // short last strip are computed and never stored (a row of the result depends on the same row of the operand only), and the strip
  // ranges of R are literals -- the dispatch on the run-time shapes sits outside, one uniform branch per step.
#define TRD_MMA_C(TGT, AS, BS)                                                                              \
  _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                                        \
    double2_t a_[4], b_[4];                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) a_[i] = *(const double2_t*)((AS) + oa[hh] + i * 256);      \
    _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) b_[cb] = *(const double2_t*)((BS) + wc0 * 16 + ob[cb][hh]); \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
      _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) {                                                    \
        TGT[i][cb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a_[i][0], b_[cb][0], TGT[i][cb], 0, 0, 0);          \
        TGT[i][cb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a_[i][1], b_[cb][1], TGT[i][cb], 0, 0, 0);          \
      }                                                                                                     \
    }                                                                                                       \
  }
#define TRD_MMA_R(TGT, AS, BS, C0, NC)                                                                      \
  _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                                        \
    const double2_t a_ = *(const double2_t*)((AS) + oa[hh] + wc0 * 16);                                     \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                          \
      if (c >= (C0) && c < (NC)) {                                                                          \
        double2_t b_[4];                                                                                     \
        _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) b_[cb] = *(const double2_t*)((BS) + c * 256 + ob[cb][hh]); \
        _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) {                                                  \
          TGT[c][cb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a_[0], b_[cb][0], TGT[c][cb], 0, 0, 0);           \
          TGT[c][cb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a_[1], b_[cb][1], TGT[c][cb], 0, 0, 0);           \
        }                                                                                                   \
      }                                                                                                     \
    }                                                                                                       \
  }
  // the two slabs of a step into one set: C; the triangular product of a full tile, first and second step; the last tile, N strips wide
#define TRD_STEP_C(TGT) { TRD_MMA_C(TGT, A0, Bc) TRD_MMA_C(TGT, A0 + ASL, Bc + 1024) }
#define TRD_STEP_D0(TGT) { TRD_MMA_R(TGT, At, Bc, 0, 4) TRD_MMA_R(TGT, At + ASL, Bc + 1024, 1, 4) }
#define TRD_STEP_D1(TGT) { TRD_MMA_R(TGT, At + 2 * ASL, Bc, 2, 4) TRD_MMA_R(TGT, At + 3 * ASL, Bc + 1024, 3, 4) }
#define TRD_STEP_RL(N) { TRD_MMA_R(acc0, A0, Bc, 0, N) TRD_MMA_R(acc0, A0 + ASL, Bc + 1024, 0, N) }

  __syncthreads();
  TMPC_TC0()
  TRD_ISSUE()
  if (DP > 2) TRD_ISSUE()
  TRD_LOAD_C(0, acc0)
  if (nt > 2) TRD_LOAD_C(1, acc1) else TRD_ZERO(acc1)
  if (nt > 3) TRD_LOAD_C(2, acc2) else TRD_ZERO(acc2)
  if (nt > 4) TRD_LOAD_C(3, acc3) else TRD_ZERO(acc3)
  TRD_PARK_C(acc0)                                          // T_0 = E_0 (the set holds -E_0)
  TRD_ZERO(acc0)
  // a step: wait for its slabs, let every wave arrive (the buffer of the step before is free then), start the DMA of the next step
#define TRD_STEP_BEGIN()                                                                                    \
    { const int k_ = ndone % DP; vm_wait_le(vmtot - (k_ == 0 ? vmk0 : k_ == 1 ? vmk1 : vmk2)); }            \
    TMPC_TC(2, 0)                                                                                           \
    __builtin_amdgcn_s_barrier();                                                                           \
    TMPC_TC(2, 1)                                                                                           \
    TRD_ISSUE()                                                                                             \
    TMPC_TC(2, 2)                                                                                           \
    const double* Bc = Bs + (ndone % DP) * 2048;                                                            \
    ++ndone;
  // the two steps of the update of a full tile (C layout) / of the last tile (R layout, set 0) by X_i
#define TRD_PAIR_C(TGT)                                                                                     \
  _Pragma("unroll 1") for (int cs = 0; cs < 2; ++cs) {                                                                          \
    TRD_STEP_BEGIN()                                                                                        \
    const double* A0 = At + 2 * cs * ASL;                                                                   \
    TRD_STEP_C(TGT)                                                                                         \
    TMPC_TC(2, 3)                                                                                           \
  }
#define TRD_PAIR_RL(N)                                                                                      \
  if (POTRF) {                       /* the diagonal tile: X_i X_i', both operands in LDS already */          \
    __builtin_amdgcn_s_barrier();                                                                           \
    if (rw) {                                                                                               \
      _Pragma("unroll 1") for (int sl = 0; sl < 4; ++sl) { const double* As_ = At + sl * ASL; TRD_MMA_R(acc0, As_, As_, 0, N) } \
    }                                                                                                       \
  } else {                                                                                                  \
    _Pragma("unroll 1") for (int cs = 0; cs < 2; ++cs) {                                                    \
      TRD_STEP_BEGIN()                                                                                      \
      const double* A0 = At + 2 * cs * ASL;                                                                 \
      if (rw) TRD_STEP_RL(N)                                                                                \
      TMPC_TC(2, 3)                                                                                         \
    }                                                                                                       \
  }
  // the triangular product of a full tile in set TGT, X_i out and (negated) into LDS
#define TRD_DIAG(TGT)                                                                                       \
  {                                                                                                         \
    { TRD_STEP_BEGIN() if (rw) TRD_STEP_D0(TGT) TMPC_TC(2, 3) }                                             \
    { TRD_STEP_BEGIN() if (rw) TRD_STEP_D1(TGT) TMPC_TC(2, 3) }                                             \
    TRD_STORE_R(ci, 4, TGT, +)                                                                              \
    if (ci + 1 < nt) {                                                                                      \
      __builtin_amdgcn_s_barrier();                         /* every wave is done with T_i */               \
      TRD_PARK_R(4, TGT, +)                                                                                 \
    }                                                                                                       \
  }
  for (int ci = 0; ci < ntd; ++ci) {
    const int nci = TRD_NB(ci) >> 4;
    // T_i times the lower-triangular inverse: strip c needs the slabs <= c
    if (nci == 4) {
      if (ci == 0 || ci == nlast) TRD_DIAG(acc0) else if (ci == 1) TRD_DIAG(acc1) else if (ci == 2) TRD_DIAG(acc2) else TRD_DIAG(acc3)
    } else {                                                // a narrow last tile (always set 0): run-time strip range
      for (int cs = 0; cs < ((nci + 1) >> 1); ++cs) {
        TRD_STEP_BEGIN()
        const double* A0 = At + 2 * cs * ASL;
        if (rw) {
          TRD_MMA_R(acc0, A0, Bc, 2 * cs, nci)
          if (2 * cs + 1 < nci) TRD_MMA_R(acc0, A0 + ASL, Bc + 1024, 2 * cs + 1, nci)
        }
        TMPC_TC(2, 3)
      }
      TRD_STORE_R(ci, nci, acc0, +)
    }
    if (ci == 0 && nt > 1) TRD_LOAD_R(nlast, ncl, acc0)     // E_last: lands during the updates of the tiles before it
    if (ci + 1 >= nt) break;
    // X_i into the tiles to its right
    if (ci < 1 && 1 < nlast) TRD_PAIR_C(acc1)
    if (ci < 2 && 2 < nlast) TRD_PAIR_C(acc2)
    if (ci < 3 && 3 < nlast) TRD_PAIR_C(acc3)
    if (ncl == 3) TRD_PAIR_RL(3) else if (ncl == 4) TRD_PAIR_RL(4) else if (ncl == 2) TRD_PAIR_RL(2) else TRD_PAIR_RL(1)
    // block row i finished: set i + 1 holds T_{i+1}
    const int t = ci + 1;
    __builtin_amdgcn_s_barrier();                           // every wave is done with X_i
    if (POTRF && t == nlast) break;                         // the diagonal tile stays in its set
    if (t == nlast) { TRD_PARK_R(ncl, acc0, -) TRD_ZERO(acc0) }
    else if (t == 1) { TRD_PARK_C(acc1) TRD_ZERO(acc1) }
    else if (t == 2) { TRD_PARK_C(acc2) TRD_ZERO(acc2) }
    else { TRD_PARK_C(acc3) TRD_ZERO(acc3) }
    TMPC_TC(2, 4)
  }
  if (POTRF) {                                              // D_rr - sum_i X_i X_i' (set 0 holds its negative) straight into the LDS image of the tile Cholesky
    // (round 3: it used to go back to memory and be re-read by wg_potrf_inv one barrier later -- a store-to-load round trip through L2 per
    // tile, ~10 k of the ~95 k cycles of a tile).  Every wave has passed the barrier behind the last product that read the A operand, the
    // image (64 x LDP from the start of the LDS block) may overwrite it.
    if (rw) {
      double* Simg = lds;
      const int row = wc0 + 4 * fq + fk;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c < ncl) {
#pragma unroll
          for (int e = 0; e < 4; ++e) Simg[row * LDP + 16 * c + 4 * fj + e] = -acc0[c][e];
        }
      }
    }
  }
#undef TRD_DIAG
#undef TRD_PAIR_RL
#undef TRD_PAIR_C
#undef TRD_STEP_BEGIN
#undef TRD_STEP_RL
#undef TRD_STEP_D1
#undef TRD_STEP_D0
#undef TRD_STEP_C
#undef TRD_MMA_R
#undef TRD_MMA_C
#undef TRD_ISSUE
#undef TRD_ZERO
#undef TRD_PARK_R
#undef TRD_PARK_C
#undef TRD_STORE_R
#undef TRD_LOAD_R
#undef TRD_LOAD_C
#undef TRD_NB
}

__global__ void __launch_bounds__(256, 2) k_cr_trsm_dma(WS w, Dims dm, CrDev cr, int eoff, int nelim, int count) {
  const int dp = dm.dp;
  const int nst = (dp + 63) / 64;                           // strips per edge
  const int per = 2 * nst;
  const int it = cr_item(count * nelim * per);
  if (it < 0) return;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int g = it / per, r = it - g * per;
  const int b = cr.alist[g / nelim];
  const int* er = cr.elim + (size_t)(eoff + g % nelim) * CR_EW;
  const int which = r / nst, strip = r - which * nst;
  const int slot = which ? er[CE_EB] : er[CE_EA];
  if (slot < 0) return;
  const int node = er[CE_NODE];
  const double* Dk = w.D + ((size_t)b * dm.p + node) * (size_t)dp * dp;
  const double* Li = w.Linv + ((size_t)b * dm.p + node) * dm.nt * TB * TB;
  const int r0 = strip * 64;
  const bool lp = cr_lowp(w, b);
  const int ld32 = cr_ld32(dm);
  trd_strip<false>(cr_edge(w, dm, b, slot) + (size_t)r0 * dp, Dk, Li, (dp - r0 < 64) ? dp - r0 : 64, dm.nt, dp, it, lds,
                   lp ? cr_edge32(w, dm, b, slot) + (size_t)r0 * ld32 : nullptr, ld32);
}

// ---- phase 1 on the same sweep: block row r of the factor is the triangular solve of the row strip r of D against the rows above it,
// its diagonal tile D_rr - sum_i X_i X_i' comes out of the same accumulators and goes through the 64 x 64 tile Cholesky + inverse.
// One workgroup per block, the block rows one after the other (each needs the finished rows above).
constexpr int potrf_dma_lds_doubles() { return FACT_LDS_DOUBLES > trd_lds_doubles() ? FACT_LDS_DOUBLES : trd_lds_doubles(); }
__global__ void __launch_bounds__(256, 2) k_cr_potrf_dma(WS w, Dims dm, CrDev cr, int eoff, int nelim, int count) {
  const int it = cr_item(count * nelim);
  if (it < 0) return;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int b = cr.alist[it / nelim];
  const int* er = cr.elim + (size_t)(eoff + it % nelim) * CR_EW;
  const int node = er[CE_NODE], dp = dm.dp;
  double* Dk = w.D + ((size_t)b * dm.p + node) * (size_t)dp * dp;
  double* Li = w.Linv + ((size_t)b * dm.p + node) * dm.nt * TB * TB;
  const double* dref = w.Ddiag + ((size_t)b * dm.p + node) * dp;
  double* stat = lds + FACT_LDS_DOUBLES - 8;                // frozen pivots / smallest pivot ratio of the block so far (kept in LDS: the sweep needs every register)
  if (threadIdx.x == 0) { stat[0] = 0.0; stat[1] = 1.0; }
  for (int r = 0; r < dm.nt; ++r) {
    const int r0 = 64 * r, nb = (dp - r0 < 64) ? dp - r0 : 64;
    if (r > 0) {
      trd_strip<true>(Dk + (size_t)r0 * dp, Dk, Li, nb, r + 1, dp, it, lds);
      __syncthreads();                                      // the tile is in its LDS image, every wave is done with the slabs in LDS
    }
    wg_potrf_inv(Dk + (size_t)r0 * dp + r0, dp, Li + (size_t)r * TB * TB, dref + r0, nb, lds, nullptr, stat, (it >> 5) & 3, r > 0);
  }
  if (threadIdx.x == 0) {
    const int nbad = (int)stat[0];
    const double minr = stat[1];
    if (nbad) atomicAdd(w.iprob + (size_t)b * IS + I_NSHIFT, nbad);
    if (w.prob && minr < 1.0) atomic_min_pos(w.prob + (size_t)b * PS + P_MINPIV, minr);
  }
}

// ---- phase 3: symmetric updates of the surviving neighbours and the fill edges, one 64 x 64 output tile per workgroup
// (register-staged core of tmpc_factor.h: the fallback / debug path; the product path is k_cr_update_dma below)
template <bool USE_MFMA, int NS>
__global__ void __launch_bounds__(256, 2) k_cr_update(WS w, Dims dm, CrDev cr, int eoff, int nelim, int uoff, int nupd, int count, int mt) {
  const int nm = (dm.dp + mt - 1) / mt;
  const int ntl = nm * (nm + 1) / 2, ntf = nm * nm;
  const int per = nupd * ntl + nelim * ntf;
  const int it = cr_item(count * per);
  if (it < 0) return;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int b = cr.alist[it / per];
  int r = it % per;
  const int dp = dm.dp;
  const size_t bs = (size_t)dp * dp;
  // Item order inside a problem: node by node along the cycle -- the fill tiles of eliminated node j, then the update tiles of
  // the surviving node that follows it (update record j + 1): the O blocks of node j feed both, and the second operand of that
  // update is the first O block of node j + 1, whose fill tiles come next.  Workgroups of one XCD run consecutive items, so each
  // O block is fetched from HBM once and then served by that L2 (item order by kind re-fetched every block 3-4 times).
  const int slot_items = ntf + ntl;
  int j = r / slot_items, rr = r - j * slot_items;
  bool is_upd;
  {
    // slots j < min(nelim, nupd) hold ntf fill items and ntl update items; the longer list continues alone
    const int nmin = nelim < nupd ? nelim : nupd;
    const int full = nmin * slot_items;
    if (r < full) { is_upd = rr >= ntf; if (is_upd) rr -= ntf; }
    else if (nelim > nupd) { const int q = r - full; j = nmin + q / ntf; rr = q % ntf; is_upd = false; }
    else { const int q = r - full; j = nmin + q / ntl; rr = q % ntl; is_upd = true; }
  }
  if (is_upd) {
    const int* ur = cr.upd + (size_t)(uoff + (j + 1) % nupd) * CR_UW;
    int t = rr, tm = 0;
    while (t > tm) { t -= tm + 1; ++tm; }               // t -> (tm, tn), tn <= tm
    const int tn = t;
    const int m0 = tm * mt, n0 = tn * mt;
    const int M = (dp - m0 < mt) ? dp - m0 : mt, N = (dp - n0 < mt) ? dp - n0 : mt;
    double* C = w.D + ((size_t)b * dm.p + ur[CU_NODE]) * bs + (size_t)m0 * dp + n0;
    const double* O0 = cr_edge(w, dm, b, ur[CU_E0]);
    wg_gemm_nt<USE_MFMA, 2, 2, 2, NS>(C, dp, O0 + (size_t)m0 * dp, dp, O0 + (size_t)n0 * dp, dp, M, N, dp, GM_SUB, tm == tn, lds);
    if (ur[CU_E1] >= 0) {
      const double* O1 = cr_edge(w, dm, b, ur[CU_E1]);
      wg_gemm_nt<USE_MFMA, 2, 2, 2, NS>(C, dp, O1 + (size_t)m0 * dp, dp, O1 + (size_t)n0 * dp, dp, M, N, dp, GM_SUB, tm == tn, lds);
    }
  } else {
    const int* er = cr.elim + (size_t)(eoff + j) * CR_EW;
    if (er[CE_FILL] < 0) return;
    const int t = rr;
    const int tm = t / nm, tn = t - tm * nm;
    const int m0 = tm * mt, n0 = tn * mt;
    const int M = (dp - m0 < mt) ? dp - m0 : mt, N = (dp - n0 < mt) ? dp - n0 : mt;
    const double* Ox = cr_edge(w, dm, b, er[CE_FX] ? er[CE_EB] : er[CE_EA]);
    const double* Oy = cr_edge(w, dm, b, er[CE_FX] ? er[CE_EA] : er[CE_EB]);
    double* C = cr_edge(w, dm, b, er[CE_FILL]) + (size_t)m0 * dp + n0;
    wg_gemm_nt<USE_MFMA, 2, 2, 2, NS>(C, dp, Ox + (size_t)m0 * dp, dp, Oy + (size_t)n0 * dp, dp, M, N, dp, er[CE_FACC] ? GM_SUB : GM_NEG, false, lds);
  }
}

// which right-hand sides a problem solves in this pass: pass 1 (main phase only) [rhs | u_tau | u_alpha] in W3; pass 2 the
// corrector rhs alone in Z (main phase) or all three (centering: no predictor).  0: nothing to do.
__device__ __forceinline__ int cr_nc(const WS& w, int b, int pass) {
  const int phase = w.iprob[(size_t)b * IS + I_PHASE];
  if (phase == PH_DONE || (pass == 1 && phase != PH_MAIN)) return 0;
  return ((pass == 1) || (phase != PH_MAIN && !w.iprob[(size_t)b * IS + I_CHORD])) ? 3 : 1;     // chord step: rhs only
}
__device__ __forceinline__ double* cr_rhs(const WS& w, const Dims& dm, int b, int node, int nc) {
  return ((nc == 3) ? w.W3 : w.Z) + ((size_t)b * dm.p + node) * dm.dp * nc;
}

// The same items on the LDS-DMA core (tmpc_gemm_dma.h): 64 x 64 tiles, the two edges of a doubly updated node as ONE K stream.
// fuse != 0: the forward substitution step of pass 1, z_s -= O_s z_i (k_cr_fwd_off), rides along in the first column tile of every row tile
// of a surviving node -- the O blocks stream through this kernel anyway, so pass 1 reads them once less (the right-hand sides of pass 1
// and k_cr_fwd_diag of this level must have run before; problems without a predictor pass are left alone).
__global__ void __launch_bounds__(256, 4) k_cr_update_dma(WS w, Dims dm, CrDev cr, int eoff, int nelim, int uoff, int nupd, int count, int fuse) {
  const int dp = dm.dp;
  const int nm = (dp + 63) / 64;
  const int ntl = nm * (nm + 1) / 2, ntf = nm * nm;
  const int per = nupd * ntl + nelim * ntf;
  const int it = cr_item(count * per);
  if (it < 0) return;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int b = cr.alist[it / per];
  if (cr_lowp(w, b)) return;                               // this problem's tiles run in single precision: k_cr_update_dma_f32
  const int r = it % per;
  const size_t bs = (size_t)dp * dp;
  const int slot_items = ntf + ntl;
  int j = r / slot_items, rr = r - j * slot_items;
  bool is_upd;
  {
    const int nmin = nelim < nupd ? nelim : nupd;
    const int full = nmin * slot_items;
    if (r < full) { is_upd = rr >= ntf; if (is_upd) rr -= ntf; }
    else if (nelim > nupd) { const int q = r - full; j = nmin + q / ntf; rr = q % ntf; is_upd = false; }
    else { const int q = r - full; j = nmin + q / ntl; rr = q % ntl; is_upd = true; }
  }
  if (is_upd) {
    const int* ur = cr.upd + (size_t)(uoff + (j + 1) % nupd) * CR_UW;
    int t = rr, tm = 0;
    while (t > tm) { t -= tm + 1; ++tm; }
    const int tn = t;
    const int m0 = tm * 64, n0 = tn * 64;
    const int M = (dp - m0 < 64) ? dp - m0 : 64, N = (dp - n0 < 64) ? dp - n0 : 64;
    double* C = w.D + ((size_t)b * dm.p + ur[CU_NODE]) * bs + (size_t)m0 * dp + n0;
    const double* O0 = cr_edge(w, dm, b, ur[CU_E0]);
    const double* O1 = ur[CU_E1] >= 0 ? cr_edge(w, dm, b, ur[CU_E1]) : nullptr;
    const bool fz = fuse && tn == 0 && cr_nc(w, b, 1) == 3;
    const double* z0 = fz ? cr_rhs(w, dm, b, ur[CU_S0], 3) : nullptr;
    const double* z1 = (fz && O1) ? cr_rhs(w, dm, b, ur[CU_S1], 3) : nullptr;
    double* yz = fz ? cr_rhs(w, dm, b, ur[CU_NODE], 3) + (size_t)m0 * 3 : nullptr;
    if (fz) wg_tile_dma<UPD_DMA_DEPTH, true>(C, dp, O0 + (size_t)m0 * dp, O0 + (size_t)n0 * dp, O1 ? O1 + (size_t)m0 * dp : nullptr, O1 ? O1 + (size_t)n0 * dp : nullptr,
                                             dp, M, N, dp, GM_SUB, tm == tn ? 0 : GM_NOTRI, it, lds, 0, z0, z1, yz, 3);
    else wg_tile_dma<UPD_DMA_DEPTH>(C, dp, O0 + (size_t)m0 * dp, O0 + (size_t)n0 * dp, O1 ? O1 + (size_t)m0 * dp : nullptr, O1 ? O1 + (size_t)n0 * dp : nullptr,
                                    dp, M, N, dp, GM_SUB, tm == tn ? 0 : GM_NOTRI, it, lds);
  } else {
    const int* er = cr.elim + (size_t)(eoff + j) * CR_EW;
    if (er[CE_FILL] < 0) return;
    const int tm = rr / nm, tn = rr - tm * nm;
    const int m0 = tm * 64, n0 = tn * 64;
    const int M = (dp - m0 < 64) ? dp - m0 : 64, N = (dp - n0 < 64) ? dp - n0 : 64;
    const double* Ox = cr_edge(w, dm, b, er[CE_FX] ? er[CE_EB] : er[CE_EA]);
    const double* Oy = cr_edge(w, dm, b, er[CE_FX] ? er[CE_EA] : er[CE_EB]);
    double* C = cr_edge(w, dm, b, er[CE_FILL]) + (size_t)m0 * dp + n0;
    wg_tile_dma<UPD_DMA_DEPTH>(C, dp, Ox + (size_t)m0 * dp, Oy + (size_t)n0 * dp, nullptr, nullptr, dp, M, N, dp, er[CE_FACC] ? GM_SUB : GM_NEG, GM_NOTRI, it, lds);
  }
}

// The same items for the problems whose Schur-complement updates run in SINGLE precision this iteration (I_LOWP, set by k_ctrl_a while mu / kappa > Opts::lowp_switch in
// the first LOWP_ITERS iterations): float32 copies of the O blocks (written next to the fp64 ones by k_cr_trsm_dma), float32 accumulation on v_mfma_f32_16x16x4f32
// (wg_tile_dma_f32), the result subtracted from / stored into the fp64 blocks.  No fused right-hand sides: k_cr_fwd_off runs for these problems (cr_factor).
__global__ void __launch_bounds__(256, 5) k_cr_update_dma_f32(WS w, Dims dm, CrDev cr, int eoff, int nelim, int uoff, int nupd, int count) {
  const int dp = dm.dp;
  const int nm = (dp + 63) / 64;
  const int ntl = nm * (nm + 1) / 2, ntf = nm * nm;
  const int per = nupd * ntl + nelim * ntf;
  const int it = cr_item(count * per);
  if (it < 0) return;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int b = cr.alist[it / per];
  if (!cr_lowp(w, b)) return;
  const int r = it % per;
  const size_t bs = (size_t)dp * dp;
  const int slot_items = ntf + ntl, ld32 = cr_ld32(dm);
  int j = r / slot_items, rr = r - j * slot_items;
  bool is_upd;
  {
    const int nmin = nelim < nupd ? nelim : nupd;
    const int full = nmin * slot_items;
    if (r < full) { is_upd = rr >= ntf; if (is_upd) rr -= ntf; }
    else if (nelim > nupd) { const int q = r - full; j = nmin + q / ntf; rr = q % ntf; is_upd = false; }
    else { const int q = r - full; j = nmin + q / ntl; rr = q % ntl; is_upd = true; }
  }
  if (is_upd) {
    const int* ur = cr.upd + (size_t)(uoff + (j + 1) % nupd) * CR_UW;
    int t = rr, tm = 0;
    while (t > tm) { t -= tm + 1; ++tm; }
    const int tn = t;
    const int m0 = tm * 64, n0 = tn * 64;
    const int M = (dp - m0 < 64) ? dp - m0 : 64, N = (dp - n0 < 64) ? dp - n0 : 64;
    double* C = w.D + ((size_t)b * dm.p + ur[CU_NODE]) * bs + (size_t)m0 * dp + n0;
    const float* P0 = cr_edge32(w, dm, b, ur[CU_E0]);
    const float* P1 = ur[CU_E1] >= 0 ? cr_edge32(w, dm, b, ur[CU_E1]) : nullptr;
    wg_tile_dma_f32<UPD_DMA_DEPTH>(C, dp, P0 + (size_t)m0 * ld32, P0 + (size_t)n0 * ld32, P1 ? P1 + (size_t)m0 * ld32 : nullptr, P1 ? P1 + (size_t)n0 * ld32 : nullptr,
                                   ld32, M, N, ld32, GM_SUB, tm == tn ? 0 : GM_NOTRI, it, lds, TMPC_ABL(dm));
  } else {
    const int* er = cr.elim + (size_t)(eoff + j) * CR_EW;
    if (er[CE_FILL] < 0) return;
    const int tm = rr / nm, tn = rr - tm * nm;
    const int m0 = tm * 64, n0 = tn * 64;
    const int M = (dp - m0 < 64) ? dp - m0 : 64, N = (dp - n0 < 64) ? dp - n0 : 64;
    const float* Px = cr_edge32(w, dm, b, er[CE_FX] ? er[CE_EB] : er[CE_EA]);
    const float* Py = cr_edge32(w, dm, b, er[CE_FX] ? er[CE_EA] : er[CE_EB]);
    double* C = cr_edge(w, dm, b, er[CE_FILL]) + (size_t)m0 * dp + n0;
    wg_tile_dma_f32<UPD_DMA_DEPTH>(C, dp, Px + (size_t)m0 * ld32, Py + (size_t)n0 * ld32, nullptr, nullptr, ld32, M, N, ld32, er[CE_FACC] ? GM_SUB : GM_NEG, GM_NOTRI, it, lds, TMPC_ABL(dm));
  }
}

// ------------------------------------------------------------------ triangular solves in the same order
// Right-hand sides R [p][dp][NC] (NC interleaved), in place.  Forward, level by level: z_i <- L_i^-1 z_i for the eliminated nodes,
// then z_s -= O_s z_i for the surviving neighbours; backward in reverse: z_i <- L_i^-T (z_i - O_a' z_a - O_b' z_b).
// The matrix-vector work is the skinny MFMA GEMM of tmpc_factor.h (wg_gemv16); every factor block is streamed once per sweep.
// LDS image: the vector of the node and one neighbour's (two regions of NCV rows), a tile-sized scratch, the slab of the skinny GEMM.
// 160 KB hold blocks up to dp = 3168 (round 4: 2384 with four-row regions; before, a third, unused vector region capped it at 1552).
// Round 5: the two vector regions hold NCV = 3 rows (the right-hand sides there are), not the NCP = 4 the operand fragments of wg_gemv16 address -- fragment column 3 reads
// whatever follows the region (LDS of this workgroup, never stored: a column of the result depends on its own operand column only).  Blocks up to dp = 3168 instead of 2384:
// Step 3 at n = 64 with nx = 40 (blocks of 2901) fits.
constexpr int NCV = 3;
constexpr int cr_solve_lds_doubles(int dp) { return 2 * NCV * (dp + 4) + NCP * (TB + 4) + 64 * GLDV + 16; }

__global__ void __launch_bounds__(256) k_cr_fwd_diag(WS w, Dims dm, CrDev cr, int eoff, int nelim, int count, int pass) {
  const int it = cr_item(count * nelim);
  if (it < 0) return;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int b = cr.alist[it / nelim];
  const int nc = cr_nc(w, b, pass);
  if (nc == 0) return;
  const int* er = cr.elim + (size_t)(eoff + it % nelim) * CR_EW;
  const int node = er[CE_NODE], dp = dm.dp, xld = dp + 4, tld = TB + 4;
  double* zc = lds; double* tmp = lds + 2 * NCV * xld; double* As = tmp + NCP * tld;
  for (int e = threadIdx.x; e < NCV * xld; e += 256) zc[e] = 0.0;
  for (int e = threadIdx.x; e < NCP * tld; e += 256) tmp[e] = 0.0;
  __syncthreads();
  double* R = cr_rhs(w, dm, b, node, nc);
  vec_g2s(zc, xld, R, dp, nc);
  __syncthreads();
  blk_fwd(zc, xld, tmp, tld, w.D + ((size_t)b * dm.p + node) * dp * dp, w.Linv + ((size_t)b * dm.p + node) * dm.nt * TB * TB, dp, As, nc);
  vec_s2g(R, zc, xld, dp, nc);
}

// lowp_only: the call inside the factorisation (cr_factor, fused forward sweep of pass 1) for the problems whose update tiles run in single precision and
// carry no right-hand sides -- every other problem's step rode in k_cr_update_dma
__global__ void __launch_bounds__(256) k_cr_fwd_off(WS w, Dims dm, CrDev cr, int uoff, int nupd, int count, int pass, int lowp_only) {
  const int it = cr_item(count * nupd);
  if (it < 0) return;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int b = cr.alist[it / nupd];
  if (lowp_only && !cr_lowp(w, b)) return;
  const int nc = cr_nc(w, b, pass);
  if (nc == 0) return;
  const int* ur = cr.upd + (size_t)(uoff + it % nupd) * CR_UW;
  const int dp = dm.dp, xld = dp + 4, tld = TB + 4;
  double* zs = lds; double* zi = zs + NCV * xld; double* As = lds + 2 * NCV * xld + NCP * tld;
  for (int e = threadIdx.x; e < 2 * NCV * xld + NCP * tld; e += 256) lds[e] = 0.0;
  __syncthreads();
  double* R = cr_rhs(w, dm, b, ur[CU_NODE], nc);
  vec_g2s(zs, xld, R, dp, nc);
  for (int q = 0; q < 2; ++q) {
    const int slot = ur[q ? CU_E1 : CU_E0];
    if (slot < 0) break;
    vec_g2s(zi, xld, cr_rhs(w, dm, b, ur[q ? CU_S1 : CU_S0], nc), dp, nc);
    __syncthreads();
    // z_s -= O_s z_i; in an iteration whose updates run in single precision the sweep reads the float32 copies of the O blocks (half the bytes)
    if (cr_lowp(w, b)) wg_gemv16<false, false, float>(zs, xld, zi, xld, cr_edge32(w, dm, b, slot), cr_ld32(dm), dp, dp, true, -1.0, As, nc);
    else wg_gemv16<false>(zs, xld, zi, xld, cr_edge(w, dm, b, slot), dp, dp, dp, true, -1.0, As, nc);
  }
  __syncthreads();
  vec_s2g(R, zs, xld, dp, nc);
}

__global__ void __launch_bounds__(256) k_cr_bwd(WS w, Dims dm, CrDev cr, int eoff, int nelim, int count, int pass) {
  const int it = cr_item(count * nelim);
  if (it < 0) return;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int b = cr.alist[it / nelim];
  const int nc = cr_nc(w, b, pass);
  if (nc == 0) return;
  const int* er = cr.elim + (size_t)(eoff + it % nelim) * CR_EW;
  const int node = er[CE_NODE], dp = dm.dp, xld = dp + 4, tld = TB + 4;
  double* zc = lds; double* zn = zc + NCV * xld; double* tmp = lds + 2 * NCV * xld; double* As = tmp + NCP * tld;
  for (int e = threadIdx.x; e < 2 * NCV * xld + NCP * tld; e += 256) lds[e] = 0.0;
  __syncthreads();
  double* R = cr_rhs(w, dm, b, node, nc);
  vec_g2s(zc, xld, R, dp, nc);
  for (int q = 0; q < 2; ++q) {
    const int slot = er[q ? CE_EB : CE_EA];
    if (slot < 0) continue;
    __syncthreads();
    vec_g2s(zn, xld, cr_rhs(w, dm, b, er[q ? CE_NB : CE_NA], nc), dp, nc);
    __syncthreads();
    // z_i -= O_x' z_x
    if (cr_lowp(w, b)) wg_gemv16<true, false, float>(zc, xld, zn, xld, cr_edge32(w, dm, b, slot), cr_ld32(dm), dp, dp, true, -1.0, As, nc);
    else wg_gemv16<true>(zc, xld, zn, xld, cr_edge(w, dm, b, slot), dp, dp, dp, true, -1.0, As, nc);
  }
  __syncthreads();
  blk_bwd(zc, xld, tmp, tld, w.D + ((size_t)b * dm.p + node) * dp * dp, w.Linv + ((size_t)b * dm.p + node) * dm.nt * TB * TB, dp, As, nc);
  vec_s2g(R, zc, xld, dp, nc);
}

}  // namespace tmpc
