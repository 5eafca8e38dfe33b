// Small dense helpers for the per-stage (nx+nu) x (nx+nu) blocks, fp64 -- the body of tmpc_small.h, included once per instantiation:
//   namespace tmpc        NMAX = 32, LD = 33, wsync() = workgroup barrier: one workgroup (64 or 256 threads) per stage, the per-stage kernels of tmpc_stage.h
//   namespace tmpc::sm8   NMAX = 8,  LD = 9,  wsync() = wave-local LDS fence: one WAVE per stage inside the persistent one-workgroup-per-problem
//                         kernel of tmpc_persist.h (round 5), where the sixteen waves of a workgroup work on different stages at the same time
// (TMPC_SM_NMAX, TMPC_SM_LD, TMPC_SM_WSYNC are set by tmpc_small.h around each inclusion.)
// One 64-lane wavefront (or four) owns one stage; matrices live in LDS with an odd leading dimension LD (the 4-row register blocks of `mm`
// hit distinct banks for ds_read_b64).  Every helper ends with wsync() unless noted.
constexpr int NMAX = TMPC_SM_NMAX;    // max stage-block size n = nx + nu + ns
constexpr int LD = TMPC_SM_LD;        // LDS leading dimension (doubles)
constexpr int MS = NMAX * LD;         // doubles per LDS matrix slot

__device__ __forceinline__ void wsync() { TMPC_SM_WSYNC(); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// value of lane l (wave-uniform l) in every lane: two v_readlane instead of the ds_bpermute pair of __shfl
__device__ __forceinline__ double wave_bcast(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// sum over the NT threads of the block (NT = 64: one wave, shuffles only; NT = 256: four wave sums joined through LDS).
// Every thread gets the result.  Ends with a barrier for NT > 64.
template <int NT>
__device__ __forceinline__ double block_sum(double v) {
  v = wave_sum(v);
  if (NT > 64) {
    __shared__ double red_[NT / 64];
    __syncthreads();                                  // (previous use of red_)
    if ((threadIdx.x & 63) == 0) red_[threadIdx.x >> 6] = v;
    __syncthreads();
    v = 0.0;
#pragma unroll
    for (int q = 0; q < NT / 64; ++q) v += red_[q];
  }
  return v;
}

// e -> (i, j) = (e / n, e % n): the element loops of the per-stage kernels run this for every element; the benchmark shape has
// n = 32 (and the n x n matrices of every shape have n <= 32), where a shift replaces the ~20-instruction integer division
__device__ __forceinline__ void ediv(int e, int n, int& i, int& j) {
  if (n == 32) { i = e >> 5; j = e & 31; } else { i = e / n; j = e - i * n; }
}

// The helpers below take the thread index `lane` in [0, NT) of a block of NT threads (NT = 64: one wave per stage, the original
// form; NT = 256: four waves share the stage's matrices -- the same LDS, four times the loads in flight and a quarter of the
// dependent work per wave; the stage kernels are latency-bound with one wave per SIMD).
// The same copy in two halves, so that several matrices can be in flight before the first one is needed: g2r issues the loads of a
// matrix of at most NMAX x NMAX elements into registers (element e = lane + q NT), r2s parks them in an LDS slot.
template <int NT>
__device__ __forceinline__ void g2r(double (&v)[NMAX * NMAX / NT], const double* __restrict__ g, int rows, int cols, int ldg, int lane) {
  const int tot = rows * cols;
#pragma unroll
  for (int q = 0; q < NMAX * NMAX / NT; ++q) {
    const int e = lane + q * NT;
    int i, j; ediv(e, cols, i, j);
    v[q] = (e < tot) ? g[(size_t)i * ldg + j] : 0.0;
  }
}
template <int NT>
__device__ __forceinline__ void r2s(double* __restrict__ s, const double (&v)[NMAX * NMAX / NT], int rows, int cols, int lane) {
  const int tot = rows * cols;
#pragma unroll
  for (int q = 0; q < NMAX * NMAX / NT; ++q) {
    const int e = lane + q * NT;
    int i, j; ediv(e, cols, i, j);
    if (e < tot) s[i * LD + j] = v[q];
  }
  wsync();
}
// global (rows x cols, row-major, ld = ldg) -> LDS slot (LD).  Coalesced along rows.
template <int NT = 64>
__device__ __forceinline__ void g2s(double* __restrict__ s, const double* __restrict__ g, int rows, int cols,
                                    int ldg, int lane) {
  const int tot = rows * cols;
  int e = lane;
  for (; e + 3 * NT < tot; e += 4 * NT) {      // four loads in flight per thread: one exposed memory latency per 4 NT elements
    int i[4], j[4]; double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int eu = e + NT * u; ediv(eu, cols, i[u], j[u]); v[u] = g[(size_t)i[u] * ldg + j[u]]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) s[i[u] * LD + j[u]] = v[u];
  }
  for (; e < tot; e += NT) {
    int i, j; ediv(e, cols, i, j);
    s[i * LD + j] = g[(size_t)i * ldg + j];
  }
  wsync();
}
template <int NT = 64>
__device__ __forceinline__ void s2g(double* __restrict__ g, const double* __restrict__ s, int rows, int cols,
                                    int ldg, int lane) {
  const int tot = rows * cols;
  for (int e = lane; e < tot; e += NT) {
    int i, j; ediv(e, cols, i, j);
    g[(size_t)i * ldg + j] = s[i * LD + j];
  }
  wsync();
}
// symmetrised store: g = (s + s')/2
template <int NT = 64>
__device__ __forceinline__ void s2g_sym(double* __restrict__ g, const double* __restrict__ s, int n, int lane) {
  const int tot = n * n;
  for (int e = lane; e < tot; e += NT) {
    int i, j; ediv(e, n, i, j);
    g[(size_t)i * n + j] = 0.5 * (s[i * LD + j] + s[j * LD + i]);
  }
  wsync();
}
template <int NT = 64>
__device__ __forceinline__ void s_sym(double* s, int n, int lane) {   // in place (s+s')/2
  const int tot = n * n;
  for (int e = lane; e < tot; e += NT) {
    int i, j; ediv(e, n, i, j);
    if (j < i) {
      const double v = 0.5 * (s[i * LD + j] + s[j * LD + i]);
      s[i * LD + j] = v;
      s[j * LD + i] = v;
    }
  }
  wsync();
}

// C (M x N) {=, +=, -=} A (M x K) * B (K x N); element (i,k) of A at A[i*ars + k*acs], (k,j) of B at
// B[k*brs + j*bcs]; C row-major with LD.  mode 0: '=', 1: '+=', 2: '-='.  M,N,K <= 32.
// Lane (li,lj) = (lane>>3, lane&7) owns the 4x4 block rows 4li.., cols 4lj..; out-of-range rows/cols read
// in-slot garbage that only reaches outputs which are never stored.
__device__ __forceinline__ void mm64(double* __restrict__ C, const double* __restrict__ A, int ars, int acs,
                                     const double* __restrict__ B, int brs, int bcs, int M, int N, int K, int mode,
                                     int lane) {
#if TMPC_SM_NMAX <= 8
  // M, N, K <= 8: one lane per entry of C; the (up to) sixteen operand reads leave together, entries beyond K are read inside the slots and masked
  {
    const int i = lane >> 3, j = lane & 7;
    const bool on = i < M && j < N;
    double acc1 = 0.0;
    if (on) {
      double a[8], b[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { a[k] = A[i * ars + k * acs]; b[k] = B[k * brs + j * bcs]; }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc1 = fma(k < K ? a[k] : 0.0, k < K ? b[k] : 0.0, acc1);
    }
    wsync();
    if (on) {
      double* p = &C[i * LD + j];
      if (mode == 0) *p = acc1;
      else if (mode == 1) *p += acc1;
      else *p -= acc1;
    }
    wsync();
    return;
  }
#endif
  const int i0 = (lane >> 3) * 4, j0 = (lane & 7) * 4;
  double acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
  if (i0 < M && j0 < N) {
    for (int k = 0; k < K; ++k) {
      double a[4], b[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) a[r] = A[(i0 + r) * ars + k * acs];
#pragma unroll
      for (int c = 0; c < 4; ++c) b[c] = B[k * brs + (j0 + c) * bcs];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = fma(a[r], b[c], acc[r][c]);
    }
  }
  wsync();   // all reads of A/B done before C (which may alias a consumed operand slot) is written
  if (i0 < M && j0 < N) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = i0 + r, j = j0 + c;
        if (i < M && j < N) {
          double* p = &C[i * LD + j];
          if (mode == 0) *p = acc[r][c];
          else if (mode == 1) *p += acc[r][c];
          else *p -= acc[r][c];
        }
      }
  }
  wsync();
}

// the same product by 256 threads: thread (ti, tj) = (tid >> 4, tid & 15) owns the 2 x 2 block rows 2 ti.., cols 2 tj..
__device__ __forceinline__ void mm256(double* __restrict__ C, const double* __restrict__ A, int ars, int acs,
                                      const double* __restrict__ B, int brs, int bcs, int M, int N, int K, int mode,
                                      int tid) {
  const int i0 = (tid >> 4) * 2, j0 = (tid & 15) * 2;
  double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0;
  if (i0 < M && j0 < N) {
    const double* ap = A + i0 * ars;
    const double* bp = B + j0 * bcs;
#pragma unroll 4
    for (int k = 0; k < K; ++k) {
      const double x0 = ap[k * acs], x1 = ap[ars + k * acs];
      const double y0 = bp[k * brs], y1 = bp[bcs + k * brs];
      a00 = fma(x0, y0, a00); a01 = fma(x0, y1, a01); a10 = fma(x1, y0, a10); a11 = fma(x1, y1, a11);
    }
  }
  wsync();   // all reads of A/B done before C (which may alias a consumed operand slot) is written
  if (i0 < M && j0 < N) {
    const double v[2][2] = {{a00, a01}, {a10, a11}};
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int i = i0 + r, j = j0 + c;
        if (i < M && j < N) {
          double* p = &C[i * LD + j];
          if (mode == 0) *p = v[r][c];
          else if (mode == 1) *p += v[r][c];
          else *p -= v[r][c];
        }
      }
  }
  wsync();
}
// The same product on the matrix cores: wave w owns the 16 x 16 tile (w >> 1, w & 1) of C and runs K / 4 v_mfma_f64_16x16x4 on
// operand fragments read straight from LDS -- 2 reads per lane and instruction, an eighth of the LDS traffic of the vector form above
// (which reads one operand per FMA and leaves the per-stage kernels LDS-bound: pre 122 -> 108 ms per step; with one wave per stage
// and two blocks per CU the same instruction was slower than the vector form).  Rows / columns beyond M / N are computed on whatever the
// 32 x 33 slots hold and not stored; K is padded with zeros to a multiple of 4.
__device__ __forceinline__ void mm256_mfma(double* __restrict__ C, const double* __restrict__ A, int ars, int acs,
                                           const double* __restrict__ B, int brs, int bcs, int M, int N, int K, int mode,
                                           int tid) {
  const int wv = tid >> 6, lane = tid & 63;
  const int ti = (wv >> 1) * 16, tj = (wv & 1) * 16;
  const int fr = lane & 15, fk = lane >> 4;
  typedef double d4_t __attribute__((ext_vector_type(4)));
  d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
  const bool on = ti < M && tj < N;
  if (on) {
    const double* ap = A + (ti + fr) * ars + fk * acs;
    const double* bp = B + fk * brs + (tj + fr) * bcs;
    for (int k0 = 0; k0 < K; k0 += 4) {
      const bool kin = k0 + fk < K;
      const double a = kin ? ap[k0 * acs] : 0.0;
      const double b = kin ? bp[k0 * brs] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
  }
  wsync();   // all reads of A/B done before C (which may alias a consumed operand slot) is written
  if (on) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = ti + fk + 4 * r, j = tj + fr;
      if (i < M && j < N) {
        double* p = &C[i * LD + j];
        if (mode == 0) *p = acc[r];
        else if (mode == 1) *p += acc[r];
        else *p -= acc[r];
      }
    }
  }
  wsync();
}
template <int NT = 64>
__device__ __forceinline__ void mm(double* __restrict__ C, const double* __restrict__ A, int ars, int acs,
                                   const double* __restrict__ B, int brs, int bcs, int M, int N, int K, int mode,
                                   int lane) {
#ifndef TMPC_MM_VALU                        // (-DTMPC_MM_VALU: the vector form, for comparison)
  if (NT == 256) { mm256_mfma(C, A, ars, acs, B, brs, bcs, M, N, K, mode, lane); return; }
#endif
  if (NT == 256) mm256(C, A, ars, acs, B, brs, bcs, M, N, K, mode, lane);
  else mm64(C, A, ars, acs, B, brs, bcs, M, N, K, mode, lane);
}

// 1/sqrt(x) for x > 0: hardware estimate (v_rsq_f64, ~2^-26) + two Newton steps
__device__ __forceinline__ double rsqrt_nr(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * (1.5 - 0.5 * x * r * r);
  r = r * (1.5 - 0.5 * x * r * r);
  return r;
}

// In-place lower Cholesky of the symmetric n x n matrix in LDS slot A (only the lower triangle is
// referenced/written).  Returns 0 on success; on a non-positive pivot the pivot is replaced by a tiny
// positive number (Cholesky-with-shift) and the return value counts such events.
// Left-looking (Crout): column j = A[:, j] - L[:, :j] L[j, :j]', two lanes per row share the dot product (k strided by
// 2), the pivot travels by shuffle, so a column costs ONE barrier and no read-modify-write of LDS (the right-looking
// version it replaces took three barriers, a square root and a division on the critical path of every column: 48 k
// cycles per 32 x 32 matrix, a quarter of k_stage_pre).
__device__ __forceinline__ int chol_lower(double* A, int n, int lane) {
  int nbad = 0;
  const int row = lane & 31, half = lane >> 5;
  const bool mine = row < n;
  for (int j = 0; j < n; ++j) {
    double acc = 0.0;
    if (mine && row >= j) {
      const double* li = A + row * LD;
      const double* lj = A + j * LD;
      double a0 = 0.0, a1 = 0.0;
      int k = half;
      for (; k + 2 < j; k += 4) { a0 = fma(li[k], lj[k], a0); a1 = fma(li[k + 2], lj[k + 2], a1); }
      for (; k < j; k += 2) a0 = fma(li[k], lj[k], a0);
      acc = a0 + a1;
    }
    acc += __shfl_xor(acc, 32, 64);
    double sij = 0.0;
    if (mine && row >= j) sij = A[row * LD + j] - acc;
    double piv = wave_bcast(sij, j);
    if (!(piv > 0.0)) { piv = 1e-300; ++nbad; }
    const double rinv = rsqrt_nr(piv);
    if (half == 0 && mine && row >= j) A[row * LD + j] = (row == j) ? piv * rinv : sij * rinv;
    wsync();
  }
  return nbad;
}

// Is I + theta W positive definite?  (W symmetric n x n in LDS, destroyed; wave-uniform answer.)  The same Crout sweep as chol_lower on the
// shifted matrix, leaving at the first non-positive pivot.  k_eigmin asks this before it computes an eigenvalue: a step length is
// only needed exactly when the step is SHORT (lambda_min(W) <= -1 / theta); a long one is clipped to 1 anyway.
__device__ __forceinline__ bool shifted_is_pd(double* A, double theta, int n, int lane) {
  const int row = lane & 31, half = lane >> 5;
  const bool mine = row < n;
  for (int e = lane; e < n * n; e += 64) { int i, j; ediv(e, n, i, j); A[i * LD + j] = theta * A[i * LD + j] + (i == j ? 1.0 : 0.0); }
  wsync();
  for (int j = 0; j < n; ++j) {
    double acc = 0.0;
    if (mine && row >= j) {
      const double* li = A + row * LD;
      const double* lj = A + j * LD;
      double a0 = 0.0, a1 = 0.0;
      int k = half;
      for (; k + 2 < j; k += 4) { a0 = fma(li[k], lj[k], a0); a1 = fma(li[k + 2], lj[k + 2], a1); }
      for (; k < j; k += 2) a0 = fma(li[k], lj[k], a0);
      acc = a0 + a1;
    }
    acc += __shfl_xor(acc, 32, 64);
    double sij = 0.0;
    if (mine && row >= j) sij = A[row * LD + j] - acc;
    const double piv = wave_bcast(sij, j);
    if (!(piv > 0.0)) return false;                        // (wave-uniform: every lane holds the same pivot)
    const double rinv = rsqrt_nr(piv);
    if (half == 0 && mine && row >= j) A[row * LD + j] = (row == j) ? piv * rinv : sij * rinv;
    wsync();
  }
  return true;
}

// Li = L^-1 for lower-triangular L (n x n, LDS).  Li gets explicit zeros above the diagonal.
// Row-wise forward substitution: row i of L^-1 from the rows above, Li[i][c] = -(sum_{c<=k<i} L[i][k] Li[k][c]) / L[i][i];
// two lanes per column share the sum, L[i][k] is a broadcast read, Li[k][c] runs along the lanes, the reciprocal
// diagonal is computed once per lane and shuffled -- one barrier per row, no division inside the loop.
__device__ __forceinline__ void tri_inv_lower(double* __restrict__ Li, const double* __restrict__ L, int n, int lane) {
  const int c = lane & 31, half = lane >> 5;
  double rdl = 1.0;
  if (lane < n) rdl = 1.0 / L[lane * LD + lane];
  for (int i = 0; i < n; ++i) {
    const double rdi = wave_bcast(rdl, i);
    double acc = 0.0;
    if (c < i) {
      const double* li = L + i * LD;
      double a0 = 0.0, a1 = 0.0;
      int k = c + half;
      for (; k + 2 < i; k += 4) { a0 = fma(li[k], Li[k * LD + c], a0); a1 = fma(li[k + 2], Li[(k + 2) * LD + c], a1); }
      for (; k < i; k += 2) a0 = fma(li[k], Li[k * LD + c], a0);
      acc = a0 + a1;
    }
    acc += __shfl_xor(acc, 32, 64);
    if (half == 0 && c < n) Li[i * LD + c] = (c < i) ? -acc * rdi : ((c == i) ? rdi : 0.0);
    wsync();
  }
}

// Two matrices at once: lanes 0..31 factor A, lanes 32..63 factor B (one lane per row, full-length dot products, pivots
// by v_readlane) -- the two dependency chains share every barrier and every latency, so the pair costs what one costs.
__device__ __forceinline__ int chol_lower_pair(double* A, double* B, int n, int lane) {
  int nbad = 0;
  const int row = lane & 31, half = (lane >> 5) & 1;
  double* Mx = half ? B : A;
  const bool mine = row < n && lane < 64;        // (blocks of more than one wave: the first wave works, the others keep the barriers)
  for (int j = 0; j < n; ++j) {
    double sij = 0.0;
    if (mine && row >= j) {
      const double* li = Mx + row * LD;
      const double* lj = Mx + j * LD;
      double a0 = 0.0, a1 = 0.0;
      int k = 0;
      for (; k + 1 < j; k += 2) { a0 = fma(li[k], lj[k], a0); a1 = fma(li[k + 1], lj[k + 1], a1); }
      if (k < j) a0 = fma(li[k], lj[k], a0);
      sij = li[j] - (a0 + a1);
    }
    const double pa = wave_bcast(sij, j), pb = wave_bcast(sij, 32 + j);      // wave-uniform source lanes
    if (!(pa > 0.0)) ++nbad;
    if (!(pb > 0.0)) ++nbad;
    double piv = half ? pb : pa;
    if (!(piv > 0.0)) piv = 1e-300;
    const double rinv = rsqrt_nr(piv);
    if (mine && row >= j) Mx[row * LD + j] = (row == j) ? piv * rinv : sij * rinv;
    wsync();
  }
  return nbad;
}
// LiA = LA^-1 (lanes 0..31) and LiB = LB^-1 (lanes 32..63), one lane per column
__device__ __forceinline__ void tri_inv_lower_pair(double* __restrict__ LiA, const double* __restrict__ LA,
                                                   double* __restrict__ LiB, const double* __restrict__ LB, int n, int lane) {
  const int c = lane & 31, half = (lane >> 5) & 1;
  const bool first = lane < 64;                  // (blocks of more than one wave: the first wave works, the others keep the barriers)
  const double* L = half ? LB : LA;
  double* Li = half ? LiB : LiA;
  double rdl = 1.0;
  if (c < n) rdl = 1.0 / L[c * LD + c];
  for (int i = 0; i < n; ++i) {
    const double ra = wave_bcast(rdl, i), rb = wave_bcast(rdl, 32 + i);
    const double rdi = half ? rb : ra;
    double acc = 0.0;
    if (c < i && first) {
      const double* li = L + i * LD;
      double a0 = 0.0, a1 = 0.0;
      int k = c;
      for (; k + 1 < i; k += 2) { a0 = fma(li[k], Li[k * LD + c], a0); a1 = fma(li[k + 1], Li[(k + 1) * LD + c], a1); }
      if (k < i) a0 = fma(li[k], Li[k * LD + c], a0);
      acc = a0 + a1;
    }
    if (c < n && first) Li[i * LD + c] = (c < i) ? -acc * rdi : ((c == i) ? rdi : 0.0);
    wsync();
  }
}

// Block-size generic forms of the pair routines.  NT = 64: the single-wave pair above.  NT = 256: matrix A on wave 0 and matrix B
// on wave 1, two lanes per row (column) each -- the dot products are half as long as in the one-lane-per-row pair form -- and the
// other two waves only keep the barriers.  The returned count of non-positive pivots is the same in every thread.
template <int NT>
__device__ __forceinline__ int chol_lower_pair_t(double* A, double* B, int n, int tid) {
  if (NT == 64) return chol_lower_pair(A, B, n, tid);
  const int wv = tid >> 6, lane = tid & 63;
  const int row = lane & 31, half = lane >> 5;
  double* Mx = wv ? B : A;
  const bool mine = (wv < 2) && row < n;
  int nbad = 0;
  for (int j = 0; j < n; ++j) {
    double acc = 0.0;
    if (mine && row >= j) {
      const double* li = Mx + row * LD;
      const double* lj = Mx + j * LD;
      double a0 = 0.0, a1 = 0.0;
      int k = half;
      for (; k + 2 < j; k += 4) { a0 = fma(li[k], lj[k], a0); a1 = fma(li[k + 2], lj[k + 2], a1); }
      for (; k < j; k += 2) a0 = fma(li[k], lj[k], a0);
      acc = a0 + a1;
    }
    acc += __shfl_xor(acc, 32, 64);
    double sij = 0.0;
    if (mine && row >= j) sij = Mx[row * LD + j] - acc;
    double piv = wave_bcast(sij, j);
    if (!(piv > 0.0)) { piv = 1e-300; if (wv < 2) ++nbad; }
    const double rinv = rsqrt_nr(piv);
    if (half == 0 && mine && row >= j) Mx[row * LD + j] = (row == j) ? piv * rinv : sij * rinv;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // each wave touches its own matrix only: in-order LDS, no workgroup barrier per column
  }
  wsync();
  return (int)block_sum<NT>((lane == 0 && wv < 2) ? (double)nbad : 0.0);
}
template <int NT>
__device__ __forceinline__ void tri_inv_lower_pair_t(double* __restrict__ LiA, const double* __restrict__ LA,
                                                     double* __restrict__ LiB, const double* __restrict__ LB, int n, int tid) {
  if (NT == 64) { tri_inv_lower_pair(LiA, LA, LiB, LB, n, tid); return; }
  const int wv = tid >> 6, lane = tid & 63;
  const int c = lane & 31, half = lane >> 5;
  const double* L = wv ? LB : LA;
  double* Li = wv ? LiB : LiA;
  const bool on = wv < 2;
  double rdl = 1.0;
  if (on && lane < n) rdl = 1.0 / L[lane * LD + lane];
  for (int i = 0; i < n; ++i) {
    const double rdi = wave_bcast(rdl, i);
    double acc = 0.0;
    if (on && c < i) {
      const double* li = L + i * LD;
      double a0 = 0.0, a1 = 0.0;
      int k = c + half;
      for (; k + 2 < i; k += 4) { a0 = fma(li[k], Li[k * LD + c], a0); a1 = fma(li[k + 2], Li[(k + 2) * LD + c], a1); }
      for (; k < i; k += 2) a0 = fma(li[k], Li[k * LD + c], a0);
      acc = a0 + a1;
    }
    acc += __shfl_xor(acc, 32, 64);
    if (on && half == 0 && c < n) Li[i * LD + c] = (c < i) ? -acc * rdi : ((c == i) ? rdi : 0.0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (as above)
  }
  wsync();
}

__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }   // LDS ops of one wave complete in order: this keeps the compiler from reordering across it

// D (16 x 16 in registers, acc[r] = D[(lane >> 4) + 4 r][lane & 15]) += sgn * A B  with A[r][k] = Ap[r * lda + k], B[k][c] = Bp[k * bk + c * bc]
__device__ __forceinline__ double4_t mm16(const double* Ap, int lda, const double* Bp, int bk, int bc, double4_t acc, double sgn, int lane) {
  const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
  for (int k0 = 0; k0 < 16; k0 += 4) {
    const double a = sgn * Ap[fr * lda + k0 + fk];
    const double bv = Bp[(k0 + fk) * bk + fr * bc];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc, 0, 0, 0);
  }
  return acc;
}
__device__ __forceinline__ double4_t load_d16(const double* Cp, int ldc, int lane) {
  const int fr = lane & 15, fk = lane >> 4;
  return (double4_t){Cp[fk * ldc + fr], Cp[(fk + 4) * ldc + fr], Cp[(fk + 8) * ldc + fr], Cp[(fk + 12) * ldc + fr]};
}
__device__ __forceinline__ void store_d16(double* Cp, int ldc, double4_t v, int lane) {
  const int fr = lane & 15, fk = lane >> 4;
  Cp[fk * ldc + fr] = v[0]; Cp[(fk + 4) * ldc + fr] = v[1]; Cp[(fk + 8) * ldc + fr] = v[2]; Cp[(fk + 12) * ldc + fr] = v[3];
}

// n = 32, four waves: the Cholesky factors of A and B (lower triangles in place) and their inverses LiA, LiB (zeros above the diagonal),
// blocked by 16.  The 32 dependent column steps of the pair routines above are the floor of k_stage_pre (each one an LDS round trip,
// a dot product, two shuffles and a reciprocal square root: 41 k + 36 k cycles per pair); by blocks the dot products are half as
// long and the off-diagonal work -- panel L21 = A21 Li11', update A22 -= L21 L21', Li21 = -Li22 (L21 Li11) -- runs on
// v_mfma_f64_16x16x4 with LDS operands, wave 0 on A and wave 1 on B.  Returns the number of non-positive pivots (same in every thread).
template <int NT>
__device__ __forceinline__ int chol_inv_pair32(double* A, double* LiA, double* B, double* LiB, int tid) {
  const int wv = tid >> 6, lane = tid & 63;
  const int o22 = 16 * LD + 16;
  int nbad = chol_lower_pair_t<NT>(A, B, 16, tid);
  tri_inv_lower_pair_t<NT>(LiA, A, LiB, B, 16, tid);
  if (wv < 2) {
    double* M = wv ? B : A; double* Li = wv ? LiB : LiA;
    double* M21 = M + 16 * LD;
    double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
    acc = mm16(M21, LD, Li, 1, LD, acc, 1.0, lane);         // L21 = A21 Li11'   (B[k][c] = Li11[c][k])
    wave_lds_sync();
    store_d16(M21, LD, acc, lane);
    wave_lds_sync();
    acc = load_d16(M + o22, LD, lane);
    acc = mm16(M21, LD, M21, 1, LD, acc, -1.0, lane);       // A22 -= L21 L21'   (B[k][c] = L21[c][k])
    wave_lds_sync();
    store_d16(M + o22, LD, acc, lane);
  }
  wsync();
  nbad += chol_lower_pair_t<NT>(A + o22, B + o22, 16, tid);
  tri_inv_lower_pair_t<NT>(LiA + o22, A + o22, LiB + o22, B + o22, 16, tid);
  if (wv < 2) {
    double* M = wv ? B : A; double* Li = wv ? LiB : LiA;
    double* M21 = M + 16 * LD; double* Li21 = Li + 16 * LD;
    double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
    acc = mm16(M21, LD, Li, LD, 1, acc, 1.0, lane);         // W = L21 Li11
    wave_lds_sync();
    store_d16(Li21, LD, acc, lane);                         // (parked in its final place)
    wave_lds_sync();
    acc = (double4_t){0.0, 0.0, 0.0, 0.0};
    acc = mm16(Li + o22, LD, Li21, LD, 1, acc, -1.0, lane); // Li21 = -Li22 W
    wave_lds_sync();
    store_d16(Li21, LD, acc, lane);
    store_d16(Li + 16, LD, (double4_t){0.0, 0.0, 0.0, 0.0}, lane);    // the block above the diagonal
  }
  wsync();
  return nbad;
}

// Cyclic (round-robin parallel-ordered) two-sided Jacobi: destroys the symmetric n x n LDS matrix A and
// leaves its eigenvalues on the diagonal.  cs: LDS scratch of >= 4*16 doubles.  One single-wave block works
// on one matrix (the block barrier is then a wave-local barrier).
// tol2: stop when off-diagonal mass <= tol2 * total mass (eigenvalue error ~ sqrt(tol2)*||A||, quadratically
// better for separated eigenvalues).
__device__ __forceinline__ void jacobi_impl(double* A, int n, double* cs, int lane, double tol2) {
#define TMPC_JSYNC() wsync()
  const int m = (n + 1) & ~1;          // players (even)
  const int np = m >> 1;               // pairs per round
  if (n == 1) return;
  for (int sweep = 0; sweep < 14; ++sweep) {
    // convergence test: off-diagonal mass vs total
    double off = 0.0, dia = 0.0;
    for (int e = lane; e < n * n; e += 64) {
      int i, j; ediv(e, n, i, j);
      const double v = A[i * LD + j];
      if (i == j) dia += v * v; else off += v * v;
    }
    off = wave_sum(off); dia = wave_sum(dia);
    if (off <= tol2 * (dia + off) || (dia + off) == 0.0) break;
    for (int r = 0; r < m - 1; ++r) {
      if (lane < np) {
        const int t = lane;
        int p = (t == 0) ? 0 : 1 + ((t - 1 + r) % (m - 1));
        const int u = m - 1 - t;
        int q = 1 + ((u - 1 + r) % (m - 1));
        if (p > q) { const int tmp = p; p = q; q = tmp; }
        double c = 1.0, s = 0.0;
        if (q < n) {
          const double app = A[p * LD + p], aqq = A[q * LD + q], apq = A[p * LD + q];
          if (fabs(apq) > 1e-300 && fabs(apq) > 1e-18 * sqrt(fabs(app * aqq))) {
            const double theta = (aqq - app) / (2.0 * apq);
            const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            c = 1.0 / sqrt(tt * tt + 1.0);
            s = tt * c;
          }
        } else { q = -1; }
        cs[4 * t + 0] = c; cs[4 * t + 1] = s;
        cs[4 * t + 2] = (double)p; cs[4 * t + 3] = (double)q;
      }
      TMPC_JSYNC();
      // columns: (A[i][p], A[i][q]) <- (c a_ip - s a_iq, s a_ip + c a_iq)
      for (int e = lane; e < np * 32; e += 64) {
        const int t = e >> 5, i = e & 31;
        const int q = (int)cs[4 * t + 3];
        if (i < n && q >= 0) {
          const double c = cs[4 * t], s = cs[4 * t + 1];
          const int p = (int)cs[4 * t + 2];
          const double aip = A[i * LD + p], aiq = A[i * LD + q];
          A[i * LD + p] = c * aip - s * aiq;
          A[i * LD + q] = s * aip + c * aiq;
        }
      }
      TMPC_JSYNC();
      // rows
      for (int e = lane; e < np * 32; e += 64) {
        const int t = e >> 5, j = e & 31;
        const int q = (int)cs[4 * t + 3];
        if (j < n && q >= 0) {
          const double c = cs[4 * t], s = cs[4 * t + 1];
          const int p = (int)cs[4 * t + 2];
          const double apj = A[p * LD + j], aqj = A[q * LD + j];
          A[p * LD + j] = c * apj - s * aqj;
          A[q * LD + j] = s * apj + c * aqj;
        }
      }
      TMPC_JSYNC();
    }
  }
  TMPC_JSYNC();
#undef TMPC_JSYNC
}
__device__ __forceinline__ void jacobi_eigvals(double* A, int n, double* cs, int lane) { jacobi_impl(A, n, cs, lane, 1e-31); }

// smallest eigenvalue of the symmetric tridiagonal matrix (dd, ee) in LDS: Sturm-count multisection over 64 shifts per round (wave-uniform result)
__device__ __forceinline__ double tridiag_lmin(const double* dd, const double* ee, int n, int lane) {
  // Gershgorin lower bound, min-diagonal upper bound of lambda_min
  double glo = 1e300, ghi = 1e300;
  if (lane < n) {
    const double el = (lane > 0) ? fabs(ee[lane - 1]) : 0.0, er = (lane < n - 1) ? fabs(ee[lane]) : 0.0;
    glo = dd[lane] - el - er; ghi = dd[lane];
  }
  double lo = wave_min(glo), hi = wave_min(ghi);
  const double scale = fmax(fabs(lo), fabs(hi));
  lo -= 1e-14 * scale + 1e-300;
  // multisection: lane l counts eigenvalues below sigma_l; lambda_min lies in the last interval with count 0
  for (int round = 0; round < 7 && (hi - lo) > 4e-16 * fmax(scale, 1e-300); ++round) {
    const double h = (hi - lo) / 65.0;
    const double sig = lo + h * (double)(lane + 1);
    // Sturm count without divisions: p_i = (d_{i-1} - sigma) p_{i-1} - e_{i-2}^2 p_{i-2}; the number of sign changes of p_0 .. p_n is
    // the number of eigenvalues below sigma (a zero takes the sign opposite to its predecessor); rescaled against overflow
    int cnt = 0;
    double pm = 1.0, pc = dd[0] - sig;
    if (pc == 0.0) pc = -1e-300;
    if (pc < 0.0) ++cnt;
    for (int i = 1; i < n; ++i) {
      const double e2 = ee[i - 1] * ee[i - 1];
      double pn = fma(dd[i] - sig, pc, -e2 * pm);
      if (pn == 0.0) pn = (pc < 0.0) ? 1e-300 : -1e-300;
      if ((pn < 0.0) != (pc < 0.0)) ++cnt;
      const double big = fmax(fabs(pn), fabs(pc));
      const double sc = (big > 1e100) ? 1e-100 : ((big < 1e-100) ? 1e100 : 1.0);
      pm = pc * sc; pc = pn * sc;
    }
    // number of shifts with zero eigenvalues below them
    const unsigned long long mask = __ballot(cnt == 0);
    const int nz = __popcll(mask);            // counts are monotone in sigma: the first nz shifts have count 0
    const double nlo = lo + h * (double)nz;   // sigma_{nz-1} (or lo)
    const double nhi = (nz < 64) ? lo + h * (double)(nz + 1) : hi;
    lo = nlo; hi = nhi;
  }
  return 0.5 * (lo + hi);
}


// The n = 32 form of the tridiagonalisation below with the matrix in REGISTERS: lane (r, h) = (lane & 31, lane >> 5) holds the 16 entries
// A[r][16 h .. 16 h + 15]; the Householder steps are unrolled 16-fold, so column j is a literal register.  Per step only v and w go
// through LDS (one store and eight 16-byte broadcast reads each) instead of two reads and one write of the whole trailing matrix:
// k_eigmin was LDS-bound (72 % of the LDS cycles busy at 16 waves per CU, Householder 148 k of its 197 k cycles per matrix).
// The vectors carry zeros outside the trailing block, so every step runs the same 48 FMAs per lane on the full rows.
// Leaves diagonal and off-diagonal in vv + 32 / vv + 64 like tridiag_min_eig.  vv: >= 128 doubles, 16-byte aligned.
__device__ __forceinline__ void tridiag_reduce32(const double* A, double* vv, int lane) {
  double* dd = vv + 32;
  double* ee = vv + 64;
  double* vb = vv;            // v (32)
  double* wb = vv + 96;       // w (32)
  const int r = lane & 31, h = lane >> 5;
  double a[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) a[q] = A[r * LD + 16 * h + q];
  wsync();
#pragma unroll 1
  for (int hj = 0; hj < 2; ++hj)            // (only the register index must be a literal: half the code of a 30-fold unroll, which was 66 KB)
#pragma unroll
  for (int qj = 0; qj < 16; ++qj) {
    const int j = 16 * hj + qj;
    if (j >= 30) break;
    // x_r = A[r][j], r > j (both halves of a row get it)
    const double xo = a[qj];
    const double xx = __shfl_xor(xo, 32, 64);
    const double xi = (r > j) ? ((h == hj) ? xo : xx) : 0.0;
    const double x0 = wave_bcast(xi, j + 1);
    const double sigma = wave_sum((h == 0 && r > j + 1) ? xi * xi : 0.0);
    if (sigma == 0.0) {                      // already tridiagonal in this column
      if (lane == 0) ee[j] = x0;
      continue;
    }
    const double mu = sqrt(x0 * x0 + sigma);
    const double v0 = (x0 <= 0.0) ? (x0 - mu) : (-sigma / (x0 + mu));
    const double beta = 2.0 * v0 * v0 / (sigma + v0 * v0);
    const double rv0 = 1.0 / v0;
    const double vr = (r == j + 1) ? 1.0 : ((r > j + 1) ? xi * rv0 : 0.0);
    if (h == 0) vb[r] = vr;
    if (lane == 0) ee[j] = mu;               // |H x| = mu e_1 (sign irrelevant for eigenvalues of the tridiagonal)
    wsync();
    double vc[16];
#pragma unroll
    for (int t = 0; t < 8; ++t) { const double2_t u = *(const double2_t*)(vb + 16 * h + 2 * t); vc[2 * t] = u[0]; vc[2 * t + 1] = u[1]; }
    // p = beta * A v over the row (v vanishes outside the trailing block)
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int q = 0; q < 16; q += 2) { a0 = fma(a[q], vc[q], a0); a1 = fma(a[q + 1], vc[q + 1], a1); }
    const double part = a0 + a1;
    const double pi = (r > j) ? beta * (part + __shfl_xor(part, 32, 64)) : 0.0;
    const double kk = 0.5 * beta * wave_sum((h == 0) ? pi * vr : 0.0);
    const double wr = (r > j) ? pi - kk * vr : 0.0;
    if (h == 0) wb[r] = wr;
    wsync();
    // A -= v w' + w v'
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const double2_t u = *(const double2_t*)(wb + 16 * h + 2 * t);
      a[2 * t] -= vr * u[0] + wr * vc[2 * t];
      a[2 * t + 1] -= vr * u[1] + wr * vc[2 * t + 1];
    }
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) if (16 * h + q == r) dd[r] = a[q];
  if (lane == 63) ee[30] = a[14];            // A[31][30]
  wsync();
}

// Smallest eigenvalue of the symmetric n x n LDS matrix A (destroyed): Householder tridiagonalisation
// (backward stable, ~4/3 n^3 flops, a quarter of the LDS traffic of a converged Jacobi) followed by Sturm-count
// multisection over 64 shifts per round.  One single-wave block; vv: LDS scratch of >= 3*32 doubles.
// Wave-uniform return value.
__device__ __forceinline__ double tridiag_min_eig(double* A, int n, double* vv, int lane) {
  double* dd = vv + 32;       // diagonal
  double* ee = vv + 64;       // off-diagonal: ee[i] couples i and i+1
  if (n == 1) { const double a00 = A[0]; if (lane == 0) dd[0] = a00; wsync(); return a00; }
#if TMPC_SM_NMAX >= 32
  if (n == 32) {
    tridiag_reduce32(A, vv, lane);
#ifdef TMPC_CYCLE_PROF
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_prof[5 * 8 + 1] += __builtin_readcyclecounter(); }
#endif
    const double l32 = tridiag_lmin(dd, ee, n, lane);
#ifdef TMPC_CYCLE_PROF
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_prof[5 * 8 + 2] += __builtin_readcyclecounter(); }
#endif
    return l32;
  }
#endif
  for (int j = 0; j + 2 < n; ++j) {
    const int m = n - j - 1;                 // length of the column below the diagonal
    const double xi = (lane < m) ? A[(j + 1 + lane) * LD + j] : 0.0;
    const double x0 = wave_bcast(xi, 0);
    const double sigma = wave_sum((lane >= 1 && lane < m) ? xi * xi : 0.0);
    if (sigma == 0.0) {                      // already tridiagonal in this column
      if (lane == 0) ee[j] = x0;
      wsync();
      continue;
    }
    const double mu = sqrt(x0 * x0 + sigma);
    const double v0 = (x0 <= 0.0) ? (x0 - mu) : (-sigma / (x0 + mu));
    const double beta = 2.0 * v0 * v0 / (sigma + v0 * v0);
    const double rv0 = 1.0 / v0;             // (one reciprocal, wave-uniform, instead of a division per lane)
    if (lane < m) vv[lane] = (lane == 0) ? 1.0 : xi * rv0;
    if (lane == 0) ee[j] = mu;               // |H x| = mu e_1 (sign irrelevant for eigenvalues of the tridiagonal)
    wsync();
    // p = beta * A22 v: two lanes per row (alternate columns), joined by one shuffle
    double pi = 0.0;
    {
      const int r = lane & 31, c0 = lane >> 5;
      double a0 = 0.0, a1 = 0.0;
      if (r < m) {
        const double* ar = A + (j + 1 + r) * LD + j + 1;
        int c = c0;
        for (; c + 2 < m; c += 4) { a0 = fma(ar[c], vv[c], a0); a1 = fma(ar[c + 2], vv[c + 2], a1); }
        for (; c < m; c += 2) a0 = fma(ar[c], vv[c], a0);
      }
      const double part = a0 + a1;
      pi = beta * (part + __shfl_xor(part, 32, 64));
    }
    const double kk = 0.5 * beta * wave_sum((lane < m) ? pi * vv[lane] : 0.0);
    wsync();
    if (lane < m) vv[96 + lane] = pi - kk * vv[lane];     // w
    wsync();
    // A22 -= v w' + w v'   (two lanes per row, alternate columns: no index division, v_r and w_r stay in registers)
    {
      const int r = lane & 31, c0 = lane >> 5;
      if (r < m) {
        const double vr = vv[r], wr = vv[96 + r];
        double* ar = A + (j + 1 + r) * LD + j + 1;
        for (int c = c0; c < m; c += 2) ar[c] -= vr * vv[96 + c] + wr * vv[c];
      }
    }
    wsync();
  }
  if (lane < n) dd[lane] = A[lane * LD + lane];
  if (lane == 0) ee[n - 2] = A[(n - 1) * LD + n - 2];
  wsync();
#ifdef TMPC_CYCLE_PROF
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_prof[5 * 8 + 1] += __builtin_readcyclecounter(); }
#endif
  const double lm_ = tridiag_lmin(dd, ee, n, lane);
#ifdef TMPC_CYCLE_PROF
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_prof[5 * 8 + 2] += __builtin_readcyclecounter(); }
#endif
  return lm_;
}
// largest eigenvalue of the tridiagonal matrix that the last tridiag_min_eig call left in vv (lambda_max(T) = -lambda_min(-T))
__device__ __forceinline__ double tridiag_max_after(double* vv, int n, int lane) {
  double* dd = vv + 32; double* ee = vv + 64;
  if (n == 1) return dd[0];
  if (lane < n) dd[lane] = -dd[lane];
  wsync();
  return -tridiag_lmin(dd, ee, n, lane);
}
// min / max of the diagonal after jacobi_eigvals (wave-uniform result)
__device__ __forceinline__ void diag_minmax(const double* A, int n, int lane, double* mn, double* mx) {
  double lo = 1e300, hi = -1e300;
  if (lane < n) { lo = A[lane * LD + lane]; hi = lo; }
  *mn = wave_min(lo); *mx = wave_max(hi);
}

// <A, B> over the n x n leading block (both LDS)
template <int NT = 64>
__device__ __forceinline__ double dot_ss(const double* A, const double* B, int n, int lane) {
  double acc = 0.0;
  for (int e = lane; e < n * n; e += NT) { int i, j; ediv(e, n, i, j); acc = fma(A[i * LD + j], B[i * LD + j], acc); }
  return block_sum<NT>(acc);
}
template <int NT = 64>
__device__ __forceinline__ double trace_s(const double* A, int n, int lane) {
  double acc = (lane < n) ? A[lane * LD + lane] : 0.0;
  return block_sum<NT>(acc);
}

