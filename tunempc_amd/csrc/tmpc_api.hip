// Host side of the MI355X convexifier: workspace, launch sequence, C ABI (include/tunempc_hip.h).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <new>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <unistd.h>
#include <vector>

#include <string>
#include <unordered_map>

#include "../../include/tunempc_hip.h"
#include "../../include/tunempc_hip_debug.h"
#include "tmpc_common.h"
#include "tmpc_small.h"
#include "tmpc_stage.h"
#include "tmpc_schur.h"
#include "tmpc_factor.h"
#include "tmpc_cr.h"
#include "tmpc_cr_small.h"
#include "tmpc_persist.h"
#include "tmpc_phi.h"
#include "tmpc_t3.h"
#include "tmpc_eig.h"
#include "tmpc_dd.h"
#include "tmpc_big.h"

using namespace tmpc;

static thread_local char g_err[512] = "";
static int set_err(const char* what, hipError_t e, int line) {
  snprintf(g_err, sizeof(g_err), "%s: %s (line %d)", what, hipGetErrorString(e), line);
  return TMPC_E_HIP;
}
#define HIPCHK(x)                                            \
  do {                                                       \
    hipError_t e_ = (x);                                     \
    if (e_ != hipSuccess) return set_err(#x, e_, __LINE__);  \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
  template <typename T> T* as() const { return (T*)p; }
};

// A handle holds up to MAXL independent LANES: each owns a slice of the workspace, a stream and a host thread while a call runs.
// A wave of problems is split over the lanes and their interior-point loops run concurrently: the matrix-core-bound kernels of the
// block factorisation of one lane overlap with the LDS / latency-bound stage kernels and the HBM-bound triangular solves of the
// other (the loops are independent: problems never interact).  Results do not depend on the number of lanes.
constexpr int MAXL = 4;
// One persistent host thread per extra lane (lanes 1 ..): created the first time a wave is split over lanes, parked on a condition variable between waves, joined
// by tmpc_destroy.  (Until round 4 every wave spawned and joined a std::thread per extra lane -- tens of microseconds per wave, on exactly the small shapes whose
// whole solve is a few milliseconds: review of round 4.)  A process forked after the worker was created has no such thread: the pid is checked at every use.
struct LaneWorker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false, done = true, stop = false;
  pid_t pid = 0;
  void start() {
    pid = getpid();
    th = std::thread([this] {
      std::unique_lock<std::mutex> lk(m);
      for (;;) {
        cv.wait(lk, [&] { return has_job || stop; });
        if (stop) return;
        std::function<void()> j = std::move(job);
        has_job = false;
        lk.unlock();
        j();
        lk.lock();
        done = true;
        cv.notify_all();
      }
    });
  }
  void submit(std::function<void()> j) {
    { std::lock_guard<std::mutex> lk(m); job = std::move(j); has_job = true; done = false; }
    cv.notify_all();
  }
  void wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return done; }); }
  void shutdown() {
    { std::lock_guard<std::mutex> lk(m); stop = true; }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
  // End of a worker's life.  In the process that created it: stop, join, delete.  In a process forked after start() the thread does not exist and the mutex /
  // condition variable were copied while the parent's worker sat in cv.wait -- destroying that condition variable may wait for a waiter that never leaves
  // (glibc) and the std::thread object must not be destroyed joinable: the object is LEAKED there on purpose (a few hundred bytes per lane; handles are not
  // meant to be used across fork).
  static void retire(LaneWorker*& wk) {
    if (!wk) return;
    if (wk->pid == getpid()) { wk->shutdown(); delete wk; }
    wk = nullptr;
  }
};

struct Lane {
  LaneWorker* worker;            // host thread of lanes 1 .. (nullptr until first used)
  WS ws;
  // lane-local copies of the user inputs when called with host pointers
  double *dA, *dB, *dH;          // device staging for the host entries (lane capacity)
  double* dG;                    // [cap][p][nr][n] staging of the equality- / active-constraint Jacobians
  int32_t* dncnt;                // [cap][p] rows of C_k per stage (Step 2)
  double* d_info;                // [cap][16]
  double* d_abk;                 // [cap][3]
  void* big_scr;                 // scratch of the generic per-stage kernels when they are forced at n <= 32 (debug flag), allocated on first use
  int32_t* d_si;                 // [cap][2]
  hipStream_t st;                // the lane's own stream
  hipEvent_t ev[8];
  double prof[16];               // see tmpc_get_profile
  std::vector<hipEvent_t> kev;   // profile mode: event pairs around the launches of one factorisation (class = index % 3)
  int last_nb;                   // problems this lane solved in the last wave (trace / multiplier read-back)
  int o32_dp;                    // block width of the last call that wrote float32 O copies (0: none yet, the array is all zero): a call with another width finds stale data where its zero padding should be
  std::unordered_map<std::string, hipGraphExec_t>* graphs;   // launch-bound shapes: one IPM iteration as a captured graph, keyed by every launch argument (run_chunk)
  char err[512];                 // error text of the lane's worker thread
};

struct tmpc_handle {
  Dims dm;          // dm.B = lane capacity (problems per lane and wave)
  int chunk;        // problems per wave = nlanes * dm.B
  int nlanes;
  Opts opt;
  int flags;
  Lane lane[MAXL];
  void* slab;
  size_t slab_bytes;
  hipEvent_t ev_in;              // inputs of a device-resident call are ready on the caller's stream
  int device;                    // HIP device the workspace lives on
  CrSched sched;                 // elimination order of the block factorisation (tmpc_cr.h)
  int* d_sched;                  // device copy: elimination records | update records | orientation
  int rs, mt;                    // rows per workgroup of k_cr_trsm / output tile edge of k_cr_update (0: chosen per launch)
  int tune_small, tune_pretest, tune_fuse, tune_graph, tune_persist;     // tmpc_set_tuning
  void* dd_slab;                 // tight mode (tmpc_set_tight): low words of the double-double planes, allocated on first use
  size_t dd_bytes;
  int tight;                     // 1: the tight phase follows the default solve
  double tight_tol;
};

// device copy of a schedule: [elim | upd | orient]
static int cr_upload(const CrSched& sc, int** out) {
  std::vector<int> flat(sc.elim);
  flat.insert(flat.end(), sc.upd.begin(), sc.upd.end());
  flat.insert(flat.end(), sc.orient.begin(), sc.orient.end());
  *out = nullptr;
  if (hipMalloc(out, flat.size() * sizeof(int)) != hipSuccess) return TMPC_E_NOMEM;
  if (hipMemcpy(*out, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { hipFree(*out); *out = nullptr; return TMPC_E_HIP; }
  return TMPC_OK;
}
static CrDev cr_dev(const CrSched& sc, const int* d_sched, const int* alist) {
  CrDev c; c.elim = d_sched; c.upd = d_sched + sc.elim.size(); c.orient = d_sched + sc.elim.size() + sc.upd.size(); c.alist = alist;
  return c;
}

// ---------------------------------------------------------------------------------- sizes
static bool dims_ok(int p, int nx, int mb) { return p >= 1 && nx >= 1 && mb >= 0 && nx + mb <= NB && nx + mb >= 1; }      // n <= 32: the tuned per-stage kernels; 32 < n <= 96: tmpc_big.h (above 64: plain model only, model_dims_ok)

static Dims make_dims(int chunk, int p, int nx, int mb, int ng = 0, int nc = 0, int step3 = 0) {
  Dims d;
  d.ng = ng; d.nr = ng + nc; d.constr = 0;
  d.nT = step3 ? (nx + mb) * (nx + mb + 1) / 2 : 0;
  d.nz = d.nr + (nc > 0 ? 2 : 0);          // room for the two epigraph variables of Step 2
  d.B = chunk; d.p = p; d.nx = nx; d.mb = mb; d.n = nx + mb;
  d.d = nx * (nx + 1) / 2;
  d.dp = (d.d + d.nz + (d.nT ? d.nT + 1 : 0) + 15) / 16 * 16;      // room for the stage-local variables inside the blocks (run_chunk narrows it when unused)
  d.nt = (d.dp + TB - 1) / TB;
  d.flags = 0;
  return d;
}

// do the interleaved factor records of k_schur (10 nx^2 doubles) fit the LDS?  (nx <= 43; otherwise they live in a per-stage global scratch)
static bool schur_in_lds(const Dims& dm) { return (size_t)sch_rec<0>() * dm.nx * dm.nx * sizeof(double) + (size_t)dm.d * sizeof(unsigned) + 64 <= 158 * 1024; }
struct Carver {
  char* base; size_t off;
  template <typename T> T* take(size_t count) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

static size_t carve(WS& w, const Dims& dm, char* base, Lane* h) {
  Carver c{base, 0};
  const size_t BP = (size_t)dm.B * dm.p, nn = (size_t)dm.n * dm.n, nxx = (size_t)dm.nx * dm.nx;
  const size_t bs = (size_t)dm.dp * dm.dp;
  w.Hb = c.take<double>(BP * nn); w.V = c.take<double>(BP * dm.nx * dm.n);
  w.P = c.take<double>(BP * nxx);
  w.X1 = c.take<double>(BP * nn); w.X2 = c.take<double>(BP * nn); w.S1 = c.take<double>(BP * nn); w.S2 = c.take<double>(BP * nn);
  w.S1i = c.take<double>(BP * nn); w.S2i = c.take<double>(BP * nn); w.L1i = c.take<double>(BP * nn); w.L2i = c.take<double>(BP * nn);
  w.LX1i = c.take<double>(BP * nn); w.LX2i = c.take<double>(BP * nn);
  w.Rd1 = c.take<double>(BP * nn); w.Rd2 = c.take<double>(BP * nn); w.T1 = c.take<double>(BP * nn); w.T2 = c.take<double>(BP * nn);
  w.dS1 = c.take<double>(BP * nn); w.dS2 = c.take<double>(BP * nn); w.dX1 = c.take<double>(BP * nn); w.dX2 = c.take<double>(BP * nn);
  w.c1 = c.take<double>(BP * nn); w.c2 = c.take<double>(BP * nn);
  w.dP = c.take<double>(BP * nxx);
  w.Wm = c.take<double>(BP * 4 * nn); w.eigmin = c.take<double>(BP * 4);
  w.KF = c.take<double>(BP * 12 * nxx); w.adjV = c.take<double>(BP * NADJ * nxx); w.adjE = c.take<double>(BP * NADJ * nxx);
  w.part = c.take<double>(BP * NPART); w.prob = c.take<double>((size_t)dm.B * PS); w.iprob = c.take<int>((size_t)dm.B * IS);
  w.D = c.take<double>(BP * bs); w.O = c.take<double>(BP * bs); w.F = c.take<double>(BP * bs);
  // float32 copies of the O factors for the single-precision updates of the early iterations (Opts::lowp_switch): Steps 1 / 2 on the tuned path (not Step 3), blocks the LDS-DMA kernels take
  w.O32 = (dm.nT == 0 && dm.n <= NMAX && dm.dp > 64 && dm.nt <= TRR_NT) ? c.take<float>(2 * BP * (size_t)dm.dp * ((dm.dp + 31) & ~31)) : nullptr;
  w.Linv = c.take<double>(BP * dm.nt * TB * TB); w.Ddiag = c.take<double>(BP * dm.dp);
  w.W3 = c.take<double>(BP * dm.dp * 3); w.U = c.take<double>(BP * dm.dp * 2); w.TU = c.take<double>(BP * dm.dp * 2);
  w.Z = c.take<double>(BP * dm.dp);
  w.active = c.take<int>(64);
  w.alist = c.take<int>((size_t)dm.B);
  w.flist = c.take<int>((size_t)dm.B);
  w.cr_orient = nullptr;
  w.trace = c.take<double>((size_t)dm.B * TRACE_LEN * TRACE_W);
  w.Hc = c.take<double>(BP * nn); w.dHc = c.take<double>(BP * nn); w.Pout = c.take<double>(BP * nxx);
  w.bscr = (dm.n > NMAX) ? c.take<double>(BP * BIG_SCR * nn) : nullptr;
  w.sscr = schur_in_lds(dm) ? nullptr : c.take<double>(BP * 10 * nxx);
  w.Dl = w.Ol = w.Fl = w.Linvl = w.KFl = w.adjVl = w.adjEl = w.W3l = w.Zl = w.Pprev = w.Pdef = w.phidef = w.Zdd = w.TUl = w.atdef = w.ddscr = nullptr; w.plist = w.pnext = nullptr;
  w.G = nullptr; w.ncnt = nullptr; w.rho = 0.0;
  w.phi = w.zph = w.dphi = w.dzph = w.corrp = w.pvec = w.psm = w.prs = w.Fg = nullptr;
  w.at = w.adt = w.aX = w.adX = w.acor = w.aSi = w.aLi = w.aLXi = w.asum = nullptr;
  if (dm.nr > 0) {
    const size_t g = dm.nr, z = dm.nz;
    w.phi = c.take<double>(BP * g); w.zph = c.take<double>(BP * g); w.dphi = c.take<double>(BP * g); w.dzph = c.take<double>(BP * g);
    w.corrp = c.take<double>(BP * g); w.Fg = c.take<double>(BP * g);
    w.pvec = c.take<double>(BP * 2 * g * (2 * dm.n + 2 * dm.nx)); w.psm = c.take<double>(BP * (z * z + 6 * z));
    w.asum = c.take<double>(BP * 5);
    if (dm.nr > NRS) w.prs = c.take<double>(BP * 4 * g * g);
    double* dG = c.take<double>(BP * g * dm.n);
    int32_t* dn = c.take<int32_t>(BP);
    if (h) { h->dG = dG; h->dncnt = dn; }
    if (dm.nz > dm.nr) {
      w.at = c.take<double>(BP * 2); w.adt = c.take<double>(BP * 2);
      w.aX = c.take<double>(BP * 2 * AE); w.adX = c.take<double>(BP * 2 * AE); w.acor = c.take<double>(BP * 2 * AE);
      w.aSi = c.take<double>(BP * 2 * AE); w.aLi = c.take<double>(BP * 2 * AE); w.aLXi = c.take<double>(BP * 2 * AE);
    }
  }
  w.t3th = w.t3z = w.t3dth = w.t3dz = w.t3cth = w.t3x = w.t3dx = w.t3cq = w.t3g = w.t3v = w.t3lam = nullptr;
  w.t3t = w.t3dt = w.t3beta = w.t3psi = w.t3phi = w.Tout = nullptr;
  if (dm.nT > 0) {
    const size_t m = dm.nT, m1 = m + 1;
    w.t3th = c.take<double>(BP * m); w.t3z = c.take<double>(BP * m); w.t3dth = c.take<double>(BP * m); w.t3dz = c.take<double>(BP * m); w.t3cth = c.take<double>(BP * m);
    w.t3x = c.take<double>(BP * m1); w.t3dx = c.take<double>(BP * m1); w.t3cq = c.take<double>(BP * m1); w.t3g = c.take<double>(BP * m1);
    w.t3v = c.take<double>(BP * m1); w.t3lam = c.take<double>(BP * m1);
    w.t3t = c.take<double>(BP); w.t3dt = c.take<double>(BP); w.t3beta = c.take<double>(BP);
    w.t3psi = c.take<double>(BP * nn); w.t3phi = c.take<double>(BP * nn); w.Tout = c.take<double>(BP * nn);
  }
  if (h) {
    h->dA = c.take<double>(BP * nxx); h->dB = c.take<double>(BP * dm.nx * std::max(dm.mb, 1)); h->dH = c.take<double>(BP * nn);
    h->d_info = c.take<double>((size_t)dm.B * TMPC_INFO_STRIDE); h->d_abk = c.take<double>((size_t)dm.B * 3);
    h->d_si = c.take<int32_t>((size_t)dm.B * 2);
  } else {
    c.take<double>(BP * nxx); c.take<double>(BP * dm.nx * std::max(dm.mb, 1)); c.take<double>(BP * nn);
    c.take<double>((size_t)dm.B * TMPC_INFO_STRIDE); c.take<double>((size_t)dm.B * 3); c.take<int32_t>((size_t)dm.B * 2);
  }
  return (c.off + 255) & ~(size_t)255;
}

// low words of the double-double planes of the tight mode (tmpc_dd.h), one slice per lane
static size_t carve_dd(WS& w, const Dims& dm, char* base) {
  Carver c{base, 0};
  const size_t BP = (size_t)dm.B * dm.p, nxx = (size_t)dm.nx * dm.nx, bs = (size_t)dm.dp * dm.dp;
  w.Dl = c.take<double>(BP * bs); w.Ol = c.take<double>(BP * bs); w.Fl = c.take<double>(BP * bs);
  w.Linvl = c.take<double>(BP * dm.nt * TB * TB);
  w.KFl = c.take<double>(BP * 12 * nxx); w.adjVl = c.take<double>(BP * NADJ * nxx); w.adjEl = c.take<double>(BP * NADJ * nxx);
  w.W3l = c.take<double>(BP * dm.dp * 3); w.Zl = c.take<double>(BP * dm.dp); w.TUl = c.take<double>(BP * dm.dp * 2);
  w.Pprev = c.take<double>(BP * nxx); w.Pdef = c.take<double>(BP * nxx);
  w.phidef = c.take<double>(BP * (size_t)dm.nr);
  w.Zdd = dm.nr > 0 ? c.take<double>(BP * 4 * (size_t)dm.n * dm.n) : nullptr;
  w.atdef = dm.nz > dm.nr ? c.take<double>(BP * 2) : nullptr;
  w.ddscr = (dm.n > NMAX) ? c.take<double>(BP * DD_SCR_MATS * 2 * (size_t)dm.n * dm.n) : nullptr;
  w.plist = c.take<int>((size_t)dm.B); w.pnext = c.take<int>((size_t)dm.B);
  return (c.off + 255) & ~(size_t)255;
}

// ---------------------------------------------------------------------------------- small output kernels
__global__ void k_output(WS w, Dims dm, double* alpha, double* beta, double* kappa, int32_t* status, int32_t* iters,
                         double* info) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= dm.B) return;
  const double* pr = w.prob + (size_t)b * PS;
  const int* ip = w.iprob + (size_t)b * IS;
  if (alpha) alpha[b] = pr[P_ALPHA_OUT];
  if (beta) beta[b] = pr[P_BETA];
  if (kappa) kappa[b] = pr[P_KAPPA];
  if (status) status[b] = ip[I_STATUS];
  if (iters) iters[b] = ip[I_ITERS];
  if (info) {
    double* o = info + (size_t)b * TMPC_INFO_STRIDE;
    o[0] = pr[P_S]; o[1] = pr[P_SBETA]; o[2] = pr[P_MINEIG_H]; o[3] = pr[P_MINEIG_HC]; o[4] = pr[P_MAXCOND];
    o[5] = pr[P_MU]; o[6] = pr[P_MUT]; o[7] = pr[P_PINF]; o[8] = pr[P_DINF]; o[9] = pr[P_RELGAP];
    o[10] = (double)ip[I_IPMSTATUS]; o[11] = (double)(ip[I_NSHIFT] + ip[I_CHOLBAD]); o[12] = (double)ip[I_NCENT];
    o[13] = (double)ip[I_EARLY]; o[14] = pr[P_STEPN]; o[15] = pr[P_MINPIV];
  }
}

// eig scan of arbitrary stage blocks (tmpc_eig_scan_host)
__global__ void __launch_bounds__(64) k_eig_scan(const double* H, double* out, int n) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int sid = blockIdx.x, lane = threadIdx.x;
  double* sH = sm; double* cs = sm + MS;
  const double* Hg = H + (size_t)sid * n * n;
  for (int e = lane; e < n * n; e += 64) { const int i = e / n, j = e - i * n; sH[i * LD + j] = 0.5 * (Hg[i * n + j] + Hg[j * n + i]); }
  wsync();
  jacobi_eigvals(sH, n, cs, lane);
  double lo = 1e300, hi = -1e300, amin = 1e300, amax = 0.0;
  if (lane < n) { const double ev = sH[lane * LD + lane]; lo = ev; hi = ev; const double a = fabs(ev); if (a != 0.0) { amin = a; amax = a; } }
  lo = wave_min(lo); hi = wave_max(hi); amin = wave_min(amin); amax = wave_max(amax);
  if (lane == 0) { double* o = out + (size_t)sid * 4; o[0] = lo; o[1] = hi; o[2] = amin; o[3] = amax; }
}

// dHc = sym(V' P+ V - E' P E) for arbitrary P (tmpc_supplement_batch_host); uses ws.V built from A,B
// dHc_k = sym(V' P_{k+1} V - E' P_k E  +  J_k' diag(w_k) J_k  +  T_k)   (convexifier.py:165-211: the rows of J are the
// equality-constraint Jacobian G_k with weights Fg_k followed by the active-constraint Jacobian C_k with weights F_k,
// zero-weight padding up to nr rows; T_k the free regularisation of Step 3).  J, wts, T may be null.
__global__ void __launch_bounds__(64) k_supplement(const double* A, const double* Bm, const double* P, double* dHc, Dims dm,
                                                   int nr, const double* J, const double* wts, const double* T) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int sid = blockIdx.x, lane = threadIdx.x;
  const int b = sid / dm.p, k = sid - b * dm.p;
  const int n = dm.n, nx = dm.nx, mb = dm.mb, nxx = nx * nx;
  double* sV = sm; double* sM = sm + MS; double* t0 = sm + 2 * MS; double* t1 = sm + 3 * MS; double* sZ = sm + 4 * MS;
  for (int e = lane; e < nx * n; e += 64) {
    const int i = e / n, j = e - i * n;
    sV[i * LD + j] = (j < nx) ? A[(size_t)sid * nxx + i * nx + j] : Bm[(size_t)sid * nx * mb + i * mb + (j - nx)];
  }
  for (int e = lane; e < n * n; e += 64) { const int i = e / n, j = e - i * n; sZ[i * LD + j] = 0.0; }
  wsync();
  const int kn = (k + 1 == dm.p) ? 0 : k + 1;
  build_M(sM, sV, t0, t1, sZ, P + (size_t)sid * nxx, P + (size_t)(b * dm.p + kn) * nxx, 0.0, n, nx, lane);
  if (J || T) {
    const double* Jg = J ? J + (size_t)sid * nr * n : nullptr;
    const double* wg = wts ? wts + (size_t)sid * nr : nullptr;
    for (int e = lane; e < n * n; e += 64) {
      const int i = e / n, j = e - i * n;
      double acc = sM[i * LD + j];
      if (Jg) for (int r = 0; r < nr; ++r) acc = fma(wg[r] * Jg[r * n + i], Jg[r * n + j], acc);
      if (T) acc += T[(size_t)sid * n * n + e];
      sM[i * LD + j] = acc;
    }
    wsync();
  }
  s2g_sym(dHc + (size_t)sid * n * n, sM, n, lane);
}

// Tracking-MPC reference (pmpc.py:961-974): W = sym(Hc)/ts, yref = wref - Hc^-1 q by Cholesky + two substitutions.
// One single-wave block per stage.
__global__ void __launch_bounds__(64) k_tracking_ref(const double* Hc, const double* q, const double* wref, double inv_ts,
                                                     double* W, double* yref, int* info, int n) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const size_t sid = blockIdx.x;
  const int lane = threadIdx.x;
  double* A = sm; double* v = sm + MS;
  const double* Hg = Hc + sid * n * n;
  for (int e = lane; e < n * n; e += 64) { const int i = e / n, j = e - i * n; A[i * LD + j] = 0.5 * (Hg[i * n + j] + Hg[j * n + i]); }
  if (lane < n) v[lane] = q[sid * n + lane];
  wsync();
  if (W) for (int e = lane; e < n * n; e += 64) { const int i = e / n, j = e - i * n; W[sid * n * n + e] = A[i * LD + j] * inv_ts; }
  wsync();
  const int nbad = chol_lower(A, n, lane);
  for (int j = 0; j < n; ++j) {             // L z = q
    if (lane == j) v[j] /= A[j * LD + j];
    wsync();
    if (lane > j && lane < n) v[lane] -= A[lane * LD + j] * v[j];
    wsync();
  }
  for (int j = n - 1; j >= 0; --j) {        // L' x = z
    if (lane == j) v[j] /= A[j * LD + j];
    wsync();
    if (lane < j) v[lane] -= A[j * LD + lane] * v[j];
    wsync();
  }
  if (lane < n) yref[sid * n + lane] = wref[sid * n + lane] - v[lane];
  if (info && lane == 0) info[sid] = nbad;
}

// Producer side (pocp.py:322-361): one wave per stage.  Active-set extraction = order-preserving compaction of the rows of C_k with
// |mu_k,i| > thr (wave ballot + prefix count), q_k = -mu_k' C_k, and the diagonal n x n block of the Lagrangian Hessian.
__global__ void __launch_bounds__(64) k_pack_sens(const double* C, const double* mu, const double* Hbig, double thr, int nh, int ncmax, int p, int n,
                                                  double* CAs, int32_t* ncnt, int32_t* idx, double* q, double* Hst) {
  const size_t sid = blockIdx.x;
  const int lane = threadIdx.x;
  const size_t b = sid / p; const int k = (int)(sid - b * p);
  if (C) {
    const double* Ck = C + sid * nh * n; const double* mk = mu + sid * nh;
    double* out = CAs + sid * (size_t)ncmax * n;
    int cnt = 0;
    for (int i0 = 0; i0 < nh; i0 += 64) {
      const int i = i0 + lane;
      const bool act = (i < nh) && (fabs(mk[i]) > thr);
      const unsigned long long m = __ballot(act);
      const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
      if (act) {
        if (idx) idx[sid * nh + pos] = i;
        if (pos < ncmax) for (int c = 0; c < n; ++c) out[(size_t)pos * n + c] = Ck[(size_t)i * n + c];
      }
      cnt += __popcll(m);
    }
    for (int e = lane + (cnt < ncmax ? cnt : ncmax) * n; e < ncmax * n; e += 64) out[e] = 0.0;      // zero padding
    if (idx) for (int i = cnt + lane; i < nh; i += 64) idx[sid * nh + i] = -1;
    if (lane == 0) ncnt[sid] = cnt;                                                                  // (> ncmax: the caller's padding was too small)
    if (q) for (int c = lane; c < n; c += 64) {
      double acc = 0.0;
      for (int i = 0; i < nh; ++i) acc = fma(mk[i], Ck[(size_t)i * n + c], acc);
      q[sid * n + c] = -acc;
    }
  } else if (q) {
    for (int c = lane; c < n; c += 64) q[sid * n + c] = 0.0;                                         // pocp.py:361: zeros without path constraints
  }
  if (Hbig) {
    const size_t ld = (size_t)p * n;
    const double* Hb = Hbig + b * ld * ld + ((size_t)k * n) * ld + (size_t)k * n;
    for (int e = lane; e < n * n; e += 64) { const int i = e / n, j = e - i * n; Hst[sid * n * n + e] = Hb[(size_t)i * ld + j]; }
  }
}

// ---------------------------------------------------------------------------------- debug kernels
__global__ void __launch_bounds__(64) k_debug_min_eig(const double* W, double* out, int n) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int mid = blockIdx.x, lane = threadIdx.x;
  double* A = sm; double* cs = A + MS;
  const double* Wg = W + (size_t)mid * n * n;
  for (int e = lane; e < n * n; e += 64) { const int i = e / n, j = e - i * n; A[i * LD + j] = 0.5 * (Wg[i * n + j] + Wg[j * n + i]); }
  wsync();
  const double lo = tridiag_min_eig(A, n, cs, lane);
  if (lane == 0) out[mid] = lo;
}

__global__ void __launch_bounds__(64) k_debug_min_eig_lane(const double* W, double* out, int n, int nmat) {      // one thread per matrix (lane_min_eig8, n <= 8)
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i < nmat) out[i] = lane_min_eig8(W + (size_t)i * n * n, n);
}

template <bool USE_MFMA>
__global__ void __launch_bounds__(256, 2) k_debug_gemm(double* C, const double* A, const double* B, int M, int N, int K, int mode, int lower) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  wg_gemm_nt<USE_MFMA>(C, N, A, K, B, K, M, N, K, mode, lower != 0, lds);
}
// the LDS-DMA tile core of the batched factorisation kernels (tmpc_gemm_dma.h): one workgroup per 64 x 64 tile, optional second operand
// pair as one K stream (A2 = A, B2 = B shifted by K2 columns: the caller passes K = K1 + K2 and the split)
__global__ void __launch_bounds__(256, 4) k_debug_gemm_dma(double* C, const double* A, const double* B, int M, int N, int K, int mode, int lower, int k1) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int nmc = (N + 63) / 64;
  const int tm = blockIdx.x / nmc, tn = blockIdx.x - tm * nmc;
  if (lower && tn > tm) return;
  const int m0 = 64 * tm, n0 = 64 * tn;
  const int Mt = M - m0 < 64 ? M - m0 : 64, Nt = N - n0 < 64 ? N - n0 : 64;
  const double* A0 = A + (size_t)m0 * K; const double* B0 = B + (size_t)n0 * K;
  if (k1 > 0 && k1 < K && 2 * k1 == K)          // two operand pairs of equal K as ONE stream
    wg_tile_dma<UPD_DMA_DEPTH>(C + (size_t)m0 * N + n0, N, A0, B0, A0 + k1, B0 + k1, K, Mt, Nt, k1, mode, (lower && tm == tn) ? 0 : GM_NOTRI, blockIdx.x, lds);
  else
    wg_tile_dma<UPD_DMA_DEPTH>(C + (size_t)m0 * N + n0, N, A0, B0, nullptr, nullptr, K, Mt, Nt, K, mode, (lower && tm == tn) ? 0 : GM_NOTRI, blockIdx.x, lds);
}
// ---------------------------------------------------------------------------------- launch configuration
static size_t slots_bytes(int s) { return (size_t)s * MS * sizeof(double); }
template <int PART> static size_t schur_lds(const Dims& dm) {
  if (!schur_in_lds(dm)) return (size_t)dm.d * sizeof(unsigned) + 64;      // factor records in global scratch (k_schur<PART, true>)
 return (size_t)sch_rec<PART>() * dm.nx * dm.nx * sizeof(double) + (size_t)dm.d * sizeof(unsigned) + 64; }
static size_t factor_lds() { return (size_t)FACT_LDS_DOUBLES * sizeof(double); }
static size_t solve_lds(const Dims& dm) { return (size_t)cr_solve_lds_doubles(dm.dp) * sizeof(double); }

// kernels whose dynamic LDS exceeds the 64 KB default: the attribute is per device
static int set_lds_attrs(int device) {
  static bool done[64] = {false};
  if (device >= 0 && device < 64 && done[device]) return TMPC_OK;
  const int big = 160 * 1024;
  HIPCHK(hipFuncSetAttribute((const void*)k_stage_pre<256>, hipFuncAttributeMaxDynamicSharedMemorySize, big - 4096));     // (these carry a few bytes of static LDS: the block reductions)
  HIPCHK(hipFuncSetAttribute((const void*)k_stage_rhs<256>, hipFuncAttributeMaxDynamicSharedMemorySize, big - 4096));     // (these carry a few bytes of static LDS: the block reductions)
  HIPCHK(hipFuncSetAttribute((const void*)k_stage_dir<256>, hipFuncAttributeMaxDynamicSharedMemorySize, big - 4096));     // (these carry a few bytes of static LDS: the block reductions)
  HIPCHK(hipFuncSetAttribute((const void*)k_final_stage, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_schur<0>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_schur<1>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_cr_potrf<true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_cr_potrf<false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_cr_trsm<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_cr_trsm<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_cr_trsm_dma, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_cr_potrf_dma, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_cr_update<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_cr_update<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_cr_fwd_diag, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_cr_fwd_off, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_cr_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_cr_small_solve, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_cr_small_factor, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_ipm_small, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_debug_gemm<true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_debug_gemm<false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_supplement, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_phi_pre<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_phi_pre<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_phi_pre<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_phi_pre<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_phi_rhs<false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_phi_dir<false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_phi_rhs<true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_phi_dir<true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_t3_schur<false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_t3_schur<true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_dd_potrf, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_dd_aug_fill, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_dd_trsm, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_dd_update, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_dd_fwd_diag, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_dd_fwd_off, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_dd_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_dd_images, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_dd_schur<0>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_dd_schur<1>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_dd_schur<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_dd_schur<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute((const void*)k_dd_polish_pre, hipFuncAttributeMaxDynamicSharedMemorySize, big - 4096));
  HIPCHK(hipFuncSetAttribute((const void*)k_polish_step, hipFuncAttributeMaxDynamicSharedMemorySize, big - 4096));
  if (device >= 0 && device < 64) done[device] = true;
  return TMPC_OK;
}

// ---------------------------------------------------------------------------------- block factorisation / solves (tmpc_cr.h)
static unsigned cr_grid(long items) { return (unsigned)((items + 7) / 8 * 8); }
// host-side bits of Dims::flags (bit 0 = TMPC_FLAG_NO_MFMA is the only one device code reads)
constexpr int DF_NO_SMALL = 2;   // tmpc_set_tuning(TMPC_TUNE_SMALL_BLOCKS, 0): the batched launch sequence also for dp = 16
constexpr int DF_LOWP = 8;       // single-precision updates may be on for some problems (Opts::lowp_switch > 0 and the handle has the float32 copies): cr_factor launches their forward-substitution steps
constexpr int DF_NO_DMA = 4;     // TMPC_DEBUG_FLAG_NO_DMA: the register-staged factorisation kernels (the path of blocks wider than 320) for every block size

// Small blocks (dp = 16): one kernel per factorisation / per solve instead of a launch sequence per level (tmpc_cr_small.h); TMPC_SMALL=0: off
static bool cr_small_levels(const Dims& dm, const CrSched& sc, CrLevs* out) {
  if ((dm.flags & DF_NO_SMALL) || dm.dp != 16 || (dm.flags & 1) || dm.p > CRS_PMAX || (int)sc.lev.size() > CRS_MAXLEV) return false;
  if (out) {
    out->n = (int)sc.lev.size();
    for (int l = 0; l < out->n; ++l) { out->v[4 * l] = sc.lev[l].eoff; out->v[4 * l + 1] = sc.lev[l].nelim; out->v[4 * l + 2] = sc.lev[l].uoff; out->v[4 * l + 3] = sc.lev[l].nupd; }
  }
  return true;
}

// Cholesky of the block-cyclic-tridiagonal Schur matrices of the `count` problems listed in alist (device), level by level.
// rs / mt: rows per workgroup of the triangular solves / edge of the output tile of the updates; 0 = by the amount of work
// (whole 128-wide pieces while every CU still gets several workgroups, 64 otherwise).  The result does not depend on them.
static void cr_factor(const WS& w, const Dims& dm, const CrSched& sc, const int* d_sched, const int* alist, int count, hipStream_t st,
                      int rs_opt, int mt_opt, std::vector<hipEvent_t>* kev = nullptr, int* nkev = nullptr, int fuse_fwd1 = 0, int nlowp = 0) {
  // nlowp: how many of the `count` problems run their Schur-complement updates in single precision (exact: k_ctrl_d / k_init_prob counted them)
  const CrDev cd = cr_dev(sc, d_sched, alist);
  int ke = 0;
  // profile mode: an event before and after every launch; pair i belongs to class i % 4 (potrf, trsm, update in fp64, update in float32)
  auto mark = [&](int cls) {
#ifdef TMPC_CYCLE_PROF
    { static int cls_ids[4] = {0, 1, 2, 3}; hipMemcpyToSymbolAsync(HIP_SYMBOL(tmpc::g_prof_cls), &cls_ids[cls + 1], sizeof(int), 0, hipMemcpyHostToDevice, st); }
#endif
    if (!kev) return;
    while ((int)kev->size() <= ke) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; kev->push_back(e); }
    (void)cls;
    hipEventRecord((*kev)[ke++], st);
  };
  const bool mf = !(dm.flags & 1);
  if (sc.prep) hipLaunchKernelGGL(k_cr_prep, dim3(count), dim3(256), 0, st, w, dm, cd, sc.prep);
  {
    CrLevs lv;
    if (!fuse_fwd1 && cr_small_levels(dm, sc, &lv)) {
      mark(0);
      hipLaunchKernelGGL(k_cr_small_factor, dim3(count), dim3(CRS_NT), (size_t)crs_factor_lds_doubles() * sizeof(double), st, w, dm, cd, lv);
      mark(0);
      if (nkev) *nkev = ke;
      return;
    }
  }
  const int nt64 = (dm.dp + 63) / 64;
  for (const CrLevel& lv : sc.lev) {
    // product path: the LDS-DMA kernels (blocks up to 320 wide); TMPC_FACTOR_DMA=0 or the no-MFMA flag: the register-staged core
    const bool use_dma = !(dm.flags & DF_NO_DMA), potrf_dma = use_dma;
    mark(0);
    if (mf && use_dma && potrf_dma && dm.nt <= TRR_NT)
      hipLaunchKernelGGL(k_cr_potrf_dma, dim3(cr_grid((long)count * lv.nelim)), dim3(256), (size_t)potrf_dma_lds_doubles() * sizeof(double), st, w, dm, cd, lv.eoff, lv.nelim, count);
    else if (mf) hipLaunchKernelGGL(k_cr_potrf<true>, dim3(cr_grid((long)count * lv.nelim)), dim3(256), factor_lds(), st, w, dm, cd, lv.eoff, lv.nelim, count);
    else hipLaunchKernelGGL(k_cr_potrf<false>, dim3(cr_grid((long)count * lv.nelim)), dim3(256), factor_lds(), st, w, dm, cd, lv.eoff, lv.nelim, count);
    mark(0);
    // fused forward substitution of pass 1 (see k_cr_update_dma): z_i <- L_i^-1 z_i for this level's nodes as soon as L_i exists
    if (fuse_fwd1) hipLaunchKernelGGL(k_cr_fwd_diag, dim3(cr_grid((long)count * lv.nelim)), dim3(256), solve_lds(dm), st, w, dm, cd, lv.eoff, lv.nelim, count, 1);
    if (lv.nupd == 0) { mark(1); mark(1); mark(2); mark(2); mark(3); mark(3); continue; }                  // last node: nothing left to update
    const long work64 = (long)count * lv.nelim * 2 * nt64;                 // 64-row strips of this level
    const int rs = rs_opt > 0 ? rs_opt : (work64 >= 16384 ? 128 : 64);
    const int mt = mt_opt > 0 ? mt_opt : 64;      // one 64 x 64 tile per workgroup: ~64 consecutive items per XCD span the tiles of 1-2 nodes, whose O blocks fit that L2
                                                  // (the 128 x 128 / 512-thread shape of wg_gemm_nt was measured slower here: 102 vs 84.5 ms per phase)
    const int nstrip = (dm.dp + rs - 1) / rs, nm = (dm.dp + mt - 1) / mt;
    const long it_trsm = (long)count * lv.nelim * 2 * nstrip;
    const long it_upd = (long)count * ((long)lv.nupd * (nm * (nm + 1) / 2) + (long)lv.nelim * nm * nm);
    mark(1);
    if (mf && use_dma && dm.nt <= TRR_NT) {
      const size_t trsm_lds = (size_t)trd_lds_doubles() * sizeof(double);
      hipLaunchKernelGGL(k_cr_trsm_dma, dim3(cr_grid((long)count * lv.nelim * 2 * nt64)), dim3(256), trsm_lds, st, w, dm, cd, lv.eoff, lv.nelim, count);
    }
    else if (mf) hipLaunchKernelGGL((k_cr_trsm<true, 2>), dim3(cr_grid(it_trsm)), dim3(256), factor_lds(), st, w, dm, cd, lv.eoff, lv.nelim, count, rs);
    else hipLaunchKernelGGL((k_cr_trsm<false, 2>), dim3(cr_grid(it_trsm)), dim3(256), factor_lds(), st, w, dm, cd, lv.eoff, lv.nelim, count, rs);
    mark(1); mark(2);
    if (mf && use_dma) {
      const int nm64 = (dm.dp + 63) / 64;
      const long it_dma = (long)count * ((long)lv.nupd * (nm64 * (nm64 + 1) / 2) + (long)lv.nelim * nm64 * nm64);
      if (nlowp < count)           // (each kernel leaves the other's problems alone)
        hipLaunchKernelGGL(k_cr_update_dma, dim3(cr_grid(it_dma)), dim3(256), (size_t)dma_lds_doubles<UPD_DMA_DEPTH>() * sizeof(double), st, w, dm, cd, lv.eoff, lv.nelim, lv.uoff, lv.nupd, count, fuse_fwd1);
      mark(2); mark(3);
      if (nlowp > 0)               // no right-hand-side mini slabs in the float32 tile: 32 KB of LDS, five workgroups per CU
        hipLaunchKernelGGL(k_cr_update_dma_f32, dim3(cr_grid(it_dma)), dim3(256), (size_t)UPD_DMA_DEPTH * DMA_SLAB * sizeof(double), st, w, dm, cd, lv.eoff, lv.nelim, lv.uoff, lv.nupd, count);
      mark(3);
    }
    else if (mf) { hipLaunchKernelGGL((k_cr_update<true, 2>), dim3(cr_grid(it_upd)), dim3(256), factor_lds(), st, w, dm, cd, lv.eoff, lv.nelim, lv.uoff, lv.nupd, count, mt); mark(2); mark(3); mark(3); }
    else { hipLaunchKernelGGL((k_cr_update<false, 2>), dim3(cr_grid(it_upd)), dim3(256), factor_lds(), st, w, dm, cd, lv.eoff, lv.nelim, lv.uoff, lv.nupd, count, mt); mark(2); mark(3); mark(3); }
    // problems whose update tiles ran in single precision carried no right-hand sides: their z_s -= O_s z_i of the fused forward sweep here, fp64 products on the float32 O copies (k_cr_fwd_off with lowp_only: the other problems are skipped)
    if (fuse_fwd1 && nlowp > 0 && mf && use_dma && lv.nupd)
      hipLaunchKernelGGL(k_cr_fwd_off, dim3(cr_grid((long)count * lv.nupd)), dim3(256), solve_lds(dm), st, w, dm, cd, lv.uoff, lv.nupd, count, 1, 1);
  }
  if (nkev) *nkev = ke;
}

// forward and backward substitution with that factor for the right-hand sides of `pass` (W3 / Z, see cr_nc)
static void cr_solve(const WS& w, const Dims& dm, const CrSched& sc, const int* d_sched, const int* alist, int count, hipStream_t st, int pass, bool skip_fwd = false) {
  const CrDev cd = cr_dev(sc, d_sched, alist);
  {
    CrLevs lv;
    if (!skip_fwd && cr_small_levels(dm, sc, &lv)) {
      hipLaunchKernelGGL(k_cr_small_solve, dim3(count), dim3(CRS_NT), (size_t)crs_solve_lds_doubles(dm.p) * sizeof(double), st, w, dm, cd, lv, pass);
      return;
    }
  }
  const size_t lds = solve_lds(dm);
  for (const CrLevel& lv : sc.lev) {
    if (skip_fwd) break;                 // the forward sweep ran inside the factorisation (cr_factor, fuse_fwd1)
    hipLaunchKernelGGL(k_cr_fwd_diag, dim3(cr_grid((long)count * lv.nelim)), dim3(256), lds, st, w, dm, cd, lv.eoff, lv.nelim, count, pass);
    if (lv.nupd) hipLaunchKernelGGL(k_cr_fwd_off, dim3(cr_grid((long)count * lv.nupd)), dim3(256), lds, st, w, dm, cd, lv.uoff, lv.nupd, count, pass, 0);
  }
  for (size_t l = sc.lev.size(); l-- > 0;) {
    const CrLevel& lv = sc.lev[l];
    hipLaunchKernelGGL(k_cr_bwd, dim3(cr_grid((long)count * lv.nelim)), dim3(256), lds, st, w, dm, cd, lv.eoff, lv.nelim, count, pass);
  }
}

// ---- the same in double-double (tight mode, tmpc_dd.h): one launch per phase and level, no small-block or fused forms
static bool ddschur_in_lds(const Dims& dm) { return (size_t)2 * ddsch_mats<0>() * dm.nx * dm.nx * sizeof(double) + (size_t)dm.d * sizeof(unsigned) + 64 <= 158 * 1024; }
template <int PART> static size_t ddschur_lds(const Dims& dm) {
  if (!ddschur_in_lds(dm)) return (size_t)dm.d * sizeof(unsigned) + 64;      // factors read from global memory (k_dd_schur<PART, true>)
  return (size_t)2 * ddsch_mats<PART>() * dm.nx * dm.nx * sizeof(double) + (size_t)dm.d * sizeof(unsigned) + 64;
}
// the dd assembly of D_k and of the coupling blocks for the stages of `grid_stages` problems-times-stages
static void dd_schur_launch(const WS& w, const Dims& dm, int grid_stages, hipStream_t st) {
  if (ddschur_in_lds(dm)) {
    hipLaunchKernelGGL(k_dd_schur<0>, dim3(grid_stages), dim3(256), ddschur_lds<0>(dm), st, w, dm);
    hipLaunchKernelGGL(k_dd_schur<1>, dim3(grid_stages), dim3(256), ddschur_lds<1>(dm), st, w, dm);
  } else {
    hipLaunchKernelGGL((k_dd_schur<0, true>), dim3(grid_stages), dim3(256), ddschur_lds<0>(dm), st, w, dm);
    hipLaunchKernelGGL((k_dd_schur<1, true>), dim3(grid_stages), dim3(256), ddschur_lds<1>(dm), st, w, dm);
  }
}
// LDS of the stage-level dd kernels: their matrices live in LDS slots at n <= 32, in WS::ddscr above
static size_t dd_stage_lds(const Dims& dm, int slots) { return dm.n > NMAX ? 64 : slots_bytes(slots); }
static void dd_factor(const WS& w, const Dims& dm, const CrSched& sc, const int* d_sched, const int* alist, int count, hipStream_t st) {
  const CrDev cd = cr_dev(sc, d_sched, alist);
  const size_t lds = (size_t)DD_FACT_LDS * sizeof(double);
  if (sc.prep) hipLaunchKernelGGL(k_dd_prep, dim3(count), dim3(256), 0, st, w, dm, cd, sc.prep);
  const int nm = (dm.dp + 63) / 64;
  for (const CrLevel& lv : sc.lev) {
    hipLaunchKernelGGL(k_dd_potrf, dim3(cr_grid((long)count * lv.nelim)), dim3(256), lds, st, w, dm, cd, lv.eoff, lv.nelim, count);
    if (lv.nupd == 0) continue;
    hipLaunchKernelGGL(k_dd_trsm, dim3(cr_grid((long)count * lv.nelim * 2 * nm)), dim3(256), lds, st, w, dm, cd, lv.eoff, lv.nelim, count);
    const long items = (long)count * ((long)lv.nupd * (nm * (nm + 1) / 2) + (long)lv.nelim * nm * nm);
    hipLaunchKernelGGL(k_dd_update, dim3(cr_grid(items)), dim3(256), lds, st, w, dm, cd, lv.eoff, lv.nelim, lv.uoff, lv.nupd, count);
  }
}
static int dd_solve(const WS& w, const Dims& dm, const CrSched& sc, const int* d_sched, const int* alist, int count, hipStream_t st, int pass, int nb_all) {
  const CrDev cd = cr_dev(sc, d_sched, alist);
  const size_t lds = (size_t)dd_solve_lds_doubles(dm.dp) * sizeof(double);
  // the right-hand sides arrive as fp64 numbers: low words zero
  HIPCHK(hipMemsetAsync(w.W3l, 0, (size_t)nb_all * dm.p * dm.dp * 3 * sizeof(double), st));
  HIPCHK(hipMemsetAsync(w.Zl, 0, (size_t)nb_all * dm.p * dm.dp * sizeof(double), st));
  for (const CrLevel& lv : sc.lev) {
    hipLaunchKernelGGL(k_dd_fwd_diag, dim3(cr_grid((long)count * lv.nelim)), dim3(256), lds, st, w, dm, cd, lv.eoff, lv.nelim, count, pass);
    if (lv.nupd) hipLaunchKernelGGL(k_dd_fwd_off, dim3(cr_grid((long)count * lv.nupd)), dim3(256), lds, st, w, dm, cd, lv.uoff, lv.nupd, count, pass);
  }
  for (size_t l = sc.lev.size(); l-- > 0;) {
    const CrLevel& lv = sc.lev[l];
    hipLaunchKernelGGL(k_dd_bwd, dim3(cr_grid((long)count * lv.nelim)), dim3(256), lds, st, w, dm, cd, lv.eoff, lv.nelim, count, pass);
  }
  return TMPC_OK;
}

// the per-stage kernels run with four waves per stage (the one-wave form of round 1 -- profiles/r2*: 7.6 % slower -- went with its environment switch in round 4)
#define TMPC_STAGE_LAUNCH(K, LDSB, ST, ...)                                                          \
  do {                                                                                               \
    hipLaunchKernelGGL((K<256>), dim3(BP), dim3(256), LDSB, ST, __VA_ARGS__);                        \
  } while (0)

// one chunk (nb = actual number of problems in this chunk, <= capacity); inputs already on device
static int run_chunk(tmpc_handle* h, Lane* ln, int nb, const double* dA, const double* dB, const double* dH, hipStream_t st,
                     const double* dG = nullptr, const int32_t* dncnt = nullptr, double rho = 0.0, bool step3 = false) {
  Dims dm = h->dm;
  dm.B = nb;
  if (!step3) dm.nT = 0;                  // a handle with room for T also serves the other models
  // a handle created with room for G / C rows also serves calls without them; Step 2 (constr) when the C counts are given
  if (!dG) { dm.ng = 0; dm.nr = 0; dm.nz = 0; }
  // rho == 0 with C counts: the beta-only objective (the other reading of convexifier.py:276-283, see tunempc_hip.h): the rows of C_k are
  // cost-free like those of G_k -- ragged rows, no norm terms (no arrow blocks, no epigraph variables)
  dm.constr = (dG && dncnt && rho > 0.0) ? 1 : 0;
  if (dG && !dncnt) { dm.nr = dm.ng; dm.nz = dm.ng; }
  if (dG && dncnt && !dm.constr) dm.nz = dm.nr;
  // stage-local multipliers ride inside the blocks (block size d + nz)
  const bool eq = dm.nr > 0;
  const bool t3 = dm.nT > 0;
  dm.dp = (dm.d + (eq ? dm.nz : 0) + (t3 ? dm.nT + 1 : 0) + 15) / 16 * 16;
  dm.nt = (dm.dp + TB - 1) / TB;
  dm.flags = (h->flags & TMPC_FLAG_NO_MFMA) | (h->tune_small ? 0 : DF_NO_SMALL) | ((h->flags & TMPC_DEBUG_FLAG_NO_DMA) ? DF_NO_DMA : 0);
  const size_t t3_lds = (size_t)(3 * (dm.nT + 1) + 8) * sizeof(double);
  const bool big = dm.n > NMAX || (h->flags & TMPC_DEBUG_FLAG_GENERIC_STAGE);      // generic per-stage kernels (tmpc_big.h)
  const size_t t3_schur_lds = (size_t)((big ? 0 : 9 * 32 * T3_LD) + 2 * (dm.nT + 1)) * sizeof(double) + (size_t)(2 * (dm.nT + 1) + 2 * (dm.d + 1)) * sizeof(short) + 64;
  if (big && !ln->ws.bscr) {                 // (debug flag at n <= 32: the scratch is not part of the workspace)
    if (hipMalloc(&ln->big_scr, (size_t)h->dm.B * dm.p * BIG_SCR * dm.n * dm.n * sizeof(double)) != hipSuccess) { snprintf(g_err, sizeof(g_err), "hipMalloc of the generic-stage scratch failed"); return TMPC_E_NOMEM; }
    ln->ws.bscr = (double*)ln->big_scr;
  }
  WS wall = ln->ws;                      // view over ALL problems of the chunk (init / final kernels)
  wall.A = dA; wall.Bm = dB; wall.H = dH; wall.G = dG; wall.ncnt = dncnt; wall.rho = rho;
  wall.cr_orient = h->d_sched + h->sched.elim.size() + h->sched.upd.size();
  WS w = wall;                           // view over the problems still iterating (per-iteration kernels map blockIdx through alist)
  w.A = nullptr; w.Bm = nullptr; w.H = nullptr;      // the caller's input buffers are read by the init / final kernels only (through `wall`): left out of the iteration view, whose
                                         // bytes are the key of the iteration graphs below, a device call whose inputs move does not capture a new graph per call
  int* alist = wall.alist;
  int* flist = wall.flist;
  wall.alist = nullptr;
  Opts o = h->opt;
  o.tight = 0; o.tight_tol = 0.0;
  // (chord steps also with stage-local multipliers since round 3: their rows are part of the frozen blocks like everything else)
  o.fast_exit = (h->flags & TMPC_FLAG_FAST_EXIT) ? 1 : 0;
  if ((h->flags & (TMPC_FLAG_NO_MFMA | TMPC_DEBUG_FLAG_NO_DMA)) || !w.O32 || t3 || dm.dp <= 64 || dm.nt > TRR_NT) o.lowp_switch = 0.0;
  if (o.lowp_switch > 0.0) {
    dm.flags |= DF_LOWP;
    // the float32 copies are laid out by this call's block width; where the last call used another one (a handle with room for rows serving the plain model, or
    // the other way round) its data sit where the zero padding of this call's rows must be
    if (ln->o32_dp != 0 && ln->o32_dp != dm.dp) HIPCHK(hipMemsetAsync(w.O32, 0, 2 * (size_t)h->dm.B * h->dm.p * h->dm.dp * ((h->dm.dp + 31) & ~31) * sizeof(float), st));
    ln->o32_dp = dm.dp;
  }
  WS wf = w; wf.alist = flist;            // view over the problems that get a new factorisation this iteration
  const int BPall = nb * dm.p;
  const bool prof = (h->flags & TMPC_FLAG_PROFILE) != 0;
  bool use_graph = h->tune_graph && dm.dp <= 64 && !(h->flags & (TMPC_FLAG_PROFILE | TMPC_DEBUG_FLAG_STOP_ASSEMBLED)) && st == ln->st;
  const int reg_max = (h->flags & TMPC_DEBUG_FLAG_NO_LIFT) ? 0 : REG_MAX;      // (debug: no diagonal lifts, frozen pivots while centering go straight to the back-off-and-step route)
  HIPCHK(hipMemsetAsync(w.active, 0, 5 * sizeof(int), st));
  HIPCHK(hipMemsetAsync(w.trace, 0, (size_t)nb * TRACE_LEN * TRACE_W * sizeof(double), st));
  const size_t big_lds = (size_t)BIG_EIG_LDS * sizeof(double);
  if (big) hipLaunchKernelGGL(kb_init_stage, dim3(BPall), dim3(256), big_lds, st, wall, dm);
  else hipLaunchKernelGGL(k_init_stage, dim3(BPall), dim3(64), slots_bytes(2), st, wall, dm);
  {
    WS wi = wall; wi.alist = alist;      // k_init_prob fills the list (its own index is blockIdx: alist is only written)
    hipLaunchKernelGGL(k_init_prob, dim3(nb), dim3(64), 0, st, wi, dm);
  }
  hipLaunchKernelGGL(k_init_state, dim3(BPall), dim3(64), 0, st, wall, dm);
  if (eq) hipLaunchKernelGGL(k_phi_init, dim3(BPall), dim3(64), 0, st, wall, dm);
  if (t3) hipLaunchKernelGGL(k_t3_init, dim3(BPall), dim3(64), 0, st, wall, dm);
  int cnt[5] = {0, 0, 0, 0, 0};           // problems still iterating / of which need a factorisation / handed to the polish / still polishing / to factor with single-precision updates
  HIPCHK(hipMemcpyAsync(cnt, w.active, 5 * sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  int active = cnt[0], nfac = cnt[0], nlowp = cnt[4];
  // main-phase cap + a centering budget per target the back-off may visit (the CPU restatement's bound).  Only members that keep backing
  // off get there -- a degenerate Step 3 member of the fuzz (kappa* = 1, damped centering steps at every target) needs ~110 iterations;
  // until round 3 the wave stopped at max_iter + center_iter + 2 = 64 and such a member came back Feasible.
  const int cap = o.max_iter + o.center_iter * (MUT_BACKOFF_MAX + 1) + 2;
  int it = 0;
  // Small problems (one 16 x 16 tile per Schur block, n <= 8, plain model): the whole loop below as ONE launch, a 16-wave workgroup per problem
  // (tmpc_persist.h; tmpc_set_tuning(TMPC_TUNE_PERSISTENT, 0): the launch sequence)
  {
    CrLevs plv;
    // Where it pays (scripts/persist_check.py, profiles/r5_persist_check.txt): the workgroup of a problem takes its stages sixteen at a time, so a
    // single long problem is slower than on the launch sequence, which spreads them over the CUs (p = 30, batch 1: 6.9 against 4.6 ms); it wins for short
    // periods at any batch (p <= 8: 1.05 - 1.2 x from 1 to 94 problems; p = 16: 0.9 - 1.0 x, profiles/r5_persist_short_periods.txt) and from ~96 problems in flight
    // on the chip whatever the period (p = 20 ... 100: 0.8 - 0.9 x at 64, 1.0 at 96, 1.2 at 128, 1.7 - 1.9 x at 256 = one workgroup per CU;
    // profiles/r5_persist_sweep.txt).  A wave is split over the lanes of the handle, each of which makes this decision for its share.
    // tune_persist: 0 never, 1 by this rule, 2 whenever the shape allows it.
    const bool pays = h->tune_persist >= 2 || dm.p <= 8 || active * h->nlanes >= 96;
    const bool persist = h->tune_persist && pays && !eq && !t3 && !big && dm.dp == 16 && dm.n <= PK_NMAX && !h->tight && !prof &&
                         !(h->flags & TMPC_DEBUG_FLAG_STOP_ASSEMBLED) && cr_small_levels(dm, h->sched, &plv);
    if (persist && active > 0) {
      const CrDev cd = cr_dev(h->sched, h->d_sched, alist);
      hipLaunchKernelGGL(k_ipm_small, dim3(active), dim3(CRS_NT), (size_t)pk_lds_doubles(dm.p) * sizeof(double), st, w, dm, o, cd, plv, h->sched.prep, reg_max,
                         h->tune_pretest ? o.chord_step : -1.0, cap);
      HIPCHK(hipGetLastError());
      ln->prof[15] += (double)active;        // (counted with or without TMPC_FLAG_PROFILE: which path a wave took -- bench.py's label)
      active = 0; nfac = 0;
    }
  }
  // one interior-point loop; ddm: the tight phase (tmpc_dd.h) -- assembly, factorisation and substitutions in double-double, every problem of the list
  auto ipm_loop = [&](const Opts& o, bool ddm, int cap) -> int {
  while (active > 0 && it < cap) {
    const int BP = active * dm.p;        // grids cover the problems still iterating only
    // Launch-bound shapes (blocks of one tile: the reference's own examples; ~35 dependent launches of a few microseconds each per iteration): the launch
    // sequence of an iteration is captured once per distinct argument set as a hipGraph and replayed -- one submission instead of ~35.  The key holds
    // every launch argument (workspace views, dimensions, options, list lengths), so a replay enqueues exactly what the code below would.
    hipGraphExec_t gexec = nullptr;
    std::string gkey;
    bool capturing = false;
    // an error return out of the captured region (HIPCHK) must not leave the lane's stream in capture mode -- every later call on it would fail
    struct CaptureGuard {
      hipStream_t st; bool* on;
      ~CaptureGuard() { if (*on) { hipGraph_t g_ = nullptr; (void)hipStreamEndCapture(st, &g_); if (g_) hipGraphDestroy(g_); (void)hipGetLastError(); } }
    } capture_guard{st, &capturing};
    int nkev = 0;
    if (use_graph && !ddm) {
      gkey.assign((const char*)&w, sizeof(WS)); gkey.append((const char*)&dm, sizeof(Dims)); gkey.append((const char*)&o, sizeof(Opts));
      const int ks[6] = {active, nfac, nb, reg_max, (int)h->flags, h->tune_pretest | (h->tune_fuse << 1) | (h->tune_small << 2)};
      gkey.append((const char*)ks, sizeof(ks));
      if (!ln->graphs) ln->graphs = new std::unordered_map<std::string, hipGraphExec_t>();
      // an argument set seen for the first time runs as plain launches and is only remembered; the second time it is captured; from then on replayed --
      // the tail of a ragged batch (a new pair of list lengths every iteration) never pays for a capture it would use once
      auto f = ln->graphs->find(gkey);
      if (f == ln->graphs->end()) {
        if (ln->graphs->size() >= 256) { for (auto& kv : *ln->graphs) if (kv.second) hipGraphExecDestroy(kv.second); ln->graphs->clear(); }
        (*ln->graphs)[gkey] = nullptr;
      } else if (f->second) gexec = f->second;
      else if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) capturing = true;
    }
    if (!gexec) {
    if (prof) HIPCHK(hipEventRecord(ln->ev[0], st));
    if (big) hipLaunchKernelGGL(kb_stage_pre, dim3(BP), dim3(256), 0, st, w, dm);
    else TMPC_STAGE_LAUNCH(k_stage_pre, slots_bytes(PRE_SLOTS), st, w, dm);
    if (eq) {        // <n above 32, more than 32 rows per stage>
      const bool bigr = dm.nr > NRS;
      if (big && bigr) hipLaunchKernelGGL((k_phi_pre<true, true>), dim3(BP), dim3(64), (size_t)phi_pre_lds(true, true) * sizeof(double), st, w, dm, 1);
      else if (big) hipLaunchKernelGGL((k_phi_pre<true, false>), dim3(BP), dim3(64), (size_t)phi_pre_lds(true, false) * sizeof(double), st, w, dm, 1);
      else if (bigr) hipLaunchKernelGGL((k_phi_pre<false, true>), dim3(BP), dim3(64), (size_t)phi_pre_lds(false, true) * sizeof(double), st, w, dm, 1);
      else hipLaunchKernelGGL((k_phi_pre<false, false>), dim3(BP), dim3(64), (size_t)phi_pre_lds(false, false) * sizeof(double), st, w, dm, 1);
    }
    if (t3) hipLaunchKernelGGL(k_t3_pre, dim3(BP), dim3(64), t3_lds, st, w, dm);
    hipLaunchKernelGGL(k_ctrl_a, dim3(active), dim3(64), 0, st, w, dm, o);
    if (prof) HIPCHK(hipEventRecord(ln->ev[1], st));
    // assembly and factorisation only for the problems that need a new one (flist; the others take a chord step)
    if (ddm) {      // (round 5: like the fp64 loop only for the problems of flist -- the others take a chord step on the double-double factor they have)
      if (nfac > 0) {
        hipLaunchKernelGGL(k_dd_images, dim3(nfac * dm.p), dim3(256), dd_stage_lds(dm, DD_IMG_SLOTS), st, wf, dm);
        dd_schur_launch(wf, dm, nfac * dm.p, st);
      }
    } else if (nfac > 0) {
      if (schur_in_lds(dm)) {
        hipLaunchKernelGGL(k_schur<0>, dim3(nfac * dm.p), dim3(SCH_NT), schur_lds<0>(dm), st, wf, dm);
        hipLaunchKernelGGL(k_schur<1>, dim3(nfac * dm.p), dim3(SCH_NT), schur_lds<1>(dm), st, wf, dm);
      } else {
        hipLaunchKernelGGL((k_schur<0, true>), dim3(nfac * dm.p), dim3(SCH_NT), schur_lds<0>(dm), st, wf, dm);
        hipLaunchKernelGGL((k_schur<1, true>), dim3(nfac * dm.p), dim3(SCH_NT), schur_lds<1>(dm), st, wf, dm);
      }
    }
    if (eq && nfac > 0 && ddm) hipLaunchKernelGGL(k_dd_aug_fill, dim3(nfac * dm.p), dim3(256), (size_t)dd_aug_lds_doubles(dm.nr, dm.n, dm.nx) * sizeof(double), st, wf, dm, 0);
    else if (eq && nfac > 0) hipLaunchKernelGGL(k_aug_fill, dim3(nfac * dm.p), dim3(64), 0, st, wf, dm);
    if (t3 && nfac > 0 && big) hipLaunchKernelGGL(k_t3_schur<true>, dim3(nfac * dm.p), dim3(256), t3_schur_lds, st, wf, dm);
    else if (t3 && nfac > 0) hipLaunchKernelGGL(k_t3_schur<false>, dim3(nfac * dm.p), dim3(256), t3_schur_lds, st, wf, dm);
    if (t3 && eq && nfac > 0) hipLaunchKernelGGL(k_t3_cross, dim3(nfac * dm.p), dim3(64), 0, st, wf, dm);
    if (h->flags & TMPC_DEBUG_FLAG_STOP_ASSEMBLED) {   // debug (tunempc_hip_debug.h; tests/tools/step3_asm_check.py): stop with the assembled, unfactored system of the first iteration in the workspace
      if (big) hipLaunchKernelGGL(kb_stage_rhs, dim3(BP), dim3(256), 0, st, w, dm, 1);
      else TMPC_STAGE_LAUNCH(k_stage_rhs, slots_bytes(RHS_SLOTS), st, w, dm, 1);
      if (t3) hipLaunchKernelGGL(k_t3_rhs, dim3(BP), dim3(64), t3_lds, st, w, dm, 1);
      hipLaunchKernelGGL(k_gather, dim3(BP), dim3(64), 0, st, w, dm, 1);
      if (t3) hipLaunchKernelGGL(k_t3_gather, dim3(BP), dim3(64), 0, st, w, dm, 1);
      HIPCHK(hipStreamSynchronize(st));
      HIPCHK(hipGetLastError());
      return TMPC_OK;
    }
    // plain model: the right-hand sides of the predictor pass are ready before the factorisation, and its forward substitution
    // rides inside it (k_cr_update_dma reads the O blocks anyway): one read of every O block less per main-phase iteration
    // (round 3: also with the multipliers of G / C -- their right-hand side rows depend on the iterate only, like the others; not with Step 3)
    const bool fuse1 = !ddm && h->tune_fuse && !(dm.flags & DF_NO_DMA) && !t3 && !(dm.flags & 1) && nfac > 0 && dm.p > 1 && dm.nt <= TRR_NT && !cr_small_levels(dm, h->sched, nullptr);
    if (fuse1) {
      if (big) hipLaunchKernelGGL(kb_stage_rhs, dim3(BP), dim3(256), 0, st, w, dm, 1);
      else TMPC_STAGE_LAUNCH(k_stage_rhs, slots_bytes(RHS_SLOTS), st, w, dm, 1);
      if (eq && big) hipLaunchKernelGGL(k_phi_rhs<true>, dim3(BP), dim3(64), slots_bytes(2), st, w, dm, 1, 1);
      else if (eq) hipLaunchKernelGGL(k_phi_rhs<false>, dim3(BP), dim3(64), slots_bytes(2), st, w, dm, 1, 1);
      hipLaunchKernelGGL(k_gather, dim3(BP), dim3(64), 0, st, w, dm, 1);
      if (eq) hipLaunchKernelGGL(k_aug_gather, dim3(BP), dim3(64), 0, st, w, dm, 1);
    }
    if (prof) HIPCHK(hipEventRecord(ln->ev[2], st));
    if (ddm) { if (nfac > 0) dd_factor(wf, dm, h->sched, h->d_sched, flist, nfac, st); }
    else if (nfac > 0) cr_factor(w, dm, h->sched, h->d_sched, flist, nfac, st, h->rs, h->mt, prof ? &ln->kev : nullptr, &nkev, fuse1 ? 1 : 0, ddm ? 0 : nlowp);
    if (prof) HIPCHK(hipEventRecord(ln->ev[3], st));
    for (int pass = 1; pass <= 2; ++pass) {
      const bool fused = (pass == 1 && fuse1);
      if (!fused && big) hipLaunchKernelGGL(kb_stage_rhs, dim3(BP), dim3(256), 0, st, w, dm, pass);
      else if (!fused) TMPC_STAGE_LAUNCH(k_stage_rhs, slots_bytes(RHS_SLOTS), st, w, dm, pass);
      if (eq && !fused && big) hipLaunchKernelGGL(k_phi_rhs<true>, dim3(BP), dim3(64), slots_bytes(2), st, w, dm, pass, 1);
      else if (eq && !fused) hipLaunchKernelGGL(k_phi_rhs<false>, dim3(BP), dim3(64), slots_bytes(2), st, w, dm, pass, 1);
      if (t3) hipLaunchKernelGGL(k_t3_rhs, dim3(BP), dim3(64), t3_lds, st, w, dm, pass);
      if (!fused) hipLaunchKernelGGL(k_gather, dim3(BP), dim3(64), 0, st, w, dm, pass);
      if (eq && !fused) hipLaunchKernelGGL(k_aug_gather, dim3(BP), dim3(64), 0, st, w, dm, pass);
      if (t3) hipLaunchKernelGGL(k_t3_gather, dim3(BP), dim3(64), 0, st, w, dm, pass);
      if (ddm) { const int rc_ = dd_solve(w, dm, h->sched, h->d_sched, alist, active, st, pass, nb); if (rc_ != TMPC_OK) return rc_; }
      else cr_solve(w, dm, h->sched, h->d_sched, alist, active, st, pass, fused);
      hipLaunchKernelGGL(k_solve_border, dim3(active), dim3(256), 0, st, w, dm, (const int*)alist, pass);
      if (eq && big) {
        hipLaunchKernelGGL(kb_phi_dm, dim3(BP), dim3(256), 0, st, w, dm, pass);
        hipLaunchKernelGGL(k_phi_dir<true>, dim3(BP), dim3(64), (size_t)PHI_DIR_LDS * sizeof(double), st, w, dm, pass, 1);
      } else if (eq) hipLaunchKernelGGL(k_phi_dir<false>, dim3(BP), dim3(64), (size_t)PHI_DIR_LDS * sizeof(double), st, w, dm, pass, 1);
      if (t3) hipLaunchKernelGGL(k_t3_dir, dim3(BP), dim3(64), t3_lds, st, w, dm, pass);
      if (big) {
        hipLaunchKernelGGL(kb_stage_dir, dim3(BP), dim3(256), 0, st, w, dm, pass);
        hipLaunchKernelGGL(kb_eigmin, dim3(BP * 4), dim3(256), big_lds, st, w, dm, pass);
      } else {
        TMPC_STAGE_LAUNCH(k_stage_dir, slots_bytes(DIR_SLOTS), st, w, dm, pass);
        if (dm.n <= 8) hipLaunchKernelGGL(k_eigmin_lane, dim3((BP * 4 + 63) / 64), dim3(64), 0, st, w, dm, pass, BP * 4);      // one thread per matrix (tmpc_stage.h)
        else hipLaunchKernelGGL(k_eigmin, dim3(BP * 4), dim3(64), (size_t)(MS + 160) * sizeof(double), st, w, dm, pass, h->tune_pretest ? o.chord_step : -1.0);
      }
      if (eq) hipLaunchKernelGGL(k_phi_steps, dim3((BPall + 63) / 64), dim3(64), 0, st, w, dm, pass);
      if (t3) hipLaunchKernelGGL(k_t3_steps, dim3(BP), dim3(64), t3_lds, st, w, dm, pass);
      if (pass == 1) {
        hipLaunchKernelGGL(k_ctrl_b, dim3(active), dim3(64), 0, st, w, dm, reg_max);
        if (prof) HIPCHK(hipEventRecord(ln->ev[4], st));
      } else {
        hipLaunchKernelGGL(k_ctrl_c, dim3(active), dim3(64), 0, st, w, dm, reg_max);
      }
    }
    TMPC_STAGE_LAUNCH(k_update, 0, st, w, dm);
    if (eq) hipLaunchKernelGGL(k_phi_update, dim3((BPall + 63) / 64), dim3(64), 0, st, w, dm);
    if (t3) hipLaunchKernelGGL(k_t3_update, dim3(BP), dim3(64), 0, st, w, dm);
    HIPCHK(hipMemsetAsync(w.active, 0, 2 * sizeof(int), st));
    if (dm.flags & DF_LOWP) HIPCHK(hipMemsetAsync(w.active + 4, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_ctrl_d, dim3((nb + 63) / 64), dim3(64), 0, st, w, dm, o);
    if (prof) HIPCHK(hipEventRecord(ln->ev[5], st));
    }      // (!gexec)
    if (capturing) {
      hipGraph_t graph = nullptr;
      capturing = false;
      const bool ok_ = hipStreamEndCapture(st, &graph) == hipSuccess && graph && hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0) == hipSuccess;
      if (graph) hipGraphDestroy(graph);
      if (!ok_) {        // nothing of this iteration ran (it was only recorded): plain launches from here on, same iteration again
        (void)hipGetLastError();
        use_graph = false; gexec = nullptr;
        continue;
      }
      (*ln->graphs)[gkey] = gexec;
    }
    if (gexec) HIPCHK(hipGraphLaunch(gexec, st));
    const int nfac_done = nfac;
    HIPCHK(hipMemcpyAsync(cnt, w.active, ((dm.flags & DF_LOWP) ? 5 : 2) * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    active = cnt[0]; nfac = cnt[1]; nlowp = (dm.flags & DF_LOWP) ? cnt[4] : 0;
    if (prof) {
      float ms;
      for (int i = 0; i < 5; ++i) { HIPCHK(hipEventElapsedTime(&ms, ln->ev[i], ln->ev[i + 1])); ln->prof[i] += ms; }
      if (nfac_done > 0) ln->prof[5] += 1.0;
      ln->prof[8] += (double)nfac_done;
      for (int i = 0; i + 1 < nkev; i += 2) { HIPCHK(hipEventElapsedTime(&ms, ln->kev[i], ln->kev[i + 1])); const int c_ = (nkev > 2) ? (i / 2) % 4 : 0; ln->prof[c_ < 3 ? 9 + c_ : 14] += ms; }
    }
    ++it;
  }
  return TMPC_OK;
  };
  { const int rc_ = ipm_loop(o, false, cap); if (rc_ != TMPC_OK) return rc_; }
  // ---- tight mode (tmpc_set_tight; plain model): restart the problems that ended Optimal towards tight_tol * kappa with the block linear
  // algebra in double-double, then the dd dual-Newton polish (tmpc_dd.h)
  // (round 5: also the models with rows -- Step 1 with G, Step 2 with C and the norm terms: the multipliers and epigraph variables ride in the augmented blocks of
  // tmpc_phi.h, their rows formed in double-double (k_dd_aug_fill), and join the polish as variables: k_dd_polish_pre, k_polish_phi, k_polish_arrows)
  if (h->tight && !t3 && wall.Dl) {
    // (round 5: chord steps also in this phase and in the polish -- a double-double factorisation costs ten fp64 ones, a step on an old one a tenth of it)
    Opts ot = o; ot.tight = 1; ot.tight_tol = h->tight_tol; ot.fast_exit = 0; ot.max_iter = 2 * o.max_iter;
    // where a factorisation costs ten fp64 ones and a step on the old one a twentieth of it, chord steps pay from a much shorter safe step on: threshold 3 instead of 10
    // (a chord step that contracts by less than 1/4 still returns to Newton).  Measured at the bench shape (profiles/r5_tight_chord.txt): 10 -> 1476, 5 -> 1507,
    // 3 -> 1547, 2 -> 1579, 1.5 -> 1585 stage-conv/s; at 2 the certified gap of one test shape leaves its bound (chord steps right before the hand-over to the polish)
    if (ot.chord_step > TIGHT_CHORD_STEP) ot.chord_step = TIGHT_CHORD_STEP;
    HIPCHK(hipMemsetAsync(w.active, 0, 4 * sizeof(int), st));
    { WS wi = wall; wi.alist = alist; wi.flist = flist; hipLaunchKernelGGL(k_tight_restart, dim3(nb), dim3(64), 0, st, wi, dm, ot); }
    HIPCHK(hipMemcpyAsync(cnt, w.active, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    active = cnt[0]; nfac = cnt[0];
    { const int rc_ = ipm_loop(ot, true, it + cap); if (rc_ != TMPC_OK) return rc_; }
    HIPCHK(hipMemcpyAsync(cnt, w.active, 3 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const int npol = cnt[2];
    // three index buffers: the problems of this step, those of the next one (written by k_polish_ctrl_b), and the ones of this step that get a new
    // factorisation (the first step: all; later only the problems whose chord step stopped contracting)
    // (plist itself is read again by the final sweep: until round 5 it served as one of the two ping-pong buffers, and a member that needed a third polish step was
    // written over its head -- the member there missed the final sweep and exported mu S(y)^-1 instead of the dual iterate of its last Newton step: dual residuals
    // 1e-3 instead of 1e-15, found with the certificate of the models with multipliers)
    int* list = alist; int* next = wall.pnext; int* fac = flist; int count = npol, nfp = npol;
    if (npol > 0) {
      HIPCHK(hipMemcpyAsync(list, wall.plist, (size_t)npol * sizeof(int), hipMemcpyDeviceToDevice, st));
      HIPCHK(hipMemcpyAsync(fac, wall.plist, (size_t)npol * sizeof(int), hipMemcpyDeviceToDevice, st));
    }
    for (int step = 0; step < POLISH_MAX && count > 0; ++step) {
      WS wp = wall; wp.alist = list;
      hipLaunchKernelGGL(k_dd_polish_pre, dim3(count * dm.p), dim3(256), dd_stage_lds(dm, DD_POL_SLOTS), st, wp, dm, 0);
      hipLaunchKernelGGL(k_polish_ctrl_a, dim3(count), dim3(64), 0, st, wp, dm);
      if (nfp > 0) {
        WS wq = wall; wq.alist = fac;
        dd_schur_launch(wq, dm, nfp * dm.p, st);
        if (eq) {        // rows of the multipliers: vectors, T_loc,loc and border entries from the fp64 roundings of X_r = mu S_r^-1, S_r^-1 (the Hessian only sets the rate)
          const bool bigr = dm.nr > NRS, bign = dm.n > NMAX;
          if (bign && bigr) hipLaunchKernelGGL((k_phi_pre<true, true>), dim3(nfp * dm.p), dim3(64), (size_t)phi_pre_lds(true, true) * sizeof(double), st, wq, dm, 1);
          else if (bign) hipLaunchKernelGGL((k_phi_pre<true, false>), dim3(nfp * dm.p), dim3(64), (size_t)phi_pre_lds(true, false) * sizeof(double), st, wq, dm, 1);
          else if (bigr) hipLaunchKernelGGL((k_phi_pre<false, true>), dim3(nfp * dm.p), dim3(64), (size_t)phi_pre_lds(false, true) * sizeof(double), st, wq, dm, 1);
          else hipLaunchKernelGGL((k_phi_pre<false, false>), dim3(nfp * dm.p), dim3(64), (size_t)phi_pre_lds(false, false) * sizeof(double), st, wq, dm, 1);
          hipLaunchKernelGGL(k_dd_aug_fill, dim3(nfp * dm.p), dim3(256), (size_t)dd_aug_lds_doubles(dm.nr, dm.n, dm.nx) * sizeof(double), st, wq, dm, 1);
        }
        dd_factor(wq, dm, h->sched, h->d_sched, fac, nfp, st);
      }
      hipLaunchKernelGGL(k_dd_gather, dim3(count * dm.p), dim3(64), 0, st, wp, dm);
      if (eq) hipLaunchKernelGGL(k_aug_gather, dim3(count * dm.p), dim3(64), 0, st, wp, dm, 2);
      { const int rc_ = dd_solve(wp, dm, h->sched, h->d_sched, list, count, st, 2, nb); if (rc_ != TMPC_OK) return rc_; }
      hipLaunchKernelGGL(k_dd_solve_border, dim3(count), dim3(256), 0, st, wp, dm, (const int*)list);
      if (eq) hipLaunchKernelGGL(k_polish_phi, dim3(count * dm.p), dim3(64), 0, st, wp, dm);
      if (eq && dm.constr) hipLaunchKernelGGL(k_polish_arrows, dim3(count * dm.p), dim3(256), 0, st, wp, dm);
      if (dm.n > NMAX) hipLaunchKernelGGL(kb_polish_step, dim3(count * dm.p), dim3(256), 0, st, wp, dm);
      else hipLaunchKernelGGL(k_polish_step, dim3(count * dm.p), dim3(256), slots_bytes(5), st, wp, dm);
      HIPCHK(hipMemsetAsync(w.active + 3, 0, sizeof(int), st));
      HIPCHK(hipMemsetAsync(w.active + 1, 0, sizeof(int), st));
      hipLaunchKernelGGL(k_polish_ctrl_b, dim3(count), dim3(64), 0, st, wp, dm, ot, (const int*)list, count, next, w.active + 3, fac, w.active + 1);
      HIPCHK(hipMemcpyAsync(cnt, w.active, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      count = cnt[3]; nfp = cnt[1];
      { int* t_ = list; list = next; next = t_; }
      ++it;
    }
    if (npol > 0) {      // final iterate of every polished problem: cone check, X_r / S_r for the outputs and the dual export
      WS wp = wall; wp.alist = wall.plist;
      hipLaunchKernelGGL(k_dd_polish_pre, dim3(npol * dm.p), dim3(256), dd_stage_lds(dm, DD_POL_SLOTS), st, wp, dm, 1);
      hipLaunchKernelGGL(k_polish_final, dim3(npol), dim3(64), 0, st, wp, dm);
    }
    hipLaunchKernelGGL(k_tight_fallback, dim3(nb), dim3(64), 0, st, wall, dm);      // members whose tight phase failed: back to the result of the default solve
  }
  ln->prof[7] += it;
  if (prof && w.O32) {      // problem-factorisations whose updates ran in single precision (profile slot 13)
    std::vector<int> hip_((size_t)nb * IS);
    HIPCHK(hipMemcpyAsync(hip_.data(), wall.iprob, hip_.size() * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int b = 0; b < nb; ++b) ln->prof[13] += (double)hip_[(size_t)b * IS + I_NLOWP];
  }
  if (big) hipLaunchKernelGGL(kb_final_stage, dim3(BPall), dim3(256), big_lds, st, wall, dm);
  else hipLaunchKernelGGL(k_final_stage, dim3(BPall), dim3(64), slots_bytes(FIN_SLOTS), st, wall, dm);
  hipLaunchKernelGGL(k_final_prob, dim3(nb), dim3(64), 0, st, wall, dm);
  HIPCHK(hipGetLastError());
  return TMPC_OK;
}

// every entry point runs on the device the handle was created on (kernel attributes and the workspace belong to it)
#define ON_DEVICE(h)                                                                              \
  do {                                                                                            \
    int cur_ = -1;                                                                                \
    if (hipGetDevice(&cur_) != hipSuccess || cur_ != (h)->device) HIPCHK(hipSetDevice((h)->device)); \
  } while (0)

// ---------------------------------------------------------------------------------- C ABI
extern "C" {

const char* tmpc_last_error(void) { return g_err; }
const char* tmpc_version(void) { return "tunempc_amd 0.2 (gfx950, fp64 MFMA, cyclic-reduction block factorisation)"; }

int tmpc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static bool rows_ok(int nx, int ng, int nc) {
  (void)nx;
  return ng >= 0 && ng <= NGM && nc >= 0 && nc <= NCM;
}
// stage blocks beyond NBM = 64: the plain model only (the multiplier and Step 3 kernels keep one lane per entry of an n-vector)
static bool model_dims_ok(int nx, int mb, int ng, int nc, int step3) { return nx + mb <= NBM || (ng == 0 && nc == 0 && !step3); }
static_assert(AEL == TMPC_ARROW_LD, "tunempc_hip.h: leading dimension of the exported arrow blocks");
static_assert(NGM == TMPC_MAX_ROWS && NCM == TMPC_MAX_ROWS, "tunempc_hip.h: row limits");
static uint64_t workspace_bytes(int chunk, int p, int nx, int mb, int ng, int nc, int step3) {
  if (chunk < 1 || !dims_ok(p, nx, mb) || !rows_ok(nx, ng, nc) || !model_dims_ok(nx, mb, ng, nc, step3)) return 0;
  WS w;
  Dims dm = make_dims(chunk, p, nx, mb, ng, nc, step3);
  if (solve_lds(dm) > 160 * 1024) return 0;      // blocks (d + multipliers + entries of T_k) beyond the LDS image of the substitution kernels
  return (uint64_t)carve(w, dm, nullptr, nullptr);
}
uint64_t tmpc_workspace_bytes_con(int chunk, int p, int nx, int mb, int ng, int nc) { return workspace_bytes(chunk, p, nx, mb, ng, nc, 0); }
uint64_t tmpc_workspace_bytes_step3(int chunk, int p, int nx, int mb) { return workspace_bytes(chunk, p, nx, mb, 0, 0, 1); }
uint64_t tmpc_workspace_bytes_step3_con(int chunk, int p, int nx, int mb, int ng, int nc) { return workspace_bytes(chunk, p, nx, mb, ng, nc, 1); }
uint64_t tmpc_workspace_bytes_eq(int chunk, int p, int nx, int mb, int ng) { return tmpc_workspace_bytes_con(chunk, p, nx, mb, ng, 0); }
uint64_t tmpc_workspace_bytes(int chunk, int p, int nx, int mb) { return tmpc_workspace_bytes_eq(chunk, p, nx, mb, 0); }

int tmpc_create(tmpc_handle** out, int chunk, int p, int nx, int mb) { return tmpc_create_eq(out, chunk, p, nx, mb, 0); }

int tmpc_create_eq(tmpc_handle** out, int chunk, int p, int nx, int mb, int ng) { return tmpc_create_con(out, chunk, p, nx, mb, ng, 0); }

static int create_handle(tmpc_handle** out, int chunk, int p, int nx, int mb, int ng, int nc, int step3, int lanes = 0);
int tmpc_create_con(tmpc_handle** out, int chunk, int p, int nx, int mb, int ng, int nc) { return create_handle(out, chunk, p, nx, mb, ng, nc, 0); }
int tmpc_create_step3(tmpc_handle** out, int chunk, int p, int nx, int mb) { return create_handle(out, chunk, p, nx, mb, 0, 0, 1); }
int tmpc_create_step3_con(tmpc_handle** out, int chunk, int p, int nx, int mb, int ng, int nc) { return create_handle(out, chunk, p, nx, mb, ng, nc, 1); }
int tmpc_create_ex(tmpc_handle** out, int chunk, int p, int nx, int mb, int ng, int nc, int step3, int lanes) {
  if (lanes < 0 || lanes > MAXL) { snprintf(g_err, sizeof(g_err), "tmpc_create_ex: lanes = %d outside 0..%d", lanes, MAXL); return TMPC_E_ARG; }
  return create_handle(out, chunk, p, nx, mb, ng, nc, step3 ? 1 : 0, lanes);
}

static int create_handle(tmpc_handle** out, int chunk, int p, int nx, int mb, int ng, int nc, int step3, int lanes) {
  if (!out) return TMPC_E_ARG;
  *out = nullptr;
  if (!dims_ok(p, nx, mb)) { snprintf(g_err, sizeof(g_err), "unsupported dims p=%d nx=%d mb=%d (need nx+mb<=%d)", p, nx, mb, NB); return TMPC_E_UNSUPPORTED; }
  if (!rows_ok(nx, ng, nc)) {
    snprintf(g_err, sizeof(g_err), "unsupported constraint rows ng=%d nc=%d (need 0<=ng<=%d, 0<=nc<=%d)", ng, nc, NGM, NCM);
    return TMPC_E_UNSUPPORTED;
  }
  if (!model_dims_ok(nx, mb, ng, nc, step3)) {
    snprintf(g_err, sizeof(g_err), "nx+mb = %d > %d is supported for the plain model only (no G / C rows, no Step 3)", nx + mb, NBM);
    return TMPC_E_UNSUPPORTED;
  }
  {                          // blocks that the LDS images of k_schur and of the substitutions can hold
    const Dims db = make_dims(1, p, nx, mb, ng, nc, step3);
    if (solve_lds(db) > 160 * 1024) {
      snprintf(g_err, sizeof(g_err), "nx=%d ng=%d nc=%d step3=%d: Schur blocks of %d (svec(P) + multipliers + entries of T_k) do not fit the LDS image of the substitution kernels (limit 3168)", nx, ng, nc, step3, db.dp);
      return TMPC_E_UNSUPPORTED;
    }
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { snprintf(g_err, sizeof(g_err), "no HIP device"); return TMPC_E_NODEVICE; }
  if (chunk <= 0) {
    size_t fr = 0, tot = 0;
    HIPCHK(hipMemGetInfo(&fr, &tot));
    const uint64_t per = workspace_bytes(1, p, nx, mb, ng, nc, step3);
    uint64_t fit = (uint64_t)(0.6 * (double)fr) / std::max<uint64_t>(per, 1);
    chunk = (int)std::max<uint64_t>(1, std::min<uint64_t>(512, fit));
  }
  tmpc_handle* h = new (std::nothrow) tmpc_handle();
  if (!h) return TMPC_E_NOMEM;
  h->slab = nullptr; h->d_sched = nullptr; h->ev_in = nullptr; h->dd_slab = nullptr; h->dd_bytes = 0; h->tight = 0; h->tight_tol = 0x1p-37;
  if (hipGetDevice(&h->device) != hipSuccess) { delete h; return TMPC_E_HIP; }
  // lanes: one by default -- measured on MI355X (profiles/r2b_lanes.txt), two or four concurrent half-waves gain nothing at the
  // bench shape (6445 / 6526 / 6399 stage-conv/s with 1 / 2 / 4 lanes): every kernel of the loop already fills the chip, so the
  // streams time-share instead of overlapping.  TMPC_LANES=n turns them on (latency-bound small batches may still profit).
  // Round 3: problems whose Schur blocks are a single tile (dp <= 64: every example of the reference) are latency-bound per launch, and two
  // lanes overlap the launch ramps of one half-wave with the kernels of the other: +8 ... +10 % at the AWE and evaporation shapes with
  // batches of 64 / 256 (repeat-timed scripts/config_sweep.py), nothing to gain at batch 1.
  int nl = (make_dims(1, p, nx, mb, ng, nc, step3).dp <= 64 && chunk >= 2) ? 2 : 1;
  if (lanes >= 1) nl = std::min(lanes, MAXL);          // tmpc_create_ex
  nl = std::max(1, std::min(nl, chunk));
  h->nlanes = nl;
  const int cap = (chunk + nl - 1) / nl;
  h->chunk = cap * nl;
  h->dm = make_dims(cap, p, nx, mb, ng, nc, step3);
  h->sched = cr_build(p);
  h->rs = 0; h->mt = 0;
  h->tune_small = 1; h->tune_pretest = 1; h->tune_fuse = 1; h->tune_graph = 1; h->tune_persist = 1;
  h->opt.tol = 0x1p-25; h->opt.center_tol = 1e-9; h->opt.max_iter = 50; h->opt.center_iter = 12;
  h->opt.fast_exit = 0;
  h->opt.lowp_switch = TMPC_LOWP_SWITCH_DEFAULT;
  h->opt.chord_step = 10.0;       // centering: re-use the factorisation once the iterate moves by < 1/10 in the local norm (profiles/r2z_chord_default.txt: +3.7 %, same answers to 1e-10); tmpc_set_tuning(TMPC_TUNE_CHORD_STEP, 0) disables
  h->flags = 0;
  WS tmp;
  const size_t lane_bytes = carve(tmp, h->dm, nullptr, nullptr);
  h->slab_bytes = lane_bytes * nl;
  if (hipMalloc(&h->slab, h->slab_bytes) != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "hipMalloc(%zu bytes) failed", h->slab_bytes);
    delete h;
    return TMPC_E_NOMEM;
  }
  int rc = cr_upload(h->sched, &h->d_sched);
  bool ok = (rc == TMPC_OK) && hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming) == hipSuccess;
  for (int l = 0; l < MAXL; ++l) { Lane& ln = h->lane[l]; ln.big_scr = nullptr; ln.o32_dp = 0; ln.graphs = nullptr; ln.st = nullptr; for (int i = 0; i < 8; ++i) ln.ev[i] = nullptr; memset(ln.prof, 0, sizeof(ln.prof)); ln.last_nb = 0; ln.err[0] = 0; }
  for (int l = 0; l < nl && ok; ++l) {
    Lane& ln = h->lane[l];
    carve(ln.ws, h->dm, (char*)h->slab + (size_t)l * lane_bytes, &ln);
    if (ln.ws.O32) ok = hipMemset(ln.ws.O32, 0, 2 * (size_t)h->dm.B * h->dm.p * h->dm.dp * ((h->dm.dp + 31) & ~31) * sizeof(float)) == hipSuccess;      // (the padding columns stay zero: k_cr_trsm_dma writes the dp real ones)
    ok = ok && hipStreamCreateWithFlags(&ln.st, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 8 && ok; ++i) ok = hipEventCreate(&ln.ev[i]) == hipSuccess;
  }
  if (ok) { rc = set_lds_attrs(h->device); ok = (rc == TMPC_OK); }
  if (!ok) { tmpc_destroy(h); return rc != TMPC_OK ? rc : TMPC_E_HIP; }
  *out = h;
  return TMPC_OK;
}

int tmpc_destroy(tmpc_handle* h) {
  if (!h) return TMPC_E_ARG;
#ifdef TMPC_EIG_DEBUG
  { int c[4]; double v[8]; hipDeviceSynchronize(); hipMemcpyFromSymbol(c, HIP_SYMBOL(g_eig_dbg), sizeof c); hipMemcpyFromSymbol(v, HIP_SYMBOL(g_eig_dbgv), sizeof v);
    fprintf(stderr, "eig pretest: passed %d (wrong %d: lambda %.6e theta %.6f pass %g phase %g)  failed %d (wrong %d: lambda %.6e theta %.6f pass %g phase %g)\n",
            c[0], c[1], v[0], v[1], v[2], v[3], c[2], c[3], v[4], v[5], v[6], v[7]); }
#endif
  for (int l = 0; l < MAXL; ++l) {
    Lane& ln = h->lane[l];
    for (int i = 0; i < 8; ++i) if (ln.ev[i]) hipEventDestroy(ln.ev[i]);
    for (hipEvent_t e : ln.kev) hipEventDestroy(e);
    if (ln.st) hipStreamDestroy(ln.st);
    if (ln.big_scr) hipFree(ln.big_scr);
    if (ln.graphs) { for (auto& kv : *ln.graphs) if (kv.second) hipGraphExecDestroy(kv.second); delete ln.graphs; }
    LaneWorker::retire(ln.worker);
  }
  if (h->ev_in) hipEventDestroy(h->ev_in);
  if (h->slab) hipFree(h->slab);
  if (h->dd_slab) hipFree(h->dd_slab);
  if (h->d_sched) hipFree(h->d_sched);
  delete h;
  return TMPC_OK;
}

int tmpc_get_chunk(tmpc_handle* h) { return h ? h->chunk : TMPC_E_ARG; }

int tmpc_set_options(tmpc_handle* h, double tol, double center_tol, int max_iter, int center_iter, int flags) {
  if (!h) return TMPC_E_ARG;
  if (flags & ~(TMPC_FLAG_NO_MFMA | TMPC_FLAG_PROFILE | TMPC_FLAG_FAST_EXIT | TMPC_DEBUG_FLAG_STOP_ASSEMBLED | TMPC_DEBUG_FLAG_NO_LIFT | TMPC_DEBUG_FLAG_NO_DMA | TMPC_DEBUG_FLAG_GENERIC_STAGE)) {
    snprintf(g_err, sizeof(g_err), "tmpc_set_options: unknown flag bits 0x%x (TMPC_FLAG_NO_MFMA = 1, TMPC_FLAG_PROFILE = 2, TMPC_FLAG_FAST_EXIT = 4)", flags);
    return TMPC_E_ARG;
  }
  if (tol > 0) h->opt.tol = tol;
  if (center_tol > 0) h->opt.center_tol = center_tol;
  if (max_iter > 0) h->opt.max_iter = max_iter;
  if (center_iter > 0) h->opt.center_iter = center_iter;
  h->flags = flags;
  return TMPC_OK;
}


// Performance knobs that used to be environment variables (round 4: part of the ABI, per handle)
int tmpc_set_tuning(tmpc_handle* h, int key, double value) {
  if (!h) return TMPC_E_ARG;
  switch (key) {
    case TMPC_TUNE_CHORD_STEP: if (!(value >= 0.0)) return TMPC_E_ARG; h->opt.chord_step = value; return TMPC_OK;
    case TMPC_TUNE_LOWP_SWITCH: if (!(value >= 0.0)) return TMPC_E_ARG; h->opt.lowp_switch = value; return TMPC_OK;
    case TMPC_TUNE_SMALL_BLOCKS: h->tune_small = value != 0.0; return TMPC_OK;
    case TMPC_TUNE_EIG_PRETEST: h->tune_pretest = value != 0.0; return TMPC_OK;
    case TMPC_TUNE_FUSE_FWD: h->tune_fuse = value != 0.0; return TMPC_OK;
    case TMPC_TUNE_GRAPH: h->tune_graph = value != 0.0; return TMPC_OK;
    case TMPC_TUNE_PERSISTENT: if (!(value >= 0.0 && value <= 2.0)) return TMPC_E_ARG; h->tune_persist = (int)value; return TMPC_OK;
    default: snprintf(g_err, sizeof(g_err), "tmpc_set_tuning: unknown key %d", key); return TMPC_E_ARG;
  }
}

// Tight-accuracy mode (see tunempc_hip.h and tmpc_dd.h).  The double-double workspace (about as large as the block storage of the handle) is
// allocated at the first enable and kept until tmpc_destroy.
int tmpc_set_tight(tmpc_handle* h, int enable, double tight_tol) {
  if (!h) return TMPC_E_ARG;
  if (!enable) { h->tight = 0; return TMPC_OK; }
  if (h->dm.nT > 0) { snprintf(g_err, sizeof(g_err), "tmpc_set_tight: Steps 1 and 2 only (no Step 3 handles)"); return TMPC_E_UNSUPPORTED; }
  if (h->dm.nr > 0 && (size_t)dd_aug_lds_doubles(h->dm.nr, h->dm.n, h->dm.nx) * sizeof(double) > 160 * 1024) {
    snprintf(g_err, sizeof(g_err), "tmpc_set_tight: %d rows per stage at nx = %d, nx + mb = %d: their double-double vectors do not fit the LDS (rows * (2 n + 2 nx) <= 4040)", h->dm.nr, h->dm.nx, h->dm.n);
    return TMPC_E_UNSUPPORTED;
  }
  if ((size_t)dd_solve_lds_doubles(h->dm.dp) * sizeof(double) > 160 * 1024) {
    snprintf(g_err, sizeof(g_err), "tmpc_set_tight: Schur blocks of %d do not fit the LDS image of the double-double substitution kernels (nx <= 51)", h->dm.dp);
    return TMPC_E_UNSUPPORTED;
  }
  if (tight_tol > 0.0 && !(tight_tol >= 0x1p-42 && tight_tol < 1.0)) { snprintf(g_err, sizeof(g_err), "tmpc_set_tight: tolerance %g outside [2^-42, 1)", tight_tol); return TMPC_E_ARG; }
  ON_DEVICE(h);
  if (!h->dd_slab) {
    WS tmp;
    const size_t lane_bytes = carve_dd(tmp, h->dm, nullptr);
    if (hipMalloc(&h->dd_slab, lane_bytes * h->nlanes) != hipSuccess) {
      h->dd_slab = nullptr;
      snprintf(g_err, sizeof(g_err), "tmpc_set_tight: hipMalloc(%zu bytes) failed", lane_bytes * h->nlanes);
      return TMPC_E_NOMEM;
    }
    h->dd_bytes = lane_bytes * h->nlanes;
    for (int l = 0; l < h->nlanes; ++l) carve_dd(h->lane[l].ws, h->dm, (char*)h->dd_slab + (size_t)l * lane_bytes);
  }
  if (tight_tol > 0.0) h->tight_tol = tight_tol;
  h->tight = 1;
  return TMPC_OK;
}

}  // extern "C"

// rows of the last wave live in the lanes that solved them: lane 0 first
template <typename F> static int for_last_wave(tmpc_handle* h, int nb, F&& f) {
  int done = 0;
  for (int l = 0; l < h->nlanes && done < nb; ++l) {
    const int m = std::min(h->lane[l].last_nb, nb - done);
    if (m > 0) { int rc = f(h->lane[l], done, m); if (rc != TMPC_OK) return rc; }
    done += m;
  }
  return done == nb ? TMPC_OK : TMPC_E_ARG;
}

extern "C" {

int tmpc_get_trace(tmpc_handle* h, int nb, double* out) {
  if (!h || !out || nb < 1 || nb > h->chunk) return TMPC_E_ARG;
  ON_DEVICE(h);
  const size_t row = (size_t)TRACE_LEN * TRACE_W;
  return for_last_wave(h, nb, [&](Lane& ln, int o, int m) {
    HIPCHK(hipMemcpy(out + (size_t)o * row, ln.ws.trace, (size_t)m * row * sizeof(double), hipMemcpyDeviceToHost));
    return TMPC_OK;
  });
}

// dual iterate of the LAST wave solved (see tunempc_hip.h)
__global__ void k_dual_scalars(WS w, int nb, double* out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const double* pr = w.prob + (size_t)b * PS;
  double* o = out + (size_t)b * 4;
  o[0] = pr[P_X0]; o[1] = pr[P_TAU]; o[2] = pr[P_ALPHA]; o[3] = pr[P_MUT];
}

int tmpc_get_dual_host(tmpc_handle* h, int nb, double* X1, double* X2, double* scal) {
  if (!h || nb < 1 || nb > h->chunk) return TMPC_E_ARG;
  ON_DEVICE(h);
  const size_t row = (size_t)h->dm.p * h->dm.n * h->dm.n;
  return for_last_wave(h, nb, [&](Lane& ln, int o, int m) {
    if (X1) HIPCHK(hipMemcpy(X1 + (size_t)o * row, ln.ws.X1, (size_t)m * row * sizeof(double), hipMemcpyDeviceToHost));
    if (X2) HIPCHK(hipMemcpy(X2 + (size_t)o * row, ln.ws.X2, (size_t)m * row * sizeof(double), hipMemcpyDeviceToHost));
    if (scal) {
      hipLaunchKernelGGL(k_dual_scalars, dim3((m + 63) / 64), dim3(64), 0, 0, ln.ws, m, ln.d_info);      // d_info: [cap][16] scratch, free between calls
      HIPCHK(hipMemcpy(scal + (size_t)o * 4, ln.d_info, (size_t)m * 4 * sizeof(double), hipMemcpyDeviceToHost));
    }
    return TMPC_OK;
  });
}

// dual side of the stage-local multipliers of the LAST wave solved (Step 1 with G, Step 2): see tunempc_hip.h
int tmpc_get_dual_con_host(tmpc_handle* h, int nb, double* phi, double* z, double* aX, double* at) {
  if (!h || nb < 1 || nb > h->chunk || h->dm.nr < 1 || !h->lane[0].ws.phi) return TMPC_E_ARG;
  ON_DEVICE(h);
  const size_t row = (size_t)h->dm.p * h->dm.nr, arow = (size_t)h->dm.p * 2 * AE, trow = (size_t)h->dm.p * 2;
  return for_last_wave(h, nb, [&](Lane& ln, int o, int m) {
    if (phi) HIPCHK(hipMemcpy(phi + o * row, ln.ws.phi, (size_t)m * row * sizeof(double), hipMemcpyDeviceToHost));
    if (z) HIPCHK(hipMemcpy(z + o * row, ln.ws.zph, (size_t)m * row * sizeof(double), hipMemcpyDeviceToHost));
    if (aX) { if (!ln.ws.aX) return (int)TMPC_E_ARG; HIPCHK(hipMemcpy(aX + o * arow, ln.ws.aX, (size_t)m * arow * sizeof(double), hipMemcpyDeviceToHost)); }
    if (at) { if (!ln.ws.at) return (int)TMPC_E_ARG; HIPCHK(hipMemcpy(at + o * trow, ln.ws.at, (size_t)m * trow * sizeof(double), hipMemcpyDeviceToHost)); }
    return (int)TMPC_OK;
  });
}

int tmpc_debug_get_multipliers(tmpc_handle* h, int nb, int nr, double* phi, double* z, double* dphi, double* dz) {
  if (!h || nb < 1 || nb > h->chunk || nr < 1 || nr > h->dm.nr || !h->lane[0].ws.phi) return TMPC_E_ARG;
  ON_DEVICE(h);
  const size_t row = (size_t)h->dm.p * nr;
  return for_last_wave(h, nb, [&](Lane& ln, int o, int m) {
    const size_t cnt = (size_t)m * row * sizeof(double);
    if (phi) HIPCHK(hipMemcpy(phi + o * row, ln.ws.phi, cnt, hipMemcpyDeviceToHost));
    if (z) HIPCHK(hipMemcpy(z + o * row, ln.ws.zph, cnt, hipMemcpyDeviceToHost));
    if (dphi) HIPCHK(hipMemcpy(dphi + o * row, ln.ws.dphi, cnt, hipMemcpyDeviceToHost));
    if (dz) HIPCHK(hipMemcpy(dz + o * row, ln.ws.dzph, cnt, hipMemcpyDeviceToHost));
    return TMPC_OK;
  });
}

int tmpc_debug_get_array(tmpc_handle* h, int which, uint64_t offset, uint64_t count, double* out) {
  if (!h || !out) return TMPC_E_ARG;
  ON_DEVICE(h);
  const WS& ws = h->lane[0].ws;                 // lane 0 (run with TMPC_LANES=1 to see a whole wave)
  const double* src = nullptr;
  switch (which) {
    case 0: src = ws.psm; break;
    case 1: src = ws.pvec; break;
    case 2: src = ws.Ddiag; break;
    case 3: src = ws.D; break;
    case 4: src = ws.part; break;
    case 5: src = ws.O; break;
    case 6: src = ws.F; break;
    case 7: src = ws.W3; break;
    case 8: src = ws.U; break;
    default: return TMPC_E_ARG;
  }
  if (!src) return TMPC_E_ARG;
  HIPCHK(hipMemcpy(out, src + offset, count * sizeof(double), hipMemcpyDeviceToHost));
  return TMPC_OK;
}

int tmpc_get_profile(tmpc_handle* h, double* out16) {
  if (!h || !out16) return TMPC_E_ARG;
  for (int i = 0; i < 16; ++i) out16[i] = 0.0;
  for (int l = 0; l < h->nlanes; ++l)
    for (int i = 0; i < 16; ++i) {
      if (i == 7) out16[i] = std::max(out16[i], h->lane[l].prof[i]); else out16[i] += h->lane[l].prof[i];
      h->lane[l].prof[i] = 0.0;
    }
  out16[12] = (double)h->nlanes;
  return TMPC_OK;
}

// ---- one call = inputs + outputs of nbt problems; a wave of up to `chunk` problems is split over the lanes
struct Call {
  int nbt;
  const double *A, *B, *H, *J; const int32_t* ncnt; double rho;
  double *Hc, *dHc, *P, *FgF, *alpha, *beta, *kappa; int32_t *status, *iters; double* info;
  bool host;           // host pointers (staged through the lane's buffers) or device pointers
  int jr;              // row stride of J / FgF in this call (0: no constraint rows)
  bool step3 = false;  // Step 3 model (plain + T)
  double* T = nullptr; // [nbt][p][n][n] output of Step 3
};

static int lane_run(tmpc_handle* h, Lane* ln, const Call& c, int off, int nb) {
  if (hipSetDevice(h->device) != hipSuccess) { snprintf(g_err, sizeof(g_err), "hipSetDevice(%d) failed", h->device); return TMPC_E_HIP; }
  const Dims& dm = h->dm;
  const size_t nn = (size_t)dm.n * dm.n, nxx = (size_t)dm.nx * dm.nx, nxm = (size_t)dm.nx * dm.mb, gn = (size_t)c.jr * dm.n;
  const size_t so = (size_t)off * dm.p, BP = (size_t)nb * dm.p;
  hipStream_t st = ln->st;
  const double *dA = c.A + so * nxx, *dB = c.B ? c.B + so * nxm : nullptr, *dH = c.H + so * nn, *dJ = c.J ? c.J + so * gn : nullptr;
  const int32_t* dn = c.ncnt ? c.ncnt + so : nullptr;
  if (c.host) {
    HIPCHK(hipMemcpyAsync(ln->dA, dA, BP * nxx * sizeof(double), hipMemcpyHostToDevice, st)); dA = ln->dA;
    if (dm.mb > 0) { HIPCHK(hipMemcpyAsync(ln->dB, dB, BP * nxm * sizeof(double), hipMemcpyHostToDevice, st)); dB = ln->dB; }
    HIPCHK(hipMemcpyAsync(ln->dH, dH, BP * nn * sizeof(double), hipMemcpyHostToDevice, st)); dH = ln->dH;
    if (dJ) { HIPCHK(hipMemcpyAsync(ln->dG, dJ, BP * gn * sizeof(double), hipMemcpyHostToDevice, st)); dJ = ln->dG; }
    if (dn) { HIPCHK(hipMemcpyAsync(ln->dncnt, dn, BP * sizeof(int32_t), hipMemcpyHostToDevice, st)); dn = ln->dncnt; }
  }
  int rc = run_chunk(h, ln, nb, dA, dB, dH, st, dJ, dn, c.rho, c.step3);
  if (rc != TMPC_OK) return rc;
  ln->last_nb = nb;
  const hipMemcpyKind kind = c.host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  Dims d2 = dm; d2.B = nb;
  hipLaunchKernelGGL(k_output, dim3((nb + 63) / 64), dim3(64), 0, st, ln->ws, d2, ln->d_abk, ln->d_abk + nb, ln->d_abk + 2 * nb,
                     ln->d_si, ln->d_si + nb, ln->d_info);
  if (c.Hc) HIPCHK(hipMemcpyAsync(c.Hc + so * nn, ln->ws.Hc, BP * nn * sizeof(double), kind, st));
  if (c.dHc) HIPCHK(hipMemcpyAsync(c.dHc + so * nn, ln->ws.dHc, BP * nn * sizeof(double), kind, st));
  if (c.P) HIPCHK(hipMemcpyAsync(c.P + so * nxx, ln->ws.Pout, BP * nxx * sizeof(double), kind, st));
  if (c.FgF && c.jr > 0) HIPCHK(hipMemcpyAsync(c.FgF + so * c.jr, ln->ws.Fg, BP * c.jr * sizeof(double), kind, st));
  if (c.T && c.step3) HIPCHK(hipMemcpyAsync(c.T + so * nn, ln->ws.Tout, BP * nn * sizeof(double), kind, st));
  if (c.alpha) HIPCHK(hipMemcpyAsync(c.alpha + off, ln->d_abk, nb * sizeof(double), kind, st));
  if (c.beta) HIPCHK(hipMemcpyAsync(c.beta + off, ln->d_abk + nb, nb * sizeof(double), kind, st));
  if (c.kappa) HIPCHK(hipMemcpyAsync(c.kappa + off, ln->d_abk + 2 * nb, nb * sizeof(double), kind, st));
  if (c.status) HIPCHK(hipMemcpyAsync(c.status + off, ln->d_si, nb * sizeof(int32_t), kind, st));
  if (c.iters) HIPCHK(hipMemcpyAsync(c.iters + off, ln->d_si + nb, nb * sizeof(int32_t), kind, st));
  if (c.info) HIPCHK(hipMemcpyAsync(c.info + (size_t)off * TMPC_INFO_STRIDE, ln->d_info, (size_t)nb * TMPC_INFO_STRIDE * sizeof(double), kind, st));
  HIPCHK(hipStreamSynchronize(st));
  return TMPC_OK;
}

// Split the batch into waves of `chunk` problems and every wave over the lanes (contiguous slices, lane 0 first); lanes 1.. run
// on their own host threads.  `user` = the caller's stream for device-resident calls: the lanes start after the work queued on
// it so far, and the call returns with every result written (the lanes have been synchronised).
static int dispatch(tmpc_handle* h, const Call& c, hipStream_t user, bool has_user) {
  if (has_user) {
    HIPCHK(hipEventRecord(h->ev_in, user));
    for (int l = 0; l < h->nlanes; ++l) HIPCHK(hipStreamWaitEvent(h->lane[l].st, h->ev_in, 0));
  }
  const bool prof = (h->flags & TMPC_FLAG_PROFILE) != 0;
  std::chrono::steady_clock::time_point t0;
  if (prof) t0 = std::chrono::steady_clock::now();
  for (int l = 0; l < h->nlanes; ++l) h->lane[l].last_nb = 0;
  // (profile mode: ONE lane -- the hipEvent phase times of concurrent lanes overlap, and their sum is not wall time: ADVICE r3)
  const int wave = prof ? h->dm.B : h->chunk;
  for (int off = 0; off < c.nbt; off += wave) {
    const int nw = std::min(wave, c.nbt - off);
    // balanced contiguous slices (a lane never gets more than its capacity dm.B)
    const int nl = prof ? 1 : std::min(h->nlanes, nw);
    int lo[MAXL + 1];
    for (int l = 0; l <= nl; ++l) lo[l] = (int)((long)nw * l / nl);
    int rcs[MAXL]; for (int l = 0; l < MAXL; ++l) { rcs[l] = TMPC_OK; h->lane[l].last_nb = 0; }
    for (int l = 1; l < nl; ++l) {
      Lane& ln = h->lane[l];
      if (ln.worker && ln.worker->pid != getpid()) LaneWorker::retire(ln.worker);      // forked child: leaked, a fresh worker is started below
      if (!ln.worker) { ln.worker = new LaneWorker(); ln.worker->start(); }
      ln.worker->submit([&, l]() {
        rcs[l] = lane_run(h, &h->lane[l], c, off + lo[l], lo[l + 1] - lo[l]);
        if (rcs[l] != TMPC_OK) snprintf(h->lane[l].err, sizeof(h->lane[l].err), "%s", g_err);
      });
    }
    rcs[0] = lane_run(h, &h->lane[0], c, off + lo[0], lo[1] - lo[0]);
    for (int l = 1; l < nl; ++l) h->lane[l].worker->wait();
    for (int l = 0; l < nl; ++l)
      if (rcs[l] != TMPC_OK) { if (l > 0) snprintf(g_err, sizeof(g_err), "%s", h->lane[l].err); return rcs[l]; }
  }
  if (prof) h->lane[0].prof[6] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return TMPC_OK;
}

int tmpc_convexify_batch_device(tmpc_handle* h, int nbt, const double* dA, const double* dB, const double* dH,
                                double* Hc, double* dHc, double* P, double* alpha, double* beta, double* kappa,
                                int32_t* status, int32_t* iters, double* info, void* stream) {
  if (h && nbt == 0) return TMPC_OK;           // empty shard
  if (!h || nbt < 1 || !dA || !dH || (h->dm.mb > 0 && !dB)) return TMPC_E_ARG;
  ON_DEVICE(h);
  Call c{nbt, dA, dB, dH, nullptr, nullptr, 0.0, Hc, dHc, P, nullptr, alpha, beta, kappa, status, iters, info, false, 0};
  return dispatch(h, c, (hipStream_t)stream, true);
}

int tmpc_convexify_con_batch_device(tmpc_handle* h, int nbt, const double* dA, const double* dB, const double* dH, const double* dJ,
                                    const int32_t* d_ncnt, double rho, double* Hc, double* dHc, double* P, double* FgF, double* alpha,
                                    double* beta, double* kappa, int32_t* status, int32_t* iters, double* info, void* stream) {
  if (h && nbt == 0) return TMPC_OK;           // empty shard
  if (!h || nbt < 1 || !dA || !dH || !dJ || (h->dm.mb > 0 && !dB)) return TMPC_E_ARG;
  const Dims& dm = h->dm;
  if (d_ncnt ? (dm.nz <= dm.nr || !(rho >= 0.0)) : dm.ng < 1) {
    snprintf(g_err, sizeof(g_err), "handle has no room for this call (Step 2 needs tmpc_create_con with nc > 0 and rho >= 0, Step 1 with G needs ng > 0)");
    return TMPC_E_ARG;
  }
  ON_DEVICE(h);
  Call c{nbt, dA, dB, dH, dJ, d_ncnt, rho, Hc, dHc, P, FgF, alpha, beta, kappa, status, iters, info, false, d_ncnt ? dm.nr : dm.ng};
  return dispatch(h, c, (hipStream_t)stream, true);
}

int tmpc_convexify_batch_host(tmpc_handle* h, int nbt, const double* A, const double* B, const double* H,
                              double* Hc, double* dHc, double* P, double* alpha, double* beta, double* kappa,
                              int32_t* status, int32_t* iters, double* info) {
  if (h && nbt == 0) return TMPC_OK;
  if (!h || nbt < 1 || !A || !H || (h->dm.mb > 0 && !B)) return TMPC_E_ARG;
  ON_DEVICE(h);
  Call c{nbt, A, B, H, nullptr, nullptr, 0.0, Hc, dHc, P, nullptr, alpha, beta, kappa, status, iters, info, true, 0};
  return dispatch(h, c, nullptr, false);
}

int tmpc_convexify_eq_batch_host(tmpc_handle* h, int nbt, const double* A, const double* B, const double* H, const double* G,
                                 double* Hc, double* dHc, double* P, double* Fg, double* alpha, double* beta, double* kappa,
                                 int32_t* status, int32_t* iters, double* info) {
  if (h && nbt == 0) return TMPC_OK;
  if (!h || nbt < 1 || !A || !H || !G || !Fg || (h->dm.mb > 0 && !B)) return TMPC_E_ARG;
  if (h->dm.ng < 1) { snprintf(g_err, sizeof(g_err), "handle was created without equality-constraint rows (use tmpc_create_eq)"); return TMPC_E_ARG; }
  ON_DEVICE(h);
  Call c{nbt, A, B, H, G, nullptr, 0.0, Hc, dHc, P, Fg, alpha, beta, kappa, status, iters, info, true, h->dm.ng};
  return dispatch(h, c, nullptr, false);
}

int tmpc_convexify_step2_batch_host(tmpc_handle* h, int nbt, const double* A, const double* B, const double* H, const double* J,
                                    const int32_t* ncnt, double rho, double* Hc, double* dHc, double* P, double* FgF, double* alpha,
                                    double* beta, double* kappa, int32_t* status, int32_t* iters, double* info) {
  if (h && nbt == 0) return TMPC_OK;
  if (!h || nbt < 1 || !A || !H || !J || !ncnt || !FgF || !(rho >= 0.0) || (h->dm.mb > 0 && !B)) return TMPC_E_ARG;      // rho = 0: beta-only objective
  const Dims& dm = h->dm;
  if (dm.nz <= dm.nr) { snprintf(g_err, sizeof(g_err), "handle was created without active-constraint rows (use tmpc_create_con with nc > 0)"); return TMPC_E_ARG; }
  const int ncmax = dm.nr - dm.ng;
  for (size_t i = 0; i < (size_t)nbt * dm.p; ++i)
    if (ncnt[i] < 0 || ncnt[i] > ncmax) { snprintf(g_err, sizeof(g_err), "ncnt[%zu]=%d outside 0..%d", i, ncnt[i], ncmax); return TMPC_E_ARG; }
  ON_DEVICE(h);
  Call c{nbt, A, B, H, J, ncnt, rho, Hc, dHc, P, FgF, alpha, beta, kappa, status, iters, info, true, dm.nr};
  return dispatch(h, c, nullptr, false);
}

int tmpc_convexify_step3_batch_host(tmpc_handle* h, int nbt, const double* A, const double* B, const double* H, double rho,
                                    double* Hc, double* dHc, double* P, double* T, double* alpha, double* beta, double* kappa,
                                    int32_t* status, int32_t* iters, double* info) {
  if (h && nbt == 0) return TMPC_OK;
  if (!h || nbt < 1 || !A || !H || !(rho > 0.0) || (h->dm.mb > 0 && !B)) return TMPC_E_ARG;
  if (h->dm.nT < 1) { snprintf(g_err, sizeof(g_err), "handle was created without room for Step 3 (use tmpc_create_step3)"); return TMPC_E_ARG; }
  ON_DEVICE(h);
  Call c{nbt, A, B, H, nullptr, nullptr, rho, Hc, dHc, P, nullptr, alpha, beta, kappa, status, iters, info, true, 0};
  c.step3 = true; c.T = T;
  return dispatch(h, c, nullptr, false);
}

// Step 3 with the multipliers of G (and, with ncnt, of C and the norm terms of Step 2) in the same solve (convexifier.py:144:
// setUpModelPicos(..., constr = constraint_contribution, force = True)).  J, ncnt, FgF as in tmpc_convexify_step2_batch_host
// (ncnt NULL: J holds only the ng rows of G, cost-free multipliers as in tmpc_convexify_eq_batch_host).
int tmpc_convexify_step3_con_batch_host(tmpc_handle* h, int nbt, const double* A, const double* B, const double* H, const double* J,
                                        const int32_t* ncnt, double rho, double* Hc, double* dHc, double* P, double* FgF, double* T,
                                        double* alpha, double* beta, double* kappa, int32_t* status, int32_t* iters, double* info) {
  if (h && nbt == 0) return TMPC_OK;
  if (!h || nbt < 1 || !A || !H || !J || !FgF || !(rho > 0.0) || (h->dm.mb > 0 && !B)) return TMPC_E_ARG;
  const Dims& dm = h->dm;
  if (dm.nT < 1 || dm.nr < 1) { snprintf(g_err, sizeof(g_err), "handle was created without room for Step 3 and constraint rows (use tmpc_create_step3_con)"); return TMPC_E_ARG; }
  if (ncnt) {
    if (dm.nz <= dm.nr) { snprintf(g_err, sizeof(g_err), "handle was created without active-constraint rows (nc > 0)"); return TMPC_E_ARG; }
    const int ncmax = dm.nr - dm.ng;
    for (size_t i = 0; i < (size_t)nbt * dm.p; ++i)
      if (ncnt[i] < 0 || ncnt[i] > ncmax) { snprintf(g_err, sizeof(g_err), "ncnt[%zu]=%d outside 0..%d", i, ncnt[i], ncmax); return TMPC_E_ARG; }
  }
  ON_DEVICE(h);
  Call c{nbt, A, B, H, J, ncnt, rho, Hc, dHc, P, FgF, alpha, beta, kappa, status, iters, info, true, ncnt ? dm.nr : dm.ng};
  c.step3 = true; c.T = T;
  return dispatch(h, c, nullptr, false);
}

// Device-resident forms of the two Step 3 entries (round 4): device pointers in and out, the lanes start after the work queued on `stream`.
// d_ncnt is checked on the device side only through the padding (rows beyond the count must be zero, as in tmpc_convexify_con_batch_device).
int tmpc_convexify_step3_batch_device(tmpc_handle* h, int nbt, const double* dA, const double* dB, const double* dH, double rho, double* Hc, double* dHc,
                                      double* P, double* T, double* alpha, double* beta, double* kappa, int32_t* status, int32_t* iters, double* info, void* stream) {
  if (h && nbt == 0) return TMPC_OK;
  if (!h || nbt < 1 || !dA || !dH || !(rho > 0.0) || (h->dm.mb > 0 && !dB)) return TMPC_E_ARG;
  if (h->dm.nT < 1) { snprintf(g_err, sizeof(g_err), "handle was created without room for Step 3 (use tmpc_create_step3)"); return TMPC_E_ARG; }
  ON_DEVICE(h);
  Call c{nbt, dA, dB, dH, nullptr, nullptr, rho, Hc, dHc, P, nullptr, alpha, beta, kappa, status, iters, info, false, 0};
  c.step3 = true; c.T = T;
  return dispatch(h, c, (hipStream_t)stream, true);
}
int tmpc_convexify_step3_con_batch_device(tmpc_handle* h, int nbt, const double* dA, const double* dB, const double* dH, const double* dJ, const int32_t* d_ncnt,
                                          double rho, double* Hc, double* dHc, double* P, double* FgF, double* T, double* alpha, double* beta, double* kappa,
                                          int32_t* status, int32_t* iters, double* info, void* stream) {
  if (h && nbt == 0) return TMPC_OK;
  if (!h || nbt < 1 || !dA || !dH || !dJ || !FgF || !(rho > 0.0) || (h->dm.mb > 0 && !dB)) return TMPC_E_ARG;
  const Dims& dm = h->dm;
  if (dm.nT < 1 || dm.nr < 1) { snprintf(g_err, sizeof(g_err), "handle was created without room for Step 3 and constraint rows (use tmpc_create_step3_con)"); return TMPC_E_ARG; }
  if (d_ncnt && dm.nz <= dm.nr) { snprintf(g_err, sizeof(g_err), "handle was created without active-constraint rows (nc > 0)"); return TMPC_E_ARG; }
  ON_DEVICE(h);
  Call c{nbt, dA, dB, dH, dJ, d_ncnt, rho, Hc, dHc, P, FgF, alpha, beta, kappa, status, iters, info, false, d_ncnt ? dm.nr : dm.ng};
  c.step3 = true; c.T = T;
  return dispatch(h, c, (hipStream_t)stream, true);
}

int tmpc_supplement_terms_batch_host(tmpc_handle* hh, int nbt, const double* A, const double* B, const double* P, int nr,
                                     const double* J, const double* wts, const double* T, double* dHc) {
  if (!hh || nbt < 1 || !A || !P || !dHc || (hh->dm.mb > 0 && !B) || nr < 0 || ((J != nullptr) != (wts != nullptr)) || (J && nr < 1)) return TMPC_E_ARG;
  hipStream_t st = 0;
  const Dims& dm = hh->dm;
  Lane* h = &hh->lane[0];                               // lane 0's buffers (this entry is not on the hot path)
  const size_t nn = (size_t)dm.n * dm.n, nxx = (size_t)dm.nx * dm.nx, nxm = (size_t)dm.nx * dm.mb;
  ON_DEVICE(hh);
  DevBuf bJ, bw;
  const size_t capst = (size_t)dm.B * dm.p;             // stages per chunk
  if (J) { HIPCHK(bJ.alloc(capst * nr * dm.n * 8)); HIPCHK(bw.alloc(capst * nr * 8)); }
  double* dJ = J ? bJ.as<double>() : nullptr; double* dw = J ? bw.as<double>() : nullptr;
  for (int off = 0; off < nbt; off += dm.B) {
    const int nb = std::min(dm.B, nbt - off);
    const size_t BP = (size_t)nb * dm.p, so = (size_t)off * dm.p;
    Dims d2 = dm; d2.B = nb;
    HIPCHK(hipMemcpyAsync(h->dA, A + so * nxx, BP * nxx * sizeof(double), hipMemcpyHostToDevice, st));
    if (dm.mb > 0) HIPCHK(hipMemcpyAsync(h->dB, B + so * nxm, BP * nxm * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(h->ws.P, P + so * nxx, BP * nxx * sizeof(double), hipMemcpyHostToDevice, st));
    if (J) {
      HIPCHK(hipMemcpyAsync(dJ, J + so * nr * dm.n, BP * nr * dm.n * 8, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(dw, wts + so * nr, BP * nr * 8, hipMemcpyHostToDevice, st));
    }
    if (T) HIPCHK(hipMemcpyAsync(h->dH, T + so * nn, BP * nn * sizeof(double), hipMemcpyHostToDevice, st));
    if (dm.n > NMAX) {      // generic form (tmpc_big.h): the supplement of P alone
      hipLaunchKernelGGL(kb_supplement, dim3((unsigned)BP), dim3(256), 0, st, (const double*)h->dA, (const double*)h->dB, (const double*)h->ws.P, h->ws.dHc, d2, h->ws.bscr,
                         nr, (const double*)(J ? dJ : nullptr), (const double*)(J ? dw : nullptr), (const double*)(T ? h->dH : nullptr));
    } else
    hipLaunchKernelGGL(k_supplement, dim3((unsigned)BP), dim3(64), slots_bytes(5), st, h->dA, h->dB, h->ws.P, h->ws.dHc, d2, nr, dJ, dw,
                       T ? h->dH : nullptr);
    HIPCHK(hipMemcpyAsync(dHc + so * nn, h->ws.dHc, BP * nn * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  return TMPC_OK;
}

int tmpc_pack_sensitivities_host(tmpc_handle* h, int nb, int nh, const double* C, const double* mu, const double* Hbig, double thr, int ncmax,
                                 double* C_As, int32_t* ncnt, int32_t* idx, double* q, double* Hst) {
  if (!h || nb < 1 || nh < 0 || ncmax < 0 || ((C != nullptr) != (mu != nullptr)) || (C && (nh < 1 || ncmax < 1 || !C_As || !ncnt)) || (Hbig && !Hst)) return TMPC_E_ARG;
  ON_DEVICE(h);
  const int p = h->dm.p, n = h->dm.n;
  const size_t BP = (size_t)nb * p, ld = (size_t)p * n;
  DevBuf bC, bmu, bH, bCA, bnc, bidx, bq, bHs;
  if (C) {
    HIPCHK(bC.alloc(BP * nh * n * 8)); HIPCHK(bmu.alloc(BP * nh * 8)); HIPCHK(bCA.alloc(BP * ncmax * n * 8)); HIPCHK(bnc.alloc(BP * 4)); HIPCHK(bidx.alloc(BP * nh * 4));
    HIPCHK(hipMemcpy(bC.p, C, BP * nh * n * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(bmu.p, mu, BP * nh * 8, hipMemcpyHostToDevice));
  }
  if (q) HIPCHK(bq.alloc(BP * n * 8));
  if (Hbig) { HIPCHK(bH.alloc((size_t)nb * ld * ld * 8)); HIPCHK(bHs.alloc(BP * n * n * 8)); HIPCHK(hipMemcpy(bH.p, Hbig, (size_t)nb * ld * ld * 8, hipMemcpyHostToDevice)); }
  hipLaunchKernelGGL(k_pack_sens, dim3((unsigned)BP), dim3(64), 0, 0, C ? bC.as<double>() : nullptr, C ? bmu.as<double>() : nullptr,
                     Hbig ? bH.as<double>() : nullptr, thr, nh, ncmax, p, n, C ? bCA.as<double>() : nullptr, C ? bnc.as<int32_t>() : nullptr,
                     (C && idx) ? bidx.as<int32_t>() : nullptr, q ? bq.as<double>() : nullptr, Hbig ? bHs.as<double>() : nullptr);
  HIPCHK(hipDeviceSynchronize());
  if (C) {
    HIPCHK(hipMemcpy(C_As, bCA.p, BP * ncmax * n * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(ncnt, bnc.p, BP * 4, hipMemcpyDeviceToHost));
    if (idx) HIPCHK(hipMemcpy(idx, bidx.p, BP * nh * 4, hipMemcpyDeviceToHost));
  }
  if (q) HIPCHK(hipMemcpy(q, bq.p, BP * n * 8, hipMemcpyDeviceToHost));
  if (Hbig) HIPCHK(hipMemcpy(Hst, bHs.p, BP * n * n * 8, hipMemcpyDeviceToHost));
  return TMPC_OK;
}

int tmpc_supplement_batch_host(tmpc_handle* h, int nbt, const double* A, const double* B, const double* P, double* dHc) {
  return tmpc_supplement_terms_batch_host(h, nbt, A, B, P, 0, nullptr, nullptr, nullptr, dHc);
}

int tmpc_eig_scan_host(tmpc_handle* hh, int nbt, const double* H, double* out) {
  if (!hh || nbt < 1 || !H || !out) return TMPC_E_ARG;
  ON_DEVICE(hh);
  Lane* h = &hh->lane[0];
  const Dims& dm = hh->dm;
  hipStream_t st = 0;
  const size_t nn = (size_t)dm.n * dm.n;
  for (int off = 0; off < nbt; off += dm.B) {
    const int nb = std::min(dm.B, nbt - off);
    const size_t BP = (size_t)nb * dm.p;
    HIPCHK(hipMemcpyAsync(h->dH, H + (size_t)off * dm.p * nn, BP * nn * sizeof(double), hipMemcpyHostToDevice, st));
    if (dm.n > NMAX) hipLaunchKernelGGL(kb_eig_scan, dim3((unsigned)BP), dim3(256), (size_t)BIG_EIG_LDS * sizeof(double), st, (const double*)h->dH, h->ws.part, dm.n);
    else hipLaunchKernelGGL(k_eig_scan, dim3((unsigned)BP), dim3(64), slots_bytes(2), st, h->dH, h->ws.part, dm.n);
    HIPCHK(hipMemcpyAsync(out + (size_t)off * dm.p * 4, h->ws.part, BP * 4 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  return TMPC_OK;
}

// out = A + V diag(max(tol - lambda, 0)) V' for nb symmetric n x n matrices (tmpc_eig.h; reference: sqp_method.py:327-403).
// No handle: any n >= 1; the device is the current one.
// device scratch of tmpc_eig_clip_host, kept between calls (the reference calls this once per SQP iteration: no hipMalloc / hipFree
// pair inside that loop); one per host thread and device, grown on demand
struct EigScratch {
  void* p = nullptr; size_t bytes = 0; int device = -1;
  ~EigScratch() { if (p) hipFree(p); }
  hipError_t reserve(size_t need) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (p && dev == device && bytes >= need) return hipSuccess;
    if (p) { hipFree(p); p = nullptr; bytes = 0; }
    e = hipMalloc(&p, need);
    if (e == hipSuccess) { bytes = need; device = dev; }
    return e;
  }
};
static thread_local EigScratch g_eig_scratch;

int tmpc_eig_clip_host(int nb, int n, const double* A, double tol, double* out, double* evals, double* reg, int32_t* sweeps) {
  if (nb < 1 || n < 1 || !A || !out) return TMPC_E_ARG;
  const size_t nn = (size_t)n * n, mat = (size_t)nb * nn;
  const size_t small = (size_t)nb * (3 + 2 * (size_t)n);    // shift | evals | lift | reg | offmax
  HIPCHK(g_eig_scratch.reserve((4 * mat + small) * 8));
  double* dA = (double*)g_eig_scratch.p; double* dU = dA + mat; double* dV = dU + mat; double* dO = dV + mat;
  double* dshift = dO + mat; double* dev = dshift + nb; double* dlift = dev + (size_t)nb * n; double* dreg = dlift + (size_t)nb * n; double* doff = dreg + nb;
  HIPCHK(hipMemcpy(dA, A, mat * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_eig_init, dim3(nb), dim3(256), 0, 0, dA, dU, dV, dshift, n);
  const int m = (n + 1) & ~1;
  const double thr = std::max(1e-15, 2.0 * sqrt((double)n) * 2.220446049250313e-16);     // ~ the rounding level of an n-long dot product (random-walk bound): smaller rotations never settle;
                                                                                          // 2 n eps (the worst-case bound) left 2e-9 relative error in the clipped eigenvalues at n = 257
  int sw = 0;
  bool converged = (n == 1);
  std::vector<double> off(nb);
  if (n > 1) {
    for (; sw < EIG_MAX_SWEEPS; ++sw) {
      HIPCHK(hipMemsetAsync(doff, 0, (size_t)nb * 8, 0));
      for (int r = 0; r < m - 1; ++r) hipLaunchKernelGGL(k_eig_round, dim3(m / 2, nb), dim3(256), 0, 0, dU, dV, n, m, r, doff, thr);
      HIPCHK(hipMemcpy(off.data(), doff, (size_t)nb * 8, hipMemcpyDeviceToHost));
      double worst = 0.0;
      for (double v : off) if (v > worst) worst = v;
      if (!(worst > thr)) { ++sw; converged = true; break; }   // a full sweep without a rotation
    }
  }
  HIPCHK(hipMemsetAsync(dreg, 0, (size_t)nb * 8, 0));
  hipLaunchKernelGGL(k_eig_values, dim3(n, nb), dim3(256), 0, 0, dU, dshift, dev, dlift, dreg, n, tol);
  hipLaunchKernelGGL(k_eig_apply, dim3((unsigned)((nn + 255) / 256), nb), dim3(256), 0, 0, dA, dV, dlift, dO, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, dO, mat * 8, hipMemcpyDeviceToHost));
  if (evals) HIPCHK(hipMemcpy(evals, dev, (size_t)nb * n * 8, hipMemcpyDeviceToHost));
  if (reg) HIPCHK(hipMemcpy(reg, dreg, (size_t)nb * 8, hipMemcpyDeviceToHost));
  if (sweeps) for (int b = 0; b < nb; ++b) sweeps[b] = sw;
  if (!converged) {                                            // outputs hold the last iterate; the caller must not take them for converged
    double worst = 0.0;
    for (double v : off) if (v > worst) worst = v;
    snprintf(g_err, sizeof(g_err), "tmpc_eig_clip_host: %d Jacobi sweeps without convergence (largest relative off-diagonal %.3e, threshold %.3e)", sw, worst, thr);
    return TMPC_E_NOCONV;
  }
  return TMPC_OK;
}

int tmpc_tracking_reference_host(tmpc_handle* h, int nstage, const double* Hc, const double* q, const double* wref,
                                 double ts, double* W, double* yref, int32_t* info) {
  if (!h || nstage < 1 || !Hc || !q || !wref || !yref || !(ts > 0.0)) return TMPC_E_ARG;
  ON_DEVICE(h);
  const int n = h->dm.n;
  const size_t nn = (size_t)n * n;
  DevBuf bH, bW, bv, bi;                                   // bv: q | wref | yref
  HIPCHK(bH.alloc((size_t)nstage * nn * 8));
  if (W) HIPCHK(bW.alloc((size_t)nstage * nn * 8));
  HIPCHK(bv.alloc((size_t)nstage * n * 3 * 8));
  HIPCHK(bi.alloc((size_t)nstage * sizeof(int)));
  double* dH = bH.as<double>(); double* dW = W ? bW.as<double>() : nullptr; double* dv = bv.as<double>(); int* di = bi.as<int>();
  HIPCHK(hipMemcpy(dH, Hc, (size_t)nstage * nn * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dv, q, (size_t)nstage * n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dv + (size_t)nstage * n, wref, (size_t)nstage * n * 8, hipMemcpyHostToDevice));
  DevBuf bscr;
  if (n > NMAX) {
    HIPCHK(bscr.alloc((size_t)nstage * nn * 8));
    hipLaunchKernelGGL(kb_tracking_ref, dim3((unsigned)nstage), dim3(256), 0, 0, (const double*)dH, (const double*)dv, (const double*)(dv + (size_t)nstage * n), 1.0 / ts, dW,
                       dv + (size_t)2 * nstage * n, di, n, bscr.as<double>());
  } else
  hipLaunchKernelGGL(k_tracking_ref, dim3((unsigned)nstage), dim3(64), (size_t)(MS + 64) * sizeof(double), 0, dH, dv, dv + (size_t)nstage * n,
                     1.0 / ts, dW, dv + (size_t)2 * nstage * n, di, n);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(yref, dv + (size_t)2 * nstage * n, (size_t)nstage * n * 8, hipMemcpyDeviceToHost));
  if (W) HIPCHK(hipMemcpy(W, dW, (size_t)nstage * nn * 8, hipMemcpyDeviceToHost));
  if (info) HIPCHK(hipMemcpy(info, di, (size_t)nstage * sizeof(int), hipMemcpyDeviceToHost));
  return TMPC_OK;
}

// ---------------------------------------------------------------------------------- debug / unit-test entries (tunempc_hip_debug.h)
int tmpc_debug_gemm_nt(tmpc_handle* h, double* C, const double* A, const double* B, int M, int N, int K, int mode, int lower) {
  const int shape = mode >> 4; mode &= 15;             // mode + 16: the LDS-DMA tile core (one workgroup per tile); + 32: the same with K split into two operand pairs
  if (!h || !C || !A || !B || M % 16 || N % 16 || K % 16 || K < 16 || mode < 0 || mode > 2 || shape > 2 || (shape == 2 && K % 32)) return TMPC_E_ARG;
  ON_DEVICE(h);
  DevBuf bC, bA, bB;
  HIPCHK(bC.alloc((size_t)M * N * 8)); HIPCHK(bA.alloc((size_t)M * K * 8)); HIPCHK(bB.alloc((size_t)N * K * 8));
  double* dC = bC.as<double>(); double* dA = bA.as<double>(); double* dB = bB.as<double>();
  HIPCHK(hipMemcpy(dC, C, (size_t)M * N * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dA, A, (size_t)M * K * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dB, B, (size_t)N * K * 8, hipMemcpyHostToDevice));
  if (shape >= 1) {
    hipLaunchKernelGGL(k_debug_gemm_dma, dim3(((M + 63) / 64) * ((N + 63) / 64)), dim3(256), (size_t)dma_lds_doubles<UPD_DMA_DEPTH>() * sizeof(double), 0,
                       dC, dA, dB, M, N, K, mode, lower, shape == 2 ? K / 2 : 0);
  } else if (h->flags & TMPC_FLAG_NO_MFMA) hipLaunchKernelGGL(k_debug_gemm<false>, dim3(1), dim3(256), factor_lds(), 0, dC, dA, dB, M, N, K, mode, lower);
  else hipLaunchKernelGGL(k_debug_gemm<true>, dim3(1), dim3(256), factor_lds(), 0, dC, dA, dB, M, N, K, mode, lower);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(C, dC, (size_t)M * N * 8, hipMemcpyDeviceToHost));
  return TMPC_OK;
}

// smallest eigenvalue of nmat symmetric n x n matrices (the step-length primitive of k_eigmin)
int tmpc_debug_min_eig(tmpc_handle* h, int nmat, int n, const double* W, double* out) {
  if (!h || nmat < 1 || n < 1 || n > NMAX || !W || !out) return TMPC_E_ARG;
  ON_DEVICE(h);
  DevBuf bW, bO;
  HIPCHK(bW.alloc((size_t)nmat * n * n * 8)); HIPCHK(bO.alloc((size_t)nmat * 8));
  HIPCHK(hipMemcpy(bW.p, W, (size_t)nmat * n * n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_debug_min_eig, dim3(nmat), dim3(64), (size_t)(MS + 160) * sizeof(double), 0, bW.as<double>(), bO.as<double>(), n);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, bO.p, (size_t)nmat * 8, hipMemcpyDeviceToHost));
  return TMPC_OK;
}

int tmpc_debug_min_eig_lane(tmpc_handle* h, int nmat, int n, const double* W, double* out) {
  if (!h || nmat < 1 || n < 1 || n > 8 || !W || !out) return TMPC_E_ARG;
  ON_DEVICE(h);
  DevBuf bW, bO;
  HIPCHK(bW.alloc((size_t)nmat * n * n * 8)); HIPCHK(bO.alloc((size_t)nmat * 8));
  HIPCHK(hipMemcpy(bW.p, W, (size_t)nmat * n * n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_debug_min_eig_lane, dim3((nmat + 63) / 64), dim3(64), 0, 0, bW.as<double>(), bO.as<double>(), n, nmat);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, bO.p, (size_t)nmat * 8, hipMemcpyDeviceToHost));
  return TMPC_OK;
}

// The elimination schedule of the block factorisation for period p (host only, no device): out = [nlev, prep, nelim, nupd,
// levels (eoff, nelim, uoff, nupd) x nlev, elimination records x 8 ints, update records x 8 ints, orientation x p].
// Returns the number of ints (call with cap = 0 to size the buffer), or TMPC_E_ARG.
int tmpc_debug_cr_schedule(int p, int32_t* out, int cap) {
  if (p < 1) return TMPC_E_ARG;
  const CrSched sc = cr_build(p);
  std::vector<int> flat;
  flat.push_back((int)sc.lev.size()); flat.push_back(sc.prep); flat.push_back((int)sc.elim.size() / CR_EW); flat.push_back((int)sc.upd.size() / CR_UW);
  for (const CrLevel& lv : sc.lev) { flat.push_back(lv.eoff); flat.push_back(lv.nelim); flat.push_back(lv.uoff); flat.push_back(lv.nupd); }
  flat.insert(flat.end(), sc.elim.begin(), sc.elim.end());
  flat.insert(flat.end(), sc.upd.begin(), sc.upd.end());
  flat.insert(flat.end(), sc.orient.begin(), sc.orient.end());
  if (out && cap >= (int)flat.size()) memcpy(out, flat.data(), flat.size() * sizeof(int));
  return (int)flat.size();
}

// stand-alone workspace for the unit test / timing of the block factorisation: nb copies of one system
struct CrBench {
  WS w; Dims dm; CrSched sc; int* d_sched = nullptr; DevBuf D, O, F, Li, dd, Z, W3, ip, pr, al;
  DevBuf O32b; int ablate = 0;
  int init(int nb, int p, int d, int flags, bool lowp = false) {
    memset(&w, 0, sizeof(w)); memset(&dm, 0, sizeof(dm));
    dm.B = nb; dm.p = p; dm.d = d; dm.dp = (d + 15) / 16 * 16; dm.nt = (dm.dp + TB - 1) / TB; dm.flags = flags & TMPC_FLAG_NO_MFMA;
    const size_t bs = (size_t)dm.dp * dm.dp, per = (size_t)p * bs;
    sc = cr_build(p);
    int rc = cr_upload(sc, &d_sched);
    if (rc != TMPC_OK) return rc;
    if (D.alloc(nb * per * 8) != hipSuccess || O.alloc(nb * per * 8) != hipSuccess || F.alloc(nb * per * 8) != hipSuccess ||
        Li.alloc((size_t)nb * p * dm.nt * TB * TB * 8) != hipSuccess || dd.alloc((size_t)nb * p * dm.dp * 8) != hipSuccess ||
        Z.alloc((size_t)nb * p * dm.dp * 8) != hipSuccess || W3.alloc((size_t)nb * p * dm.dp * 3 * 8) != hipSuccess ||
        ip.alloc((size_t)nb * IS * sizeof(int)) != hipSuccess || pr.alloc((size_t)nb * PS * 8) != hipSuccess || al.alloc((size_t)nb * sizeof(int)) != hipSuccess)
      return TMPC_E_NOMEM;
    w.D = D.as<double>(); w.O = O.as<double>(); w.F = F.as<double>(); w.Linv = Li.as<double>(); w.Ddiag = dd.as<double>();
    w.Z = Z.as<double>(); w.W3 = W3.as<double>(); w.iprob = ip.as<int>(); w.prob = pr.as<double>(); w.alist = al.as<int>();
    std::vector<int> ids(nb); for (int i = 0; i < nb; ++i) ids[i] = i;
    std::vector<double> one((size_t)nb * PS, 1.0);
    if (hipMemcpy(w.alist, ids.data(), nb * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return TMPC_E_HIP;
    if (hipMemset(w.iprob, 0, (size_t)nb * IS * sizeof(int)) != hipSuccess) return TMPC_E_HIP;       // phase = PH_MAIN
    if (hipMemcpy(w.prob, one.data(), one.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return TMPC_E_HIP;
    if (lowp && dm.dp > 64 && dm.nt <= TRR_NT) {
      dm.flags |= DF_LOWP;      // every problem with single-precision updates (timing / unit test of wg_tile_dma_f32)
      const size_t n32 = 2 * (size_t)nb * p * dm.dp * ((dm.dp + 31) & ~31);
      if (O32b.alloc(n32 * sizeof(float)) != hipSuccess || hipMemset(O32b.as<float>(), 0, n32 * sizeof(float)) != hipSuccess) return TMPC_E_NOMEM;
      w.O32 = O32b.as<float>();
#ifdef TMPC_ABLATE
      dm.flags |= ablate << 24;
#endif
      std::vector<int> hi((size_t)nb * IS, 0);
      for (int b = 0; b < nb; ++b) hi[(size_t)b * IS + I_LOWP] = 1;
      if (hipMemcpy(w.iprob, hi.data(), hi.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return TMPC_E_HIP;
    }
    return TMPC_OK;
  }
  ~CrBench() { if (d_sched) hipFree(d_sched); }
};

// host image of one system in the storage of tmpc_cr.h: D [p][dp][dp] (identity padding), edge slot k in the orientation of the schedule
static void cr_pack(const CrSched& sc, int p, int d, int dp, const double* D, const double* Ccpl, std::vector<double>& hD, std::vector<double>& hO,
                    std::vector<double>& hdd) {
  const size_t bs = (size_t)dp * dp;
  hD.assign(p * bs, 0.0); hO.assign(p * bs, 0.0); hdd.assign((size_t)p * dp, 1.0);
  for (int k = 0; k < p; ++k) {
    for (int i = 0; i < dp; ++i)
      for (int j = 0; j < dp; ++j) {
        double dv = (i == j) ? 1.0 : 0.0, cv = 0.0;
        if (i < d && j < d) { dv = D[((size_t)k * d + i) * d + j]; cv = Ccpl[((size_t)k * d + i) * d + j]; }    // Ccpl[k] = T[block k, block k+1]
        hD[k * bs + (size_t)i * dp + j] = dv;
        if (sc.orient[k]) hO[k * bs + (size_t)i * dp + j] = cv; else hO[k * bs + (size_t)j * dp + i] = cv;
      }
    for (int i = 0; i < d; ++i) hdd[(size_t)k * dp + i] = D[((size_t)k * d + i) * d + i];
  }
}

// Factor + solve one block-cyclic-tridiagonal system given dense blocks (unit test of the kernels of tmpc_cr.h):
// D [p][d][d] diagonal blocks, Ccpl [p][d][d] with Ccpl[k] = T[block k, block k+1 mod p], rhs/x [p][d].
int tmpc_debug_block_solve(tmpc_handle* h, int p, int d, const double* D, const double* Ccpl, const double* rhs, double* x, int32_t* nshift) {
  if (!h || p < 1 || d < 1 || !D || !Ccpl || !rhs || !x) return TMPC_E_ARG;
  ON_DEVICE(h);
  CrBench cb;
  int rc = cb.init(1, p, d, h->flags, h->opt.lowp_switch >= 1.0);      // (a switch >= 1 is meaningless for a solve: in the two debug entries it means 'every update in single precision')
  if (rc != TMPC_OK) return rc;
  const int dp = cb.dm.dp;
  const size_t bs = (size_t)dp * dp;
  std::vector<double> hD, hO, hdd, hz((size_t)p * dp, 0.0);
  cr_pack(cb.sc, p, d, dp, D, Ccpl, hD, hO, hdd);
  for (int k = 0; k < p; ++k) for (int i = 0; i < d; ++i) hz[(size_t)k * dp + i] = rhs[(size_t)k * d + i];
  HIPCHK(hipMemcpy(cb.w.D, hD.data(), p * bs * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(cb.w.O, hO.data(), p * bs * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(cb.w.Ddiag, hdd.data(), (size_t)p * dp * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(cb.w.Z, hz.data(), (size_t)p * dp * 8, hipMemcpyHostToDevice));
  cr_factor(cb.w, cb.dm, cb.sc, cb.d_sched, cb.w.alist, 1, 0, h->rs, h->mt, nullptr, nullptr, 0, cb.w.O32 ? 1 : 0);
  cr_solve(cb.w, cb.dm, cb.sc, cb.d_sched, cb.w.alist, 1, 0, 2);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipGetLastError());
  int hip_[IS];
  HIPCHK(hipMemcpy(hz.data(), cb.w.Z, (size_t)p * dp * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hip_, cb.w.iprob, sizeof(hip_), hipMemcpyDeviceToHost));
  for (int k = 0; k < p; ++k) for (int i = 0; i < d; ++i) x[(size_t)k * d + i] = hz[(size_t)k * dp + i];
  if (nshift) *nshift = hip_[I_NSHIFT];
  return TMPC_OK;
}

// Isolated timing of the block factorisation and of one single-right-hand-side solve: nb copies of one random SPD
// block-cyclic-tridiagonal system (restored before every repetition).  ms_out2[0] = factorisation, [1] = solve, averages over `reps`.
int tmpc_debug_factor_bench(tmpc_handle* h, int nb, int p, int d, int reps, double* ms_out2) {
  if (!h || nb < 1 || p < 1 || d < 1 || reps < 1 || !ms_out2) return TMPC_E_ARG;
  ON_DEVICE(h);
  CrBench cb;
  cb.ablate = (int)h->opt.lowp_switch - 2;      // (-DTMPC_ABLATE builds: 3, 4, 5, 6 = the ablations of wg_tile_dma_f32)
  int rc = cb.init(nb, p, d, h->flags, h->opt.lowp_switch >= 1.0);
  if (rc != TMPC_OK) return rc;
  const int dp = cb.dm.dp;
  const size_t bs = (size_t)dp * dp, per = (size_t)p * bs;
  // T = sum_k J_k' J_k + I  with J_k = [E_k G_k] on blocks (k, k+1): SPD by construction
  std::vector<double> E((size_t)p * d * d), G((size_t)p * d * d), Dd((size_t)p * d * d, 0.0), Cc((size_t)p * d * d, 0.0);
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return ((double)(st % 2000001) / 1000000.0 - 1.0) / sqrt((double)d); };
  for (auto& v : E) v = rnd();
  for (auto& v : G) v = rnd();
  for (int k = 0; k < p; ++k) {
    const int kn = (k + 1) % p;
    const double* Ek = &E[(size_t)k * d * d]; const double* Gk = &G[(size_t)k * d * d];
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < d; ++j) {
        double ee = 0, gg = 0, eg = 0;
        for (int r = 0; r < d; ++r) { ee += Ek[r * d + i] * Ek[r * d + j]; gg += Gk[r * d + i] * Gk[r * d + j]; eg += Ek[r * d + i] * Gk[r * d + j]; }
        Dd[((size_t)k * d + i) * d + j] += ee + (i == j ? 1.0 : 0.0); Dd[((size_t)kn * d + i) * d + j] += gg;
        Cc[((size_t)k * d + i) * d + j] = eg;
      }
  }
  std::vector<double> hD, hO, hdd;
  cr_pack(cb.sc, p, d, dp, Dd.data(), Cc.data(), hD, hO, hdd);
  DevBuf pD, pO;
  HIPCHK(pD.alloc(per * 8)); HIPCHK(pO.alloc(per * 8));
  HIPCHK(hipMemcpy(pD.p, hD.data(), per * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(pO.p, hO.data(), per * 8, hipMemcpyHostToDevice));
  for (int b = 0; b < nb; ++b) HIPCHK(hipMemcpy(cb.w.Ddiag + (size_t)b * p * dp, hdd.data(), (size_t)p * dp * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(cb.w.Z, 0, (size_t)nb * p * dp * 8));
  hipEvent_t e0, e1, e2; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventCreate(&e2));
  ms_out2[0] = ms_out2[1] = 0.0;
  for (int r = 0; r < reps + 1; ++r) {
    for (int b = 0; b < nb; ++b) {
      HIPCHK(hipMemcpyAsync(cb.w.D + b * per, pD.p, per * 8, hipMemcpyDeviceToDevice, 0));
      HIPCHK(hipMemcpyAsync(cb.w.O + b * per, pO.p, per * 8, hipMemcpyDeviceToDevice, 0));
    }
    HIPCHK(hipEventRecord(e0, 0));
    cr_factor(cb.w, cb.dm, cb.sc, cb.d_sched, cb.w.alist, nb, 0, h->rs, h->mt, nullptr, nullptr, 0, cb.w.O32 ? nb : 0);
    HIPCHK(hipEventRecord(e1, 0));
    cr_solve(cb.w, cb.dm, cb.sc, cb.d_sched, cb.w.alist, nb, 0, 2);
    HIPCHK(hipEventRecord(e2, 0));
    HIPCHK(hipEventSynchronize(e2));
    float ms0, ms1; HIPCHK(hipEventElapsedTime(&ms0, e0, e1)); HIPCHK(hipEventElapsedTime(&ms1, e1, e2));
    if (r > 0) { ms_out2[0] += ms0 / reps; ms_out2[1] += ms1 / reps; }
  }
  HIPCHK(hipGetLastError());
  int nsh = 0; HIPCHK(hipMemcpy(&nsh, cb.w.iprob + I_NSHIFT, sizeof(int), hipMemcpyDeviceToHost));
  if (nsh) { snprintf(g_err, sizeof(g_err), "factor bench: %d shifted pivots", nsh); }
  hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(e2);
  return TMPC_OK;
}

#ifdef TMPC_CYCLE_PROF
int tmpc_debug_cycle_prof(double* out64) {
  unsigned long long hh[64];
  HIPCHK(hipMemcpyFromSymbol(hh, HIP_SYMBOL(tmpc::g_prof), sizeof(hh)));
  for (int i = 0; i < 64; ++i) out64[i] = (double)hh[i];
  unsigned long long z[64] = {0};
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(tmpc::g_prof), z, sizeof(z)));
  return 0;
}
#endif

}  // extern "C"
