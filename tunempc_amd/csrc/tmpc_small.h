// Small dense helpers for the per-stage (nx+nu) x (nx+nu) blocks, fp64: tmpc_small_impl.h, instantiated twice (see there).
#pragma once
#include <hip/hip_runtime.h>

namespace tmpc {

constexpr int NB = 96;            // largest stage block of the generic per-stage kernels (tmpc_big.h): plain model (round 5; 64 before -- the Jacobi tile of 96 x 97 doubles is 75 KB)
constexpr int NBM = 64;           // largest stage block with stage-local multipliers or Step 3: the kernels of tmpc_phi.h / tmpc_t3.h keep one LANE per entry of their n-vectors
constexpr int BIG_SCR = 5;        // n x n scratch matrices per stage of those kernels (WS::bscr)

#define TMPC_SM_NMAX 32
#define TMPC_SM_LD 33
#define TMPC_SM_WSYNC() __syncthreads()
#include "tmpc_small_impl.h"
#undef TMPC_SM_NMAX
#undef TMPC_SM_LD
#undef TMPC_SM_WSYNC

namespace sm8 {
#define TMPC_SM_NMAX 8
#define TMPC_SM_LD 9
#define TMPC_SM_WSYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#include "tmpc_small_impl.h"
#undef TMPC_SM_NMAX
#undef TMPC_SM_LD
#undef TMPC_SM_WSYNC
}  // namespace sm8

}  // namespace tmpc
