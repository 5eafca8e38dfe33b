"""In-kernel cycle split of the register-tile GEMM core (needs the -DTMPC_CYCLE_PROF build libtunempc_hip_prof.so)."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tunempc_amd._lib as L
L.library_path = lambda: os.path.join(ROOT, 'tunempc_amd', 'lib', 'libtunempc_hip_prof.so')
from tunempc_amd._lib import HipConvexifier
h = HipConvexifier(1, 2, 4, 1)
lib = h.lib
lib.tmpc_debug_cycle_prof.argtypes = [C.POINTER(C.c_double)]
out = np.zeros(16)
names = ['3 x (LDS fetch + 100 MFMA)', 'barrier (vmcnt(0) + s_barrier)', 'DMA issue', 'LDS fetch + 100 MFMA (step 3)', 'epilogue']
for shared in (0, 16):
    for var in (1, 2):
        M, N, K = (304, 304, 304) if var == 1 else (304, 64, 256)
        lib.tmpc_debug_cycle_prof(out.ctypes.data_as(C.POINTER(C.c_double)))   # reset
        ms = h.debug_gemm_bench(256, M, N, K, var, shared, 8, 1)
        lib.tmpc_debug_cycle_prof(out.ctypes.data_as(C.POINTER(C.c_double)))
        tot = out[8:13].sum()
        print(f"var {var} {'shared' if shared else 'private'}: {ms:.3f} ms; wave 0 of block 0, cycles (2 launches): total {tot:.3e}")
        for n_, v in zip(names, out[8:13]):
            print(f"    {n_:34s} {v:.3e}  {100*v/tot:5.1f} %")
