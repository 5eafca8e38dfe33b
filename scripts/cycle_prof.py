import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tunempc_amd._lib as L
L.library_path = lambda: os.path.join(ROOT, 'tunempc_amd', 'lib', 'libtunempc_hip_prof.so')
from tunempc_amd._lib import HipConvexifier
h = HipConvexifier(2, 3, 1)
lib = h.lib
lib.tmpc_debug_cycle_prof.argtypes = [C.POINTER(C.c_double)]
out = np.zeros(16)
for nb in (64, 512):
    os.environ['TMPC_ABL'] = '6'
    lib.tmpc_debug_cycle_prof(out.ctypes.data_as(C.POINTER(C.c_double)))   # reset
    ms = h.debug_factor_bench(nb, 8, 300, reps=1)
    lib.tmpc_debug_cycle_prof(out.ctypes.data_as(C.POINTER(C.c_double)))
    tot = out[:6].sum()
    names = ['issue next-slab loads', 'C prefetch issue', 'ds_read + MFMA', 'wait loads + LDS store', 'epilogue stores', 'barrier']
    print(f"nb {nb}: kernel {ms} ms; block 0 thread 0 cycles in wg_gemm_nt (both variants' launches, 2 reps+warm): total {tot:.3e}")
    for n_, v in zip(names, out[:6]):
        print(f"    {n_:28s} {v:.3e}  {100*v/tot:5.1f} %")
