import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tunempc_amd._lib as L
L.library_path = lambda: os.path.join(ROOT, 'tunempc_amd', 'lib', 'libtunempc_hip_prof.so')
from tunempc_amd._lib import HipConvexifier
from tunempc_amd import synthetic
A, B, H = synthetic.gen_batch(100000, 64, 64, 24, 8)
h = HipConvexifier(64, 24, 8)
lib = h.lib
lib.tmpc_debug_cycle_prof.argtypes = [C.POINTER(C.c_double)]
out = np.zeros(64)
lib.tmpc_debug_cycle_prof(out.ctypes.data_as(C.POINTER(C.c_double)))
res = h.convexify_batch(A, B, H)
lib.tmpc_debug_cycle_prof(out.ctypes.data_as(C.POINTER(C.c_double)))
load, tot = out[40], out[43]
ms = out[42] - out[41]
n = res['iters'][0] * 2
print('k_eigmin block 0: per call load %.0f cycles, tridiag_min_eig %.0f (multisection %.0f, Householder %.0f)' % (load / n, tot / n, ms / n, (tot - ms) / n))
