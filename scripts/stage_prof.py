"""In-kernel cycle split of k_stage_pre (wave of block 0; needs the -DTMPC_CYCLE_PROF build libtunempc_hip_prof.so)."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tunempc_amd._lib as L
L.library_path = lambda: os.path.join(ROOT, 'tunempc_amd', 'lib', 'libtunempc_hip_prof.so')
from tunempc_amd._lib import HipConvexifier
from tunempc_amd import synthetic
p, nx, mb = 64, 24, 8
A, B, H = synthetic.gen_batch(100000, 64, p, nx, mb)
h = HipConvexifier(p, nx, mb)
lib = h.lib
lib.tmpc_debug_cycle_prof.argtypes = [C.POINTER(C.c_double)]
out = np.zeros(64)
lib.tmpc_debug_cycle_prof(out.ctypes.data_as(C.POINTER(C.c_double)))
res = h.convexify_batch(np.tile(A, (8, 1, 1, 1)), np.tile(B, (8, 1, 1, 1)), np.tile(H, (8, 1, 1, 1)))
lib.tmpc_debug_cycle_prof(out.ctypes.data_as(C.POINTER(C.c_double)))
names = ['global -> LDS loads', 'build_M', 'residuals / copies', 'chol_lower (x4)', 'tri_inv_lower (x4)', 'LDS -> global stores', "mm Li'Li (x2)", 'Kronecker factors, Phi, Psi (mm)']
o8 = out[32:40]          # the factorisation marks leave class 3 selected: slots 3*8 + 8 .. 15
tot = o8.sum()
print('k_stage_pre, stage 0 of problem 0, %d launches: %.3e cycles per launch' % (res['iters'][0], tot / res['iters'][0]))
for n_, v in zip(names, o8):
    print(f'    {n_:36s} {v:.3e}  {100 * v / tot:5.1f} %')
