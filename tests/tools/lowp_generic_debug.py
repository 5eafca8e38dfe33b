"""Checker (test infrastructure, run on the GPU box): the generic and the tuned stage kernels on one shape with the single-precision updates on / off,
against the numpy oracle -- how the retry of a frozen pivot moves the point (round 6; tests/test_gpu_parity.py runs the comparison with lowp_switch = 0)."""
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import numpy as np, convexify_oracle as co
from tunempc_amd._lib import HipConvexifier
def rel(a,b): return np.linalg.norm(a-b)/np.linalg.norm(b)
seed,nb,p,nx,mb = 70,3,16,12,4
A,B,H = co.gen_batch(seed,nb,p,nx,mb)
res={}
for name,flags,sw in (('tuned lowp',0,None),('generic lowp',64,None),('tuned fp64',0,0.0),('generic fp64',64,0.0)):
    h=HipConvexifier(p,nx,mb,chunk=nb,flags=flags)
    if sw is not None: h.set_tuning(lowp_switch=sw)
    res[name]=h.convexify_batch(A,B,H); h.close()
    print(name,'iters',res[name]['iters'],'status',res[name]['status'])
import cpu_ipm
ref=cpu_ipm.convexify_batch(A,B,H,threads=3)
for a in res:
    print(a,'vs cpu_ipm',[f"{rel(res[a]['Hc'][b],ref['Hc'][b]):.1e}" for b in range(nb)])
for a,b_ in (('tuned lowp','generic lowp'),('tuned fp64','generic fp64'),('tuned lowp','tuned fp64'),('generic lowp','generic fp64')):
    print(a,'vs',b_,[f"{rel(res[a]['Hc'][b],res[b_]['Hc'][b]):.1e}" for b in range(nb)])
