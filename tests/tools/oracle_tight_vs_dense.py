"""CPU only: the numpy oracle's tight mode (Step 1 with G, Step 2, beta-only objective, Step 3) against the dense solver of oracle/reference_sdp.py on random small problems:
objective values to the dense solver's tolerance.  python tests/tools/oracle_tight_vs_dense.py  (round 5: 33 members compared, all Optimal, worst 5.0e-8 -- a Step 2 member
with rho = 1e-3 whose dense objective moves 0.00214349356821 / ...243776 / ...233568 at tol 1e-8 / 1e-9 / 1e-10 towards the oracle's 0.00214349233162 (2^-37) and ...233142 (2^-41):
the dense solver's tolerance, not the oracle, sets the distance)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'oracle'))
import numpy as np, convexify_oracle as co, reference_sdp as rs
rng=np.random.default_rng(12345)
bad=[]; worst=0; n_ok=0; stat={}
t0=time.time()
for case in range(60):
    p=int(rng.integers(1,5)); nx=int(rng.integers(2,5)); mb=int(rng.integers(1,3)); n=nx+mb
    model=['G','step2','beta','step3'][int(rng.integers(0,4))]
    ng=int(rng.integers(0,3)); nc=int(rng.integers(1,3))
    if model=='G': ng=max(ng,1)
    A,B,H=co.gen_batch(int(rng.integers(1,1<<30)),1,p,nx,mb); A,B,H=A[0],B[0],H[0]
    if np.linalg.eigvalsh(H)[:,0].min()>0: continue
    G=rng.standard_normal((p,ng,n)) if ng else None
    ncs=rng.integers(0,nc+1,size=p)
    C=[rng.standard_normal((ncs[k],n)) if ncs[k] else None for k in range(p)]
    rho=float(rng.choice([1e-3,1e-2,1e-1]))
    if model=='G': kw=dict(G=G); dk=dict(G=None if G is None else list(G),constr=False)
    elif model=='step2': kw=dict(G=G,C=C,rho=rho); dk=dict(G=None if G is None else list(G),C=C,rho=rho,constr=True)
    elif model=='beta': kw=dict(G=G,C=C,rho=0.0); dk=dict(G=None if G is None else list(G),C=C,rho=0.0,constr=True)
    else: kw=dict(G=G,rho=rho,force=True); dk=dict(G=None if G is None else list(G),rho=rho,constr=False,force=True)
    try:
        r=co.sdp_step1(A,B,H,dict(tol=2.0**-37,tight=True),**kw)
    except Exception as e:
        bad.append((case,model,p,nx,mb,ng,'EXC',repr(e)[:80])); continue
    stat[r['ipm_status']]=stat.get(r['ipm_status'],0)+1
    if r['ipm_status']!='optimal' or r['mu_target']>2.0**-36*max(1,r['kappa']):
        bad.append((case,model,p,nx,mb,ng,r['ipm_status'],r['mu_target'])); continue
    Q=[H[k][:nx,:nx] for k in range(p)]; R=[H[k][nx:,nx:] for k in range(p)]; N=[H[k][:nx,nx:] for k in range(p)]
    d=rs.solve_step(list(A),list(B),Q,R,N,tol=1e-9,**dk)
    if d['solver_status']!='optimal': continue
    obj=r.get('objective',r['beta'])
    e=abs(obj-d['objective'])/obj
    worst=max(worst,e); n_ok+=1
    if e>5e-8: bad.append((case,model,p,nx,mb,ng,'obj',e))
print('compared',n_ok,'worst',worst,'status',stat,'bad',bad,'%.0fs'%(time.time()-t0))
