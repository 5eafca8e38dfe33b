"""ctypes loader of the C++/OpenMP CPU baseline (oracle/cpu_ipm/cpu_ipm.cpp)  --  TEST / MEASUREMENT INFRASTRUCTURE ONLY.
Only tests/ and bench.py's cpu_baseline leg import this; the product (tunempc_amd) never does."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'libcpu_ipm.so')
_lib = None


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in ('cpu_ipm.cpp', 'cpu_ipm_con.h')]
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(['make', '-C', HERE, '-s'] + (['-B'] if force else []))
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        lib = C.CDLL(LIB)
        dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int32)
        lib.cpu_ipm_convexify_batch.restype = C.c_int
        lib.cpu_ipm_convexify_batch.argtypes = [C.c_int] * 4 + [dp, dp, dp, C.c_double, C.c_int, dp, dp, ip, ip]
        lib.cpu_ipm_convexify_batch2.restype = C.c_int
        lib.cpu_ipm_convexify_batch2.argtypes = [C.c_int] * 4 + [dp, dp, dp, C.c_double, C.c_int, C.c_int, dp, dp, ip, ip, dp]
        lib.cpu_ipm_convexify_con_batch.restype = C.c_int
        lib.cpu_ipm_convexify_con_batch.argtypes = [C.c_int] * 6 + [dp, dp, dp, dp, ip, C.c_double, C.c_int, C.c_double, C.c_int, dp, dp, dp, dp, dp, dp, ip, ip]
        lib.cpu_ipm_convexify_con_batch2.restype = C.c_int
        lib.cpu_ipm_convexify_con_batch2.argtypes = [C.c_int] * 6 + [dp, dp, dp, dp, ip, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, ip, ip, dp]
        lib.cpu_ipm_max_threads.restype = C.c_int
        _lib = lib
    return _lib


def convexify_batch(A, B, H, tol=0.0, threads=1, tight=False):
    """A [nb,p,nx,nx], B [nb,p,nx,mb], H [nb,p,n,n] -> dict(Hc, kappa, status, iters); `threads` OpenMP threads, one problem each.
    tight: the tight-accuracy mode (double-double block linear algebra + dual-Newton polish; adds mu_t, dd_iters, polish_steps, stepn)."""
    lib = load()
    A = np.ascontiguousarray(A, dtype=np.float64); B = np.ascontiguousarray(B, dtype=np.float64); H = np.ascontiguousarray(H, dtype=np.float64)
    nb, p, nx, _ = A.shape
    mb = B.shape[3]
    Hc = np.empty_like(H); kappa = np.empty(nb); status = np.empty(nb, np.int32); iters = np.empty(nb, np.int32)
    d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    i = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    if tight:
        info = np.zeros((nb, 4))
        rc = lib.cpu_ipm_convexify_batch2(nb, p, nx, mb, d(A), d(B), d(H), float(tol), int(threads), 1, d(Hc), d(kappa), i(status), i(iters), d(info))
        if rc != 0:
            raise RuntimeError('cpu_ipm_convexify_batch2 failed: %d' % rc)
        return dict(Hc=Hc, kappa=kappa, status=status, iters=iters, mu_t=info[:, 0], dd_iters=info[:, 1].astype(int),
                    polish_steps=info[:, 2].astype(int), stepn=info[:, 3])
    rc = lib.cpu_ipm_convexify_batch(nb, p, nx, mb, d(A), d(B), d(H), float(tol), int(threads), d(Hc), d(kappa), i(status), i(iters))
    if rc != 0:
        raise RuntimeError('cpu_ipm_convexify_batch failed: %d' % rc)
    return dict(Hc=Hc, kappa=kappa, status=status, iters=iters)


def convexify_con_batch(A, B, H, J=None, ng=0, ncnt=None, rho=None, cost_free=False, force=False, tol=0.0, threads=1, tight=False):
    """The models with rows (cpu_ipm_con.h; restatement of convexify_oracle.sdp_step1(G=, C=, rho=, force=, cost_free=)).
    J [nb,p,ng+nc,n]: the `ng` rows of G_k first, then the rows of C_k padded to nc; ncnt [nb,p] active rows of C_k (None: all nc).
    The rows of C_k take part when rho is given (Step 2, `constr=True`) or cost_free (the beta-only objective); force (with rho): Step 3.
    tight: the tight-accuracy mode (Steps 1 / 2; with force: NotImplementedError -- the numpy oracle has that mode for Step 3); adds mu_t, dd_iters, polish_steps, stepn.
    Returns dict(Hc, P, FgF [nb,p,ng+nc] (un-scaled multipliers, zero padding), T, kappa, objective, status, iters)."""
    lib = load()
    A = np.ascontiguousarray(A, dtype=np.float64); B = np.ascontiguousarray(B, dtype=np.float64); H = np.ascontiguousarray(H, dtype=np.float64)
    nb, p, nx, _ = A.shape
    mb = B.shape[3]; n = nx + mb
    if J is None:
        J = np.zeros((nb, p, 0, n))
    J = np.ascontiguousarray(J, dtype=np.float64)
    nJ = J.shape[2]; nc = nJ - ng
    constr = nc > 0 and (rho is not None or cost_free)
    flags = (1 if constr else 0) | (2 if cost_free else 0) | (4 if (force and rho is not None) else 0)
    if ncnt is not None:
        ncnt = np.ascontiguousarray(ncnt, dtype=np.int32)
    Hc = np.empty_like(H); P = np.empty((nb, p, nx, nx)); FgF = np.zeros((nb, p, nJ)); T = np.zeros((nb, p, n, n))
    kappa = np.empty(nb); obj = np.empty(nb); status = np.empty(nb, np.int32); iters = np.empty(nb, np.int32)
    d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    i = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    if tight:
        if flags & 4:
            raise NotImplementedError('cpu_ipm: Step 3 in the tight mode exists in the numpy oracle only')
        info = np.zeros((nb, 4))
        rc = lib.cpu_ipm_convexify_con_batch2(nb, p, nx, mb, int(ng), int(nc), d(A), d(B), d(H), d(J) if nJ else None, i(ncnt) if ncnt is not None else None,
                                              float(rho if rho is not None else 0.0), flags, float(tol), int(threads), 1, d(Hc), d(P), d(FgF) if nJ else None, d(T),
                                              d(kappa), d(obj), i(status), i(iters), d(info))
        if rc != 0:
            raise RuntimeError('cpu_ipm_convexify_con_batch2 failed: %d' % rc)
        return dict(Hc=Hc, P=P, FgF=FgF, T=T, kappa=kappa, objective=obj, status=status, iters=iters, mu_t=info[:, 0], dd_iters=info[:, 1].astype(int),
                    polish_steps=info[:, 2].astype(int), stepn=info[:, 3])
    rc = lib.cpu_ipm_convexify_con_batch(nb, p, nx, mb, int(ng), int(nc), d(A), d(B), d(H), d(J) if nJ else None, i(ncnt) if ncnt is not None else None,
                                         float(rho if rho is not None else 0.0), flags, float(tol), int(threads), d(Hc), d(P), d(FgF) if nJ else None, d(T),
                                         d(kappa), d(obj), i(status), i(iters))
    if rc != 0:
        raise RuntimeError('cpu_ipm_convexify_con_batch failed: %d' % rc)
    return dict(Hc=Hc, P=P, FgF=FgF, T=T, kappa=kappa, objective=obj, status=status, iters=iters)


def max_threads():
    return int(load().cpu_ipm_max_threads())
