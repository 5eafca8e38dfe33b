"""Matrix helpers on the hot path (behaviour of reference tunempc/mtools.py:33-41).  numpy only -- the reference module
also imports casadi for `tracking_cost`, which is not on this path."""
import numpy as np


def symmetrize(S):
    """Symmetric part of a square matrix (mtools.py:33-36)."""
    S = np.asarray(S)
    return 0.5 * (S + S.T)


def buildHessian(Q, R, N):
    """Stage Hessian [[Q, N], [N', R]] from its blocks (mtools.py:38-41)."""
    Q = np.asarray(Q); R = np.asarray(R); N = np.asarray(N)
    return np.block([[Q, N], [N.T, R]])
