"""Matrix helpers on the hot path (reference: tunempc/mtools.py:33-41).  numpy only -- the reference module
also imports casadi for `tracking_cost`, which is not on this path."""
import numpy as np


def symmetrize(S):
    """(S + S')/2   (mtools.py:33-36)"""
    return (S + S.T) / 2.0


def buildHessian(Q, R, N):
    """[[Q, N], [N', R]]   (mtools.py:38-41)"""
    return np.vstack((np.hstack((Q, N)), np.hstack((N.T, R))))
