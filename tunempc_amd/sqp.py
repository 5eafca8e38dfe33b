"""Consumer-side row N4: the Hessian regularisation of the reference's SQP method
(tunempc/sqp_method.py:327-403, `Sqp.__regularize_hessian`) with its eigenvalue work on the GPU.

    regularize_hessian(H, jacg_active, regularization='reduced', tol=...)
        'reduced': Z = null_space(jacg_active), Hr = Z' H Z; eigenvalues of Hr below tol are lifted to tol,
                   H += Z (evec diag(evmod - eva) evec^-1) Z', H = (H + H')/2                      (sqp_method.py:337-372)
        'full'   : the same lift on the eigenvalues of H itself                                   (sqp_method.py:375-394)
        otherwise: H unchanged, reg = 0                                                           (sqp_method.py:396-397)
    returns (H, reg) with reg = max(evmod - eva) (the reference stores it in self.__reg / self._reg).

The null space (an SVD of the active-constraint Jacobian, scipy.linalg.null_space as in the reference) and the two thin
products with Z stay host numpy; the symmetric eigen-decomposition, clip and reconstruction of the r x r (or full) matrix are
tmpc_eig_clip_host (include/tunempc_hip.h, tunempc_amd/csrc/tmpc_eig.h).  The reference calls the general `eig` on a matrix that is
symmetric by construction; its eigenvalues are real and evec^-1 = evec', which is what the symmetric solver returns.

A decomposition that did not converge (TMPC_E_NOCONV: 40 Jacobi sweeps exhausted) raises `_lib.EigNotConverged` out of
regularize_hessian: a Hessian "regularised" with unconverged eigenvectors would silently hand the QP an indefinite matrix."""
import numpy as np

from tunempc_amd import _lib
from tunempc_amd.logger import Logger


def regularize_hessian(H, jacg_active=None, regularization='reduced', tol=1e-8):
    H = np.array(H, dtype=np.float64)
    reg = 0.0
    if regularization == 'reduced':
        from scipy.linalg import null_space
        Z = null_space(np.atleast_2d(np.asarray(jacg_active, dtype=np.float64)))
        if Z.shape[1] != 0:                                      # sqp_method.py:346-350: nothing to do for an empty reduced space
            Hr = Z.T @ H @ Z
            res = _lib.eig_clip(Hr, tol)
            if res['evals'].min() < tol:                         # sqp_method.py:348, 353
                reg = res['reg']
                H = H + Z @ (res['out'] - 0.5 * (Hr + Hr.T)) @ Z.T   # Z dHr Z'
                H = (H + H.T) / 2.0
                chk = _lib.eig_clip(Z.T @ H @ Z, tol)['evals']   # sqp_method.py:369-373
                for e in chk:
                    if e < tol / 1e2:
                        Logger.logger.warning('Regularization of reduced Hessian failed. Eigenvalue: {}'.format(e))
    elif regularization == 'full':
        res = _lib.eig_clip(H, tol)
        reg = res['reg']
        H = (res['out'] + res['out'].T) / 2.0
    return H, reg
