"""CPU restatement (TEST INFRASTRUCTURE ONLY -- nothing under tunempc_amd/ may import this) of the Hessian regularisation of
the reference's SQP method, tunempc/sqp_method.py:327-403 (`Sqp.__regularize_hessian`), as plain numpy/scipy.
Parity unpinned: the reference holds no vector for this routine and its class needs CasADi (absent here); the restatement follows
the source line by line (the general `eig` of the reference replaced by `eigh`, identical for the symmetric matrices it is given)."""
import numpy as np
from scipy.linalg import null_space


def eig_clip(A, tol):
    """sqp_method.py:352-361 / :378-389: eigenvalues below tol -> tol.  Returns (A + evec diag(evmod - eva) evec', eva, max(evmod - eva))."""
    A = 0.5 * (np.asarray(A, dtype=np.float64) + np.asarray(A, dtype=np.float64).T)
    eva, evec = np.linalg.eigh(A)
    evmod = np.where(eva < tol, tol, eva)
    deva = evmod - eva
    return A + (evec * deva) @ evec.T, eva, float(deva.max())


def regularize_hessian(H, jacg_active=None, regularization='reduced', tol=1e-8):
    H = np.array(H, dtype=np.float64)
    reg = 0.0
    if regularization == 'reduced':                              # sqp_method.py:337-372
        Z = null_space(np.atleast_2d(np.asarray(jacg_active, dtype=np.float64)))
        if Z.shape[1] != 0:
            Hr = Z.T @ H @ Z
            out, eva, lift = eig_clip(Hr, tol)
            if eva.min() < tol:
                reg = lift
                H = H + Z @ (out - 0.5 * (Hr + Hr.T)) @ Z.T
                H = (H + H.T) / 2.0
    elif regularization == 'full':                               # sqp_method.py:375-394
        out, eva, reg = eig_clip(H, tol)
        H = (out + out.T) / 2.0
    return H, reg
