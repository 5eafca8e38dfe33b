"""Caller-side drop-in for `Tuner.convexify` (reference: tunempc/tuner.py:134-160).

The reference's `Tuner` is CasADi/IPOPT host code that stays as it is (BASELINE.json north_star); the
only lines that touch the hot path are the slicing of Q, R, N out of S['H'] (:145-147), the call
(:157) and the assembly of S['Hc'] (:158).  `tuner_convexify` is exactly those lines over the
sensitivities dict S produced by `Pocp.get_sensitivities` (pocp.py:261-362), so that a maintainer can
replace the body of `Tuner.convexify` by one call."""
import numpy as np

from . import convexifier
from .logger import Logger


def _full(m):
    return m.full() if hasattr(m, 'full') else np.asarray(m, dtype=np.float64)


def tuner_convexify(S, nx, p, rho=1.0, force=False, solver='hip'):
    """Compute positive definite stage cost matrices for a tracking NMPC scheme that is locally first-order
    equivalent to economic MPC (tuner.py:134-160).  S: dict with 'A','B','H' (lists of p matrices) and
    optionally 'C_As', 'G'.  Stores and returns S['Hc'] (list of p ndarrays)."""
    H = [_full(S['H'][i]) for i in range(p)]
    Q = [H[i][:nx, :nx] for i in range(p)]          # tuner.py:145
    R = [H[i][nx:, nx:] for i in range(p)]          # tuner.py:146
    N = [H[i][:nx, nx:] for i in range(p)]          # tuner.py:147
    opts = {'rho': rho, 'solver': solver, 'force': force}
    Logger.logger.info(60 * '=')
    Logger.logger.info(15 * ' ' + 'Convexify Lagrangian Hessians...')
    Logger.logger.info(60 * '=')
    Logger.logger.info('')
    A = [_full(a) for a in S['A']]
    B = [_full(b) for b in S['B']]
    dHc, _, _, _ = convexifier.convexify(A, B, Q, R, N, C=S.get('C_As'), G=S.get('G'), opts=opts)
    if isinstance(dHc, np.ndarray):
        # already-convex early exit: the reference returns bare zero arrays (convexifier.py:85) and then fails on
        # dHc[i] in tuner.py:158; here the zero supplement is applied per stage instead (documented deviation).
        dHc = [dHc for _ in range(p)]
    S['Hc'] = [H[i] + dHc[i] for i in range(p)]      # tuner.py:158
    return S['Hc']
