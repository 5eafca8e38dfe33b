"""CPU restatement of the consumer arithmetic of the tuned matrices (TEST INFRASTRUCTURE ONLY -- imported by tests/
and nothing else; the product path is tunempc_amd/pmpc.py -> tmpc_tracking_reference_host).

Follows tunempc/pmpc.py:961-974 (per-stage update inside `Pmpc.step`; the same formula at set-up, :594-609):
    W    = Href / ts
    yref = vertcat(xref, uref) - inv(Href / ts) @ qref.T / ts
and the per-phase rotation of the periodic lists, pmpc.py:773-781.  Parity: pinned only by the closed form itself
(plain numpy linear algebra); the reference module needs casadi + acados and cannot be imported here."""
import numpy as np


def tracking_reference(Hc, q, wref, ts):
    """Hc [..., n, n], q/wref [..., n] -> W [..., n, n], yref [..., n]  (pmpc.py:961-974)."""
    Hc = np.asarray(Hc, dtype=np.float64); q = np.asarray(q, dtype=np.float64); wref = np.asarray(wref, dtype=np.float64)
    Hs = 0.5 * (Hc + np.swapaxes(Hc, -1, -2))
    W = Hs / ts
    yref = wref - np.linalg.solve(W, q[..., None])[..., 0] / ts
    return W, yref


def rotate(lst, N):
    """pmpc.py:773-781: out[k][j] = lst[(k + j) % Nref]."""
    Nref = len(lst)
    return [[lst[(k + j) % Nref] for j in range(N)] for k in range(Nref)]
