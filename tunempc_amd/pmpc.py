"""Consumer side of the tuned matrices: how the reference's `Pmpc` uses Hc_k and q_k in a tracking MPC
(reference: tunempc/pmpc.py).

Only the arithmetic between `convexify()` and the acados solver is mirrored here -- the NLP/acados set-up of `Pmpc`
stays host code in the reference:
  * `rotate_tuning`     -- per-phase rotation of the periodic lists, pmpc.py:773-781;
  * `tracking_reference` -- W_k = Hc_k / ts and yref_k = wref_k - (Hc_k/ts)^-1 q_k / ts, pmpc.py:961-974 (first stage
    at set-up: :594-609), batched over all stages on the GPU (one Cholesky solve per stage, `k_tracking_ref`).
There is no CPU fallback: without the HIP library / a gfx950 device the calls raise."""
import numpy as np

from ._lib import HipConvexifier

_handles = {}


def _handle(n):
    # the tracking reference only needs n = nx + mb; nx/mb split and period are irrelevant for this entry
    if n not in _handles:
        _handles[n] = HipConvexifier(1, n, 0, chunk=1)
    return _handles[n]


def _full(m):
    return m.full() if hasattr(m, 'full') else np.asarray(m, dtype=np.float64)


def rotate_tuning(H, q, N):
    """pmpc.py:773-781: for every phase k of the Nref-periodic reference, the N stage matrices / gradients seen by the
    horizon starting at k:  Href[k][j] = H[(k + j) % Nref],  qref[k][j] = q[(k + j) % Nref]."""
    Nref = len(H)
    if len(q) != Nref:
        raise AssertionError('H and q must have the same period')
    Href = [[H[(k + j) % Nref] for j in range(N)] for k in range(Nref)]
    qref = [[q[(k + j) % Nref] for j in range(N)] for k in range(Nref)]
    return Href, qref


def tracking_reference(Hc, q, wref, ts):
    """pmpc.py:961-974 for a whole period at once.  Hc: list of Nref (n x n) tuned matrices, q: list of Nref gradient
    rows (1 x n or n), wref: list of Nref reference points vertcat(xref, uref) (n or n x 1), ts: sampling time.
    Returns (W, yref): lists of Nref arrays, W_k = Hc_k / ts (the acados 'W'), yref_k = wref_k - Hc_k^-1 q_k."""
    Hs = np.stack([_full(h) for h in Hc])
    n = Hs.shape[-1]
    qs = np.stack([np.reshape(_full(v), (n,)) for v in q])
    ws = np.stack([np.reshape(_full(v), (n,)) for v in wref])
    W, yref, info = _handle(n).tracking_reference(Hs, qs, ws, ts)
    if np.any(info != 0):
        raise ValueError('tracking_reference: stage matrices %s are not positive definite' % np.flatnonzero(info).tolist())
    return [W[k] for k in range(len(Hc))], [yref[k] for k in range(len(Hc))]
