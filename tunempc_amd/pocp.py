"""Producer side of the hot path: the post-processing of the NLP sensitivities that feeds `convexify()` (reference:
tunempc/pocp.py:261-362, `Pocp.get_sensitivities`).

The CasADi evaluations (Jacobians of the dynamics and constraints, Lagrangian Hessian) stay host code in the reference;
mirrored here is the array arithmetic between their `.full()` results and the convexifier (SURVEY.md section 8f, row
N2), plus the packing of one or many sensitivity dicts into the contiguous batched layout of the C ABI
(include/tunempc_hip.h):
  * `active_set`      -- rows of C_k with |mu_k,i| > mu_tresh, `None` when a stage has none   (pocp.py:322-340)
  * `stage_hessians`  -- diagonal blocks of the Lagrangian Hessian                              (pocp.py:350-355)
  * `cost_gradient`   -- q_k = -mu_k' C_k, zeros without path constraints                        (pocp.py:357-361)
  * `pack_batch`      -- list of S dicts -> A [B,p,nx,nx], B [B,p,nx,mb], H [B,p,n,n], q [B,p,n], ragged C_As padded to
                         [B,p,nc_max,n] with counts nc [B,p], G [B,p,ng,n]
These are O(p n^2) index/GEMV operations on host data (the reference does them in Python as well); the O(p n^6)
work behind them is the HIP path."""
import numpy as np

MU_TRESH = 1e-15            # pocp.py:73: threshold for active constraint detection


def _full(m):
    return m.full() if hasattr(m, 'full') else np.asarray(m, dtype=np.float64)


def active_set(C, mu, mu_tresh=MU_TRESH):
    """pocp.py:322-340.  C: list of p (nh x n) constraint Jacobians, mu: list of p multiplier vectors (nh).
    Returns (C_As, indeces_As): C_As[k] = (nc_k x n) rows with |mu_k,i| > mu_tresh or None, and their indices."""
    if len(C) != len(mu):
        raise AssertionError('Input data lists should have same length!')
    C_As, idx_As = [], []
    for k in range(len(C)):
        Ck = _full(C[k]); mk = np.reshape(_full(mu[k]), (-1,))
        if Ck.shape[0] != mk.shape[0]:
            raise AssertionError('one multiplier per constraint row expected at stage %d' % k)
        index_active = [i for i in range(mk.shape[0]) if np.abs(mk[i]) > mu_tresh]
        C_As.append(Ck[index_active, :].copy() if index_active else None)
        idx_As.append(index_active)
    return C_As, idx_As


def stage_hessians(H, M, N):
    """pocp.py:350-355: S['H'] = [H[i*M:(i+1)*M, i*M:(i+1)*M] for i in range(N)], M = nx + nu (+ ns)."""
    H = _full(H)
    if H.shape[0] < N * M or H.shape[1] < N * M:
        raise AssertionError('Hessian too small for %d stages of size %d' % (N, M))
    return [H[i * M:(i + 1) * M, i * M:(i + 1) * M].copy() for i in range(N)]


def cost_gradient(mu, C, N=None, n=None):
    """pocp.py:357-361: q_k = -mu_k' C_k (1 x n); zeros((1, n)) for every stage when there are no path constraints."""
    if C is None:
        return [np.zeros((1, n)) for _ in range(N)]
    return [-(np.reshape(_full(mu[k]), (1, -1)) @ _full(C[k])) for k in range(len(C))]


def pack_batch(S_list, nx):
    """Pack sensitivity dicts (keys 'A','B','H', optional 'q','C_As','G'; lists of p matrices as produced by
    get_sensitivities) of B tuning problems with common shapes into the contiguous arrays of the batched C ABI.
    Returns a dict with A, B, H, q, and -- when present -- C (zero-padded to nc_max rows), nc (int32 counts) and G."""
    if not S_list:
        raise AssertionError('at least one problem expected')
    p = len(S_list[0]['A'])
    out = {}
    for key in ('A', 'B', 'H'):
        arr = []
        for S in S_list:
            if len(S[key]) != p:
                raise AssertionError('Input data lists should have same length!')
            mats = [_full(m) for m in S[key]]
            if any(m.shape != mats[0].shape for m in mats):
                raise AssertionError('Data matrices should have same size along trajectory.')
            arr.append(np.stack(mats))
        out[key] = np.ascontiguousarray(np.stack(arr), dtype=np.float64)
    nb, n = len(S_list), out['H'].shape[-1]
    if out['A'].shape[-2:] != (nx, nx) or out['B'].shape[-2] != nx or out['B'].shape[-1] != n - nx:
        raise AssertionError('A (nx x nx), B (nx x (n - nx)), H (n x n) expected')
    out['q'] = np.zeros((nb, p, n))
    for b, S in enumerate(S_list):
        if S.get('q') is not None:
            out['q'][b] = np.stack([np.reshape(_full(v), (n,)) for v in S['q']])
    if any(S.get('C_As') is not None for S in S_list):
        nc = np.zeros((nb, p), dtype=np.int32)
        for b, S in enumerate(S_list):
            if S.get('C_As') is not None:
                nc[b] = [0 if c is None else _full(c).shape[0] for c in S['C_As']]
        C = np.zeros((nb, p, max(int(nc.max()), 1), n))
        for b, S in enumerate(S_list):
            if S.get('C_As') is not None:
                for k, c in enumerate(S['C_As']):
                    if c is not None:
                        C[b, k, :nc[b, k]] = _full(c)
        out['C'], out['nc'] = C, nc
    if any(S.get('G') is not None for S in S_list):
        if not all(S.get('G') is not None for S in S_list):
            raise AssertionError('Input arguments should be of same type!')
        out['G'] = np.ascontiguousarray(np.stack([np.stack([_full(g) for g in S['G']]) for S in S_list]), dtype=np.float64)
    return out


def pack_batch_device(handle, C=None, mu=None, Hbig=None, mu_tresh=MU_TRESH, ncmax=None, nb=None):
    """The same post-processing on the GPU for a whole batch (tmpc_pack_sensitivities_host through `handle`, a HipConvexifier of the
    problem shape): dense inputs C [nb,p,nh,n], mu [nb,p,nh], Hbig [nb,p*n,p*n] as the CasADi `.full()` calls deliver them, outputs in
    the batched layout of the C ABI (C_As zero-padded with counts `nc`, q, H).  Bit-identical to `active_set` / `cost_gradient` /
    `stage_hessians` (tests/test_gpu_parity.py::test_producer_packing_on_device)."""
    return handle.pack_sensitivities(C=C, mu=mu, Hbig=Hbig, thr=mu_tresh, ncmax=ncmax, nb=nb)
