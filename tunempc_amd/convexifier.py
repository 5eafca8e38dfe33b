"""Convexification code -- MI355X-native drop-in for `tunempc/convexifier.py`.

Same entry point, argument meaning, return structure, log lines and exceptions as the reference
(`convexify`, convexifier.py:36-163); the SDP that the reference hands to PICOS -> CVXOPT/MOSEK
(`setUpModelPicos` :213-308, `solveSDP` :359-372) is solved by the HIP kernels behind
include/tunempc_hip.h.  `opts['solver']` selects the backend: 'hip' (default here; the reference's
'mosek' / 'cvxopt' strings are accepted and mapped to 'hip' because those solvers do not exist on this
stack).  There is no CPU path in this package.

Scope: Step 1 (eta_F = 0, eta_T = 0), with the equality-constraint multipliers Fg
(convexifier.py:249-255) when G is given (up to NG_MAX rows per stage), and Step 2 (eta_F = 1: multipliers F of the
active constraints C and the rho-norm terms, convexifier.py:116-131) when Step 1 is infeasible and C is given, and
Step 3 (opts['force']: the regularisation T with rho*||T_k||_F, convexifier.py:137-147, together with the multipliers of G / C
when they are given) when the earlier steps stay infeasible.
"""
import numpy as np

from . import mtools
from . import preprocessing
from .logger import Logger
from ._lib import HipConvexifier, STATUS_NAMES, load_library

_HANDLES = {}        # (p, nx, mb, ng) -> HipConvexifier; at most _MAX_HANDLES shapes stay resident (least recently used goes first)
_MAX_HANDLES = 4
_MAX_CHUNK = 512     # problems per workspace chunk the drop-in paths ever ask for (larger batches run chunk by chunk)
DEFAULT_TOL = 2.0 ** -25


NG_MAX = 31      # equality-constraint rows per stage the HIP path eliminates (tmpc_common.h: NGM)
N_TUNED, N_ROWS_MAX, N_MAX = 32, 64, 96          # stage-block sizes: tuned kernels / generic per-stage kernels for every model (Step 3 while its blocks fit) / for the plain model (round 5)
NC_MAX = 31      # active-constraint rows per stage in Step 2 (tmpc_common.h: NCM)


# Mixed precision of the drop-in entry points (round 6; include/tunempc_hip.h: TMPC_TUNE_LOWP_SWITCH).  None: the library default -- the Schur-complement updates of
# a problem's early main-phase iterations (barrier parameter above 1e-5 kappa) run on the fp32 matrix cores, everything that defines the returned point stays fp64.
# 0.0: fp64 throughout (the arithmetic of rounds 1-5, bit for bit).  A module attribute, not an `opts` key: the reference's opts (convexifier.py:36) know nothing of it.
LOWP_SWITCH = None
LOWP_SWITCH_DEFAULT = 1e-5      # TMPC_LOWP_SWITCH_DEFAULT


def _handle(p, nx, mb, ng=0, nc=0, nb=1, step3=False, plain=False, exact=False):
    """One cached handle per problem shape.  Its workspace is sized for the batch actually asked for (next power of two of nb,
    at most _MAX_CHUNK), not for the 60 %-of-free-HBM default of tmpc_create: a single-problem convexify() at nx=24, p=64
    pins 0.15 GB instead of 75 GB.  A handle with room for G / C rows also serves the calls without them, so Step 1 and Step 2
    of one convexify() share it; it is rebuilt only when a call needs more rows or a larger chunk.
    plain=True: a handle of its own without any room for rows; exact=True: a handle of its own with room for exactly the rows asked for.  The
    tight-accuracy mode runs on such handles only: a shared one that an earlier call of the same shape grew rows on could refuse the mode (its
    double-double vectors of the rows must fit the LDS) or not, depending on the cache history -- ADVICE r4."""
    if plain:
        ng = nc = 0
    if nx + mb > N_MAX:
        raise NotImplementedError('the HIP path handles stage blocks up to nx + nu = {} (got {})'.format(N_MAX, nx + mb))
    if nx + mb > N_ROWS_MAX and (ng or nc or step3) and not plain:
        raise NotImplementedError('stage blocks beyond nx + nu = {} are handled for the plain model only: no G / C rows, no Step 3 (got nx + nu = {}, ng = {}, nc = {}, '
                                  'force = {})'.format(N_ROWS_MAX, nx + mb, ng, nc, bool(step3)))
    if step3 and load_library().tmpc_workspace_bytes_step3_con(1, p, nx, mb, ng, nc) == 0:
        # 32 < n <= 64 runs on the generic per-stage kernels (csrc/tmpc_big.h) for every step; the blocks of Step 3 carry the n(n+1)/2 entries of T_k and must fit
        # the LDS image of the substitution kernels (nx(nx+1)/2 + n(n+1)/2 + 1 + rows <= 3168: n = 64 with nx = 40 fits -- 2901 --, n = 64 with nx = 48 does not)
        raise NotImplementedError('Step 3 (force=True) at nx = {}, nx + nu = {}: Schur blocks of nx(nx+1)/2 + n(n+1)/2 + 1 + multipliers exceed what the '
                                  'substitution kernels hold in LDS (3168)'.format(nx, nx + mb))
    key = (p, nx, mb, ng, bool(step3), bool(plain), nc if exact else -1)   # the ng rows of G are live in every call of a handle; the room for C rows is padded per stage by ncnt
    want = 1
    while want < min(max(int(nb), 1), _MAX_CHUNK):
        want *= 2
    h = _HANDLES.pop(key, None)
    if h is not None and (h.nc < nc or h.chunk < want):
        nc, want = max(nc, h.nc), max(want, h.chunk)
        h.close()
        h = None
    if h is None:
        while len(_HANDLES) >= _MAX_HANDLES:
            _HANDLES.pop(next(iter(_HANDLES))).close()
        h = HipConvexifier(p, nx, mb, chunk=want, ng=ng, nc=nc, step3=step3)
    _HANDLES[key] = h           # most recently used last
    h.set_tuning(lowp_switch=LOWP_SWITCH_DEFAULT if LOWP_SWITCH is None else float(LOWP_SWITCH))      # (every fetch: a cached handle must not keep an earlier setting)
    return h


def release_handles():
    """Free every cached workspace (device memory) of the drop-in entry points."""
    while _HANDLES:
        _HANDLES.popitem()[1].close()


def _rows_supported(nx, ng, nc):
    """tmpc_create_con: up to NG_MAX + NC_MAX stage-local multipliers ride in each block of the factorisation."""
    return ng <= NG_MAX and nc <= NC_MAX


def _to_array(m):
    """numpy / np.matrix / CasADi DM (via .full()) -> 2-D float64 ndarray."""
    if hasattr(m, 'full'):
        m = m.full()
    return np.atleast_2d(np.asarray(m, dtype=np.float64))


def convexify_batch(A, B, H, tol=None, handle=None, G=None, nc_hint=0, tight=None):
    """Batched Step 1.  A [nb,p,nx,nx], B [nb,p,nx,mb], H [nb,p,n,n] -> dict with
    Hc, dHc [nb,p,n,n], P [nb,p,nx,nx], alpha, beta, kappa [nb], status [nb] (0 Optimal, 1 Feasible,
    2 Infeasible; convexifier.py:442-451), iters [nb], info [nb,16].
    G [nb,p,ng,n] (optional): equality-constraint Jacobians; their multipliers Fg [nb,p,ng] (convexifier.py:249-255)
    join Step 1 and are returned as 'Fg'; dHc then includes G' diag(Fg) G (convexifier.py:196-197).
    tight: None / False: the default accuracy (mu_target = 2^-25 kappa); True or a tolerance: the tight-accuracy mode of the library (tmpc_set_tight:
    continuation to tight_tol * kappa, default 2^-37, in double-double arithmetic; up to nx = 51; with rows of G while their double-double vectors fit the LDS: rows * (2 n + 2 nx) <= 4040)."""
    A = np.asarray(A, dtype=np.float64); B = np.asarray(B, dtype=np.float64); H = np.asarray(H, dtype=np.float64)
    nb, p, nx, _ = A.shape
    mb = B.shape[3]
    ng = 0
    if G is not None:
        G = np.asarray(G, dtype=np.float64)
        ng = G.shape[2]
        if ng > NG_MAX:
            raise NotImplementedError('the HIP path handles up to {} equality-constraint rows per stage (got {})'.format(NG_MAX, ng))
    if handle is not None:
        h = handle
    elif tight and ng > 0:
        h = _handle(p, nx, mb, ng, 0, nb, exact=True)      # never the shared handle of the shape (room for C rows it may have grown counts against the LDS of the mode)
    elif tight:
        h = _handle(p, nx, mb, 0, 0, nb, plain=True)
    else:
        h = _handle(p, nx, mb, ng, nc_hint, nb)            # nc_hint: room for the C rows of a Step 2 that may follow (same handle)
    global _LAST_HANDLE
    _LAST_HANDLE = h
    if handle is None or tol is not None:
        h.set_options(tol=tol if tol is not None else DEFAULT_TOL)      # per call: a cached handle never keeps an earlier caller's tolerance
    if tight:
        h.set_tight(True, None if tight is True else float(tight))
    try:
        return h.convexify_eq_batch(A, B, H, G) if ng > 0 else h.convexify_batch(A, B, H)
    finally:
        if tight:
            h.set_tight(False)                    # a cached handle never keeps an earlier caller's mode


def convexify_step2_batch(A, B, H, C, ncnt, rho, G=None, tol=None, handle=None, tight=None):
    """Batched Step 2 model (convexifier.py:116-131, setUpModelPicos with constr=True): A, B, H as in convexify_batch;
    C [nb,p,nc,n] active-constraint Jacobians zero-padded to nc rows, ncnt [nb,p] rows present per stage (0: C_k is None);
    G [nb,p,ng,n] optional.  Returns the dict of convexify_batch plus 'F' [nb,p,nc] (zeros in the padding) and, with G, 'Fg'.
    rho = 0: the beta-only objective (cost-free multipliers; see `convexify`, opts['objective']).
    tight: as in convexify_batch (round 5: the mode covers this model, either objective)."""
    A = np.asarray(A, dtype=np.float64); B = np.asarray(B, dtype=np.float64); H = np.asarray(H, dtype=np.float64)
    C = np.asarray(C, dtype=np.float64); ncnt = np.asarray(ncnt, dtype=np.int32)
    nb, p, nx, _ = A.shape
    mb = B.shape[3]
    nc = C.shape[2]
    ng = 0 if G is None else np.shape(G)[2]
    if nc < 1 or not _rows_supported(nx, ng, nc):
        raise NotImplementedError('the HIP path handles up to {} equality- and 1..{} active-constraint rows per stage '
                                  '(got ng={}, nc={})'.format(NG_MAX, NC_MAX, ng, nc))
    if ncnt.shape != (nb, p) or (ncnt < 0).any() or (ncnt > nc).any():
        raise ValueError('ncnt must be an int array [nb, p] with 0 <= ncnt <= C.shape[2] = {}'.format(nc))
    h = handle or (_handle(p, nx, mb, ng, nc, nb, exact=True) if tight else _handle(p, nx, mb, ng, nc, nb))
    if handle is None or tol is not None:
        h.set_options(tol=tol if tol is not None else DEFAULT_TOL)
    # the handle may have more room than this call needs (it is shared between the steps): zero rows are padding
    J = np.zeros((nb, p, h.ng + h.nc, nx + mb))
    if ng:
        J[:, :, :ng] = np.asarray(G, dtype=np.float64)
    J[:, :, h.ng:h.ng + nc] = C
    global _LAST_HANDLE
    _LAST_HANDLE = h
    if tight:
        h.set_tight(True, None if tight is True else float(tight))
    try:
        out = h.convexify_step2_batch(A, B, H, J, ncnt, rho)
    finally:
        if tight:
            h.set_tight(False)
    FgF = out.pop('FgF')
    out['F'] = FgF[:, :, h.ng:h.ng + nc]
    if ng:
        out['Fg'] = FgF[:, :, :ng]
    return out


def convexify_step3_batch(A, B, H, rho, tol=None, handle=None, G=None, C=None, ncnt=None):
    """Batched Step 3 model (convexifier.py:137-147, setUpModelPicos with force=True): A, B, H as in convexify_batch.
    Returns the dict of convexify_batch plus 'T' [nb,p,n,n] (every entry > 0); dHc includes T (convexifier.py:202-203).
    With G [nb,p,ng,n] and / or C [nb,p,nc,n] + ncnt [nb,p] (as in convexify_step2_batch) their multipliers join the same solve
    (convexifier.py:144 passes constr = constraint_contribution): 'Fg' / 'F' are returned as well and dHc includes their terms."""
    A = np.asarray(A, dtype=np.float64); B = np.asarray(B, dtype=np.float64); H = np.asarray(H, dtype=np.float64)
    nb, p, nx, _ = A.shape
    mb = B.shape[3]
    ng = 0 if G is None else np.shape(G)[2]
    nc = 0 if C is None else np.shape(C)[2]
    global _LAST_HANDLE
    if ng == 0 and nc == 0:
        h = handle or _handle(p, nx, mb, 0, 0, nb, step3=True)
        if handle is None or tol is not None:
            h.set_options(tol=tol if tol is not None else DEFAULT_TOL)
        _LAST_HANDLE = h
        return h.convexify_step3_batch(A, B, H, rho)
    if not _rows_supported(nx, ng, nc):
        raise NotImplementedError('the HIP path handles up to {} equality- and {} active-constraint rows per stage (got ng={}, nc={})'.format(NG_MAX, NC_MAX, ng, nc))
    if nc:
        if ncnt is None:
            raise ValueError('convexify_step3_batch: C needs ncnt [nb, p], the rows of C_k present per stage (0 where C_k is None)')
        ncnt = np.asarray(ncnt, dtype=np.int32)
        if ncnt.shape != (nb, p) or (ncnt < 0).any() or (ncnt > nc).any():
            raise ValueError('ncnt must be an int array [nb, p] with 0 <= ncnt <= C.shape[2] = {}'.format(nc))
    h = handle or _handle(p, nx, mb, ng, nc, nb, step3=True)
    if handle is None or tol is not None:
        h.set_options(tol=tol if tol is not None else DEFAULT_TOL)
    _LAST_HANDLE = h
    if nc:
        J = np.zeros((nb, p, h.ng + h.nc, nx + mb))
        if ng:
            J[:, :, :ng] = np.asarray(G, dtype=np.float64)
        J[:, :, h.ng:h.ng + nc] = np.asarray(C, dtype=np.float64)
        out = h.convexify_step3_con_batch(A, B, H, J, ncnt, rho)
    else:
        out = h.convexify_step3_con_batch(A, B, H, np.asarray(G, dtype=np.float64), None, rho)
    FgF = out.pop('FgF')
    if nc:
        out['F'] = FgF[:, :, h.ng:h.ng + nc]
    if ng:
        out['Fg'] = FgF[:, :, :ng]
    return out


def convexify_steps_batch(A, B, H, G=None, C=None, ncnt=None, rho=1e-3, tol=None, tight=None):
    """The step logic of convexify() (convexifier.py:98-131) for a whole batch: Step 1 for every problem (with the equality-
    constraint multipliers when G is given), then the Step 2 model for the members that came back Infeasible, when C is given.
    Inputs as in convexify_batch / convexify_step2_batch.  Returns the dict of convexify_batch with the Step 2 results merged
    in, plus 'step' [nb] (0: already convex, 1, 2), 'F' [nb,p,nc] (zeros where Step 1 sufficed) and, with G, 'Fg'.
    Members that are still Infeasible keep status 2 (convexify() raises for them; a batch does not abort).
    tight: as in convexify_batch, for both steps (round 5)."""
    A = np.asarray(A, dtype=np.float64); B = np.asarray(B, dtype=np.float64); H = np.asarray(H, dtype=np.float64)
    out = convexify_batch(A, B, H, tol=tol, G=G, tight=tight)
    nb, p = A.shape[:2]
    out['step'] = np.where(out['info'][:, 13] != 0.0, 0, 1).astype(np.int32)
    if C is None:
        return out
    if ncnt is None:
        raise ValueError('convexify_steps_batch: C needs ncnt [nb, p], the rows of C_k present per stage (0 where C_k is None)')
    C = np.asarray(C, dtype=np.float64); ncnt = np.asarray(ncnt, dtype=np.int32)
    out['F'] = np.zeros((nb, p, C.shape[2]))
    redo = np.where(out['status'] == 2)[0]
    if redo.size:
        r2 = convexify_step2_batch(A[redo], B[redo], H[redo], C[redo], ncnt[redo], rho, G=None if G is None else np.asarray(G)[redo], tol=tol, tight=tight)
        for key, val in r2.items():
            out[key][redo] = val
        out['step'][redo] = 2
    return out


_LAST_HANDLE = None      # handle of the most recent batch call (its iteration trace is the solver log of convexify())


_IPM_STATUS = {0: 'optimal', 1: 'optimal_inaccurate', 2: 'not converged', 3: 'optimal', 4: 'optimal'}     # info[10]; plays the role of M.status (convexifier.py:365, :443); 3: TMPC_FLAG_FAST_EXIT stopped after the first full centering step, 4: tight mode fell back to the default point


def _log_solution(res):
    """The log lines of solveSDP (convexifier.py:365-370) and check_convergence (:441-453) for one solved problem."""
    status = STATUS_NAMES[int(res['status'][0])]
    ipm = _IPM_STATUS.get(int(res['info'][0, 10]), 'unknown')
    # solver verbosity (convexifier.py:87-91: opts['verbose'] = 1 below INFO level): the iteration log of the device IPM
    if Logger.logger.getEffectiveLevel() < 20 and _LAST_HANDLE is not None:
        Logger.logger.debug(' it  phase        mu       kappa       pinf       dinf   step_p   step_d  rel.step')
        for r in _LAST_HANDLE.trace(1)[0]:
            if r[0] > 0:
                Logger.logger.debug('{:3d} {:>6s} {:9.2e} {:11.4e} {:10.2e} {:10.2e} {:8.4f} {:8.4f} {:9.2e}'.format(
                    int(r[0]), 'main' if r[1] < 1 else ('chord' if r[1] % 1 else ('polish' if r[1] >= 3 else 'center')), r[2], r[3], r[4], r[5], min(r[6], 1.0), min(r[7], 1.0), r[8]))
    if ipm == 'optimal':
        Logger.logger.debug('SDP solution:')
        Logger.logger.debug('alpha: {}'.format(res['alpha'][0]))
        Logger.logger.debug('beta: {}'.format(res['beta'][0]))
    else:
        Logger.logger.debug('solution status: {} ...'.format(ipm))
    if status in ['Optimal', 'Feasible']:
        Logger.logger.info('{} solution found.'.format(status))
        Logger.logger.info('Maximum condition number: {}'.format(res['info'][0, 4]))
        Logger.logger.info('Minimum eigenvalue: {}'.format(res['info'][0, 3]))
    else:
        Logger.logger.info('SDP solver status: {}'.format(ipm))
        Logger.logger.info('Minimum eigenvalue: {}'.format(res['info'][0, 3]))
        Logger.logger.info('!! Problem infeasible !!')
    return status


def convexify(A, B, Q, R, N, G=None, C=None, opts={'rho': 1e-3, 'solver': 'hip', 'force': False}):
    """ Convexify the indefinite Hessian "H" of the system with the discrete time dynamics

        x_{k+1} = A x_k + B u_k

    so that the solution of the LQR problem based on the convexified Hessian "H + dH" yields the same
    trajectory as the LQR-solution of the indefinite problem  (convexifier.py:36-58).

    :param A: system matrix            :param B: input matrix
    :param Q: weighting matrix Q (nx,nx)   :param R: (nu,nu)   :param N: (nx,nu)
    :param C: jacobian of active constraints at steady state (nc, nx+nu)
    :param G: jacobian of equality constraints at steady state (ng, nx+nu)
    :param opts: tuning options {'rho', 'solver', 'force'}  (never mutated, unlike convexifier.py:89-91); 'tight': True or a tolerance -- Steps 1 and 2 in the
                 tight-accuracy mode of the library (relative gap on kappa N * 7e-12 instead of N * 3e-8, the accuracy MOSEK / CVXOPT stop at; with G or C
                 while rows * (2 n + 2 nx) <= 4040; not with 'force': Step 3 has no tight mode -- NotImplementedError); one more key,
                 'objective': 'paper' (default) | 'beta'.  The reference assembles the Step 2/3 objective with
                 `picos.sum(obj, abs(rho*F[i]))` (convexifier.py:276-285).  In PICOS 1.2.0 the second positional parameter of
                 picos.sum may be an iterator label rather than a summand (unverifiable here, PICOS is not
                 installed): then the solver minimises beta alone and the multipliers are cost-free.  'paper' is the objective
                 of the paper (eq. 20a): beta + rho * sum(||F_k|| + ||Fg_k|| [+ ||T_k||]); 'beta' is the other reading, for
                 Step 2 (the rows of C_k act like rows of G_k).  For Step 3 a cost-free T_k has no counterpart here.
    :return: Convexified Hessian supplement "dH": (dHc, dQc, dRc, dNc), lists of p arrays.
    """
    arg = {'A': A, 'B': B, 'Q': Q, 'R': R, 'N': N}
    if C is None:
        Logger.logger.info('Convexifier called w/o active constraints at steady state')
    else:
        arg['C'] = C
    if G is not None:
        arg['G'] = G

    arg = preprocessing.input_checks(arg)
    period = len(arg['A'])
    Logger.logger.info('Convexify Hessians along {:d}-periodic steady state trajectory.'.format(period))

    As = np.stack([_to_array(a) for a in arg['A']])
    Bs = np.stack([_to_array(b) for b in arg['B']])
    nx = As.shape[1]
    nu = Bs.shape[2]
    Hs = np.stack([mtools.buildHessian(_to_array(q), _to_array(r), _to_array(n_))
                   for q, r, n_ in zip(arg['Q'], arg['R'], arg['N'])])

    if (opts or {}).get('tight') and (opts or {}).get('force', False):
        raise NotImplementedError("opts['tight'] covers Steps 1 and 2; it cannot be combined with opts['force'] (Step 3 has no tight mode)")
    solver = (opts or {}).get('solver', 'hip')
    if solver not in ('hip', 'mosek', 'cvxopt'):
        raise ValueError("unknown solver '{}' (this build provides 'hip')".format(solver))
    objective = (opts or {}).get('objective', 'paper')
    if objective not in ('paper', 'beta'):
        raise ValueError("unknown objective '{}' ('paper': beta + rho * norm terms, 'beta': beta alone)".format(objective))
    rho2 = 0.0 if objective == 'beta' else (opts or {}).get('rho', 1e-3)      # Step 2: rho = 0 is the beta-only model

    Gs = None
    if 'G' in arg:        # the multipliers Fg_k >= 0 belong to every step, Step 1 included (convexifier.py:249-255)
        Gs = np.stack([_to_array(g) for g in arg['G']])
        if Gs.shape[1] == 0:
            Gs = None
    Cl = rows = None
    nc = 0
    if 'C' in arg:
        Cl = [None if c is None else _to_array(c) for c in arg['C']]
        rows = [0 if c is None else c.shape[0] for c in Cl]
        nc = max(1, max(rows))

    if Gs is not None and Gs.shape[1] > NG_MAX:
        raise NotImplementedError('the HIP path handles up to {} equality-constraint rows per stage (got {})'.format(NG_MAX, Gs.shape[1]))

    # check if hessian is already convex (convexifier.py:82-85: before anything is constructed or logged about the SDP):
    # batched eigenvalue scan on the device
    h = _handle(period, nx, nu, 0 if Gs is None else Gs.shape[1], nc if (nc <= NC_MAX and nx + nu <= N_ROWS_MAX) else 0, 1)
    if h.eig_scan(Hs[None])[0, :, 0].min() > 0:
        Logger.logger.info('Provided hessian(s) are already positive definite. No convexification needed!')
        return np.zeros((nx + nu, nx + nu)), np.zeros((nx, nx)), np.zeros((nu, nu)), np.zeros((nx, nu))

    Logger.logger.info('Construct SDP...')
    Logger.logger.info('')
    Logger.logger.info(50 * '*')
    Logger.logger.info('Step 1: (η_F = 0), (η_T = 0)')
    Logger.logger.info('solving SDP...')
    res = convexify_batch(As[None], Bs[None], Hs[None], G=None if Gs is None else Gs[None], nc_hint=nc if (nc <= NC_MAX and nx + nu <= N_ROWS_MAX) else 0, tight=(opts or {}).get('tight'))      # (no room for C rows beyond n = 64: Step 1 then runs on the plain handle and may well succeed -- ADVICE r5)

    if res['info'][0, 13] != 0.0:      # (the library's own pre-check; same answer as the scan above)
        Logger.logger.info('Provided hessian(s) are already positive definite. No convexification needed!')
        return np.zeros((nx + nu, nx + nu)), np.zeros((nx, nx)), np.zeros((nu, nu)), np.zeros((nx, nu))

    status = _log_solution(res)
    if status in ['Optimal', 'Feasible']:
        Logger.logger.info('EQUIVALENCE TYPE A')
        Logger.logger.info(50 * '*')

    if status == 'Infeasible' and 'C' in arg:        # convexifier.py:116-131
        Logger.logger.info(50 * '*')
        Logger.logger.info('Step 2: (η_F = 1), (η_T = 0)')
        Cp = np.zeros((period, nc, nx + nu))
        for k, c in enumerate(Cl):
            if rows[k]:
                Cp[k, :rows[k]] = c
        Logger.logger.info('solving SDP...')
        res = convexify_step2_batch(As[None], Bs[None], Hs[None], Cp[None], np.asarray(rows, np.int32)[None],
                                    rho2, G=None if Gs is None else Gs[None], tight=(opts or {}).get('tight'))
        status = _log_solution(res)
        if status in ['Optimal', 'Feasible']:
            Logger.logger.info('EQUIVALENCE TYPE B')
            Logger.logger.info(50 * '*')

    if status == 'Infeasible':
        Logger.logger.warning('!! Strict dissipativity does not hold locally !!')
        Logger.logger.warning('!! The provided indefinite LQ MPC problem is not stabilising !!')
        Logger.logger.warning(50 * '*')
        if (opts or {}).get('force', False):                                  # convexifier.py:137-147
            if objective == 'beta':
                # With a cost-free T_k the SDP of Step 3 is degenerate: T_k alone can produce any M_k, the optimum is kappa* = 1 with M_k = I, and the
                # result is Hc_k = I / (s alpha) for every stage whatever H was, with alpha undetermined (tests/test_oracle.py::
                # test_step3_with_cost_free_T_is_degenerate).  Not a model worth a kernel: refuse rather than return c * I.
                raise NotImplementedError("opts['objective'] = 'beta' covers Step 2; Step 3 with a cost-free T_k (the beta-only reading of "
                                          "convexifier.py:284-285) is degenerate -- its optimum is Hc_k = c * I for every stage, see INTEGRATION.md -- "
                                          "use the default objective")
            Logger.logger.info('Step 3: (η_F = 1), (η_T = 1)')
            Logger.logger.info('Enforcing convexification...')
            Logger.logger.info('solving SDP...')
            if 'C' in arg:                                                    # convexifier.py:144: constr = constraint_contribution
                Cp = np.zeros((period, nc, nx + nu))
                for k, c in enumerate(Cl):
                    if rows[k]:
                        Cp[k, :rows[k]] = c
                res = convexify_step3_batch(As[None], Bs[None], Hs[None], (opts or {}).get('rho', 1e-3), G=None if Gs is None else Gs[None],
                                            C=Cp[None], ncnt=np.asarray(rows, np.int32)[None])
            else:
                res = convexify_step3_batch(As[None], Bs[None], Hs[None], (opts or {}).get('rho', 1e-3), G=None if Gs is None else Gs[None])
            status = _log_solution(res)
            Logger.logger.warning(50 * '*')
        else:
            Logger.logger.warning('Consider operating the system at another orbit of different period p')
            Logger.logger.warning('Convexification and stabilization of the MPC scheme can be enforced by enabling "force"-flag.')
            Logger.logger.warning('In this case there are no guarantees of (local, first-order) equivalence.')
            raise ValueError('Convexification is not possible if the system is not optimally operated at the optimal orbit.')

    Logger.logger.info('')
    Logger.logger.info('Hessians convexified.')
    Logger.logger.info('')

    dH = res['dHc'][0]
    dHc = [dH[k].copy() for k in range(period)]
    dQc = [d[:nx, :nx] for d in dHc]
    dRc = [d[nx:, nx:] for d in dHc]
    dNc = [d[:nx, nx:] for d in dHc]
    return dHc, dQc, dRc, dNc


def convexHessianSuppl(A, B, Q, R, N, dP, G=None, Fg=None, C=None, F=None, T=None):
    """Construct the convexified Hessian supplement from dP (convexifier.py:165-211), on the GPU:
    dHc_k = sym(A'P+A - P etc. [+ G_k' diag(Fg_k) G_k] [+ C_k' diag(F_k) C_k] [+ T_k]).  As in the reference the G term
    needs Fg, the C term is applied only when F is given and C_k is not None (:198-201)."""
    As = np.stack([_to_array(a) for a in A]); Bs = np.stack([_to_array(b) for b in B])
    Ps = np.stack([_to_array(p_) for p_ in dP])
    period, nx, _ = As.shape
    n = nx + Bs.shape[2]
    rows = [[] for _ in range(period)]; wts = [[] for _ in range(period)]
    if G:
        for i in range(period):
            Gi = _to_array(G[i]); fi = np.reshape(_to_array(Fg[i]), (-1,))
            assert Gi.shape == (fi.shape[0], n), 'G_k (ng x n) and Fg_k (ng) expected'
            rows[i].append(Gi); wts[i].append(fi)
    if F:
        for i in range(period):
            if C[i] is not None:
                Ci = _to_array(C[i]); fi = np.reshape(_to_array(F[i]), (-1,))
                assert Ci.shape == (fi.shape[0], n), 'C_k (nc_k x n) and F_k (nc_k) expected'
                rows[i].append(Ci); wts[i].append(fi)
    nr = max(sum(r.shape[0] for r in rows[i]) for i in range(period))
    J = W = None
    if nr > 0:
        J = np.zeros((1, period, nr, n)); W = np.zeros((1, period, nr))
        for i in range(period):
            if rows[i]:
                Ji = np.vstack(rows[i]); J[0, i, :Ji.shape[0]] = Ji; W[0, i, :Ji.shape[0]] = np.concatenate(wts[i])
    Ts = np.stack([_to_array(t) for t in T])[None] if T else None
    dH = _handle(period, nx, Bs.shape[2], 0, 0, 1).supplement_terms_batch(As[None], Bs[None], Ps[None], J, W, Ts)[0]
    dHc = [dH[k] for k in range(period)]
    return dHc, [d[:nx, :nx] for d in dHc], [d[nx:, nx:] for d in dHc], [d[:nx, nx:] for d in dHc]
