"""CPU tests of the tight-accuracy mode of the checkers (oracle/ddnum.py, convexify_oracle.sdp_step1(tight=True), oracle/cpu_ipm tight).

What ties them down:
  * ddnum (double-double on numpy arrays) against mpmath at 50 digits;
  * the dd block solve of the oracle against the same IPM with the blocks assembled and factored in 40-digit mpmath arithmetic
    (tests/tools/tight_probe.py: the experiment that located the fp64 wall);
  * the C++ port against the numpy oracle (same iteration counts, Hc to 1e-8, kappa to 1e-12);
  * properties no solver can fake: kappa decreases with the tolerance and stays inside the default's certified gap; the identity family
    (Hc = I, kappa* = 1); reproducibility of the returned point on inputs 1e-14 apart (the polish is what makes the point DEFINED:
    1e-7 without it at mu = 2e-12, 1e-12 with it)."""
import os
import sys

import numpy as np
import pytest

import convexify_oracle as co
import cpu_ipm
import ddnum as dn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
hc_of = lambda A, B, H, r: H + co.symmetrize(co.calH(A, B, r['P']))


def test_ddnum_against_mpmath():
    mp = pytest.importorskip('mpmath')
    mp.mp.dps = 50
    rng = np.random.default_rng(0)
    d = 10
    G = rng.standard_normal((d, d)); A = G @ G.T + 1e-6 * np.eye(d)
    Am = mp.matrix(A.tolist())
    val = lambda x, i, j: mp.mpf(float(x.hi[i, j])) + mp.mpf(float(x.lo[i, j]))
    L = dn.cholesky(dn.DD(A)); Lm = mp.cholesky(Am)
    assert max(abs(val(L, i, j) - Lm[i, j]) for i in range(d) for j in range(d)) < 1e-28
    Bm = rng.standard_normal((d, 2))
    X = dn.solve_lower(L, dn.solve_lower(L, Bm), trans=True)
    Xm = mp.inverse(Am) * mp.matrix(Bm.tolist())
    assert max(abs(val(X, i, j) - Xm[i, j]) / abs(Xm[i, j]) for i in range(d) for j in range(2)) < 1e-26
    C = dn.matmul_nt(dn.DD(A), dn.DD(A)); Cm = Am * Am.T
    assert max(abs(val(C, i, j) - Cm[i, j]) / abs(Cm[i, j]) for i in range(d) for j in range(d)) < 1e-30
    q = dn.DD(np.array([1.0])) / dn.DD(np.array([3.0]))
    assert abs(mp.mpf(float(q.hi[0])) + mp.mpf(float(q.lo[0])) - mp.mpf(1) / 3) < 1e-31
    s = dn.DD(np.array([2.0])).sqrt()
    assert abs(mp.mpf(float(s.hi[0])) + mp.mpf(float(s.lo[0])) - mp.sqrt(2)) < 1e-30


def test_fp64_wall_and_dd_block_solve_against_mpmath():
    """below the wall the fp64 path needs shifts and backs its target off; with the blocks in dd the same iteration arrives, and agrees with
    the blocks in 40-digit arithmetic to the reproducibility floor of the fp64 stage arithmetic (no polish in either run)"""
    pytest.importorskip('mpmath')
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'tools'))
    import tight_probe as tp
    A, B, H, _, _ = co.gen_problem(3, 6, 4, 2)
    tol = 2.0 ** -33
    r64 = co.sdp_step1(A, B, H, dict(tol=2.0 ** -37, max_iter=80, center_iter=20))
    assert r64['shift'] > 0.0 and r64['mu_target'] > 2.0 ** -37 * r64['kappa'] * 1.5           # the wall: shifted factorisations, backed-off target
    old = co.POLISH_ENTER
    co.POLISH_ENTER = 0.0                                                                      # (no hand-over: compare the primal-dual iterations themselves)
    try:
        rdd = co.sdp_step1(A, B, H, dict(tol=tol, max_iter=80, center_iter=20, tight=True))
    finally:
        co.POLISH_ENTER = old
    rmp = co.sdp_step1(A, B, H, dict(tol=tol, max_iter=80, center_iter=20, _assemble=tp.assemble_mp, _chol_cls=tp.MpSystem))
    assert rdd['ipm_status'] == rmp['ipm_status'] == 'optimal' and rdd['shift'] == 0.0 and rdd['dd_iters'] >= 3
    assert rdd['mu_target'] == rmp['mu_target'] <= 2.0 ** -33 * rdd['kappa'] * 1.5
    assert rel(hc_of(A, B, H, rdd), hc_of(A, B, H, rmp)) < 1e-8
    assert abs(rdd['kappa'] - rmp['kappa']) < 1e-12


@pytest.mark.parametrize('seed,p,nx,mb,lt', [(3, 6, 4, 2, 37), (11, 8, 4, 1, 37), (21, 5, 5, 2, 41), (8, 2, 3, 2, 35), (5, 16, 4, 1, 37)])
def test_cpu_port_tight_matches_oracle(seed, p, nx, mb, lt):
    A, B, H, _, _ = co.gen_problem(seed, p, nx, mb)
    r = co.sdp_step1(A, B, H, dict(tol=2.0 ** -lt, tight=True))
    c = cpu_ipm.convexify_batch(A[None], B[None], H[None], tol=2.0 ** -lt, threads=2, tight=True)
    assert r['ipm_status'] == 'optimal' and c['status'][0] == 0
    assert r['iters'] == c['iters'][0] and r['dd_iters'] == c['dd_iters'][0]
    assert r['mu_target'] == c['mu_t'][0]
    assert rel(hc_of(A, B, H, r), c['Hc'][0]) < 1e-8
    assert abs(r['kappa'] - c['kappa'][0]) < 1e-12 * r['kappa']


def test_tight_point_is_reproducible_and_inside_the_default_gap():
    A, B, H, _, _ = co.gen_problem(11, 8, 4, 1)
    rng = np.random.default_rng(1)
    H2 = co.symmetrize(H * (1 + 1e-14 * rng.standard_normal(H.shape)))
    r0 = co.sdp_step1(A, B, H)
    N = 2 * 8 * 5 + 1
    prev = r0['kappa']
    for lt in (33, 37, 41):
        r = co.sdp_step1(A, B, H, dict(tol=2.0 ** -lt, tight=True)); r2 = co.sdp_step1(A, B, H2, dict(tol=2.0 ** -lt, tight=True))
        assert r['ipm_status'] == 'optimal' and r['polish_steps'] >= 1
        assert rel(hc_of(A, B, H, r), hc_of(A, B, H2, r2)) < 1e-10                 # (1e-7 ... 4e-6 at 2^-41 without the polish)
        assert r['kappa'] <= prev + 1e-13                                          # kappa decreases along the path ...
        assert r0['kappa'] - r['kappa'] <= 1.5 * N * r0['mu_target']              # ... and stays inside the gap the default certifies
        prev = r['kappa']
        # the returned dual iterate is on the central path: X_r S_r = mu I to the accuracy fp64 can show
        V = np.concatenate([A, B], axis=2)
        Pb = r['P'] * (r['s'] * r['alpha'])
        M = r['alpha'] * r['s'] * H + np.swapaxes(V, 1, 2) @ np.roll(Pb, -1, axis=0) @ V
        M[:, :4, :4] -= Pb
        S1 = M - np.eye(5)
        assert np.abs(r['X1'] @ S1 - r['mu_target'] * np.eye(5)).max() < 1e-3 * r['mu_target']


def test_tight_identity_family():
    A, B, H, _, _ = co.gen_problem(77, 5, 4, 2, identity=True)
    c = cpu_ipm.convexify_batch(A[None], B[None], H[None], tol=2.0 ** -37, threads=2, tight=True)
    assert c['status'][0] == 0
    assert np.abs(c['Hc'][0] - np.eye(6)).max() < 1e-10
    assert 0.0 <= c['kappa'][0] - 1.0 <= 61 * 2.0 ** -37 * 1.5


def test_tight_batch_of_the_cpu_port():
    """outer parallelism (batch >= threads) and inner parallelism (batch < threads) give the same numbers"""
    A, B, H = co.gen_batch(900, 3, 6, 5, 2)
    a = cpu_ipm.convexify_batch(A, B, H, tol=2.0 ** -37, threads=2, tight=True)
    b = cpu_ipm.convexify_batch(A, B, H, tol=2.0 ** -37, threads=4, tight=True)
    assert (a['status'] == 0).all() and (b['status'] == 0).all()
    assert np.array_equal(a['Hc'], b['Hc']) and np.array_equal(a['kappa'], b['kappa'])


# ----------------------------------------------------------------------------- round 5: the oracle's tight mode on the models with rows
def _rows_case(seed, p, nx, mb, ng, nc, model):
    n = nx + mb
    rng = np.random.default_rng(seed)
    A, B, H = co.gen_batch(400 + seed, 1, p, nx, mb)
    A, B, H = A[0], B[0], H[0]
    G = rng.standard_normal((p, ng, n)) if ng else None
    ncs = rng.integers(0, nc + 1, size=p) if nc else [0] * p
    C = [rng.standard_normal((ncs[k], n)) if ncs[k] else None for k in range(p)] if nc else None
    rho = 0.0 if model == 'beta' else 1e-2
    kw = dict(G=G)
    if model != 'G':
        kw.update(C=C, rho=rho)
    return A, B, H, G, C, rho, kw


@pytest.mark.parametrize('seed,p,nx,mb,ng,nc,model', [(0, 3, 3, 2, 2, 0, 'G'), (1, 4, 3, 2, 1, 2, 'step2'), (2, 2, 4, 2, 0, 2, 'step2'), (3, 3, 3, 1, 1, 2, 'beta')])
def test_tight_with_rows_against_the_dense_solver(seed, p, nx, mb, ng, nc, model):
    """sdp_step1(tight=True) with rows of G, with the Step 2 model and with the beta-only objective (multipliers and the epigraph variables of the norm terms as variables
    of the dd dual-Newton polish) against oracle/reference_sdp.py -- the dense restatement of the COMPLETE reference model with a dense Mehrotra solver that shares nothing
    with the structured one: the objective value (beta + rho * norm terms, solver-independent) agrees to the dense solver's tolerance, an order and more closer than the
    default mode's; and the returned point is reproducible on inputs 1e-14 apart."""
    import reference_sdp as rs
    A, B, H, G, C, rho, kw = _rows_case(seed, p, nx, mb, ng, nc, model)
    r = co.sdp_step1(A, B, H, dict(tol=2.0 ** -37, tight=True), **kw)
    r0 = co.sdp_step1(A, B, H, **kw)
    assert r['ipm_status'] == r0['ipm_status'] == 'optimal' and r['polish_steps'] >= 1
    Q = [H[k][:nx, :nx] for k in range(p)]; R = [H[k][nx:, nx:] for k in range(p)]; N = [H[k][:nx, nx:] for k in range(p)]
    d = rs.solve_step(list(A), list(B), Q, R, N, G=None if G is None else list(G), C=C, rho=rho, constr=(model != 'G'), tol=1e-9)
    assert d['solver_status'] == 'optimal'
    obj, obj0 = r.get('objective', r['beta']), r0.get('objective', r0['beta'])
    assert obj <= obj0                                                           # further down the central path
    assert abs(obj - d['objective']) <= 5e-8 * obj                               # (measured 3e-10 ... 1.2e-8: the dense solver stops at 1e-9)
    assert abs(obj - d['objective']) < 0.1 * abs(obj0 - d['objective'])          # the default mode is 4e-7 ... 1e-6 away
    r2 = co.sdp_step1(A * (1 + 1e-14), B, H, dict(tol=2.0 ** -37, tight=True), **kw)
    assert np.abs(r2['P'] - r['P']).max() <= 1e-9 * np.abs(r['P']).max()
    if G is not None:
        assert np.abs(r2['Fg'] - r['Fg']).max() <= 1e-9 * max(1.0, np.abs(r['Fg']).max())
    if C is not None:
        for k in range(p):
            if C[k] is not None:
                assert np.abs(r2['F'][k] - r['F'][k]).max() <= 1e-9 * max(1.0, np.abs(r['F'][k]).max())


@pytest.mark.parametrize('seed,p,nx,mb,ng,nc', [(0, 2, 2, 1, 0, 0), (1, 3, 3, 2, 0, 0), (2, 2, 3, 1, 1, 0), (3, 2, 2, 2, 1, 2)])
def test_tight_step3_against_the_dense_solver(seed, p, nx, mb, ng, nc):
    """The oracle's tight mode on the Step 3 model (convexifier.py:137-147: the entries of T_k as multipliers, their Frobenius norm through a second-order cone whose
    Jordan inverse and log-det barrier join the dd polish) against the dense solver of oracle/reference_sdp.py: objective to its tolerance, an order and more closer than the
    default mode.  (The HIP library has no tight mode for Step 3 yet: tmpc_set_tight refuses such handles; this pins the definition it will have to meet.)"""
    import reference_sdp as rs
    n = nx + mb
    rng = np.random.default_rng(9 + seed)
    A, B, H = co.gen_batch(500 + seed, 1, p, nx, mb)
    A, B, H = A[0], B[0], H[0]
    G = rng.standard_normal((p, ng, n)) if ng else None
    ncs = rng.integers(0, nc + 1, size=p) if nc else [0] * p
    C = [rng.standard_normal((ncs[k], n)) if ncs[k] else None for k in range(p)] if nc else None
    kw = dict(G=G, rho=1e-2, force=True)
    if C is not None:
        kw.update(C=C)
    r0 = co.sdp_step1(A, B, H, **kw)
    r = co.sdp_step1(A, B, H, dict(tol=2.0 ** -37, tight=True), **kw)
    assert r['ipm_status'] == r0['ipm_status'] == 'optimal' and r['polish_steps'] >= 1 and (r['T'] > 0).all()
    Q = [H[k][:nx, :nx] for k in range(p)]; R = [H[k][nx:, nx:] for k in range(p)]; N = [H[k][:nx, nx:] for k in range(p)]
    d = rs.solve_step(list(A), list(B), Q, R, N, G=None if G is None else list(G), C=C, rho=1e-2, constr=C is not None, force=True, tol=1e-9)
    assert d['solver_status'] == 'optimal'
    assert r['objective'] <= r0['objective']
    assert abs(r['objective'] - d['objective']) <= 5e-8 * r['objective']               # (measured 1e-10 ... 8e-9)
    assert abs(r['objective'] - d['objective']) < 0.1 * abs(r0['objective'] - d['objective'])
    r2 = co.sdp_step1(A * (1 + 1e-14), B, H, dict(tol=2.0 ** -37, tight=True), **kw)
    assert np.abs(r2['T'] - r['T']).max() <= 1e-7 * np.abs(r['T']).max()


@pytest.mark.parametrize('name', ['tight_plain_n6', 'tight_eq_term_n5', 'tight_step2_with_g_n6', 'tight_step3_n5'])
def test_tight_golden_vectors(name):
    """tests/golden/tight_*.npz (make_golden.py tight): the oracle's tight mode reproduces the committed outputs (generated with the dense-solver and reproducibility checks
    on); the GPU suite compares the HIP library with the same files (tests/test_gpu_tight.py::test_tight_golden_vectors_on_the_gpu)."""
    g = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
    A, B, H = g['A'], g['B'], g['H']
    for b in range(A.shape[0]):
        kw = {}
        if 'G' in g.files:
            kw['G'] = g['G'][b]
        if 'C' in g.files:
            ncs = g['ncnt'][b]
            kw.update(C=[g['C'][b, k, :ncs[k]] if ncs[k] else None for k in range(A.shape[1])], rho=float(g['rho']))
        if 'T' in g.files:                       # Step 3 (the library has no tight mode for it yet: the vector pins the oracle's definition)
            kw.update(rho=float(g['rho']), force=True)
        r = co.sdp_step1(A[b], B[b], H[b], dict(tol=float(g['tol']), tight=True), **kw)
        assert r['ipm_status'] == 'optimal' and r['mu_target'] == g['mu_target'][b]
        if 'T' in g.files:
            assert np.abs(r['T'] - g['T'][b]).max() <= 1e-9 * np.abs(g['T'][b]).max() and abs(r['kappa'] - g['kappa'][b]) <= 1e-11 * g['kappa'][b]
            continue
        assert np.abs(r['P'] - g['P'][b]).max() <= 1e-10 * np.abs(g['P'][b]).max() and abs(r['kappa'] - g['kappa'][b]) <= 1e-12 * g['kappa'][b]
        if 'Fg' in g.files:
            assert np.abs(r['Fg'] - g['Fg'][b]).max() <= 1e-10 * max(1.0, np.abs(g['Fg'][b]).max())


# ----------------------------------------------------------------------------- round 6: the C++ port's tight mode on the models with rows
def _rows_inputs(seed, nb, p, nx, mb, ng, ncs):
    n = nx + mb
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    rng = np.random.default_rng(seed + 5)
    nc = max(max(ncs), 1)
    G = rng.standard_normal((nb, p, ng, n))
    C = np.zeros((nb, p, nc, n)); ncnt = np.tile(np.asarray(ncs, np.int32), (nb, 1))
    for b in range(nb):
        for k in range(p):
            C[b, k, :ncs[k]] = rng.standard_normal((ncs[k], n))
    return A, B, H, G, C, ncnt


@pytest.mark.parametrize('mode,seed,nb,p,nx,mb,ng,ncs,rho', [
    ('eq', 20, 2, 3, 3, 2, 2, [0, 0, 0], 0.0), ('eq', 9, 2, 6, 5, 2, 3, [0] * 6, 0.0), ('step2', 20, 2, 3, 3, 2, 0, [2, 0, 1], 1e-3),
    ('step2', 7, 2, 5, 4, 2, 3, [0, 3, 1, 2, 3], 1e-2), ('step2', 30, 2, 2, 3, 1, 1, [1, 1], 1e-3), ('step2', 20, 1, 1, 3, 1, 0, [2], 1.0),
    ('beta', 7, 2, 5, 4, 2, 3, [0, 3, 1, 2, 3], 0.0)])
def test_cpu_ipm_tight_mode_with_rows_matches_the_oracle(mode, seed, nb, p, nx, mb, ng, ncs, rho):
    """oracle/cpu_ipm/cpu_ipm_con.h in the tight mode (stage-local rows of the augmented blocks in double-double, as tmpc_dd.h's k_dd_aug_fill; dd dual-Newton polish
    in (tau, alpha, P, phi, t)) against the numpy oracle's tight mode (dense border columns in fp64, `_polish_dd`): the same central-path point at 2^-37 kappa."""
    import cpu_ipm
    TT = 2.0 ** -37
    A, B, H, G, C, ncnt = _rows_inputs(seed, nb, p, nx, mb, ng, ncs)
    J = np.concatenate([G, C], axis=2)
    kw = dict(eq={}, step2=dict(rho=rho), beta=dict(cost_free=True))[mode]
    o = cpu_ipm.convexify_con_batch(A, B, H, G if mode == 'eq' else J, ng=ng, ncnt=None if mode == 'eq' else ncnt, tol=TT, tight=True, **kw)
    for b in range(nb):
        if np.linalg.eigvalsh(H[b]).min() > 0:
            continue
        Cl = None if mode == 'eq' else [C[b, k, :ncnt[b, k]] if ncnt[b, k] else None for k in range(p)]
        Gb = G[b] if ng else None
        r = co.sdp_step1(A[b], B[b], H[b], dict(tol=TT, tight=True), G=Gb, C=Cl, **kw)
        st, dHc = co.check_convergence(A[b], B[b], H[b], r['P'], r['ipm_status'], G=Gb, Fg=r.get('Fg'), C=Cl, F=r.get('F'))[:2]
        assert int(o['status'][b]) == int(st) == 0 and r['ipm_status'] == 'optimal' and o['mu_t'][b] == r['mu_target']
        assert np.linalg.norm(o['Hc'][b] - H[b] - dHc) <= 1e-9 * np.linalg.norm(H[b] + dHc)
        assert abs(o['kappa'][b] - r['kappa']) <= 1e-10 * r['kappa']
        if ng:
            assert np.abs(o['FgF'][b, :, :ng] - r['Fg']).max() <= 1e-8 * max(1.0, np.abs(r['Fg']).max())


def test_cpu_ipm_tight_mode_refuses_step3():
    import cpu_ipm
    A, B, H = co.gen_batch(0, 1, 2, 2, 1)
    with pytest.raises(NotImplementedError):
        cpu_ipm.convexify_con_batch(A, B, H, rho=1e-3, force=True, tight=True)
