"""The C++/OpenMP CPU baseline (oracle/cpu_ipm, bench.py's cpu_baseline leg) against the numpy oracle: same algorithm, same answers."""
import numpy as np
import pytest

import convexify_oracle as co
import cpu_ipm


@pytest.mark.parametrize('seed,nb,p,nx,mb', [(0, 3, 3, 3, 2), (20, 4, 1, 3, 1), (30, 4, 2, 3, 1), (5, 2, 16, 3, 2), (11, 1, 6, 12, 4)])
def test_cpu_ipm_matches_oracle(seed, nb, p, nx, mb):
    A, B, H = co.gen_batch(seed, nb, p, nx, mb)
    o = cpu_ipm.convexify_batch(A, B, H, threads=2)
    for b in range(nb):
        r = co.convexify_arrays(A[b], B[b], H[b])
        assert int(o['status'][b]) == int(r['status'])
        assert np.linalg.norm(o['Hc'][b] - r['Hc']) <= 1e-8 * np.linalg.norm(r['Hc'])       # the parity bar of the GPU tests
        assert abs(o['kappa'][b] - r['kappa']) <= 1e-9 * max(1.0, r['kappa'])


def test_cpu_ipm_threads_do_not_change_results():
    A, B, H = co.gen_batch(77, 4, 4, 4, 2)
    o1 = cpu_ipm.convexify_batch(A, B, H, threads=1)
    o4 = cpu_ipm.convexify_batch(A, B, H, threads=4)
    assert np.array_equal(o1['Hc'], o4['Hc']) and np.array_equal(o1['iters'], o4['iters'])


def test_cpu_ipm_infeasible_member():
    A, B, H = co.gen_batch(70, 2, 2, 2, 1)
    A[1] = 0.5 * np.eye(2); B[1] = 0.0
    H[1] = co.build_hessian(np.eye(2), np.array([[-1.0]]), np.zeros((2, 1)))
    o = cpu_ipm.convexify_batch(A, B, H)
    assert int(o['status'][1]) == 2 and int(o['status'][0]) == 0


def test_cpu_ipm_hard_targets_back_off_like_the_oracle():
    """cond(Hhat) = 1e5 (scripts/robustness_sweep.py): the Schur matrix turns numerically singular before the default mu_t.  Both
    restatements back mu_t off by powers of two, taking the step of the shifted factorisation, and end Optimal (round 3)."""
    from tunempc_amd import synthetic
    for b in (2, 5):
        A, B, H = synthetic.gen_problem(7000 + 17 * b, 30, 4, 1, sigP=10.0, cond_exp=5, rad=0.5)
        o = cpu_ipm.convexify_batch(A[None], B[None], H[None])
        r = co.convexify_arrays(A, B, H)
        assert int(o['status'][0]) == int(r['status']) == 0
        assert abs(o['kappa'][0] - r['kappa']) <= 1e-4 * r['kappa']              # within N mu_t of each other whatever the back-off count


# ----------------------------------------------------------------------------- the models with rows (oracle/cpu_ipm/cpu_ipm_con.h, round 6)
def _row_inputs(seed, nb, p, nx, mb, ng, ncs):
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


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


TIE = 2e-9      # the bar the review of round 5 set for "tied to the numpy oracle" (measured: 1e-16 ... 2e-11)


def _check_rows(o, b, A, B, H, r, G, Cl, ng, ncnt):
    st, dHc = co.check_convergence(A, B, H, r['P'], r['ipm_status'], G=G, Fg=r.get('Fg'), C=Cl, F=r.get('F'), T=r.get('T'))[:2]
    assert int(o['status'][b]) == int(st) == 0 and int(o['iters'][b]) == r['iters']
    assert _rel(o['Hc'][b], H + dHc) <= TIE and _rel(o['P'][b], r['P']) <= 10 * TIE
    assert abs(o['kappa'][b] - r['kappa']) <= 1e-10 * max(1.0, r['kappa'])
    if 'objective' in r:
        assert abs(o['objective'][b] - r['objective']) <= 1e-10 * abs(r['objective'])
    p = A.shape[0]
    for k in range(p):
        if ng:
            assert np.linalg.norm(o['FgF'][b, k, :ng] - r['Fg'][k]) <= 10 * TIE * max(1.0, np.linalg.norm(r['Fg'][k]))
        if Cl is not None and Cl[k] is not None:
            assert np.linalg.norm(o['FgF'][b, k, ng:ng + ncnt[k]] - r['F'][k]) <= 10 * TIE * max(1.0, np.linalg.norm(r['F'][k]))
            assert not o['FgF'][b, k, ng + ncnt[k]:].any()
    if 'T' in r:
        assert _rel(o['T'][b], r['T']) <= 10 * TIE


@pytest.mark.parametrize('seed,nb,p,nx,mb,ng,ncs,rho', [
    (20, 2, 3, 3, 2, 0, [2, 0, 1], 1e-3), (0, 1, 3, 3, 2, 2, [1, 2, 0], 1.0), (30, 2, 2, 3, 1, 1, [1, 1], 1e-3), (20, 1, 1, 3, 1, 0, [2], 1.0),
    (7, 2, 5, 4, 2, 3, [0, 3, 1, 2, 3], 1e-2), (11, 2, 4, 6, 3, 2, [4, 0, 8, 1], 1e-3), (13, 1, 3, 8, 2, 2, [0, 0, 0], 1e-2)])
@pytest.mark.parametrize('cost_free', [False, True])
def test_cpu_ipm_step2_matches_oracle(seed, nb, p, nx, mb, ng, ncs, rho, cost_free):
    """Step 2 (convexifier.py:116-131, :258-266, :276-283; either reading of the objective): the stage-local elimination of cpu_ipm_con.h against the
    border columns of the numpy oracle -- the cases of tests/test_gpu_parity.py::test_step2_parity_vs_oracle (p = 1, p = 2, ragged C_k, stages without C_k)."""
    A, B, H, G, C, ncnt = _row_inputs(seed, nb, p, nx, mb, ng, ncs)
    kw = dict(cost_free=True) if cost_free else dict(rho=rho)
    o = cpu_ipm.convexify_con_batch(A, B, H, np.concatenate([G, C], axis=2), ng=ng, ncnt=ncnt, threads=2, **kw)
    for b in range(nb):
        Cl = [C[b, k, :ncnt[b, k]] if ncnt[b, k] else None for k in range(p)]
        Gb = G[b] if ng else None
        r = co.sdp_step1(A[b], B[b], H[b], G=Gb, C=Cl, **kw)
        _check_rows(o, b, A[b], B[b], H[b], r, Gb, Cl, ng, ncnt[b])


@pytest.mark.parametrize('seed,nb,p,nx,mb,ng', [(20, 2, 3, 3, 2, 2), (30, 2, 2, 3, 1, 2), (20, 1, 1, 3, 1, 1), (9, 2, 6, 5, 2, 3)])
def test_cpu_ipm_equality_term_matches_oracle(seed, nb, p, nx, mb, ng):
    """Step 1 with the cost-free multipliers of G_k (convexifier.py:249-255, :346-347)."""
    A, B, H, G, _, _ = _row_inputs(seed, nb, p, nx, mb, ng, [0] * p)
    o = cpu_ipm.convexify_con_batch(A, B, H, G, ng=ng)
    for b in range(nb):
        r = co.sdp_step1(A[b], B[b], H[b], G=G[b])
        _check_rows(o, b, A[b], B[b], H[b], r, G[b], None, ng, None)


@pytest.mark.parametrize('seed,b,p,nx,mb,ng,ncs,rho', [(0, 0, 2, 2, 1, 0, None, 1e-3), (1, 1, 3, 3, 2, 0, None, 1e-2), (2, 1, 1, 3, 1, 0, None, 1e-3), (6, 0, 4, 4, 3, 0, None, 1.0),
                                                     (4, 0, 2, 2, 1, 1, [0, 0], 1e-2), (3, 0, 1, 3, 1, 0, [2], 1e-1), (8, 0, 3, 3, 2, 1, [2, 0, 1], 1e-2)])
def test_cpu_ipm_step3_matches_oracle(seed, b, p, nx, mb, ng, ncs, rho):
    """Step 3 (convexifier.py:137-147, :269-273, :284-285): the entries of T_k as stage-local multipliers, rho ||T_k||_F as a second-order cone; alone and
    together with the rows of G_k / C_k (`constr` as left by Step 2, :144)."""
    A, B, H, G, C, ncnt = _row_inputs(seed, b + 1, p, nx, mb, ng, ncs or [0] * p)
    A, B, H, G, C, ncnt = (x[b:b + 1] for x in (A, B, H, G, C, ncnt))
    if ncs is None:
        o = cpu_ipm.convexify_con_batch(A, B, H, rho=rho, force=True)
        r = co.sdp_step1(A[0], B[0], H[0], rho=rho, force=True)
        _check_rows(o, 0, A[0], B[0], H[0], r, None, None, 0, None)
    else:
        o = cpu_ipm.convexify_con_batch(A, B, H, np.concatenate([G, C], axis=2), ng=ng, ncnt=ncnt, rho=rho, force=True)
        Cl = [C[0, k, :ncnt[0, k]] if ncnt[0, k] else None for k in range(p)]
        Gb = G[0] if ng else None
        r = co.sdp_step1(A[0], B[0], H[0], G=Gb, C=Cl, rho=rho, force=True)
        _check_rows(o, 0, A[0], B[0], H[0], r, Gb, Cl, ng, ncnt[0])


def test_cpu_ipm_rows_early_exit_and_padding():
    """An already convex member leaves through the pre-check (convexifier.py:82-85) with zero multipliers; rows beyond ncnt are never read."""
    A, B, H, G, C, ncnt = _row_inputs(0, 3, 3, 3, 2, 1, [1, 0, 2])
    assert np.linalg.eigvalsh(H[1]).min() > 0          # (member 1 of this seed is convex as generated)
    J = np.concatenate([G, C], axis=2)
    o = cpu_ipm.convexify_con_batch(A, B, H, J, ng=1, ncnt=ncnt, rho=1e-2)
    assert int(o['iters'][1]) == 0 and not o['FgF'][1].any() and np.array_equal(o['Hc'][1], co.symmetrize(H[1]))
    J2 = J.copy()
    for k in range(3):
        J2[:, k, 1 + ncnt[0, k]:] = 7.0               # garbage in the padding
    o2 = cpu_ipm.convexify_con_batch(A, B, H, J2, ng=1, ncnt=ncnt, rho=1e-2)
    assert np.array_equal(o['Hc'], o2['Hc'])
