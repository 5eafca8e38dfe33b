"""CPU tests of oracle/reference_sdp.py: the dense restatement of the COMPLETE reference model (Steps 1-3 with G/C/T,
convexifier.py:36-163, :213-357).  Checked against the structured Step-1 oracle, constructed cases whose answer is known
by construction, and the invariants of SURVEY.md section 8c.  (The HIP path covers Step 1; Steps 2/3 are the next scope row.)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import convexify_oracle as co          # noqa: E402
import reference_sdp as rs             # noqa: E402


def _split(H, nx):
    return [h[:nx, :nx] for h in H], [h[nx:, nx:] for h in H], [h[:nx, nx:] for h in H]


def _rblock_problem(p=2, nx=2, nu=1, seed=0):
    """B_k = 0 and R_k < 0: the R block of Hc_k is alpha*R_k + B'PB = alpha*R_k, never positive definite without help
    (the infeasible family of SURVEY.md 8c(3)); Q_k = I, N_k = 0 keep everything else trivially fine."""
    rng = np.random.default_rng(seed)
    A = [0.5 * np.eye(nx) + 0.1 * rng.standard_normal((nx, nx)) for _ in range(p)]
    B = [np.zeros((nx, nu)) for _ in range(p)]
    Q = [np.eye(nx) for _ in range(p)]; R = [-0.5 * np.eye(nu) for _ in range(p)]; N = [np.zeros((nx, nu)) for _ in range(p)]
    Cu = [np.hstack([np.zeros((nu, nx)), np.eye(nu)]) for _ in range(p)]          # rows that reach exactly the input directions
    return A, B, Q, R, N, Cu


def test_model_sizes_match_the_reference_count():
    """variables = 2 + p d (+ p ng) (+ sum nc_i) (+ p n(n+1)/2)   plus one epigraph variable per norm term."""
    A, B, Q, R, N, Cu = _rblock_problem(p=3, nx=2, nu=1)
    p, nx, n, d = 3, 2, 3, 3
    M, _, _ = rs.set_up_model(A, B, Q, R, N, constr=False)
    assert M.m == 2 + p * d and len(M.blocks) == 2 * p and len(M.lp_rows) == 1
    G = [np.ones((2, n)) for _ in range(p)]
    C = [Cu[0], None, np.vstack([Cu[2], Cu[2]])]
    M, _, _ = rs.set_up_model(A, B, Q, R, N, G=G, C=C, constr=True, force=True)
    nvar = 2 + p * d + p * 2 + (1 + 0 + 2) + p * n * (n + 1) // 2
    nnorm = 2 + p + p                                   # F (two stages have constraints), Fg, T
    assert M.m == nvar + nnorm
    assert len(M.blocks) == 2 * p + nnorm
    assert len(M.lp_rows) == 1 + p * 2 + 3 + p * n * (n + 1) // 2


@pytest.mark.parametrize('seed,p,nx,nu', [(20, 2, 2, 1), (0, 3, 3, 2), (30, 2, 3, 1), (20, 1, 3, 1)])
def test_step1_agrees_with_structured_oracle(seed, p, nx, nu):
    A, B, H = co.gen_problem(seed, p, nx, nu)[:3]
    Q, R, N = _split(H, nx)
    res = rs.solve_step(list(A), list(B), Q, R, N, constr=False)
    ref = co.convexify_arrays(A, B, H)
    assert res['status'] == 'Optimal' and res['solver_status'] == 'optimal'
    s, sbeta = co.auto_scaling(H)
    assert abs(sbeta * res['beta'] / ref['kappa'] - 1.0) < 1e-6                 # same optimal value kappa* = s_beta beta*
    for k in range(p):
        ev = np.linalg.eigvalsh(res['Hc'][k])
        assert ev[0] > 0 and ev[-1] / ev[0] <= ref['kappa'] * (1 + 1e-6)
        assert np.allclose(res['Hc'][k], H[k] + res['dHc'][k])
    chk = co.convex_hessian_suppl(A, B, np.stack(res['dP']))[0]                  # Hc - H is calH(P*) exactly (invariant 2 of 8c)
    assert np.allclose(np.stack(res['dHc']), chk, atol=1e-12)


def test_step2_active_constraints_rescue_an_infeasible_step1():
    A, B, Q, R, N, Cu = _rblock_problem()
    r1 = rs.solve_step(A, B, Q, R, N, constr=False)
    assert r1['status'] == 'Infeasible'
    dHc, dQc, dRc, dNc, info = rs.convexify_reference(A, B, Q, R, N, C=Cu, opts={'rho': 1e-3})
    assert info['step'] == 2 and info['status'] == 'Optimal'
    res = info['result']
    for k in range(len(A)):
        assert np.linalg.eigvalsh(res['Hc'][k])[0] > 0
        assert res['F'][k][0] > 0.5                       # the multiplier has to lift R = -0.5 above zero
        assert np.allclose(dRc[k], Cu[k][:, 2:].T @ np.diagflat(res['F'][k]) @ Cu[k][:, 2:], atol=1e-9)   # B = 0: the R supplement is C'FC alone
    with pytest.raises(ValueError, match='Convexification is not possible'):
        rs.convexify_reference(A, B, Q, R, N, C=None)


def test_step3_force_regularises_and_without_force_raises():
    A, B, Q, R, N, _ = _rblock_problem()
    with pytest.raises(ValueError, match='Convexification is not possible'):
        rs.convexify_reference(A, B, Q, R, N, opts={'force': False})
    dHc, _, dRc, _, info = rs.convexify_reference(A, B, Q, R, N, opts={'force': True, 'rho': 1e-3})
    assert info['step'] == 3 and info['status'] in ('Optimal', 'Feasible')
    res = info['result']
    for k in range(len(A)):
        assert np.linalg.eigvalsh(res['Hc'][k])[0] > 0
        assert (res['T'][k] >= -1e-9).all() and res['T'][k][2, 2] > 0.5            # T > 0 elementwise (:273) and large where R needs it


def test_equality_constraint_term_acts_already_in_step1():
    """With G given, Fg >= 0 enters Step 1 without a cost term (convexifier.py:249-255, :276-283 only under `constr`)."""
    A, B, Q, R, N, Cu = _rblock_problem()
    dHc, _, dRc, _, info = rs.convexify_reference(A, B, Q, R, N, G=Cu)
    assert info['step'] == 1 and info['status'] == 'Optimal'
    res = info['result']
    assert all(res['Fg'][k][0] > 0.5 for k in range(len(A)))
    assert all(np.linalg.eigvalsh(res['Hc'][k])[0] > 0 for k in range(len(A)))


def test_early_exit_returns_bare_zero_arrays():
    A, B, Q, R, N, _ = _rblock_problem()
    R = [np.eye(1) for _ in A]
    out = rs.convexify_reference(A, B, Q, R, N)
    assert out[4]['step'] == 0 and out[0].shape == (3, 3) and not out[0].any() and out[2].shape == (1, 1)


@pytest.mark.parametrize('name,kw', [('n1_step2_active_constraints', 'C'), ('n1_step1_equality_term', 'G'), ('n1_step3_force', None)])
def test_steps23_golden_vectors(name, kw):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
    p = g['A'].shape[0]
    lst = lambda a: [a[k] for k in range(p)]
    extra = {kw: lst(g['Cu'])} if kw else {}
    out = rs.convexify_reference(lst(g['A']), lst(g['B']), lst(g['Q']), lst(g['R']), lst(g['N']),
                                 opts={'rho': float(g['rho']), 'force': bool(g['force'])}, **extra)
    info = out[4]
    assert info['step'] == int(g['step']) and info['status'] == str(g['status'])
    assert abs(info['result']['objective'] / float(g['objective']) - 1.0) < 1e-6
    assert abs(info['kappa'] / float(g['kappa']) - 1.0) < 1e-4


@pytest.mark.parametrize('seed,p,nx,nu,ng', [(20, 3, 3, 2, 2), (0, 3, 3, 2, 1), (30, 2, 3, 1, 3)])
def test_structured_oracle_with_equality_term_matches_dense_model(seed, p, nx, nu, ng):
    """Step 1 with G (cost-free multipliers Fg >= 0, convexifier.py:249-255): the structured oracle (extra border columns)
    and the dense restatement reach the same optimal value; the extra freedom can only lower kappa*."""
    A, B, H = co.gen_problem(seed, p, nx, nu)[:3]
    G = np.random.default_rng(seed + 1).standard_normal((p, ng, nx + nu))
    r = co.convexify_arrays(A, B, H, G=G)
    r0 = co.convexify_arrays(A, B, H)
    Q, R, N = _split(H, nx)
    ref = rs.solve_step(list(A), list(B), Q, R, N, G=[g for g in G], constr=False)
    assert r['status'] == co.STATUS_OPTIMAL and ref['status'] == 'Optimal'
    assert abs(r['kappa'] / (ref['beta'] * r['sbeta']) - 1.0) < 1e-5
    assert r['kappa'] <= r0['kappa'] * (1 + 1e-9)
    assert (r['Fg'] >= 0).all()
    ev = np.linalg.eigvalsh(r['Hc'])
    assert ev.min() > 0 and (ev[:, -1] / ev[:, 0]).max() <= r['kappa'] * (1 + 1e-8)
    assert np.abs(r['Hc'] - H - co.convex_hessian_suppl(A, B, r['P'], G=G, Fg=r['Fg'])[0]).max() < 1e-12


@pytest.mark.parametrize('seed,p,nx,nu,ng,ncs,rho', [(20, 3, 3, 2, 0, [2, 0, 1], 1e-3), (0, 3, 3, 2, 2, [1, 2, 0], 1.0), (30, 2, 3, 1, 1, [1, 1], 1e-3),
                                                   (20, 1, 3, 1, 0, [2], 1.0)])
def test_structured_oracle_step2_matches_dense_model(seed, p, nx, nu, ng, ncs, rho):
    """Step 2 model (constr=True, convexifier.py:258-283): multipliers F_k >= 0 of ragged active-constraint Jacobians (None
    entries included), norms rho*||F_k|| and rho*||Fg_k|| in the objective.  The structured oracle (stage-local border columns,
    arrow LMIs for the norms) and the dense restatement reach the same objective."""
    A, B, H = co.gen_problem(seed, p, nx, nu)[:3]
    rng = np.random.default_rng(seed + 5)
    G = rng.standard_normal((p, ng, nx + nu)) if ng else None
    C = [rng.standard_normal((c, nx + nu)) if c else None for c in ncs]
    r = co.sdp_step1(A, B, H, G=G, C=C, rho=rho)
    Q, R, N = _split(H, nx)
    ref = rs.solve_step(list(A), list(B), Q, R, N, G=None if G is None else [g for g in G], C=C, rho=rho, constr=True)
    assert r['ipm_status'] == 'optimal' and ref['status'] == 'Optimal'
    assert abs(r['objective'] / ref['objective'] - 1.0) < 5e-6
    assert all((f is None) == (c is None) and (f is None or (f >= 0).all()) for f, c in zip(r['F'], C))
    dHc = co.convex_hessian_suppl(A, B, r['P'], G=G, Fg=r.get('Fg'), C=C, F=r['F'])[0]
    ev = np.linalg.eigvalsh(H + dHc)
    assert ev.min() > 0 and (ev[:, -1] / ev[:, 0]).max() <= r['kappa'] * (1 + 1e-8)


def test_structured_oracle_takes_step2_when_step1_is_infeasible():
    """The dense model's vector: B = 0, R < 0 -> Step 1 infeasible, Step 2 feasible through C (convexifier.py:116-131)."""
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'n1_step2_active_constraints.npz'))
    p = g['A'].shape[0]
    H = np.stack([co.build_hessian(g['Q'][k], g['R'][k], g['N'][k]) for k in range(p)])
    C = [g['Cu'][k] for k in range(p)]
    assert co.convexify_arrays(g['A'], g['B'], H)['status'] == co.STATUS_INFEASIBLE
    r = co.convexify_arrays(g['A'], g['B'], H, C=C, rho=float(g['rho']))
    assert r['step'] == 2 == int(g['step']) and r['status'] == co.STATUS_OPTIMAL
    assert abs(r['objective'] / float(g['objective']) - 1.0) < 2e-6 and abs(r['kappa'] / float(g['kappa']) - 1.0) < 1e-4
    out = co.convexify([a for a in g['A']], [b for b in g['B']], [q for q in g['Q']], [x for x in g['R']], [x for x in g['N']],
                       C=C, opts={'rho': float(g['rho'])})
    assert np.abs(np.stack(out[0]) - r['dHc']).max() == 0.0 and all(x[0, 0] > 0.5 for x in out[2])


@pytest.mark.parametrize('seed,p,nx,nu,ng,ncs,rho', [(20, 3, 3, 2, 0, None, 1e-3), (0, 2, 2, 1, 1, [1, 0], 1.0), (30, 2, 3, 1, 0, [1, 1], 1e-3), (20, 1, 3, 1, 0, None, 1e-3)])
def test_structured_oracle_step3_matches_dense_model(seed, p, nx, nu, ng, ncs, rho):
    """Step 3 model (force=True, convexifier.py:269-273, :284-285, :352-353): T_k symmetric, every entry > 0, rho*||T_k||_F in
    the objective, with and without the Step 2 terms.  The n(n+1)/2 entries of T_k are stage-local multipliers whose direction
    is a basis matrix; the structured oracle and the dense restatement reach the same objective."""
    A, B, H = co.gen_problem(seed, p, nx, nu)[:3]
    rng = np.random.default_rng(seed + 5)
    G = rng.standard_normal((p, ng, nx + nu)) if ng else None
    C = [rng.standard_normal((c, nx + nu)) if c else None for c in ncs] if ncs else None
    r = co.sdp_step1(A, B, H, G=G, C=C, rho=rho, force=True)
    Q, R, N = _split(H, nx)
    ref = rs.solve_step(list(A), list(B), Q, R, N, G=None if G is None else [g for g in G], C=C, rho=rho, constr=C is not None, force=True)
    assert r['ipm_status'] == 'optimal' and ref['status'] == 'Optimal'
    assert abs(r['objective'] / ref['objective'] - 1.0) < 1e-5
    assert (r['T'] > 0).all() and np.abs(r['T'] - np.swapaxes(r['T'], 1, 2)).max() == 0.0
    dHc = co.convex_hessian_suppl(A, B, r['P'], G=G, Fg=r.get('Fg'), C=C, F=r.get('F'), T=r['T'])[0]
    ev = np.linalg.eigvalsh(H + dHc)
    assert ev.min() > 0 and (ev[:, -1] / ev[:, 0]).max() <= r['kappa'] * (1 + 1e-8)


def test_structured_oracle_takes_step3_when_forced():
    """The dense model's vector: B = 0, R < 0, no constraints -> Steps 1 infeasible, force -> Step 3 (convexifier.py:137-147)."""
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'n1_step3_force.npz'))
    p = g['A'].shape[0]
    H = np.stack([co.build_hessian(g['Q'][k], g['R'][k], g['N'][k]) for k in range(p)])
    with pytest.raises(ValueError, match='Convexification is not possible'):
        co.convexify([a for a in g['A']], [b for b in g['B']], [q for q in g['Q']], [x for x in g['R']], [x for x in g['N']])
    r = co.convexify_arrays(g['A'], g['B'], H, rho=float(g['rho']), force=True)
    assert r['step'] == 3 == int(g['step']) and r['status'] == co.STATUS_OPTIMAL
    assert abs(r['objective'] / float(g['objective']) - 1.0) < 2e-6 and abs(r['kappa'] / float(g['kappa']) - 1.0) < 1e-4
    out = co.convexify([a for a in g['A']], [b for b in g['B']], [q for q in g['Q']], [x for x in g['R']], [x for x in g['N']],
                       opts={'rho': float(g['rho']), 'force': True})
    assert np.abs(np.stack(out[0]) - r['dHc']).max() == 0.0 and all(x[0, 0] > 0.5 for x in out[2])
