"""Row N4: the SQP Hessian regularisation (tunempc/sqp_method.py:327-403).  CPU: the oracle against closed forms and the properties
the reference relies on; GPU: tmpc_eig_clip_host and the host mirror tunempc_amd/sqp.py against the oracle."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import sqp_oracle as so  # noqa: E402


def sym_with_eigs(rng, eigs):
    n = len(eigs)
    Q = np.linalg.qr(rng.standard_normal((n, n)))[0]
    return (Q * np.asarray(eigs)) @ Q.T


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


# ----------------------------------------------------------------------------- oracle (CPU)
def test_oracle_clip_known_answer():
    """diag(-2, 0.5, 3) rotated: clipping at tol = 1 must give eigenvalues (1, 1, 3) in the same eigenbasis; reg = 3."""
    rng = np.random.default_rng(0)
    Q = np.linalg.qr(rng.standard_normal((3, 3)))[0]
    A = (Q * np.array([-2.0, 0.5, 3.0])) @ Q.T
    out, eva, reg = so.eig_clip(A, 1.0)
    assert rel(out, (Q * np.array([1.0, 1.0, 3.0])) @ Q.T) < 1e-13 and abs(reg - 3.0) < 1e-12
    assert np.allclose(np.sort(eva), [-2.0, 0.5, 3.0])


def test_oracle_full_and_reduced_modes():
    rng = np.random.default_rng(1)
    n, mc = 12, 4
    H = sym_with_eigs(rng, np.linspace(-1.0, 2.0, n))
    J = rng.standard_normal((mc, n))
    Hf, regf = so.regularize_hessian(H, regularization='full', tol=1e-6)
    assert np.linalg.eigvalsh(Hf).min() > 1e-6 * (1 - 1e-6) and abs(regf - (1e-6 + 1.0)) < 1e-10
    Hr, regr = so.regularize_hessian(H, J, 'reduced', tol=1e-6)
    from scipy.linalg import null_space
    Z = null_space(J)
    assert np.linalg.eigvalsh(Z.T @ Hr @ Z).min() > 1e-6 * (1 - 1e-6)
    # the correction lives in the null space only: J-range components of H are untouched
    P = np.eye(n) - Z @ Z.T
    assert rel(P @ Hr @ P, P @ H @ P) < 1e-12 and regr > 0
    # a Hessian that is positive on the null space already is returned as it came (sqp_method.py:353)
    Hp = H + 5.0 * np.eye(n)
    Hq, regq = so.regularize_hessian(Hp, J, 'reduced', tol=1e-6)
    assert np.array_equal(Hq, Hp) and regq == 0.0
    Hn, regn = so.regularize_hessian(H, J, 'none', tol=1e-6)
    assert np.array_equal(Hn, H) and regn == 0.0


def test_host_mirror_imports_without_gpu():
    from tunempc_amd import sqp
    assert callable(sqp.regularize_hessian)


# ----------------------------------------------------------------------------- GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize('n,seed', [(1, 0), (2, 1), (5, 2), (33, 3), (100, 4), (257, 5)])
def test_eig_clip_parity(n, seed):
    from tunempc_amd import _lib
    rng = np.random.default_rng(100 + seed)
    eigs = rng.standard_normal(n) * 3.0
    if n >= 5:
        eigs[0], eigs[1] = 1.5, -1.5              # a +- pair (what a plain one-sided Jacobi on an indefinite matrix would mix)
        eigs[2] = eigs[3] = 0.25                  # a repeated eigenvalue
    A = sym_with_eigs(rng, eigs)
    tol = 1e-2
    res = _lib.eig_clip(A, tol)
    ref, eva, reg = so.eig_clip(A, tol)
    assert rel(res['out'], ref) < 1e-11
    assert np.abs(np.sort(res['evals']) - eva).max() < 1e-11 * max(1.0, np.abs(eva).max())
    assert abs(res['reg'] - reg) < 1e-11 * max(1.0, reg)
    assert np.linalg.eigvalsh(res['out']).min() > tol * (1 - 1e-9)
    assert res['sweeps'] <= 20


@pytest.mark.gpu
def test_eig_clip_batch_and_identity():
    from tunempc_amd import _lib
    rng = np.random.default_rng(7)
    A = np.stack([sym_with_eigs(rng, rng.standard_normal(24)) for _ in range(5)])
    res = _lib.eig_clip(A, 1e-6)
    for b in range(5):
        assert rel(res['out'][b], so.eig_clip(A[b], 1e-6)[0]) < 1e-11
    # positive definite input: untouched (up to symmetrisation), reg = 0
    P = sym_with_eigs(rng, np.linspace(1.0, 4.0, 16))
    r2 = _lib.eig_clip(P, 1e-6)
    assert rel(r2['out'], P) < 1e-15 and r2['reg'] == 0.0


@pytest.mark.gpu
def test_regularize_hessian_mirror_parity():
    """tunempc_amd.sqp.regularize_hessian against the restatement of sqp_method.py:327-403, both modes, at an NLP-like size."""
    from tunempc_amd import sqp
    rng = np.random.default_rng(11)
    n, mc = 90, 30
    H = sym_with_eigs(rng, np.concatenate([-np.abs(rng.standard_normal(10)), np.abs(rng.standard_normal(n - 10)) + 0.1]))
    J = rng.standard_normal((mc, n))
    for mode in ('reduced', 'full', 'none'):
        Hg, rg = sqp.regularize_hessian(H, J, mode, tol=1e-7)
        Ho, ro = so.regularize_hessian(H, J, mode, tol=1e-7)
        assert rel(Hg, Ho) < 1e-11 and abs(rg - ro) < 1e-11 * max(1.0, ro)
    Hp = H + 10.0 * np.eye(n)
    Hq, rq = sqp.regularize_hessian(Hp, J, 'reduced', tol=1e-7)
    assert np.array_equal(Hq, Hp) and rq == 0.0
