"""Seeded synthetic tuning problems: "random SPD-perturbed Hessians".
Per problem b: rng = default_rng(base_seed + b); A_k Gaussian scaled to spectral radius 0.9, B_k Gaussian/sqrt(nx),
SPD Hhat_k with cond <= 10, symmetric Gaussian Phat_k, H_k = Hhat_k - (V_k' Phat_{k+1} V_k - E' Phat_k E):
indefinite, yet Step 1 of the convexifier is strictly feasible by construction."""
import numpy as np


def gen_problem(seed, p, nx, mb, sigP=1.0, identity=False, cond_exp=1.0, rad=0.9):
    """cond_exp: cond(Hhat_k) <= 10**cond_exp; rad: spectral radius of A_k (the knobs of scripts/robustness_sweep.py: the defaults are the
    benchmark distribution)."""
    rng = np.random.default_rng(seed)
    n = nx + mb
    A = np.zeros((p, nx, nx)); B = np.zeros((p, nx, mb)); Phat = np.zeros((p, nx, nx)); Hhat = np.zeros((p, n, n))
    for k in range(p):
        a = rng.standard_normal((nx, nx)) / np.sqrt(nx)
        rho = np.max(np.abs(np.linalg.eigvals(a)))
        A[k] = a * (rad / rho)
        B[k] = rng.standard_normal((nx, mb)) / np.sqrt(nx)
        W, _ = np.linalg.qr(rng.standard_normal((n, n)))
        lam = np.ones(n) if identity else 10.0 ** rng.uniform(0, cond_exp, n)
        Hhat[k] = (W * lam) @ W.T
        pk = rng.standard_normal((nx, nx)); Phat[k] = sigP * (pk + pk.T) / 2
    V = np.concatenate([A, B], axis=2)
    dH = np.swapaxes(V, 1, 2) @ np.roll(Phat, -1, axis=0) @ V
    dH[:, :nx, :nx] -= Phat
    H = Hhat - dH
    H = (H + np.swapaxes(H, 1, 2)) / 2.0
    return A, B, H


def gen_batch(base_seed, nb, p, nx, mb, **kw):
    out = [gen_problem(base_seed + b, p, nx, mb, **kw) for b in range(nb)]
    return tuple(np.stack([o[i] for o in out]) for i in range(3))
