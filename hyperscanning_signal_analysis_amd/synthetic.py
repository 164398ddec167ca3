"""Seeded synthetic workloads -- the build's own generator, not reference code.

Used by bench.py, the parity tests and tests/golden/make_golden.py.  NumPy only (no GPU, no oracle).
"""
from __future__ import annotations

import numpy as np

__all__ = ["synthetic_var_dyad", "northstar_freqs", "NORTHSTAR"]

NORTHSTAR = dict(m=64, fs=500.0, window=1000, hop=500, p=8, F=256, T=300_000)


def northstar_freqs(F: int = 256):
    """0.5 Hz grid 0.5 .. 128 Hz (the reference's default 0.5-Hz step, mtmvar.py:1014)."""
    return 0.5 * np.arange(1, F + 1)

def synthetic_var_dyad(dyad: int, m: int = 64, p: int = 8, T: int = 300_000, fs: float = 500.0,
                       burn: int = 2000, density: float = 0.05, coupling: float = 0.05,
                       target_radius: float = 0.95):
    """Seeded stable VAR(p) recording used by bench and parity tests.

    Diagonal AR(2) resonators (f0 ~ U(4, 40) Hz, r ~ U(0.80, 0.95)), sparse N(0,1)*coupling
    off-diagonal terms on every lag, companion spectral radius rescaled to `target_radius`
    (A_k <- A_k * g**k), unit-variance innovations, burn-in discarded, each channel z-scored.
    This is the build's own workload generator, not reference code.
    """
    rng = np.random.default_rng(1234 + dyad)
    A = np.zeros((p, m, m))
    f0 = rng.uniform(4.0, 40.0, m)
    r = rng.uniform(0.80, 0.95, m)
    idx = np.arange(m)
    A[0, idx, idx] = 2 * r * np.cos(2 * np.pi * f0 / fs)
    if p > 1:
        A[1, idx, idx] = -r ** 2
    mask = rng.random((p, m, m)) < density
    off = coupling * rng.standard_normal((p, m, m)) * mask
    off[:, idx, idx] = 0.0
    A += off
    comp = np.zeros((m * p, m * p))
    comp[:m, :] = np.concatenate(list(A), axis=1)
    comp[m:, :-m] = np.eye(m * (p - 1))
    rho = np.max(np.abs(np.linalg.eigvals(comp)))
    g = min(1.0, target_radius / rho)
    A = A * (g ** np.arange(1, p + 1))[:, None, None]
    n_tot = T + burn
    e = rng.standard_normal((n_tot, m))
    x = np.zeros((n_tot, m))
    # x[t] = e[t] + [x[t-1], ..., x[t-p]] @ W,  W = vstack_k A_k^T  (one (p*m) x m mat-vec per sample)
    W = np.ascontiguousarray(np.concatenate([A[k].T for k in range(p)], axis=0))
    for t in range(p, n_tot):
        x[t] = e[t] + x[t - p:t][::-1].reshape(-1) @ W
    x = x[burn:].T
    x = (x - x.mean(axis=1, keepdims=True)) / x.std(axis=1, keepdims=True)
    return np.ascontiguousarray(x)
