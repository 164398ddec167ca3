"""CPU oracle for the MVAR / ffDTF hot path  --  TEST INFRASTRUCTURE ONLY.

This file restates, in plain NumPy, the arithmetic of the reference hot path
(`/root/reference/src/mtmvar.py:35-284, 551-601` and the window generator
`/root/reference/src/eeg_alpha_ibi_ffdtf.py:451-518`).  It is the *checker* for the
HIP kernels: only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import it.  The product package never imports anything from `oracle/`; it fails
loudly when the HIP library is missing instead of falling back to this code.

Pinning: every function here is checked against outputs of the reference itself
(imported in the build container by `tests/golden/make_golden.py`, which wrote the
fixtures under `tests/golden/*.npz`) in `tests/test_oracle_golden.py`.

Two flavours are provided where it matters for timing:
  * `*_loop`  -- the reference's own loop structure (per-frequency Python loop, per-(i,j)
                 normalisation loop).  This is what `bench.py` times as the CPU baseline
                 (`cpu_baseline.kind = "port"`), because the reference source itself
                 cannot travel to the GPU box.
  * vectorised -- identical math, batched LAPACK calls; used by the parity tests so they
                 finish in seconds.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "lag_covariances", "count_corr", "ar_coeff", "mvar_transfer_function",
    "mvar_transfer_function_loop", "dtf_multivariate", "full_freq_dtf",
    "full_freq_dtf_loop", "multivariate_spectra", "mvar_criterion",
    "window_positions", "create_windows", "sliding_ffdtf",
]


# --------------------------------------------------------------------------- a1
def lag_covariances(x: np.ndarray, p: int) -> np.ndarray:
    """R_l = X[:, :n-l] @ X[:, l:].T / n for l = 0..p (biased, not demeaned).

    Follows mtmvar.py:57-59 (lags 1..p, `corr_scale = 1/n`, `nn = n-k-1`) and
    mtmvar.py:72-73 (lag 0).  3-D input `(m, n, trials)` is averaged over trials
    (mtmvar.py:78-85).  Returns `(p+1, m, m)`.
    """
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 2:
        x = x[:, :, None]
    m, n, trials = x.shape
    R = np.zeros((p + 1, m, m))
    for t in range(trials):
        xt = x[:, :, t]
        for l in range(p + 1):
            R[l] += xt[:, : n - l] @ xt[:, l:].T * (1.0 / n)
    if trials > 1:
        R /= trials
    return R


def count_corr(x: np.ndarray, ip: int, iwhat: int = 1):
    """Block-Toeplitz normal equations (mtmvar.py:35-87), `iwhat == 1` only (Q8).

    r_left block (a, b) = R_{a-b} (a > b), R_{b-a}.T (a < b), R_0 (a == b);
    r_right block k = R_{k+1}; r = R_0.
    """
    if iwhat != 1:
        raise NotImplementedError("only the biased estimator (iwhat=1) is on the hot path")
    R = lag_covariances(x, ip)
    m = R.shape[1]
    r_left = np.zeros((m * ip, m * ip))
    r_right = np.zeros((m * ip, m))
    for a in range(ip):
        r_right[a * m:(a + 1) * m] = R[a + 1]
        for b in range(ip):
            blk = R[a - b] if a >= b else R[b - a].T
            r_left[a * m:(a + 1) * m, b * m:(b + 1) * m] = blk
    return r_left, r_right, R[0].copy()


# --------------------------------------------------------------------------- a2
def ar_coeff(data: np.ndarray, model_order: int = 5):
    """Yule-Walker fit (mtmvar.py:90-123): returns `(m, m, p)` coefficients and `(m, m)` V."""
    data = np.asarray(data, dtype=np.float64)
    if data.ndim < 3:
        data = data[:, :, None]
    m = data.shape[0]
    r_left, r_right, r_zero = count_corr(data, model_order, 1)
    x = np.linalg.solve(r_left, r_right).T                    # mtmvar.py:116
    variance = r_zero - x.dot(r_right)                        # mtmvar.py:119
    ar = x.reshape(m, model_order, m).transpose((0, 2, 1))    # mtmvar.py:122
    return ar, variance


# --------------------------------------------------------------------------- a3
def _twiddles(p: int, freqs: np.ndarray, fs: float) -> np.ndarray:
    """z[k, f] = exp(-(k+1) * 2*pi*1j * freqs / fs), same operation order as mtmvar.py:153."""
    freqs = np.asarray(freqs)
    z = np.zeros((p, len(freqs)), dtype=complex)
    for k in range(1, p + 1):
        z[k - 1, :] = np.exp(-k * 2 * np.pi * 1j * freqs / fs)
    return z


def mvar_transfer_function_loop(ar_coeffs, freqs, fs):
    """Reference loop structure (mtmvar.py:126-162): one `np.linalg.inv` per frequency."""
    p = ar_coeffs.shape[2]
    F = len(freqs)
    m = ar_coeffs.shape[0]
    H = np.zeros((m, m, F), dtype=complex)
    A = np.zeros((m, m, F), dtype=complex)
    z = _twiddles(p, freqs, fs)
    for fi in range(F):
        a = np.eye(m, dtype=complex)
        for k in range(p):
            a -= ar_coeffs[:, :, k] * z[k, fi].item()
        H[:, :, fi] = np.linalg.inv(a)
        A[:, :, fi] = a
    return H, A


def mvar_transfer_function(ar_coeffs, freqs, fs):
    """Vectorised form of mtmvar.py:126-162 (batched inverse over frequencies).

    The accumulation `a -= ar[:, :, k] * z[k]` is kept in lag order so the rounding of
    A(f) is identical to the loop form.
    """
    p = ar_coeffs.shape[2]
    m = ar_coeffs.shape[0]
    F = len(freqs)
    z = _twiddles(p, freqs, fs)
    A = np.broadcast_to(np.eye(m, dtype=complex)[:, :, None], (m, m, F)).copy()
    for k in range(p):
        A -= ar_coeffs[:, :, k, None] * z[k][None, None, :]
    H = np.linalg.inv(A.transpose(2, 0, 1)).transpose(1, 2, 0)
    return np.ascontiguousarray(H), A


# --------------------------------------------------------------------------- a4 / a5
def dtf_multivariate(signals, freqs, fs, optimal_model_order):
    """|H|^2, un-normalised (mtmvar.py:204-234, quirk Q2).  Order must be given."""
    ar, _ = ar_coeff(signals, optimal_model_order)
    H, _ = mvar_transfer_function(ar, freqs, fs)
    return np.abs(H) ** 2


def full_freq_dtf(signals, freqs, fs, optimal_model_order):
    """ffDTF (mtmvar.py:237-284): ff[i,j,f] = dtf[i,j,f] / sum_{j',f'} dtf[i,j',f']."""
    dtf = dtf_multivariate(signals, freqs, fs, optimal_model_order)
    den = dtf.sum(axis=(1, 2))
    return dtf / den[:, None, None]


def full_freq_dtf_loop(signals, freqs, fs, optimal_model_order):
    """Same result with the reference's loop structure; this is the timed CPU baseline."""
    ar, _ = ar_coeff(signals, optimal_model_order)
    H, _ = mvar_transfer_function_loop(ar, freqs, fs)
    dtf = np.abs(H) ** 2
    m, _, F = dtf.shape
    ff = np.zeros((m, m, F))
    for i in range(m):                       # mtmvar.py:281-283
        for j in range(m):
            ff[i, j, :] = dtf[i, j, :] / np.sum(dtf[i, :, :])
    return ff


# --------------------------------------------------------------------------- a6
def multivariate_spectra(signals, freqs, fs, optimal_model_order):
    """S(f) = H V H.T -- plain transpose, no conjugate (mtmvar.py:199, quirk Q3)."""
    ar, V = ar_coeff(signals, optimal_model_order)
    H, _ = mvar_transfer_function(ar, freqs, fs)
    Hf = H.transpose(2, 0, 1)                                  # (F, m, m)
    S = Hf @ (V[None] @ Hf.transpose(0, 2, 1))
    return np.ascontiguousarray(S.transpose(1, 2, 0))


# --------------------------------------------------------------------------- a7
def partial_coherence(spectra):
    """kappa_ij = M_ij / sqrt(M_ii M_jj), M_ij = det(S without row i and column j) (no cofactor sign), 1 on the
    diagonal, 0 where the denominator is zero (reference: src/mtmvar.py:287-338, minors :302-322, ratio :324-336).
    Restated with the same minors-by-determinant arithmetic (vectorised over frequency)."""
    S = np.asarray(spectra, dtype=np.complex128)
    n, _, F = S.shape
    minors = np.ones((n, n, F), dtype=np.complex128)
    if n > 1:
        St = np.ascontiguousarray(S.transpose(2, 0, 1))
        for i in range(n):
            rows = [r for r in range(n) if r != i]
            for j in range(n):
                cols = [c for c in range(n) if c != j]
                minors[i, j] = np.linalg.det(St[:, rows][:, :, cols])
    kappa = np.zeros((n, n, F), dtype=np.complex128)
    for i in range(n):
        for j in range(n):
            if i == j:
                kappa[i, j] = 1.0
            else:
                den = np.sqrt(minors[i, i] * minors[j, j])
                with np.errstate(divide="ignore", invalid="ignore"):
                    kappa[i, j] = np.where(den != 0, minors[i, j] / den, 0)
    return kappa


def direct_dtf(signals, freqs, fs, optimal_model_order):
    """dDTF = ffDTF * |kappa| (src/mtmvar.py:341-385, product at :383)."""
    S = multivariate_spectra(signals, freqs, fs, optimal_model_order)
    return full_freq_dtf(signals, freqs, fs, optimal_model_order) * np.abs(partial_coherence(S))


def gen_partial_directed_coherence(signals, freqs, fs, optimal_model_order):
    """GPDC_ij = (|A_ij| / sigma_i) / sqrt(sum_k |A_kj|^2 / sigma_k^2) (src/mtmvar.py:388-468, loops :452-466)."""
    ar, V = ar_coeff(signals, optimal_model_order)
    _, A = mvar_transfer_function(ar, freqs, fs)
    s2 = np.diag(V)
    num = np.abs(A) / np.sqrt(s2)[:, None, None]
    den = np.sqrt(np.sum(np.abs(A) ** 2 / s2[:, None, None], axis=0))          # (j, f)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(den[None] != 0, num / den[None], 0.0)


def mvar_criterion(data, max_model_order, crit_type="AIC"):
    """AIC / HQ / SC curves (mtmvar.py:551-601, quirk Q7).  Returns (crit, range, p_opt)."""
    m, n = data.shape
    rng = np.arange(1, max_model_order + 1, dtype=int)
    crit = np.zeros(max_model_order)
    for p in rng:
        _, V = ar_coeff(data, int(p))
        if crit_type == "AIC":
            pen = 2 * p * m ** 2 / n
        elif crit_type == "HQ":
            pen = 2 * np.log(np.log(n)) * p * m ** 2 / n
        elif crit_type == "SC":
            pen = np.log(n) * p * m ** 2 / n
        else:
            raise ValueError("Invalid criterion type. Choose from 'AIC', 'HQ', 'SC'.")
        crit[p - 1] = np.log(np.linalg.det(V)) + pen
    return crit, rng, rng[np.argmin(crit)]


# --------------------------------------------------------------------------- a8
def window_positions(T: int, n_windows: int = 3, window_size=None):
    """Start positions of `_create_windows` (eeg_alpha_ibi_ffdtf.py:451-518), same errors."""
    if window_size is None:
        if T % n_windows != 0:
            raise ValueError(
                f"Cannot evenly divide signal of length {T} into {n_windows} "
                f"non-overlapping windows. Provide a specific window_size.")
        window_size = T // n_windows
    else:
        min_required = (T + n_windows - 1) // n_windows
        if window_size < min_required:
            raise ValueError(
                f"window_size={window_size} is too short. To cover {T} samples with "
                f"{n_windows} windows without leaving gaps, the minimum window_size is {min_required}.")
        if window_size > T:
            raise ValueError(f"window_size ({window_size}) cannot exceed signal length ({T}).")
    max_start = T - window_size
    if max_start < n_windows - 1 and n_windows > 1:
        raise ValueError(
            f"window_size={window_size} is too large to generate {n_windows} "
            f"distinct windows. Decrease window_size or n_windows.")
    if n_windows == 1:
        pos = np.array([0], dtype=int)
    else:
        pos = np.linspace(0, max_start, n_windows, dtype=int)
    return pos, int(window_size)


def create_windows(signals, n_windows=3, window_size=None):
    pos, w = window_positions(signals.shape[1], n_windows, window_size)
    return [signals[:, s:s + w] for s in pos]


def sliding_ffdtf(x, window_size, n_windows, p, freqs, fs, loop=False):
    """ffDTF of every window of one recording `(m, T)`; returns `(n_windows, m, m, F)`."""
    fn = full_freq_dtf_loop if loop else full_freq_dtf
    return np.stack([fn(w, freqs, fs, p) for w in create_windows(x, n_windows, window_size)])
