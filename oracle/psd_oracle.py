"""CPU restatement of the multitaper PSD used by the reference's `compute_psd_multitaper`
(/root/reference/src/psd.py:7-33, call site scripts/.../01_compute_psd.py:80 with fmin=1, fmax=30, bandwidth=2).

TEST INFRASTRUCTURE ONLY -- never imported by the product path.

PARITY UNPINNED.  The arithmetic lives in the third-party dependency mne==1.11.0
(`mne.time_frequency.psd_array_multitaper`, requirements.txt:11), which is not installed here and cannot be
fetched; the reference holds no test, golden vector or stored output for it.  This file restates the
published algorithm with mne's documented defaults (remove_dc=True, adaptive=False, low_bias=True,
normalization='length'):

    half_nbw = bandwidth * n_times / (2 sfreq);  K_max = int(2 half_nbw)
    tapers, eigenvalues = DPSS(n_times, half_nbw, K_max)   (scipy.signal.windows.dpss, norm=2, sym=False)
    low_bias: keep the tapers with eigenvalue > 0.9 (at least the best one)
    X_k = rfft((x - mean(x)) * taper_k), DC (and Nyquist for even n) scaled by 1/sqrt(2)
    psd = 2 / sum_k eig_k * sum_k eig_k |X_k|^2  on  fmin <= f <= fmax,  f = rfftfreq(n_times, 1/sfreq)
"""
from __future__ import annotations

import numpy as np
from scipy.signal.windows import dpss


def mt_params(n_times: int, sfreq: float, bandwidth: float, low_bias: bool = True):
    half_nbw = float(bandwidth) * n_times / (2.0 * sfreq)
    k_max = max(int(2 * half_nbw), 1)
    tapers, eig = dpss(n_times, half_nbw, k_max, sym=False, norm=2, return_ratios=True)
    tapers, eig = np.atleast_2d(tapers), np.atleast_1d(eig)
    if low_bias:
        idx = eig > 0.9
        if not idx.any():
            idx = np.zeros_like(idx)
            idx[np.argmax(eig)] = True
        tapers, eig = tapers[idx], eig[idx]
    return np.ascontiguousarray(tapers), eig


def psd_array_multitaper(data, sfreq, fmin=0.0, fmax=np.inf, bandwidth=None):
    x = np.asarray(data, dtype=np.float64)
    n_times = x.shape[-1]
    if bandwidth is None:
        bandwidth = 8.0 * sfreq / n_times               # mne's default: half_nbw = 4
    tapers, eig = mt_params(n_times, sfreq, bandwidth)
    freqs = np.fft.rfftfreq(n_times, 1.0 / sfreq)
    mask = (freqs >= fmin) & (freqs <= fmax)
    x = x - x.mean(axis=-1, keepdims=True)
    X = np.fft.rfft(x[:, None, :] * tapers[None], n=n_times)          # (ch, K, nfreq)
    X[..., 0] /= np.sqrt(2.0)
    if n_times % 2 == 0:
        X[..., -1] /= np.sqrt(2.0)
    w2 = eig[None, :, None]
    psd = (w2 * (X.real ** 2 + X.imag ** 2)).sum(axis=1) * (2.0 / eig.sum())
    return psd[:, mask], freqs[mask]


def compute_psd_multitaper(data, sfreq, fmin, fmax, bandwidth):
    """Same return order as the reference wrapper (src/psd.py:30-33): (freqs, psd)."""
    psd, freqs = psd_array_multitaper(data, sfreq, fmin, fmax, bandwidth)
    return freqs, psd
