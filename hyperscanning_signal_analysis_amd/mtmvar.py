"""Drop-in replacements for the MVAR / DTF functions of the reference's `src/mtmvar.py`.

Same names, parameters, defaults, return shapes, printed messages and error behaviour as
/root/reference/src/mtmvar.py:35-284 and :551-601; the arithmetic runs on the MI355X through
libhypermvar.so (see `engine.py`).  NumPy arrays in, freshly allocated NumPy arrays out.

Differences that are deliberate and documented in DESIGN.md:
  * `count_corr(..., iwhat=2)` (never used by the reference, quirk Q8) is not provided;
  * `mvar_criterion` gets all orders 1..pmax from ONE factorisation at pmax (the block LDL^T partial
    sums are exactly the lower-order residual covariances) instead of pmax separate fits;
  * `multivariate_spectra` and `full_freq_dtf` share nothing in the reference (each refits the model);
    here each call still fits once, but `mvar_analysis()` returns both from a single fit.
Supported sizes: 1..64 channels, model order 1..32.
"""
from __future__ import annotations

import numpy as np
import torch

from .engine import Engine, SingularMatrixError, default_engine

__all__ = ["count_corr", "ar_coeff", "mvar_transfer_function", "multivariate_spectra", "dtf_multivariate",
           "full_freq_dtf", "mvar_criterion", "mvar_analysis", "lag_covariances", "compute_and_plot_mvar",
           "mvar_plot"]


# ----------------------------------------------------------------------------- internals
def _as_trials(data, eng: Engine):
    """(m, n) or (m, n, trials) ndarray -> device tensor (trials, m, n)."""
    data = np.asarray(data, dtype=np.float64)
    if data.ndim == 2:
        data = data[:, :, None]
    if data.ndim != 3:
        raise ValueError("signals must have shape (channels, samples) or (channels, samples, trials)")
    x = torch.as_tensor(np.ascontiguousarray(data.transpose(2, 0, 1))).to(eng.device)
    return x


def _lagcov_mean(data, p: int, eng: Engine):
    """Trial-averaged lag covariances in MP layout: (1, p+1, MP, MP).  mtmvar.py:54-85."""
    x = _as_trials(data, eng)
    trials, m, n = x.shape
    if n <= p:
        raise ValueError(f"need more samples ({n}) than the model order ({p})")
    rec = torch.arange(trials, dtype=torch.int64, device=eng.device)
    start = torch.zeros(trials, dtype=torch.int64, device=eng.device)
    R = eng.lagcov(x, rec, start, n, p)
    if trials > 1:
        R = eng.trial_mean(R, m)                          # mtmvar.py:78-85
    return R, m, n


def _fit(data, p: int, eng: Engine, want_logdet=False):
    R, m, n = _lagcov_mean(data, p, eng)
    ar, V, logdet, info = eng.yw_solve(R, m, want_logdet)
    eng.raise_on_info(info, "ar_coeff")
    return ar, V, logdet, m, n


def _order(signals, max_model_order, optimal_model_order, crit_type, plot, comment, style):
    if optimal_model_order is None:
        _, _, optimal_model_order = mvar_criterion(signals, max_model_order, crit_type, plot)
        if style == "spectra":
            print('Optimal model order for all channels: p = ', str(optimal_model_order))
        else:
            comment_str = '' if comment is None else comment + ' '
            print(f'Optimal model order for all {comment_str}channels: p = {optimal_model_order}')
    else:
        if style == "spectra":
            print('Using provided model order: p = ', str(optimal_model_order))
        else:
            print(f'Using provided model order: p = {optimal_model_order}')
    return int(optimal_model_order)


# ----------------------------------------------------------------------------- public API
def lag_covariances(x, p, engine: Engine | None = None):
    """R_l = X[:, :n-l] X[:, l:].T / n, l = 0..p, trial-averaged; (p+1, m, m).  (mtmvar.py:57-59,72-73)"""
    eng = engine or default_engine()
    R, m, _ = _lagcov_mean(x, int(p), eng)
    return R[0, :, :m, :m].cpu().numpy()


def count_corr(x, ip, iwhat):
    """Block-Toeplitz normal equations (mtmvar.py:35-87): returns (r_left, r_right, r)."""
    if iwhat != 1:
        raise NotImplementedError("only the biased estimator (iwhat=1) is implemented (the reference never uses 2)")
    R = lag_covariances(x, ip)
    m = R.shape[1]
    r_left = np.zeros((m * ip, m * ip))
    r_right = np.zeros((m * ip, m))
    for a in range(ip):
        r_right[a * m:(a + 1) * m] = R[a + 1]
        for b in range(ip):
            r_left[a * m:(a + 1) * m, b * m:(b + 1) * m] = R[a - b] if a >= b else R[b - a].T
    return r_left, r_right, R[0].copy()


def ar_coeff(data, model_order=5):
    """MVAR coefficients (channels, channels, model_order) and residual covariance (mtmvar.py:90-123)."""
    eng = default_engine()
    ar, V, _, m, _ = _fit(data, int(model_order), eng)
    return ar[0, :m, :m, :].cpu().numpy(), V[0, :m, :m].cpu().numpy()


def mvar_transfer_function(ar_coeffs, freqs, fs):
    """H(f) = inv(I - sum_k A_k exp(-2 pi i f k / fs)) and A(f); both (chan, chan, len(freqs)) complex.

    mtmvar.py:126-162.  Raises numpy.linalg.LinAlgError('Singular matrix') like np.linalg.inv.
    """
    eng = default_engine()
    ar_coeffs = np.asarray(ar_coeffs, dtype=np.float64)
    m, _, p = ar_coeffs.shape
    mp = eng.pad(m)
    arp = torch.zeros(1, mp, mp, p, dtype=torch.float64, device=eng.device)
    arp[0, :m, :m, :] = torch.as_tensor(ar_coeffs).to(eng.device)
    tw = eng.twiddles(freqs, fs, p)
    out = eng.transfer(arp, m, tw, want_P=False, want_H=True, want_A=True)
    eng.raise_on_info(out["info"], "mvar_transfer_function (inverse of A(f))", per_item=len(np.atleast_1d(freqs)))
    H = eng.to_mmf_complex(out["H"], m)[0].cpu().numpy()
    A = eng.to_mmf_complex(out["A"], m)[0].cpu().numpy()
    return H, A


def mvar_analysis(signals, freqs, fs, model_order, want=("ffdtf", "spectra"), engine: Engine | None = None):
    """One fit, several products: any of 'ar', 'V', 'H', 'A', 'dtf', 'ffdtf', 'spectra', 'pcoh', 'ddtf', 'gpdc'
    as a dict.  `engine`: run on this Engine instead of the process-wide default (extra keyword, not in the reference).

    Not in the reference (which refits for every product, mtmvar.py:165-284); this is what the pipeline
    mirror uses so that ffDTF and spectra share the lag covariances, the solve and the inverses.
    """
    eng = engine or default_engine()
    p = int(model_order)
    ar, V, _, m, _ = _fit(signals, p, eng)
    tw = eng.twiddles(freqs, fs, p)
    need_S = any(k in want for k in ("spectra", "pcoh", "ddtf"))
    need_H = ("H" in want) or need_S
    need_P = any(k in want for k in ("dtf", "ffdtf", "ddtf"))
    t = eng.transfer(ar, m, tw, want_P=need_P, want_H=need_H, want_A=any(k in want for k in ("A", "gpdc")))
    eng.raise_on_info(t["info"], "mvar_transfer_function (inverse of A(f))", per_item=len(np.atleast_1d(freqs)))
    res = {}
    if "ar" in want:
        res["ar"] = ar[0, :m, :m, :].cpu().numpy()
    if "V" in want:
        res["V"] = V[0, :m, :m].cpu().numpy()
    if "H" in want:
        res["H"] = eng.to_mmf_complex(t["H"], m)[0].cpu().numpy()
    if "A" in want:
        res["A"] = eng.to_mmf_complex(t["A"], m)[0].cpu().numpy()
    if "dtf" in want:
        res["dtf"] = eng.normalise(t["P"], t["rowsum"], m, normalise=False)[0][0].cpu().numpy()
    ff = None
    if "ffdtf" in want or "ddtf" in want:
        ff = eng.normalise(t["P"], t["rowsum"], m, normalise=True)[0]
    if "ffdtf" in want:
        res["ffdtf"] = ff[0].cpu().numpy()
    if need_S:
        S = eng.spectra(t["H"], V, m)
        if "spectra" in want:
            res["spectra"] = eng.to_mmf_complex(S, m)[0].cpu().numpy()
        if "pcoh" in want or "ddtf" in want:
            kappa, info = eng.partial_coherence(S, m)
            eng.raise_on_info(info, "partial_coherence")
            if "pcoh" in want:
                res["pcoh"] = kappa[0].cpu().numpy()
            if "ddtf" in want:
                res["ddtf"] = eng.ddtf(ff, kappa)[0].cpu().numpy()
    if "gpdc" in want:
        res["gpdc"] = eng.gpdc(t["A"], V, m)[0].cpu().numpy()
    return res


def multivariate_spectra(signals, freqs, fs, max_model_order=20, optimal_model_order=None, crit_type='AIC'):
    """S(f) = H V H.T (plain transpose, mtmvar.py:199); (N_chan, N_chan, N_f) complex128."""
    p = _order(signals, max_model_order, optimal_model_order, crit_type, True, None, "spectra")
    return mvar_analysis(signals, np.asarray(freqs), fs, p, want=("spectra",))["spectra"]


def dtf_multivariate(signals, freqs, fs, max_model_order=20, optimal_model_order=None, crit_type='AIC', comment=None):
    """Un-normalised |H|^2 (mtmvar.py:204-234, quirk Q2); (N_chan, N_chan, N_f) float64."""
    p = _order(signals, max_model_order, optimal_model_order, crit_type, False, comment, "dtf")
    return mvar_analysis(signals, np.asarray(freqs), fs, p, want=("dtf",))["dtf"]


def full_freq_dtf(signals, freqs, fs, max_model_order=20, optimal_model_order=None, crit_type='AIC'):
    """ffDTF_ij(f) = |H_ij(f)|^2 / sum_f sum_k |H_ik(f)|^2 (mtmvar.py:237-284)."""
    p = _order(signals, max_model_order, optimal_model_order, crit_type, False, None, "dtf")
    return mvar_analysis(signals, np.asarray(freqs), fs, p, want=("ffdtf",))["ffdtf"]


def partial_coherence(spectra):
    """kappa_ij = M_ij / sqrt(M_ii M_jj) with M the minors of the spectral matrix (mtmvar.py:287-338), computed
    from the inverse: M_ij = (-1)^(i+j) det(S) (S^-1)_ji.  (N_chan, N_chan, N_f) complex128 in and out."""
    import torch
    eng = default_engine()
    S = np.ascontiguousarray(np.asarray(spectra, dtype=np.complex128))
    n_chan = S.shape[0]
    if S.ndim != 3 or S.shape[1] != n_chan:
        raise ValueError("spectra must have shape (N_chan, N_chan, N_f)")
    eng.pad(n_chan)
    Sk = eng.pack_complex(torch.from_numpy(S)[None].to(eng.device))
    kappa, info = eng.partial_coherence(Sk, n_chan)
    eng.raise_on_info(info, "partial_coherence")
    return kappa[0].cpu().numpy()


def direct_dtf(signals, freqs, fs, max_model_order=20, optimal_model_order=None, crit_type='AIC'):
    """dDTF = ffDTF * |partial coherence| (mtmvar.py:341-385); one fit instead of the reference's two."""
    p = _order(signals, max_model_order, optimal_model_order, crit_type, True, None, "spectra")
    _order(signals, max_model_order, p if optimal_model_order is None else optimal_model_order, crit_type, False,
           None, "dtf")
    return mvar_analysis(signals, np.asarray(freqs), fs, p, want=("ddtf",))["ddtf"]


def gen_partial_directed_coherence(signals, freqs, fs, max_model_order=20, optimal_model_order=None,
                                   crit_type='AIC'):
    """GPDC_ij(f) = |A_ij|/sigma_i / sqrt(sum_k |A_kj|^2/sigma_k^2) (mtmvar.py:388-468)."""
    p = _order(signals, max_model_order, optimal_model_order, crit_type, False, None, "spectra")
    return mvar_analysis(signals, np.asarray(freqs), fs, p, want=("gpdc",))["gpdc"]


def mvar_criterion(data, max_model_order, crit_type='AIC', plot=False, engine: Engine | None = None):
    """AIC / HQ / SC over p = 1..max_model_order (mtmvar.py:551-601): (crit, p_range, optimal_order)."""
    data = np.asarray(data, dtype=np.float64)
    n_channels, n_samples = data.shape                       # 2-D only, like the reference
    model_order_range = np.arange(1, max_model_order + 1, dtype=int)
    if crit_type == 'AIC':
        pen = 2 * model_order_range * n_channels ** 2 / n_samples
    elif crit_type == 'HQ':
        pen = 2 * np.log(np.log(n_samples)) * model_order_range * n_channels ** 2 / n_samples
    elif crit_type == 'SC':
        pen = np.log(n_samples) * model_order_range * n_channels ** 2 / n_samples
    else:
        raise ValueError("Invalid criterion type. Choose from 'AIC', 'HQ', 'SC'.")
    eng = engine or default_engine()
    _, _, logdet, _, _ = _fit(data, int(max_model_order), eng, want_logdet=True)
    crit = logdet[0].cpu().numpy() + pen
    best = int(np.argmin(crit))                     # first minimum, like the reference's argmin (Q7)
    p_opt = model_order_range[best]
    if plot:
        _criterion_figure(model_order_range, crit, best, crit_type)
    return crit, model_order_range, p_opt


def _criterion_figure(orders, crit, best, crit_type):
    """The figure `mvar_criterion(..., plot=True)` pops up: criterion against model order, the minimum marked."""
    import matplotlib.pyplot as plt
    fig, ax = plt.subplots()
    ax.plot(orders, crit, "o-")
    ax.plot([orders[best]], [crit[best]], "ro")
    ax.set(xlabel="Model order p", ylabel=f"{crit_type} criterion",
           title=f"MVAR Model Order Selection ({crit_type}). The best order = {orders[best]}")
    ax.grid(True)
    plt.show()
    return fig


def mvar_plot(spectra, ff_dtf, freqs, chan_names=None, top_title=""):
    """Minimal m x m grid (|S_ii| on the diagonal, ffDTF j -> i elsewhere); own code, matplotlib only."""
    import matplotlib.pyplot as plt
    m = ff_dtf.shape[0]
    names = list(chan_names) if chan_names is not None else [str(k) for k in range(m)]
    fig, axs = plt.subplots(m, m, figsize=(10, 10), squeeze=False)
    top = float(ff_dtf[~np.eye(m, dtype=bool)].max()) if m > 1 else 1.0
    for i in range(m):
        for j in range(m):
            ax = axs[i, j]
            if i == j:
                ax.plot(freqs, np.abs(spectra[i, i, :]), color="k")
            else:
                ax.fill_between(freqs, ff_dtf[i, j, :], color="C0")
                ax.set_ylim(0, top)
            ax.tick_params(labelsize=5)
            if i == 0:
                ax.set_title(names[j], fontsize=7)
            if j == 0:
                ax.set_ylabel(names[i], fontsize=7)
    fig.suptitle(top_title, fontsize=9)
    return fig


def compute_and_plot_mvar(ncdf_path, channel_subset=None, max_model_order=20, optimal_model_order=None,
                          crit_type="AIC", freq_min=1.0, freq_max=40.0, freq_step=0.5, low_cutoff_hz=None,
                          high_cutoff_hz=None, plot=True, plot_loaded_signal=False, loaded_signal_max_channels=19,
                          loaded_signal_spacing=8.0, loaded_signal_figsize=(16.0, 9.0), loader=None):
    """Load one EEG NetCDF file, compute ffDTF and multivariate spectra, optionally plot (mtmvar.py:1006-1128).

    Returns (ff_dtf, spectra, chan_names, crit, model_order_range, p_opt) like the reference.  The
    reference imports its loader from a module that does not define it; here it is wired
    to `eeg_io.load_eeg_signals` (same semantics as `src/mne_bridge.py:113-223`), or to `loader=` if given.
    One fit feeds both products.
    """
    if loader is None:
        from .eeg_io import load_eeg_signals as loader
    signals, chan_names, fs, time_s, event_duration_s = loader(
        ncdf_path, channel_subset=channel_subset, low_cutoff_hz=low_cutoff_hz, high_cutoff_hz=high_cutoff_hz)
    stem = ncdf_path.stem if hasattr(ncdf_path, "stem") else ncdf_path
    if plot_loaded_signal:
        import matplotlib.pyplot as plt
        k = min(loaded_signal_max_channels, signals.shape[0])
        plt.figure(figsize=loaded_signal_figsize)
        for c in range(k):
            plt.plot(time_s, signals[c] - c * loaded_signal_spacing, lw=0.6)
        plt.yticks([-c * loaded_signal_spacing for c in range(k)], chan_names[:k])
        plt.axvline(event_duration_s, color="r", lw=0.8)
        plt.title(f"Loaded EEG signal — {stem}")
        plt.show()
    freqs = np.arange(freq_min, freq_max + freq_step, freq_step)
    print(f"\n  Channels : {chan_names}")
    print(f"  Signals  : {signals.shape}  fs={fs} Hz")
    if optimal_model_order is None:
        crit, model_order_range, p_opt = mvar_criterion(signals, max_model_order, crit_type, plot=False)
        print(f"  {crit_type} optimal model order: p = {p_opt}")
    else:
        p_opt = optimal_model_order
        print(f"  Using fixed model order: p = {p_opt}")
        crit = np.array([])
        model_order_range = np.array([])
    print("  Computing ffDTF ...")
    print("  Computing multivariate spectra ...")
    res = mvar_analysis(signals, freqs, fs, int(p_opt), want=("ffdtf", "spectra"))
    ff_dtf, spectra = res["ffdtf"], res["spectra"]
    if plot:
        import matplotlib.pyplot as plt
        mvar_plot(spectra, ff_dtf, freqs, chan_names, top_title=f"MVAR Spectra and ff_DTF — {stem}")
        plt.suptitle(f"{stem}  (fs={fs:.0f} Hz, {signals.shape[0]} ch, {signals.shape[1]} samp)", fontsize=8, y=1.01)
        plt.tight_layout()
        plt.show()
    return ff_dtf, spectra, chan_names, crit, model_order_range, p_opt
