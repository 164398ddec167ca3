"""NetCDF front-end feeding the MVAR engine: the semantics of the reference's `load_eeg_signals`
(/root/reference/src/mne_bridge.py:113-223) -- Butterworth-4 zero-phase high/low-pass, 50 Hz notch
(Q = 15) when 50 Hz is below Nyquist, trim to [0, event_duration], drop the M1/M2 mastoids, optional channel
subset, per-channel z-score -- split into a pure-array function (`preprocess_eeg`, testable without
xarray) and thin file readers (xarray imported lazily; the reference's files are NETCDF4_CLASSIC written by
`DataArray.to_netcdf`, /root/reference/src/export.py:606).

Host-side O(n) DSP with the same SciPy calls as the reference; this is SURVEY.md section 8(f) row 2 (the
caller side of the hot path), not GPU work.
"""
from __future__ import annotations

import numpy as np

__all__ = ["filter_eeg", "preprocess_eeg", "load_eeg_signals", "load_dyad_block"]


def filter_eeg(data_tc, fs, low_cutoff_hz=None, high_cutoff_hz=None):
    """The filter sequence of `load_eeg_signals` (mne_bridge.py:152-183) on (time, channel) samples: zero-phase
    Butterworth-4 high-pass, low-pass, then the 50 Hz notch (Q = 15) when 50 Hz is below Nyquist."""
    from scipy.signal import butter, filtfilt, iirnotch
    # filtered channel by channel along a CONTIGUOUS time axis: the same recurrences on the same numbers as
    # filtfilt(..., axis=0) on the (time, channel) array -- bit-identical -- at half the time (110 001 x 34: 0.56 -> 0.28 s
    # per filter), which is what bounds a config-5 batch once the GPU work is out of the way
    x = np.ascontiguousarray(np.asarray(data_tc, dtype=np.float64).T)
    nyq = fs / 2.0
    for cutoff, kind, label in ((low_cutoff_hz, "highpass", "low_cutoff_hz"), (high_cutoff_hz, "lowpass", "high_cutoff_hz")):
        if cutoff is None:
            continue
        wn = float(cutoff) / nyq
        if not 0.0 < wn < 1.0:
            raise ValueError(f"Invalid {label}={cutoff}. Must satisfy 0 < cutoff < {nyq:.3f} Hz.")
        b, a = butter(4, wn, btype=kind)
        x = filtfilt(b, a, x, axis=-1)
    if 50.0 < nyq:
        b, a = iirnotch(50.0, Q=15, fs=fs)
        x = filtfilt(b, a, x, axis=-1)
    return x.T


def preprocess_eeg(data_tc, time_s, channel_names, fs, event_duration_s=None, channel_subset=None,
                   low_cutoff_hz=None, high_cutoff_hz=None, source="<array>"):
    """(time, channel) samples -> (signals (n_chan, n_samp) z-scored, names, time_s trimmed)."""
    x = filter_eeg(data_tc, fs, low_cutoff_hz, high_cutoff_hz)
    t = np.asarray(time_s, dtype=np.float64)
    names = [str(c) for c in channel_names]
    if event_duration_s is None:
        event_duration_s = float(t[-1])
    keep_t = (t >= 0.0) & (t <= event_duration_s)
    x, t = x[keep_t], t[keep_t]
    keep_c = [k for k, c in enumerate(names) if c not in ("M1", "M2")]
    if channel_subset is not None:
        avail = [names[k] for k in keep_c]
        keep_c = [names.index(c) for c in channel_subset if c in avail]
        if not keep_c:
            raise ValueError(f"None of the requested channels {channel_subset} found in {source}. Available: {avail}")
    sig = np.ascontiguousarray(x[:, keep_c].T)
    sd = np.std(sig, axis=1, keepdims=True)
    sd[sd == 0] = 1.0
    sig = (sig - np.mean(sig, axis=1, keepdims=True)) / sd
    return sig, [names[k] for k in keep_c], t


def load_eeg_signals(ncdf_path, channel_subset=None, low_cutoff_hz=None, high_cutoff_hz=None):
    """(signals, channel_names, fs, time_s, event_duration_s) of one exported EEG NetCDF file."""
    try:
        import xarray as xr
    except ImportError as e:  # pragma: no cover
        raise ImportError("load_eeg_signals needs xarray + netCDF4 to read the reference's .nc files") from e
    da = xr.open_dataarray(ncdf_path)
    try:
        if "time" not in da.dims or "channel" not in da.dims:
            raise ValueError(f"Expected 'time' and 'channel' dimensions in {ncdf_path}, got {da.dims}")
        fs = float(da.attrs.get("sampling_freq", da.attrs.get("sampling_frequency_Hz", 128.0)))
        t = da.coords["time"].values
        dur = float(da.attrs.get("event_duration", t[-1]))
        sig, names, t = preprocess_eeg(da.transpose("time", "channel").values, t, da.coords["channel"].values, fs,
                                       dur, channel_subset, low_cutoff_hz, high_cutoff_hz, source=str(ncdf_path))
    finally:
        da.close()
    return sig, names, fs, t, dur


def load_dyad_block(child_path, caregiver_path, **kw):
    """np.vstack([child, caregiver]) on their common length: the (2 x n_chan, T) block the sliding-window
    engine takes (BASELINE.json config 2); returns (block, names, fs)."""
    a, na, fs_a, _, _ = load_eeg_signals(child_path, **kw)
    b, nb, fs_b, _, _ = load_eeg_signals(caregiver_path, **kw)
    if fs_a != fs_b:
        raise ValueError(f"sampling rates differ: {fs_a} vs {fs_b}")
    T = min(a.shape[1], b.shape[1])
    return np.vstack([a[:, :T], b[:, :T]]), [f"ch_{c}" for c in na] + [f"cg_{c}" for c in nb], fs_a
