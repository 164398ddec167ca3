"""`preprocess_eeg`: the array part of the reference's load_eeg_signals (src/mne_bridge.py:113-223)."""
import numpy as np
import pytest

from hyperscanning_signal_analysis_amd.eeg_io import preprocess_eeg


def make(fs=256.0, dur=20.0, pad=2.0):
    t = np.arange(-pad, dur + pad, 1.0 / fs)
    rng = np.random.default_rng(0)
    names = ["Fz", "Cz", "M1", "Pz", "M2", "Oz"]
    x = rng.standard_normal((t.size, len(names)))
    x[:, 1] += 4 * np.sin(2 * np.pi * 50 * t)          # mains on Cz
    x[:, 3] += 3.0                                       # offset on Pz
    return x, t, names, fs, dur


def test_trim_drop_zscore_and_notch():
    x, t, names, fs, dur = make()
    sig, ch, tt = preprocess_eeg(x, t, names, fs, event_duration_s=dur)
    assert ch == ["Fz", "Cz", "Pz", "Oz"]                                   # mastoids dropped
    assert tt[0] >= 0.0 and tt[-1] <= dur and sig.shape == (4, tt.size)     # margins trimmed
    assert np.allclose(sig.mean(axis=1), 0, atol=1e-12) and np.allclose(sig.std(axis=1), 1, atol=1e-12)
    spec = np.abs(np.fft.rfft(sig[1]))
    f = np.fft.rfftfreq(sig.shape[1], 1 / fs)
    assert spec[np.argmin(np.abs(f - 50))] < 0.2 * np.median(spec[(f > 20) & (f < 40)]) * 10   # 50 Hz line gone
    sig2, ch2, _ = preprocess_eeg(x, t, names, fs, dur, channel_subset=["Oz", "Fz", "XX", "M1"])
    assert ch2 == ["Oz", "Fz"] and np.allclose(sig2[0], sig[3]) and np.allclose(sig2[1], sig[0])


def test_filters_and_errors():
    x, t, names, fs, dur = make()
    sig, _, _ = preprocess_eeg(x, t, names, fs, dur, low_cutoff_hz=1.0, high_cutoff_hz=30.0)
    spec = np.abs(np.fft.rfft(sig[0]))
    f = np.fft.rfftfreq(sig.shape[1], 1 / fs)
    assert spec[f > 60].max() < 0.05 * spec[(f > 2) & (f < 25)].max()
    with pytest.raises(ValueError, match="Invalid low_cutoff_hz"):
        preprocess_eeg(x, t, names, fs, dur, low_cutoff_hz=500.0)
    with pytest.raises(ValueError, match="Invalid high_cutoff_hz"):
        preprocess_eeg(x, t, names, fs, dur, high_cutoff_hz=0.0)
    with pytest.raises(ValueError, match="None of the requested channels"):
        preprocess_eeg(x, t, names, fs, dur, channel_subset=["M1", "nope"])
    flat = x.copy()
    flat[:, 0] = 7.0                                     # dead channel: std 0 -> left at zero, no NaN
    s3, _, _ = preprocess_eeg(flat, t, names, 90.0, dur)  # fs = 90: Nyquist 45 < 50, notch skipped
    assert np.all(s3[0] == 0.0) and np.isfinite(s3).all()
