"""Pins the CPU oracle (oracle/mvar_oracle.py) to outputs of the reference itself.

The .npz files under tests/golden/ were produced by tests/golden/make_golden.py, which imported
/root/reference/src/mtmvar.py and /root/reference/src/eeg_alpha_ibi_ffdtf.py in the build container.
Tolerances are ~1e-10: the oracle repeats the reference's LAPACK calls in the same order.
"""
import numpy as np
import pytest

from oracle import mvar_oracle as O


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_g1_config1(golden):
    g = golden("g1_config1.npz")
    x, fs, freqs = g["x"], float(g["fs"]), g["freqs"]
    rl, rr, r0 = O.count_corr(x, 4, 1)
    assert rel(rl, g["r_left"]) < 1e-14 and rel(rr, g["r_right"]) < 1e-14 and rel(r0, g["r"]) < 1e-14
    ar, V = O.ar_coeff(x, 4)
    assert rel(ar, g["ar"]) < 1e-12 and rel(V, g["V"]) < 1e-12
    for fn in (O.mvar_transfer_function, O.mvar_transfer_function_loop):
        H, A = fn(g["ar"], freqs, fs)
        assert rel(A, g["A"]) < 1e-14 and rel(H, g["H"]) < 1e-12
    assert rel(O.dtf_multivariate(x, freqs, fs, 4), g["dtf"]) < 1e-11
    assert np.allclose(O.full_freq_dtf(x, freqs, fs, 4), g["ffdtf"], rtol=1e-10, atol=0)
    assert np.allclose(O.full_freq_dtf_loop(x, freqs, fs, 4), g["ffdtf"], rtol=1e-10, atol=0)
    assert rel(O.multivariate_spectra(x, freqs, fs, 4), g["spectra"]) < 1e-11
    for c in ("AIC", "HQ", "SC"):
        crit, rng, popt = O.mvar_criterion(x, 10, c)
        assert np.allclose(crit, g[f"crit_{c}"], rtol=1e-11, atol=1e-12)
        assert int(popt) == int(g[f"crit_{c}_popt"])
    popt = int(g["crit_AIC_popt"])
    assert np.allclose(O.full_freq_dtf(x, freqs, fs, popt), g["ffdtf_auto"], rtol=1e-10, atol=0)


def test_g2_northstar_window(golden):
    g = golden("g2_northstar.npz")
    x, fs, freqs, p = g["x"], float(g["fs"]), g["freqs"], int(g["p"])
    assert rel(O.lag_covariances(x, p), g["R"]) < 1e-14
    ar, V = O.ar_coeff(x, p)
    assert rel(ar, g["ar"]) < 1e-11 and rel(V, g["V"]) < 1e-11
    ff = O.full_freq_dtf(x, freqs, fs, p)
    assert np.allclose(ff[:, :, ::16], g["ffdtf_sub"], rtol=1e-9, atol=0)
    assert np.allclose(ff.sum(axis=2), g["ffdtf_fsum"], rtol=1e-10, atol=0)
    assert abs(ff.sum() - float(g["ffdtf_sum"])) < 1e-10 and abs(ff.sum() - 64.0) < 1e-10
    assert abs((ff ** 2).sum() - float(g["ffdtf_sqsum"])) < 1e-12
    den = O.dtf_multivariate(x, freqs, fs, p).sum(axis=(1, 2))
    assert np.allclose(den, g["den"], rtol=1e-10)
    S = O.multivariate_spectra(x, freqs, fs, p)
    assert rel(S[:, :, ::64], g["spectra_sub"]) < 1e-10


def test_g3_overlapping_windows(golden):
    g = golden("g3_overlap.npz")
    x, fs, freqs, p = g["x"], float(g["fs"]), g["freqs"], int(g["p"])
    ff = O.sliding_ffdtf(x, 1000, 3, p, freqs, fs)
    for i in range(3):
        assert np.allclose(ff[i][:, :, ::32], g[f"ffdtf_sub{i}"], rtol=1e-9, atol=0)
        assert np.allclose(ff[i].sum(axis=2), g[f"ffdtf_fsum{i}"], rtol=1e-10, atol=0)
        ar, _ = O.ar_coeff(x[:, 500 * i:500 * i + 1000], p)
        assert rel(ar, g[f"ar{i}"]) < 1e-11


def test_g4_config4_windows(golden):
    g = golden("g4_config4.npz")
    x, fs = g["x"], float(g["fs"])
    freqs = np.arange(1.0, (fs / 2 - 0.1) + 0.1, 0.1)               # quirk Q9
    assert len(freqs) == 30 and np.array_equal(freqs, g["freqs"])
    for tag, (nw, ws) in {"a": (3, None), "b": (5, 160)}.items():
        pos, w = O.window_positions(x.shape[1], nw, ws)
        assert np.array_equal(pos, g[f"starts_{tag}"]) and w == int(g[f"wsize_{tag}"])
        for i, win in enumerate(O.create_windows(x, nw, ws)):
            assert np.allclose(O.full_freq_dtf(win, freqs, fs, 5), g[f"ff_{tag}"][i], rtol=1e-9, atol=0)
            assert rel(O.multivariate_spectra(win, freqs, fs, 5), g[f"sp_{tag}"][i]) < 1e-10
    assert np.allclose(O.full_freq_dtf(x, freqs, fs, 5), g["ff_global"], rtol=1e-9, atol=0)
    _, _, popt = O.mvar_criterion(x, 20, "AIC")
    assert int(popt) == int(g["p_opt_auto"])
    assert np.allclose(O.full_freq_dtf(x, freqs, fs, int(popt)), g["ff_global_auto"], rtol=1e-9, atol=0)


def test_g5_multitrial(golden):
    g = golden("g5_multitrial.npz")
    rl, rr, r0 = O.count_corr(g["x"], 3, 1)
    assert rel(rl, g["r_left"]) < 1e-13 and rel(rr, g["r_right"]) < 1e-13 and rel(r0, g["r"]) < 1e-13
    ar, V = O.ar_coeff(g["x"], 3)
    assert rel(ar, g["ar"]) < 1e-12 and rel(V, g["V"]) < 1e-12
    H, _ = O.mvar_transfer_function(ar, g["freqs"], float(g["fs"]))
    assert rel(H, g["H"]) < 1e-11


def test_g6_errors(golden):
    g = golden("g6_errors.npz")
    assert str(g["deadchan_raises"]).startswith("LinAlgError")
    with pytest.raises(np.linalg.LinAlgError):
        O.ar_coeff(g["xz"], 3)
    with pytest.raises(ValueError, match="Invalid criterion type"):
        O.mvar_criterion(np.random.default_rng(0).standard_normal((3, 200)), 3, "BIC")
    cases = [(481, 3, None), (480, 3, 100), (480, 3, 481), (480, 5, 478)]
    for (T, nw, ws), msg in zip(cases, g["window_errors"]):
        with pytest.raises(ValueError) as e:
            O.window_positions(T, nw, ws)
        assert str(e.value) == str(msg)


def test_g7_connectivity_measures(golden):
    """partial_coherence / direct_dtf / gen_partial_directed_coherence (src/mtmvar.py:287-468)."""
    g = golden("g7_connectivity.npz")
    for tag in "abc":
        x, fs, freqs, p = g[f"x_{tag}"], float(g[f"fs_{tag}"]), g[f"freqs_{tag}"], int(g[f"p_{tag}"])
        assert rel(O.partial_coherence(g[f"spectra_{tag}"]), g[f"pcoh_{tag}"]) < 1e-12
        assert rel(O.direct_dtf(x, freqs, fs, p), g[f"ddtf_{tag}"]) < 1e-10
        assert rel(O.gen_partial_directed_coherence(x, freqs, fs, p), g[f"gpdc_{tag}"]) < 1e-12
    assert rel(O.partial_coherence(g["Z"]), g["pcoh_Z"]) < 1e-13
    assert np.array_equal(O.partial_coherence(np.full((1, 1, 3), 2.0 + 1.0j)), g["pcoh_1x1"])
