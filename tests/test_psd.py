"""Multitaper PSD.  PARITY UNPINNED: mne==1.11.0, which holds the arithmetic of the
reference's `compute_psd_multitaper` (src/psd.py:30-32), is not available offline and the reference stores no
output of it.  What is checked: the CPU restatement against analytic properties (CPU), and the GPU path (hipFFT +
HIP kernels through the C ABI) against that restatement (GPU)."""
import numpy as np
import pytest

from oracle import psd_oracle as P


def _signal(n_ch=5, n=6000, fs=500.0, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    x = rng.standard_normal((n_ch, n)) + 3.0
    x[1] += 4.0 * np.sin(2 * np.pi * 10.0 * t)
    return x, fs


def test_oracle_properties_cpu():
    x, fs = _signal()
    freqs, psd = P.compute_psd_multitaper(x, fs, 1.0, 30.0, 2.0)
    assert psd.shape == (5, len(freqs)) and freqs[0] >= 1.0 and freqs[-1] <= 30.0
    assert np.all(psd > 0)
    assert abs(freqs[np.argmax(psd[1])] - 10.0) <= 1.0                 # the 10 Hz line, smeared by the 2 Hz bandwidth
    # white noise of unit variance: one-sided density per bin equals 2 (normalization 'length', no /sfreq);
    # averaged over the band and the noise channels it is within a few percent
    assert abs(psd[[0, 2, 3, 4]].mean() / 2.0 - 1.0) < 0.12
    # the mean is removed, scaling is quadratic, channels are independent
    f2, p2 = P.compute_psd_multitaper(3.0 * x + 100.0, fs, 1.0, 30.0, 2.0)
    assert np.allclose(p2, 9.0 * psd, rtol=1e-9)
    tapers, eig = P.mt_params(6000, fs, 2.0)
    assert tapers.shape[0] == len(eig) and np.all(eig > 0.9) and len(eig) >= 20
    assert np.allclose((tapers ** 2).sum(axis=1), 1.0, atol=2e-3)       # unit energy up to the periodic-window truncation


@pytest.mark.gpu
@pytest.mark.parametrize("n_ch,n,fs,bw", [(5, 6000, 500.0, 2.0), (19, 4097, 128.0, 1.0), (2, 30_001, 500.0, 2.0),
                                           (4, 20_011, 500.0, 2.0), (3, 4100, 250.0, 3.0)])
def test_gpu_psd_matches_restatement(n_ch, n, fs, bw):
    """Lengths whose prime factors are all <= 13 go through batched real-to-complex FFTs; the others (4 097 = 17 * 241,
    20 011 prime, 4 100 = 4 * 25 * 41: a segment cut by event times has any length) through the pruned chirp-z transform
    of csrc/psd.hip when the wanted band is narrow enough (two tapers per complex transform, an odd taper count leaves the
    last one half empty), else through rocFFT's own Bluestein (30 001): the same numbers on the same bins."""
    from hyperscanning_signal_analysis_amd.psd import compute_psd_multitaper
    x, _ = _signal(n_ch, n, fs, seed=n_ch)
    freqs, psd = compute_psd_multitaper(x, fs, 1.0, 30.0, bw)
    fo, po = P.compute_psd_multitaper(x, fs, 1.0, 30.0, bw)
    assert np.array_equal(freqs, fo)
    assert np.abs(psd - po).max() <= 1e-9 * np.abs(po).max()
    assert np.allclose(psd, po, rtol=1e-7, atol=1e-12 * np.abs(po).max())


@pytest.mark.gpu
def test_gpu_psd_chunked_channels_and_edges():
    from hyperscanning_signal_analysis_amd.psd import average_psd_across_conditions, compute_psd_multitaper
    x, fs = _signal(7, 3000, 250.0, seed=9)
    f1, p1 = compute_psd_multitaper(x, fs, 0.0, 125.0, 2.0)                       # DC and Nyquist bins included
    f2, p2 = compute_psd_multitaper(x, fs, 0.0, 125.0, 2.0, max_workspace_bytes=1)  # one channel per chunk
    assert np.array_equal(p1, p2)
    fo, po = P.compute_psd_multitaper(x, fs, 0.0, 125.0, 2.0)
    assert np.abs(p1 - po).max() <= 1e-9 * np.abs(po).max()
    assert np.allclose(average_psd_across_conditions({"a": p1, "b": 3 * p1}), 2 * p1)
    with pytest.raises(ValueError):
        average_psd_across_conditions({})
    # the same on the chirp-z path (4 100 samples): the DC bin included, one channel per chunk, run twice (cached tables)
    x, fs = _signal(5, 4100, 250.0, seed=10)
    f3, p3 = compute_psd_multitaper(x, fs, 0.0, 20.0, 2.0)
    f4, p4 = compute_psd_multitaper(x, fs, 0.0, 20.0, 2.0, max_workspace_bytes=1)
    fo, po = P.compute_psd_multitaper(x, fs, 0.0, 20.0, 2.0)
    assert f3[0] == 0.0 and np.array_equal(p3, p4) and np.array_equal(f3, fo)
    assert np.abs(p3 - po).max() <= 1e-9 * np.abs(po).max()


@pytest.mark.gpu
@pytest.mark.parametrize("sym", [False, True])
@pytest.mark.parametrize("M,NW,K", [(16, 1.5, 2), (33, 1.0, 1), (64, 2.0, 4), (1000, 4.0, 8), (999, 2.5, 5), (4096, 8.0, 16), (6000, 12.0, 24), (20_000, 40.0, 80)])
def test_gpu_dpss_matches_scipy(M, NW, K, sym):
    """`hmv_dpss_f64` (Sturm-count multisection + inverse iteration on the commuting tridiagonal matrix, SciPy's sign
    convention, ratios through hipFFT) against scipy.signal.windows.dpss -- the generator mne calls.  Eigenvectors of
    well separated eigenvalues: agreement ~1e-11; ratios ~1e-13."""
    from scipy.signal.windows import dpss
    from hyperscanning_signal_analysis_amd.psd import dpss_device
    ref_t, ref_r = dpss(M, NW, K, sym=sym, norm=2, return_ratios=True)
    t, r = dpss_device(M, NW, K, sym)
    t, r = t.cpu().numpy(), r.cpu().numpy()
    assert t.shape == (K, M) and r.shape == (K,)
    if sym:
        assert np.abs((t ** 2).sum(axis=1) - 1.0).max() < 1e-12                   # unit 2-norm
        assert np.abs(t @ t.T - np.eye(K)).max() < 1e-9                           # orthogonal without re-orthogonalisation
    assert np.abs(t - ref_t).max() < 1e-9, np.abs(t - ref_t).max()
    assert np.abs(r - ref_r).max() < 1e-11, np.abs(r - ref_r).max()


@pytest.mark.gpu
def test_gpu_dpss_refuses_bad_arguments_and_psd_uses_device_tapers(tmp_path, monkeypatch):
    """hmv_dpss_f64 argument checks surface as exceptions; compute_psd_multitaper takes its tapers from the device
    generator (a taper file appears in the cache directory, and a second call reads it back to the same result)."""
    from hyperscanning_signal_analysis_amd import psd as P
    with pytest.raises(Exception):
        P.dpss_device(100, 60.0, 4)            # half-bandwidth product >= n_times / 2
    with pytest.raises(Exception):
        P.dpss_device(100, 4.0, 101)           # more tapers than samples
    monkeypatch.setenv("HYPERMVAR_DPSS_CACHE", str(tmp_path))
    P._TAPERS.clear()
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 1500))
    f1, p1 = P.compute_psd_multitaper(x, 250.0, 1.0, 40.0, 2.0)
    files = list(tmp_path.glob("dpss_n1500_*.npz"))
    assert len(files) == 1
    P._TAPERS.clear()                          # second call: tapers from the file
    f2, p2 = P.compute_psd_multitaper(x, 250.0, 1.0, 40.0, 2.0)
    assert np.array_equal(f1, f2) and np.array_equal(p1, p2)


@pytest.mark.gpu
def test_gpu_psd_many_segment_lengths_bounded_caches():
    """A batch meets a new segment length with almost every file: FFT plans and chirp-z tables are cached up to a bound and
    rebuilt past it.  30 lengths (smooth and awkward alternating), each against the restatement, the first one again at the
    end (its plan and tables were dropped in between)."""
    from hyperscanning_signal_analysis_amd.psd import compute_psd_multitaper
    first = None
    for k in range(30):
        n = 2000 + 37 * k + (k % 2)
        x, fs = _signal(2, n, 250.0, seed=100 + k)
        f, p = compute_psd_multitaper(x, fs, 1.0, 30.0, 2.0)
        fo, po = P.compute_psd_multitaper(x, fs, 1.0, 30.0, 2.0)
        assert np.array_equal(f, fo) and np.abs(p - po).max() <= 1e-9 * np.abs(po).max(), n
        if first is None:
            first = (x, fs, p)
    f, p = compute_psd_multitaper(first[0], first[1], 1.0, 30.0, 2.0)
    assert np.array_equal(p, first[2])
