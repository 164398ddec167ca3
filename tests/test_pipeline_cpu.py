"""Host-side logic of the EEG_IBI_FFDTF_Pipeline mirror (no GPU): file discovery, skip rules, the scalar
DSP helpers and their error behaviour (reference: src/eeg_alpha_ibi_ffdtf.py:122-199, 271-448)."""
import numpy as np
import pytest

from hyperscanning_signal_analysis_amd.eeg_alpha_ibi_ffdtf import EEG_IBI_FFDTF_Pipeline


def make_tree(root, dyads=("W_001", "W_002"), films=("Peppa", "Brave"), skip=()):
    for kind in ("EEG", "IBI"):
        for d in dyads:
            for role, sub in (("ch", "child"), ("cg", "caregiver")):
                folder = root / kind / d / sub
                folder.mkdir(parents=True, exist_ok=True)
                for f in films:
                    if (kind, d, role, f) in skip:
                        continue
                    (folder / f"{d}_{kind}_{role}_{f}.nc").write_bytes(b"")
    return root


def test_file_discovery_and_smoke_mode(tmp_path, capsys):
    make_tree(tmp_path)
    p = EEG_IBI_FFDTF_Pipeline(tmp_path, tmp_path / "out", ["Peppa"], plot_global_enabled=False)
    assert p.dyads_to_process == ["W_001", "W_002"]
    assert len(p.eeg_files) == 4 and len(p.ibi_files) == 4
    assert "FULL ANALYSIS" in capsys.readouterr().out
    f, ok = p._find_file(p.eeg_files, "W_002", "Peppa", "cg")
    assert ok and f.name == "W_002_EEG_cg_Peppa.nc"
    assert p._find_file(p.eeg_files, "W_009", "Peppa", "cg") == (None, False)
    p2 = EEG_IBI_FFDTF_Pipeline(tmp_path, tmp_path / "out", ["Peppa", "Brave"], smoke_test=True, smoke_dyads_n=1)
    assert p2.dyads_to_process == ["W_001"] and len(p2.eeg_files) == 4
    assert p2.freq_max == pytest.approx(3.9) and len(p2._freqs()) == 30 and p2._freqs()[-1] == 3.9000000000000026
    with pytest.raises(FileNotFoundError):
        EEG_IBI_FFDTF_Pipeline(tmp_path, tmp_path / "out", ["Nothing"])


def test_dsp_helpers(tmp_path):
    make_tree(tmp_path)
    p = EEG_IBI_FFDTF_Pipeline(tmp_path, tmp_path / "out", ["Peppa"])
    rng = np.random.default_rng(0)
    fs = 128.0
    t = np.arange(int(80 * fs)) / fs
    eeg = rng.standard_normal((19, t.size))
    eeg[3] += 5 * np.sin(2 * np.pi * 10 * t)            # strong alpha on "F4"
    names = [f"C{k}" for k in range(19)]
    names[2], names[3] = "F3", "F4"
    filt = p._alpha_bandpass_filter(eeg, fs)
    assert filt.shape == eeg.shape
    spec = np.abs(np.fft.rfft(filt[3]))
    f = np.fft.rfftfreq(t.size, 1 / fs)
    assert spec[(f > 20)].max() < 1e-2 * spec.max() and abs(f[np.argmax(spec)] - 10) < 0.1
    faa = p._compute_asymmetry(filt, names)
    assert faa.shape == (t.size,) and np.median(faa) > 1.0           # right (F4) alpha dominates
    with pytest.raises(ValueError, match="not found"):
        p._compute_asymmetry(filt, [f"C{k}" for k in range(19)])
    with pytest.raises(ValueError, match="metric"):
        p._compute_asymmetry(filt, names, metric="x")
    ds = p._downsample_signal(faa, fs, 8.0)
    assert ds.shape == (640,)
    with pytest.raises(ValueError, match="lower"):
        p._downsample_signal(faa, 8, 8)
    with pytest.raises(ValueError, match="divisible"):
        p._downsample_signal(faa, 100, 8)
    crop = p._crop_signal(ds, 8.0, 10, 60)
    assert crop.shape == (480,) and crop[0] == ds[80]
    with pytest.raises(ValueError, match="too short"):
        p._crop_signal(ds[:400], 8.0, 10, 60)
    wins = p._create_windows(np.zeros((4, 480)), 3, None)
    assert [w.shape for w in wins] == [(4, 160)] * 3


def test_faa_chain_matches_reference_outputs(tmp_path, golden):
    """g8: the reference's own _alpha_bandpass_filter / _compute_asymmetry / _downsample_signal / _crop_signal
    (src/eeg_alpha_ibi_ffdtf.py:271-448) run on seeded inputs; the mirror's SciPy chain must reproduce them."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from g8_inputs import FS_EEG, NAMES, g8_inputs
    g, inp = golden("g8_faa_chain.npz"), g8_inputs()
    make_tree(tmp_path)
    p = EEG_IBI_FFDTF_Pipeline(tmp_path, tmp_path / "out", ["Peppa"])
    rows = []
    for role in ("ch", "cg"):
        filt = p._alpha_bandpass_filter(inp[f"eeg_{role}"], FS_EEG)
        assert np.allclose(filt[[2, 3]], g[f"filt_{role}"], rtol=1e-9, atol=1e-12)
        faa = p._compute_asymmetry(filt, NAMES, metric="amp")
        assert np.allclose(faa, g[f"faa_{role}"], rtol=1e-8, atol=1e-10)
        assert np.allclose(p._compute_asymmetry(filt, NAMES, metric="power"), g[f"faa_pow_{role}"], rtol=1e-8, atol=1e-10)
        ds = p._downsample_signal(faa, FS_EEG, 8.0)
        assert np.allclose(ds, g[f"faa_ds_{role}"], rtol=1e-8, atol=1e-10)
        crop = p._crop_signal(ds, 8.0, 10, 60)
        assert np.allclose(crop, g[f"faa_crop_{role}"], rtol=1e-8, atol=1e-10)
        rows += [crop, inp[f"ibi8_{role}"]]
    block = np.vstack(rows)
    block = (block - np.mean(block, axis=1, keepdims=True)) / np.std(block, axis=1, keepdims=True)
    assert np.allclose(block, g["block"], rtol=1e-8, atol=1e-9)
    # the numerical part of _compute_ffDTF on this block is pinned on the oracle (CPU) and on the GPU path
    from oracle import mvar_oracle as O
    freqs = p._freqs()
    assert np.allclose(O.full_freq_dtf(g["block"], freqs, 8.0, 5), g["ff_block"], rtol=1e-9, atol=1e-13)
    assert np.allclose(O.multivariate_spectra(g["block"], freqs, 8.0, 5), g["sp_block"], rtol=1e-9, atol=1e-12)
    assert int(g["p_block"]) == 5
