"""EEG_IBI_FFDTF_Pipeline mirror end to end on the GPU with synthetic recordings (config 4 of
BASELINE.json): .npz layout of the reference (eeg_alpha_ibi_ffdtf.py:637-658) and parity of every
window with the oracle on the same pre-processed block."""
import json

import numpy as np
import pytest
import torch

from oracle import mvar_oracle as O
from tests.test_pipeline_cpu import make_tree

pytestmark = pytest.mark.gpu


def synthetic_loader(eeg_file, ibi_file, role):
    seed = abs(hash((eeg_file.name, role))) % (2 ** 31)
    rng = np.random.default_rng(seed)
    fs = 128.0
    n = int(80 * fs)
    t = np.arange(n) / fs
    names = ["Fp1", "Fp2", "F3", "F4", "C3", "C4", "P3", "P4", "O1", "O2", "F7", "F8", "T3", "T4", "T5", "T6", "Fz", "Cz", "Pz"]
    eeg = rng.standard_normal((19, n))
    eeg[2] += (1.5 + np.sin(2 * np.pi * 0.21 * t)) * np.sin(2 * np.pi * 10 * t)
    eeg[3] += (1.5 + np.cos(2 * np.pi * 0.13 * t)) * np.sin(2 * np.pi * 10.5 * t)
    ibi = 0.8 + 0.05 * np.sin(2 * np.pi * 0.25 * t) + 0.01 * np.cumsum(rng.standard_normal(n)) / np.sqrt(n)
    return t, eeg, fs, names, ibi[None, :], fs, 80.0


def test_run_pipeline_matches_oracle(tmp_path):
    from hyperscanning_signal_analysis_amd.eeg_alpha_ibi_ffdtf import EEG_IBI_FFDTF_Pipeline
    root = make_tree(tmp_path / "data", dyads=("W_001", "W_002"), films=("Peppa",),
                     skip={("IBI", "W_002", "cg", "Peppa")})
    out = tmp_path / "out"
    pipe = EEG_IBI_FFDTF_Pipeline(root, out, ["Peppa"], n_windows=5, window_size=160, ar_p=5,
                                  plot_global_enabled=False, save_global_enabled=False,
                                  plot_windowed_enabled=False, save_windowed_enabled=False,
                                  loader=synthetic_loader)
    pipe.run_pipeline()
    assert not (out / "W_002").exists()                              # missing file -> [SKIP]
    z = np.load(out / "W_001" / "W_001_Peppa_ffDTF.npz", allow_pickle=False)
    assert sorted(z.files) == sorted(["ff_dtf_global", "spectra_global", "ff_dtf_windowed", "spectra_windowed",
                                      "p_opt_g", "p_opt_w", "meta"])
    assert z["ff_dtf_windowed"].shape == (5, 4, 4, 30) and z["spectra_windowed"].dtype == np.complex128
    assert z["ff_dtf_global"].shape == (4, 4, 30) and int(z["p_opt_g"]) == 5 and list(z["p_opt_w"]) == [5] * 5
    meta = json.loads(str(z["meta"]))
    assert meta["chan_names"] == ["faa_ch", "ibi_ch", "faa_cg", "ibi_cg"] and meta["fs"] == 8.0
    # rebuild the 4 x 480 block with the pipeline's own host DSP and check the GPU numerics against the oracle
    blocks = []
    for role, rname in (("ch", "Child"), ("cg", "Care Giver")):
        _, eeg, fs, names, ibi, fs_ibi, _ = synthetic_loader(root / "EEG" / "W_001" / ("child" if role == "ch" else "caregiver") / f"W_001_EEG_{role}_Peppa.nc", None, rname)
        faa, ibic = pipe._preprocess(eeg, fs, names, ibi, fs_ibi)
        blocks += [faa, ibic]
    sig = np.vstack(blocks)
    sig = (sig - sig.mean(axis=1, keepdims=True)) / sig.std(axis=1, keepdims=True)
    freqs = np.arange(1.0, (8.0 / 2 - 0.1) + 0.1, 0.1)
    for k, w in enumerate(O.create_windows(sig, 5, 160)):
        assert np.allclose(z["ff_dtf_windowed"][k], O.full_freq_dtf(w, freqs, 8.0, 5), rtol=1e-7, atol=1e-12)
        ref = O.multivariate_spectra(w, freqs, 8.0, 5)
        assert np.abs(z["spectra_windowed"][k] - ref).max() <= 1e-8 * np.abs(ref).max()
    assert np.allclose(z["ff_dtf_global"], O.full_freq_dtf(sig, freqs, 8.0, 5), rtol=1e-7, atol=1e-12)


def test_order_selection_path(tmp_path):
    from hyperscanning_signal_analysis_amd.eeg_alpha_ibi_ffdtf import EEG_IBI_FFDTF_Pipeline
    root = make_tree(tmp_path / "data", dyads=("W_001",), films=("Peppa",))
    pipe = EEG_IBI_FFDTF_Pipeline(root, tmp_path / "out", ["Peppa"], ar_p=None, loader=synthetic_loader)
    x = np.random.default_rng(1).standard_normal((4, 480))
    x[:, 1:] += 0.5 * x[:, :-1]
    ff, sp, p_opt = pipe._compute_ffDTF("D", x, list("abcd"), 8.0, plot=False)
    _, _, want = O.mvar_criterion(x, 20, "AIC")
    assert int(p_opt) == int(want)
    assert np.allclose(ff, O.full_freq_dtf(x, pipe._freqs(), 8.0, int(want)), rtol=1e-7, atol=1e-12)


def test_compute_and_plot_mvar_with_injected_loader(capsys):
    """a11: single-file driver (mtmvar.py:1006-1128) with a working loader (reference import is broken, Q6)."""
    from hyperscanning_signal_analysis_amd import mtmvar as M
    rng = np.random.default_rng(4)
    x = rng.standard_normal((5, 3000))
    x[:, 1:] += 0.7 * x[:, :-1]
    x = (x - x.mean(axis=1, keepdims=True)) / x.std(axis=1, keepdims=True)
    names = ["Fz", "Cz", "Pz", "C3", "C4"]

    def loader(path, channel_subset=None, low_cutoff_hz=None, high_cutoff_hz=None):
        return x, names, 128.0, np.arange(3000) / 128.0, 3000 / 128.0

    ff, sp, ch, crit, rng_, p_opt = M.compute_and_plot_mvar("W_001_EEG_ch_Peppa.nc", optimal_model_order=None,
                                                            max_model_order=8, plot=False, loader=loader)
    out = capsys.readouterr().out
    assert "AIC optimal model order" in out and ch == names
    freqs = np.arange(1.0, 40.0 + 0.5, 0.5)
    _, _, want = O.mvar_criterion(x, 8, "AIC")
    assert int(p_opt) == int(want) and len(crit) == 8 and list(rng_) == list(range(1, 9))
    assert np.allclose(ff, O.full_freq_dtf(x, freqs, 128.0, int(want)), rtol=1e-7, atol=1e-12)
    ref = O.multivariate_spectra(x, freqs, 128.0, int(want))
    assert np.abs(sp - ref).max() <= 1e-8 * np.abs(ref).max()
    ff2, _, _, crit2, rng2, p2 = M.compute_and_plot_mvar("x.nc", optimal_model_order=3, plot=False, loader=loader)
    assert p2 == 3 and crit2.size == 0 and rng2.size == 0 and ff2.shape == (5, 5, 79)


def test_compute_ffdtf_on_reference_block(tmp_path, golden):
    """g8: the reference's own `_compute_ffDTF` outputs on the 4 x 480 block of its preprocessing chain."""
    from hyperscanning_signal_analysis_amd.eeg_alpha_ibi_ffdtf import EEG_IBI_FFDTF_Pipeline
    g = golden("g8_faa_chain.npz")
    root = make_tree(tmp_path / "data")
    pipe = EEG_IBI_FFDTF_Pipeline(root, tmp_path / "out", ["Peppa"], ar_p=5, plot_global_enabled=False,
                                  save_global_enabled=False, plot_windowed_enabled=False,
                                  save_windowed_enabled=False)
    ff, sp, p_opt = pipe._compute_ffDTF("W_000", g["block"], ["faa_ch", "ibi_ch", "faa_cg", "ibi_cg"], 8.0,
                                        plot=False, save_plot=False)
    assert p_opt == int(g["p_block"])
    assert np.abs(ff - g["ff_block"]).max() <= 1e-9 * np.abs(g["ff_block"]).max()
    assert np.abs(sp - g["sp_block"]).max() <= 1e-9 * np.abs(g["sp_block"]).max()


@pytest.mark.parametrize("pinned", [False, True])
def test_streamed_recordings_equal_resident_ones(pinned):
    """`Engine.stream_dyads` (H2D of recording d + 1 and D2H of recording d - 1 under the compute of recording d, three HIP
    streams, two buffer slots reused round robin) returns what the resident path computes, bit for bit, for every
    recording -- from NumPy arrays (staged through pinned buffers) and from pinned tensors (copied as they are); a
    recording with a dead channel comes back NaN-filled in its windows without stopping the stream."""
    from hyperscanning_signal_analysis_amd import distributed as hdist
    from hyperscanning_signal_analysis_amd.engine import default_engine
    from hyperscanning_signal_analysis_amd.sliding import regular_grid, window_items, window_positions
    from hyperscanning_signal_analysis_amd.synthetic import synthetic_var_dyad
    eng = default_engine()
    m, T, n, p = 64, 9000, 1000, 8
    freqs = 0.5 * np.arange(1, 65)
    pos, n = window_positions(T, 2 * T // n - 1, n)
    recs = [synthetic_var_dyad(50 + d, m=m, p=4, T=T, burn=300) for d in range(5)]
    recs[3] = recs[3].copy()
    recs[3][7] = 0.0                                               # a dead channel: singular fits
    feed = [torch.from_numpy(r).pin_memory() for r in recs] if pinned else recs
    tl = []
    host_out = torch.empty(5, len(pos), m, m, 5, dtype=torch.float64).pin_memory() if pinned else None
    got = eng.stream_dyads(feed, n, pos, p, freqs, 500.0, depth=2, timeline=tl, out=host_out)
    assert len(got) == 5 and [e[1] for e in tl if e[0] == "collected"] == [0, 1, 2, 3, 4]
    lo, hi = hdist.band_bins(freqs)
    rec_i, st_i = window_items(1, pos, eng.device)
    for d, r in enumerate(recs):
        ff = eng.sliding_ffdtf(eng.to_device(r[None]), rec_i, st_i, n, p, freqs, 500.0, check="nan", grid=regular_grid(pos, n, p))
        want = eng.band_sums(ff, lo, hi).cpu().numpy()
        assert got[d].shape == want.shape == (len(pos), m, m, len(hdist.DEFAULT_BANDS))
        assert np.array_equal(got[d], want, equal_nan=True), d
        assert np.isnan(want).all() == (d == 3)
    # other pipeline depths (1: strictly serial, the download of d queued before d + 1 starts; 3: the default), a consumer of
    # the full array on the device (keep_full: then the bands are summed from the array, the two-kernel route) and a
    # caller-chosen reduction: the same numbers
    for kw in ({"depth": 1}, {}, {"keep_full": lambda d, ff: seen.append((d, tuple(ff.shape)))},
               {"reduce": lambda ff: ff.sum(dim=3)}):
        seen = []
        again = eng.stream_dyads(feed, n, pos, p, freqs, 500.0, **kw)
        for d in range(5):
            if "reduce" in kw:
                assert again[d].shape == (len(pos), m, m) and (d == 3 or np.allclose(again[d].sum(axis=2), 1.0, atol=1e-12))
            else:
                assert np.array_equal(again[d], got[d], equal_nan=True), (kw, d)
        if "keep_full" in kw:
            assert seen == [(d, (len(pos), m, m, len(freqs))) for d in range(5)]


def _npz_equal(a, b):
    za, zb = np.load(a, allow_pickle=False), np.load(b, allow_pickle=False)
    assert sorted(za.files) == sorted(zb.files)
    for k in za.files:
        if k != "meta":
            assert np.array_equal(za[k], zb[k]), (a.name, k)


@pytest.mark.parametrize("ar_p", [5, None])
def test_pipeline_batched_over_items_equals_one_item_at_a_time(tmp_path, ar_p):
    """BASELINE config 4 as a batch: the windows and global blocks of 200 dyad x film items in one GPU batch per block
    shape and model order (AIC order selection included when ar_p is None: all blocks from one fit at order 20) give the
    same files, bit for bit, as the reference's loop structure (one item at a time)."""
    from hyperscanning_signal_analysis_amd.eeg_alpha_ibi_ffdtf import EEG_IBI_FFDTF_Pipeline
    n_dyads = 100 if ar_p is not None else 6
    dyads = tuple(f"W_{k:03d}" for k in range(1, n_dyads + 1))
    root = make_tree(tmp_path / "data", dyads=dyads, films=("Peppa", "Brave"))
    kw = dict(n_windows=5, window_size=160, ar_p=ar_p, plot_global_enabled=False, save_global_enabled=False,
              plot_windowed_enabled=False, save_windowed_enabled=False, loader=synthetic_loader)
    EEG_IBI_FFDTF_Pipeline(root, tmp_path / "batched", ["Peppa", "Brave"], batch_items=256, **kw).run_pipeline()
    EEG_IBI_FFDTF_Pipeline(root, tmp_path / "single", ["Peppa", "Brave"], batch_items=1, **kw).run_pipeline()
    files = sorted((tmp_path / "batched").rglob("*.npz"))
    assert len(files) == 2 * n_dyads
    for f in files:
        _npz_equal(f, tmp_path / "single" / f.parent.name / f.name)
    if ar_p is None:                       # the batched order selection is the criterion of the oracle, block by block
        z = np.load(files[0], allow_pickle=False)
        assert z["p_opt_w"].shape == (5,) and 1 <= int(z["p_opt_g"]) <= 20
