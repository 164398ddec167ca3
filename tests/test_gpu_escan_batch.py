"""BASELINE config 5 in miniature: a synthetic tree of exported per-task EEG files (the reference's layout and
attributes), two complete dyads x three segments, one dyad without a caregiver file and one unreadable dyad, through
`escan_batch.run` on the GPU with an injected NumPy reader (xarray / netCDF4 are absent).  Every window is compared
with the oracle on the block the front-end built; bookkeeping (skip-if-exists, [SKIP], failed-dyad log, rank
sharding) is checked as the reference's batch drivers do it."""
import json

import numpy as np
import pytest
import torch

from oracle import mvar_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from hyperscanning_signal_analysis_amd import escan_batch as EB
    from hyperscanning_signal_analysis_amd import distributed as hd

FS = 128.0
CHANS = ["Fp1", "Fp2", "F3", "F4", "M1", "C3", "C4", "M2", "O1", "O2"]          # two mastoids to be dropped


def _write(path, seed, dur_s, events, json_attr):
    rng = np.random.default_rng(seed)
    n = int(dur_s * FS)
    t = np.arange(n) / FS - 2.0                                  # 2 s margin before task onset
    x = rng.standard_normal((n, len(CHANS)))
    x[1:] += 0.6 * x[:-1]
    x[:, 1:] += 0.3 * x[:, :-1]
    ev = [{"name": nm, "start_s": 100.0 + st, "start_rel_s": st, "duration_s": du} for nm, st, du in events]
    attrs = {"sampling_freq": FS, "who": path.stem.split("_")[3], "dyad_id": "_".join(path.stem.split("_")[:2]),
             "task_events_structure": json.dumps(ev) if json_attr else "LIST"}
    path.parent.mkdir(parents=True, exist_ok=True)
    with open(path, "wb") as f:
        np.savez(f, data=x, time=t, channels=np.asarray(CHANS), attrs=json.dumps(attrs), events=json.dumps(ev))


def _reader(path):
    with np.load(path, allow_pickle=False) as z:
        attrs = json.loads(str(z["attrs"]))
        if attrs["task_events_structure"] == "LIST":             # the attribute as a decoded list (ncdf.py:88-91)
            attrs["task_events_structure"] = json.loads(str(z["events"]))
        return {"data_tc": z["data"], "time": z["time"], "channels": [str(c) for c in z["channels"]], "attrs": attrs}


@pytest.fixture()
def tree(tmp_path):
    root = tmp_path / "UNIWAW_imported"
    movies = [("Peppa", 1.0, 11.0), ("Brave", 14.0, 9.5)]
    talk = [("talk_1", 0.5, 12.0)]
    for d, dy in enumerate(["W_003", "W_010"]):
        for r, (code, role) in enumerate((("ch", "child"), ("cg", "caregiver"))):
            _write(root / "EEG" / dy / role / f"{dy}_EEG_{code}_passive_movies.nc", 10 * d + r, 30.0, movies, r == 0)
            _write(root / "EEG" / dy / role / f"{dy}_EEG_{code}_talk.nc", 100 + 10 * d + r, 16.0, talk, r == 1)
    _write(root / "EEG" / "W_020" / "child" / "W_020_EEG_ch_talk.nc", 7, 16.0, talk, True)        # no caregiver
    bad = root / "EEG" / "W_030" / "child" / "W_030_EEG_ch_talk.nc"
    bad.parent.mkdir(parents=True)
    bad.write_bytes(b"not a file the reader understands")
    (root / "EEG" / "W_030" / "caregiver").mkdir()
    (root / "EEG" / "W_030" / "caregiver" / "W_030_EEG_cg_talk.nc").write_bytes(b"neither")
    return root


def test_batch_matches_oracle_per_window_and_keeps_books(tree, tmp_path, capsys):
    out = tmp_path / "ffdtf_out"
    freqs = np.arange(1.0, 33.0, 1.0)                            # 32 points: the in-kernel normalisation path
    res = EB.run(tree, out, window_s=2.0, overlap=0.5, model_order=3, freqs=freqs, low_cutoff_hz=1.0, high_cutoff_hz=45.0,
                 save_full=True, with_psd=True, psd_fmin=2.0, psd_fmax=30.0, psd_bandwidth=2.0, reader=_reader)
    log = capsys.readouterr().out
    assert res["done"] == ["W_003", "W_010"] and res["skipped"] == ["W_020"] and [d for d, _ in res["failed"]] == ["W_030"]
    assert "[SKIP] W_020 talk: missing caregiver file" in log and "Failed: W_030" in log and "[SAVED]" in log
    assert "Failed dyads:" in (out / "batch.log").read_text() and "W_030" in (out / "batch.log").read_text()
    found = EB.discover_dyads(tree)
    for dy in ("W_003", "W_010"):
        z = np.load(out / f"{dy}_ffdtf.npz", allow_pickle=False)
        meta = json.loads(str(z["meta"]))
        assert [s["event"] for s in meta["segments"]] == ["Peppa", "Brave", "talk_1"]
        assert list(z["channels"]) == [f"{c}_{r}" for r in ("ch", "cg") for c in CHANS if c not in ("M1", "M2")]
        for seg in meta["segments"]:
            key = f"{seg['task']}/{seg['event']}"
            recs = {r: _reader(found[dy][seg["task"]][r]) for r in ("ch", "cg")}
            block, names, fs = EB.segment_block(recs["ch"], recs["cg"], seg["start_s"], seg["duration_s"], 1.0, 45.0)
            assert block.shape == (16, seg["samples"]) and fs == FS and seg["window"] == 256
            assert np.abs(block.mean(axis=1)).max() < 1e-12 and np.abs(block.std(axis=1) - 1).max() < 1e-12
            # inclusive cut start <= t <= start + duration (io_utils.py:148-150)
            assert seg["samples"] == int(np.sum((recs["ch"]["time"] >= seg["start_s"]) &
                                                (recs["ch"]["time"] <= seg["start_s"] + seg["duration_s"])))
            ff, starts = z[f"{key}/ffdtf"], z[f"{key}/starts"]
            assert seg["windows"] == (seg["samples"] - 256) // 128 + 1 == len(starts)
            # fixed 50 % hop; the tail shorter than one hop is dropped (sliding.hop_positions)
            assert starts.tolist() == list(range(0, 128 * len(starts), 128)) and 0 <= seg["samples"] - (starts[-1] + 256) < 128
            ref = np.stack([O.full_freq_dtf(block[:, s:s + 256], freqs, fs, 3) for s in starts])
            assert np.abs(ff - ref).max() / np.abs(ref).max() < 1e-9 and np.allclose(ff, ref, rtol=1e-5, atol=1e-12)
            lo, hi = hd.band_bins(freqs)
            bands = np.stack([ff[..., a:b].sum(-1) for a, b in zip(lo, hi)], axis=-1)
            assert np.allclose(z[f"{key}/ffdtf_bands"], bands, rtol=1e-12, atol=1e-300)
            assert z[f"{key}/psd"].shape == (16, len(z[f"{key}/psd_freqs"])) and (z[f"{key}/psd"] > 0).all()
    # second run: finished dyads are skipped, nothing is recomputed
    res2 = EB.run(tree, out, model_order=3, freqs=freqs, reader=_reader, verbose=False)
    assert res2["done"] == [] and sorted(res2["skipped"]) == ["W_003", "W_010", "W_020"]
    # rank sharding: two ranks split the four dyads 2 + 2 and write disjoint files
    r0 = EB.run(tree, tmp_path / "o2", model_order=3, freqs=freqs, reader=_reader, world=2, rank=0, verbose=False)
    r1 = EB.run(tree, tmp_path / "o2", model_order=3, freqs=freqs, reader=_reader, world=2, rank=1, verbose=False)
    assert r0["done"] == ["W_003", "W_010"] and r1["done"] == [] and [d for d, _ in r1["failed"]] == ["W_030"]
    assert (tmp_path / "o2" / "batch_rank1.log").exists()


def _write_full(path, seed, fs, dur_s, events, n_eeg=32):
    """one exported task file at BASELINE config-5 size: 32 EEG channels + the two mastoids at 500 Hz"""
    rng = np.random.default_rng(seed)
    chans = [f"E{k:02d}" for k in range(n_eeg)] + ["M1", "M2"]
    n = int(dur_s * fs)
    t = np.arange(n) / fs - 2.0
    x = rng.standard_normal((n, len(chans)))
    x[1:] += 0.7 * x[:-1]
    x[:, 1:n_eeg] += 0.2 * x[:, :n_eeg - 1]
    ev = [{"name": nm, "start_s": 50.0 + st, "start_rel_s": st, "duration_s": du} for nm, st, du in events]
    attrs = {"sampling_freq": fs, "task_events_structure": json.dumps(ev)}
    path.parent.mkdir(parents=True, exist_ok=True)
    with open(path, "wb") as f:
        np.savez(f, data=x, time=t, channels=np.asarray(chans), attrs=json.dumps(attrs), events=json.dumps(ev))


def make_config5_tree(root, dyads, fs=500.0, jitter=False):
    """SECORE (one 220 s event, /root/reference/src/secore_loader.py:117), passive movies (three films) and a free talk.
    jitter: every dyad's events last a different number of samples (up to 0.9 s more: the same window counts), as real
    event durations do -- every segment then has a length nobody met before (DPSS tapers, FFT plans, chirp-z tables)."""
    base = {"secore": (230.0, [("secore", 1.0, 220.0)]),
            "passive_movies": (215.0, [("Peppa", 1.0, 60.0), ("Incredibles", 70.0, 60.0), ("Brave", 140.0, 60.0)]),
            "talk": (190.0, [("talk_1", 1.0, 180.0)])}
    for d, dy in enumerate(dyads):
        tasks = base
        if jitter:
            tasks, e = {}, 0
            for task, (dur, evs) in base.items():
                out = []
                for name, start, length in evs:
                    out.append((name, start, length + ((37 * d + 113 * e) % 450) / fs))
                    e += 1
                tasks[task] = (dur, out)
        for r, (code, role) in enumerate((("ch", "child"), ("cg", "caregiver"))):
            for k, (task, (dur, ev)) in enumerate(tasks.items()):
                _write_full(Path(root) / "EEG" / dy / role / f"{dy}_EEG_{code}_{task}.nc", 1000 * d + 10 * k + r, fs, dur, ev)
    return Path(root)


from pathlib import Path  # noqa: E402


def test_config5_full_size_dyad(tmp_path):
    """ONE dyad at BASELINE config-5 size -- 2 x 32 channels at 500 Hz, SECORE 220 s + three 60 s films + 180 s of talk,
    2 s windows with 50 % overlap, p = 8, 256 frequencies, multitaper PSD beside it -- through the pipelined front-end:
    every task file filtered once, 5 segments, 575 windows; spot windows against the oracle on the block the front-end
    built, every ffDTF row of every window sums to one (band sums add up), the host / GPU time split is reported."""
    root = make_config5_tree(tmp_path / "tree", ["W_101"])
    out = tmp_path / "out"
    timing = {}
    # (band-pass 1 - 120 Hz for a 0.5 - 128 Hz grid.  With the 45 Hz low-pass that EEG work often uses, order-8 normal
    # equations of 500 Hz data have cond ~ 2e13: LAPACK's LU and Cholesky solves then differ by 2e-4 in the coefficients
    # and 8e-4 in ffDTF from one another -- no parity statement is possible there, whatever computes it)
    res = EB.run(root, out, window_s=2.0, overlap=0.5, model_order=8, low_cutoff_hz=1.0, high_cutoff_hz=120.0,
                 with_psd=True, psd_fmin=1.0, psd_fmax=30.0, psd_bandwidth=2.0, reader=_reader, verbose=False, timing=timing,
                 bands=((0.5, 4.0), (4.0, 8.0), (8.0, 13.0), (13.0, 30.0), (30.0, 128.5)))
    assert res["done"] == ["W_101"] and not res["failed"]
    z = np.load(out / "W_101_ffdtf.npz", allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    assert [s["event"] for s in meta["segments"]] == ["Peppa", "Incredibles", "Brave", "secore", "talk_1"]
    assert [s["windows"] for s in meta["segments"]] == [59, 59, 59, 219, 179] and not meta["failed_segments"]
    assert len(z["channels"]) == 64 and len(z["freqs"]) == 256
    found = EB.discover_dyads(root)
    for seg in (meta["segments"][0], meta["segments"][3]):
        key = f"{seg['task']}/{seg['event']}"
        recs = {r: _reader(found["W_101"][seg["task"]][r]) for r in ("ch", "cg")}
        block, names, fs = EB.segment_block(recs["ch"], recs["cg"], seg["start_s"], seg["duration_s"], 1.0, 120.0)
        assert block.shape == (64, seg["samples"]) and fs == 500.0
        bands, starts = z[f"{key}/ffdtf_bands"], z[f"{key}/starts"]
        assert bands.shape == (seg["windows"], 64, 64, 5) and np.isfinite(bands).all()
        # the five bands tile the whole grid: every row of every window sums to one
        assert np.abs(bands.sum(axis=(2, 3)) - 1.0).max() < 1e-12
        lo, hi = hd.band_bins(z["freqs"])
        for k in (0, len(starts) // 2, len(starts) - 1):
            ref = O.full_freq_dtf(block[:, starts[k]:starts[k] + 1000], z["freqs"], fs, 8)
            refb = np.stack([ref[..., a:b].sum(-1) for a, b in zip(lo, hi)], axis=-1)
            assert np.abs(bands[k] - refb).max() / np.abs(refb).max() < 1e-7          # cond ~ 3e6 here
        assert z[f"{key}/psd"].shape[0] == 64 and (z[f"{key}/psd"] > 0).all()
    assert timing["host_prepare_s"] > 0 and timing["gpu_s"] > 0 and 0 < timing["gpu_busy_fraction_of_wall"] <= 1
