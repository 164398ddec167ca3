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
