"""ESCan batch front-end: exported per-task EEG NetCDF files of many dyads -> sliding-window ffDTF (and multitaper
PSD) per dyad x task segment on the MI355X.  BASELINE.json config 5 / SURVEY.md section 8(f) row 2.

What it mirrors of the reference (nothing here is on the GPU hot path; it FEEDS it):
  * file layout written by `export_passive_and_talk_data` (/root/reference/src/export.py:467-609):
        <root>/EEG/<DYAD>/<child|caregiver>/<DYAD>_EEG_<ch|cg>_<task>.nc      (variable `signals`, dims (time, channel))
    with attrs `sampling_freq`, `task_events_structure` (a list of {name, start_s, start_rel_s, duration_s}, possibly
    JSON-encoded: /root/reference/src/export.py:246-288, 328-340; decoding as /root/reference/src/ncdf.py:5-19, 88-91);
  * segmenting a task file by its events like `load_eeg_nc` + `trim_to_event_window`
    (/root/reference/src/io_utils.py:113-118, 130-150: `start_rel_s`, inclusive `start <= t <= start + duration`) and the
    region logic of `_build_task_regions_from_xarray` (/root/reference/src/ncdf.py:93-111);
  * preprocessing with the semantics of `load_eeg_signals` (/root/reference/src/mne_bridge.py:113-223: zero-phase
    Butterworth-4 high / low pass and the 50 Hz notch on the WHOLE file, then the cut, M1/M2 dropped, per-channel z-score),
    child block stacked on caregiver block as in BASELINE config 2;
  * windows by the reference's `_create_windows` rule (src/eeg_alpha_ibi_ffdtf.py:451-518) with
    n_windows = 2 T / W - 1 (2 s windows, 50 % overlap when T is a multiple of the hop);
  * per dyad output file + batch bookkeeping of the reference's drivers: results saved immediately per dyad
    (src/eeg_alpha_ibi_ffdtf.py:637-658), `[SKIP]` for missing inputs (:674-684), failed dyads collected and appended to
    a log file (/root/reference/scripts/export_dyade_to_ncdf_by_task_batch.py:93-116).  Skip-if-exists is an addition.

The NetCDF reader is injectable (`reader=`): xarray / netCDF4 are not installed in the build image, so the default
reader imports them lazily and the tests inject a NumPy reader over a synthetic tree.

Multi-GPU: dyads are sharded over ranks (`world`, `rank`), no collective; every rank writes its own dyad files.
"""
from __future__ import annotations

import json
import re
import time
from pathlib import Path

import numpy as np

from . import distributed as hdist
from .eeg_io import filter_eeg
from .engine import default_engine
from .sliding import hop_positions, regular_grid, window_items

__all__ = ["discover_dyads", "decode_events", "segment_block", "run", "xarray_reader"]

ROLES = (("ch", "child"), ("cg", "caregiver"))
_FILE_RE = re.compile(r"^(?P<dyad>.+)_EEG_(?P<role>ch|cg)_(?P<task>.+)$")


def xarray_reader(path):
    """Default reader: {data_tc (time, channel), time, channels, attrs} of one exported file (needs xarray + netCDF4)."""
    try:
        import xarray as xr
    except ImportError as e:  # pragma: no cover
        raise ImportError("reading the reference's .nc files needs xarray + netCDF4; pass reader= otherwise") from e
    da = xr.load_dataarray(str(path))
    return {"data_tc": da.transpose("time", "channel").values, "time": da.coords["time"].values,
            "channels": [str(c) for c in da.coords["channel"].values], "attrs": dict(da.attrs)}


def discover_dyads(root, tasks=None):
    """{dyad: {task: {"ch": path, "cg": path}}} for every <root>/EEG/<dyad>/<child|caregiver>/*_EEG_<ch|cg>_<task>.nc."""
    base = Path(root) / "EEG"
    found = {}
    if not base.is_dir():
        raise FileNotFoundError(f"EEG folder not found: {base}")
    for p in sorted(base.rglob("*.nc")):
        m = _FILE_RE.match(p.stem)
        if m is None:
            continue
        if tasks is not None and m.group("task") not in tasks:
            continue
        found.setdefault(m.group("dyad"), {}).setdefault(m.group("task"), {})[m.group("role")] = p
    return found


def decode_events(attrs):
    """task_events_structure -> [(name, start_rel_s, duration_s)]; accepts the list or its JSON string (ncdf.py:95-97)."""
    ev = attrs.get("task_events_structure", [])
    if isinstance(ev, (str, bytes)):
        ev = json.loads(ev) if str(ev).strip() else []
    out = []
    for k, e in enumerate(ev or []):
        if not isinstance(e, dict):
            continue
        start = e.get("start_rel_s", e.get("start_s", 0.0))
        out.append((str(e.get("name", f"event_{k + 1}")), float(start), float(e.get("duration_s", 0.0))))
    return out


def _filtered(rec, low_cutoff_hz, high_cutoff_hz):
    """Whole-file filtering + mastoid drop with load_eeg_signals' semantics, WITHOUT its trim and z-score."""
    fs = float(rec["attrs"].get("sampling_freq", rec["attrs"].get("sampling_frequency_Hz", 128.0)))
    x = filter_eeg(rec["data_tc"], fs, low_cutoff_hz, high_cutoff_hz)
    names = [str(c) for c in rec["channels"]]
    keep = [k for k, c in enumerate(names) if c not in ("M1", "M2")]
    return x[:, keep], [names[k] for k in keep], np.asarray(rec["time"], dtype=np.float64), fs


def segment_block(child, caregiver, start_s, duration_s, low_cutoff_hz=None, high_cutoff_hz=None, channel_subset=None):
    """(block (2 n_ch, T) z-scored, names, fs) of one event: both members filtered over their whole file, cut to
    start <= t <= start + duration (io_utils.py:148-150), z-scored per channel inside the segment, stacked child
    first on their common length."""
    parts, names, fss = [], [], []
    for role, rec in (("ch", child), ("cg", caregiver)):
        x, ch, t, fs = _filtered(rec, low_cutoff_hz, high_cutoff_hz)
        if channel_subset is not None:
            idx = [ch.index(c) for c in channel_subset if c in ch]
            if not idx:
                raise ValueError(f"None of the requested channels {channel_subset} found. Available: {ch}")
            x, ch = x[:, idx], [ch[k] for k in idx]
        mask = (t >= start_s) & (t <= start_s + duration_s)
        seg = np.ascontiguousarray(x[mask].T)
        sd = np.std(seg, axis=1, keepdims=True)
        sd[sd == 0] = 1.0
        parts.append((seg - np.mean(seg, axis=1, keepdims=True)) / sd)
        names += [f"{c}_{role}" for c in ch]
        fss.append(fs)
    if fss[0] != fss[1]:
        raise ValueError(f"sampling rates differ: {fss[0]} vs {fss[1]}")
    T = min(p.shape[1] for p in parts)
    return np.vstack([p[:, :T] for p in parts]), names, fss[0]


def run(root, out_dir, tasks=None, window_s=2.0, overlap=0.5, model_order=8, freqs=None, bands=hdist.DEFAULT_BANDS,
        low_cutoff_hz=None, high_cutoff_hz=None, channel_subset=None, with_psd=False, psd_fmin=1.0, psd_fmax=30.0,
        psd_bandwidth=2.0, save_full=False, skip_existing=True, reader=None, engine=None, world=1, rank=0,
        verbose=True):
    """Process every dyad under <root>/EEG.  Per dyad one `<out_dir>/<dyad>_ffdtf.npz` with, per segment `<task>/<event>`:
        <seg>/ffdtf_bands   (windows, n, n, n_bands)   band-integrated ffDTF of every window
        <seg>/ffdtf         (windows, n, n, F)         only with save_full=True (8.4 MB per window at 64 channels)
        <seg>/starts        (windows,)                 first sample of every window inside the segment
        <seg>/psd, <seg>/psd_freqs                     multitaper PSD of the segment block (with_psd=True)
      plus `channels`, `freqs`, `meta` (JSON).  Returns {"done": [...], "skipped": [...], "failed": [(dyad, error)]}.
    A window whose fit is singular is NaN-filled, not fatal (the reference would raise and lose the dyad)."""
    eng = engine or default_engine()
    reader = reader or xarray_reader
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    say = print if verbose else (lambda *a, **k: None)
    tree = discover_dyads(root, tasks)
    dyads = sorted(tree)
    mine = [dyads[k] for k in hdist.shard_dyads(len(dyads), world, rank)]
    say(f"[INFO] {len(dyads)} dyads under {root}; rank {rank}/{world} takes {len(mine)}")
    done, skipped, failed = [], [], []
    for dyad in mine:
        target = out_dir / f"{dyad}_ffdtf.npz"
        if skip_existing and target.exists():
            say(f"[SKIP] {dyad}: {target.name} exists")
            skipped.append(dyad)
            continue
        try:
            result, meta = {}, {"dyad": dyad, "segments": [], "model_order": int(model_order), "window_s": window_s,
                                "overlap": overlap, "created": time.strftime("%Y-%m-%dT%H:%M:%S")}
            names_out, freqs_out = None, None
            for task, files in sorted(tree[dyad].items()):
                if "ch" not in files or "cg" not in files:
                    say(f"[SKIP] {dyad} {task}: missing {'child' if 'ch' not in files else 'caregiver'} file")
                    continue
                recs = {r: reader(files[r]) for r in ("ch", "cg")}
                for name, start, dur in decode_events(recs["ch"]["attrs"]):
                    block, names, fs = segment_block(recs["ch"], recs["cg"], start, dur, low_cutoff_hz, high_cutoff_hz,
                                                     channel_subset)
                    W = int(round(window_s * fs))
                    hop = max(1, int(round(W * (1.0 - overlap))))
                    T = block.shape[1]
                    if T < W:
                        say(f"[SKIP] {dyad} {task}/{name}: {T} samples < one window ({W})")
                        continue
                    pos = hop_positions(T, W, hop)       # fixed hop; the tail shorter than one hop is dropped
                    n_win = len(pos)
                    f = np.asarray(freqs if freqs is not None else np.arange(0.5, min(fs / 2.0, 128.0) + 1e-9, 0.5))
                    xd = eng.to_device(block[None])
                    rec_i, st_i = window_items(1, pos, eng.device)
                    ff = eng.sliding_ffdtf(xd, rec_i, st_i, W, int(model_order), f, fs, check="nan",
                                           grid=regular_grid(pos, W, int(model_order)))
                    lo, hi = hdist.band_bins(f, bands)
                    key = f"{task}/{name}"
                    result[f"{key}/ffdtf_bands"] = eng.band_sums(ff, lo, hi).cpu().numpy()
                    if save_full:
                        result[f"{key}/ffdtf"] = ff.cpu().numpy()
                    result[f"{key}/starts"] = np.asarray(pos)
                    if with_psd:
                        from .psd import compute_psd_multitaper
                        pf, psd = compute_psd_multitaper(block, fs, psd_fmin, psd_fmax, psd_bandwidth, engine=eng)
                        result[f"{key}/psd"], result[f"{key}/psd_freqs"] = psd, pf
                    n_bad = int(np.isnan(result[f"{key}/ffdtf_bands"]).any(axis=(1, 2, 3)).sum())
                    meta["segments"].append({"task": task, "event": name, "start_s": start, "duration_s": dur, "fs": fs,
                                             "samples": int(T), "windows": int(n_win), "window": int(W),
                                             "singular_windows": n_bad})
                    names_out, freqs_out = names, f
                    say(f"[OK] {dyad} {key}: {block.shape[0]} ch x {T} samples, {n_win} windows"
                        + (f", {n_bad} singular" if n_bad else ""))
            if not meta["segments"]:
                say(f"[SKIP] {dyad}: no complete child + caregiver segment")
                skipped.append(dyad)
                continue
            np.savez_compressed(target, channels=np.asarray(names_out), freqs=freqs_out,
                                bands=np.asarray(bands, dtype=np.float64), meta=json.dumps(meta), **result)
            say(f"[SAVED] {target}")
            done.append(dyad)
        except Exception as e:                      # one bad dyad does not stop the batch (export_..._batch.py:93-99)
            failed.append((dyad, f"{type(e).__name__}: {e}"))
            say(f"Failed: {dyad} -> {e}")
    say(f"Finished. Success: {len(done)}, Skipped: {len(skipped)}, Failed: {len(failed)}")
    with open(out_dir / ("batch.log" if world == 1 else f"batch_rank{rank}.log"), "a", encoding="utf-8") as log:
        log.write(f"Finished. Success: {len(done)}, Skipped: {len(skipped)}, Failed: {len(failed)}\n")
        if failed:
            log.write("Failed dyads:\n")
            for dyad, err in failed:
                log.write(f"  - {dyad}: {err}\n")
    return {"done": done, "skipped": skipped, "failed": failed}
