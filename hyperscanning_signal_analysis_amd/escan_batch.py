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

__all__ = ["discover_dyads", "decode_events", "segment_block", "prepare_dyad", "run", "savez_fast", "xarray_reader"]

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


def prepare_dyad(dyad, files_by_task, reader, low_cutoff_hz=None, high_cutoff_hz=None, channel_subset=None, say=None):
    """Host part of one dyad: every task file is read and filtered ONCE (both members), then cut into its events, z-scored
    per segment and stacked child on caregiver.  Returns {"segments": [{task, event, start_s, duration_s, block, names, fs}],
    "notes": [printed lines], "host_s": seconds}.  Pure NumPy / SciPy: runs in a worker thread beside the GPU work of the
    previous dyad (filtfilt releases the GIL)."""
    t0 = time.perf_counter()
    segs, notes = [], []
    for task, files in sorted(files_by_task.items()):
        if "ch" not in files or "cg" not in files:
            notes.append(f"[SKIP] {dyad} {task}: missing {'child' if 'ch' not in files else 'caregiver'} file")
            continue
        recs = {r: reader(files[r]) for r in ("ch", "cg")}
        filt = {}
        for r in ("ch", "cg"):
            x, ch, t, fs = _filtered(recs[r], low_cutoff_hz, high_cutoff_hz)         # whole file, once
            if channel_subset is not None:
                idx = [ch.index(c) for c in channel_subset if c in ch]
                if not idx:
                    raise ValueError(f"None of the requested channels {channel_subset} found. Available: {ch}")
                x, ch = x[:, idx], [ch[k] for k in idx]
            filt[r] = (x, ch, t, fs)
        if filt["ch"][3] != filt["cg"][3]:
            raise ValueError(f"sampling rates differ: {filt['ch'][3]} vs {filt['cg'][3]}")
        for name, start, dur in decode_events(recs["ch"]["attrs"]):
            parts, names = [], []
            for r in ("ch", "cg"):
                x, ch, t, fs = filt[r]
                mask = (t >= start) & (t <= start + dur)
                seg = np.ascontiguousarray(x[mask].T)
                sd = np.std(seg, axis=1, keepdims=True)
                sd[sd == 0] = 1.0
                parts.append((seg - np.mean(seg, axis=1, keepdims=True)) / sd)
                names += [f"{c}_{r}" for c in ch]
            T = min(q.shape[1] for q in parts)
            segs.append({"task": task, "event": name, "start_s": start, "duration_s": dur, "fs": filt["ch"][3],
                         "block": np.vstack([q[:, :T] for q in parts]), "names": names})
    return {"segments": segs, "notes": notes, "host_s": time.perf_counter() - t0}


def savez_fast(path, compresslevel=0, **arrays):
    """np.savez / np.savez_compressed with a choice of deflate level (same .npz container, np.load reads it).  float64
    connectivity values do not compress -- 5 % at NumPy's fixed level 6 (88.8 of 94.2 MB for the band sums of one
    full-size dyad) for 3.9 s of a host core, more than the dyad's preprocessing and 100x its GPU time; level 1 is hardly
    faster (3.4 s).  0 = stored (default), 1..9 = deflate."""
    import zipfile
    method = zipfile.ZIP_STORED if int(compresslevel) <= 0 else zipfile.ZIP_DEFLATED
    kw = {} if method == zipfile.ZIP_STORED else {"compresslevel": int(compresslevel)}
    with zipfile.ZipFile(path, mode="w", compression=method, allowZip64=True, **kw) as zf:
        for key, val in arrays.items():
            with zf.open(key + ".npy", mode="w", force_zip64=True) as f:
                np.lib.format.write_array(f, np.asanyarray(val), allow_pickle=False)


def run(root, out_dir, tasks=None, window_s=2.0, overlap=0.5, model_order=8, freqs=None, bands=hdist.DEFAULT_BANDS,
        low_cutoff_hz=None, high_cutoff_hz=None, channel_subset=None, with_psd=False, psd_fmin=1.0, psd_fmax=30.0,
        psd_bandwidth=2.0, save_full=False, skip_existing=True, reader=None, engine=None, world=1, rank=0,
        verbose=True, prefetch=2, timing=None, save_workers=4, compresslevel=0):
    """Process every dyad under <root>/EEG.  Per dyad one `<out_dir>/<dyad>_ffdtf.npz` with, per segment `<task>/<event>`:
        <seg>/ffdtf_bands   (windows, n, n, n_bands)   band-integrated ffDTF of every window
        <seg>/ffdtf         (windows, n, n, F)         only with save_full=True (8.4 MB per window at 64 channels)
        <seg>/starts        (windows,)                 first sample of every window inside the segment
        <seg>/psd, <seg>/psd_freqs                     multitaper PSD of the segment block (with_psd=True)
      plus `channels`, `freqs`, `meta` (JSON).  Returns {"done": [...], "skipped": [...], "failed": [(dyad, error)],
      "timing": {...}}.
    A window whose fit is singular is NaN-filled, not fatal (the reference would raise and lose the dyad); a segment that
    cannot be processed is logged in the dyad's meta and does not discard the dyad's other segments.
    Pipeline: reading + filtering + cutting of the next `prefetch` dyads runs in worker threads while the GPU works on the
    current one (every task file is filtered once, not once per event); per segment the block crosses PCIe once and the
    PSD runs on a second HIP stream beside the MVAR kernels.  `timing` (optional dict) receives the host / GPU split."""
    import concurrent.futures as cf

    import torch
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
    todo = []
    for dyad in mine:
        if skip_existing and (out_dir / f"{dyad}_ffdtf.npz").exists():
            say(f"[SKIP] {dyad}: {dyad}_ffdtf.npz exists")
            skipped.append(dyad)
        else:
            todo.append(dyad)
    tm = {"wall_s": 0.0, "host_prepare_s": 0.0, "wait_for_host_s": 0.0, "gpu_s": 0.0, "save_s": 0.0, "dyads": []}
    t_all = time.perf_counter()
    psd_stream = torch.cuda.Stream(eng.device) if with_psd else None
    # two pools: the writes (3 - 4 s of zlib per full-size dyad when compressed) must not take the workers that prepare the
    # next dyads, or the GPU waits for its input behind somebody's output
    pool = cf.ThreadPoolExecutor(max_workers=max(1, int(prefetch)))
    save_pool = cf.ThreadPoolExecutor(max_workers=max(1, int(save_workers)))
    futures, saves = {}, []

    def submit(k):
        if k < len(todo) and k not in futures:
            futures[k] = pool.submit(prepare_dyad, todo[k], tree[todo[k]], reader, low_cutoff_hz, high_cutoff_hz, channel_subset)

    for k in range(min(len(todo), max(1, int(prefetch)))):
        submit(k)
    try:
        for k, dyad in enumerate(todo):
            target = out_dir / f"{dyad}_ffdtf.npz"
            try:
                tw = time.perf_counter()
                prep = futures.pop(k).result()
                wait_s = time.perf_counter() - tw
                submit(k + max(1, int(prefetch)))
                for line in prep["notes"]:
                    say(line)
                result, meta = {}, {"dyad": dyad, "segments": [], "failed_segments": [], "model_order": int(model_order),
                                    "window_s": window_s, "overlap": overlap, "created": time.strftime("%Y-%m-%dT%H:%M:%S")}
                names_out, freqs_out = None, None
                tg = time.perf_counter()
                pending = []                                       # device results of this dyad, fetched after the last launch
                for seg in prep["segments"]:
                    key = f"{seg['task']}/{seg['event']}"
                    try:
                        block, fs = seg["block"], seg["fs"]
                        W = int(round(window_s * fs))
                        hop = max(1, int(round(W * (1.0 - overlap))))
                        T = block.shape[1]
                        if T < W:
                            say(f"[SKIP] {dyad} {key}: {T} samples < one window ({W})")
                            continue
                        pos = hop_positions(T, W, hop)            # fixed hop; the tail shorter than one hop is dropped
                        f = np.asarray(freqs if freqs is not None else np.arange(0.5, min(fs / 2.0, 128.0) + 1e-9, 0.5))
                        xd = eng.to_device(block[None])
                        rec_i, st_i = window_items(1, pos, eng.device)
                        psd_dev = None
                        if with_psd:                               # second stream, same resident block
                            from .psd import compute_psd_multitaper_device
                            psd_stream.wait_stream(torch.cuda.current_stream(eng.device))
                            with torch.cuda.stream(psd_stream):
                                pf, psd_dev = compute_psd_multitaper_device(xd[0], fs, psd_fmin, psd_fmax, psd_bandwidth, engine=eng)
                            xd.record_stream(psd_stream)
                        lo, hi = hdist.band_bins(f, bands)
                        grid_w = regular_grid(pos, W, int(model_order))
                        if save_full:
                            ff, bad = eng.sliding_ffdtf(xd, rec_i, st_i, W, int(model_order), f, fs, check="mask", grid=grid_w)
                            bsum = eng.band_sums(ff, lo, hi)
                            ff.masked_fill_(bad.view(-1, 1, 1, 1), float("nan"))
                        else:          # the reduced product straight from K3's row workers: the full array is never written
                            ff = None
                            bsum, bad = eng.sliding_ffdtf(xd, rec_i, st_i, W, int(model_order), f, fs, check="mask",
                                                          grid=grid_w, bands=(lo, hi))
                        bsum.masked_fill_(bad.view(-1, 1, 1, 1), float("nan"))
                        pending.append((seg, key, pos, W, f, bsum, ff if save_full else None, bad, psd_dev, pf if with_psd else None))
                    except Exception as e:                         # one bad segment does not discard the dyad
                        meta["failed_segments"].append({"segment": key, "error": f"{type(e).__name__}: {e}"})
                        say(f"Failed segment: {dyad} {key} -> {e}")
                if psd_stream is not None:
                    torch.cuda.current_stream(eng.device).wait_stream(psd_stream)
                for seg, key, pos, W, f, bsum, ff, bad, psd_dev, pf in pending:
                    result[f"{key}/ffdtf_bands"] = bsum.cpu().numpy()
                    if ff is not None:
                        result[f"{key}/ffdtf"] = ff.cpu().numpy()
                    result[f"{key}/starts"] = np.asarray(pos)
                    if psd_dev is not None:
                        result[f"{key}/psd"], result[f"{key}/psd_freqs"] = psd_dev.cpu().numpy(), pf
                    n_bad = int(bad.sum().item())
                    T = seg["block"].shape[1]
                    meta["segments"].append({"task": seg["task"], "event": seg["event"], "start_s": seg["start_s"],
                                             "duration_s": seg["duration_s"], "fs": seg["fs"], "samples": int(T),
                                             "windows": int(len(pos)), "window": int(W), "singular_windows": n_bad})
                    names_out, freqs_out = seg["names"], f
                    say(f"[OK] {dyad} {key}: {seg['block'].shape[0]} ch x {T} samples, {len(pos)} windows"
                        + (f", {n_bad} singular" if n_bad else ""))
                torch.cuda.synchronize(eng.device)
                gpu_s = time.perf_counter() - tg
                if not meta["segments"]:
                    say(f"[SKIP] {dyad}: no complete child + caregiver segment")
                    skipped.append(dyad)
                    continue
                # the write goes to a worker pool of its own (stored by default; deflate -- compresslevel > 0 -- costs 3 - 4 s of
                # a core per full-size dyad, zlib outside the GIL): the GPU and the next dyad's host work do not wait for it
                def _save(target=target, names_out=names_out, freqs_out=freqs_out, meta=meta, result=result):
                    t0 = time.perf_counter()
                    tmp = target.with_suffix(".tmp.npz")
                    savez_fast(tmp, compresslevel, channels=np.asarray(names_out), freqs=freqs_out,
                               bands=np.asarray(bands, dtype=np.float64), meta=np.asarray(json.dumps(meta)), **result)
                    tmp.replace(target)                       # a file that exists is complete (skip-if-exists relies on it)
                    return time.perf_counter() - t0
                saves.append((dyad, target, save_pool.submit(_save)))
                save_s = 0.0
                done.append(dyad)
                tm["host_prepare_s"] += prep["host_s"]; tm["wait_for_host_s"] += wait_s
                tm["gpu_s"] += gpu_s; tm["save_s"] += save_s
                tm["dyads"].append({"dyad": dyad, "host_prepare_s": prep["host_s"], "waited_for_host_s": wait_s,
                                    "gpu_s": gpu_s, "save_s": save_s, "segments": len(meta["segments"])})
            except Exception as e:                  # one bad dyad does not stop the batch (export_..._batch.py:93-99)
                failed.append((dyad, f"{type(e).__name__}: {e}"))
                say(f"Failed: {dyad} -> {e}")
                submit(k + max(1, int(prefetch)))
        for dyad, target, fut in saves:                   # every file on disk before the call returns
            try:
                tm["save_s"] += fut.result()
                say(f"[SAVED] {target}")
            except Exception as e:
                done.remove(dyad)
                failed.append((dyad, f"{type(e).__name__}: {e}"))
                say(f"Failed: {dyad} -> {e}")
    finally:
        pool.shutdown(wait=True, cancel_futures=True)
        save_pool.shutdown(wait=True)
    tm["wall_s"] = time.perf_counter() - t_all
    tm["gpu_busy_fraction_of_wall"] = tm["gpu_s"] / tm["wall_s"] if tm["wall_s"] > 0 else 0.0
    if timing is not None:
        timing.update(tm)
    say(f"Finished. Success: {len(done)}, Skipped: {len(skipped)}, Failed: {len(failed)}")
    with open(out_dir / ("batch.log" if world == 1 else f"batch_rank{rank}.log"), "a", encoding="utf-8") as log:
        log.write(f"Finished. Success: {len(done)}, Skipped: {len(skipped)}, Failed: {len(failed)}\n")
        if failed:
            log.write("Failed dyads:\n")
            for dyad, err in failed:
                log.write(f"  - {dyad}: {err}\n")
    return {"done": done, "skipped": skipped, "failed": failed, "timing": tm}
