"""Cross-modal FAA + IBI ffDTF pipeline with the interface of the reference's
`EEG_IBI_FFDTF_Pipeline` (/root/reference/src/eeg_alpha_ibi_ffdtf.py:29-806).

Same constructor arguments, method names, result keys and `.npz` layout; what changes is WHERE the
MVAR work runs: all windows of a dyad x film (plus the global segment) go to the MI355X as ONE batch
(K1 -> K2 -> K3 -> K4/K5, one fit shared by ffDTF and spectra) instead of one Python call per window
and product.  The scalar DSP in front of it (alpha band-pass, Hilbert envelope, FAA, polyphase
down-sampling, crop, z-score) stays on the host with the same SciPy calls the reference uses -- it is
O(n) work on 2 x 19 channels and is listed as "next" in section 8(f).

Extra, optional constructor arguments (defaults keep the drop-in behaviour):
    loader   callable(eeg_path, ibi_path, role) -> the 7-tuple of `_load_eeg_and_ibi`; default reads the
             reference's NetCDF files with xarray (imported lazily: xarray is not needed to import this
             module or to run the numerics).
    engine   a `hyperscanning_signal_analysis_amd.engine.Engine` (default: the process-wide one).
    batch_items  how many dyad x film items are pre-processed on the host before their MVAR work goes to the GPU as ONE
             batch (every window and every global block of all of them: one lag-covariance, one Yule-Walker, one
             transfer-function launch per block shape and model order, the AIC order selection included).  Default 256;
             1 = one dyad x film at a time, as the reference loops.  Results and files do not depend on it (bit for bit).
"""
from __future__ import annotations

import json
from datetime import datetime
from pathlib import Path

import numpy as np

from . import mtmvar
from .sliding import create_windows as _create_windows_impl

__all__ = ["EEG_IBI_FFDTF_Pipeline"]


class EEG_IBI_FFDTF_Pipeline:
    def __init__(self, cleaned_signals_folder: Path, output_ffDTF_folder: Path, target_events: list,
                 smoke_test: bool = False, smoke_dyads_n: int = 1,
                 left_frontal_eeg_channel: str = "F3", right_frontal_eeg_channel: str = "F4",
                 fs_downsampled: float = 8.0, n_windows: int = 3, window_size: int = None, ar_p: int = 5,
                 plot_global_enabled: bool = True, save_global_enabled: bool = True,
                 plot_windowed_enabled: bool = True, save_windowed_enabled: bool = True,
                 loader=None, engine=None, batch_items: int = 256):
        self.cleaned_signals_folder = Path(cleaned_signals_folder)
        self.output_ffDTF_folder = Path(output_ffDTF_folder)
        self.target_events = target_events
        self.smoke_test = smoke_test
        self.smoke_dyads_n = smoke_dyads_n
        self.left_chan = left_frontal_eeg_channel
        self.right_chan = right_frontal_eeg_channel
        self.fs_ds = float(fs_downsampled)
        # frequency grid 1.0 .. fs/2 - 0.1 in 0.1 Hz steps (eeg_alpha_ibi_ffdtf.py:100-103, quirk Q9)
        self.freq_min = 1.0
        self.freq_step = 0.1
        self.freq_max = self.fs_ds / 2.0 - self.freq_step
        self.n_windows = n_windows
        self.window_size = window_size
        self.ar_p = ar_p
        self.plot_global_enabled = plot_global_enabled
        self.save_global_enabled = save_global_enabled
        self.plot_windowed_enabled = plot_windowed_enabled
        self.save_windowed_enabled = save_windowed_enabled
        self._loader = loader
        self._engine = engine
        self.batch_items = max(1, int(batch_items))
        self.eeg_files = []
        self.ibi_files = []
        self.dyads_to_process = []
        self._prepare_file_lists()

    # ------------------------------------------------------------------ file discovery (host)
    @staticmethod
    def _dyad_of(path: Path) -> str:
        parts = path.stem.split("_")
        return f"{parts[0]}_{parts[1]}" if len(parts) >= 2 else path.stem

    def _prepare_file_lists(self):
        """Collect <root>/EEG/**/*.nc and <root>/IBI/**/*.nc for the target events (ref :122-180)."""
        found = {}
        dyads = set()
        for kind in ("EEG", "IBI"):
            folder = self.cleaned_signals_folder / kind
            files = sorted(p for p in folder.rglob("*.nc")
                           if kind in p.name and any(ev in p.name for ev in self.target_events))
            if not files:
                raise FileNotFoundError(f"No {kind} files found for events {self.target_events} under: {folder}")
            found[kind] = files
            dyads.update(self._dyad_of(p) for p in files)
        all_dyads = sorted(dyads)
        self.dyads_to_process = all_dyads[:self.smoke_dyads_n] if self.smoke_test else all_dyads
        keep = set(self.dyads_to_process)
        self.eeg_files = [p for p in found["EEG"] if self._dyad_of(p) in keep]
        self.ibi_files = [p for p in found["IBI"] if self._dyad_of(p) in keep]
        mode = f"SMOKE TEST (first {self.smoke_dyads_n} dyads)" if self.smoke_test else "FULL ANALYSIS"
        print(f"\n=== Initialization Complete: {mode} ===")
        print(f" [INFO] Target events : {self.target_events}")
        print(f" [INFO] Dyads loaded  : {len(self.dyads_to_process)}/{len(all_dyads)} ({', '.join(self.dyads_to_process)})")
        print(f" [OK]   EEG files     : {len(self.eeg_files)}/{len(found['EEG'])} ready")
        print(f" [OK]   IBI files     : {len(self.ibi_files)}/{len(found['IBI'])} ready\n")

    def _find_file(self, file_list, dyad, film, role):
        """(path, True) for the single file of dyad/film/role, (None, False) if absent (ref :183-199)."""
        hits = [f for f in file_list if dyad in f.name and f"_{film}" in f.name and f"_{role}_" in f.name]
        if not hits:
            return None, False
        if len(hits) > 1:
            raise ValueError(f"Found multiple files for dyad: {dyad}, film: {film}, role: {role} -> {hits}")
        return hits[0], True

    def _load_eeg_and_ibi(self, eeg_file, ibi_file, role):
        """(time_s, eeg (ch, n), fs_eeg, channel_names, ibi, fs_ibi, event_duration_s)  (ref :202-268)."""
        if self._loader is not None:
            return self._loader(eeg_file, ibi_file, role)
        try:
            import xarray as xr
        except ImportError as e:  # pragma: no cover
            raise ImportError("reading the reference's NetCDF files needs xarray + netCDF4; "
                              "pass loader=... to read another container") from e
        with xr.open_dataarray(eeg_file) as da:
            eeg = da.values.T.copy()
            time_s = da.coords["time"].values.copy()
            names = da.coords["channel"].values.tolist()
            dur = float(da.attrs["event_duration_s"])
            raw = da.attrs.get("sampling_freq") or da.attrs.get("sfreq")
            if raw is None:
                print(f" [WARN] {role}: Missing EEG sampling freq. Defaulting to 128.0 Hz")
            fs_eeg = 128.0 if raw is None else float(raw)
        with xr.open_dataarray(ibi_file) as da:
            ibi = da.values.T.copy()
            raw = da.attrs.get("sampling_freq") or da.attrs.get("sfreq")
            if raw is None:
                print(f" [INFO] {role}: Missing IBI sampling freq. Copying from EEG ({fs_eeg} Hz)")
            fs_ibi = fs_eeg if raw is None else float(raw)
        return time_s, eeg, fs_eeg, names, ibi, fs_ibi, dur

    # ------------------------------------------------------------------ scalar DSP (host, SciPy)
    def _alpha_bandpass_filter(self, data, fs, lowcut=8, highcut=12, order=4, axis=-1):
        """Zero-phase Butterworth band-pass, second-order sections (ref :271-311)."""
        from scipy.signal import butter, sosfiltfilt
        sos = butter(order, [lowcut / (0.5 * fs), highcut / (0.5 * fs)], btype="band", output="sos")
        return sosfiltfilt(sos, data, axis=axis)

    def _compute_asymmetry(self, filtered_eeg, channel_names, metric="amp"):
        """FAA(t) = log(env_right + 1e-12) - log(env_left + 1e-12), Hilbert envelopes (ref :314-365)."""
        from scipy.fft import next_fast_len
        from scipy.signal import hilbert
        try:
            li, ri = channel_names.index(self.left_chan), channel_names.index(self.right_chan)
        except ValueError as e:
            raise ValueError(f"Channels {self.left_chan} or {self.right_chan} not found: {e}")
        n = filtered_eeg.shape[1]
        nfft = next_fast_len(n)
        env = [np.abs(hilbert(filtered_eeg[k, :], N=nfft)[:n]) for k in (li, ri)]
        if metric == "power":
            env = [e ** 2 for e in env]
        elif metric != "amp":
            raise ValueError("metric must be 'power' or 'amp'")
        return np.log(env[1] + 1e-12) - np.log(env[0] + 1e-12)

    def _downsample_signal(self, signal, fs=128, fs_new=8):
        """Integer-factor polyphase decimation with its anti-aliasing FIR (ref :368-406)."""
        from scipy.signal import resample_poly
        if fs_new >= fs:
            raise ValueError("fs_new must be lower than fs")
        ratio = fs / fs_new
        if not np.isclose(ratio, round(ratio)):
            raise ValueError("fs must be divisible by fs_new")
        return resample_poly(signal, up=1, down=int(round(ratio)))

    def _crop_signal(self, signal, fs, drop_front_sec=10, keep_duration_sec=60):
        """Drop the first `drop_front_sec`, keep the next `keep_duration_sec` (ref :409-448)."""
        need = drop_front_sec + keep_duration_sec
        have = signal.shape[-1] / fs
        if have < need:
            raise ValueError(
                f"Cropping failed: The signal is too short. "
                f"It needs to be at least {need} seconds long, "
                f"but the provided signal is only {have:.2f} seconds.")
        return signal[..., int(drop_front_sec * fs):int(need * fs)]

    def _create_windows(self, signals, n_windows=3, window_size=None):
        """`n_windows` views spanning the signal; same positions and ValueErrors as the reference (:451-518)."""
        return _create_windows_impl(signals, n_windows, window_size)

    # ------------------------------------------------------------------ MVAR on the GPU
    def _freqs(self):
        return np.arange(self.freq_min, self.freq_max + self.freq_step, self.freq_step)

    def _order_for(self, signals, max_model_order, crit_type):
        if self.ar_p is None:
            _, _, p_opt = mtmvar.mvar_criterion(signals, max_model_order, crit_type, plot=False, engine=self._engine)
            return int(p_opt)
        return self.ar_p

    def _compute_ffDTF(self, dyad, signals, chan_names, fs, max_model_order=20, crit_type="AIC",
                       plot=True, save_plot=False, save_path=None, fig_name=None):
        """(ff_dtf, spectra, p_opt) of one segment (ref :521-634); ONE fit feeds both products."""
        freqs = self._freqs()
        p_opt = self._order_for(signals, max_model_order, crit_type)
        res = mtmvar.mvar_analysis(signals, freqs, fs, p_opt, want=("ffdtf", "spectra"), engine=self._engine)
        ff_dtf, spectra = res["ffdtf"], res["spectra"]
        if plot or save_plot:
            self._figure(dyad, spectra, ff_dtf, freqs, chan_names, plot, save_plot, save_path, fig_name)
        return ff_dtf, spectra, p_opt

    def _orders_for_blocks(self, blocks, max_model_order=20, crit_type="AIC"):
        """Model order of every block: the fixed `ar_p`, or the criterion's argmin -- for ALL blocks of one shape from one
        batched fit at `max_model_order` (the recursion's error covariances of every lower order: K2's `vq_logdet`)
        instead of one `mvar_criterion` call per block (ref :585-589, 726-728; mtmvar.py:551-601)."""
        if self.ar_p is not None:
            return [self.ar_p] * len(blocks)
        _, _, popt = self._criterion_for_blocks(blocks, max_model_order, crit_type)
        return popt

    def _criterion_for_blocks(self, blocks, max_model_order=20, crit_type="AIC"):
        """(crit curves, order range, argmin order) per block, batched by block shape; same numbers as
        `mtmvar.mvar_criterion` block by block."""
        import torch
        from .engine import default_engine
        if crit_type not in ("AIC", "HQ", "SC"):
            raise ValueError("Invalid criterion type. Choose from 'AIC', 'HQ', 'SC'.")
        eng = self._engine or default_engine()
        rng = np.arange(1, max_model_order + 1, dtype=int)
        crit_all, popt = [None] * len(blocks), [None] * len(blocks)
        shapes = {}
        for k, b in enumerate(blocks):
            shapes.setdefault(np.asarray(b).shape, []).append(k)
        for (m, n), idx in shapes.items():
            x = eng.to_device(np.stack([np.asarray(blocks[k], dtype=np.float64) for k in idx]))
            B = len(idx)
            rec = torch.arange(B, dtype=torch.int64, device=eng.device)
            st = torch.zeros(B, dtype=torch.int64, device=eng.device)
            R = eng.lagcov(x, rec, st, n, int(max_model_order))
            _, _, logdet, info = eng.yw_solve(R, m, want_logdet=True)
            eng.raise_on_info(info, "ar_coeff (Yule-Walker solve)")
            if crit_type == "AIC":
                pen = 2 * rng * m ** 2 / n
            elif crit_type == "HQ":
                pen = 2 * np.log(np.log(n)) * rng * m ** 2 / n
            else:
                pen = np.log(n) * rng * m ** 2 / n
            crit = logdet.cpu().numpy() + pen[None, :]
            for j, k in enumerate(idx):
                crit_all[k] = crit[j]
                popt[k] = int(rng[int(np.argmin(crit[j]))])          # first minimum, like the reference's argmin
        return crit_all, rng, popt

    def _compute_blocks(self, blocks, orders, fs):
        """ffDTF and spectra of many (m, n) blocks: every group of equal shape and order is ONE GPU batch (each block its
        own "recording": K1 -> K2 -> K3 -> K4 / K5).  Returns (ff list, spectra list) in the order of `blocks`."""
        import torch
        from .engine import default_engine
        eng = self._engine or default_engine()
        freqs = self._freqs()
        ff_out, sp_out = [None] * len(blocks), [None] * len(blocks)
        groups = {}
        for k, (b, p) in enumerate(zip(blocks, orders)):
            groups.setdefault((np.asarray(b).shape, int(p)), []).append(k)
        for ((m, n), p), idx in groups.items():
            x = eng.to_device(np.stack([np.asarray(blocks[k], dtype=np.float64) for k in idx]))
            B = len(idx)
            rec = torch.arange(B, dtype=torch.int64, device=eng.device)
            start = torch.zeros(B, dtype=torch.int64, device=eng.device)
            R = eng.lagcov(x, rec, start, n, p)
            ar, V, _, info = eng.yw_solve(R, m)
            eng.raise_on_info(info, "ar_coeff (Yule-Walker solve)")
            t = eng.transfer(ar, m, eng.twiddles(freqs, fs, p), want_P=True, want_H=True)
            eng.raise_on_info(t["info"], "mvar_transfer_function (inverse of A(f))", per_item=len(freqs))
            ff = eng.normalise(t["P"], t["rowsum"], m)[0].cpu().numpy()
            sp = eng.to_mmf_complex(eng.spectra(t["H"], V, m), m).cpu().numpy()
            for j, k in enumerate(idx):
                ff_out[k], sp_out[k] = ff[j], sp[j]
        return ff_out, sp_out

    def _compute_ffDTF_batch(self, windows, fs, max_model_order=20, crit_type="AIC"):
        """All windows of one dyad x film: (ff list, spectra list, orders)."""
        orders = self._orders_for_blocks(windows, max_model_order, crit_type)
        ff, sp = self._compute_blocks(windows, orders, fs)
        return ff, sp, orders

    def _figure(self, dyad, spectra, ff_dtf, freqs, chan_names, show, save, save_path, fig_name):
        """m x m grid: |S_ii(f)| on the diagonal, ffDTF_ij(f) (j -> i) elsewhere."""
        import matplotlib.pyplot as plt
        fig_name = fig_name or f"{dyad}_ffDTF.png"
        m = ff_dtf.shape[0]
        fig, axs = plt.subplots(m, m, figsize=(2.2 * m, 1.8 * m), squeeze=False)
        top = float(ff_dtf[~np.eye(m, dtype=bool)].max()) if m > 1 else 1.0
        for i in range(m):
            for j in range(m):
                ax = axs[i, j]
                if i == j:
                    ax.plot(freqs, np.abs(spectra[i, i, :]), color="k")
                else:
                    ax.fill_between(freqs, ff_dtf[i, j, :], color="C0")
                    ax.set_ylim(0, top)
                if i == 0:
                    ax.set_title(chan_names[j], fontsize=8)
                if j == 0:
                    ax.set_ylabel(chan_names[i], fontsize=8)
        fig.suptitle(fig_name[:-4])
        fig.tight_layout()
        if save and save_path is not None:
            Path(save_path).mkdir(parents=True, exist_ok=True)
            fig.savefig(Path(save_path) / fig_name, dpi=300, bbox_inches="tight")
        if show:
            plt.show()
        plt.close("all")

    # ------------------------------------------------------------------ output
    def _save_single_result(self, dyad, film, result):
        """One compressed .npz per dyad x film, same keys as the reference (:637-658)."""
        out_dir = Path(self.output_ffDTF_folder) / dyad
        out_dir.mkdir(parents=True, exist_ok=True)
        path = out_dir / f"{dyad}_{film}_ffDTF.npz"
        mv = result["mvar"]
        np.savez_compressed(
            path,
            ff_dtf_global=mv["ff_dtf_global"], spectra_global=mv["spectra_global"],
            ff_dtf_windowed=np.array(mv["ff_dtf_windowed"]), spectra_windowed=np.array(mv["spectra_windowed"]),
            p_opt_g=mv["p_opt_g"], p_opt_w=mv["p_opt_w"], meta=json.dumps(result["meta"]))
        print(f"[SAVED] {dyad} | {film} --> {path}\n")

    def _preprocess(self, eeg, fs_eeg, names, ibi, fs_ibi):
        faa = self._compute_asymmetry(self._alpha_bandpass_filter(eeg, fs_eeg), names, metric="amp")
        faa = self._crop_signal(self._downsample_signal(faa, fs_eeg, self.fs_ds), self.fs_ds, 10, 60)
        ibi = self._crop_signal(self._downsample_signal(np.squeeze(ibi), fs_ibi, self.fs_ds), self.fs_ds, 10, 60)
        return faa, ibi

    def _prepare_item(self, dyad, film):
        """Host part of one dyad x film (ref :669-719): find and load the four files, FAA + IBI -> z-scored 4 x 480 block.
        Returns None when a file is missing (the reference's [SKIP])."""
        print(f"--- Processing dyad: {dyad} | Film: {film} ---")
        paths, missing = {}, []
        for kind, files in (("EEG", self.eeg_files), ("IBI", self.ibi_files)):
            for role in ("ch", "cg"):
                path, ok = self._find_file(files, dyad, film, role)
                paths[(kind, role)] = path
                if not ok:
                    missing.append(f"{kind} ({role})")
        if missing:
            order = ["EEG (ch)", "IBI (ch)", "EEG (cg)", "IBI (cg)"]
            missing.sort(key=order.index)
            print(f" [SKIP] Missing files: {', '.join(missing)} -> Skipping {film}")
            return None
        for kind, role in (("EEG", "ch"), ("IBI", "ch"), ("EEG", "cg"), ("IBI", "cg")):
            print(f" [OK] Loaded {kind} ({role}) : {paths[(kind, role)].name}")
        _, eeg_ch, fs_eeg, names, ibi_ch, fs_ibi, _ = self._load_eeg_and_ibi(paths[("EEG", "ch")], paths[("IBI", "ch")], role="Child")
        _, eeg_cg, fs_eeg, names, ibi_cg, fs_ibi, _ = self._load_eeg_and_ibi(paths[("EEG", "cg")], paths[("IBI", "cg")], role="Care Giver")
        faa_ch, ibi_ch_c = self._preprocess(eeg_ch, fs_eeg, names, ibi_ch, fs_ibi)
        faa_cg, ibi_cg_c = self._preprocess(eeg_cg, fs_eeg, names, ibi_cg, fs_ibi)
        sig = np.vstack([faa_ch, ibi_ch_c, faa_cg, ibi_cg_c])
        sig = (sig - np.mean(sig, axis=1, keepdims=True)) / np.std(sig, axis=1, keepdims=True)
        print(" [OK] Pre-processing complete (Alpha -> FAA -> Downsample -> Crop -> Z-Score)")
        return {"dyad": dyad, "film": film, "sig": sig, "fs_eeg": fs_eeg,
                "windows": self._create_windows(sig, self.n_windows, self.window_size)}

    def _finish_items(self, items):
        """GPU part and output of a batch of prepared items: the windows and the global blocks of ALL of them as one batch
        per block shape and order (ref :726-772 once per item), then figures and one .npz per item (ref :774-800)."""
        names4 = ["faa_ch", "ibi_ch", "faa_cg", "ibi_cg"]
        if not items:
            return
        globals_ = [it["sig"] for it in items]
        if self.ar_p is not None:                       # informational, as in the reference (:726-728)
            _, _, suggested = self._criterion_for_blocks(globals_, 20, "AIC")
        blocks, owner = [], []
        for k, it in enumerate(items):
            for w in it["windows"]:
                blocks.append(w); owner.append((k, "w"))
            blocks.append(it["sig"]); owner.append((k, "g"))
        orders = self._orders_for_blocks(blocks)
        ff, sp = self._compute_blocks(blocks, orders, self.fs_ds)
        for k, it in enumerate(items):
            dyad, film = it["dyad"], it["film"]
            mine = [j for j, (kk, _) in enumerate(owner) if kk == k]
            jw, jg = [j for j in mine if owner[j][1] == "w"], [j for j in mine if owner[j][1] == "g"][0]
            if self.ar_p is not None:
                print(f" [INFO] AIC suggested p={suggested[k]} for global signal. Forcing fixed p={self.ar_p}.")
            out_dir = self.output_ffDTF_folder / dyad
            print(f" [INFO] Computing windowed ffDTF ({len(jw)} windows)...")
            ff_w, sp_w, p_w = [ff[j] for j in jw], [sp[j] for j in jw], [orders[j] for j in jw]
            if self.plot_windowed_enabled or self.save_windowed_enabled:
                for i in range(len(jw)):
                    self._figure(dyad, sp_w[i], ff_w[i], self._freqs(), names4, self.plot_windowed_enabled,
                                 self.save_windowed_enabled, out_dir, f"{dyad}_{film}_win{i}_ffDTF.png")
            print(" [INFO] Computing global ffDTF...")
            ff_g, sp_g, p_g = ff[jg], sp[jg], orders[jg]
            if self.plot_global_enabled or self.save_global_enabled:
                self._figure(dyad, sp_g, ff_g, self._freqs(), names4, self.plot_global_enabled, self.save_global_enabled,
                             out_dir, f"{dyad}_{film}_ffDTF_global.png")
            result = {
                "mvar": {"ff_dtf_global": ff_g, "spectra_global": sp_g, "ff_dtf_windowed": ff_w,
                         "spectra_windowed": sp_w, "p_opt_g": p_g, "p_opt_w": p_w},
                "meta": {"dyad": dyad, "film": film, "fs": self.fs_ds, "fs_original": it["fs_eeg"],
                         "chan_names": names4, "faa_chan_names": (self.left_chan, self.right_chan),
                         "windowing": {"n_windows": self.n_windows, "window_size": self.window_size},
                         "computed_at": datetime.now().isoformat()},
            }
            self._save_single_result(dyad, film, result)

    def run_pipeline(self):
        """Per dyad x film: load 4 files, FAA + IBI -> 4 x 480 z-scored block, windowed + global ffDTF / spectra, save
        (ref :661-806).  Missing files are skipped with a [SKIP] line.  The host part runs item by item; the MVAR work of
        up to `batch_items` items goes to the GPU together (a 4-channel, 160-sample window is ~1 us of arithmetic: one item
        at a time is pure launch latency); with batch_items = 1 the order of the printed lines is the reference's."""
        if not self.dyads_to_process:
            raise RuntimeError("No loaded dyads. Check the files.")
        items = []
        for dyad in self.dyads_to_process:
            for film in self.target_events:
                it = self._prepare_item(dyad, film)
                if it is None:
                    continue
                items.append(it)
                if len(items) >= self.batch_items:
                    self._finish_items(items)
                    items = []
        self._finish_items(items)
