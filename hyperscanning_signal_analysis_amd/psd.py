"""Multitaper PSD on the GPU with the reference's call signature (src/psd.py:7-50).

PARITY UNPINNED: the reference delegates to mne==1.11.0 (`psd_array_multitaper`), which is not available
offline; the kernel path restates that algorithm with mne's defaults (see include/hypermvar.h,
`hmv_psd_multitaper_f64`, and oracle/psd_oracle.py).  The DPSS tapers come from the device too (`hmv_dpss_f64`: the
algorithm of scipy.signal.windows.dpss, which mne calls, restated for the GPU) and are cached per (n_times,
time-half-bandwidth) in memory (on disk too when $HYPERMVAR_DPSS_CACHE names a directory); everything per sample runs on
the device:
taper products, batched real-to-complex FFTs (hipFFT) and the eigenvalue-weighted power sum.
"""
from __future__ import annotations

import os
import tempfile

import numpy as np
import torch

from . import _lib
from .engine import default_engine

__all__ = ["compute_psd_multitaper", "compute_psd_multitaper_device", "average_psd_across_conditions", "dpss_device"]


def _taper_cache_dir():
    """Where DPSS tapers are kept between processes: $HYPERMVAR_DPSS_CACHE (unset or '' = no disk cache).  Opt-in since
    the tapers of a length come off the GPU in ~0.15 s: reading 390 MB (439 tapers x 110 000 points) back from disk is
    not faster, and a batch of real recordings -- every segment a length of its own -- would write that much per segment."""
    return os.environ.get("HYPERMVAR_DPSS_CACHE") or None


def dpss_device(n_times: int, half_nbw: float, k_max: int, sym: bool = False, engine=None):
    """(tapers (k_max, n_times), concentration ratios (k_max,)) as device tensors: scipy.signal.windows.dpss(n_times,
    half_nbw, k_max, sym=sym, norm=2, return_ratios=True) computed on the GPU (`hmv_dpss_f64`)."""
    eng = engine or default_engine()
    nbytes = int(eng.lib.hmv_dpss_workspace_bytes(int(n_times), int(k_max), int(bool(sym))))
    if nbytes < 0:
        raise ValueError("dpss: bad n_times / k_max")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=eng.device)
    tapers = eng.empty(int(k_max), int(n_times))
    ratios = eng.empty(int(k_max))
    with torch.cuda.device(eng.device):
        rc = eng.lib.hmv_dpss_f64(int(n_times), float(half_nbw), int(k_max), int(bool(sym)), tapers.data_ptr(), ratios.data_ptr(),
                                  ws.data_ptr(), nbytes, eng.stream())
    _lib.check(rc, "hmv_dpss_f64")
    return tapers, ratios


_TAPERS = {}        # (device, n_times, half_nbw) -> (tapers, weights) on that device


def _tapers(n_times: int, half_nbw: float, eng):
    """(tapers (K, n_times), sqrt(eigenvalues)) on the engine's device, as mne selects them (low_bias: eigenvalue > 0.9).
    A batch meets the same (length, bandwidth) again and again: cached in memory per process and on disk across
    processes and ranks (plain .npz, written atomically); computed on the GPU otherwise (SciPy's host DPSS costs 55 s
    for a 220 s segment at 500 Hz)."""
    key = (str(eng.device), int(n_times), float(half_nbw))
    hit = _TAPERS.get(key)
    if hit is not None:
        return hit
    path = None
    cdir = _taper_cache_dir()
    if cdir is not None:
        path = os.path.join(cdir, f"dpss_n{n_times}_nw{half_nbw!r}.npz")
        if os.path.exists(path):
            try:
                with np.load(path, allow_pickle=False) as z:
                    out = (eng.to_device(z["tapers"]), eng.to_device(z["weights"]))
                    _TAPERS[key] = out
                    return out
            except Exception:          # unreadable or half-written file: recompute
                pass
    k_max = max(int(2 * half_nbw), 1)
    tapers, eig = dpss_device(n_times, half_nbw, k_max, False, eng)
    eig_h = eig.cpu().numpy()                          # k_max numbers: selection and weights on the host
    keep = eig_h > 0.9                                 # low_bias=True
    if not keep.any():
        keep = np.zeros_like(keep)
        keep[np.argmax(eig_h)] = True
    idx = torch.as_tensor(np.flatnonzero(keep)).to(eng.device)
    out = (tapers.index_select(0, idx).contiguous(), eng.to_device(np.sqrt(eig_h[keep])))
    if len(_TAPERS) >= 8:
        _TAPERS.pop(next(iter(_TAPERS)))
    _TAPERS[key] = out
    if path is not None:
        try:
            os.makedirs(cdir, exist_ok=True)
            tmp = f"{path}.{os.getpid()}.tmp"
            with open(tmp, "wb") as f:
                np.savez(f, tapers=out[0].cpu().numpy(), weights=out[1].cpu().numpy())
            os.replace(tmp, path)
        except OSError:                # read-only or full disk: the cache is an optimisation only
            pass
    return out


def compute_psd_multitaper_device(xd: torch.Tensor, sfreq, fmin, fmax, bandwidth, max_workspace_bytes: int = 8 << 30,
                                  engine=None):
    """(freqs (host), psd (device tensor (n_channels, n_freqs))) of a block that already sits on the device, on the
    CURRENT stream and without a host copy: what the batch front-end runs beside the MVAR work of the same block."""
    eng = engine or default_engine()
    assert xd.dim() == 2 and xd.dtype == torch.float64 and xd.is_cuda and xd.stride(1) == 1
    n_ch, n_times = xd.shape
    half_nbw = float(bandwidth) * n_times / (2.0 * float(sfreq))
    td, wd = _tapers(n_times, half_nbw, eng)
    K = td.shape[0]
    freqs = np.fft.rfftfreq(n_times, 1.0 / float(sfreq))
    sel = np.flatnonzero((freqs >= fmin) & (freqs <= fmax))
    if sel.size == 0:
        return freqs[sel], eng.empty(n_ch, 0)
    lo, hi = int(sel[0]), int(sel[-1])
    per_ch = int(eng.lib.hmv_psd_workspace_bytes(1, n_times, K))
    ch_chunk = int(max(1, min(n_ch, max_workspace_bytes // max(per_ch, 1))))
    nbytes = int(eng.lib.hmv_psd_workspace_bytes(ch_chunk, n_times, K))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=eng.device)
    out = eng.empty(n_ch, hi - lo + 1)
    with torch.cuda.device(eng.device):
        rc = eng.lib.hmv_psd_multitaper_f64(xd.data_ptr(), n_ch, n_times, xd.stride(0), td.data_ptr(), wd.data_ptr(), K,
                                            lo, hi, out.data_ptr(), ws.data_ptr(), nbytes, ch_chunk, eng.stream())
    _lib.check(rc, "hmv_psd_multitaper_f64")
    ws.record_stream(torch.cuda.current_stream(eng.device))
    return freqs[lo:hi + 1], out


def compute_psd_multitaper(data, sfreq, fmin, fmax, bandwidth, max_workspace_bytes: int = 8 << 30, engine=None):
    """(freqs, psd): psd (n_channels, n_freqs) on fmin <= f <= fmax, like src/psd.py:7-33."""
    eng = engine or default_engine()
    x = np.ascontiguousarray(np.asarray(data, dtype=np.float64))
    if x.ndim != 2:
        raise ValueError("data must have shape (n_channels, n_times)")
    n_ch, n_times = x.shape
    half_nbw = float(bandwidth) * n_times / (2.0 * float(sfreq))
    td, wd = _tapers(n_times, half_nbw, eng)
    K = td.shape[0]
    freqs = np.fft.rfftfreq(n_times, 1.0 / float(sfreq))
    sel = np.flatnonzero((freqs >= fmin) & (freqs <= fmax))
    if sel.size == 0:
        return freqs[sel], np.zeros((n_ch, 0))
    lo, hi = int(sel[0]), int(sel[-1])
    per_ch = int(eng.lib.hmv_psd_workspace_bytes(1, n_times, K))
    ch_chunk = int(max(1, min(n_ch, max_workspace_bytes // max(per_ch, 1))))
    nbytes = int(eng.lib.hmv_psd_workspace_bytes(ch_chunk, n_times, K))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=eng.device)
    xd = eng.to_device(x)
    out = eng.empty(n_ch, hi - lo + 1)
    with torch.cuda.device(eng.device):
        rc = eng.lib.hmv_psd_multitaper_f64(xd.data_ptr(), n_ch, n_times, n_times, td.data_ptr(), wd.data_ptr(), K,
                                            lo, hi, out.data_ptr(), ws.data_ptr(), nbytes, ch_chunk, eng.stream())
    _lib.check(rc, "hmv_psd_multitaper_f64")
    return freqs[lo:hi + 1], out.cpu().numpy()


def average_psd_across_conditions(psd_dict):
    """Arithmetic mean of equally shaped PSD arrays (src/psd.py:36-50)."""
    if not psd_dict:
        raise ValueError('psd_dict is empty; no conditions to average PSD over.')
    return np.mean(np.stack(list(psd_dict.values()), axis=0), axis=0)
