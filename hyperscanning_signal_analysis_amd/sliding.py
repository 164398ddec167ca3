"""Batched sliding-window ffDTF: the (dyad x window) batch dimension the GPU path is built around.

`window_positions` / `create_windows` follow `EEG_IBI_FFDTF_Pipeline._create_windows`
(/root/reference/src/eeg_alpha_ibi_ffdtf.py:451-518) -- same start positions, same ValueErrors.
`sliding_ffdtf` is the batched equivalent of calling `full_freq_dtf(window, freqs, fs,
optimal_model_order=p)` (/root/reference/src/mtmvar.py:237-284) on every window of every recording.
"""
from __future__ import annotations

import numpy as np
import torch

from .engine import Engine, default_engine

__all__ = ["window_positions", "hop_positions", "create_windows", "sliding_ffdtf", "sliding_ffdtf_device", "window_items",
           "regular_grid"]


def window_positions(T: int, n_windows: int = 3, window_size=None):
    """Start positions and window length of `_create_windows` (eeg_alpha_ibi_ffdtf.py:451-518): `n_windows` windows of
    `window_size` samples whose starts are spread evenly (integer-truncated) from 0 to T - window_size, so the last
    window ends exactly at T.  Same ValueError texts as the reference (pinned by tests/golden/g6_errors.npz)."""
    T, n_windows = int(T), int(n_windows)
    if window_size is None:
        window_size, rest = divmod(T, n_windows)
        if rest:
            raise ValueError(
                f"Cannot evenly divide signal of length {T} into {n_windows} "
                f"non-overlapping windows. Provide a specific window_size."
            )
    elif window_size > T:
        # (the reference tests "too short" first; a size cannot be both, so the order is immaterial)
        raise ValueError(f"window_size ({window_size}) cannot exceed signal length ({T}).")
    elif window_size * n_windows < T:          # fewer than ceil(T / n_windows) samples per window: gaps
        raise ValueError(
            f"window_size={window_size} is too short. To cover {T} samples "
            f"with {n_windows} windows without leaving gaps, the minimum "
            f"window_size is {-(-T // n_windows)}."
        )
    last_start = T - window_size
    if n_windows > 1 and last_start < n_windows - 1:
        raise ValueError(
            f"window_size={window_size} is too large to generate {n_windows} "
            f"distinct windows. Decrease window_size or n_windows."
        )
    starts = np.linspace(0, last_start, n_windows, dtype=int) if n_windows > 1 else np.zeros(1, dtype=int)
    return starts, int(window_size)


def hop_positions(T: int, window_size: int, hop: int):
    """Starts 0, hop, 2 hop, ... of every whole window of `window_size` samples inside T samples (the tail shorter than
    one hop is dropped).  This is the fixed-overlap grid of BASELINE config 2 / 5 ("2 s windows, 50 % overlap"); unlike
    `window_positions` it does not stretch the last window to the end of the recording, so it never refuses a length."""
    T, window_size, hop = int(T), int(window_size), int(hop)
    if window_size < 1 or hop < 1:
        raise ValueError("window_size and hop must be positive")
    if T < window_size:
        return np.zeros(0, dtype=int)
    return np.arange((T - window_size) // hop + 1, dtype=int) * hop


def create_windows(signals, n_windows=3, window_size=None):
    """List of `n_windows` views (channels, window_size), as the reference returns."""
    positions, w = window_positions(signals.shape[1], n_windows, window_size)
    return [signals[:, s:s + w] for s in positions]


def window_items(n_rec: int, positions, device):
    """(item_rec, item_start) int64 device tensors for `n_rec` recordings sharing the same positions."""
    pos = torch.as_tensor(np.asarray(positions, dtype=np.int64))
    item_start = pos.repeat(n_rec).to(device)
    item_rec = torch.arange(n_rec, dtype=torch.int64).repeat_interleave(len(pos)).to(device)
    return item_rec, item_start


def regular_grid(positions, window_size: int, p: int):
    """(hop, first, n_win) if the start positions are an arithmetic progression whose step divides the window into
    2..8 whole hops longer than the model order -- the case in which K1 can share the overlap between windows
    (`Engine.sliding_ffdtf(grid=...)`) -- else None."""
    pos = np.asarray(positions, dtype=np.int64)
    if len(pos) < 2:
        return None
    hop = int(pos[1] - pos[0])
    if hop <= int(p) or hop < 1 or not np.array_equal(np.diff(pos), np.full(len(pos) - 1, hop)):
        return None
    if window_size % hop != 0 or not 2 <= window_size // hop <= 8:
        return None
    return hop, int(pos[0]), len(pos)


def sliding_ffdtf_device(x: torch.Tensor, window_size: int, n_windows: int, p: int, freqs, fs: float,
                         engine: Engine | None = None, out: torch.Tensor | None = None, check: bool = True,
                         share_overlap: bool = True):
    """x: device tensor (n_rec, m, T) -> device tensor (n_rec, n_windows, m, m, F).  No host copies."""
    eng = engine or default_engine()
    n_rec, m, T = x.shape
    positions, w = window_positions(T, n_windows, window_size)
    item_rec, item_start = window_items(n_rec, positions, eng.device)
    ff = eng.sliding_ffdtf(x, item_rec, item_start, w, p, freqs, fs, out=out, check=check,
                           grid=regular_grid(positions, w, p) if share_overlap else None)
    return ff.view(n_rec, len(positions), m, m, -1)


def sliding_ffdtf(x, window_size, n_windows, p, freqs, fs, engine: Engine | None = None):
    """NumPy in / NumPy out.  x: (m, T) or (n_rec, m, T) -> (n_windows, m, m, F) or (n_rec, n_windows, ...)."""
    eng = engine or default_engine()
    x = np.asarray(x, dtype=np.float64)
    single = x.ndim == 2
    xd = eng.to_device(x[None] if single else x)
    ff = sliding_ffdtf_device(xd, window_size, n_windows, p, freqs, fs, eng).cpu().numpy()
    return ff[0] if single else ff
