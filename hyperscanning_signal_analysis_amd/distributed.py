"""Multi-GPU layout of the sliding-window path: one process per GPU, dyads (or window ranges) sharded
across ranks with NO data-path collective, and one gather to rank 0 at the end of a job.

Every (dyad, window) is independent in the reference (it refits each window from scratch,
/root/reference/src/eeg_alpha_ibi_ffdtf.py:741-755), so the only exchange is the final collection of
results.  Full-resolution ffDTF is 8.4 MB per window (5 GB per 10-minute dyad): it stays rank-local
(each rank can write its own per-dyad files, as the reference saves one .npz per dyad x film,
eeg_alpha_ibi_ffdtf.py:637-658); what is gathered over RCCL/xGMI is the band-integrated product
(m, m, n_bands) per window -- the quantity the reference's graph plots integrate anyway
(/root/reference/src/mtmvar.py:984-987).

Works with backend "nccl" (= RCCL on ROCm) on GPUs and with "gloo" on CPU tensors (tests).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

__all__ = ["shard_range", "shard_dyads", "shard_windows", "shard_window_items", "band_bins", "band_integrate",
           "gather_to_root", "DEFAULT_BANDS"]

# delta, theta, alpha, beta, gamma edges in Hz (inclusive low, exclusive high)
DEFAULT_BANDS = ((0.5, 4.0), (4.0, 8.0), (8.0, 13.0), (13.0, 30.0), (30.0, 128.5))


def shard_range(n: int, world: int, rank: int):
    """Contiguous, balanced [lo, hi) share of n units for `rank` (first n % world ranks get one more)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_dyads(n_dyads: int, world: int, rank: int):
    lo, hi = shard_range(n_dyads, world, rank)
    return list(range(lo, hi))


def shard_windows(positions, world: int, rank: int):
    """Window-range sharding of ONE recording (read-only halo: windows overlap, nothing is exchanged)."""
    lo, hi = shard_range(len(positions), world, rank)
    return np.asarray(positions)[lo:hi]


def shard_window_items(positions, window_size: int, world: int, rank: int):
    """Window-range shard of ONE recording for `rank`: (sample_lo, sample_hi, rebased_starts, (win_lo, win_hi)).

    The rank needs only samples [sample_lo, sample_hi) of the recording -- its own windows plus the read-only
    overlap with the neighbour's first window -- and finds its windows at `rebased_starts` inside that slice
    (secondary partitioning: BASELINE config 2 at 2 / 4 / 8 GPUs).  Windows of a shard computed
    from the slice equal the same windows computed from the whole recording bit for bit: every kernel reads only
    the samples of its own window.  An empty shard (more ranks than windows) gives (0, 0, [], (lo, lo))."""
    pos = np.asarray(positions, dtype=np.int64)
    lo, hi = shard_range(len(pos), world, rank)
    if hi == lo:
        return 0, 0, pos[:0], (lo, lo)
    mine = pos[lo:hi]
    s_lo, s_hi = int(mine.min()), int(mine.max()) + int(window_size)
    return s_lo, s_hi, mine - s_lo, (lo, hi)


def band_bins(freqs, bands=DEFAULT_BANDS):
    """Bin ranges [lo, hi) of each band on the frequency grid (empty band: lo == hi).  Host-side index logic."""
    f = np.asarray(freqs, dtype=np.float64)
    lo, hi = [], []
    for b_lo, b_hi in bands:
        idx = np.flatnonzero((f >= b_lo) & (f < b_hi))
        if len(idx) == 0:
            lo.append(0); hi.append(0)
        else:
            if not np.array_equal(idx, np.arange(idx[0], idx[-1] + 1)):
                raise ValueError("band_bins needs an ascending frequency grid")
            lo.append(int(idx[0])); hi.append(int(idx[-1]) + 1)
    return np.asarray(lo, dtype=np.int32), np.asarray(hi, dtype=np.int32)


def band_integrate(ff: torch.Tensor, freqs, bands=DEFAULT_BANDS, engine=None) -> torch.Tensor:
    """(..., m, m, F) device tensor -> (..., m, m, n_bands): sum of ffDTF over the bins of each band
    (`hmv_band_sums_f64`; the product has no CPU path)."""
    from .engine import default_engine
    eng = engine or default_engine()
    lo, hi = band_bins(freqs, bands)
    return eng.band_sums(ff, lo, hi)


def gather_to_root(local: torch.Tensor, dst: int = 0):
    """One gather of equally-shaped per-rank tensors to `dst`; returns (world, *shape) on dst, None elsewhere."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local.unsqueeze(0)
    world, rank = dist.get_world_size(), dist.get_rank()
    local = local.contiguous()
    if rank == dst:
        buf = [torch.empty_like(local) for _ in range(world)]
        dist.gather(local, gather_list=buf, dst=dst)
        return torch.stack(buf, dim=0)
    dist.gather(local, gather_list=None, dst=dst)
    return None
