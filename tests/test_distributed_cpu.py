"""Multi-rank layout on CPU (gloo, world_size 2): sharding is balanced and disjoint, the single gather to
rank 0 returns the per-rank results in rank order, band bins and window-range shards are right."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hyperscanning_signal_analysis_amd import distributed as hd


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 64, 599):
        for world in (1, 2, 3, 8):
            spans = [hd.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert hd.shard_dyads(64, 8, 3) == list(range(24, 32))
    pos = np.arange(0, 5000, 500)
    assert list(hd.shard_windows(pos, 2, 1)) == list(pos[5:])


def test_band_bins_tile_the_grid():
    freqs = 0.5 * np.arange(1, 257)
    lo, hi = hd.band_bins(freqs)
    assert lo.dtype == np.int32 and len(lo) == len(hd.DEFAULT_BANDS)
    for b, (f_lo, f_hi) in enumerate(hd.DEFAULT_BANDS):
        sel = np.flatnonzero((freqs >= f_lo) & (freqs < f_hi))
        assert (lo[b], hi[b]) == (sel[0], sel[-1] + 1)
    assert lo[0] == 0 and hi[-1] == 256 and np.array_equal(lo[1:], hi[:-1])      # the bands tile 0.5 .. 128 Hz
    lo2, hi2 = hd.band_bins(freqs, bands=((200.0, 300.0),))                        # empty band
    assert lo2[0] == hi2[0]


def test_shard_window_items_rebases_starts():
    pos = np.arange(0, 300_000 - 999, 500)                    # 599 windows of 1000, hop 500
    covered = []
    for world in (1, 2, 4, 8, 3):
        covered = []
        for r in range(world):
            s_lo, s_hi, starts, (w_lo, w_hi) = hd.shard_window_items(pos, 1000, world, r)
            assert s_lo == pos[w_lo] and s_hi == pos[w_hi - 1] + 1000
            assert starts[0] == 0 and np.array_equal(starts + s_lo, pos[w_lo:w_hi])
            assert s_hi - s_lo == 500 * (w_hi - w_lo) + 500           # own windows + the one-hop halo
            covered += list(range(w_lo, w_hi))
        assert covered == list(range(599))
    s_lo, s_hi, starts, span = hd.shard_window_items(pos[:2], 1000, 4, 3)          # more ranks than windows
    assert (s_lo, s_hi, len(starts)) == (0, 0, 0) and span[0] == span[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dyads = hd.shard_dyads(5, world, rank)                      # rank 0: [0,1,2], rank 1: [3,4]
        # every rank "computes" a result that encodes its dyads; shapes must match for the gather
        local = torch.full((3, 2, 2, 5), float("nan"), dtype=torch.float64)
        for k, d in enumerate(dyads):
            local[k] = d
        got = hd.gather_to_root(local, dst=0)
        if rank == 0:
            assert got.shape == (world, 3, 2, 2, 5)
            np.save(os.path.join(out_dir, "gathered.npy"), got.numpy())
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


def test_gather_to_root_gloo_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = np.load(tmp_path / "gathered.npy")
    assert np.all(g[0, :3] == np.array([0, 1, 2])[:, None, None, None])
    assert np.all(g[1, :2] == np.array([3, 4])[:, None, None, None]) and np.isnan(g[1, 2]).all()


def test_gather_without_process_group_is_identity():
    t = torch.arange(6.0).view(2, 3)
    assert torch.equal(hd.gather_to_root(t), t.unsqueeze(0))
