"""Multi-rank layout on CPU (gloo, world_size 2): sharding is balanced and disjoint, the single gather to
rank 0 returns the per-rank results in rank order, band integration matches a NumPy reduction."""
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


def test_band_integrate_matches_numpy():
    rng = np.random.default_rng(0)
    ff = rng.random((3, 4, 4, 256))
    freqs = 0.5 * np.arange(1, 257)
    got = hd.band_integrate(torch.as_tensor(ff), freqs).numpy()
    assert got.shape == (3, 4, 4, len(hd.DEFAULT_BANDS))
    for b, (lo, hi) in enumerate(hd.DEFAULT_BANDS):
        sel = (freqs >= lo) & (freqs < hi)
        assert np.allclose(got[..., b], ff[..., sel].sum(-1), rtol=1e-13)
    assert np.allclose(got.sum(-1), ff.sum(-1), rtol=1e-12)      # the bands tile 0.5 .. 128 Hz


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
