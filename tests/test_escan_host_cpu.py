"""Host-side logic of the batch front-end and of the window grids (no GPU): file discovery in the reference's export
tree, decoding of `task_events_structure`, segment cutting / preprocessing (`escan_batch`), and when a window grid counts
as regular (`sliding.regular_grid`)."""
import numpy as np
import pytest

from hyperscanning_signal_analysis_amd import escan_batch as EB
from hyperscanning_signal_analysis_amd.sliding import hop_positions, regular_grid, window_positions
from tests.test_gpu_escan_batch import CHANS, FS, _reader, _write


@pytest.fixture()
def tree(tmp_path):
    root = tmp_path / "UNIWAW_imported"
    for dy in ("W_003", "W_010"):
        for r, (code, role) in enumerate((("ch", "child"), ("cg", "caregiver"))):
            _write(root / "EEG" / dy / role / f"{dy}_EEG_{code}_passive_movies.nc", r, 30.0,
                   [("Peppa", 1.0, 11.0), ("Brave", 14.0, 9.5)], r == 0)
            _write(root / "EEG" / dy / role / f"{dy}_EEG_{code}_talk.nc", 10 + r, 16.0, [("talk_1", 0.5, 12.0)], r == 1)
    _write(root / "EEG" / "W_020" / "child" / "W_020_EEG_ch_talk.nc", 7, 16.0, [("talk_1", 0.5, 12.0)], True)
    (root / "EEG" / "W_020" / "child" / "notes.txt").write_text("not an export")
    return root


def test_discovery_and_event_decoding(tree):
    found = EB.discover_dyads(tree)
    assert sorted(found) == ["W_003", "W_010", "W_020"]
    assert sorted(found["W_003"]) == ["passive_movies", "talk"] and sorted(found["W_003"]["talk"]) == ["cg", "ch"]
    assert list(found["W_020"]["talk"]) == ["ch"]
    assert list(EB.discover_dyads(tree, tasks=("talk",))["W_010"]) == ["talk"]
    rec = _reader(found["W_003"]["passive_movies"]["ch"])                 # attribute as a JSON string
    assert EB.decode_events(rec["attrs"]) == [("Peppa", 1.0, 11.0), ("Brave", 14.0, 9.5)]
    rec = _reader(found["W_003"]["passive_movies"]["cg"])                 # attribute as a decoded list
    assert EB.decode_events(rec["attrs"]) == [("Peppa", 1.0, 11.0), ("Brave", 14.0, 9.5)]
    assert EB.decode_events({"task_events_structure": ""}) == [] and EB.decode_events({}) == []
    assert EB.decode_events({"task_events_structure": [{"start_s": 3.0, "duration_s": 2.0}, "junk"]}) == [("event_1", 3.0, 2.0)]
    with pytest.raises(FileNotFoundError):
        EB.discover_dyads(tree / "nowhere")


def test_segment_block_cut_zscore_and_stack(tree):
    found = EB.discover_dyads(tree)
    recs = {r: _reader(found["W_003"]["talk"][r]) for r in ("ch", "cg")}
    block, names, fs = EB.segment_block(recs["ch"], recs["cg"], 0.5, 12.0, 1.0, 45.0)
    keep = [c for c in CHANS if c not in ("M1", "M2")]
    assert names == [f"{c}_ch" for c in keep] + [f"{c}_cg" for c in keep] and fs == FS
    t = recs["ch"]["time"]
    assert block.shape == (16, int(np.sum((t >= 0.5) & (t <= 12.5))))                     # inclusive cut
    assert np.abs(block.mean(axis=1)).max() < 1e-12 and np.abs(block.std(axis=1) - 1).max() < 1e-12
    sub, names2, _ = EB.segment_block(recs["ch"], recs["cg"], 0.5, 12.0, channel_subset=["F3", "F4", "XX"])
    assert names2 == ["F3_ch", "F4_ch", "F3_cg", "F4_cg"] and sub.shape[0] == 4
    with pytest.raises(ValueError, match="None of the requested channels"):
        EB.segment_block(recs["ch"], recs["cg"], 0.5, 12.0, channel_subset=["XX"])
    with pytest.raises(ValueError, match="Invalid high_cutoff_hz"):
        EB.segment_block(recs["ch"], recs["cg"], 0.5, 12.0, high_cutoff_hz=100.0)


def test_regular_grid_detection():
    pos, w = window_positions(300_000, 599, 1000)
    assert regular_grid(pos, w, 8) == (500, 0, 599)
    assert regular_grid(pos + 250, w, 8) == (500, 250, 599)
    assert regular_grid(np.arange(0, 4000, 250), 1000, 8) == (250, 0, 16)                # 75 % overlap: 4 hops
    assert regular_grid(np.arange(0, 4000, 1000), 1000, 8) is None                        # no overlap: nothing to share
    assert regular_grid(np.arange(0, 4000, 300), 1000, 8) is None                         # hop does not divide the window
    assert regular_grid(np.arange(0, 4000, 100), 1000, 8) is None                         # more than 8 hops per window
    assert regular_grid(np.arange(0, 400, 5), 40, 8) is None                              # hop not longer than the order
    assert regular_grid([0, 500, 1001], 1000, 8) is None and regular_grid([0], 1000, 8) is None
    pos2, w2 = window_positions(480, 5, 160)                                               # linspace(.., dtype=int) grid of G4
    assert regular_grid(pos2, w2, 5) == (80, 0, 5)


@pytest.mark.parametrize("T,W,hop,want", [(1200, 1000, 500, [0]), (1400, 1000, 500, [0]), (1500, 1000, 500, [0, 500]),
                                          (2300, 1000, 1000, [0, 1000]), (5001, 1000, 500, list(range(0, 4001, 500))),
                                          (999, 1000, 500, []), (1000, 1000, 500, [0])])
def test_fixed_hop_grid_never_refuses_a_segment_length(T, W, hop, want):
    """Segments are cut with an inclusive time mask (dur * fs + 1 samples) and are almost never a whole number of hops:
    the fixed-hop grid drops the tail and stays regular (so K1 can share the overlap), where the reference's
    `_create_windows` arithmetic (`window_positions`) would refuse W < T < 1.5 W or any T that is not a multiple of W
    at zero overlap -- the failure that used to discard a whole dyad."""
    pos = hop_positions(T, W, hop)
    assert pos.tolist() == want
    assert all(s + W <= T for s in pos)
    if len(pos) >= 2 and W % hop == 0 and 2 <= W // hop <= 8:
        assert regular_grid(pos, W, 8) == (hop, 0, len(pos))
    if T in (1200, 1400, 2300):
        with pytest.raises(ValueError):
            window_positions(T, (T - W) // hop + 1, W)
