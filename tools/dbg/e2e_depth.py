#!/usr/bin/env python3
"""Engine.stream_dyads at several pipeline depths against the resident rate (one box, one process):
python tools/dbg/e2e_depth.py [dyads]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hyperscanning_signal_analysis_amd import distributed as hdist                              # noqa: E402
from hyperscanning_signal_analysis_amd.engine import Engine                                     # noqa: E402
from hyperscanning_signal_analysis_amd.sliding import regular_grid, window_items, window_positions   # noqa: E402
from hyperscanning_signal_analysis_amd.synthetic import NORTHSTAR, northstar_freqs, synthetic_var_dyad   # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ns = NORTHSTAR
m, fs, w, p, F = ns["m"], ns["fs"], ns["window"], ns["p"], ns["F"]
T = 300000
pos, w = window_positions(T, 2 * T // w - 1, w)
freqs = northstar_freqs(F)
eng = Engine(max_workspace_bytes=64 << 30)
if os.environ.get("E2E_COPY_PRIO") is not None:      # A/B: priority of the two copy streams (-1 high, 0 normal)
    pr = int(os.environ["E2E_COPY_PRIO"])
    eng._copy = (torch.cuda.Stream(eng.device, priority=pr), torch.cuda.Stream(eng.device, priority=pr))
fdev = eng.to_device(freqs)
xh = synthetic_var_dyad(0, m=m, p=p, T=T, fs=fs)
x = eng.to_device(xh[None])
rec, st = window_items(1, pos, eng.device)
grid = regular_grid(pos, w, p)
lo, hi = hdist.band_bins(freqs)
out = eng.empty(len(pos), m, m, F)
res = {}
for name, fn in (("resident_full", lambda: eng.sliding_ffdtf(x, rec, st, w, p, fdev, fs, out=out, check=False, grid=grid)),
                 ("resident_bands", lambda: eng.sliding_ffdtf(x, rec, st, w, p, fdev, fs, check=False, grid=grid, bands=(lo, hi)))):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    res[name + "_ms"] = (time.perf_counter() - t0) / 8 * 1e3
src = torch.from_numpy(xh).pin_memory()
host_out = torch.empty(E, len(pos), m, m, len(lo), dtype=torch.float64).pin_memory()
for depth in (2, 3):
    eng.stream_dyads([src] * depth, w, pos, p, fdev, fs, out=host_out, depth=depth)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.stream_dyads([src] * E, w, pos, p, fdev, fs, out=host_out, depth=depth)
    torch.cuda.synchronize()
    res["stream_depth%d_ms_per_dyad" % depth] = (time.perf_counter() - t0) / E * 1e3
tl = []
eng.stream_dyads([src] * 8, w, pos, p, fdev, fs, out=host_out, depth=3, timeline=tl)
res["device_timeline_depth3"] = [{"d": e[1], **{k: round(v, 2) for k, v in e[2].items()}} for e in tl if e[0] == "device_ms"]
# raw PCIe rates of this box
d = eng.empty(m, T)
torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(src, non_blocking=True); torch.cuda.synchronize()
res["h2d_GBps"] = src.numel() * 8 / (time.perf_counter() - t0) / 1e9
r = eng.empty(len(pos), m, m, len(lo))
torch.cuda.synchronize(); t0 = time.perf_counter(); host_out[0].copy_(r, non_blocking=True); torch.cuda.synchronize()
res["d2h_GBps"] = r.numel() * 8 / (time.perf_counter() - t0) / 1e9
print(json.dumps(res))
