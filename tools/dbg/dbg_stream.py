"""Where the time of Engine.stream_dyads goes: copies alone, compute alone, pipelined."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hyperscanning_signal_analysis_amd.engine import Engine
from hyperscanning_signal_analysis_amd import distributed as hdist
from hyperscanning_signal_analysis_amd.sliding import window_positions, window_items, regular_grid
from hyperscanning_signal_analysis_amd.synthetic import synthetic_var_dyad, northstar_freqs
eng = Engine()
m, T, w, p = 64, 300000, 1000, 8
x = synthetic_var_dyad(0, m=m, p=p, T=T)
pos, w = window_positions(T, 599, w)
freqs = northstar_freqs(256)
xp = torch.from_numpy(x).pin_memory()
xd = eng.empty(1, m, T)
def tm(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("H2D 154 MB pinned: %.2f ms" % tm(lambda: xd[0].copy_(xp, non_blocking=True)))
rec, st = window_items(1, pos, eng.device)
fd = eng.to_device(freqs)
ff = eng.empty(599, m, m, 256)
g = regular_grid(pos, w, p)
print("compute (resident): %.2f ms" % tm(lambda: eng.sliding_ffdtf(xd, rec, st, w, p, fd, 500.0, out=ff, check=False, grid=g)))
print("compute + nan-async: %.2f ms" % tm(lambda: eng.sliding_ffdtf(xd, rec, st, w, p, fd, 500.0, out=ff, check="nan-async", grid=g)))
lo, hi = hdist.band_bins(freqs)
print("band sums: %.2f ms" % tm(lambda: eng.band_sums(ff, lo, hi)))
red = eng.band_sums(ff, lo, hi)
hp = torch.empty(red.shape, dtype=red.dtype).pin_memory()
print("D2H 98 MB pinned: %.2f ms" % tm(lambda: hp.copy_(red, non_blocking=True)))
feed = [xp] * 8
out = torch.empty(8, *red.shape, dtype=torch.float64).pin_memory()
eng.stream_dyads(feed[:2], w, pos, p, fd, 500.0, out=out); torch.cuda.synchronize()
for depth in (2, 3):
    tl = []
    t0 = time.perf_counter(); eng.stream_dyads(feed, w, pos, p, fd, 500.0, out=out, timeline=tl, depth=depth); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("stream_dyads depth %d: %.2f ms per dyad" % (depth, dt / 8 * 1e3), [(e[0][0], e[1], round((e[2] - t0) * 1e3, 1)) for e in tl])
