"""Side measurement for the SURVEY 8(f) rank-4 measures: dDTF (= ffDTF x |partial coherence|) and GPDC per window
at the north-star shape (m = 64, n = 1000, p = 8, F = 256) on the GPU, and at m = 8 against the reference's
minors-by-determinant algorithm (oracle restatement) on the host."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hyperscanning_signal_analysis_amd.engine import Engine
from hyperscanning_signal_analysis_amd.synthetic import synthetic_var_dyad, northstar_freqs
from hyperscanning_signal_analysis_amd.sliding import window_items, window_positions

eng = Engine()


def measures(xd, rec, st, n, p, freqs, fs, m):
    R = eng.lagcov(xd, rec, st, n, p)
    ar, V, _, _ = eng.yw_solve(R, m)
    t = eng.transfer(ar, m, eng.twiddles(freqs, fs, p), want_P=True, want_H=True, want_A=True)
    ff = eng.normalise(t["P"], t["rowsum"], m)[0]
    S = eng.spectra(t["H"], V, m)
    kappa, _ = eng.partial_coherence(S, m)
    return ff * kappa.abs(), eng.gpdc(t["A"], V, m)


W = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = synthetic_var_dyad(0, T=500 * (W + 1))
xd = eng.to_device(x[None])
pos, w = window_positions(x.shape[1], W, 1000)
rec, st = window_items(1, pos, eng.device)
freqs = northstar_freqs(256)
measures(xd, rec, st, w, 8, freqs, 500.0, 64); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): d, g = measures(xd, rec, st, w, 8, freqs, 500.0, 64)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"GPU  m=64 p=8 F=256: dDTF + GPDC of {W} windows in {dt*1e3:.1f} ms -> {W/dt:,.0f} windows/s")

from oracle import mvar_oracle as O
m8 = x[:8, :1000]
f8 = freqs[::16]
t0 = time.perf_counter()
O.direct_dtf(m8, f8, 500.0, 8); O.gen_partial_directed_coherence(m8, f8, 500.0, 8)
tc = time.perf_counter() - t0
x8 = eng.to_device(np.ascontiguousarray(x[:8])[None])
measures(x8, rec, st, w, 8, f8, 500.0, 8); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): measures(x8, rec, st, w, 8, f8, 500.0, 8)
torch.cuda.synchronize()
dg = (time.perf_counter() - t0) / 3
print(f"m=8 p=8 F=16: host (minors by determinant, oracle) {1/tc:,.1f} windows/s; GPU batch of {W}: {W/dg:,.0f} windows/s")
