"""Side measurement for the multitaper PSD (BASELINE.json config 5 shape: 64 channels, 220 s @ 500 Hz,
fmin 1, fmax 30, bandwidth 2 Hz -> 439 tapers) on the GPU, and the NumPy restatement on a few channels."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hyperscanning_signal_analysis_amd.psd import compute_psd_multitaper, _tapers

n_ch, n, fs = 64, 110_000, 500.0
x = np.random.default_rng(0).standard_normal((n_ch, n))
t0 = time.perf_counter(); tp, w = _tapers(n, 2.0 * n / (2 * fs)); t_tap = time.perf_counter() - t0
print(f"DPSS tapers on the host (SciPy, cached afterwards): {tp.shape[0]} x {n} in {t_tap:.1f} s")
compute_psd_multitaper(x, fs, 1.0, 30.0, 2.0); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): f, p = compute_psd_multitaper(x, fs, 1.0, 30.0, 2.0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"GPU: {n_ch} channels x {n} samples, {tp.shape[0]} tapers, {len(f)} bins: {dt*1e3:.0f} ms per recording (incl. H2D of tapers)")
from oracle import psd_oracle as P
t0 = time.perf_counter(); P.compute_psd_multitaper(x[:2], fs, 1.0, 30.0, 2.0); tc = (time.perf_counter() - t0 - t_tap) / 2
print(f"host NumPy restatement: {tc:.1f} s per channel (taper time subtracted) -> {tc*n_ch:.0f} s per recording")
