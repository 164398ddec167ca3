"""Side measurement for BASELINE.json config 4 (cross-modal path): many 4-channel windows (160 samples @ 8 Hz,
p = 5, 30 frequencies) as one GPU batch -- ffDTF only (fused C call) and ffDTF + spectra (staged calls)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hyperscanning_signal_analysis_amd.engine import Engine

n_items = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
eng = Engine()
rng = np.random.default_rng(0)
x = rng.standard_normal((n_items, 4, 160))
x[:, :, 1:] += 0.6 * x[:, :, :-1]
xd = eng.to_device(x)
rec = torch.arange(n_items, dtype=torch.int64, device=eng.device)
st = torch.zeros(n_items, dtype=torch.int64, device=eng.device)
freqs = np.arange(1.0, (8.0 / 2 - 0.1) + 0.1, 0.1)
for name, fn in (("ffDTF (fused)", lambda: eng.sliding_ffdtf(xd, rec, st, 160, 5, freqs, 8.0, check=False)),):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {n_items} windows of 4x160, p=5, F=30: {dt*1e3:.2f} ms -> {n_items/dt:,.0f} windows/s")

def both():
    R = eng.lagcov(xd, rec, st, 160, 5)
    ar, V, _, info = eng.yw_solve(R, 4)
    t = eng.transfer(ar, 4, eng.twiddles(freqs, 8.0, 5), want_P=True, want_H=True)
    ff = eng.normalise(t["P"], t["rowsum"], 4)[0]
    sp = eng.to_mmf_complex(eng.spectra(t["H"], V, 4), 4)
    return ff, sp
both(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): both()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"ffDTF + spectra (staged): {dt*1e3:.2f} ms -> {n_items/dt:,.0f} windows/s")
from oracle import mvar_oracle as O
t0 = time.perf_counter()
for k in range(200):
    O.full_freq_dtf(x[k], freqs, 8.0, 5); O.multivariate_spectra(x[k], freqs, 8.0, 5)
print(f"CPU oracle (vectorised NumPy, ffDTF + spectra): {200/(time.perf_counter()-t0):,.0f} windows/s")

# ---- the pipeline mirror itself: dyad x film items per second, batched over items against one item at a time
import tempfile
from pathlib import Path
from hyperscanning_signal_analysis_amd.eeg_alpha_ibi_ffdtf import EEG_IBI_FFDTF_Pipeline
from tests.test_pipeline_cpu import make_tree
from tests.test_gpu_pipeline import synthetic_loader
with tempfile.TemporaryDirectory() as td:
    td = Path(td)
    dy = tuple(f"W_{k:03d}" for k in range(1, 101))
    root = make_tree(td / "data", dyads=dy, films=("Peppa", "Brave"))
    kw = dict(n_windows=3, ar_p=5, plot_global_enabled=False, save_global_enabled=False, plot_windowed_enabled=False,
              save_windowed_enabled=False, loader=synthetic_loader)
    import contextlib, io
    for bi in (1, 256):
        pipe = EEG_IBI_FFDTF_Pipeline(root, td / f"o{bi}", ["Peppa", "Brave"], batch_items=bi, **kw)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            items = [pipe._prepare_item(d, f) for d in pipe.dyads_to_process for f in pipe.target_events]
        t1 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            for k in range(0, len(items), bi):
                pipe._finish_items(items[k:k + bi])
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"run_pipeline, batch_items={bi}: {len(items)} dyad x film items: host DSP {t1 - t0:.2f} s, GPU MVAR + .npz "
              f"{t2 - t1:.2f} s -> {len(items) / (t2 - t1):,.0f} items/s in the MVAR + output phase")
