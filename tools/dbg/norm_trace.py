"""Diagnostic (libhypermvar_trace.so, -DHMV_NORM_TRACE): when / where each window was normalised inside K3."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["HYPERMVAR_LIB"] = os.path.join(ROOT, "hyperscanning_signal_analysis_amd", "libhypermvar_trace.so")
from hyperscanning_signal_analysis_amd.engine import Engine
from hyperscanning_signal_analysis_amd.sliding import window_items, window_positions
from hyperscanning_signal_analysis_amd.synthetic import synthetic_var_dyad, northstar_freqs
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 119
eng = Engine(max_workspace_bytes=64 << 30)
eng.lib.hmv_debug_set_tf_stamps.argtypes = [ctypes.c_void_p]
T = 500 * (nw + 1)
x = synthetic_var_dyad(0, T=T)
xd = eng.to_device(x[None])
freqs = eng.to_device(northstar_freqs(256))
pos, w = window_positions(T, nw, 1000)
rec, st = window_items(1, pos, eng.device)
out = eng.empty(nw, 64, 64, 256)
stamps = torch.zeros(nw * 4, 4, dtype=torch.int64, device=eng.device)
eng.lib.hmv_debug_set_tf_stamps(stamps.data_ptr())
for rep in range(3):
    stamps.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); b.record(); torch.cuda.synchronize()
    eng.sliding_ffdtf(xd, rec, st, w, 8, freqs, 500.0, out=out, check=False, k3_events=(a.cuda_event, b.cuda_event))
    torch.cuda.synchronize()
print("K3 ms", a.elapsed_time(b))
s = stamps.cpu().numpy().reshape(nw, 4, 4)[: nw - 8]
t0 = s[:, :, 0].min()
start = (s[:, :, 0].min(axis=1) - t0) / 100.0      # us
den = (s[:, :, 1].max(axis=1) - s[:, :, 0].min(axis=1)) / 100.0
dur = (s[:, :, 2].max(axis=1) - s[:, :, 0].min(axis=1)) / 100.0
hw = s[:, 0, 3]
xcc = hw & 0xf
hwid = (hw >> 32) & 0xffffffff
cu = (hwid >> 8) & 0xf
se = (hwid >> 13) & 0x7
print("window  start_us  den_us  total_us  xcc se cu")
for k in range(0, len(s), max(1, len(s) // 40)):
    print(f"{k:5d} {start[k]:9.1f} {den[k]:7.1f} {dur[k]:9.1f}   {xcc[k]} {se[k]} {cu[k]}")
print("duration us: median %.1f  mean %.1f  max %.1f" % (np.median(dur), dur.mean(), dur.max()))
print("xcc histogram", np.bincount(xcc.astype(int), minlength=8))
print("last end (us)", ((s[:, :, 2].max(axis=1) - t0) / 100.0).max())
conc = [(((s[:, :, 0].min(axis=1) <= tt) & (s[:, :, 2].max(axis=1) > tt)).sum()) for tt in s[:, :, 0].min(axis=1)]
print("concurrent normalisers when each starts: max", max(conc), "mean", np.mean(conc))
