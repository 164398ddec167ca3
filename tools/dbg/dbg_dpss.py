"""Debug aid: GPU DPSS against SciPy, per taper."""
import sys, time
import numpy as np, torch
from scipy.signal.windows import dpss
from hyperscanning_signal_analysis_amd.psd import dpss_device
for M, NW, K in [(64, 2.0, 4), (1000, 4.0, 8), (999, 2.5, 5), (6000, 12.0, 24), (20000, 40.0, 80)]:
  for sym in (False, True):
    rt, rr = dpss(M, NW, K, sym=sym, norm=2, return_ratios=True)
    t, r = dpss_device(M, NW, K, sym)
    t, r = t.cpu().numpy(), r.cpu().numpy()
    e = np.abs(t - rt).max(axis=1)
    print(M, NW, K, sym, "taper err", e.max(), "argmax", e.argmax(), "ratio err", np.abs(r - rr).max(), "orth", np.abs(t @ t.T - np.eye(K)).max(), flush=True)
    if e.max() > 1e-8:
        print("  per taper", e[:8], "\n  ratios", r[:6], rr[:6])
if len(sys.argv) > 1:
    M, NW = 110_000, 220.0
    K = 440
    torch.cuda.synchronize(); t0 = time.time(); t, r = dpss_device(M, NW, K); torch.cuda.synchronize(); print("M=110000 K=440:", time.time() - t0, "s")
    t0 = time.time(); t, r = dpss_device(M, NW, K); torch.cuda.synchronize(); print("again:", time.time() - t0, "s", "kept", int((r > 0.9).sum()))
    tt = t[:8].cpu().numpy(); print("norms", (tt ** 2).sum(axis=1))
    t0 = time.time(); rt, rr = dpss(M, NW, K, sym=False, norm=2, return_ratios=True); print("scipy:", time.time() - t0, "s")
    print("M=110000 taper err", np.abs(t.cpu().numpy() - rt).max(), "ratio err", np.abs(r.cpu().numpy() - rr).max())
