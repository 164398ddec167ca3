import sys, numpy as np, torch
sys.path.insert(0, ".")
from hyperscanning_signal_analysis_amd import _lib
from hyperscanning_signal_analysis_amd.engine import default_engine
from hyperscanning_signal_analysis_amd.sliding import window_items, window_positions
from hyperscanning_signal_analysis_amd.synthetic import synthetic_var_dyad
eng = default_engine()
for (m, n, p, F, nw) in [(48, 600, 4, 16, 400), (48, 600, 4, 48, 120), (33, 300, 2, 16, 400), (64, 600, 4, 16, 400), (19, 400, 3, 32, 400), (4, 160, 5, 32, 500), (64, 1000, 8, 256, 59)]:
    T = n * (nw + 1) // 2
    x = synthetic_var_dyad(11, m=m, p=min(p, 4), T=T, burn=300)
    freqs = np.linspace(0.5, 120.0, F)
    xd = eng.to_device(x[None])
    pos, w = window_positions(T, nw, n)
    rec, st = window_items(1, pos, eng.device)
    fused = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0)
    again = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0)
    plain = eng.sliding_ffdtf(xd, rec, st, w, p, freqs, 500.0, flags=_lib.FLAG_UNFUSED_NORM)
    torch.cuda.synchronize()
    d = (fused != plain)
    print((m, n, p, F, nw), "fused==plain", bool(torch.equal(fused, plain)), "fused==again", bool(torch.equal(fused, again)),
          "n diff", int(d.sum()))
    if d.any():
        idx = d.nonzero()
        print(" windows", idx[:, 0].unique()[:20].tolist(), "...", int(idx[:, 0].max()))
        print(" rows", idx[:, 1].unique().tolist())
        print(" cols", idx[:, 2].unique().tolist())
        print(" f", idx[:, 3].unique().tolist())
        k = idx[0]
        print(" first", k.tolist(), float(fused[tuple(k)]), float(plain[tuple(k)]))
        rel = ((fused - plain).abs() / plain.abs().clamp_min(1e-300)).max()
        print(" max rel", float(rel))
