import sys, numpy as np, torch
sys.path.insert(0, ".")
from hyperscanning_signal_analysis_amd import _lib
from hyperscanning_signal_analysis_amd.engine import Engine
from hyperscanning_signal_analysis_amd.sliding import window_items, window_positions
from hyperscanning_signal_analysis_amd.synthetic import synthetic_var_dyad, northstar_freqs
eng = Engine(max_workspace_bytes=64 << 30)
x = synthetic_var_dyad(0, T=60_000)
xd = eng.to_device(x[None])
freqs = eng.to_device(northstar_freqs(256))
for nw in (9, 10, 12, 16, 24, 40, 72, 119):
    T = 500 * (nw + 1)
    pos, w = window_positions(T, nw, 1000)
    rec, st = window_items(1, pos, eng.device)
    out = eng.empty(nw, 64, 64, 256)
    res = {}
    for tag, fl in (("fused", 0), ("unfused", _lib.FLAG_UNFUSED_NORM)):
        ts = []
        for rep in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); b.record(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.sliding_ffdtf(xd[:, :, :T], rec, st, w, 8, freqs, 500.0, out=out, check=False, flags=fl,
                              k3_events=(a.cuda_event, b.cuda_event))
            e1.record(); torch.cuda.synchronize()
            ts.append((a.elapsed_time(b), e0.elapsed_time(e1)))
        res[tag] = min(t[0] for t in ts[1:]), min(t[1] for t in ts[1:])
    print(f"nw={nw:4d} fused windows={max(0, nw-8):4d}  K3 fused {res['fused'][0]:.3f} ms  unfused {res['unfused'][0]:.3f} ms   "
          f"call fused {res['fused'][1]:.3f}  unfused {res['unfused'][1]:.3f}")
