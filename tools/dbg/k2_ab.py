#!/usr/bin/env python3
"""K2 alone at the north-star batch (599 windows x 64 channels, p = 8): the three forms of the Yule-Walker solve timed
with HIP events on one box, and their distance to each other and to the oracle.
    python tools/dbg/k2_ab.py [n_windows]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hyperscanning_signal_analysis_amd import _lib                                     # noqa: E402
from hyperscanning_signal_analysis_amd.engine import Engine                            # noqa: E402
from hyperscanning_signal_analysis_amd.sliding import window_items, window_positions   # noqa: E402
from hyperscanning_signal_analysis_amd.synthetic import synthetic_var_dyad             # noqa: E402

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 599
m, p, n = 64, 8, 1000
T = n * (nw + 1) // 2
eng = Engine()
x = synthetic_var_dyad(0, m=m, p=p, T=T, fs=500.0)
xd = eng.to_device(x[None])
pos, w = window_positions(T, nw, n)
rec, st = window_items(1, pos, eng.device)
R = eng.lagcov(xd, rec, st, w, p)
mp = eng.pad(m)
wsd = int(eng.lib.hmv_yw_workspace_doubles(m, p))
ws = torch.zeros(nw * wsd, dtype=torch.float64, device=eng.device)
res = {"windows": nw}
out = {}
for name, form, flags in (("lwr_pipelined", 3, 0), ("lwr_first_form", 0, 0), ("ldlt_tile_chain", 1, 0)):
    eng.lib.hmv_set_tuning(_lib.TUNE_YW_FORM, form)
    ar = eng.empty(nw, mp, mp, p); V = eng.empty(nw, mp, mp); info = eng.empty(nw, dtype=torch.int32)
    ld = eng.empty(nw, p)

    def run(logdet=False):
        rc = eng.lib.hmv_yw_solve_f64(R.data_ptr(), nw, m, p, ws.data_ptr(), ar.data_ptr(), V.data_ptr(),
                                      ld.data_ptr() if logdet else 0, info.data_ptr(), flags, eng.stream())
        assert rc == 0
    run(); run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    res[name + "_ms"] = e0.elapsed_time(e1) / 10
    run(True)
    torch.cuda.synchronize()
    out[name] = (ar.clone(), V.clone(), ld.clone())
    res[name + "_bad_windows"] = int((info != 0).sum())
eng.lib.hmv_set_tuning(_lib.TUNE_YW_FORM, 0)
if os.environ.get("K2_STAMPS"):          # a library built with -DHMV_LWR2_STAMP: cycles of wave 0 per section
    ws.zero_()
    ar = eng.empty(nw, mp, mp, p); V = eng.empty(nw, mp, mp); info = eng.empty(nw, dtype=torch.int32)
    eng.lib.hmv_set_tuning(_lib.TUNE_YW_FORM, 3)
    eng.lib.hmv_yw_solve_f64(R.data_ptr(), nw, m, p, ws.data_ptr(), ar.data_ptr(), V.data_ptr(), 0, info.data_ptr(), 0, eng.stream())
    torch.cuda.synchronize()
    eng.lib.hmv_set_tuning(_lib.TUNE_YW_FORM, 0)
    tiles = wsd // (mp * mp)
    st8 = ws.view(nw, tiles, mp * mp)[:, tiles - 1, :8].contiguous().view(torch.int64).cpu().numpy().astype(np.float64)
    names = ["setup+barriers", "inverses", "A_new,Vf", "forward side", "D pass", "B_new,Vb,backward", "logdet,V,emit", "(inside the products: k-step loops)"]
    res["stamps_mean_kcycles"] = {n: round(float(st8[:, k].mean()) / 1e3, 1) for k, n in enumerate(names)}
    res["stamps_total_kcycles"] = round(float(st8[:, :7].sum(axis=1).mean()) / 1e3, 1)
ref = out["ldlt_tile_chain"]
for name in ("lwr_pipelined", "lwr_first_form"):
    for k, what in enumerate(("ar", "V", "logdet")):
        res[f"{name}_vs_ldlt_{what}"] = float((out[name][k] - ref[k]).abs().max() / ref[k].abs().max())
from oracle import mvar_oracle as O                                                     # noqa: E402
for k in (0, nw // 2):
    aro, Vo = O.ar_coeff(x[:, pos[k]:pos[k] + w], p)
    for name in out:
        res[f"{name}_vs_oracle_ar_w{k}"] = float(np.abs(out[name][0][k, :m, :m].cpu().numpy() - aro).max() / np.abs(aro).max())
print(json.dumps(res, indent=1))
