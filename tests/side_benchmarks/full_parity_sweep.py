"""Full-size parity sweep (BASELINE config 2): EVERY one of the 599 windows of a 10-minute 64-channel dyad, GPU path
(default flags: K1 on shared hop blocks, K2 tile chain, K3 with the normalisation inside) against the oracle's
vectorised restatement on the host, window by window.  ~2 minutes of host time with 16 worker processes; prints the
worst window.  Run on the GPU box: python tests/side_benchmarks/full_parity_sweep.py [dyad]"""
import multiprocessing as mp
import os
import sys
import time

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mvar_oracle as O
from hyperscanning_signal_analysis_amd.synthetic import NORTHSTAR, northstar_freqs, synthetic_var_dyad

_X = None
_FREQS = northstar_freqs(256)


def _ref(s):
    return O.full_freq_dtf(_X[:, s:s + 1000], _FREQS, 500.0, 8)


def main():
    global _X
    dyad = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    _X = synthetic_var_dyad(dyad)
    T = _X.shape[1]
    pool = mp.get_context("fork").Pool(16)            # forked BEFORE anything touches the GPU
    import torch
    from hyperscanning_signal_analysis_amd.engine import Engine
    from hyperscanning_signal_analysis_amd.sliding import regular_grid, window_items, window_positions
    pos, w = window_positions(T, 2 * T // 1000 - 1, 1000)
    eng = Engine(max_workspace_bytes=64 << 30)
    xd = eng.to_device(_X[None])
    rec, st = window_items(1, pos, eng.device)
    ff = eng.sliding_ffdtf(xd, rec, st, w, 8, _FREQS, 500.0, grid=regular_grid(pos, w, 8))
    torch.cuda.synchronize()
    t0 = time.time()
    worst_max, worst_elem, worst_k = 0.0, 0.0, -1
    row_err = 0.0
    for k0 in range(0, len(pos), 64):
        ks = list(range(k0, min(len(pos), k0 + 64)))
        refs = pool.map(_ref, [int(pos[k]) for k in ks])
        got = ff[k0:k0 + len(ks)].cpu().numpy()
        for k, r, g in zip(ks, refs, got):
            e_max = np.abs(g - r).max() / np.abs(r).max()
            e_el = (np.abs(g - r) / np.abs(r)).max()
            row_err = max(row_err, np.abs(g.sum(axis=(1, 2)) - 1).max())
            if e_max > worst_max:
                worst_max, worst_k = e_max, k
            worst_elem = max(worst_elem, e_el)
    pool.close()
    print(f"dyad {dyad}: {len(pos)} windows x 64 x 64 x 256 compared with oracle.full_freq_dtf in {time.time() - t0:.0f} s")
    print(f"  worst window (max-norm relative): {worst_max:.3e} at window {worst_k}   [contract 1e-5, test guard 1e-9]")
    print(f"  worst single element (elementwise relative, values down to {float(ff.min()):.1e}): {worst_elem:.3e}")
    print(f"  worst |row sum - 1|: {row_err:.2e}")
    assert worst_max < 1e-9 and worst_elem < 1e-5


if __name__ == "__main__":
    main()
