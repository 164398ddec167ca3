#!/usr/bin/env python3
"""All-cores CPU figure for bench.py's cpu_baseline: the oracle's vectorised restatement (oracle.full_freq_dtf)
in one process per core, one BLAS thread each, on windows of a recording handed over as a .npy file.  Runs as a
CHILD process of bench.py (no GPU, no torch): prints one JSON object.

    python tools/cpu_allcores.py <x.npy> <window> <hop> <p> <n_freqs> <fs> <processes> <seconds>
"""
import json
import multiprocessing as mp
import os
import sys
import time

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mvar_oracle as O      # bench.py's cpu_baseline leg: the oracle as the thing timed on the host

_X = None


def _work(args):
    starts, w, p, freqs, fs, budget = args
    done, t0 = 0, time.perf_counter()
    for s in starts:
        O.full_freq_dtf(_X[:, s:s + w], freqs, fs, p)
        done += 1
        if time.perf_counter() - t0 > budget:
            break
    return done, time.perf_counter() - t0


def main():
    global _X
    path, w, hop, p, F, fs, procs, budget = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), \
        int(sys.argv[5]), float(sys.argv[6]), int(sys.argv[7]), float(sys.argv[8])
    _X = np.load(path)
    freqs = 0.5 * np.arange(1, F + 1)
    starts = np.arange(0, _X.shape[1] - w + 1, hop)
    shares = [np.roll(starts, -k * 7)[:4096] for k in range(procs)]
    with mp.get_context("fork").Pool(procs) as pool:           # workers inherit _X; nothing here touches a GPU
        res = pool.map(_work, [(sh, w, p, freqs, fs, budget) for sh in shares])
    busy = max(r[1] for r in res)
    n = int(sum(r[0] for r in res))
    print(json.dumps({"value": n / busy, "unit": "windows/s", "processes": procs, "blas_threads": 1, "windows": n,
                      "seconds": busy, "what": "oracle.full_freq_dtf (vectorised restatement), one process per core"}))


if __name__ == "__main__":
    main()
