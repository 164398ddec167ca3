"""Side measurement for BASELINE.json config 5 on ONE GPU: a synthetic tree of N dyads at full size (2 x 32 channels +
mastoids at 500 Hz; SECORE 220 s, three 60 s films, 180 s of talk per dyad) through `escan_batch.run` -- host
preprocessing in worker threads beside the GPU's MVAR / ffDTF + PSD work.  Prints the host / GPU time split as JSON
(kept as profiles/r03_escan_config5.json).  Usage: bench_escan_config5.py [dyads] [prefetch]"""
import json, os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hyperscanning_signal_analysis_amd import escan_batch as EB
from tests.test_gpu_escan_batch import _reader, make_config5_tree

n_dyads = int(sys.argv[1]) if len(sys.argv) > 1 else 6
prefetch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
with tempfile.TemporaryDirectory() as td:
    t0 = time.perf_counter()
    root = make_config5_tree(os.path.join(td, "tree"), [f"W_{k:03d}" for k in range(1, n_dyads + 1)])
    t_make = time.perf_counter() - t0
    rows = []
    root_j = make_config5_tree(os.path.join(td, "tree_jitter"), [f"W_{k:03d}" for k in range(1, n_dyads + 1)], jitter=True)
    plan = ((1, True), (prefetch, True), (prefetch, False), (prefetch, "jitter"))
    if len(sys.argv) > 3 and sys.argv[3] == "jitter-only":
        plan = ((prefetch, "jitter"),)
    for pf, psd in plan:
        timing = {}
        out = os.path.join(td, f"out_{pf}_{psd}")
        if pf == 1:                                   # the first pass also pays the DPSS tapers (cached afterwards)
            EB.run(root, os.path.join(td, "warm"), tasks=("talk",), with_psd=True, reader=_reader, verbose=False, prefetch=1)
        realistic = (psd == "jitter")        # every segment a length of its own, as real event durations are
        psd = bool(psd)
        res = EB.run(root_j if realistic else root, out, window_s=2.0, overlap=0.5, model_order=8, low_cutoff_hz=1.0, high_cutoff_hz=120.0,
                     with_psd=psd, psd_fmin=1.0, psd_fmax=30.0, psd_bandwidth=2.0, reader=_reader, verbose=False,
                     timing=timing, prefetch=pf)
        assert len(res["done"]) == n_dyads, res
        windows = 3 * 59 + 219 + 179
        rows.append({"prefetch_threads": pf, "with_psd": psd, "segment_lengths": "different for every dyad" if realistic else
                     "the same for every dyad (tapers, plans and tables cached)", "dyads": n_dyads, "windows_per_dyad": windows,
                     "wall_s": timing["wall_s"], "dyads_per_s": n_dyads / timing["wall_s"],
                     "windows_per_s": n_dyads * windows / timing["wall_s"],
                     "host_prepare_s_sum": timing["host_prepare_s"], "waited_for_host_s": timing["wait_for_host_s"],
                     "gpu_s_sum": timing["gpu_s"], "save_s_sum": timing["save_s"],
                     "gpu_busy_fraction_of_wall": timing["gpu_busy_fraction_of_wall"]})
    print(json.dumps({"what": "escan_batch.run, BASELINE config 5 shape on one MI355X, synthetic tree (NumPy reader)",
                      "host_cores": len(os.sched_getaffinity(0)), "tree_generation_s": t_make, "runs": rows}, indent=1))
