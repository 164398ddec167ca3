#!/usr/bin/env python3
"""MVAR + ffDTF windows/sec on MI355X -- the metric of BASELINE.json.

Workload (BASELINE.json configs[1], SURVEY.md section 8(d) "C2"): per GPU ONE synthetic dyad,
2 x 32 channels @ 500 Hz, 10 minutes (T = 300 000), 2 s windows with 50 % overlap (599 windows),
MVAR order p = 8, 256-point frequency grid 0.5 .. 128 Hz, float64.  One "step" = one pass of the hot path
(K1 lag covariance -> K2 Yule-Walker -> K3 transfer inverse + |H|^2 -> K4 ffDTF normalisation) over
those 599 windows, input already resident in HBM, output left in HBM as (599, 64, 64, 256) float64.
N > 1: weak scaling, rank r processes dyad r (no data-path collective); after the K timed steps ONE
RCCL gather of the band-integrated ffDTF to rank 0 (inside the timed region).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` for the dominant
kernel (K3, timed with HIP events recorded by the library around its launch inside the timed region)
and `cpu_baseline` (the NumPy port of the reference's loop structure, oracle/mvar_oracle.py, on a
bounded sample of the same windows, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from hyperscanning_signal_analysis_amd import distributed as hdist          # noqa: E402
from hyperscanning_signal_analysis_amd.engine import Engine                 # noqa: E402
from hyperscanning_signal_analysis_amd.sliding import window_items, window_positions  # noqa: E402
from hyperscanning_signal_analysis_amd.synthetic import NORTHSTAR, northstar_freqs, synthetic_var_dyad  # noqa: E402

# Algorithmic flops per window; K3 = A(f) build + complex inverses + |H|^2 + row sums
FLOP_LAGCOV = 73.43e6
FLOP_YW = 78.29e6
FLOP_AF = 33.55e6
FLOP_INV = 536.87e6
FLOP_NORM = 5.24e6
FLOP_WINDOW = FLOP_LAGCOV + FLOP_YW + FLOP_AF + FLOP_INV + FLOP_NORM          # 727.4 MFLOP
FLOP_K3_WINDOW = FLOP_AF + FLOP_INV + 3.0 * 64 * 64 * 256                    # 573.6 MFLOP
PEAK_F64_TFLOPS = 78.6          # MI355X spec, vector = matrix f64 (measured ceiling ~65: DESIGN.md)


def cpu_baseline(x_host, positions, w, p, freqs, fs, budget_s=15.0, max_windows=256):
    """Reference-style (per-frequency Python loop) NumPy port timed on this host, 1 BLAS thread."""
    from oracle import mvar_oracle as O      # the ONLY place bench.py touches oracle/
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except Exception:                         # pragma: no cover
        import contextlib
        ctx = contextlib.nullcontext()
    done, t0 = 0, time.perf_counter()
    with ctx:
        for s in positions[:max_windows]:
            O.full_freq_dtf_loop(x_host[:, s:s + w], freqs, fs, p)
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "windows/s", "cores": 1, "kind": "port",
            "sample": f"first {done} of the {len(positions)} windows of dyad 0, oracle.full_freq_dtf_loop "
                      f"(reference loop structure, NumPy/OpenBLAS 1 thread), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--minutes", type=float, default=10.0, help="recording length per dyad (default 10)")
    ap.add_argument("--single-stream", action="store_true", help="do not split K2 over two HIP streams")
    ap.add_argument("--unfused-norm", action="store_true",
                    help="ffDTF normalisation as a separate pass (K4) instead of inside K3 (A/B measurements)")
    ap.add_argument("--yw-one-launch", action="store_true",
                    help="K2 as one workgroup per window in one launch instead of the chain of tile launches (A/B)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="process-group backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 code path "
                         "with several ranks on ONE GPU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    rehearsal = (args.backend == "gloo")                 # every rank on GPU 0, host-side collectives
    dev = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    ns = NORTHSTAR
    m, fs, w, p, F = ns["m"], ns["fs"], ns["window"], ns["p"], ns["F"]
    T = int(round(args.minutes * 60 * fs))
    n_windows = 2 * T // w - 1
    positions, w = window_positions(T, n_windows, w)
    freqs = northstar_freqs(F)

    x_host = synthetic_var_dyad(rank, m=m, p=p, T=T, fs=fs)           # weak scaling: dyad index = rank
    eng = Engine(device=dev, max_workspace_bytes=64 << 30)
    x = eng.to_device(x_host[None])                                     # (1, 64, T) resident in HBM
    item_rec, item_start = window_items(1, positions, dev)
    fdev = eng.to_device(freqs)
    out = eng.empty(n_windows, m, m, F)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:                                                     # create the hipEvent_t handles
        a.record(); b.record()
    torch.cuda.synchronize()

    chunk = n_windows                       # one chunk per step: K1, K3, K4 are launched once per step
    k3_windows = n_windows                  # windows of the K3 launch the events bracket
    two_streams = not args.single_stream    # K2 as two half-batches on two HIP streams (library option)

    from hyperscanning_signal_analysis_amd import _lib as hlib
    flags = (hlib.FLAG_UNFUSED_NORM if args.unfused_norm else 0) | (hlib.FLAG_YW_ONE_LAUNCH if args.yw_one_launch else 0)

    def step(k3_events=None):
        eng.sliding_ffdtf(x, item_rec, item_start, w, p, fdev, fs, out=out, check=False,
                          chunk=chunk, k3_events=k3_events, overlap=two_streams, flags=flags)

    def barrier():
        if world > 1:
            if rehearsal:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step((ev[k][0].cuda_event, ev[k][1].cuda_event))
    if world > 1:                                                       # the single gather at the end
        bands = hdist.band_integrate(out, freqs)
        gathered = hdist.gather_to_root(bands.cpu() if rehearsal else bands, dst=0)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # sanity inside the bench: rows of every window sum to one, nothing singular
    rowsum_err = float((out.sum(dim=(2, 3)) - 1.0).abs().max().item())
    if not os.environ.get("HYPERMVAR_BENCH_NOCHECK"):          # diagnostic kernel variants only
        assert rowsum_err < 1e-9, f"ffDTF rows do not sum to 1 ({rowsum_err})"

    if rank == 0:
        k3_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        windows_total = world * n_windows * args.steps
        value = windows_total / dt
        achieved = FLOP_K3_WINDOW * k3_windows / (k3_ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "k3_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "MVAR+ffDTF windows/sec, 2x32-ch dyad p=8, 256 freqs",
            "value": value, "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: 1 dyad/GPU, 2x32 ch @500 Hz, %g min, 2 s windows 50%% overlap "
                                   "(%d windows), MVAR p=8, 256 freqs 0.5-128 Hz" % (args.minutes, n_windows),
                       "windows_per_step_per_gpu": n_windows, "k2": "one workgroup per window, one launch" if args.yw_one_launch else "tile launches on %d stream(s)" % (2 if two_streams else 1),
                       "normalisation": "separate K4 pass" if args.unfused_norm else "inside K3 (last arriver)",
                       "parallelism": f"dyad-sharded x{world}",
                       "gather": "band-integrated ffDTF to rank 0 (once, timed)" if world > 1 else "none"},
            "roofline": {"bound": "mfma", "kernel": "tf_inv_kernel<4, false> (K3)", "achieved": achieved,
                         "peak": PEAK_F64_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F64_TFLOPS,
                         "traffic": traffic, "k3_ms_per_launch": k3_ms,
                         "flop_per_launch": FLOP_K3_WINDOW * k3_windows, "windows_per_launch": k3_windows},
            "path_tflops": FLOP_WINDOW * value / world / 1e12,
            "path_frac_of_peak": FLOP_WINDOW * value / world / 1e12 / PEAK_F64_TFLOPS,
        }
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(x_host, positions, w, p, freqs, fs)
        elif not args.no_cpu_baseline:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
