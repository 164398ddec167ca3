#!/usr/bin/env python3
"""MVAR + ffDTF windows/sec on MI355X -- the metric of BASELINE.json.

Workload (BASELINE.json configs[1], SURVEY.md section 8(d) "C2"): per GPU ONE synthetic dyad,
2 x 32 channels @ 500 Hz, 10 minutes (T = 300 000), 2 s windows with 50 % overlap (599 windows),
MVAR order p = 8, 256-point frequency grid 0.5 .. 128 Hz, float64.  One "step" = one pass of the hot path
(K1 lag covariance -> K2 Yule-Walker -> K3 transfer inverse + |H|^2 + ffDTF normalisation, the rows of the
last windows of the batch by a small kernel right behind it) over those windows, input already resident in HBM, output left in HBM as
(windows, 64, 64, 256) float64.
`--dyads-per-gpu D` puts D dyads on every GPU (D x 599 windows per step, processed in chunks of one dyad);
`--gpus 8 --dyads-per-gpu 8` is BASELINE.json configs[2] ("C3": 64 dyads, dyad-sharded across 8 GPUs).
N > 1: weak scaling, rank r processes dyads r*D .. r*D+D-1 (no data-path collective); after the K timed steps
ONE RCCL gather of the band-integrated ffDTF to rank 0 (inside the timed region).
`--shard windows` is the strong-scaling mode (secondary partitioning: C2 at 2 / 4 / 8 GPUs): ONE dyad,
rank r uploads only the samples of its window range (read-only halo of one window) and computes those windows; same
gather; `"scaling": "strong"`.
Side measurements on the same line (N = 1, after the headline's timed region; `--no-side` skips them):
`end_to_end` -- recordings streamed from pinned host memory through `Engine.stream_dyads` (H2D of dyad d+1 and D2H of
the band-integrated result of dyad d-1 under the compute of dyad d), windows/s including PCIe both ways;
`with_spectra` -- ffDTF and the spectra S = H V H^T of every window from one fit.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` for the dominant
kernel (K3, timed with HIP events recorded by the library around its launch inside the timed region)
and `cpu_baseline` (the NumPy port of the reference's loop structure, oracle/mvar_oracle.py, on a
bounded sample of the same windows, rank 0 only; the vectorised restatement and an all-cores run are
reported as sub-fields).  The timed steps run with check=False (the singularity check is a device
synchronisation, not work); after the timed region one extra step is checked: both info arrays zero, every
ffDTF row sums to one, the in-kernel normalisation equals the separate K4 pass bit for bit on the whole
output, and one window equals the oracle.
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

from hyperscanning_signal_analysis_amd import _lib as hlib                  # noqa: E402
from hyperscanning_signal_analysis_amd import distributed as hdist          # noqa: E402
from hyperscanning_signal_analysis_amd.engine import Engine                 # noqa: E402
from hyperscanning_signal_analysis_amd.sliding import regular_grid, window_items, window_positions  # noqa: E402
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


def cpu_baseline(x_host, positions, w, p, freqs, fs, budget_s=12.0, max_windows=256):
    """CPU numbers beside the GPU one, all on bounded samples of dyad 0's windows:
    value      the reference-style port (per-frequency Python loop, per-(i,j) normalisation loop), 1 BLAS thread;
    vectorised the same math with batched LAPACK calls, 1 process / 1 BLAS thread;
    all_cores  the vectorised restatement in one process per core (multiprocessing), 1 BLAS thread each."""
    from oracle import mvar_oracle as O      # the ONLY place bench.py touches oracle/
    from threadpoolctl import threadpool_limits
    out = {}
    with threadpool_limits(limits=1):
        for key, fn, budget in (("loop", O.full_freq_dtf_loop, budget_s), ("vec", O.full_freq_dtf, budget_s / 3)):
            done, t0 = 0, time.perf_counter()
            for s in positions[:max_windows]:
                fn(x_host[:, s:s + w], freqs, fs, p)
                done += 1
                if time.perf_counter() - t0 > budget:
                    break
            out[key] = (done, time.perf_counter() - t0)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    procs = max(1, min(cores, 16))            # the GPU box gives one GPU's job a 16-core share
    allc = None
    try:                                      # a CHILD process (NumPy + oracle only, forks its own workers)
        import subprocess
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "x.npy")
            np.save(path, np.ascontiguousarray(x_host[:, :min(x_host.shape[1], 64 * w)]))
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_allcores.py"), path, str(w),
                                str(int(positions[1] - positions[0]) if len(positions) > 1 else w), str(p),
                                str(len(freqs)), str(fs), str(procs), str(budget_s / 2)],
                               capture_output=True, text=True, timeout=120)
        allc = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:                    # pragma: no cover  (never let the side measurement kill the bench)
        allc = {"error": repr(e)}
    d, dt = out["loop"]
    dv, dtv = out["vec"]
    return {"value": d / dt, "unit": "windows/s", "cores": 1, "kind": "port",
            "sample": f"first {d} of the {len(positions)} windows of dyad 0, oracle.full_freq_dtf_loop "
                      f"(reference loop structure, NumPy/OpenBLAS 1 thread), {dt:.1f} s",
            "vectorised": {"value": dv / dtv, "unit": "windows/s", "cores": 1, "blas_threads": 1,
                           "sample": f"first {dv} windows, oracle.full_freq_dtf (batched LAPACK), {dtv:.1f} s"},
            "all_cores": allc, "host_cores_visible": cores}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--minutes", type=float, default=10.0, help="recording length per dyad (default 10)")
    ap.add_argument("--dyads-per-gpu", type=int, default=1,
                    help="dyads per GPU (8 with --gpus 8 = BASELINE config 3: 64 dyads); one chunk per dyad")
    ap.add_argument("--single-stream", action="store_true", help="do not give the library its second HIP stream")
    ap.add_argument("--unfused-norm", action="store_true",
                    help="ffDTF normalisation as a separate pass (K4) instead of inside K3 (A/B measurements)")
    ap.add_argument("--yw-one-launch", action="store_true",
                    help="K2 as one workgroup per window in one launch instead of the chain of tile launches (A/B)")
    ap.add_argument("--direct-lagcov", action="store_true",
                    help="K1 sums every window from its own samples instead of sharing the 50 %% overlap (A/B)")
    ap.add_argument("--with-spectra", action="store_true",
                    help="side measurement: ffDTF AND multivariate spectra S = H V H^T of every window from one fit "
                         "(what the reference's orchestrators always compute together); reported as `with_spectra`")
    ap.add_argument("--shard", default="dyads", choices=("dyads", "windows"),
                    help="N > 1: dyads = weak scaling (one or D dyads per rank); windows = strong scaling (ONE dyad, every "
                         "rank a window range of it)")
    ap.add_argument("--no-side", action="store_true", help="skip the side measurements (end_to_end, with_spectra)")
    ap.add_argument("--stream-dyads", type=int, default=0,
                    help="recordings streamed in the end_to_end side measurement (default: max(4, dyads per GPU))")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="process-group backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 code path "
                         "with several ranks on ONE GPU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    rehearsal = (args.backend == "gloo")                 # every rank on GPU 0, host-side collectives
    if world > 1:                                         # rendezvous BEFORE anything touches the GPU
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
    dev = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)
    if world > 1 and not rehearsal:
        dist.init_process_group("nccl", device_id=dev)

    ns = NORTHSTAR
    m, fs, w, p, F = ns["m"], ns["fs"], ns["window"], ns["p"], ns["F"]
    T = int(round(args.minutes * 60 * fs))
    n_windows = 2 * T // w - 1
    positions, w = window_positions(T, n_windows, w)
    freqs = northstar_freqs(F)
    D = max(1, args.dyads_per_gpu)
    strong = (args.shard == "windows")
    eng = Engine(device=dev, max_workspace_bytes=64 << 30)
    fdev = eng.to_device(freqs)
    if strong:
        # strong scaling: ONE dyad (index 0, the same on every rank); this rank uploads only its window range's samples
        assert D == 1, "--shard windows is one dyad"
        x_hosts = [synthetic_var_dyad(0, m=m, p=p, T=T, fs=fs)]
        s_lo, s_hi, my_pos, (w_lo, w_hi) = hdist.shard_window_items(positions, w, world, rank)
        x = eng.to_device(np.ascontiguousarray(x_hosts[0][None, :, s_lo:s_hi]))
        item_rec, item_start = window_items(1, my_pos, dev)
        n_items = int(w_hi - w_lo)
        all_positions, positions = positions, my_pos
    else:
        # weak scaling: dyad index = rank * D + d
        x_hosts = [synthetic_var_dyad(rank * D + d, m=m, p=p, T=T, fs=fs) for d in range(D)]
        x = eng.to_device(np.stack(x_hosts))                            # (D, 64, T) resident in HBM
        item_rec, item_start = window_items(D, positions, dev)
        n_items = D * n_windows
        all_positions = positions
    out = eng.empty(n_items, m, m, F)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:                                                     # create the hipEvent_t handles
        a.record(); b.record()
    torch.cuda.synchronize()

    chunk = n_items if strong else n_windows   # one chunk per dyad: K1, K3 are launched once per dyad and step
    k3_windows = chunk                         # windows of the K3 launch the events bracket (the last chunk's)
    two_streams = not args.single_stream
    flags = (hlib.FLAG_UNFUSED_NORM if args.unfused_norm else 0) | (hlib.FLAG_YW_ONE_LAUNCH if args.yw_one_launch else 0)

    grid = None if args.direct_lagcov else regular_grid(positions, w, p)     # (hop, first, n_win): K1 shares the overlap

    def step(k3_events=None, check=False, dst=None, fl=None):
        return eng.sliding_ffdtf(x, item_rec, item_start, w, p, fdev, fs, out=out if dst is None else dst, check=check,
                                 chunk=chunk, k3_events=k3_events, overlap=two_streams,
                                 flags=flags if fl is None else fl, return_ar=check, grid=grid)

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
        bands = hdist.band_integrate(out, freqs, engine=eng)
        if strong:                                                      # window ranges differ by one: equal shapes for the gather
            n_max = -(-n_windows // world)
            if bands.shape[0] < n_max:
                bands = torch.cat([bands, bands.new_zeros(n_max - bands.shape[0], *bands.shape[1:])])
        gathered = hdist.gather_to_root(bands.cpu() if rehearsal else bands, dst=0)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        if rank == 0:
            assert gathered.shape[0] == world and bool(torch.isfinite(gathered).all())

    # ---- after the timed region: one checked step (diagnostic kernel variants may skip it)
    checks = {}
    if os.environ.get("HYPERMVAR_BENCH_NOCHECK"):
        checks = {"skipped": True}
    else:
        _, _, _, (info_yw, info_tf) = step(check=True)                 # raises on any singular window
        checks["info_all_zero"] = not (bool(info_yw.any()) or bool(info_tf.any()))
        rowsum_err = float((out.sum(dim=(2, 3)) - 1.0).abs().max().item())
        assert rowsum_err < 1e-9, f"ffDTF rows do not sum to 1 ({rowsum_err})"
        checks["max_row_sum_error"] = rowsum_err
        other = eng.empty(n_items, m, m, F)                             # the other normalisation path, every dyad
        eng.sliding_ffdtf(x, item_rec, item_start, w, p, fdev, fs, out=other, check=False,
                          chunk=chunk, overlap=two_streams, flags=flags ^ hlib.FLAG_UNFUSED_NORM, grid=grid)
        checks["fused_equals_separate_normalisation_bitwise"] = bool(torch.equal(other, out))
        assert checks["fused_equals_separate_normalisation_bitwise"], "in-kernel and separate normalisation differ"
        del other
        if rank == 0:
            from oracle import mvar_oracle as O                          # checker only
            kw = n_items // 2 if strong else n_windows // 2
            a0 = int(all_positions[kw + (w_lo if strong else 0)])
            ref = O.full_freq_dtf(x_hosts[0][:, a0:a0 + w], freqs, fs, p)
            got = out[kw].cpu().numpy()
            checks["oracle_window_rel_err"] = float(np.abs(got - ref).max() / np.abs(ref).max())
            assert checks["oracle_window_rel_err"] < 1e-9

    # ---- side measurement: recordings streamed from (pinned) host memory, PCIe both ways inside the clock
    e2e_res = None
    side = (world == 1 and not strong and not args.no_side)
    if side:
        E = args.stream_dyads or max(16, D)
        feed_src = [torch.from_numpy(xh).pin_memory() for xh in x_hosts]          # what a loader would hand over
        feed = [feed_src[d % len(feed_src)] for d in range(E)]
        nb = len(hdist.DEFAULT_BANDS)
        host_out = torch.empty(E, n_windows, m, m, nb, dtype=torch.float64).pin_memory()   # the caller's result buffer
        eng.stream_dyads(feed[:2], w, positions, p, fdev, fs, out=host_out)        # warm-up: buffers, streams
        torch.cuda.synchronize()
        te0 = time.perf_counter()
        red = eng.stream_dyads(feed, w, positions, p, fdev, fs, out=host_out)
        torch.cuda.synchronize()
        te = time.perf_counter() - te0
        ref_b = hdist.band_integrate(out[:n_windows], freqs, engine=eng).cpu().numpy()
        e2e_res = {"windows_per_s": E * n_windows / te, "dyads_streamed": E, "ms_per_dyad": te / E * 1e3,
                   "fraction_of_resident_rate": (E * n_windows / te) / (world * n_items * args.steps / dt),
                   "h2d_bytes_per_dyad": int(x_hosts[0].nbytes), "d2h_bytes_per_dyad": int(red[0].nbytes),
                   "what": "Engine.stream_dyads: pinned host recordings -> H2D on a copy stream under the compute of the "
                           "previous dyad -> band-integrated ffDTF (windows, 64, 64, 5) D2H on a third stream",
                   "first_dyad_equals_resident_result_bitwise": bool(np.array_equal(red[0], ref_b))}
        assert e2e_res["first_dyad_equals_resident_result_bitwise"], "streamed and resident results differ"
        del feed, feed_src, red, host_out

    spectra_res = None
    if args.with_spectra or side:            # ffDTF + spectra of dyad 0 from one fit, after the headline measurement
        nsp = n_windows
        S_out = eng.empty(nsp, m, m, F, 2)
        ff_sp = eng.empty(nsp, m, m, F)

        def sp_step():
            return eng.sliding_ffdtf_spectra(x[:1], item_rec[:nsp], item_start[:nsp], w, p, fdev, fs, chunk=nsp,
                                             check=False, out_ff=ff_sp, out_S=S_out, grid=grid if not strong else None)
        sp_step()
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        for _ in range(4):
            sp_step()
        torch.cuda.synchronize()
        ts = (time.perf_counter() - ts0) / 4
        spectra_res = {"windows_per_s": nsp / ts, "ms_per_window": ts / nsp * 1e3, "windows": nsp,
                       # (same K1 form as the headline path: the same bits, i.e. 0.0)
                       "ffdtf_max_rel_diff_to_headline_path": float((ff_sp - out[:nsp]).abs().max() / out[:nsp].abs().max()),
                       # algorithmic: S = H V H^T in full, V real = six real 64^3 products per (window, frequency); the
                       # kernel computes the upper triangle only (4.5 products)
                       "flop_per_window": FLOP_WINDOW + 805.3e6,
                       "tflops": (FLOP_WINDOW + 805.3e6) * nsp / ts / 1e12}
        del S_out, ff_sp

    if rank == 0:
        k3_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        windows_total = world * n_items * args.steps
        value = windows_total / dt
        achieved = FLOP_K3_WINDOW * k3_windows / (k3_ms * 1e-3) / 1e12
        k3_form = int(eng.lib.hmv_get_tuning(hlib.TUNE_K3_FORM))
        k3_name = "tf_inv_kernel<4, false>" if k3_form == 1 else "tf_inv64_asm_kernel"
        # HBM bytes per K3 launch from the PMC passes of tools/profile_round.sh: valid only for the kernel it was measured
        # on (the file names it) -- a stale figure is dropped, not reported
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "k3_traffic.json")
        if os.path.exists(pmc):
            try:
                tj = json.load(open(pmc))
                if k3_name.split("<")[0] in tj.get("kernel", "") and not args.unfused_norm:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_src = "profiles/k3_traffic.json (%s, %s)" % (tj.get("round", "?"), tj.get("kernel", "?"))
                else:
                    traffic_src = "none: profiles/k3_traffic.json was measured on '%s'" % tj.get("kernel", "?")
            except Exception:
                traffic = None
        res = {
            "metric": "MVAR+ffDTF windows/sec, 2x32-ch dyad p=8, 256 freqs",
            "value": value, "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("C2 strong scaling: ONE dyad, window ranges over %d GPU(s), 2x32 ch @500 Hz, %g min, 2 s "
                                    "windows 50%% overlap (%d windows), MVAR p=8, 256 freqs 0.5-128 Hz"
                                    % (world, args.minutes, n_windows)) if strong else
                                   "C%d: %d dyad(s)/GPU, 2x32 ch @500 Hz, %g min, 2 s windows 50%% overlap "
                                   "(%d windows per dyad), MVAR p=8, 256 freqs 0.5-128 Hz"
                                   % (2 if D == 1 else 3, D, args.minutes, n_windows),
                       "windows_per_step_per_gpu": n_items, "dyads_per_gpu": D,
                       "k1": "every window from its own samples" if grid is None else
                             "hop blocks summed once, shared by the overlapping windows (hop %d)" % grid[0],
                       "k2": "block LDL^T, one workgroup per window, one launch" if args.yw_one_launch
                             else ("block LDL^T (HYPERMVAR_YW_FORM=1)" if int(eng.lib.hmv_get_tuning(hlib.TUNE_YW_FORM)) == 1
                                   else "block Levinson-Whittle recursion, one workgroup per window, one launch"),
                       "pivot_tau": eng.pivot_tau,
                       "normalisation": "separate K4 pass" if args.unfused_norm else "inside K3 (rows of window w by "
                                        "workgroups of window w+lag)",
                       "timed_steps_check_singularity": False,
                       "parallelism": (f"window-range-sharded x{world}" if strong else f"dyad-sharded x{world}"),
                       "gather": "band-integrated ffDTF to rank 0 (once, timed)" if world > 1 else "none"},
            "roofline": {"bound": "mfma", "kernel": "%s (K3%s)"
                                  % (k3_name, "" if args.unfused_norm else ", incl. the in-kernel ffDTF normalisation"),
                         "achieved": achieved,
                         "peak": PEAK_F64_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F64_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_src, "k3_ms_per_launch": k3_ms,
                         "flop_per_launch": FLOP_K3_WINDOW * k3_windows, "windows_per_launch": k3_windows},
            # ALGORITHMIC flops of the direct form (727.4 MFLOP per window); with the shared hop blocks
            # K1 executes about half of its 73.4 MFLOP, i.e. the work actually done is ~5 % less than this figure counts
            "path_tflops": FLOP_WINDOW * value / world / 1e12,
            "path_frac_of_peak": FLOP_WINDOW * value / world / 1e12 / PEAK_F64_TFLOPS,
            "path_flops_are": "algorithmic (direct-form K1), not executed",
            "checks_after_timed_region": checks,
        }
        if spectra_res is not None:
            res["with_spectra"] = spectra_res
        if e2e_res is not None:
            res["end_to_end"] = e2e_res
        if not args.no_cpu_baseline:           # rank 0's host cores, whatever N is
            res["cpu_baseline"] = cpu_baseline(x_hosts[0], all_positions, w, p, freqs, fs)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
