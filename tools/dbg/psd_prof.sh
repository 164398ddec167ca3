#!/bin/bash
# PSD timing and the kernel statistics of a config-5 batch whose segment lengths differ from dyad to dyad (tapers, FFT plans and
# chirp-z tables are new for every segment): bash tools/dbg/psd_prof.sh
python tests/side_benchmarks/bench_psd.py 2>/dev/null | head -3
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_escan -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tests/side_benchmarks/bench_escan_config5.py 4 8 jitter-only > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f=glob.glob("gpurun_out/prof_escan/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)/1e6
print("all kernels: %.1f ms"%tot)
for r in rows[:14]: print(r["Name"][:70], r["Calls"], "tot %.1f ms"%(float(r["TotalDurationNs"])/1e6), "avg %.3f"%(float(r["AverageNs"])/1e6))
PY
