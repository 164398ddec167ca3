python -m pytest tests -m gpu -x -q > gpurun_out/gputest_r03e.log 2>&1; tail -4 gpurun_out/gputest_r03e.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side > gpurun_out/bench_r03e.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/bench_r03e.json')); print(d['value'], d['ms_per_step'], d['roofline']['k3_ms_per_launch'], d['checks_after_timed_region'])"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_tail -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f=glob.glob("gpurun_out/prof_tail/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]: print(r["Name"][:60], r["Calls"], "%.3f"%(float(r["AverageNs"])/1e6))
PY
