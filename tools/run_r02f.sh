#!/bin/bash
set -eo pipefail
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02f
mkdir -p "$OUT"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > "$OUT/pytest.txt" 2>&1 || { tail -40 "$OUT/pytest.txt"; exit 1; }
tail -3 "$OUT/pytest.txt"
python bench.py --steps 10 --warmup 3 > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err" || { tail -20 "$OUT/bench_n1.err"; exit 1; }
python -c "import json;r=json.load(open('$OUT/bench_n1.json'));print('N=1', r['value'], r['ms_per_step'], r['roofline']['frac'], r['checks_after_timed_region'], {k:(v if not isinstance(v,dict) else v.get('value')) for k,v in r['cpu_baseline'].items() if k!='sample'})"
python bench.py --steps 3 --warmup 1 --dyads-per-gpu 3 --no-cpu-baseline > "$OUT/bench_d3.json" 2> "$OUT/bench_d3.err" || { tail -20 "$OUT/bench_d3.err"; exit 1; }
python -c "import json;r=json.load(open('$OUT/bench_d3.json'));print('D=3', r['value'], r['ms_per_step'], r['config']['workload'])"
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_gloo2.json" 2> "$OUT/bench_gloo2.err" || { tail -30 "$OUT/bench_gloo2.err"; exit 1; }
tail -1 "$OUT/bench_gloo2.json" | python -c "import json,sys;r=json.loads(sys.stdin.read());print('gloo x2 on one GPU', r['value'], r['ms_per_step'], r['n_gpus'], r['config']['gather'])"
