#!/bin/bash
set -eo pipefail
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02g
mkdir -p "$OUT"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > "$OUT/pytest.txt" 2>&1 || { tail -40 "$OUT/pytest.txt"; exit 1; }
tail -3 "$OUT/pytest.txt"
for lg in 1 2 3; do
  echo "== lag group $lg"
  HYPERMVAR_LAG_GROUP=$lg bash tools/prof_stats.sh r02g_lg$lg | grep -E "lagcov"
  for rep in 1 2; do HYPERMVAR_LAG_GROUP=$lg python bench.py --steps 10 --warmup 3 --no-cpu-baseline | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('bench ms/step %.3f K3 %.3f' % (r['ms_per_step'], r['roofline']['k3_ms_per_launch']))"; done
done
