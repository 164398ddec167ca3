#!/bin/bash
# rocprofv3 kernel stats of one bench command: bash tools/prof_stats.sh <tag> [bench args...]
set -eo pipefail
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run --output-format csv -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > "$OUT/stats.log" 2>&1
cd "$REPO"
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
cp "$f" "$OUT/kernel_stats.csv"
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:10.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms  {r['Percentage']}%")
PY
tail -1 "$OUT/stats.log"
