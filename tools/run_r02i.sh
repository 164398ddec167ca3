#!/bin/bash
set -eo pipefail
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02i
mkdir -p "$OUT"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > "$OUT/pytest.txt" 2>&1 || { tail -40 "$OUT/pytest.txt"; exit 1; }
tail -3 "$OUT/pytest.txt"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --with-spectra > "$OUT/spectra.json" 2> "$OUT/spectra.err" || { tail "$OUT/spectra.err"; exit 1; }
python -c "import json;r=json.load(open('$OUT/spectra.json'));print('with_spectra', r['with_spectra'])"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats_spectra" -o run --output-format csv -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --with-spectra > "$OUT/stats_spectra.log" 2>&1
cd $REPO
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r02i/stats_spectra/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:10.1f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
