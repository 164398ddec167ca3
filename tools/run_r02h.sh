#!/bin/bash
set -eo pipefail
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02h
mkdir -p "$OUT"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > "$OUT/pytest.txt" 2>&1 || { tail -40 "$OUT/pytest.txt"; exit 1; }
tail -3 "$OUT/pytest.txt"
for rep in 1 2 3; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/shared_$rep.json" 2> "$OUT/shared_$rep.err" || { tail "$OUT/shared_$rep.err"; exit 1; }
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --direct-lagcov > "$OUT/direct_$rep.json" 2> "$OUT/direct_$rep.err"
done
python - <<'PY'
import json,glob,os
out=os.path.join(os.getcwd(),"gpurun_out","r02h")
for tag in ("shared","direct"):
    v=[json.load(open(f)) for f in sorted(glob.glob(f"{out}/{tag}_[0-9].json"))]
    print(tag, ["%.3f ms (K3 %.3f) %.0f w/s"%(r["ms_per_step"], r["roofline"]["k3_ms_per_launch"], r["value"]) for r in v])
PY
bash tools/prof_stats.sh r02h_shared | grep -E "lagc|yw_|tf_inv|norm"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --with-spectra > "$OUT/spectra.json" 2> "$OUT/spectra.err" || { tail "$OUT/spectra.err"; exit 1; }
python -c "import json;r=json.load(open('$OUT/spectra.json'));print('with_spectra', r['with_spectra'])"
