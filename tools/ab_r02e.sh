#!/bin/bash
# A/B on one box: K2 one launch (workgroup per window) vs round-1 tile launch chain.
set -eo pipefail
REPO=$(pwd)
OUT=$REPO/gpurun_out/ab_r02e
mkdir -p "$OUT"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > "$OUT/pytest.txt" 2>&1 || { tail -30 "$OUT/pytest.txt"; exit 1; }
tail -3 "$OUT/pytest.txt"
for rep in 1 2 3; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/new_$rep.json" 2> "$OUT/new_$rep.err"
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --yw-tiled > "$OUT/tiled_$rep.json" 2> "$OUT/tiled_$rep.err"
  echo "rep $rep done"
done
python - <<'PY'
import json,glob,os
out=os.path.join(os.getcwd(),"gpurun_out","ab_r02e")
for tag in ("new","tiled"):
    v=[json.load(open(f)) for f in sorted(glob.glob(f"{out}/{tag}_[0-9].json"))]
    print(tag, ["%.3f ms (K3 %.3f)"%(r["ms_per_step"], r["roofline"]["k3_ms_per_launch"]) for r in v])
PY
bash tools/prof_stats.sh r02e_k2new > "$OUT/stats.txt" 2>&1; grep -E "yw_|lagcov|tf_inv|norm" "$OUT/stats.txt"
