#!/bin/bash
set -eo pipefail
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02j
mkdir -p "$OUT"
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > "$OUT/pytest.txt" 2>&1 || { tail -40 "$OUT/pytest.txt"; exit 1; }
tail -3 "$OUT/pytest.txt"
for rep in 1 2 3; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/new_$rep.json" 2> "$OUT/new_$rep.err" || { tail "$OUT/new_$rep.err"; exit 1; }
  HYPERMVAR_LIB=$REPO/hyperscanning_signal_analysis_amd/libhypermvar_ywold.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/old_$rep.json" 2> "$OUT/old_$rep.err"
done
python - <<'PY'
import json,glob,os
out=os.path.join(os.getcwd(),"gpurun_out","r02j")
for tag in ("new","old"):
    v=[json.load(open(f)) for f in sorted(glob.glob(f"{out}/{tag}_[0-9].json"))]
    print(tag, ["%.3f ms (K3 %.3f) %.0f w/s"%(r["ms_per_step"], r["roofline"]["k3_ms_per_launch"], r["value"]) for r in v])
PY
bash tools/prof_stats.sh r02j | grep -E "yw_|lagc"
