#!/bin/bash
# A/B on one box: write-through publish + in-K3 normalisation vs separate K4; publish cost alone (no normaliser).
set -eo pipefail
REPO=$(pwd)
OUT=$REPO/gpurun_out/ab_r02d
mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "normalisation_inside or northstar or two_stream or edge_cases or g3 or g4 or multi_dyad" > "$OUT/pytest.txt" 2>&1 || { tail -30 "$OUT/pytest.txt"; exit 1; }
tail -3 "$OUT/pytest.txt"
NN=$REPO/hyperscanning_signal_analysis_amd/libhypermvar_nonorm.so
for rep in 1 2 3; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/fused_$rep.json" 2> "$OUT/fused_$rep.err"
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --unfused-norm > "$OUT/unfused_$rep.json" 2> "$OUT/unfused_$rep.err"
  echo "rep $rep done"
done
python - <<'PY'
import json,glob,os
out=os.path.join(os.getcwd(),"gpurun_out","ab_r02d")
for tag in ("fused","unfused"):
    v=[json.load(open(f)) for f in sorted(glob.glob(f"{out}/{tag}_[0-9].json"))]
    print(tag, ["%.3f ms (K3 %.3f)"%(r["ms_per_step"], r["roofline"]["k3_ms_per_launch"]) for r in v])
PY
