#!/bin/bash
# K3's normalisation lag (windows between a window's matrices and its row tasks): bench at several values, one box.
for round in 1 2; do
for lag in ${LAGS:-24 32 40}; do
  HYPERMVAR_NORM_LAG=$lag python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-side > gpurun_out/lag_$lag.json
  python - <<PY
import json
d = json.load(open("gpurun_out/lag_$lag.json")); print("lag $lag", round(d["ms_per_step"], 3), round(d["roofline"]["k3_ms_per_launch"], 3))
PY
done; done
