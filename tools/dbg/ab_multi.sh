#!/bin/bash
# bench-only comparison of several variant libraries on one box: bash tools/dbg/ab_multi.sh name1 name2 ...
# (names as in libhypermvar_<name>.so; "base" = the regular library), two rounds.
set -eo pipefail
for round in $(seq 1 ${ROUNDS:-2}); do
  for n in "$@"; do
    if [ "$n" = base ]; then unset HYPERMVAR_LIB; else export HYPERMVAR_LIB=$PWD/hyperscanning_signal_analysis_amd/libhypermvar_$n.so; fi
    python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline > gpurun_out/abm_$n.json
    python - <<PY
import json
d = json.load(open("gpurun_out/abm_$n.json")); print("$n", round(d["ms_per_step"], 3), round(d["roofline"]["k3_ms_per_launch"], 3))
PY
  done
done
