#!/bin/bash
# Same-box A/B of two builds of the library: parity tests on the variant, then bench base / variant / base.
# usage (on the GPU box, repo root): bash tools/dbg/ab.sh <variant .so> [extra bench flags]
set -eo pipefail
L=$PWD/$1; shift
HYPERMVAR_LIB=$L timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --deselect tests/test_gpu_parity.py::test_native_library_is_what_runs 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/ab_base.json
HYPERMVAR_LIB=$L python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/ab_var.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/ab_base2.json
HYPERMVAR_LIB=$L python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/ab_var2.json
python - <<PY
import json
for n in ("ab_base", "ab_var", "ab_base2", "ab_var2"):
    d = json.load(open("gpurun_out/%s.json" % n)); print(n, round(d["ms_per_step"], 3), round(d["roofline"]["k3_ms_per_launch"], 3), d["checks_after_timed_region"])
PY
