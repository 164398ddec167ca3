#!/bin/bash
# kernel trace of one bench run with a variant library: bash tools/dbg/trace_variant.sh <variant .so> <tag>
set -eo pipefail
REPO=$(pwd); export HYPERMVAR_LIB=$REPO/$1; TAG=$2
OUT=$REPO/gpurun_out/trace_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT" -o run --output-format csv -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/log.txt" 2>&1
cd "$REPO"
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/run_kernel_trace.csv")))
idx = [i for i, r in enumerate(rows) if "tf_inv" in r["Kernel_Name"]]
a, b = idx[1], idx[2]            # one steady-state step: from the end of K3 of step 1 to the end of K3 of step 2
base = int(rows[a]["End_Timestamp"])
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]) - base, int(r["End_Timestamp"]) - base
    print(f"{r['Kernel_Name'][:52]:52s} q{r['Queue_Id']} start {s/1e3:9.1f} us dur {(e-s)/1e3:8.1f} grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])} vgpr {r['VGPR_Count']} lds {r['LDS_Block_Size']}")
PY
