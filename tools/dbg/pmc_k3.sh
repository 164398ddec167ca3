#!/bin/bash
# SQ counters of K3 in two passes (8 SQ slots each) + GRBM_GUI_ACTIVE for the effective clock; run on the GPU box from the
# repo root: bash tools/dbg/pmc_k3.sh <tag>.  Output: gpurun_out/pmc_<tag>/{a,b}/... and a condensed table on stdout.
set -eo pipefail
TAG=${1:-x}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
PMCB="$REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES \
    -d "$OUT/a" -o run --output-format csv -- python3 $PMCB > "$OUT/a.log" 2>&1 || echo "pass a failed"
rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
    -d "$OUT/b" -o run --output-format csv -- python3 $PMCB > "$OUT/b.log" 2>&1 || echo "pass b failed"
cd "$REPO"
python3 - <<PY
import csv, glob, collections
for p in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "tf_inv" in k:
                acc[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        for c, v in sorted(d.items()):
            print(p, k, c, "%.4g" % (sum(v) / len(v)), "n=%d" % len(v))
PY
