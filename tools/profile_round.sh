#!/bin/bash
# Round profile set, run ON the GPU box from the repo root:
#     gpurun --timeout 900 -- 'bash tools/profile_round.sh r01'
# Four separate rocprofv3 passes of the same bench command (kernel trace + stats; FETCH_SIZE;
# WRITE_SIZE; SQ counters) -- counters never share a pass with each other's groups or with a
# runtime trace -- then the plain bench line.  tools/summarise_profiles.py turns the raw output
# under gpurun_out/ into the committed files under profiles/.
set -eo pipefail
TAG=${1:-r03}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="$REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side"
PMCB="$REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run --output-format csv -- python3 $BENCH > "$OUT/stats.log" 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o run --output-format csv -- python3 $PMCB > "$OUT/fetch.log" 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o run --output-format csv -- python3 $PMCB > "$OUT/write.log" 2>&1
echo "write pass done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES \
    -d "$OUT/sq" -o run --output-format csv -- python3 $PMCB > "$OUT/sq.log" 2>&1 || echo "sq pass failed (non-fatal)"
echo "sq pass done"
rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d "$OUT/sq2" -o run --output-format csv -- python3 $PMCB > "$OUT/sq2.log" 2>&1 || echo "sq2 pass failed (non-fatal)"
echo "sq2 pass done"
cd "$REPO"
python3 bench.py --steps 5 --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.err"
cat "$OUT/bench.json"
# side measurement: the ffDTF + spectra path (kernel stats only)
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats_spectra" -o run --output-format csv -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side --with-spectra > "$OUT/stats_spectra.log" 2>&1 || echo "spectra stats pass failed (non-fatal)"
cd "$REPO"
