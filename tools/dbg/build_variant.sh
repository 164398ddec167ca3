#!/bin/bash
# Build hyperscanning_signal_analysis_amd/libhypermvar_<name>.so from the WORKING TREE's tf_inv.hip (extra -D flags
# allowed) and the regular objects of the other files: the variant for tools/dbg/ab.sh; `make` is not touched.
set -eo pipefail
NAME=$1; shift
cd "$(dirname "$0")/../../hyperscanning_signal_analysis_amd/csrc"
mkdir -p build/var
SRC=${SRC:-tf_inv}
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -mllvm -simplifycfg-sink-common=false -mllvm -simplifycfg-hoist-common=false \
    "$@" -Rpass-analysis=kernel-resource-usage -c $SRC.hip -o build/var/${SRC}_$NAME.o 2> build/var/${SRC}_$NAME.remarks || { tail -20 build/var/${SRC}_$NAME.remarks; exit 1; }
python3 - <<PY
import re
txt = open("build/var/${SRC}_$NAME.remarks").read()
for blk in re.split(r"remark: Function Name: ", txt)[1:]:
    name = blk.split()[0]
    g = lambda k: re.search(k + r": (\d+)", blk)
    v, sp, lds, occ = g("VGPRs"), g("VGPRs Spill"), g("LDS Size \[bytes/block\]"), g("Occupancy \[waves/SIMD\]")
    if "tf_inv_kernelILi4ELb0" in name or "tf_inv64_asm" in name or (sp and int(sp.group(1)) > 0):
        print(name[:60], "VGPRs", v.group(1), "spill", sp.group(1), "LDS", lds.group(1), "occ", occ.group(1))
PY
OBJS=""
for f in lagcov yw_solve yw_lwr yw_lwr2 tf_inv ffdtf_norm spectra connect psd dpss capi; do
  if [ $f = $SRC ]; then OBJS="$OBJS build/var/${f}_$NAME.o"; else OBJS="$OBJS build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o ../libhypermvar_$NAME.so -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib
ls -la ../libhypermvar_$NAME.so
