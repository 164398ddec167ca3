#!/bin/bash
# A/B variant of the hand-scheduled K3 body: bash tools/dbg/k3a_variant.sh <name> <generator options, comma-separated | none> [-D...]
# -> hyperscanning_signal_analysis_amd/libhypermvar_<name>.so (compare with tools/dbg/ab_multi.sh base <name> ...).
set -eo pipefail
NAME=$1; OPTS=$2; shift 2
HERE="$(cd "$(dirname "$0")" && pwd)"
CSRC="$HERE/../../hyperscanning_signal_analysis_amd/csrc"
mkdir -p "$CSRC/build/var"
python3 "$CSRC/gen/k3gen.py" --opts "$OPTS" --out "$CSRC/build/var/body_$NAME.inc"
bash "$HERE/build_variant.sh" "$NAME" -DHMV_K3A_INC="\"build/var/body_$NAME.inc\"" "$@"
