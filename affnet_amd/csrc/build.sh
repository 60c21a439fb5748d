#!/bin/bash
# Builds libaffnet_hip.so for gfx950 in-tree (affnet_amd/libaffnet_hip.so): product kernels + the stamped debug instantiations only.
# AFFNET_PROBES=1: builds affnet_amd/libaffnet_hip_probes.so instead - the same sources with -DAFFNET_PROBES plus split_probe.hip, i.e. the
# library with the probe kernels of the tuning / measurement tools (include/affnet_hip_probes.h); objects in obj_probes/.
# -ffp-contract=off: the detector / sampler reproduce the reference's fp32 operation sequence
# exactly; fused multiply-adds appear only where written as fmaf().
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $EXTRA_HIPCC_FLAGS"
FILES="config_fill context pyramid detect laf_ops cnn32 pipeline match handcrafted fullconv"
OBJ="$HERE/obj"
OUT="$HERE/../libaffnet_hip.so"
if [ "$AFFNET_PROBES" = "1" ]; then
  FLAGS="$FLAGS -DAFFNET_PROBES"
  FILES="$FILES debug split_probe"
  OBJ="$HERE/obj_probes"
  OUT="$HERE/../libaffnet_hip_probes.so"
fi
mkdir -p "$OBJ"
pids=()
for f in $FILES; do
  if [ ! -f "$OBJ/$f.o" ] || [ "$HERE/$f.hip" -nt "$OBJ/$f.o" ] || [ "$HERE/common.h" -nt "$OBJ/$f.o" ] || [ "$HERE/cnn_mfma.h" -nt "$OBJ/$f.o" ] || [ "$HERE/shape_filter.h" -nt "$OBJ/$f.o" ] || [ "$HERE/../../include/affnet_hip.h" -nt "$OBJ/$f.o" ] || [ "$HERE/../../include/affnet_hip_debug.h" -nt "$OBJ/$f.o" ] || [ "$HERE/../../include/affnet_hip_probes.h" -nt "$OBJ/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$OBJ/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
objs=""
for f in $FILES; do objs="$objs $OBJ/$f.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" $objs
echo "built $OUT"
