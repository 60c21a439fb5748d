#!/bin/bash
# Builds libaffnet_hip.so for gfx950 in-tree (affnet_amd/libaffnet_hip.so).
# -ffp-contract=off: the detector / sampler reproduce the reference's fp32 operation sequence
# exactly; fused multiply-adds appear only where written as fmaf().
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../libaffnet_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $EXTRA_HIPCC_FLAGS"
mkdir -p "$HERE/obj"
pids=()
for f in config_fill context pyramid detect laf_ops cnn32 pipeline match handcrafted debug fullconv split_probe; do
  if [ ! -f "$HERE/obj/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/obj/$f.o" ] || [ "$HERE/common.h" -nt "$HERE/obj/$f.o" ] || [ "$HERE/cnn_mfma.h" -nt "$HERE/obj/$f.o" ] || [ "$HERE/../../include/affnet_hip.h" -nt "$HERE/obj/$f.o" ] || [ "$HERE/../../include/affnet_hip_debug.h" -nt "$HERE/obj/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$HERE/obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$HERE"/obj/*.o
echo "built $OUT"
