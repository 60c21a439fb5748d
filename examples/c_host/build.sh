#!/bin/bash
# Builds examples/c_host/extract (plain C99, gcc) against the in-tree libaffnet_hip.so and the HIP runtime.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$HERE/../.."
ROCM="${ROCM_PATH:-/opt/rocm}"
gcc -std=c99 -O2 -Wall -Wextra -Wno-unused-parameter -o "$HERE/extract" "$HERE/extract.c" -I "$ROOT/include" -I "$ROCM/include" \
    -L "$ROOT/affnet_amd" -laffnet_hip -L "$ROCM/lib" -lamdhip64 -Wl,-rpath,'$ORIGIN/../../affnet_amd' -Wl,-rpath,"$ROCM/lib"
echo "built $HERE/extract"
