#!/bin/bash
# A/B of two builds of the library on the same box: default vs affnet_amd/libaffnet_hip_b.so (tools/ab_trunk_exact.py), alternating
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
for i in 1 2 3; do python tools/ab_trunk_exact.py 2>/dev/null | tail -n 1; AFFNET_HIP_LIB=$PWD/affnet_amd/libaffnet_hip_b.so python tools/ab_trunk_exact.py 2>/dev/null | tail -n 1; done
