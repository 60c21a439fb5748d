#!/bin/bash
# A/B of builds of the library on the same box: the default one vs every affnet_amd/libaffnet_hip_b*.so (AFFNET_HIP_LIB), stage times of a short headline bench, two rounds
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
run() { echo -n "$1: "; env $2 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-split3 --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_image']; print('%.0f kp/s aff %.4f ori %.4f hard %.4f' % (d['value'], s['affnet'], s['orinet'], s['hardnet_trunk']))"; }
for i in 1 2; do run default X=0; for f in affnet_amd/libaffnet_hip_b*.so; do run $(basename $f) AFFNET_HIP_LIB=$PWD/$f; done; done
