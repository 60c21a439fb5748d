#!/bin/bash
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
run() { echo -n "$1: "; env $1 timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-split3 --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_image']; print('%.0f kp/s aff %.4f ori %.4f hard %.4f' % (d['value'], s['affnet'], s['orinet'], s['hardnet_trunk']))"; }
run "X=0"
run "AFFNET_TRUNK_PERSIST=7"
run "AFFNET_TRUNK_PERSIST=1"
run "AFFNET_TRUNK_PERSIST=6 AFFNET_TRUNK_DELAY=0"
