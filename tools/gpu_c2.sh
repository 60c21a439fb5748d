#!/bin/bash
# configs[1] single-image latency A/B: environment variants of the latency path (one line each: eager ms, graph ms)
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
run() { echo -n "$1: "; env $1 timeout 300 python bench.py --config2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eager %.4f  graph %s' % (d['value'], d['hip_graph']))"; }
for v in "$@"; do run "$v"; run "$v"; done
