#!/bin/bash
# After a change of bench.py only (no kernel change): refresh the two lines of the evidence set it affects - the driver's command and the 1-rank RCCL path.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
bench() { local n=$1; shift; ( time timeout 900 python3 bench.py "$@" ) > gpurun_out/$n.out 2> gpurun_out/$n.err; echo "$n exit: $?"; cp gpurun_out/bench_detail.json gpurun_out/$n.detail.json 2>/dev/null; tail -n 1 gpurun_out/$n.out | cut -c1-200; grep "^real" gpurun_out/$n.err; }
bench bench_driver --gpus 1 --steps 20 --warmup 5
AFFNET_BENCH_SELF_GATHER=1 bench bench_self_gather_rccl_1rank --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-other-configs --no-split3 --verify-gather all
wc -l gpurun_out/bench_self_gather_rccl_1rank.out
