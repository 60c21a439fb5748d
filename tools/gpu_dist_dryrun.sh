#!/bin/bash
# Dry run of bench.py's N > 1 code path on the single-GPU box: 2 ranks share cuda:0, backend gloo (no RCCL between two
# processes on one device).  Validates rank / world bookkeeping, record packing, the gather and the JSON line.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp AFFNET_BENCH_BACKEND=gloo AFFNET_BENCH_ONE_DEVICE=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --batch 16 --chunk 8 > gpurun_out/bench_dist_dryrun.log 2>&1
echo "exit $?"; grep '^{' gpurun_out/bench_dist_dryrun.log | cut -c1-700; tail -3 gpurun_out/bench_dist_dryrun.log | cut -c1-300
# the RCCL flavour of the same path (stream-ordered async all-gather, no host sync inside the timed region) in a 1-rank group
unset AFFNET_BENCH_BACKEND AFFNET_BENCH_ONE_DEVICE
AFFNET_BENCH_SELF_GATHER=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_self_gather.log 2>&1
echo "self-gather exit $?"; grep '^{' gpurun_out/bench_self_gather.log | cut -c1-300; grep -i "error\|Traceback" gpurun_out/bench_self_gather.log | head -5
