#!/bin/bash
# bench.py's N > 1 code path with REAL kernels on the single-GPU box: N self-spawned ranks share cuda:0, backend gloo (RCCL cannot
# put two ranks on one device).  Rank 0 re-computes every gathered record of the last step itself (--verify-gather all): rank / world
# bookkeeping, record packing, the exchange, global order and content.  Then the RCCL flavour of the same path (stream-ordered async
# all-gather, no host sync inside the timed region) in a 1-rank group, also verified.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
for N in 2 4; do
  AFFNET_BENCH_BACKEND=gloo AFFNET_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus $N --steps 2 --warmup 1 --batch 8 --chunk 8 --no-secondary --verify-gather all > gpurun_out/bench_dist_dryrun_$N.log 2>&1
  echo "N=$N exit $?"; grep '^{' gpurun_out/bench_dist_dryrun_$N.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['gather_check'], d['exchange'], d['ms_per_step_per_rank'])"; tail -2 gpurun_out/bench_dist_dryrun_$N.log | cut -c1-300
done
AFFNET_BENCH_SELF_GATHER=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --verify-gather all > gpurun_out/bench_self_gather.log 2>&1
echo "self-gather exit $?"; grep '^{' gpurun_out/bench_self_gather.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['gather_check'], d['exchange'])"; grep -i "error\|Traceback" gpurun_out/bench_self_gather.log | head -5
# the other exchange mode through RCCL as well (round 5: gather to rank 0 is the default, --gather all = all_gather)
AFFNET_BENCH_SELF_GATHER=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-other-configs --no-split3 --gather all --verify-gather all > gpurun_out/bench_self_gather_all.log 2>&1
echo "self-gather (all_gather) exit $?"; grep '^{' gpurun_out/bench_self_gather_all.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['gather_check'], d['exchange'])"; grep -i "error\|Traceback" gpurun_out/bench_self_gather_all.log | head -5
