#!/bin/bash
# One gpurun call: GPU parity tests + smoke + short bench + rocprofv3 kernel stats.  Outputs under gpurun_out/.
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== rocminfo"; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx950|Compute Unit" | head -6
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -q -m gpu -s -p no:cacheprovider ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?"
tail -n 60 gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?"; tail -n 5 gpurun_out/smoke.log
if [ -z "$SKIP_BENCH" ]; then
echo "== bench"
timeout 600 python bench.py --steps ${BENCH_STEPS:-2} --warmup 1 ${BENCH_ARGS} > gpurun_out/bench.log 2>&1; echo "bench exit: $?"; tail -n 3 gpurun_out/bench.log
fi
if [ -n "$DO_PROF" ]; then
echo "== rocprofv3 kernel stats"
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o run -- python bench.py --steps 1 --warmup 1 --batch 16 --no-cpu-baseline > gpurun_out/prof.log 2>&1; echo "prof exit: $?"
find gpurun_out/prof -name "*stats*" | head; 
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 25 "$f"
fi
