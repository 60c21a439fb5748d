#!/bin/bash
# round 5, first GPU call: all GPU tests with the float64-referee accounting (parity report + the HIP path's rows dumped for offline analysis),
# smoke, and the default bench line.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
export AFFNET_PARITY_REPORT=$PWD/gpurun_out/parity_report.json AFFNET_DUMP_ROWS=$PWD/gpurun_out/rows
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider -rA > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?"; tail -n 3 gpurun_out/pytest_gpu.log | cut -c1-300
grep -h "FAILED\|Error" gpurun_out/pytest_gpu.log | head -20 | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?"; tail -n 2 gpurun_out/smoke.log
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1; echo "bench exit: $?"; grep '^{' gpurun_out/bench_default.log | cut -c1-600; grep "^real" gpurun_out/bench_default.log
