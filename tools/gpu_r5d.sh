#!/bin/bash
# round 5 evidence, part B (+ a re-run of the default bench line with the referee's thread cap): rocprofv3 kernel stats, PMC passes, calibration, probes
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
rm -rf gpurun_out/prof* gpurun_out/pmc* gpurun_out/calib_*
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1; echo "bench exit: $?"; grep "^real" gpurun_out/bench_default.log; grep '^{' gpurun_out/bench_default.log | cut -c1-200
bash tools/gpu_full_prof.sh
