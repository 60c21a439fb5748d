#!/bin/bash
# OnePassSIR path: parity tests, a throughput line and kernel stats.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
export AFFNET_PARITY_REPORT=$PWD/gpurun_out/parity_report_onepass.json
timeout 900 python -m pytest tests/test_gpu_onepass.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -rA -k "onepass or local_norm or dense_affnet or nms2d or pyramid_variants" > gpurun_out/pytest_onepass.log 2>&1; echo "pytest exit: $?"; tail -n 12 gpurun_out/pytest_onepass.log | cut -c1-250
timeout 300 python bench.py --onepass --steps 3 --warmup 2 > gpurun_out/bench_onepass.log 2>&1; echo "bench onepass exit: $?"; grep '^{' gpurun_out/bench_onepass.log | cut -c1-900
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_onepass -o run -- python bench.py --onepass --steps 1 --warmup 1 > gpurun_out/prof_onepass.log 2>&1; echo "prof exit: $?"
f=$(find gpurun_out/prof_onepass -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 22 "$f" | cut -c1-180
