#!/bin/bash
# round 5, second GPU call: CPU conv-order probe on the GPU box's host, the GPU tests touched by the centroid-order change + the C host
# program, parity report, rows dump.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
lscpu | grep "Model name" > gpurun_out/cpu_conv_order_gpubox.txt
timeout 300 python tools/probes/cpu_conv_order.py 16 >> gpurun_out/cpu_conv_order_gpubox.txt 2>&1; tail -n 12 gpurun_out/cpu_conv_order_gpubox.txt | cut -c1-250
export AFFNET_PARITY_REPORT=$PWD/gpurun_out/parity_report.json AFFNET_DUMP_ROWS=$PWD/gpurun_out/rows
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider -rA > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?"; tail -n 3 gpurun_out/pytest_gpu.log | cut -c1-300
grep -h "^FAILED\|^ERROR" gpurun_out/pytest_gpu.log | head -20 | cut -c1-300
