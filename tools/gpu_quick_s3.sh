#!/bin/bash
# Quick check of the split-operand trunks (EXTRA_K adds a pytest -k expression): tests of both arithmetic modes at small sizes, net / phase timings, short bench
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
export AFFNET_PARITY_REPORT=$PWD/gpurun_out/parity_report_c.json
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "split3 or jit or synthetic_golden or graf_img1_golden or batched_launches or graph_replay or trunk_layer or cnn_outputs or ${EXTRA_K:-zzzz}" > gpurun_out/pytest_gpu_c.log 2>&1; echo "pytest exit: $?"; tail -n 3 gpurun_out/pytest_gpu_c.log | cut -c1-300
grep -n "^E " gpurun_out/pytest_gpu_c.log | head -12 | cut -c1-400
timeout 200 python tools/s3_net_timing.py > gpurun_out/split3_net_timing.txt 2>&1; tail -n 5 gpurun_out/split3_net_timing.txt
timeout 200 python tools/s3_phase_timing.py > gpurun_out/split3_phase_timing.txt 2>&1; tail -n 22 gpurun_out/split3_phase_timing.txt
( time timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-secondary ) > gpurun_out/bench_c.log 2>&1; echo "bench exit: $?"
grep '^{' gpurun_out/bench_c.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('value',d['value'], d['stage_ms_per_image'])
a=d['arith_fp32_split3']; print('split', a.get('value'), a.get('stage_ms_per_image'), a.get('roofline',{}).get('frac'), a.get('error'))
"
