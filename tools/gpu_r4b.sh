#!/bin/bash
# Round 4, second GPU call: split-loop probe (old vs new loop on the real layer shapes), the tests that failed on the first bar, bench line.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
export AFFNET_PARITY_REPORT=$PWD/gpurun_out/parity_report_b.json
timeout 300 tools/probes/s3_loop_probe 20 1024 > gpurun_out/s3_loop_probe.txt 2>&1; echo "probe exit $?"; cat gpurun_out/s3_loop_probe.txt | cut -c1-260
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "metric_configuration or odd_sized or graf_img1_n2000 or config5" > gpurun_out/pytest_gpu_b.log 2>&1; echo "pytest exit: $?"; tail -n 4 gpurun_out/pytest_gpu_b.log | cut -c1-400
grep -n "^E " gpurun_out/pytest_gpu_b.log | head -12 | cut -c1-600
( time timeout 900 python bench.py --no-cpu-baseline --no-other-configs ) > gpurun_out/bench_b.log 2>&1; echo "bench exit: $?"
grep '^{' gpurun_out/bench_b.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('value',d['value'], d['stage_ms_per_image'])
a=d['arith_fp32_split3']; print('split', a.get('value'), a.get('stage_ms_per_image'), a.get('roofline',{}).get('frac'), a.get('error'))
"
