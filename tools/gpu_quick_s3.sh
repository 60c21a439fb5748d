#!/bin/bash
# Quick check of the split-operand trunks (EXTRA_K adds a pytest -k expression): tests of both arithmetic modes at small sizes, net / phase timings, short bench
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 120 tools/probes/f16_split_probe 20000 > gpurun_out/f16_split_probe.txt 2>&1; head -n 8 gpurun_out/f16_split_probe.txt | cut -c1-200
[ -n "$LOOP_PROBE" ] && { timeout 300 tools/probes/s3_loop_probe 20 1024 $LOOP_PROBE > gpurun_out/s3_loop_probe_$LOOP_PROBE.txt 2>&1; cut -c1-230 gpurun_out/s3_loop_probe_$LOOP_PROBE.txt; }
export AFFNET_PARITY_REPORT=$PWD/gpurun_out/parity_report_c.json
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "split or jit or synthetic_golden or graf_img1_golden or batched_launches or graph_replay or trunk_layer or cnn_outputs or ${EXTRA_K:-zzzz}" > gpurun_out/pytest_gpu_c.log 2>&1; echo "pytest exit: $?"; tail -n 3 gpurun_out/pytest_gpu_c.log | cut -c1-300
grep -n "^E " gpurun_out/pytest_gpu_c.log | head -12 | cut -c1-400
timeout 200 python tools/s3_net_timing.py > gpurun_out/split3_net_timing.txt 2>&1; tail -n 5 gpurun_out/split3_net_timing.txt
timeout 200 python tools/s3_phase_timing.py > gpurun_out/split3_phase_timing.txt 2>&1; tail -n 22 gpurun_out/split3_phase_timing.txt
( time timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-secondary ) > gpurun_out/bench_c.log 2>&1; echo "bench exit: $?"
grep '^{' gpurun_out/bench_c.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('value',d['value'], d['stage_ms_per_image'])
for m in ('arith_fp32_split3', 'arith_fp32_split2h'):
    a=d.get(m, {}); print(m, a.get('value'), a.get('stage_ms_per_image'), a.get('roofline',{}).get('frac'), a.get('error'), (a.get('parity_check') or {}).get('note', '')[:80])
"
