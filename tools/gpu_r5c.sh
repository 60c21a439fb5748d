#!/bin/bash
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
export AFFNET_DUMP_ROWS=$PWD/gpurun_out/rows
timeout 200 python tools/ab_variant.py > gpurun_out/ab_variant.txt 2>&1; cat gpurun_out/ab_variant.txt | grep -v amdgpu.ids | tail -n 12
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1; echo "bench exit: $?"; grep "^real" gpurun_out/bench_default.log
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench_default.log') if x.startswith('{')]
d=json.loads(l[-1]); pc=d['parity_check']
print(d['value'], d['roofline']['frac'], {k:v for k,v in pc.items() if k not in ('unmatched_rows','bar','reference')})
for m in ('arith_fp32_split3','arith_fp32_split2h'):
    print(m, d[m]['value'], {k:v for k,v in d[m]['parity_check'].items() if k in ('pass','matched','unmatched_keys','unmatched_unexplained','laf_max_px','rows_worse_than_cpu_vs_fp64','rows_outside_1e-3_unexplained')})
PY
