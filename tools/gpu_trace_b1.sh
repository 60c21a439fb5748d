#!/bin/bash
# Kernel traces of the two non-headline BASELINE configs: configs[1] (--config2: single image, B = 1, eager + HIP graph) and
# configs[4] (--config5: 4K, 8000 kp).  Output: gpurun_out/prof_c2, gpurun_out/prof_c5 (rocprofv3 kernel trace + stats), plus the
# unprofiled bench lines.  tools/gap_table.py turns the B = 1 trace into the per-stage gap table.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 300 python bench.py --config2 > gpurun_out/bench_config2.log 2>&1; echo "config2 exit: $?"; grep '^{' gpurun_out/bench_config2.log | cut -c1-1500
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c2 -o run -- python bench.py --config2 --steps 5 > gpurun_out/prof_c2.log 2>&1; echo "prof c2 exit: $?"
if [ "$SKIP_C5" != "1" ]; then
timeout 400 python bench.py --config5 --no-cpu-baseline --no-secondary > gpurun_out/bench_config5.log 2>&1; echo "config5 exit: $?"; grep '^{' gpurun_out/bench_config5.log | cut -c1-1500
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -o run -- python bench.py --config5 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/prof_c5.log 2>&1; echo "prof c5 exit: $?"
fi
ls -la gpurun_out/prof_c2 gpurun_out/prof_c5 2>/dev/null | head -20
