#!/bin/bash
# detector / pyramid iteration: exactness tests + per-kernel stats
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "blur or pyramid or full_path_synthetic or batched or edge or 4k" > gpurun_out/pytest_det.log 2>&1; echo "pytest exit: $?"; tail -n 3 gpurun_out/pytest_det.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o run -- python bench.py --steps 1 --warmup 1 --batch 32 --chunk 16 --no-cpu-baseline --pipeline 0 > gpurun_out/prof.log 2>&1; echo "prof exit: $?"
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -v "at::\|rocclr\|shape_\|select_emit\|scale_lafs\|apply_rot\|level_sel\|finish" "$f" | head -n 14 | cut -c1-150
