#!/bin/bash
# Full check: all GPU tests + smoke + default bench (with cpu baseline) + rocprof stats + PMC passes.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?"; tail -n 6 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?"; tail -n 1 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench_default.log 2>&1; echo "bench exit: $?"; grep '^{' gpurun_out/bench_default.log | cut -c1-1500
CMD="python bench.py --steps 1 --warmup 1 --batch 64 --chunk 32 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o run -- $CMD > gpurun_out/prof.log 2>&1; echo "prof exit: $?"
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 12 "$f" | cut -c1-200
i=1
for SET in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d gpurun_out/pmc$i -o run -- $CMD > gpurun_out/pmc$i.log 2>&1; echo "pmc$i exit $?"
  i=$((i+1))
done
python tools/pmc_summary.py gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3 gpurun_out/pmc4 > gpurun_out/pmc_summary.txt 2>&1
head -16 gpurun_out/pmc_summary.txt
