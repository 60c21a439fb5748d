#!/bin/bash
# Round 4, first GPU call: the whole GPU suite (both arithmetic modes), smoke, the default bench line (co-reported arith_fp32_split3,
# other_configs), then diagnostics of the split-operand trunk (per-wave phase stamps, wait-state PMC pass).
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
export AFFNET_PARITY_REPORT=$PWD/gpurun_out/parity_report.json
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?"; tail -n 4 gpurun_out/pytest_gpu.log | cut -c1-400
grep -n "^E " gpurun_out/pytest_gpu.log | head -12 | cut -c1-400
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?"; tail -n 2 gpurun_out/smoke.log
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1; echo "bench exit: $?"; grep '^{' gpurun_out/bench_default.log | cut -c1-300; tail -n 4 gpurun_out/bench_default.log | cut -c1-200
timeout 200 python tools/s3_phase_timing.py > gpurun_out/split3_phase_timing.txt 2>&1; tail -n 24 gpurun_out/split3_phase_timing.txt
timeout 200 python tools/s3_net_timing.py > gpurun_out/split3_net_timing.txt 2>&1; tail -n 5 gpurun_out/split3_net_timing.txt
CMD="python bench.py --arith fp32_split3 --steps 1 --warmup 1 --batch 64 --chunk 32 --no-cpu-baseline --no-secondary --no-other-configs"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmc_s3w -o run -- $CMD > gpurun_out/pmc_s3w.log 2>&1; echo "pmc split3 waits exit $?"
python tools/pmc_biggest.py $(dirname $(find gpurun_out/pmc_s3w -name run_counter_collection.csv | head -1)) 'cnn32_trunk' > gpurun_out/pmc_s3w_summary.txt 2>&1; head -n 8 gpurun_out/pmc_s3w_summary.txt | cut -c1-600
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/pmc_s3i -o run -- $CMD > gpurun_out/pmc_s3i.log 2>&1; echo "pmc split3 insts exit $?"
python tools/pmc_biggest.py $(dirname $(find gpurun_out/pmc_s3i -name run_counter_collection.csv | head -1)) 'cnn32_trunk' > gpurun_out/pmc_s3i_summary.txt 2>&1; head -n 8 gpurun_out/pmc_s3i_summary.txt | cut -c1-600
