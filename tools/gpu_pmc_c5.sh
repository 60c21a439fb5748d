#!/bin/bash
# PMC passes (separate; --kernel-trace only) of the 4K configuration for the scale-space kernels.  Output: gpurun_out/pmc_c5_N/
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
CMD="python bench.py --config5 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary"
i=1
for SET in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  rm -rf gpurun_out/pmc_c5_$i
  timeout 400 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d gpurun_out/pmc_c5_$i -o run -- $CMD > gpurun_out/pmc_c5_$i.log 2>&1; echo "pmc$i exit $?"
  i=$((i+1))
done
python tools/pmc_biggest.py gpurun_out/pmc_c5_1 gpurun_out/pmc_c5_2 gpurun_out/pmc_c5_3 "blur2d|hessian|level_resolve|select_rank"
