#!/bin/bash
# PMC counters (separate passes; --kernel-trace only, no other trace domains) for one short batched bench run,
# then the per-phase s_memtime breakdown of the CNN kernels.  Outputs under gpurun_out/pmcN/.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
CMD="python bench.py --steps 1 --warmup 1 --batch 16 --chunk 16 --no-cpu-baseline"
i=1
for SET in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d gpurun_out/pmc$i -o run -- $CMD > gpurun_out/pmc$i.log 2>&1; echo "pmc$i exit $?"
  i=$((i+1))
done
python tools/pmc_summary.py gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3 gpurun_out/pmc4 > gpurun_out/pmc_summary.txt 2>&1
cat gpurun_out/pmc_summary.txt | head -60
python tools/cnn_phase_timing.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/cnn_phase_timing.txt
