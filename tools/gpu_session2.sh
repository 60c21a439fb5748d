#!/bin/bash
# streams sweep + PMC counters for the CNN / dense kernels.  Outputs under gpurun_out/.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?"; tail -n 3 gpurun_out/pytest_gpu.log
for S in 1 2 4; do
  timeout 300 python bench.py --steps 2 --warmup 1 --batch 32 --streams $S --no-cpu-baseline > gpurun_out/bench_s$S.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/bench_s$S.log") if x.startswith("{")]
d=json.loads(l[-1]); print("streams $S: %.0f kp/s, %.3f ms/img" % (d["value"], d["ms_per_image"]), d["stage_ms_per_image"], "trunk TF %.1f" % d["roofline"]["achieved"])
PY
done
# PMC pass 1: SQ counters (MFMA busy etc.)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc1 -o run -- python bench.py --steps 1 --warmup 0 --batch 4 --no-cpu-baseline > gpurun_out/pmc1.log 2>&1; echo "pmc1 exit $?"
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pmc2 -o run -- python bench.py --steps 1 --warmup 0 --batch 4 --no-cpu-baseline > gpurun_out/pmc2.log 2>&1; echo "pmc2 exit $?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc3 -o run -- python bench.py --steps 1 --warmup 0 --batch 4 --no-cpu-baseline > gpurun_out/pmc3.log 2>&1; echo "pmc3 exit $?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc4 -o run -- python bench.py --steps 1 --warmup 0 --batch 4 --no-cpu-baseline > gpurun_out/pmc4.log 2>&1; echo "pmc4 exit $?"
ls gpurun_out/pmc1 gpurun_out/pmc3 | head
