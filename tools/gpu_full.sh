#!/bin/bash
# Full evidence run on an MI355X box, part A: all GPU tests (with the parity report) + smoke + the default bench line (cpu baseline,
# parity_check, secondary rooflines, arith_fp32_split3 / arith_fp32_split2h co-reports, other_configs) + the other modes, each in every arithmetic mode where it
# applies + the N-rank path on one device.  Part B (tools/gpu_full_prof.sh): rocprofv3 kernel stats, PMC passes, counter calibration.
# Everything lands in gpurun_out/; copy what is to be judged into profiles/ with tools/collect_profiles.sh <tag>.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
export AFFNET_PARITY_REPORT=$PWD/gpurun_out/parity_report.json AFFNET_DUMP_ROWS=$PWD/gpurun_out/rows
rm -rf gpurun_out/rows
# which fp32 summation orders does THIS host's torch CPU use (the live oracle's host: reference-side host variation)
lscpu | grep "Model name" > gpurun_out/cpu_conv_order_gpubox.txt; timeout 300 python tools/probes/cpu_conv_order.py 16 >> gpurun_out/cpu_conv_order_gpubox.txt 2>&1; tail -n 3 gpurun_out/cpu_conv_order_gpubox.txt | cut -c1-250
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -rA > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?"; tail -n 3 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?"; tail -n 2 gpurun_out/smoke.log
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1; echo "bench exit: $?"; grep '^{' gpurun_out/bench_default.log | cut -c1-400; grep "^real" gpurun_out/bench_default.log
for A in fp32 fp32_split3 fp32_split2h; do
  S=""; [ "$A" = "fp32_split3" ] && S="_split3"; [ "$A" = "fp32_split2h" ] && S="_split2h"
  timeout 300 python bench.py --config2 --arith $A > gpurun_out/bench_config2$S.log 2>&1; echo "config2 $A exit: $?"; grep '^{' gpurun_out/bench_config2$S.log | cut -c1-300
  timeout 600 python bench.py --config5 --arith $A --no-cpu-baseline > gpurun_out/bench_config5$S.log 2>&1; echo "config5 $A exit: $?"; grep '^{' gpurun_out/bench_config5$S.log | cut -c1-300
  timeout 300 python bench.py --onepass --arith $A --no-cpu-baseline > gpurun_out/bench_onepass$S.log 2>&1; echo "onepass $A exit: $?"; grep '^{' gpurun_out/bench_onepass$S.log | cut -c1-300
done
timeout 300 python bench.py --arith fp32_split3 --no-cpu-baseline --no-secondary --no-other-configs > gpurun_out/bench_split3.log 2>&1; echo "split3 exit: $?"; grep '^{' gpurun_out/bench_split3.log | cut -c1-300
timeout 300 python bench.py --arith fp32_split2h --no-cpu-baseline --no-secondary --no-other-configs > gpurun_out/bench_split2h.log 2>&1; echo "split2h exit: $?"; grep '^{' gpurun_out/bench_split2h.log | cut -c1-300
timeout 300 python bench.py --include-h2d --no-cpu-baseline --no-secondary --no-other-configs --no-split3 > gpurun_out/bench_h2d.log 2>&1; echo "h2d exit: $?"; grep '^{' gpurun_out/bench_h2d.log | cut -c1-300
# N > 1 path on one device: self-spawned ranks, RCCL cannot share one GPU between ranks -> gloo for the exchange, real kernels, every
# gathered record re-computed by rank 0 (--verify-gather all); then the 1-rank RCCL flavour, verified too
bash tools/gpu_dist_dryrun.sh
timeout 120 python bench.py --gpus 2 --steps 1 --warmup 0 > gpurun_out/bench_gpus2_refused.log 2>&1; echo "gpus2 on a 1-GPU box exit (2 = refused loudly): $?"; tail -n 2 gpurun_out/bench_gpus2_refused.log
timeout 200 python tools/ab_variant.py 2>&1 | grep -v amdgpu.ids > gpurun_out/ab_variant.txt; tail -n 4 gpurun_out/ab_variant.txt
