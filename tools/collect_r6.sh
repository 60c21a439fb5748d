#!/bin/bash
# Copies the judged summaries of a tools/gpu_r6_evidence.sh run from gpurun_out/ (scratch) into profiles/ (tracked).  Usage: collect_r6.sh r06_s2
T=${1:?tag, e.g. r06_s2}
cp gpurun_out/parity_report.json profiles/${T}_parity_report.json
(grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -n 1; grep -E "^(PASSED|FAILED)" gpurun_out/pytest_gpu.log; tail -n 2 gpurun_out/smoke.log) > profiles/${T}_pytest_gpu_summary.txt
for n in bench_driver bench_config2 bench_config5 bench_onepass bench_gpus2_onedevice_gloo bench_self_gather_rccl_1rank; do
  [ -f gpurun_out/$n.out ] && tail -n 1 gpurun_out/$n.out > profiles/${T}_${n}_line.json
  [ -f gpurun_out/$n.detail.json ] && cp gpurun_out/$n.detail.json profiles/${T}_${n}_detail.json
done
tail -n 1 gpurun_out/bench_gpus2_refused.log > profiles/${T}_bench_gpus2_refused_on_1gpu_box.txt
python tools/trim_trace.py gpurun_out/prof_c2/run_kernel_trace.csv profiles/${T}_config2_kernel_trace.csv
cp gpurun_out/prof_c2/run_kernel_stats.csv profiles/${T}_config2_kernel_stats.csv
cp gpurun_out/gap_table.md profiles/${T}_config2_gap_table.md
cp gpurun_out/prof/run_kernel_stats.csv profiles/${T}_kernel_stats.csv
[ -f gpurun_out/traffic.json ] && cp gpurun_out/traffic.json profiles/${T}_traffic.json
[ -f gpurun_out/pmc_matrix_pipe.txt ] && cp gpurun_out/pmc_matrix_pipe.txt profiles/${T}_pmc_matrix_pipe.txt
cp gpurun_out/host.txt profiles/${T}_host.txt
python tools/kernel_resources.py > profiles/${T}_kernel_resources.md
ls -la profiles/${T}_*
