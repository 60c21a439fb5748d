#!/bin/bash
# Copies the judged summaries of a tools/gpu_full.sh run from gpurun_out/ (scratch) into profiles/ (tracked).  Usage: collect_profiles.sh r02_s1
T=${1:?tag, e.g. r02_s1}
cp gpurun_out/parity_report.json profiles/${T}_parity_report.json
grep '^{' gpurun_out/bench_default.log > profiles/${T}_bench_default.json
grep '^{' gpurun_out/bench_config2.log > profiles/${T}_bench_config2_latency.json
grep '^{' gpurun_out/bench_h2d.log > profiles/${T}_bench_include_h2d.json
for N in 2 4; do grep '^{' gpurun_out/bench_dist_dryrun_$N.log > profiles/${T}_bench_gpus${N}_onedevice_gloo_verify_gather.json; done
grep '^{' gpurun_out/bench_self_gather.log > profiles/${T}_bench_self_gather_rccl_1rank_verify_gather.json
python tools/trim_trace.py gpurun_out/prof_c2/run_kernel_trace.csv profiles/${T}_config2_kernel_trace.csv
cp gpurun_out/prof_c2/run_kernel_stats.csv profiles/${T}_config2_kernel_stats.csv
cp gpurun_out/gap_table.md profiles/${T}_config2_gap_table.md
cp gpurun_out/prof_c5/run_kernel_stats.csv profiles/${T}_config5_kernel_stats.csv
tail -n 1 gpurun_out/bench_gpus2_refused.log > profiles/${T}_bench_gpus2_refused_on_1gpu_box.txt
cp gpurun_out/prof/run_kernel_stats.csv profiles/${T}_kernel_stats.csv
grep '^{' gpurun_out/bench_onepass.log > profiles/${T}_bench_onepass.json
grep '^{' gpurun_out/bench_config5.log > profiles/${T}_bench_config5.json
cp gpurun_out/fetch_calibration.json profiles/${T}_fetch_calibration.json
cp gpurun_out/pmc_summary.txt profiles/${T}_pmc_summary.txt
sed "s#gpurun_out/fetch_calibration.json#profiles/${T}_fetch_calibration.json#" gpurun_out/traffic.json > profiles/${T}_traffic.json
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -n 1 > profiles/${T}_pytest_gpu_summary.txt
grep -E "^(PASSED|FAILED)" gpurun_out/pytest_gpu.log >> profiles/${T}_pytest_gpu_summary.txt
tail -n 1 gpurun_out/smoke.log >> profiles/${T}_pytest_gpu_summary.txt
# arith fp32_split3 (the other arithmetic mode of the boundary): its own bench lines, kernel stats, PMC summary, timings
grep '^{' gpurun_out/bench_split3.log > profiles/${T}_split3_bench.json
for C in config2 config5 onepass; do [ -f gpurun_out/bench_${C}_split3.log ] && grep '^{' gpurun_out/bench_${C}_split3.log > profiles/${T}_split3_bench_${C}.json; done
cp gpurun_out/prof_split3/run_kernel_stats.csv profiles/${T}_split3_kernel_stats.csv
(cat gpurun_out/split3_phase_timing.txt; echo; cat gpurun_out/split3_net_timing.txt) | grep -v amdgpu.ids > profiles/${T}_split3_phase_and_net_timing.txt
[ -f gpurun_out/pmc_split3_summary.txt ] && cp gpurun_out/pmc_split3_summary.txt profiles/${T}_split3_pmc_summary.txt
# arith fp32_split2h
grep '^{' gpurun_out/bench_split2h.log > profiles/${T}_split2h_bench.json
for C in config2 config5 onepass; do [ -f gpurun_out/bench_${C}_split2h.log ] && grep '^{' gpurun_out/bench_${C}_split2h.log > profiles/${T}_split2h_bench_${C}.json; done
[ -f gpurun_out/prof_split2h/run_kernel_stats.csv ] && cp gpurun_out/prof_split2h/run_kernel_stats.csv profiles/${T}_split2h_kernel_stats.csv
[ -f gpurun_out/pmc_split2h_summary.txt ] && cp gpurun_out/pmc_split2h_summary.txt profiles/${T}_split2h_pmc_summary.txt
[ -f gpurun_out/f16_split_probe.txt ] && cp gpurun_out/f16_split_probe.txt profiles/${T}_f16_split_probe.txt
[ -f gpurun_out/s3_loop_probe_h.txt ] && cp gpurun_out/s3_loop_probe_h.txt profiles/${T}_s3_loop_probe_split2h.txt
[ -f gpurun_out/s3_loop_probe.txt ] && cp gpurun_out/s3_loop_probe.txt profiles/${T}_s3_loop_probe.txt
grep -v "^ *value" gpurun_out/clock_watch.txt > profiles/${T}_clock_power_exact_vs_split.txt; grep "value" gpurun_out/clock_watch.txt >> profiles/${T}_clock_power_exact_vs_split.txt
# BASELINE configs[4] counters
[ -f gpurun_out/config5_traffic.json ] && sed "s#gpurun_out/fetch_calibration.json#profiles/${T}_fetch_calibration.json#" gpurun_out/config5_traffic.json > profiles/${T}_config5_traffic.json
[ -f gpurun_out/pmc_config5_summary.txt ] && cp gpurun_out/pmc_config5_summary.txt profiles/${T}_config5_pmc_summary.txt
[ -f gpurun_out/cpu_conv_order_gpubox.txt ] && cp gpurun_out/cpu_conv_order_gpubox.txt profiles/${T}_cpu_conv_order_gpubox.txt
[ -f gpurun_out/ab_variant.txt ] && cp gpurun_out/ab_variant.txt profiles/${T}_ab_variant.txt
[ -f gpurun_out/bench_self_gather_all.log ] && grep '^{' gpurun_out/bench_self_gather_all.log > profiles/${T}_bench_self_gather_rccl_1rank_all_gather_verify_gather.json
# the same GPU rows against the reference on THIS host (the authoring container = the host of tests/golden): needs the oracle, runs here
[ -d gpurun_out/rows ] && python tests/offline_parity_account.py gpurun_out/rows profiles/${T}_offline_parity_account_authoring_host.json | tail -n 1
python tools/kernel_resources.py > profiles/${T}_kernel_resources.md
python tools/roofline_table.py profiles/${T} > profiles/${T}_roofline_table.md
ls -la profiles/${T}_*
