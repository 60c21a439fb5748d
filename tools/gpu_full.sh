#!/bin/bash
# Full evidence run on an MI355X box: all GPU tests (with the parity report) + smoke + default bench (cpu baseline, parity_check,
# secondary rooflines) + config2 latency + H2D-inclusive line + 2-rank dry run of the self-spawn path + rocprof stats + counter
# calibration + PMC passes.  Everything lands in gpurun_out/; copy what is to be judged into profiles/ (tools/collect_profiles.sh).
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
export AFFNET_PARITY_REPORT=$PWD/gpurun_out/parity_report.json
SKIP_PMC=${SKIP_PMC:-0}
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -rA > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?"; tail -n 4 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?"; tail -n 1 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; echo "bench exit: $?"; grep '^{' gpurun_out/bench_default.log | cut -c1-600
timeout 300 python bench.py --config2 > gpurun_out/bench_config2.log 2>&1; echo "config2 exit: $?"; grep '^{' gpurun_out/bench_config2.log | cut -c1-400
timeout 300 python bench.py --include-h2d --no-cpu-baseline --no-secondary > gpurun_out/bench_h2d.log 2>&1; echo "h2d exit: $?"; grep '^{' gpurun_out/bench_h2d.log | cut -c1-300
# N > 1 path on one device: self-spawned ranks, RCCL cannot share one GPU between ranks -> gloo for the exchange, real kernels, every
# gathered record re-computed by rank 0 (--verify-gather all); then the 1-rank RCCL flavour, verified too
bash tools/gpu_dist_dryrun.sh
cp gpurun_out/bench_dist_dryrun_2.log gpurun_out/bench_spawn2_onedev.log
timeout 300 python bench.py --onepass > gpurun_out/bench_onepass.log 2>&1; echo "onepass exit: $?"; grep '^{' gpurun_out/bench_onepass.log | cut -c1-300
timeout 600 python bench.py --config5 > gpurun_out/bench_config5.log 2>&1; echo "config5 exit: $?"; grep '^{' gpurun_out/bench_config5.log | cut -c1-300
timeout 120 python bench.py --gpus 2 --steps 1 --warmup 0 > gpurun_out/bench_gpus2_refused.log 2>&1; echo "gpus2 on a 1-GPU box exit (2 = refused loudly): $?"; tail -n 2 gpurun_out/bench_gpus2_refused.log
# kernel traces of the two non-headline configs: B = 1 latency (eager + graph) with its gap table, and 4K
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c2 -o run -- python bench.py --config2 --steps 5 > gpurun_out/prof_c2.log 2>&1; echo "prof config2 exit: $?"
(python tools/gap_table.py gpurun_out/prof_c2/run_kernel_trace.csv; echo; python tools/gap_table.py gpurun_out/prof_c2/run_kernel_trace.csv --graph) > gpurun_out/gap_table.md 2>&1; tail -n 11 gpurun_out/gap_table.md
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -o run -- python bench.py --config5 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/prof_c5.log 2>&1; echo "prof config5 exit: $?"
CMD="python bench.py --steps 1 --warmup 1 --batch 64 --chunk 32 --no-cpu-baseline --no-secondary --no-split3"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o run -- $CMD > gpurun_out/prof.log 2>&1; echo "prof exit: $?"
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 12 "$f" | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_onepass -o run -- python bench.py --onepass --steps 1 --warmup 1 > gpurun_out/prof_onepass.log 2>&1; echo "prof onepass exit: $?"
# EXPLORATORY split-operand path (never `value`): its own bench lines, kernel stats, matrix-pipe PMC pass, phase / per-net timings, the
# MFMA / VALU overlap probe and the clock / power watch
timeout 300 python bench.py --split3 --no-cpu-baseline --no-secondary > gpurun_out/bench_split3.log 2>&1; echo "split3 exit: $?"; grep '^{' gpurun_out/bench_split3.log | cut -c1-300
timeout 300 python bench.py --config2 --split3 > gpurun_out/bench_config2_split3.log 2>&1; echo "config2 split3 exit: $?"; grep '^{' gpurun_out/bench_config2_split3.log | cut -c1-300
timeout 200 python tools/s3_phase_timing.py > gpurun_out/split3_phase_timing.txt 2>&1; tail -n 14 gpurun_out/split3_phase_timing.txt
timeout 200 python tools/s3_net_timing.py > gpurun_out/split3_net_timing.txt 2>&1; tail -n 4 gpurun_out/split3_net_timing.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_split3 -o run -- python bench.py --split3 --steps 1 --warmup 1 --batch 64 --chunk 32 --no-cpu-baseline --no-secondary > gpurun_out/prof_split3.log 2>&1; echo "prof split3 exit: $?"
if [ "$SKIP_PMC" != "1" ]; then
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_split3 -o run -- python bench.py --split3 --steps 1 --warmup 1 --batch 64 --chunk 32 --no-cpu-baseline --no-secondary > gpurun_out/pmc_split3.log 2>&1; echo "pmc split3 exit $?"
  python tools/pmc_biggest.py $(dirname $(find gpurun_out/pmc_split3 -name run_counter_collection.csv | head -1)) 'cnn32_trunk' > gpurun_out/pmc_split3_summary.txt 2>&1; head -n 12 gpurun_out/pmc_split3_summary.txt
fi
(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu_overlap tools/probes/mfma_valu_overlap.hip 2>/dev/null && timeout 120 /tmp/mfma_valu_overlap) > gpurun_out/mfma_valu_overlap.txt 2>&1; tail -n 4 gpurun_out/mfma_valu_overlap.txt
bash tools/clock_watch.sh > gpurun_out/clock_watch.txt 2>&1; cat gpurun_out/clock_watch.txt
if [ "$SKIP_PMC" != "1" ]; then
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib_f -o run -- python tools/fetch_calib.py run > gpurun_out/calib_f.log 2>&1; echo "calib fetch exit $?"
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/calib_w -o run -- python tools/fetch_calib.py run > gpurun_out/calib_w.log 2>&1; echo "calib write exit $?"
  python tools/fetch_calib.py reduce $(dirname $(find gpurun_out/calib_f -name run_counter_collection.csv | head -1)) $(dirname $(find gpurun_out/calib_w -name run_counter_collection.csv | head -1)) > gpurun_out/fetch_calibration.json 2> gpurun_out/calib_reduce.log; head -c 1500 gpurun_out/fetch_calibration.json
  i=1
  for SET in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAVES" \
             "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d gpurun_out/pmc$i -o run -- $CMD > gpurun_out/pmc$i.log 2>&1; echo "pmc$i exit $?"
    i=$((i+1))
  done
  P() { dirname $(find gpurun_out/pmc$1 -name run_counter_collection.csv | head -1); }
  python tools/pmc_summary.py $(P 1) $(P 2) $(P 3) $(P 4) > gpurun_out/pmc_summary.txt 2>&1
  python tools/pmc_traffic.py $(P 3) $(P 4) 32 gpurun_out/fetch_calibration.json > gpurun_out/traffic.json 2> gpurun_out/traffic.log
  head -16 gpurun_out/pmc_summary.txt
  # the stand-alone sampler (secondary_rooflines section of bench.py): kernel stats + FETCH / WRITE passes of a run that includes it
  CMD2="python bench.py --steps 1 --warmup 1 --batch 32 --chunk 32 --no-cpu-baseline --no-split3"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sampler -o run -- $CMD2 > gpurun_out/prof_sampler.log 2>&1; echo "prof sampler exit $?"
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_sampler_f -o run -- $CMD2 > gpurun_out/pmc_sampler_f.log 2>&1; echo "pmc sampler fetch exit $?"
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_sampler_w -o run -- $CMD2 > gpurun_out/pmc_sampler_w.log 2>&1; echo "pmc sampler write exit $?"
  python tools/pmc_traffic.py gpurun_out/pmc_sampler_f gpurun_out/pmc_sampler_w 32 gpurun_out/fetch_calibration.json > gpurun_out/traffic_sampler.json 2>> gpurun_out/traffic.log
  grep -A8 "grid_sample_kernel" gpurun_out/traffic_sampler.json | head -12; grep "grid_sample_kernel" gpurun_out/prof_sampler/run_kernel_stats.csv | cut -c1-200
fi
