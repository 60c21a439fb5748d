#!/bin/bash
# Round 6 evidence set (ONE per round, after the last code change): GPU tests + smoke, the DRIVER'S bench command (stdout = the compact line, stderr / detail
# file = the full record), the other single-GPU configurations, the N-rank path on one device, rocprofv3 kernel stats of the bench command, the configs[1]
# trace + gap table, and the counter passes behind `roofline.traffic` (separate --pmc passes, --kernel-trace only).  tools/collect_r6.sh <tag> copies the
# judged summaries into profiles/.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
export AFFNET_PARITY_REPORT=$PWD/gpurun_out/parity_report.json
lscpu | grep "Model name" > gpurun_out/host.txt
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -rA > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?"; tail -n 2 gpurun_out/pytest_gpu.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?"; tail -n 2 gpurun_out/smoke.log
bench() { # name, args...: stdout / stderr apart, the detail file next to them
  local n=$1; shift
  ( time timeout 900 python3 bench.py "$@" ) > gpurun_out/$n.out 2> gpurun_out/$n.err; echo "$n exit: $?"; cp gpurun_out/bench_detail.json gpurun_out/$n.detail.json 2>/dev/null
  tail -n 1 gpurun_out/$n.out | cut -c1-330; grep "^real" gpurun_out/$n.err
}
bench bench_driver --gpus 1 --steps 20 --warmup 5
bench bench_config2 --config2
bench bench_config5 --config5 --no-cpu-baseline
bench bench_onepass --onepass --no-cpu-baseline
AFFNET_BENCH_BACKEND=gloo AFFNET_BENCH_ONE_DEVICE=1 bench bench_gpus2_onedevice_gloo --gpus 2 --steps 2 --warmup 1 --batch 8 --chunk 8 --no-secondary --verify-gather all
AFFNET_BENCH_SELF_GATHER=1 bench bench_self_gather_rccl_1rank --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-other-configs --no-split3 --verify-gather all
timeout 120 python bench.py --gpus 2 --steps 1 --warmup 0 > gpurun_out/bench_gpus2_refused.log 2>&1; echo "gpus2 on a 1-GPU box exit (2 = refused loudly): $?"
# profiles
COMMON="--steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-other-configs --no-split3"
CMD="python bench.py $COMMON --batch 64 --chunk 32"
rm -rf gpurun_out/prof gpurun_out/prof_c2 gpurun_out/pmc2b gpurun_out/pmc3 gpurun_out/pmc4
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c2 -o run -- python bench.py --config2 --steps 5 > gpurun_out/prof_c2.log 2>&1; echo "prof config2 exit: $?"
(python tools/gap_table.py gpurun_out/prof_c2/run_kernel_trace.csv; echo; python tools/gap_table.py gpurun_out/prof_c2/run_kernel_trace.csv --graph) > gpurun_out/gap_table.md 2>&1; tail -n 11 gpurun_out/gap_table.md
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o run -- $CMD > gpurun_out/prof.log 2>&1; echo "prof exit: $?"
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 8 "$f" | cut -c1-200
if [ "$SKIP_PMC" != "1" ]; then
  SET3="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
  P() { dirname $(find gpurun_out/$1 -name run_counter_collection.csv | head -1); }
  run_pmc() { timeout 400 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d gpurun_out/$1 -o run -- $3 > gpurun_out/$1.log 2>&1; echo "$1 exit $?"; }
  run_pmc pmc2b "$SET3" "$CMD"; run_pmc pmc3 "FETCH_SIZE" "$CMD"; run_pmc pmc4 "WRITE_SIZE" "$CMD"
  CAL=$(ls profiles/*fetch_calibration.json profiles/archive/*fetch_calibration.json 2>/dev/null | sort | tail -n 1)
  python tools/pmc_traffic.py $(P pmc3) $(P pmc4) 32 $CAL > gpurun_out/traffic.json 2> gpurun_out/traffic.log; head -c 600 gpurun_out/traffic.json; echo
  python tools/pmc_biggest.py $(P pmc2b) 'cnn32_trunk|hardnet_head|blur2d|hessian' > gpurun_out/pmc_matrix_pipe.txt 2>&1; head -n 8 gpurun_out/pmc_matrix_pipe.txt | cut -c1-250
fi
