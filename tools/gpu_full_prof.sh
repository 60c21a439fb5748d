#!/bin/bash
# Full evidence run, part B: rocprofv3 kernel stats of the bench commands, PMC passes (separate passes, --kernel-trace only) of both
# arithmetic modes and of the 4K configuration, counter calibration, phase / net timings, loop probe, clock / power watch.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
SKIP_PMC=${SKIP_PMC:-0}
COMMON="--steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-other-configs"
CMD="python bench.py $COMMON --batch 64 --chunk 32 --no-split3"
CMD3="python bench.py $COMMON --batch 64 --chunk 32 --arith fp32_split3"
CMD2H="python bench.py $COMMON --batch 64 --chunk 32 --arith fp32_split2h"
CMD5="python bench.py --config5 $COMMON --no-split3"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c2 -o run -- python bench.py --config2 --steps 5 > gpurun_out/prof_c2.log 2>&1; echo "prof config2 exit: $?"
(python tools/gap_table.py gpurun_out/prof_c2/run_kernel_trace.csv; echo; python tools/gap_table.py gpurun_out/prof_c2/run_kernel_trace.csv --graph) > gpurun_out/gap_table.md 2>&1; tail -n 11 gpurun_out/gap_table.md
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -o run -- $CMD5 > gpurun_out/prof_c5.log 2>&1; echo "prof config5 exit: $?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o run -- $CMD > gpurun_out/prof.log 2>&1; echo "prof exit: $?"
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 12 "$f" | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_split3 -o run -- $CMD3 > gpurun_out/prof_split3.log 2>&1; echo "prof split3 exit: $?"
f=$(find gpurun_out/prof_split3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 8 "$f" | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_split2h -o run -- $CMD2H > gpurun_out/prof_split2h.log 2>&1; echo "prof split2h exit: $?"
f=$(find gpurun_out/prof_split2h -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 8 "$f" | cut -c1-200
timeout 300 python tools/s3_phase_timing.py > gpurun_out/split3_phase_timing.txt 2>&1; grep -c "total per patch" gpurun_out/split3_phase_timing.txt
timeout 200 python tools/s3_net_timing.py > gpurun_out/split3_net_timing.txt 2>&1; tail -n 5 gpurun_out/split3_net_timing.txt
timeout 100 tools/probes/s3_loop_probe 20 1024 > gpurun_out/s3_loop_probe.txt 2>&1; echo "loop probe exit $?"
timeout 100 tools/probes/s3_loop_probe 20 1024 h > gpurun_out/s3_loop_probe_h.txt 2>&1; echo "loop probe (two fp16 terms) exit $?"
timeout 100 tools/probes/f16_split_probe 20000 > gpurun_out/f16_split_probe.txt 2>&1; head -n 3 gpurun_out/f16_split_probe.txt | cut -c1-200
bash tools/clock_watch.sh > gpurun_out/clock_watch.txt 2>&1; tail -n 6 gpurun_out/clock_watch.txt
if [ "$SKIP_PMC" != "1" ]; then
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib_f -o run -- python tools/fetch_calib.py run > gpurun_out/calib_f.log 2>&1; echo "calib fetch exit $?"
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/calib_w -o run -- python tools/fetch_calib.py run > gpurun_out/calib_w.log 2>&1; echo "calib write exit $?"
  python tools/fetch_calib.py reduce $(dirname $(find gpurun_out/calib_f -name run_counter_collection.csv | head -1)) $(dirname $(find gpurun_out/calib_w -name run_counter_collection.csv | head -1)) > gpurun_out/fetch_calibration.json 2> gpurun_out/calib_reduce.log; head -c 600 gpurun_out/fetch_calibration.json; echo
  SET1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAVES"
  SET2="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
  SET3="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
  P() { dirname $(find gpurun_out/$1 -name run_counter_collection.csv | head -1); }
  run_pmc() { timeout 400 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d gpurun_out/$1 -o run -- $3 > gpurun_out/$1.log 2>&1; echo "$1 exit $?"; }
  # exact fp32 MFMA path: wait / busy counters, LDS / VALU counters, matrix-pipe busy, FETCH, WRITE
  run_pmc pmc1 "$SET1" "$CMD"; [ "$SKIP_SET2" != "1" ] && run_pmc pmc2 "$SET2" "$CMD"; run_pmc pmc2b "$SET3" "$CMD"; run_pmc pmc3 "FETCH_SIZE" "$CMD"; run_pmc pmc4 "WRITE_SIZE" "$CMD"
  [ "$SKIP_SET2" = "1" ] && { mkdir -p gpurun_out/pmc2; cp -r gpurun_out/pmc2b/* gpurun_out/pmc2/; }
  python tools/pmc_summary.py $(P pmc1) $(P pmc2) $(P pmc2b) $(P pmc3) $(P pmc4) > gpurun_out/pmc_summary.txt 2>&1; head -12 gpurun_out/pmc_summary.txt | cut -c1-300
  python tools/pmc_traffic.py $(P pmc3) $(P pmc4) 32 gpurun_out/fetch_calibration.json > gpurun_out/traffic_exact.json 2> gpurun_out/traffic.log
  # arith fp32_split3: the same passes
  run_pmc pmc_s3_1 "$SET1" "$CMD3"; run_pmc pmc_s3_2 "$SET3" "$CMD3"; run_pmc pmc_s3_3 "FETCH_SIZE" "$CMD3"; run_pmc pmc_s3_4 "WRITE_SIZE" "$CMD3"
  python tools/pmc_summary.py $(P pmc_s3_1) $(P pmc_s3_2) $(P pmc_s3_3) $(P pmc_s3_4) > gpurun_out/pmc_split3_summary.txt 2>&1; head -12 gpurun_out/pmc_split3_summary.txt | cut -c1-300
  python tools/pmc_traffic.py $(P pmc_s3_3) $(P pmc_s3_4) 32 gpurun_out/fetch_calibration.json > gpurun_out/traffic_split3.json 2>> gpurun_out/traffic.log
  # arith fp32_split2h: the same passes
  run_pmc pmc_h2_1 "$SET1" "$CMD2H"; run_pmc pmc_h2_2 "$SET3" "$CMD2H"; run_pmc pmc_h2_3 "FETCH_SIZE" "$CMD2H"; run_pmc pmc_h2_4 "WRITE_SIZE" "$CMD2H"
  python tools/pmc_summary.py $(P pmc_h2_1) $(P pmc_h2_2) $(P pmc_h2_3) $(P pmc_h2_4) > gpurun_out/pmc_split2h_summary.txt 2>&1; head -12 gpurun_out/pmc_split2h_summary.txt | cut -c1-300
  python tools/pmc_traffic.py $(P pmc_h2_3) $(P pmc_h2_4) 32 gpurun_out/fetch_calibration.json > gpurun_out/traffic_split2h.json 2>> gpurun_out/traffic.log
  python - <<'PY'
import json
a = json.load(open("gpurun_out/traffic_exact.json"))
for f in ("gpurun_out/traffic_split3.json", "gpurun_out/traffic_split2h.json"):
    for k, v in json.load(open(f))["kernels"].items():
        if k not in a["kernels"]:                      # the split-operand instantiations (trunks <.., 3> / <.., 2>, head, dense convs)
            a["kernels"][k] = v
a["note"] = "kernels of the exact-fp32 run (bench.py --no-split3) plus the split-operand kernels of the --arith fp32_split3 / fp32_split2h runs; 32 images per launch in all"
json.dump(a, open("gpurun_out/traffic.json", "w"), indent=1)
PY
  # BASELINE configs[4] (4K, 8 images per launch): FETCH / WRITE of the scale-space kernels and the trunks
  run_pmc pmc_c5_f "FETCH_SIZE" "$CMD5"; run_pmc pmc_c5_w "WRITE_SIZE" "$CMD5"; run_pmc pmc_c5_b "$SET3" "$CMD5"
  python tools/pmc_traffic.py $(P pmc_c5_f) $(P pmc_c5_w) 8 gpurun_out/fetch_calibration.json > gpurun_out/config5_traffic.json 2>> gpurun_out/traffic.log
  python tools/pmc_summary.py $(P pmc_c5_b) $(P pmc_c5_f) $(P pmc_c5_w) > gpurun_out/pmc_config5_summary.txt 2>&1; head -8 gpurun_out/pmc_config5_summary.txt | cut -c1-300
fi
