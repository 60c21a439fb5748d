#!/bin/bash
# Iteration loop of round 3: all GPU tests (or -k "$1"), the B = 1 latency line, a short default bench line, the B = 1 kernel trace.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
export AFFNET_PARITY_REPORT=$PWD/gpurun_out/parity_report.json
if [ -n "$1" ]; then K=(-k "$1"); else K=(); fi
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x "${K[@]}" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?"; tail -n 5 gpurun_out/pytest_gpu.log | cut -c1-300
grep -n "^E " gpurun_out/pytest_gpu.log | head -12 | cut -c1-300
timeout 300 python bench.py --config2 > gpurun_out/bench_config2.log 2>&1; echo "config2 exit: $?"
python - <<'PY'
import json
for l in open("gpurun_out/bench_config2.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("config2: eager %.3f ms (min %.3f)  graph %s" % (d["value"], d["warm_ms_min"], d["hip_graph"]))
PY
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/bench_quick.log 2>&1; echo "bench exit: $?"
python - <<'PY'
import json
for l in open("gpurun_out/bench_quick.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("%.0f kp/s" % d["value"], d["stage_ms_per_image"], "trunk %.1f TF" % d["roofline"]["achieved"])
PY
if [ "$SKIP_TRACE" != "1" ]; then
rm -rf gpurun_out/prof_c2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c2 -o run -- python bench.py --config2 --steps 5 > gpurun_out/prof_c2.log 2>&1; echo "prof c2 exit: $?"
python tools/gap_table.py gpurun_out/prof_c2/run_kernel_trace.csv --graph > gpurun_out/gap_table_graph.md 2>&1; tail -n 12 gpurun_out/gap_table_graph.md
fi
