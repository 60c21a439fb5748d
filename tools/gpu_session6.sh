#!/bin/bash
# batched path: tests + smoke + chunk/stream sweep + rocprof stats.  Outputs under gpurun_out/.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?"; tail -n 15 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?"; tail -n 2 gpurun_out/smoke.log
run() { # name, args
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $2 > gpurun_out/bench_$1.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/bench_$1.log") if x.startswith("{")]
if not l: print("$1: FAILED"); print(open("gpurun_out/bench_$1.log").read()[-1500:])
else:
    d=json.loads(l[-1]); print("$1: %.0f kp/s, %.3f ms/img" % (d["value"], d["ms_per_image"]), d["stage_ms_per_image"], "trunk TF %.1f" % d["roofline"]["achieved"])
PY
}
run c16s1 "--chunk 16 --streams 1"
run c16s2 "--chunk 16 --streams 2"
run c8s2 "--chunk 8 --streams 2"
run c32s1 "--chunk 32 --streams 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o run -- python bench.py --steps 1 --warmup 1 --batch 32 --chunk 16 --no-cpu-baseline > gpurun_out/prof.log 2>&1; echo "prof exit: $?"
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 30 "$f"
