#!/bin/bash
# bench variants.  Outputs under gpurun_out/.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
run() { # name, args
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $2 > gpurun_out/bench_$1.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/bench_$1.log") if x.startswith("{")]
if not l: print("$1: FAILED"); print(open("gpurun_out/bench_$1.log").read()[-1500:])
else:
    d=json.loads(l[-1]); print("$1: %.0f kp/s, %.3f ms/img" % (d["value"], d["ms_per_image"]), d["stage_ms_per_image"], "trunk TF %.1f all-CNN TF %.1f" % (d["roofline"]["achieved"], d["roofline"]["all_cnn_tflops"]))
PY
}
run nopipe16 "--chunk 16 --pipeline 0"
run pipe16 "--chunk 16 --pipeline 1"
run nopipe16b "--chunk 16 --pipeline 0"
run pipe16b "--chunk 16 --pipeline 1"
run pipe32 "--chunk 32 --pipeline 1"
run nopipe32 "--chunk 32 --pipeline 0"
run s2 "--chunk 16 --streams 2"
