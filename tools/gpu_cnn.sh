#!/bin/bash
# CNN kernel iteration: CNN parity tests + phase timing + bench.  Outputs under gpurun_out/.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "${PYTEST_K:-cnn or mfma or full_path_synthetic or batched}" > gpurun_out/pytest_cnn.log 2>&1; echo "pytest exit: $?"; tail -n 4 gpurun_out/pytest_cnn.log
python tools/cnn_phase_timing.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/cnn_phase_timing.txt
run() { # name, args
  timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline $2 > gpurun_out/bench_$1.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/bench_$1.log") if x.startswith("{")]
if not l: print("$1: FAILED"); print(open("gpurun_out/bench_$1.log").read()[-1500:])
else:
    d=json.loads(l[-1]); print("$1: %.0f kp/s, %.3f ms/img" % (d["value"], d["ms_per_image"]), d["stage_ms_per_image"], "trunk TF %.1f all-CNN TF %.1f" % (d["roofline"]["achieved"], d["roofline"]["all_cnn_tflops"]))
PY
}
run pipe16 "--chunk 16 --pipeline 1"
run nopipe16 "--chunk 16 --pipeline 0"
