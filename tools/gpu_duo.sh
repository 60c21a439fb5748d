#!/bin/bash
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
run() { # name, env, args
  env $2 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline $3 > gpurun_out/bench_$1.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/bench_$1.log") if x.startswith("{")]
if not l: print("$1: FAILED"); print(open("gpurun_out/bench_$1.log").read()[-1500:])
else:
    d=json.loads(l[-1]); print("$1: %.0f kp/s, %.3f ms/img" % (d["value"], d["ms_per_image"]), d["stage_ms_per_image"], "trunk TF %.1f all-CNN TF %.1f" % (d["roofline"]["achieved"], d["roofline"]["all_cnn_tflops"]))
PY
}
run lockstep "AFFNET_CNN_DUO=2" "--pipeline 0"
run solo "AFFNET_CNN_DUO=0" "--pipeline 0"
run anti "AFFNET_CNN_DUO=1" "--pipeline 0"
AFFNET_CNN_DUO=2 timeout 300 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "cnn or full_path_synthetic or batched" 2>&1 | tail -2
