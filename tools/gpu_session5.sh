#!/bin/bash
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?"; tail -n 3 gpurun_out/pytest_gpu.log
python tools/cnn_phase_timing.py 2>&1 | grep -v amdgpu.ids
run() { # name, env, args
  env $2 timeout 300 python bench.py --steps 2 --warmup 1 --batch 32 --no-cpu-baseline $3 > gpurun_out/bench_$1.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/bench_$1.log") if x.startswith("{")]
if not l: print("$1: FAILED"); print(open("gpurun_out/bench_$1.log").read()[-800:])
else:
    d=json.loads(l[-1]); print("$1: %.0f kp/s, %.3f ms/img" % (d["value"], d["ms_per_image"]), d["stage_ms_per_image"], "trunk TF %.1f" % d["roofline"]["achieved"])
PY
}
run base "A=1" "--pipeline 0"
