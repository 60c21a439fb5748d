#!/bin/bash
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
run() { # name, args
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $2 > gpurun_out/bench_$1.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/bench_$1.log") if x.startswith("{")]
if not l: print("$1: FAILED"); print(open("gpurun_out/bench_$1.log").read()[-1500:])
else:
    d=json.loads(l[-1]); print("$1: %.0f kp/s, %.3f ms/img" % (d["value"], d["ms_per_image"]))
PY
}
run warm "--pipeline 0"
for i in 1 2 3; do
run pipe_$i "--pipeline 1"
run nopipe_$i "--pipeline 0"
run pipe32_$i "--pipeline 1 --chunk 32"
done
