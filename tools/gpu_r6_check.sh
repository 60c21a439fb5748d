#!/bin/bash
# Round 6 iteration loop: GPU tests (all, or -k "$1"), then the DRIVER'S bench command with stdout / stderr kept apart: the last stdout line must be
# the compact (< 4 KB) JSON line, the full record goes to gpurun_out/bench_detail.json.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
export AFFNET_PARITY_REPORT=$PWD/gpurun_out/parity_report.json
if [ -n "$1" ]; then K=(-k "$1"); else K=(); fi
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider "${K[@]}" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?"; tail -n 5 gpurun_out/pytest_gpu.log | cut -c1-300
grep -n "^E " gpurun_out/pytest_gpu.log | head -20 | cut -c1-400
grep -n "^th mode\|vs golden" gpurun_out/pytest_gpu.log | cut -c1-400 | head -40
if [ "$SKIP_BENCH" != "1" ]; then
t0=$(date +%s)
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.out 2> gpurun_out/bench_driver.err; echo "bench exit: $? in $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
last = open("gpurun_out/bench_driver.out").read().rstrip("\n").splitlines()[-1]
print("last stdout line: %d bytes" % len(last))
d = json.loads(last)
print(last)
PY
fi
