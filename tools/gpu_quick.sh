#!/bin/bash
# Short iteration loop: a subset of the GPU parity tests (-k "$1") and a short bench line with the secondary rooflines.
mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "$1" > gpurun_out/pytest_quick.log 2>&1; echo "pytest exit: $?"; tail -n 4 gpurun_out/pytest_quick.log | cut -c1-300
grep -n "^E " gpurun_out/pytest_quick.log | head -8 | cut -c1-300
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline $2 > gpurun_out/bench_quick.log 2>&1; echo "bench exit: $?"
python - <<'PY'
import json
for l in open("gpurun_out/bench_quick.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("%.0f kp/s" % d["value"], d["stage_ms_per_image"], "trunk %.1f TF" % d["roofline"]["achieved"])
        for s in d.get("secondary_rooflines", []):
            print("   %-70s %.4f ms/img  %.0f GB/s" % (s["kernel"][:70], s["ms_per_image"], s["achieved"]))
PY
