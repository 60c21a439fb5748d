#!/bin/bash
# Shader clock and socket power while the batched bench runs (exact fp32 path, then the two split-operand arithmetic modes): is the
# split path power limited?  Samples rocm-smi every ~0.25 s in the background; prints the median / max over the samples taken while
# the GPU was busy (> 50 % of the peak power seen).
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
for mode in exact split3 split2h; do
  flag=""; [ $mode = split3 ] && flag="--arith fp32_split3"; [ $mode = split2h ] && flag="--arith fp32_split2h"
  ( while true; do rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done ) > gpurun_out/clock_$mode.jsonl &
  W=$!
  timeout 300 python bench.py $flag --steps 40 --warmup 3 --no-cpu-baseline --no-secondary --no-other-configs --no-split3 > gpurun_out/clock_bench_$mode.log 2>&1
  kill $W; wait $W 2>/dev/null
  python - "$mode" <<'PY'
import json, sys, re, statistics
mode = sys.argv[1]
sc, pw = [], []
for ln in open("gpurun_out/clock_%s.jsonl" % mode):
    try: d = json.loads(ln)
    except Exception: continue
    c = d.get("card0", {})
    s = [v for k, v in c.items() if k.startswith("sclk")]
    p = [v for k, v in c.items() if "ower" in k and "(W)" in k]
    if s and p:
        m = re.search(r"(\d+)\s*Mhz", s[0], re.I)
        try: sc.append(int(m.group(1))); pw.append(float(p[0]))
        except Exception: pass
if pw:
    busy = [i for i, p in enumerate(pw) if p > 0.5 * max(pw)]
    print("%s: %d samples (%d busy): sclk median %d MHz (min %d, max %d), power median %.0f W (max %.0f W)" % (
        mode, len(pw), len(busy), statistics.median(sc[i] for i in busy), min(sc[i] for i in busy), max(sc[i] for i in busy),
        statistics.median(pw[i] for i in busy), max(pw)))
else:
    print(mode, "no samples parsed; first line:", open("gpurun_out/clock_%s.jsonl" % mode).readline()[:300])
PY
  grep '^{' gpurun_out/clock_bench_$mode.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  value', d['value'], d['stage_ms_per_image'])"
done
