#!/bin/bash
# Shader clock and socket power while ONLY the pyramid (blur kernels) runs in a loop at 4K: is the blur power limited?
export TMPDIR=/tmp
( while true; do rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n'; echo; sleep 0.2; done ) > /tmp/pyr_power.jsonl &
W=$!
PYR_LOOPS=${PYR_LOOPS:-40} timeout 120 python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import affnet_amd
from affnet_amd import engine
from affnet_amd._lib import lib, ptr, check
dev = torch.device("cuda", 0)
x = (torch.rand(8, 1, 2160, 3840) * 255.0).to(dev)
det = affnet_amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=2000, border=5, num_Baum_iters=0).to(dev)
ctx = det._context(x, allow_batch=True)
st = engine.stream_of(dev)
for _ in range(3):
    check(lib.affnet_pyramid_build(ctx.handle, ptr(x), st), ctx.handle, "pyramid_build")
torch.cuda.synchronize()
t0 = time.time()
for rep in range(int(os.environ.get("PYR_LOOPS", "40"))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        check(lib.affnet_pyramid_build(ctx.handle, ptr(x), st), ctx.handle, "pyramid_build")
    e1.record(); torch.cuda.synchronize()
    if rep % 8 == 0:
        print("t = %.1f s: pyramid 3840x2160 x 8: %.4f ms per image" % (time.time() - t0, e0.elapsed_time(e1) / 100 / 8))
PY
kill $W; wait $W 2>/dev/null
python - <<'PY'
import json, re, statistics
sc, pw = [], []
for ln in open("/tmp/pyr_power.jsonl"):
    try: d = json.loads(ln)
    except Exception: continue
    c = d.get("card0", {})
    s = [v for k, v in c.items() if k.startswith("sclk")]
    p = [v for k, v in c.items() if "ower" in k and "(W)" in k]
    if s and p:
        m = re.search(r"(\d+)\s*Mhz", s[0], re.I)
        try: sc.append(int(m.group(1))); pw.append(float(p[0]))
        except Exception: pass
busy = [i for i, p in enumerate(pw) if p > 0.6 * max(pw)]
print("pyramid loop: %d samples (%d busy): sclk median %d MHz (min %d, max %d), power median %.0f W (max %.0f W)" % (
    len(pw), len(busy), statistics.median(sc[i] for i in busy), min(sc[i] for i in busy), max(sc[i] for i in busy),
    statistics.median(pw[i] for i in busy), max(pw)))
PY
