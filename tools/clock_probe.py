#!/usr/bin/env python
"""Shader clock / power while one MFMA-loop probe variant runs back to back (is the loop clock-limited or issue-limited?).
Usage: clock_probe.py LAYER PROBE [SECONDS [BLOCKS]]   LAYER 1 / 5 = HardNet conv1 / conv5, 13 / 15 = AffNet conv3 / conv5"""
import os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import os as _os
_os.environ.setdefault("AFFNET_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "affnet_amd", "libaffnet_hip_probes.so"))   # probe kernels live there (include/affnet_hip_probes.h)
import affnet_amd
from affnet_amd._lib import lib, ptr
layer, probe = int(sys.argv[1]), int(sys.argv[2])
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
dev = torch.device("cuda:0")
if layer >= 10:
    H = affnet_amd.AffNetFast(); H.load_state_dict(torch.load(os.path.join(ROOT, "pretrained/AffNet.pth"), map_location="cpu", weights_only=False)["state_dict"]); H.to(dev)
else:
    H = affnet_amd.HardNet(); H.load_state_dict(affnet_amd.synthetic_hardnet_state(0)); H.to(dev)
pk = H.packed_weights(dev)
out = torch.zeros(2, device=dev)
reps, blocks = 40, int(sys.argv[4]) if len(sys.argv) > 4 else 256 * 4
mfmas = 2304 if layer >= 10 else 9216
lib.affnet_cnn32_probe(ptr(pk), layer, probe, 2, blocks, ptr(out), None); torch.cuda.synchronize()
t0 = time.time(); n = 0; smi = None; ms = []
while time.time() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.affnet_cnn32_probe(ptr(pk), layer, probe, reps, blocks, ptr(out), None)
    e1.record()
    if smi is None and time.time() - t0 > secs / 2:
        smi = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1) / 20)
tf = [blocks * reps * mfmas * 2048.0 / (m * 1e-3) / 1e12 for m in ms]
print("layer %d probe %d blocks %d: first %.1f TF, last %.1f TF, min %.1f max %.1f" % (layer, probe, blocks, tf[0], tf[-1], min(tf), max(tf)))
for line in (smi or "").splitlines():
    if "sclk" in line or "Power" in line or "fclk" in line or "mclk" in line:
        print("   ", line.strip())
