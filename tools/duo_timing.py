#!/usr/bin/env python
"""Slot timeline of the anti-phase AffNet kernel (cnn16_duo_kernel): per tick, time each group spends working vs waiting."""
import os, sys
import numpy as np
import torch
os.environ["AFFNET_CNN_DUO_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import affnet_amd
from affnet_amd._lib import lib, ptr
dev = torch.device("cuda:0")
n = 48000
p = (torch.rand(n, 1, 32, 32) * 255).to(dev)
A = affnet_amd.AffNetFast(); A.load_state_dict(torch.load(os.path.join(ROOT, "pretrained/AffNet.pth"), map_location="cpu", weights_only=False)["state_dict"]); A.to(dev)
A(p); torch.cuda.synchronize()
st = torch.zeros(256 * 2 * 256 * 2 + 256 * 2 * 64 * 8, dtype=torch.int64, device=dev)
lib.affnet_cnn32_debug_timing(ptr(st))
A(p); torch.cuda.synchronize()
lib.affnet_cnn32_debug_timing(None)
raw = st.cpu().numpy()
t = raw[:262144].reshape(256, 2, 256, 2).astype(np.float64)
sub = raw[262144:].reshape(256, 2, 64, 8).astype(np.float64)[:, :, 2:60, :7]
d = np.diff(sub, axis=3).mean(axis=(0, 2))
for g in (0, 1):
    print('group %d E* split: c5 store+sync1 %.0f | head loop+sums %.0f | sync2 %.0f | tail math %.0f | prologue (3 syncs) %.0f | conv0 %.0f' % ((g,) + tuple(d[g])))
names = ["E*", "M1", "E1", "M2", "E2", "M3", "E3", "M4", "E4", "M5"]
for g in (0, 1):
    arrive, leave = t[:, g, :, 0], t[:, g, :, 1]
    off = 1 if g == 1 else 0                      # group 1 has one extra initial tick
    work = arrive[:, 1 + off:101 + off] - leave[:, off:100 + off]      # slot s = between tick s-1 release and tick s arrival
    wait = leave[:, 1 + off:101 + off] - arrive[:, 1 + off:101 + off]
    w = work.reshape(256, 10, 10).mean(axis=(0, 1)); ww = wait.reshape(256, 10, 10).mean(axis=(0, 1))
    print("group %d: slot work / wait-at-tick (cycles)" % g)
    print("   " + "  ".join("%s %.0f/%.0f" % (nm, a, b) for nm, a, b in zip(names, np.roll(w, 0), np.roll(ww, 0))))
    print("   per patch: work %.0f  wait %.0f  total %.0f" % (w.sum(), ww.sum(), w.sum() + ww.sum()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); A(p); e1.record(); torch.cuda.synchronize()
print("48000 patches: %.3f ms -> %.1f TFLOP/s" % (e0.elapsed_time(e1), 48000 * 19193856.0 / e0.elapsed_time(e1) / 1e9))
