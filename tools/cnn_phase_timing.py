#!/usr/bin/env python
"""Per-phase cycle breakdown of the fused CNN trunk kernels (s_memtime stamps, tuning aid).
Runs each net on 2048 random patches and prints mean cycles per phase per wave."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import affnet_amd
from affnet_amd._lib import lib, ptr

dev = torch.device("cuda:0")
n = 2048
p = (torch.rand(n, 1, 32, 32) * 255).to(dev)
A = affnet_amd.AffNetFast(); A.load_state_dict(torch.load(os.path.join(ROOT, "pretrained/AffNet.pth"), map_location="cpu", weights_only=False)["state_dict"]); A.to(dev)
H = affnet_amd.HardNet(); H.load_state_dict(affnet_amd.synthetic_hardnet_state(0)); H.to(dev)
names = ["input+norm", "conv0", "conv1 mfma", "conv1 store", "conv2 mfma", "conv2 store", "conv3 mfma", "conv3 store",
         "conv4 mfma", "conv4 store", "conv5 mfma", "conv5 store"]
for net, nm, nw in [(A, "AffNet", 8), (H, "HardNet", int(os.environ.get("AFFNET_HARDNET_WAVES", "8")))]:
    net(p); torch.cuda.synchronize()
    st = torch.zeros(n * nw * 16, dtype=torch.int64, device=dev)
    lib.affnet_cnn32_debug_timing(ptr(st))
    net(p); torch.cuda.synchronize()
    lib.affnet_cnn32_debug_timing(None)
    t = st.cpu().numpy().reshape(n, nw, 16).astype(np.float64)
    nst = 12 if nm == "HardNet" else 13
    d = np.diff(t[:, :, :nst], axis=2)              # cycles per phase per wave
    wg = t[:, :, :nst].max(axis=1) - t[:, :, 0:1].min(axis=1)   # per patch: boundary times relative to WG start
    print("== %s (%d waves): mean phase cycles per wave (s_memtime ticks, 100 MHz const clock?)" % (nm, nw))
    tot = (t[:, :, nst - 1].max(axis=1) - t[:, :, 0].min(axis=1)).mean()
    span = (t[:, :, nst - 1].max() - t[:, :, 0].min())
    print("  kernel span %.0f ticks for %d patches -> %.0f ticks / patch / CU-slot (256 CUs)" % (span, n, span / (n / 256.0)))
    for i in range(nst - 1):
        print("  %-12s mean %9.0f  max-over-waves %9.0f" % (names[i], d[:, :, i].mean(), d[:, :, i].max(axis=1).mean()))
    print("  total per patch (WG) %.0f ticks" % tot)
