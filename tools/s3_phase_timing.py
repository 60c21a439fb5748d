#!/usr/bin/env python
"""Phase breakdown (s_memtime stamps) of the split-operand trunks (arith fp32_split3 / fp32_split2h) next to the exact fp32 HardNet trunk (tuning aid)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import affnet_amd
from affnet_amd._lib import lib, ptr
from affnet_amd import engine

dev = torch.device("cuda:0")
n, nw = int(os.environ.get("PHASE_PATCHES", "2048")), 8
p = (torch.rand(n, 1, 32, 32) * 255).to(dev)
H = affnet_amd.HardNet(); H.load_state_dict(affnet_amd.synthetic_hardnet_state(0)); H.to(dev)
A = affnet_amd.AffNetFast(); A.load_state_dict(torch.load(os.path.join(ROOT, "pretrained/AffNet.pth"), map_location="cpu", weights_only=False)["state_dict"]); A.to(dev)
O = affnet_amd.OriNetFast(PS=32); O.load_state_dict(torch.load(os.path.join(ROOT, "pretrained/OriNet.pth"), map_location="cpu", weights_only=False)["state_dict"]); O.to(dev)
names = ["input+norm", "conv0", "conv1 mfma", "conv1 store", "conv2 mfma", "conv2 store", "conv3 mfma", "conv3 store",
         "conv4 mfma", "conv4 store", "conv5 mfma"]
SPLIT_MODES = tuple(int(v) for v in os.environ.get("PHASE_ARITH", "1,2").split(","))      # AFFNET_ARITH_* codes of the split modes to stamp
nets = [("HardNet", H, (0,) + SPLIT_MODES)] + ([("AffNet", A, SPLIT_MODES), ("OriNet", O, SPLIT_MODES)] if os.environ.get("PHASE_ALL", "1") == "1" else [])
for nm, net, modes in nets:
  for split in modes:
    net.arith = split                      # stand-alone calls run on the (device, arith) utility context: no shared handle is switched in place
    ctx = engine.utility_ctx(dev, split)
    lib.affnet_debug_split3_variant(ctx, int(os.environ.get("PHASE_VARIANT", "0")) if split else 0)     # bit 0: alternating wave priorities in the HardNet loops
    net(p); torch.cuda.synchronize()
    st = torch.zeros(n * nw * 32, dtype=torch.int64, device=dev)
    lib.affnet_cnn32_debug_timing(ctx, ptr(st))
    net(p); torch.cuda.synchronize()
    lib.affnet_cnn32_debug_timing(ctx, None)
    t = st.cpu().numpy().reshape(n, nw, 32).astype(np.float64)
    d = np.diff(t[:, :, :12], axis=2)
    print("== %s %s: mean ticks per phase per wave (shader cycles)" % (nm, {0: "exact fp32", 1: "split operands (arith fp32_split3)", 2: "split operands (arith fp32_split2h)"}[split]))
    for i in range(11):
        print("  %-12s mean %8.0f  max-over-waves %8.0f" % (names[i], d[:, :, i].mean(), d[:, :, i].max(axis=1).mean()))
    last = 13 if nm != "HardNet" else 11
    if nm != "HardNet":
        print("  %-12s mean %8.0f" % ("head partials", (t[:, :, 13] - t[:, :, 11]).mean()))
    print("  total per patch %.0f ticks" % (t[:, :, last].max(axis=1) - t[:, :, 0].min(axis=1)).mean())
    if split and nm == "HardNet":
        # per wave: which SIMD it ran on (HW_ID bits 5:4 on gfx9) and how long each MFMA loop took
        hw = st.cpu().numpy().reshape(n, nw, 32)[:, :, 14]
        simd = (hw >> 4) & 3
        print("  wave -> SIMD (patch 0): %s ; same mapping in %.0f %% of the patches" % (simd[0].tolist(), 100.0 * (simd == simd[0]).all(axis=1).mean()))
        for i in (2, 4, 6, 8, 10):
            print("  %-12s per wave: %s" % (names[i], " ".join("%6.0f" % d[:, w, i].mean() for w in range(nw))))
    big = (torch.rand(48000, 1, 32, 32) * 255).to(dev)
    net(big); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); net(big); e1.record(); torch.cuda.synchronize()
    print("  48000 patches: %.3f ms" % e0.elapsed_time(e1))
for m in SPLIT_MODES:
    lib.affnet_debug_split3_variant(engine.utility_ctx(dev, m), 0)
