#!/usr/bin/env python
"""Wall time of each trunk alone on 48000 patches: arith fp32 (exact fp32 MFMA), fp32_split3 and fp32_split2h (split operands); min of 5 launches each."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import affnet_amd
from affnet_amd._lib import lib
from affnet_amd import engine

dev = torch.device("cuda:0")
A = affnet_amd.AffNetFast(); A.load_state_dict(torch.load(os.path.join(ROOT, "pretrained/AffNet.pth"), map_location="cpu", weights_only=False)["state_dict"]); A.to(dev)
O = affnet_amd.OriNetFast(PS=32); O.load_state_dict(torch.load(os.path.join(ROOT, "pretrained/OriNet.pth"), map_location="cpu", weights_only=False)["state_dict"]); O.to(dev)
H = affnet_amd.HardNet(); H.load_state_dict(affnet_amd.synthetic_hardnet_state(0)); H.to(dev)
big = (torch.rand(48000, 1, 32, 32) * 255).to(dev)
for arith in (0, 1, 2, 1, 2):          # AFFNET_ARITH_*: 0 exact, 1 three bf16 terms, 2 two fp16 terms
    row = []
    for nm, net in (("AffNet", A), ("OriNet", O), ("HardNet", H)):
        net.arith = arith              # stand-alone calls run on the (device, arith) utility context: no shared handle is switched in place
        net(big); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); net(big); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        row.append("%s %.3f ms" % (nm, best))
    print({0: "exact        ", 1: "fp32_split3  ", 2: "fp32_split2h "}[arith], " | ".join(row))
