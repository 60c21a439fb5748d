#!/usr/bin/env python
"""Wall time of each trunk alone on 48000 patches, arith fp32 (exact fp32 MFMA) vs arith fp32_split3 (split operands) (min of 5 launches each)."""
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
ctx = engine.utility_ctx(dev)        # the nets' stand-alone calls (arith "fp32") run on this context: switch IT for the A/B
for split in (0, 1, 3, 1, 3):          # 0 exact, 1 split operands, 3 split with the variant bits of AFFNET_S3_VARIANT (default 1: alternating wave priorities in the HardNet loops)
    lib.affnet_set_arith(ctx, 1 if split else 0)
    lib.affnet_debug_split3_variant(ctx, int(os.environ.get("AFFNET_S3_VARIANT", "1")) if split == 3 else 0)
    row = []
    for nm, net in (("AffNet", A), ("OriNet", O), ("HardNet", H)):
        net(big); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); net(big); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        row.append("%s %.3f ms" % (nm, best))
    print({0: "exact        ", 1: "split3       ", 3: "split3 variant"}[split], " | ".join(row))
lib.affnet_set_arith(ctx, 0); lib.affnet_debug_split3_variant(ctx, 0)
