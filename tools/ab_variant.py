#!/usr/bin/env python
"""A/B of the split-operand trunks' variant bits (affnet_debug_split3_variant; bit 0 = the two waves of a SIMD alternate at the higher issue
priority inside the HardNet loops) in both split arithmetic modes: min of 7 launches on 48000 patches each, three rounds."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import affnet_amd
from affnet_amd._lib import lib
from affnet_amd import engine
dev = torch.device("cuda:0")
H = affnet_amd.HardNet(); H.load_state_dict(affnet_amd.synthetic_hardnet_state(0)); H.to(dev)
big = (torch.rand(48000, 1, 32, 32) * 255).to(dev)
for mode in (1, 2):
    H.arith = mode
    ctx = engine.utility_ctx(dev, mode)
    for rep in range(3):
        for v in (0, 1):
            lib.affnet_debug_split3_variant(ctx, v)
            H(big); torch.cuda.synchronize()
            best = 1e9
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); H(big); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            print("arith %s variant %d (1 = alternating wave priorities): HardNet 48000 patches %.3f ms" % ({1: "fp32_split3", 2: "fp32_split2h"}[mode], v, best))
    lib.affnet_debug_split3_variant(ctx, 0)
