import os, sys, torch
sys.path.insert(0, os.getcwd())
import affnet_amd
from affnet_amd._lib import lib
from affnet_amd import engine
dev = torch.device("cuda:0")
H = affnet_amd.HardNet(); H.load_state_dict(affnet_amd.synthetic_hardnet_state(0)); H.to(dev)
big = (torch.rand(48000, 1, 32, 32) * 255).to(dev)
H.arith = "fp32_split2h"               # the net's stand-alone calls then run on the (device, split2h) utility context - nothing shared is switched
ctx = engine.utility_ctx(dev, 2)
for rep in range(3):
    for v in (0, 2):
        lib.affnet_debug_split3_variant(ctx, v)
        H(big); torch.cuda.synchronize()
        best = 1e9
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); H(big); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print("variant", v, "(2 = no NaN->inf step)", "HardNet split2h 48000 patches: %.3f ms" % best)
lib.affnet_debug_split3_variant(ctx, 0)
