#!/usr/bin/env python
"""Exact-fp32 trunks on 48000 patch tensors each (no pyramid sampling): min of 5 launches, ms.  A/B aid: AFFNET_HIP_LIB selects the library."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import affnet_amd
dev = torch.device("cuda:0")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A = affnet_amd.AffNetFast(PS=32); A.load_state_dict(torch.load(os.path.join(root, "pretrained", "AffNet.pth"), map_location="cpu", weights_only=False)["state_dict"]); A.to(dev)
O = affnet_amd.OriNetFast(PS=32); O.load_state_dict(torch.load(os.path.join(root, "pretrained", "OriNet.pth"), map_location="cpu", weights_only=False)["state_dict"]); O.to(dev)
H = affnet_amd.HardNet(); H.load_state_dict(affnet_amd.synthetic_hardnet_state(0)); H.to(dev)
big = (torch.rand(48000, 1, 32, 32) * 255).to(dev)
out = []
for name, net in (("AffNet", A), ("OriNet", O), ("HardNet", H)):
    net(big); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); net(big); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    out.append("%s %.3f" % (name, best))
print(os.environ.get("AFFNET_HIP_LIB", "default"), " ".join(out))
