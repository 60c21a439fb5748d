"""Times affnet_pyramid_build alone (HIP events): python tools/pyr_time.py [H W B]   (default 2160 3840 8)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import affnet_amd  # noqa: E402
from affnet_amd import engine  # noqa: E402
from affnet_amd._lib import lib, ptr, check  # noqa: E402

H, W, B = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (2160, 3840, 8)
dev = torch.device("cuda", 0)
x = (torch.rand(B, 1, H, W) * 255.0).to(dev)
det = affnet_amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=2000, border=5, num_Baum_iters=0).to(dev)
ctx = det._context(x, allow_batch=True)
st = engine.stream_of(dev)
for _ in range(3):
    check(lib.affnet_pyramid_build(ctx.handle, ptr(x), st), ctx.handle, "pyramid_build")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    check(lib.affnet_pyramid_build(ctx.handle, ptr(x), st), ctx.handle, "pyramid_build")
e1.record()
torch.cuda.synchronize()
print("pyramid %dx%d x %d: %.4f ms per image" % (W, H, B, e0.elapsed_time(e1) / 20 / B))
