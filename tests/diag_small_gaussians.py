#!/usr/bin/env python
"""TEST INFRASTRUCTURE (uses the CPU oracle; not collected by pytest, run by hand on the GPU box: python tests/diag_small_gaussians.py).
Box-side diagnosis: for small Gaussians (K = 3, 5, 7) compare (a) the HIP blur, (b) the oracle's conv2d on THIS host's CPU and
(c) a numpy emulation of the row-major fmaf chain.  Tells whether a mismatch is the GPU kernel's or this CPU's oneDNN kernel choice."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (this file lives in tests/)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import affnet_oracle as orc
from affnet_amd import engine
from affnet_amd._lib import lib, ptr, check
from affnet_amd.host_plan import gaussian_taps

x = orc.synthetic_image(240, 320, 1)
ctx = engine.utility_ctx(torch.device("cuda:0"))
print(torch.__config__.show().split("\n")[3:6], torch.get_num_threads())
for sigma in [0.306, 0.3856, 0.4859, 0.6122, 0.7, 0.9, 1.2263]:
    taps = gaussian_taps(sigma)
    k = taps.shape[0]
    R = k // 2
    xin = x[0, 0].cuda().contiguous()
    out = torch.empty_like(xin)
    buf = (C.c_float * (k * k))(*taps.reshape(-1).tolist())
    check(lib.affnet_gauss_blur(ctx, ptr(xin), ptr(out), 240, 320, buf, k, None), ctx, "blur")
    torch.cuda.synchronize()
    g = out.cpu().numpy()
    o = orc.gaussian_blur(x, sigma)[0, 0].numpy()
    torch.set_num_threads(1)
    o1 = orc.gaussian_blur(x, sigma)[0, 0].numpy()
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    xp = F.pad(x, (R, R, R, R), "replicate")[0, 0].numpy()
    acc = np.zeros((240, 320), dtype=np.float32)
    for i in range(k):
        for j in range(k):
            acc = (xp[i:i + 240, j:j + 320].astype(np.float64) * np.float64(taps[i, j]) + acc.astype(np.float64)).astype(np.float32)
    print("sigma %.4f K=%d: gpu vs chain %g | oracle(cpu, all threads) vs chain %g | oracle(1 thread) vs chain %g | gpu vs oracle %g"
          % (sigma, k, np.abs(g - acc).max(), np.abs(o - acc).max(), np.abs(o1 - acc).max(), np.abs(g - o).max()))

# ---- threshold-mode shape filter: which rows does the GPU keep that the oracle drops (or vice versa), and how close to the thresholds?
import affnet_amd
sd = {k: torch.load(os.path.join(ROOT, "pretrained", k + ".pth"), map_location="cpu", weights_only=False)["state_dict"] for k in ("AffNet", "OriNet")}
A = affnet_amd.AffNetFast(PS=32); A.load_state_dict(sd["AffNet"]); A = A.cuda()
det = affnet_amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, th=-1, AffNet=A).cuda()
L, r = det(x.cuda())
ex = orc.OracleExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=1, th=-1, affnet_sd=sd["AffNet"])
Lw, rw = ex(x)
key = lambda a: [tuple(int(v) for v in row) for row in np.asarray(a)]
kg, kw = set(key(det.last_ids.cpu().numpy())), set(key(ex.keys.numpy()))
d = ex.detected
with torch.no_grad():
    patches = orc.extract_from_pyramid(ex.scale_pyr, d["oct"], d["lev"], d["lafs"], 32)
    Aw = orc.affnet_batched(sd["AffNet"], patches, 256)
    Ag = A(patches.cuda()).cpu()
new = torch.cat([torch.bmm(Aw, d["lafs"][:, :, :2]), d["lafs"][:, :, 2:]], dim=2)
l1, l2 = orc.eig2x2(Aw)
ratio = torch.abs(l1 / (l2 + 1e-8))
pts = torch.tensor([[-1.0, -1, 1, 1], [-1, 1, -1, 1], [1, 1, 1, 1]]).unsqueeze(0)
Hm = torch.cat([new, torch.tensor([0.0, 0, 1]).view(1, 1, 3).repeat(new.size(0), 1, 1)], dim=1)
corners = torch.bmm(Hm, pts.expand(new.size(0), 3, 4))[:, :2, :]
allk = key(torch.stack([d["oct"].long(), d["lev"].long(), d["pix"].long()], 1).numpy())
print("th mode: gpu rows %d, oracle rows %d; only gpu %d, only oracle %d" % (len(kg), len(kw), len(kg - kw), len(kw - kg)))
for i, k in enumerate(allk):
    if (k in kg) != (k in kw):
        print("  key", k, "in gpu" if k in kg else "in oracle", "| ratio %.7f | corner min %.9f max %.9f | A cpu %s | A gpu-cpu max %.3g"
              % (float(ratio[i]), float(corners[i].min()), float(corners[i].max()), Aw[i].reshape(-1).tolist(), float((Ag[i] - Aw[i]).abs().max())))
