#!/usr/bin/env python
"""Which fp32 summation order does THIS host's torch CPU conv2d use for the detector's 27-tap centroid (HandCraftedModules.py:279: a
(1, 3, h, w) tensor convolved with 3 x 3 x 3 weights)?  CPU only, no GPU, no oracle.  For each map size the conv2d output is compared bit
for bit with fmaf chains in the six loop orders of (level c, ky k, kx l).  Expected (ATen Convolution.cpp use_mkldnn + oneDNN's direct
convolution for 3 input channels): 'ckl' when 3 h w <= 20480 (native im2col + sgemm), 'klc' above (oneDNN)."""
import itertools
import sys

import numpy as np
import torch
import torch.nn.functional as F


def fma32(a, b, c):
    return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(np.float32)      # products of two fp32 are exact in fp64


def orders_matching(h, w, seed=0):
    g = torch.Generator().manual_seed(seed)
    r3 = (torch.rand(1, 3, h, w, generator=g) ** 4 * 1000).float()
    off = torch.linspace(-0.5, 1.5, 3)
    wts = torch.zeros(3, 3, 3, 3)
    wts[0] = torch.tensor([2.0158737, 2.5398417, 3.2]).view(3, 1, 1).expand(3, 3, 3)
    wts[1] = off.view(1, 3, 1).expand(3, 3, 3)
    wts[2] = off.view(1, 1, 3).expand(3, 3, 3)
    out = F.conv2d(r3, wts, padding=1)[0].numpy()
    xp = F.pad(r3, (1, 1, 1, 1))[0].numpy()
    wn = wts.numpy()
    good = []
    for perm in itertools.permutations("ckl"):
        acc = np.zeros((3, h, w), np.float32)
        for a in range(3):
            for b in range(3):
                for d in range(3):
                    i = dict(zip(perm, (a, b, d)))
                    xs = xp[i["c"], i["k"]:i["k"] + h, i["l"]:i["l"] + w]
                    for oc in range(3):
                        acc[oc] = fma32(xs, wn[oc, i["c"], i["k"], i["l"]], acc[oc])
        if np.array_equal(acc, out):
            good.append("".join(perm))
    return good


if __name__ == "__main__":
    torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
    print("torch", torch.__version__, "threads", torch.get_num_threads())
    print(torch.__config__.show().split("\n")[3][:160])
    for h, w in [(24, 32), (48, 64), (82, 83), (83, 83), (96, 128), (192, 256), (384, 512), (598, 1000), (768, 1024)]:
        print("%4d x %4d (3hw = %7d, %s 20480): fmaf-chain orders equal to conv2d bit for bit: %s" % (h, w, 3 * h * w, ">" if 3 * h * w > 20480 else "<=", orders_matching(h, w)))
    # affine_grid (LAF.py:313-324): theta (n,2,3) x base grid through bmm - does it equal fmaf(t01, v, fmaf(t00, u, ...)) style chains?
    g = torch.Generator().manual_seed(1)
    th = (torch.rand(64, 2, 3, generator=g) * 600).float()
    grid = F.affine_grid(th, torch.Size((64, 1, 32, 32)), align_corners=False).numpy()
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    from affnet_amd import engine
    u = engine.base_grid(32)                   # torch's own base grid values (linspace(-1, 1, 32) * 31 / 32), as the HIP sampler uses them
    t = th.numpy()
    cands = {}
    X, Y = np.meshgrid(u, u)
    for n in range(64):
        for row in (0, 1):
            a, b, c = t[n, row]
            f64 = lambda v: v.astype(np.float64)
            c1 = (f64(X) * np.float64(a)).astype(np.float32)                                   # ((a x) + b y) + c, products rounded
            c1 = (f64(c1) + f64((f64(Y) * np.float64(b)).astype(np.float32))).astype(np.float32)
            c1 = (f64(c1) + np.float64(c)).astype(np.float32)
            c2 = fma32(np.ones_like(X), c, fma32(Y, b, (f64(X) * np.float64(a)).astype(np.float32)))   # fma(1, c, fma(y, b, x a))
            c3 = fma32(Y, b, fma32(X, a, np.full_like(X, c)))                                        # fma(y, b, fma(x, a, c))
            c4 = fma32(np.ones_like(X), c, fma32(Y, b, fma32(X, a, np.zeros_like(X))))
            for name, cv in (("plain (ax + by) + c", c1), ("fma(1,c,fma(y,b,x*a))", c2), ("fma(y,b,fma(x,a,c))", c3), ("fma chain from 0: x,y,1", c4)):
                cands[name] = cands.get(name, 0) + int((cv != grid[n, :, :, row]).sum())
    print("affine_grid vs candidate orders, mismatching elements of %d:" % (64 * 2 * 1024), cands)
