"""Deterministic synthetic benchmark images (SURVEY.md section 8d / BASELINE.md section 4): multi-octave
bilinear noise, min-max scaled to 0..255 float32.  Generated on the host with a seeded
torch.Generator, then uploaded; identical to the generator the parity oracle uses."""
import math

import torch
import torch.nn.functional as F


def synthetic_image(h, w, seed):
    g = torch.Generator().manual_seed(int(seed))
    acc = torch.zeros(1, 1, h, w)
    s = 1
    while min(h, w) / s >= 4:
        hh, ww = -(-h // s), -(-w // s)
        n = torch.rand(1, 1, hh, ww, generator=g)
        acc += math.sqrt(s) * F.interpolate(n, size=(h, w), mode="bilinear", align_corners=False)
        s *= 2
    acc = (acc - acc.min()) / (acc.max() - acc.min()) * 255.0
    return acc.float().contiguous()


def synthetic_hardnet_state(seed=0):
    """Stand-in HardNet weights (the reference's HardNet++.pth is a missing blob): seeded normal conv
    weights (He scale), BN running_mean ~ U(-.3,.3), running_var ~ U(.2,.6).  Same generator as the
    parity oracle, so descriptors are comparable."""
    g = torch.Generator().manual_seed(seed)
    widths = [(1, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128)]
    sd = {}
    for (ci, bi), (cin, cout) in zip([(0, 1), (3, 4), (6, 7), (9, 10), (12, 13), (15, 16)], widths):
        sd["features.%d.weight" % ci] = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (cin * 9))
        sd["features.%d.running_mean" % bi] = torch.rand(cout, generator=g) * 0.6 - 0.3
        sd["features.%d.running_var" % bi] = torch.rand(cout, generator=g) * 0.4 + 0.2
    sd["features.19.weight"] = torch.randn(128, 128, 8, 8, generator=g) * math.sqrt(1.0 / (128 * 64))
    sd["features.20.running_mean"] = torch.rand(128, generator=g) * 0.6 - 0.3
    sd["features.20.running_var"] = torch.rand(128, generator=g) * 0.4 + 0.2
    return sd
