#!/usr/bin/env python
"""graf 1 <-> 6 matching check = test() of the reference (train_AffNet_test_on_graffity.py:262-339), on the MI355X:
detect + AffNet + OriNet + HardNet on both images (one batched call when they have the same size), SNN ratio matching
(MFMA distance kernel, no N x N matrix in HBM), homography consistency at 6 px.

    python examples/graf_matching/match_graf.py [IMG1 IMG2 H1to2 [N [HARDNET.pth]]]

Defaults: tests/golden/graf_img1.png, graf_img6.png, graf_H1to6p, N = 3000.  The reference's HardNet++.pth is a missing
blob; without a checkpoint the seeded synthetic HardNet is used, whose descriptors are not discriminative (few matches)."""
import os
import sys

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import affnet_amd  # noqa: E402
from affnet_amd.ReprojectionStuff import match_snn, get_GT_correspondence_indexes  # noqa: E402


def load_grayscale_var(fname):        # train_AffNet_test_on_graffity.py:246-253
    img = np.mean(np.array(Image.open(fname).convert("RGB")), axis=2)
    return torch.from_numpy(img.astype(np.float32)).view(1, 1, img.shape[0], img.shape[1])


def main(argv):
    gd = os.path.join(ROOT, "tests", "golden")
    f1, f2, fh = (argv + [None] * 3)[:3]
    f1, f2, fh = f1 or os.path.join(gd, "graf_img1.png"), f2 or os.path.join(gd, "graf_img6.png"), fh or os.path.join(gd, "graf_H1to6p")
    n = int(argv[3]) if len(argv) > 3 else 3000
    dev = torch.device("cuda:0")
    ld = lambda p: torch.load(p, map_location="cpu", weights_only=False)["state_dict"]
    A = affnet_amd.AffNetFast(PS=32); A.load_state_dict(ld(os.path.join(ROOT, "pretrained", "AffNet.pth")))
    O = affnet_amd.OriNetFast(PS=32); O.load_state_dict(ld(os.path.join(ROOT, "pretrained", "OriNet.pth")))
    Hn = affnet_amd.HardNet(); Hn.load_state_dict(ld(argv[4]) if len(argv) > 4 else affnet_amd.synthetic_hardnet_state(0))
    A, O, Hn = A.to(dev), O.to(dev), Hn.to(dev)
    det = affnet_amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=n, border=5, num_Baum_iters=1, AffNet=A, OriNet=O).to(dev)
    img1, img2 = load_grayscale_var(f1).to(dev), load_grayscale_var(f2).to(dev)
    H1to2 = torch.from_numpy(np.loadtxt(fh)).float()
    for do_ori, tag in ((True, "graf1-6"), (False, "ori graf1-6")):
        if img1.shape == img2.shape:
            r1, r2 = det.run_batch(torch.cat([img1, img2], 0), do_ori=do_ori, desc=Hn)
        else:
            r1, r2 = det.run(img1, do_ori=do_ori, desc=Hn), det.run(img2, do_ori=do_ori, desc=Hn)
        t1, t2, _, _ = match_snn(r1["descriptors"], r2["descriptors"], 0.8)
        _, plain, _ = get_GT_correspondence_indexes(r1["LAFs"][t1], r2["LAFs"][t2], H1to2, dist_threshold=6)
        print("Test on %s, %d tentatives %d true matches %s  inl.ratio" % (tag, t1.numel(), plain.numel(), str(plain.numel() / max(1, t1.numel()))[:5]))


if __name__ == "__main__":
    main(sys.argv[1:])
