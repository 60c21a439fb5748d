#!/usr/bin/env python
"""Hessian detector + 16 Baumberg shape iterations (MI355X): the flow of the reference's examples/hesaffnet/hesaffBaum.py.

    python hesaffBaum.py IMAGE OUT.txt NFEATS

Shape estimation uses the hand-crafted AffineShapeEstimator(patch_size=19) slot, ellipses come from LAFs2ellT on the device.
(As shipped, the reference script stops with a TypeError: Utils.py:54 hands a kwargs dict positionally to
AffineShapeEstimator.forward(self, x).  This is the pipeline it describes.)"""
import os
import sys

import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import affnet_amd  # noqa: E402
from affnet_amd.HandCraftedModules import AffineShapeEstimator  # noqa: E402
from affnet_amd.LAF import LAFs2ellT  # noqa: E402
from hesaffnet import read_gray, write_oxford  # noqa: E402


def main(argv):
    if len(argv) != 3 or not argv[2].isdigit():
        print("Wrong input format. Try python hesaffBaum.py imgs/cat.png cat.txt 2000")
        return 1
    extractor = affnet_amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=int(argv[2]), border=5, num_Baum_iters=16,
                                                          AffNet=AffineShapeEstimator(patch_size=19)).cuda()
    with torch.no_grad():
        frames, _ = extractor(read_gray(argv[0]).cuda())
    write_oxford(argv[1], LAFs2ellT(frames).cpu().numpy())
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
