#!/usr/bin/env python
"""OnePassSIR (threshold 28.41) + OriNet + HardNet, LAFs saved as `lafs1.npy` (MI355X): the flow of the reference's
examples/hesaffnet/extract_geomOriTh.py - `th = 28.41` hard-wired (its line 31), AffNet + OriNetFast slots, `np.save('lafs1.npy', LAFs)`
as its only live output (line 88; the ellipse / .mat writers are commented out there).

    python extract_geomOriTh.py IMAGE OUT_PREFIX [HARDNET.pth]

Writes OUT_PREFIX.lafs1.npy (the (N, 2, 3) pixel LAFs; the reference drops `lafs1.npy` into the working directory) and
OUT_PREFIX.desc.npy (descriptors, which the reference computes and discards)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from extract_geom_and_desc_upisup import read_gray  # noqa: E402
from extract_geom_and_desc_upisupTh import build  # noqa: E402

TH = 28.41      # extract_geomOriTh.py:31


def main(argv):
    if len(argv) < 2:
        print("Wrong input format. Try python extract_geomOriTh.py imgs/cat.png cat")
        return 1
    detector, descriptor = build(TH, argv[2] if len(argv) > 2 else None, with_orinet=True)
    with torch.no_grad():
        frames, _ = detector(read_gray(argv[0]).cuda(), do_ori=True)
        descs = descriptor(detector.extract_patches_from_pyr(frames, PS=32))
    np.save(argv[1] + ".lafs1.npy", frames.cpu().numpy())
    np.save(argv[1] + ".desc.npy", descs.cpu().numpy())
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
