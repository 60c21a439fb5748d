#!/usr/bin/env python
"""OnePassSIR detector in THRESHOLD mode + HardNet descriptors (MI355X): the flow of the reference's
examples/hesaffnet/extract_geom_and_desc_upisupTh.py - the same script as extract_geom_and_desc_upisup.py with `num_features = -1, th = argv[3]`
(its line 63): every scale-space maximum whose response exceeds the threshold is kept.

    python extract_geom_and_desc_upisupTh.py IMAGE OUT.txt TH [HARDNET.pth]

Output as in the budget variant: Oxford ellipse file + OUT.txt.desc.npy.  (Same caveats: AffNetFastFullConv stands in for the reference's
non-existent `AffNetFastFullAff`; HardNet++.pth is a missing blob, seeded synthetic weights without a checkpoint argument.)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
import affnet_amd  # noqa: E402
from affnet_amd.LAF import LAFs2ellT  # noqa: E402
from extract_geom_and_desc_upisup import read_gray  # noqa: E402


def build(threshold, hardnet_ckpt=None, with_orinet=False):
    dense_shape = affnet_amd.AffNetFastFullConv(PS=32)
    dense_shape.load_state_dict(torch.load(os.path.join(REPO, "pretrained", "AffNet.pth"), map_location="cpu", weights_only=False)["state_dict"])
    descriptor = affnet_amd.HardNet()
    if hardnet_ckpt:
        descriptor.load_state_dict(torch.load(hardnet_ckpt, map_location="cpu", weights_only=False)["state_dict"])
    else:
        print("no HardNet checkpoint given: seeded synthetic HardNet weights")
        descriptor.load_state_dict(affnet_amd.synthetic_hardnet_state(0))
    ori = None
    if with_orinet:
        ori = affnet_amd.OriNetFast(PS=32)
        ori.load_state_dict(torch.load(os.path.join(REPO, "pretrained", "OriNet.pth"), map_location="cpu", weights_only=False)["state_dict"])
    detector = affnet_amd.OnePassSIR(mrSize=5.192, num_features=-1, th=threshold, border=15, num_Baum_iters=1, AffNet=dense_shape, OriNet=ori).cuda()
    return detector, descriptor.cuda()


def main(argv):
    try:
        image_path, out_path, threshold = argv[0], argv[1], float(argv[2])
    except (IndexError, ValueError):
        print("Wrong input format. Try python extract_geom_and_desc_upisupTh.py imgs/cat.png cat.txt 5.3333")
        return 1
    detector, descriptor = build(threshold, argv[3] if len(argv) > 3 else None)
    with torch.no_grad():
        frames, _ = detector(read_gray(image_path).cuda())
        descs = descriptor(detector.extract_patches_from_pyr(frames, PS=32))
        ells = LAFs2ellT(frames).cpu().numpy()
    with open(out_path, "w") as f:
        f.write("1.0\n%d\n" % len(ells))
        np.savetxt(f, ells, delimiter=" ", fmt="%10.10f")
    np.save(out_path + ".desc.npy", descs.cpu().numpy())
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
