#!/usr/bin/env python
"""OnePassSIR detector + HardNet descriptors command line (MI355X): the flow of the reference's
examples/hesaffnet/extract_geom_and_desc_upisup.py.

    python extract_geom_and_desc_upisup.py IMAGE OUT.txt NFEATS [HARDNET.pth]

The reference script cannot run as shipped (Python-2 prints; it imports a class `AffNetFastFullAff` that does not exist in
architectures.py); the fully-convolutional AffNet that does exist, AffNetFastFullConv, takes the shipped AffNet.pth (same `features`
layout) and is used here.  Output = Oxford ellipse file ("1.0", count, `x y a b c` rows) from LAFs2ellT, plus OUT.txt.desc.npy with
the (N,128) descriptors.  HardNet++.pth is not part of the reference snapshot: without a checkpoint argument seeded synthetic weights
are used (and said so)."""
import os
import sys

import numpy as np
import torch
from PIL import Image

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, REPO)
import affnet_amd  # noqa: E402
from affnet_amd.LAF import LAFs2ellT  # noqa: E402


def read_gray(path):
    rgb = np.asarray(Image.open(path).convert("RGB"), dtype=np.float64)
    return torch.from_numpy(rgb.mean(axis=2).astype(np.float32))[None, None]


def main(argv):
    if len(argv) not in (3, 4) or not argv[2].isdigit():
        print("Wrong input format. Try python extract_geom_and_desc_upisup.py imgs/cat.png cat.txt 2000")
        return 1
    image_path, out_path, budget = argv[0], argv[1], int(argv[2])
    dense_shape = affnet_amd.AffNetFastFullConv(PS=32)
    dense_shape.load_state_dict(torch.load(os.path.join(REPO, "pretrained", "AffNet.pth"), map_location="cpu", weights_only=False)["state_dict"])
    descriptor = affnet_amd.HardNet()
    if len(argv) == 4:
        descriptor.load_state_dict(torch.load(argv[3], map_location="cpu", weights_only=False)["state_dict"])
    else:
        print("no HardNet checkpoint given: seeded synthetic HardNet weights")
        descriptor.load_state_dict(affnet_amd.synthetic_hardnet_state(0))
    detector = affnet_amd.OnePassSIR(mrSize=5.192, num_features=budget, border=15, num_Baum_iters=1, AffNet=dense_shape).cuda()
    descriptor = descriptor.cuda()
    with torch.no_grad():
        frames, _ = detector(read_gray(image_path).cuda())                    # orientation by the default OrientationDetector(19)
        descs = descriptor(detector.extract_patches_from_pyr(frames, PS=32))
        ells = LAFs2ellT(frames).cpu().numpy()
    with open(out_path, "w") as f:
        f.write("1.0\n%d\n" % len(ells))
        np.savetxt(f, ells, delimiter=" ", fmt="%10.10f")
    np.save(out_path + ".desc.npy", descs.cpu().numpy())
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
