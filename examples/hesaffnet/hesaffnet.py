#!/usr/bin/env python
"""Hessian-AffNet detector command line (MI355X): same contract as the reference's examples/hesaffnet/hesaffnet.py.

    python hesaffnet.py IMAGE OUT.txt NFEATS

Output = Oxford ellipse text file: line 1 "1.0", line 2 the count, then one `x y a b c` row per region.
The reference script passes th = -1 (hesaffnet.py:26,50), which switches the extractor to threshold mode: NFEATS is
ignored and every scale-space maximum is kept.  That default is kept; HESAFFNET_TH=none selects the feature budget
instead (th=None, what the reference's test() uses)."""
import os
import sys

import numpy as np
import torch
from PIL import Image

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, REPO)
import affnet_amd  # noqa: E402
from affnet_amd.LAF import LAFs2ell  # noqa: E402


def read_gray(path):
    """RGB -> per-pixel channel mean, float32 0..255, shape (1,1,H,W) (hesaffnet.py:35-39)."""
    rgb = np.asarray(Image.open(path).convert("RGB"), dtype=np.float64)
    gray = rgb.mean(axis=2).astype(np.float32)
    return torch.from_numpy(gray)[None, None]


def write_oxford(path, ellipses):
    with open(path, "w") as f:
        f.write("1.0\n%d\n" % len(ellipses))
        np.savetxt(f, ellipses, delimiter=" ", fmt="%10.10f")


def main(argv):
    if len(argv) != 3 or not argv[2].lstrip("-").isdigit():
        print("Wrong input format. Try python hesaffnet.py imgs/cat.png cat.txt 2000")
        return 1
    image_path, out_path, budget = argv[0], argv[1], int(argv[2])
    threshold = None if os.environ.get("HESAFFNET_TH", "").lower() == "none" else -1
    shape_net = affnet_amd.AffNetFast(PS=32)
    state = torch.load(os.path.join(REPO, "pretrained", "AffNet.pth"), map_location="cpu", weights_only=False)
    shape_net.load_state_dict(state["state_dict"])
    extractor = affnet_amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=budget, border=5, num_Baum_iters=1,
                                                          th=threshold, AffNet=shape_net).cuda()
    with torch.no_grad():
        frames, _ = extractor(read_gray(image_path).cuda())
    write_oxford(out_path, LAFs2ell(frames.cpu().numpy()))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
