"""Golden vectors on odd-sized inputs the reference itself ships: examples/hesaffnet/img/cat.png (598 x 1000) and fox1.png
(1000 x 563), plus a synthetic 641 x 481 image (odd level 0) - the UNMODIFIED reference, 2000 keypoints, full path
detect + AffNet + OriNet + HardNet (seeded synthetic HardNet weights: HardNet++.pth is a missing blob upstream).
Run in the authoring container only:

    python tests/golden/make_golden_oddsize.py

The two PNGs are byte copies of the reference's data files (the GPU box has no /root/reference).  Rows are in the reference's
output order; tests match rows through the response bit pattern and, for exact response ties, the frame centre (tests/_rowmatch.py)."""
import os
import shutil
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import affnet_oracle as orc  # noqa: E402  (synthetic input + synthetic HardNet weights only)
import ref_harness as rh  # noqa: E402
from conftest import load_gray  # noqa: E402  (hesaffnet.py:35-39 loader)


def main():
    ns = rh.import_reference()
    A = ns.architectures.AffNetFast(PS=32); A.load_state_dict(rh.load_state_dict("AffNet.pth")); A.eval()
    O = ns.architectures.OriNetFast(PS=32); O.load_state_dict(rh.load_state_dict("OriNet.pth")); O.eval()
    Hn = ns.HardNet.HardNet(); Hn.load_state_dict(orc.synthetic_hardnet_state(0)); Hn.eval()
    cases = []
    for name in ("cat", "fox1"):
        src = os.path.join(rh.REF_ROOT, "examples", "hesaffnet", "img", name + ".png")
        dst = os.path.join(HERE, "hesaffnet_%s.png" % name)
        shutil.copyfile(src, dst)
        os.chmod(dst, 0o644)
        cases.append(("hesaffnet_" + name, load_gray(dst)))
    cases.append(("synth_481x641_s5", orc.synthetic_image(481, 641, 5)))
    for name, x in cases:
        det = ns.SparseImgRepresenter.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=2000, border=5, num_Baum_iters=1, AffNet=A, OriNet=O)
        with torch.no_grad(), rh.quiet():
            L, r = det(x, do_ori=True)
            D = Hn(det.extract_patches_from_pyr(L, PS=32))
        np.savez_compressed(os.path.join(HERE, name + "_n2000.npz"), LAFs=L.numpy(), resp=r.numpy(), desc=D.numpy(), hw=np.array(x.shape[2:]))
        print("written:", name, tuple(x.shape), L.shape, D.shape, "unique responses:", len(np.unique(r.numpy())))


if __name__ == "__main__":
    main()
