"""Golden vectors of hesaffnet.py EXACTLY AS THE REFERENCE SHIPS IT (examples/hesaffnet/hesaffnet.py:24-60): `th = -1` => `num = -1`
(SparseImgRepresenter.py:33-37): no feature budget - every 3-D maximum of the response that survives the shape filter is returned, in
(octave, level, pixel) order; AffNetFast in the AffNet slot, no OriNet, do_ori = False; then LAFs2ell (LAF.py:225-240) = the rows of
the Oxford file the script writes.  The UNMODIFIED reference on its own input images at their own sizes: test-graf/img1.png (800x640,
~7000 rows) and examples/hesaffnet/img/{cat,fox1}.png.  Run in the authoring container only:

    python tests/golden/make_golden_thmode.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_harness as rh  # noqa: E402

IMAGES = (("graf_img1", "graf_img1.png"), ("cat", "hesaffnet_cat.png"), ("fox1", "hesaffnet_fox1.png"))


def load_gray(path):
    from PIL import Image
    img = np.mean(np.array(Image.open(path).convert("RGB")), axis=2)  # hesaffnet.py:35-36
    return torch.from_numpy(img.astype(np.float32)).view(1, 1, img.shape[0], img.shape[1])


def main():
    ns = rh.import_reference()
    A = ns.architectures.AffNetFast(PS=32); A.load_state_dict(rh.load_state_dict("AffNet.pth")); A.eval()
    for tag, fname in IMAGES:
        x = load_gray(os.path.join(HERE, fname))
        # hesaffnet.py:50 (nfeats is what argv[3] says; with th = -1 it is ignored: SparseImgRepresenter.py:33-35)
        HA = ns.SparseImgRepresenter.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=2000, border=5, num_Baum_iters=1, th=-1, AffNet=A)
        with torch.no_grad(), rh.quiet():
            L, r = HA(x)
        ells = ns.LAF.LAFs2ell(L.numpy())
        np.savez_compressed(os.path.join(HERE, "thmode_%s.npz" % tag), LAFs=L.numpy(), resp=r.numpy(), ells=ells.astype(np.float64),
                            hw=np.array([x.size(2), x.size(3)]))
        print("written:", tag, tuple(x.shape), "rows", L.shape[0])


if __name__ == "__main__":
    main()
