"""Golden vectors for SURVEY section 8f row 2 (hand-crafted default slot fillers), from the UNMODIFIED reference classes
HandCraftedModules.OrientationDetector / AffineShapeEstimator:
  * unit level: both modules on 48 random 19x19 patches;
  * full path with the default-constructed extractor (no AffNet / OriNet arguments): do_ori=True, num_Baum_iters=0;
  * Baumberg iterations: as shipped the reference raises a TypeError here (Utils.py:54 passes a kwargs dict positionally to
    AffineShapeEstimator.forward(self, x)); the class is used behind a one-line subclass whose forward ignores extra arguments.

    python tests/golden/make_golden_handcrafted.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import affnet_oracle as orc  # noqa: E402
import ref_harness as rh  # noqa: E402


def main():
    ns = rh.import_reference()
    HC = ns.HandCraftedModules
    g = torch.Generator().manual_seed(13)
    rp = torch.rand(48, 1, 19, 19, generator=g) * 255.0
    rp[40:] = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(rp[40:], (2, 2, 2, 2), "replicate"), 5, 1)   # smooth patches
    out = {"patches": rp.numpy()}
    with torch.no_grad():
        out["ori_angles"] = HC.OrientationDetector(patch_size=19)(rp).numpy()
        out["baum_A"] = HC.AffineShapeEstimator(patch_size=19)(rp).numpy()
    x = orc.synthetic_image(240, 320, 1)
    SSAPE = ns.SparseImgRepresenter.ScaleSpaceAffinePatchExtractor
    det = SSAPE(mrSize=5.192, num_features=300, border=5, num_Baum_iters=0)            # default slots
    with torch.no_grad(), rh.quiet():
        L, r = det(x, do_ori=True)
    out["default_LAFs"], out["default_resp"] = L.numpy(), r.numpy()
    out["default_ellT"] = ns.LAF.LAFs2ellT(L.clone()).numpy()            # LAF.py:35-51

    class BaumShim(HC.AffineShapeEstimator):
        def forward(self, x, *ignored):
            return super(BaumShim, self).forward(x)

    det = SSAPE(mrSize=5.192, num_features=300, border=5, num_Baum_iters=4, AffNet=BaumShim(patch_size=19))
    with torch.no_grad(), rh.quiet():
        L, r = det(x, do_ori=False)
    out["baum4_LAFs"], out["baum4_resp"] = L.numpy(), r.numpy()
    # hesaffBaum.py:40 as shipped: num_Baum_iters = 16 (same shim), and the file content it writes: LAFs2ellT (hesaffBaum.py:47)
    det = SSAPE(mrSize=5.192, num_features=300, border=5, num_Baum_iters=16, AffNet=BaumShim(patch_size=19))
    with torch.no_grad(), rh.quiet():
        L, r = det(x, do_ori=False)
    out["baum16_LAFs"], out["baum16_resp"] = L.numpy(), r.numpy()
    out["baum16_ellT"] = ns.LAF.LAFs2ellT(L.clone()).numpy()
    np.savez_compressed(os.path.join(HERE, "handcrafted_slots.npz"), **out)
    print("written; default path %d LAFs, Baumberg x4 %d LAFs, x16 %d LAFs" % (out["default_LAFs"].shape[0], out["baum4_LAFs"].shape[0], out["baum16_LAFs"].shape[0]))


if __name__ == "__main__":
    main()
