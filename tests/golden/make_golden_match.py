"""Golden vectors for SURVEY section 8f row 1 (descriptor matching + homography check), produced by the UNMODIFIED
reference functions Losses.distance_matrix_vector / ReprojectionStuff.get_GT_correspondence_indexes and the matching
lines of train_AffNet_test_on_graffity.py:292-300 executed verbatim on CPU.

    python tests/golden/make_golden_match.py

Inputs: graf img1 / img6 (500 keypoints each, full path with the synthetic HardNet), H1to6p."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import affnet_oracle as orc  # noqa: E402
import ref_harness as rh  # noqa: E402
from make_golden import load_gray  # noqa: E402


def main():
    ns = rh.import_reference()
    import Losses
    import ReprojectionStuff
    aff_sd, ori_sd = rh.load_state_dict("AffNet.pth"), rh.load_state_dict("OriNet.pth")
    A = ns.architectures.AffNetFast(PS=32); A.load_state_dict(aff_sd); A.eval()
    O = ns.architectures.OriNetFast(PS=32); O.load_state_dict(ori_sd); O.eval()
    Hn = ns.HardNet.HardNet(); Hn.load_state_dict(orc.synthetic_hardnet_state(0)); Hn.eval()
    out = {}
    feats = []
    for name in ("graf_img1.png", "graf_img6.png"):
        x = load_gray(os.path.join(HERE, name))
        det = ns.SparseImgRepresenter.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=500, border=5, num_Baum_iters=1,
                                                                      AffNet=A, OriNet=O)
        with torch.no_grad(), rh.quiet():
            L, r = det(x, do_ori=True)
            D = Hn(det.extract_patches_from_pyr(L, PS=32))
        feats.append((L, D))
    (L1, D1), (L2, D2) = feats
    H = torch.from_numpy(np.loadtxt(os.path.join(HERE, "graf_H1to6p"))).float()
    # train_AffNet_test_on_graffity.py:292-305, verbatim modulo .cuda()
    SNN_threshold = 0.8
    dist_matrix = Losses.distance_matrix_vector(D1, D2)
    out["dist_head"] = dist_matrix[:8, :8].numpy().copy()
    min_dist, idxs_in_2 = torch.min(dist_matrix, 1)
    dist_matrix[:, idxs_in_2] = 100000
    min_2nd_dist, idxs_2nd_in_2 = torch.min(dist_matrix, 1)
    mask = (min_dist / (min_2nd_dist + 1e-8)) <= SNN_threshold
    tent1 = torch.arange(0, idxs_in_2.size(0))[mask].long()
    tent2 = idxs_in_2[mask].long()
    gd, plain, in2 = ReprojectionStuff.get_GT_correspondence_indexes(L1[tent1], L2[tent2], H, dist_threshold=6)
    out.update(LAFs1=L1.numpy(), desc1=D1.numpy(), LAFs2=L2.numpy(), desc2=D2.numpy(), H=H.numpy(), min_dist=min_dist.numpy(),
               idx=idxs_in_2.numpy(), min_2nd=min_2nd_dist.numpy(), tent1=tent1.numpy(), tent2=tent2.numpy(), gt_dist=gd.numpy(),
               gt_plain=plain.numpy(), gt_idx=in2.numpy(),
               reproj=ReprojectionStuff.reprojectLAFs(L2[tent2], torch.inverse(H)).numpy())
    np.savez_compressed(os.path.join(HERE, "match_graf16_n500.npz"), **out)
    print("tentatives %d, homography-consistent %d" % (tent1.numel(), plain.numel()))


if __name__ == "__main__":
    main()
