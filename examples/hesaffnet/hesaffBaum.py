#!/usr/bin/env python
"""hesaffBaum.py of the reference (examples/hesaffnet/hesaffBaum.py:24-50) on the MI355X: Hessian detector + 16 Baumberg
shape iterations with the hand-crafted AffineShapeEstimator(patch_size=19), Oxford ellipse output through LAFs2ellT.

    python hesaffBaum.py IMG OUT NFEATS

(The reference script raises a TypeError as shipped - Utils.py:54 passes a kwargs dict positionally to
AffineShapeEstimator.forward(self, x); this is the flow it intends.)"""
import os
import sys

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from affnet_amd.SparseImgRepresenter import ScaleSpaceAffinePatchExtractor  # noqa: E402
from affnet_amd.HandCraftedModules import AffineShapeEstimator  # noqa: E402
from affnet_amd.LAF import LAFs2ellT  # noqa: E402
from affnet_amd.Utils import line_prepender  # noqa: E402

try:
    input_img_fname, output_fname, nfeats = sys.argv[1], sys.argv[2], int(sys.argv[3])
except Exception:
    print("Wrong input format. Try python hesaffBaum.py imgs/cat.png cat.txt 2000")
    sys.exit(1)

img = np.mean(np.array(Image.open(input_img_fname).convert("RGB")), axis=2)
var_image_reshape = torch.from_numpy(img.astype(np.float32)).view(1, 1, img.shape[0], img.shape[1]).cuda()
HA = ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=nfeats, border=5, num_Baum_iters=16,
                                    AffNet=AffineShapeEstimator(patch_size=19)).cuda()
with torch.no_grad():
    LAFs, resp = HA(var_image_reshape)
ells = LAFs2ellT(LAFs).cpu().numpy()
np.savetxt(output_fname, ells, delimiter=" ", fmt="%10.10f")
line_prepender(output_fname, str(len(ells)))
line_prepender(output_fname, "1.0")
