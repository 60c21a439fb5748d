#!/usr/bin/env python
"""Hessian-AffNet detector CLI - the reference's examples/hesaffnet/hesaffnet.py on MI355X.

    python hesaffnet.py imgs/cat.png cat.txt 2000

Same arguments, same loader (RGB -> channel mean -> float32 0..255), same constructor call, same output
(Oxford ellipse text file: "1.0", count, then `x y a b c` rows).  Like the reference (hesaffnet.py:26,50)
the default threshold th = -1 is passed, which makes the extractor ignore `nfeats` and keep every maximum;
set HESAFFNET_TH=none to use the feature budget instead (th=None, the form the reference's test() uses).
"""
import os
import sys

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from affnet_amd.SparseImgRepresenter import ScaleSpaceAffinePatchExtractor  # noqa: E402
from affnet_amd.LAF import LAFs2ell  # noqa: E402
from affnet_amd.Utils import line_prepender  # noqa: E402
from affnet_amd.architectures import AffNetFast  # noqa: E402

th = -1  # hesaffnet.py:26
if os.environ.get("HESAFFNET_TH", "").lower() == "none":
    th = None
try:
    input_img_fname = sys.argv[1]
    output_fname = sys.argv[2]
    nfeats = int(sys.argv[3])
except Exception:
    print("Wrong input format. Try python hesaffnet.py imgs/cat.png cat.txt 2000")
    sys.exit(1)

img = Image.open(input_img_fname).convert("RGB")
img = np.mean(np.array(img), axis=2)
var_image_reshape = torch.from_numpy(img.astype(np.float32)).view(1, 1, img.shape[0], img.shape[1])

AffNetPix = AffNetFast(PS=32)
weightd_fname = os.path.join(ROOT, "pretrained", "AffNet.pth")
checkpoint = torch.load(weightd_fname, map_location="cpu", weights_only=False)
AffNetPix.load_state_dict(checkpoint["state_dict"])
AffNetPix.eval()

HA = ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=nfeats, border=5, num_Baum_iters=1, th=th, AffNet=AffNetPix)
HA = HA.cuda()
var_image_reshape = var_image_reshape.cuda()
with torch.no_grad():
    LAFs, resp = HA(var_image_reshape)
ells = LAFs2ell(LAFs.data.cpu().numpy())

np.savetxt(output_fname, ells, delimiter=" ", fmt="%10.10f")
line_prepender(output_fname, str(len(ells)))
line_prepender(output_fname, "1.0")
