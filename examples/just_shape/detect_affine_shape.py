#!/usr/bin/env python
"""AffNet on an HPatches-style patch column - the reference's examples/just_shape/detect_affine_shape.py
on MI355X.

    python detect_affine_shape.py imgs/ref.png out.txt

The column (h x w grey image, h a multiple of w) is cut into w x w tiles; each tile is resized to 32x32
(bilinear; identity when w == 32), divided by 255 and fed to AffNet in batches of 128; rows of the
output text file are `a11 a12 a21 a22` ('%10.5f').  cv2 is not available in this image, so the tile
resize uses PIL (same bilinear kernel for the identity / integer cases the tests cover).
"""
import os
import sys

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from affnet_amd.architectures import AffNetFast  # noqa: E402

PS = 32
model = AffNetFast(PS=PS)
checkpoint = torch.load(os.path.join(ROOT, "pretrained", "AffNet.pth"), map_location="cpu", weights_only=False)
model.load_state_dict(checkpoint["state_dict"])
model.eval()
model.cuda()

try:
    input_img_fname = sys.argv[1]
    output_fname = sys.argv[2]
except Exception:
    print("Wrong input format. Try ./detect_affine_shape.py imgs/ref.png out.txt")
    sys.exit(1)

image = np.array(Image.open(input_img_fname).convert("L"))
h, w = image.shape
n_patches = h // w
patches = np.ndarray((n_patches, 1, PS, PS), dtype=np.float32)
for i in range(n_patches):
    patch = image[i * w:(i + 1) * w, 0:w]
    if w != PS:
        patch = np.array(Image.fromarray(patch).resize((PS, PS), Image.BILINEAR))
    patches[i, 0, :, :] = patch / 255.0
descriptors_for_net = np.zeros((n_patches, 4))
bs = 128
for st in range(0, n_patches, bs):
    data_a = torch.from_numpy(patches[st:st + bs]).cuda()
    with torch.no_grad():
        out_a = model(data_a)
    descriptors_for_net[st:st + bs, :] = out_a.data.cpu().numpy().reshape(-1, 4)
np.savetxt(output_fname, descriptors_for_net, delimiter=" ", fmt="%10.5f")
