#!/usr/bin/env python
"""AffNet shapes for an HPatches-style patch column (MI355X): same contract as the reference's
examples/just_shape/detect_affine_shape.py.

    python detect_affine_shape.py COLUMN.png OUT.txt

COLUMN is an h x w grey image with h a multiple of w: every w x w tile is one patch.  Tiles are brought to 32 x 32
(bilinear, identity when w == 32), scaled to 0..1 and pushed through AffNet 128 at a time; OUT gets one row
`a11 a12 a21 a22` per patch ('%10.5f').  PIL does the resize (cv2 is not part of this image)."""
import os
import sys

import numpy as np
import torch
from PIL import Image

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, REPO)
import affnet_amd  # noqa: E402

SIDE, CHUNK = 32, 128


def tiles_of(column):
    """(h, w) uint8 column -> (n, 1, 32, 32) float32 in 0..1."""
    h, w = column.shape
    tiles = column[: (h // w) * w].reshape(h // w, w, w)
    if w != SIDE:
        tiles = np.stack([np.asarray(Image.fromarray(t).resize((SIDE, SIDE), Image.BILINEAR)) for t in tiles])
    return torch.from_numpy(tiles.astype(np.float32) / 255.0)[:, None]


def main(argv):
    if len(argv) != 2:
        print("Wrong input format. Try ./detect_affine_shape.py imgs/ref.png out.txt")
        return 1
    net = affnet_amd.AffNetFast(PS=SIDE)
    net.load_state_dict(torch.load(os.path.join(REPO, "pretrained", "AffNet.pth"), map_location="cpu", weights_only=False)["state_dict"])
    net.cuda()
    patches = tiles_of(np.asarray(Image.open(argv[0]).convert("L")))
    with torch.no_grad():
        shapes = [net(patches[i:i + CHUNK].cuda()).reshape(-1, 4).cpu() for i in range(0, len(patches), CHUNK)]
    np.savetxt(argv[1], torch.cat(shapes).numpy() if shapes else np.zeros((0, 4)), delimiter=" ", fmt="%10.5f")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
