import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def weights():
    """AffNet / OriNet shipped checkpoints (byte copies under pretrained/) and the synthetic
    HardNet stand-in (HardNet++.pth is a missing blob in the reference)."""
    import torch
    import affnet_oracle as orc
    out = {}
    for k in ("AffNet", "OriNet"):
        ck = torch.load(os.path.join(ROOT, "pretrained", k + ".pth"), map_location="cpu", weights_only=False)
        out[k] = ck["state_dict"]
    out["HardNet"] = orc.synthetic_hardnet_state(0)
    return out


def load_gray(path):
    """hesaffnet.py:35-39 loader: RGB -> mean over channels -> float32 (1,1,H,W), 0..255."""
    import numpy as np
    import torch
    from PIL import Image
    img = np.mean(np.array(Image.open(path).convert("RGB")), axis=2)
    return torch.from_numpy(img.astype(np.float32)).view(1, 1, img.shape[0], img.shape[1])
