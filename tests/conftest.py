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


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a host without a GPU: the gpu-marked tests are skipped (not errors), so the CPU suite is reachable without -m."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (run with -m gpu on the GPU box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


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


# ---- parity report: the GPU parity tests record what they measured (pytest -q swallows prints); written once per session
PARITY_REPORT = {}


def record_parity(name, **numbers):
    PARITY_REPORT[name] = numbers


def pytest_sessionfinish(session, exitstatus):
    if not PARITY_REPORT:
        return
    import json
    path = os.environ.get("AFFNET_PARITY_REPORT", os.path.join(ROOT, "gpurun_out", "parity_report.json"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        doc = {"what": "measured parity of the HIP path vs oracle/affnet_oracle.py (run live on this host) and vs tests/golden "
                       "(unmodified reference on the authoring host); written by tests/test_gpu_parity.py",
               "exitstatus": int(exitstatus), "cases": PARITY_REPORT}
        try:
            import torch
            if torch.cuda.is_available():
                doc["device"] = torch.cuda.get_device_name(0)
        except Exception:
            pass
        json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
