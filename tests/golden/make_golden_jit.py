"""Second, independently produced CNN oracle: the reference's own TorchScript traces of AffNet and OriNet
(/root/reference/convertJIT/AffNetJIT.pt, OriNetJIT.pt - SURVEY.md section 2 row 8 / section 8c), run UNMODIFIED on the patches of
tests/golden/cnn_random_patches.npz.  The traces were made by the reference's author from the same checkpoints
(convertJIT/convert_OriNet_and_AffNet_to_JIT.ipynb cells 0-4) and return the RAW network outputs:

    AffNetJIT(patches) -> (n, 3) = (1 + x0, x1, 1 + x2): the matrix entries before rectifyAffineTransformationUpIsUp
    OriNetJIT(patches) -> (n, 2) = the (y, x) vector before atan2

    python tests/golden/make_golden_jit.py       -> tests/golden/cnn_jit_raw.npz

Only runs in the authoring container (the GPU box has no /root/reference); nothing of the traces is copied - only their outputs.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_harness as rh  # noqa: E402


def main():
    p = torch.from_numpy(np.load(os.path.join(HERE, "cnn_random_patches.npz"))["patches"])
    out = {}
    for key, name in (("affnet_raw", "AffNetJIT.pt"), ("orinet_raw", "OriNetJIT.pt")):
        m = rh.load_jit_trace(name)
        with torch.no_grad():
            out[key] = m(p).numpy()
        print(key, out[key].shape, out[key][:2])
    np.savez_compressed(os.path.join(HERE, "cnn_jit_raw.npz"), **out)


if __name__ == "__main__":
    main()
