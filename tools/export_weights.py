#!/usr/bin/env python
"""Checkpoint (.pth with a `state_dict`, reference key layout) -> flat AFNW0001 weight file for non-Python callers of the C ABI
(examples/c_host/extract.c):   python tools/export_weights.py affnet|orinet|hardnet IN.pth OUT.afnw"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from affnet_amd import _lib, engine  # noqa: E402

KINDS = {"affnet": _lib.NET_AFFNET, "orinet": _lib.NET_ORINET, "hardnet": _lib.NET_HARDNET}


def main(argv):
    if len(argv) != 3 or argv[0] not in KINDS:
        print(__doc__)
        return 1
    sd = torch.load(argv[1], map_location="cpu", weights_only=False)
    sd = sd.get("state_dict", sd)
    n = engine.save_flat_weights(KINDS[argv[0]], sd, argv[2])
    print("%s: %d floats" % (argv[2], n))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
