#!/usr/bin/env python
"""MFMA-loop probe: one HardNet conv layer's loop in isolation (affnet_cnn32_probe) with / without its loads."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import os as _os
_os.environ.setdefault("AFFNET_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "affnet_amd", "libaffnet_hip_probes.so"))   # probe kernels live there (include/affnet_hip_probes.h)
import affnet_amd
from affnet_amd._lib import lib, ptr
dev = torch.device("cuda:0")
H = affnet_amd.HardNet(); H.load_state_dict(affnet_amd.synthetic_hardnet_state(0)); H.to(dev)
pk = H.packed_weights(dev)
out = torch.zeros(2, device=dev)
reps, blocks = 40, 256 * 4
mfma = {1: 9216, 5: 9216}          # MFMAs per workgroup per repetition (conv1 / conv5 of HardNet)
for layer in (1, 5):
    for probe, name in ((0, "full loop"), (1, "no weight loads"), (2, "no activation loads"), (3, "MFMA + addressing only"),
                        (4, "full, accumulators in AGPRs"), (8, "full, conflict-free LDS pattern"), (9, "no weights, conflict-free LDS")):
        lib.affnet_cnn32_probe(ptr(pk), layer, probe, 2, blocks, ptr(out), None); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.affnet_cnn32_probe(ptr(pk), layer, probe, reps, blocks, ptr(out), None); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        tf = blocks * reps * mfma[layer] * 2048.0 / (ms * 1e-3) / 1e12
        print("conv%d  %-34s %8.3f ms  %.1f TFLOP/s  (%.1f %% of 157.3)" % (layer, name, ms, tf, 100 * tf / 157.3))
