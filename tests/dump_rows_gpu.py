#!/usr/bin/env python
"""GPU side of the wide parity sweep (round 5): runs the fused path on a list of images beyond the ones the tests use and writes the rows
(ids, LAFs, responses, descriptors) as .npz - no oracle needed here.  tests/offline_parity_account.py then compares them with the reference
on the authoring host (the host of tests/golden) and accounts for every key and row.

    python tests/dump_rows_gpu.py gpurun_out/rows_sweep         (on the GPU box)
    python tests/offline_parity_account.py gpurun_out/rows_sweep profiles/rNN_offline_parity_sweep_authoring_host.json    (here)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import affnet_amd  # noqa: E402
from conftest import load_gray  # noqa: E402

DEV = "cuda:0"


def main(out):
    os.makedirs(out, exist_ok=True)
    sd = {k: torch.load(os.path.join(ROOT, "pretrained", k + ".pth"), map_location="cpu", weights_only=False)["state_dict"] for k in ("AffNet", "OriNet")}
    A = affnet_amd.AffNetFast(PS=32); A.load_state_dict(sd["AffNet"]); A = A.to(DEV)
    O = affnet_amd.OriNetFast(PS=32); O.load_state_dict(sd["OriNet"]); O = O.to(DEV)
    H = affnet_amd.HardNet(); H.load_state_dict(affnet_amd.synthetic_hardnet_state(0)); H = H.to(DEV)
    if len(sys.argv) > 2 and sys.argv[2] == "metric_batch":
        # every image of the metric's batch (BASELINE configs[2]: seeds 0 .. 63 of bench.py), as ONE 32-image batched call per half like the benchmark
        for half in (0, 1):
            seeds = list(range(32 * half, 32 * half + 32))
            xb = torch.cat([affnet_amd.synthetic_image(768, 1024, s) for s in seeds], 0).to(DEV)
            for arith in ("fp32", "fp32_split2h"):
                det = affnet_amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=2000, border=5, num_Baum_iters=1, AffNet=A, OriNet=O, arith=arith).to(DEV)
                for s, r in zip(seeds, det.run_batch(xb, do_ori=True, desc=H)):
                    np.savez_compressed(os.path.join(out, "sweep_synth_768x1024_s%d_n2000%s.npz" % (s, "" if arith == "fp32" else "__arith_" + arith)),
                                        ids=r["ids"].cpu().numpy(), LAFs=r["LAFs"].cpu().numpy(), resp=r["responses"].cpu().numpy())
                del det
            print("metric batch, seeds %d .. %d" % (seeds[0], seeds[-1]), flush=True)
        return
    cases = [("sweep_synth_768x1024_s%d_n2000" % s, lambda s=s: affnet_amd.synthetic_image(768, 1024, s), 2000) for s in range(100, 124)]
    cases += [("sweep_synth_768x1024_s%d_n%d" % (s, n), lambda s=s: affnet_amd.synthetic_image(768, 1024, s), n) for s, n in ((7, 500), (8, 1000), (9, 4000))]
    cases += [("sweep_synth_480x640_s%d_n1000" % s, lambda s=s: affnet_amd.synthetic_image(480, 640, s), 1000) for s in (40, 41, 42, 43)]
    cases += [("sweep_synth_2160x3840_s%d_n8000" % s, lambda s=s: affnet_amd.synthetic_image(2160, 3840, s), 8000) for s in (1, 2, 3)]
    cases += [("sweep_graf_img6_n2000", lambda: load_gray(os.path.join(HERE, "golden", "graf_img6.png")), 2000),
              ("sweep_hesaffnet_cat_n1000", lambda: load_gray(os.path.join(HERE, "golden", "hesaffnet_cat.png")), 1000)]
    for name, img, n in cases:
        x = img().to(DEV)
        for arith in ("fp32", "fp32_split3", "fp32_split2h"):
            if arith != "fp32" and not (name.endswith("s100_n2000") or name.endswith("s101_n2000") or "graf_img6" in name or name.endswith("s1_n8000")):
                continue                                  # the split modes on a subset (every full-path test already runs them)
            det = affnet_amd.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=n, border=5, num_Baum_iters=1, AffNet=A, OriNet=O, arith=arith).to(DEV)
            r = det.run(x, do_ori=True, desc=H)
            np.savez_compressed(os.path.join(out, name + ("" if arith == "fp32" else "__arith_" + arith) + ".npz"), ids=r["ids"].cpu().numpy(),
                                LAFs=r["LAFs"].cpu().numpy(), resp=r["responses"].cpu().numpy(), desc=r["descriptors"].cpu().numpy())
            del det
        print(name, tuple(x.shape[2:]), n, flush=True)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "rows_sweep"))
