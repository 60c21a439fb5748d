"""Golden vectors of the OnePassSIR path from the UNMODIFIED reference classes: AffNetFastFullConv
(architectures.py:629-674, loaded with the shipped AffNet.pth - same `features` layout), LocalNorm2d, and OnePassSIR itself
(OnePassSIR.py executed in memory with its single Python-2 print statement rewritten, oracle/ref_harness.py:import_onepass_sir).
Authoring container only:   python tests/golden/make_golden_onepass.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import affnet_oracle as orc  # noqa: E402
import ref_harness as rh  # noqa: E402


def main():
    ns = rh.import_reference()
    sir = rh.import_onepass_sir()
    aff_sd, ori_sd = rh.load_state_dict("AffNet.pth"), rh.load_state_dict("OriNet.pth")
    FC = ns.architectures.AffNetFastFullConv(); FC.load_state_dict(aff_sd); FC.eval()
    O = ns.architectures.OriNetFast(PS=32); O.load_state_dict(ori_sd); O.eval()
    x = orc.synthetic_image(240, 320, 1)
    xs = orc.synthetic_image(131, 97, 4)            # ragged: partial tiles in every dense layer
    out = {}
    with torch.no_grad(), rh.quiet():
        out["norm_240x320"] = FC.lrn(x)[0, 0].numpy()
        out["map_240x320_sub"] = FC(x)[0, :, ::4, ::4].numpy()          # every 4th pixel of the (4,h,w) map
        out["norm_131x97"] = FC.lrn(xs)[0, 0].numpy()
        out["map_131x97"] = FC(xs)[0].numpy()
        det = sir.OnePassSIR(mrSize=5.192, num_features=300, border=15, num_Baum_iters=1, AffNet=FC, OriNet=O)
        L, r = det(x, do_ori=True)
        out["LAFs_n300"], out["resp_n300"] = L.numpy(), r.numpy()
        det = sir.OnePassSIR(mrSize=5.192, num_features=300, border=15, num_Baum_iters=1, AffNet=FC, OriNet=O)
        L, r = det(x, do_ori=False)
        out["LAFs_n300_noori"] = L.numpy()
        det = sir.OnePassSIR(mrSize=5.192, num_features=5000, border=15, num_Baum_iters=1, AffNet=FC, OriNet=O)   # fewer detections than the budget
        L, r = det(x, do_ori=False)
        out["LAFs_all_noori"], out["resp_all"] = L.numpy(), r.numpy()
    np.savez_compressed(os.path.join(HERE, "onepass_synth.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
