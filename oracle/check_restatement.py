"""Pins oracle/affnet_oracle.py against the UNMODIFIED reference, bit for bit.

Runs only where /root/reference exists (the authoring container).  Exit code 0 = every
compared tensor is bit-identical.  Usage: python oracle/check_restatement.py [--fast]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import affnet_oracle as orc  # noqa: E402
import onepass_oracle as opo  # noqa: E402
import ref_harness as rh  # noqa: E402


def same(name, a, b):
    a, b = np.asarray(a), np.asarray(b)
    ok = a.shape == b.shape and np.array_equal(a, b)
    md = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) if a.shape == b.shape and a.size else -1
    print("%-34s %s  shape=%s maxdiff=%g" % (name, "IDENTICAL" if ok else "DIFFERS", a.shape, md))
    return ok


def run(fast=False):
    ns = rh.import_reference()
    from PIL import Image
    torch.manual_seed(0)
    aff_sd, ori_sd = rh.load_state_dict("AffNet.pth"), rh.load_state_dict("OriNet.pth")
    hard_sd = orc.synthetic_hardnet_state(0)
    A = ns.architectures.AffNetFast(PS=32); A.load_state_dict(aff_sd); A.eval()
    O = ns.architectures.OriNetFast(PS=32); O.load_state_dict(ori_sd); O.eval()
    Hn = ns.HardNet.HardNet(); Hn.load_state_dict(hard_sd); Hn.eval()
    cases = [("synthetic 240x320", orc.synthetic_image(240, 320, 1), 300)]
    if not fast:
        img = np.mean(np.array(Image.open(os.path.join(rh.REF_ROOT, "test-graf/img1.png")).convert("RGB")), axis=2)
        cases.append(("graf img1", torch.from_numpy(img.astype(np.float32)).view(1, 1, *img.shape), 2000))
    ok = True
    for name, x, n in cases:
        print("== %s, N=%d" % (name, n))
        det = ns.SparseImgRepresenter.ScaleSpaceAffinePatchExtractor(
            mrSize=5.192, num_features=n, border=5, num_Baum_iters=1, AffNet=A, OriNet=O)
        with torch.no_grad(), rh.quiet():
            L, r = det(x, do_ori=True)
            P = det.extract_patches_from_pyr(L, PS=32)
            D = Hn(P)
        o = orc.OracleExtractor(mrSize=5.192, num_features=n, border=5, num_Baum_iters=1,
                                affnet_sd=aff_sd, orinet_sd=ori_sd)
        L2, r2, P2, D2 = orc.describe(x, o, hard_sd, do_ori=True, ps=32)
        for oi in range(len(det.scale_pyr)):
            for li in range(len(det.scale_pyr[oi])):
                ok &= np.array_equal(det.scale_pyr[oi][li].numpy(), o.scale_pyr[oi][li].numpy())
        print("pyramid identical so far:", ok)
        ok &= same("sigmas", np.array(det.sigmas), np.array(o.sigmas))
        ok &= same("LAFs", L.numpy(), L2.numpy())
        ok &= same("responses", r.numpy(), r2.numpy())
        ok &= same("patches", P.numpy(), P2.numpy())
        ok &= same("descriptors", D.numpy(), D2.numpy())
        # threshold mode (hesaffnet.py as shipped: th = -1 -> num = -1)
        if name.startswith("synthetic"):
            det = ns.SparseImgRepresenter.ScaleSpaceAffinePatchExtractor(
                mrSize=5.192, num_features=n, border=5, num_Baum_iters=1, th=-1, AffNet=A)
            with torch.no_grad(), rh.quiet():
                L, r = det(x)
            o = orc.OracleExtractor(mrSize=5.192, num_features=n, border=5, num_Baum_iters=1, th=-1, affnet_sd=aff_sd)
            L2, r2 = o(x)
            ok &= same("th=-1 LAFs", L.numpy(), L2.numpy())
            ok &= same("th=-1 responses", r.numpy(), r2.numpy())
            ok &= same("LAFs2ell", ns.LAF.LAFs2ell(L.numpy()), orc.lafs_to_ellipses(L2.numpy()))
    # constructor variants the mirror accepts: nlevels = 1 (35-tap Gaussian), init_sigma <= 0.5 (octave 0 unblurred, own blur sequence),
    # and the 1024x768 synthetic image of BASELINE configs[2] (the metric's configuration)
    x = orc.synthetic_image(240, 320, 1)
    for kw in (dict(nlevels=1), dict(init_sigma=0.4), dict(init_sigma=0.5, nlevels=2)):
        det = ns.SparseImgRepresenter.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=0, **kw)
        with torch.no_grad(), rh.quiet():
            L, r = det(x)
        o = orc.OracleExtractor(mrSize=5.192, num_features=300, border=5, num_Baum_iters=0, **kw)
        L2, r2 = o(x)
        pyr_ok = all(torch.equal(a, b) for oi in range(len(det.scale_pyr)) for a, b in zip(det.scale_pyr[oi], o.scale_pyr[oi]))
        print("== variant %s: pyramid identical: %s" % (kw, pyr_ok))
        ok &= pyr_ok and len(det.scale_pyr) == len(o.scale_pyr)
        ok &= same("  LAFs", L.numpy(), L2.numpy())
        ok &= same("  responses", r.numpy(), r2.numpy())
    if not fast:
        x = orc.synthetic_image(768, 1024, 1)
        print("== synthetic 768x1024 seed 1 (configs[2]), N=2000")
        det = ns.SparseImgRepresenter.ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=2000, border=5, num_Baum_iters=1, AffNet=A, OriNet=O)
        with torch.no_grad(), rh.quiet():
            L, r = det(x, do_ori=True)
            D = Hn(det.extract_patches_from_pyr(L, PS=32))
        o = orc.OracleExtractor(mrSize=5.192, num_features=2000, border=5, num_Baum_iters=1, affnet_sd=aff_sd, orinet_sd=ori_sd)
        L2, r2, P2, D2 = orc.describe(x, o, hard_sd, do_ori=True, ps=32)
        ok &= same("LAFs", L.numpy(), L2.numpy())
        ok &= same("responses", r.numpy(), r2.numpy())
        ok &= same("descriptors", D.numpy(), D2.numpy())
    # second opinion on the two CNN trunks: the reference's own TorchScript traces (convertJIT/*.pt), run live
    print("== AffNet / OriNet restatement vs the reference's TorchScript traces (raw outputs; float noise allowed: std_mean vs mean + std)")
    rp = torch.rand(64, 1, 32, 32, generator=torch.Generator().manual_seed(7)) * 255
    with torch.no_grad():
        for nm, got in (("AffNetJIT.pt", orc.affnet_raw(aff_sd, rp)), ("OriNetJIT.pt", orc.orinet_vector(ori_sd, rp))):
            d = float((rh.load_jit_trace(nm)(rp) - got).abs().max())
            print("  %-40s max abs diff %.3g %s" % (nm, d, "ok" if d < 2e-6 else "MISMATCH"))
            ok &= d < 2e-6
    # OnePassSIR (fully-convolutional AffNet once per octave) - oracle/onepass_oracle.py
    print("== OnePassSIR path (AffNetFastFullConv with the shipped AffNet.pth, border = 15 as in the reference's scripts)")
    x = orc.synthetic_image(240, 320, 1)
    FC = ns.architectures.AffNetFastFullConv(); FC.load_state_dict(aff_sd); FC.eval()
    with torch.no_grad():
        ok &= same("LocalNorm2d(33)", FC.lrn(x).numpy(), opo.local_norm2d(x).numpy())
        ok &= same("AffNetFastFullConv", FC(x).numpy(), opo.affnet_fullconv_forward(aff_sd, x).numpy())
    nms2 = ns.HandCraftedModules.NMS2d.__new__(ns.HandCraftedModules.NMS2d)      # as shipped the ctor raises under py3 (float padding)
    torch.nn.Module.__init__(nms2)
    nms2.MP, nms2.eps, nms2.th = torch.nn.MaxPool2d(3, stride=1, return_indices=False, padding=1), 1e-5, 0
    r = orc.hessian_response(orc.gaussian_blur(x, 1.6), 1.6)
    ok &= same("NMS2d (int padding shim)", nms2(r).numpy(), opo.nms2d(r).numpy())
    sir = rh.import_onepass_sir()
    cases = [(300, True, x), (5000, False, x)]
    if not fast:
        cases.append((2000, True, orc.synthetic_image(768, 1024, 1)))
    for n, do_ori, xx in cases:
        det = sir.OnePassSIR(mrSize=5.192, num_features=n, border=15, num_Baum_iters=1, AffNet=FC, OriNet=O)
        with torch.no_grad(), rh.quiet():
            L, r = det(xx, do_ori=do_ori)
        o = opo.OnePassOracle(mrSize=5.192, num_features=n, border=15, affnet_sd=aff_sd, orinet_sd=ori_sd)
        L2, r2 = o(xx, do_ori=do_ori)
        ok &= same("OnePassSIR %dx%d N=%d LAFs" % (xx.size(3), xx.size(2), n), L.numpy(), L2.numpy())
        ok &= same("OnePassSIR responses", r.numpy(), r2.numpy())
    print("ALL IDENTICAL" if ok else "MISMATCH")
    return ok


if __name__ == "__main__":
    sys.exit(0 if run("--fast" in sys.argv) else 1)
