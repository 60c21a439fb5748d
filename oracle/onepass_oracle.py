"""CPU oracle for the OnePassSIR path: fully-convolutional AffNet evaluated once per octave
instead of once per patch.

TEST INFRASTRUCTURE - NOT PRODUCT CODE (same rules as oracle/affnet_oracle.py: only tests/, smoke() and bench.py's
cpu_baseline leg may import it).

Restates, function by function (paths relative to the reference repo root):
  * LocalNorm2d                         architectures.py:21-31
  * AffNetFastFullConv.forward          architectures.py:629-674 (+ rectifyAffineTransformationUpIsUpFullyConv LAF.py:293-297)
  * NMS2d                               HandCraftedModules.py:194-206 (as shipped it raises under Python 3: `padding = kernel_size/2`
                                        is a float; restated with the integer padding Python 2 produced)
  * NMS3dAndComposeAAff.forward         HandCraftedModules.py:292-363 (+ sc_y_x_and_A2LAFs LAF.py:442-449)
  * OnePassSIR.multiScaleDetectorAff / getOrientation / forward        OnePassSIR.py:53-153

Pinning: oracle/check_restatement.py compares every function here bit for bit with the reference's importable classes
(AffNetFastFullConv, NMS3dAndComposeAAff, LocalNorm2d) and with OnePassSIR itself, whose source is executed IN MEMORY with its one
Python-2 statement (`print time.time() - t, ...`, OnePassSIR.py:144) rewritten - the same kind of one-line shim the Baumberg
estimator needed.  The reference ships no weights for a fully-convolutional AffNet (its scripts import a non-existent
`AffNetFastFullAff`); AffNetFastFullConv has exactly AffNetFast's `features` layout, so the shipped AffNet.pth loads into it
unchanged - that is the weight set used here and in the tests.
"""
import torch
import torch.nn.functional as F

import affnet_oracle as orc


def local_norm2d(x, ks=33):
    """architectures.py:21-31: (x - mean_33x33) / (sqrt(|E[x^2] - mean^2|) + 1e-10), reflect padding, clamped to [-6, 6]."""
    pd = int(ks / 2)
    mean = F.avg_pool2d(F.pad(x, (pd, pd, pd, pd), "reflect"), ks, stride=1, padding=0)
    sq = F.avg_pool2d(F.pad(x * x, (pd, pd, pd, pd), "reflect"), ks, stride=1, padding=0)
    return torch.clamp((x - mean) / (torch.sqrt(torch.abs(sq - mean * mean)) + 1e-10), min=-6.0, max=6.0)


def rectify_up_is_up_fully_conv(A):
    """LAF.py:293-297; A is (n,4,h,w)."""
    det = torch.sqrt(torch.abs(A[:, 0:1] * A[:, 3:4] - A[:, 1:2] * A[:, 2:3] + 1e-10))
    b2a2 = torch.sqrt(A[:, 1:2] * A[:, 1:2] + A[:, 0:1] * A[:, 0:1])
    return torch.cat([(b2a2 / det).contiguous(), 0 * det.contiguous(),
                      (A[:, 3:4] * A[:, 1:2] + A[:, 2:3] * A[:, 0:1]) / (b2a2 * det), (det / b2a2).contiguous()], dim=1)


def affnet_fullconv_features(sd, x):
    """The dense trunk + 8x8 head on the locally normalised, reflect-padded image (architectures.py:632-653,667-668).
    Returns (norm_inp, ff) - ff is (1,3,h/4-ish,w/4-ish)."""
    norm_inp = local_norm2d(x, 33)
    y = F.pad(norm_inp, (14, 14, 14, 14), "reflect")
    y = orc.cnn_trunk(sd, y)
    return norm_inp, F.conv2d(y, sd["features.19.weight"], sd["features.19.bias"])


def affnet_fullconv_forward(sd, x):
    """architectures.py:666-674: (1,1,H,W) image -> (1,4,H,W) per-pixel rectified affine shape (a11, 0, a21, a22)."""
    _, ff = affnet_fullconv_features(sd, x)
    xy = torch.tanh(F.interpolate(ff, size=(x.size(2), x.size(3)), mode="bilinear", align_corners=False))
    a0bc = torch.cat([1.0 + xy[:, 0:1].contiguous(), 0 * xy[:, 1:2].contiguous(), xy[:, 1:2].contiguous(), 1.0 + xy[:, 2:].contiguous()],
                     dim=1).contiguous()
    return rectify_up_is_up_fully_conv(a0bc).contiguous()


def nms2d(x, threshold=0.0, kernel_size=3):
    """HandCraftedModules.py:194-206 with padding = kernel_size // 2 (the Python-2 value of `kernel_size/2`)."""
    mp = F.max_pool2d(x, kernel_size, stride=1, padding=kernel_size // 2)
    eps = 1e-5
    if threshold > eps:
        return x * (x > threshold).float() * ((x + eps - mp) > 0).float()
    return ((x - mp + eps) > 0).float() * x


def nms3d_compose_aff(low, cur, high, num_features, octave_map, scales, mr_size, aff_resp):
    """HandCraftedModules.py:292-363 (NMS3dAndComposeAAff.forward): identical to NMS3dAndComposeA up to the LAF composition,
    which scales the per-pixel A matrix of the octave's affine map by the centroid scale (sc_y_x_and_A2LAFs, LAF.py:442-449)."""
    resp, lafs, om, idx = orc.nms3d_compose(low, cur, high, num_features, octave_map, scales, mr_size)
    if resp is None:
        return None, None, None, None
    A = aff_resp.view(4, -1).t()[idx, :].view(-1, 2, 2)
    s = lafs[:, 0, 0].clone()
    out = torch.cat([s.view(-1, 1, 1).expand_as(A) * A, lafs[:, :, 2:]], dim=2)
    return resp, out, om, idx


class OnePassOracle(object):
    """OnePassSIR (OnePassSIR.py:14-153) with RespNet = HessianResp, AffNet = AffNetFastFullConv state dict, OriNet = OriNetFast
    state dict (None: the default OrientationDetector(19))."""

    def __init__(self, border=16, num_features=500, mrSize=3.0, nlevels=3, init_sigma=1.6, th=None, affnet_sd=None, orinet_sd=None):
        self.mrSize, self.b, self.num = mrSize, border, num_features
        self.nlevels, self.init_sigma = nlevels, init_sigma
        self.th = th
        if th is not None:                      # OnePassSIR.py:31-34
            self.num = -1
        else:
            self.th = 0
        self.aff, self.ori = affnet_sd, orinet_sd
        self.scale_pyr = self.sigmas = self.pix_dists = None
        self.keys = None
        self.aff_maps = None

    def multi_scale_detector_aff(self, x, num_features):
        """OnePassSIR.py:53-115."""
        pyr, sigmas, dists = orc.scale_pyramid(x, self.nlevels, self.init_sigma, self.b)
        self.scale_pyr, self.sigmas, self.pix_dists = pyr, sigmas, dists
        self.aff_maps = []
        resp_l, laf_l, oct_l, lev_l, pix_l = [], [], [], [], []
        for o, levels in enumerate(pyr):
            omap = (levels[0] * 0).byte()
            amap = affnet_fullconv_forward(self.aff, levels[0])                      # :69 AffNet(octave[0])
            self.aff_maps.append(amap)
            rmaps = [torch.clamp(orc.hessian_response(levels[l], sigmas[o][l]) - self.th, min=0) for l in range(len(levels))]
            for l in range(1, len(levels) - 1):
                r, lafs, om, pix = nms3d_compose_aff(rmaps[l - 1], rmaps[l], rmaps[l + 1], num_features, omap, sigmas[o][l - 1:l + 2],
                                                     self.mrSize, amap)
                if r is None:
                    continue
                omap = om
                ok = orc.inside_image(torch.cat([lafs[:, :2, :2] * 3.0, lafs[:, :, 2:]], dim=2))      # :91 (the 3.0 is hard-coded)
                resp_l.append(r[ok])
                laf_l.append(lafs[ok])
                oct_l.append(torch.full((int(ok.sum()),), float(o)))
                lev_l.append(torch.full((int(ok.sum()),), float(l - 1)))
                pix_l.append(pix[ok].long())
        resp, lafs, octs, levs, pixs = torch.cat(resp_l), torch.cat(laf_l), torch.cat(oct_l), torch.cat(lev_l), torch.cat(pix_l)
        if 0 < num_features < resp.numel():
            resp, sel = torch.topk(resp, k=num_features)
            lafs, octs, levs, pixs = lafs[sel], octs[sel], levs[sel], pixs[sel]
        return resp, lafs, octs, levs, pixs

    def forward(self, x, do_ori=True):
        """OnePassSIR.py:139-153.  Returns (LAFs px (N,2,3), responses (N,))."""
        with torch.no_grad():
            resp, lafs, octs, levs, pixs = self.multi_scale_detector_aff(x, self.num)
            lafs = lafs.clone()
            lafs[:, 0:2, 0:2] = self.mrSize * lafs[:, :, 0:2]
            self.detected = {"resp": resp.clone(), "lafs": lafs.clone()}
            if do_ori:                                                             # :116-129 (getOrientation)
                ps = 32 if self.ori is not None else 19
                patches = orc.extract_from_pyramid(self.scale_pyr, octs, levs, lafs, ps)
                R = orc.orinet_forward(self.ori, patches) if self.ori is not None else orc.angles_to_rotation(orc.orientation_detector(patches))
                lafs = torch.cat([torch.bmm(lafs[:, :, :2], R), lafs[:, :, 2:]], dim=2)
            self.keys = torch.stack([octs.long(), levs.long(), pixs.long()], dim=1)
            return orc.denormalize_lafs(lafs, x.size(3), x.size(2)), resp

    __call__ = forward
    # descriptor patches exactly as in the patch-based extractor (OnePassSIR.py:130-138 == SparseImgRepresenter.py:181-188)
    level_for_lafs = orc.OracleExtractor.level_for_lafs
    extract_patches_from_pyr = orc.OracleExtractor.extract_patches_from_pyr
