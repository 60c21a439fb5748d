"""CPU oracle for the ScaleSpaceAffinePatchExtractor hot path.

TEST INFRASTRUCTURE - NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import this file.  The product path
(affnet_amd/) never imports it and raises if the HIP library is missing.

What it is: a functional restatement (torch CPU fp32 ops + numpy) of the reference
algorithm, one function per reference function, each citing the reference file:line
it follows (paths relative to the reference repo root).  The same ATen CPU operators
are used in the same order as the reference, so on the same host the restatement is
bit-identical to the reference run under Python 3 / torch 2.x (py3 true-division
semantics, align_corners=False, CPU float->uint8 wrap: SURVEY.md Appendix A).

Pinning: tests/golden/*.npz were produced by tests/golden/make_golden.py, which runs
the UNMODIFIED reference (oracle/ref_harness.py) in the authoring container;
tests/test_oracle_golden.py checks this file against them, and
oracle/check_restatement.py re-checks bit-identity whenever /root/reference is
present.  The reference repo itself ships no expected-value fixtures for this path
(SURVEY.md section 4), and HardNet++.pth is absent, so HardNet is pinned on seeded synthetic
weights only.

Every stage additionally returns integer keys (octave, level, flat pixel index) for
each keypoint so that parity tests can match keypoints exactly instead of by
nearest-neighbour.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------
# Gaussian scale pyramid
# ----------------------------------------------------------------------------------------


def gauss_kernel_2d(sigma):
    """k x k float32 Gaussian, sum-normalised in float64.  Utils.py:92-114 (CircularGaussKernel
    with kernlen=None, circ_zeros=False, norm=True) and Utils.py:155-161 (calculate_weights).
    Under Python 3 `kernlen / 2` is a true division (5.5 for 11 taps)."""
    klen = int(2.0 * 3.0 * sigma + 1.0)
    if klen % 2 == 0:
        klen += 1
    half = klen / 2
    ax = np.linspace(-half, half, klen)
    gx, gy = np.meshgrid(ax, ax, sparse=False, indexing="xy")
    ker = np.exp(-((gx ** 2 + gy ** 2) / (2.0 * sigma * sigma)))
    ker /= np.sum(ker)
    pad = int(np.floor(float(klen) / 2.0))
    return ker.astype(np.float32), pad


def gaussian_blur(x, sigma):
    """Replicate-pad + full 2-D cross-correlation.  Utils.py:162-166."""
    ker, pad = gauss_kernel_2d(sigma)
    w = torch.from_numpy(ker).view(1, 1, ker.shape[0], ker.shape[1])
    return F.conv2d(F.pad(x, (pad, pad, pad, pad), "replicate"), w, padding=0)


def pyramid_plan(height, width, n_levels=3, init_sigma=1.6, border=5):
    """Octave sizes, blur sigmas, level sigmas and pixel distances without touching pixels.
    HandCraftedModules.py:14-21 (ctor) and :23-56 (loop structure / stop rule)."""
    step = 2 ** (1.0 / float(n_levels))
    min_size = 2 * border + 2 + 1
    plan = {"first_blur": None, "octaves": []}
    cur_sigma = 0.5
    if init_sigma > cur_sigma:
        plan["first_blur"] = float(np.sqrt(init_sigma ** 2 - cur_sigma ** 2))
        cur_sigma = init_sigma
    h, w = int(height), int(width)
    pix = 1.0
    while True:
        lev_sigmas = [cur_sigma]
        blur_sigmas = []
        for _ in range(1, n_levels + 2):
            blur_sigmas.append(float(cur_sigma * np.sqrt(step * step - 1.0)))
            cur_sigma *= step
            lev_sigmas.append(cur_sigma)
        plan["octaves"].append({"h": h, "w": w, "pix_dist": pix, "blur_sigmas": blur_sigmas,
                                "level_sigmas": lev_sigmas})
        nh, nw = (h - 1) // 2 + 1, (w - 1) // 2 + 1  # avg_pool2d(k=1, s=2, p=0) output size
        pix *= 2.0
        cur_sigma = init_sigma
        if nh <= min_size or nw <= min_size:
            break
        h, w = nh, nw
    return plan


def scale_pyramid(x, n_levels=3, init_sigma=1.6, border=5):
    """HandCraftedModules.py:23-56 (ScalePyramid.forward).  Returns (pyr, sigmas, pix_dists)
    as nested python lists exactly like the reference."""
    plan = pyramid_plan(x.size(2), x.size(3), n_levels, init_sigma, border)
    cur = gaussian_blur(x, plan["first_blur"]) if plan["first_blur"] is not None else x
    pyr, sigmas, dists = [], [], []
    for oi, octv in enumerate(plan["octaves"]):
        levels = [cur]
        nxt = None
        for i, bs in enumerate(octv["blur_sigmas"], start=1):
            cur = gaussian_blur(cur, bs)
            levels.append(cur)
            if i == n_levels:
                nxt = F.avg_pool2d(cur, kernel_size=1, stride=2, padding=0)
        pyr.append(levels)
        sigmas.append(list(octv["level_sigmas"]))
        dists.append([octv["pix_dist"]] * len(levels))
        cur = nxt
    return pyr, sigmas, dists


# ----------------------------------------------------------------------------------------
# Hessian response, 3-D NMS, centroid -> LAF
# ----------------------------------------------------------------------------------------

_K_GX = torch.tensor([[[[0.5, 0.0, -0.5]]]], dtype=torch.float32)
_K_GY = _K_GX.permute(0, 1, 3, 2).contiguous()
_K_GXX = torch.tensor([[[[1.0, -2.0, 1.0]]]], dtype=torch.float32)
_K_GYY = _K_GXX.permute(0, 1, 3, 2).contiguous()


def hessian_response(x, sigma):
    """|gxx*gyy - gxy^2| * sigma^4 with replicate borders.  HandCraftedModules.py:62-78."""
    gxx = F.conv2d(F.pad(x, (1, 1, 0, 0), "replicate"), _K_GXX)
    gyy = F.conv2d(F.pad(x, (0, 0, 1, 1), "replicate"), _K_GYY)
    gx = F.conv2d(F.pad(x, (1, 1, 0, 0), "replicate"), _K_GX)
    gxy = F.conv2d(F.pad(gx, (0, 0, 1, 1), "replicate"), _K_GY)
    return torch.abs(gxx * gyy - gxy * gxy) * (sigma ** 4)


def zero_border(x, b):
    """Utils.py:140-148 (zero_response_at_border)."""
    if b < x.size(3) and b < x.size(2):
        x[:, :, 0:b, :] = 0
        x[:, :, x.size(2) - b:, :] = 0
        x[:, :, :, 0:b] = 0
        x[:, :, :, x.size(3) - b:] = 0
        return x
    return x * 0


def centroid_weights(scales):
    """(3,3,3,3) weights of the response-weighted centroid.  Utils.py:116-138
    (generate_2dgrid/generate_3dgrid, centered=True; py3: offsets -0.5, 0.5, 1.5) and
    HandCraftedModules.py:266-270."""
    off = torch.linspace(-3 / 2 + 1, 3 / 2, 3)
    g = torch.zeros(3, 3, 3, 3)
    g[0] = torch.tensor(scales, dtype=torch.float32).view(3, 1, 1).expand(3, 3, 3)
    g[1] = off.view(1, 3, 1).expand(3, 3, 3)
    g[2] = off.view(1, 1, 3).expand(3, 3, 3)
    return g


def nms3d_compose(low, cur, high, num_features, octave_map, scales, mr_size):
    """One detection level.  HandCraftedModules.py:240-291 (NMS3dAndComposeA.forward),
    NMS3d :215-220, sc_y_x2LAFs LAF.py:431-441.
    Returns (responses (k,), LAFs (k,2,3) normalised, new octave_map, flat pixel idx (k,))
    or (None, None, None, None) when <= 1 positive response survives."""
    h, w = cur.size(2), cur.size(3)
    r3 = torch.cat([low, cur, high], dim=1)
    vol = r3.unsqueeze(1)
    pooled = F.max_pool3d(vol, 3, stride=1, padding=(0, 1, 1))
    kept = (((vol - pooled + 1e-5) > 0).float() * vol).squeeze(1)[:, 1:2, :, :]
    v = zero_border(kept, int(mr_size)) * (1.0 - octave_map.float())
    n_pos = (v > 0).float().sum().item()
    if n_pos <= 1:
        return None, None, None, None
    octave_map = (octave_map.float() + v.float()).byte()
    flat = v.view(-1)
    if 0 < num_features < n_pos:
        resp, idx = torch.topk(flat, k=num_features, dim=0)
    else:
        idx = flat.nonzero().squeeze()
        resp = flat[idx]
    wts = centroid_weights(scales)
    syx = F.conv2d(r3, wts, padding=1) / (F.conv2d(r3, torch.ones(3, 3, 3, 3), padding=1) + 1e-8)
    yy = torch.linspace(0, h - 1, h).view(h, 1).expand(h, w)
    xx = torch.linspace(0, w - 1, w).view(1, w).expand(h, w)
    syx[0, 1] = syx[0, 1] + yy
    syx[0, 2] = syx[0, 2] + xx
    syx = syx.view(3, -1).t()[idx, :]
    syx[:, 0] = syx[:, 0] / float(min(h, w))
    syx[:, 1] = syx[:, 1] / float(h)
    syx[:, 2] = syx[:, 2] / float(w)
    k = syx.size(0)
    lafs = torch.zeros(k, 2, 3)
    lafs[:, 0, 0] = syx[:, 0]
    lafs[:, 1, 1] = syx[:, 0]
    lafs[:, 0, 2] = syx[:, 2]
    lafs[:, 1, 2] = syx[:, 1]
    return resp, lafs, octave_map, idx.view(-1)


def multi_scale_detector(x, num_features, n_levels=3, init_sigma=1.6, border=5, mr_size=3.0, th=0.0, resp_fn=None):
    """SparseImgRepresenter.py:53-111.  Returns dict with pyramid and candidate arrays.  resp_fn = the RespNet slot
    (:38-41): callable(level (1,1,h,w), sigma) -> (1,1,h,w); None = HessianResp."""
    hessian = hessian_response if resp_fn is None else resp_fn
    pyr, sigmas, dists = scale_pyramid(x, n_levels, init_sigma, border)
    resp_l, laf_l, oct_l, lev_l, pix_l = [], [], [], [], []
    for o, levels in enumerate(pyr):
        omap = (levels[0] * 0).byte()
        rmaps = [torch.clamp(hessian(levels[l], sigmas[o][l]) - th, min=0) for l in range(len(levels))]
        for l in range(1, len(levels) - 1):
            r, lafs, om, pix = nms3d_compose(rmaps[l - 1], rmaps[l], rmaps[l + 1], num_features, omap,
                                             sigmas[o][l - 1:l + 2], mr_size)
            if r is None:
                continue
            omap = om
            resp_l.append(r)
            laf_l.append(lafs)
            oct_l.append(torch.full((r.numel(),), float(o)))
            lev_l.append(torch.full((r.numel(),), float(l - 1)))  # "prevBlur": SparseImgRepresenter.py:94
            pix_l.append(pix.long())
    resp = torch.cat(resp_l)
    lafs = torch.cat(laf_l)
    octs = torch.cat(oct_l)
    levs = torch.cat(lev_l)
    pixs = torch.cat(pix_l)
    if 0 < num_features < resp.numel():
        resp, sel = torch.topk(resp, k=num_features)
        lafs, octs, levs, pixs = lafs[sel], octs[sel], levs[sel], pixs[sel]
    return {"pyr": pyr, "sigmas": sigmas, "pix_dists": dists, "resp": resp, "lafs": lafs,
            "oct": octs, "lev": levs, "pix": pixs}


# ----------------------------------------------------------------------------------------
# LAF algebra + patch sampler
# ----------------------------------------------------------------------------------------


def laf_scale_coef(w, h, inverse=False):
    """LAF.py:407-429 (denormalizeLAFs / normalizeLAFs)."""
    w, h = float(w), float(h)
    m = min(h, w)
    c = torch.ones(1, 2, 3) * m
    c[0, 0, 2] = w
    c[0, 1, 2] = h
    if inverse:
        c = torch.ones(1, 2, 3) / m
        c[0, 0, 2] = 1.0 / w
        c[0, 1, 2] = 1.0 / h
    return c


def denormalize_lafs(lafs, w, h):
    return laf_scale_coef(w, h).expand(lafs.size(0), 2, 3) * lafs


def normalize_lafs(lafs, w, h):
    return laf_scale_coef(w, h, inverse=True).expand(lafs.size(0), 2, 3) * lafs


def extract_patches(img, lafs, ps):
    """affine_grid + grid_sample (bilinear, zeros padding, align_corners=False).
    LAF.py:313-324 (grid), :326-362 (chunks of 32 - chunking does not change values),
    :364-372."""
    h, w = img.size(2), img.size(3)
    n = lafs.size(0)
    if n == 0:
        return torch.zeros(0, img.size(1), ps, ps)
    theta = lafs * laf_scale_coef(float(w), float(h)).expand(n, 2, 3)
    grid = F.affine_grid(theta, torch.Size((n, 1, ps, ps)), align_corners=False)
    grid[:, :, :, 0] = 2.0 * grid[:, :, :, 0] / float(w) - 1.0
    grid[:, :, :, 1] = 2.0 * grid[:, :, :, 1] / float(h) - 1.0
    out = torch.zeros(n, img.size(1), ps, ps)
    for s in range(0, n, 32):
        e = min(n, s + 32)
        out[s:e] = F.grid_sample(img.expand(e - s, img.size(1), h, w), grid[s:e], mode="bilinear",
                                 padding_mode="zeros", align_corners=False)
    return out


def extract_from_pyramid(pyr, octs, levs, lafs, ps):
    """LAF.py:376-404 (inverted index + per-level extraction)."""
    out = torch.zeros(lafs.size(0), 1, ps, ps)
    for o in range(len(pyr)):
        for l in range(len(pyr[o])):
            sel = torch.nonzero((octs == o) & (levs == l)).view(-1)
            if sel.numel() == 0:
                continue
            out[sel] = extract_patches(pyr[o][l], lafs[sel], ps)
    return out


def rectify_up_is_up(A):
    """LAF.py:285-291."""
    det = torch.sqrt(torch.abs(A[:, 0, 0] * A[:, 1, 1] - A[:, 1, 0] * A[:, 0, 1] + 1e-10))
    b2a2 = torch.sqrt(A[:, 0, 1] * A[:, 0, 1] + A[:, 0, 0] * A[:, 0, 0])
    r0 = torch.stack([b2a2 / det, 0 * det], dim=1)
    r1 = torch.stack([(A[:, 1, 1] * A[:, 0, 1] + A[:, 1, 0] * A[:, 0, 0]) / (b2a2 * det), det / b2a2], dim=1)
    return torch.stack([r0, r1], dim=1)


def eig2x2(A):
    """Utils.py:168-175 (batch_eig2x2)."""
    tr = A[:, 0, 0] + A[:, 1, 1]
    d1 = tr * tr - 4 * (A[:, 0, 0] * A[:, 1, 1] - A[:, 1, 0] * A[:, 0, 1])
    ok = (d1 > 0).float()
    d = torch.sqrt(torch.abs(d1))
    l1 = ok * (tr + d) / 2.0 + 1000.0 * (1.0 - ok)
    l2 = ok * (tr - d) / 2.0 + 0.0001 * (1.0 - ok)
    return l1, l2


def frame_corners(lafs):
    """LAF.py:91-104 (LAFs_to_H_frames + the product checkTouchBoundary thresholds): the four corners (+-1, +-1) of every frame in the
    LAFs' own (normalised) units, (n, 2, 4); dtype follows the input."""
    pts = torch.tensor([[-1.0, -1, 1, 1], [-1, 1, -1, 1], [1, 1, 1, 1]], dtype=lafs.dtype).unsqueeze(0)
    n = lafs.size(0)
    Hm = torch.cat([lafs, torch.tensor([0.0, 0, 1], dtype=lafs.dtype).view(1, 1, 3).repeat(n, 1, 1)], dim=1)
    return torch.bmm(Hm, pts.expand(n, 3, 4))[:, :2, :]


def inside_image(lafs):
    """LAF.py:91-104 (LAFs_to_H_frames + checkTouchBoundary) on normalised LAFs."""
    out = frame_corners(lafs)
    return ~(((out > 1.0).int() + (out < 0.0).int()).sum(dim=1).sum(dim=1) > 0)


def rotation_matrix(angle):
    """LAF.py:276-283."""
    a = angle.view(-1, 1, 1)
    s, c = torch.sin(a), torch.cos(a)
    return torch.cat([torch.cat([c, s], dim=2), torch.cat([-s, c], dim=2)], dim=1)


def lafs_to_ellipses(lafs):
    """LAF.py:225-240 (LAFs2ell): per-row numpy SVD -> Oxford ellipse (x y a b c).  The input
    dtype is preserved for the arithmetic (float32 in -> float32 SVD), the result array is
    float64, exactly like the reference."""
    lafs = np.asarray(lafs)
    lafs = lafs.reshape(-1, 2, 3)
    ell = np.zeros((len(lafs), 5))
    for i in range(len(lafs)):
        L = lafs[i].copy()
        sc = np.sqrt(L[0, 0] * L[1, 1] - L[0, 1] * L[1, 0] + 1e-10)
        u, W, _ = np.linalg.svd(L[0:2, 0:2] / sc, full_matrices=True)
        W[0] = 1.0 / (W[0] * W[0] * sc * sc)
        W[1] = 1.0 / (W[1] * W[1] * sc * sc)
        A = np.matmul(np.matmul(u, np.diag(W)), u.transpose())
        ell[i, 0], ell[i, 1] = L[0, 2], L[1, 2]
        ell[i, 2], ell[i, 3], ell[i, 4] = A[0, 0], A[0, 1], A[1, 1]
    return ell


# ----------------------------------------------------------------------------------------
# CNNs (state-dict driven, eval mode)
# ----------------------------------------------------------------------------------------


def input_norm(x):
    """architectures.py:235-239 / HardNet.py:92-96: per-patch mean / unbiased std + 1e-7."""
    flat = x.view(x.size(0), -1)
    mp = torch.mean(flat, dim=1)
    sp = torch.std(flat, dim=1) + 1e-7
    return (x - mp.view(-1, 1, 1, 1)) / sp.view(-1, 1, 1, 1)


_TRUNK = [(0, 1, 1), (3, 4, 1), (6, 7, 2), (9, 10, 1), (12, 13, 2), (15, 16, 1)]  # conv idx, bn idx, stride


def cnn_trunk(sd, x):
    """Six conv3x3(pad 1, no bias) + BatchNorm(eval, affine=False, eps 1e-5) + ReLU blocks.
    architectures.py:207-224 == :36-53, HardNet.py:67-84.  Dropout is identity in eval."""
    for ci, bi, st in _TRUNK:
        x = F.conv2d(x, sd["features.%d.weight" % ci], None, stride=st, padding=1)
        x = F.batch_norm(x, sd["features.%d.running_mean" % bi], sd["features.%d.running_var" % bi],
                         None, None, False, 0.1, 1e-5)
        x = F.relu(x)
    return x


def affnet_forward(sd, patches):
    """architectures.py:204-252 (AffNetFast.forward)."""
    y = cnn_trunk(sd, input_norm(patches))
    y = torch.tanh(F.conv2d(y, sd["features.19.weight"], sd["features.19.bias"]))
    xy = F.adaptive_avg_pool2d(y, 1).view(-1, 3)
    n = xy.size(0)
    A = torch.zeros(n, 2, 2)
    A[:, 0, 0] = 1.0 + xy[:, 0]
    A[:, 1, 0] = xy[:, 1]
    A[:, 1, 1] = 1.0 + xy[:, 2]
    return rectify_up_is_up(A)


def affnet_raw(sd, patches):
    """architectures.py:227-229,240-246: the three numbers AffNetFast builds its matrix from, (1 + x0, x1, 1 + x2) = (a11, a21, a22) BEFORE
    rectifyAffineTransformationUpIsUp.  This is what the reference's own TorchScript trace convertJIT/AffNetJIT.pt returns
    (convertJIT/convert_OriNet_and_AffNet_to_JIT.ipynb), i.e. the quantity a second, independently produced CNN oracle can be compared on."""
    y = cnn_trunk(sd, input_norm(patches))
    y = torch.tanh(F.conv2d(y, sd["features.19.weight"], sd["features.19.bias"]))
    xy = F.adaptive_avg_pool2d(y, 1).view(-1, 3)
    return xy + torch.tensor([1.0, 0.0, 1.0])


def affnet_batched(sd, patches, bs=256):
    """Utils.py:37-66 (batched_forward, chunks of 256)."""
    n = patches.size(0)
    if n <= bs:
        return affnet_forward(sd, patches)
    out = torch.zeros(n, 2, 2)
    for s in range(0, n, bs):
        e = min(n, s + bs)
        out[s:e] = affnet_forward(sd, patches[s:e])
    return out


def orinet_vector(sd, patches):
    """architectures.py:33-80: the (n, 2) mean vector OriNetFast feeds to atan2 (head conv 8x8 padding 1, tanh, average pool).
    Its length says how well conditioned the angle is: the parity report records it for LAF rows outside 1e-3 px."""
    y = cnn_trunk(sd, input_norm(patches))
    y = torch.tanh(F.conv2d(y, sd["features.19.weight"], sd["features.19.bias"], padding=1))
    return F.adaptive_avg_pool2d(y, 1).view(-1, 2)


def orinet_forward(sd, patches, return_rot=True, keep=None):
    """architectures.py:33-82 (OriNetFast.forward).  keep: optional dict that receives the pre-atan2 vector under "vec"."""
    xy = orinet_vector(sd, patches)
    if keep is not None:
        keep["vec"] = xy.clone()
    ang = torch.atan2(xy[:, 0] + 1e-8, xy[:, 1] + 1e-8)
    return rotation_matrix(ang) if return_rot else ang


def hardnet_forward(sd, patches):
    """HardNet.py:61-101; L2Norm :12-19 (eps 1e-8)."""
    y = cnn_trunk(sd, input_norm(patches))
    y = F.conv2d(y, sd["features.19.weight"], None)
    y = F.batch_norm(y, sd["features.20.running_mean"], sd["features.20.running_var"], None, None, False, 0.1, 1e-5)
    y = y.view(y.size(0), -1)
    nrm = torch.sqrt(torch.sum(y * y, dim=1) + 1e-8)
    return y / nrm.unsqueeze(-1)


def synthetic_hardnet_state(seed=0):
    """HardNet++.pth is a missing blob (.MISSING_LARGE_BLOBS).  Deterministic stand-in weights
    (SURVEY.md section 8d): seeded normal conv weights scaled ~He, BN running_mean ~ U(-.3,.3),
    running_var ~ U(.2,.6).  Written with an explicit generator so it does not depend on
    nn.Module default-init internals."""
    g = torch.Generator().manual_seed(seed)
    widths = [(1, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128)]
    sd = {}
    for (ci, bi, _), (cin, cout) in zip(_TRUNK, widths):
        sd["features.%d.weight" % ci] = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (cin * 9))
        sd["features.%d.running_mean" % bi] = torch.rand(cout, generator=g) * 0.6 - 0.3
        sd["features.%d.running_var" % bi] = torch.rand(cout, generator=g) * 0.4 + 0.2
    sd["features.19.weight"] = torch.randn(128, 128, 8, 8, generator=g) * math.sqrt(1.0 / (128 * 64))
    sd["features.20.running_mean"] = torch.rand(128, generator=g) * 0.6 - 0.3
    sd["features.20.running_var"] = torch.rand(128, generator=g) * 0.4 + 0.2
    return sd


# ----------------------------------------------------------------------------------------
# Pipeline
# ----------------------------------------------------------------------------------------


class OracleExtractor(object):
    """Restates ScaleSpaceAffinePatchExtractor (SparseImgRepresenter.py:14-209) for the
    configuration the hot path uses: RespNet = HessianResp, AffNet = AffNetFast state dict, OriNet = OriNetFast state
    dict.  Without a state dict the reference's default slot fillers are used (SparseImgRepresenter.py:42-49):
    OrientationDetector(patch_size=19) and AffineShapeEstimator(patch_size=19) - the latter called as
    `AffNet(patches)`: as shipped, batched_forward passes a kwargs dict positionally (Utils.py:54) which
    AffineShapeEstimator.forward(self, x) rejects with a TypeError, so the Baumberg path is pinned against the reference
    class behind a one-line `forward(self, x, *ignored)` shim (tests/golden/make_golden_handcrafted.py)."""

    def __init__(self, border=16, num_features=500, patch_size=32, mrSize=3.0, nlevels=3,
                 num_Baum_iters=0, init_sigma=1.6, th=None, affnet_sd=None, orinet_sd=None,
                 reproduce_wasted_extraction=False, resp_fn=None):
        self.mrSize, self.b, self.num = mrSize, border, num_features
        self.nlevels, self.iters, self.init_sigma = nlevels, num_Baum_iters, init_sigma
        self.th = th
        if th is not None:  # SparseImgRepresenter.py:33-37
            self.num = -1
        else:
            self.th = 0
        self.aff, self.ori = affnet_sd, orinet_sd
        self.resp_fn = resp_fn
        self.PS = 32
        self.waste = reproduce_wasted_extraction
        self.scale_pyr = self.sigmas = self.pix_dists = None
        self.keys = None

    def _affine_shape(self, det, n_out):
        """SparseImgRepresenter.py:113-165 for num_Baum_iters == 1 ... k."""
        resp, lafs, octs, levs, pixs = det["resp"], det["lafs"], det["oct"], det["lev"], det["pix"]
        ps = self.PS if self.aff is not None else 19
        shape = (lambda p: affnet_batched(self.aff, p, 256)) if self.aff is not None else affine_shape_estimator
        patches = extract_from_pyramid(self.scale_pyr, octs, levs, lafs, ps)
        base = torch.eye(2).unsqueeze(0).expand(lafs.size(0), 2, 2)
        new = lafs
        for i in range(self.iters):
            A = shape(patches)
            base = torch.bmm(A, base)
            new = torch.cat([torch.bmm(base, lafs[:, :, 0:2]), lafs[:, :, 2:]], dim=2)
            if i != self.iters - 1:
                patches = extract_from_pyramid(self.scale_pyr, octs, levs, new, ps)
        l1, l2 = eig2x2(base)
        ratio = torch.abs(l1 / (l2 + 1e-8))
        good = ((ratio < 6.0) & (ratio > (1.0 / 6.0))) & inside_image(new)
        # what the hard decisions of :147-162 were taken on, per CANDIDATE (row order of self.detected) - read by the parity tests to trace every
        # keypoint that only one side returns to the decision that differs (oracle/fp64_referee.py); no effect on the results
        self.shape_stage = {"A": base.clone(), "frames": new.clone(), "ratio": ratio.clone(), "good": good.clone(), "n_out": n_out}
        if n_out > 0 and good.float().sum().item() > n_out:
            resp, sel = torch.topk(resp * good.float(), k=n_out)
        else:
            sel = torch.nonzero(good).view(-1)
            resp = resp[sel]
        base, lafs = base[sel], lafs[sel]
        new = torch.cat([torch.bmm(base, lafs[:, :, 0:2]), lafs[:, :, 2:]], dim=2)
        return resp, new, octs[sel], levs[sel], pixs[sel]

    def _orientation(self, lafs, octs, levs):
        """SparseImgRepresenter.py:167-180."""
        ps = self.PS if self.ori is not None else 19
        patches = extract_from_pyramid(self.scale_pyr, octs, levs, lafs, ps)
        keep = {}
        R = orinet_forward(self.ori, patches, keep=keep) if self.ori is not None else angles_to_rotation(orientation_detector(patches))
        self.ori_vec = keep.get("vec")           # (n, 2) OriNet output before atan2, row order of the returned LAFs (None: hand-crafted detector)
        lafs = torch.cat([torch.bmm(lafs[:, :, :2], R), lafs[:, :, 2:]], dim=2)
        if self.waste:  # :178-179, result discarded by the reference
            extract_from_pyramid(self.scale_pyr, octs, levs, lafs, ps)
        return lafs

    def forward(self, x, do_ori=False):
        """SparseImgRepresenter.py:189-209.  Returns (LAFs px (N,2,3), responses (N,))."""
        with torch.no_grad():
            pre = int(1.5 * self.num) if self.iters > 0 else self.num
            det = multi_scale_detector(x, pre, self.nlevels, self.init_sigma, self.b, self.mrSize, self.th, self.resp_fn)
            self.scale_pyr, self.sigmas, self.pix_dists = det["pyr"], det["sigmas"], det["pix_dists"]
            det["lafs"][:, 0:2, 0:2] = self.mrSize * det["lafs"][:, :, 0:2]
            self.detected = {k: det[k].clone() for k in ("resp", "lafs", "oct", "lev", "pix")}
            resp, lafs, octs, levs, pixs = det["resp"], det["lafs"], det["oct"], det["lev"], det["pix"]
            if self.iters > 0:
                resp, lafs, octs, levs, pixs = self._affine_shape(det, self.num)
            self.shaped = {"resp": resp.clone(), "lafs": lafs.clone()}
            if do_ori:
                lafs = self._orientation(lafs, octs, levs)
            self.keys = torch.stack([octs.long(), levs.long(), pixs.long()], dim=1)
            out = denormalize_lafs(lafs, x.size(3), x.size(2))
            self.last_lafs_px = out.numpy().copy()       # row order of self.keys (read by oracle/fp64_referee.py)
            return out, resp

    __call__ = forward

    def level_for_lafs(self, dlafs, ps):
        """LAF.py:450-472 (get_LAFs_scales + get_pyramid_and_level_index_for_LAFs): float64
        argmin over sigma_l * 2^o, first minimum wins."""
        sc = torch.sqrt(torch.abs(dlafs[:, 0, 0] * dlafs[:, 1, 1] - dlafs[:, 0, 1] * dlafs[:, 1, 0]) + 1e-12)
        need = (sc / ps).numpy().astype(np.float64)
        full, oi, li = [], [], []
        for o in range(len(self.sigmas)):
            full += list(np.array(self.sigmas[o]) * np.array(self.pix_dists[o]))
            oi += [o] * len(self.sigmas[o])
            li += list(range(len(self.sigmas[o])))
        d = np.abs(np.array(full).reshape(-1, 1) - need.reshape(1, -1))  # cdist on 1-D points
        best = d.argmin(axis=0)
        return torch.tensor(oi)[best], torch.tensor(li)[best]

    def extract_patches_from_pyr(self, dlafs, PS=41):
        """SparseImgRepresenter.py:181-188."""
        with torch.no_grad():
            octs, levs = self.level_for_lafs(dlafs, PS)
            w0, h0 = self.scale_pyr[0][0].size(3), self.scale_pyr[0][0].size(2)
            return extract_from_pyramid(self.scale_pyr, octs, levs, normalize_lafs(dlafs, w0, h0), PS)


def describe(x, extractor, hardnet_sd, do_ori=True, ps=32):
    """train_OriNet_test_on_graffity.py:293-298 (get_geometry_and_descriptors)."""
    lafs, resp = extractor(x, do_ori=do_ori)
    patches = extractor.extract_patches_from_pyr(lafs, PS=ps)
    with torch.no_grad():
        desc = hardnet_forward(hardnet_sd, patches)
    return lafs, resp, patches, desc


def detect_affine_shape(affnet_sd, patch_column_u8, ps=32):
    """examples/just_shape/detect_affine_shape.py:36-70: split an HPatches-style column of w x w
    tiles, resize to 32x32 (identity when w == 32), /255, AffNet in batches of 128, rows
    `a11 a12 a21 a22`."""
    h, w = patch_column_u8.shape
    n = h // w
    if w != ps:
        raise NotImplementedError("oracle covers w == PS (cv2.resize is the identity)")
    patches = np.stack([patch_column_u8[i * w:(i + 1) * w, 0:w] for i in range(n)]).astype(np.float32) / 255.0
    t = torch.from_numpy(patches).view(n, 1, ps, ps)
    outs = []
    with torch.no_grad():
        for s in range(0, n, 128):
            outs.append(affnet_forward(affnet_sd, t[s:s + 128]).reshape(-1, 4))
    return torch.cat(outs).numpy()


def synthetic_image(h, w, seed):
    """Deterministic multi-octave noise image, 0..255 float32 (SURVEY.md section 8d / BASELINE.md section 4)."""
    g = torch.Generator().manual_seed(int(seed))
    acc = torch.zeros(1, 1, h, w)
    s = 1
    while min(h, w) / s >= 4:
        hh, ww = -(-h // s), -(-w // s)
        n = torch.rand(1, 1, hh, ww, generator=g)
        acc += math.sqrt(s) * F.interpolate(n, size=(h, w), mode="bilinear", align_corners=False)
        s *= 2
    acc = (acc - acc.min()) / (acc.max() - acc.min()) * 255.0
    return acc.float().contiguous()


# ----------------------------------------------------------------------------------------
# descriptor matching (first consumer of the descriptors)
# ----------------------------------------------------------------------------------------


def distance_matrix_vector(anchor, positive):
    """(n1,d),(n2,d) -> (n1,n2) sqrt(|a|^2 + |b|^2 - 2 a.b + 1e-6).  Losses.py:5-13."""
    d1_sq = torch.sum(anchor * anchor, dim=1).unsqueeze(-1)
    d2_sq = torch.sum(positive * positive, dim=1).unsqueeze(-1)
    eps = 1e-6
    return torch.sqrt((d1_sq.repeat(1, positive.size(0)) + torch.t(d2_sq.repeat(1, anchor.size(0)))
                       - 2.0 * torch.bmm(anchor.unsqueeze(0), torch.t(positive).unsqueeze(0)).squeeze(0)) + eps)


def match_snn(desc1, desc2, snn_threshold=0.8):
    """Second-nearest-neighbour ratio test exactly as train_AffNet_test_on_graffity.py:292-300 does it:
    the "second nearest" is the minimum over the columns that are NOBODY's nearest neighbour
    (`dist_matrix[:, idxs_in_2] = 100000` masks whole columns for every row).
    Returns (min_dist, idx_in_2, min_2nd_dist, tent_in_1, tent_in_2)."""
    dist = distance_matrix_vector(desc1, desc2)
    min_dist, idxs_in_2 = torch.min(dist, 1)
    dist[:, idxs_in_2] = 100000
    min_2nd_dist, _ = torch.min(dist, 1)
    mask = (min_dist / (min_2nd_dist + 1e-8)) <= snn_threshold
    tent_in_1 = torch.arange(0, idxs_in_2.size(0))[mask].long()
    tent_in_2 = idxs_in_2[mask].long()
    return min_dist, idxs_in_2, min_2nd_dist, tent_in_1, tent_in_2


def lin_h(H, x, y):
    """Local affine approximation of a homography at (x, y).  ReprojectionStuff.py:9-21."""
    A = torch.zeros(x.size(0), 2, 2)
    den = x * H[2, 0] + y * H[2, 1] + H[2, 2]
    num1_densq = (x * H[0, 0] + y * H[0, 1] + H[0, 2]) / (den * den)
    num2_densq = (x * H[1, 0] + y * H[1, 1] + H[1, 2]) / (den * den)
    A[:, 0, 0] = H[0, 0] / den - num1_densq * H[2, 0]
    A[:, 0, 1] = H[0, 1] / den - num1_densq * H[2, 1]
    A[:, 1, 0] = H[1, 0] / den - num2_densq * H[2, 0]
    A[:, 1, 1] = H[1, 1] / den - num2_densq * H[2, 1]
    return A


def reproject_lafs(lafs1, H1to2):
    """Pixel LAFs (n,2,3) through a homography.  ReprojectionStuff.py:23-40 + LAF.py:91-95."""
    n = lafs1.size(0)
    lhf = torch.cat([lafs1, torch.Tensor([0, 0, 1]).unsqueeze(0).unsqueeze(0).repeat(n, 1, 1)], dim=1)
    xy1 = torch.bmm(H1to2.expand(n, 3, 3), lhf[:, :, 2:])
    xy1 = xy1 / xy1[:, 2:, :].expand(n, 3, 1)
    As = lin_h(H1to2, lafs1[:, 0, 2], lafs1[:, 1, 2])
    AF = torch.bmm(As, lhf[:, 0:2, 0:2])
    return torch.cat([AF, xy1[:, :2, :]], dim=2)


def distance_matrix_vector_reproj(anchor, positive):
    """ReprojectionStuff.py:78-86 - NOT the Losses.py function of the same name: the result is (n_positive, n_anchor),
    with abs() under the root and eps = 1e-12."""
    d1_sq = torch.sum(anchor * anchor, dim=1)
    d2_sq = torch.sum(positive * positive, dim=1)
    eps = 1e-12
    return torch.sqrt(torch.abs((d1_sq.expand(positive.size(0), anchor.size(0)) +
                                 torch.t(d2_sq.expand(anchor.size(0), positive.size(0)))
                                 - 2.0 * torch.bmm(positive.unsqueeze(0), torch.t(anchor).unsqueeze(0)).squeeze(0)) + eps))


def get_gt_correspondence_indexes(lafs1, lafs2, H1to2, dist_threshold=4):
    """ReprojectionStuff.py:126-137: centres of lafs2 reprojected into image 1; for every lafs1 centre the nearest
    reprojected centre (ReprojectionStuff's own distance_matrix_vector returns (n1, n2), so torch.min(dist, 1) runs over
    the reprojected set; the |a|^2+|b|^2-2ab expansion in fp32 is only good to ~0.1 px at ~800 px coordinates), kept if
    <= dist_threshold."""
    lafs2_in_1 = reproject_lafs(lafs2, torch.inverse(H1to2))
    c1 = lafs1[:, :, 2]
    c2 = lafs2_in_1[:, 0:2, 2]
    dist = distance_matrix_vector_reproj(c2, c1)
    min_dist, idxs_in_2 = torch.min(dist, 1)
    plain = torch.arange(0, idxs_in_2.size(0))
    mask = min_dist <= dist_threshold
    return min_dist[mask], plain[mask], idxs_in_2[mask]


def match_and_verify(lafs1, desc1, lafs2, desc2, H1to2, snn_threshold=0.8, dist_threshold=6):
    """The evaluation of train_AffNet_test_on_graffity.py:290-305 (test()): tentatives + homography-consistent matches."""
    md, idx, md2, t1, t2 = match_snn(desc1, desc2, snn_threshold)
    gd, plain, in2 = get_gt_correspondence_indexes(lafs1[t1], lafs2[t2], H1to2, dist_threshold)
    return dict(min_dist=md, idx=idx, min_2nd=md2, tent1=t1, tent2=t2, gt_dist=gd, gt_plain=plain, gt_idx=in2)


# ----------------------------------------------------------------------------------------
# the hand-crafted default slot fillers
# ----------------------------------------------------------------------------------------


def circular_gauss_kernel(kernlen=None, circ_zeros=False, sigma=None, norm=True):
    """Utils.py:92-114 under Python 3 (`kernlen / 2` is a true division)."""
    if kernlen is None:
        kernlen = int(2.0 * 3.0 * sigma + 1.0)
        if kernlen % 2 == 0:
            kernlen = kernlen + 1
    half = kernlen / 2
    r2 = float(half * half)
    sigma2 = 0.9 * r2 if sigma is None else 2.0 * sigma * sigma
    x = np.linspace(-half, half, kernlen)
    xv, yv = np.meshgrid(x, x, sparse=False, indexing="xy")
    distsq = xv ** 2 + yv ** 2
    kernel = np.exp(-(distsq / sigma2))
    if circ_zeros:
        kernel *= (distsq <= r2).astype(np.float32)
    if norm:
        kernel /= np.sum(kernel)
    return kernel


def orientation_detector(x, num_ang_bins=36):
    """Dominant gradient orientation of (n,1,PS,PS) patches -> angles (n,).  HandCraftedModules.py:133-192
    (OrientationDetector.forward): only the lower bin bo0 is accumulated (with weight (1 - wo1) * mag), the smoothing conv1d
    zero-pads (not circular), argmax -> angle = -(2 pi idx / 36 - pi)."""
    ps = x.size(2)
    wx = torch.tensor([[[[0.5, 0, -0.5]]]])
    wy = torch.tensor([[[[0.5], [0], [-0.5]]]])
    gk = 10.0 * torch.from_numpy(circular_gauss_kernel(kernlen=ps).astype(np.float32))
    gx = F.conv2d(F.pad(x, (1, 1, 0, 0), "replicate"), wx)
    gy = F.conv2d(F.pad(x, (0, 0, 1, 1), "replicate"), wy)
    mag = torch.sqrt(gx * gx + gy * gy + 1e-10)
    mag = mag * gk.unsqueeze(0).unsqueeze(0).expand_as(mag)
    ori = torch.atan2(gy, gx)
    o_big = float(num_ang_bins) * (ori + 1.0 * math.pi) / (2.0 * math.pi)
    bo0 = torch.floor(o_big)
    wo1 = o_big - bo0
    bo0 = bo0 % num_ang_bins
    wo0 = (1.0 - wo1) * mag
    bins = [F.adaptive_avg_pool2d((bo0 == i).float() * wo0, (1, 1)) for i in range(num_ang_bins)]
    bins = torch.cat(bins, 1).view(-1, 1, num_ang_bins)
    bins = F.conv1d(bins, torch.tensor([[[0.33, 0.34, 0.33]]]), padding=1)
    _, idx = bins.view(-1, num_ang_bins).max(1)
    return -((2.0 * float(np.pi) * idx.float() / float(num_ang_bins)) - float(math.pi))


def angles_to_rotation(angles):
    """LAF.py:306-311 (angles2A)."""
    c, s = torch.cos(angles).view(-1, 1, 1), torch.sin(angles).view(-1, 1, 1)
    return torch.cat([torch.cat([c, s], dim=2), torch.cat([-s, c], dim=2)], dim=1)


def affine_shape_estimator(x):
    """Baumberg second-moment shape of (n,1,PS,PS) patches -> rectified (n,2,2).  HandCraftedModules.py:81-132
    (AffineShapeEstimator.forward + invSqrt) + LAF.py:299-302 (abc2A) + :285-291 (rectify)."""
    ps = x.size(2)
    wx = torch.tensor([[[[-1.0, 0, 1.0]]]])
    wy = torch.tensor([[[[-1.0], [0], [1.0]]]])
    gk = torch.from_numpy(circular_gauss_kernel(kernlen=ps, sigma=(ps / 2) / 3.0).astype(np.float32))
    gx = F.conv2d(F.pad(x, (1, 1, 0, 0), "replicate"), wx)
    gy = F.conv2d(F.pad(x, (0, 0, 1, 1), "replicate"), wy)
    g = gk.unsqueeze(0).unsqueeze(0).expand_as(gx)
    a = (gx * gx * g).view(x.size(0), -1).mean(dim=1)
    b = (gx * gy * g).view(x.size(0), -1).mean(dim=1)
    c = (gy * gy * g).view(x.size(0), -1).mean(dim=1)
    eps = 1e-12
    mask = (b != 0).float()
    r1 = mask * (c - a) / (2.0 * b + eps)
    t1 = torch.sign(r1) / (torch.abs(r1) + torch.sqrt(1.0 + r1 * r1))
    r = 1.0 / torch.sqrt(1.0 + t1 * t1)
    t = t1 * r
    r = r * mask + 1.0 * (1.0 - mask)
    t = t * mask
    xx = 1.0 / torch.sqrt(r * r * a - 2.0 * r * t * b + t * t * c)
    zz = 1.0 / torch.sqrt(t * t * a + 2.0 * r * t * b + r * r * c)
    d = torch.sqrt(xx * zz)
    xx = xx / d
    zz = zz / d
    na = r * r * xx + t * t * zz
    nb = -r * t * xx + t * r * zz
    nc = t * t * xx + r * r * zz
    A = torch.cat([torch.cat([na.view(-1, 1, 1), nb.view(-1, 1, 1)], dim=2), torch.cat([nb.view(-1, 1, 1), nc.view(-1, 1, 1)], dim=2)], dim=1)
    return rectify_up_is_up(A)


# ----------------------------------------------------------------------------------------
# LAF -> Oxford ellipse on tensors
# ----------------------------------------------------------------------------------------


def bsvd2x2(As):
    """Batched closed-form 2x2 SVD.  LAF.py:106-144."""
    Su = torch.bmm(As, As.permute(0, 2, 1))
    phi = 0.5 * torch.atan2(Su[:, 0, 1] + Su[:, 1, 0] + 1e-12, Su[:, 0, 0] - Su[:, 1, 1] + 1e-12)
    U = torch.zeros(As.size(0), 2, 2)
    U[:, 0, 0] = torch.cos(phi); U[:, 1, 1] = torch.cos(phi); U[:, 0, 1] = -torch.sin(phi); U[:, 1, 0] = torch.sin(phi)
    Sw = torch.bmm(As.permute(0, 2, 1), As)
    theta = 0.5 * torch.atan2(Sw[:, 0, 1] + Sw[:, 1, 0] + 1e-12, Sw[:, 0, 0] - Sw[:, 1, 1] + 1e-12)
    W = torch.zeros(As.size(0), 2, 2)
    W[:, 0, 0] = torch.cos(theta); W[:, 1, 1] = torch.cos(theta); W[:, 0, 1] = -torch.sin(theta); W[:, 1, 0] = torch.sin(theta)
    SUsum = Su[:, 0, 0] + Su[:, 1, 1]
    SUdif = torch.sqrt((Su[:, 0, 0] - Su[:, 1, 1]) ** 2 + 4 * Su[:, 0, 1] * Su[:, 1, 0] + 1e-12)
    SIG = torch.zeros(As.size(0), 2, 2)
    SIG[:, 0, 0] = torch.sqrt((SUsum + SUdif) / 2.0)
    SIG[:, 1, 1] = torch.sqrt((SUsum - SUdif) / 2.0)
    S = torch.bmm(torch.bmm(U.permute(0, 2, 1), As), W)
    C = torch.sign(S)
    C[:, 0, 1] = 0
    C[:, 1, 0] = 0
    return U, SIG, torch.bmm(W, C)


def lafs_to_ellipses_t(lafs):
    """(n,2,3) pixel LAFs -> (n,5) Oxford ellipses x y a b c on tensors.  LAF.py:35-51 (LAFs2ellT)."""
    ell = torch.zeros((len(lafs), 5))
    scale = torch.sqrt(lafs[:, 0, 0] * lafs[:, 1, 1] - lafs[:, 0, 1] * lafs[:, 1, 0] + 1e-10)
    u, W, _ = bsvd2x2(lafs[:, 0:2, 0:2] / scale.view(-1, 1, 1).repeat(1, 2, 2))
    W[:, 0, 0] = 1.0 / (scale * scale * W[:, 0, 0] ** 2)
    W[:, 1, 1] = 1.0 / (scale * scale * W[:, 1, 1] ** 2)
    A = torch.bmm(torch.bmm(u, W), u.permute(0, 2, 1))
    ell[:, 0], ell[:, 1] = lafs[:, 0, 2], lafs[:, 1, 2]
    ell[:, 2], ell[:, 3], ell[:, 4] = A[:, 0, 0], A[:, 0, 1], A[:, 1, 1]
    return ell
