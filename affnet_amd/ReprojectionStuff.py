"""Homography reprojection of LAFs and ground-truth correspondences with the reference's names
(ReprojectionStuff.py:23-40,126-137), plus the SNN matcher of train_AffNet_test_on_graffity.py:292-300 as one call."""
import ctypes as C

import numpy as np
import torch

from . import engine
from ._lib import lib, check, ptr


def _h9(H):
    h = np.ascontiguousarray(torch.as_tensor(H).detach().cpu().numpy(), dtype=np.float32).reshape(9)
    return (C.c_float * 9)(*h.tolist())


def reprojectLAFs(LAFs1, H1to2, return_LHFs=False):
    """Pixel LAFs (n,2,3) cuda -> reprojected LAFs (n,2,3) (or (n,3,3) homogeneous frames)."""
    engine.require_cuda(LAFs1, "LAFs1")
    lafs = LAFs1.contiguous().float()
    out = torch.empty_like(lafs)
    n = lafs.size(0)
    if n:
        ctx = engine.utility_ctx(lafs.device)
        check(lib.affnet_reproject_lafs(ctx, ptr(lafs), n, _h9(H1to2), ptr(out), engine.stream_of(lafs.device)), ctx, "affnet_reproject_lafs")
    if return_LHFs:
        last = torch.tensor([0.0, 0.0, 1.0], device=lafs.device).view(1, 1, 3).repeat(n, 1, 1)
        return torch.cat([out, last], dim=1)
    return out


def get_GT_correspondence_indexes(LAFs1, LAFs2, H1to2, dist_threshold=4):
    """ReprojectionStuff.py:126-137: (min_dist[mask], plain_indxs_in1[mask], idxs_in_2[mask])."""
    engine.require_cuda(LAFs1, "LAFs1")
    Hinv = torch.inverse(torch.as_tensor(H1to2).detach().cpu().float())        # 3x3 on the host, as the reference (LAPACK)
    l2_in_1 = reprojectLAFs(LAFs2, Hinv)
    l1 = LAFs1.contiguous().float()
    nq, nr = l1.size(0), l2_in_1.size(0)       # rows = image-1 LAFs, minimum over the reprojected image-2 LAFs (:131-132)
    dev = l1.device
    md = torch.empty(nq, dtype=torch.float32, device=dev)
    idx = torch.empty(nq, dtype=torch.int32, device=dev)
    if nq and nr:
        ctx = engine.utility_ctx(dev)
        check(lib.affnet_centre_nn(ctx, ptr(l1), nq, ptr(l2_in_1), nr, ptr(md), ptr(idx), engine.stream_of(dev)), ctx, "affnet_centre_nn")
    mask = md <= dist_threshold
    plain = torch.arange(0, nq, device=dev)
    return md[mask], plain[mask], idx[mask].long()


def match_snn(descriptors1, descriptors2, SNN_threshold=0.8):
    """train_AffNet_test_on_graffity.py:292-300 in one fused call (no n1 x n2 matrix in HBM).
    Returns (tent_matches_in_1, tent_matches_in_2, min_dist, min_2nd_dist) - the first two int64 like the reference."""
    engine.require_cuda(descriptors1, "descriptors1")
    engine.require_cuda(descriptors2, "descriptors2")
    a, b = descriptors1.contiguous().float(), descriptors2.contiguous().float()
    n1, n2, dev = a.size(0), b.size(0), a.device
    if n2 == 0:
        raise ValueError("match_snn: descriptors2 is empty (the reference's torch.min over an empty dimension raises too)")
    if n1 == 0:
        e = torch.empty(0, dtype=torch.int64, device=dev)
        return e, e.clone(), torch.empty(0, device=dev), torch.empty(0, device=dev)
    md = torch.empty(n1, dtype=torch.float32, device=dev)
    md2 = torch.empty(n1, dtype=torch.float32, device=dev)
    idx = torch.empty(n1, dtype=torch.int32, device=dev)
    tent = torch.empty(n1, 2, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    scratch = torch.empty(lib.affnet_match_scratch_bytes(n1, n2), dtype=torch.uint8, device=dev)
    ctx = engine.utility_ctx(dev)
    rc = lib.affnet_match_snn(ctx, ptr(a), n1, ptr(b), n2, a.size(1), float(SNN_threshold), ptr(md), ptr(idx), ptr(md2), ptr(tent), ptr(cnt),
                              ptr(scratch), engine.stream_of(dev))
    check(rc, ctx, "affnet_match_snn")
    k = int(cnt.item())
    return tent[:k, 0].long(), tent[:k, 1].long(), md, md2
