"""LAF helpers with the reference's names and argument meaning (LAF.py), backed by HIP kernels.

Only what the hot path and its callers need: extract_patches (:364-372), denormalizeLAFs /
normalizeLAFs (:407-429), LAFs2ell (:225-240, host, Oxford ellipse text format),
convertLAFs_to_A23format (:200-223)."""
import numpy as np
import torch

from . import engine
from ._lib import lib, check, ptr


def extract_patches(img, LAFs, PS=32, bs=32):
    """img (1,1,h,w) cuda fp32; LAFs (n,2,3) normalised -> (n,1,PS,PS).  `bs` (chunk size of the
    reference's batched_grid_apply) is accepted and ignored: one launch samples every patch."""
    engine.require_cuda(img, "img")
    if img.dim() != 4 or img.size(0) != 1 or img.size(1) != 1:
        raise ValueError("extract_patches expects a (1,1,h,w) image")
    img = img.contiguous().float()
    lafs = LAFs.to(img.device, torch.float32).contiguous()
    n, h, w = lafs.size(0), img.size(2), img.size(3)
    out = torch.empty(n, 1, PS, PS, dtype=torch.float32, device=img.device)
    if n:
        ctx = engine.utility_ctx(img.device)
        rc = lib.affnet_laf_grid_sample(ctx, ptr(img), h, w, ptr(lafs), n, PS, ptr(out), engine.stream_of(img.device))
        check(rc, ctx, "affnet_laf_grid_sample")
    return out


def _scale(LAFs, w, h, inverse):
    engine.require_cuda(LAFs, "LAFs")
    lafs = LAFs.contiguous().float()
    out = torch.empty_like(lafs)
    n = lafs.size(0)
    if n:
        ctx = engine.utility_ctx(lafs.device)
        rc = lib.affnet_scale_lafs(ctx, ptr(lafs), ptr(out), None, n, int(w), int(h), int(inverse), engine.stream_of(lafs.device))
        check(rc, ctx, "affnet_scale_lafs")
    return out


def denormalizeLAFs(LAFs, w, h):
    return _scale(LAFs, w, h, 0)


def normalizeLAFs(LAFs, w, h):
    return _scale(LAFs, w, h, 1)


def LAFs2ellT(LAFs):
    """(n,2,3) cuda pixel LAFs -> (n,5) Oxford ellipses on the device (LAF.py:35-51, closed-form 2x2 SVD :106-144)."""
    engine.require_cuda(LAFs, "LAFs")
    lafs = LAFs.contiguous().float()
    n = lafs.size(0)
    out = torch.zeros(n, 5, dtype=torch.float32, device=lafs.device)
    if n:
        ctx = engine.utility_ctx(lafs.device)
        check(lib.affnet_lafs_to_ellipses(ctx, ptr(lafs), None, n, ptr(out), engine.stream_of(lafs.device)), ctx, "affnet_lafs_to_ellipses")
    return out


def convertLAFs_to_A23format(LAFs):
    sh = LAFs.shape
    if len(sh) == 3 and sh[1] == 2 and sh[2] == 3:
        return np.array(LAFs, copy=True)
    out = np.zeros((sh[0], 2, 3))
    if len(sh) == 2 and sh[1] == 7:      # x y scale a11 a12 a21 a22
        out[:, 0, 2], out[:, 1, 2] = LAFs[:, 0], LAFs[:, 1]
        out[:, 0, 0], out[:, 0, 1] = LAFs[:, 2] * LAFs[:, 3], LAFs[:, 2] * LAFs[:, 4]
        out[:, 1, 0], out[:, 1, 1] = LAFs[:, 2] * LAFs[:, 5], LAFs[:, 2] * LAFs[:, 6]
        return out
    if len(sh) == 2 and sh[1] == 6:      # x y s*a11 s*a12 s*a21 s*a22
        out[:, 0, 2], out[:, 1, 2] = LAFs[:, 0], LAFs[:, 1]
        out[:, 0, 0], out[:, 0, 1], out[:, 1, 0], out[:, 1, 1] = LAFs[:, 2], LAFs[:, 3], LAFs[:, 4], LAFs[:, 5]
        return out
    raise ValueError("Unknown LAF format")


def LAFs2ell(in_LAFs):
    """(n,2,3) numpy LAFs -> (n,5) Oxford ellipses x y a b c (host; per-row SVD in the input dtype)."""
    lafs = convertLAFs_to_A23format(np.asarray(in_LAFs))
    ell = np.zeros((len(lafs), 5))
    for i in range(len(lafs)):
        L = lafs[i].copy()
        sc = np.sqrt(L[0, 0] * L[1, 1] - L[0, 1] * L[1, 0] + 1e-10)
        u, W, _ = np.linalg.svd(L[0:2, 0:2] / sc, full_matrices=True)
        W[0] = 1.0 / (W[0] * W[0] * sc * sc)
        W[1] = 1.0 / (W[1] * W[1] * sc * sc)
        A = np.matmul(np.matmul(u, np.diag(W)), u.transpose())
        ell[i] = [L[0, 2], L[1, 2], A[0, 0], A[0, 1], A[1, 1]]
    return ell
