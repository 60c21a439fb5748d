"""The reference's hand-crafted slot fillers with their constructor and call signature (HandCraftedModules.py:81-192),
executed by csrc/handcrafted.hip.  A default-constructed ScaleSpaceAffinePatchExtractor uses them
(SparseImgRepresenter.py:42-49): OrientationDetector(patch_size=19), AffineShapeEstimator(patch_size=19)."""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib, engine
from ._lib import lib, check, ptr
from .host_plan import circular_gauss_kernel

HC_ORIENTATION, HC_BAUMBERG = 0, 1


class _HipHandCrafted(nn.Module):
    KIND = None

    def __init__(self, patch_size):
        super(_HipHandCrafted, self).__init__()
        if patch_size != 19:
            raise NotImplementedError("the HIP hand-crafted slot kernels are specialised for 19x19 patches "
                                      "(the size ScaleSpaceAffinePatchExtractor and hesaffBaum.py use), got %r" % (patch_size,))
        self.PS = patch_size
        self._window = None

    def window(self):
        """19x19 Gaussian window as a ctypes float array (kept alive by the module)."""
        if self._window is None:
            w = np.ascontiguousarray(self._make_window().astype(np.float32).reshape(-1))
            self._window = (C.c_float * w.size)(*w.tolist())
        return self._window

    def _run(self, x, want_angles=False):
        engine.require_cuda(x, "patches")
        if x.dim() == 4:
            x = x[:, 0]
        if tuple(x.shape[1:]) != (19, 19):
            raise ValueError("expected (n,1,19,19) patches, got %s" % (tuple(x.shape),))
        x = x.contiguous().float()
        n, dev = x.size(0), x.device
        out = torch.empty(n, 2, 2, dtype=torch.float32, device=dev)
        ang = torch.empty(n, dtype=torch.float32, device=dev) if want_angles else None
        if n:
            ctx = engine.utility_ctx(dev)
            rc = lib.affnet_handcrafted_forward(ctx, self.KIND, ptr(x), n, C.cast(self.window(), C.c_void_p), ptr(out), ptr(ang),
                                                engine.stream_of(dev))
            check(rc, ctx, "affnet_handcrafted_forward")
        return out, ang


class OrientationDetector(_HipHandCrafted):
    KIND = HC_ORIENTATION

    def __init__(self, mrSize=3.0, patch_size=None):
        super(OrientationDetector, self).__init__(32 if patch_size is None else patch_size)
        self.mrSize, self.num_ang_bins = mrSize, 36

    def _make_window(self):
        return 10.0 * circular_gauss_kernel(kernlen=self.PS)              # HandCraftedModules.py:152

    def forward(self, x, return_rot_matrix=False):
        R, ang = self._run(x, want_angles=not return_rot_matrix)
        return R if return_rot_matrix else ang


class AffineShapeEstimator(_HipHandCrafted):
    KIND = HC_BAUMBERG

    def __init__(self, threshold=0.001, patch_size=19):
        super(AffineShapeEstimator, self).__init__(patch_size)
        self.threshold = threshold

    def _make_window(self):
        return circular_gauss_kernel(kernlen=self.PS, sigma=(self.PS / 2) / 3.0)   # HandCraftedModules.py:90

    def forward(self, x, *ignored):
        """(n,1,19,19) -> (n,2,2) rectified shape.  Extra positional arguments are ignored: the reference's batched_forward
        passes a kwargs dict positionally (Utils.py:54), which its own AffineShapeEstimator.forward rejects (TypeError)."""
        return self._run(x)[0]


class NMS2d(nn.Module):
    """HandCraftedModules.py:194-206: keeps x where x - maxpool3x3(x) + 1e-5 > 0 (and x > threshold when threshold > 1e-5).  The
    reference's constructor raises under Python 3 (`padding = kernel_size/2` is a float); the integer padding it meant is used."""

    def __init__(self, kernel_size=3, threshold=0):
        super(NMS2d, self).__init__()
        if kernel_size != 3:
            raise NotImplementedError("the HIP kernel is the 3 x 3 NMS the reference uses")
        self.eps, self.th = 1e-5, threshold

    def forward(self, x):
        engine.require_cuda(x, "response map")
        if x.dim() != 4 or x.size(0) != 1 or x.size(1) != 1:
            raise ValueError("expected a (1,1,H,W) response map")
        x = x.contiguous().float()
        out = torch.empty_like(x)
        ctx = engine.utility_ctx(x.device)
        check(lib.affnet_nms2d(ctx, ptr(x), ptr(out), x.size(2), x.size(3), float(self.th), engine.stream_of(x.device)), ctx, "affnet_nms2d")
        return out
