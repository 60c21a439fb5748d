"""OnePassSIR with the reference's constructor and forward() (OnePassSIR.py:14-153), executed on MI355X by libaffnet_hip.so:
the affine shape comes from ONE dense evaluation of a fully-convolutional AffNet per pyramid octave instead of one CNN
evaluation per candidate patch.

* AffNet = affnet_amd.architectures.AffNetFastFullConv (native): one C call builds the pyramid, evaluates the dense net on level 0
  of every octave (csrc/fullconv.hip) and runs the detector with the per-level top-k, the frame boundary test and the LAF
  composition s * A_map[pixel] * mrSize (csrc/detect.hip); a second C call does orientation, denormalisation and descriptors;
* AffNet = any callable image (1,1,h,w) -> (1,4,h,w) (the reference's slot signature, OnePassSIR.py:69): evaluated per octave by this
  mirror, the maps are copied into the workspace and the same detector runs on them.

As in the reference every pyramid octave must be at least 34 px wide / high (LocalNorm2d(33) reflect-pads by 16): construct with
border >= 15 like the reference's scripts (extract_geom_and_desc_upisup.py:63).  The reference's own default AffNet slot
(AffineShapeEstimator, a per-patch module) cannot produce a dense map and fails there as well: AffNet is required here.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib, engine
from ._lib import lib, check, ptr
from .architectures import _HipPatchNet, AffNetFastFullConv
from .HandCraftedModules import OrientationDetector, _HipHandCrafted


class OnePassSIR(nn.Module):
    def __init__(self, border=16, num_features=500, patch_size=32, mrSize=3.0, nlevels=3, th=None, num_Baum_iters=0, init_sigma=1.6,
                 RespNet=None, OriNet=None, AffNet=None, arith="fp32"):
        super(OnePassSIR, self).__init__()
        self.arith = arith           # "fp32" (exact fp32 MFMA, default) / "fp32_split3" / "fp32_split2h": see ScaleSpaceAffinePatchExtractor
        _lib.arith_code(arith)
        self.mrSize, self.PS, self.b = mrSize, patch_size, border
        self.num, self.th = num_features, th
        if th is not None:          # OnePassSIR.py:31-34
            self.num = -1
        else:
            self.th = 0
        self.nlevels, self.num_Baum_iters, self.init_sigma = nlevels, num_Baum_iters, init_sigma
        self.RespNet = RespNet       # None = the built-in Hessian response; otherwise any callable RespNet(level (1,1,h,w), sigma) ->
                                     # (1,1,h,w) (OnePassSIR.py:24,38-41,71), evaluated per pyramid level by this mirror
        if AffNet is None:
            raise ValueError("OnePassSIR needs a dense AffNet (AffNetFastFullConv or any image -> (1,4,h,w) callable); the reference's default "
                             "AffineShapeEstimator is a per-patch module and fails in OnePassSIR.py:69 as well")
        self.AffNet = AffNet
        # OnePassSIR.py:44-47; a foreign OriNet (any callable patches (n,1,PS,PS) -> angles (n,) or rotations (n,2,2) with a .PS
        # attribute, OnePassSIR.py:117-130) runs between the stages (_staged_orientation)
        self.OriNet = OriNet if OriNet is not None else OrientationDetector(patch_size=19)
        self.scale_pyr = self.sigmas = self.pix_dists = self.aff_maps = None
        self.max_keep, self.raw_div = 16384, 4
        self._ctx = self._ctx_key = None
        self.last_ids = None

    def _context(self, x):
        engine.require_cuda(x, "image")
        if x.dim() != 4 or x.size(1) != 1 or x.size(0) < 1:
            raise ValueError("expected a (B,1,H,W) image batch (B = 1: the reference's shape)")
        key = (x.size(0), x.size(2), x.size(3), x.device, self.num, float(self.th), self.mrSize, self.b, self.init_sigma, self.nlevels, self.max_keep,
               self.raw_div)
        if self._ctx is not None and self._ctx_key == key and self._ctx.arith != _lib.arith_code(self.arith):
            self._ctx.set_arith(self.arith)
        if self._ctx is None or self._ctx_key != key:
            self._ctx = engine.Context(x.size(2), x.size(3), x.device, self.nlevels, self.init_sigma, self.b, self.mrSize, float(self.th), self.num,
                                       self.num, self.max_keep, batch=x.size(0), baum_iters=0, raw_div=self.raw_div, onepass=True, arith=self.arith)
            self._ctx_key = key
        return self._ctx

    def enqueue(self, x, do_ori=True, desc=None):
        """Whole path without host synchronisation; x (B,1,H,W).  Returns capacity-sized device tensors + the device row counts."""
        if do_ori and not isinstance(self.OriNet, (_HipPatchNet, _HipHandCrafted)):      # before anything is enqueued
            raise NotImplementedError("enqueue() is the no-synchronisation path: a foreign OriNet runs Python between the stages - use run() / forward()")
        ctx = self._context(x)
        dev, st = x.device, engine.stream_of(x.device)
        img = x.contiguous().float()
        B, F = x.size(0), ctx.cap_final
        lafs = torch.empty(B, F, 2, 3, dtype=torch.float32, device=dev)
        resp = torch.empty(B, F, dtype=torch.float32, device=dev)
        ids = torch.empty(B, F, 3, dtype=torch.int32, device=dev)
        count = torch.zeros(B, dtype=torch.int32, device=dev)
        dsc = torch.empty(B, F, 128, dtype=torch.float32, device=dev) if desc is not None else None
        self._detect(ctx, img, st)
        nets = _lib.Nets()
        if do_ori:
            if isinstance(self.OriNet, _HipHandCrafted):
                nets.h_orientation_window = C.cast(self.OriNet.window(), C.c_void_p)
            else:
                nets.d_orinet = self.OriNet.packed_weights(dev).data_ptr()
        nets.d_hardnet = desc.packed_weights(dev).data_ptr() if desc is not None else None
        check(lib.affnet_describe_detected(ctx.handle, C.byref(nets), int(bool(do_ori)), ptr(lafs), ptr(resp), ptr(ids), ptr(dsc), ptr(count), st),
              ctx.handle, "affnet_describe_detected")
        return {"LAFs": lafs, "responses": resp, "ids": ids, "descriptors": dsc, "count": count, "overflow": ctx.counter_view(0), "_img": img}

    def _detect(self, ctx, img, st):
        """Detector half into the context's internal list: pyramid, dense affine maps (native net on the device, or a foreign dense
        AffNet evaluated per octave, OnePassSIR.py:69), response maps of a custom RespNet slot if any, per-level top-k + boundary test."""
        dev, B = img.device, img.size(0)
        native = isinstance(self.AffNet, AffNetFastFullConv)
        packed = ptr(self.AffNet.packed_weights(dev)) if native else None
        if native and self.RespNet is None:
            check(lib.affnet_detect_image_onepass(ctx.handle, packed, ptr(img), st), ctx.handle, "affnet_detect_image_onepass")
            return
        check(lib.affnet_pyramid_build(ctx.handle, ptr(img), st), ctx.handle, "affnet_pyramid_build")
        if not native:              # foreign dense AffNet slot: any callable image -> (1,4,h,w)
            with torch.no_grad():
                for b in range(B):
                    pyr, maps = ctx.pyramid_views(b), ctx.affmap_views(b)
                    for o in range(len(pyr)):
                        maps[o].copy_(self.AffNet(pyr[o][0]).to(dev, torch.float32))
        if self.RespNet is None:
            check(lib.affnet_detect_image_onepass(ctx.handle, None, None, st), ctx.handle, "affnet_detect_image_onepass")
            return
        stride = lib.affnet_pyramid_image_stride(ctx.handle)
        base = lib.affnet_pyramid_level_offset(ctx.handle, 0, 0)
        rmaps = torch.zeros(B * stride, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for b in range(B):
                pyr = ctx.pyramid_views(b)
                for o, (h, w) in enumerate(ctx.plan.sizes):
                    for l in range(ctx.plan.levels_per_octave):
                        off = b * stride + lib.affnet_pyramid_level_offset(ctx.handle, o, l) - base
                        rmaps[off:off + h * w].copy_(self.RespNet(pyr[o][l], ctx.plan.sigmas[o][l]).reshape(-1).to(dev, torch.float32))
        check(lib.affnet_detect_image_onepass_responses(ctx.handle, packed, ptr(rmaps), st), ctx.handle, "affnet_detect_image_onepass_responses")
        self._rmaps = rmaps         # stays alive until the kernels that read it have run

    def _staged_orientation(self, x, desc):
        """Foreign OriNet slot (OnePassSIR.py:117-130 getOrientation with any callable): detector on the device, the slot called on the
        patch tensor, rotation / denormalisation / descriptors by the stage kernels the fused path is made of."""
        ctx = self._context(x)
        dev, st = x.device, engine.stream_of(x.device)
        img = x.contiguous().float()
        P = ctx.cap_pre
        self._detect(ctx, img, st)
        resp = torch.empty(P, dtype=torch.float32, device=dev)
        lafs = torch.empty(P, 2, 3, dtype=torch.float32, device=dev)
        ids = torch.empty(P, 3, dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        check(lib.affnet_detected_list(ctx.handle, ptr(resp), ptr(lafs), ptr(ids), ptr(cnt), st), ctx.handle, "affnet_detected_list")
        ctx.read_counts()
        n = int(cnt.item())
        PS = self.OriNet.PS
        patches = torch.empty(n, 1, PS, PS, dtype=torch.float32, device=dev)
        check(lib.affnet_pyr_grid_sample(ctx.handle, ptr(lafs), ptr(ids), ptr(cnt), n, PS, ptr(patches), st), ctx.handle, "affnet_pyr_grid_sample")
        with torch.no_grad():
            ang = self.OriNet(patches)
        if ang.dim() <= 2:          # angles -> rotation matrices (LAF.py:306-311)
            c, s = torch.cos(ang).view(-1, 1, 1), torch.sin(ang).view(-1, 1, 1)
            ang = torch.cat([torch.cat([c, s], 2), torch.cat([-s, c], 2)], 1)
        R = ang.to(dev, torch.float32).contiguous()
        check(lib.affnet_apply_rotation(ctx.handle, ptr(lafs), ptr(R), ptr(cnt), n, st), ctx.handle, "affnet_apply_rotation")
        out = torch.empty_like(lafs)
        check(lib.affnet_scale_lafs(ctx.handle, ptr(lafs), ptr(out), ptr(cnt), P, x.size(3), x.size(2), 0, st), ctx.handle, "affnet_scale_lafs")
        dsc = None
        if desc is not None:
            lids = torch.empty(n, 3, dtype=torch.int32, device=dev)
            norm = torch.empty(n, 2, 3, dtype=torch.float32, device=dev)
            dp = torch.empty(n, 1, 32, 32, dtype=torch.float32, device=dev)
            check(lib.affnet_level_select(ctx.handle, ptr(out), None, n, 32, ptr(lids), ptr(norm), st), ctx.handle, "affnet_level_select")
            check(lib.affnet_pyr_grid_sample(ctx.handle, ptr(norm), ptr(lids), None, n, 32, ptr(dp), st), ctx.handle, "affnet_pyr_grid_sample")
            with torch.no_grad():
                dsc = desc(dp)
        return {"LAFs": out[:n].unsqueeze(0), "responses": resp[:n].unsqueeze(0), "ids": ids[:n].unsqueeze(0),
                "descriptors": None if dsc is None else dsc.unsqueeze(0), "count": cnt}

    def run(self, x, do_ori=True, desc=None):
        """x (1,1,H,W) -> dict(LAFs px (N,2,3), responses (N,), ids (N,3) = (octave, level - 1, pixel), descriptors (N,128) | None)."""
        if x.size(0) != 1:
            raise ValueError("run() / forward() are batch-size-1 like the reference; use enqueue() for batches")
        staged = do_ori and not isinstance(self.OriNet, (_HipPatchNet, _HipHandCrafted))
        if staged:
            r = self._staged_orientation(x, desc)
        else:
            r = self.enqueue(x, do_ori=do_ori, desc=desc)
        ctx = self._ctx
        self.scale_pyr = ctx.pyramid_views()
        self.sigmas = [list(s) for s in ctx.plan.sigmas]
        self.pix_dists = [list(p) for p in ctx.plan.pix_dists]
        self.aff_maps = ctx.affmap_views()
        counts = ctx.read_counts()      # raises on capacity overflow / "no keypoints detected"; [1] = rows of the (single) image after the fused describe call
        n = int(r["count"].item()) if staged else int(counts[1])
        self.last_ids = r["ids"][0, :n]
        dsc = r["descriptors"]
        return {"LAFs": r["LAFs"][0, :n], "responses": r["responses"][0, :n], "ids": r["ids"][0, :n],
                "descriptors": None if dsc is None else dsc[0, :n]}

    def forward(self, x, do_ori=True):
        """OnePassSIR.py:139-153: (LAFs in pixels (N,2,3), responses (N,))."""
        r = self.run(x, do_ori=do_ori)
        return r["LAFs"], r["responses"]

    def extract_patches_from_pyr(self, dLAFs, PS=41):
        """OnePassSIR.py:130-138: pixel LAFs (N,2,3) -> (N,1,PS,PS) from the best pyramid level of the LAST forward()."""
        if self._ctx is None or self.scale_pyr is None:
            raise RuntimeError("call forward() first: the pyramid of the last image is reused (stateful like the reference)")
        ctx = self._ctx
        if ctx.batch != 1:
            raise RuntimeError("extract_patches_from_pyr works on the pyramid of a single-image forward()")
        engine.require_cuda(dLAFs, "dLAFs")
        dev, st = dLAFs.device, engine.stream_of(dLAFs.device)
        lafs = dLAFs.contiguous().float()
        n = lafs.size(0)
        out = torch.empty(n, 1, PS, PS, dtype=torch.float32, device=dev)
        if n == 0:
            return out
        ids = torch.empty(n, 3, dtype=torch.int32, device=dev)
        norm = torch.empty(n, 2, 3, dtype=torch.float32, device=dev)
        check(lib.affnet_level_select(ctx.handle, ptr(lafs), None, n, PS, ptr(ids), ptr(norm), st), ctx.handle, "affnet_level_select")
        check(lib.affnet_pyr_grid_sample(ctx.handle, ptr(norm), ptr(ids), None, n, PS, ptr(out), st), ctx.handle, "affnet_pyr_grid_sample")
        return out
