"""ScaleSpaceAffinePatchExtractor with the reference's constructor, forward() and
extract_patches_from_pyr() (SparseImgRepresenter.py:14-209), executed on MI355X by
libaffnet_hip.so.

* native slots (affnet_amd.architectures.AffNetFast / OriNetFast, default HessianResp): ONE C call,
  `affnet_extract_features`, runs pyramid -> detector -> AffNet -> filter -> OriNet ->
  denormalise with no host synchronisation; the only read-back is the final row count;
* foreign slots (any callable with the reference's slot signature): the same
  stages are driven one by one through the C ABI and the slot is called on device tensors.

Default slots as in the reference (SparseImgRepresenter.py:42-49): OriNet=None -> OrientationDetector(patch_size=19),
AffNet=None -> AffineShapeEstimator(patch_size=19) (affnet_amd.HandCraftedModules, csrc/handcrafted.hip); a custom
RespNet callable is evaluated per pyramid level and the detector runs on its response maps.  nlevels 1..6 (the reference's default
is 3; nlevels = 1 blurs with a 35 x 35 Gaussian), any init_sigma (init_sigma <= 0.5: octave 0 keeps the unblurred image and its own blur
sequence, HandCraftedModules.py:25-31).
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib, engine
from ._lib import lib, check, ptr
from .architectures import _HipPatchNet
from .HandCraftedModules import AffineShapeEstimator, OrientationDetector, _HipHandCrafted


class ScaleSpaceAffinePatchExtractor(nn.Module):
    def __init__(self, border=16, num_features=500, patch_size=32, mrSize=3.0, nlevels=3, num_Baum_iters=0,
                 init_sigma=1.6, th=None, RespNet=None, OriNet=None, AffNet=None, arith="fp32"):
        super(ScaleSpaceAffinePatchExtractor, self).__init__()
        self.arith = arith           # arithmetic of the native CNN slots' contractions: "fp32" = exact fp32 MFMA (default), "fp32_split3" = fp32 as
                                     # three bf16 terms (exact), six products on the bf16 MFMA, "fp32_split2h" = two fp16 terms (23 of 24 bits), three
                                     # products on the fp16 MFMA; fp32 accumulate (include/affnet_hip.h AFFNET_ARITH_*; same parity bars)
        _lib.arith_code(arith)       # raises on an unknown mode
        self.mrSize, self.PS, self.b = mrSize, patch_size, border
        self.num, self.nlevels = num_features, nlevels
        self.num_Baum_iters, self.init_sigma = num_Baum_iters, init_sigma
        self.th = th
        if th is not None:        # SparseImgRepresenter.py:33-37: a threshold disables the feature budget
            self.num = -1
        else:
            self.th = 0
        self.RespNet = RespNet       # None = the built-in Hessian response fused into the detector kernel; otherwise any callable
                                     # RespNet(level (1,1,h,w), sigma) -> (1,1,h,w) (SparseImgRepresenter.py:38-41), run per level
        if not 1 <= nlevels <= 6:
            raise NotImplementedError("nlevels must be 1..6 (levels per octave = nlevels + 2 <= 8)")
        self.OriNet = OriNet if OriNet is not None else OrientationDetector(patch_size=19)       # SparseImgRepresenter.py:42-45
        self.AffNet = AffNet if AffNet is not None else AffineShapeEstimator(patch_size=19)      # :46-49
        self.scale_pyr = self.sigmas = self.pix_dists = None
        self._ctx = None
        self._ctx_key = None
        self.last_ids = None       # (N,3) int32 (octave, level-1, pixel) of the detections returned last
        self.max_keep = 16384      # row capacity in threshold mode (num = -1)
        self.raw_div = 4           # raw-maxima list capacity per octave = h*w / raw_div (overflow -> AffnetHipError, never truncation)
        self.lazy_shape_rows = -1  # fused path: AffNet first runs on the 1.2 N best candidates, on the rest only if needed (identical rows);
                                   # 0 = all 1.5 N candidates at once like the reference (affnet_config.lazy_shape_rows)

    # ------------------------------------------------------------------------------------------
    def _context(self, x, allow_batch=False):
        engine.require_cuda(x, "image")
        if x.dim() != 4 or x.size(1) != 1 or x.size(0) < 1 or (x.size(0) != 1 and not allow_batch):
            raise ValueError("expected a (1,1,H,W) image; forward() is batch-size-1 like the reference "
                             "(HandCraftedModules.py:283-284) - use enqueue()/run_batch() for (B,1,H,W) batches")
        pre = int(1.5 * self.num) if self.num_Baum_iters > 0 else self.num
        key = (x.size(0), x.size(2), x.size(3), x.device, pre, self.num, float(self.th), self.mrSize, self.b, self.init_sigma,
               self.nlevels, self.max_keep, self.num_Baum_iters, self.raw_div, self.lazy_shape_rows)
        if self._ctx is not None and self._ctx_key == key and self._ctx.arith != _lib.arith_code(self.arith):
            self._ctx.set_arith(self.arith)          # same buffers, other arithmetic: no new context
        if self._ctx is None or self._ctx_key != key:
            self._ctx = engine.Context(x.size(2), x.size(3), x.device, self.nlevels, self.init_sigma, self.b, self.mrSize,
                                       float(self.th), self.num, pre, self.max_keep, batch=x.size(0), baum_iters=self.num_Baum_iters, raw_div=self.raw_div,
                                       lazy_shape_rows=self.lazy_shape_rows, arith=self.arith)
            self._ctx_key = key
        return self._ctx

    def _publish_pyramid(self, ctx):
        self.scale_pyr = ctx.pyramid_views()
        self.sigmas = [list(s) for s in ctx.plan.sigmas]
        self.pix_dists = [list(p) for p in ctx.plan.pix_dists]

    def _response_pyramid(self, ctx, img):
        """Custom RespNet slot: builds the pyramid, evaluates RespNet on every level of every image and returns the response
        maps laid out like the pyramid (what affnet_detect_responses / affnet_detect_image_responses expect)."""
        dev = img.device
        st = engine.stream_of(dev)
        check(lib.affnet_pyramid_build(ctx.handle, ptr(img), st), ctx.handle, "affnet_pyramid_build")
        stride = lib.affnet_pyramid_image_stride(ctx.handle)
        base = lib.affnet_pyramid_level_offset(ctx.handle, 0, 0)
        resp = torch.zeros(ctx.batch * stride, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for b in range(ctx.batch):
                pyr = ctx.pyramid_views(b)
                for o, (h, w) in enumerate(ctx.plan.sizes):
                    for l in range(ctx.plan.levels_per_octave):
                        off = b * stride + lib.affnet_pyramid_level_offset(ctx.handle, o, l) - base
                        r = self.RespNet(pyr[o][l], ctx.plan.sigmas[o][l])
                        resp[off:off + h * w].copy_(r.reshape(-1).to(dev, torch.float32))
        return resp

    @staticmethod
    def _native(slot):
        return slot is None or isinstance(slot, (_HipPatchNet, _HipHandCrafted))

    def enqueue(self, x, do_ori=False, desc=None, det_stream=None, input_ready=None):
        """Enqueues the whole fused path and returns capacity-sized device tensors plus the device row
        counts - no host synchronisation (throughput use).  x may be a (B,1,H,W) batch of equally sized
        images (BASELINE configs[2]; the reference loops over images in Python): every kernel launch then
        covers the B images, results get a leading batch dimension and `count` is (B,).  With `det_stream` (a torch.cuda.Stream) the
        pyramid + detector run there and the CNN stages on the current stream, ordered by events, so that
        two extractor objects alternating over a stream of images overlap the latency-bound detector of
        image i+1 with the MFMA-bound CNN stages of image i.  `input_ready`: None = the detector stream first waits for
        everything enqueued so far on the current stream (safe default: x may still be in flight there); a
        torch.cuda.Event = wait for that event only; False = x is already resident (no wait)."""
        ctx = self._context(x, allow_batch=True)
        dev = x.device
        if not (self._native(self.AffNet) and self._native(self.OriNet)):
            raise NotImplementedError("the fused path needs the native AffNetFast / OriNetFast slots (foreign slots: forward())")
        img = x.contiguous().float()
        B, F = x.size(0), ctx.cap_final
        lafs = torch.empty(B, F, 2, 3, dtype=torch.float32, device=dev)
        resp = torch.empty(B, F, dtype=torch.float32, device=dev)
        ids = torch.empty(B, F, 3, dtype=torch.int32, device=dev)
        count = torch.zeros(B, dtype=torch.int32, device=dev)
        dsc = torch.empty(B, F, 128, dtype=torch.float32, device=dev) if desc is not None else None
        nets = self._nets(dev, do_ori, desc)
        if self.RespNet is not None:
            rmaps = self._response_pyramid(ctx, img)
            check(lib.affnet_detect_image_responses(ctx.handle, ptr(rmaps), engine.stream_of(dev)), ctx.handle, "affnet_detect_image_responses")
            rc = lib.affnet_describe_detected(ctx.handle, C.byref(nets), int(bool(do_ori)), ptr(lafs), ptr(resp), ptr(ids), ptr(dsc),
                                              ptr(count), engine.stream_of(dev))
            check(rc, ctx.handle, "affnet_describe_detected")
        elif det_stream is None:
            rc = lib.affnet_extract_features(ctx.handle, C.byref(nets), ptr(img), int(bool(do_ori)), ptr(lafs), ptr(resp), ptr(ids),
                                             ptr(dsc), ptr(count), engine.stream_of(dev))
            check(rc, ctx.handle, "affnet_extract_features")
        else:
            cur = torch.cuda.current_stream(dev)
            if input_ready is None:
                det_stream.wait_stream(cur)                  # image upload / previous users of x
            elif input_ready is not False:
                det_stream.wait_event(input_ready)
            if getattr(self, "_busy", None) is not None:
                det_stream.wait_event(self._busy)            # workspace still read by the previous describe
            rc = lib.affnet_detect_image(ctx.handle, ptr(img), C.c_void_p(det_stream.cuda_stream))
            check(rc, ctx.handle, "affnet_detect_image")
            ev = torch.cuda.Event()
            ev.record(det_stream)
            cur.wait_event(ev)
            rc = lib.affnet_describe_detected(ctx.handle, C.byref(nets), int(bool(do_ori)), ptr(lafs), ptr(resp), ptr(ids), ptr(dsc),
                                              ptr(count), engine.stream_of(dev))
            check(rc, ctx.handle, "affnet_describe_detected")
            self._busy = torch.cuda.Event()
            self._busy.record(cur)
        if B == 1:      # the reference's shapes
            lafs, resp, ids, dsc = lafs[0], resp[0], ids[0], (None if dsc is None else dsc[0])
        # "overflow": (B,) device view of the capacity-overflow flags (non-zero = a fixed-capacity list overflowed and this image's rows are
        # truncated) - valid once the enqueued work has completed; run() / run_batch() / bench.py check it through affnet_read_counts
        return {"LAFs": lafs, "responses": resp, "ids": ids, "descriptors": dsc, "count": count, "overflow": ctx.counter_view(0), "_img": img}

    def _nets(self, dev, do_ori, desc):
        nets = _lib.Nets()
        if self.num_Baum_iters > 0:
            if isinstance(self.AffNet, _HipHandCrafted):
                nets.h_baumberg_window = C.cast(self.AffNet.window(), C.c_void_p)
            else:
                nets.d_affnet = self.AffNet.packed_weights(dev).data_ptr()
        if do_ori:
            if isinstance(self.OriNet, _HipHandCrafted):
                nets.h_orientation_window = C.cast(self.OriNet.window(), C.c_void_p)
            else:
                nets.d_orinet = self.OriNet.packed_weights(dev).data_ptr()
        nets.d_hardnet = desc.packed_weights(dev).data_ptr() if desc is not None else None
        return nets

    def capture(self, x, do_ori=False, desc=None):
        """Captures the whole fused path for images of x's shape into ONE HIP graph (affnet_graph_capture_extract) and returns a
        CapturedPath: `.image` is the static input tensor, `.launch(x)` copies x into it and replays the graph with a single launch
        on the current stream (returns the same dict as enqueue()), `.run(x)` adds the count read-back and slicing of run().  For
        callers that process one image at a time (hesaffnet.py:35-60): ~45 kernel launches per image become one."""
        return CapturedPath(self, x, do_ori, desc)

    def run_batch(self, x, do_ori=False, desc=None):
        """(B,1,H,W) -> list of B per-image dicts (LAFs px (n_b,2,3), responses, ids, descriptors): one fused
        enqueue for the whole batch, then ONE host read-back of the B row counts."""
        r = self.enqueue(x, do_ori=do_ori, desc=desc)
        ctx = self._ctx
        ctx.read_counts(allow_empty=True)   # surfaces capacity overflow; images without detections are legal in a batch
        cnt = r["count"].cpu().tolist()
        if x.size(0) == 1:
            r = {k: (v.unsqueeze(0) if isinstance(v, torch.Tensor) and k not in ("count", "overflow", "_img") else v) for k, v in r.items()}
        out = []
        for b, n in enumerate(cnt):
            dsc = r["descriptors"]
            out.append({"LAFs": r["LAFs"][b, :n], "responses": r["responses"][b, :n], "ids": r["ids"][b, :n],
                        "descriptors": None if dsc is None else dsc[b, :n]})
        return out

    def run(self, x, do_ori=False, desc=None):
        """Fused path.  Returns dict(LAFs px (N,2,3), responses (N,), ids (N,3), descriptors (N,128)|None).
        `desc`: affnet_amd.HardNet.HardNet or None."""
        if x.dim() == 4 and x.size(0) != 1:
            raise ValueError("run() is for a single (1,1,H,W) image; use run_batch() for (B,1,H,W) batches")
        r = self.enqueue(x, do_ori=do_ori, desc=desc)
        ctx = self._ctx
        self._publish_pyramid(ctx)
        # the one host read-back: raises on capacity overflow and (AffnetEmptyError, "no keypoints detected") on an image without
        # detections, like the reference; its second entry is the row count of the (single) image - no second synchronising .item()
        n = int(ctx.read_counts()[1])
        self.last_ids = r["ids"][:n]
        dsc = r["descriptors"]
        return {"LAFs": r["LAFs"][:n], "responses": r["responses"][:n], "ids": r["ids"][:n],
                "descriptors": None if dsc is None else dsc[:n]}

    # ------------------------------------------------------------------------------------------
    def _staged(self, x, do_ori):
        """Stage-by-stage path for foreign AffNet / OriNet slots (same kernels, slot called on tensors)."""
        ctx = self._context(x)
        dev, st = x.device, engine.stream_of(x.device)
        img = x.contiguous().float()
        P, F = ctx.cap_pre, ctx.cap_final
        rmaps = None
        if self.RespNet is not None:
            rmaps = self._response_pyramid(ctx, img)
        else:
            check(lib.affnet_pyramid_build(ctx.handle, ptr(img), st), ctx.handle, "affnet_pyramid_build")
        self._publish_pyramid(ctx)
        resp = torch.empty(P, dtype=torch.float32, device=dev)
        lafs = torch.empty(P, 2, 3, dtype=torch.float32, device=dev)
        ids = torch.empty(P, 3, dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        if rmaps is not None:
            check(lib.affnet_detect_responses(ctx.handle, ptr(rmaps), ptr(resp), ptr(lafs), ptr(ids), ptr(cnt), st), ctx.handle,
                  "affnet_detect_responses")
        else:
            check(lib.affnet_detect(ctx.handle, ptr(resp), ptr(lafs), ptr(ids), ptr(cnt), st), ctx.handle, "affnet_detect")
        n = int(cnt.item())
        ctx.read_counts()                   # raises AffnetEmptyError ("no keypoints detected") / on capacity overflow
        if self.num_Baum_iters > 0:
            PS = self.AffNet.PS
            patches = torch.empty(n, 1, PS, PS, dtype=torch.float32, device=dev)

            def shape_pass(frames):             # one evaluation of the slot on patches cut along `frames` (P,2,3), rows < n
                check(lib.affnet_pyr_grid_sample(ctx.handle, ptr(frames), ptr(ids), ptr(cnt), n, PS, ptr(patches), st), ctx.handle,
                      "affnet_pyr_grid_sample")
                with torch.no_grad():
                    a = torch.cat([self.AffNet(patches[s:s + 256], {}) for s in range(0, n, 256)], 0)   # Utils.py:37-66
                full = torch.zeros(P, 2, 2, dtype=torch.float32, device=dev)
                full[:n] = a.to(dev, torch.float32)
                return full
            A = shape_pass(lafs)                # base_A = bmm(A_0, I) = A_0
            if self.num_Baum_iters > 1:
                # SparseImgRepresenter.py:127-146: patches re-extracted along [base_A * LAF | centre], base_A = A_i * base_A - the steps
                # the fused path runs between its shape passes (affnet_shape_iterate), here around a foreign slot
                frames = torch.empty(P, 2, 3, dtype=torch.float32, device=dev)
                for _ in range(1, self.num_Baum_iters):
                    check(lib.affnet_shape_iterate(ctx.handle, None, ptr(A), ptr(lafs), ptr(cnt), 0, ptr(frames), st), ctx.handle, "affnet_shape_iterate")
                    Ai = shape_pass(frames)
                    check(lib.affnet_shape_iterate(ctx.handle, ptr(Ai), ptr(A), ptr(lafs), ptr(cnt), 1, ptr(frames), st), ctx.handle, "affnet_shape_iterate")
            Afull = A
            r2 = torch.empty(F, dtype=torch.float32, device=dev)
            l2 = torch.empty(F, 2, 3, dtype=torch.float32, device=dev)
            i2 = torch.empty(F, 3, dtype=torch.int32, device=dev)
            c2 = torch.zeros(1, dtype=torch.int32, device=dev)
            check(lib.affnet_shape_filter_select(ctx.handle, ptr(resp), ptr(lafs), ptr(ids), ptr(Afull), ptr(cnt), ptr(r2), ptr(l2),
                                                 ptr(i2), ptr(c2), st), ctx.handle, "affnet_shape_filter_select")
            n = int(c2.item())
            resp, lafs, ids, cnt = r2, l2, i2, c2
        if do_ori:
            PS = self.OriNet.PS
            patches = torch.empty(n, 1, PS, PS, dtype=torch.float32, device=dev)
            if n:
                check(lib.affnet_pyr_grid_sample(ctx.handle, ptr(lafs), ptr(ids), ptr(cnt), n, PS, ptr(patches), st), ctx.handle,
                      "affnet_pyr_grid_sample")
            with torch.no_grad():
                ang = self.OriNet(patches)
            if ang.dim() <= 2:   # angles -> rotation matrices (LAF.py:306-311)
                c, s = torch.cos(ang).view(-1, 1, 1), torch.sin(ang).view(-1, 1, 1)
                ang = torch.cat([torch.cat([c, s], 2), torch.cat([-s, c], 2)], 1)
            R = ang.to(dev, torch.float32).contiguous()
            if n:
                check(lib.affnet_apply_rotation(ctx.handle, ptr(lafs), ptr(R), ptr(cnt), n, st), ctx.handle, "affnet_apply_rotation")
        out = torch.empty_like(lafs)
        check(lib.affnet_scale_lafs(ctx.handle, ptr(lafs), ptr(out), ptr(cnt), lafs.size(0), x.size(3), x.size(2), 0, st), ctx.handle,
              "affnet_scale_lafs")
        self.last_ids = ids[:n]
        return out[:n], resp[:n]

    def forward(self, x, do_ori=False):
        """x (1,1,H,W) fp32 0..255 on the GPU -> (LAFs in pixels (N,2,3), responses (N,))."""
        aff_native = self.num_Baum_iters == 0 or self._native(self.AffNet)
        ori_native = (not do_ori) or self._native(self.OriNet)
        if aff_native and ori_native:
            r = self.run(x, do_ori=do_ori)
            return r["LAFs"], r["responses"]
        return self._staged(x, do_ori)

    def extract_patches_from_pyr(self, dLAFs, PS=41):
        """Pixel LAFs (N,2,3) -> (N,1,PS,PS) sampled from the best pyramid level of the LAST forward()
        (SparseImgRepresenter.py:181-188; level choice on the device instead of host scipy)."""
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
        self.last_level_ids = ids
        return out


class CapturedPath(object):
    def __init__(self, det, x, do_ori, desc):
        if not (det._native(det.AffNet) and det._native(det.OriNet)) or det.RespNet is not None:
            raise NotImplementedError("graph capture needs the native slots (foreign slots run Python between the stages)")
        ctx = det._context(x, allow_batch=True)
        dev = x.device
        self.det, self.ctx, self.do_ori, self.desc = det, ctx, bool(do_ori), desc
        B, F = x.size(0), ctx.cap_final
        self.image = x.contiguous().float().clone()
        self.out = {"LAFs": torch.empty(B, F, 2, 3, dtype=torch.float32, device=dev), "responses": torch.empty(B, F, dtype=torch.float32, device=dev),
                    "ids": torch.empty(B, F, 3, dtype=torch.int32, device=dev), "count": torch.zeros(B, dtype=torch.int32, device=dev),
                    "descriptors": torch.empty(B, F, 128, dtype=torch.float32, device=dev) if desc is not None else None}
        self._nets = det._nets(dev, do_ori, desc)
        # The graph bakes in the ADDRESSES of the packed weight blobs: hold references (a net whose weights change later builds a new
        # blob and would free the captured one under the graph) and remember what they were built from, so that launch() can refuse to
        # replay stale weights instead of silently doing so.
        self._mods = [m for m in (det.AffNet, det.OriNet if do_ori else None, desc) if m is not None and hasattr(m, "packed_weights")]
        self._blobs = [m.packed_weights(dev) for m in self._mods]
        self._stamps = [m._weights_stamp() for m in self._mods]
        self._arith = ctx.arith                                # the graph bakes in the kernels of this arithmetic mode
        self._stream = torch.cuda.Stream(device=dev)           # capture needs an explicit stream; nothing executes on it
        torch.cuda.synchronize(dev)
        o = self.out
        check(lib.affnet_graph_capture_extract(ctx.handle, C.byref(self._nets), ptr(self.image), int(self.do_ori), ptr(o["LAFs"]), ptr(o["responses"]),
                                               ptr(o["ids"]), ptr(o["descriptors"]), ptr(o["count"]), C.c_void_p(self._stream.cuda_stream)),
              ctx.handle, "affnet_graph_capture_extract")

    def launch(self, x=None, check_weights=True):
        """One graph launch on the current stream; no host synchronisation.  x (same shape as at capture) is copied into `.image` first.
        (The captured path contains only kernels of this library - its fills and copies are kernels too: hipMemsetAsync nodes of a
        captured graph share blit state with eager null-stream memsets on ROCm 7.2 and faulted on replay.)
        check_weights: compare the nets' parameter stamps with the ones at capture (walks every parameter and buffer of up to three nets:
        ~50 us of Python per replay) and refuse to replay stale weights; latency-critical loops whose weights are frozen pass False."""
        if check_weights:
            for m, stamp in zip(self._mods, self._stamps):
                if m._weights_stamp() != stamp:
                    raise RuntimeError("the weights of %s changed after capture(): the graph holds the old packed weights - capture again" % type(m).__name__)
        if self.ctx.arith != self._arith:
            raise RuntimeError("the arithmetic mode of the extractor changed after capture(): the graph runs the mode it was captured with - capture again")
        if x is not None:
            self.image.copy_(x, non_blocking=True)
        busy = getattr(self.det, "_busy", None)
        if busy is not None:                                        # an enqueue() with a detector stream may still read the shared workspace
            torch.cuda.current_stream(self.image.device).wait_event(busy)
        check(lib.affnet_graph_launch(self.ctx.handle, engine.stream_of(self.image.device)), self.ctx.handle, "affnet_graph_launch")
        out = dict(self.out, overflow=self.ctx.counter_view(0), _img=self.image)       # the same keys as enqueue()
        if self.image.size(0) == 1:
            return {k: (v[0] if (v is not None and k not in ("count", "overflow", "_img")) else v) for k, v in out.items()}
        return out

    def run(self, x=None, check_weights=True):
        """launch() + the one count read-back: dict(LAFs px (N,2,3), responses, ids, descriptors) of a single image."""
        if self.image.size(0) != 1:
            raise ValueError("run() is for single images; use launch() for batches")
        r = self.launch(x, check_weights=check_weights)
        self.det._publish_pyramid(self.ctx)
        n = int(self.ctx.read_counts()[1])
        dsc = r["descriptors"]
        return {"LAFs": r["LAFs"][:n], "responses": r["responses"][:n], "ids": r["ids"][:n], "descriptors": None if dsc is None else dsc[:n]}


def get_geometry_and_descriptors(img, det, desc, do_ori=True):
    """train_OriNet_test_on_graffity.py:293-298.  With native nets this is one fused C call."""
    from .HardNet import HardNet
    if isinstance(desc, HardNet) and det._native(det.AffNet) and det._native(det.OriNet):
        r = det.run(img, do_ori=do_ori, desc=desc)
        return r["LAFs"], r["descriptors"]
    with torch.no_grad():
        LAFs, resp = det(img, do_ori=do_ori)
        patches = det.extract_patches_from_pyr(LAFs, PS=32)
        return LAFs, desc(patches)
