"""Thin Python owner of libaffnet_hip contexts, workspaces and packed weights.
PyTorch is used for device memory and streams only."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr
from .host_plan import PyramidPlan

_UTILITY = {}


def require_cuda(t, what="tensor"):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError("affnet_amd: %s must live on the MI355X (tensor.cuda()); this implementation has no CPU path" % what)


def utility_ctx(device, arith=0):
    """Context without a pyramid, for stand-alone stage calls (one per device and arithmetic mode)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, _lib.arith_code(arith))
    if key not in _UTILITY:
        h = C.c_void_p()
        check(lib.affnet_ctx_create(C.byref(h), idx, None), None, "affnet_ctx_create(utility)")
        check(lib.affnet_set_arith(h, key[1]), h, "affnet_set_arith")
        _UTILITY[key] = h
    h = _UTILITY[key]
    if lib.affnet_get_arith(h) != key[1]:       # a tool switched the shared handle in place (affnet_set_arith) and did not switch back
        check(lib.affnet_set_arith(h, key[1]), h, "affnet_set_arith")
    return h


def stream_of(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Context(object):
    """One extractor instance on one device: config, workspace, pyramid views."""

    def __init__(self, height, width, device, n_levels, init_sigma, border, mr_size, threshold,
                 num_features, num_prefilter, max_keep=16384, batch=1, baum_iters=0, raw_div=4, onepass=False, lazy_shape_rows=-1, arith=0):
        self.plan = PyramidPlan(height, width, n_levels, init_sigma, border)
        self.batch = int(batch)
        self.onepass = bool(onepass)
        self.cfg = self.plan.fill_config(mr_size, threshold, num_features, num_prefilter, max_keep, raw_div=raw_div, batch=self.batch, baum_iters=baum_iters,
                                         onepass=onepass, lazy_shape_rows=lazy_shape_rows, arith=arith)
        self.arith = _lib.arith_code(arith)
        self.device = device
        self.handle = C.c_void_p()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        rc = lib.affnet_ctx_create(C.byref(self.handle), idx, C.byref(self.cfg))
        check(rc, self.handle, "affnet_ctx_create")
        nbytes = lib.affnet_workspace_bytes(self.handle)
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=device)
        check(lib.affnet_bind_workspace(self.handle, ptr(self.workspace), nbytes), self.handle, "affnet_bind_workspace")
        self.cap_pre = lib.affnet_capacity_prefilter(self.handle)
        self.cap_final = lib.affnet_capacity_final(self.handle)
        self._ws_f32 = self.workspace.view(torch.float32)

    def set_arith(self, arith):
        """Switches the arithmetic of this context's CNN contractions (affnet_set_arith); 'fp32' restores the default bit for bit."""
        self.arith = _lib.arith_code(arith)
        check(lib.affnet_set_arith(self.handle, self.arith), self.handle, "affnet_set_arith")

    def pyramid_views(self, image=0):
        """scale_pyr[o][l] of image `image` of the batch as (1,1,h,w) views into the workspace
        (SparseImgRepresenter.py:55)."""
        pyr = []
        base = image * lib.affnet_pyramid_image_stride(self.handle)
        for o, (h, w) in enumerate(self.plan.sizes):
            levels = []
            for l in range(self.plan.levels_per_octave):
                off = base + lib.affnet_pyramid_level_offset(self.handle, o, l)
                levels.append(self._ws_f32[off:off + h * w].view(1, 1, h, w))
            pyr.append(levels)
        return pyr

    def affmap_views(self, image=0):
        """OnePassSIR contexts: the dense affine-shape map of every octave of image `image` as (1,4,h,w) views into the workspace."""
        base = image * lib.affnet_affmap_image_stride(self.handle)
        out = []
        for o, (h, w) in enumerate(self.plan.sizes):
            off = base + lib.affnet_affmap_offset(self.handle, o)
            out.append(self._ws_f32[off:off + 4 * h * w].view(1, 4, h, w))
        return out

    def counter_view(self, which=0):
        """(B,) int32 device view of a per-image counter in the workspace (0 = capacity-overflow flag, 1 = rows after detection, 2 = rows
        after the shape filter): callers of enqueue() can test it on the device or copy it asynchronously - no host synchronisation."""
        off, stride = lib.affnet_counter_offset(self.handle, which), lib.affnet_counter_stride(self.handle)
        return self.workspace.view(torch.int32)[off:off + stride * self.batch:stride]

    def read_counts(self, allow_empty=False):
        """The one host read-back: [rows after detection, rows after the shape filter, overflow flag, raw maxima] summed over
        the batch.  Raises AffnetHipError on a capacity overflow (results would be truncated) and AffnetEmptyError when no
        image produced a detection (allow_empty=True: return the zeros instead - batches may legitimately hold flat images)."""
        out = (C.c_int32 * 4)()
        rc = lib.affnet_read_counts(self.handle, C.byref(out), stream_of(self.device))
        if not (allow_empty and rc == _lib.ERR_EMPTY):
            check(rc, self.handle, "affnet_read_counts")
        return list(out)

    def __del__(self):
        try:
            if self.handle:
                lib.affnet_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def pack_state_dict(kind, sd):
    """state dict (reference key layout) -> packed fp32 blob (CPU tensor)."""
    conv_idx, bn_idx = (0, 3, 6, 9, 12, 15), (1, 4, 7, 10, 13, 16)

    def f32(name):
        return sd[name].detach().to("cpu", torch.float32).contiguous()

    keep = []
    convs = [f32("features.%d.weight" % i) for i in conv_idx]
    means = [f32("features.%d.running_mean" % i) for i in bn_idx]
    vars_ = [f32("features.%d.running_var" % i) for i in bn_idx]
    head_w = f32("features.19.weight")
    head_b = f32("features.19.bias") if "features.19.bias" in sd else None
    if kind == _lib.NET_AFFNET_FULLCONV and head_b is None:
        raise ValueError("AffNetFastFullConv: features.19.bias missing from the state dict")
    hbm = f32("features.20.running_mean") if kind == _lib.NET_HARDNET else None
    hbv = f32("features.20.running_var") if kind == _lib.NET_HARDNET else None
    keep += convs + means + vars_ + [head_w, head_b, hbm, hbv]
    arr = lambda ts: (C.c_void_p * 6)(*[t.data_ptr() for t in ts])
    n = lib.affnet_cnn32_packed_floats(kind)
    out = torch.empty(n, dtype=torch.float32)
    rc = lib.affnet_cnn32_pack_weights(kind, arr(convs), arr(means), arr(vars_), ptr(head_w), ptr(head_b), ptr(hbm), ptr(hbv), ptr(out))
    check(rc, None, "affnet_cnn32_pack_weights")
    del keep
    return out


def save_flat_weights(kind, sd, path):
    """state dict (reference key layout) -> the flat AFNW0001 file a non-Python caller of the C ABI feeds to affnet_cnn32_pack_weights
    (examples/c_host/extract.c): magic, int32 kind, int32 float count, then float32 conv weights features.{0,3,6,9,12,15}, BN running means
    features.{1,4,7,10,13,16}, BN running variances, head features.19.weight, features.19.bias (AffNet / OriNet) or features.20.running_mean /
    running_var (HardNet)."""
    conv_idx, bn_idx = (0, 3, 6, 9, 12, 15), (1, 4, 7, 10, 13, 16)
    names = ["features.%d.weight" % i for i in conv_idx] + ["features.%d.running_mean" % i for i in bn_idx] + \
            ["features.%d.running_var" % i for i in bn_idx] + ["features.19.weight"]
    names += ["features.20.running_mean", "features.20.running_var"] if kind == _lib.NET_HARDNET else ["features.19.bias"]
    flat = np.concatenate([sd[n].detach().to("cpu", torch.float32).contiguous().numpy().reshape(-1) for n in names])
    with open(path, "wb") as f:
        f.write(b"AFNW0001")
        f.write(np.array([kind, flat.size], dtype=np.int32).tobytes())
        f.write(flat.astype(np.float32).tobytes())
    return flat.size


def cnn_forward(kind, packed, patches, scratch=None, arith=0):
    """patches (n,1,32,32) or (n,32,32) cuda fp32 -> (n,2,2) / (n,128)."""
    require_cuda(patches, "patches")
    if patches.dim() == 4:
        if patches.size(1) != 1:
            raise ValueError("expected single-channel patches")
        patches = patches[:, 0]
    if tuple(patches.shape[1:]) != (32, 32):
        raise ValueError("the HIP CNN kernels are specialised for 32x32 patches, got %s" % (tuple(patches.shape),))
    patches = patches.contiguous().float()
    n = patches.size(0)
    dev = patches.device
    out = torch.empty((n, 128) if kind == _lib.NET_HARDNET else (n, 2, 2), dtype=torch.float32, device=dev)
    if n == 0:
        return out
    if scratch is None:       # HardNet: conv5 tensors + split-K partials of the head GEMM; AffNet / OriNet: per-wave head partials
        scratch = torch.empty(n * ((8192 + 512) if kind == _lib.NET_HARDNET else 144), dtype=torch.float32, device=dev)
    ctx = utility_ctx(dev, arith)
    rc = lib.affnet_cnn32_forward(ctx, kind, ptr(packed), ptr(patches), None, n, ptr(out), ptr(scratch), stream_of(dev))
    check(rc, ctx, "affnet_cnn32_forward")
    return out


def fullconv_forward(packed, img, arith=0):
    """(1,1,H,W) cuda fp32 -> (1,4,H,W): dense AffNetFastFullConv map (architectures.py:666-674)."""
    img = img.contiguous().float()
    h, w = img.size(2), img.size(3)
    nbytes = lib.affnet_fullconv_scratch_bytes(h, w)
    if nbytes == 0:
        raise ValueError("image %dx%d too small for AffNetFastFullConv (LocalNorm2d(33) needs H, W >= 34; the reference raises too)" % (w, h))
    dev = img.device
    scratch = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    out = torch.empty(1, 4, h, w, dtype=torch.float32, device=dev)
    ctx = utility_ctx(dev, arith)
    check(lib.affnet_fullconv_forward(ctx, ptr(packed), ptr(img), h, w, ptr(out), ptr(scratch), stream_of(dev)), ctx, "affnet_fullconv_forward")
    return out


def local_norm(img):
    """LocalNorm2d(33) (architectures.py:21-31) of a (1,1,H,W) cuda image."""
    require_cuda(img, "image")
    img = img.contiguous().float()
    out = torch.empty_like(img)
    ctx = utility_ctx(img.device)
    check(lib.affnet_local_norm(ctx, ptr(img), ptr(out), img.size(2), img.size(3), stream_of(img.device)), ctx, "affnet_local_norm")
    return out


def base_grid(ps):
    buf = (C.c_float * ps)()
    check(lib.affnet_host_base_grid(ps, buf), None, "affnet_host_base_grid")
    return np.array(buf, dtype=np.float32)
