"""AffNetFast / OriNetFast with the reference's constructor, state-dict keys and call
signature (architectures.py:204-252, :33-82), executed by the fused HIP trunk kernel.

The nn.Sequential below is only a parameter container with the reference's indices
(`features.{0,3,...}.weight`, `features.{1,4,...}.running_*`, `features.19.{weight,bias}`), so
`load_state_dict(torch.load('AffNet.pth')['state_dict'])` works unchanged; it is never executed.
"""
import torch
import torch.nn as nn

from . import _lib, engine


def _container(widths, head_out, head_kernel, head_pad, head_bias, head_bn=False):
    layers, cin = [], 1
    for i, cout in enumerate(widths):
        stride = 2 if i in (2, 4) else 1
        layers += [nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False), nn.BatchNorm2d(cout, affine=False), nn.ReLU()]
        cin = cout
    layers += [nn.Dropout(0.25), nn.Conv2d(cin, head_out, head_kernel, stride=1, padding=head_pad, bias=head_bias)]
    layers += [nn.BatchNorm2d(head_out, affine=False)] if head_bn else [nn.Tanh(), nn.AdaptiveAvgPool2d(1)]
    return nn.Sequential(*layers)


class _HipPatchNet(nn.Module):
    KIND = None

    def __init__(self):
        super(_HipPatchNet, self).__init__()
        self._packed = None            # BN-folded, MFMA-ordered blob on the device
        self.arith = "fp32"            # stand-alone calls of this net: "fp32" (exact fp32 MFMA, default), "fp32_split3" (fp32 = 3 x bf16 split
                                       # operands) or "fp32_split2h" (fp32 = 2 x fp16 split operands; include/affnet_hip.h AFFNET_ARITH_*).  Inside an extractor the extractor's `arith` decides.
        self._packed_version = None    # _weights_stamp() it was built from

    def _weights_stamp(self):
        """Changes whenever a parameter or buffer is replaced (.to(other device), .float()) or modified in place through the tensor itself
        (load_state_dict, nn.init.*_ / no_grad in-place ops on the parameter, BN running-stat updates: they bump the tensor's `_version`).
        NOT detected: writes through `.data` (`p.data.copy_()`, `nn.init.orthogonal(m.weight.data)` - the reference's own weights_init
        style): `.data` has a version counter of its own, the parameter's stays put, and the cached blob would keep serving the old
        weights - call invalidate_packed() after such edits.  A no-op `.to(same device)` - e.g. moving an extractor that holds this
        net - leaves the stamp unchanged, so the packed blob is NOT rebuilt (and never freed) under kernels that may still be reading
        it on another stream."""
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def invalidate_packed(self):
        """Drops the cached BN-folded weight blob (it is rebuilt on the next call).  Required after weight edits through `.data`
        (see _weights_stamp); a CapturedPath taken before must be captured again."""
        self._packed = None

    def packed_weights(self, device):
        """BN-folded, MFMA-ordered weight blob on `device` (cached until the parameters / buffers change)."""
        stamp = self._weights_stamp()
        if self._packed is None or self._packed_version != stamp or self._packed.device != device:
            blob = engine.pack_state_dict(self.KIND, self.state_dict())
            self._packed = blob.to(device)
            self._packed_version = stamp
        return self._packed

    def _run(self, patches):
        if self.training:
            raise RuntimeError("affnet_amd nets are inference-only (call .eval()); training is out of scope")
        engine.require_cuda(patches, "patches")
        return engine.cnn_forward(self.KIND, self.packed_weights(patches.device), patches, arith=self.arith)


class AffNetFast(_HipPatchNet):
    KIND = _lib.NET_AFFNET

    def __init__(self, PS=32):
        super(AffNetFast, self).__init__()
        if PS != 32:
            raise NotImplementedError("the shipped AffNet.pth and the HIP kernel are for PS=32")
        self.features = _container([16, 16, 32, 32, 64, 64], 3, 8, 0, True)
        self.PS = PS
        self.halfPS = int(PS / 2)
        self.eval()

    def forward(self, input, return_A_matrix=False):
        """(n,1,32,32) -> (n,2,2) rectified affine shape.  The second positional argument is the
        kwargs dict batched_forward passes positionally (Utils.py:54)."""
        return self._run(input)


class AffNetFastFullConv(_HipPatchNet):
    """architectures.py:629-674: the fully-convolutional AffNet of the OnePassSIR path.  Same `features` layout as AffNetFast, so
    `load_state_dict(torch.load('AffNet.pth')['state_dict'])` works (the reference ships no dedicated checkpoint).
    forward((1,1,H,W) image, 0..255) -> (1,4,H,W) per-pixel rectified shape (a11, 0, a21, a22); H, W >= 34."""
    KIND = _lib.NET_AFFNET_FULLCONV

    def __init__(self, PS=32, stride=2):
        super(AffNetFastFullConv, self).__init__()
        if PS != 32 or stride != 2:
            raise NotImplementedError("the HIP kernels are specialised for PS=32, stride=2 (the reference's defaults)")
        self.features = _container([16, 16, 32, 32, 64, 64], 3, 8, 0, True)
        self.features = nn.Sequential(*list(self.features.children())[:20])   # the reference's Sequential ends with the 8x8 conv
        self.stride, self.PS = stride, PS
        self.halfPS = int(PS / 2)
        self.eval()

    def forward(self, input, return_A_matrix=False):
        if self.training:
            raise RuntimeError("affnet_amd nets are inference-only (call .eval()); training is out of scope")
        engine.require_cuda(input, "image")
        if input.dim() != 4 or input.size(0) != 1 or input.size(1) != 1:
            raise ValueError("expected a (1,1,H,W) image")
        return engine.fullconv_forward(self.packed_weights(input.device), input, arith=self.arith)


class OriNetFast(_HipPatchNet):
    KIND = _lib.NET_ORINET

    def __init__(self, PS=32):
        super(OriNetFast, self).__init__()
        if PS != 32:
            raise NotImplementedError("the shipped OriNet.pth and the HIP kernel are for PS=32")
        self.features = _container([16, 16, 32, 32, 64, 64], 2, int(PS / 4), 1, True)
        self.PS = PS
        self.halfPS = int(PS / 4)
        self.eval()

    def forward(self, input, return_rot_matrix=True):
        R = self._run(input)
        if return_rot_matrix:
            return R
        return torch.atan2(R[:, 0, 1], R[:, 0, 0])
