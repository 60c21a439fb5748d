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
        self._packed = None       # (device, tensor)
        self._packed_version = -1
        self._version = 0

    def _bump(self):
        self._version += 1

    def load_state_dict(self, *a, **k):
        r = super(_HipPatchNet, self).load_state_dict(*a, **k)
        self._bump()
        return r

    def _apply(self, fn, *a, **k):
        r = super(_HipPatchNet, self)._apply(fn, *a, **k)
        self._bump()
        return r

    def _weights_stamp(self):
        """Changes whenever a parameter or buffer is replaced (load_state_dict / .to()) or modified in place
        (p.data.copy_(), BN running-stat updates, direct state-dict tensor edits all bump the tensor's `_version`)."""
        return (self._version,) + tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def invalidate_packed(self):
        """Drops the cached BN-folded weight blob (it is rebuilt on the next call)."""
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
        return engine.cnn_forward(self.KIND, self.packed_weights(patches.device), patches)


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
