"""HardNet descriptor with the reference's state-dict layout (HardNet.py:61-101), executed by the
fused HIP trunk kernel + head GEMM (BN and L2 normalisation fused)."""
from . import _lib
from .architectures import _HipPatchNet, _container


class HardNet(_HipPatchNet):
    KIND = _lib.NET_HARDNET

    def __init__(self):
        super(HardNet, self).__init__()
        self.features = _container([32, 32, 64, 64, 128, 128], 128, 8, 0, False, head_bn=True)
        self.features[18].p = 0.1
        self.PS = 32
        self.eval()

    def forward(self, input):
        """(n,1,32,32) -> (n,128) L2-normalised descriptors."""
        return self._run(input)
