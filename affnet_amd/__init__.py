"""affnet_amd - MI355X-native implementation of the ScaleSpaceAffinePatchExtractor hot path of
ducha-aiki/affnet behind the reference's Python API.  All device work lives in
libaffnet_hip.so (affnet_amd/csrc, C ABI in include/affnet_hip.h); importing this package
fails loudly if the library has not been built."""
from . import _lib  # noqa: F401  (raises ImportError when libaffnet_hip.so is missing)
from .SparseImgRepresenter import ScaleSpaceAffinePatchExtractor, get_geometry_and_descriptors  # noqa: F401
from .architectures import AffNetFast, OriNetFast, AffNetFastFullConv  # noqa: F401
from .OnePassSIR import OnePassSIR  # noqa: F401
from .HardNet import HardNet  # noqa: F401
from . import LAF, HandCraftedModules, Losses, ReprojectionStuff  # noqa: F401
from .synthetic import synthetic_image, synthetic_hardnet_state  # noqa: F401

__version__ = "0.1.0"
