"""Import the UNMODIFIED reference (/root/reference) in this container.

TEST INFRASTRUCTURE ONLY.  This file exists so that tests/golden/make_golden.py and
oracle/check_restatement.py can run the real reference (Python 3.10 + torch CPU) to
(a) generate the committed golden vectors and (b) pin oracle/affnet_oracle.py
bit-for-bit.  /root/reference does not exist on the GPU box, so nothing under tests
marked gpu, bench.py or __graft_entry__.smoke() may import this module.

Recipe = SURVEY.md Appendix B: stub the two import-only third-party modules
(`cv2` at Utils.py:6,10-11 and `torchvision.transforms` at architectures.py:13),
put the reference on sys.path, load checkpoints with weights_only=False.
"""
import contextlib
import io
import os
import sys
import types

REF_ROOT = os.environ.get("AFFNET_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF_ROOT, "SparseImgRepresenter.py"))


def _install_stubs():
    if "cv2" not in sys.modules:
        cv2 = types.ModuleType("cv2")
        cv2.INTER_LINEAR = 1

        def _resize(x, dsize=None, interpolation=None):  # identity for 32x32 inputs
            if tuple(x.shape[:2]) != tuple(dsize):
                raise NotImplementedError("cv2 stub: resize only supports identity")
            return x

        cv2.resize = _resize
        sys.modules["cv2"] = cv2
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvt = types.ModuleType("torchvision.transforms")
        tv.transforms = tvt
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tvt


def import_reference():
    """Returns a namespace with the reference's hot-path modules."""
    if not available():
        raise RuntimeError("reference not found at %s" % REF_ROOT)
    sys.dont_write_bytecode = True
    _install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import matplotlib
    matplotlib.use("Agg")
    ns = types.SimpleNamespace()
    import SparseImgRepresenter, HandCraftedModules, LAF, Utils, architectures, HardNet
    ns.SparseImgRepresenter = SparseImgRepresenter
    ns.HandCraftedModules = HandCraftedModules
    ns.LAF = LAF
    ns.Utils = Utils
    ns.architectures = architectures
    ns.HardNet = HardNet
    return ns


@contextlib.contextmanager
def quiet():
    """The reference prints stage timings unconditionally (SparseImgRepresenter.py:163-164,197-202)."""
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        yield buf


def load_state_dict(name):
    import torch
    ck = torch.load(os.path.join(REF_ROOT, "pretrained", name), map_location="cpu", weights_only=False)
    return ck["state_dict"]


def load_jit_trace(name):
    """The reference's own TorchScript traces of the two shape / orientation CNNs (convertJIT/AffNetJIT.pt, OriNetJIT.pt), loaded
    where they lie: a second CNN oracle that shares no code with oracle/affnet_oracle.py."""
    import torch
    m = torch.jit.load(os.path.join(REF_ROOT, "convertJIT", name), map_location="cpu")
    m.eval()
    return m


def import_onepass_sir():
    """OnePassSIR.py cannot be imported under Python 3: its forward() holds ONE Python-2 statement
    (`print time.time() - t, 'detection multiscale'`, OnePassSIR.py:144).  The source is read from the reference, that single
    statement is rewritten IN MEMORY and the module is executed - nothing is written anywhere, no reference source enters this
    repository.  Returns the module (class OnePassSIR)."""
    import_reference()
    src = open(os.path.join(REF_ROOT, "OnePassSIR.py")).read()
    bad = "print time.time() - t, 'detection multiscale'"
    assert src.count(bad) == 1, "OnePassSIR.py changed: expected exactly one Python-2 print statement"
    mod = types.ModuleType("OnePassSIR")
    exec(compile(src.replace(bad, "pass"), os.path.join(REF_ROOT, "OnePassSIR.py"), "exec"), mod.__dict__)
    return mod
