"""distance_matrix_vector with the reference's name and meaning (Losses.py:5-13), on the MI355X matrix cores."""
import torch

from . import engine
from ._lib import lib, check, ptr


def distance_matrix_vector(anchor, positive):
    """(n1,128),(n2,128) cuda fp32 -> (n1,n2) sqrt(|a|^2 + |b|^2 - 2 a.b + 1e-6)."""
    engine.require_cuda(anchor, "anchor")
    engine.require_cuda(positive, "positive")
    a, b = anchor.contiguous().float(), positive.contiguous().float()
    n1, n2 = a.size(0), b.size(0)
    out = torch.empty(n1, n2, dtype=torch.float32, device=a.device)
    if n1 and n2:
        scratch = torch.empty(lib.affnet_match_scratch_bytes(n1, n2), dtype=torch.uint8, device=a.device)
        ctx = engine.utility_ctx(a.device)
        rc = lib.affnet_distance_matrix(ctx, ptr(a), n1, ptr(b), n2, a.size(1), ptr(out), ptr(scratch), engine.stream_of(a.device))
        check(rc, ctx, "affnet_distance_matrix")
    return out
