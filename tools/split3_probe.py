"""EXPLORATORY (never the headline): fp32 operands as three bf16 terms on the bf16 matrix cores.
On the MI355X: (1) numerics of C = A B^T at the HardNet head's K = 8192 and a conv layer's K = 1152 - exact-fp32 MFMA chain (today's
arithmetic) vs 6 / 9 split terms vs plain bf16, each against float64; (2) sustained rate of the inner-loop shape a trunk layer would
have on split operands (6 / 9 terms) next to the fp32 16x16x4 loop over the same tiles.  Prints one JSON line.
    python tools/split3_probe.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("AFFNET_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "affnet_amd", "libaffnet_hip_probes.so"))   # probe kernels live there (include/affnet_hip_probes.h)
from affnet_amd._lib import lib, ptr  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    out = {"what": "fp32 = 3 x bf16 split operands on v_mfma_f32_16x16x32_bf16: numerics and loop-rate probes (the product kernels of AFFNET_ARITH_FP32_SPLIT3 are the trunks)", "numerics": [], "rate": []}
    for name, K, sb in (("HardNet head GEMM, K = 8192", 8192, 0.02), ("3x3 conv over 128 channels, K = 1152", 1152, 0.05)):
        A = torch.clamp(torch.randn(256, K, generator=g), min=0).contiguous()
        Bt = (torch.randn(128, K, generator=g) * sb).contiguous()
        ref = A.double() @ Bt.double().t()
        scale = A.double().abs() @ Bt.double().abs().t()
        rec = {"case": name}
        Ad, Bd = A.to(dev), Bt.to(dev)
        for mode, label in ((0, "fp32_mfma_chain"), (1, "split_6_terms"), (2, "split_9_terms"), (3, "bf16_1_term")):
            C = torch.zeros(256, 128, device=dev)
            assert lib.affnet_split3_gemm(ptr(Ad), ptr(Bd), 256, 128, K, mode, ptr(C), None) == 0
            torch.cuda.synchronize()
            err = (C.cpu().double() - ref).abs()
            rec[label] = {"max_abs": float(err.max()), "max_rel_to_sum_abs": float((err / scale).max())}
        # the fp32 chain must be what the CPU's sequential fmaf chain gives
        out["numerics"].append(rec)
    sink = torch.zeros(2, device=dev)
    reps, blocks = 4000, 256
    for terms in (1, 6, 9):
        for _ in range(2):
            assert lib.affnet_split3_rate(reps, terms, blocks, ptr(sink), None) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            assert lib.affnet_split3_rate(reps, terms, blocks, ptr(sink), None) == 0
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        macs = blocks * 8 * reps * 4 * 16 * 16 * 32                      # fp32-EQUIVALENT multiply-adds (the split terms are overhead, not work)
        out["rate"].append({"terms": terms, "ms": ms, "fp32_equivalent_tflops": 2.0 * macs / (ms * 1e-3) / 1e12,
                            "what": "fp32 16x16x4 MFMA loop" if terms == 1 else "%d bf16 16x16x32 MFMAs per tile and 32 k" % terms})
    base = out["rate"][0]["fp32_equivalent_tflops"]
    for r in out["rate"]:
        r["vs_fp32_loop"] = r["fp32_equivalent_tflops"] / base
    print(json.dumps(out))


if __name__ == "__main__":
    main()
