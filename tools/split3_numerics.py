"""EXPLORATORY: fp32 = three bf16 terms.  CPU emulation of an fp32 dot product computed as 6 or 9 bf16 x bf16
products with fp32 accumulation (what v_mfma_f32_16x16x32_bf16 would do on operands split x = x0 + x1 + x2, each term rounded to bf16)
against the exact-fp32 chain the product path uses today and against float64.  Shapes: the HardNet head (K = 8192) and a 3x3 conv
over 128 channels (K = 1152), operands with the magnitudes of standardised activations / BN-folded weights.
    python tools/split3_numerics.py            -> one JSON line"""
import json

import numpy as np


def bf16_round(x):
    """fp32 -> nearest-even bf16, returned as fp32."""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def split3(x):
    x0 = bf16_round(x)
    r1 = (x - x0).astype(np.float32)          # exact in fp32
    x1 = bf16_round(r1)
    r2 = (r1 - x1).astype(np.float32)
    x2 = bf16_round(r2)
    return x0, x1, x2, float(np.abs((r2 - x2)).max() / max(np.abs(x).max(), 1e-30))


def dot_fp32_chain(a, b):
    """sequential fmaf chain in fp32 (the exact-fp32 MFMA path), vectorised over rows."""
    acc = np.zeros(a.shape[0], dtype=np.float32)
    for k in range(a.shape[1]):
        acc = (acc.astype(np.float64) + a[:, k].astype(np.float64) * b[k]).astype(np.float32)      # fma: one rounding
    return acc


def dot_split(a, b, terms):
    """sum over the listed (i, j) term pairs of bf16 x bf16 products (exact in fp32), accumulated in fp32 in blocks of 32 k
    (one MFMA step: products of a block summed in float64 here - the hardware's internal tree is at least fp32 - then rounded)."""
    A, B = split3(a)[:3], split3(b)[:3]
    acc = np.zeros(a.shape[0], dtype=np.float32)
    for k0 in range(0, a.shape[1], 32):
        for i, j in terms:
            blk = (A[i][:, k0:k0 + 32].astype(np.float64) * B[j][k0:k0 + 32].astype(np.float64)).sum(axis=1)
            acc = (acc.astype(np.float64) + blk).astype(np.float32)
    return acc


def main():
    rng = np.random.default_rng(0)
    six = [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]
    nine = six + [(1, 2), (2, 1), (2, 2)]
    out = {"what": "relative error of a K-term dot product vs float64, max over 256 rows: exact-fp32 fmaf chain (today's MFMA path) vs bf16 3-way split "
                   "operands with 6 / 9 product terms and fp32 accumulation per 32-k MFMA step (CPU emulation)", "cases": []}
    for name, K, sa, sb in (("HardNet head (K = 8192): ReLU activations x BN-folded weights", 8192, 1.0, 0.02),
                            ("3x3 conv over 128 channels (K = 1152)", 1152, 1.0, 0.05)):
        a = np.maximum(rng.standard_normal((256, K)).astype(np.float32) * sa, 0).astype(np.float32)
        b = (rng.standard_normal(K) * sb).astype(np.float32)
        ref = a.astype(np.float64) @ b.astype(np.float64)
        scale = np.abs(a.astype(np.float64)) @ np.abs(b.astype(np.float64))          # sum |a||b|: the natural error scale of a dot product
        rec = {"case": name, "split_residual_rel": split3(a)[3]}
        for label, val in (("fp32_chain", dot_fp32_chain(a, b)), ("split_6_terms", dot_split(a, b, six)), ("split_9_terms", dot_split(a, b, nine)),
                           ("bf16_1_term", dot_split(a, b, [(0, 0)]))):
            err = np.abs(val.astype(np.float64) - ref)
            rec[label] = {"max_abs": float(err.max()), "max_rel_to_sum_abs": float((err / scale).max())}
        out["cases"].append(rec)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
