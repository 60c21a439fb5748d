#!/usr/bin/env python
"""CPU emulation of the three summations the arithmetic modes of the boundary perform (include/affnet_hip.h AFFNET_ARITH_*), against fp64, on
conv-like dot products (post-ReLU activations of mixed magnitude x weights ~N(0, 0.05)), K = 144 .. 8192:
  fp32 chain    : acc = fma(x_k, w_k, acc) in k order                                 (AFFNET_ARITH_FP32_MFMA, v_mfma_f32_16x16x4_f32)
  bf16 x 3      : x, w as three bf16 terms, six products i + j <= 2 per 32-chunk      (AFFNET_ARITH_FP32_SPLIT3)
  fp16 x 2      : x, 2^e w as two fp16 terms, three products per 32-chunk, times 2^-e (AFFNET_ARITH_FP32_SPLIT2H), and the same WITHOUT the weight scale
A 32-wide chunk of exact products is summed exactly and rounded once into the fp32 accumulator (the matrix instruction's own summation order
inside a k = 32 step is not modelled).  Prints rms and max of |result - fp64| / sum|x||w|.   profiles/r04_s4_split2h_numerics.txt is this output."""
import numpy as np
import torch


def bf16(x):
    return torch.from_numpy(x.astype(np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def split_bf3(x):
    x = x.astype(np.float32); a = bf16(x); r = (x - a).astype(np.float32); b = bf16(r); c = bf16((r - b).astype(np.float32))
    return a, b, c


def split_h2(x):
    x = x.astype(np.float32); a = x.astype(np.float16).astype(np.float32); b = (x - a).astype(np.float32).astype(np.float16).astype(np.float32)
    return a, b


def chunked(xa, wa, pairs, K, M):
    acc = np.zeros(M, np.float32)
    for k0 in range(0, K, 32):
        for (i, j) in pairs:
            acc = (acc + (xa[j][:, k0:k0 + 32].astype(np.float64) @ wa[i][k0:k0 + 32].astype(np.float64))).astype(np.float32)
    return acc


def errors(K, M=4096, rng=None):
    """rms / max of |result - fp64| / sum|x||w| for the four summations on M conv-like dot products of length K."""
    rng = rng or np.random.default_rng(0)
    X = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32) * rng.choice([1e-3, 1, 1, 3], size=(M, K)).astype(np.float32)
    W = (rng.standard_normal(K) * 0.05).astype(np.float32)
    ref = X.astype(np.float64) @ W.astype(np.float64)
    den = np.abs(X).astype(np.float64) @ np.abs(W).astype(np.float64)
    acc = np.zeros(M, np.float32)
    for k in range(K):
        acc = (acc + X[:, k].astype(np.float64) * W[k]).astype(np.float32)            # one rounding per fma
    e32 = np.abs(acc - ref) / den
    e3 = np.abs(chunked(split_bf3(X), split_bf3(W), ((0, 0), (1, 0), (2, 0), (0, 1), (1, 1), (0, 2)), K, M) - ref) / den
    s = 2.0 ** (13 - np.floor(np.log2(np.abs(W).max())))
    e2 = np.abs(chunked(split_h2(X), split_h2(W * np.float32(s)), ((0, 0), (1, 0), (0, 1)), K, M) / np.float32(s) - ref) / den
    e2n = np.abs(chunked(split_h2(X), split_h2(W), ((0, 0), (1, 0), (0, 1)), K, M) - ref) / den
    return {name: (float(np.sqrt((e ** 2).mean())), float(e.max())) for name, e in (("fp32_chain", e32), ("bf16x3", e3), ("fp16x2", e2), ("fp16x2_unscaled", e2n))}


if __name__ == "__main__":
    stream = np.random.default_rng(0)                 # one stream for the whole table
    for K in (144, 288, 576, 1152, 8192):
        r = errors(K, rng=stream)
        print("K %5d | fp32 chain rms %.2e max %.2e | bf16 x 3 rms %.2e max %.2e | fp16 x 2 (weights x 2^e) rms %.2e max %.2e | fp16 x 2 unscaled rms %.2e max %.2e"
              % ((K,) + r["fp32_chain"] + r["bf16x3"] + r["fp16x2"] + r["fp16x2_unscaled"]))
