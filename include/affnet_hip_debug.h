/*
 * affnet_hip_debug.h - debug, self-test and tuning entry points of libaffnet_hip.so.
 *
 * NOT part of the drop-in boundary (include/affnet_hip.h): nothing here replaces a reference
 * function.  These symbols exist for the parity tests (layer-by-layer activation dumps, MFMA
 * fragment-layout self-test) and for the measurement tools under tools/ (in-kernel phase stamps,
 * isolated MFMA loops, known-byte-count streaming kernels that calibrate the rocprofv3
 * FETCH_SIZE / WRITE_SIZE counters).  A product build may drop them.
 */
#ifndef AFFNET_HIP_DEBUG_H
#define AFFNET_HIP_DEBUG_H

#include "affnet_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Parity aid: runs the trunk of `net_kind` on ONE patch (d_patch (32,32) fp32) and copies the activation tensor after
 * trunk layer `layer` (0..5, post BN+ReLU, (C,H,W) fp32) to d_out.  Localises a kernel bug to one layer of
 * architectures.py:207-229 / HardNet.py:67-89.  Exact-fp32 contexts only (AFFNET_ARITH_FP32_MFMA): the split-operand trunks have no
 * per-layer dump, a context in another arithmetic mode gets AFFNET_ERR_INVALID instead of the exact path's activations. */
int affnet_cnn32_debug_layer(affnet_ctx* ctx, int net_kind, const float* d_packed, const float* d_patch,
                             int layer, float* d_out, void* stream);

/* Tuning aid, PER CONTEXT: while d_stamps != NULL every CNN launch made through `ctx` runs the stamped instantiation of
 * the trunk kernel and writes s_memtime stamps [patch][wave][32] (uint64) at its phase boundaries (0 start, 1 input
 * ready, 2 conv0 done, then per conv layer k = 1..5: 2k+1 MFMA loop done, 2k+2 outputs stored; 14 / 15 = HW_ID / XCC_ID;
 * 16..22 sub-phases of the input stage).  NULL switches it off.  Other contexts are not affected. */
int affnet_cnn32_debug_timing(affnet_ctx* ctx, unsigned long long* d_stamps);

/* Tuning aid: runs the MFMA loop of one layer in isolation `reps` times per workgroup on `n_blocks` workgroups (same LDS
 * footprint as the trunk kernel).  layer 1 / 5: HardNet conv1 / conv5 (d_packed = HardNet's packed weights); 13 / 14 / 15:
 * AffNet conv3 as 2 x 2 tiles / conv3 as 4 x 1 tiles / conv5 (d_packed = AffNet's).  probe bit 0: no weight loads inside
 * the loop, bit 1: no activation loads, bit 2: accumulators in AGPRs, bit 3: lane-consecutive LDS read pattern (HardNet
 * layers only) - separates matrix-pipe issue efficiency from L2 / LDS effects.  d_out: 2 floats (sink). */
int affnet_cnn32_probe(const float* d_packed, int layer, int probe, int reps, int n_blocks, float* d_out, void* stream);

/* 16x16x4 fp32 MFMA layout self-test: d_out (16,16) = A (16,4) * B (4,16). */
int affnet_selftest_mfma(const float* d_A, const float* d_B, float* d_out, void* stream);

/* Counter calibration (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern"):
 * streams exactly n_bytes (a multiple of 64 KiB) with `width` bytes per lane per load (4, 8 or 16), fully coalesced.
 *   mode 0: read d_src, reduce, one 4-byte store per workgroup to d_dst     (known READ bytes  = n_bytes)
 *   mode 1: write d_dst with a lane pattern, no reads                        (known WRITE bytes = n_bytes)
 *   mode 2: 64 x 64 fp32 tiles with a (halo)-pixel apron through LDS like the blur / Hessian tile loaders: image
 *           (n_bytes / 4 / 4096 rows of 4096 px), 4-byte loads, known unique bytes = n_bytes (halo re-reads hit L2)
 * Run under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE`; tools/fetch_calib.py turns the counters into the per-width
 * factors that tools/pmc_traffic.py applies. */
int affnet_debug_stream(const void* d_src, void* d_dst, size_t n_bytes, int width, int mode, int halo, void* stream);

/* Numerics / rate probes of the split-operand arithmetic (AFFNET_ARITH_FP32_SPLIT3): fp32 operands as three bf16 terms on v_mfma_f32_16x16x32_bf16 (csrc/split_probe.hip).
 *   affnet_split3_gemm: d_C (M x N) = d_A (M x K) * d_Bt^T (d_Bt: N x K), M, N multiples of 16, K of 32.  mode 0 = the exact-fp32
 *     v_mfma_f32_16x16x4_f32 chain (today's arithmetic), 1 = six split terms, 2 = nine, 3 = the leading bf16 term only.
 *   affnet_split3_rate: sustained rate of the inner-loop shape a trunk layer would have (fragments from LDS, 4 pixel tiles x 1 channel
 *     tile); terms = 6 / 9 on bf16 MFMA, 1 = the fp32 16x16x4 loop over the same tiles.  One launch = n_blocks x 8 waves x reps x 4 tiles
 *     x (16 x 16 x 32) multiply-adds.  d_out: 2 floats (sink). */
/* Tuning aid for AFFNET_ARITH_FP32_SPLIT3 (include/affnet_hip.h: affnet_set_arith is the product switch): variant bits of the
 * split-operand trunk launches of this context.  bit 0 (value 1): alternating wave priorities in the HardNet loops (A/B aid of
 * tools/s3_net_timing.py; results identical; default off).  bit 1 (value 2): the HardNet loops of AFFNET_ARITH_FP32_SPLIT2H skip their
 * NaN -> +inf step (A/B of its cost only, tools/ab_nan_step.py: an out-of-range activation could then be hidden by a ReLU).  Does not
 * change the arithmetic mode. */
int affnet_debug_split3_variant(affnet_ctx* ctx, int bits);
int affnet_split3_gemm(const float* d_A, const float* d_Bt, int M, int N, int K, int mode, float* d_C, void* stream);
int affnet_split3_rate(int reps, int terms, int n_blocks, float* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AFFNET_HIP_DEBUG_H */
