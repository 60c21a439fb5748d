/*
 * affnet_hip_debug.h - debug, self-test and tuning entry points of libaffnet_hip.so.
 *
 * NOT part of the drop-in boundary (include/affnet_hip.h): nothing here replaces a reference
 * function.  These symbols exist for the parity tests (layer-by-layer activation dumps, MFMA
 * fragment-layout self-test) and for the in-kernel phase stamps of tools/cnn_phase_timing.py / s3_phase_timing.py.  The
 * probe KERNELS of the tuning tools (isolated MFMA loops, counter-calibration streams, split-arithmetic
 * GEMM / rate probes) are not in libaffnet_hip.so at all: include/affnet_hip_probes.h,
 * libaffnet_hip_probes.so (AFFNET_PROBES=1 bash affnet_amd/csrc/build.sh).
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

/* 16x16x4 fp32 MFMA layout self-test: d_out (16,16) = A (16,4) * B (4,16). */
int affnet_selftest_mfma(const float* d_A, const float* d_B, float* d_out, void* stream);

/* Tuning aid for AFFNET_ARITH_FP32_SPLIT3 (include/affnet_hip.h: affnet_set_arith is the product switch): variant bits of the
 * split-operand trunk launches of this context.  bit 0 (value 1): alternating wave priorities in the HardNet loops (A/B aid of
 * tools/s3_net_timing.py; results identical; default off).  bit 1 (value 2): the HardNet loops of AFFNET_ARITH_FP32_SPLIT2H skip their
 * NaN -> +inf step (A/B of its cost only, tools/ab_nan_step.py: an out-of-range activation could then be hidden by a ReLU).  Does not
 * change the arithmetic mode. */
int affnet_debug_split3_variant(affnet_ctx* ctx, int bits);

#ifdef __cplusplus
}
#endif
#endif /* AFFNET_HIP_DEBUG_H */
