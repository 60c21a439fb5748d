/*
 * affnet_hip_probes.h - probe kernels of the tuning / measurement tools under tools/.
 *
 * NOT in libaffnet_hip.so: these entry points exist only in libaffnet_hip_probes.so, the same sources compiled with -DAFFNET_PROBES
 * (`AFFNET_PROBES=1 bash affnet_amd/csrc/build.sh`; __graft_entry__.build() builds it next to the product library so that the tools
 * travel to the GPU box).  The tools select it with AFFNET_HIP_LIB=<path> (affnet_amd/_lib.py).  The shipped library holds product
 * kernels (+ the stamped debug instantiations of affnet_hip_debug.h) only.
 */
#ifndef AFFNET_HIP_PROBES_H
#define AFFNET_HIP_PROBES_H

#include "affnet_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Tuning aid: runs the MFMA loop of one layer in isolation `reps` times per workgroup on `n_blocks` workgroups (same LDS
 * footprint as the trunk kernel).  layer 1 / 5: HardNet conv1 / conv5 (d_packed = HardNet's packed weights); 13 / 14 / 15:
 * AffNet conv3 as 2 x 2 tiles / conv3 as 4 x 1 tiles / conv5 (d_packed = AffNet's).  probe bit 0: no weight loads inside
 * the loop, bit 1: no activation loads, bit 2: accumulators in AGPRs, bit 3: lane-consecutive LDS read pattern (HardNet
 * layers only) - separates matrix-pipe issue efficiency from L2 / LDS effects.  d_out: 2 floats (sink). */
int affnet_cnn32_probe(const float* d_packed, int layer, int probe, int reps, int n_blocks, float* d_out, void* stream);

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
int affnet_split3_gemm(const float* d_A, const float* d_Bt, int M, int N, int K, int mode, float* d_C, void* stream);
int affnet_split3_rate(int reps, int terms, int n_blocks, float* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AFFNET_HIP_PROBES_H */
