/*
 * affnet_hip.h - C ABI of libaffnet_hip.so: the MI355X (gfx950) implementation of the
 * ScaleSpaceAffinePatchExtractor hot path of ducha-aiki/affnet.
 *
 * The reference has no FFI: its "plugin API" for this path is Python duck typing
 * (SURVEY.md section 8b).  Each entry point below names the reference function(s) it replaces
 * (file:line relative to the reference repo).  The host-side mirror that binds these
 * symbols with ctypes is affnet_amd/_lib.py; INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *  - every `d_*` pointer is a DEVICE pointer owned by the caller (torch tensors in the
 *    Python mirror); the library never allocates or frees device memory;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); every call only
 *    enqueues work on it and returns - no host synchronisation, no host read-back, except
 *    the explicitly named affnet_read_counts();
 *  - all functions return AFFNET_OK (0) or a negative error code; affnet_last_error(ctx)
 *    returns a human readable message for the last failure on that context;
 *  - a context is not re-entrant: one thread / one stream at a time per context (the
 *    reference object is stateful in the same way, SparseImgRepresenter.py:55); DIFFERENT contexts may be
 *    driven from different host threads / streams concurrently - the library keeps no process-global state;
 *  - a context is bound to one HIP device (affnet_ctx_create): every entry point makes that device current for
 *    the duration of the call and restores the caller's current device before returning, so a multi-GPU process
 *    does not have to hipSetDevice around library calls.  Pointers must belong to the context's device;
 *  - batching: a context created with cfg->batch = B processes B equally sized images per call.  Every
 *    per-image array then has a leading batch dimension: d_img (B,H,W); row arrays (B,cap,...) with image
 *    b's rows at b*cap (cap = affnet_capacity_prefilter / _final as documented per entry point, or the
 *    n_max argument); d_count (B).  Rows >= count[b] of image b are zero.  B = 1 is the reference's shape;
 *  - all image / LAF / descriptor arithmetic is IEEE fp32, compiled with
 *    -ffp-contract=off; fused multiply-adds are used only where the reference's CPU
 *    kernels use them ("bit-exact detector").
 */
#ifndef AFFNET_HIP_H
#define AFFNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFFNET_OK 0
#define AFFNET_ERR_INVALID (-1)   /* bad argument / unsupported configuration          */
#define AFFNET_ERR_HIP (-2)       /* a HIP runtime call failed (message has details)   */
#define AFFNET_ERR_CAPACITY (-3)  /* a fixed-capacity device list overflowed           */
#define AFFNET_ERR_EMPTY (-4)     /* no detections in any image of the call: returned by
                                   * affnet_read_counts (reference: torch.cat([]) raises,
                                   * SparseImgRepresenter.py:100)                       */

#define AFFNET_MAX_OCTAVES 16
#define AFFNET_MAX_LEVELS 8       /* n_levels + 2 <= 8                                 */
#define AFFNET_MAX_TAPS 37        /* Gaussian kernels up to 37 x 37 (nlevels = 1: 35)  */

/* Network kinds for the 32x32-patch CNNs. */
#define AFFNET_NET_AFFNET 0       /* architectures.py:204-252 AffNetFast               */
#define AFFNET_NET_ORINET 1       /* architectures.py:33-82   OriNetFast               */
#define AFFNET_NET_HARDNET 2      /* HardNet.py:61-101        HardNet                  */
#define AFFNET_NET_AFFNET_FULLCONV 3 /* architectures.py:629-674 AffNetFastFullConv: same state-dict layout as AffNetFast (the
                                      * shipped AffNet.pth loads unchanged), evaluated densely on whole images (OnePassSIR)   */

/* Arithmetic of the CNN contractions (the 3x3 conv layers conv1..conv5 of AffNetFast / OriNetFast / HardNet / AffNetFastFullConv and the
 * HardNet 8x8 head; architectures.py:204-252,33-82,629-674, HardNet.py:61-101).  Everything else - pyramid, detector, sampler, conv0, the
 * AffNet / OriNet heads, normalisations - is IEEE fp32 in both modes.
 *   FP32_MFMA   (default): v_mfma_f32_16x16x4_f32, every product and sum in fp32 (an fmaf chain in k order).
 *   FP32_SPLIT3 : every fp32 operand x is carried as x = x0 + x1 + x2 with each term rounded to bf16 (exact: 3 x 8 significand bits);
 *                 a product w * a is the six bf16 products w_i * a_j with i + j <= 2 on v_mfma_f32_16x16x32_bf16 (each exact in the
 *                 fp32 accumulator), accumulated in fp32.  Operands stay fp32 at every interface (weights, activations between layers,
 *                 outputs); the dropped terms are <= 2^-24 relative per product, i.e. this is another fp32 summation, not narrower
 *                 arithmetic (K = 8192 dot product: 2.6e-7 of sum|a||b| vs 2.1e-7 for the fmaf chain).  Results
 *                 differ from FP32_MFMA by summation order only (descriptors ~1e-6); the parity bars are the same for both modes.
 *   FP32_SPLIT2H: every fp32 operand x is carried as x ~ h + l with h = fp16(x), l = fp16(x - h) (round to nearest even both; x - h is
 *                 exact): 11 + 11 significand bits + the sign of the remainder = 23 of fp32's 24 bits, |x - h - l| <= 2^-23 |x| (rms
 *                 ~2^-24.4) for |x| >= 2^-2; the operands are NOT scaled per element, so below 2^-2 the low term fp16(x - h) is a subnormal
 *                 fp16 and the bound becomes ABSOLUTE: |x - h - l| <= 2^-25 whatever |x| (2^-22 |x| at 2^-3, 2^-17 |x| at 2^-8).  A product w * a is the THREE fp16 products w_h a_h + w_l a_h + w_h a_l on v_mfma_f32_16x16x32_f16 (each exact
 *                 in the fp32 accumulator; the dropped w_l a_l is <= 2^-24 relative), accumulated in fp32 - half the matrix instructions
 *                 of FP32_SPLIT3.  NOT bit-faithful fp32 operands like FP32_SPLIT3, but the same error class as an fp32 summation: a
 *                 K-term dot product against fp64 on the same data: 2.0e-8 rms of sum|w||a| vs 3.6e-8 for the fmaf chain of FP32_MFMA
 *                 and 2.4e-8 for FP32_SPLIT3 (the accumulation roundings dominate all three).  fp16 has a 5-bit
 *                 exponent, so the weights of a layer are packed times a power of two 2^e that puts the layer's largest |w| into
 *                 [2^13, 2^14) and the layer's sums are multiplied by 2^-e (both exact); activations are used as they are: |a| < 65504
 *                 is REQUIRED (the reference's nets are BatchNorm-ed, |a| stays below ~50; a larger value becomes inf and the output
 *                 non-finite - HardNet descriptors NaN, AffNet / OriNet outputs NaN or saturated - it does not pass as a plausible number) and an activation below 2^-2 keeps an
 *                 ABSOLUTE error of 2^-25 instead of a relative one, as stated above (subnormal fp16 operands are honoured by the MFMA,
 *                 tools/probes/f16_split_probe.hip).  Against the O(1) sums of the BatchNorm-ed layers of the shipped nets an absolute 2^-25
 *                 per activation is fp32-level; a net whose activations are UNIFORMLY small (all |a| << 2^-2) loses relative precision in
 *                 this mode - use AFFNET_ARITH_FP32_SPLIT3 (exact operands) or the default there.  Same interfaces (fp32
 *                 everywhere), same parity bars. */
#define AFFNET_ARITH_FP32_MFMA 0
#define AFFNET_ARITH_FP32_SPLIT3 1
#define AFFNET_ARITH_FP32_SPLIT2H 2

typedef struct affnet_ctx affnet_ctx;

/*
 * Static description of one extractor instance.  The host mirror fills it with the
 * reference's own formulas (ScalePyramid.__init__/forward, HandCraftedModules.py:14-56;
 * CircularGaussKernel, Utils.py:92-114) so that no Gaussian / sigma formula is baked into
 * device code (SURVEY.md section 7 "hard parts": py2-vs-py3 semantics live on the host).
 */
typedef struct affnet_config {
    int32_t height, width;                 /* input image size                                         */
    int32_t n_octaves;                     /* result of the stop rule (HandCraftedModules.py:50)       */
    int32_t levels_per_octave;             /* nLevels + 2 (5)                                          */
    int32_t oct_h[AFFNET_MAX_OCTAVES];     /* octave sizes: avg_pool2d(k=1,s=2) => ceil(h/2)           */
    int32_t oct_w[AFFNET_MAX_OCTAVES];
    float level_sigma[AFFNET_MAX_OCTAVES][AFFNET_MAX_LEVELS];  /* sigmas[o][l]  (centroid weights)    */
    float level_sigma4[AFFNET_MAX_OCTAVES][AFFNET_MAX_LEVELS]; /* float32(sigma**4) (HessianResp)     */
    double level_sigma_px[AFFNET_MAX_OCTAVES][AFFNET_MAX_LEVELS]; /* sigma*pix_dist, float64 (LAF.py:459) */
    int32_t first_blur_taps;               /* 0 = no initial blur (init_sigma <= 0.5)                  */
    float first_blur[AFFNET_MAX_TAPS * AFFNET_MAX_TAPS];       /* k x k, row-major                    */
    int32_t level_blur_taps[AFFNET_MAX_LEVELS];                /* blur producing level l (l>=1)       */
    float level_blur[AFFNET_MAX_LEVELS][AFFNET_MAX_TAPS * AFFNET_MAX_TAPS];
    /* Octave 0 only, when its blurs differ from the later octaves' (init_sigma <= 0.5: octave 0 starts at curSigma = 0.5,
     * every later octave restarts at init_sigma, HandCraftedModules.py:25-31,49).  level_blur0_taps[1] == 0: octave 0 uses
     * level_blur like every other octave (the usual case, init_sigma > 0.5). */
    int32_t level_blur0_taps[AFFNET_MAX_LEVELS];
    float level_blur0[AFFNET_MAX_LEVELS][AFFNET_MAX_TAPS * AFFNET_MAX_TAPS];
    int32_t lazy_shape_rows;               /* fused path, AffNetFast, one shape iteration: AffNet first runs on this many of the response-sorted
                                            * candidates and on the rest only if fewer than N of them survive the shape filter (device-side
                                            * decision, identical output rows).  0 = evaluate all C candidates at once like the reference,
                                            * < 0 = default (1.2 N)                                                              */
    int32_t onepass;                       /* != 0: context for the OnePassSIR path (OnePassSIR.py:14-153): the workspace also holds a
                                            * dense affine-shape map per octave + the dense net's scratch; num_prefilter must equal
                                            * num_features (no shape-filter stage) and every octave must be >= 34 px (LocalNorm2d(33)
                                            * reflect-pads by 16: the reference's scripts use border = 15)                       */
    float mr_size;                         /* mrSize (ctor kwarg, SparseImgRepresenter.py:19)          */
    float threshold;                       /* th; responses are clamp(resp - th, 0) (:77)              */
    int32_t num_features;                  /* N; <= 0: keep everything (th given => num = -1, :33-35)  */
    int32_t num_prefilter;                 /* C = int(1.5 N) if Baumberg iters > 0 else N (:192-194)   */
    int32_t max_raw_per_octave_div;        /* raw-maxima capacity of octave o = h*w / div (default 4)  */
    int32_t max_keep;                      /* capacity of the selected list when N <= 0                */
    int32_t batch;                         /* images per call, all of size height x width (<= 0: 1).
                                            * The reference is batch-size-1 (HandCraftedModules.py:283-284) and
                                            * loops in Python; here one launch covers the batch (BASELINE configs[2]). */
    int32_t baum_iters;                    /* num_Baum_iters (ctor kwarg, SparseImgRepresenter.py:22): AffNet shape iterations with
                                            * re-extraction (:127-146); <= 0: 1 when AffNet weights are passed, else none    */
    int32_t arith;                         /* AFFNET_ARITH_*: arithmetic of the CNN contractions of this context's calls (0 = exact fp32
                                            * MFMA, the default); may be changed later with affnet_set_arith                        */
} affnet_config;

/* Fills *cfg for an height x width image with the reference's own formulas, so that a caller that is not Python can create a context
 * without re-deriving them: the pyramid stop rule (minSize = 2 border + 3), octave sizes, level sigmas, the incremental blur sigmas and
 * their k x k Gaussian tap tables (k = int(6 sigma + 1) | 1, taps at linspace(-k/2, k/2, k): Python-3 true division), sigma^4 and
 * sigma x pixel distance.  Replaces ScalePyramid.__init__ / forward's bookkeeping (HandCraftedModules.py:14-56) and
 * CircularGaussKernel / GaussianBlur.calculate_weights (Utils.py:92-114,155-161).  All intermediates are doubles evaluated like the
 * numpy expressions of the reference (affnet_amd/host_plan.py is the Python mirror's restatement; a CPU test compares the two structs
 * byte for byte).  Arguments = the ScaleSpaceAffinePatchExtractor ctor kwargs (SparseImgRepresenter.py:15-24): n_levels (nlevels, 3),
 * init_sigma (1.6), border, mr_size (mrSize), threshold (th; 0 when None), num_features (N; <= 0 with a threshold),
 * num_prefilter (C = int(1.5 N) when baum_iters > 0, else N), batch (images per call), baum_iters (num_Baum_iters).
 * The remaining fields get their defaults (max_raw_per_octave_div 4, max_keep 16384, lazy_shape_rows -1, onepass 0,
 * arith AFFNET_ARITH_FP32_MFMA) and may be changed before affnet_ctx_create.
 * Returns AFFNET_ERR_INVALID for n_levels outside 1..AFFNET_MAX_LEVELS-2, more than AFFNET_MAX_OCTAVES octaves or a Gaussian of more than
 * AFFNET_MAX_TAPS taps (there is no context yet to hold a message). */
int affnet_config_fill(affnet_config* cfg, int height, int width, int n_levels, double init_sigma, int border, double mr_size,
                       double threshold, int num_features, int num_prefilter, int batch, int baum_iters);

/* ---- context ------------------------------------------------------------------------------- */

/* Creates a context bound to HIP device `device` (one context per (device, stream) user).
 * cfg == NULL creates a utility context for the stand-alone stage calls that do not touch the
 * pyramid (gauss_blur, hessian_response, laf_grid_sample, cnn32_forward, apply_rotation, scale_lafs). */
int affnet_ctx_create(affnet_ctx** out, int device, const affnet_config* cfg);
void affnet_ctx_destroy(affnet_ctx* ctx);
const char* affnet_last_error(const affnet_ctx* ctx);
/* Library / build identification, e.g. "affnet_hip 0.1 gfx950". */
const char* affnet_version(void);
/* Arithmetic mode (AFFNET_ARITH_*) of the CNN contractions launched through this context from now on (utility contexts too).  The
 * packed weight blobs serve all modes; switching back to AFFNET_ARITH_FP32_MFMA restores the default path bit for bit.  A graph
 * captured earlier (affnet_graph_capture_extract) keeps the mode it was captured with. */
int affnet_set_arith(affnet_ctx* ctx, int arith);
int affnet_get_arith(const affnet_ctx* ctx);

/* Bytes of caller-owned device workspace the context needs (pyramid + detector lists +
 * CNN scratch).  Offsets into it are exposed so the host mirror can present the pyramid as
 * tensors (`scale_pyr`, SparseImgRepresenter.py:55). */
size_t affnet_workspace_bytes(const affnet_ctx* ctx);
/* d_workspace must be 256-byte aligned device memory of the context's device (checked with hipPointerGetAttributes). */
int affnet_bind_workspace(affnet_ctx* ctx, void* d_workspace, size_t bytes);
/* Float offset (from the workspace base) of pyramid level (o, l) of image 0; -1 if out of range. */
int64_t affnet_pyramid_level_offset(const affnet_ctx* ctx, int octave, int level);
/* Floats between the pyramids of consecutive images of the batch. */
int64_t affnet_pyramid_image_stride(const affnet_ctx* ctx);
/* Images per call (cfg->batch). */
int affnet_batch(const affnet_ctx* ctx);
/* Maximum number of rows the detector / shape stages can emit (C and N capacities). */
int affnet_capacity_prefilter(const affnet_ctx* ctx);
int affnet_capacity_final(const affnet_ctx* ctx);

/* ---- stage entry points -------------------------------------------------------------------- */

/* Gaussian blur of one image: replicate padding + full 2-D k x k cross-correlation
 * accumulated with fmaf in row-major tap order (bit-identical to the reference's conv2d on
 * CPU).  Replaces Utils.py:150-166 (GaussianBlur.forward).  `h_taps` is a HOST pointer to
 * k*k floats. */
int affnet_gauss_blur(affnet_ctx* ctx, const float* d_in, float* d_out, int h, int w,
                      const float* h_taps, int k, void* stream);

/* Whole scale pyramid into the bound workspace.  Replaces ScalePyramid.forward,
 * HandCraftedModules.py:23-56.  d_img: (H, W) fp32, 0..255. */
int affnet_pyramid_build(affnet_ctx* ctx, const float* d_img, void* stream);

/* Hessian response of one level (for the RespNet slot / tests).  Replaces
 * HessianResp.forward, HandCraftedModules.py:74-78.  sigma4 = float32(sigma**4). */
int affnet_hessian_response(affnet_ctx* ctx, const float* d_in, float* d_out, int h, int w,
                            float sigma4, void* stream);

/* Detector on the pyramid in the workspace: Hessian response, clamp(resp - th, 0), 3-D NMS,
 * border zeroing, octaveMap masking (uint8 wrap emulated), response-weighted 27-tap
 * centroid, per-level and global top-k (C = num_prefilter), x mrSize.
 * Replaces SparseImgRepresenter.py:53-111,198 + HandCraftedModules.py:208-291 +
 * Utils.py:116-148 + LAF.py:431-441.
 * Outputs (capacity affnet_capacity_prefilter rows; rows >= count are zero):
 *   d_resp (cap) fp32, d_lafs (cap,2,3) fp32 normalised LAFs (A scaled by mrSize),
 *   d_ids (cap,3) int32 = (octave, level-1 "prevBlur", flat pixel index),
 *   d_count (1) int32 number of valid rows.
 * Row order = the reference's: descending response when more than C candidates exist,
 * otherwise (octave, level, pixel) order. */
int affnet_detect(affnet_ctx* ctx, float* d_resp, float* d_lafs, int32_t* d_ids,
                  int32_t* d_count, void* stream);

/* Same detector on RESPONSE maps supplied by the caller - the RespNet slot of the reference (SparseImgRepresenter.py:24,38-41:
 * any callable RespNet(level (1,1,h,w), sigma) -> (1,1,h,w)).  d_responses is laid out like the pyramid in the workspace
 * (affnet_pyramid_level_offset - offset of level (0,0), affnet_pyramid_image_stride); clamp(r - th, 0) is applied here. */
int affnet_detect_responses(affnet_ctx* ctx, const float* d_responses, float* d_resp, float* d_lafs, int32_t* d_ids,
                            int32_t* d_count, void* stream);

/* Affine patch sampler: affine_grid + bilinear grid_sample, zeros padding,
 * align_corners=False, including the reference's fp32 coordinate round trip.
 * Replaces LAF.py:313-324,326-372 (generate_patch_grid_from_normalized_LAFs,
 * batched_grid_apply, extract_patches).  d_img (h,w) one level; d_lafs (n,2,3) normalised;
 * d_out (n, ps, ps). */
int affnet_laf_grid_sample(affnet_ctx* ctx, const float* d_img, int h, int w,
                           const float* d_lafs, int n, int ps, float* d_out, void* stream);

/* Same, gathering each patch from pyramid level d_ids[i] = (octave, level, *) of the bound
 * workspace; only rows < *d_count are written (d_count may be NULL => n_max rows).
 * Replaces LAF.py:376-404 (inverted index + per-level extraction). */
int affnet_pyr_grid_sample(affnet_ctx* ctx, const float* d_lafs, const int32_t* d_ids,
                           const int32_t* d_count, int n_max, int ps, float* d_out, void* stream);

/* ---- CNNs ---------------------------------------------------------------------------------- */

/* Number of floats of the packed (BN-folded, MFMA-ordered) weight blob of a network. */
size_t affnet_cnn32_packed_floats(int net_kind);
/* Packs a state dict on the HOST.  conv_w[i] (i=0..5): trunk conv weights (Cout,Cin,3,3);
 * bn_mean[i], bn_var[i]: running stats (eps 1e-5, affine=False) folded into weight+bias;
 * head_w / head_b: final conv (AffNet (3,64,8,8)+bias, OriNet (2,64,8,8)+bias,
 * HardNet (128,128,8,8), no bias, followed by BN head_bn_mean/var).
 * h_out receives affnet_cnn32_packed_floats(kind) floats; the caller uploads them.
 * Replaces the nn.Sequential definitions architectures.py:207-229,36-59, HardNet.py:67-89. */
int affnet_cnn32_pack_weights(int net_kind, const float* const* conv_w, const float* const* bn_mean,
                              const float* const* bn_var, const float* head_w, const float* head_b,
                              const float* head_bn_mean, const float* head_bn_var, float* h_out);

/* Forward of one network on n 32x32 patches (d_patches (n,32,32) fp32, raw intensities; the
 * per-patch mean / unbiased-std normalisation is fused).
 *   AffNet : d_out (n,2,2) rectified affine shape      (architectures.py:246-252, LAF.py:285-291)
 *   OriNet : d_out (n,2,2) rotation matrix             (architectures.py:76-82,  LAF.py:276-283)
 *   HardNet: d_out (n,128) L2-normalised descriptor    (HardNet.py:98-101)
 * Only rows < *d_count are computed when d_count != NULL.  d_scratch: HardNet n*(8192+512) floats (conv5 tensor = A
 * operand of the head GEMM + its split-K partials), AffNet / OriNet n*144 floats (per-wave partial sums of the head). */
int affnet_cnn32_forward(affnet_ctx* ctx, int net_kind, const float* d_packed, const float* d_patches,
                         const int32_t* d_count, int n_max, float* d_out, float* d_scratch, void* stream);

/* Same, but each 32x32 input patch is sampled on the fly from the pyramid in the workspace
 * (fused affnet_pyr_grid_sample + affnet_cnn32_forward; no patch tensor in HBM). */
int affnet_cnn32_forward_pyr(affnet_ctx* ctx, int net_kind, const float* d_packed, const float* d_lafs,
                             const int32_t* d_ids, const int32_t* d_count, int n_max, float* d_out,
                             float* d_scratch, void* stream);

/* ---- fully-convolutional AffNet + 2-D NMS (the OnePassSIR path) ------------------------------- */

/* LocalNorm2d(33): (x - mean33) / (sqrt|E33[x^2] - mean33^2| + 1e-10) clamped to [-6, 6], reflect padding; the 33 x 33 box sums are
 * accumulated in the reference's CPU order (bit-identical sums; the normalised value agrees to 1 ulp).  Replaces
 * architectures.py:21-31.  h, w >= 17. */
int affnet_local_norm(affnet_ctx* ctx, const float* d_in, float* d_out, int h, int w, void* stream);

/* Bytes of device scratch affnet_fullconv_forward needs for an h x w image (0 if the image is too small: h, w >= 34). */
size_t affnet_fullconv_scratch_bytes(int h, int w);

/* Dense affine-shape map of one image: d_img (h, w) fp32 0..255 -> d_out (4, h, w) planar (a11, 0, a21, a22) = the rectified
 * per-pixel shape matrix.  d_packed: affnet_cnn32_pack_weights(AFFNET_NET_AFFNET_FULLCONV, ...) blob on the device.
 * Replaces AffNetFastFullConv.forward, architectures.py:666-674 (LocalNorm2d, reflect pad 14, six conv+BN+ReLU, 8x8 head,
 * bilinear upsampling, tanh, rectifyAffineTransformationUpIsUpFullyConv LAF.py:293-297). */
int affnet_fullconv_forward(affnet_ctx* ctx, const float* d_packed, const float* d_img, int h, int w, float* d_out, float* d_scratch,
                            void* stream);

/* NMS2d: out = x where x - maxpool3x3(x) + 1e-5 > 0 (and x > threshold when threshold > 1e-5), else 0.
 * Replaces HandCraftedModules.py:194-206 (as shipped its constructor raises under Python 3: `padding = kernel_size/2`). */
int affnet_nms2d(affnet_ctx* ctx, const float* d_in, float* d_out, int h, int w, float threshold, void* stream);

/* ---- hand-crafted slot fillers --------------------------------------- */

#define AFFNET_HC_ORIENTATION 0   /* HandCraftedModules.py:133-192 OrientationDetector(patch_size=19)    */
#define AFFNET_HC_BAUMBERG 1      /* HandCraftedModules.py:81-132  AffineShapeEstimator(patch_size=19)   */

/* n 19x19 patches (d_patches (n,19,19) fp32) -> d_out (n,2,2): rotation matrix [[cos,sin],[-sin,cos]] of the dominant
 * gradient orientation (kind 0; d_angles (n) optional) or the rectified Baumberg shape matrix (kind 1).
 * h_weights: HOST pointer to the 19x19 Gaussian window (kind 0: 10 * CircularGaussKernel(kernlen=19); kind 1:
 * CircularGaussKernel(kernlen=19, sigma=19/2/3), Utils.py:92-114 - computed by the host mirror). */
int affnet_handcrafted_forward(affnet_ctx* ctx, int kind, const float* d_patches, int n, const float* h_weights, float* d_out,
                               float* d_angles, void* stream);
/* Same, each patch sampled from the pyramid in the workspace (rows < d_count[image]). */
int affnet_handcrafted_forward_pyr(affnet_ctx* ctx, int kind, const float* d_lafs, const int32_t* d_ids, const int32_t* d_count,
                                   int n_max, const float* h_weights, float* d_out, void* stream);

/* ---- LAF stages ---------------------------------------------------------------------------- */

/* base_A = A; new_LAF = [A * LAF_2x2 | centre]; keep rows with 1/6 < |l1/(l2+1e-8)| < 6 and
 * all four frame corners inside [0,1]^2; if more than N survive take the N largest
 * responses (bad rows zeroed first), else keep survivors in order.
 * Replaces SparseImgRepresenter.py:121-162 (num_Baum_iters == 1), Utils.py:168-175,
 * LAF.py:91-104.  Inputs have *d_count_in valid rows (capacity C); outputs capacity N
 * (or C when N <= 0). */
int affnet_shape_filter_select(affnet_ctx* ctx, const float* d_resp_in, const float* d_lafs_in,
                               const int32_t* d_ids_in, const float* d_A, const int32_t* d_count_in,
                               float* d_resp_out, float* d_lafs_out, int32_t* d_ids_out,
                               int32_t* d_count_out, void* stream);

/* One step of the iterated shape estimation (num_Baum_iters > 1, SparseImgRepresenter.py:127-146), rows < d_count[image] of
 * capacity affnet_capacity_prefilter(ctx):
 *   mode 0: d_lafs_out = [d_base * LAF_2x2 | centre]                                   (the frames the next patches are cut from)
 *   mode 1: d_base = d_A * d_base (in place), then d_lafs_out as above                 (d_A = the shape net's output on those patches)
 * d_A may be NULL in mode 0. */
int affnet_shape_iterate(affnet_ctx* ctx, const float* d_A, float* d_base, const float* d_lafs, const int32_t* d_count, int mode,
                         float* d_lafs_out, void* stream);

/* LAF_2x2 <- LAF_2x2 * R for rows < *d_count.  SparseImgRepresenter.py:173-177. */
int affnet_apply_rotation(affnet_ctx* ctx, float* d_lafs, const float* d_R, const int32_t* d_count,
                          int n_max, void* stream);

/* LAFs <- LAFs * [[m,m,W],[m,m,H]] (inverse=0) or / (inverse=1), m = min(H,W).
 * LAF.py:407-429 (denormalizeLAFs / normalizeLAFs). */
int affnet_scale_lafs(affnet_ctx* ctx, const float* d_in, float* d_out, const int32_t* d_count, int n_max,
                      int w, int h, int inverse, void* stream);

/* Pixel LAFs (n,2,3) -> Oxford ellipses (n,5) = x y a b c with [a b; b c] = (A A^T)^-1 through the closed-form 2x2 SVD.
 * Replaces LAF.py:35-51 (LAFs2ellT) + :106-144 (bsvd2x2).  Rows >= count are zero. */
int affnet_lafs_to_ellipses(affnet_ctx* ctx, const float* d_lafs, const int32_t* d_count, int n_max, float* d_out, void* stream);

/* Pyramid level for descriptor patches: argmin over (o,l) of |sigma[o][l]*2^o - sqrt|det A|/PS|
 * in float64, first minimum wins; also writes normalised LAFs (by pyr[0][0] size).
 * Replaces LAF.py:450-472 (host scipy cdist round trip) + SparseImgRepresenter.py:181-188.
 * d_lafs_px (n,2,3) pixel LAFs -> d_ids (n,3) (octave, level, 0), d_lafs_norm (n,2,3). */
int affnet_level_select(affnet_ctx* ctx, const float* d_lafs_px, const int32_t* d_count, int n_max,
                        int ps, int32_t* d_ids, float* d_lafs_norm, void* stream);

/* ---- descriptor matching ------------------------------------------ */

/* Bytes of device scratch affnet_match_snn / affnet_distance_matrix need for n1 x n2 descriptors. */
size_t affnet_match_scratch_bytes(int n1, int n2);

/* Full distance matrix sqrt(|a|^2 + |b|^2 - 2 a.b + 1e-6): d_out (n1, n2).  Replaces Losses.py:5-13
 * (distance_matrix_vector).  dim must be 128. */
int affnet_distance_matrix(affnet_ctx* ctx, const float* d_desc1, int n1, const float* d_desc2, int n2, int dim, float* d_out,
                           void* d_scratch, void* stream);

/* Nearest / "second nearest" neighbour ratio test exactly as train_AffNet_test_on_graffity.py:292-300 does it (the second
 * minimum runs over the columns that are nobody's nearest neighbour: dist_matrix[:, idxs_in_2] = 100000); the distance
 * matrix is never materialised.  Outputs: d_min_dist (n1), d_idx (n1), d_min2_dist (n1), d_tent (n1,2) int32 = the pairs
 * (i, idx[i]) with min/(min2+1e-8) <= snn_threshold in row order, d_count (1). */
int affnet_match_snn(affnet_ctx* ctx, const float* d_desc1, int n1, const float* d_desc2, int n2, int dim, float snn_threshold,
                     float* d_min_dist, int32_t* d_idx, float* d_min2_dist, int32_t* d_tent, int32_t* d_count, void* d_scratch,
                     void* stream);

/* Pixel LAFs (n,2,3) through the homography h_H (HOST pointer, 9 floats row-major): centre = H x / z, shape = linH(H, x) A.
 * Replaces ReprojectionStuff.py:9-40 (linH, reprojectLAFs). */
int affnet_reproject_lafs(affnet_ctx* ctx, const float* d_lafs, int n, const float* h_H, float* d_out, void* stream);

/* For every query LAF (image-1 LAFs): nearest reference LAF centre (reprojected image-2 LAFs), distance computed like
 * ReprojectionStuff.py:78-86 on the 2-D centres (sqrt(| |a|^2 + |p|^2 - 2 p.a | + 1e-12), fp32).
 * Replaces the distance / min part of ReprojectionStuff.py:126-137 (get_GT_correspondence_indexes). */
int affnet_centre_nn(affnet_ctx* ctx, const float* d_query_lafs, int nq, const float* d_ref_lafs, int nr, float* d_min_dist,
                     int32_t* d_idx, void* stream);

/* ---- fused pipeline ------------------------------------------------------------------------ */

typedef struct affnet_nets {
    const float* d_affnet;   /* packed AffNet weights, NULL => num_Baum_iters = 0            */
    const float* d_orinet;   /* packed OriNet weights, NULL => do_ori must be 0              */
    const float* d_hardnet;  /* packed HardNet weights, NULL => no descriptors               */
    /* The reference's default slot fillers (SparseImgRepresenter.py:42-49), used when the CNN of the slot is NULL:       */
    const float* h_orientation_window;  /* HOST, 361 floats: OrientationDetector(19) when do_ori and d_orinet == NULL     */
    const float* h_baumberg_window;     /* HOST, 361 floats: AffineShapeEstimator(19) when baum_iters > 0, d_affnet NULL   */
} affnet_nets;

/* Whole path, image resident -> results resident, zero host synchronisation:
 * pyramid -> detect -> [AffNet shape + filter] -> [OriNet] -> denormalise ->
 * [level select -> sample -> HardNet].
 * Replaces ScaleSpaceAffinePatchExtractor.forward (SparseImgRepresenter.py:189-209) +
 * extract_patches_from_pyr (:181-188) + HardNet forward = get_geometry_and_descriptors
 * (train_OriNet_test_on_graffity.py:293-298).
 * Outputs (capacity affnet_capacity_final rows): d_lafs_px (cap,2,3), d_resp (cap),
 * d_ids (cap,3) (octave, level-1, pixel) of the detection, d_desc (cap,128) or NULL,
 * d_count (1). */
int affnet_extract_features(affnet_ctx* ctx, const affnet_nets* nets, const float* d_img, int do_ori,
                            float* d_lafs_px, float* d_resp, int32_t* d_ids, float* d_desc,
                            int32_t* d_count, void* stream);

/* The two halves of affnet_extract_features, so that a caller can software-pipeline images over two
 * streams (detector of image i+1 next to the CNN stages of image i; the caller orders the streams with
 * events and must not start affnet_detect_image on a context before the previous
 * affnet_describe_detected on it has finished):
 *   affnet_detect_image      = pyramid + detector into the context's internal candidate list;
 *   affnet_describe_detected = AffNet shape + filter, OriNet, denormalise, level select, HardNet. */
int affnet_detect_image(affnet_ctx* ctx, const float* d_img, void* stream);
/* Detector half on caller-supplied response maps (custom RespNet slot; the pyramid must have been built with
 * affnet_pyramid_build): candidates go to the internal list consumed by affnet_describe_detected. */
int affnet_detect_image_responses(affnet_ctx* ctx, const float* d_responses, void* stream);
int affnet_describe_detected(affnet_ctx* ctx, const affnet_nets* nets, int do_ori, float* d_lafs_px, float* d_resp,
                             int32_t* d_ids, float* d_desc, int32_t* d_count, void* stream);

/* OnePassSIR detector half (context created with cfg->onepass): replaces OnePassSIR.multiScaleDetectorAff + the x mrSize of
 * OnePassSIR.forward (OnePassSIR.py:53-115,146), NMS3dAndComposeAAff (HandCraftedModules.py:292-363), sc_y_x_and_A2LAFs
 * (LAF.py:442-449) and the boundary test of OnePassSIR.py:91.
 *   d_img != NULL: the pyramid is built first; NULL: it has been built with affnet_pyramid_build;
 *   d_packed_fullconv != NULL (affnet_cnn32_pack_weights(AFFNET_NET_AFFNET_FULLCONV, ...)): the dense affine-shape map of every
 *   octave is computed from its level 0 (OnePassSIR.py:69); NULL: the caller has written the maps of a foreign dense AffNet,
 *   planar (4, h_o, w_o) per octave, at affnet_affmap_offset(ctx, o) (+ affnet_affmap_image_stride per image).
 * Follow with affnet_describe_detected(nets->d_affnet = NULL, ...): orientation, denormalisation, descriptors. */
int affnet_detect_image_onepass(affnet_ctx* ctx, const float* d_packed_fullconv, const float* d_img, void* stream);
/* Same with a custom RespNet slot (OnePassSIR.py:24,38-41,71: RespNet(level, sigma) per pyramid level): the pyramid has been built
 * with affnet_pyramid_build and d_responses holds the slot's response maps laid out like the pyramid (affnet_pyramid_level_offset /
 * affnet_pyramid_image_stride); only clamp(r - th, 0) is applied to them. */
int affnet_detect_image_onepass_responses(affnet_ctx* ctx, const float* d_packed_fullconv, const float* d_responses, void* stream);

/* Copies the context's internal detection list (what affnet_detect_image / _responses / _onepass produced and
 * affnet_describe_detected consumes) into caller buffers of capacity affnet_capacity_prefilter(ctx) rows per image: responses,
 * NORMALISED LAFs (x mrSize), (octave, level, pixel) ids, row counts.  For callers that run a foreign (Python) OriNet / AffNet /
 * descriptor between the stages (SparseImgRepresenter.py:38-49, OnePassSIR.py:44-47) on exactly the rows the fused path would use. */
int affnet_detected_list(affnet_ctx* ctx, float* d_resp, float* d_lafs, int32_t* d_ids, int32_t* d_count, void* stream);
/* Float offset (from the workspace base) of the (4, h_o, w_o) affine-shape map of octave o of image 0, and the floats between the
 * maps of consecutive images; -1 / 0 when the context has no OnePassSIR areas. */
int64_t affnet_affmap_offset(const affnet_ctx* ctx, int octave);
int64_t affnet_affmap_image_stride(const affnet_ctx* ctx);

/* The whole path as ONE HIP graph.  affnet_graph_capture_extract stream-captures affnet_extract_features with exactly these buffers
 * (same arguments; `stream` must be an explicit, non-null stream; nothing executes during capture) and instantiates the graph;
 * affnet_graph_launch replays it: one launch instead of ~45 for callers that process one image at a time
 * (hesaffnet.py:35-60 per image).  The captured buffers (image, outputs, workspace, packed weights) must stay alive and in place;
 * new image content is copied into the captured d_img before a launch.  A later capture on the same context replaces the graph. */
int affnet_graph_capture_extract(affnet_ctx* ctx, const affnet_nets* nets, const float* d_img, int do_ori, float* d_lafs_px, float* d_resp,
                                 int32_t* d_ids, float* d_desc, int32_t* d_count, void* stream);
int affnet_graph_launch(affnet_ctx* ctx, void* stream);

/* Stage timing with HIP events recorded on the caller's stream around the stages of
 * affnet_extract_features (no host synchronisation while enabled; a ring of 256 calls).
 * Stages: 0 pyramid, 1 detector, 2 AffNet trunk(+sampling), 3 shape filter/select, 4 OriNet(+rotation),
 * 5 denormalise + level select (with a native OriNet and descriptors this work is done by OriNet's finish kernel, i.e. inside
 * stage 4, and stage 5 is empty), 6 HardNet trunk (+sampling), 7 HardNet head GEMM. */
#define AFFNET_PROFILE_STAGES 8
int affnet_profile_enable(affnet_ctx* ctx, int on);
/* After the caller synchronised the stream(s): sums the elapsed milliseconds per stage over the
 * calls recorded since the last read, returns the number of calls in *n_calls and clears the ring. */
int affnet_profile_read(affnet_ctx* ctx, double sum_ms[AFFNET_PROFILE_STAGES], int32_t* n_calls);

/* The one optional read-back: copies counters to the host after synchronising `stream`:
 * out[0] = rows after detection, out[1] = rows after shape filter, out[2] = capacity-overflow
 * flag, out[3] = raw maxima found (sums / OR over the images of the batch).
 * Returns AFFNET_ERR_CAPACITY when a fixed-capacity device list overflowed (results would be truncated: treat as an
 * error), AFFNET_ERR_EMPTY when no image of the call produced a detection (out[] is still filled), else AFFNET_OK. */
int affnet_read_counts(affnet_ctx* ctx, int32_t out[4], void* stream);

/* The same counters without a host synchronisation: int32 offset (from the workspace base) of a per-image device counter of image 0
 * and the int32 stride between images.  which: 0 = capacity-overflow flag, 1 = rows after detection, 2 = rows after the shape filter,
 * 3 = candidates the shape CNN was actually evaluated on (cfg->lazy_shape_rows).
 * Valid once the enqueued work has completed on the stream; -1 for a context without a workspace layout. */
int64_t affnet_counter_offset(const affnet_ctx* ctx, int which);
int64_t affnet_counter_stride(const affnet_ctx* ctx);

/* Host helper (no GPU): out[ps] = affine_grid base coordinates (linspace(-1,1,ps)*(ps-1))/ps with
 * torch's CPU rounding; exported so the CPU test-suite can pin it against torch.linspace. */
int affnet_host_base_grid(int ps, float* out);

#ifdef __cplusplus
}
#endif
#endif /* AFFNET_HIP_H */
