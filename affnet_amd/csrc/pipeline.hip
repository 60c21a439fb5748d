// Fused whole-path entry point: image resident -> (LAFs, responses, descriptors) resident with zero
// host synchronisation.  Replaces ScaleSpaceAffinePatchExtractor.forward
// (SparseImgRepresenter.py:189-209), extract_patches_from_pyr (:181-188) and the HardNet call of
// get_geometry_and_descriptors (train_OriNet_test_on_graffity.py:293-298).
//
// Every stage reads its row count from device memory (counter block in the workspace) and is
// launched with the capacity-sized grid, so data-dependent sizes never travel to the host; the
// reference synchronises 18x per image on .item() (HandCraftedModules.py:252), once more on
// num_survived (SparseImgRepresenter.py:151) and round-trips LAF scales through scipy on the
// host (LAF.py:466).  The reference's discarded extra patch extraction (:178-179) is dropped.
#include "common.h"
#include "shape_filter.h"

void aff_prof_mark(affnet_ctx* ctx, int idx, hipStream_t st) {
    if (!ctx->prof_on || ctx->prof_calls >= PROF_RING) return;
    (void)hipEventRecord(ctx->prof_ev[(size_t)ctx->prof_calls * PROF_EVENTS + idx], st);
}

extern "C" int affnet_profile_enable(affnet_ctx* ctx, int on) {
    if (!ctx) return AFFNET_ERR_INVALID;
    if (on && ctx->prof_ev.empty()) {
        ctx->prof_ev.resize((size_t)PROF_RING * PROF_EVENTS);
        for (auto& e : ctx->prof_ev) AFF_HIP(ctx, hipEventCreate(&e));
    }
    ctx->prof_on = on != 0;
    ctx->prof_calls = 0;
    return AFFNET_OK;
}

extern "C" int affnet_profile_read(affnet_ctx* ctx, double sum_ms[AFFNET_PROFILE_STAGES], int32_t* n_calls) {
    AFF_DEVICE(ctx);
    if (!ctx || !sum_ms || !n_calls) return AFFNET_ERR_INVALID;
    for (int s = 0; s < AFFNET_PROFILE_STAGES; ++s) sum_ms[s] = 0.0;
    *n_calls = ctx->prof_calls;
    for (int c = 0; c < ctx->prof_calls; ++c)
        for (int s = 0; s < AFFNET_PROFILE_STAGES; ++s) {
            float ms = 0.f;
            const size_t b = (size_t)c * PROF_EVENTS;
            // the detector may run on another stream than the CNN stages: its end has its own event (9)
            AFF_HIP(ctx, hipEventElapsedTime(&ms, ctx->prof_ev[b + s], ctx->prof_ev[b + (s == 1 ? 9 : s + 1)]));
            sum_ms[s] += ms;
        }
    ctx->prof_calls = 0;
    return AFFNET_OK;
}

// AffNet iterations > 1 (laf_ops.hip)
int aff_shape_iterate(affnet_ctx* ctx, const float* A, float* base, const float* lafs, const int32_t* count, int mode, float* lafs_out,
                      hipStream_t st);

// hand-crafted slot fillers (handcrafted.hip)
int aff_handcrafted_launch(affnet_ctx* ctx, int kind, const float* patches, const float* lafs, const int32_t* ids, const int32_t* count,
                           int n_max, const float* h_weights, float* out, float* out_angle, hipStream_t st);

// HardNet with a stage mark between trunk and head (cnn32.hip)
int aff_hardnet_forward_pyr_marked(affnet_ctx* ctx, const float* packed, const float* lafs, const int32_t* ids, const int32_t* count,
                                   int n_max, float* out, float* scratch, hipStream_t st);

// shape stage in steps + row-windowed CNN launches (laf_ops.hip / cnn32.hip): lazy evaluation of the shape CNN
int aff_shape_filter_begin(affnet_ctx* ctx, hipStream_t st);
int aff_shape_filter_rows(affnet_ctx* ctx, const float* resp, const float* lafs, const float* A, const int32_t* count, int row_begin, int row_end,
                          bool lazy, hipStream_t st, bool freeze = false);
int aff_shape_select(affnet_ctx* ctx, const float* d_resp_in, const float* d_lafs_in, const int32_t* d_ids_in, const float* d_A,
                     const int32_t* d_count_in, float* d_resp_out, float* d_lafs_out, int32_t* d_ids_out, int32_t* d_count_out, hipStream_t st);
int aff_cnn_forward_pyr_rows(affnet_ctx* ctx, int kind, const float* packed, const float* lafs, const int32_t* ids, const int32_t* count, int n_max,
                             float* out, float* scratch, int row_begin, int row_count, const int32_t* skip_cnt, int skip_n, hipStream_t st);

// fused launches of the one-image-per-call latency path (cnn32.hip / laf_ops.hip)
int aff_affnet_filter_rows(affnet_ctx* ctx, const float* packed, const float* resp, const float* lafs, const int32_t* ids, const int32_t* count,
                           float* out, float* scratch, int row_begin, int row_count, bool lazy, int shape_op, hipStream_t st);
int aff_orinet_rotate(affnet_ctx* ctx, const float* packed, float* lafs, const int32_t* ids, const int32_t* count, int n_max, float* out, float* scratch,
                      hipStream_t st, const DenormSel* denorm);
void aff_denorm_sel_fill(affnet_ctx* ctx, int ps, float* d_lafs_px, int32_t* d_ids, float* d_lafs_norm, DenormSel* ds);
int aff_denorm_level_select(affnet_ctx* ctx, const float* d_lafs_norm_in, float* d_lafs_px, const int32_t* d_count, int n_max, int ps, int32_t* d_ids,
                            float* d_lafs_norm, hipStream_t st);

int aff_detect_impl(affnet_ctx* ctx, const float* d_responses, float* d_resp, float* d_lafs, int32_t* d_ids, int32_t* d_count, void* stream);

// Detector half for a custom RespNet slot: the caller has built the pyramid (affnet_pyramid_build), evaluated its RespNet on
// every level and hands over the response pyramid; candidates go to the context's internal list for
// affnet_describe_detected.
extern "C" int affnet_detect_image_responses(affnet_ctx* ctx, const float* d_responses, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !ctx->ws || !d_responses) return aff_fail(ctx, AFFNET_ERR_INVALID, "detect_image_responses: context not bound or null responses");
    hipStream_t st = (hipStream_t)stream;
    aff_prof_mark(ctx, 0, st);
    aff_prof_mark(ctx, 1, st);
    int rc = aff_detect_impl(ctx, d_responses, ctx->st_det_resp, ctx->st_det_lafs, ctx->st_det_ids, ctx->st_det_count, stream);
    if (rc) return rc;
    aff_prof_mark(ctx, 9, st);
    return AFFNET_OK;
}

extern "C" int affnet_detect_image(affnet_ctx* ctx, const float* d_img, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !ctx->ws || !d_img) return aff_fail(ctx, AFFNET_ERR_INVALID, "detect_image: context not bound or null image");
    hipStream_t st = (hipStream_t)stream;
    aff_prof_mark(ctx, 0, st);
    int rc = affnet_pyramid_build(ctx, d_img, stream);
    if (rc) return rc;
    aff_prof_mark(ctx, 1, st);
    rc = affnet_detect(ctx, ctx->st_det_resp, ctx->st_det_lafs, ctx->st_det_ids, ctx->st_det_count, stream);
    if (rc) return rc;
    aff_prof_mark(ctx, 9, st);
    return AFFNET_OK;
}

int aff_detect_onepass_impl(affnet_ctx* ctx, const float* d_packed_fullconv, const float* d_responses, hipStream_t st);

// OnePassSIR detector half (OnePassSIR.py:53-115,146): pyramid (when d_img != NULL; NULL = already built with
// affnet_pyramid_build), dense AffNetFastFullConv map per octave (when d_packed_fullconv != NULL; NULL = the caller wrote the maps
// of a foreign dense AffNet into the workspace at affnet_affmap_offset), Hessian / NMS / per-level top-k / boundary test / global
// top-k, LAFs = mrSize * s * A_map[pixel].  Candidates go to the internal list consumed by affnet_describe_detected, which is
// then called with nets->d_affnet == NULL (no per-patch shape stage): OriNet, denormalisation, level select, HardNet as usual.
extern "C" int affnet_detect_image_onepass(affnet_ctx* ctx, const float* d_packed_fullconv, const float* d_img, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !ctx->ws) return aff_fail(ctx, AFFNET_ERR_INVALID, "detect_image_onepass: context not bound");
    hipStream_t st = (hipStream_t)stream;
    aff_prof_mark(ctx, 0, st);
    if (d_img) {
        int rc = affnet_pyramid_build(ctx, d_img, stream);
        if (rc) return rc;
    }
    aff_prof_mark(ctx, 1, st);
    int rc = aff_detect_onepass_impl(ctx, d_packed_fullconv, nullptr, st);
    if (rc) return rc;
    aff_prof_mark(ctx, 9, st);
    return AFFNET_OK;
}

extern "C" int affnet_detect_image_onepass_responses(affnet_ctx* ctx, const float* d_packed_fullconv, const float* d_responses, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !ctx->ws || !d_responses) return aff_fail(ctx, AFFNET_ERR_INVALID, "detect_image_onepass_responses: context not bound or null responses");
    hipStream_t st = (hipStream_t)stream;
    aff_prof_mark(ctx, 0, st);
    aff_prof_mark(ctx, 1, st);
    int rc = aff_detect_onepass_impl(ctx, d_packed_fullconv, d_responses, st);
    if (rc) return rc;
    aff_prof_mark(ctx, 9, st);
    return AFFNET_OK;
}

extern "C" int affnet_detected_list(affnet_ctx* ctx, float* d_resp, float* d_lafs, int32_t* d_ids, int32_t* d_count, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !ctx->ws || !d_resp || !d_lafs || !d_ids || !d_count) return aff_fail(ctx, AFFNET_ERR_INVALID, "detected_list: context not bound or null output");
    hipStream_t st = (hipStream_t)stream;
    const size_t P = (size_t)ctx->B * ctx->cap_pre;
    int rc = aff_copy_async(ctx, d_resp, ctx->st_det_resp, P * sizeof(float), st);
    if (!rc) rc = aff_copy_async(ctx, d_lafs, ctx->st_det_lafs, P * 6 * sizeof(float), st);
    if (!rc) rc = aff_copy_async(ctx, d_ids, ctx->st_det_ids, P * 3 * sizeof(int32_t), st);
    if (!rc) rc = aff_copy_async(ctx, d_count, ctx->st_det_count, (size_t)ctx->B * sizeof(int32_t), st);
    return rc;
}

extern "C" int affnet_shape_iterate(affnet_ctx* ctx, const float* d_A, float* d_base, const float* d_lafs, const int32_t* d_count, int mode,
                                    float* d_lafs_out, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !ctx->ws || !d_base || !d_lafs || !d_count || !d_lafs_out || (mode != 0 && mode != 1) || (mode == 1 && !d_A))
        return aff_fail(ctx, AFFNET_ERR_INVALID, "shape_iterate: bad argument");
    return aff_shape_iterate(ctx, d_A, d_base, d_lafs, d_count, mode, d_lafs_out, (hipStream_t)stream);
}

extern "C" int affnet_describe_detected(affnet_ctx* ctx, const affnet_nets* nets, int do_ori, float* d_lafs_px, float* d_resp,
                                        int32_t* d_ids, float* d_desc, int32_t* d_count, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !ctx->ws || !nets || !d_lafs_px || !d_resp || !d_ids || !d_count)
        return aff_fail(ctx, AFFNET_ERR_INVALID, "describe_detected: context not bound or null argument");
    if (do_ori && !nets->d_orinet && !nets->h_orientation_window)
        return aff_fail(ctx, AFFNET_ERR_INVALID, "describe_detected: do_ori needs OriNet weights or the OrientationDetector window");
    if (d_desc && !nets->d_hardnet) return aff_fail(ctx, AFFNET_ERR_INVALID, "describe_detected: descriptors need HardNet weights");
    hipStream_t st = (hipStream_t)stream;
    const int P = ctx->cap_pre, F = ctx->cap_final;
    const size_t B = (size_t)ctx->B;
    int rc;
    int32_t* det_count = ctx->st_det_count;
    aff_prof_mark(ctx, 2, st);
    const bool baumberg = !nets->d_affnet && nets->h_baumberg_window && ctx->cfg.baum_iters > 0;
    // one shape pass: AffNetFast (32x32 patches, MFMA) or the hand-crafted AffineShapeEstimator (19x19 patches)
    auto shape_pass = [&](const float* lafs, float* A_out) -> int {
        if (baumberg)
            return aff_handcrafted_launch(ctx, AFFNET_HC_BAUMBERG, nullptr, lafs, ctx->st_det_ids, det_count, P, nets->h_baumberg_window, A_out,
                                          nullptr, st);
        return affnet_cnn32_forward_pyr(ctx, AFFNET_NET_AFFNET, nets->d_affnet, lafs, ctx->st_det_ids, det_count, P, A_out, ctx->st_hard_scratch,
                                        stream);   // head partials (P x 144 floats) share the HardNet scratch
    };
    // Lazy shape evaluation.  The reference runs AffNet on all C = 1.5 N candidates and keeps the N best survivors of the shape
    // filter (SparseImgRepresenter.py:113-162).  The candidates arrive sorted by response (top-k order), so the N best survivors are
    // simply the FIRST N survivors: AffNet runs on the first `lazy` candidates (1.2 N: the filter passes ~87 % on the benchmark
    // images, 2000 survivors need ~2330 candidates), and a second launch covers the rest only for images that do not have their
    // N survivors yet (or whose detections are not response-sorted: fewer than C candidates) - decided on the device, no host
    // synchronisation.  Output rows are identical to the all-at-once evaluation; ~20 % of the AffNet patches are never computed.
    const int N = ctx->cfg.num_features;
    const int lazy = (nets->d_affnet && ctx->cfg.baum_iters <= 1 && N > 0 && ctx->cfg.lazy_shape_rows != 0)
                         ? (ctx->cfg.lazy_shape_rows > 0 ? ctx->cfg.lazy_shape_rows : N + (N + 4) / 5) : 0;
    if (lazy > 0 && lazy < P) {
        // four launches: trunk + (finish + shape filter) per pass; the first trunk launch clears the survivor counters, the second
        // freezes the first pass's survivor count (were: trunk, finish, clear, begin, filter, freeze, trunk, finish, filter)
        rc = aff_affnet_filter_rows(ctx, nets->d_affnet, ctx->st_det_resp, ctx->st_det_lafs, ctx->st_det_ids, det_count, ctx->st_A, ctx->st_hard_scratch,
                                    0, lazy, false, 1, st);
        if (rc) return rc;
        rc = aff_affnet_filter_rows(ctx, nets->d_affnet, ctx->st_det_resp, ctx->st_det_lafs, ctx->st_det_ids, det_count, ctx->st_A, ctx->st_hard_scratch,
                                    lazy, P - lazy, true, 2, st);
        if (rc) return rc;
        aff_prof_mark(ctx, 3, st);
        rc = aff_shape_select(ctx, ctx->st_det_resp, ctx->st_det_lafs, ctx->st_det_ids, ctx->st_A, det_count, d_resp, ctx->st_lafs_shaped, d_ids,
                              d_count, st);
        if (rc) return rc;
    } else if (nets->d_affnet || baumberg) {
        rc = shape_pass(ctx->st_det_lafs, ctx->st_A);
        if (rc) return rc;
        // num_Baum_iters > 1 (SparseImgRepresenter.py:127-146): base_A = A_i * base_A, patches re-extracted from
        // [base_A * LAF | centre] on the same pyramid level.  st_A holds base_A (= A_0 after the first pass: bmm(A, I) = A).
        const int iters = ctx->cfg.baum_iters > 0 ? ctx->cfg.baum_iters : 1;
        for (int it = 1; it < iters; ++it) {
            rc = aff_shape_iterate(ctx, nullptr, ctx->st_A, ctx->st_det_lafs, det_count, 0, ctx->st_lafs_iter, st);
            if (rc) return rc;
            rc = shape_pass(ctx->st_lafs_iter, ctx->st_A2);
            if (rc) return rc;
            rc = aff_shape_iterate(ctx, ctx->st_A2, ctx->st_A, ctx->st_det_lafs, det_count, 1, ctx->st_lafs_iter, st);
            if (rc) return rc;
        }
        aff_prof_mark(ctx, 3, st);
        rc = affnet_shape_filter_select(ctx, ctx->st_det_resp, ctx->st_det_lafs, ctx->st_det_ids, ctx->st_A, det_count, d_resp,
                                        ctx->st_lafs_shaped, d_ids, d_count, stream);
        if (rc) return rc;
    } else {
        aff_prof_mark(ctx, 3, st);
        // num_Baum_iters == 0: detections pass through unchanged (C == N)
        if (F != P) return aff_fail(ctx, AFFNET_ERR_INVALID, "describe_detected: without AffNet num_prefilter must equal num_features");
        { int crc = aff_copy_async(ctx, d_resp, ctx->st_det_resp, B * F * sizeof(float), st); if (crc) return crc; }
        { int crc = aff_copy_async(ctx, ctx->st_lafs_shaped, ctx->st_det_lafs, B * F * 6 * sizeof(float), st); if (crc) return crc; }
        { int crc = aff_copy_async(ctx, d_ids, ctx->st_det_ids, B * F * 3 * sizeof(int32_t), st); if (crc) return crc; }
        { int crc = aff_copy_async(ctx, d_count, det_count, B * sizeof(int32_t), st); if (crc) return crc; }
        { int crc = aff_copy2d_async(ctx, ctx->cnt + CNT_SHAPED, CNT_TOTAL * sizeof(int32_t), det_count, sizeof(int32_t), sizeof(int32_t), B, st); if (crc) return crc; }
    }
    aff_prof_mark(ctx, 4, st);
    bool denorm_done = false;          // OriNet's finish kernel has denormalised the frames and chosen their levels (descriptor path)
    if (do_ori) {
        if (nets->d_orinet) {       // LAF <- LAF * R inside OriNet's finish kernel; with descriptors also denormalisation + level choice
            DenormSel ds;
            if (d_desc) aff_denorm_sel_fill(ctx, 32, d_lafs_px, ctx->st_lvl_ids, ctx->st_lafs_norm, &ds);
            rc = aff_orinet_rotate(ctx, nets->d_orinet, ctx->st_lafs_shaped, d_ids, d_count, F, ctx->st_R, ctx->st_hard_scratch, st, d_desc ? &ds : nullptr);
            if (rc) return rc;
            denorm_done = d_desc != nullptr;
        } else {
            rc = aff_handcrafted_launch(ctx, AFFNET_HC_ORIENTATION, nullptr, ctx->st_lafs_shaped, d_ids, d_count, F, nets->h_orientation_window,
                                        ctx->st_R, nullptr, st);
            if (rc) return rc;
            rc = affnet_apply_rotation(ctx, ctx->st_lafs_shaped, ctx->st_R, d_count, F, stream);
            if (rc) return rc;
        }
    }
    aff_prof_mark(ctx, 5, st);
    if (d_desc) {
        // denormalise + level choice in one launch; descriptor rows past the row count are cleared by hardnet_finish_kernel
        if (!denorm_done) {
            rc = aff_denorm_level_select(ctx, ctx->st_lafs_shaped, d_lafs_px, d_count, F, 32, ctx->st_lvl_ids, ctx->st_lafs_norm, st);
            if (rc) return rc;
        }
        aff_prof_mark(ctx, 6, st);
        rc = aff_hardnet_forward_pyr_marked(ctx, nets->d_hardnet, ctx->st_lafs_norm, ctx->st_lvl_ids, d_count, F, d_desc,
                                            ctx->st_hard_scratch, st);
        if (rc) return rc;
    } else {
        rc = affnet_scale_lafs(ctx, ctx->st_lafs_shaped, d_lafs_px, d_count, F, ctx->cfg.width, ctx->cfg.height, 0, stream);
        if (rc) return rc;
        aff_prof_mark(ctx, 6, st);
        aff_prof_mark(ctx, 7, st);
    }
    aff_prof_mark(ctx, 8, st);
    if (ctx->prof_on && ctx->prof_calls < PROF_RING) ++ctx->prof_calls;
    return AFFNET_OK;
}

extern "C" int affnet_extract_features(affnet_ctx* ctx, const affnet_nets* nets, const float* d_img, int do_ori, float* d_lafs_px,
                                       float* d_resp, int32_t* d_ids, float* d_desc, int32_t* d_count, void* stream) {
    AFF_DEVICE(ctx);
    int rc = affnet_detect_image(ctx, d_img, stream);
    if (rc) return rc;
    return affnet_describe_detected(ctx, nets, do_ori, d_lafs_px, d_resp, d_ids, d_desc, d_count, stream);
}


// ---- the whole path as ONE HIP graph ---------------------------------------------------------------------------------
// affnet_extract_features enqueues ~45 kernels / memsets without ever touching the host, so for a fixed set of buffers it can be
// stream-captured once and replayed with a single hipGraphLaunch: what a latency-bound caller (one image at a time, BASELINE
// configs[1]) pays per call drops from ~45 launches to one.  Throughput callers (32 images per call) are GPU bound and gain nothing.
extern "C" int affnet_graph_capture_extract(affnet_ctx* ctx, const affnet_nets* nets, const float* d_img, int do_ori, float* d_lafs_px, float* d_resp,
                                            int32_t* d_ids, float* d_desc, int32_t* d_count, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !ctx->ws) return aff_fail(ctx, AFFNET_ERR_INVALID, "graph_capture: context not bound");
    hipStream_t st = (hipStream_t)stream;
    if (!st) return aff_fail(ctx, AFFNET_ERR_INVALID, "graph_capture: needs an explicit (non-null) stream - the legacy null stream cannot be captured");
    if (ctx->prof_on) return aff_fail(ctx, AFFNET_ERR_INVALID, "graph_capture: switch stage profiling off first (its events are per call)");
    if (ctx->graph_exec) { (void)hipGraphExecDestroy(ctx->graph_exec); ctx->graph_exec = nullptr; }
    if (ctx->graph) { (void)hipGraphDestroy(ctx->graph); ctx->graph = nullptr; }
    AFF_HIP(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = affnet_extract_features(ctx, nets, d_img, do_ori, d_lafs_px, d_resp, d_ids, d_desc, d_count, stream);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &g);       // always end the capture, also after a failed enqueue
    if (rc != AFFNET_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess || !g) return aff_fail(ctx, AFFNET_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
    ctx->graph = g;
    AFF_HIP(ctx, hipGraphInstantiate(&ctx->graph_exec, ctx->graph, nullptr, nullptr, 0));
    return AFFNET_OK;
}

extern "C" int affnet_graph_launch(affnet_ctx* ctx, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !ctx->graph_exec) return aff_fail(ctx, AFFNET_ERR_INVALID, "graph_launch: nothing captured (affnet_graph_capture_extract first)");
    AFF_HIP(ctx, hipGraphLaunch(ctx->graph_exec, (hipStream_t)stream));
    return AFFNET_OK;
}
