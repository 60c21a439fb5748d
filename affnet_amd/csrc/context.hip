// Context, workspace layout and small host helpers of libaffnet_hip.so.
#include <stdarg.h>

#include "common.h"

int aff_fail(affnet_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

// ---- fills / copies ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void aff_zero_kernel(unsigned char* __restrict__ p, size_t bytes) {
    // 16-byte stores over the aligned body, byte stores for the (rare) unaligned head / tail
    const size_t head = ((16 - ((size_t)p & 15)) & 15) < bytes ? ((16 - ((size_t)p & 15)) & 15) : bytes;
    const size_t n16 = (bytes - head) / 16, tail0 = head + n16 * 16;
    uint4* q = reinterpret_cast<uint4*>(p + head);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) q[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0) {
        for (size_t i = threadIdx.x; i < head; i += 256) p[i] = 0;
        for (size_t i = tail0 + threadIdx.x; i < bytes; i += 256) p[i] = 0;
    }
}

// Several fills in ONE launch (blockIdx.y = segment): at one image per call every launch costs ~5 us whatever it does, and the
// detector alone cleared six areas.
__global__ __launch_bounds__(256) void aff_zero_multi_kernel(AffZeroSegs z) {
    unsigned char* p = z.p[blockIdx.y];
    const size_t bytes = z.bytes[blockIdx.y];
    const size_t head = ((16 - ((size_t)p & 15)) & 15) < bytes ? ((16 - ((size_t)p & 15)) & 15) : bytes;
    const size_t n16 = (bytes - head) / 16, tail0 = head + n16 * 16;
    uint4* q = reinterpret_cast<uint4*>(p + head);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) q[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0) {
        for (size_t i = threadIdx.x; i < head; i += 256) p[i] = 0;
        for (size_t i = tail0 + threadIdx.x; i < bytes; i += 256) p[i] = 0;
    }
}

int aff_zero_multi_async(affnet_ctx* ctx, const AffZeroSegs& z, hipStream_t st) {
    if (z.overflow) return aff_fail(ctx, AFFNET_ERR_INVALID, "internal: more than 8 areas in one fused clear (AffZeroSegs)");
    if (z.n <= 0) return AFFNET_OK;
    size_t blocks = 1;
    for (int i = 0; i < z.n; ++i) {
        const size_t b = (z.bytes[i] / 16 + 255) / 256;
        blocks = b > blocks ? b : blocks;
    }
    blocks = blocks > 2048 ? 2048 : blocks;
    hipLaunchKernelGGL(aff_zero_multi_kernel, dim3((unsigned)blocks, (unsigned)z.n), dim3(256), 0, st, z);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

__global__ __launch_bounds__(256) void aff_copy2d_kernel(uint32_t* __restrict__ dst, size_t dpitch_w, const uint32_t* __restrict__ src, size_t spitch_w,
                                                         size_t width_w) {
    const uint32_t* s = src + (size_t)blockIdx.y * spitch_w;
    uint32_t* d = dst + (size_t)blockIdx.y * dpitch_w;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < width_w; i += (size_t)gridDim.x * 256) d[i] = s[i];
}

int aff_zero_async(affnet_ctx* ctx, void* dst, size_t bytes, hipStream_t st) {
    if (!dst || bytes == 0) return AFFNET_OK;
    size_t blocks = (bytes / 16 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    hipLaunchKernelGGL(aff_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned char*)dst, bytes);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

int aff_copy2d_async(affnet_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, size_t rows, hipStream_t st) {
    if (rows == 0 || width_bytes == 0) return AFFNET_OK;
    if (!dst || !src || ((size_t)dst & 3) || ((size_t)src & 3) || (dpitch & 3) || (spitch & 3) || (width_bytes & 3) || rows > 65535)
        return aff_fail(ctx, AFFNET_ERR_INVALID, "copy2d: pointers, pitches and width must be multiples of 4 bytes, rows <= 65535");
    size_t bx = (width_bytes / 4 + 255) / 256;
    bx = bx < 1 ? 1 : (bx > 2048 ? 2048 : bx);
    hipLaunchKernelGGL(aff_copy2d_kernel, dim3((unsigned)bx, (unsigned)rows), dim3(256), 0, st, (uint32_t*)dst, dpitch / 4, (const uint32_t*)src, spitch / 4,
                       width_bytes / 4);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

int aff_copy_async(affnet_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t st) {
    return aff_copy2d_async(ctx, dst, bytes, src, bytes, bytes, 1, st);
}

extern "C" const char* affnet_version(void) { return "affnet_hip 0.2 (gfx950, hipcc, fp32 MFMA 16x16x4)"; }

extern "C" const char* affnet_last_error(const affnet_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

void aff_base_grid(int ps, float* base) {
    // torch.linspace(-1, 1, ps) on CPU: step = (end-start)/(steps-1) in fp32; element i is
    // fma(step, i, start) for i < steps/2 and fma(-step, steps-1-i, end) otherwise; then
    // affine_grid (align_corners=False) multiplies by (ps-1) and divides by ps.
    if (ps == 1) { base[0] = 0.0f; return; }  // linspace(-1,1,1) = [-1]; * 0 / 1 = -0 -> 0 contribution either way
    const float step = 2.0f / (float)(ps - 1);
    const int half = ps / 2;
    for (int i = 0; i < ps; ++i) {
        float v = (i < half) ? fmaf(step, (float)i, -1.0f) : fmaf(-step, (float)(ps - 1 - i), 1.0f);
        base[i] = (v * (float)(ps - 1)) / (float)ps;
    }
}

extern "C" int affnet_host_base_grid(int ps, float* out) {  // exported for the CPU-side mirror test
    if (ps < 1 || !out) return AFFNET_ERR_INVALID;
    aff_base_grid(ps, out);
    return AFFNET_OK;
}

static int validate(affnet_ctx* ctx, const affnet_config* c) {
    if (c->height < 8 || c->width < 8) return aff_fail(ctx, AFFNET_ERR_INVALID, "image %dx%d too small", c->height, c->width);
    if (c->batch > 4096) return aff_fail(ctx, AFFNET_ERR_INVALID, "batch=%d (max 4096)", c->batch);
    if (c->arith < AFFNET_ARITH_FP32_MFMA || c->arith > AFFNET_ARITH_FP32_SPLIT2H) return aff_fail(ctx, AFFNET_ERR_INVALID, "arith=%d (AFFNET_ARITH_*)", c->arith);
    if (c->n_octaves < 1 || c->n_octaves > AFFNET_MAX_OCTAVES) return aff_fail(ctx, AFFNET_ERR_INVALID, "n_octaves=%d", c->n_octaves);
    if (c->levels_per_octave < 3 || c->levels_per_octave > AFFNET_MAX_LEVELS)
        return aff_fail(ctx, AFFNET_ERR_INVALID, "levels_per_octave=%d", c->levels_per_octave);
    if (c->oct_h[0] != c->height || c->oct_w[0] != c->width) return aff_fail(ctx, AFFNET_ERR_INVALID, "octave 0 size mismatch");
    for (int o = 1; o < c->n_octaves; ++o)
        if (c->oct_h[o] != (c->oct_h[o - 1] - 1) / 2 + 1 || c->oct_w[o] != (c->oct_w[o - 1] - 1) / 2 + 1)
            return aff_fail(ctx, AFFNET_ERR_INVALID, "octave %d size is not ceil(prev/2)", o);
    if (c->first_blur_taps != 0 && (c->first_blur_taps % 2 == 0 || c->first_blur_taps > AFFNET_MAX_TAPS))
        return aff_fail(ctx, AFFNET_ERR_INVALID, "first_blur_taps=%d", c->first_blur_taps);
    for (int l = 1; l < c->levels_per_octave; ++l)
        if (c->level_blur_taps[l] % 2 == 0 || c->level_blur_taps[l] < 1 || c->level_blur_taps[l] > AFFNET_MAX_TAPS)
            return aff_fail(ctx, AFFNET_ERR_INVALID, "level_blur_taps[%d]=%d", l, c->level_blur_taps[l]);
    if (c->level_blur0_taps[1] != 0)
        for (int l = 1; l < c->levels_per_octave; ++l)
            if (c->level_blur0_taps[l] % 2 == 0 || c->level_blur0_taps[l] < 1 || c->level_blur0_taps[l] > AFFNET_MAX_TAPS)
                return aff_fail(ctx, AFFNET_ERR_INVALID, "level_blur0_taps[%d]=%d", l, c->level_blur0_taps[l]);
    return AFFNET_OK;
}

extern "C" int affnet_ctx_create(affnet_ctx** out, int device, const affnet_config* cfg) {
    if (!out) return AFFNET_ERR_INVALID;
    affnet_ctx* ctx = new affnet_ctx();
    ctx->device = device;
    if (!cfg) {  // utility context: stand-alone stage calls (blur, sampler, CNNs on patch tensors), no pyramid
        memset(&ctx->cfg, 0, sizeof(ctx->cfg));
        *out = ctx;
        return AFFNET_OK;
    }
    ctx->cfg = *cfg;
    ctx->arith = cfg->arith;
    int rc = validate(ctx, cfg);
    if (rc != AFFNET_OK) { *out = ctx; return rc; }
    const affnet_config& c = ctx->cfg;
    const int div = c.max_raw_per_octave_div > 0 ? c.max_raw_per_octave_div : 4;
    size_t pyr = 0, mapb = 0, raw = 0;
    for (int o = 0; o < c.n_octaves; ++o) {
        OctaveGeom& g = ctx->oct[o];
        g.h = c.oct_h[o]; g.w = c.oct_w[o];
        g.pyr_off = (int64_t)pyr; pyr += (size_t)c.levels_per_octave * g.h * g.w;
        g.map_off = (int64_t)mapb; mapb += aff_align((size_t)g.h * g.w, 16);
        int cap = (int)((size_t)g.h * g.w / div); if (cap < 256) cap = 256;
        g.raw_off = (int64_t)raw; g.raw_cap = cap; raw += cap;
    }
    ctx->B = c.batch > 0 ? c.batch : 1;
    const size_t B = (size_t)ctx->B;
    ctx->pyr_floats = pyr; ctx->map_bytes = mapb; ctx->raw_total = raw; ctx->cand_cap = raw;
    ctx->pyr_stride = aff_align(pyr * sizeof(float)) / sizeof(float);
    ctx->map_stride = aff_align(mapb);
    ctx->raw_stride = raw;
    const int keep = c.max_keep > 0 ? c.max_keep : 65536;
    ctx->cap_pre = c.num_prefilter > 0 ? c.num_prefilter : keep;
    ctx->cap_final = c.num_features > 0 ? c.num_features : ctx->cap_pre;
    if (ctx->cap_final > ctx->cap_pre) ctx->cap_final = ctx->cap_pre;
    size_t off = 0;
    ctx->off_pyr = off; off += B * ctx->pyr_stride * sizeof(float);
    ctx->off_map = off; off += B * ctx->map_stride;
    ctx->off_raw = off; off += aff_align(B * raw * sizeof(RawMax));
    ctx->off_cnt = off; off += aff_align(B * CNT_TOTAL * sizeof(int32_t));
    ctx->off_hist = off; off += aff_align(B * SEL_HIST_BINS * sizeof(uint32_t));   // top-digit histogram of the global top-k (detect.hip)
    ctx->off_cand = off; off += aff_align(B * ctx->cand_cap * 7 * sizeof(float));   // resp + syx[3] + ids[3]
    ctx->off_sel = off; off += aff_align(B * (size_t)ctx->cap_pre * 7 * sizeof(float));
    ctx->off_stage = off;
    const size_t P = B * (size_t)ctx->cap_pre, F = B * (size_t)ctx->cap_final;
    off += aff_align(P * 10 * sizeof(float));          // det resp(1) + lafs(6) + ids(3)
    off += aff_align(P * 4 * sizeof(float));           // A
    off += aff_align(P * 10 * sizeof(float));          // A of the current iteration (4) + re-extraction LAFs (6)
    off += aff_align(P * 2 * sizeof(float));           // key, good
    off += aff_align(P * sizeof(int32_t));             // rank / pos
    off += aff_align(B * sizeof(int32_t));             // detector row counts
    off += aff_align(F * 6 * sizeof(float));           // shaped lafs (normalised)
    off += aff_align(F * 4 * sizeof(float));           // R
    off += aff_align(F * 9 * sizeof(float));           // lafs_norm(6) + lvl ids(3)
    {   // HardNet trunk output + split-K head partials; also holds the AffNet (P rows) / OriNet head partials (144 floats each)
        const size_t hard = F * (8192 + 4 * 128), aff = P * 144;
        off += aff_align((hard > aff ? hard : aff) * sizeof(float));
    }
    if (c.onepass) {
        // OnePassSIR: dense affine-shape maps of every octave, the dense net's scratch (octave 0 is the largest), a second
        // candidate list (7 floats per row like the first) and the per-(octave, level) top-k table
        if (c.num_prefilter != c.num_features) { *out = ctx; return aff_fail(ctx, AFFNET_ERR_INVALID, "onepass: num_prefilter must equal num_features"); }
        size_t aff = 0;
        for (int o = 0; o < c.n_octaves; ++o) {
            if (c.oct_h[o] < 34 || c.oct_w[o] < 34) {
                *out = ctx;
                return aff_fail(ctx, AFFNET_ERR_INVALID, "onepass: octave %d is %dx%d, LocalNorm2d(33) needs >= 34 px (the reference raises too: use "
                                "border >= 15 like its scripts, so that the pyramid stops earlier)", o, c.oct_w[o], c.oct_h[o]);
            }
            ctx->aff_off[o] = aff; aff += aff_align((size_t)4 * c.oct_h[o] * c.oct_w[o] * sizeof(float)) / sizeof(float);
        }
        ctx->aff_stride = aff;
        ctx->dense_stride = affnet_fullconv_scratch_bytes(c.oct_h[0], c.oct_w[0]) / sizeof(float);
        ctx->off_affmap = off; off += aff_align(B * aff * sizeof(float));
        ctx->off_dense = off; off += aff_align(B * ctx->dense_stride * sizeof(float));
        ctx->off_cand2 = off; off += aff_align(B * ctx->cand_cap * 7 * sizeof(float));
        ctx->off_lvltab = off; off += aff_align(B * (size_t)AFFNET_MAX_OCTAVES * AFFNET_MAX_LEVELS * 4 * sizeof(int32_t));
    }
    ctx->ws_bytes = off;
    *out = ctx;
    return AFFNET_OK;
}

extern "C" void affnet_ctx_destroy(affnet_ctx* ctx) { delete ctx; }

extern "C" int affnet_set_arith(affnet_ctx* ctx, int arith) {
    if (!ctx) return AFFNET_ERR_INVALID;
    if (arith < AFFNET_ARITH_FP32_MFMA || arith > AFFNET_ARITH_FP32_SPLIT2H) return aff_fail(ctx, AFFNET_ERR_INVALID, "arith=%d (AFFNET_ARITH_*)", arith);
    ctx->arith = arith;
    ctx->cfg.arith = arith;
    return AFFNET_OK;
}
extern "C" int affnet_get_arith(const affnet_ctx* ctx) { return ctx ? ctx->arith : AFFNET_ERR_INVALID; }

extern "C" size_t affnet_workspace_bytes(const affnet_ctx* ctx) { return ctx ? ctx->ws_bytes : 0; }

extern "C" int affnet_capacity_prefilter(const affnet_ctx* ctx) { return ctx ? ctx->cap_pre : 0; }
extern "C" int affnet_capacity_final(const affnet_ctx* ctx) { return ctx ? ctx->cap_final : 0; }

extern "C" int64_t affnet_pyramid_level_offset(const affnet_ctx* ctx, int octave, int level) {
    if (!ctx || octave < 0 || octave >= ctx->cfg.n_octaves || level < 0 || level >= ctx->cfg.levels_per_octave) return -1;
    const OctaveGeom& g = ctx->oct[octave];
    return (int64_t)(ctx->off_pyr / sizeof(float)) + g.pyr_off + (int64_t)level * g.h * g.w;
}

extern "C" int64_t affnet_pyramid_image_stride(const affnet_ctx* ctx) { return ctx ? (int64_t)ctx->pyr_stride : 0; }

// int32 offset (from the workspace base) of a per-image device counter of image 0; consecutive images are affnet_counter_stride()
// int32 apart.  which: 0 = capacity-overflow flag (non-zero: a fixed-capacity list overflowed, results are truncated),
// 1 = rows after detection, 2 = rows after the shape filter, 3 = candidates the shape CNN was evaluated on (lazy evaluation).  Lets a caller test the flags on the device / read them with its own
// asynchronous copy instead of the synchronising affnet_read_counts.
extern "C" int64_t affnet_counter_offset(const affnet_ctx* ctx, int which) {
    if (!ctx || ctx->ws_bytes == 0 || which < 0 || which > 3) return -1;
    const int idx = which == 0 ? CNT_OVERFLOW : (which == 1 ? CNT_DET : (which == 2 ? CNT_SHAPED : CNT_AFF_EVAL));
    return (int64_t)(ctx->off_cnt / sizeof(int32_t)) + idx;
}
extern "C" int64_t affnet_counter_stride(const affnet_ctx* ctx) { return ctx ? (int64_t)CNT_TOTAL : 0; }

extern "C" int64_t affnet_affmap_offset(const affnet_ctx* ctx, int octave) {
    if (!ctx || !ctx->cfg.onepass || octave < 0 || octave >= ctx->cfg.n_octaves) return -1;
    return (int64_t)(ctx->off_affmap / sizeof(float)) + (int64_t)ctx->aff_off[octave];
}
extern "C" int64_t affnet_affmap_image_stride(const affnet_ctx* ctx) { return (ctx && ctx->cfg.onepass) ? (int64_t)ctx->aff_stride : 0; }
extern "C" int affnet_batch(const affnet_ctx* ctx) { return ctx ? ctx->B : 0; }

extern "C" int affnet_bind_workspace(affnet_ctx* ctx, void* d_workspace, size_t bytes) {
    if (!ctx) return AFFNET_ERR_INVALID;
    if (!d_workspace || bytes < ctx->ws_bytes)
        return aff_fail(ctx, AFFNET_ERR_INVALID, "workspace too small: %zu < %zu", bytes, ctx->ws_bytes);
    if (((uintptr_t)d_workspace & 255) != 0) return aff_fail(ctx, AFFNET_ERR_INVALID, "workspace must be 256-byte aligned");
    {   // the workspace must live on the context's device: a pointer of another GPU would fault (or silently run there)
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, d_workspace) == hipSuccess) {
            if (at.type == hipMemoryTypeDevice && at.device != ctx->device)
                return aff_fail(ctx, AFFNET_ERR_INVALID, "workspace belongs to device %d, the context to device %d", at.device, ctx->device);
        } else {
            (void)hipGetLastError();   // not a HIP allocation / no GPU present (CPU-side layout tests): nothing to check
        }
    }
    char* b = (char*)d_workspace;
    ctx->ws = b;
    ctx->pyr = (float*)(b + ctx->off_pyr);
    ctx->omap = (uint8_t*)(b + ctx->off_map);
    ctx->raw = (RawMax*)(b + ctx->off_raw);
    ctx->cnt = (int32_t*)(b + ctx->off_cnt);
    ctx->sel_hist = (uint32_t*)(b + ctx->off_hist);
    const size_t B = (size_t)ctx->B;
    float* cand = (float*)(b + ctx->off_cand);
    const size_t CC = B * ctx->cand_cap;
    ctx->cand_resp = cand; ctx->cand_syx = cand + CC; ctx->cand_ids = (int32_t*)(cand + 4 * CC);
    float* sel = (float*)(b + ctx->off_sel);
    const size_t P = B * (size_t)ctx->cap_pre, F = B * (size_t)ctx->cap_final;
    ctx->sel_resp = sel; ctx->sel_syx = sel + P; ctx->sel_ids = (int32_t*)(sel + 4 * P);
    char* s = b + ctx->off_stage;
    ctx->st_det_resp = (float*)s; ctx->st_det_lafs = (float*)s + P; ctx->st_det_ids = (int32_t*)((float*)s + 7 * P);
    s += aff_align(P * 10 * sizeof(float));
    ctx->st_A = (float*)s; s += aff_align(P * 4 * sizeof(float));
    ctx->st_A2 = (float*)s; ctx->st_lafs_iter = (float*)s + 4 * P; s += aff_align(P * 10 * sizeof(float));
    ctx->st_key = (float*)s; ctx->st_good = (int32_t*)((float*)s + P); s += aff_align(P * 2 * sizeof(float));
    ctx->st_rank = (int32_t*)s; s += aff_align(P * sizeof(int32_t));
    ctx->st_det_count = (int32_t*)s; s += aff_align(B * sizeof(int32_t));
    ctx->st_lafs_shaped = (float*)s; s += aff_align(F * 6 * sizeof(float));
    ctx->st_R = (float*)s; s += aff_align(F * 4 * sizeof(float));
    ctx->st_lafs_norm = (float*)s; ctx->st_lvl_ids = (int32_t*)((float*)s + 6 * F); s += aff_align(F * 9 * sizeof(float));
    ctx->st_hard_scratch = (float*)s;
    if (ctx->cfg.onepass) {
        ctx->affmap = (float*)(b + ctx->off_affmap);
        ctx->dense = (float*)(b + ctx->off_dense);
        float* c2 = (float*)(b + ctx->off_cand2);
        ctx->cand2_resp = c2; ctx->cand2_syx = c2 + CC; ctx->cand2_ids = (int32_t*)(c2 + 4 * CC);
        ctx->lvltab = (int32_t*)(b + ctx->off_lvltab);
    }
    return AFFNET_OK;
}

extern "C" int affnet_read_counts(affnet_ctx* ctx, int32_t out[4], void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !ctx->ws) return AFFNET_ERR_INVALID;
    std::vector<int32_t> all((size_t)ctx->B * CNT_TOTAL);
    AFF_HIP(ctx, hipMemcpyAsync(all.data(), ctx->cnt, all.size() * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    AFF_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    int32_t host[CNT_TOTAL];
    memset(host, 0, sizeof(host));
    int raw = 0;
    for (int b = 0; b < ctx->B; ++b) {
        const int32_t* h = all.data() + (size_t)b * CNT_TOTAL;
        host[CNT_DET] += h[CNT_DET]; host[CNT_SHAPED] += h[CNT_SHAPED]; host[CNT_OVERFLOW] |= h[CNT_OVERFLOW];
        for (int o = 0; o < ctx->cfg.n_octaves; ++o) raw += h[CNT_RAW0 + o];
    }
    out[0] = host[CNT_DET]; out[1] = host[CNT_SHAPED]; out[2] = host[CNT_OVERFLOW];
    out[3] = raw;
    if (host[CNT_OVERFLOW]) return aff_fail(ctx, AFFNET_ERR_CAPACITY, "a fixed-capacity detector list overflowed (flag %d)", host[CNT_OVERFLOW]);
    if (host[CNT_DET] == 0)
        return aff_fail(ctx, AFFNET_ERR_EMPTY, "no keypoints detected in %d image(s) (the reference raises in torch.cat, SparseImgRepresenter.py:100)", ctx->B);
    return AFFNET_OK;
}
