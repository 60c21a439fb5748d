// LAF-driven patch sampler and LAF algebra kernels for gfx950.
//
// Replaces LAF.py:313-324,326-372 (affine grid + grid_sample), :376-404 (pyramid gather),
// :407-429 (de/normalise), :450-472 (level choice via host scipy cdist),
// SparseImgRepresenter.py:121-162 (shape compose + filter + top-N), :173-177 (apply rotation),
// Utils.py:168-175 (batch_eig2x2), LAF.py:91-104 (checkTouchBoundary).
#include <math.h>

#include "common.h"
#include "shape_filter.h"

#define MAX_PS 64

struct BaseGrid { float v[MAX_PS]; };

void aff_fill_pyr_table(const affnet_ctx* ctx, PyrTable* t) {
    memset(t, 0, sizeof(*t));
    t->n_octaves = ctx->cfg.n_octaves; t->n_levels = ctx->cfg.levels_per_octave;
    t->img_stride = ctx->pyr_stride;
    for (int o = 0; o < t->n_octaves; ++o) {
        const OctaveGeom& g = ctx->oct[o];
        t->h[o] = g.h; t->w[o] = g.w;
        for (int l = 0; l < t->n_levels; ++l) t->lvl[o][l] = ctx->pyr + g.pyr_off + (size_t)l * g.h * g.w;
    }
}

// One workgroup per patch.  The affine footprint of a patch is an axis-aligned box of the level image: when it is small (most
// detector-level patches: <= 2 x ps x ps pixels) its rows are staged in LDS with coalesced row-segment loads and the four taps of
// every sample come from there (common.h: aff_tile_*); large footprints (a 32 x 32 patch drawn from a 160 x 160 px region of octave 0
// touches a sixth of it) keep the four predicated gathers per sample.  Both paths are bit-identical.
#define GS_TILE_CAP 4096
__global__ __launch_bounds__(256) void grid_sample_kernel(const float* __restrict__ img, int h, int w, PyrTable pt, int use_pyr,
                                                          const float* __restrict__ lafs, const int32_t* __restrict__ ids,
                                                          const int32_t* __restrict__ d_count, int n_max, int ps, BaseGrid bg,
                                                          float* __restrict__ out) {
    const int p = blockIdx.x;
    const size_t bi = blockIdx.y;                     // image of the batch; rows of image bi start at bi * n_max
    const int n = d_count ? min(d_count[bi], n_max) : n_max;
    if (p >= n) return;
    lafs += bi * n_max * 6; out += bi * n_max * (size_t)(ps * ps);
    if (use_pyr) {
        ids += bi * n_max * 3;
        int o = ids[3 * p], l = ids[3 * p + 1];
        o = o < 0 ? 0 : (o >= pt.n_octaves ? pt.n_octaves - 1 : o);
        l = l < 0 ? 0 : (l >= pt.n_levels ? pt.n_levels - 1 : l);
        img = pt.lvl[o][l] + bi * pt.img_stride; h = pt.h[o]; w = pt.w[o];
    }
    const float* L = lafs + 6 * (size_t)p;
    const float m = (float)(h < w ? h : w);
    const float t00 = L[0] * m, t01 = L[1] * m, t02 = L[2] * (float)w;
    const float t10 = L[3] * m, t11 = L[4] * m, t12 = L[5] * (float)h;
    float* dst = out + (size_t)p * ps * ps;
    __shared__ float tile[GS_TILE_CAP];
    const AffTile tl = aff_tile_box(t00, t01, t02, t10, t11, t12, ps, GS_TILE_CAP);     // uniform across the workgroup
    if (tl.tw > 0) {
        aff_tile_load<256>(tile, tl, img, h, w, threadIdx.x);
        __syncthreads();
        for (int i = threadIdx.x; i < ps * ps; i += 256) {
            const int r = i / ps, c = i - r * ps;
            dst[i] = aff_sample_bilinear_tile(tile, tl, img, h, w, t00, t01, t02, t10, t11, t12, bg.v[c], bg.v[r]);
        }
        return;
    }
    for (int i = threadIdx.x; i < ps * ps; i += 256) {
        const int r = i / ps, c = i - r * ps;
        dst[i] = aff_sample_bilinear(img, h, w, t00, t01, t02, t10, t11, t12, bg.v[c], bg.v[r]);
    }
}

static int sample_common(affnet_ctx* ctx, const float* img, int h, int w, int use_pyr, const float* lafs, const int32_t* ids,
                         const int32_t* cnt, int n, int ps, float* out, hipStream_t st) {
    if (ps < 1 || ps > MAX_PS) return aff_fail(ctx, AFFNET_ERR_INVALID, "patch size %d not in 1..%d", ps, MAX_PS);
    if (n <= 0) return AFFNET_OK;
    BaseGrid bg;
    memset(&bg, 0, sizeof(bg));
    aff_base_grid(ps, bg.v);
    PyrTable pt;
    if (use_pyr) aff_fill_pyr_table(ctx, &pt); else memset(&pt, 0, sizeof(pt));
    hipLaunchKernelGGL(grid_sample_kernel, dim3(n, use_pyr ? ctx->B : 1), dim3(256), 0, st, img, h, w, pt, use_pyr, lafs, ids, cnt, n, ps, bg,
                       out);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

extern "C" int affnet_laf_grid_sample(affnet_ctx* ctx, const float* d_img, int h, int w, const float* d_lafs, int n, int ps,
                                      float* d_out, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !d_img || !d_lafs || !d_out || h < 1 || w < 1 || n < 0) return aff_fail(ctx, AFFNET_ERR_INVALID, "laf_grid_sample: bad argument");
    return sample_common(ctx, d_img, h, w, 0, d_lafs, nullptr, nullptr, n, ps, d_out, (hipStream_t)stream);
}

extern "C" int affnet_pyr_grid_sample(affnet_ctx* ctx, const float* d_lafs, const int32_t* d_ids, const int32_t* d_count, int n_max,
                                      int ps, float* d_out, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !ctx->ws || !d_lafs || !d_ids || !d_out || n_max < 0) return aff_fail(ctx, AFFNET_ERR_INVALID, "pyr_grid_sample: bad argument");
    return sample_common(ctx, nullptr, 0, 0, 1, d_lafs, d_ids, d_count, n_max, ps, d_out, (hipStream_t)stream);
}

// ---- shape compose + filter ------------------------------------------------------------------------
// Rows [row_begin, row_end) of every image; the other rows keep what they hold (the caller zeroes key / good first).  With the
// lazy-evaluation predicate (skip_cnt != NULL, see cnn32.hip CnnArgs) the pass does nothing for an image that already has its
// skip_n survivors - its remaining candidates were never run through the shape CNN and stay "not good".  Survivors are counted
// into CNT_SURVIVED here (one atomic per wave); thread 0 of a second pass records how many candidates were evaluated at all.
__global__ __launch_bounds__(256) void shape_filter_kernel(const float* __restrict__ resp, const float* __restrict__ lafs,
                                                           const float* __restrict__ A, const int32_t* __restrict__ d_count,
                                                           int n_max, float* __restrict__ key, int32_t* __restrict__ good, int row_begin,
                                                           int row_end, int32_t* cnt, const int32_t* skip_cnt, int skip_n) {
    const int i = row_begin + blockIdx.x * 256 + threadIdx.x;
    {
        const size_t bi = blockIdx.y;
        resp += bi * n_max; lafs += bi * n_max * 6; A += bi * n_max * 4; d_count += bi; key += bi * n_max; good += bi * n_max;
        cnt += bi * CNT_TOTAL;
    }
    const int n = min(*d_count, n_max);
    bool skip = false;
    if (skip_cnt) {
        const int32_t* c = skip_cnt + (size_t)blockIdx.y * CNT_TOTAL;
        skip = c[CNT_SEL_MODE] == 1 && c[CNT_SURVIVED1] >= skip_n;      // frozen after the first pass (shape_freeze_kernel)
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt[CNT_AFF_EVAL] = skip ? min(n, row_begin) : min(n, row_end);
    if (skip || i >= row_end || i >= n) return;
    aff_shape_filter_row(resp, lafs, A[4 * i], A[4 * i + 1], A[4 * i + 2], A[4 * i + 3], i, key, good, cnt);
}

// Selection + emission in ONE launch, one 1024-thread workgroup per image (were: clear, 2-D rank sort with atomics, emit):
//   survivors > N  -> top-N of key = response * good (descending; ties by row index) - torch.topk branch (:151-153)
//   otherwise      -> stable compaction of the good rows                              - nonzero branch (:154-156)
// The detector hands over rows sorted by descending response (CNT_SEL_MODE == 1) whenever it had more candidates than C, and then
// "the N largest keys" are simply the first N good rows: both branches are a prefix count of the good flags.  Only when the rows
// are NOT sorted (fewer than C candidates, or caller-supplied rows of affnet_shape_filter_select that are not in descending response
// order: checked here row by row, the flag of the last detector call alone is not trusted) and there are more than N survivors - or
// the N-th good row's key is not positive, so that the zero keys of rejected rows compete with it - the keys are ranked by comparison
// (brute force from LDS tiles; rare and small).
__global__ __launch_bounds__(1024) void shape_select_kernel(const float* __restrict__ resp, const float* __restrict__ lafs,
                                                            const int32_t* __restrict__ ids, const float* __restrict__ A,
                                                            const float* __restrict__ key, const int32_t* __restrict__ good,
                                                            const int32_t* __restrict__ d_count, int n_max, int N, int out_cap, float* out_resp,
                                                            float* out_lafs, int32_t* out_ids, int32_t* out_count, int32_t* cnt) {
    __shared__ int s_wsum[16];
    __shared__ int s_general;
    __shared__ float t_key[1024];
    {
        const size_t bi = blockIdx.x;
        resp += bi * n_max; lafs += bi * n_max * 6; ids += bi * n_max * 3; A += bi * n_max * 4; key += bi * n_max;
        good += bi * n_max; d_count += bi; cnt += bi * CNT_TOTAL;
        out_resp += bi * out_cap; out_lafs += bi * out_cap * 6; out_ids += bi * out_cap * 3; out_count += bi;
    }
    const int n = min(*d_count, n_max);
    const int surv = cnt[CNT_SURVIVED];
    const bool topk = (N > 0) && (surv > N);
    const bool sorted = cnt[CNT_SEL_MODE] == 1;
    const int n_out = topk ? N : (surv < out_cap ? surv : out_cap);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) {
        *out_count = n_out; cnt[CNT_SHAPED] = n_out;
        if (!topk && surv > out_cap) atomicOr(&cnt[CNT_OVERFLOW], 8);
        s_general = (topk && !sorted) ? 1 : 0;
    }
    __syncthreads();
    // rows past the output count: zero (the caller's buffers are not cleared separately)
    for (int r = n_out + t; r < out_cap; r += 1024) {
        out_resp[r] = 0.0f;
#pragma unroll
        for (int k = 0; k < 6; ++k) out_lafs[6 * (size_t)r + k] = 0.0f;
        out_ids[3 * r] = 0; out_ids[3 * r + 1] = 0; out_ids[3 * r + 2] = 0;
    }
    // exclusive prefix count of the good flags: thread t owns rows [t * seg, (t + 1) * seg)
    const int seg = (n + 1023) / 1024;
    const int r0 = t * seg, r1 = min(r0 + seg, n);
    int mine = 0;
    bool unsorted = false;      // CNT_SEL_MODE is what the context's LAST detector call left behind; the rows handed to the public stage entry
                                // (affnet_shape_filter_select) may be anything: the prefix shortcut is taken only if they really are sorted
    for (int r = r0; r < r1; ++r) {
        mine += good[r] != 0;
        if (topk && sorted && r > 0 && !(resp[r] <= resp[r - 1])) unsorted = true;
    }
    if (unsorted) s_general = 1;     // (thread 0's initialisation of s_general precedes the first barrier below)
    int inc = mine;
#pragma unroll
    for (int ofs = 1; ofs < 64; ofs <<= 1) {
        const int v = __shfl_up(inc, ofs, 64);
        if (lane >= ofs) inc += v;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    int before = inc - mine;
    for (int w = 0; w < wave; ++w) before += s_wsum[w];
    if (topk && sorted && before < N && N <= before + mine) {          // the thread that owns the N-th good row: is its key positive?
        int c = before;
        for (int r = r0; r < r1; ++r)
            if (good[r] != 0 && ++c == N && !(key[r] > 0.0f)) s_general = 1;
    }
    __syncthreads();
    auto emit = [&](int i, int p) {
        out_resp[p] = topk ? key[i] : resp[i];
        const float a00 = A[4 * i], a01 = A[4 * i + 1], a10 = A[4 * i + 2], a11 = A[4 * i + 3];
        const float* L = lafs + 6 * (size_t)i;
        float* O = out_lafs + 6 * (size_t)p;
        O[0] = fmaf(a01, L[3], a00 * L[0]); O[1] = fmaf(a01, L[4], a00 * L[1]); O[2] = L[2];
        O[3] = fmaf(a11, L[3], a10 * L[0]); O[4] = fmaf(a11, L[4], a10 * L[1]); O[5] = L[5];
        out_ids[3 * p] = ids[3 * i]; out_ids[3 * p + 1] = ids[3 * i + 1]; out_ids[3 * p + 2] = ids[3 * i + 2];
    };
    if (!s_general) {
        int p = before;
        for (int r = r0; r < r1; ++r)
            if (good[r] != 0) { if (p < n_out) emit(r, p); ++p; }
        return;
    }
    // general top-N: rank of row i = rows j with key_j > key_i, or equal key and j < i (the rejected rows' zero keys take part)
    for (int i0 = 0; i0 < n; i0 += 1024) {                              // uniform trip count: barriers inside
        const int i = i0 + t;
        const float ki = i < n ? key[i] : 0.0f;
        int r = 0;
        for (int j0 = 0; j0 < n; j0 += 1024) {
            __syncthreads();
            t_key[t] = (j0 + t < n) ? key[j0 + t] : -INFINITY;
            __syncthreads();
            const int m = min(1024, n - j0);
            for (int q = 0; q < m; ++q) { const float kj = t_key[q]; r += (kj > ki) || (kj == ki && (j0 + q) < i); }
        }
        if (i < n && r < N) emit(i, r);
    }
}

__global__ void shape_freeze_kernel(int32_t* cnt) {        // survivors of the first pass, read by the second pass's predicates
    cnt += blockIdx.x * CNT_TOTAL;
    cnt[CNT_SURVIVED1] = cnt[CNT_SURVIVED];
}

__global__ void shape_begin_kernel(int32_t* cnt) {
    cnt += blockIdx.x * CNT_TOTAL;
    cnt[CNT_SURVIVED] = 0; cnt[CNT_SURVIVED1] = 0; cnt[CNT_AFF_EVAL] = 0;
}

// The shape stage in three steps, so that the fused pipeline can evaluate the shape CNN lazily (pipeline.hip):
//   begin  : key / good of every row = 0 ("not evaluated, not good"), survivor counters = 0
//   rows   : filter rows [row_begin, row_end) of every image (optionally under the lazy predicate), counting survivors;
//            freeze = true publishes the survivor count for the predicates of a following pass
//   select : top-N / compaction of the good rows -> outputs
int aff_shape_filter_begin(affnet_ctx* ctx, hipStream_t st) {
    const size_t P = (size_t)ctx->B * ctx->cap_pre;
    int rc = aff_zero_async(ctx, ctx->st_key, P * 2 * sizeof(float), st);        // st_key and st_good are adjacent (context.hip)
    if (rc) return rc;
    hipLaunchKernelGGL(shape_begin_kernel, dim3(ctx->B), dim3(1), 0, st, ctx->cnt);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

int aff_shape_filter_rows(affnet_ctx* ctx, const float* resp, const float* lafs, const float* A, const int32_t* count, int row_begin, int row_end,
                          bool lazy, hipStream_t st, bool freeze = false) {
    if (row_end > row_begin) {
        hipLaunchKernelGGL(shape_filter_kernel, dim3(aff_cdiv(row_end - row_begin, 256), ctx->B), dim3(256), 0, st, resp, lafs, A, count, ctx->cap_pre,
                           ctx->st_key, ctx->st_good, row_begin, row_end, ctx->cnt, lazy ? ctx->cnt : nullptr, ctx->cfg.num_features);
        AFF_LAUNCH_CHECK(ctx);
    }
    if (freeze) {
        hipLaunchKernelGGL(shape_freeze_kernel, dim3(ctx->B), dim3(1), 0, st, ctx->cnt);
        AFF_LAUNCH_CHECK(ctx);
    }
    return AFFNET_OK;
}

int aff_shape_select(affnet_ctx* ctx, const float* d_resp_in, const float* d_lafs_in, const int32_t* d_ids_in, const float* d_A,
                     const int32_t* d_count_in, float* d_resp_out, float* d_lafs_out, int32_t* d_ids_out, int32_t* d_count_out, hipStream_t st) {
    hipLaunchKernelGGL(shape_select_kernel, dim3(ctx->B), dim3(1024), 0, st, d_resp_in, d_lafs_in, d_ids_in, d_A, ctx->st_key, ctx->st_good, d_count_in,
                       ctx->cap_pre, ctx->cfg.num_features, ctx->cap_final, d_resp_out, d_lafs_out, d_ids_out, d_count_out, ctx->cnt);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

extern "C" int affnet_shape_filter_select(affnet_ctx* ctx, const float* d_resp_in, const float* d_lafs_in, const int32_t* d_ids_in,
                                          const float* d_A, const int32_t* d_count_in, float* d_resp_out, float* d_lafs_out,
                                          int32_t* d_ids_out, int32_t* d_count_out, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !ctx->ws || !d_resp_in || !d_lafs_in || !d_ids_in || !d_A || !d_count_in || !d_resp_out || !d_lafs_out || !d_ids_out ||
        !d_count_out)
        return aff_fail(ctx, AFFNET_ERR_INVALID, "shape_filter_select: bad argument");
    hipStream_t st = (hipStream_t)stream;
    int rc = aff_shape_filter_begin(ctx, st);
    if (rc) return rc;
    rc = aff_shape_filter_rows(ctx, d_resp_in, d_lafs_in, d_A, d_count_in, 0, ctx->cap_pre, false, st);
    if (rc) return rc;
    return aff_shape_select(ctx, d_resp_in, d_lafs_in, d_ids_in, d_A, d_count_in, d_resp_out, d_lafs_out, d_ids_out, d_count_out, st);
}

// ---- AffNet iterations (num_Baum_iters > 1, SparseImgRepresenter.py:127-146) ------------------------------------
// mode 0: lafs_out = [base * LAF_2x2 | centre]                         (new_LAFs for the re-extraction, :137)
// mode 1: base = A * base (bmm, k ascending, fused accumulate), then lafs_out as above   (:136-137)
__global__ __launch_bounds__(256) void shape_iterate_kernel(const float* __restrict__ A, float* __restrict__ base, const float* __restrict__ lafs,
                                                            const int32_t* __restrict__ d_count, int n_max, int mode,
                                                            float* __restrict__ lafs_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const size_t bi = blockIdx.y;
    const int n = min(d_count[bi], n_max);
    if (i >= n) return;
    const size_t r = bi * n_max + i;
    float b00 = base[4 * r], b01 = base[4 * r + 1], b10 = base[4 * r + 2], b11 = base[4 * r + 3];
    if (mode == 1) {
        const float a00 = A[4 * r], a01 = A[4 * r + 1], a10 = A[4 * r + 2], a11 = A[4 * r + 3];
        const float n00 = fmaf(a01, b10, a00 * b00), n01 = fmaf(a01, b11, a00 * b01);
        const float n10 = fmaf(a11, b10, a10 * b00), n11 = fmaf(a11, b11, a10 * b01);
        b00 = n00; b01 = n01; b10 = n10; b11 = n11;
        base[4 * r] = b00; base[4 * r + 1] = b01; base[4 * r + 2] = b10; base[4 * r + 3] = b11;
    }
    const float* L = lafs + 6 * r;
    float* O = lafs_out + 6 * r;
    O[0] = fmaf(b01, L[3], b00 * L[0]); O[1] = fmaf(b01, L[4], b00 * L[1]); O[2] = L[2];
    O[3] = fmaf(b11, L[3], b10 * L[0]); O[4] = fmaf(b11, L[4], b10 * L[1]); O[5] = L[5];
}

int aff_shape_iterate(affnet_ctx* ctx, const float* A, float* base, const float* lafs, const int32_t* count, int mode, float* lafs_out,
                      hipStream_t st) {
    hipLaunchKernelGGL(shape_iterate_kernel, dim3(aff_cdiv(ctx->cap_pre, 256), ctx->B), dim3(256), 0, st, A, base, lafs, count, ctx->cap_pre, mode,
                       lafs_out);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

// ---- small elementwise LAF kernels ---------------------------------------------------------------
__global__ void apply_rotation_kernel(float* __restrict__ lafs, const float* __restrict__ R, const int32_t* __restrict__ d_count,
                                      int n_max) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t bi = blockIdx.y;
    const int n = d_count ? min(d_count[bi], n_max) : n_max;
    if (i >= n) return;
    float* L = lafs + 6 * (bi * n_max + i);
    const float* r = R + 4 * (bi * n_max + i);
    const float l00 = L[0], l01 = L[1], l10 = L[3], l11 = L[4];
    L[0] = fmaf(l01, r[2], l00 * r[0]); L[1] = fmaf(l01, r[3], l00 * r[1]);
    L[3] = fmaf(l11, r[2], l10 * r[0]); L[4] = fmaf(l11, r[3], l10 * r[1]);
}

extern "C" int affnet_apply_rotation(affnet_ctx* ctx, float* d_lafs, const float* d_R, const int32_t* d_count, int n_max, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !d_lafs || !d_R || n_max < 0) return aff_fail(ctx, AFFNET_ERR_INVALID, "apply_rotation: bad argument");
    if (n_max == 0) return AFFNET_OK;
    hipLaunchKernelGGL(apply_rotation_kernel, dim3(aff_cdiv(n_max, 256), ctx->B), dim3(256), 0, (hipStream_t)stream, d_lafs, d_R, d_count, n_max);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

__global__ void scale_lafs_kernel(const float* __restrict__ in, float* __restrict__ out, const int32_t* __restrict__ d_count, int n_max,
                                  float c_a, float c_x, float c_y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t bi = blockIdx.y;
    const int n = d_count ? min(d_count[bi], n_max) : n_max;
    if (i >= n_max) return;
    const float* L = in + 6 * (bi * n_max + i);
    float* O = out + 6 * (bi * n_max + i);
    if (i >= n) { O[0] = O[1] = O[2] = O[3] = O[4] = O[5] = 0.f; return; }
    O[0] = c_a * L[0]; O[1] = c_a * L[1]; O[2] = c_x * L[2];
    O[3] = c_a * L[3]; O[4] = c_a * L[4]; O[5] = c_y * L[5];
}

extern "C" int affnet_scale_lafs(affnet_ctx* ctx, const float* d_in, float* d_out, const int32_t* d_count, int n_max, int w, int h,
                                 int inverse, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !d_in || !d_out || n_max < 0 || w < 1 || h < 1) return aff_fail(ctx, AFFNET_ERR_INVALID, "scale_lafs: bad argument");
    if (n_max == 0) return AFFNET_OK;
    const float fw = (float)w, fh = (float)h, m = fw < fh ? fw : fh;
    // normalizeLAFs: ones/min_size in fp32, 1.0/w and 1.0/h are python doubles stored to fp32 (LAF.py:424-426)
    const float ca = inverse ? 1.0f / m : m;
    const float cx = inverse ? (float)(1.0 / (double)fw) : fw;
    const float cy = inverse ? (float)(1.0 / (double)fh) : fh;
    hipLaunchKernelGGL(scale_lafs_kernel, dim3(aff_cdiv(n_max, 256), ctx->B), dim3(256), 0, (hipStream_t)stream, d_in, d_out, d_count, n_max, ca, cx,
                       cy);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}


__global__ void level_select_kernel(const float* __restrict__ lafs_px, const int32_t* __restrict__ d_count, int n_max, float ps,
                                    LevelTable lt, float ca, float cx, float cy, int32_t* __restrict__ ids, float* __restrict__ lafs_norm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t bi = blockIdx.y;
    const int n = d_count ? min(d_count[bi], n_max) : n_max;
    if (i >= n) return;
    lafs_px += bi * n_max * 6; ids += bi * n_max * 3; lafs_norm += bi * n_max * 6;
    const float* L = lafs_px + 6 * (size_t)i;
    // get_LAFs_scales (LAF.py:450-451) in fp32, then / PS in fp32, then float64 |a - b| argmin (cdist on 1-D points)
    const float p1 = L[0] * L[4], p2 = L[1] * L[3];
    const float sc = sqrtf(fabsf(p1 - p2) + 1e-12f);
    const double need = (double)(sc / ps);
    int best = 0;
    double bd = INFINITY;
    const int tot = lt.n_oct * lt.n_lvl;
    for (int k = 0; k < tot; ++k) {
        const double df = lt.sig[k] - need;
        const double d = sqrt(df * df);               // scipy cdist 'euclidean' on 1-D points
        if (d < bd) { bd = d; best = k; }
    }
    ids[3 * i] = best / lt.n_lvl; ids[3 * i + 1] = best % lt.n_lvl; ids[3 * i + 2] = 0;
    float* O = lafs_norm + 6 * (size_t)i;
    O[0] = ca * L[0]; O[1] = ca * L[1]; O[2] = cx * L[2];
    O[3] = ca * L[3]; O[4] = ca * L[4]; O[5] = cy * L[5];
}

// scale_lafs_kernel (denormalizeLAFs) + level_select_kernel in one launch (the fused pipeline): out_px = denormalised frames, then
// the level choice and the re-normalised frames from out_px's values exactly as the two kernels compute them.
__global__ void denorm_level_select_kernel(const float* __restrict__ in, float* __restrict__ out_px, const int32_t* __restrict__ d_count, int n_max,
                                           float c_a, float c_x, float c_y, float ps, LevelTable lt, float ca, float cx, float cy,
                                           int32_t* __restrict__ ids, float* __restrict__ lafs_norm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t bi = blockIdx.y;
    const int n = d_count ? min(d_count[bi], n_max) : n_max;
    if (i >= n_max) return;
    const float* L = in + 6 * (bi * n_max + i);
    float* P = out_px + 6 * (bi * n_max + i);
    if (i >= n) { P[0] = P[1] = P[2] = P[3] = P[4] = P[5] = 0.f; return; }
    aff_denorm_level_row(L[0], L[1], L[2], L[3], L[4], L[5], c_a, c_x, c_y, ps, lt, ca, cx, cy, P, ids + 3 * (bi * n_max + i), lafs_norm + 6 * (bi * n_max + i));
}

// The constants of denorm_level_select_kernel for this context, for a kernel that fuses the step (OriNet's finish kernel, cnn32.hip).
void aff_denorm_sel_fill(affnet_ctx* ctx, int ps, float* d_lafs_px, int32_t* d_ids, float* d_lafs_norm, DenormSel* ds) {
    const affnet_config& c = ctx->cfg;
    ds->lt.n_oct = c.n_octaves; ds->lt.n_lvl = c.levels_per_octave;
    for (int o = 0; o < ds->lt.n_oct; ++o)
        for (int l = 0; l < ds->lt.n_lvl; ++l) ds->lt.sig[o * ds->lt.n_lvl + l] = c.level_sigma_px[o][l];
    const float fw = (float)c.width, fh = (float)c.height, m = fw < fh ? fw : fh;
    ds->out_px = d_lafs_px; ds->ids = d_ids; ds->lafs_norm = d_lafs_norm;
    ds->c_a = m; ds->c_x = fw; ds->c_y = fh; ds->ps = (float)ps;
    ds->ca = 1.0f / m; ds->cx = (float)(1.0 / (double)fw); ds->cy = (float)(1.0 / (double)fh);
}

int aff_denorm_level_select(affnet_ctx* ctx, const float* d_lafs_norm_in, float* d_lafs_px, const int32_t* d_count, int n_max, int ps, int32_t* d_ids,
                            float* d_lafs_norm, hipStream_t st) {
    if (n_max == 0) return AFFNET_OK;
    LevelTable lt;
    const affnet_config& c = ctx->cfg;
    lt.n_oct = c.n_octaves; lt.n_lvl = c.levels_per_octave;
    for (int o = 0; o < lt.n_oct; ++o)
        for (int l = 0; l < lt.n_lvl; ++l) lt.sig[o * lt.n_lvl + l] = c.level_sigma_px[o][l];
    const float fw = (float)c.width, fh = (float)c.height, m = fw < fh ? fw : fh;
    hipLaunchKernelGGL(denorm_level_select_kernel, dim3(aff_cdiv(n_max, 256), ctx->B), dim3(256), 0, st, d_lafs_norm_in, d_lafs_px, d_count, n_max, m, fw,
                       fh, (float)ps, lt, 1.0f / m, (float)(1.0 / (double)fw), (float)(1.0 / (double)fh), d_ids, d_lafs_norm);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

extern "C" int affnet_level_select(affnet_ctx* ctx, const float* d_lafs_px, const int32_t* d_count, int n_max, int ps, int32_t* d_ids,
                                   float* d_lafs_norm, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !d_lafs_px || !d_ids || !d_lafs_norm || n_max < 0 || ps < 1) return aff_fail(ctx, AFFNET_ERR_INVALID, "level_select: bad argument");
    if (n_max == 0) return AFFNET_OK;
    LevelTable lt;
    const affnet_config& c = ctx->cfg;
    lt.n_oct = c.n_octaves; lt.n_lvl = c.levels_per_octave;
    for (int o = 0; o < lt.n_oct; ++o)
        for (int l = 0; l < lt.n_lvl; ++l) lt.sig[o * lt.n_lvl + l] = c.level_sigma_px[o][l];
    const float fw = (float)c.width, fh = (float)c.height, m = fw < fh ? fw : fh;
    hipLaunchKernelGGL(level_select_kernel, dim3(aff_cdiv(n_max, 256), ctx->B), dim3(256), 0, (hipStream_t)stream, d_lafs_px, d_count, n_max,
                       (float)ps, lt, 1.0f / m, (float)(1.0 / (double)fw), (float)(1.0 / (double)fh), d_ids, d_lafs_norm);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

// ---- LAF -> Oxford ellipse on the device ----------------------------------------------------
// Replaces LAF.py:35-51 (LAFs2ellT) + :106-144 (bsvd2x2, closed-form batched 2x2 SVD), operation by operation in fp32.
__global__ void lafs2ell_kernel(const float* __restrict__ lafs, const int32_t* __restrict__ d_count, int n_max, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t bi = blockIdx.y;
    const int n = d_count ? min(d_count[bi], n_max) : n_max;
    if (i >= n_max) return;
    float* E = out + 5 * (bi * n_max + i);
    if (i >= n) { E[0] = E[1] = E[2] = E[3] = E[4] = 0.f; return; }
    const float* L = lafs + 6 * (bi * n_max + i);
    const float scale = sqrtf((L[0] * L[4] - L[1] * L[3]) + 1e-10f);
    const float a00 = L[0] / scale, a01 = L[1] / scale, a10 = L[3] / scale, a11 = L[4] / scale;
    // Su = As * As^T (bmm: k ascending, fused accumulate)
    const float su00 = fmaf(a01, a01, a00 * a00), su01 = fmaf(a01, a11, a00 * a10), su10 = fmaf(a11, a01, a10 * a00),
                su11 = fmaf(a11, a11, a10 * a10);
    const float phi = 0.5f * atan2f((su01 + su10) + 1e-12f, (su00 - su11) + 1e-12f);
    const float cp = cosf(phi), sp = sinf(phi);                       // U = [[cp, -sp], [sp, cp]]
    const float susum = su00 + su11;
    const float dif = su00 - su11;
    const float sudif = sqrtf((dif * dif + (4.0f * su01) * su10) + 1e-12f);
    const float sig0 = sqrtf((susum + sudif) / 2.0f), sig1 = sqrtf((susum - sudif) / 2.0f);
    // W' = diag(1 / (scale^2 sig^2));  A = U W' U^T
    const float w0 = 1.0f / ((scale * scale) * (sig0 * sig0)), w1 = 1.0f / ((scale * scale) * (sig1 * sig1));
    // (U W')[r][c] = U[r][c] * w_c (bmm with a diagonal: the other term is an exact +0)
    const float m00 = cp * w0, m01 = -sp * w1, m10 = sp * w0, m11 = cp * w1;
    // A = M U^T : A[r][c] = M[r][0] U[c][0] + M[r][1] U[c][1]
    E[0] = L[2]; E[1] = L[5];
    E[2] = fmaf(m01, -sp, m00 * cp);
    E[3] = fmaf(m01, cp, m00 * sp);
    E[4] = fmaf(m11, cp, m10 * sp);
}

extern "C" int affnet_lafs_to_ellipses(affnet_ctx* ctx, const float* d_lafs, const int32_t* d_count, int n_max, float* d_out, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !d_lafs || !d_out || n_max < 0) return aff_fail(ctx, AFFNET_ERR_INVALID, "lafs_to_ellipses: bad argument");
    if (n_max == 0) return AFFNET_OK;
    hipLaunchKernelGGL(lafs2ell_kernel, dim3(aff_cdiv(n_max, 256), ctx->B), dim3(256), 0, (hipStream_t)stream, d_lafs, d_count, n_max, d_out);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}
