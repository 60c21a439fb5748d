// Descriptor matching for gfx950 (the first consumer of the descriptors).
//
// Replaces Losses.py:5-13 (distance_matrix_vector), the second-nearest-neighbour ratio test of
// train_AffNet_test_on_graffity.py:292-300 and ReprojectionStuff.py:9-40,126-137 (linH, reprojectLAFs,
// get_GT_correspondence_indexes).
//
// The n1 x n2 distance matrix (36 MB at 3000 x 3000) is never materialised: rowmin_dist_kernel computes 64 x 16 tiles
// of a.b on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, K = descriptor length), turns them into
// sqrt((|a|^2 + |b|^2) - 2 a.b + 1e-6) in the reference's operation order and keeps a running (min, argmin) per row.
// The reference finds the "second nearest" by overwriting the COLUMNS of all nearest neighbours with 100000
// (dist_matrix[:, idxs_in_2] = 100000) and taking the row minimum again: the same kernel runs a second time with a
// per-column `used` mask, so the quirk (a column masked for every row, not only for its own) is reproduced.
#include <math.h>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ x, int n, int dim, float* __restrict__ out) {
    // one wavefront per row; torch.sum(a * a, dim=1)
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    float s = 0.f;
    for (int k = lane; k < dim; k += 64) { const float v = x[(size_t)row * dim + k]; s = fmaf(v, v, s); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[row] = s;
}

// Workgroup = 4 wavefronts = 64 rows of `a`; wave w owns rows 16w..16w+15 and walks all columns of `b` in tiles of 16.
// MFMA roles: A operand = a rows (lane (m, kq) holds a[row m][16 t + 4 kq + j]), B operand = b rows (columns of the
// distance matrix); the K order is permuted identically on both sides, which a dot product does not see.
// Result tile: lane (n = l & 15, g = l >> 4) holds D[row 4 g + r][col n], r = 0..3.
template <int DIM>
__global__ __launch_bounds__(256) void rowmin_dist_kernel(const float* __restrict__ a, int n1, const float* __restrict__ b, int n2,
                                                          const float* __restrict__ a_sq, const float* __restrict__ b_sq,
                                                          const uint8_t* __restrict__ used, float* __restrict__ out_min,
                                                          int32_t* __restrict__ out_idx, float* __restrict__ out_full) {
    constexpr int NG = DIM / 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * 64 + wave * 16;
    if (row0 >= n1) return;
    f32x4 fa[NG];
    {
        const int r = min(row0 + m, n1 - 1);
#pragma unroll
        for (int t = 0; t < NG; ++t) fa[t] = *reinterpret_cast<const f32x4*>(&a[(size_t)r * DIM + 16 * t + 4 * kq]);
    }
    float asq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) asq[r] = a_sq[min(row0 + 4 * kq + r, n1 - 1)];
    float best[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int bidx[4] = {0, 0, 0, 0};
    for (int c0 = 0; c0 < n2; c0 += 16) {
        const int col = c0 + m;
        const int cc = min(col, n2 - 1);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        f32x4 fb[NG];
#pragma unroll
        for (int t = 0; t < NG; ++t) fb[t] = *reinterpret_cast<const f32x4*>(&b[(size_t)cc * DIM + 16 * t + 4 * kq]);
#pragma unroll
        for (int t = 0; t < NG; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[t][j], fb[t][j], acc, 0, 0, 0);
        const float bsq = b_sq[cc];
        const bool masked = used && used[cc];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // Losses.py:12-13: sqrt((d1_sq + d2_sq) - 2.0 * a.b + eps)
            float d = sqrtf(((asq[r] + bsq) - 2.0f * acc[r]) + 1e-6f);
            if (masked) d = 100000.0f;
            // strict <: the first minimum wins; a NaN (the fp32 expansion can go below -1e-6 for near-identical vectors)
            // is sticky, as in torch.min
            if (col < n2 && (d < best[r] || (d != d && best[r] == best[r]))) { best[r] = d; bidx[r] = col; }
            if (out_full && col < n2 && row0 + 4 * kq + r < n1) out_full[(size_t)(row0 + 4 * kq + r) * n2 + col] = d;
        }
    }
    // minimum over the 16 column lanes (same g): butterfly inside each row of 16 lanes, ties -> smaller column
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best[r], o, 64);
            const int oi = __shfl_xor(bidx[r], o, 64);
            const bool onan = ob != ob, bnan = best[r] != best[r];
            if ((onan && (!bnan || oi < bidx[r])) || (!onan && !bnan && (ob < best[r] || (ob == best[r] && oi < bidx[r])))) {
                best[r] = ob; bidx[r] = oi;
            }
        }
        const int row = row0 + 4 * kq + r;
        if (m == 0 && row < n1 && out_min) { out_min[row] = best[r]; out_idx[row] = bidx[r]; }
    }
}

__global__ void mark_used_kernel(const int32_t* __restrict__ idx, int n1, uint8_t* __restrict__ used) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n1) used[idx[i]] = 1;
}

// One workgroup: ratio test + order-preserving compaction of the tentative matches (train_AffNet_test_on_graffity.py:296-300).
__global__ __launch_bounds__(1024) void snn_select_kernel(const float* __restrict__ min1, const float* __restrict__ min2,
                                                          const int32_t* __restrict__ idx, int n1, float thr, int32_t* __restrict__ tent,
                                                          int32_t* __restrict__ count) {
    __shared__ int s_wave[16];
    __shared__ int s_base;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i0 = 0; i0 < n1; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        bool keep = false;
        if (i < n1) keep = (min1[i] / (min2[i] + 1e-8f)) <= thr;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        if (keep) {
            const int slot = off + __popcll(bal & ((1ull << lane) - 1ull));
            tent[2 * slot] = i; tent[2 * slot + 1] = idx[i];
        }
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += s_wave[w]; s_base += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = s_base;
}

extern "C" int affnet_match_snn(affnet_ctx* ctx, const float* d_desc1, int n1, const float* d_desc2, int n2, int dim, float snn_threshold,
                                float* d_min_dist, int32_t* d_idx, float* d_min2_dist, int32_t* d_tent, int32_t* d_count, void* d_scratch,
                                void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !d_desc1 || !d_desc2 || !d_min_dist || !d_idx || !d_min2_dist || !d_tent || !d_count || !d_scratch || n1 < 0 || n2 < 1)
        return aff_fail(ctx, AFFNET_ERR_INVALID, "match_snn: bad argument");
    if (dim != 128) return aff_fail(ctx, AFFNET_ERR_INVALID, "match_snn: descriptor length %d (only 128 is built)", dim);
    hipStream_t st = (hipStream_t)stream;
    { int zrc = aff_zero_async(ctx, d_count, sizeof(int32_t), st); if (zrc) return zrc; }
    if (n1 == 0) return AFFNET_OK;
    float* a_sq = (float*)d_scratch;
    float* b_sq = a_sq + n1;
    uint8_t* used = (uint8_t*)(b_sq + n2);
    int32_t* idx2 = (int32_t*)(used + aff_align((size_t)n2, 16));
    { int zrc = aff_zero_async(ctx, used, (size_t)n2, st); if (zrc) return zrc; }
    hipLaunchKernelGGL(sqnorm_kernel, dim3(aff_cdiv(n1, 4)), dim3(256), 0, st, d_desc1, n1, dim, a_sq);
    hipLaunchKernelGGL(sqnorm_kernel, dim3(aff_cdiv(n2, 4)), dim3(256), 0, st, d_desc2, n2, dim, b_sq);
    hipLaunchKernelGGL(rowmin_dist_kernel<128>, dim3(aff_cdiv(n1, 64)), dim3(256), 0, st, d_desc1, n1, d_desc2, n2, a_sq, b_sq,
                       (const uint8_t*)nullptr, d_min_dist, d_idx, (float*)nullptr);
    hipLaunchKernelGGL(mark_used_kernel, dim3(aff_cdiv(n1, 256)), dim3(256), 0, st, d_idx, n1, used);
    hipLaunchKernelGGL(rowmin_dist_kernel<128>, dim3(aff_cdiv(n1, 64)), dim3(256), 0, st, d_desc1, n1, d_desc2, n2, a_sq, b_sq,
                       (const uint8_t*)used, d_min2_dist, idx2, (float*)nullptr);
    hipLaunchKernelGGL(snn_select_kernel, dim3(1), dim3(1024), 0, st, d_min_dist, d_min2_dist, d_idx, n1, snn_threshold, d_tent, d_count);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

// Full distance matrix (Losses.py:5-13) for callers that want it: d_out (n1, n2).
extern "C" int affnet_distance_matrix(affnet_ctx* ctx, const float* d_desc1, int n1, const float* d_desc2, int n2, int dim, float* d_out,
                                      void* d_scratch, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !d_desc1 || !d_desc2 || !d_out || !d_scratch || n1 < 0 || n2 < 0) return aff_fail(ctx, AFFNET_ERR_INVALID, "distance_matrix: bad argument");
    if (dim != 128) return aff_fail(ctx, AFFNET_ERR_INVALID, "distance_matrix: descriptor length %d (only 128 is built)", dim);
    if (n1 == 0 || n2 == 0) return AFFNET_OK;
    hipStream_t st = (hipStream_t)stream;
    float* a_sq = (float*)d_scratch;
    float* b_sq = a_sq + n1;
    hipLaunchKernelGGL(sqnorm_kernel, dim3(aff_cdiv(n1, 4)), dim3(256), 0, st, d_desc1, n1, dim, a_sq);
    hipLaunchKernelGGL(sqnorm_kernel, dim3(aff_cdiv(n2, 4)), dim3(256), 0, st, d_desc2, n2, dim, b_sq);
    hipLaunchKernelGGL(rowmin_dist_kernel<128>, dim3(aff_cdiv(n1, 64)), dim3(256), 0, st, d_desc1, n1, d_desc2, n2, a_sq, b_sq,
                       (const uint8_t*)nullptr, (float*)nullptr, (int32_t*)nullptr, d_out);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

extern "C" size_t affnet_match_scratch_bytes(int n1, int n2) {
    return (size_t)(n1 + n2) * sizeof(float) + aff_align((size_t)n2, 16) + (size_t)n1 * sizeof(int32_t) + 64;
}

// ---- homography reprojection of LAFs + nearest reprojected centre ---------------------------------------------
struct Hom { float h[9]; };

// ReprojectionStuff.py:9-40: centre through H (bmm row-by-column, k ascending), local affine linH at the centre times the
// LAF's 2x2 part.
__global__ void reproject_lafs_kernel(const float* __restrict__ lafs, int n, Hom H, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* L = lafs + 6 * (size_t)i;
    const float x = L[2], y = L[5];
    const float* h = H.h;
    // xy1 = H * [x, y, 1]^T, then / z
    const float X = fmaf(h[2], 1.0f, fmaf(h[1], y, h[0] * x));
    const float Y = fmaf(h[5], 1.0f, fmaf(h[4], y, h[3] * x));
    const float Z = fmaf(h[8], 1.0f, fmaf(h[7], y, h[6] * x));
    // linH (:9-21), operation by operation
    const float den = (x * h[6] + y * h[7]) + h[8];
    const float n1d = ((x * h[0] + y * h[1]) + h[2]) / (den * den);
    const float n2d = ((x * h[3] + y * h[4]) + h[5]) / (den * den);
    const float a00 = h[0] / den - n1d * h[6], a01 = h[1] / den - n1d * h[7];
    const float a10 = h[3] / den - n2d * h[6], a11 = h[4] / den - n2d * h[7];
    float* O = out + 6 * (size_t)i;
    O[0] = fmaf(a01, L[3], a00 * L[0]); O[1] = fmaf(a01, L[4], a00 * L[1]); O[2] = X / Z;
    O[3] = fmaf(a11, L[3], a10 * L[0]); O[4] = fmaf(a11, L[4], a10 * L[1]); O[5] = Y / Z;
}

extern "C" int affnet_reproject_lafs(affnet_ctx* ctx, const float* d_lafs, int n, const float* h_H, float* d_out, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !d_lafs || !h_H || !d_out || n < 0) return aff_fail(ctx, AFFNET_ERR_INVALID, "reproject_lafs: bad argument");
    if (n == 0) return AFFNET_OK;
    Hom H;
    memcpy(H.h, h_H, sizeof(H.h));
    hipLaunchKernelGGL(reproject_lafs_kernel, dim3(aff_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, d_lafs, n, H, d_out);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

// For every query LAF centre (image-1 LAFs): nearest reference centre (the reprojected image-2 LAFs) with
// ReprojectionStuff.py:78-86's own distance_matrix_vector on 2-D points: sqrt(|(|a|^2 + |p|^2) - 2 p.a| + 1e-12) with
// a = reference ("anchor"), p = query ("positive"), in fp32 - at ~800 px coordinates that expansion is only good to
// ~0.1 px, and the 6 px consistency threshold of test() is applied to exactly this value.  ReprojectionStuff.py:126-137.
__global__ void centre_nn_kernel(const float* __restrict__ q, int nq, const float* __restrict__ ref, int nr, float* __restrict__ out_min,
                                 int32_t* __restrict__ out_idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const float px = q[6 * (size_t)i + 2], py = q[6 * (size_t)i + 5];
    const float psq = px * px + py * py;
    float best = INFINITY;
    int bi = 0;
    for (int j = 0; j < nr; ++j) {
        const float ax = ref[6 * (size_t)j + 2], ay = ref[6 * (size_t)j + 5];
        const float asq = ax * ax + ay * ay;
        const float dot = fmaf(py, ay, px * ax);
        const float d = sqrtf(fabsf((asq + psq) - 2.0f * dot) + 1e-12f);
        if (d < best) { best = d; bi = j; }
    }
    out_min[i] = best; out_idx[i] = bi;
}

extern "C" int affnet_centre_nn(affnet_ctx* ctx, const float* d_query_lafs, int nq, const float* d_ref_lafs, int nr, float* d_min_dist,
                                int32_t* d_idx, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !d_query_lafs || !d_ref_lafs || !d_min_dist || !d_idx || nq < 0 || nr < 1)
        return aff_fail(ctx, AFFNET_ERR_INVALID, "centre_nn: bad argument");
    if (nq == 0) return AFFNET_OK;
    hipLaunchKernelGGL(centre_nn_kernel, dim3(aff_cdiv(nq, 128)), dim3(128), 0, (hipStream_t)stream, d_query_lafs, nq, d_ref_lafs, nr, d_min_dist,
                       d_idx);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}
