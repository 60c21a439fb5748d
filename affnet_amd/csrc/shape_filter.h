// Shape filter of one candidate row, shared by shape_filter_kernel (laf_ops.hip) and the AffNet finish kernel (cnn32.hip) that
// fuses it: SparseImgRepresenter.py:121-162 (compose), Utils.py:168-175 (batch_eig2x2), LAF.py:98-104 (checkTouchBoundary), op by op.
#pragma once
#include "common.h"

// resp / lafs / key / good / cnt point at the image's rows; (a00, a01, a10, a11) = the row's shape matrix.  Writes key[i] / good[i] and
// counts survivors into cnt[CNT_SURVIVED].
__device__ __forceinline__ void aff_shape_filter_row(const float* __restrict__ resp, const float* __restrict__ lafs, float a00, float a01, float a10,
                                                     float a11, int i, float* __restrict__ key, int32_t* __restrict__ good, int32_t* cnt) {
    // base_A = bmm(A, I) = A exactly (SparseImgRepresenter.py:136, one iteration)
    const float* L = lafs + 6 * (size_t)i;
    // new_LAF = [base_A * LAF_2x2 | centre]: bmm row-by-column, k ascending, fused accumulate
    const float n00 = fmaf(a01, L[3], a00 * L[0]), n01 = fmaf(a01, L[4], a00 * L[1]);
    const float n10 = fmaf(a11, L[3], a10 * L[0]), n11 = fmaf(a11, L[4], a10 * L[1]);
    const float cx = L[2], cy = L[5];
    // batch_eig2x2 (Utils.py:168-175), op by op
    const float tr = a00 + a11;
    const float p1 = a00 * a11, p2 = a10 * a01;
    const float d1 = tr * tr - 4.0f * (p1 - p2);
    const float mk = d1 > 0.f ? 1.0f : 0.0f;
    const float dl = sqrtf(fabsf(d1));
    const float l1 = mk * (tr + dl) / 2.0f + 1000.0f * (1.0f - mk);
    const float l2 = mk * (tr - dl) / 2.0f + 0.0001f * (1.0f - mk);
    const float ratio = fabsf(l1 / (l2 + 1e-8f));
    bool ok = (ratio < 6.0f) && (ratio > (float)(1.0 / 6.0));
    // checkTouchBoundary (LAF.py:98-104): corners (+-1,+-1) of the frame must stay inside [0,1]^2
    const float px[4] = {-1.f, -1.f, 1.f, 1.f}, py[4] = {-1.f, 1.f, -1.f, 1.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float ox = fmaf(cx, 1.0f, fmaf(n01, py[k], n00 * px[k]));
        const float oy = fmaf(cy, 1.0f, fmaf(n11, py[k], n10 * px[k]));
        if (ox > 1.0f || ox < 0.0f || oy > 1.0f || oy < 0.0f) ok = false;
    }
    good[i] = ok ? 1 : 0;
    key[i] = resp[i] * (ok ? 1.0f : 0.0f);
    if (ok) atomicAdd(&cnt[CNT_SURVIVED], 1);
}

// Denormalisation (LAF.py:407-417) + pyramid-level choice (LAF.py:450-472, float64 |a - b| argmin like scipy's cdist on 1-D points) + re-normalised frame
// of ONE row, shared by denorm_level_select_kernel (laf_ops.hip) and OriNet's finish kernel (cnn32.hip), which fuses it behind the rotation.
struct LevelTable { double sig[AFFNET_MAX_OCTAVES * AFFNET_MAX_LEVELS]; int n_oct, n_lvl; };
struct DenormSel {           // per-launch constants + outputs (pointers of image 0, rows of image b at b * n_max); out_px == NULL: not fused
    float* out_px; int32_t* ids; float* lafs_norm;
    float c_a, c_x, c_y, ps, ca, cx, cy;
    LevelTable lt;
};
__device__ __forceinline__ void aff_denorm_level_row(float l0, float l1, float l2, float l3, float l4, float l5, float c_a, float c_x, float c_y, float ps,
                                                     const LevelTable& lt, float ca, float cx, float cy, float* __restrict__ P, int32_t* __restrict__ I,
                                                     float* __restrict__ O) {
    const float q0 = c_a * l0, q1 = c_a * l1, q2 = c_x * l2, q3 = c_a * l3, q4 = c_a * l4, q5 = c_y * l5;
    P[0] = q0; P[1] = q1; P[2] = q2; P[3] = q3; P[4] = q4; P[5] = q5;
    const float p1 = q0 * q4, p2 = q1 * q3;
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
    I[0] = best / lt.n_lvl; I[1] = best % lt.n_lvl; I[2] = 0;
    O[0] = ca * q0; O[1] = ca * q1; O[2] = cx * q2;
    O[3] = ca * q3; O[4] = ca * q4; O[5] = cy * q5;
}

// The same row with the level search spread over the lanes of a wavefront (OriNet's finish kernel owns one wave per row: a single lane walking the 30 - 48
// float64 distances made that kernel 4 x longer at 64 000 rows per launch).  EVERY lane calls it with the same row values; lane k evaluates level k's distance
// with the expression of the sequential loop, the minimum is reduced with ties going to the LOWER index - what the loop's strict `<` selects - and lane 0 writes.
__device__ __forceinline__ void aff_denorm_level_row_wave(int lane, float l0, float l1, float l2, float l3, float l4, float l5, float c_a, float c_x, float c_y, float ps,
                                                          const LevelTable& lt, float ca, float cx, float cy, float* __restrict__ P, int32_t* __restrict__ I,
                                                          float* __restrict__ O) {
    const float q0 = c_a * l0, q1 = c_a * l1, q2 = c_x * l2, q3 = c_a * l3, q4 = c_a * l4, q5 = c_y * l5;
    const float p1 = q0 * q4, p2 = q1 * q3;
    const float sc = sqrtf(fabsf(p1 - p2) + 1e-12f);
    const double need = (double)(sc / ps);
    const int tot = lt.n_oct * lt.n_lvl;
    double bd = INFINITY;
    int best = 0x7fffffff;
    for (int k = lane; k < tot; k += 64) {           // tot <= 128: at most two levels per lane, ascending
        const double df = lt.sig[k] - need;
        const double d = sqrt(df * df);
        if (d < bd) { bd = d; best = k; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double od = __shfl_xor(bd, off, 64);
        const int ok = __shfl_xor(best, off, 64);
        if (od < bd || (od == bd && ok < best)) { bd = od; best = ok; }
    }
    if (lane != 0) return;
    if (best == 0x7fffffff) best = 0;                // a NaN frame: no distance compares below infinity - the sequential loop stays at level 0
    P[0] = q0; P[1] = q1; P[2] = q2; P[3] = q3; P[4] = q4; P[5] = q5;
    I[0] = best / lt.n_lvl; I[1] = best % lt.n_lvl; I[2] = 0;
    O[0] = ca * q0; O[1] = ca * q1; O[2] = cx * q2;
    O[3] = ca * q3; O[4] = ca * q4; O[5] = cy * q5;
}
