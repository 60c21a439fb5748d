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
