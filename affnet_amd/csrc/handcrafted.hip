// The reference's hand-crafted default slot fillers for gfx950: OrientationDetector
// (HandCraftedModules.py:133-192, dominant gradient orientation from a 36-bin histogram) and AffineShapeEstimator
// (:81-132, Baumberg second-moment shape) on 19x19 patches - the slots a default-constructed
// ScaleSpaceAffinePatchExtractor uses (SparseImgRepresenter.py:42-49) and hesaffBaum.py runs for 16 iterations.
//
// One wavefront per patch: the 361 pixels are sampled from the pyramid (or loaded) into LDS, gradients use replicate
// padding like F.pad(..., 'replicate'), every elementwise step follows the reference's fp32 operation order
// (-ffp-contract=off); only the 361-term means differ in summation order.  The Gaussian weight tables are computed on the
// host with the reference's CircularGaussKernel formula (host_plan.py) and passed by value.
#include <math.h>

#include "common.h"

#define HC_PS 19
#define HC_N (HC_PS * HC_PS)

struct HcTables {
    float gk[HC_N];        // orientation: 10 * CircularGaussKernel(kernlen=19); Baumberg: CircularGaussKernel(19, sigma=19/2/3)
    float base[HC_PS];     // affine_grid base coordinates for PS = 19
};

__device__ __forceinline__ float hc_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// KIND 0: orientation -> out (n,2,2) rotation [[cos, sin], [-sin, cos]] (LAF.py:306-311) and out_angle (n) if non-null
// KIND 1: Baumberg    -> out (n,2,2) rectified shape matrix
template <int KIND>
__global__ __launch_bounds__(64) void hc19_kernel(const float* __restrict__ patches, PyrTable pt, const float* __restrict__ lafs,
                                                  const int32_t* __restrict__ ids, const int32_t* __restrict__ count, int n_max, HcTables tb,
                                                  float* __restrict__ out, float* __restrict__ out_angle) {
    __shared__ float px[HC_N];
    __shared__ float wv[HC_N];
    __shared__ int bn[HC_N];
    __shared__ float hist[40];
    const size_t bi = blockIdx.y;
    const int n = count ? min(count[bi], n_max) : n_max;
    if ((int)blockIdx.x >= n) return;
    const size_t pidx = bi * n_max + blockIdx.x;
    const int lane = threadIdx.x;
    if (patches) {
        for (int p = lane; p < HC_N; p += 64) px[p] = patches[pidx * HC_N + p];
    } else {
        int o = ids[3 * pidx], l = ids[3 * pidx + 1];
        o = o < 0 ? 0 : (o >= pt.n_octaves ? pt.n_octaves - 1 : o);
        l = l < 0 ? 0 : (l >= pt.n_levels ? pt.n_levels - 1 : l);
        const float* img = pt.lvl[o][l] + bi * pt.img_stride;
        const int h = pt.h[o], w = pt.w[o];
        const float* L = lafs + 6 * pidx;
        const float m = (float)(h < w ? h : w);
        const float t00 = L[0] * m, t01 = L[1] * m, t02 = L[2] * (float)w;
        const float t10 = L[3] * m, t11 = L[4] * m, t12 = L[5] * (float)h;
        for (int p = lane; p < HC_N; p += 64) {
            const int r = p / HC_PS, c = p - r * HC_PS;
            px[p] = aff_sample_bilinear(img, h, w, t00, t01, t02, t10, t11, t12, tb.base[c], tb.base[r]);
        }
    }
    __syncthreads();
    float sa = 0.f, sb = 0.f, sc = 0.f;
    for (int p = lane; p < HC_N; p += 64) {
        const int r = p / HC_PS, c = p - r * HC_PS;
        const int cm = c > 0 ? c - 1 : 0, cp = c < HC_PS - 1 ? c + 1 : HC_PS - 1;     // replicate padding
        const int rm = r > 0 ? r - 1 : 0, rp = r < HC_PS - 1 ? r + 1 : HC_PS - 1;
        const float xl = px[r * HC_PS + cm], xr = px[r * HC_PS + cp], yu = px[rm * HC_PS + c], yd = px[rp * HC_PS + c];
        if (KIND == 0) {
            const float gx = 0.5f * xl - 0.5f * xr, gy = 0.5f * yu - 0.5f * yd;       // taps [0.5, 0, -0.5]
            float mag = sqrtf((gx * gx + gy * gy) + 1e-10f);
            mag = mag * tb.gk[p];
            const float ori = atan2f(gy, gx);
            const float o_big = (36.0f * (ori + 3.14159274f)) / 6.28318548f;          // float(36) * (ori + pi) / (2 pi), fp32 scalars
            float b0 = floorf(o_big);
            const float w1 = o_big - b0;
            b0 = fmodf(b0, 36.0f);
            if (b0 < 0.0f) b0 += 36.0f;                                               // torch's % is a floored modulo
            bn[p] = (int)b0;
            wv[p] = (1.0f - w1) * mag;
        } else {
            const float gx = xr - xl, gy = yd - yu;                                   // taps [-1, 0, 1]
            const float g = tb.gk[p];
            sa += (gx * gx) * g; sb += (gx * gy) * g; sc += (gy * gy) * g;
        }
    }
    if (KIND == 0) {
        __syncthreads();
        if (lane < 36) {                     // deterministic: each bin is summed in pixel order by one lane
            float s = 0.f;
            for (int p = 0; p < HC_N; ++p) s += (bn[p] == lane) ? wv[p] : 0.0f;
            hist[lane + 1] = s / (float)HC_N;                                         // adaptive_avg_pool2d -> mean
        }
        if (lane == 36) { hist[0] = 0.f; hist[37] = 0.f; }                            // conv1d zero padding
        __syncthreads();
        float sm = -INFINITY;
        if (lane < 36) sm = fmaf(0.33f, hist[lane + 2], fmaf(0.34f, hist[lane + 1], 0.33f * hist[lane]));
        // argmax, first maximum wins
        float bv = sm;
        int bidx = lane;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bidx, o, 64);
            if (ov > bv || (ov == bv && oi < bidx)) { bv = ov; bidx = oi; }
        }
        if (lane == 0) {
            const float ang = -(((6.28318548f * (float)bidx) / 36.0f) - 3.14159274f);
            const float sn = sinf(ang), cs = cosf(ang);
            float* o = out + 4 * pidx;
            o[0] = cs; o[1] = sn; o[2] = -sn; o[3] = cs;
            if (out_angle) out_angle[pidx] = ang;
        }
    } else {
        sa = hc_wave_sum(sa); sb = hc_wave_sum(sb); sc = hc_wave_sum(sc);
        if (lane == 0) {
            const float a = sa / (float)HC_N, b = sb / (float)HC_N, c = sc / (float)HC_N;
            // invSqrt (HandCraftedModules.py:93-118), operation by operation
            const float mask = (b != 0.0f) ? 1.0f : 0.0f;
            const float r1 = (mask * (c - a)) / (2.0f * b + 1e-12f);
            const float sg = (r1 > 0.0f) ? 1.0f : ((r1 < 0.0f) ? -1.0f : 0.0f);
            const float t1 = sg / (fabsf(r1) + sqrtf(1.0f + r1 * r1));
            float r = 1.0f / sqrtf(1.0f + t1 * t1);
            float t = t1 * r;
            r = r * mask + 1.0f * (1.0f - mask);
            t = t * mask;
            float x = 1.0f / sqrtf(((r * r) * a - ((2.0f * r) * t) * b) + (t * t) * c);
            float z = 1.0f / sqrtf(((t * t) * a + ((2.0f * r) * t) * b) + (r * r) * c);
            const float d = sqrtf(x * z);
            x = x / d; z = z / d;
            const float na = (r * r) * x + (t * t) * z;
            const float nb = ((-r) * t) * x + (t * r) * z;
            const float nc = (t * t) * x + (r * r) * z;
            // abc2A (LAF.py:299-302) + rectifyAffineTransformationUpIsUp (:285-291)
            const float a00 = na, a01 = nb, a10 = nb, a11 = nc;
            const float det = sqrtf(fabsf(a00 * a11 - a10 * a01 + 1e-10f));
            const float b2a2 = sqrtf(a01 * a01 + a00 * a00);
            float* o = out + 4 * pidx;
            o[0] = b2a2 / det; o[1] = 0.0f * det;
            o[2] = (a11 * a01 + a10 * a00) / (b2a2 * det); o[3] = det / b2a2;
        }
    }
}

int aff_handcrafted_launch(affnet_ctx* ctx, int kind, const float* patches, const float* lafs, const int32_t* ids, const int32_t* count,
                           int n_max, const float* h_weights, float* out, float* out_angle, hipStream_t st) {
    if (kind < 0 || kind > 1) return aff_fail(ctx, AFFNET_ERR_INVALID, "handcrafted: unknown kind %d", kind);
    if (!h_weights || !out || n_max < 0 || (!patches && (!lafs || !ids))) return aff_fail(ctx, AFFNET_ERR_INVALID, "handcrafted: null argument");
    if (!patches && !ctx->ws) return aff_fail(ctx, AFFNET_ERR_INVALID, "handcrafted: sampling from the pyramid needs a bound workspace");
    if (n_max == 0) return AFFNET_OK;
    HcTables tb;
    memcpy(tb.gk, h_weights, sizeof(tb.gk));
    aff_base_grid(HC_PS, tb.base);
    PyrTable pt;
    if (!patches) aff_fill_pyr_table(ctx, &pt); else memset(&pt, 0, sizeof(pt));
    const dim3 grid(n_max, patches ? 1 : ctx->B);
    if (kind == 0) hipLaunchKernelGGL(hc19_kernel<0>, grid, dim3(64), 0, st, patches, pt, lafs, ids, count, n_max, tb, out, out_angle);
    else hipLaunchKernelGGL(hc19_kernel<1>, grid, dim3(64), 0, st, patches, pt, lafs, ids, count, n_max, tb, out, out_angle);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

extern "C" int affnet_handcrafted_forward(affnet_ctx* ctx, int kind, const float* d_patches, int n, const float* h_weights, float* d_out,
                                          float* d_angles, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !d_patches) return aff_fail(ctx, AFFNET_ERR_INVALID, "handcrafted_forward: null argument");
    return aff_handcrafted_launch(ctx, kind, d_patches, nullptr, nullptr, nullptr, n, h_weights, d_out, d_angles, (hipStream_t)stream);
}

extern "C" int affnet_handcrafted_forward_pyr(affnet_ctx* ctx, int kind, const float* d_lafs, const int32_t* d_ids, const int32_t* d_count,
                                              int n_max, const float* h_weights, float* d_out, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx) return AFFNET_ERR_INVALID;
    return aff_handcrafted_launch(ctx, kind, nullptr, d_lafs, d_ids, d_count, n_max, h_weights, d_out, nullptr, (hipStream_t)stream);
}
