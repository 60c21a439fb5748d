// Hessian scale-space detector for gfx950: response + 3-D NMS + octaveMap + centroid + top-k.
//
// Replaces SparseImgRepresenter.py:53-111,198 (multiScaleDetector, x mrSize),
// HandCraftedModules.py:58-78 (HessianResp), :208-291 (NMS3d, NMS3dAndComposeA),
// Utils.py:116-148 (grids, zero_response_at_border), LAF.py:431-441 (sc_y_x2LAFs).
//
// Structure (per image):
//   1. hessian_nms_kernel, one launch per octave: a workgroup stages a 64x16 tile (+2 halo) of
//      all five blurred levels in LDS, computes the five Hessian response tiles (+1 halo) into
//      LDS - they never touch HBM - runs the 3x3x3 NMS for the three detection levels and
//      appends every surviving maximum (with its 27-tap response-weighted centroid) to a
//      per-octave raw list.  The reference instead materialises 30 response maps, runs
//      max_pool3d, two whole-map 3->3 channel convolutions and a top-k over H*W per level.
//   2. level_resolve_kernel, two data-parallel passes per detection level: replays the reference's sequential
//      level loop on the sparse raw list: v = nms * (1 - float(octaveMap)), the `<= 1 positive`
//      skip rule, octaveMap = uint8(int64(float(octaveMap) + v)) (mod-256 wrap emulated,
//      HandCraftedModules.py:248-256) and emits accepted candidates (v != 0).
//   3. select_*: global top-C.  Taking top-C per level and then top-C of the union
//      (reference) selects the same set as one top-C over all candidates, so the per-level
//      top-k is skipped; a radix select finds the C-th largest response, a rank sort orders the
//      survivors (descending response, or (octave, level, pixel) order when nothing is cut).
//
// Arithmetic is the reference's fp32 sequence exactly (compiled with -ffp-contract=off):
// gxx = (l - 2c) + r, gxy from replicate-padded gx, |gxx*gyy - gxy*gxy| * float32(sigma^4),
// keep iff (c - max27) + 1e-5f > 0, centroid sums as fmaf chains in the order of the reference's CPU conv2d for that map size
// ((level, ky, kx) up to 6826 px, (ky, kx, level) above: see the centroid pass).
#include <math.h>
#include <stdlib.h>

#include "common.h"

#define HT_X 64
#define HT_Y 16
#define HX_W (HT_X + 4)   // blurred tile with 2-px halo
#define HX_H (HT_Y + 4)
#define HX_S 70           // LDS row stride of the blurred tile: thread t = 14 * row + segment reads X at 70 * row + 5 * segment + k, i.e. bank
                          // 5 * t + const (mod 32) - distinct over any 32 consecutive threads (stride 68 gave 2-way conflicts on the 105
                          // window reads per thread that dominate this kernel's LDS traffic: 45 % of its LDS cycles)
#define HR_W (HT_X + 2)   // response tile with 1-px halo
#define HR_H (HT_Y + 2)
#define HR_S (HR_W + 2)   // row stride 68 floats: rows 16-byte aligned, so that the NMS reads its 6-float windows as b128 + b64
#define HN_CAP 320        // per-workgroup staging capacity (1024 px x 3 levels; ~2% are maxima)

struct HessOct {           // one octave of the launch
    const float* levels;   // 5 blurred levels of this octave, contiguous
    RawMax* raw;
    int32_t* raw_cnt;
    int h, w, raw_cap;
    int centroid_order;        // summation order of the 27-tap centroid: 1 = (ky, kx, level) (oneDNN direct convolution), 0 = (level, ky, kx) (ATen im2col + sgemm); host decides
    int tiles_x, tile_begin;   // tiles per row; first flat tile index (blockIdx.x) of this octave
    float sigma[AFFNET_MAX_LEVELS];
    float sigma4[AFFNET_MAX_LEVELS];
};

// ONE launch covers every octave (blockIdx.x = flat tile index over all octaves, largest octave first; blockIdx.z = image): six
// launches per image were 85 us at one image per call, the small octaves (<= 40 tiles) each as long as the first (latency of one
// workgroup), and in batched calls their tails no longer idle the chip.
struct HessParams {
    HessOct oct[AFFNET_MAX_OCTAVES];
    int n_oct, n_levels;   // n_levels = levels_per_octave (5)
    int n_tiles, tiles_per_wg;   // tiles of all octaves; consecutive tiles one workgroup walks
    float th;
    int border;            // int(mrSize)
    int32_t* overflow;
    // batch (blockIdx.z = image): per-image strides of the pyramid (floats), the raw lists (entries), the counters
    size_t levels_stride, raw_stride;
    int precomputed;       // != 0: `levels` holds RESPONSE maps of a custom RespNet slot (same layout); only clamp(r - th, 0) applies
};

__device__ __forceinline__ float hessian_at(const float* __restrict__ X, int ty, int tx, float s4, float th) {
    // X: blurred tile, row stride HX_W; (ty,tx) centre in tile coordinates (>= 1 from each edge)
    const float c = X[ty * HX_W + tx];
    const float gxx = (X[ty * HX_W + tx - 1] - 2.0f * c) + X[ty * HX_W + tx + 1];
    const float gyy = (X[(ty - 1) * HX_W + tx] - 2.0f * c) + X[(ty + 1) * HX_W + tx];
    const float gx_up = 0.5f * X[(ty - 1) * HX_W + tx - 1] - 0.5f * X[(ty - 1) * HX_W + tx + 1];
    const float gx_dn = 0.5f * X[(ty + 1) * HX_W + tx - 1] - 0.5f * X[(ty + 1) * HX_W + tx + 1];
    const float gxy = 0.5f * gx_up - 0.5f * gx_dn;
    const float t1 = gxx * gyy;
    const float t2 = gxy * gxy;
    const float r = fabsf(t1 - t2) * s4;
    return fmaxf(r - th, 0.0f);
}

// Uniform (scalar) state of one tile: its octave's geometry, level pointers, raw list and sigma tables.
template <int NL>
struct HessTile {
    const float* levels; RawMax* raw; int32_t* raw_cnt;
    int h, w, raw_cap, x0, y0, centroid_order;
    float sigma[NL], sigma4[NL];
};

template <int NL>
__device__ __forceinline__ void hess_tile_setup(const HessParams& hp, int flat_tile, HessTile<NL>& t) {
    int oi = 0;
    while (oi + 1 < hp.n_oct && flat_tile >= hp.oct[oi + 1].tile_begin) ++oi;      // uniform: scalar loads from the kernel arguments
    t.h = hp.oct[oi].h; t.w = hp.oct[oi].w; t.raw_cap = hp.oct[oi].raw_cap; t.centroid_order = hp.oct[oi].centroid_order;
#pragma unroll
    for (int l = 0; l < NL; ++l) { t.sigma[l] = hp.oct[oi].sigma[l]; t.sigma4[l] = hp.oct[oi].sigma4[l]; }
    const int tile = flat_tile - hp.oct[oi].tile_begin, tiles_x = hp.oct[oi].tiles_x;
    t.x0 = (tile % tiles_x) * HT_X; t.y0 = (tile / tiles_x) * HT_Y;
    t.levels = hp.oct[oi].levels + blockIdx.z * hp.levels_stride;
    t.raw = hp.oct[oi].raw + blockIdx.z * hp.raw_stride;
    t.raw_cnt = hp.oct[oi].raw_cnt + blockIdx.z * CNT_TOTAL;
}

#define HESS_NLD ((HX_H * HX_W + 255) / 256)
// All NL x 6 loads of a thread are issued before the first one is consumed (the element -> pixel mapping is the same for every
// level).  Written as load-then-store per element the compiler put s_waitcnt vmcnt(0) behind each load: 30 serialized HBM round
// trips per thread, ~29 us per workgroup, 0.55 TB/s for the whole kernel.
template <int NL>
__device__ __forceinline__ void hess_tile_load(const HessTile<NL>& t, float (&tmp)[NL][HESS_NLD]) {
    int goff[HESS_NLD];
#pragma unroll
    for (int k = 0; k < HESS_NLD; ++k) {
        const int i = threadIdx.x + 256 * k;
        const int ty = i / HX_W, tx = i - ty * HX_W;
        int gy = t.y0 + ty - 2, gx = t.x0 + tx - 2;
        gy = gy < 0 ? 0 : (gy >= t.h ? t.h - 1 : gy);   // replicate padding of the Hessian filters
        gx = gx < 0 ? 0 : (gx >= t.w ? t.w - 1 : gx);
        goff[k] = (i < HX_H * HX_W) ? gy * t.w + gx : 0;
    }
    const size_t lvl_stride = (size_t)t.h * t.w;
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int k = 0; k < HESS_NLD; ++k) tmp[l][k] = t.levels[l * lvl_stride + goff[k]];
}

// NL = levels per octave = nLevels + 2 (5 for the reference's default nlevels = 3; 3..8 are instantiated).
// A workgroup walks hp.tiles_per_wg consecutive tiles and requests the blurred levels of the NEXT tile (registers) right after the
// current one's are in LDS, so the HBM round trip runs under the response / NMS / centroid work instead of in front of it
// (PMC, 4K batch: SQ_WAIT_ANY 45 % of the wave cycles with one tile per workgroup and 3 workgroups per CU, 0.9 TB/s).
template <int NL>
__global__ __launch_bounds__(256, (NL <= 5 ? 3 : 1)) void hessian_nms_kernel(HessParams hp) {
    // LDS: NL blurred tiles (20x68) + NL response tiles (18x67)
    __shared__ __attribute__((aligned(16))) float X[NL][HX_H * HX_S];
    __shared__ __attribute__((aligned(16))) float Rr[NL][HR_H * HR_S];
    // Maxima found by this workgroup are staged and appended with ONE global atomic (a single contended counter retires
    // only ~90 atomics/us: per-candidate atomics cost 150 us on octave 0).  The staging list aliases the blurred tiles,
    // which are dead once the responses are in Rr: 51 KB of LDS -> 3 workgroups per CU instead of 2.
    RawMax* s_list = reinterpret_cast<RawMax*>(&X[0][0]);
    static_assert(sizeof(RawMax) * HN_CAP <= sizeof(float) * NL * HX_H * HX_S, "staging list must fit in the tile area");
    // Maxima are sparse (~2 % of the pixels per level): the test runs on every pixel, but the 27-tap centroid only on the hits, which
    // are first QUEUED in LDS (behind the staging list) and then worked off one per thread.  Computed inside the test loop, every
    // wavefront with a single hit among its 64 lanes x 4 pixels x (NL - 2) levels walked the whole centroid code for it.
    // Queue entry: level << 10 | row << 6 | column, and from bit 16 on the mask of the LOWER detection levels at which the same pixel is
    // a maximum too (bit l' - 1): their responses go into RawMax::prev for the octaveMap replay.
    uint32_t* s_queue = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(&X[0][0]) + sizeof(RawMax) * HN_CAP);
    static_assert(sizeof(RawMax) * HN_CAP + 4 * HT_X * HT_Y * (NL - 2) <= sizeof(float) * NL * HX_H * HX_S, "staging list + queue must fit in the tile area");
    __shared__ int s_n, s_base;
    // sigma table of the current tile's octave for the centroid pass, whose level index is a run-time value (queue entry): read from
    // LDS.  (Indexing the tile's scalar state with it - also through a chain of selects, which the compiler folds back into an indexed
    // load - put the whole HessTile into scratch memory: 21 dwords stored per thread per tile, 24 MB of HBM writes per 1024x768 image.)
    __shared__ float s_sigma[AFFNET_MAX_LEVELS];
    const int first = blockIdx.x * hp.tiles_per_wg;
    const int last = min(first + hp.tiles_per_wg, hp.n_tiles);
    int32_t* const overflow = hp.overflow + blockIdx.z * CNT_TOTAL;
    float tmp[NL][HESS_NLD];
    if (!hp.precomputed) {
        HessTile<NL> nxt;
        hess_tile_setup<NL>(hp, first, nxt);
        hess_tile_load<NL>(nxt, tmp);
    }
    for (int ft = first; ft < last; ++ft) {
        HessTile<NL> p;                                 // this tile (uniform: scalar registers)
        hess_tile_setup<NL>(hp, ft, p);
        if (ft > first) __syncthreads();               // the previous tile's staging list / queue (alias X) and Rr are consumed
        if (threadIdx.x == 0) {
            s_n = 0;
#pragma unroll
            for (int l = 0; l < NL; ++l) s_sigma[l] = p.sigma[l];
        }
        if (!hp.precomputed) {
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
                for (int k = 0; k < HESS_NLD; ++k) {
                    const int i = threadIdx.x + 256 * k;
                    if (i < HX_H * HX_W) X[l][(i / HX_W) * HX_S + (i % HX_W)] = tmp[l][k];
                }
        }
        __syncthreads();
        if (ft + 1 < last && !hp.precomputed) {
            HessTile<NL> nxt;                           // only its geometry and level pointer live on, inside the addresses of the loads
            hess_tile_setup<NL>(hp, ft + 1, nxt);
            hess_tile_load<NL>(nxt, tmp);               // in flight until the next iteration's LDS stores
        }
        const int h = p.h, w = p.w, x0 = p.x0, y0 = p.y0;
        const size_t lvl_stride = (size_t)h * w;
        // Responses: one thread = 5 consecutive pixels of one response row (18 rows x 14 segments = 252 threads), all NL levels.
        // The 3 x 7 window of the blurred tile is read once into registers (21 LDS reads for 5 responses instead of 45); every
        // response is the same expression tree as hessian_at().
        if (hp.precomputed) {
            for (int l = 0; l < NL; ++l)
                for (int i = threadIdx.x; i < HR_H * HR_W; i += 256) {
                    const int ry = i / HR_W, rx = i - ry * HR_W;
                    const int gy = y0 + ry - 1, gx = x0 + rx - 1;
                    float r = -INFINITY;                   // outside the image: -inf for max_pool3d padding
                    if (gy >= 0 && gy < h && gx >= 0 && gx < w) r = fmaxf(p.levels[l * lvl_stride + (size_t)gy * w + gx] - hp.th, 0.0f);   // SparseImgRepresenter.py:77
                    Rr[l][ry * HR_S + rx] = r;
                }
        } else if (threadIdx.x < HR_H * 14) {
            const int ry = threadIdx.x / 14, rx0 = (threadIdx.x - ry * 14) * 5;
            const int gy = y0 + ry - 1;
            const bool row_in = gy >= 0 && gy < h;
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const float s4 = p.sigma4[l];
                const float* xr = &X[l][ry * HX_S + rx0];  // top-left of the window: response (ry, rx) is centred on X (ry + 1, rx + 1)
                float u[7], c[7], d[7];
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    const bool ok = rx0 + k < HX_W;
                    u[k] = ok ? xr[k] : 0.0f; c[k] = ok ? xr[HX_S + k] : 0.0f; d[k] = ok ? xr[2 * HX_S + k] : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const int rx = rx0 + k;
                    if (rx >= HR_W) break;
                    const int gx = x0 + rx - 1;
                    const float cc = c[k + 1];
                    const float gxx = (c[k] - 2.0f * cc) + c[k + 2];
                    const float gyy = (u[k + 1] - 2.0f * cc) + d[k + 1];
                    const float gx_up = 0.5f * u[k] - 0.5f * u[k + 2];
                    const float gx_dn = 0.5f * d[k] - 0.5f * d[k + 2];
                    const float gxy = 0.5f * gx_up - 0.5f * gx_dn;
                    const float t1 = gxx * gyy;
                    const float t2 = gxy * gxy;
                    const float r = fmaxf(fabsf(t1 - t2) * s4 - hp.th, 0.0f);
                    Rr[l][ry * HR_S + rx] = (row_in && gx >= 0 && gx < w) ? r : -INFINITY;   // outside the image: -inf (max_pool3d padding)
                }
            }
        }
        __syncthreads();
        {
            const bool border_ok = (hp.border < w) && (hp.border < h);
            // NMS: each thread owns 4 pixels of one row.  Per level the 3 x 6 response window is read once; the three-row column
            // maxima are shared by the 4 pixels and by the NL - 2 detection levels (max is order-independent: same values).
            int tn = threadIdx.x;                 // opaque per tile like tq below: the 12 queue codes (l << 10 | ty << 6 | tx) are loop invariants too
            asm volatile("" : "+v"(tn));
            const int ty = tn >> 4, txb = (tn & 15) * 4;
            const int gy = y0 + ty;
            // m5[l][q]: 3 x 3 maximum around pixel q of level l.  (Kept per level as the six column maxima cm[l][0..5] and reduced per
            // pixel later, the 30 + 20 live registers of this pass put the NL = 5 instantiation 5 VGPRs over the 168 that three
            // workgroups per CU allow: 5 spilled registers, 24 B of scratch per thread.  max is order-independent: same values.)
            float m5[NL][4], ctr[NL][4];
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                // top-left of the window of pixel txb (response tile has a 1-px halo); txb is a multiple of 4 and the rows are 16-byte
                // aligned: one 16-byte + one 8-byte read per row (conflict-free: the 16 lanes of a row read 64 consecutive floats)
                // instead of six 4-byte reads at a 4-float lane stride (8 of 32 banks: 2..4-way conflicts, 37 % of the LDS cycles)
                const float* r = &Rr[l][ty * HR_S + txb];
                float win[3][6];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const float4 q4 = *reinterpret_cast<const float4*>(r + dy * HR_S);
                    const float2 q2 = *reinterpret_cast<const float2*>(r + dy * HR_S + 4);
                    win[dy][0] = q4.x; win[dy][1] = q4.y; win[dy][2] = q4.z; win[dy][3] = q4.w; win[dy][4] = q2.x; win[dy][5] = q2.y;
                }
                float cm[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    cm[k] = fmaxf(fmaxf(win[0][k], win[1][k]), win[2][k]);
                    if (k >= 1 && k <= 4) ctr[l][k - 1] = win[1][k];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) m5[l][q] = fmaxf(fmaxf(cm[q], cm[q + 1]), cm[q + 2]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int tx = txb + q, gx = x0 + tx;
                if (gx >= w || gy >= h) break;
                const bool in_border = !border_ok || gy < hp.border || gy >= h - hp.border || gx < hp.border || gx >= w - hp.border;
                if (in_border) continue;                      // zero_response_at_border -> nms value 0 -> never a candidate
                unsigned hits = 0;
#pragma unroll
                for (int l = 1; l <= NL - 2; ++l) {
                    const float c = ctr[l][q];
                    const float M = fmaxf(fmaxf(m5[l - 1][q], m5[l][q]), m5[l + 1][q]);
                    const float d = c - M;
                    const float e = d + 1e-5f;
                    if (!(e > 0.0f) || c == 0.0f) continue;    // keep * x == 0 -> contributes nothing anywhere
                    s_queue[atomicAdd(&s_n, 1)] = (hits << 16) | (unsigned)((l << 10) | (ty << 6) | tx);
                    hits |= 1u << (l - 1);
                }
            }
        }
        __syncthreads();
        const int n_q = s_n;
        if (n_q == 0) continue;                               // uniform
        // The thread index of the centroid / copy-out part is made opaque per tile: derived from threadIdx.x, its loop-invariant addresses
        // (queue slots, staging-list rows at 1 KB steps) were hoisted out of the tile loop and held in registers across the whole tile -
        // what pushed the NL = 5 instantiation 5 VGPRs over the 168 of three workgroups per CU (5 spills, 24 B of scratch per thread).
        int tq = threadIdx.x;
        asm volatile("" : "+v"(tq));
        for (int e = tq; e < n_q; e += 256) {
            const unsigned code = s_queue[e];
            const int l = (code >> 10) & 63, qy = (code >> 6) & 15, qx = code & 63;
            const unsigned lower = code >> 16;
            const int py = y0 + qy, px = x0 + qx;
            const float c = Rr[l][(qy + 1) * HR_S + qx + 1];
            // 27-tap centroid on the UNMASKED responses, zero padding (HandCraftedModules.py:279): two conv2d calls of a (1, 3, h, w) tensor with
            // 3 x 3 x 3 weights on the reference's CPU.  Their fp32 summation order depends on the map size - ATen's use_mkldnn() sends a
            // batch-1 3 x 3 convolution to oneDNN only when the input has more than 20480 elements (Convolution.cpp: "for some case, native is
            // faster"): the native path (im2col + sgemm, K = 27) accumulates in (level, ky, kx) order, oneDNN's direct convolution for 3 input
            // channels in (ky, kx, level) order, both as fmaf chains (verified bit for bit on 300 x 500 / 40 x 50 maps ON THE AUTHORING HOST = the host of
            // the golden vectors, tools/probes/cpu_conv_order.py; the small-map branch is host-specific: MKL's sgemm on the GPU box's EPYC sums
            // differently, profiles/archive/r05_s1_cpu_conv_order_gpubox.txt).  The predicate is evaluated on the host in 64 bits (detect_candidates:
            // HessOct::centroid_order; AFFNET_CENTROID_ORDER=onednn|native pins one order for a reference torch build that dispatches differently).
            // A one-ulp difference of a sub-pixel centre moves the sampled patch enough to shift a sensitive frame by 1e-3 px (round 5: the
            // float64 referee traced every LAF row outside 1e-3 px to this), so both orders are reproduced.
            float ns = 0.f, ny = 0.f, nx = 0.f, den = 0.f;
#define AFF_CENTROID_TAP(dl, ky, kx)                                                        \
            {                                                                               \
                const float sg = s_sigma[l - 1 + (dl)];                                     \
                const float oy = ((ky) == 0) ? -0.5f : ((ky) == 1 ? 0.5f : 1.5f);           \
                const float ox = ((kx) == 0) ? -0.5f : ((kx) == 1 ? 0.5f : 1.5f);           \
                float r = Rr[l - 1 + (dl)][(qy + (ky)) * HR_S + qx + (kx)];                 \
                if (r == -INFINITY) r = 0.0f; /* conv2d zero padding */                     \
                ns = fmaf(r, sg, ns);                                                       \
                ny = fmaf(r, oy, ny);                                                       \
                nx = fmaf(r, ox, nx);                                                       \
                den = fmaf(r, 1.0f, den);                                                   \
            }
            if (p.centroid_order) {                    // oneDNN: (ky, kx, level)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int dl = 0; dl < 3; ++dl) AFF_CENTROID_TAP(dl, ky, kx)
            } else {                                   // native im2col + sgemm: (level, ky, kx)
#pragma unroll
                for (int dl = 0; dl < 3; ++dl)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) AFF_CENTROID_TAP(dl, ky, kx)
            }
#undef AFF_CENTROID_TAP
            const float dd = den + 1e-8f;
            float cs = ns / dd, cy = ny / dd, cx = nx / dd;
            cy = cy + (float)py;
            cx = cx + (float)px;
            const float msz = (float)(h < w ? h : w);
            RawMax rm;
            rm.pix = py * w + px;
            rm.lvl = l;
            rm.val = c;
            rm.s = cs / msz;
            rm.y = cy / (float)h;
            rm.x = cx / (float)w;
            rm.prev[0] = (l >= 2 && ((lower >> (l - 2)) & 1u)) ? Rr[l - 1][(qy + 1) * HR_S + qx + 1] : 0.0f;
            rm.prev[1] = (l >= 3 && ((lower >> (l - 3)) & 1u)) ? Rr[l - 2][(qy + 1) * HR_S + qx + 1] : 0.0f;
            if (e < HN_CAP) {
                s_list[e] = rm;
            } else {                                   // staging full (pathological tile): direct append
                const int slot = atomicAdd(p.raw_cnt, 1);
                if (slot < p.raw_cap) p.raw[slot] = rm;
                else atomicOr(overflow, 1);
            }
        }
        __syncthreads();
        const int n_loc = n_q < HN_CAP ? n_q : HN_CAP;
        if (tq == 0) s_base = atomicAdd(p.raw_cnt, n_loc);
        __syncthreads();
        const int base = s_base;
        for (int i = tq; i < n_loc; i += 256) {
            if (base + i < p.raw_cap) p.raw[base + i] = s_list[i];
            else atomicOr(overflow, 1);
        }
    }
}

// ---- standalone Hessian response (RespNet slot / tests) -------------------------------------------
__global__ __launch_bounds__(256) void hessian_resp_kernel(const float* __restrict__ in, float* __restrict__ out, int h, int w,
                                                          float s4) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int xm = x > 0 ? x - 1 : 0, xp = x < w - 1 ? x + 1 : w - 1;
    const int ym = y > 0 ? y - 1 : 0, yp = y < h - 1 ? y + 1 : h - 1;
    const float c = in[(size_t)y * w + x];
    const float gxx = (in[(size_t)y * w + xm] - 2.0f * c) + in[(size_t)y * w + xp];
    const float gyy = (in[(size_t)ym * w + x] - 2.0f * c) + in[(size_t)yp * w + x];
    const float gu = 0.5f * in[(size_t)ym * w + xm] - 0.5f * in[(size_t)ym * w + xp];
    const float gd = 0.5f * in[(size_t)yp * w + xm] - 0.5f * in[(size_t)yp * w + xp];
    const float gxy = 0.5f * gu - 0.5f * gd;
    const float t1 = gxx * gyy, t2 = gxy * gxy;
    out[(size_t)y * w + x] = fabsf(t1 - t2) * s4;
}

extern "C" int affnet_hessian_response(affnet_ctx* ctx, const float* d_in, float* d_out, int h, int w, float sigma4, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !d_in || !d_out || h < 1 || w < 1) return aff_fail(ctx, AFFNET_ERR_INVALID, "hessian_response: bad argument");
    hipLaunchKernelGGL(hessian_resp_kernel, dim3(aff_cdiv(w, 64), aff_cdiv(h, 4)), dim3(256), 0, (hipStream_t)stream, d_in, d_out, h,
                       w, sigma4);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

// ---- sequential octaveMap replay on the sparse raw list -------------------------------------------
struct ResolveParams {
    RawMax* raw[AFFNET_MAX_OCTAVES];
    uint8_t* omap[AFFNET_MAX_OCTAVES];
    int raw_cap[AFFNET_MAX_OCTAVES];
    int n_detect_levels;          // nLevels (3)
    int32_t* cnt;                 // counter block
    float* cand_resp; float* cand_syx; int32_t* cand_ids;
    int cand_cap;
    size_t raw_stride, map_stride;   // batch (blockIdx.y = image)
};

// The reference's level loop is sequential only ACROSS levels (level l sees the octaveMap written by level l-1); inside a
// level every maximum is independent except for the `<= 1 positive -> skip the level` rule, which needs the level's count
// first.  So each detection level is two data-parallel passes over the raw lists of all octaves and images:
//   mode 0 (count): v = nms * (1 - float(octaveMap)) > 0  -> per-(octave, level) counter
//   mode 1 (apply): if the count is > 1: octaveMap = uint8(int64(float(octaveMap) + v)), emit the candidate (v != 0)
// (the former one-workgroup-per-octave replay was a chain of dependent global round trips: 0.25 ms per 16 images at
// 1024x768, 1.3 ms per 4 images at 4K).
__global__ __launch_bounds__(256) void level_resolve_kernel(ResolveParams p, int l, int mode) {
    const int o = blockIdx.y;
    const size_t img = blockIdx.z;
    int32_t* cnt = p.cnt + img * CNT_TOTAL;
    int n = cnt[CNT_RAW0 + o];
    if (n > p.raw_cap[o]) n = p.raw_cap[o];
    const int lane = threadIdx.x & 63;
    __shared__ int s_wcnt[4], s_wbase;
    int32_t* pos = cnt + CNT_POS0 + (l - 1) * AFFNET_MAX_OCTAVES + o;
    if (mode == 1 && *pos <= 1) return;                 // HandCraftedModules.py:252-254: level skipped, octaveMap unchanged
    uint8_t* omap = p.omap[o] + img * p.map_stride;
    const RawMax* raw = p.raw[o] + img * p.raw_stride;
    int local = 0;
    for (int i0 = blockIdx.x * 256; i0 < n; i0 += gridDim.x * 256) {     // grid-stride: uniform trip count per workgroup
        const int i = i0 + threadIdx.x;
        bool mine = i < n;
        RawMax r;
        if (mine) r = raw[i];
        mine = mine && r.lvl == l;
        const float mval = mine ? (float)omap[r.pix] : 0.0f;
        const float v = mine ? r.val * (1.0f - mval) : 0.0f;
        if (mode == 0) {
            local += (mine && v > 0.0f) ? 1 : 0;
            continue;
        }
        bool emit = false;
        if (mine) {
            const float sum = mval + v;
            omap[r.pix] = (uint8_t)(long long)sum;     // float -> int64 -> uint8 wrap, as torch's CPU .byte()
            emit = v != 0.0f;
        }
        // one global atomic per WORKGROUP iteration (the image's candidate counter is a single address: ~90 atomics / us; one per
        // wavefront made the apply passes of a 4K batch atomic-bound)
        const unsigned long long bal = __ballot(emit);
        if (lane == 0) s_wcnt[threadIdx.x >> 6] = __popcll(bal);
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tot = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
            s_wbase = tot ? atomicAdd(&cnt[CNT_CAND], tot) : 0;
        }
        __syncthreads();
        int wbase = s_wbase;
        for (int wv = 0; wv < (int)(threadIdx.x >> 6); ++wv) wbase += s_wcnt[wv];
        __syncthreads();                                // s_wcnt / s_wbase are rewritten by the next iteration
        if (emit) {
            const int slot = wbase + __popcll(bal & ((1ull << lane) - 1ull));
            if (slot < p.cand_cap) {
                float* cr = p.cand_resp + img * p.cand_cap;
                float* cs = p.cand_syx + img * p.cand_cap * 3;
                int32_t* ci = p.cand_ids + img * p.cand_cap * 3;
                cr[slot] = v;
                cs[3 * slot] = r.s; cs[3 * slot + 1] = r.y; cs[3 * slot + 2] = r.x;
                ci[3 * slot] = o; ci[3 * slot + 1] = l - 1; ci[3 * slot + 2] = r.pix;
            } else {
                atomicOr(&cnt[CNT_OVERFLOW], 2);
            }
        }
    }
    if (mode == 0) {
        const unsigned long long bal = __ballot(local > 0);
        if (bal) {
#pragma unroll
            for (int ofs = 32; ofs > 0; ofs >>= 1) local += __shfl_xor(local, ofs, 64);
            if (lane == 0) atomicAdd(pos, local);
        }
    }
}

__device__ __forceinline__ uint32_t order_key(float f) {   // larger float -> larger uint
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---- the same replay in TWO launches (<= 3 detection levels) --------------------------------------
// The octaveMap of a pixel changes only through that pixel's own maxima, and a raw maximum carries the responses of the same pixel at
// the lower detection levels (RawMax::prev).  So its masked response v is a function of (val, prev, which lower levels were applied),
// and the only global facts are the per-level skip decisions "<= 1 positive -> level skipped", each depending on the lower levels'
// decisions.  resolve_count_kernel counts the positives of every level under EVERY combination of lower-level decisions (1 + 2 + 4
// counters per octave); resolve_apply_kernel derives the actual decisions from them and emits the candidates.  The octaveMap itself is
// never materialised (nor cleared: 11 MB per 4K image).  Five dependent launches and five passes over the raw lists become two.
__device__ __forceinline__ float omap_step(float m, float pv) {             // octaveMap value of the pixel after a level with NMS'ed value pv
    const float v = pv * (1.0f - m);
    return (float)(uint8_t)(long long)(m + v);                                // float -> int64 -> uint8 wrap, as torch's CPU .byte()
}
// masked response of a raw maximum given which lower levels were applied (a1: level 1, a2: level 2)
__device__ __forceinline__ float masked_value(const RawMax& r, bool a1, bool a2) {
    float m = 0.0f;
    if (r.lvl == 2) { if (a1 && r.prev[0] != 0.0f) m = omap_step(m, r.prev[0]); }
    else if (r.lvl == 3) {
        if (a1 && r.prev[1] != 0.0f) m = omap_step(m, r.prev[1]);            // level 1 first, then level 2: the reference's order
        if (a2 && r.prev[0] != 0.0f) m = omap_step(m, r.prev[0]);
    }
    return r.val * (1.0f - m);
}

__global__ __launch_bounds__(256) void resolve_count_kernel(ResolveParams p) {
    const int o = blockIdx.y;
    const size_t img = blockIdx.z;
    int32_t* cnt = p.cnt + img * CNT_TOTAL;
    int n = cnt[CNT_RAW0 + o];
    if (n > p.raw_cap[o]) n = p.raw_cap[o];
    if ((int)blockIdx.x * 256 >= n) return;
    const RawMax* raw = p.raw[o] + img * p.raw_stride;
    int c[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const RawMax r = raw[i];
        if (r.lvl == 1) c[0] += r.val > 0.0f;
        else if (r.lvl == 2) { c[1] += masked_value(r, false, false) > 0.0f; c[2] += masked_value(r, true, false) > 0.0f; }
        else {
#pragma unroll
            for (int h = 0; h < 4; ++h) c[3 + h] += masked_value(r, h & 1, h >> 1) > 0.0f;
        }
    }
    __shared__ int s_c[7];
    if (threadIdx.x < 7) s_c[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        int v = c[k];
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) v += __shfl_xor(v, ofs, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_c[k], v);
    }
    __syncthreads();
    if (threadIdx.x < 7 && s_c[threadIdx.x]) atomicAdd(&cnt[CNT_HYP0 + 8 * o + threadIdx.x], s_c[threadIdx.x]);
}

__global__ __launch_bounds__(256) void resolve_apply_kernel(ResolveParams p, uint32_t* __restrict__ ghist) {
    const int o = blockIdx.y;
    const size_t img = blockIdx.z;
    int32_t* cnt = p.cnt + img * CNT_TOTAL;
    int n = cnt[CNT_RAW0 + o];
    if (n > p.raw_cap[o]) n = p.raw_cap[o];
    // the reference's sequential decisions (HandCraftedModules.py:252-254: a level with <= 1 positive is skipped, octaveMap unchanged)
    const int32_t* hy = cnt + CNT_HYP0 + 8 * o;
    const int n1 = hy[0];
    const bool a1 = n1 > 1;
    const int n2 = hy[1 + (a1 ? 1 : 0)];
    const bool a2 = n2 > 1;
    const int n3 = hy[3 + (a1 ? 1 : 0) + (a2 ? 2 : 0)];
    const bool a3 = n3 > 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {                                // per-level positives, as the multi-pass replay records them
        cnt[CNT_POS0 + 0 * AFFNET_MAX_OCTAVES + o] = n1;
        if (p.n_detect_levels > 1) cnt[CNT_POS0 + 1 * AFFNET_MAX_OCTAVES + o] = n2;
        if (p.n_detect_levels > 2) cnt[CNT_POS0 + 2 * AFFNET_MAX_OCTAVES + o] = n3;
    }
    if ((int)blockIdx.x * 256 >= n) return;
    const RawMax* raw = p.raw[o] + img * p.raw_stride;
    const int lane = threadIdx.x & 63;
    __shared__ int s_wcnt[4], s_wbase;
    __shared__ uint32_t s_hist[SEL_HIST_BINS];
    if (ghist) {
        for (int i = threadIdx.x; i < SEL_HIST_BINS; i += 256) s_hist[i] = 0;
        ghist += img * SEL_HIST_BINS;
    }
    __syncthreads();
    for (int i0 = blockIdx.x * 256; i0 < n; i0 += gridDim.x * 256) {          // grid-stride: uniform trip count per workgroup
        const int i = i0 + threadIdx.x;
        RawMax r;
        r.lvl = 0; r.val = 0.f; r.prev[0] = r.prev[1] = 0.f; r.pix = 0; r.s = r.y = r.x = 0.f;
        if (i < n) r = raw[i];
        const bool applied = r.lvl == 1 ? a1 : (r.lvl == 2 ? a2 : (r.lvl == 3 ? a3 : false));
        const float v = applied ? masked_value(r, a1, a2) : 0.0f;
        const bool emit = applied && v != 0.0f;
        // one global atomic per WORKGROUP iteration (the image's candidate counter is a single address: ~90 atomics / us)
        const unsigned long long bal = __ballot(emit);
        if (lane == 0) s_wcnt[threadIdx.x >> 6] = __popcll(bal);
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tot = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
            s_wbase = tot ? atomicAdd(&cnt[CNT_CAND], tot) : 0;
        }
        __syncthreads();
        int wbase = s_wbase;
        for (int wv = 0; wv < (int)(threadIdx.x >> 6); ++wv) wbase += s_wcnt[wv];
        __syncthreads();                                // s_wcnt / s_wbase are rewritten by the next iteration
        if (emit) {
            const int slot = wbase + __popcll(bal & ((1ull << lane) - 1ull));
            if (slot < p.cand_cap) {
                float* cr = p.cand_resp + img * p.cand_cap;
                float* cs = p.cand_syx + img * p.cand_cap * 3;
                int32_t* ci = p.cand_ids + img * p.cand_cap * 3;
                cr[slot] = v;
                cs[3 * slot] = r.s; cs[3 * slot + 1] = r.y; cs[3 * slot + 2] = r.x;
                ci[3 * slot] = o; ci[3 * slot + 1] = r.lvl - 1; ci[3 * slot + 2] = r.pix;
                if (ghist) atomicAdd(&s_hist[order_key(v) >> 21], 1u);
            } else {
                atomicOr(&cnt[CNT_OVERFLOW], 2);
            }
        }
    }
    if (ghist) {
        __syncthreads();
        for (int i = threadIdx.x; i < SEL_HIST_BINS; i += 256)
            if (s_hist[i]) atomicAdd(&ghist[i], s_hist[i]);
    }
}

// ---- global top-C --------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long ord_key(const int32_t* ids) {  // (octave, level, pixel) lexicographic
    return ((unsigned long long)(uint32_t)ids[0] << 40) | ((unsigned long long)(uint32_t)ids[1] << 32) | (uint32_t)ids[2];
}

// Radix select of the C-th largest response, digits of 11 / 11 / 10 bits.
//   select_hist_kernel    many workgroups per image: LDS histogram of the FIRST digit of every candidate, non-empty bins added to
//                         the image's global histogram (sel_hist, zeroed with the counters);
//   select_prepare_kernel one workgroup per image: picks the first digit from that histogram without touching the candidates,
//                         then ONE pass over them collects the keys of that bucket in LDS (when they fit) and the second and third
//                         digit are decided from there.
// (The former single-workgroup select made four passes over the candidates with an LDS atomic per key and let thread 0 walk the
// 256 bins after each: 35 us for 10^4 candidates, 250 us for the 8 x 2.5 * 10^5 of a 4K batch - on 8 of 256 CUs.)
__global__ __launch_bounds__(256) void select_hist_kernel(const float* __restrict__ resp, const int32_t* __restrict__ cnt, int cand_cap, int C,
                                                          uint32_t* __restrict__ ghist) {
    __shared__ uint32_t hist[SEL_HIST_BINS];
    resp += (size_t)blockIdx.y * cand_cap;
    cnt += blockIdx.y * CNT_TOTAL;
    ghist += (size_t)blockIdx.y * SEL_HIST_BINS;
    int n = cnt[CNT_CAND];
    if (n > cand_cap) n = cand_cap;
    if (!(C > 0 && n > C) || (int)blockIdx.x * 1024 >= n) return;          // keep-all mode needs no threshold; no work for this workgroup
    for (int i = threadIdx.x; i < SEL_HIST_BINS; i += 256) hist[i] = 0;
    __syncthreads();
    for (int i0 = blockIdx.x * 1024; i0 < n; i0 += gridDim.x * 1024) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * 256 + threadIdx.x; v[u] = i < n ? resp[i] : 0.0f; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * 256 + (int)threadIdx.x < n) atomicAdd(&hist[order_key(v[u]) >> 21], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SEL_HIST_BINS; i += 256)
        if (hist[i]) atomicAdd(&ghist[i], hist[i]);
}

// Descending walk over hist[nbins] (nbins = 1024 or 2048, in LDS) by the whole 1024-thread workgroup: finds the bin d with
// sum(hist[d+1 ..]) < need <= sum(hist[d ..]) and writes out[0] = d, out[1] = need - sum(hist[d+1 ..]) (1-based rank inside bin d).
// The caller guarantees sum(hist) >= need >= 1.  Ends with a barrier.
__device__ __forceinline__ void radix_pick(const uint32_t* hist, int nbins, uint32_t need, uint32_t* s_wsum, uint32_t* out) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = nbins >> 10;
    const int b0 = nbins - 1 - t * per;
    const uint32_t c0 = hist[b0], c1 = per == 2 ? hist[b0 - 1] : 0u;
    const uint32_t mine = c0 + c1;
    uint32_t inc = mine;
#pragma unroll
    for (int ofs = 1; ofs < 64; ofs <<= 1) {
        const uint32_t v = __shfl_up(inc, ofs, 64);
        if (lane >= ofs) inc += v;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    uint32_t before = inc - mine;
    for (int w = 0; w < wave; ++w) before += s_wsum[w];
    if (before < need && need <= before + mine) {                          // exactly one thread
        if (need <= before + c0) { out[0] = (uint32_t)b0; out[1] = need - before; }
        else { out[0] = (uint32_t)(b0 - 1); out[1] = need - before - c0; }
    }
    __syncthreads();
}

#define SEL_LIST_CAP 8192       // keys of the first digit's bucket kept in LDS (32 KB); larger buckets are re-read from global memory
__global__ __launch_bounds__(1024) void select_prepare_kernel(const float* __restrict__ resp, const int32_t* __restrict__ ids, int32_t* cnt, int cand_cap,
                                                              int C, int sel_cap, const uint32_t* __restrict__ ghist) {
    __shared__ uint32_t hist[SEL_HIST_BINS];
    __shared__ uint32_t list[SEL_LIST_CAP];
    __shared__ uint32_t s_wsum[16], s_pick[2], s_ln;
    resp += (size_t)blockIdx.x * cand_cap;           // blockIdx.x = image
    ids += (size_t)blockIdx.x * cand_cap * 3;
    cnt += blockIdx.x * CNT_TOTAL;
    ghist += (size_t)blockIdx.x * SEL_HIST_BINS;
    int n = cnt[CNT_CAND];
    if (n > cand_cap) n = cand_cap;
    if (threadIdx.x == 0) {
        cnt[CNT_SEL] = 0; cnt[CNT_EQ_TAKEN] = 0;
    }
    if (!(C > 0 && n > C)) {
        if (threadIdx.x == 0) {
            cnt[CNT_SEL_MODE] = 0; cnt[CNT_SEL_THRESH] = 0; cnt[CNT_SEL_NEED_EQ] = 0;
            if (n > sel_cap) atomicOr(&cnt[CNT_OVERFLOW], 4);
        }
        return;
    }
    // digit 1 (key bits 31..21): the histogram is already there
    for (int i = threadIdx.x; i < SEL_HIST_BINS; i += 1024) hist[i] = ghist[i];
    if (threadIdx.x == 0) s_ln = 0;
    __syncthreads();
    radix_pick(hist, SEL_HIST_BINS, (uint32_t)C, s_wsum, s_pick);
    const uint32_t d1 = s_pick[0];
    uint32_t need = s_pick[1];
    const bool in_lds = hist[d1] <= SEL_LIST_CAP;     // uniform
    __syncthreads();
    // digit 2 (bits 20..10) over the keys of bucket d1; they are copied to LDS on the way when they fit
    for (int i = threadIdx.x; i < SEL_HIST_BINS; i += 1024) hist[i] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 8 * 1024) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * 1024 + threadIdx.x; v[u] = i < n ? resp[i] : 0.0f; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t k = order_key(v[u]);
            if (i0 + u * 1024 + (int)threadIdx.x < n && (k >> 21) == d1) {
                atomicAdd(&hist[(k >> 10) & 2047u], 1u);
                if (in_lds) list[atomicAdd(&s_ln, 1u)] = k;
            }
        }
    }
    __syncthreads();
    radix_pick(hist, 2048, need, s_wsum, s_pick);
    const uint32_t d2 = s_pick[0];
    need = s_pick[1];
    const uint32_t pre = (d1 << 11) | d2;             // key >> 10 of the threshold
    __syncthreads();
    // digit 3 (bits 9..0)
    for (int i = threadIdx.x; i < 1024; i += 1024) hist[i] = 0;
    __syncthreads();
    if (in_lds) {
        const int ln = (int)s_ln;
        for (int i = threadIdx.x; i < ln; i += 1024) {
            const uint32_t k = list[i];
            if ((k >> 10) == pre) atomicAdd(&hist[k & 1023u], 1u);
        }
    } else {
        for (int i0 = 0; i0 < n; i0 += 8 * 1024) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * 1024 + threadIdx.x; v[u] = i < n ? resp[i] : 0.0f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t k = order_key(v[u]);
                if (i0 + u * 1024 + (int)threadIdx.x < n && (k >> 10) == pre) atomicAdd(&hist[k & 1023u], 1u);
            }
        }
    }
    __syncthreads();
    radix_pick(hist, 1024, need, s_wsum, s_pick);
    const uint32_t T = (pre << 10) | s_pick[0];
    const uint32_t need_eq = s_pick[1], eq_total = hist[s_pick[0]];
    __syncthreads();
    if (threadIdx.x == 0) {
        cnt[CNT_SEL_MODE] = 1;
        cnt[CNT_SEL_THRESH] = (int32_t)T;               // key of the C-th largest response
        cnt[CNT_SEL_NEED_EQ] = (int32_t)need_eq;        // how many elements equal to it are inside the top C
        cnt[CNT_SEL_EQ_TOTAL] = (int32_t)eq_total;      // how many elements equal it at all (last digit: bin = full key)
        cnt[CNT_SEL_TIE_LO] = -1; cnt[CNT_SEL_TIE_HI] = -1;   // every tie is taken
    }
    if (eq_total == need_eq) return;                    // (uniform) the usual case: all ties at the threshold are inside the top C
    // More ties than needed (torch.topk's choice among equal values is unspecified): the first need_eq of them in (octave, level,
    // pixel) order, found by a second radix select - over the 44-bit order keys of the tied rows, ascending, 4 digits of 11 bits.
    // O(n) per digit whatever the number of ties (a scan of the whole list per tied row was O(n * ties): seconds on a flat
    // response map whose clamp plateau ties most of the candidates).
    unsigned long long prefix = 0ull, mask = 0ull;
    uint32_t need_desc = eq_total - need_eq + 1;        // the need_eq-th smallest key = the (eq_total - need_eq + 1)-th largest
    for (int shift = 33; shift >= 0; shift -= 11) {
        for (int i = threadIdx.x; i < SEL_HIST_BINS; i += 1024) hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 1024) {
            if (order_key(resp[i]) != T) continue;
            const unsigned long long o = ord_key(ids + 3 * i);
            if ((o & mask) == prefix) atomicAdd(&hist[(uint32_t)(o >> shift) & 2047u], 1u);
        }
        __syncthreads();
        radix_pick(hist, SEL_HIST_BINS, need_desc, s_wsum, s_pick);
        prefix |= (unsigned long long)s_pick[0] << shift;
        mask |= 2047ull << shift;
        need_desc = s_pick[1];
        __syncthreads();
    }
    if (threadIdx.x == 0) { cnt[CNT_SEL_TIE_LO] = (int32_t)(uint32_t)prefix; cnt[CNT_SEL_TIE_HI] = (int32_t)(uint32_t)(prefix >> 32); }
}

__global__ __launch_bounds__(256) void select_compact_kernel(const float* __restrict__ resp, const float* __restrict__ syx,
                                                             const int32_t* __restrict__ ids, int32_t* cnt, int cand_cap,
                                                             float* sel_resp, float* sel_syx, int32_t* sel_ids, int sel_cap) {
    {
        const size_t img = blockIdx.y;
        resp += img * cand_cap; syx += img * cand_cap * 3; ids += img * cand_cap * 3;
        cnt += img * CNT_TOTAL;
        sel_resp += img * sel_cap; sel_syx += img * sel_cap * 3; sel_ids += img * sel_cap * 3;
    }
    int n = cnt[CNT_CAND];
    if (n > cand_cap) n = cand_cap;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    bool take = true;
    if (cnt[CNT_SEL_MODE] == 1) {
        const uint32_t k = order_key(resp[i]), T = (uint32_t)cnt[CNT_SEL_THRESH];
        take = k > T;
        if (k == T) {
            // Ties at the threshold: deterministic - the first NEED_EQ of them in (octave, level, pixel) order = the tied rows whose
            // order key does not exceed the key select_prepare_kernel found (all ones when every tie is taken: the usual case)
            const unsigned long long lim = ((unsigned long long)(uint32_t)cnt[CNT_SEL_TIE_HI] << 32) | (uint32_t)cnt[CNT_SEL_TIE_LO];
            take = ord_key(ids + 3 * i) <= lim;
        }
    }
    if (!take) return;
    const int slot = atomicAdd(&cnt[CNT_SEL], 1);
    if (slot >= sel_cap) return;                        // overflow already flagged by select_prepare
    sel_resp[slot] = resp[i];
    sel_syx[3 * slot] = syx[3 * i]; sel_syx[3 * slot + 1] = syx[3 * i + 1]; sel_syx[3 * slot + 2] = syx[3 * i + 2];
    sel_ids[3 * slot] = ids[3 * i]; sel_ids[3 * slot + 1] = ids[3 * i + 1]; sel_ids[3 * slot + 2] = ids[3 * i + 2];
}

// Rank sort, split over a 2-D grid: block (bx, by) counts, for its 256 rows i, how many of the 256 rows j
// of chunk by come before them; partial ranks are accumulated with integer atomics (exact, order
// independent).  mode 1: descending response (ties: key order); mode 0: (octave, level, pixel)
// ascending = the reference's concatenation order.
__global__ __launch_bounds__(256) void select_rank_kernel(const float* __restrict__ sel_resp, const int32_t* __restrict__ sel_ids,
                                                          const int32_t* __restrict__ cnt, int sel_cap, int32_t* __restrict__ rank) {
    __shared__ __attribute__((aligned(16))) float t_resp[256];
    __shared__ __attribute__((aligned(16))) unsigned long long t_ord[256];
    {
        const size_t img = blockIdx.z;
        sel_resp += img * sel_cap; sel_ids += img * sel_cap * 3; cnt += img * CNT_TOTAL; rank += img * sel_cap;
    }
    int n = cnt[CNT_SEL];
    if (n > sel_cap) n = sel_cap;
    const int mode = cnt[CNT_SEL_MODE];
    const int i = blockIdx.x * 256 + threadIdx.x, base = blockIdx.y * 256;
    if (blockIdx.x * 256 >= n || base >= n) return;
    const int j = base + threadIdx.x;
    // rows past the end of the list: -inf never precedes a (finite) response, the all-ones key never precedes a key
    t_resp[threadIdx.x] = j < n ? sel_resp[j] : -INFINITY;
    t_ord[threadIdx.x] = j < n ? ord_key(sel_ids + 3 * j) : ~0ull;
    const float ri = i < n ? sel_resp[i] : 0.0f;
    const unsigned long long oi = i < n ? ord_key(sel_ids + 3 * i) : 0ull;
    __syncthreads();
    if (i >= n) return;
    int r = 0;
    if (mode == 1) {
        // 16-byte broadcast reads, 4 comparisons each (one LDS read + wait per element made this loop 20 us for 3000 rows);
        // equal responses (practically only the row itself) are resolved by key order in a second, rarely taken loop
        const float4* tr = reinterpret_cast<const float4*>(t_resp);
        int eq = 0;
#pragma unroll 8
        for (int t = 0; t < 64; ++t) {
            const float4 q = tr[t];
            r += (q.x > ri) + (q.y > ri) + (q.z > ri) + (q.w > ri);
            eq += (q.x == ri) + (q.y == ri) + (q.z == ri) + (q.w == ri);
        }
        const bool self_here = (i >= base) && (i < base + 256);
        if (eq > (self_here ? 1 : 0)) {
            for (int t = 0; t < 256; ++t) r += (t_resp[t] == ri && t_ord[t] < oi);
        }
    } else {
        const ulonglong2* to = reinterpret_cast<const ulonglong2*>(t_ord);
#pragma unroll 8
        for (int t = 0; t < 128; ++t) {
            const ulonglong2 q = to[t];
            r += (q.x < oi) + (q.y < oi);
        }
    }
    if (r) atomicAdd(&rank[i], r);
}

__global__ __launch_bounds__(256) void select_emit_kernel(const float* __restrict__ sel_resp, const float* __restrict__ sel_syx,
                                                          const int32_t* __restrict__ sel_ids, int32_t* cnt, int sel_cap,
                                                          const int32_t* __restrict__ rank, float mr, float* out_resp,
                                                          float* out_lafs, int32_t* out_ids, int32_t* out_count) {
    {
        const size_t img = blockIdx.y;
        sel_resp += img * sel_cap; sel_syx += img * sel_cap * 3; sel_ids += img * sel_cap * 3; cnt += img * CNT_TOTAL;
        rank += img * sel_cap; out_resp += img * sel_cap; out_lafs += img * sel_cap * 6; out_ids += img * sel_cap * 3;
        if (out_count) out_count += img;
    }
    int n = cnt[CNT_SEL];
    if (n > sel_cap) n = sel_cap;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) { cnt[CNT_DET] = n; if (out_count) *out_count = n; }
    if (i >= n) return;
    const int r = rank[i];
    out_resp[r] = sel_resp[i];
    const float s = sel_syx[3 * i], y = sel_syx[3 * i + 1], x = sel_syx[3 * i + 2];
    float* L = out_lafs + 6 * (size_t)r;
    const float sm = mr * s;                            // LAFs[:,0:2,0:2] *= mrSize (SparseImgRepresenter.py:198)
    L[0] = sm; L[1] = mr * 0.0f; L[2] = x;
    L[3] = mr * 0.0f; L[4] = sm; L[5] = y;
    out_ids[3 * r] = sel_ids[3 * i]; out_ids[3 * r + 1] = sel_ids[3 * i + 1]; out_ids[3 * r + 2] = sel_ids[3 * i + 2];
}

// ---- OnePassSIR: per-level top-k + frame boundary test on the candidate list, LAFs composed with the dense affine map ------------
// (HandCraftedModules.py:292-363 NMS3dAndComposeAAff, OnePassSIR.py:87-93, LAF.py:442-449 sc_y_x_and_A2LAFs)
struct AffMaps {
    const float* map[AFFNET_MAX_OCTAVES];   // octave o: planar (4, h, w) = (a11, a12, a21, a22) per pixel, image 0
    int hw[AFFNET_MAX_OCTAVES];
    size_t img_stride;
};

// One workgroup per (octave, detection level, image).  The reference takes the top `num_features` responses of a LEVEL when the
// level has more positive maxima than that (HandCraftedModules.py:323-327) - before OnePassSIR drops frames that touch the image
// boundary - so the per-level cut cannot be folded into the global top-k as in the patch-based detector.  MSB-first 8-bit radix
// select over the level's candidates -> tab[(o * MAX_LEVELS + l - 1) * 4 ..] = {mode, threshold key, ties to take, ties in total}.
__global__ __launch_bounds__(1024) void onepass_level_select_kernel(const float* __restrict__ resp, const int32_t* __restrict__ ids, const int32_t* cnt,
                                                                    int cand_cap, int N, int n_detect, int32_t* tab) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_prefix, s_mask, s_need;
    const int o = blockIdx.x / n_detect, l1 = blockIdx.x - o * n_detect;     // l1 = level - 1 (the id stored with a candidate)
    resp += (size_t)blockIdx.y * cand_cap; ids += (size_t)blockIdx.y * cand_cap * 3;
    cnt += blockIdx.y * CNT_TOTAL;
    tab += ((size_t)blockIdx.y * AFFNET_MAX_OCTAVES * AFFNET_MAX_LEVELS + o * AFFNET_MAX_LEVELS + l1) * 4;
    int n = cnt[CNT_CAND];
    if (n > cand_cap) n = cand_cap;
    const int n_pos = cnt[CNT_POS0 + l1 * AFFNET_MAX_OCTAVES + o];
    if (!(N > 0 && N < n_pos)) {
        if (threadIdx.x == 0) { tab[0] = 0; tab[1] = 0; tab[2] = 0; tab[3] = 0; }
        return;
    }
    if (threadIdx.x == 0) { s_prefix = 0; s_mask = 0; s_need = (uint32_t)N; }
    __syncthreads();
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += 1024) hist[i] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix, mask = s_mask;
        for (int i = threadIdx.x; i < n; i += 1024) {
            if (ids[3 * i] != o || ids[3 * i + 1] != l1) continue;
            const uint32_t k = order_key(resp[i]);
            if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t need = s_need;
            int d = 255;
            for (; d > 0; --d) {
                if (hist[d] >= need) break;
                need -= hist[d];
            }
            s_need = need;
            s_prefix = prefix | ((uint32_t)d << shift);
            s_mask = mask | (255u << shift);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { tab[0] = 1; tab[1] = (int32_t)s_prefix; tab[2] = (int32_t)s_need; tab[3] = (int32_t)hist[s_prefix & 255u]; }
}

__global__ __launch_bounds__(256) void onepass_filter_kernel(const float* __restrict__ resp, const float* __restrict__ syx, const int32_t* __restrict__ ids,
                                                             int32_t* cnt, int cand_cap, const int32_t* tab, AffMaps am, float* out_resp, float* out_syx,
                                                             int32_t* out_ids) {
    {
        const size_t img = blockIdx.y;
        resp += img * cand_cap; syx += img * cand_cap * 3; ids += img * cand_cap * 3; cnt += img * CNT_TOTAL;
        tab += img * AFFNET_MAX_OCTAVES * AFFNET_MAX_LEVELS * 4;
        out_resp += img * cand_cap; out_syx += img * cand_cap * 3; out_ids += img * cand_cap * 3;
    }
    int n = cnt[CNT_CAND];
    if (n > cand_cap) n = cand_cap;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool keep = i < n;
    int o = 0, l1 = 0, pix = 0;
    float s = 0.f, y = 0.f, x = 0.f, r = 0.f;
    if (keep) {
        o = ids[3 * i]; l1 = ids[3 * i + 1]; pix = ids[3 * i + 2];
        r = resp[i];
        const int32_t* t = tab + (o * AFFNET_MAX_LEVELS + l1) * 4;
        if (t[0] == 1) {                                       // this level keeps only its top num_features responses
            const uint32_t k = order_key(r), T = (uint32_t)t[1];
            keep = k > T;
            if (k == T && t[3] != t[2]) {                      // more ties than needed: the first ones in pixel order (deterministic)
                int before = 0;
                for (int j = 0; j < n; ++j) before += (ids[3 * j] == o && ids[3 * j + 1] == l1 && order_key(resp[j]) == T && ids[3 * j + 2] < pix) ? 1 : 0;
                keep = before < t[2];
            } else if (k == T) {
                keep = true;
            }
        }
    }
    if (keep) {
        s = syx[3 * i]; y = syx[3 * i + 1]; x = syx[3 * i + 2];
        const float* m = am.map[o] + blockIdx.y * am.img_stride;
        const int hw = am.hw[o];
        // LAF = [s * A | (x, y)] (LAF.py:442-449); OnePassSIR.py:91 tests the frame with its 2x2 part times 3.0 (hard-coded)
        const float b00 = (s * m[pix]) * 3.0f, b01 = (s * m[hw + pix]) * 3.0f, b10 = (s * m[2 * hw + pix]) * 3.0f, b11 = (s * m[3 * hw + pix]) * 3.0f;
        const float px[4] = {-1.f, -1.f, 1.f, 1.f}, py[4] = {-1.f, 1.f, -1.f, 1.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {                          // checkTouchBoundary (LAF.py:98-104), bmm order like shape_filter_kernel
            const float ox = fmaf(x, 1.0f, fmaf(b01, py[k], b00 * px[k]));
            const float oy = fmaf(y, 1.0f, fmaf(b11, py[k], b10 * px[k]));
            if (ox > 1.0f || ox < 0.0f || oy > 1.0f || oy < 0.0f) keep = false;
        }
    }
    const unsigned long long bal = __ballot(keep);
    int wbase = 0;
    if (bal) {
        if (lane == 0) wbase = atomicAdd(&cnt[CNT_CAND2], __popcll(bal));
        wbase = __shfl(wbase, 0, 64);
    }
    if (keep) {
        const int slot = wbase + __popcll(bal & ((1ull << lane) - 1ull));     // slot < cand_cap: cand2 has the capacity of the first list
        out_resp[slot] = r;
        out_syx[3 * slot] = s; out_syx[3 * slot + 1] = y; out_syx[3 * slot + 2] = x;
        out_ids[3 * slot] = o; out_ids[3 * slot + 1] = l1; out_ids[3 * slot + 2] = pix;
    }
}

__global__ void onepass_adopt_count_kernel(int32_t* cnt) {    // the global selection below runs on the filtered list
    cnt += blockIdx.x * CNT_TOTAL;
    cnt[CNT_CAND] = cnt[CNT_CAND2];
}

// select_emit_kernel with the OnePassSIR LAF composition: A = s * A_map[pixel] (LAF.py:444), then x mrSize (OnePassSIR.py:146).
__global__ __launch_bounds__(256) void select_emit_onepass_kernel(const float* __restrict__ sel_resp, const float* __restrict__ sel_syx,
                                                                  const int32_t* __restrict__ sel_ids, int32_t* cnt, int sel_cap,
                                                                  const int32_t* __restrict__ rank, float mr, AffMaps am, float* out_resp,
                                                                  float* out_lafs, int32_t* out_ids, int32_t* out_count) {
    {
        const size_t img = blockIdx.y;
        sel_resp += img * sel_cap; sel_syx += img * sel_cap * 3; sel_ids += img * sel_cap * 3; cnt += img * CNT_TOTAL;
        rank += img * sel_cap; out_resp += img * sel_cap; out_lafs += img * sel_cap * 6; out_ids += img * sel_cap * 3;
        if (out_count) out_count += img;
    }
    int n = cnt[CNT_SEL];
    if (n > sel_cap) n = sel_cap;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) { cnt[CNT_DET] = n; if (out_count) *out_count = n; }
    if (i >= n) return;
    const int r = rank[i];
    out_resp[r] = sel_resp[i];
    const float s = sel_syx[3 * i], y = sel_syx[3 * i + 1], x = sel_syx[3 * i + 2];
    const int o = sel_ids[3 * i], pix = sel_ids[3 * i + 2];
    const float* m = am.map[o] + blockIdx.y * am.img_stride;
    const int hw = am.hw[o];
    float* L = out_lafs + 6 * (size_t)r;
    L[0] = mr * (s * m[pix]); L[1] = mr * (s * m[hw + pix]); L[2] = x;
    L[3] = mr * (s * m[2 * hw + pix]); L[4] = mr * (s * m[3 * hw + pix]); L[5] = y;
    out_ids[3 * r] = o; out_ids[3 * r + 1] = sel_ids[3 * i + 1]; out_ids[3 * r + 2] = pix;
}

// Stage 1 + 2 of the detector (shared by the patch-based and the OnePassSIR path): raw 3-D maxima per octave, then the
// sequential-in-level octaveMap replay -> candidate list (ctx->cand_*, CNT_CAND) and per-level positive counts (CNT_POS0).
// d_responses == NULL: Hessian responses computed from the pyramid in the workspace; otherwise response maps of a custom
// RespNet slot, laid out like the pyramid (image stride = affnet_pyramid_image_stride, level offsets as the pyramid's).
static int detect_candidates(affnet_ctx* ctx, const float* d_responses, AffZeroSegs z, hipStream_t st, bool fold_hist) {
    const affnet_config& c = ctx->cfg;
    const int NLv = c.levels_per_octave;
    if (NLv < 3 || NLv > 8) return aff_fail(ctx, AFFNET_ERR_INVALID, "detect: levels_per_octave = %d (3..8 supported)", NLv);
    const int B = ctx->B;
    const bool two_pass = NLv - 2 <= 3;          // <= 3 detection levels: resolve_count / resolve_apply, no octaveMap in memory
    z.add(ctx->cnt, (size_t)B * CNT_TOTAL * sizeof(int32_t));
    if (!two_pass) z.add(ctx->omap, (size_t)B * ctx->map_stride);
    z.add(ctx->sel_hist, (size_t)B * SEL_HIST_BINS * sizeof(uint32_t));
    { int zrc = aff_zero_multi_async(ctx, z, st); if (zrc) return zrc; }          // counters, octaveMap, histogram + the caller's areas: one launch
    ResolveParams rp;
    memset(&rp, 0, sizeof(rp));
    HessParams hp;
    memset(&hp, 0, sizeof(hp));
    hp.n_oct = c.n_octaves; hp.n_levels = NLv;
    hp.th = c.threshold;
    hp.border = (int)c.mr_size;
    hp.overflow = ctx->cnt + CNT_OVERFLOW;
    hp.levels_stride = ctx->pyr_stride; hp.raw_stride = ctx->raw_stride;
    hp.precomputed = d_responses ? 1 : 0;
    int n_tiles = 0;
    int centroid_pin = -1;                       // -1 = follow ATen's dispatch rule per octave (the default: what the golden vectors were produced with)
    if (const char* e = getenv("AFFNET_CENTROID_ORDER")) { if (!strcmp(e, "onednn")) centroid_pin = 1; else if (!strcmp(e, "native")) centroid_pin = 0; }
    for (int o = 0; o < c.n_octaves; ++o) {
        const OctaveGeom& g = ctx->oct[o];
        HessOct& ho = hp.oct[o];
        ho.levels = (d_responses ? d_responses : ctx->pyr) + g.pyr_off;
        ho.h = g.h; ho.w = g.w;
        for (int l = 0; l < NLv; ++l) { ho.sigma[l] = c.level_sigma[o][l]; ho.sigma4[l] = c.level_sigma4[o][l]; }
        ho.raw = ctx->raw + g.raw_off; ho.raw_cap = g.raw_cap;
        ho.centroid_order = centroid_pin >= 0 ? centroid_pin : (3LL * (long long)g.h * (long long)g.w > 20480LL ? 1 : 0);     // ATen use_mkldnn(): batch-1 3 x 3 conv goes to oneDNN above 20480 input elements
        ho.raw_cnt = ctx->cnt + CNT_RAW0 + o;
        ho.tiles_x = aff_cdiv(g.w, HT_X); ho.tile_begin = n_tiles;
        n_tiles += ho.tiles_x * aff_cdiv(g.h, HT_Y);
        rp.raw[o] = ho.raw; rp.omap[o] = ctx->omap + g.map_off; rp.raw_cap[o] = g.raw_cap;
    }
    {
        // big grids: 4 tiles per workgroup (next tile's loads under this tile's work); small grids: one tile each (latency)
        hp.n_tiles = n_tiles;
        hp.tiles_per_wg = ((long long)n_tiles * B >= 8192) ? 2 : 1;     // measured at 4K: 1 -> 0.202, 2 -> 0.195, 4 -> 0.196 ms per image
        if (const char* e = getenv("AFFNET_HESS_TPW")) { const int v = atoi(e); if (v >= 1 && v <= 64) hp.tiles_per_wg = v; }   // tuning aid
        const dim3 hgrid(aff_cdiv(n_tiles, hp.tiles_per_wg), 1, B);
        switch (NLv) {
            case 3: hipLaunchKernelGGL(hessian_nms_kernel<3>, hgrid, dim3(256), 0, st, hp); break;
            case 4: hipLaunchKernelGGL(hessian_nms_kernel<4>, hgrid, dim3(256), 0, st, hp); break;
            case 5: hipLaunchKernelGGL(hessian_nms_kernel<5>, hgrid, dim3(256), 0, st, hp); break;
            case 6: hipLaunchKernelGGL(hessian_nms_kernel<6>, hgrid, dim3(256), 0, st, hp); break;
            case 7: hipLaunchKernelGGL(hessian_nms_kernel<7>, hgrid, dim3(256), 0, st, hp); break;
            default: hipLaunchKernelGGL(hessian_nms_kernel<8>, hgrid, dim3(256), 0, st, hp); break;
        }
        AFF_LAUNCH_CHECK(ctx);
    }
    rp.n_detect_levels = NLv - 2; rp.cnt = ctx->cnt;
    rp.cand_resp = ctx->cand_resp; rp.cand_syx = ctx->cand_syx; rp.cand_ids = ctx->cand_ids; rp.cand_cap = (int)ctx->cand_cap;
    rp.raw_stride = ctx->raw_stride; rp.map_stride = ctx->map_stride;
    {
        int max_cap = 0;
        for (int o = 0; o < c.n_octaves; ++o) max_cap = ctx->oct[o].raw_cap > max_cap ? ctx->oct[o].raw_cap : max_cap;
        // up to 128 workgroups per (octave, image), grid-stride over the raw list (the count is on the device: workgroups past the end
        // leave at once).  (A cap of 32 left octave 0 of a 4K batch to 8 x 32 workgroups: 42 us per image for the five passes.)
        const int rb = aff_cdiv(max_cap, 256);
        const dim3 rgrid(rb < 128 ? rb : 128, c.n_octaves, B);
        if (two_pass) {
            hipLaunchKernelGGL(resolve_count_kernel, rgrid, dim3(256), 0, st, rp);
            hipLaunchKernelGGL(resolve_apply_kernel, rgrid, dim3(256), 0, st, rp, fold_hist ? ctx->sel_hist : (uint32_t*)nullptr);
        } else {
            for (int l = 1; l <= rp.n_detect_levels; ++l) {
                hipLaunchKernelGGL(level_resolve_kernel, rgrid, dim3(256), 0, st, rp, l, 0);
                hipLaunchKernelGGL(level_resolve_kernel, rgrid, dim3(256), 0, st, rp, l, 1);
            }
        }
    }
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

// Stage 3: global top-C of a candidate list (CNT_CAND rows of resp / syx / ids) -> ranked rows in ctx->sel_* (CNT_SEL, st_rank).
static int select_top(affnet_ctx* ctx, const float* resp, const float* syx, const int32_t* ids, hipStream_t st, bool have_hist) {
    const affnet_config& c = ctx->cfg;
    const int B = ctx->B;
    if (!have_hist) {        // (the two-launch resolve fills the first-digit histogram while it emits the candidates)
        int hb = aff_cdiv((int)ctx->cand_cap, 1024);
        hb = hb < 1 ? 1 : (hb > 128 ? 128 : hb);
        hipLaunchKernelGGL(select_hist_kernel, dim3(hb, B), dim3(256), 0, st, resp, ctx->cnt, (int)ctx->cand_cap, c.num_prefilter, ctx->sel_hist);
        AFF_LAUNCH_CHECK(ctx);
    }
    hipLaunchKernelGGL(select_prepare_kernel, dim3(B), dim3(1024), 0, st, resp, ids, ctx->cnt, (int)ctx->cand_cap, c.num_prefilter, ctx->cap_pre,
                       ctx->sel_hist);
    AFF_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(select_compact_kernel, dim3(aff_cdiv((int)ctx->cand_cap, 256), B), dim3(256), 0, st, resp, syx, ids, ctx->cnt,
                       (int)ctx->cand_cap, ctx->sel_resp, ctx->sel_syx, ctx->sel_ids, ctx->cap_pre);
    AFF_LAUNCH_CHECK(ctx);
    const int nb = aff_cdiv(ctx->cap_pre, 256);
    hipLaunchKernelGGL(select_rank_kernel, dim3(nb, nb, B), dim3(256), 0, st, ctx->sel_resp, ctx->sel_ids, ctx->cnt, ctx->cap_pre, ctx->st_rank);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

// Output rows past the row count and the rank accumulator are cleared together with the detector's own areas (detect_candidates).
static AffZeroSegs clear_outputs(affnet_ctx* ctx, float* d_resp, float* d_lafs, int32_t* d_ids) {
    const size_t P = (size_t)ctx->B * ctx->cap_pre;
    AffZeroSegs z;
    if ((char*)d_lafs == (char*)d_resp + P * sizeof(float) && (char*)d_ids == (char*)d_lafs + P * 6 * sizeof(float)) {
        z.add(d_resp, P * 10 * sizeof(float));        // the context's own detection list is one contiguous area
    } else {
        z.add(d_resp, P * sizeof(float)); z.add(d_lafs, P * 6 * sizeof(float)); z.add(d_ids, P * 3 * sizeof(int32_t));
    }
    z.add(ctx->st_rank, P * sizeof(int32_t));
    return z;
}

int aff_detect_impl(affnet_ctx* ctx, const float* d_responses, float* d_resp, float* d_lafs, int32_t* d_ids, int32_t* d_count, void* stream) {
    if (!ctx || !ctx->ws || !d_resp || !d_lafs || !d_ids) return aff_fail(ctx, AFFNET_ERR_INVALID, "detect: context not bound or null output");
    hipStream_t st = (hipStream_t)stream;
    const bool fold = ctx->cfg.levels_per_octave - 2 <= 3;
    int rc = detect_candidates(ctx, d_responses, clear_outputs(ctx, d_resp, d_lafs, d_ids), st, fold);
    if (rc) return rc;
    rc = select_top(ctx, ctx->cand_resp, ctx->cand_syx, ctx->cand_ids, st, fold);
    if (rc) return rc;
    const int nb = aff_cdiv(ctx->cap_pre, 256);
    hipLaunchKernelGGL(select_emit_kernel, dim3(nb, ctx->B), dim3(256), 0, st, ctx->sel_resp, ctx->sel_syx, ctx->sel_ids, ctx->cnt, ctx->cap_pre,
                       ctx->st_rank, ctx->cfg.mr_size, d_resp, d_lafs, d_ids, d_count);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

int aff_fullconv_launch(affnet_ctx* ctx, const float* packed, const float* img, size_t img_stride, int h, int w, float* out, size_t out_stride,
                        float* scratch, size_t scratch_stride, int B, hipStream_t st);

// OnePassSIR.multiScaleDetectorAff (OnePassSIR.py:53-115) on the pyramid in the workspace.  d_packed_fullconv != NULL: the dense
// AffNetFastFullConv maps of every octave are computed here (level 0 of each octave, OnePassSIR.py:69); NULL: the caller has written
// them into the workspace (affnet_affmap_offset) - the slot form for a foreign dense AffNet.  Results go to the context's internal
// detection list, consumed by affnet_describe_detected (with nets->d_affnet == NULL: no per-patch shape stage).
int aff_detect_onepass_impl(affnet_ctx* ctx, const float* d_packed_fullconv, const float* d_responses, hipStream_t st) {
    if (!ctx || !ctx->ws) return aff_fail(ctx, AFFNET_ERR_INVALID, "detect_onepass: context not bound");
    if (!ctx->cfg.onepass) return aff_fail(ctx, AFFNET_ERR_INVALID, "detect_onepass: the context was not created with cfg.onepass");
    const affnet_config& c = ctx->cfg;
    const int B = ctx->B;
    AffMaps am;
    memset(&am, 0, sizeof(am));
    am.img_stride = ctx->aff_stride;
    for (int o = 0; o < c.n_octaves; ++o) {
        const OctaveGeom& g = ctx->oct[o];
        am.map[o] = ctx->affmap + ctx->aff_off[o];
        am.hw[o] = g.h * g.w;
        if (d_packed_fullconv) {
            int rc = aff_fullconv_launch(ctx, d_packed_fullconv, ctx->pyr + g.pyr_off, ctx->pyr_stride, g.h, g.w, ctx->affmap + ctx->aff_off[o],
                                         ctx->aff_stride, ctx->dense, ctx->dense_stride, B, st);
            if (rc) return rc;
        }
    }
    int rc = detect_candidates(ctx, d_responses, clear_outputs(ctx, ctx->st_det_resp, ctx->st_det_lafs, ctx->st_det_ids), st, false);
    if (rc) return rc;
    const int n_detect = c.levels_per_octave - 2;
    hipLaunchKernelGGL(onepass_level_select_kernel, dim3(c.n_octaves * n_detect, B), dim3(1024), 0, st, ctx->cand_resp, ctx->cand_ids, ctx->cnt,
                       (int)ctx->cand_cap, c.num_features, n_detect, ctx->lvltab);
    AFF_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(onepass_filter_kernel, dim3(aff_cdiv((int)ctx->cand_cap, 256), B), dim3(256), 0, st, ctx->cand_resp, ctx->cand_syx, ctx->cand_ids,
                       ctx->cnt, (int)ctx->cand_cap, ctx->lvltab, am, ctx->cand2_resp, ctx->cand2_syx, ctx->cand2_ids);
    AFF_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(onepass_adopt_count_kernel, dim3(B), dim3(1), 0, st, ctx->cnt);
    AFF_LAUNCH_CHECK(ctx);
    rc = select_top(ctx, ctx->cand2_resp, ctx->cand2_syx, ctx->cand2_ids, st, false);
    if (rc) return rc;
    const int nb = aff_cdiv(ctx->cap_pre, 256);
    hipLaunchKernelGGL(select_emit_onepass_kernel, dim3(nb, B), dim3(256), 0, st, ctx->sel_resp, ctx->sel_syx, ctx->sel_ids, ctx->cnt, ctx->cap_pre,
                       ctx->st_rank, c.mr_size, am, ctx->st_det_resp, ctx->st_det_lafs, ctx->st_det_ids, ctx->st_det_count);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

extern "C" int affnet_detect(affnet_ctx* ctx, float* d_resp, float* d_lafs, int32_t* d_ids, int32_t* d_count, void* stream) {
    AFF_DEVICE(ctx);
    return aff_detect_impl(ctx, nullptr, d_resp, d_lafs, d_ids, d_count, stream);
}

extern "C" int affnet_detect_responses(affnet_ctx* ctx, const float* d_responses, float* d_resp, float* d_lafs, int32_t* d_ids,
                                       int32_t* d_count, void* stream) {
    AFF_DEVICE(ctx);
    if (!d_responses) return aff_fail(ctx, AFFNET_ERR_INVALID, "detect_responses: null response pyramid");
    return aff_detect_impl(ctx, d_responses, d_resp, d_lafs, d_ids, d_count, stream);
}
