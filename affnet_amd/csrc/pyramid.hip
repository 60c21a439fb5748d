// Gaussian scale pyramid for gfx950.
//
// Replaces Utils.py:150-166 (GaussianBlur.forward) and HandCraftedModules.py:23-56
// (ScalePyramid.forward).  The reference blurs with a full k x k 2-D cross-correlation
// (F.conv2d) after replicate padding; on CPU that kernel accumulates the taps with fused
// multiply-adds in row-major tap order starting from 0.  This kernel performs exactly the same
// fmaf chain per output pixel, so every pyramid level is bit-identical to the reference's CPU
// result (DESIGN.md "bit-exact detector"); a separable blur would be ~6x fewer MACs but moves
// levels by ~2e-4 and flips keypoints.
//
// Layout: one workgroup = 256 threads = 64 x 64 output tile; each thread owns a 4 (columns) x 4 (rows) register tile and
// walks the 4 + K - 1 input rows it needs once: one (K+3)-wide register window per input row (16-byte LDS reads) feeds
// up to 4 x 4 x K fmaf.  Input rows arrive in ascending order, so every output pixel still accumulates its taps in
// row-major order - the chain is unchanged, only shared loads are reused (the 1-row version was LDS-bound: 1.45 LDS bytes per
// fmaf, SQ_LDS_BANK_CONFLICT 62 % of LDS cycles; this one reads a quarter of that and its row stride == 0 (mod 16 floats)
// is conflict free for ds_read_b128).  Taps arrive as a by-value kernel argument (scalar loads -> SGPR operands of
// v_fmac_f32).  Optional fused stride-2 decimation writes the next octave's level 0 (F.avg_pool2d(k=1, s=2),
// HandCraftedModules.py:46-47).
#include "common.h"

#define BT_X 64
// BT_R = output rows per thread (tile height BT_Y = 16 * BT_R).  4 is the throughput shape (one (K+3)-wide register window per input
// row feeds 4 output rows); 1 is the latency shape for small grids: a 64 x 16 tile is a quarter of the serial work per workgroup and
// four times as many workgroups - at one 800 x 640 image per call the 64 x 64 tiling put 130 workgroups on 256 CUs and every one
// of the 25 dependent blur launches took 9-17 us whatever the octave (profiles/r03_s0_config2_gap_table.md).  The per-pixel fmaf
// chain is the same in both shapes.

template <int K>
struct Taps { float w[K * K]; };

// acc = fma(a, w, acc) as ONE v_fmac_f32 with the (wave-uniform) tap in an SGPR: written as asm so that the vectoriser does not
// pair it with its neighbour and rebuild the register pairs with v_mov (IEEE fma: bit-identical to fmaf)
#define constexpr_fmac(acc, a, w) asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "s"(w), "v"(a))

template <int K, int BT_R>
struct BlurTile {
    static constexpr int R = K / 2;
    static constexpr int LW = BT_X + 2 * R;               // tile width incl. halo
    static constexpr int LS = (LW + 15) & ~15;            // row stride: multiple of 16 floats (16-B aligned rows, conflict-free b128)
    static constexpr int LH = 16 * BT_R + 2 * R;
    static constexpr int FLOATS = LH * LS;
};

// One output tile (tile coordinates bx, by) of one blur; `tile` = BlurTile<K, BT_R>::FLOATS floats of LDS.
template <int K, int BT_R>
__device__ __forceinline__ void blur2d_tile(float* tile, const float* __restrict__ in, float* __restrict__ out, float* __restrict__ dec_out, int h, int w,
                                            int w2, int bx, int by, const Taps<K>& taps) {
    constexpr int R = K / 2;
    constexpr int BT_Y = 16 * BT_R;
    constexpr int LW = BlurTile<K, BT_R>::LW, LS = BlurTile<K, BT_R>::LS, LH = BlurTile<K, BT_R>::LH;
    const int x0 = bx * BT_X, y0 = by * BT_Y;
    {
        // Tile loader: wavefront v takes tile rows v, v + 4, ...; lane c loads column c and, for c < K - 1, column 64 + c.  The row
        // (replicate-clamped) is wave-uniform, so a load's address is a scalar row base + the lane's clamped column computed once:
        // no vector arithmetic per load.  (An element-per-thread mapping i = tid + 256 k cost a division by the tile width, two
        // clamps and a 64-bit multiply-add per load - 15 vector instructions, several quarter-rate - a quarter of the kernel's VALU
        // cycles.)  Every load of a thread is in flight before the first LDS store; rows below the image repeat its last row.
        constexpr int RPW = (LH + 3) / 4;
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
        int gx0 = x0 - R + lane, gx1 = x0 - R + 64 + lane;
        gx0 = gx0 < 0 ? 0 : (gx0 >= w ? w - 1 : gx0);          // replicate padding
        gx1 = gx1 < 0 ? 0 : (gx1 >= w ? w - 1 : gx1);
        const bool second = lane < LW - 64;
        float v0[RPW], v1[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            int gy = y0 - R + wave + 4 * r;
            gy = gy < 0 ? 0 : (gy >= h ? h - 1 : gy);
            const float* rp = in + (size_t)gy * w;
            v0[r] = rp[gx0];
            if (second) v1[r] = rp[gx1];
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = wave + 4 * r;
            if (r < RPW - 1 || row < LH) {
                tile[row * LS + lane] = v0[r];
                if (second) tile[row * LS + 64 + lane] = v1[r];
            }
        }
    }
    __syncthreads();
    const int tx = (threadIdx.x & 15) * 4, ty = (threadIdx.x >> 4) * BT_R;
    if (y0 + ty >= h) return;
    // The four outputs of a thread are two packed pairs (x, x+1), (x+2, x+3): tap j multiplies the input pairs (r[j], r[j+1]) and
    // (r[j+2], r[j+3]) of the row window r[0 .. K+2].  Even j finds them as E[k] = (r[2k], r[2k+1]), the halves of the aligned
    // 16-byte LDS reads: one v_pk_fma_f32 per pair.  Odd j straddles those registers; its four products are plain v_fmac_f32 on
    // the single halves (same fp32 VALU rate as the packed form: 2 x 2 cycles against 4), with the tap as the SGPR operand.
    // (Round 2 read the odd pairs O[k] = (r[2k+1], r[2k+2]) a second time from LDS with 4-byte reads: lanes 16 bytes apart hit 8
    // of the 32 banks, two tile rows per 32-lane group the same 8: 4-way conflicts on 16 reads per row - 65 % of the LDS cycles
    // of the K = 11 kernel, and what bound the latency shape.)  Per accumulator the fmaf chain is unchanged: taps in row-major order.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc01[BT_R], acc23[BT_R];
#pragma unroll
    for (int rr = 0; rr < BT_R; ++rr) { acc01[rr] = (f32x2){0.f, 0.f}; acc23[rr] = (f32x2){0.f, 0.f}; }
    constexpr int NV = (K + 3 + 3) / 4;            // float4 loads covering K+3 values
    // Input-row loop: rolled in the throughput shape (register window + K scalar taps per (row, output row) pair); fully unrolled in
    // the latency shape.
    constexpr int UNR = (BT_R == 1) ? K : 1;
#pragma unroll UNR
    for (int i = 0; i < BT_R + K - 1; ++i) {
        f32x2 E[NV * 2];
        const float4* row = reinterpret_cast<const float4*>(&tile[(ty + i) * LS + tx]);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 q = row[v];
            E[2 * v] = (f32x2){q.x, q.y}; E[2 * v + 1] = (f32x2){q.z, q.w};
        }
#pragma unroll
        for (int rr = 0; rr < BT_R; ++rr) {
            const int ti = i - rr;                  // tap row for output row rr (uniform across the workgroup)
            if (ti < 0 || ti >= K) continue;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const float wt = taps.w[ti * K + j];
                if (j & 1) {
                    constexpr_fmac(acc01[rr].x, E[(j - 1) / 2].y, wt);
                    constexpr_fmac(acc01[rr].y, E[(j + 1) / 2].x, wt);
                    constexpr_fmac(acc23[rr].x, E[(j + 1) / 2].y, wt);
                    constexpr_fmac(acc23[rr].y, E[(j + 3) / 2].x, wt);
                } else {
                    const f32x2 wv = (f32x2){wt, wt};
                    acc01[rr] = __builtin_elementwise_fma(E[j / 2], wv, acc01[rr]);
                    acc23[rr] = __builtin_elementwise_fma(E[j / 2 + 1], wv, acc23[rr]);
                }
            }
        }
    }
    float acc[BT_R][4];
#pragma unroll
    for (int rr = 0; rr < BT_R; ++rr) { acc[rr][0] = acc01[rr].x; acc[rr][1] = acc01[rr].y; acc[rr][2] = acc23[rr].x; acc[rr][3] = acc23[rr].y; }
    const int x = x0 + tx;
#pragma unroll
    for (int rr = 0; rr < BT_R; ++rr) {
        const int y = y0 + ty + rr;
        if (y >= h) break;
        if (x + 3 < w && (w & 3) == 0 && ((size_t)out & 15) == 0) {
            *reinterpret_cast<float4*>(&out[(size_t)y * w + x]) = make_float4(acc[rr][0], acc[rr][1], acc[rr][2], acc[rr][3]);
            if (dec_out && !(y & 1)) {
                dec_out[(size_t)(y >> 1) * w2 + (x >> 1)] = acc[rr][0];
                dec_out[(size_t)(y >> 1) * w2 + (x >> 1) + 1] = acc[rr][2];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (x + q < w) {
                    out[(size_t)y * w + x + q] = acc[rr][q];
                    if (dec_out && !(y & 1) && !((x + q) & 1)) dec_out[(size_t)(y >> 1) * w2 + ((x + q) >> 1)] = acc[rr][q];
                }
            }
        }
    }
}

template <int K, int BT_R>
__global__ __launch_bounds__(256) void blur2d_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                     float* __restrict__ dec_out, int h, int w, int w2, size_t in_stride,
                                                     size_t out_stride, Taps<K> taps) {
    __shared__ __attribute__((aligned(16))) float tile[BlurTile<K, BT_R>::FLOATS];
    in += blockIdx.z * in_stride;                   // blockIdx.z = image of the batch
    out += blockIdx.z * out_stride;
    if (dec_out) dec_out += blockIdx.z * out_stride;
    blur2d_tile<K, BT_R>(tile, in, out, dec_out, h, w, w2, blockIdx.x, blockIdx.y, taps);
}

// Two INDEPENDENT blurs in one launch: the last level of octave o (L-1 from L-2) and the first blur of octave o + 1 (level 1 from the
// decimated level 0) both depend only on octave o's level L-2.  blockIdx.x < a.tiles: a tile of job a, else of job b (tiles_x per
// job).  One launch less per octave on the serial chain of the pyramid (5 of 25 at 6 octaves).
struct BlurJob { const float* in; float* out; int h, w, tiles_x, tiles; };
template <int KA, int KB, int BT_R>
__global__ __launch_bounds__(256) void blur2d_pair_kernel(BlurJob a, BlurJob b, size_t stride, Taps<KA> ta, Taps<KB> tb) {
    constexpr int FL = BlurTile<KA, BT_R>::FLOATS > BlurTile<KB, BT_R>::FLOATS ? BlurTile<KA, BT_R>::FLOATS : BlurTile<KB, BT_R>::FLOATS;
    __shared__ __attribute__((aligned(16))) float tile[FL];
    const size_t img = blockIdx.z * stride;
    if ((int)blockIdx.x < a.tiles) {
        const int t = blockIdx.x;
        blur2d_tile<KA, BT_R>(tile, a.in + img, a.out + img, nullptr, a.h, a.w, 0, t % a.tiles_x, t / a.tiles_x, ta);
    } else {
        const int t = (int)blockIdx.x - a.tiles;
        blur2d_tile<KB, BT_R>(tile, b.in + img, b.out + img, nullptr, b.h, b.w, 0, t % b.tiles_x, t / b.tiles_x, tb);
    }
}

// The reference's default schedule (init_sigma 1.6, 3 levels): 15 x 15 taps for the last level, 9 x 9 for level 1.  Other tap pairs
// take the two separate launches.
static bool launch_blur_pair(const float* in_a, float* out_a, int ha, int wa, const float* taps_a, int ka, const float* in_b, float* out_b, int hb, int wb,
                             const float* taps_b, int kb, int batch, size_t stride, hipStream_t st) {
    if (ka != 15 || kb != 9) return false;
    Taps<15> ta; Taps<9> tb;
    memcpy(ta.w, taps_a, sizeof(ta.w)); memcpy(tb.w, taps_b, sizeof(tb.w));
    const int t64 = (aff_cdiv(wa, BT_X) * aff_cdiv(ha, 64) + aff_cdiv(wb, BT_X) * aff_cdiv(hb, 64)) * batch;
    const int ty = t64 >= 1024 ? 64 : 16;
    BlurJob a{in_a, out_a, ha, wa, aff_cdiv(wa, BT_X), aff_cdiv(wa, BT_X) * aff_cdiv(ha, ty)};
    BlurJob b{in_b, out_b, hb, wb, aff_cdiv(wb, BT_X), aff_cdiv(wb, BT_X) * aff_cdiv(hb, ty)};
    const dim3 grid(a.tiles + b.tiles, 1, batch);
    if (ty == 64) hipLaunchKernelGGL((blur2d_pair_kernel<15, 9, 4>), grid, dim3(256), 0, st, a, b, stride, ta, tb);
    else hipLaunchKernelGGL((blur2d_pair_kernel<15, 9, 1>), grid, dim3(256), 0, st, a, b, stride, ta, tb);
    return true;
}

template <int K>
static void launch_blur(const float* in, float* out, float* dec, int h, int w, int batch, size_t in_stride, size_t out_stride,
                        const float* taps, hipStream_t st) {
    Taps<K> t;
    memcpy(t.w, taps, sizeof(float) * K * K);
    const int tiles64 = aff_cdiv(w, BT_X) * aff_cdiv(h, 64) * batch;
    if (tiles64 >= 1024) {       // >= 4 workgroups per CU with the tall tile: throughput shape
        dim3 grid(aff_cdiv(w, BT_X), aff_cdiv(h, 64), batch);
        hipLaunchKernelGGL((blur2d_kernel<K, 4>), grid, dim3(256), 0, st, in, out, dec, h, w, (w - 1) / 2 + 1, in_stride, out_stride, t);
    } else {
        dim3 grid(aff_cdiv(w, BT_X), aff_cdiv(h, 16), batch);
        hipLaunchKernelGGL((blur2d_kernel<K, 1>), grid, dim3(256), 0, st, in, out, dec, h, w, (w - 1) / 2 + 1, in_stride, out_stride, t);
    }
}

// batch images per launch: image b reads in + b*in_stride and writes out / dec + b*out_stride (floats)
static int blur_dispatch(affnet_ctx* ctx, const float* in, float* out, float* dec, int h, int w, const float* taps, int k,
                         hipStream_t st, int batch = 1, size_t in_stride = 0, size_t out_stride = 0) {
    switch (k) {
        case 3: launch_blur<3>(in, out, dec, h, w, batch, in_stride, out_stride, taps, st); break;
        case 5: launch_blur<5>(in, out, dec, h, w, batch, in_stride, out_stride, taps, st); break;
        case 7: launch_blur<7>(in, out, dec, h, w, batch, in_stride, out_stride, taps, st); break;
        case 9: launch_blur<9>(in, out, dec, h, w, batch, in_stride, out_stride, taps, st); break;
        case 11: launch_blur<11>(in, out, dec, h, w, batch, in_stride, out_stride, taps, st); break;
        case 13: launch_blur<13>(in, out, dec, h, w, batch, in_stride, out_stride, taps, st); break;
        case 15: launch_blur<15>(in, out, dec, h, w, batch, in_stride, out_stride, taps, st); break;
        case 17: launch_blur<17>(in, out, dec, h, w, batch, in_stride, out_stride, taps, st); break;
        case 19: launch_blur<19>(in, out, dec, h, w, batch, in_stride, out_stride, taps, st); break;
        case 21: launch_blur<21>(in, out, dec, h, w, batch, in_stride, out_stride, taps, st); break;
        case 23: launch_blur<23>(in, out, dec, h, w, batch, in_stride, out_stride, taps, st); break;
        case 25: launch_blur<25>(in, out, dec, h, w, batch, in_stride, out_stride, taps, st); break;
        case 27: launch_blur<27>(in, out, dec, h, w, batch, in_stride, out_stride, taps, st); break;
        case 29: launch_blur<29>(in, out, dec, h, w, batch, in_stride, out_stride, taps, st); break;
        case 31: launch_blur<31>(in, out, dec, h, w, batch, in_stride, out_stride, taps, st); break;
        case 33: launch_blur<33>(in, out, dec, h, w, batch, in_stride, out_stride, taps, st); break;
        case 35: launch_blur<35>(in, out, dec, h, w, batch, in_stride, out_stride, taps, st); break;
        case 37: launch_blur<37>(in, out, dec, h, w, batch, in_stride, out_stride, taps, st); break;
        case 1: {   // a 1 x 1 "Gaussian" (sigma < 1/6): conv2d with the single tap w = 1 -> fma(x, 1, 0) = x
            launch_blur<1>(in, out, dec, h, w, batch, in_stride, out_stride, taps, st); break;
        }
        default:
            return aff_fail(ctx, AFFNET_ERR_INVALID, "unsupported Gaussian size %d (supported: odd 1..37)", k);
    }
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

extern "C" int affnet_gauss_blur(affnet_ctx* ctx, const float* d_in, float* d_out, int h, int w, const float* h_taps, int k,
                                 void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !d_in || !d_out || !h_taps || h < 1 || w < 1) return aff_fail(ctx, AFFNET_ERR_INVALID, "gauss_blur: bad argument");
    return blur_dispatch(ctx, d_in, d_out, nullptr, h, w, h_taps, k, (hipStream_t)stream);
}

extern "C" int affnet_pyramid_build(affnet_ctx* ctx, const float* d_img, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !ctx->ws || !d_img) return aff_fail(ctx, AFFNET_ERR_INVALID, "pyramid_build: context not bound or null image");
    hipStream_t st = (hipStream_t)stream;
    const affnet_config& c = ctx->cfg;
    const int L = c.levels_per_octave;
    const int dec_level = L - 2;  // i == nLevels (HandCraftedModules.py:46)
    int first_level = 1;             // 2 when the previous octave's paired launch already produced this octave's level 1
    for (int o = 0; o < c.n_octaves; ++o) {
        const OctaveGeom& g = ctx->oct[o];
        float* base = ctx->pyr + g.pyr_off;
        const size_t lvl = (size_t)g.h * g.w;
        int next_first_level = 1;
        if (o == 0) {
            if (c.first_blur_taps > 0) {
                int rc = blur_dispatch(ctx, d_img, base, nullptr, g.h, g.w, c.first_blur, c.first_blur_taps, st, ctx->B, lvl,
                                       ctx->pyr_stride);
                if (rc) return rc;
            } else {
                int crc = aff_copy2d_async(ctx, base, ctx->pyr_stride * sizeof(float), d_img, lvl * sizeof(float), lvl * sizeof(float), (size_t)ctx->B, st);
                if (crc) return crc;
            }
        }
        for (int l = first_level; l < L; ++l) {
            float* dec = nullptr;
            if (l == dec_level && o + 1 < c.n_octaves) dec = ctx->pyr + ctx->oct[o + 1].pyr_off;
            const bool own = (o == 0 && c.level_blur0_taps[1] > 0);      // octave 0 with its own blur sequence (init_sigma <= 0.5)
            const float* taps = own ? c.level_blur0[l] : c.level_blur[l];
            const int k = own ? c.level_blur0_taps[l] : c.level_blur_taps[l];
            if (l == L - 1 && dec_level == L - 2 && o + 1 < c.n_octaves) {
                // this octave's last level and the next octave's level 1 in one launch (both read what level L-2 produced)
                const OctaveGeom& g1 = ctx->oct[o + 1];
                float* base1 = ctx->pyr + g1.pyr_off;
                if (launch_blur_pair(base + (l - 1) * lvl, base + l * lvl, g.h, g.w, taps, k, base1, base1 + (size_t)g1.h * g1.w, g1.h, g1.w, c.level_blur[1],
                                     c.level_blur_taps[1], ctx->B, ctx->pyr_stride, st)) {
                    AFF_LAUNCH_CHECK(ctx);
                    next_first_level = 2;
                    continue;
                }
            }
            int rc = blur_dispatch(ctx, base + (l - 1) * lvl, base + l * lvl, dec, g.h, g.w, taps, k, st, ctx->B, ctx->pyr_stride, ctx->pyr_stride);
            if (rc) return rc;
        }
        first_level = next_first_level;
    }
    return AFFNET_OK;
}
