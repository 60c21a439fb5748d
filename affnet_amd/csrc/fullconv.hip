// Fully-convolutional AffNet for gfx950: the dense affine-shape map of OnePassSIR.
//
// Replaces architectures.py:21-31 (LocalNorm2d), :629-674 (AffNetFastFullConv.forward) incl. LAF.py:293-297
// (rectifyAffineTransformationUpIsUpFullyConv), and HandCraftedModules.py:194-206 (NMS2d).
//
//   image (h x w, fp32 0..255)
//     -> local_norm_kernel      : (x - mean33) / (sqrt|E33[x^2] - mean33^2| + 1e-10) clamped to +-6, reflect padding.  The two
//                                 33 x 33 box sums are accumulated per output pixel sequentially in row-major order in fp32 -
//                                 exactly what ATen's CPU avg_pool2d does - so the sums are the reference's bit for bit and the
//                                 normalised image agrees to 1 ulp (x^2 up to 65025 against window variances of a few units:
//                                 E[x^2] - mean^2 cancels catastrophically, any other summation order moves it by 1e-3);
//     -> dense_conv0_kernel     : reflect-pad by 14 (fused into the tile loader), conv 1 -> 16 on the matrix cores (K = 9 -> 12)
//     -> dense_conv_kernel x 5  : 16->16, 16->32 /2, 32->32, 32->64 /2, 64->64; one workgroup = one 32 / 16 / 8-pixel square
//                                 input tile (+1 px apron of REAL neighbours) staged in LDS in the trunk's channel-interleaved
//                                 layout and contracted by the SAME conv3x3_mfma instantiations as the per-patch AffNet trunk
//                                 (cnn_mfma.h); activations travel between layers as [C/4][Y][X] float4 planes in HBM
//     -> fullconv_head_kernel   : 8 x 8 valid conv 64 -> 3 (+ bias) on MFMA with the taps of a kernel row folded into N
//     -> fullconv_finish_kernel : bilinear upsampling to h x w (align_corners = False), tanh, [[1+x0, 0],[x1, 1+x2]],
//                                 up-is-up rectification -> planar (4, h, w) map (a11, 0, a21, a22)
//
// Algorithmic work at 1024 x 768 (5 octaves): 23.1 GFLOP, ~0.5 GB of activation traffic, 0.34 ms on MI355X - against 57.6 GFLOP /
// 0.43 ms for 3000 per-patch AffNet evaluations.
#include "cnn_mfma.h"

struct DenseGeom {
    int h, w;          // image
    int Hp, Wp;        // reflect-padded by 14: conv0 / conv1 resolution
    int H2, W2;        // after the stride-2 conv2
    int H4, W4;        // after the stride-2 conv4
    int Hf, Wf;        // after the 8 x 8 valid head
};

static DenseGeom dense_geom(int h, int w) {
    DenseGeom g;
    g.h = h; g.w = w;
    g.Hp = h + 28; g.Wp = w + 28;
    g.H2 = (g.Hp - 1) / 2 + 1; g.W2 = (g.Wp - 1) / 2 + 1;
    g.H4 = (g.H2 - 1) / 2 + 1; g.W4 = (g.W2 - 1) / 2 + 1;
    g.Hf = g.H4 - 7; g.Wf = g.W4 - 7;
    return g;
}

// floats of scratch per image: normalised image + two ping-pong activation buffers (the largest tensor is 16 x Hp x Wp)
static size_t dense_scratch_floats(int h, int w) {
    const DenseGeom g = dense_geom(h, w);
    const size_t norm = aff_align((size_t)h * w * sizeof(float)) / sizeof(float);
    const size_t act = aff_align((size_t)16 * g.Hp * g.Wp * sizeof(float)) / sizeof(float);
    return norm + 2 * act;
}

extern "C" size_t affnet_fullconv_scratch_bytes(int h, int w) {
    if (h < 34 || w < 34) return 0;
    return dense_scratch_floats(h, w) * sizeof(float);
}

__device__ __forceinline__ int reflect_idx(int t, int n) {      // F.pad(..., 'reflect'): -1 -> 1, n -> n - 2
    t = t < 0 ? -t : t;
    return t >= n ? 2 * (n - 1) - t : t;
}

// ---- LocalNorm2d(33) ---------------------------------------------------------------------------------
#define LN_K 33
#define LN_R 16
#define LN_T 64                     // output tile side
#define LN_LW (LN_T + 2 * LN_R)     // 96
__global__ __launch_bounds__(256) void local_norm_kernel(const float* __restrict__ in, float* __restrict__ out, int h, int w, size_t in_stride,
                                                         size_t out_stride) {
    __shared__ __attribute__((aligned(16))) float tile[LN_LW * LN_LW];
    in += blockIdx.z * in_stride;
    out += blockIdx.z * out_stride;
    const int x0 = blockIdx.x * LN_T, y0 = blockIdx.y * LN_T;
    for (int i = threadIdx.x; i < LN_LW * LN_LW; i += 256) {
        const int ty = i / LN_LW, tx = i - ty * LN_LW;
        // tile (ty, tx) = source (y0 + ty - 16, x0 + tx - 16); rows / columns past h + 15 / w + 15 are never part of a valid window
        const int gy = reflect_idx(min(y0 + ty, h + 2 * LN_R - 1) - LN_R, h), gx = reflect_idx(min(x0 + tx, w + 2 * LN_R - 1) - LN_R, w);
        tile[i] = in[(size_t)gy * w + gx];
    }
    __syncthreads();
    const int tx = (threadIdx.x & 15) * 4, ty = (threadIdx.x >> 4) * 4;
    if (y0 + ty >= h || x0 + tx >= w) return;
    // 4 x 4 outputs per thread; input rows arrive in ascending order, every window is summed left to right: each accumulator
    // sees its 33 x 33 values in row-major order = the order of ATen's CPU avg_pool2d loop (acc = acc + v, fp32)
    // (sum, sum of squares) of one output travel as a float2 and grow with one v_pk_add_f32 per window element
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 s12[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) s12[r][q] = (f32x2){0.f, 0.f};
#pragma unroll 1
    for (int i = 0; i < 4 + LN_K - 1; ++i) {
        f32x2 vv[LN_K + 3];
        const float4* row = reinterpret_cast<const float4*>(&tile[(ty + i) * LN_LW + tx]);
#pragma unroll
        for (int k = 0; k < (LN_K + 3) / 4; ++k) {
            const float4 t = row[k];
            vv[4 * k] = (f32x2){t.x, t.x * t.x}; vv[4 * k + 1] = (f32x2){t.y, t.y * t.y};      // x * x rounded to fp32 first, like the
            vv[4 * k + 2] = (f32x2){t.z, t.z * t.z}; vv[4 * k + 3] = (f32x2){t.w, t.w * t.w};  // reference's `x*x` tensor
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ti = i - r;                                   // window row of output row r (uniform across the workgroup)
            if (ti < 0 || ti >= LN_K) continue;
#pragma unroll
            for (int j = 0; j < LN_K; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) s12[r][q] = s12[r][q] + vv[j + q];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = y0 + ty + r;
        if (y >= h) break;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int x = x0 + tx + q;
            if (x >= w) break;
            const float xv = tile[(ty + r + LN_R) * LN_LW + tx + q + LN_R];
            const float mean = s12[r][q].x / 1089.0f, sq = s12[r][q].y / 1089.0f;
            const float mm = mean * mean;
            const float sd = sqrtf(fabsf(sq - mm)) + 1e-10f;
            const float nv = (xv - mean) / sd;
            out[(size_t)y * w + x] = fminf(fmaxf(nv, -6.0f), 6.0f);
        }
    }
}

// ---- dense layers ------------------------------------------------------------------------------------
struct DenseArgs {
    const float* in;       // conv0: normalised image (h x w); other layers: [CIN/4][Hin][Win] float4 planes
    float* out;            // [COUT/4][Hout][Wout] float4 planes
    const float* W;        // packed weights of this layer
    const float* bias;
    int Hin, Win, Hout, Wout;
    int h, w;              // conv0 only: un-padded image size
    size_t in_stride, out_stride;   // floats between consecutive images of the batch
};

// Epilogue: bias + ReLU and one 16-byte store per tile: lane (n = pixel of the 16-pixel tile, g) owns channels 4g..4g+3 of
// N-tile j = one float4 of plane group (ng * TN + j) * 4 + g.
template <int HOUT, int TM, int TN, bool ADD_BIAS>
__device__ __forceinline__ void store_tiles_dense(float* __restrict__ out, int Hout, int Wout, int Y0, int X0, const f32x4 (&bias)[TN],
                                                  const f32x4 (&acc)[TM][TN], int wave, int lane) {
    constexpr int MT = HOUT * HOUT / 16, MG = MT / TM;
    const int mg = wave % MG, ng = wave / MG;
    const int n = lane & 15, g = lane >> 4;
    const size_t plane = (size_t)Hout * Wout * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = (mg * TM + i) * 16 + n;
        const int oy = p / HOUT, ox = p - oy * HOUT;
        const int Y = Y0 + oy, X = X0 + ox;
        if (Y >= Hout || X >= Wout) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x4 v = acc[i][j];
            if (ADD_BIAS) v += bias[j];
            v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
            *reinterpret_cast<f32x4*>(out + ((size_t)((ng * TN + j) * 4 + g)) * plane + ((size_t)Y * Wout + X) * 4) = v;
        }
    }
}

// conv0: 1 -> 16 on a 32 x 32 tile of the reflect-padded normalised image; zero padding (the conv's own) outside Hp x Wp.
__global__ __launch_bounds__(512, 4) void dense_conv0_kernel(DenseArgs a) {
    constexpr int NW = 8, CB = 16, TM = 8, TN = 1;
    __shared__ __attribute__((aligned(16))) float patch[WP32 * WP32];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* in = a.in + blockIdx.z * a.in_stride;
    float* out = a.out + blockIdx.z * a.out_stride;
    const int X0 = blockIdx.x * 32, Y0 = blockIdx.y * 32;
    float w0[3][TN];
    f32x4 bias0[TN];
    conv0_load_w<NW, CB, TM, TN>(a.W, a.bias, w0, bias0, wave, lane);
    for (int i = tid; i < WP32 * WP32; i += 512) {
        const int ty = i / WP32, tx = i - ty * WP32;
        const int Y = Y0 + ty - 1, X = X0 + tx - 1;                    // coordinates in the padded (Hp x Wp) image
        float v = 0.0f;
        if (Y >= 0 && Y < a.Hin && X >= 0 && X < a.Win) v = in[(size_t)reflect_idx(Y - 14, a.h) * a.w + reflect_idx(X - 14, a.w)];
        patch[i] = v;
    }
    __syncthreads();
    f32x4 acc[TM][TN];
    conv0_mfma<NW, CB, TM, TN>(patch, w0, bias0, acc, wave, lane);      // accumulators start at the bias
    store_tiles_dense<32, TM, TN, false>(out, a.Hout, a.Wout, Y0, X0, bias0, acc, wave, lane);
}

// conv1..5: one workgroup = one LI::H-square INPUT tile -> (LI::H / STRIDE)-square output tile, all COUT channels.
template <int CIN, int COUT, int STRIDE, typename LI, int TM, int TN, int GRP, bool ROLL>
__global__ __launch_bounds__(512, 4) void dense_conv_kernel(DenseArgs a) {
    constexpr int NW = 8, T = LI::H, HOUT = T / STRIDE, NG4 = CIN / 4;
    __shared__ __attribute__((aligned(16))) float act[NG4 * LI::PSG];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* in = a.in + blockIdx.z * a.in_stride;
    float* out = a.out + blockIdx.z * a.out_stride;
    const int X0 = blockIdx.x * T, Y0 = blockIdx.y * T;                 // input-tile origin
    f32x4 b0[ROLL ? 1 : GRP][TN], bias[TN];
    prefetch_b0<NW, COUT, HOUT, TM, TN, (ROLL ? 1 : GRP)>(a.W, b0, wave, lane);
    prefetch_bias<NW, HOUT, TM, TN>(a.bias, bias, wave, lane);
    // stage the tile + 1-px apron: (T + 2)^2 float4 per plane group, rows of consecutive float4 -> coalesced 16-byte loads
    constexpr int TW = T + 2, NPOS = TW * TW;
    const size_t plane = (size_t)a.Hin * a.Win * 4;
    for (int i = tid; i < NG4 * NPOS; i += 512) {
        const int g = i / NPOS, r = i - g * NPOS;
        const int ty = r / TW, tx = r - ty * TW;
        const int Y = Y0 + ty - 1, X = X0 + tx - 1;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (Y >= 0 && Y < a.Hin && X >= 0 && X < a.Win) v = *reinterpret_cast<const f32x4*>(in + g * plane + ((size_t)Y * a.Win + X) * 4);
        *reinterpret_cast<f32x4*>(&act[g * LI::PSG + (ty * LI::WP + tx) * 4]) = v;
    }
    __syncthreads();
    f32x4 acc[TM][TN];
    if constexpr (ROLL) conv3x3_mfma_roll<NW, CIN, COUT, LI, STRIDE, TM, TN>(act, a.W, reinterpret_cast<const f32x4 (&)[1][TN]>(b0), acc, wave, lane);
    else conv3x3_mfma<NW, CIN, COUT, LI, STRIDE, TM, TN, GRP>(act, a.W, b0, acc, wave, lane);
    store_tiles_dense<HOUT, TM, TN, true>(out, a.Hout, a.Wout, Y0 / STRIDE, X0 / STRIDE, bias, acc, wave, lane);
}

// ---- AFFNET_ARITH_FP32_SPLIT3: conv1 .. conv5 of the dense net on split operands -----------------------------------------------------
// Same tiles-through-LDS scheme; the loader splits every fp32 activation ONCE into three bf16 terms while it stages the input tile (+ 1-px
// apron of real neighbours) into the term-interleaved cells of LayQ, then the per-patch trunks' own loop (cnn_mfma.h: conv3x3_mfma_s3q) runs on
// it.  Tiles are LQ::H x LQ::W input pixels (conv1 / conv2: 16 x 32 - the 32 x 32 tile of the exact path would need 111 KB pre-split).
// Activations stay fp32 [C/4][Y][X] float4 planes in HBM between the layers, conv0 and the 8 x 8 head stay fp32 MFMA.
template <int HOUT, int WOUT, int TM, int TN>
__device__ __forceinline__ void store_tiles_dense_rect(float* __restrict__ out, int Hout, int Wout, int Y0, int X0, const f32x4 (&bias)[TN],
                                                       const f32x4 (&acc)[TM][TN], int wave, int lane) {
    constexpr int MT = HOUT * WOUT / 16, MG = MT / TM;
    const int mg = wave % MG, ng = wave / MG;
    const int n = lane & 15, g = lane >> 4;
    const size_t plane = (size_t)Hout * Wout * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = (mg * TM + i) * 16 + n;
        const int oy = p / WOUT, ox = p - oy * WOUT;
        const int Y = Y0 + oy, X = X0 + ox;
        if (Y >= Hout || X >= Wout) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x4 v = acc[i][j] + bias[j];
            v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
            *reinterpret_cast<f32x4*>(out + ((size_t)((ng * TN + j) * 4 + g)) * plane + ((size_t)Y * Wout + X) * 4) = v;
        }
    }
}

template <int CIN, int COUT, int STRIDE, typename LQ, int TM, int TN>
__global__ __launch_bounds__(512, 4) void dense_conv_s3_kernel(DenseArgs a) {
    constexpr int NW = 8, TH = LQ::H, TWD = LQ::W, HOUT = TH / STRIDE, WOUT = TWD / STRIDE, NG4 = CIN / 4;
    static_assert(LQ::C == CIN && LQ::BYTES <= 80 * 1024, "two workgroups per CU");
    __shared__ __attribute__((aligned(16))) char actb[LQ::BYTES];
    float* act = reinterpret_cast<float*>(actb);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* in = a.in + blockIdx.z * a.in_stride;
    float* out = a.out + blockIdx.z * a.out_stride;
    const int X0 = blockIdx.x * TWD, Y0 = blockIdx.y * TH;              // input-tile origin
    S3W<TN> w0;
    f32x4 bias[TN];
    s3_prefetch_w0<NW, CIN, COUT, HOUT * WOUT / 16, TM, TN, LQ::TERMS>(a.W, w0, wave, lane);
    {
        constexpr int MG = (HOUT * WOUT / 16) / TM;                       // rectangular tile: the wave's N-tile group is wave / MG
#pragma unroll
        for (int j = 0; j < TN; ++j) bias[j] = *reinterpret_cast<const f32x4*>(&a.bias[((wave / MG) * TN + j) * 16 + 4 * (lane >> 4)]);
    }
    // stage the tile + 1-px apron pre-split: one float4 = 4 channels of one pixel = half a cell of 8 channels
    constexpr int TWA = TWD + 2, NPOS = (TH + 2) * TWA;
    const size_t plane = (size_t)a.Hin * a.Win * 4;
    for (int i = tid; i < NG4 * NPOS; i += 512) {
        const int g = i / NPOS, r = i - g * NPOS;
        const int ty = r / TWA, tx = r - ty * TWA;
        const int Y = Y0 + ty - 1, X = X0 + tx - 1;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (Y >= 0 && Y < a.Hin && X >= 0 && X < a.Win) v = *reinterpret_cast<const f32x4*>(in + g * plane + ((size_t)Y * a.Win + X) * 4);
        char* dst = actb + (g >> 1) * LQ::GS + LQ::at(ty, tx) + (g & 1) * 8;
        split_store4<LQ::TERMS, LQ::TSTEP>(dst, v);
    }
    __syncthreads();
    f32x4 acc[TM][TN];
    conv3x3_mfma_s3q<NW, CIN, COUT, LQ, STRIDE, TM, TN>(act, a.W, w0, acc, wave, lane, false);
    store_tiles_dense_rect<HOUT, WOUT, TM, TN>(out, a.Hout, a.Wout, Y0 / STRIDE, X0 / STRIDE, bias, acc, wave, lane);
}

// 8 x 8 valid head 64 -> 3 (+ bias) on the matrix cores.  A direct GEMM would use 3 of the 16 MFMA rows; instead the taps of one
// kernel ROW are folded into the N dimension: P[(o, kx)][y][x'] = sum over (ky, c) of W[o][c][ky][kx] * in[c][y + ky][x'] is a
// GEMM with N = 3 * 8 = 24 (two 16-row tiles, 75 % useful), K = 8 * 64 = 512, and out[o][y][x] = bias + sum over kx of
// P[(o, kx)][y][x + kx] is eight shifted adds per output.  One wave = one output row segment: 64 consecutive x' (four 16-pixel
// MFMA tiles) -> 57 outputs; activation fragments are 16-byte loads straight from the [C/4][Y][X] float4 planes (16 consecutive
// pixels = 256 contiguous bytes per channel quad), weights [ky][c/16][kq][n (32)][4] stream from L1 / L2.
#define FH_SEG 57
__global__ __launch_bounds__(256) void fullconv_head_kernel(const float* __restrict__ in, const float* __restrict__ hw, const float* __restrict__ hb,
                                                            float* __restrict__ out, int H4, int W4, int Hf, int Wf, size_t in_stride,
                                                            size_t out_stride) {
    __shared__ float P[4][32 * 64];
    in += blockIdx.z * in_stride;
    out += blockIdx.z * out_stride;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = lane & 15, kq = lane >> 4;
    const int y = min((int)blockIdx.y * 4 + wave, Hf - 1);             // tail rows recompute row Hf - 1 (never stored)
    const int x0 = blockIdx.x * FH_SEG;
    const size_t plane = (size_t)H4 * W4 * 4;
    int xo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xo[i] = min(x0 + i * 16 + m, W4 - 1) * 4;   // columns past the tensor: clamped, their P values are never read
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ky = 0; ky < 8; ++ky) {
        const float* rowp = in + (size_t)(y + ky) * W4 * 4;
#pragma unroll
        for (int G = 0; G < 4; ++G) {
            f32x4 fb[4], fa[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const f32x4*>(rowp + (size_t)(G * 4 + kq) * plane + xo[i]);
#pragma unroll
            for (int j = 0; j < 2; ++j) fa[j] = *reinterpret_cast<const f32x4*>(hw + ((((ky * 4 + G) * 4 + kq) * 32) + j * 16 + m) * 4);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j][s4], fb[i][s4], acc[i][j], 0, 0, 0);
        }
    }
    // acc[i][j][r] = P[n = 16 j + 4 kq + r][pixel 16 i + m]
    float* Pw = P[wave];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Pw[(16 * j + 4 * kq + r) * 64 + 16 * i + m] = acc[i][j][r];
    __syncthreads();
    const int yy = blockIdx.y * 4 + wave, x = x0 + lane;
    if (yy >= Hf || lane >= FH_SEG || x >= Wf) return;
    const size_t o = (size_t)yy * Wf + x, pl = (size_t)Hf * Wf;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float sacc = 0.f;
#pragma unroll
        for (int kx = 0; kx < 8; ++kx) sacc += Pw[(c * 8 + kx) * 64 + lane + kx];
        out[c * pl + o] = sacc + hb[c];
    }
}

// Bilinear upsampling (F.upsample / interpolate, align_corners = False: src = scale * (dst + 0.5) - 0.5 clamped at 0), tanh,
// a0bc composition and rectifyAffineTransformationUpIsUpFullyConv (LAF.py:293-297) -> (4, h, w).
__global__ __launch_bounds__(256) void fullconv_finish_kernel(const float* __restrict__ ff, float* __restrict__ out, int Hf, int Wf, int h, int w,
                                                              size_t in_stride, size_t out_stride) {
    ff += blockIdx.z * in_stride;
    out += blockIdx.z * out_stride;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float sh = (float)Hf / (float)h, sw = (float)Wf / (float)w;
    float fy = sh * ((float)y + 0.5f) - 0.5f, fx = sw * ((float)x + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
    int y0 = (int)fy, x0 = (int)fx;
    y0 = y0 > Hf - 1 ? Hf - 1 : y0; x0 = x0 > Wf - 1 ? Wf - 1 : x0;
    const int y1 = y0 + (y0 < Hf - 1 ? 1 : 0), x1 = x0 + (x0 < Wf - 1 ? 1 : 0);
    const float ly1 = fminf(fmaxf(fy - (float)y0, 0.f), 1.f), lx1 = fminf(fmaxf(fx - (float)x0, 0.f), 1.f);
    const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const size_t pl = (size_t)Hf * Wf;
    float t[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* p = ff + c * pl;
        const float top = lx0 * p[(size_t)y0 * Wf + x0] + lx1 * p[(size_t)y0 * Wf + x1];
        const float bot = lx0 * p[(size_t)y1 * Wf + x0] + lx1 * p[(size_t)y1 * Wf + x1];
        t[c] = tanhf(ly0 * top + ly1 * bot);
    }
    const float a00 = 1.0f + t[0], a01 = 0.0f * t[1], a10 = t[1], a11 = 1.0f + t[2];
    const float det = sqrtf(fabsf(a00 * a11 - a01 * a10 + 1e-10f));
    const float b2a2 = sqrtf(a01 * a01 + a00 * a00);
    const size_t o = (size_t)y * w + x, P = (size_t)h * w;
    out[o] = b2a2 / det;
    out[P + o] = 0.0f * det;
    out[2 * P + o] = (a11 * a01 + a10 * a00) / (b2a2 * det);
    out[3 * P + o] = det / b2a2;
}

// NMS2d (HandCraftedModules.py:194-206): keep x where x - max3x3 + 1e-5 > 0 (-inf padding), optionally x > th.
__global__ __launch_bounds__(256) void nms2d_kernel(const float* __restrict__ in, float* __restrict__ out, int h, int w, float th) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    float m = -INFINITY;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) m = fmaxf(m, in[(size_t)yy * w + xx]);
        }
    const float v = in[(size_t)y * w + x];
    float r;
    if (th > 1e-5f) r = v * ((v > th) ? 1.0f : 0.0f) * ((((v + 1e-5f) - m) > 0.0f) ? 1.0f : 0.0f);
    else r = ((((v - m) + 1e-5f) > 0.0f) ? 1.0f : 0.0f) * v;
    out[(size_t)y * w + x] = r;
}

// ---- host side ---------------------------------------------------------------------------------------
// B images of one size: image b reads img + b * img_stride, writes out + b * out_stride (4 * h * w floats each) and uses
// scratch + b * scratch_stride (dense_scratch_floats(h, w) floats each).
int aff_fullconv_launch(affnet_ctx* ctx, const float* packed, const float* img, size_t img_stride, int h, int w, float* out, size_t out_stride,
                        float* scratch, size_t scratch_stride, int B, hipStream_t st) {
    if (h < 34 || w < 34)
        return aff_fail(ctx, AFFNET_ERR_INVALID, "fullconv: image %dx%d too small (LocalNorm2d(33) reflect-pads by 16; the reference raises as well)", w, h);
    const DenseGeom g = dense_geom(h, w);
    if (g.Hf < 1 || g.Wf < 1) return aff_fail(ctx, AFFNET_ERR_INVALID, "fullconv: image %dx%d too small for the 8x8 head", w, h);
    const NetLayout L = net_layout(AFFNET_NET_AFFNET_FULLCONV);
    const size_t norm_f = aff_align((size_t)h * w * sizeof(float)) / sizeof(float);
    const size_t act_f = aff_align((size_t)16 * g.Hp * g.Wp * sizeof(float)) / sizeof(float);
    float* norm = scratch;
    float* bufA = scratch + norm_f;
    float* bufB = bufA + act_f;
    hipLaunchKernelGGL(local_norm_kernel, dim3(aff_cdiv(w, LN_T), aff_cdiv(h, LN_T), B), dim3(256), 0, st, img, norm, h, w, img_stride, scratch_stride);
    AFF_LAUNCH_CHECK(ctx);
    DenseArgs a;
    a.in_stride = scratch_stride; a.out_stride = scratch_stride; a.h = h; a.w = w;
    auto layer = [&](int i, const float* in, float* o, int Hin, int Win, int Hout, int Wout) {
        a.in = in; a.out = o; a.W = packed + L.w_off[i]; a.bias = packed + L.b_off[i];
        a.Hin = Hin; a.Win = Win; a.Hout = Hout; a.Wout = Wout;
    };
    layer(0, norm, bufA, g.Hp, g.Wp, g.Hp, g.Wp);
    hipLaunchKernelGGL(dense_conv0_kernel, dim3(aff_cdiv(g.Wp, 32), aff_cdiv(g.Hp, 32), B), dim3(512), 0, st, a);
    AFF_LAUNCH_CHECK(ctx);
    if (ctx->arith == AFFNET_ARITH_FP32_SPLIT3 || ctx->arith == AFFNET_ARITH_FP32_SPLIT2H) {
        // conv1 .. conv5 on split operands: three bf16 terms per fp32 operand and six bf16 MFMAs per product, or (SPLIT2H) two fp16 terms and three
        // fp16 MFMAs; fp32 accumulate
        const bool h2 = ctx->arith == AFFNET_ARITH_FP32_SPLIT2H;
        auto layer3 = [&](int i, const float* in, float* o, int Hin, int Win, int Hout, int Wout) {
            layer(i, in, o, Hin, Win, Hout, Wout);
            a.W = packed + (h2 ? L.w_h2[i] : L.w_s3[i]);
        };
        // (register blockings as in the per-patch split trunks: one channel tile per wave for conv3 / conv4, tools/probes/s3_loop_probe)
        // Q1: conv1 input tile, 16 rows x 32 columns, 16 channels (59 KB); Q2: conv2 (stride 2); Q3: conv3 / conv4 input tiles (61 KB); Q5: conv5 (61 KB)
#define DENSE_S3(CI, CO, STR, H_, W_, WP_, GREM, WPR, GREMR, TM_, TN_, GX, GY)   /* LayQ (three terms): WP_, GREM; LayR (two terms): WPR, GREMR - cnn_mfma.h */             \
        do {                                                                                                                                     \
            if (h2) hipLaunchKernelGGL((dense_conv_s3_kernel<CI, CO, STR, LayR<H_, W_, WPR, CI, GREMR>, TM_, TN_>), dim3(GX, GY, B), dim3(512), 0, st, a);   \
            else hipLaunchKernelGGL((dense_conv_s3_kernel<CI, CO, STR, LayQ<H_, W_, WP_, CI, GREM, 3>, TM_, TN_>), dim3(GX, GY, B), dim3(512), 0, st, a);  \
            AFF_LAUNCH_CHECK(ctx);                                                                                                               \
        } while (0)
        layer3(1, bufA, bufB, g.Hp, g.Wp, g.Hp, g.Wp);
        DENSE_S3(16, 16, 1, 16, 32, 34, 0, 34, 0, 4, 1, aff_cdiv(g.Wp, 32), aff_cdiv(g.Hp, 16));
        layer3(2, bufB, bufA, g.Hp, g.Wp, g.H2, g.W2);
        DENSE_S3(16, 32, 2, 16, 32, 34, 16, 34, 16, 2, 1, aff_cdiv(g.Wp, 32), aff_cdiv(g.Hp, 16));
        layer3(3, bufA, bufB, g.H2, g.W2, g.H2, g.W2);
        DENSE_S3(32, 32, 1, 16, 16, 18, 0, 20, 0, 4, 1, aff_cdiv(g.W2, 16), aff_cdiv(g.H2, 16));
        layer3(4, bufB, bufA, g.H2, g.W2, g.H4, g.W4);
        DENSE_S3(32, 64, 2, 16, 16, 18, 0, 20, 16, 2, 1, aff_cdiv(g.W2, 16), aff_cdiv(g.H2, 16));
        layer3(5, bufA, bufB, g.H4, g.W4, g.H4, g.W4);
        DENSE_S3(64, 64, 1, 8, 8, 16, 128, 12, 0, 2, 1, aff_cdiv(g.W4, 8), aff_cdiv(g.H4, 8));
#undef DENSE_S3
    } else {
    layer(1, bufA, bufB, g.Hp, g.Wp, g.Hp, g.Wp);
    hipLaunchKernelGGL((dense_conv_kernel<16, 16, 1, LayC0, 8, 1, 1, true>), dim3(aff_cdiv(g.Wp, 32), aff_cdiv(g.Hp, 32), B), dim3(512), 0, st, a);
    AFF_LAUNCH_CHECK(ctx);
    layer(2, bufB, bufA, g.Hp, g.Wp, g.H2, g.W2);
    hipLaunchKernelGGL((dense_conv_kernel<16, 32, 2, LayC1, 4, 1, 1, false>), dim3(aff_cdiv(g.Wp, 32), aff_cdiv(g.Hp, 32), B), dim3(512), 0, st, a);
    AFF_LAUNCH_CHECK(ctx);
    layer(3, bufA, bufB, g.H2, g.W2, g.H2, g.W2);
    hipLaunchKernelGGL((dense_conv_kernel<32, 32, 1, LayC2, 4, 1, 1, false>), dim3(aff_cdiv(g.W2, 16), aff_cdiv(g.H2, 16), B), dim3(512), 0, st, a);
    AFF_LAUNCH_CHECK(ctx);
    layer(4, bufB, bufA, g.H2, g.W2, g.H4, g.W4);
    hipLaunchKernelGGL((dense_conv_kernel<32, 64, 2, LayC3, 2, 1, 2, false>), dim3(aff_cdiv(g.W2, 16), aff_cdiv(g.H2, 16), B), dim3(512), 0, st, a);
    AFF_LAUNCH_CHECK(ctx);
    layer(5, bufA, bufB, g.H4, g.W4, g.H4, g.W4);
    hipLaunchKernelGGL((dense_conv_kernel<64, 64, 1, LayC4, 2, 1, 2, false>), dim3(aff_cdiv(g.W4, 8), aff_cdiv(g.H4, 8), B), dim3(512), 0, st, a);
    AFF_LAUNCH_CHECK(ctx);
    }
    hipLaunchKernelGGL(fullconv_head_kernel, dim3(aff_cdiv(g.Wf, FH_SEG), aff_cdiv(g.Hf, 4), B), dim3(256), 0, st, bufB, packed + L.head_w, packed + L.head_b,
                       bufA, g.H4, g.W4, g.Hf, g.Wf, scratch_stride, scratch_stride);
    AFF_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(fullconv_finish_kernel, dim3(aff_cdiv(w, 64), aff_cdiv(h, 4), B), dim3(256), 0, st, bufA, out, g.Hf, g.Wf, h, w, scratch_stride,
                       out_stride);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

extern "C" int affnet_local_norm(affnet_ctx* ctx, const float* d_in, float* d_out, int h, int w, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !d_in || !d_out) return aff_fail(ctx, AFFNET_ERR_INVALID, "local_norm: null argument");
    if (h < 17 || w < 17) return aff_fail(ctx, AFFNET_ERR_INVALID, "local_norm: image %dx%d too small for a reflect padding of 16", w, h);
    hipLaunchKernelGGL(local_norm_kernel, dim3(aff_cdiv(w, LN_T), aff_cdiv(h, LN_T), 1), dim3(256), 0, (hipStream_t)stream, d_in, d_out, h, w, (size_t)0,
                       (size_t)0);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}

extern "C" int affnet_fullconv_forward(affnet_ctx* ctx, const float* d_packed, const float* d_img, int h, int w, float* d_out, float* d_scratch,
                                       void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !d_packed || !d_img || !d_out || !d_scratch) return aff_fail(ctx, AFFNET_ERR_INVALID, "fullconv_forward: null argument");
    return aff_fullconv_launch(ctx, d_packed, d_img, 0, h, w, d_out, 0, d_scratch, 0, 1, (hipStream_t)stream);
}

extern "C" int affnet_nms2d(affnet_ctx* ctx, const float* d_in, float* d_out, int h, int w, float threshold, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !d_in || !d_out || h < 1 || w < 1) return aff_fail(ctx, AFFNET_ERR_INVALID, "nms2d: bad argument");
    hipLaunchKernelGGL(nms2d_kernel, dim3(aff_cdiv(w, 64), aff_cdiv(h, 4)), dim3(256), 0, (hipStream_t)stream, d_in, d_out, h, w, threshold);
    AFF_LAUNCH_CHECK(ctx);
    return AFFNET_OK;
}
