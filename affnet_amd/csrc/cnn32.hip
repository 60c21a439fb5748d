// 32x32-patch CNNs (AffNetFast / OriNetFast / HardNet) for gfx950 on the fp32 matrix cores.
//
// Replaces architectures.py:204-252 (AffNetFast), :33-82 (OriNetFast), HardNet.py:61-101
// (HardNet) incl. input_norm, eval-mode BatchNorm (folded into weights + bias at pack time),
// ReLU, the heads, rectifyAffineTransformationUpIsUp (LAF.py:285-291), get_rotation_matrix
// (LAF.py:276-283) and L2Norm (HardNet.py:12-19).
//
// Design (one workgroup = 8 wavefronts = one patch, whole trunk resident on the CU):
//   * the patch is sampled (or loaded), standardised (mean / unbiased std + 1e-7) and conv0
//     (K = 9, VALU) writes its planes into ONE LDS activation buffer, planar [c][H+2][W+2] with a
//     zero halo (so the 3x3 taps are plain address offsets and padding costs nothing);
//   * conv1..conv5 are implicit GEMMs on v_mfma_f32_16x16x4_f32 (exact fp32, 32 cycles):
//     M = output pixels (16 consecutive pixels per tile -> conflict-free ds_read_b32 A fragments:
//     lane = (pixel, k) with k = 4 consecutive input channels one plane apart), N = 16 output
//     channels per tile, K = (tap, cin).  B fragments stream from the packed weights in L2
//     ([k][n], n fastest -> coalesced 64-byte segments).  The 8 waves split the M x N tile grid
//     (TM x TN register blocking per wave);
//   * a layer's complete output lives in the accumulators (<= 64 VGPR/lane) until every wave has
//     finished reading the input; then bias + ReLU are applied and the planes are written back IN
//     PLACE over the input.  One activation buffer (<= 148 KB of the 160 KB LDS), no HBM traffic
//     between layers: per patch the kernel reads 4 KB (or samples the pyramid) + the L2-resident
//     weights and writes 16 B (AffNet/OriNet) or the 32 KB conv5 tensor (HardNet);
//   * AffNet / OriNet heads (K = 4096, N <= 3) run on the VALU in the same kernel; the HardNet
//     head (8192 x 128, 4.2 MB of weights) is a separate GEMM over all patches so the weights are
//     read once per 16 patches instead of once per patch, with BN + L2 normalisation fused.
#include <math.h>
#include <stdlib.h>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Plane strides (floats) of the padded 34x34 / 18x18 / 10x10 planes, all == 17 (mod 32): the 4 k-planes of
// an A fragment (lanes 16..31 read the plane after lanes 0..15) land 17 banks apart -> at most 1 of 16 banks
// collides inside a 32-lane half (PMC: SQ_LDS_BANK_CONFLICT was 47% of LDS cycles with stride == 5 mod 32),
// the stride-2 layers are conflict free (even vs odd banks), and the epilogue stores (16 channels of one
// pixel per 16 lanes) hit 16 distinct banks because 17 is odd.
#define PST32 1169
#define PST16 337
#define PST8 113
#define WP32 34
#define WP16 18
#define WP8 10
#define HEAD_K 8192

// ---- packed weight layout --------------------------------------------------------------------------
struct NetLayout {
    int cb;                 // base width: 16 (AffNet/OriNet) or 32 (HardNet)
    int cin[6], cout[6];
    size_t w_off[6], b_off[6];
    size_t head_w, head_b;  // head weights / bias (HardNet: BN-folded [8192][128] + bias[128])
    size_t total;
};

static NetLayout net_layout(int kind) {
    NetLayout L;
    L.cb = (kind == AFFNET_NET_HARDNET) ? 32 : 16;
    const int ch[7] = {1, L.cb, L.cb, 2 * L.cb, 2 * L.cb, 4 * L.cb, 4 * L.cb};
    size_t off = 0;
    for (int i = 0; i < 6; ++i) {
        L.cin[i] = ch[i]; L.cout[i] = ch[i + 1];
        L.w_off[i] = off; off += (size_t)9 * ch[i] * ch[i + 1];
        L.b_off[i] = off; off += ch[i + 1];
        off = (off + 3) & ~(size_t)3;
    }
    L.head_w = off;
    if (kind == AFFNET_NET_AFFNET) { off += 3 * 4096; L.head_b = off; off += 4; }
    else if (kind == AFFNET_NET_ORINET) { off += 2 * 4096; L.head_b = off; off += 4; }
    else { off += (size_t)HEAD_K * 128; L.head_b = off; off += 128; }
    L.total = off;
    return L;
}

extern "C" size_t affnet_cnn32_packed_floats(int net_kind) {
    if (net_kind < 0 || net_kind > 2) return 0;
    return net_layout(net_kind).total;
}

extern "C" int affnet_cnn32_pack_weights(int kind, const float* const* conv_w, const float* const* bn_mean, const float* const* bn_var,
                                         const float* head_w, const float* head_b, const float* head_bn_mean, const float* head_bn_var,
                                         float* out) {
    if (kind < 0 || kind > 2 || !conv_w || !bn_mean || !bn_var || !head_w || !out) return AFFNET_ERR_INVALID;
    const NetLayout L = net_layout(kind);
    memset(out, 0, L.total * sizeof(float));
    for (int i = 0; i < 6; ++i) {
        const int ci = L.cin[i], co = L.cout[i];
        for (int n = 0; n < co; ++n) {
            const float s = 1.0f / sqrtf(bn_var[i][n] + 1e-5f);      // BatchNorm2d(affine=False), eps 1e-5, eval mode
            out[L.b_off[i] + n] = -bn_mean[i][n] * s;
            for (int c = 0; c < ci; ++c)
                for (int t = 0; t < 9; ++t) {
                    const float w = conv_w[i][((size_t)n * ci + c) * 9 + t] * s;
                    if (i == 0) out[L.w_off[0] + (size_t)n * 9 + t] = w;                 // [n][tap]
                    else out[L.w_off[i] + ((size_t)t * ci + c) * co + n] = w;           // [tap][cin][n]
                }
        }
    }
    if (kind == AFFNET_NET_HARDNET) {
        if (!head_bn_mean || !head_bn_var) return AFFNET_ERR_INVALID;
        for (int n = 0; n < 128; ++n) {
            const float s = 1.0f / sqrtf(head_bn_var[n] + 1e-5f);
            out[L.head_b + n] = -head_bn_mean[n] * s;
            for (int k = 0; k < HEAD_K; ++k) out[L.head_w + (size_t)k * 128 + n] = head_w[(size_t)n * HEAD_K + k] * s;
        }
    } else {
        const int no = kind == AFFNET_NET_AFFNET ? 3 : 2;
        if (!head_b) return AFFNET_ERR_INVALID;
        memcpy(out + L.head_w, head_w, (size_t)no * 4096 * sizeof(float));
        memcpy(out + L.head_b, head_b, no * sizeof(float));
    }
    return AFFNET_OK;
}

struct NetOffsets {        // device-side copy of the offsets (by-value kernel argument)
    int w[6], b[6], head_w, head_b;
};

static NetOffsets to_offsets(const NetLayout& L) {
    NetOffsets o;
    for (int i = 0; i < 6; ++i) { o.w[i] = (int)L.w_off[i]; o.b[i] = (int)L.b_off[i]; }
    o.head_w = (int)L.head_w; o.head_b = (int)L.head_b;
    return o;
}

// ---- device helpers ------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum `v` over the NW wavefronts of the workgroup; red must hold >= NW floats.  Two barriers.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += red[w];
    return t;
}

// Zero the 1-pixel halo of `cout` planes of an (H+2)x(H+2) padded layout.
template <int H, int PST, int NTHR>
__device__ __forceinline__ void zero_halo(float* act, int cout) {
    constexpr int WP = H + 2;
    constexpr int CELLS = 4 * (H + 1);
    for (int i = threadIdx.x; i < cout * CELLS; i += NTHR) {
        const int c = i / CELLS, e = i - c * CELLS;
        int y, x;
        if (e < WP) { y = 0; x = e; }
        else if (e < 2 * WP) { y = H + 1; x = e - WP; }
        else { const int r = e - 2 * WP; y = 1 + (r >> 1); x = (r & 1) ? H + 1 : 0; }
        act[c * PST + y * WP + x] = 0.0f;
    }
}

// Implicit-GEMM 3x3 convolution (padding 1) of the planar LDS tensor `act` ([CIN][HIN+2][HIN+2],
// plane stride PSI) with packed weights Wg [9*CIN][COUT]; leaves the TM x TN tiles of this wave in
// `acc` (pre-activation, no bias).  HOUT = HIN / STRIDE.
template <int NW, int CIN, int COUT, int HOUT, int STRIDE, int TM, int TN, int PSI, int WPI, int UNROLL>
__device__ __forceinline__ void conv3x3_mfma(const float* act, const float* __restrict__ Wg, f32x4 (&acc)[TM][TN], int wave, int lane) {
    constexpr int MT = HOUT * HOUT / 16, NT = COUT / 16;
    constexpr int MG = MT / TM, NG = NT / TN;
    static_assert(MG * NG == NW, "the waves must tile the layer exactly");
    static_assert(CIN % 4 == 0 && (CIN / 4) % UNROLL == 0, "bad unroll");
    const int mg = wave % MG, ng = wave / MG;
    const int m = lane & 15, kq = lane >> 4;
    int a_base[TM], b_base[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = (mg * TM + i) * 16 + m;
        const int oy = p / HOUT, ox = p - oy * HOUT;
        a_base[i] = kq * PSI + (oy * STRIDE) * WPI + ox * STRIDE;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) b_base[j] = kq * COUT + (ng * TN + j) * 16 + m;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // K = (tap, cin) is walked in chunks of 4*UNROLL; a chunk never straddles a tap (CIN % (4*UNROLL) == 0).
    // The B fragments (global / L2, ~1 us latency under load) of chunk c+1 are requested BEFORE the MFMAs of
    // chunk c and consumed one iteration later; A fragments come from LDS right before use.  (Also double
    // buffering A in registers was measured SLOWER: hipcc turns the ping-pong into 64 v_mov per chunk pair and
    // waits for the new loads at the end of the chunk.)
    constexpr int KSTEP = 4 * UNROLL;
    constexpr int NCHUNK = 9 * CIN / KSTEP;
    float bn[UNROLL][TN];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
        for (int j = 0; j < TN; ++j) bn[u][j] = Wg[(4 * u) * COUT + b_base[j]];
#pragma unroll 1
    for (int ch = 0; ch < NCHUNK; ++ch) {
        const int k0 = ch * KSTEP;
        const int tap = k0 / CIN, c0 = k0 - tap * CIN;
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;          // tap / 3, tap % 3 for tap < 9
        const int a_off = c0 * PSI + ky * WPI + kx;
        float a[UNROLL][TM], b[UNROLL][TN];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int j = 0; j < TN; ++j) b[u][j] = bn[u][j];
        const int kn = (ch + 1 < NCHUNK) ? k0 + KSTEP : k0;         // last chunk re-reads itself (in bounds)
        const float* wn = Wg + (size_t)kn * COUT;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int j = 0; j < TN; ++j) bn[u][j] = wn[(4 * u) * COUT + b_base[j]];
        __builtin_amdgcn_sched_barrier(0);                           // keep the prefetch ahead of this chunk's MFMAs
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int i = 0; i < TM; ++i) a[u][i] = act[(4 * u) * PSI + a_off + a_base[i]];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
    }
}

// Epilogue: bias + ReLU, write the wave's tiles into the padded planar LDS layout of the NEXT layer.
template <int COUT, int HOUT, int TM, int TN, int PSO>
__device__ __forceinline__ void store_tiles_lds(float* act, const float* __restrict__ bias, const f32x4 (&acc)[TM][TN], int wave, int lane) {
    constexpr int MT = HOUT * HOUT / 16, MG = MT / TM;
    constexpr int WPO = HOUT + 2;
    const int mg = wave % MG, ng = wave / MG;
    const int n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ch = (ng * TN + j) * 16 + n;
        const float bv = bias[ch];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = (mg * TM + i) * 16 + 4 * g + r;
                const int oy = p / HOUT, ox = p - oy * HOUT;
                act[ch * PSO + (oy + 1) * WPO + ox + 1] = fmaxf(acc[i][j][r] + bv, 0.0f);
            }
        }
    }
}

// Same for the last trunk layer of HardNet: [c][8][8] flattened = the head GEMM's K order.
template <int COUT, int TM, int TN>
__device__ __forceinline__ void store_tiles_global(float* __restrict__ dst, const float* __restrict__ bias, const f32x4 (&acc)[TM][TN],
                                                   int wave, int lane) {
    constexpr int MT = 4, MG = MT / TM;
    const int mg = wave % MG, ng = wave / MG;
    const int n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ch = (ng * TN + j) * 16 + n;
        const float bv = bias[ch];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int p = (mg * TM + i) * 16 + 4 * g;
            float4 v;
            v.x = fmaxf(acc[i][j][0] + bv, 0.f); v.y = fmaxf(acc[i][j][1] + bv, 0.f);
            v.z = fmaxf(acc[i][j][2] + bv, 0.f); v.w = fmaxf(acc[i][j][3] + bv, 0.f);
            *reinterpret_cast<float4*>(dst + ch * 64 + p) = v;
        }
    }
}

struct PyrSrc {            // pyramid sampling source (fused sampler)
    const float* lvl[AFFNET_MAX_OCTAVES][AFFNET_MAX_LEVELS];
    int h[AFFNET_MAX_OCTAVES], w[AFFNET_MAX_OCTAVES];
    int n_octaves, n_levels;
    size_t img_stride;     // floats between the pyramids of consecutive images (batch)
    float base[32];        // affine_grid base coordinates for PS = 32
};

struct CnnArgs {
    const float* packed;
    NetOffsets off;
    const float* patches;  // (n,32,32) or NULL -> sample from the pyramid
    const float* lafs;     // normalised LAFs when sampling
    const int32_t* ids;    // (octave, level, *) when sampling
    const int32_t* count;
    int n_max;
    float* out;            // AffNet/OriNet: (n,2,2); HardNet: trunk output (n,8192)
    int dbg_layer;         // >= 0: dump activations after this trunk layer of patch 0 and exit
    float* dbg_out;
    unsigned long long* dbg_time;   // != NULL: s_memtime stamps [patch][wave][16] at the phase boundaries (tuning aid)
};

#define CNN_STAMP(k)                                                                                     \
    do {                                                                                                 \
        if (a.dbg_time && lane == 0) a.dbg_time[((size_t)pidx * NW + wave) * 16 + (k)] = __builtin_readcyclecounter(); \
    } while (0)

template <int CB>
struct TrunkLds {
    static constexpr int ACT = CB * PST32;
    static constexpr int PATCH = WP32 * WP32;
    static constexpr int W0 = 12 * 32;                  // conv0 taps + bias, 12 floats per output channel
    static constexpr int TOTAL = ACT + PATCH + 64 + W0;
};

template <int C, int H, int PST, int NTHR>
__device__ __forceinline__ void dump_planes(const float* act, float* dst) {
    constexpr int WP = H + 2;
    for (int i = threadIdx.x; i < C * H * H; i += NTHR) {
        const int c = i / (H * H), r = i - c * H * H, y = r / H, x = r - y * H;
        dst[i] = act[c * PST + (y + 1) * WP + x + 1];
    }
}

// KIND: 0 AffNet, 1 OriNet, 2 HardNet (CB = 16 / 16 / 32).  NW = wavefronts per workgroup (8 or 16).
// AffNet / OriNet: 8 waves, 79 KB LDS -> 2 workgroups per CU (4 waves / SIMD).  HardNet needs 154 KB LDS
// (1 workgroup per CU): 16 waves give the CU 4 waves / SIMD to cover LDS / L2 latency and barriers.
template <int KIND, int NW>
__global__ __launch_bounds__(NW * 64, (KIND == AFFNET_NET_HARDNET) ? NW / 4 : 4) void cnn32_trunk_kernel(CnnArgs a, PyrSrc ps) {
    constexpr int CB = (KIND == AFFNET_NET_HARDNET) ? 32 : 16;
    constexpr int NTHR = NW * 64;
    constexpr int PPT = 1024 / NTHR;                    // input pixels per thread (2 or 1)
    constexpr int RPT = 32 / PPT;                       // patch rows covered by one pass of the workgroup
    // per-wave register blocking (TM x TN tiles of 16 px x 16 ch); MG * NG == NW for every layer
    constexpr int T1M = (CB == 16) ? 8 : 64 / NW, T1N = CB / 16;
    constexpr int T2M = (CB == 16) ? 2 : 32 / NW, T2N = 2;
    constexpr int T4M = (CB == 16) ? 2 : 32 / NW, T4N = 1;
    __shared__ __attribute__((aligned(16))) float lds[TrunkLds<CB>::TOTAL];
    float* act = lds;
    float* patch = lds + TrunkLds<CB>::ACT;
    float* red = patch + TrunkLds<CB>::PATCH;
    float* w0s = red + 64;                              // [CB][12]: 9 taps, bias, 2 pad (16-byte rows)
    // grid = (n_max, batch): row blockIdx.x of image blockIdx.y; global row = image * n_max + row
    const int n = a.count ? min(a.count[blockIdx.y], a.n_max) : a.n_max;
    if ((int)blockIdx.x >= n) return;
    const size_t pidx = (size_t)blockIdx.y * a.n_max + blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    CNN_STAMP(0);

    // ---- input: load or sample 1024 pixels (PPT per thread), standardise, store padded ----------------
    float v[PPT];
    if (a.patches) {
        const float* src = a.patches + pidx * 1024;
#pragma unroll
        for (int q = 0; q < PPT; ++q) v[q] = src[tid + q * NTHR];
    } else {
        int o = a.ids[3 * pidx], l = a.ids[3 * pidx + 1];
        o = o < 0 ? 0 : (o >= ps.n_octaves ? ps.n_octaves - 1 : o);
        l = l < 0 ? 0 : (l >= ps.n_levels ? ps.n_levels - 1 : l);
        const float* img = ps.lvl[o][l] + blockIdx.y * ps.img_stride;
        const int h = ps.h[o], w = ps.w[o];
        const float* L = a.lafs + 6 * pidx;
        const float m = (float)(h < w ? h : w);
        const float t00 = L[0] * m, t01 = L[1] * m, t02 = L[2] * (float)w;
        const float t10 = L[3] * m, t11 = L[4] * m, t12 = L[5] * (float)h;
#pragma unroll
        for (int q = 0; q < PPT; ++q)
            v[q] = aff_sample_bilinear(img, h, w, t00, t01, t02, t10, t11, t12, ps.base[tid & 31], ps.base[(tid >> 5) + q * RPT]);
    }
    for (int i = tid; i < WP32 * WP32; i += NTHR) patch[i] = 0.0f;   // halo (interior overwritten below)
    // conv0 taps + bias -> LDS once per workgroup (as uniform scalar loads inside the channel loop they cost
    // 21-34k cycles per patch: a chain of dependent L2 round trips)
    for (int i = tid; i < CB * 12; i += NTHR) {
        const int c = i / 12, t = i - c * 12;
        w0s[i] = t < 9 ? a.packed[a.off.w[0] + c * 9 + t] : (t == 9 ? a.packed[a.off.b[0] + c] : 0.0f);
    }
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < PPT; ++q) sum += v[q];
    const float mean = block_sum<NW>(sum, red) * (1.0f / 1024.0f);
    float sq = 0.f;
#pragma unroll
    for (int q = 0; q < PPT; ++q) { v[q] -= mean; sq += v[q] * v[q]; }
    const float var = block_sum<NW>(sq, red) * (1.0f / 1023.0f);       // torch.std: unbiased
    const float sd = sqrtf(var) + 1e-7f;
#pragma unroll
    for (int q = 0; q < PPT; ++q) patch[((tid >> 5) + q * RPT + 1) * WP32 + (tid & 31) + 1] = v[q] / sd;
    zero_halo<32, PST32, NTHR>(act, CB);
    __syncthreads();
    CNN_STAMP(1);

    // ---- conv0: 1 -> CB, K = 9, VALU ---------------------------------------------------------------
    {
        float q[PPT][9];
#pragma unroll
        for (int qq = 0; qq < PPT; ++qq) {
            const int y = (tid >> 5) + qq * RPT, x = tid & 31;
#pragma unroll
            for (int t = 0; t < 9; ++t) q[qq][t] = patch[(y + t / 3) * WP32 + x + t % 3];
        }
#pragma unroll 4
        for (int c = 0; c < CB; ++c) {
            const float4 wa = *reinterpret_cast<const float4*>(&w0s[c * 12]);        // broadcast reads
            const float4 wb = *reinterpret_cast<const float4*>(&w0s[c * 12 + 4]);
            const float4 wc = *reinterpret_cast<const float4*>(&w0s[c * 12 + 8]);
            const float wt[9] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w, wc.x};
#pragma unroll
            for (int qq = 0; qq < PPT; ++qq) {
                const int y = (tid >> 5) + qq * RPT, x = tid & 31;
                float sacc = wc.y;                                                   // bias
#pragma unroll
                for (int t = 0; t < 9; ++t) sacc = fmaf(q[qq][t], wt[t], sacc);
                act[c * PST32 + (y + 1) * WP32 + x + 1] = fmaxf(sacc, 0.0f);
            }
        }
    }
    __syncthreads();
    if (a.dbg_layer == 0) { dump_planes<CB, 32, PST32, NTHR>(act, a.dbg_out); return; }
    CNN_STAMP(2);

    // ---- conv1: CB -> CB @32x32 --------------------------------------------------------------------
    {
        f32x4 acc[T1M][T1N];
        conv3x3_mfma<NW, CB, CB, 32, 1, T1M, T1N, PST32, WP32, 4>(act, a.packed + a.off.w[1], acc, wave, lane);
        CNN_STAMP(3);
        __syncthreads();
        store_tiles_lds<CB, 32, T1M, T1N, PST32>(act, a.packed + a.off.b[1], acc, wave, lane);   // halo already zero
        __syncthreads();
        CNN_STAMP(4);
    }
    if (a.dbg_layer == 1) { dump_planes<CB, 32, PST32, NTHR>(act, a.dbg_out); return; }

    // ---- conv2: CB -> 2CB, stride 2 @16x16 -----------------------------------------------------------
    {
        f32x4 acc[T2M][T2N];
        conv3x3_mfma<NW, CB, 2 * CB, 16, 2, T2M, T2N, PST32, WP32, 4>(act, a.packed + a.off.w[2], acc, wave, lane);
        CNN_STAMP(5);
        __syncthreads();
        zero_halo<16, PST16, NTHR>(act, 2 * CB);
        store_tiles_lds<2 * CB, 16, T2M, T2N, PST16>(act, a.packed + a.off.b[2], acc, wave, lane);
        __syncthreads();
        CNN_STAMP(6);
    }
    if (a.dbg_layer == 2) { dump_planes<2 * CB, 16, PST16, NTHR>(act, a.dbg_out); return; }

    // ---- conv3: 2CB -> 2CB @16x16 --------------------------------------------------------------------
    {
        f32x4 acc[T2M][T2N];
        conv3x3_mfma<NW, 2 * CB, 2 * CB, 16, 1, T2M, T2N, PST16, WP16, 8>(act, a.packed + a.off.w[3], acc, wave, lane);
        CNN_STAMP(7);
        __syncthreads();
        store_tiles_lds<2 * CB, 16, T2M, T2N, PST16>(act, a.packed + a.off.b[3], acc, wave, lane);
        __syncthreads();
        CNN_STAMP(8);
    }
    if (a.dbg_layer == 3) { dump_planes<2 * CB, 16, PST16, NTHR>(act, a.dbg_out); return; }

    // ---- conv4: 2CB -> 4CB, stride 2 @8x8 --------------------------------------------------------------
    {
        f32x4 acc[T4M][T4N];
        conv3x3_mfma<NW, 2 * CB, 4 * CB, 8, 2, T4M, T4N, PST16, WP16, 8>(act, a.packed + a.off.w[4], acc, wave, lane);
        CNN_STAMP(9);
        __syncthreads();
        zero_halo<8, PST8, NTHR>(act, 4 * CB);
        store_tiles_lds<4 * CB, 8, T4M, T4N, PST8>(act, a.packed + a.off.b[4], acc, wave, lane);
        __syncthreads();
        CNN_STAMP(10);
    }
    if (a.dbg_layer == 4) { dump_planes<4 * CB, 8, PST8, NTHR>(act, a.dbg_out); return; }

    // ---- conv5: 4CB -> 4CB @8x8 ------------------------------------------------------------------------
    {
        f32x4 acc[T4M][T4N];
        conv3x3_mfma<NW, 4 * CB, 4 * CB, 8, 1, T4M, T4N, PST8, WP8, 8>(act, a.packed + a.off.w[5], acc, wave, lane);
        CNN_STAMP(11);
        if (KIND == AFFNET_NET_HARDNET && a.dbg_layer < 0) {
            store_tiles_global<4 * CB, T4M, T4N>(a.out + pidx * HEAD_K, a.packed + a.off.b[5], acc, wave, lane);
            return;
        }
        __syncthreads();
        store_tiles_lds<4 * CB, 8, T4M, T4N, PST8>(act, a.packed + a.off.b[5], acc, wave, lane);
        __syncthreads();
        CNN_STAMP(12);
    }
    if (a.dbg_layer == 5) { dump_planes<4 * CB, 8, PST8, NTHR>(act, a.dbg_out); return; }

    // ---- heads (AffNet / OriNet), VALU -----------------------------------------------------------------
    if (KIND == AFFNET_NET_AFFNET) {
        // conv 64 -> 3, 8x8 valid, + bias -> tanh -> [[1+x0, 0],[x1, 1+x2]] -> up-is-up rectification
        const float* hw = a.packed + a.off.head_w;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = tid; e < 4096; e += NTHR) {
            const int c = e >> 6, y = (e >> 3) & 7, x = e & 7;
            const float vv = act[c * PST8 + (y + 1) * WP8 + x + 1];
            s0 = fmaf(vv, hw[e], s0); s1 = fmaf(vv, hw[4096 + e], s1); s2 = fmaf(vv, hw[8192 + e], s2);
        }
        s0 = block_sum<NW>(s0, red); s1 = block_sum<NW>(s1, red); s2 = block_sum<NW>(s2, red);
        if (tid == 0) {
            const float* hb = a.packed + a.off.head_b;
            const float x0 = tanhf(s0 + hb[0]), x1 = tanhf(s1 + hb[1]), x2 = tanhf(s2 + hb[2]);
            const float a00 = 1.0f + x0, a01 = 0.0f * x0, a10 = x1, a11 = 1.0f + x2;
            // rectifyAffineTransformationUpIsUp (LAF.py:285-291)
            const float det = sqrtf(fabsf(a00 * a11 - a10 * a01 + 1e-10f));
            const float b2a2 = sqrtf(a01 * a01 + a00 * a00);
            float* o = a.out + 4 * pidx;
            o[0] = b2a2 / det; o[1] = 0.0f * det;
            o[2] = (a11 * a01 + a10 * a00) / (b2a2 * det); o[3] = det / b2a2;
        }
    } else if (KIND == AFFNET_NET_ORINET) {
        // conv 64 -> 2, 8x8, padding 1 (= our zero halo) -> 3x3 -> tanh -> mean -> atan2 -> rotation
        const float* hw = a.packed + a.off.head_w;
        float s[2][9];
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int q = 0; q < 9; ++q) s[o][q] = 0.f;
        for (int e = tid; e < 4096; e += NTHR) {
            const int c = e >> 6, ky = (e >> 3) & 7, kx = e & 7;
            const float w0 = hw[e], w1 = hw[4096 + e];
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const float vv = act[c * PST8 + (q / 3 + ky) * WP8 + (q % 3) + kx];
                s[0][q] = fmaf(vv, w0, s[0][q]); s[1][q] = fmaf(vv, w1, s[1][q]);
            }
        }
        // reduce the 18 partial sums: wave shuffles, then one LDS exchange ([wave][18]) - 2 barriers in total
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int q = 0; q < 9; ++q) s[o][q] = wave_sum(s[o][q]);
        __syncthreads();
        float* red2 = patch;                                          // the input patch is dead by now: reuse its LDS
        if (lane == 0) {
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int q = 0; q < 9; ++q) red2[wave * 18 + o * 9 + q] = s[o][q];
        }
        __syncthreads();
        if (tid == 0) {
            float t0 = 0.f, t1 = 0.f;
            const float* hb = a.packed + a.off.head_b;
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                float r0 = 0.f, r1 = 0.f;
                for (int wv = 0; wv < NW; ++wv) { r0 += red2[wv * 18 + q]; r1 += red2[wv * 18 + 9 + q]; }
                t0 += tanhf(r0 + hb[0]); t1 += tanhf(r1 + hb[1]);
            }
            const float yv = t0 / 9.0f, xv = t1 / 9.0f;                       // AdaptiveAvgPool2d(1)
            const float ang = atan2f(yv + 1e-8f, xv + 1e-8f);                 // architectures.py:78
            const float sn = sinf(ang), cs = cosf(ang);
            float* o = a.out + 4 * pidx;
            o[0] = cs; o[1] = sn; o[2] = -sn; o[3] = cs;                      // LAF.py:276-283
        }
    }
}

// ---- HardNet head: (n x 8192) x (8192 x 128) GEMM + BN bias + L2 normalisation ----------------------
// Split-K GEMM on the fp32 matrix cores.  One workgroup = 256 threads = 32 patches x 128 outputs x one
// quarter of K (2048): wave w owns N-tiles 2w, 2w+1 for both 16-patch M-tiles (4 accumulators, each A and
// B fragment is used twice).  The A slab (32 x 128) is staged through LDS with 16-byte loads / stores; B
// ([k][n], BN-folded) streams from L2.  Partial sums go to a scratch [4][n][128] with plain stores (no float
// atomics: bit-reproducible); hardnet_finish_kernel adds them in fixed order, adds the bias and
// L2-normalises.  ceil(n/32) x 4 workgroups (252 for 2000 patches) fill the 256 CUs.
#define HEAD_MP 32
#define HEAD_KSPLIT 4
#define HEAD_KC 128
#define HEAD_AS (HEAD_KC + 4)    // row stride: 16-byte aligned rows, 2-way worst-case bank conflicts
__global__ __launch_bounds__(256) void hardnet_head_kernel(const float* __restrict__ trunk, const float* __restrict__ Bw,
                                                           const int32_t* __restrict__ count, int n_max, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float As[HEAD_MP * HEAD_AS];
    const int n = count ? min(count[blockIdx.z], n_max) : n_max;      // blockIdx.z = image of the batch
    const int p0 = blockIdx.x * HEAD_MP;
    if (p0 >= n) return;
    trunk += (size_t)blockIdx.z * n_max * HEAD_K;
    const size_t rows_total = (size_t)gridDim.z * n_max;
    const int kbeg = blockIdx.y * (HEAD_K / HEAD_KSPLIT);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m = lane & 15, kq = lane >> 4;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = kbeg; k0 < kbeg + HEAD_K / HEAD_KSPLIT; k0 += HEAD_KC) {
        __syncthreads();
        // stage 32 x 128 floats = 1024 float4, 4 per thread (coalesced: 32 consecutive float4 per row)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = tid + 256 * r;
            const int row = f >> 5, c4 = f & 31;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p0 + row < n) v = *reinterpret_cast<const float4*>(trunk + (size_t)(p0 + row) * HEAD_K + k0 + 4 * c4);
            *reinterpret_cast<float4*>(&As[row * HEAD_AS + 4 * c4]) = v;
        }
        __syncthreads();
        const float* bptr = Bw + (size_t)(k0 + kq) * 128 + wave * 32 + m;
#pragma unroll 8
        for (int kk = 0; kk < HEAD_KC; kk += 4) {
            const float a0 = As[m * HEAD_AS + kk + kq], a1 = As[(16 + m) * HEAD_AS + kk + kq];
            const float b0 = bptr[(size_t)kk * 128], b1 = bptr[(size_t)kk * 128 + 16];
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    // acc[i][j][r]: patch p0 + 16 i + 4 (lane>>4) + r, channel 32 wave + 16 j + (lane & 15)
    const int g = lane >> 4;
    float* dst = partial + ((size_t)blockIdx.y * rows_total + (size_t)blockIdx.z * n_max) * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = p0 + 16 * i + 4 * g + r;
            if (row >= n) continue;
            dst[(size_t)row * 128 + wave * 32 + m] = acc[i][0][r];
            dst[(size_t)row * 128 + wave * 32 + 16 + m] = acc[i][1][r];
        }
}

// One wavefront per patch: sum the K-split partials in fixed order, + BN bias, L2 normalise (eps 1e-8).
__global__ __launch_bounds__(256) void hardnet_finish_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                                             const int32_t* __restrict__ count, int n_max, float* __restrict__ out) {
    const int n = count ? min(count[blockIdx.y], n_max) : n_max;      // blockIdx.y = image of the batch
    const int lrow = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (lrow >= n) return;
    const size_t rows_total = (size_t)gridDim.y * n_max, row = (size_t)blockIdx.y * n_max + lrow;
    float v0 = 0.f, v1 = 0.f;
#pragma unroll
    for (int s = 0; s < HEAD_KSPLIT; ++s) {
        const float* p = partial + ((size_t)s * rows_total + row) * 128;
        v0 += p[lane]; v1 += p[64 + lane];
    }
    v0 += bias[lane]; v1 += bias[64 + lane];
    const float tot = wave_sum(v0 * v0 + v1 * v1);
    const float nrm = sqrtf(tot + 1e-8f);                                  // L2Norm (HardNet.py:15-18)
    out[row * 128 + lane] = v0 / nrm;
    out[row * 128 + 64 + lane] = v1 / nrm;
}

// ---- host entry points -------------------------------------------------------------------------------
void aff_fill_pyr_src(const affnet_ctx* ctx, PyrSrc* t) {
    memset(t, 0, sizeof(*t));
    if (ctx->ws) {
        t->n_octaves = ctx->cfg.n_octaves; t->n_levels = ctx->cfg.levels_per_octave;
        t->img_stride = ctx->pyr_stride;
        for (int o = 0; o < t->n_octaves; ++o) {
            const OctaveGeom& g = ctx->oct[o];
            t->h[o] = g.h; t->w[o] = g.w;
            for (int l = 0; l < t->n_levels; ++l) t->lvl[o][l] = ctx->pyr + g.pyr_off + (size_t)l * g.h * g.w;
        }
    }
    aff_base_grid(32, t->base);
}

static unsigned long long* g_dbg_time = nullptr;   // tuning aid, see affnet_cnn32_debug_timing

static int cnn_launch(affnet_ctx* ctx, int kind, const float* packed, const float* patches, const float* lafs, const int32_t* ids,
                      const int32_t* count, int n_max, float* out, float* scratch, int dbg_layer, float* dbg_out, hipStream_t st,
                      bool mark_head = false) {
    if (kind < 0 || kind > 2) return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32: unknown net kind %d", kind);
    if (!packed || !out || n_max < 0 || (!patches && (!lafs || !ids))) return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32: null argument");
    if (!patches && !ctx->ws) return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32: sampling from the pyramid needs a bound workspace");
    if (kind == AFFNET_NET_HARDNET && dbg_layer < 0 && !scratch) return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32: HardNet needs d_scratch (n*(8192+512) floats)");
    if (n_max == 0) return AFFNET_OK;
    const NetLayout L = net_layout(kind);
    CnnArgs a;
    a.packed = packed; a.off = to_offsets(L); a.patches = patches; a.lafs = lafs; a.ids = ids; a.count = count; a.n_max = n_max;
    a.out = (kind == AFFNET_NET_HARDNET) ? scratch : out;
    a.dbg_layer = dbg_layer; a.dbg_out = dbg_out; a.dbg_time = g_dbg_time;
    PyrSrc ps;
    aff_fill_pyr_src(ctx, &ps);
    const int B = patches ? 1 : ctx->B;                  // patch tensors are single-"image"; pyramid sampling covers the batch
    const dim3 grid(n_max, B);
    static const int hard_waves = []() { const char* e = getenv("AFFNET_HARDNET_WAVES"); return (e && atoi(e) == 16) ? 16 : 8; }();
    if (kind == AFFNET_NET_AFFNET) hipLaunchKernelGGL((cnn32_trunk_kernel<AFFNET_NET_AFFNET, 8>), grid, dim3(512), 0, st, a, ps);
    else if (kind == AFFNET_NET_ORINET) hipLaunchKernelGGL((cnn32_trunk_kernel<AFFNET_NET_ORINET, 8>), grid, dim3(512), 0, st, a, ps);
    else if (hard_waves == 8) hipLaunchKernelGGL((cnn32_trunk_kernel<AFFNET_NET_HARDNET, 8>), grid, dim3(512), 0, st, a, ps);
    else hipLaunchKernelGGL((cnn32_trunk_kernel<AFFNET_NET_HARDNET, 16>), grid, dim3(1024), 0, st, a, ps);
    AFF_LAUNCH_CHECK(ctx);
    if (mark_head) aff_prof_mark(ctx, 7, st);
    if (kind == AFFNET_NET_HARDNET && dbg_layer < 0) {
        float* partial = scratch + (size_t)B * n_max * HEAD_K;   // [HEAD_KSPLIT][B * n_max][128] behind the trunk output
        hipLaunchKernelGGL(hardnet_head_kernel, dim3(aff_cdiv(n_max, HEAD_MP), HEAD_KSPLIT, B), dim3(256), 0, st, scratch, packed + L.head_w,
                           count, n_max, partial);
        AFF_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(hardnet_finish_kernel, dim3(aff_cdiv(n_max, 4), B), dim3(256), 0, st, partial, packed + L.head_b, count, n_max, out);
        AFF_LAUNCH_CHECK(ctx);
    }
    return AFFNET_OK;
}

extern "C" int affnet_cnn32_forward(affnet_ctx* ctx, int net_kind, const float* d_packed, const float* d_patches, const int32_t* d_count,
                                    int n_max, float* d_out, float* d_scratch, void* stream) {
    if (!ctx || !d_patches) return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32_forward: null argument");
    return cnn_launch(ctx, net_kind, d_packed, d_patches, nullptr, nullptr, d_count, n_max, d_out, d_scratch, -1, nullptr, (hipStream_t)stream);
}

extern "C" int affnet_cnn32_forward_pyr(affnet_ctx* ctx, int net_kind, const float* d_packed, const float* d_lafs, const int32_t* d_ids,
                                        const int32_t* d_count, int n_max, float* d_out, float* d_scratch, void* stream) {
    if (!ctx) return AFFNET_ERR_INVALID;
    return cnn_launch(ctx, net_kind, d_packed, nullptr, d_lafs, d_ids, d_count, n_max, d_out, d_scratch, -1, nullptr, (hipStream_t)stream);
}

int aff_hardnet_forward_pyr_marked(affnet_ctx* ctx, const float* packed, const float* lafs, const int32_t* ids, const int32_t* count,
                                   int n_max, float* out, float* scratch, hipStream_t st) {
    return cnn_launch(ctx, AFFNET_NET_HARDNET, packed, nullptr, lafs, ids, count, n_max, out, scratch, -1, nullptr, st, true);
}

extern "C" int affnet_cnn32_debug_timing(unsigned long long* d_stamps) {
    g_dbg_time = d_stamps;   // device buffer of n_patches * waves * 16 uint64, or NULL to switch the stamps off
    return AFFNET_OK;
}

extern "C" int affnet_cnn32_debug_layer(affnet_ctx* ctx, int net_kind, const float* d_packed, const float* d_patch, int layer, float* d_out,
                                        void* stream) {
    if (!ctx || !d_patch || !d_out || layer < 0 || layer > 5) return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32_debug_layer: bad argument");
    return cnn_launch(ctx, net_kind, d_packed, d_patch, nullptr, nullptr, nullptr, 1, d_out, nullptr, layer, d_out, (hipStream_t)stream);
}

// ---- MFMA layout self-test ------------------------------------------------------------------------------
__global__ void mfma_selftest_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out) {
    const int lane = threadIdx.x, m = lane & 15, kq = lane >> 4;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m * 4 + kq], B[kq * 16 + m], c, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(4 * kq + r) * 16 + m] = c[r];
}

extern "C" int affnet_selftest_mfma(const float* d_A, const float* d_B, float* d_out, void* stream) {
    if (!d_A || !d_B || !d_out) return AFFNET_ERR_INVALID;
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_A, d_B, d_out);
    return hipGetLastError() == hipSuccess ? AFFNET_OK : AFFNET_ERR_HIP;
}
