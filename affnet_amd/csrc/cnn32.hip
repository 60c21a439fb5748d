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

// LDS activation layout: channel-interleaved by 4.  A tensor [C][H][H] lives as C/4 "plane groups"; element
// (c, y, x) of the zero-haloed (H+2)-square image sits at  (c/4)*PSG + ((y+1)*WP + x+1)*4 + c%4  (floats).  One
// ds_read_b128 of lane (pixel m, kq) then delivers the A operands of FOUR MFMA k-steps (channels 4*(4G+kq)+j,
// j = 0..3) - a quarter of the LDS instructions and half the LDS cycles of per-k-step ds_read_b32, which is what
// kept the matrix pipe waiting: under 16-32 waves of MFMA loops an LDS read returns after several hundred cycles and
// lgkmcnt (4 bits) cannot cover more than 15 reads in flight.  The packed weights are interleaved the same way, so a
// B fragment is one coalesced global_load_dwordx4 per 4 k-steps (1 KB per wave instruction).
//
// Each buffer is written by one layer's epilogue and read by the next layer's implicit GEMM; (WP, PSG) are chosen for
// the READER (ds_read_b128 services lanes {0-3,12-15,20-27},{4-11,16-19,28-31},... per cycle over 64 banks):
//   stride-1 reader: 16 consecutive pixels = 64 banks; the kq = 1 lanes of a group must land on the other half:
//                    H = 32/16: PSG == 0 (mod 64);  H = 8 (a tile = 2 rows): WP = 16, PSG == 32 (mod 64);
//   stride-2 reader: pixels 2 apart hit banks == 0..3 (mod 8) -> PSG == 4 (mod 8) (and WP == 0 (mod 8) when a tile
//                    spans two output rows, 16 -> 8) - these also make the epilogue's ds_write_b32 conflict free.
//
// Operand roles: the WEIGHTS are the MFMA "A" operand (rows = 16 output channels) and the ACTIVATIONS the "B" operand
// (columns = 16 pixels), i.e. each 16x16 tile is out^T[channel][pixel].  A lane then owns FOUR CONSECUTIVE CHANNELS of one
// pixel (rows 4g..4g+3 of column lane&15) = exactly one float4 of the interleaved layout, so the epilogue is one
// conflict-free ds_write_b128 per tile instead of four 4-way-conflicting ds_write_b32.
// LDS reads of the MFMA loops go through an explicit 32-bit LDS byte address: one address VGPR per K group (made
// opaque to the optimiser) + a compile-time immediate per tile.  Left to itself the compiler folds the chunk constant
// into every tile offset, overflows the 16-bit DS offset field and spends one v_add_u32 per ds_read_b128 inside the
// MFMA stream (conv1: 8 per 64 MFMAs, -12 % MFMA rate in tools/mfma_probe.py).
typedef __attribute__((address_space(3))) const f32x4 LdsF4;
__device__ __forceinline__ unsigned lds_byte_addr(const float* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const float*)p;
}
__device__ __forceinline__ f32x4 lds_read4(unsigned byte_addr) { return *(LdsF4*)(size_t)byte_addr; }

// Weight fragments of the MFMA loops come through a buffer descriptor: the lane offset is one loop-invariant 32-bit
// VGPR, the chunk offset an SGPR, the tile offset an immediate - no 64-bit VALU pointer arithmetic between the MFMAs and
// half the address payload of a flat global_load_dwordx4 per instruction.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t weight_rsrc(const float* base, int n_floats) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, n_floats * 4, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_read4(__amdgpu_buffer_rsrc_t r, int lane_byte_off, int uniform_byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_byte_off, uniform_byte_off, 0));
}

template <int H_, int WP_, int PSG_>
struct Lay {
    static constexpr int H = H_, WP = WP_, PSG = PSG_;
    __device__ static __forceinline__ int at(int c, int y, int x) { return (c >> 2) * PSG + ((y + 1) * WP + x + 1) * 4 + (c & 3); }
};
typedef Lay<32, 34, 4672> LayC0;   // conv0 out -> conv1 (stride 1)
typedef Lay<32, 34, 4628> LayC1;   // conv1 out -> conv2 (stride 2)
typedef Lay<16, 18, 1344> LayC2;   // conv2 out -> conv3 (stride 1)
typedef Lay<16, 24, 1732> LayC3;   // conv3 out -> conv4 (stride 2, tile = 2 output rows)
typedef Lay<8, 16, 672> LayC4;     // conv4 out -> conv5 (stride 1, tile = 2 rows)
typedef Lay<8, 16, 672> LayC5;     // conv5 out -> AffNet / OriNet heads
#define WP32 34
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
        L.w_off[i] = off; off += (i == 0) ? (size_t)12 * ch[1] : (size_t)9 * ch[i] * ch[i + 1];   // conv0: K = 9 padded to 12
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
                    if (i == 0) out[L.w_off[0] + (size_t)t * co + n] = w;               // [tap (12, rows 9..11 zero)][n]
                    else                                                                // [tap][G = c/16][kq = (c/4)%4][n][j = c%4]
                        out[L.w_off[i] + ((((size_t)t * (ci / 16) + c / 16) * 4 + (c / 4) % 4) * co + n) * 4 + c % 4] = w;
                }
        }
    }
    if (kind == AFFNET_NET_HARDNET) {
        if (!head_bn_mean || !head_bn_var) return AFFNET_ERR_INVALID;
        for (int n = 0; n < 128; ++n) {
            const float s = 1.0f / sqrtf(head_bn_var[n] + 1e-5f);
            out[L.head_b + n] = -head_bn_mean[n] * s;
            // K order of the head GEMM = the trunk kernel's output order k = pixel * 128 + channel; stored interleaved by 4 like
            // the conv weights, [k/16][(k/4)%4][n][k%4], so that one 16-byte load per lane is the B fragment of 4 MFMA k-steps
            for (int c = 0; c < 128; ++c)
                for (int pp = 0; pp < 64; ++pp) {
                    const size_t k = (size_t)pp * 128 + c;
                    out[L.head_w + (((k >> 4) * 4 + ((k >> 2) & 3)) * 128 + n) * 4 + (k & 3)] = head_w[(size_t)n * HEAD_K + c * 64 + pp] * s;
                }
        }
    } else {
        const int no = kind == AFFNET_NET_AFFNET ? 3 : 2;
        if (!head_b) return AFFNET_ERR_INVALID;
        // [o][pixel p][channel c]: (pixel, 4 consecutive channels) = what one lane of the conv5 epilogue owns (head_partials)
        for (int o = 0; o < no; ++o)
            for (int c = 0; c < 64; ++c)
                for (int pp = 0; pp < 64; ++pp) out[L.head_w + (size_t)o * 4096 + pp * 64 + c] = head_w[((size_t)o * 64 + c) * 64 + pp];
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
// Wave-wide sum on the VALU only (DPP row reductions + 4 readlanes).  __shfl_xor compiles to ds_bpermute_b32,
// i.e. six DEPENDENT trips through the LDS queue, which the MFMA loops of the co-resident waves keep hundreds of
// requests deep: the two input-norm reductions cost ~15k cycles per patch that way (profiles/r01_s2b_cnn_phase_timing).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xB1>(v);     // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);     // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);    // row_half_mirror: 8 lanes
    v = dpp_add<0x140>(v);    // row_mirror: every lane holds the sum of its row of 16
    const int iv = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(iv, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(iv, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(iv, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}

// Sum `v` over the NW wavefronts of the workgroup.  `slot` must hold NW floats that nothing else touches during the
// kernel (each reduction of a kernel gets its own slot), so ONE barrier suffices.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* slot) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += slot[w];
    return t;
}

// Zero the 1-pixel halo of the C/4 plane groups of layout L (16-byte stores).
template <typename L, int NTHR>
__device__ __forceinline__ void zero_halo(float* act, int channels, int tid = threadIdx.x) {
    constexpr int H = L::H, CELLS = 4 * (H + 1);
    const int groups = channels >> 2;
    for (int i = tid; i < groups * CELLS; i += NTHR) {
        const int g = i / CELLS, e = i - g * CELLS;
        int y, x;
        if (e < H + 2) { y = 0; x = e; }
        else if (e < 2 * (H + 2)) { y = H + 1; x = e - (H + 2); }
        else { const int r = e - 2 * (H + 2); y = 1 + (r >> 1); x = (r & 1) ? H + 1 : 0; }
        *reinterpret_cast<f32x4*>(&act[g * L::PSG + (y * L::WP + x) * 4]) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

// Plane groups (16 input channels = 4 MFMA k-steps) per pipeline chunk: grow the chunk until it holds `target`
// MFMAs, as long as the two A register sets stay within `max_a_regs` VGPRs.
constexpr int pick_groups(int cin, int tm, int tn, int target, int max_a_regs) {
    int g = 1;
    while (g * 2 <= cin / 16 && (cin / 16) % (g * 2) == 0 && 4 * tm * tn * g < target && 2 * (g * 2) * tm * 4 <= max_a_regs) g *= 2;
    return g;
}

// Implicit-GEMM 3x3 convolution (padding 1) of the LDS tensor `act` (layout LI, CIN channels) with packed weights
// Wg [tap][CIN/16][kq][COUT][4]; leaves the TM x TN tiles (16 px x 16 ch) of this wave in `acc` (pre-activation, no
// bias).  HOUT = LI::H / STRIDE.  K is walked in chunks of GRP plane groups (a chunk never straddles a tap).
// Software pipeline, one chunk deep, two statically named register sets: while the MFMAs of chunk c issue, the A
// (ds_read_b128) and B (global_load_dwordx4) fragments of chunk c+1 are in flight; loads and MFMAs interleave per
// plane group so that few LDS reads are outstanding at any wait (lgkmcnt has 4 bits).
template <int GRP, int TM, int TN>
struct Frag {
    f32x4 a[GRP][TM];
    f32x4 b[GRP][TN];
};

// B fragments of chunk 0 of a layer (and its bias values): they do not depend on the activations, so the kernel requests
// them BEFORE the barriers / epilogue of the previous layer and their L2 latency (1-2k cycles under load) is hidden.
template <int NW, int COUT, int HOUT, int TM, int TN, int GRP>
__device__ __forceinline__ void prefetch_b0(const float* __restrict__ Wg, f32x4 (&b0)[GRP][TN], int wave, int lane) {
    constexpr int MG = (HOUT * HOUT / 16) / TM;
    const int ng = wave / MG, m = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int u = 0; u < GRP; ++u)
#pragma unroll
        for (int j = 0; j < TN; ++j) b0[u][j] = *reinterpret_cast<const f32x4*>(&Wg[u * 16 * COUT + (kq * COUT + (ng * TN + j) * 16 + m) * 4]);
}
template <int NW, int HOUT, int TM, int TN>
__device__ __forceinline__ void prefetch_bias(const float* __restrict__ bias, f32x4 (&bv)[TN], int wave, int lane) {
    constexpr int MG = (HOUT * HOUT / 16) / TM;
    const int ng = wave / MG;
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const f32x4*>(&bias[(ng * TN + j) * 16 + 4 * (lane >> 4)]);   // channels 4g..4g+3
}

// PROBE (tuning aid, affnet_cnn32_probe): bit 0 = skip the weight loads, bit 1 = skip the activation loads inside the loop,
// bit 3 = activation reads from lane-consecutive addresses (bank-conflict-free reference pattern).
template <int NW, int CIN, int COUT, typename LI, int STRIDE, int TM, int TN, int GRP, int PROBE = 0>
__device__ __forceinline__ void conv3x3_mfma(const float* act, const float* __restrict__ Wg, const f32x4 (&b0)[GRP][TN],
                                             f32x4 (&acc)[TM][TN], int wave, int lane) {
    constexpr int HOUT = LI::H / STRIDE;
    constexpr int MT = HOUT * HOUT / 16, NT = COUT / 16;
    constexpr int MG = MT / TM, NG = NT / TN;
    constexpr int NGRP = CIN / 16, NCHUNK = 9 * NGRP / GRP;
    static_assert(MG * NG == NW, "the waves must tile the layer exactly");
    static_assert(CIN % 16 == 0 && NGRP % GRP == 0, "bad chunking");
    const int mg = wave % MG, ng = wave / MG;
    const int m = lane & 15, kq = lane >> 4;
    // lane address of tile 0 / N-tile 0; the other tiles of the wave sit at compile-time offsets (immediates)
    static_assert(HOUT == 8 || (TM * 16) % HOUT == 0 || HOUT % (TM * 16) == 0, "tile offsets must be wave-uniform constants");
    int a_lane;
    {
        const int p = mg * TM * 16 + m;
        const int oy = p / HOUT, ox = p - oy * HOUT;
        a_lane = kq * LI::PSG + ((oy * STRIDE) * LI::WP + ox * STRIDE) * 4;
    }
    const int b_lane = (kq * COUT + ng * TN * 16 + m) * 4;
    if (PROBE & 8) a_lane = lane * 4;                      // probe: 64 consecutive float4 per read = the conflict-free ideal
    const unsigned a_addr0 = lds_byte_addr(act) + a_lane * 4;
    auto a_imm = [](int i) {                               // byte offset of tile i from tile 0 (compile-time after unrolling)
        return 4 * ((PROBE & 8) ? i * 256 : (HOUT == 8 ? i * 2 * STRIDE * LI::WP * 4 : (((i * 16) / HOUT) * STRIDE * LI::WP + ((i * 16) % HOUT) * STRIDE) * 4));
    };
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Frag<GRP, TM, TN> f0, f1;

    auto a_chunk_off = [](int ch) {
        const int q0 = ch * GRP;
        const int tap = q0 / NGRP, g0 = q0 - tap * NGRP;
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;          // tap / 3, tap % 3 for tap < 9
        return g0 * 4 * LI::PSG + (ky * LI::WP + kx) * 4;
    };
    const __amdgpu_buffer_rsrc_t wrsrc = weight_rsrc(Wg, 9 * CIN * COUT);
    auto load_group = [&](Frag<GRP, TM, TN>& f, int u, int a_off, int w_off) {
        if (!(PROBE & 2)) {
            unsigned ab = a_addr0 + (a_off + u * 4 * LI::PSG) * 4;
            asm("" : "+v"(ab));
#pragma unroll
            for (int i = 0; i < TM; ++i) f.a[u][i] = lds_read4(ab + a_imm(i));
        }
        if (!(PROBE & 1)) {
#pragma unroll
            for (int j = 0; j < TN; ++j) f.b[u][j] = buf_read4(wrsrc, b_lane * 4 + j * 256, (w_off + u * 16 * COUT) * 4);
        }
    };
    auto mfma_group = [&](const Frag<GRP, TM, TN>& f, int u) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.b[u][j][s4], f.a[u][i][s4], acc[i][j], 0, 0, 0);   // W^T x act
    };
    // compute the chunk held in `cur` while loading chunk `nxt_ch` into `nxt`
    auto stage = [&](const Frag<GRP, TM, TN>& cur, Frag<GRP, TM, TN>& nxt, int nxt_ch) {
        const int a_off = a_chunk_off(nxt_ch);
        const int w_off = nxt_ch * (GRP * 16 * COUT);
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            load_group(nxt, u, a_off, w_off);
            mfma_group(cur, u);
            // Schedule: the TM ds_read_b128 and TN global_load_dwordx4 of this group are spread BETWEEN its MFMAs (one load
            // after every Q MFMAs) instead of being issued as a burst in front of them.  A wave cannot issue an MFMA while it
            // issues a load (a 1 KB dwordx4 wave-load holds the issue slot for tens of cycles); with bursts at the group
            // boundaries both waves of a SIMD tended to be in their bursts together and the pipe idled ~8 % of the loop
            // (tools/mfma_probe.py: conv1 81.7 % -> 88.5 % of peak with the weight loads removed).
            constexpr int NM = 4 * TM * TN, NL = ((PROBE & 2) ? 0 : TM) + ((PROBE & 1) ? 0 : TN), Q = NL ? NM / (NL + 1) : NM;
            // weight loads (L2, long latency) first, activation loads (LDS) after them
#pragma unroll
            for (int l = 0; l < ((PROBE & 1) ? 0 : TN); ++l) {
                if (l == 0) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                else __builtin_amdgcn_sched_group_barrier(0x008, Q, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
#pragma unroll
            for (int l = 0; l < ((PROBE & 2) ? 0 : TM); ++l) {
                __builtin_amdgcn_sched_group_barrier(0x008, Q, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);      // whatever MFMAs remain
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    {
        const int a_off = a_chunk_off(0);
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            const unsigned ab = a_addr0 + (a_off + u * 4 * LI::PSG) * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) f0.a[u][i] = lds_read4(ab + a_imm(i));
#pragma unroll
            for (int j = 0; j < TN; ++j) f0.b[u][j] = b0[u][j];
        }
    }
    if (PROBE) f1 = f0;
    // (The two waves of a workgroup that share a SIMD do not advance evenly - the older one wins the arbitration and leaves
    // the loop ~12 % earlier.  Alternating s_setprio between them evens that out but the pair's finish time, set by the
    // MFMA pipe, does not move: measured, not kept.)
#pragma unroll 1
    for (int ch = 0; ch + 1 < NCHUNK; ch += 2) {
        stage(f0, f1, ch + 1);
        stage(f1, f0, (ch + 2 < NCHUNK) ? ch + 2 : ch + 1);         // past the end: re-read the last chunk (in bounds, unused)
    }
    if (NCHUNK & 1) {
#pragma unroll
        for (int u = 0; u < GRP; ++u) mfma_group(f0, u);
    }
}

// Same contraction with ONE A register set that is reloaded in place (for TM = 8 under a 128-VGPR budget, where two
// sets of 8 float4 do not fit): chunk = one plane group; the tiles are processed in pairs - 8 * TN MFMAs on 2 * TN
// independent accumulators - and as soon as a pair's MFMAs have issued, its two A registers are reloaded with the next
// chunk's data, so every load is (TM - 2) / TM of a chunk ahead of its use.  B fragments keep two sets.
template <int NW, int CIN, int COUT, typename LI, int STRIDE, int TM, int TN>
__device__ __forceinline__ void conv3x3_mfma_roll(const float* act, const float* __restrict__ Wg, const f32x4 (&b0)[1][TN],
                                                  f32x4 (&acc)[TM][TN], int wave, int lane) {
    constexpr int HOUT = LI::H / STRIDE;
    constexpr int MT = HOUT * HOUT / 16, NT = COUT / 16;
    constexpr int MG = MT / TM, NG = NT / TN;
    constexpr int NGRP = CIN / 16, NCHUNK = 9 * NGRP;
    static_assert(MG * NG == NW && TM % 2 == 0 && CIN % 16 == 0, "bad tiling");
    const int mg = wave % MG, ng = wave / MG;
    const int m = lane & 15, kq = lane >> 4;
    // lane address of tile 0 / N-tile 0; the other tiles of the wave sit at compile-time offsets (immediates)
    static_assert(HOUT == 8 || (TM * 16) % HOUT == 0 || HOUT % (TM * 16) == 0, "tile offsets must be wave-uniform constants");
    int a_lane;
    {
        const int p = mg * TM * 16 + m;
        const int oy = p / HOUT, ox = p - oy * HOUT;
        a_lane = kq * LI::PSG + ((oy * STRIDE) * LI::WP + ox * STRIDE) * 4;
    }
    const int b_lane = (kq * COUT + ng * TN * 16 + m) * 4;
    const unsigned a_addr0 = lds_byte_addr(act) + a_lane * 4;
    auto a_imm = [](int i) {                               // byte offset of tile i from tile 0 (compile-time after unrolling)
        return 4 * (HOUT == 8 ? i * 2 * STRIDE * LI::WP * 4 : (((i * 16) / HOUT) * STRIDE * LI::WP + ((i * 16) % HOUT) * STRIDE) * 4);
    };
    const __amdgpu_buffer_rsrc_t wrsrc = weight_rsrc(Wg, 9 * CIN * COUT);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto a_chunk_off = [](int ch) {
        const int tap = ch / NGRP, g0 = ch - tap * NGRP;
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
        return g0 * 4 * LI::PSG + (ky * LI::WP + kx) * 4;
    };
    f32x4 fa[TM], fb0[TN], fb1[TN];
    {
        const int a_off = a_chunk_off(0);
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = lds_read4(a_addr0 + a_off * 4 + a_imm(i));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb0[j] = b0[0][j];
    }
    auto chunk = [&](const f32x4 (&bc)[TN], f32x4 (&bn)[TN], int nxt_ch) {
        unsigned ab = a_addr0 + a_chunk_off(nxt_ch) * 4;
        asm("" : "+v"(ab));
#pragma unroll
        for (int j = 0; j < TN; ++j) bn[j] = buf_read4(wrsrc, b_lane * 4 + j * 256, nxt_ch * (16 * COUT) * 4);
#pragma unroll
        for (int ip = 0; ip < TM; ip += 2) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int i = ip; i < ip + 2; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bc[j][s4], fa[i][s4], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = ip; i < ip + 2; ++i) fa[i] = lds_read4(ab + a_imm(i));
        }
    };
#pragma unroll 1
    for (int ch = 0; ch + 1 < NCHUNK; ch += 2) {
        chunk(fb0, fb1, ch + 1);
        chunk(fb1, fb0, (ch + 2 < NCHUNK) ? ch + 2 : ch + 1);
    }
    if (NCHUNK & 1) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb0[j][s4], fa[i][s4], acc[i][j], 0, 0, 0);
    }
}

// conv0 (1 -> COUT channels, K = 9 taps padded to 12) on the matrix cores as well: A[m][k] = the padded standardised
// patch at (pixel m, tap k), B = packed [12][COUT] taps (rows 9..11 zero), accumulators start at the bias, so the result
// is the fmaf chain bias, tap 0, ..., tap 8 (+ three exact fma(x, 0, acc)).  3 MFMAs per tile instead of 9 * COUT VALU
// FMAs per pixel behind dependent LDS weight reads.
template <int NW, int COUT, int TM, int TN>
__device__ __forceinline__ void conv0_load_w(const float* __restrict__ W0, const float* __restrict__ bias, float (&b)[3][TN],
                                             f32x4 (&bv)[TN], int wave, int lane) {
    constexpr int MG = 64 / TM;
    const int ng = wave / MG, m = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = (ng * TN + j) * 16 + m;
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) b[s3][j] = W0[(4 * s3 + kq) * COUT + n];
        bv[j] = *reinterpret_cast<const f32x4*>(&bias[(ng * TN + j) * 16 + 4 * kq]);     // accumulator rows 4g..4g+3 = channels
    }
}

template <int NW, int COUT, int TM, int TN>
__device__ __forceinline__ void conv0_mfma(const float* patch, const float (&b)[3][TN], const f32x4 (&bv)[TN], f32x4 (&acc)[TM][TN],
                                           int wave, int lane) {
    constexpr int MT = 64, NT = COUT / 16, MG = MT / TM, NG = NT / TN;
    static_assert(MG * NG == NW, "the waves must tile the layer exactly");
    const int mg = wave % MG;
    const int m = lane & 15, kq = lane >> 4;
    int toff[3];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        const int t = 4 * s3 + kq;
        toff[s3] = t < 9 ? (t / 3) * WP32 + (t % 3) : 0;            // taps 9..11: any valid address (weight is zero)
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = bv[j];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = (mg * TM + i) * 16 + m;
        const int base = (p >> 5) * WP32 + (p & 31);
        float av[3];
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) av[s3] = patch[base + toff[s3]];
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s3][j], av[s3], acc[i][j], 0, 0, 0);
    }
}

// Epilogue: (+ bias,) ReLU, write the wave's tiles into the LDS layout LO read by the NEXT layer: lane (n = pixel of the
// tile, g) holds channels 4g..4g+3 -> one float4 of plane group (N-tile * 4 + g).
template <int COUT, typename LO, int TM, int TN, bool ADD_BIAS = true>
__device__ __forceinline__ void store_tiles_lds(float* act, const f32x4 (&bias)[TN], const f32x4 (&acc)[TM][TN], int wave, int lane) {
    constexpr int HOUT = LO::H;
    constexpr int MT = HOUT * HOUT / 16, MG = MT / TM;
    const int mg = wave % MG, ng = wave / MG;
    const int n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = (mg * TM + i) * 16 + n;
        const int oy = p / HOUT, ox = p - oy * HOUT;
        const int pbase = ((oy + 1) * LO::WP + ox + 1) * 4;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x4 v = acc[i][j];
            if (ADD_BIAS) v += bias[j];
            v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
            *reinterpret_cast<f32x4*>(&act[((ng * TN + j) * 4 + g) * LO::PSG + pbase]) = v;
        }
    }
}

// Same for the last trunk layer of HardNet: global [pixel p][channel c] = the head GEMM's K order (float4 = 4 channels).
template <int COUT, int TM, int TN>
__device__ __forceinline__ void store_tiles_global(float* __restrict__ dst, const f32x4 (&bias)[TN], const f32x4 (&acc)[TM][TN],
                                                   int wave, int lane) {
    constexpr int MT = 4, MG = MT / TM;
    const int mg = wave % MG, ng = wave / MG;
    const int n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = (mg * TM + i) * 16 + n;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x4 v = acc[i][j] + bias[j];
            v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
            *reinterpret_cast<f32x4*>(dst + p * COUT + (ng * TN + j) * 16 + 4 * g) = v;
        }
    }
}

// AffNet / OriNet heads, first half, straight from the conv5 accumulators (no conv5 tensor in HBM): a lane owns channels
// c4..c4+3 of pixel p of each of its tiles = one float4 of the head weights [o][pixel][channel]; it forms its share of
// every head dot product, the wave reduces them and lane 0 writes the wave's partial sums to part[wave][*].  The eight
// partials per patch are combined in fixed order by cnn16_finish_kernel (bit-reproducible, no atomics).
//   AffNet: 3 outputs  = conv 64 -> 3, 8x8 valid                   (architectures.py:227-229)      part[8][4]
//   OriNet: 2 x 9      = conv 64 -> 2, 8x8, padding 1 -> 3x3 map   (architectures.py:56-58)        part[8][18]
#define HEAD_PART_AFF 32
#define HEAD_PART_ORI 144
template <int KIND, int TM>
__device__ __forceinline__ void head_partials(const float* __restrict__ hw, const f32x4 (&bias)[1], const f32x4 (&acc)[TM][1],
                                              float* __restrict__ part, int wave, int lane) {
    constexpr int MT = 4, MG = MT / TM;
    const int mg = wave % MG, ng = wave / MG;
    const int n = lane & 15, g = lane >> 4;
    const int c4 = ng * 16 + 4 * g;
    f32x4 v[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        v[i] = acc[i][0] + bias[0];
        v[i].x = fmaxf(v[i].x, 0.0f); v[i].y = fmaxf(v[i].y, 0.0f); v[i].z = fmaxf(v[i].z, 0.0f); v[i].w = fmaxf(v[i].w, 0.0f);
    }
    if (KIND == AFFNET_NET_AFFNET) {
        const __amdgpu_buffer_rsrc_t r = weight_rsrc(hw, 3 * 4096);
        f32x4 w[3][TM];
#pragma unroll
        for (int o = 0; o < 3; ++o)
#pragma unroll
            for (int i = 0; i < TM; ++i) w[o][i] = buf_read4(r, (n * 64 + c4) * 4, (o * 4096 + (mg * TM + i) * 16 * 64) * 4);
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            float sacc = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) sacc = fmaf(v[i][j], w[o][i][j], sacc);
            sacc = wave_sum(sacc);
            if (lane == 0) part[wave * 4 + o] = sacc;
        }
    } else {
        // taps outside the 8x8 kernel: the lane offset is pushed past the end of the buffer, the load returns 0 (no branch,
        // no 64-bit address arithmetic, no memory traffic for those lanes)
        const __amdgpu_buffer_rsrc_t r = weight_rsrc(hw, 2 * 4096);
        int toff[9][TM];
#pragma unroll
        for (int q = 0; q < 9; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int p = (mg * TM + i) * 16 + n, py = p >> 3, px = p & 7;
                const int ky = py - q / 3 + 1, kx = px - q % 3 + 1;                    // padding 1: tap that sees this pixel
                const bool ok = ky >= 0 && ky < 8 && kx >= 0 && kx < 8;
                toff[q][i] = ok ? ((ky * 8 + kx) * 64 + c4) * 4 : 0x40000000;
            }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            f32x4 w[9][TM];
#pragma unroll
            for (int q = 0; q < 9; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i) w[q][i] = buf_read4(r, toff[q][i], o * 4096 * 4);
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                float sacc = 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sacc = fmaf(v[i][j], w[q][i][j], sacc);
                sacc = wave_sum(sacc);
                if (lane == 0) part[wave * 18 + o * 9 + q] = sacc;
            }
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
    unsigned long long* dbg_time;   // != NULL: s_memtime stamps [patch][wave][32] at the phase boundaries (tuning aid)
};

#define CNN_STAMP(k)                                                                                     \
    do {                                                                                                 \
        if (STAMPS && a.dbg_time && lane == 0) a.dbg_time[((size_t)pidx * NW + wave) * 32 + (k)] = __builtin_readcyclecounter(); \
    } while (0)

template <int CB>
struct TrunkLds {
    static constexpr int ACT = (CB / 4) * LayC0::PSG;   // the largest layout (conv0 output); all later ones are smaller
    static constexpr int PATCH = WP32 * WP32;
    static constexpr int RED = 256;                     // reduction slots (block_sum, head exchanges)
    static constexpr int TOTAL = ACT + PATCH + RED;
};

template <int C, typename L, int NTHR>
__device__ __forceinline__ void dump_planes(const float* act, float* dst) {
    constexpr int H = L::H;
    for (int i = threadIdx.x; i < C * H * H; i += NTHR) {
        const int c = i / (H * H), r = i - c * H * H, y = r / H, x = r - y * H;
        dst[i] = act[L::at(c, y, x)];
    }
}

// One workgroup = one patch through one trunk.  KIND: 0 AffNet, 1 OriNet, 2 HardNet (CB = 16 / 16 / 32); NW = 8 wavefronts.
// AffNet / OriNet: 79 KB LDS -> 2 workgroups per CU (4 waves / SIMD, 128 VGPRs); HardNet: 154 KB LDS -> 1 workgroup per
// CU (2 waves / SIMD, 256 VGPRs).
// STAMPS = debug instantiation: the s_memtime phase stamps of tools/cnn_phase_timing.py and the per-layer activation dumps
// of affnet_cnn32_debug_layer exist only there (26 stamp sites = 26 predicated stores + branches in every wave otherwise).
template <int KIND, int NW, bool STAMPS>
__global__ __launch_bounds__(NW * 64, (KIND == AFFNET_NET_HARDNET) ? NW / 4 : 4) void cnn32_trunk_kernel(CnnArgs a, PyrSrc ps) {
    constexpr int CB = (KIND == AFFNET_NET_HARDNET) ? 32 : 16;
    constexpr int NTHR = NW * 64;
    constexpr int PPT = 1024 / NTHR;                    // input pixels per thread (2 or 1)
    constexpr int RPT = 32 / PPT;                       // patch rows covered by one pass of the workgroup
    // per-wave register blocking (TM x TN tiles of 16 px x 16 ch); MG * NG == NW for every layer
    constexpr int T1M = (CB == 16) ? 8 : 64 / NW, T1N = CB / 16;
    // conv2 / conv3: ONE channel tile per wave and as many pixel tiles as that allows - activation fragments come from LDS
    // (nearly free), weight fragments are 1 KB global loads whose cost shows in the MFMA rate: 4 x 1 instead of 2 x 2 took the
    // isolated AffNet conv3 loop from 121 to 146 TFLOP/s (tools/clock_probe.py 13 / 14)
    constexpr int T2M = (CB == 16) ? 4 : 64 / NW, T2N = 1;
    constexpr int T4M = (CB == 16) ? 2 : 32 / NW, T4N = 1;
    // plane groups (4 k-steps each) per pipeline chunk; VGPR budget 128 at 4 waves / SIMD, 256 at 2
    constexpr int AREG = (NW == 8 && CB == 32) ? 128 : 48;
    constexpr bool ROLL1 = (T1M * 8 > AREG);            // conv1: two A sets of T1M float4 do not fit -> rolling single set
    constexpr int G1 = pick_groups(CB, T1M, T1N, 32, AREG), G2 = pick_groups(CB, T2M, T2N, 32, AREG);
    constexpr int G3 = pick_groups(2 * CB, T2M, T2N, 32, AREG), G4 = pick_groups(2 * CB, T4M, T4N, 32, AREG);
    constexpr int G5 = pick_groups(4 * CB, T4M, T4N, 32, AREG);
    static_assert((CB / 4) * LayC1::PSG <= TrunkLds<CB>::ACT && (CB / 2) * LayC3::PSG <= TrunkLds<CB>::ACT, "LDS layout");
    __shared__ __attribute__((aligned(16))) float lds[TrunkLds<CB>::TOTAL];
    float* act = lds;
    float* patch = lds + TrunkLds<CB>::ACT;
    float* red = patch + TrunkLds<CB>::PATCH;
    // grid = (n_max, batch): row blockIdx.x of image blockIdx.y; global row = image * n_max + row
    const int n = a.count ? min(a.count[blockIdx.y], a.n_max) : a.n_max;
    if ((int)blockIdx.x >= n) return;
    const size_t pidx = (size_t)blockIdx.y * a.n_max + blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // Issue priority (HardNet only, one workgroup per CU): the short latency-bound phases (input, conv0, epilogues) run at
    // priority 3, the MFMA loops at 0: +2% (130 -> 133 TFLOP/s).  For AffNet / OriNet (two workgroups per CU) it is
    // zero-sum: the non-MFMA phases of one workgroup get 2x faster (with equal priorities the arbiter prefers the OLDER
    // waves, so a young workgroup next to an older one in its MFMA loop crawls: 3.7k vs 0.5k cycles per block reduction),
    // but their VALU instructions then displace the other workgroup's MFMA issue slots (-5% overall), so it stays off.
    constexpr bool PRIO = (KIND == AFFNET_NET_HARDNET);
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    CNN_STAMP(0);
    if (STAMPS && a.dbg_time && lane == 0) {   // where this workgroup runs (tuning aid: per-CU timelines)
        a.dbg_time[((size_t)pidx * NW + wave) * 32 + 14] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        a.dbg_time[((size_t)pidx * NW + wave) * 32 + 15] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    }
    // conv0 taps + bias and the first weight chunk of conv1: requested now, consumed after the input phase
    float w0[3][T1N];
    f32x4 bias0[T1N];
    conv0_load_w<NW, CB, T1M, T1N>(a.packed + a.off.w[0], a.packed + a.off.b[0], w0, bias0, wave, lane);
    f32x4 b1[ROLL1 ? 1 : G1][T1N];
    prefetch_b0<NW, CB, 32, T1M, T1N, (ROLL1 ? 1 : G1)>(a.packed + a.off.w[1], b1, wave, lane);

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
    CNN_STAMP(16);
    // halo of the padded patch (4 x 33 cells) and of the CB activation planes; interiors are written below / by conv0
    if (tid < 4 * 33) {
        const int e = tid;
        const int y = e < 34 ? 0 : (e < 68 ? 33 : 1 + ((e - 68) >> 1)), x = e < 34 ? e : (e < 68 ? e - 34 : ((e - 68) & 1) * 33);
        patch[y * WP32 + x] = 0.0f;
    }
    zero_halo<LayC0, NTHR>(act, CB);
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < PPT; ++q) sum += v[q];
    const float mean = block_sum<NW>(sum, red) * (1.0f / 1024.0f);
    CNN_STAMP(17);
    float sq = 0.f;
#pragma unroll
    for (int q = 0; q < PPT; ++q) { v[q] -= mean; sq += v[q] * v[q]; }
    const float var = block_sum<NW>(sq, red + NW) * (1.0f / 1023.0f);       // torch.std: unbiased
    const float sd = sqrtf(var) + 1e-7f;
    CNN_STAMP(18);
#pragma unroll
    for (int q = 0; q < PPT; ++q) patch[((tid >> 5) + q * RPT + 1) * WP32 + (tid & 31) + 1] = v[q] / sd;
    __syncthreads();
    CNN_STAMP(1);

    // ---- conv0: 1 -> CB, K = 9 (padded to 12), MFMA; reads `patch`, writes `act`: no barrier in between ----
    f32x4 bias1[T1N];
    {
        f32x4 acc[T1M][T1N];
        conv0_mfma<NW, CB, T1M, T1N>(patch, w0, bias0, acc, wave, lane);
        CNN_STAMP(19);
        prefetch_bias<NW, 32, T1M, T1N>(a.packed + a.off.b[1], bias1, wave, lane);
        store_tiles_lds<CB, LayC0, T1M, T1N, false>(act, bias0, acc, wave, lane);
        CNN_STAMP(20);
    }
    __syncthreads();
    if (STAMPS && a.dbg_layer == 0) { dump_planes<CB, LayC0, NTHR>(act, a.dbg_out); return; }
    CNN_STAMP(2);

    // Every layer: MFMA loop -> request the next layer's first weight chunk and bias -> barrier (all waves done reading
    // the input) -> zero the halo of the OUTPUT layout, bias + ReLU + store in place -> barrier.
    // ---- conv1: CB -> CB @32x32 --------------------------------------------------------------------
    f32x4 b2[G2][T2N];
    f32x4 bias2[T2N];
    {
        f32x4 acc[T1M][T1N];
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        if (ROLL1) conv3x3_mfma_roll<NW, CB, CB, LayC0, 1, T1M, T1N>(act, a.packed + a.off.w[1], reinterpret_cast<const f32x4 (&)[1][T1N]>(b1), acc, wave, lane);
        else conv3x3_mfma<NW, CB, CB, LayC0, 1, T1M, T1N, G1>(act, a.packed + a.off.w[1], reinterpret_cast<const f32x4 (&)[G1][T1N]>(b1), acc, wave, lane);
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        CNN_STAMP(3);
        prefetch_b0<NW, 2 * CB, 16, T2M, T2N, G2>(a.packed + a.off.w[2], b2, wave, lane);
        prefetch_bias<NW, 16, T2M, T2N>(a.packed + a.off.b[2], bias2, wave, lane);
        __syncthreads();
        CNN_STAMP(21);
        zero_halo<LayC1, NTHR>(act, CB);
        store_tiles_lds<CB, LayC1, T1M, T1N>(act, bias1, acc, wave, lane);
        CNN_STAMP(22);
        __syncthreads();
        CNN_STAMP(4);
    }
    if (STAMPS && a.dbg_layer == 1) { dump_planes<CB, LayC1, NTHR>(act, a.dbg_out); return; }

    // ---- conv2: CB -> 2CB, stride 2 @16x16 -----------------------------------------------------------
    f32x4 b3[G3][T2N];
    f32x4 bias3[T2N];
    {
        f32x4 acc[T2M][T2N];
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        conv3x3_mfma<NW, CB, 2 * CB, LayC1, 2, T2M, T2N, G2>(act, a.packed + a.off.w[2], b2, acc, wave, lane);
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        CNN_STAMP(5);
        prefetch_b0<NW, 2 * CB, 16, T2M, T2N, G3>(a.packed + a.off.w[3], b3, wave, lane);
        prefetch_bias<NW, 16, T2M, T2N>(a.packed + a.off.b[3], bias3, wave, lane);
        __syncthreads();
        zero_halo<LayC2, NTHR>(act, 2 * CB);
        store_tiles_lds<2 * CB, LayC2, T2M, T2N>(act, bias2, acc, wave, lane);
        __syncthreads();
        CNN_STAMP(6);
    }
    if (STAMPS && a.dbg_layer == 2) { dump_planes<2 * CB, LayC2, NTHR>(act, a.dbg_out); return; }

    // ---- conv3: 2CB -> 2CB @16x16 --------------------------------------------------------------------
    f32x4 b4[G4][T4N];
    f32x4 bias4[T4N];
    {
        f32x4 acc[T2M][T2N];
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        conv3x3_mfma<NW, 2 * CB, 2 * CB, LayC2, 1, T2M, T2N, G3>(act, a.packed + a.off.w[3], b3, acc, wave, lane);
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        CNN_STAMP(7);
        prefetch_b0<NW, 4 * CB, 8, T4M, T4N, G4>(a.packed + a.off.w[4], b4, wave, lane);
        prefetch_bias<NW, 8, T4M, T4N>(a.packed + a.off.b[4], bias4, wave, lane);
        __syncthreads();
        zero_halo<LayC3, NTHR>(act, 2 * CB);
        store_tiles_lds<2 * CB, LayC3, T2M, T2N>(act, bias3, acc, wave, lane);
        __syncthreads();
        CNN_STAMP(8);
    }
    if (STAMPS && a.dbg_layer == 3) { dump_planes<2 * CB, LayC3, NTHR>(act, a.dbg_out); return; }

    // ---- conv4: 2CB -> 4CB, stride 2 @8x8 --------------------------------------------------------------
    f32x4 b5[G5][T4N];
    f32x4 bias5[T4N];
    {
        f32x4 acc[T4M][T4N];
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        conv3x3_mfma<NW, 2 * CB, 4 * CB, LayC3, 2, T4M, T4N, G4>(act, a.packed + a.off.w[4], b4, acc, wave, lane);
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        CNN_STAMP(9);
        prefetch_b0<NW, 4 * CB, 8, T4M, T4N, G5>(a.packed + a.off.w[5], b5, wave, lane);
        prefetch_bias<NW, 8, T4M, T4N>(a.packed + a.off.b[5], bias5, wave, lane);
        __syncthreads();
        zero_halo<LayC4, NTHR>(act, 4 * CB);
        store_tiles_lds<4 * CB, LayC4, T4M, T4N>(act, bias4, acc, wave, lane);
        __syncthreads();
        CNN_STAMP(10);
    }
    if (STAMPS && a.dbg_layer == 4) { dump_planes<4 * CB, LayC4, NTHR>(act, a.dbg_out); return; }

    // ---- conv5: 4CB -> 4CB @8x8 ------------------------------------------------------------------------
    {
        f32x4 acc[T4M][T4N];
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        conv3x3_mfma<NW, 4 * CB, 4 * CB, LayC4, 1, T4M, T4N, G5>(act, a.packed + a.off.w[5], b5, acc, wave, lane);
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        CNN_STAMP(11);
        if (!STAMPS || a.dbg_layer < 0) {
            if constexpr (KIND == AFFNET_NET_HARDNET)   // conv5 tensor -> HBM as [pixel][channel]; the head GEMM runs over all patches
                store_tiles_global<4 * CB, T4M, T4N>(a.out + pidx * (64 * 4 * CB), bias5, acc, wave, lane);
            else                                        // per-wave partial sums of the head's dot products
                head_partials<KIND, T4M>(a.packed + a.off.head_w, bias5, acc,
                                         a.out + pidx * (KIND == AFFNET_NET_AFFNET ? HEAD_PART_AFF : HEAD_PART_ORI), wave, lane);
            CNN_STAMP(13);
            return;
        }
        __syncthreads();
        store_tiles_lds<4 * CB, LayC5, T4M, T4N>(act, bias5, acc, wave, lane);   // debug dump only
        __syncthreads();
        CNN_STAMP(12);
    }
    if (STAMPS && a.dbg_layer == 5) { dump_planes<4 * CB, LayC5, NTHR>(act, a.dbg_out); return; }
}

// ---- AffNet / OriNet heads, second half: combine the eight per-wave partials of a patch, one thread per patch -------
//   AffNet : + bias -> tanh -> [[1+x0, 0],[x1, 1+x2]] -> rectifyAffineTransformationUpIsUp
//            (architectures.py:227-229,246-252, LAF.py:285-291)
//   OriNet : + bias -> tanh -> mean over the 3x3 map -> atan2 -> rotation (architectures.py:56-58,76-82, LAF.py:276-283)
template <int KIND>
__global__ __launch_bounds__(256) void cnn16_finish_kernel(const float* __restrict__ part, const float* __restrict__ hb,
                                                           const int32_t* __restrict__ count, int n_max, float* __restrict__ out) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    const int n = count ? min(count[blockIdx.y], n_max) : n_max;
    if (row >= n) return;
    const size_t pidx = (size_t)blockIdx.y * n_max + row;
    float* o = out + 4 * pidx;
    if (KIND == AFFNET_NET_AFFNET) {
        const f32x4* pp = reinterpret_cast<const f32x4*>(part + pidx * HEAD_PART_AFF);
        f32x4 r[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) r[w] = pp[w];
        const f32x4 s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        const float x0 = tanhf(s.x + hb[0]), x1 = tanhf(s.y + hb[1]), x2 = tanhf(s.z + hb[2]);
        const float a00 = 1.0f + x0, a01 = 0.0f * x0, a10 = x1, a11 = 1.0f + x2;
        const float det = sqrtf(fabsf(a00 * a11 - a10 * a01 + 1e-10f));
        const float b2a2 = sqrtf(a01 * a01 + a00 * a00);
        o[0] = b2a2 / det; o[1] = 0.0f * det;
        o[2] = (a11 * a01 + a10 * a00) / (b2a2 * det); o[3] = det / b2a2;
    } else {
        const float* pp = part + pidx * HEAD_PART_ORI;
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            float r0 = 0.f, r1 = 0.f;
#pragma unroll
            for (int w = 0; w < 8; w += 2) { r0 += pp[w * 18 + q] + pp[(w + 1) * 18 + q]; r1 += pp[w * 18 + 9 + q] + pp[(w + 1) * 18 + 9 + q]; }
            t0 += tanhf(r0 + hb[0]); t1 += tanhf(r1 + hb[1]);
        }
        const float yv = t0 / 9.0f, xv = t1 / 9.0f;                       // AdaptiveAvgPool2d(1)
        const float ang = atan2f(yv + 1e-8f, xv + 1e-8f);                 // architectures.py:78
        const float sn = sinf(ang), cs = cosf(ang);
        o[0] = cs; o[1] = sn; o[2] = -sn; o[3] = cs;
    }
}

// ---- HardNet head: (n x 8192) x (8192 x 128) GEMM + BN bias + L2 normalisation ----------------------
// Split-K GEMM on the fp32 matrix cores.  One workgroup = 256 threads = 64 patches x 128 outputs x one quarter of K
// (2048): wave w owns N-tiles 2w, 2w+1 for all four 16-patch M-tiles (8 accumulators).  K is walked in the conv loops'
// interleaved order (k = 16 G + 4 kq + j belongs to k-step j of lane group kq), so per 16 k a wave issues 4
// ds_read_b128 (A, from the LDS slab) + 2 buffer_load_dwordx4 (B, BN-folded weights [k/16][kq][n][4] from L2) for 32
// MFMAs.  The A slab (64 x 128) is fetched one iteration ahead into registers (buffer loads: rows >= n read as zero)
// and written to LDS with 16-byte stores.  Partial sums go to a scratch [4][n][128] with plain stores (no float atomics:
// bit-reproducible); hardnet_finish_kernel adds them in fixed order, adds the bias and L2-normalises.
#define HEAD_MP 64
#define HEAD_KSPLIT 4
#define HEAD_KC 128
#define HEAD_AS (HEAD_KC + 4)    // row stride: 16-byte aligned rows
__global__ __launch_bounds__(256, 2) void hardnet_head_kernel(const float* __restrict__ trunk, const float* __restrict__ Bw,
                                                              const int32_t* __restrict__ count, int n_max, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float As[HEAD_MP * HEAD_AS];
    const int n = count ? min(count[blockIdx.z], n_max) : n_max;      // blockIdx.z = image of the batch
    const int p0 = blockIdx.x * HEAD_MP;
    if (p0 >= n) return;
    const size_t rows_total = (size_t)gridDim.z * n_max;
    const int kbeg = blockIdx.y * (HEAD_K / HEAD_KSPLIT);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rA = weight_rsrc(trunk + (size_t)blockIdx.z * n_max * HEAD_K, n * HEAD_K);   // rows >= n -> 0
    const __amdgpu_buffer_rsrc_t rB = weight_rsrc(Bw, HEAD_K * 128);
    int offA[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int f = tid + 256 * r, row = f >> 5, c4 = f & 31;       // 32 consecutive float4 = one 512-byte row segment
        offA[r] = ((p0 + row) * HEAD_K + 4 * c4) * 4;
    }
    const int offB = ((kq * 128) + wave * 32 + m) * 16;
    const unsigned a_addr = lds_byte_addr(As) + (m * HEAD_AS + 4 * kq) * 4;
    f32x4 acc[4][2], stage[8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 8; ++r) stage[r] = buf_read4(rA, offA[r], kbeg * 4);
#pragma unroll 1
    for (int k0 = kbeg; k0 < kbeg + HEAD_K / HEAD_KSPLIT; k0 += HEAD_KC) {
        __syncthreads();                                              // the previous slab has been consumed
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int f = tid + 256 * r, row = f >> 5, c4 = f & 31;
            *reinterpret_cast<f32x4*>(&As[row * HEAD_AS + 4 * c4]) = stage[r];
        }
        __syncthreads();
        if (k0 + HEAD_KC < kbeg + HEAD_K / HEAD_KSPLIT) {
#pragma unroll
            for (int r = 0; r < 8; ++r) stage[r] = buf_read4(rA, offA[r], (k0 + HEAD_KC) * 4);
        }
        f32x4 fa[2][4], fb[2][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[0][i] = lds_read4(a_addr + i * 16 * HEAD_AS * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[0][j] = buf_read4(rB, offB + j * 256, k0 * 512);
#pragma unroll
        for (int g = 0; g < HEAD_KC / 16; ++g) {
            const int cur = g & 1, nxt = cur ^ 1;
            if (g + 1 < HEAD_KC / 16) {
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[nxt][i] = lds_read4(a_addr + i * 16 * HEAD_AS * 4 + (g + 1) * 64);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[nxt][j] = buf_read4(rB, offB + j * 256, (k0 + 16 * (g + 1)) * 512);
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][i][s4], fb[cur][j][s4], acc[i][j], 0, 0, 0);
        }
    }
    // acc[i][j][r]: patch p0 + 16 i + 4 (lane>>4) + r, channel 32 wave + 16 j + (lane & 15)
    const int g = lane >> 4;
    float* dst = partial + ((size_t)blockIdx.y * rows_total + (size_t)blockIdx.z * n_max) * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i)
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

static int cnn_launch(affnet_ctx* ctx, int kind, const float* packed, const float* patches, const float* lafs, const int32_t* ids,
                      const int32_t* count, int n_max, float* out, float* scratch, int dbg_layer, float* dbg_out, hipStream_t st,
                      bool mark_head = false) {
    if (kind < 0 || kind > 2) return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32: unknown net kind %d", kind);
    if (!packed || !out || n_max < 0 || (!patches && (!lafs || !ids))) return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32: null argument");
    if (!patches && !ctx->ws) return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32: sampling from the pyramid needs a bound workspace");
    if (dbg_layer < 0 && !scratch)
        return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32: d_scratch is required (HardNet n*(8192+512) floats, AffNet / OriNet n*144 floats)");
    if (n_max == 0) return AFFNET_OK;
    if (kind == AFFNET_NET_HARDNET && n_max > 65535) return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32: n_max=%d (HardNet head: max 65535 rows per image)", n_max);
    const NetLayout L = net_layout(kind);
    CnnArgs a;
    a.packed = packed; a.off = to_offsets(L); a.patches = patches; a.lafs = lafs; a.ids = ids; a.count = count; a.n_max = n_max;
    a.out = (dbg_layer < 0) ? scratch : out;              // trunk kernels: HardNet conv5 tensor / AffNet, OriNet head partials
    a.dbg_layer = dbg_layer; a.dbg_out = dbg_out; a.dbg_time = ctx->dbg_time;
    PyrSrc ps;
    aff_fill_pyr_src(ctx, &ps);
    const int B = patches ? 1 : ctx->B;                  // patch tensors are single-"image"; pyramid sampling covers the batch
    const dim3 grid(n_max, B);
    // (Tried and removed: two AffNet patches per persistent 16-wave workgroup in anti-phase - correct but 8 % slower, the
    // small-tile loops reach 85-90 % of the pipe rate with two waves per SIMD; 16-wave HardNet workgroups - slower too.)
#define TRUNK_LAUNCH(K) do { if (a.dbg_time || dbg_layer >= 0) hipLaunchKernelGGL((cnn32_trunk_kernel<K, 8, true>), grid, dim3(512), 0, st, a, ps); \
                             else hipLaunchKernelGGL((cnn32_trunk_kernel<K, 8, false>), grid, dim3(512), 0, st, a, ps); } while (0)
    if (kind == AFFNET_NET_AFFNET) TRUNK_LAUNCH(AFFNET_NET_AFFNET);
    else if (kind == AFFNET_NET_ORINET) TRUNK_LAUNCH(AFFNET_NET_ORINET);
    else TRUNK_LAUNCH(AFFNET_NET_HARDNET);
#undef TRUNK_LAUNCH
    AFF_LAUNCH_CHECK(ctx);
    if (kind != AFFNET_NET_HARDNET && dbg_layer < 0) {       // combine the per-wave head partials in `scratch`
        const dim3 hgrid(aff_cdiv(n_max, 256), B);
        if (kind == AFFNET_NET_AFFNET)
            hipLaunchKernelGGL((cnn16_finish_kernel<AFFNET_NET_AFFNET>), hgrid, dim3(256), 0, st, scratch, packed + L.head_b, count, n_max, out);
        else
            hipLaunchKernelGGL((cnn16_finish_kernel<AFFNET_NET_ORINET>), hgrid, dim3(256), 0, st, scratch, packed + L.head_b, count, n_max, out);
        AFF_LAUNCH_CHECK(ctx);
    }
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
    AFF_DEVICE(ctx);
    if (!ctx || !d_patches) return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32_forward: null argument");
    return cnn_launch(ctx, net_kind, d_packed, d_patches, nullptr, nullptr, d_count, n_max, d_out, d_scratch, -1, nullptr, (hipStream_t)stream);
}

extern "C" int affnet_cnn32_forward_pyr(affnet_ctx* ctx, int net_kind, const float* d_packed, const float* d_lafs, const int32_t* d_ids,
                                        const int32_t* d_count, int n_max, float* d_out, float* d_scratch, void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx) return AFFNET_ERR_INVALID;
    return cnn_launch(ctx, net_kind, d_packed, nullptr, d_lafs, d_ids, d_count, n_max, d_out, d_scratch, -1, nullptr, (hipStream_t)stream);
}

int aff_hardnet_forward_pyr_marked(affnet_ctx* ctx, const float* packed, const float* lafs, const int32_t* ids, const int32_t* count,
                                   int n_max, float* out, float* scratch, hipStream_t st) {
    return cnn_launch(ctx, AFFNET_NET_HARDNET, packed, nullptr, lafs, ids, count, n_max, out, scratch, -1, nullptr, st, true);
}

extern "C" int affnet_cnn32_debug_timing(affnet_ctx* ctx, unsigned long long* d_stamps) {
    if (!ctx) return AFFNET_ERR_INVALID;
    ctx->dbg_time = d_stamps;   // device buffer of n_patches * waves * 32 uint64, or NULL to switch the stamps off (this context only)
    return AFFNET_OK;
}

extern "C" int affnet_cnn32_debug_layer(affnet_ctx* ctx, int net_kind, const float* d_packed, const float* d_patch, int layer, float* d_out,
                                        void* stream) {
    AFF_DEVICE(ctx);
    if (!ctx || !d_patch || !d_out || layer < 0 || layer > 5) return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32_debug_layer: bad argument");
    return cnn_launch(ctx, net_kind, d_packed, d_patch, nullptr, nullptr, nullptr, 1, d_out, nullptr, layer, d_out, (hipStream_t)stream);
}

// ---- tuning aid: one HardNet layer's MFMA loop in isolation (no barriers, no epilogue), repeated ----------------------
template <int LAYER, int PROBE>
__global__ __launch_bounds__(512, 2) void cnn32_probe_kernel(const float* __restrict__ packed, NetOffsets off, int reps, float* __restrict__ out) {
    constexpr int CB = 32, NW = 8;
    __shared__ __attribute__((aligned(16))) float lds[TrunkLds<CB>::TOTAL];     // same footprint as the trunk: 1 workgroup / CU
    for (int i = threadIdx.x; i < TrunkLds<CB>::TOTAL; i += 512) lds[i] = 0.001f * (float)(i & 255);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float sink = 0.f;
    if (PROBE & 4) asm volatile("; accumulators in AGPRs" ::"a"(sink));     // any 'a' operand switches the MFMAs to their AGPR form
    for (int r = 0; r < reps; ++r) {
        if (LAYER == 1) {
            f32x4 acc[8][2], b0[1][2];
            prefetch_b0<NW, CB, 32, 8, 2, 1>(packed + off.w[1], b0, wave, lane);
            conv3x3_mfma<NW, CB, CB, LayC0, 1, 8, 2, 1, (PROBE & 11)>(lds, packed + off.w[1], b0, acc, wave, lane);
#pragma unroll
            for (int i = 0; i < 8; ++i) sink += acc[i][0][0] + acc[i][1][3];
        } else {
            f32x4 acc[4][1], b0[2][1];
            prefetch_b0<NW, 4 * CB, 8, 4, 1, 2>(packed + off.w[5], b0, wave, lane);
            conv3x3_mfma<NW, 4 * CB, 4 * CB, LayC4, 1, 4, 1, 2, (PROBE & 11)>(lds, packed + off.w[5], b0, acc, wave, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) sink += acc[i][0][0] + acc[i][0][3];
        }
    }
    if (sink == 12345.678f) out[0] = sink;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = sink;
}

// Same for the 16-channel trunks (AffNet / OriNet shapes, 79 KB of LDS -> two workgroups per CU, 128 VGPRs).
template <int LAYER, int PROBE>
__global__ __launch_bounds__(512, 4) void cnn16_probe_kernel(const float* __restrict__ packed, NetOffsets off, int reps, float* __restrict__ out) {
    constexpr int CB = 16, NW = 8;
    __shared__ __attribute__((aligned(16))) float lds[TrunkLds<CB>::TOTAL];
    for (int i = threadIdx.x; i < TrunkLds<CB>::TOTAL; i += 512) lds[i] = 0.001f * (float)(i & 255);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float sink = 0.f;
    for (int r = 0; r < reps; ++r) {
        if (LAYER == 3) {
            f32x4 acc[2][2], b0[2][2];
            prefetch_b0<NW, 2 * CB, 16, 2, 2, 2>(packed + off.w[3], b0, wave, lane);
            conv3x3_mfma<NW, 2 * CB, 2 * CB, LayC2, 1, 2, 2, 2, PROBE>(lds, packed + off.w[3], b0, acc, wave, lane);
            sink += acc[0][0][0] + acc[1][1][3] + acc[0][1][1] + acc[1][0][2];
        } else if (LAYER == 4) {    // conv3 again, 4 pixel tiles x 1 channel tile per wave: half the weight loads per MFMA
            f32x4 acc[4][1], b0[1][1];
            prefetch_b0<NW, 2 * CB, 16, 4, 1, 1>(packed + off.w[3], b0, wave, lane);
            conv3x3_mfma<NW, 2 * CB, 2 * CB, LayC2, 1, 4, 1, 1, PROBE>(lds, packed + off.w[3], b0, acc, wave, lane);
            sink += acc[0][0][0] + acc[1][0][3] + acc[2][0][1] + acc[3][0][2];
        } else {
            f32x4 acc[2][1], b0[2][1];
            prefetch_b0<NW, 4 * CB, 8, 2, 1, 2>(packed + off.w[5], b0, wave, lane);
            conv3x3_mfma<NW, 4 * CB, 4 * CB, LayC4, 1, 2, 1, 2, PROBE>(lds, packed + off.w[5], b0, acc, wave, lane);
            sink += acc[0][0][0] + acc[1][0][3];
        }
    }
    if (sink == 12345.678f) out[0] = sink;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = sink;
}

// layer: 1 (HardNet conv1, TM 8 x TN 2), 5 (HardNet conv5, TM 4 x TN 1, 2 groups / chunk) with HardNet's packed weights;
// 13 / 15 (AffNet conv3, TM 2 x TN 2 / conv5, TM 2 x TN 1) with AffNet's.  probe: PROBE bits; d_out: 2 floats.
extern "C" int affnet_cnn32_probe(const float* d_packed_hardnet, int layer, int probe, int reps, int n_blocks, float* d_out, void* stream) {
    if (!d_packed_hardnet || !d_out || probe < 0 || probe > 15) return AFFNET_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (layer == 13 || layer == 14 || layer == 15) {
        if (probe > 3) return AFFNET_ERR_INVALID;
        const NetOffsets off16 = to_offsets(net_layout(AFFNET_NET_AFFNET));
#define PROBE16(L, P) if (layer == 10 + L && probe == P) hipLaunchKernelGGL((cnn16_probe_kernel<L, P>), dim3(n_blocks), dim3(512), 0, st, d_packed_hardnet, off16, reps, d_out)
        PROBE16(3, 0); PROBE16(3, 1); PROBE16(3, 2); PROBE16(3, 3); PROBE16(4, 0); PROBE16(4, 1); PROBE16(4, 2); PROBE16(4, 3); PROBE16(5, 0); PROBE16(5, 1); PROBE16(5, 2); PROBE16(5, 3);
#undef PROBE16
        return hipGetLastError() == hipSuccess ? AFFNET_OK : AFFNET_ERR_HIP;
    }
    if (layer != 1 && layer != 5) return AFFNET_ERR_INVALID;
    const NetOffsets off = to_offsets(net_layout(AFFNET_NET_HARDNET));
#define PROBE_CASE(L, P) if (layer == L && probe == P) hipLaunchKernelGGL((cnn32_probe_kernel<L, P>), dim3(n_blocks), dim3(512), 0, st, d_packed_hardnet, off, reps, d_out)
    PROBE_CASE(1, 0); PROBE_CASE(1, 1); PROBE_CASE(1, 2); PROBE_CASE(1, 3); PROBE_CASE(1, 4); PROBE_CASE(1, 8); PROBE_CASE(1, 9);
    PROBE_CASE(5, 0); PROBE_CASE(5, 1); PROBE_CASE(5, 2); PROBE_CASE(5, 3); PROBE_CASE(5, 4); PROBE_CASE(5, 8); PROBE_CASE(5, 9);
#undef PROBE_CASE
    return hipGetLastError() == hipSuccess ? AFFNET_OK : AFFNET_ERR_HIP;
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
