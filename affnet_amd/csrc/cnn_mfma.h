// Shared MFMA building blocks of the 32x32-patch trunks (cnn32.hip) and the fully-convolutional AffNet (fullconv.hip):
// LDS activation layouts, weight fragment loads, the software-pipelined implicit-GEMM 3x3 convolution on
// v_mfma_f32_16x16x4_f32, conv0 on the matrix cores and the tile epilogues.  gfx950 only.
#pragma once
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
    size_t w_s3[6];         // AFFNET_ARITH_FP32_SPLIT3 (0 = none): conv weights once more as three bf16 terms, [tap][cin/32][term][kq][cout][8]
    size_t head_s3;         // HardNet only: the BN-folded head weights as three bf16 terms, [k/32][term][kq][n 128][8]  (k = pixel * 128 + channel)
    size_t w_h2[6];         // AFFNET_ARITH_FP32_SPLIT2H (0 = none): the same layers as TWO fp16 terms of 2^e * w (e per layer: the largest |w| of the layer lands in
                            // [2^13, 2^14)), same fragment order with 2 terms, followed by 4 floats whose first is 2^-e (the loop's output scale)
    size_t head_h2;         // HardNet only: the head weights as two fp16 terms, [k/32][term][kq][n 128][8] + 4 floats (2^-e first)
    size_t total;
};

// floats occupied by the split copy of a cin x cout 3x3 layer: 9 taps x (cin / 32) groups x 3 terms x 4 lane groups x cout x 8 bf16 (= 4 floats)
constexpr size_t s3_floats(int cin, int cout, int terms = 3) { return cin == 16 ? (size_t)5 * terms * 4 * cout * 4      // 16 input channels: two taps per k = 32 step, 9 taps in 5 steps
                                                                                 : (size_t)9 * (cin / 32) * terms * 4 * cout * 4; }
#define H2_TAIL 4               // floats behind a two-term copy: [0] = 2^-e, the power of two that undoes the copy's scale (exact)
#define S3_LAYER_MASK 0x3E      // which layers have a split copy / run on split operands (bit i = conv i): conv1 .. conv5 of HardNet

static inline NetLayout net_layout(int kind) {
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
    else if (kind == AFFNET_NET_AFFNET_FULLCONV) { off += 8 * 64 * 32; L.head_b = off; off += 4; }   // [ky][c / 16][(c / 4) % 4][n = o * 8 + kx (32)][c % 4]
    else if (kind == AFFNET_NET_ORINET) { off += 2 * 4096; L.head_b = off; off += 4; }
    else { off += (size_t)HEAD_K * 128; L.head_b = off; off += 128; }
    for (int i = 0; i < 6; ++i) {
        L.w_s3[i] = 0;
        const bool has = (kind == AFFNET_NET_HARDNET && ((S3_LAYER_MASK >> i) & 1)) ||
                         ((kind == AFFNET_NET_AFFNET || kind == AFFNET_NET_ORINET || kind == AFFNET_NET_AFFNET_FULLCONV) && i >= 1);     // 16-channel trunks: conv1 .. conv5
        if (has) { L.w_s3[i] = off; off += s3_floats(L.cin[i], L.cout[i]); }
    }
    L.head_s3 = 0;
    if (kind == AFFNET_NET_HARDNET) { L.head_s3 = off; off += (size_t)HEAD_K * 128 * 3 / 2; }
    for (int i = 0; i < 6; ++i) {
        L.w_h2[i] = 0;
        if (L.w_s3[i]) { L.w_h2[i] = off; off += s3_floats(L.cin[i], L.cout[i], 2) + H2_TAIL; }
    }
    L.head_h2 = 0;
    if (kind == AFFNET_NET_HARDNET) { L.head_h2 = off; off += (size_t)HEAD_K * 128 + H2_TAIL; }
    L.total = off;
    return L;
}

struct NetOffsets {        // device-side copy of the offsets (by-value kernel argument)
    int w[6], b[6], head_w, head_b;
    int w_s3[6];           // the split copy of the ACTIVE arithmetic mode (three bf16 terms or two fp16 terms)
    int head_s3;
};

static inline NetOffsets to_offsets(const NetLayout& L, int arith = AFFNET_ARITH_FP32_SPLIT3) {
    NetOffsets o;
    const bool h2 = arith == AFFNET_ARITH_FP32_SPLIT2H;
    for (int i = 0; i < 6; ++i) { o.w[i] = (int)L.w_off[i]; o.b[i] = (int)L.b_off[i]; }
    o.head_w = (int)L.head_w; o.head_b = (int)L.head_b;
    for (int i = 0; i < 6; ++i) o.w_s3[i] = (int)(h2 ? L.w_h2[i] : L.w_s3[i]);
    o.head_s3 = (int)(h2 ? L.head_h2 : L.head_s3);
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

// Same with the lane index made opaque: the compiler then recomputes 4 * (lane >> 4) here instead of keeping it in a register across the
// whole kernel (at the 128-VGPR cap of the 16-channel split trunks that one value was spilled to scratch).
template <int NW, int HOUT, int TM, int TN>
__device__ __forceinline__ void prefetch_bias_fresh(const float* __restrict__ bias, f32x4 (&bv)[TN], int wave, int lane) {
    asm volatile("" : "+v"(lane));
    prefetch_bias<NW, HOUT, TM, TN>(bias, bv, wave, lane);
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

// ---- AFFNET_ARITH_FP32_SPLIT3: the same contraction on split operands (fp32 = three bf16 terms) --------------------------------------
// (round 3's loops on separate term planes, LayB / conv3x3_mfma_s3p: kept for tools/probes/s3_loop_probe.hip, which measures them against
// the round-4 loops the trunks use: LayQ / conv3x3_mfma_s3q below)
// x = x0 + x1 + x2 with every term rounded to bf16 is exact for a 24-bit significand, every bf16 x bf16 product is exact in the fp32
// accumulator of v_mfma_f32_16x16x32_bf16, and the six products with i + j <= 2 carry an fp32 product to 2^-25 relative - at 16x the
// rate of the fp32 MFMA.  Weights are split at pack time: Ws [tap][CIN/32][term][kq][COUT][8 bf16]; activations are split ONCE, by the
// epilogue of the layer that produces them, into three bf16 planes (LayB) - the first version kept them fp32 in LDS and split each
// fragment in registers when it was read (9 taps x NG channel groups times per element): 4.5 VALU instructions per MFMA, matrix pipe
// 49 % busy (round 3, removed).  Accumulator layout = conv3x3_mfma's.  Selected per context (affnet_set_arith).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Exact remainder of a pair after its bf16 roundings u = (bf16(v.x), bf16(v.y)): v_dot2c_f32_bf16 with the constant pair (-1, 0) /
// (0, -1) is  v.x - float(u.lo) + 0 * float(u.hi)  in one instruction; the difference is representable, so the rounding mode does not
// matter.  Checked bit-for-bit against the shift / subtract form on 2^24 random bit patterns incl. denormals - they differ only where
// bf16(v) overflows to inf.  The constants come from s_mov through an asm so that the compiler cannot fold them into an inline
// operand: it encodes the packed bf16 (-1, 0) as the inline constant -1.0, which the hardware does not read as bf16 (the low-half
// remainders came out unchanged).
// Measured alternatives when the split still sat inside the MFMA loops (HardNet conv1, cycles of the slower wave of a SIMD): shift /
// mask + v_pk_add_f32 54.9 k, this form 48.8 k, shift / mask + two unpacked v_sub_f32 50.2 k (pre-split: 38.3 k incl. a second conv0
// pass).  tools/probes/mfma_valu_overlap.hip shows why none of them hides under the matrix pipe: with two waves per SIMD, 12 bf16 MFMAs + 36 VALU instructions take 528-553 cycles for v_fmac / v_cvt_pk /
// v_lshlrev (MFMAs alone 428, the VALU alone 190-330) and ~1000 cycles for v_dot2c / v_pk_add_f32 - VALU work next to bf16 MFMAs is at
// best half hidden, so the remedy is not to have it in the loop (pre-split layouts below).
__device__ __forceinline__ void split_remainder(f32x2& v, unsigned u) {
    unsigned c0, c1;
    asm("s_mov_b32 %0, 0xbf80" : "=s"(c0));
    asm("s_mov_b32 %0, 0xbf800000" : "=s"(c1));
    const bf16x2 b = __builtin_bit_cast(bf16x2, u);
    v.x = __builtin_amdgcn_fdot2_f32_bf16(b, __builtin_bit_cast(bf16x2, c0), v.x, false);
    v.y = __builtin_amdgcn_fdot2_f32_bf16(b, __builtin_bit_cast(bf16x2, c1), v.y, false);
}

// ---- AFFNET_ARITH_FP32_SPLIT2H: fp32 = two fp16 terms, three products ------------------------------------------------------------------
// x ~ h + l, h = fp16(x), l = fp16(x - h) (both round-to-nearest-even; x - h is exact in fp32): 11 + 11 bits + the remainder's sign = 23 of
// fp32's 24 significand bits, |x - h - l| <= 2^-23 |x| for |x| >= 2^-2 and <= 2^-25 ABSOLUTE below (activations are not scaled: the low term is then a subnormal fp16).  w a = w_h a_h + w_l a_h + w_h a_l (+ w_l a_l <= 2^-24, dropped) on v_mfma_f32_16x16x32_f16: every fp16 x fp16 product is exact
// in the fp32 accumulator and subnormal fp16 inputs are honoured (tools/probes/f16_split_probe.hip), so small activations keep an ABSOLUTE
// error of 2^-25.  Half the matrix instructions of the three-term bf16 scheme; the loops and epilogues are the same code with TERMS = 2, on
// a layout of their own (LayR below: 16-byte pixels, two thirds of LayQ's bytes, conflict free for every reader).
// Weights are packed times 2^e per layer (net_layout: w_h2) so that both terms sit in fp16's normal range; the loop multiplies its sums by
// 2^-e (exact).  Activations are not scaled: |a| < 65504 required (include/affnet_hip.h).
// The pair split is written in assembly: this compiler's own lowering of  <2 x float> -> <2 x half>  followed by element reads uses the LOW
// half for both elements (v_cvt_pk_f16_f32, then v_perm / v_cvt_f32_f16 of the low half twice - found by the probe: every odd element wrong).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split_h2_pair(f32x2 v, unsigned& hi, unsigned& lo) {
    float r0, r1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v.x), "v"(v.y));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(v.x));                      // x - float(hi.lo16): exact
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(v.y));       // y - float(hi.hi16)
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(r0), "v"(r1));
}

// Split four fp32 values (a lane's four consecutive channels) into TERMS terms and store term t at dst + 16 t (8 bytes each: half a cell)
template <int TERMS, int TSTEP = 16>
__device__ __forceinline__ void split_store4(char* dst, f32x4 v) {
    f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
    if constexpr (TERMS == 2) {
        unsigned h0, l0, h1, l1;
        split_h2_pair(lo, h0, l0);
        split_h2_pair(hi, h1, l1);
        *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(dst + TSTEP) = make_uint2(l0, l1);
    } else {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const unsigned u0 = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2));
            const unsigned u1 = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2));
            *reinterpret_cast<uint2*>(dst + t * TSTEP) = make_uint2(u0, u1);
            if (t < 2) { split_remainder(lo, u0); split_remainder(hi, u1); }
        }
    }
}

// one matrix instruction of the split arithmetic: fragments travel as 16 bytes, the term count picks the operand type
template <int TERMS>
__device__ __forceinline__ f32x4 split_mfma(bf16x8 w, bf16x8 a, f32x4 c) {
    if constexpr (TERMS == 2) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, c, 0, 0, 0);
}

// ---- pre-split activations: a layer's output stored as three bf16 planes -------------------------------------------------------
// Splitting in the reading loop is redundant: the NG waves that share a pixel tile (different output channels) each split the same
// fragment - 4.5 VALU instructions per MFMA in the first version (PMC: 8.0e9 VALU vs 1.8e9 MFMA instructions per 32-image HardNet
// launch, VALU and matrix pipe each ~50 % busy, one after the other).  Where the buffer allows (1.5x the fp32 size), the EPILOGUE
// splits every output element once and the next layer reads ready bf16 fragments: element (term t, channel c, y, x) of an H x H layer
// sits at  t * TS + (c / 8) * GS + ((y + 1) * WP + x + 1) * 16 + (c % 8) * 2  bytes - one ds_read_b128 = the 8 channels of one term
// of one pixel = a lane's B fragment of a k = 32 step.
// Bank conflicts (ds_read_b128 is served in four groups of 16 lanes - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... - over a 256-byte
// bank window): pixels {0-3, 12-15} of lane quarter kq and pixels {4-11} of quarter kq + 1 share a group, so the 8-channel group stride GS
// decides whether their 16-byte cells collide.  The first version had GS = 64 (mod 256) everywhere: 56 % of the LDS-array cycles of the
// split HardNet launch were conflict cycles (SQ_LDS_BANK_CONFLICT 2.9e9 of SQ_LDS_IDX_ACTIVE 5.1e9).  GS is rounded up to a multiple of
// 256 plus GREM, chosen per READER:  16 pixels contiguous (stride-1 reader of a 16 / 32-wide layer): GREM = 0;  2 rows x 8 pixels (8-wide
// layer, row stride 256 bytes = WP 16): GREM = 128;  16 pixels at a 32-byte pitch (stride-2 reader of a 32-wide layer): GREM = 16.
template <int H_, int W_, int WP_, int C_, int GREM_ = 0>
struct LayB {
    static constexpr int H = H_, W = W_, WP = WP_, C = C_;      // H rows x W columns (+ a one-cell halo), row stride WP cells
    static constexpr int GS = (((H_ + 2) * WP_ * 16 + 255) / 256) * 256 + GREM_;      // bytes per 8-channel group
    static constexpr int TS = (C_ / 8) * GS;              // bytes per term
    static constexpr int BYTES = 3 * TS;
};

template <typename L, int NTHR>
__device__ __forceinline__ void zero_halo_b(float* act, int tid = threadIdx.x) {
    constexpr int H = L::H, W = L::W, CELLS = 2 * (W + 2) + 2 * H, PLANES = 3 * (L::C / 8);
    char* base = reinterpret_cast<char*>(act);
    for (int i = tid; i < PLANES * CELLS; i += NTHR) {
        const int g = i / CELLS, e = i - g * CELLS;      // g = term * (C / 8) + group: planes are contiguous (TS = groups * GS)
        int y, x;
        if (e < W + 2) { y = 0; x = e; }
        else if (e < 2 * (W + 2)) { y = H + 1; x = e - (W + 2); }
        else { const int r = e - 2 * (W + 2); y = 1 + (r >> 1); x = (r & 1) ? W + 1 : 0; }
        *reinterpret_cast<f32x4*>(base + (size_t)g * L::GS + (y * L::WP + x) * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

// (+ bias,) ReLU, split into three bf16 terms and store ONE pixel tile: the lane's pixel sits at byte offset `cell` of a plane, its
// four channels are 4 g .. 4 g + 3 of channel tiles nt0 .. nt0 + TN - 1
template <typename LO, int TN, bool ADD_BIAS = true>
__device__ __forceinline__ void split_store_tile(char* base, int cell, int nt0, const f32x4 (&bias)[TN], const f32x4 (&acc)[TN], int g) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        f32x4 v = acc[j];
        if (ADD_BIAS) v += bias[j];
        v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
        const int c0 = (nt0 + j) * 16 + 4 * g;                           // first of this lane's 4 channels
        char* dst = base + (c0 >> 3) * LO::GS + cell + (c0 & 4) * 2;
        f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const unsigned u0 = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2));
            const unsigned u1 = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2));
            *reinterpret_cast<uint2*>(dst + t * LO::TS) = make_uint2(u0, u1);
            if (t < 2) { split_remainder(lo, u0); split_remainder(hi, u1); }
        }
    }
}

// Epilogue of a layer whose output goes to a pre-split layout LO: the waves' tiles cover HT rows (default: all of LO) starting at row_off
template <int COUT, typename LO, int TM, int TN, int HT = LO::H>
__device__ __forceinline__ void store_tiles_split(float* act, const f32x4 (&bias)[TN], const f32x4 (&acc)[TM][TN], int wave, int lane, int row_off = 0) {
    constexpr int W = LO::W;
    constexpr int MT = HT * W / 16, MG = MT / TM;
    const int mg = wave % MG, ng = wave / MG;
    const int n = lane & 15, g = lane >> 4;
    char* base = reinterpret_cast<char*>(act);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = (mg * TM + i) * 16 + n;
        const int oy = p / W, ox = p - oy * W;
        split_store_tile<LO, TN>(base, ((oy + row_off + 1) * LO::WP + ox + 1) * 16, ng * TN, bias, acc[i], g);
    }
}

// The contraction on a PRE-SPLIT input (layout LI = LayB): no VALU in the loop - three ds_read_b128 per pixel tile and k = 32 step.
template <int NW, int CIN, int COUT, typename LI, int STRIDE, int TM, int TN>
__device__ __forceinline__ void conv3x3_mfma_s3p(const float* act, const float* __restrict__ Ws, f32x4 (&acc)[TM][TN], int wave, int lane, bool alt_prio) {
    constexpr int HOUT = LI::H / STRIDE, WOUT = LI::W / STRIDE;
    constexpr int MT = HOUT * WOUT / 16, NT = COUT / 16;
    constexpr int MG = MT / TM, NG = NT / TN;
    constexpr int NG32 = CIN / 32, NS = 9 * NG32;
    static_assert(MG * NG == NW && CIN % 32 == 0 && LI::C == CIN, "bad tiling for the split-operand loop");
    const int mg = wave % MG, ng = wave / MG;
    const int m = lane & 15, kq = lane >> 4;
    int a_lane;                                                            // bytes
    {
        const int p = mg * TM * 16 + m;
        const int oy = p / WOUT, ox = p - oy * WOUT;
        a_lane = kq * LI::GS + ((oy * STRIDE) * LI::WP + ox * STRIDE) * 16;
    }
    const unsigned a_addr0 = lds_byte_addr(act) + a_lane;
    auto a_imm = [](int i) { return 16 * (WOUT == 8 ? i * 2 * STRIDE * LI::WP : (((i * 16) / WOUT) * STRIDE * LI::WP + ((i * 16) % WOUT) * STRIDE)); };
    constexpr int WS_FLOATS = 9 * (CIN / 32) * 3 * 4 * COUT * 4;
    const __amdgpu_buffer_rsrc_t wrsrc = weight_rsrc(Ws, WS_FLOATS);
    const int w_lane = (kq * COUT + ng * TN * 16 + m) * 16;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 w[2][3][TN];
    auto load_w = [&](bf16x8 (&dst)[3][TN], int s) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                dst[t][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_lane + j * 256, ((s * 3 + t) * 4 * COUT) * 16, 0));
    };
    // Rotating schedule: the fragments of pixel tile i for step s + 1 are requested right after tile i's MFMAs of step s were issued
    // (same registers - the matrix pipe has read them by then), so every ds_read has the other tiles' MFMAs (>= 100 issue cycles) to
    // complete.  The first version read a step's 3 TM fragments at its top and waited: a wave stalled once per step and the two waves
    // of a SIMD drifted apart (conv3: faster wave 30 k cycles, slower 35 k; matrix-pipe floor 27.6 k).
    auto frag_addr = [&](int s) {
        const int tap = s / NG32, G = s - tap * NG32;
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
        return a_addr0 + 4 * G * LI::GS + (ky * LI::WP + kx) * 16;
    };
    bf16x8 a[TM][3];
    auto read_tile = [&](unsigned ab, int i) {
#pragma unroll
        for (int t = 0; t < 3; ++t) a[i][t] = __builtin_bit_cast(bf16x8, lds_read4(ab + a_imm(i) + t * LI::TS));
    };
    auto step = [&](const bf16x8 (&wc)[3][TN], int s_next) {
        if (alt_prio) {                       // the two waves of a SIMD (w, w + 4) take turns at the higher priority, one step each
            if ((s_next ^ (wave >> 2)) & 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        const unsigned ab = frag_addr(s_next);
        constexpr int TW[6] = {0, 1, 2, 0, 1, 0}, TA[6] = {2, 1, 0, 1, 0, 0};      // smallest terms first
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[TW[t]][j], a[i][TA[t]], acc[i][j], 0, 0, 0);
            read_tile(ab, i);
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * TN, 0);      // this tile's MFMAs ...
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);           // ... then its three reads for the next step
        }
    };
    load_w(w[0], 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) read_tile(frag_addr(0), i);
#pragma unroll 1
    for (int s = 0; s + 1 < NS; s += 2) {
        load_w(w[1], s + 1);
        step(w[0], s + 1);
        load_w(w[0], (s + 2 < NS) ? s + 2 : s + 1);
        step(w[1], (s + 2 < NS) ? s + 2 : s + 1);                        // the last step re-reads its own fragments (never used)
    }
    if (NS & 1) step(w[0], NS - 1);
    if (alt_prio) __builtin_amdgcn_s_setprio(0);
}

// Pre-split input with 16 channels (AffNet / OriNet conv1, conv2): one k = 32 step = two taps x 16 channels
// (lane group kq: tap 2 s + (kq >> 1), channel group kq & 1), fragments read ready from a LayB layout with the rotating schedule.
template <int NW, int COUT, typename LI, int STRIDE, int TM, int TN>
__device__ __forceinline__ void conv3x3_mfma_s3p_c16(const float* act, const float* __restrict__ Ws, f32x4 (&acc)[TM][TN], int wave, int lane, bool alt_prio) {
    constexpr int HOUT = LI::H / STRIDE, WOUT = LI::W / STRIDE;
    constexpr int MT = HOUT * WOUT / 16, NT = COUT / 16;
    constexpr int MG = MT / TM, NG = NT / TN;
    static_assert(MG * NG == NW && LI::C == 16, "bad tiling for the split-operand loop");
    const int mg = wave % MG, ng = wave / MG;
    const int m = lane & 15, kq = lane >> 4;
    int a_lane;                                                            // bytes
    {
        const int p = mg * TM * 16 + m;
        const int oy = p / WOUT, ox = p - oy * WOUT;
        a_lane = (kq & 1) * LI::GS + ((oy * STRIDE) * LI::WP + ox * STRIDE) * 16;
    }
    const unsigned a_addr0 = lds_byte_addr(act) + a_lane;
    auto a_imm = [](int i) { return 16 * (WOUT == 8 ? i * 2 * STRIDE * LI::WP : (((i * 16) / WOUT) * STRIDE * LI::WP + ((i * 16) % WOUT) * STRIDE)); };
    constexpr int WS_FLOATS = 5 * 3 * 4 * COUT * 4;
    const __amdgpu_buffer_rsrc_t wrsrc = weight_rsrc(Ws, WS_FLOATS);
    const int w_lane = (kq * COUT + ng * TN * 16 + m) * 16;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto frag_addr = [&](int s) {
        const int ta = 2 * s, tb = (2 * s + 1 < 9) ? 2 * s + 1 : 8;                   // the pad half-step re-reads tap 8 (times zero weights)
        const int off_a = ((ta / 3) * LI::WP + ta % 3) * 16, off_b = ((tb / 3) * LI::WP + tb % 3) * 16;
        return a_addr0 + ((kq >> 1) ? off_b : off_a);
    };
    bf16x8 a[TM][3];
    auto read_tile = [&](unsigned ab, int i) {
#pragma unroll
        for (int t = 0; t < 3; ++t) a[i][t] = __builtin_bit_cast(bf16x8, lds_read4(ab + a_imm(i) + t * LI::TS));
    };
#pragma unroll
    for (int i = 0; i < TM; ++i) read_tile(frag_addr(0), i);
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        bf16x8 w[3][TN];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                w[t][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_lane + j * 256, ((s * 3 + t) * 4 * COUT) * 16, 0));
        const unsigned abn = frag_addr(s + 1 < 5 ? s + 1 : s);
        if (alt_prio) {
            if ((s ^ (wave >> 2)) & 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        constexpr int TW[6] = {0, 1, 2, 0, 1, 0}, TA[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[TW[t]][j], a[i][TA[t]], acc[i][j], 0, 0, 0);
            if (s + 1 < 5) {
                read_tile(abn, i);
                __builtin_amdgcn_sched_group_barrier(0x008, 6 * TN, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            }
        }
    }
    if (alt_prio) __builtin_amdgcn_s_setprio(0);
}

// ---- round 4: term-interleaved pre-split layout + term-major MFMA order ------------------------------------------------------------
// LayB keeps the three bf16 planes of a layer TS bytes apart.  2 * TS exceeds the 16-bit DS offset field for every layer here, so each
// fragment read of the loops above needed its own v_add_u32 (3 per pixel tile and step) and the hazard s_nops the compiler puts between an
// MFMA and a VALU write of a register the MFMA reads - inside the MFMA stream.  LayQ interleaves the terms INSIDE the pixel cell:
// element (term t, channel c, y, x) sits at  (c / 8) * GS + ((y + 1) * WP + x + 1) * 48 + t * 16 + (c % 8) * 2  bytes, so the three
// terms of a lane's fragment are one address + immediates 0 / 16 / 32, the tiles of a wave further immediates, and a k = 32 step costs ONE
// v_add (step offset, an SGPR) instead of 3 TM.  Bank behaviour of the 48-byte pixel pitch (ds_read_b128, 16-lane service groups, tools/
// bank model in DESIGN.md): 16 pixels of a row at 48 B cover the 64 banks exactly once (12 m mod 64 are the sixteen multiples of 4), so
// the stride-1 readers are conflict free with GS = 0 (mod 256); 16 pixels at a 96 B pitch (stride-2 reader of a 32-wide layer) with
// GS = 16 (mod 256); two rows of 8 pixels with a 768 B row pitch (8-wide layers, WP = 16) with GS = 128 (mod 256).  Only the stride-2
// reader of the 16-wide layers (conv4) keeps 2-way conflicts: its row pitch 2 * 18 * 48 B would have to be a multiple of 256 B (WP = 24:
// 166 KB for the 64-channel layers).
template <int H_, int W_, int WP_, int C_, int GREM_ = 0, int TERMS_ = 3>
struct LayQ {
    static constexpr int H = H_, W = W_, WP = WP_, C = C_;      // H rows x W columns (+ a one-cell halo), row stride WP cells of 48 bytes
    static constexpr int GREM = GREM_;
    static constexpr int TERMS = TERMS_;                        // 3 bf16 terms or 2 fp16 terms per element (the cell keeps three 16-byte slots either way)
    static constexpr int CELL = 48;
    // address pitches (bytes) every user goes through: right neighbour, the pixel below, next term of the same pixel; 16-byte slots per pixel
    static constexpr int PIXB = CELL, ROWB = WP_ * CELL, TSTEP = 16, SLOTS = 3;
    __device__ __host__ static constexpr int at(int y, int x) { return y * ROWB + x * PIXB; }      // (y, x) counted from the top-left halo cell
    static constexpr int GS = (((H_ + 2) * ROWB + 255) / 256) * 256 + GREM_;      // bytes per 8-channel group
    static constexpr int BYTES = (C_ / 8) * GS;
};

// Two-term layout with a 16-byte pixel pitch (AFFNET_ARITH_FP32_SPLIT2H): the hi cells and the lo cells of a ROW sit side by side - element (term t, channel c, y, x) at
//   (c / 8) * GS + (y + 1) * ROWB + t * (WP * 16) + (x + 1) * 16 + (c % 8) * 2,   ROWB = 2 * WP * 16 bytes
// - two thirds of LayQ's bytes (HardNet's conv0 output of a WHOLE patch fits: 148 KB), the second term still one immediate away, and with the pixel pitch a
// divisor of the 256-byte bank window every reader of the trunks can be made conflict free (ds_read_b128 service groups {0-3, 12-15, 20-27} ..., 16-byte slots
// mod 16; kq = lane quarter, its 8-channel group GS bytes further):
//   stride 1, one-row tiles (32- / 16-wide layers): pixels 0-3, 12-15 of quarter kq and 4-11 of kq + 1 fill the 16 slots once with GS = 0 (mod 256);
//   stride 1, two rows x 8 pixels (8-wide layers): the second row must start 8 slots later: ROWB / 16 = 2 WP = 8 (mod 16): WP = 12, GS = 0 (mod 256);
//   stride 2, one-row tiles (32-wide input): pixel pitch 32 B = even slots only; quarter kq + 1 takes the odd ones with GS = 16 (mod 256);
//   stride 2, two rows x 8 (16-wide input): rows 2 apart must start on the same slot: 4 WP = 0 (mod 16): WP = 20, and GS = 16 (mod 256) - the reader that
//   kept 2-way conflicts in LayQ (conv4).
// HALLOC_ = rows the group stride is sized for (default H + 2): a 16-row VIEW of a 32-row layout (the half-patch loops of conv1 / conv2) has HALLOC_ = 34
template <int H_, int W_, int WP_, int C_, int GREM_ = 0, int HALLOC_ = H_ + 2>
struct LayR {
    static constexpr int H = H_, W = W_, WP = WP_, C = C_, GREM = GREM_;
    static constexpr int TERMS = 2, SLOTS = 2;
    static constexpr int PIXB = 16, ROWB = 2 * WP_ * 16, TSTEP = WP_ * 16;
    __device__ __host__ static constexpr int at(int y, int x) { return y * ROWB + x * PIXB; }
    static constexpr int GS = ((HALLOC_ * ROWB + 255) / 256) * 256 + GREM_;
    static constexpr int BYTES = (C_ / 8) * GS;
};

template <typename L, int NTHR>
__device__ __forceinline__ void zero_halo_q(float* act, int tid = threadIdx.x) {
    constexpr int H = L::H, W = L::W, CELLS = 2 * (W + 2) + 2 * H, GROUPS = L::C / 8;
    char* base = reinterpret_cast<char*>(act);
    for (int i = tid; i < GROUPS * CELLS * L::SLOTS; i += NTHR) {    // one 16-byte store = one term of one halo cell
        const int t = i % L::SLOTS, ce = i / L::SLOTS;
        const int g = ce / CELLS, e = ce - g * CELLS;
        int y, x;
        if (e < W + 2) { y = 0; x = e; }
        else if (e < 2 * (W + 2)) { y = H + 1; x = e - (W + 2); }
        else { const int r = e - 2 * (W + 2); y = 1 + (r >> 1); x = (r & 1) ? W + 1 : 0; }
        *reinterpret_cast<f32x4*>(base + (size_t)g * L::GS + L::at(y, x) + t * L::TSTEP) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

// (+ bias,) ReLU, split into three bf16 terms, store ONE pixel tile into a LayQ layout: `cell` = byte offset of the lane's pixel cell
template <typename LO, int TN, bool ADD_BIAS = true>
__device__ __forceinline__ void split_store_tile_q(char* base, int cell, int nt0, const f32x4 (&bias)[TN], const f32x4 (&acc)[TN], int g) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        f32x4 v = acc[j];
        if (ADD_BIAS) v += bias[j];
        v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
        const int c0 = (nt0 + j) * 16 + 4 * g;                           // first of this lane's 4 channels
        char* dst = base + (c0 >> 3) * LO::GS + cell + (c0 & 4) * 2;
        split_store4<LO::TERMS, LO::TSTEP>(dst, v);
    }
}

template <int COUT, typename LO, int TM, int TN, int HT = LO::H>
__device__ __forceinline__ void store_tiles_split_q(float* act, const f32x4 (&bias)[TN], const f32x4 (&acc)[TM][TN], int wave, int lane, int row_off = 0) {
    constexpr int W = LO::W;
    constexpr int MT = HT * W / 16, MG = MT / TM;
    const int mg = wave % MG, ng = wave / MG;
    const int n = lane & 15, g = lane >> 4;
    char* base = reinterpret_cast<char*>(act);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = (mg * TM + i) * 16 + n;
        const int oy = p / W, ox = p - oy * W;
        split_store_tile_q<LO, TN>(base, LO::at(oy + row_off + 1, ox + 1), ng * TN, bias, acc[i], g);
    }
}

template <int N> struct IntC { static constexpr int v = N; };

// Weight fragments of one k = 32 step of a split layer: [term][channel tile] (two-term arithmetic leaves w[2] unused: no registers)
template <int TN>
struct S3W {
    bf16x8 w[3][TN];
};

// Lane part of the weight address of a split layer (bytes): the step offset is an SGPR, the tile offset an immediate
template <int NW, int COUT, int MG, int TN>
__device__ __forceinline__ int s3_w_lane(int wave, int lane) {
    return (((lane >> 4) * COUT) + (wave / MG) * TN * 16 + (lane & 15)) * 16;
}
template <int COUT, int TN, int TERMS = 3>
__device__ __forceinline__ void s3_load_w(S3W<TN>& dst, __amdgpu_buffer_rsrc_t wrsrc, int w_lane, int s) {
#pragma unroll
    for (int t = 0; t < TERMS; ++t)
#pragma unroll
        for (int j = 0; j < TN; ++j)
            dst.w[t][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_lane + j * 256, ((s * TERMS + t) * 4 * COUT) * 16, 0));
}
// The first step's weights of a layer do not depend on the activations: requested BEFORE the barriers / epilogue of the previous layer
// (1 - 2 k cycles of L2 latency under load, otherwise paid with an idle matrix pipe at the head of every loop).
template <int NW, int CIN, int COUT, int HOUT_TILES, int TM, int TN, int TERMS = 3>
__device__ __forceinline__ void s3_prefetch_w0(const float* __restrict__ Ws, S3W<TN>& w0, int wave, int lane) {
    constexpr int MG = HOUT_TILES / TM;
    constexpr int WS_FLOATS = (CIN == 16 ? 5 : 9 * (CIN / 32)) * TERMS * 4 * COUT * 4;
    s3_load_w<COUT, TN, TERMS>(w0, weight_rsrc(Ws, WS_FLOATS), s3_w_lane<NW, COUT, MG, TN>(wave, lane), 0);
}

// The contraction on a term-interleaved pre-split input (LI = LayQ).  MFMA order is TERM-MAJOR: a term pair (w_i, a_j) runs over all
// TM x TN tiles of the wave before the next pair starts, so consecutive MFMAs never share an accumulator (the tile-major order of
// conv3x3_mfma_s3p chained six dependent MFMAs per accumulator, two chains interleaved) and the activation terms retire one after the
// other: pairs (w0 a0, w1 a0, w2 a0) - a0 dead, its registers take the NEXT step's a0 - (w0 a1, w1 a1) - a1 reloaded - (w0 a2) - a2
// reloaded.  Every fragment read has >= 24 MFMAs (a0), 32 (a1), 40 (a2) of this wave between issue and use.  The order of the six pairs
// inside a step does not matter numerically: the fp32 accumulator already carries the sum over all previous steps.
// C16 = true: 16 input channels (AffNet / OriNet conv1, conv2): one k = 32 step = two taps x 16 channels (lane quarter kq: tap 2 s + (kq >> 1),
// channel group kq & 1), 5 steps, the pad half-step re-reads tap 8 against zero weights.
// PROBE (tools/probes/s3_loop_probe.hip only): bit 0 = no weight loads inside the loop, bit 1 = no fragment reloads, bits 4.. = s_nop pacing
template <int NW, int CIN, int COUT, typename LI, int STRIDE, int TM, int TN, int PROBE = 0>
__device__ __forceinline__ void conv3x3_mfma_s3q(const float* act, const float* __restrict__ Ws, const S3W<TN>& w_first, f32x4 (&acc)[TM][TN],
                                                 int wave, int lane, int variant) {
    const bool alt_prio = (variant & 1) != 0;      // affnet_debug_split3_variant bits: 0 = alternating wave priorities, 1 = (A/B only) skip the NaN -> inf step of the two-term loops
    constexpr bool C16 = (CIN == 16);
    constexpr int TERMS = LI::TERMS;                                       // 3: bf16 terms, six products; 2: fp16 terms, three products (w_h a_h, w_l a_h, w_h a_l)
    constexpr int HOUT = LI::H / STRIDE, WOUT = LI::W / STRIDE;
    constexpr int MT = HOUT * WOUT / 16, NT = COUT / 16;
    constexpr int MG = MT / TM, NG = NT / TN;
    constexpr int NG32 = C16 ? 1 : CIN / 32, NS = C16 ? 5 : 9 * NG32;
    static_assert(MG * NG == NW && (C16 || CIN % 32 == 0) && LI::C == CIN && (TERMS == 2 || TERMS == 3), "bad tiling for the split-operand loop");
    const int mg = wave % MG;
    const int m = lane & 15, kq = lane >> 4;
    int a_lane;                                                            // bytes
    {
        const int p = mg * TM * 16 + m;
        const int oy = p / WOUT, ox = p - oy * WOUT;
        a_lane = (C16 ? (kq & 1) : kq) * LI::GS + LI::at(oy * STRIDE, ox * STRIDE);
    }
    const unsigned a_addr0 = lds_byte_addr(act) + a_lane;
    auto a_imm = [](int i) { return WOUT == 8 ? LI::at(i * 2 * STRIDE, 0) : LI::at(((i * 16) / WOUT) * STRIDE, ((i * 16) % WOUT) * STRIDE); };
    static_assert(LI::at(TM * 2 * STRIDE, 0) + 2 * LI::TSTEP < 65536, "tile immediates must fit the DS offset field");
    constexpr int WS_FLOATS = (C16 ? 5 : 9 * NG32) * TERMS * 4 * COUT * 4;
    const __amdgpu_buffer_rsrc_t wrsrc = weight_rsrc(Ws, WS_FLOATS);
    const int w_lane = s3_w_lane<NW, COUT, MG, TN>(wave, lane);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto frag_addr = [&](int s) -> unsigned {
        if (C16) {
            const int ta = 2 * s, tb = (2 * s + 1 < 9) ? 2 * s + 1 : 8;
            const int off_a = LI::at(ta / 3, ta % 3), off_b = LI::at(tb / 3, tb % 3);
            return a_addr0 + ((kq >> 1) ? off_b : off_a);
        }
        const int tap = s / NG32, G = s - tap * NG32;
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
        return a_addr0 + 4 * G * LI::GS + LI::at(ky, kx);
    };
    bf16x8 a[TM][TERMS];
    S3W<TN> wb[2];
    wb[0] = w_first;
    {
        const unsigned ab = frag_addr(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int t = 0; t < TERMS; ++t) a[i][t] = __builtin_bit_cast(bf16x8, lds_read4(ab + a_imm(i) + t * LI::TSTEP));
    }
    auto pair_mfma = [&](const S3W<TN>& wc, int tw, int ta) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = split_mfma<TERMS>(wc.w[tw][j], a[i][ta], acc[i][j]);
                if constexpr ((PROBE >> 4) != 0) __builtin_amdgcn_sched_barrier(0);
                if constexpr ((PROBE >> 4) != 0) asm volatile("s_nop %0" ::"n"((PROBE >> 4) - 1));
                if constexpr ((PROBE >> 4) != 0) __builtin_amdgcn_sched_barrier(0);
            }
    };
    auto reload = [&](unsigned ab, int t) {
        if constexpr (PROBE & 2) return;
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i][t] = __builtin_bit_cast(bf16x8, lds_read4(ab + a_imm(i) + t * LI::TSTEP));
    };
    constexpr int NT_ = TM * TN;
    // the term pairs of one step in term-major order, every activation term reloaded (for the next step) right after its last use;
    // then the pinned interleaving: NLEAD weight loads, one after each of the first MFMAs, and the fragment reads where their registers die
    auto pairs = [&](const S3W<TN>& wc, unsigned ab, bool load_next) {
        if constexpr (TERMS == 3) {
            pair_mfma(wc, 0, 0); pair_mfma(wc, 1, 0); pair_mfma(wc, 2, 0);
            if (load_next) reload(ab, 0);
            pair_mfma(wc, 0, 1); pair_mfma(wc, 1, 1);
            if (load_next) reload(ab, 1);
            pair_mfma(wc, 0, 2);
            if (load_next) reload(ab, 2);
        } else {
            pair_mfma(wc, 0, 0); pair_mfma(wc, 1, 0);
            if (load_next) reload(ab, 0);
            pair_mfma(wc, 0, 1);
            if (load_next) reload(ab, 1);
        }
    };
    auto pin = [&](auto n_lead_c) {
        constexpr int n_lead = decltype(n_lead_c)::v;
#pragma unroll
        for (int l = 0; l < n_lead; ++l) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, TERMS * NT_ - n_lead, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
        if constexpr (TERMS == 3) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT_, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NT_, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
    };
    // one step: weights of step s_next requested first (spread between the first MFMAs), fragments of s_next as the terms retire
    const int wave_hi = __builtin_amdgcn_readfirstlane(wave >> 2);      // scalar: the priority switch below must not become a divergent branch
    // (Tried in round 4, profiles/r04_s3_s3_loop_probe_fair_priorities.txt: "fair" priorities - the two waves of a SIMD publish the step they are in
    // through LDS and the one that is AHEAD lowers its issue priority, so that the pair advances together instead of the older wave taking ~3/4 of the
    // pipe and the younger one finishing alone.  Slower on every shape (conv3 87.9 % of the pipe floor vs 91.5 %, conv5 77.8 vs 87.0): the arbiter's
    // oldest-first order wastes less than two waves that stall on the same things at the same time.  Removed.)
    auto step = [&](const S3W<TN>& wc, S3W<TN>& wn, int s_next, bool load_next) {
        if (alt_prio) {                       // the two waves of a SIMD (w, w + 4) take turns at the higher priority, one step each
            if ((s_next ^ wave_hi) & 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        unsigned ab = frag_addr(s_next);
        asm("" : "+v"(ab));
        if constexpr (PROBE & 1) { if (load_next) wn = wc; }
        else { if (load_next) s3_load_w<COUT, TN, TERMS>(wn, wrsrc, w_lane, s_next); }
        pairs(wc, ab, load_next);
        if (load_next && PROBE == 0) pin(IntC<TERMS * TN>{});
        __builtin_amdgcn_sched_barrier(0);
    };
    static_assert(TERMS * TN <= TERMS * TM * TN, "weight loads are spread over the first term pairs");
    auto finish = [&]() {
        if constexpr (TERMS == 2) {           // the two-term weights are packed times 2^e (fp16 range): undo it, exactly
            const float osc = Ws[WS_FLOATS];
            // An activation beyond fp16's range (|a| >= 65520) splits into (inf, -inf) and its products sum to NaN - which the ReLU of every epilogue
            // (v_max_f32 returns the non-NaN operand) would turn into a plausible 0.  NaN -> +inf here: +inf survives bias + ReLU, splits into (inf, NaN)
            // in the next layer and arrives at the outputs as inf / NaN (HardNet: NaN descriptors) instead of passing silently.
            if (variant & 2) {                                             // (A/B of this step's cost: tools/ab_nan_step.py)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] *= osc;
                return;
            }
            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) { acc[i][j] *= osc; sum += acc[i][j]; }
            const float s1 = (sum.x + sum.y) + (sum.z + sum.w);            // NaN iff a NaN (or +inf and -inf) is among the wave's sums: rare path below
            if (__builtin_amdgcn_ballot_w64(s1 != s1) != 0) {
                const float pinf = __builtin_inff();
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        f32x4 v = acc[i][j];
                        v.x = (v.x == v.x) ? v.x : pinf; v.y = (v.y == v.y) ? v.y : pinf; v.z = (v.z == v.z) ? v.z : pinf; v.w = (v.w == v.w) ? v.w : pinf;
                        acc[i][j] = v;
                    }
            }
        }
    };
    if constexpr (C16) {
        // 16 input channels (5 steps, fully unrolled; AffNet / OriNet conv1, conv2 at 128 VGPRs and four waves per SIMD): ONE weight set,
        // the next step's fragments are requested when the current step's last term pair has been issued - the other waves of the SIMD
        // cover the L2 round trip (a second set cost 8 - 10 spilled registers here)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const bool more = s + 1 < NS;
            unsigned ab = frag_addr(more ? s + 1 : s);
            asm("" : "+v"(ab));
            pairs(wb[0], ab, more);
            if (more) {
                pin(IntC<0>{});
                __builtin_amdgcn_sched_barrier(0);
                s3_load_w<COUT, TN, TERMS>(wb[0], wrsrc, w_lane, s + 1);
            }
        }
        finish();
        return;
    }
    // (Tried for the two-term loops, whose steps have half the matrix time to cover the L2 round trip of the weight fragments: weights requested TWO steps
    // ahead through three rotating sets.  profiles/r04_s4_s3_loop_probe_split2h_prefetch_depth.txt: -4 .. +2.5 % per layer shape, HardNet trunk 9.08 vs 9.13 ms
    // per 48000 patches at 255 instead of 190 VGPRs - the weights are not what these loops wait for.  Removed.)
#pragma unroll 1
    for (int s = 0; s + 2 < NS; s += 2) {
        step(wb[0], wb[1], s + 1, true);
        step(wb[1], wb[0], s + 2, true);
    }
    if (NS & 1) {                             // NS odd: the loop left the last step in wb[0]
        step(wb[0], wb[1], NS - 1, false);
    } else {
        step(wb[0], wb[1], NS - 1, true);
        step(wb[1], wb[0], NS - 1, false);
    }
    if (alt_prio) __builtin_amdgcn_s_setprio(0);
    finish();
}

// The same into a term-interleaved layout (LayQ).
template <int NW, typename LQH, int TN>
__device__ __forceinline__ void conv0_half_split_q(const float* patch, const float (&b)[3][TN], const f32x4 (&bv)[TN], float* act, int pass,
                                                   int wave, int lane) {
    static_assert(NW == 8 && LQH::H == 16 && LQH::W == 32 && LQH::C == 16 * TN, "half-patch conv0: 8 waves, all channel tiles in every wave");
    const int m = lane & 15, kq = lane >> 4;
    int toff[3];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        const int t = 4 * s3 + kq;
        toff[s3] = t < 9 ? (t / 3) * WP32 + (t % 3) : 0;
    }
    char* base = reinterpret_cast<char*>(act);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        int row_l, x0;
        if (k < 4) { const int T = 4 * wave + k; row_l = T >> 1; x0 = (T & 1) * 16; }
        else { row_l = pass ? -1 : 16; x0 = (wave & 1) * 16; }
        const int pb = (row_l + 16 * pass) * WP32 + x0 + m;
        float av[3];
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) av[s3] = patch[pb + toff[s3]];
        f32x4 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = bv[j];
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s3][j], av[s3], acc[j], 0, 0, 0);
        split_store_tile_q<LQH, TN, false>(base, LQH::at(row_l + 1, x0 + m + 1), 0, bv, acc, kq);
    }
}

// conv0 of a WHOLE 32 x 32 patch straight into a pre-split layout L (32 rows x 32 columns; two-term arithmetic: LayR holds it in 145 KB for 32 channels)
template <int NW, typename L, int TN>
__device__ __forceinline__ void conv0_whole_split_q(const float* patch, const float (&b)[3][TN], const f32x4 (&bv)[TN], float* act, int wave, int lane) {
    static_assert(NW == 8 && L::H == 32 && L::W == 32 && L::C == 16 * TN, "whole-patch conv0: 8 waves x 8 pixel tiles, all channel tiles in every wave");
    const int m = lane & 15, kq = lane >> 4;
    int toff[3];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        const int t = 4 * s3 + kq;
        toff[s3] = t < 9 ? (t / 3) * WP32 + (t % 3) : 0;
    }
    char* base = reinterpret_cast<char*>(act);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int T = 8 * wave + k, row = T >> 1, x0 = (T & 1) * 16;
        const int pb = row * WP32 + x0 + m;
        float av[3];
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) av[s3] = patch[pb + toff[s3]];
        f32x4 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = bv[j];
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s3][j], av[s3], acc[j], 0, 0, 0);
        split_store_tile_q<L, TN, false>(base, L::at(row + 1, x0 + m + 1), 0, bv, acc, kq);
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

// conv0 of HALF a 32 x 32 patch straight into a pre-split layout LBH (16 rows + halo rows, 32 columns): pass 0 = rows 0 .. 15 plus row
// 16 into the bottom halo row, pass 1 = rows 16 .. 31 plus row 15 into the top halo row.  34 tiles of 16 pixels over 8 waves: four each
// and one of the two extra-row tiles (computed by four waves each, identical values).
template <int NW, typename LBH, int TN>
__device__ __forceinline__ void conv0_half_split(const float* patch, const float (&b)[3][TN], const f32x4 (&bv)[TN], float* act, int pass,
                                                 int wave, int lane) {
    static_assert(NW == 8 && LBH::H == 16 && LBH::W == 32 && LBH::C == 16 * TN, "half-patch conv0: 8 waves, all channel tiles in every wave");
    const int m = lane & 15, kq = lane >> 4;
    int toff[3];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        const int t = 4 * s3 + kq;
        toff[s3] = t < 9 ? (t / 3) * WP32 + (t % 3) : 0;
    }
    char* base = reinterpret_cast<char*>(act);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        int row_l, x0;
        if (k < 4) { const int T = 4 * wave + k; row_l = T >> 1; x0 = (T & 1) * 16; }
        else { row_l = pass ? -1 : 16; x0 = (wave & 1) * 16; }
        const int pb = (row_l + 16 * pass) * WP32 + x0 + m;
        float av[3];
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) av[s3] = patch[pb + toff[s3]];
        f32x4 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = bv[j];
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s3][j], av[s3], acc[j], 0, 0, 0);
        split_store_tile<LBH, TN, false>(base, ((row_l + 1) * LBH::WP + x0 + m + 1) * 16, 0, bv, acc, kq);
    }
}

// fp32 epilogue of a half-patch layer (tiles cover 16 rows x 32 columns, all channel tiles in one wave): + bias, ReLU, rows row_off ..
template <typename LO, int TM, int TN>
__device__ __forceinline__ void store_half_lds(float* act, const f32x4 (&bias)[TN], const f32x4 (&acc)[TM][TN], int wave, int lane, int row_off) {
    const int n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = (wave * TM + i) * 16 + n;
        const int oy = (p >> 5) + row_off, ox = p & 31;
        const int pbase = ((oy + 1) * LO::WP + ox + 1) * 4;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x4 v = acc[i][j] + bias[j];
            v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
            *reinterpret_cast<f32x4*>(&act[(j * 4 + g) * LO::PSG + pbase]) = v;
        }
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

