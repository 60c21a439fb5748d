// 32x32-patch CNNs (AffNetFast / OriNetFast / HardNet) for gfx950 on the fp32 matrix cores.
//
// Replaces architectures.py:204-252 (AffNetFast), :33-82 (OriNetFast), HardNet.py:61-101
// (HardNet) incl. input_norm, eval-mode BatchNorm (folded into weights + bias at pack time),
// ReLU, the heads, rectifyAffineTransformationUpIsUp (LAF.py:285-291), get_rotation_matrix
// (LAF.py:276-283) and L2Norm (HardNet.py:12-19).
//
// Design (one workgroup = 8 wavefronts = one patch, whole trunk resident on the CU; details in DESIGN.md section 4):
//   * the patch is sampled from the pyramid (or loaded), standardised (mean / unbiased std + 1e-7, DPP wave reductions)
//     and stored as a zero-haloed 34 x 34 LDS tile;
//   * ALL six convolutions run on v_mfma_f32_16x16x4_f32 (exact fp32): conv0 with K = 9 taps padded to 12 and the
//     accumulators initialised with the bias, conv1..5 as implicit GEMMs (cnn_mfma.h: conv3x3_mfma) with the WEIGHTS as the
//     MFMA A operand and the ACTIVATIONS as the B operand, so a lane ends up with 4 consecutive channels of one pixel;
//   * activations live in ONE LDS buffer, channel-interleaved by 4 ((c/4)*PSG + pixel*4 + c%4): one ds_read_b128 per lane =
//     the activation operands of four k-steps, one ds_write_b128 per tile in the epilogue (bias + ReLU), written IN PLACE
//     over the layer's input after a barrier.  No HBM traffic between layers;
//   * packed weights [tap][cin/16][kq][cout][4] stream from L2 through a buffer descriptor (one buffer_load_dwordx4 per
//     lane = the weight operands of four k-steps), software-pipelined one chunk ahead, loads interleaved between the MFMAs;
//   * heads: HardNet stores its conv5 tile [pixel][channel] to HBM and an 8192 x 128 split-K MFMA GEMM over all patches
//     (hardnet_head_kernel + hardnet_finish_kernel: BN bias + L2 norm) follows; AffNet / OriNet reduce their heads' dot
//     products per wave straight from the conv5 accumulators (head_partials) and cnn16_finish_kernel combines the eight
//     partials per patch in fixed order (tanh, rectification / atan2).
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

#include "cnn_mfma.h"
#include "shape_filter.h"

extern "C" size_t affnet_cnn32_packed_floats(int net_kind) {
    if (net_kind < 0 || net_kind > AFFNET_NET_AFFNET_FULLCONV) return 0;
    return net_layout(net_kind).total;
}

extern "C" int affnet_cnn32_pack_weights(int kind, const float* const* conv_w, const float* const* bn_mean, const float* const* bn_var,
                                         const float* head_w, const float* head_b, const float* head_bn_mean, const float* head_bn_var,
                                         float* out) {
    if (kind < 0 || kind > AFFNET_NET_AFFNET_FULLCONV || !conv_w || !bn_mean || !bn_var || !head_w || !out) return AFFNET_ERR_INVALID;
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
    // split copies for AFFNET_ARITH_FP32_SPLIT3 (conv1 .. conv5 = S3_LAYER_MASK, all three nets): the same BN-folded fp32 weight as three bf16 terms (nearest even,
    // exact remainders), [tap][cin / 32][term][kq][cout][8]: lane (cout, kq) of the bf16 MFMA's A operand = 8 consecutive input channels
    for (int i = 1; i < 6; ++i) {
        if (!L.w_s3[i]) continue;
        const int ci = L.cin[i], co = L.cout[i];
        uint16_t* dst = reinterpret_cast<uint16_t*>(out + L.w_s3[i]);
        auto bf16_rne = [](float x) -> uint32_t { uint32_t u; memcpy(&u, &x, 4); return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u; };
        for (int n = 0; n < co; ++n) {
            const float sc = 1.0f / sqrtf(bn_var[i][n] + 1e-5f);
            for (int c = 0; c < ci; ++c)
                for (int t9 = 0; t9 < 9; ++t9) {
                    float r = conv_w[i][((size_t)n * ci + c) * 9 + t9] * sc;          // the value the fp32 path uses
                    for (int term = 0; term < 3; ++term) {
                        const uint32_t hb = bf16_rne(r);
                        float hf; memcpy(&hf, &hb, 4);
                        r -= hf;                                                       // exact
                        if (ci == 16) {                                                // two taps per k = 32 step: [step][term][kq][cout][8]
                            const int st = t9 / 2, kq = (t9 % 2) * 2 + c / 8, j = c % 8;
                            dst[((((size_t)st * 3 + term) * 4 + kq) * co + n) * 8 + j] = (uint16_t)(hb >> 16);
                        } else {
                            const int G = c / 32, kq = (c % 32) / 8, j = c % 8;
                            dst[(((((size_t)t9 * (ci / 32) + G) * 3 + term) * 4 + kq) * co + n) * 8 + j] = (uint16_t)(hb >> 16);
                        }
                    }
                }
        }
    }
    // two-term copies for AFFNET_ARITH_FP32_SPLIT2H: 2^e * w as hi = fp16(2^e w), lo = fp16(2^e w - hi) (nearest even), e per layer such that the
    // largest |w| of the layer lands in [2^13, 2^14) - both terms of every weight down to 2^-15 of the largest then sit in fp16's normal range.
    // Same fragment order with two terms; 2^-e (what the loop multiplies its sums with) follows the copy.
    auto f16_bits = [](float x) -> uint16_t { const _Float16 h = (_Float16)x; uint16_t u; memcpy(&u, &h, 2); return u; };
    auto f16_val = [](uint16_t u) -> float { _Float16 h; memcpy(&h, &u, 2); return (float)h; };
    for (int i = 1; i < 6; ++i) {
        if (!L.w_h2[i]) continue;
        const int ci = L.cin[i], co = L.cout[i];
        uint16_t* dst = reinterpret_cast<uint16_t*>(out + L.w_h2[i]);
        float wmax = 0.0f;
        for (int n = 0; n < co; ++n) {
            const float sc = 1.0f / sqrtf(bn_var[i][n] + 1e-5f);
            for (int k = 0; k < ci * 9; ++k) wmax = fmaxf(wmax, fabsf(conv_w[i][(size_t)n * ci * 9 + k] * sc));
        }
        const int e = (wmax > 0.0f && std::isfinite(wmax)) ? std::min(100, std::max(-100, 13 - ilogbf(wmax))) : 0;      // (clamped: 2^-e must stay a normal fp32)
        out[L.w_h2[i] + s3_floats(ci, co, 2)] = ldexpf(1.0f, -e);
        for (int n = 0; n < co; ++n) {
            const float sc = 1.0f / sqrtf(bn_var[i][n] + 1e-5f);
            for (int c = 0; c < ci; ++c)
                for (int t9 = 0; t9 < 9; ++t9) {
                    const float w = ldexpf(conv_w[i][((size_t)n * ci + c) * 9 + t9] * sc, e);      // the value the fp32 path uses, times 2^e (exact)
                    const uint16_t hb = f16_bits(w), lb = f16_bits(w - f16_val(hb));
                    for (int term = 0; term < 2; ++term) {
                        const uint16_t v = term ? lb : hb;
                        if (ci == 16) {
                            const int st = t9 / 2, kq = (t9 % 2) * 2 + c / 8, j = c % 8;
                            dst[((((size_t)st * 2 + term) * 4 + kq) * co + n) * 8 + j] = v;
                        } else {
                            const int G = c / 32, kq = (c % 32) / 8, j = c % 8;
                            dst[(((((size_t)t9 * (ci / 32) + G) * 2 + term) * 4 + kq) * co + n) * 8 + j] = v;
                        }
                    }
                }
        }
    }
    if (kind == AFFNET_NET_HARDNET) {
        if (!head_bn_mean || !head_bn_var) return AFFNET_ERR_INVALID;
        {
            uint16_t* hh2 = reinterpret_cast<uint16_t*>(out + L.head_h2);
            float wmax = 0.0f;
            for (int n = 0; n < 128; ++n) {
                const float sc = 1.0f / sqrtf(head_bn_var[n] + 1e-5f);
                for (int k = 0; k < HEAD_K; ++k) wmax = fmaxf(wmax, fabsf(head_w[(size_t)n * HEAD_K + k] * sc));
            }
            const int e = (wmax > 0.0f && std::isfinite(wmax)) ? std::min(100, std::max(-100, 13 - ilogbf(wmax))) : 0;      // (clamped: 2^-e must stay a normal fp32)
            out[L.head_h2 + (size_t)HEAD_K * 128] = ldexpf(1.0f, -e);
            for (int n = 0; n < 128; ++n) {
                const float sc = 1.0f / sqrtf(head_bn_var[n] + 1e-5f);
                for (int c = 0; c < 128; ++c)
                    for (int pp = 0; pp < 64; ++pp) {
                        const size_t k = (size_t)pp * 128 + c;
                        const float w = ldexpf(head_w[(size_t)n * HEAD_K + c * 64 + pp] * sc, e);
                        const uint16_t hb = f16_bits(w), lb = f16_bits(w - f16_val(hb));
                        hh2[(((((k >> 5) * 2 + 0) * 4 + ((k & 31) >> 3)) * 128 + n) << 3) + (k & 7)] = hb;
                        hh2[(((((k >> 5) * 2 + 1) * 4 + ((k & 31) >> 3)) * 128 + n) << 3) + (k & 7)] = lb;
                    }
            }
        }
        uint16_t* hs3 = reinterpret_cast<uint16_t*>(out + L.head_s3);
        auto bf16_rne_h = [](float x) -> uint32_t { uint32_t u; memcpy(&u, &x, 4); return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u; };
        for (int n = 0; n < 128; ++n) {
            const float s = 1.0f / sqrtf(head_bn_var[n] + 1e-5f);
            out[L.head_b + n] = -head_bn_mean[n] * s;
            // split copy of the same BN-folded weights (AFFNET_ARITH_FP32_SPLIT3): three bf16 terms, exact remainders, in the B-fragment order
            // of hardnet_head_s3_kernel: [k / 32][term][kq = (k % 32) / 8][n][k % 8]
            for (int c = 0; c < 128; ++c)
                for (int pp = 0; pp < 64; ++pp) {
                    const size_t k = (size_t)pp * 128 + c;
                    float r = head_w[(size_t)n * HEAD_K + c * 64 + pp] * s;
                    for (int term = 0; term < 3; ++term) {
                        const uint32_t hb = bf16_rne_h(r);
                        float hf; memcpy(&hf, &hb, 4);
                        r -= hf;
                        hs3[(((((k >> 5) * 3 + term) * 4 + ((k & 31) >> 3)) * 128 + n) << 3) + (k & 7)] = (uint16_t)(hb >> 16);
                    }
                }
            // K order of the head GEMM = the trunk kernel's output order k = pixel * 128 + channel; stored interleaved by 4 like
            // the conv weights, [k/16][(k/4)%4][n][k%4], so that one 16-byte load per lane is the B fragment of 4 MFMA k-steps
            for (int c = 0; c < 128; ++c)
                for (int pp = 0; pp < 64; ++pp) {
                    const size_t k = (size_t)pp * 128 + c;
                    out[L.head_w + (((k >> 4) * 4 + ((k >> 2) & 3)) * 128 + n) * 4 + (k & 3)] = head_w[(size_t)n * HEAD_K + c * 64 + pp] * s;
                }
        }
    } else if (kind == AFFNET_NET_AFFNET_FULLCONV) {
        if (!head_b) return AFFNET_ERR_INVALID;
        // dense 8 x 8 head (architectures.py:652) as the A operand of fullconv_head_kernel's GEMM: rows n = o * 8 + kx (24 of 32
        // used, the rest stay zero), K = (ky, c): [ky][c / 16][(c / 4) % 4][n][c % 4]
        for (int o = 0; o < 3; ++o)
            for (int c = 0; c < 64; ++c)
                for (int ky = 0; ky < 8; ++ky)
                    for (int kx = 0; kx < 8; ++kx)
                        out[L.head_w + ((((size_t)ky * 4 + c / 16) * 4 + (c / 4) % 4) * 32 + o * 8 + kx) * 4 + c % 4] =
                            head_w[((size_t)o * 64 + c) * 64 + ky * 8 + kx];
        memcpy(out + L.head_b, head_b, 3 * sizeof(float));
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

// OriNet head through LDS (round 4).  head_partials<ORINET> makes every lane fetch the weight vector of each of the 2 x 9 (output, tap)
// pairs for its own pixel: 36 buffer_load_dwordx4 per wave, 295 KB of L2 -> L1 traffic per patch for 32 KB of distinct weights - the head
// took 14.4 k cycles per workgroup against AffNet's 3.3 k (tools/s3_phase_timing.py), L1-bound.  Here the roles are swapped: a lane owns
// the WEIGHT position (ky, kx) = its tile pixel and 4 channels, loads those weights once per output (4 loads) and reads the nine shifted
// ACTIVATIONS from a zero-haloed 10 x 10 copy of the conv5 output in LDS (the activation buffer is dead after the conv5 loop):
//   out[o][qy][qx] = sum over (ky, kx, c) of  W[o][ky][kx][c] * A[qy + ky - 1][qx + kx - 1][c]      (A = 0 outside the 8 x 8 map)
// Same products as before, grouped by weight position instead of activation position; same [wave][o * 9 + q] partial layout.
#define ORI_HP 68        // floats per pixel of the LDS copy (64 channels + 4: consecutive pixels 4 banks apart)
template <int TM, int NTHR>
__device__ __forceinline__ void head_partials_ori_lds(const float* __restrict__ hw, const f32x4 (&bias)[1], const f32x4 (&acc)[TM][1],
                                                      float* __restrict__ part, float* act, int wave, int lane, int tid) {
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
    const __amdgpu_buffer_rsrc_t r = weight_rsrc(hw, 2 * 4096);
    f32x4 w[2][TM];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int i = 0; i < TM; ++i) w[o][i] = buf_read4(r, (n * 64 + c4) * 4, (o * 4096 + (mg * TM + i) * 16 * 64) * 4);
    __syncthreads();                                                     // every wave has finished reading the conv5 input from `act`
    for (int e = tid; e < 36 * 16; e += NTHR) {                          // zero halo of the 10 x 10 grid: 36 pixels x 16 float4
        const int hp = e >> 4, q4 = e & 15;
        const int y = hp < 10 ? 0 : (hp < 20 ? 9 : 1 + ((hp - 20) >> 1)), x = hp < 10 ? hp : (hp < 20 ? hp - 10 : ((hp - 20) & 1) * 9);
        *reinterpret_cast<f32x4*>(&act[(y * 10 + x) * ORI_HP + 4 * q4]) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    int pbase[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = (mg * TM + i) * 16 + n;
        pbase[i] = ((p >> 3) * 10 + (p & 7)) * ORI_HP + c4;             // (ky, kx) in padded coordinates of tap q = (0, 0)
        *reinterpret_cast<f32x4*>(&act[pbase[i] + 11 * ORI_HP]) = v[i];  // interior pixel (py + 1, px + 1)
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(&act[pbase[i] + ((q / 3) * 10 + q % 3) * ORI_HP]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { s0 = fmaf(av[j], w[0][i][j], s0); s1 = fmaf(av[j], w[1][i][j], s1); }
        }
        s0 = wave_sum(s0);
        s1 = wave_sum(s1);
        if (lane == 0) { part[wave * 18 + q] = s0; part[wave * 18 + 9 + q] = s1; }
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
    int s3_alt;            // tuning variant bits of the split-operand trunks (affnet_debug_split3_variant): bit 0 = the two waves of a SIMD alternate at the higher priority inside the HardNet loops
    float* dbg_out;
    unsigned long long* dbg_time;   // != NULL: s_memtime stamps [patch][wave][32] at the phase boundaries (tuning aid)
    // Row window [row_begin, row_begin + gridDim.x) of every image, and the lazy-evaluation predicate of the fused pipeline: when
    // skip_cnt != NULL the launch does nothing for an image whose detections are response-sorted (CNT_SEL_MODE == 1) and whose first
    // pass already produced skip_n survivors of the shape filter (CNT_SURVIVED1) - pipeline.hip, affnet_describe_detected.
    int row_begin;
    const int32_t* skip_cnt;
    int skip_n;
    // Shape-stage bookkeeping done by thread 0 of workgroup (0, image) of the AffNet trunk launches (each was a 5 us launch of its
    // own): shape_op 1 = first pass: survivor / evaluation counters = 0; 2 = second (lazy) pass: freeze the first pass's survivor
    // count (CNT_SURVIVED1) for the predicate of the finish + filter kernel that follows this launch.  The trunk workgroups
    // themselves test CNT_SURVIVED, which nothing changes while a trunk launch runs.
    int32_t* shape_cnt;
    int shape_op;
};
// (the split-operand variants are template instantiations: cnn32_trunk_kernel<KIND, NW, STAMPS, S3>, S3 = 3 bf16 terms or 2 fp16 terms; 0 = exact)

__device__ __forceinline__ bool lazy_skip(const int32_t* skip_cnt, int skip_n, int image, int which = CNT_SURVIVED1) {
    if (!skip_cnt) return false;
    const int32_t* c = skip_cnt + (size_t)image * CNT_TOTAL;
    return c[CNT_SEL_MODE] == 1 && c[which] >= skip_n;
}

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
template <int KIND, int NW, bool STAMPS, int S3 = 0>
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
    const int prow = blockIdx.x + a.row_begin;
    if (KIND == AFFNET_NET_AFFNET && a.shape_cnt && blockIdx.x == 0 && threadIdx.x == 0) {
        int32_t* c = a.shape_cnt + (size_t)blockIdx.y * CNT_TOTAL;
        if (a.shape_op == 1) { c[CNT_SURVIVED] = 0; c[CNT_SURVIVED1] = 0; c[CNT_AFF_EVAL] = 0; }
        else if (a.shape_op == 2) c[CNT_SURVIVED1] = c[CNT_SURVIVED];
    }
    if (prow >= n || lazy_skip(a.skip_cnt, a.skip_n, blockIdx.y, CNT_SURVIVED)) return;
    const size_t pidx = (size_t)blockIdx.y * a.n_max + prow;
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
    constexpr bool HALF = S3 != 0;                                // split-operand arithmetic: conv0 .. conv2 in two half-patch passes
    constexpr int TERMS = S3 ? S3 : 3;                            // terms per operand of the split arithmetic (3 bf16 / 2 fp16)
    static_assert(S3 == 0 || S3 == 2 || S3 == 3, "S3 = number of terms of the split arithmetic");
    // three bf16 terms: term-interleaved 48-byte cells (LayQ), conv0 .. conv2 in two half-patch passes; two fp16 terms: 16-byte pixels, the two terms of a row side by
    // side (LayR; per-reader row pitch / group stride), conv0 once for the whole patch
    typedef LayQ<16, 32, 34, CB, 0, 3> LQH;                      // three terms: conv0 output of half a patch, pre-split; read by conv1 (stride 1)
    typedef LayQ<16, 32, 34, CB, 16, 3> LQH2;                    // three terms: conv1 output of half a patch; read by conv2 at stride 2
    // two-term arithmetic: LayR's 16-byte pixels hold conv0's / conv1's output of the WHOLE patch (145 KB for 32 channels, 72.5 KB for 16), so conv0 runs once; conv1 / conv2
    // keep their two half-patch LOOPS (same register blockings) on 16-row views of the whole layouts - no second conv0 pass, no halo-row fix-ups between the halves
    using LR0 = LayR<32, 32, 34, CB, 0>;                          // conv0 output, read by conv1 (stride 1)
    using LR0H = LayR<16, 32, 34, CB, 0, 34>;                     // its 16-row view
    using LR1 = LayR<32, 32, 34, CB, 16>;                         // conv1 output, read by conv2 at stride 2
    using LR1H = LayR<16, 32, 34, CB, 16, 34>;
    constexpr bool WHOLE = (TERMS == 2) && HALF;
    static_assert(!WHOLE || (LR0::BYTES <= TrunkLds<CB>::ACT * 4 && LR1::BYTES <= TrunkLds<CB>::ACT * 4 && LR0H::GS == LR0::GS && LR1H::GS == LR1::GS), "whole-patch split layouts");
    if constexpr (WHOLE) zero_halo_q<LR0, NTHR>(act);
    else if constexpr (HALF) zero_halo_q<LQH, NTHR>(act);
    else zero_halo<LayC0, NTHR>(act, CB);
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
    if constexpr (!HALF) {
        f32x4 acc[T1M][T1N];
        conv0_mfma<NW, CB, T1M, T1N>(patch, w0, bias0, acc, wave, lane);
        CNN_STAMP(19);
        prefetch_bias<NW, 32, T1M, T1N>(a.packed + a.off.b[1], bias1, wave, lane);
        store_tiles_lds<CB, LayC0, T1M, T1N, false>(act, bias0, acc, wave, lane);
        CNN_STAMP(20);
        __syncthreads();
    }
    if (STAMPS && a.dbg_layer == 0) { dump_planes<CB, LayC0, NTHR>(act, a.dbg_out); return; }
    if (!HALF) CNN_STAMP(2);

    // AFFNET_ARITH_FP32_SPLIT3 (affnet_set_arith): conv1 .. conv5 on split operands - every fp32 operand as three bf16 terms, six
    // v_mfma_f32_16x16x32_bf16 per product, fp32 accumulate.  conv0 .. conv4 write their outputs PRE-SPLIT into term-interleaved cells (LayQ),
    // conv1 .. conv5 read ready fragments (conv3x3_mfma_s3q): no VALU work inside the MFMA loops ("Split-operand trunks").
    // The first weight fragments of a loop are requested before the barriers / epilogue in front of it.
    // HardNet (one workgroup per CU): optionally (affnet_debug_split3_variant bit 0) the two waves of a SIMD take turns at the higher priority
    // inside the loops.  Round 3's tile-major loops gained 2.5 % from it; with the term-major loops it costs 1 % (default off).
    const int s3_alt = KIND == AFFNET_NET_HARDNET ? a.s3_alt : 0;       // variant bits for the loops (conv3x3_mfma_s3q)
    if constexpr (S3 != 0 && KIND == AFFNET_NET_HARDNET) {
        // conv2 / conv3 outputs, 64 channels @16x16 (122 KB / 90 KB): read at stride 1 (conv3) and, conv3's output written in place, at stride 2 (conv4).  No group
        // stride serves both readers (tools/lds_bank_model.py): GS = 0 (mod 256) leaves conv4's two-row reader with 2-way conflicts, GS = 16 conv3's one-row reader.
        // LayR takes conv4's here: its 4 x 1 tiles are the more LDS-bound (probe: -1350 cycles for conv4, +300 for conv3's 4 x 2); a second layout for conv3's
        // output with the other stride cost 0.8 k cycles per patch for zeroing its halo again (measured)
        using LQ2 = std::conditional_t<TERMS == 2, LayR<16, 16, 20, 2 * CB, 16>, LayQ<16, 16, 18, 2 * CB, 0, 3>>;
        using LQ4 = std::conditional_t<TERMS == 2, LayR<8, 8, 12, 4 * CB, 0>, LayQ<8, 8, 16, 4 * CB, 128, 3>>;        // conv4 output: 128 channels @8x8 (122 KB / 60 KB)
        static_assert(LQ2::BYTES <= TrunkLds<CB>::ACT * 4 && LQ4::BYTES <= TrunkLds<CB>::ACT * 4 && LQH::BYTES <= TrunkLds<CB>::ACT * 4 &&
                      LQH2::BYTES <= TrunkLds<CB>::ACT * 4, "pre-split layouts must fit the activation buffer");
        char* base = reinterpret_cast<char*>(act);
        // three terms: conv0 + conv1 in two half-patch passes - the pre-split conv0 output of 32 channels @32x32 would be 222 KB, half of it (16 rows +
        // a halo row either side) is 115 KB; two terms (WHOLE): 145 KB, conv0 runs once and the two half loops of conv1 / conv2 read 16-row views
        f32x4 acc_a[4][2], acc_b[4][2];
        // register blockings per layer from tools/probes/s3_loop_probe (profiles/r04_s3_s3_loop_probe_tilings.txt): conv1 / conv3 4 pixel tiles x 2 channel
        // tiles per wave; conv2 / conv4 / conv5 4 x 1 (one weight fragment feeds four pixel tiles: 85.0 / 86.0 / 89.0 % of the pipe floor vs 78.5 / 83.8 /
        // 87.0 % for 2 x 2)
        S3W<2> wf1;
        S3W<1> wf2;
        f32x4 acc2_a[4][1], acc2_b[4][1], bias2[1];
        f32x4 acc2w[4][2], bias2w[2];                                    // (two-term flow: conv2 as one 4 x 2 loop)
        if constexpr (WHOLE) {
            // two terms: conv0 once, then conv1 as ONE loop over the whole patch (8 pixel tiles x 2 channel tiles per wave: the weights stream once, not once per half)
            // and conv2 as one 4 x 2 loop
            f32x4 acc1[8][2];
            S3W<2> wf2w;
            s3_prefetch_w0<NW, CB, CB, 64, 8, 2, TERMS>(a.packed + a.off.w_s3[1], wf1, wave, lane);
            prefetch_bias<NW, 32, 8, 2>(a.packed + a.off.b[1], bias1, wave, lane);
            conv0_whole_split_q<NW, LR0, 2>(patch, w0, bias0, act, wave, lane);
            __syncthreads();
            CNN_STAMP(2);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            conv3x3_mfma_s3q<NW, CB, CB, LR0, 1, 8, 2>(act, a.packed + a.off.w_s3[1], wf1, acc1, wave, lane, s3_alt);
            if (PRIO) __builtin_amdgcn_s_setprio(3);
            CNN_STAMP(3);
            s3_prefetch_w0<NW, CB, 2 * CB, 16, 4, 2, TERMS>(a.packed + a.off.w_s3[2], wf2w, wave, lane);
            prefetch_bias<NW, 16, 4, 2>(a.packed + a.off.b[2], bias2w, wave, lane);
            __syncthreads();
            zero_halo_q<LR1, NTHR>(act);                                 // another group stride than LR0 (the stride-2 reader's): the halo cells move
            store_tiles_split_q<CB, LR1, 8, 2>(act, bias1, acc1, wave, lane);
            __syncthreads();
            CNN_STAMP(4);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            conv3x3_mfma_s3q<NW, CB, 2 * CB, LR1, 2, 4, 2>(act, a.packed + a.off.w_s3[2], wf2w, acc2w, wave, lane, s3_alt);
            if (PRIO) __builtin_amdgcn_s_setprio(3);
        } else {
            s3_prefetch_w0<NW, CB, CB, 32, 4, 2, TERMS>(a.packed + a.off.w_s3[1], wf1, wave, lane);
            prefetch_bias<NW, 32, 8, 2>(a.packed + a.off.b[1], bias1, wave, lane);
            conv0_half_split_q<NW, LQH, 2>(patch, w0, bias0, act, 0, wave, lane);
            __syncthreads();
            CNN_STAMP(2);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            conv3x3_mfma_s3q<NW, CB, CB, LQH, 1, 4, 2>(act, a.packed + a.off.w_s3[1], wf1, acc_a, wave, lane, s3_alt);
            if (PRIO) __builtin_amdgcn_s_setprio(3);
            __syncthreads();
            if (tid < LQH::SLOTS * 4 * 32) {                             // pass 0 left conv0 row 16 in the bottom halo row: zero again (slots x 4 groups x 32 cells)
                const int t = tid / 128, g = (tid >> 5) & 3, x = tid & 31;
                *reinterpret_cast<f32x4*>(base + g * LQH::GS + LQH::at(17, x + 1) + t * LQH::TSTEP) = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            conv0_half_split_q<NW, LQH, 2>(patch, w0, bias0, act, 1, wave, lane);
            __syncthreads();
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            conv3x3_mfma_s3q<NW, CB, CB, LQH, 1, 4, 2>(act, a.packed + a.off.w_s3[1], wf1, acc_b, wave, lane, s3_alt);
            if (PRIO) __builtin_amdgcn_s_setprio(3);
            CNN_STAMP(3);
            s3_prefetch_w0<NW, CB, 2 * CB, 8, 4, 1, TERMS>(a.packed + a.off.w_s3[2], wf2, wave, lane);
            bias2[0] = *reinterpret_cast<const f32x4*>(&a.packed[a.off.b[2] + (wave >> 1) * 16 + 4 * (lane >> 4)]);      // MG = 8 tiles / 4 = 2: channel tile = wave / 2
            __syncthreads();
            // conv1's output goes back into the same half layout, pre-split, and conv2 (stride 2: output rows 0 .. 7 read input rows
            // -1 .. 15, rows 8 .. 15 read 15 .. 31) runs in two passes as well
            zero_halo_q<LQH2, NTHR>(act);                                // another group stride than LQH (bank conflicts of the stride-2 reader)
            store_tiles_split_q<CB, LQH2, 4, 2>(act, bias1, acc_a, wave, lane);
            __syncthreads();
            CNN_STAMP(4);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            conv3x3_mfma_s3q<NW, CB, 2 * CB, LQH2, 2, 4, 1>(act, a.packed + a.off.w_s3[2], wf2, acc2_a, wave, lane, s3_alt);
            if (PRIO) __builtin_amdgcn_s_setprio(3);
            __syncthreads();
            store_tiles_split_q<CB, LQH2, 4, 2>(act, bias1, acc_b, wave, lane);
            if (wave == 7) {                                             // conv1 row 15 (tiles 2, 3 of wave 7 in pass 0) = the top halo row of pass 1
                const int n = lane & 15;
#pragma unroll
                for (int i = 2; i < 4; ++i) split_store_tile_q<LQH2, 2>(base, LQH2::at(0, (i - 2) * 16 + n + 1), 0, bias1, acc_a[i], lane >> 4);
            }
            __syncthreads();
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            conv3x3_mfma_s3q<NW, CB, 2 * CB, LQH2, 2, 4, 1>(act, a.packed + a.off.w_s3[2], wf2, acc2_b, wave, lane, s3_alt);
            if (PRIO) __builtin_amdgcn_s_setprio(3);
        }
        CNN_STAMP(5);
        S3W<2> wf3;
        S3W<1> wf4, wf5;
        f32x4 bias3[2], bias4[1], bias5s[1];
        s3_prefetch_w0<NW, 2 * CB, 2 * CB, 16, 4, 2, TERMS>(a.packed + a.off.w_s3[3], wf3, wave, lane);
        prefetch_bias<NW, 16, 4, 2>(a.packed + a.off.b[3], bias3, wave, lane);
        __syncthreads();
        zero_halo_q<LQ2, NTHR>(act);
        if constexpr (WHOLE) store_tiles_split_q<2 * CB, LQ2, 4, 2>(act, bias2w, acc2w, wave, lane);
        else {
            store_tiles_split_q<2 * CB, LQ2, 4, 1, 8>(act, bias2, acc2_a, wave, lane, 0);
            store_tiles_split_q<2 * CB, LQ2, 4, 1, 8>(act, bias2, acc2_b, wave, lane, 8);
        }
        __syncthreads();
        CNN_STAMP(6);
        {
            f32x4 acc_[4][2];                                            // conv3: 64 -> 64 @16x16
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            conv3x3_mfma_s3q<NW, 2 * CB, 2 * CB, LQ2, 1, 4, 2>(act, a.packed + a.off.w_s3[3], wf3, acc_, wave, lane, s3_alt);
            if (PRIO) __builtin_amdgcn_s_setprio(3);
            CNN_STAMP(7);
            s3_prefetch_w0<NW, 2 * CB, 4 * CB, 4, 4, 1, TERMS>(a.packed + a.off.w_s3[4], wf4, wave, lane);
            prefetch_bias<NW, 8, 4, 1>(a.packed + a.off.b[4], bias4, wave, lane);
            __syncthreads();
            store_tiles_split_q<2 * CB, LQ2, 4, 2>(act, bias3, acc_, wave, lane);      // same layout in place: the halo is still zero
            __syncthreads();
            CNN_STAMP(8);
        }
        {
            f32x4 acc_[4][1];                                            // conv4: 64 -> 128, stride 2 -> 8x8
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            conv3x3_mfma_s3q<NW, 2 * CB, 4 * CB, LQ2, 2, 4, 1>(act, a.packed + a.off.w_s3[4], wf4, acc_, wave, lane, s3_alt);
            if (PRIO) __builtin_amdgcn_s_setprio(3);
            CNN_STAMP(9);
            s3_prefetch_w0<NW, 4 * CB, 4 * CB, 4, 4, 1, TERMS>(a.packed + a.off.w_s3[5], wf5, wave, lane);
            prefetch_bias<NW, 8, 4, 1>(a.packed + a.off.b[5], bias5s, wave, lane);
            __syncthreads();
            zero_halo_q<LQ4, NTHR>(act);
            store_tiles_split_q<4 * CB, LQ4, 4, 1>(act, bias4, acc_, wave, lane);
            __syncthreads();
            CNN_STAMP(10);
        }
        {
            f32x4 acc5[4][1];                                            // conv5: 128 -> 128 @8x8, conv5 tensor -> HBM for the head GEMM
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            conv3x3_mfma_s3q<NW, 4 * CB, 4 * CB, LQ4, 1, 4, 1>(act, a.packed + a.off.w_s3[5], wf5, acc5, wave, lane, s3_alt);
            if (PRIO) __builtin_amdgcn_s_setprio(3);
            CNN_STAMP(11);
            store_tiles_global<4 * CB, 4, 1>(a.out + pidx * (64 * 4 * CB), bias5s, acc5, wave, lane);
        }
        return;
    }

    if constexpr (S3 != 0 && CB == 16) {
        // AffNet / OriNet on split operands, same structure as the HardNet branch: conv0 .. conv2 in two half-patch passes on pre-split
        // layouts (conv1 / conv2 have 16 input channels: two taps per k = 32 step), conv3 .. conv5 whole.
        // conv2 / conv3 outputs: 32 channels @16x16 (61 KB / 45 KB).  LayR: GS = 0 (mod 256) here - with two workgroups per CU conv3's one-row reader is LDS-bound and
        // the 2-way conflicts of the HardNet branch's choice cost it 10 % (probe: 7.1 k vs 6.5 k cycles), while conv4 (2 x 1 tiles) is the same with or without its own
        using LQ2 = std::conditional_t<TERMS == 2, LayR<16, 16, 20, 2 * CB, 0>, LayQ<16, 16, 18, 2 * CB, 0, 3>>;
        using LQ4 = std::conditional_t<TERMS == 2, LayR<8, 8, 12, 4 * CB, 0>, LayQ<8, 8, 16, 4 * CB, 128, 3>>;        // conv4 output: 64 channels @8x8 (61 KB / 30 KB)
        static_assert(LQH::BYTES <= TrunkLds<CB>::ACT * 4 && LQH2::BYTES <= TrunkLds<CB>::ACT * 4 && LQ2::BYTES <= TrunkLds<CB>::ACT * 4 && LQ4::BYTES <= TrunkLds<CB>::ACT * 4,
                      "pre-split layouts must fit the activation buffer");
        char* base = reinterpret_cast<char*>(act);
        // (tried in round 4: the phases outside the MFMA loops at a higher issue priority than the loops - with two workgroups per CU a young
        // workgroup crawls through input / conv0 / epilogues next to an older one in its loops, conv0 of half a patch takes 9 - 10 k cycles for
        // ~150 instructions per wave.  Zero-sum as in the exact path: 4.12 vs 4.10 - 4.13 ms per 48000 patches.  Removed.)
        f32x4 acc_a[4][1], acc_b[4][1];
        // (128 VGPRs at two workgroups per CU: a loop's first weight fragments are requested right in front of it here - held across the
        // previous epilogue like in the HardNet branch they cost 17 / 23 spilled registers)
        S3W<1> wf1, wf2;
        f32x4 acc2_a[2][1], acc2_b[2][1], bias2[1];
        if constexpr (WHOLE) {
            prefetch_bias_fresh<NW, 32, 8, 1>(a.packed + a.off.b[1], bias1, wave, lane);
            conv0_whole_split_q<NW, LR0, 1>(patch, w0, bias0, act, wave, lane);
            s3_prefetch_w0<NW, CB, CB, 32, 4, 1, TERMS>(a.packed + a.off.w_s3[1], wf1, wave, lane);
            __syncthreads();
            CNN_STAMP(2);
            conv3x3_mfma_s3q<NW, CB, CB, LR0H, 1, 4, 1>(act, a.packed + a.off.w_s3[1], wf1, acc_a, wave, lane, false);
            conv3x3_mfma_s3q<NW, CB, CB, LR0H, 1, 4, 1>(act + LR0::at(16, 0) / 4, a.packed + a.off.w_s3[1], wf1, acc_b, wave, lane, false);
            CNN_STAMP(3);
            {
                int l2 = lane;
                asm volatile("" : "+v"(l2));
                bias2[0] = *reinterpret_cast<const f32x4*>(&a.packed[a.off.b[2] + (wave >> 2) * 16 + 4 * (l2 >> 4)]);
            }
            __syncthreads();
            zero_halo_q<LR1, NTHR>(act);                                     // another group stride than LR0 (the stride-2 reader's): the halo cells move
            store_tiles_split_q<CB, LR1, 4, 1, 16>(act, bias1, acc_a, wave, lane, 0);
            store_tiles_split_q<CB, LR1, 4, 1, 16>(act, bias1, acc_b, wave, lane, 16);
            s3_prefetch_w0<NW, CB, 2 * CB, 8, 2, 1, TERMS>(a.packed + a.off.w_s3[2], wf2, wave, lane);
            __syncthreads();
            CNN_STAMP(4);
            conv3x3_mfma_s3q<NW, CB, 2 * CB, LR1H, 2, 2, 1>(act, a.packed + a.off.w_s3[2], wf2, acc2_a, wave, lane, false);
            conv3x3_mfma_s3q<NW, CB, 2 * CB, LR1H, 2, 2, 1>(act + LR1::at(16, 0) / 4, a.packed + a.off.w_s3[2], wf2, acc2_b, wave, lane, false);
        } else {
            prefetch_bias_fresh<NW, 32, 8, 1>(a.packed + a.off.b[1], bias1, wave, lane);
            conv0_half_split_q<NW, LQH, 1>(patch, w0, bias0, act, 0, wave, lane);
            s3_prefetch_w0<NW, CB, CB, 32, 4, 1, TERMS>(a.packed + a.off.w_s3[1], wf1, wave, lane);
            __syncthreads();
            CNN_STAMP(2);
            conv3x3_mfma_s3q<NW, CB, CB, LQH, 1, 4, 1>(act, a.packed + a.off.w_s3[1], wf1, acc_a, wave, lane, false);
            __syncthreads();
            if (tid < LQH::SLOTS * 2 * 32) {                                 // pass 0 left conv0 row 16 in the bottom halo row: zero again (slots x 2 groups x 32 cells)
                const int t = tid >> 6, g = (tid >> 5) & 1, x = tid & 31;
                *reinterpret_cast<f32x4*>(base + g * LQH::GS + LQH::at(17, x + 1) + t * LQH::TSTEP) = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            conv0_half_split_q<NW, LQH, 1>(patch, w0, bias0, act, 1, wave, lane);
            s3_prefetch_w0<NW, CB, CB, 32, 4, 1, TERMS>(a.packed + a.off.w_s3[1], wf1, wave, lane);
            __syncthreads();
            conv3x3_mfma_s3q<NW, CB, CB, LQH, 1, 4, 1>(act, a.packed + a.off.w_s3[1], wf1, acc_b, wave, lane, false);
            CNN_STAMP(3);
            {
                int l2 = lane;
                asm volatile("" : "+v"(l2));
                bias2[0] = *reinterpret_cast<const f32x4*>(&a.packed[a.off.b[2] + (wave >> 2) * 16 + 4 * (l2 >> 4)]);
            }
            __syncthreads();
            zero_halo_q<LQH2, NTHR>(act);                                    // another group stride than LQH (bank conflicts of the stride-2 reader)
            store_tiles_split_q<CB, LQH2, 4, 1>(act, bias1, acc_a, wave, lane);
            s3_prefetch_w0<NW, CB, 2 * CB, 8, 2, 1, TERMS>(a.packed + a.off.w_s3[2], wf2, wave, lane);
            __syncthreads();
            CNN_STAMP(4);
            conv3x3_mfma_s3q<NW, CB, 2 * CB, LQH2, 2, 2, 1>(act, a.packed + a.off.w_s3[2], wf2, acc2_a, wave, lane, false);
            __syncthreads();
            store_tiles_split_q<CB, LQH2, 4, 1>(act, bias1, acc_b, wave, lane);
            if (wave == 7) {                                                 // conv1 row 15 (tiles 2, 3 of wave 7 in pass 0) = the top halo row of pass 1
                const int n = lane & 15;
#pragma unroll
                for (int i = 2; i < 4; ++i) split_store_tile_q<LQH2, 1>(base, LQH2::at(0, (i - 2) * 16 + n + 1), 0, bias1, acc_a[i], lane >> 4);
            }
            s3_prefetch_w0<NW, CB, 2 * CB, 8, 2, 1, TERMS>(a.packed + a.off.w_s3[2], wf2, wave, lane);
            __syncthreads();
            conv3x3_mfma_s3q<NW, CB, 2 * CB, LQH2, 2, 2, 1>(act, a.packed + a.off.w_s3[2], wf2, acc2_b, wave, lane, false);
        }
        CNN_STAMP(5);
        // conv3 / conv4: four / two pixel tiles x ONE channel tile per wave (probe: conv3 64.8 % of the pipe floor vs 58.2 % for 2 x 2, conv4 60.6 % vs
        // 39.4 % for 1 x 2 - a weight fragment that feeds a single pixel tile leaves the loop waiting on L2)
        S3W<1> wf3, wf4;
        f32x4 bias3[1], bias4[1];
        __syncthreads();
        zero_halo_q<LQ2, NTHR>(act);
        store_tiles_split_q<2 * CB, LQ2, 2, 1, 8>(act, bias2, acc2_a, wave, lane, 0);
        store_tiles_split_q<2 * CB, LQ2, 2, 1, 8>(act, bias2, acc2_b, wave, lane, 8);
        s3_prefetch_w0<NW, 2 * CB, 2 * CB, 16, 4, 1, TERMS>(a.packed + a.off.w_s3[3], wf3, wave, lane);
        prefetch_bias_fresh<NW, 16, 4, 1>(a.packed + a.off.b[3], bias3, wave, lane);
        __syncthreads();
        CNN_STAMP(6);
        {
            f32x4 acc_[4][1];                                            // conv3: 32 -> 32 @16x16
            conv3x3_mfma_s3q<NW, 2 * CB, 2 * CB, LQ2, 1, 4, 1>(act, a.packed + a.off.w_s3[3], wf3, acc_, wave, lane, false);
            CNN_STAMP(7);
            __syncthreads();
            store_tiles_split_q<2 * CB, LQ2, 4, 1>(act, bias3, acc_, wave, lane);      // in place: the halo is still zero
            s3_prefetch_w0<NW, 2 * CB, 4 * CB, 4, 2, 1, TERMS>(a.packed + a.off.w_s3[4], wf4, wave, lane);
            prefetch_bias_fresh<NW, 8, 2, 1>(a.packed + a.off.b[4], bias4, wave, lane);
            __syncthreads();
            CNN_STAMP(8);
        }
        S3W<T4N> wf5;
        f32x4 bias5s[T4N];
        {
            f32x4 acc_[2][1];                                            // conv4: 32 -> 64, stride 2 -> 8x8
            conv3x3_mfma_s3q<NW, 2 * CB, 4 * CB, LQ2, 2, 2, 1>(act, a.packed + a.off.w_s3[4], wf4, acc_, wave, lane, false);
            CNN_STAMP(9);
            __syncthreads();
            zero_halo_q<LQ4, NTHR>(act);
            store_tiles_split_q<4 * CB, LQ4, 2, 1>(act, bias4, acc_, wave, lane);
            s3_prefetch_w0<NW, 4 * CB, 4 * CB, 4, T4M, T4N, TERMS>(a.packed + a.off.w_s3[5], wf5, wave, lane);
            prefetch_bias_fresh<NW, 8, T4M, T4N>(a.packed + a.off.b[5], bias5s, wave, lane);
            __syncthreads();
            CNN_STAMP(10);
        }
        {
            f32x4 acc5[T4M][T4N];                                        // conv5: 64 -> 64 @8x8 in the exact path's tiling (the heads read it)
            conv3x3_mfma_s3q<NW, 4 * CB, 4 * CB, LQ4, 1, T4M, T4N>(act, a.packed + a.off.w_s3[5], wf5, acc5, wave, lane, false);
            CNN_STAMP(11);
            if constexpr (KIND != AFFNET_NET_HARDNET) {
                int lane_h = lane;                                       // opaque: 4 * (lane >> 4) is recomputed here, not carried (and spilled) from the kernel's top
                asm volatile("" : "+v"(lane_h));
                if constexpr (KIND == AFFNET_NET_ORINET)
                    head_partials_ori_lds<T4M, NTHR>(a.packed + a.off.head_w, bias5s, acc5, a.out + pidx * HEAD_PART_ORI, act, wave, lane_h, tid);
                else
                    head_partials<KIND, T4M>(a.packed + a.off.head_w, bias5s, acc5, a.out + pidx * HEAD_PART_AFF, wave, lane_h);
            }
            CNN_STAMP(13);
        }
        return;
    }

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
            else if constexpr (KIND == AFFNET_NET_ORINET)   // per-wave partial sums of the head's dot products: weights once, shifted activations from LDS
                head_partials_ori_lds<T4M, NTHR>(a.packed + a.off.head_w, bias5, acc, a.out + pidx * HEAD_PART_ORI, act, wave, lane, tid);
            else
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

// ---- AffNet / OriNet heads, second half: combine the eight per-wave partials of a patch ---------------------------------
//   AffNet : + bias -> tanh -> [[1+x0, 0],[x1, 1+x2]] -> rectifyAffineTransformationUpIsUp
//            (architectures.py:227-229,246-252, LAF.py:285-291); one thread per patch; optionally the shape filter of the row
//            (laf_ops.hip: aff_shape_filter_row) in the same kernel - the fused pipeline's finish + filter
//   OriNet : + bias -> tanh -> mean over the 3x3 map -> atan2 -> rotation (architectures.py:56-58,76-82, LAF.py:276-283); one
//            WAVEFRONT per patch: lane q < 18 adds the eight partials of tap q (18 consecutive floats per wave partial: coalesced;
//            one thread per patch read 144 floats at a 576-byte stride, 5.8x overfetch, 18 us for 2000 patches), the nine tanh
//            values of each output are added in tap order as before; optionally LAF <- LAF * R in the same kernel
//            (SparseImgRepresenter.py:173-177).
struct ShapeFuse {           // finish + shape filter in one kernel (pointers of image 0; strides like shape_filter_kernel)
    const float* resp; const float* lafs; float* key; int32_t* good; int32_t* cnt;
};
__global__ __launch_bounds__(256) void affnet_finish_kernel(const float* __restrict__ part, const float* __restrict__ hb,
                                                            const int32_t* __restrict__ count, int n_max, float* __restrict__ out, int row_begin,
                                                            int row_end, const int32_t* __restrict__ skip_cnt, int skip_n, ShapeFuse sf) {
    const int row = row_begin + blockIdx.x * 256 + threadIdx.x;
    const int n_img = count ? min(count[blockIdx.y], n_max) : n_max;
    const int n = min(n_img, row_end);
    const bool skip = lazy_skip(skip_cnt, skip_n, blockIdx.y);
    const size_t pidx = (size_t)blockIdx.y * n_max + row;
    if (sf.key) {
        if (blockIdx.x == 0 && threadIdx.x == 0)
            sf.cnt[(size_t)blockIdx.y * CNT_TOTAL + CNT_AFF_EVAL] = skip ? min(n_img, row_begin) : min(n_img, row_end);
        if (skip && row < n) { sf.key[pidx] = 0.0f; sf.good[pidx] = 0; }      // never evaluated: "not good" (no separate clearing pass)
    }
    if (row >= n || skip) return;
    float* o = out + 4 * pidx;
    const f32x4* pp = reinterpret_cast<const f32x4*>(part + pidx * HEAD_PART_AFF);
    f32x4 r[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) r[w] = pp[w];
    const f32x4 s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    const float x0 = tanhf(s.x + hb[0]), x1 = tanhf(s.y + hb[1]), x2 = tanhf(s.z + hb[2]);
    const float a00 = 1.0f + x0, a01 = 0.0f * x0, a10 = x1, a11 = 1.0f + x2;
    const float det = sqrtf(fabsf(a00 * a11 - a10 * a01 + 1e-10f));
    const float b2a2 = sqrtf(a01 * a01 + a00 * a00);
    const float o0 = b2a2 / det, o1 = 0.0f * det, o2 = (a11 * a01 + a10 * a00) / (b2a2 * det), o3 = det / b2a2;
    o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
    if (sf.key) {
        const size_t bi = blockIdx.y;
        aff_shape_filter_row(sf.resp + bi * n_max, sf.lafs + bi * n_max * 6, o0, o1, o2, o3, row, sf.key + bi * n_max, sf.good + bi * n_max,
                             sf.cnt + bi * CNT_TOTAL);
    }
}

__global__ __launch_bounds__(256) void orinet_finish_kernel(const float* __restrict__ part, const float* __restrict__ hb,
                                                            const int32_t* __restrict__ count, int n_max, float* __restrict__ out, int row_begin,
                                                            int row_end, float* __restrict__ rot_lafs, DenormSel ds) {
    const int row = row_begin + blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = min(count ? min(count[blockIdx.y], n_max) : n_max, row_end);
    if (row >= n) {
        // fused denormalisation (denorm_level_select_kernel's convention): pixel frames past the row count are cleared
        if (ds.out_px && row < n_max && lane < 6) ds.out_px[6 * ((size_t)blockIdx.y * n_max + row) + lane] = 0.f;
        return;
    }
    const size_t pidx = (size_t)blockIdx.y * n_max + row;
    const float* pp = part + pidx * HEAD_PART_ORI;
    float th = 0.f;
    if (lane < 18) {
        float p[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) p[w] = pp[w * 18 + lane];
        float r = 0.f;
#pragma unroll
        for (int w = 0; w < 8; w += 2) r += p[w] + p[w + 1];
        th = tanhf(r + hb[lane >= 9 ? 1 : 0]);
    }
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int q = 0; q < 9; ++q) { t0 += __shfl(th, q, 64); t1 += __shfl(th, 9 + q, 64); }
    if (lane != 0 && !(rot_lafs && ds.out_px)) return;                // (the fused level search below uses the whole wave: every lane carries the same row values)
    const float yv = t0 / 9.0f, xv = t1 / 9.0f;                       // AdaptiveAvgPool2d(1)
    const float ang = atan2f(yv + 1e-8f, xv + 1e-8f);                 // architectures.py:78
    const float sn = sinf(ang), cs = cosf(ang);
    float* o = out + 4 * pidx;
    if (lane == 0) { o[0] = cs; o[1] = sn; o[2] = -sn; o[3] = cs; }
    if (rot_lafs) {                                                   // apply_rotation_kernel (laf_ops.hip), same fmaf order
        float* L = rot_lafs + 6 * pidx;
        const float l00 = L[0], l01 = L[1], l10 = L[3], l11 = L[4], lx = L[2], ly = L[5];
        const float r0 = fmaf(l01, -sn, l00 * cs), r1 = fmaf(l01, cs, l00 * sn), r3 = fmaf(l11, -sn, l10 * cs), r4 = fmaf(l11, cs, l10 * sn);
        // the one-image latency path: denormalisation + pyramid-level choice of the row right here (was a launch of its own; same values, the level search
        // spread over the wave).  Every lane has read the frame BEFORE lane 0 overwrites it.
        if (ds.out_px)
            aff_denorm_level_row_wave(lane, r0, r1, lx, r3, r4, ly, ds.c_a, ds.c_x, ds.c_y, ds.ps, ds.lt, ds.ca, ds.cx, ds.cy, ds.out_px + 6 * pidx, ds.ids + 3 * pidx,
                                      ds.lafs_norm + 6 * pidx);
        if (lane == 0) { L[0] = r0; L[1] = r1; L[3] = r3; L[4] = r4; }
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
#define HEAD_KSPLIT 4
#define HEAD_KC 128
#define HEAD_AS (HEAD_KC + 4)    // row stride: 16-byte aligned rows
// MP = patches per workgroup: 64 (4 M-tiles per wave) is the throughput shape; 32 / 16 give small calls 2x / 4x as many workgroups
// (one image with 2000 keypoints is 32 x 4 workgroups of the 64-patch shape on 256 CUs: 85 us at 49 TFLOP/s).  The K order of every
// output's sum is the same for all three, so results do not depend on the shape.
template <int MP>
__global__ __launch_bounds__(256, 2) void hardnet_head_kernel(const float* __restrict__ trunk, const float* __restrict__ Bw,
                                                              const int32_t* __restrict__ count, int n_max, float* __restrict__ partial) {
    constexpr int MI = MP / 16;                  // M-tiles per wave
    constexpr int NA = MP * HEAD_KC / 4 / 256;   // float4 of the A slab per thread
    __shared__ __attribute__((aligned(16))) float As[MP * HEAD_AS];
    const int n = count ? min(count[blockIdx.z], n_max) : n_max;      // blockIdx.z = image of the batch
    const int p0 = blockIdx.x * MP;
    if (p0 >= n) return;
    const size_t rows_total = (size_t)gridDim.z * n_max;
    const int kbeg = blockIdx.y * (HEAD_K / HEAD_KSPLIT);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rA = weight_rsrc(trunk + (size_t)blockIdx.z * n_max * HEAD_K, n * HEAD_K);   // rows >= n -> 0
    const __amdgpu_buffer_rsrc_t rB = weight_rsrc(Bw, HEAD_K * 128);
    int offA[NA];
#pragma unroll
    for (int r = 0; r < NA; ++r) {
        const int f = tid + 256 * r, row = f >> 5, c4 = f & 31;       // 32 consecutive float4 = one 512-byte row segment
        offA[r] = ((p0 + row) * HEAD_K + 4 * c4) * 4;
    }
    const int offB = ((kq * 128) + wave * 32 + m) * 16;
    const unsigned a_addr = lds_byte_addr(As) + (m * HEAD_AS + 4 * kq) * 4;
    f32x4 acc[MI][2], stage[NA];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < NA; ++r) stage[r] = buf_read4(rA, offA[r], kbeg * 4);
#pragma unroll 1
    for (int k0 = kbeg; k0 < kbeg + HEAD_K / HEAD_KSPLIT; k0 += HEAD_KC) {
        __syncthreads();                                              // the previous slab has been consumed
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            const int f = tid + 256 * r, row = f >> 5, c4 = f & 31;
            *reinterpret_cast<f32x4*>(&As[row * HEAD_AS + 4 * c4]) = stage[r];
        }
        __syncthreads();
        if (k0 + HEAD_KC < kbeg + HEAD_K / HEAD_KSPLIT) {
#pragma unroll
            for (int r = 0; r < NA; ++r) stage[r] = buf_read4(rA, offA[r], (k0 + HEAD_KC) * 4);
        }
        f32x4 fa[2][MI], fb[2][2];
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[0][i] = lds_read4(a_addr + i * 16 * HEAD_AS * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[0][j] = buf_read4(rB, offB + j * 256, k0 * 512);
#pragma unroll
        for (int g = 0; g < HEAD_KC / 16; ++g) {
            const int cur = g & 1, nxt = cur ^ 1;
            if (g + 1 < HEAD_KC / 16) {
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[nxt][i] = lds_read4(a_addr + i * 16 * HEAD_AS * 4 + (g + 1) * 64);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[nxt][j] = buf_read4(rB, offB + j * 256, (k0 + 16 * (g + 1)) * 512);
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][i][s4], fb[cur][j][s4], acc[i][j], 0, 0, 0);
        }
    }
    // acc[i][j][r]: patch p0 + 16 i + 4 (lane>>4) + r, channel 32 wave + 16 j + (lane & 15)
    const int g = lane >> 4;
    float* dst = partial + ((size_t)blockIdx.y * rows_total + (size_t)blockIdx.z * n_max) * 128;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = p0 + 16 * i + 4 * g + r;
            if (row >= n) continue;
            dst[(size_t)row * 128 + wave * 32 + m] = acc[i][0][r];
            dst[(size_t)row * 128 + wave * 32 + 16 + m] = acc[i][1][r];
        }
}

// The same GEMM on split operands (AFFNET_ARITH_FP32_SPLIT3): the A slab is split ONCE per element while it is staged into LDS (each conv5
// element belongs to exactly one workgroup: M-tile x K-quarter), stored as term-interleaved 48-byte cells of 8 consecutive k
// (row pitch 128 k * 6 B + 16 B: the 16 rows of an M-tile fall into 16 different 16-byte bank slots), B = the pre-split head weights
// [k / 32][term][kq][n][8] straight from L2.  Six v_mfma_f32_16x16x32_bf16 per fp32 product in term-major order, fp32 accumulate; same
// partial-sum scratch and finish kernel as the exact path.
// TERMS = 2 (AFFNET_ARITH_FP32_SPLIT2H): two fp16 terms, three v_mfma_f32_16x16x32_f16 per product, the same cells with the third slot unused; the head
// weights are packed times 2^e, the partial sums are multiplied by 2^-e (behind the weights) before they are stored.
#define HEAD_S3_ROWB (HEAD_KC * 6 + 16)
template <int MP, int TERMS = 3>
__global__ __launch_bounds__(256, 2) void hardnet_head_s3_kernel(const float* __restrict__ trunk, const float* __restrict__ Bw3,
                                                                 const int32_t* __restrict__ count, int n_max, float* __restrict__ partial) {
    constexpr int MI = MP / 16;
    constexpr int NA = MP * HEAD_KC / 4 / 256;
    __shared__ __attribute__((aligned(16))) char As[MP * HEAD_S3_ROWB];
    const int n = count ? min(count[blockIdx.z], n_max) : n_max;
    const int p0 = blockIdx.x * MP;
    if (p0 >= n) return;
    const size_t rows_total = (size_t)gridDim.z * n_max;
    const int kbeg = blockIdx.y * (HEAD_K / HEAD_KSPLIT);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rA = weight_rsrc(trunk + (size_t)blockIdx.z * n_max * HEAD_K, n * HEAD_K);   // rows >= n -> 0
    const __amdgpu_buffer_rsrc_t rB = weight_rsrc(Bw3, HEAD_K * 128 * TERMS / 2);
    int offA[NA];
#pragma unroll
    for (int r = 0; r < NA; ++r) {
        const int f = tid + 256 * r, row = f >> 5, c4 = f & 31;
        offA[r] = ((p0 + row) * HEAD_K + 4 * c4) * 4;
    }
    const int offB = ((kq * 128) + wave * 32 + m) * 16;
    const unsigned a_addr = lds_byte_addr(reinterpret_cast<const float*>(As)) + m * HEAD_S3_ROWB + kq * 48;
    f32x4 acc[MI][2], stage[NA];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < NA; ++r) stage[r] = buf_read4(rA, offA[r], kbeg * 4);
#pragma unroll 1
    for (int k0 = kbeg; k0 < kbeg + HEAD_K / HEAD_KSPLIT; k0 += HEAD_KC) {
        __syncthreads();                                              // the previous slab has been consumed
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            const int f = tid + 256 * r, row = f >> 5, c4 = f & 31;
            char* dst = As + row * HEAD_S3_ROWB + (c4 >> 1) * 48 + (c4 & 1) * 8;      // cell = 8 consecutive k, this float4 = its lower / upper half
            split_store4<TERMS>(dst, stage[r]);
        }
        __syncthreads();
        if (k0 + HEAD_KC < kbeg + HEAD_K / HEAD_KSPLIT) {
#pragma unroll
            for (int r = 0; r < NA; ++r) stage[r] = buf_read4(rA, offA[r], (k0 + HEAD_KC) * 4);
        }
        bf16x8 fb[2][TERMS][2];                                       // [buffer][term][N-tile]
        auto load_b = [&](int buf, int ks) {
#pragma unroll
            for (int t = 0; t < TERMS; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    fb[buf][t][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rB, offB + j * 256, ((ks * TERMS + t) * 4 * 128) * 16, 0));
        };
        load_b(0, k0 >> 5);
#pragma unroll
        for (int s = 0; s < HEAD_KC / 32; ++s) {
            const int cur = s & 1;
            if (s + 1 < HEAD_KC / 32) load_b(cur ^ 1, (k0 >> 5) + s + 1);
            bf16x8 fa[MI][TERMS];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int t = 0; t < TERMS; ++t) fa[i][t] = __builtin_bit_cast(bf16x8, lds_read4(a_addr + i * 16 * HEAD_S3_ROWB + s * 192 + t * 16));
            // term pairs (a_i, b_j), i + j <= TERMS - 1, term-major
            constexpr int NPAIR = TERMS == 3 ? 6 : 3;
            constexpr int TA3[6] = {0, 0, 0, 1, 1, 2}, TB3[6] = {0, 1, 2, 0, 1, 0}, TA2[3] = {0, 0, 1}, TB2[3] = {0, 1, 0};
#pragma unroll
            for (int q = 0; q < NPAIR; ++q)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = split_mfma<TERMS>(fa[i][TERMS == 3 ? TA3[q] : TA2[q]], fb[cur][TERMS == 3 ? TB3[q] : TB2[q]][j], acc[i][j]);
        }
    }
    if constexpr (TERMS == 2) {
        const float osc = Bw3[(size_t)HEAD_K * 128];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] *= osc;
    }
    // acc[i][j][r]: patch p0 + 16 i + 4 (lane>>4) + r, channel 32 wave + 16 j + (lane & 15)
    const int g = lane >> 4;
    float* dst = partial + ((size_t)blockIdx.y * rows_total + (size_t)blockIdx.z * n_max) * 128;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = p0 + 16 * i + 4 * g + r;
            if (row >= n) continue;
            dst[(size_t)row * 128 + wave * 32 + m] = acc[i][0][r];
            dst[(size_t)row * 128 + wave * 32 + 16 + m] = acc[i][1][r];
        }
}

// One wavefront per patch: sum the K-split partials in fixed order, + BN bias, L2 normalise (eps 1e-8).  Rows past the image's row
// count are cleared here (the caller's descriptor buffer needs no separate fill).
__global__ __launch_bounds__(256) void hardnet_finish_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                                             const int32_t* __restrict__ count, int n_max, float* __restrict__ out) {
    const int n = count ? min(count[blockIdx.y], n_max) : n_max;      // blockIdx.y = image of the batch
    const int lrow = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (lrow >= n_max) return;
    const size_t rows_total = (size_t)gridDim.y * n_max, row = (size_t)blockIdx.y * n_max + lrow;
    if (lrow >= n) { out[row * 128 + lane] = 0.0f; out[row * 128 + 64 + lane] = 0.0f; return; }
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
                      bool mark_head = false, int row_begin = 0, int row_count = -1, const int32_t* skip_cnt = nullptr, int skip_n = 0,
                      const ShapeFuse* fuse = nullptr, int shape_op = 0, float* rot_lafs = nullptr, const DenormSel* denorm = nullptr) {
    if (kind < 0 || kind > 2) return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32: unknown net kind %d", kind);
    if (!packed || !out || n_max < 0 || (!patches && (!lafs || !ids))) return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32: null argument");
    if (!patches && !ctx->ws) return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32: sampling from the pyramid needs a bound workspace");
    if (dbg_layer < 0 && !scratch)
        return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32: d_scratch is required (HardNet n*(8192+512) floats, AffNet / OriNet n*144 floats)");
    if (dbg_layer >= 0 && ctx->arith != AFFNET_ARITH_FP32_MFMA)       // the split trunks have no per-layer dump: the exact kernel would answer
        return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32: layer dumps exist for AFFNET_ARITH_FP32_MFMA only (context is in arithmetic mode %d)", ctx->arith);
    if (n_max == 0) return AFFNET_OK;
    if (kind == AFFNET_NET_HARDNET && n_max > 65535) return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32: n_max=%d (HardNet head: max 65535 rows per image)", n_max);
    const NetLayout L = net_layout(kind);
    CnnArgs a;
    a.packed = packed; a.off = to_offsets(L); a.patches = patches; a.lafs = lafs; a.ids = ids; a.count = count; a.n_max = n_max;
    a.out = (dbg_layer < 0) ? scratch : out;              // trunk kernels: HardNet conv5 tensor / AffNet, OriNet head partials
    a.dbg_layer = dbg_layer; a.dbg_out = dbg_out; a.dbg_time = ctx->dbg_time;
    a.row_begin = row_begin; a.skip_cnt = skip_cnt; a.skip_n = skip_n;
    a.shape_cnt = (fuse && shape_op) ? fuse->cnt : nullptr; a.shape_op = shape_op;
    if (row_count < 0) row_count = n_max - row_begin;
    if (row_begin < 0 || row_count < 0 || row_begin + row_count > n_max) return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32: bad row window");
    if (row_count == 0) return AFFNET_OK;
    PyrSrc ps;
    aff_fill_pyr_src(ctx, &ps);
    const int B = patches ? 1 : ctx->B;                  // patch tensors are single-"image"; pyramid sampling covers the batch
    const dim3 grid(row_count, B);
    // (Tried and removed: two AffNet patches per persistent 16-wave workgroup in anti-phase - correct but 8 % slower, the
    // small-tile loops reach 85-90 % of the pipe rate with two waves per SIMD; 16-wave HardNet workgroups - slower too.)
#define TRUNK_LAUNCH(K) do { if (a.dbg_time || dbg_layer >= 0) hipLaunchKernelGGL((cnn32_trunk_kernel<K, 8, true>), grid, dim3(512), 0, st, a, ps); \
                             else hipLaunchKernelGGL((cnn32_trunk_kernel<K, 8, false>), grid, dim3(512), 0, st, a, ps); } while (0)
    a.s3_alt = ctx->split3_variant;
    const bool h2 = ctx->arith == AFFNET_ARITH_FP32_SPLIT2H;
    const bool split = ctx->arith == AFFNET_ARITH_FP32_SPLIT3 || h2;
    a.off = to_offsets(L, ctx->arith);                                   // the split copy of the active mode
    const bool s3 = split && !a.dbg_time && dbg_layer < 0;             // conv1 .. conv5 on split operands (affnet_set_arith)
#define SPLIT_LAUNCH(K, ST) do { if (h2) hipLaunchKernelGGL((cnn32_trunk_kernel<K, 8, ST, 2>), grid, dim3(512), 0, st, a, ps); \
                                 else hipLaunchKernelGGL((cnn32_trunk_kernel<K, 8, ST, 3>), grid, dim3(512), 0, st, a, ps); } while (0)
    if (split && a.dbg_time && dbg_layer < 0 && kind == AFFNET_NET_HARDNET) SPLIT_LAUNCH(AFFNET_NET_HARDNET, true);      // phase stamps of the split-operand trunks (tuning aid)
    else if (split && a.dbg_time && dbg_layer < 0 && kind == AFFNET_NET_AFFNET) SPLIT_LAUNCH(AFFNET_NET_AFFNET, true);
    else if (split && a.dbg_time && dbg_layer < 0 && kind == AFFNET_NET_ORINET) SPLIT_LAUNCH(AFFNET_NET_ORINET, true);
    else if (s3 && kind == AFFNET_NET_AFFNET) SPLIT_LAUNCH(AFFNET_NET_AFFNET, false);
    else if (s3 && kind == AFFNET_NET_ORINET) SPLIT_LAUNCH(AFFNET_NET_ORINET, false);
    else if (s3) SPLIT_LAUNCH(AFFNET_NET_HARDNET, false);
#undef SPLIT_LAUNCH
    else if (kind == AFFNET_NET_AFFNET) TRUNK_LAUNCH(AFFNET_NET_AFFNET);
    else if (kind == AFFNET_NET_ORINET) TRUNK_LAUNCH(AFFNET_NET_ORINET);
    else TRUNK_LAUNCH(AFFNET_NET_HARDNET);
#undef TRUNK_LAUNCH
    AFF_LAUNCH_CHECK(ctx);
    if (kind != AFFNET_NET_HARDNET && dbg_layer < 0) {       // combine the per-wave head partials in `scratch`
        if (kind == AFFNET_NET_AFFNET) {
            ShapeFuse sf;
            memset(&sf, 0, sizeof(sf));
            if (fuse) sf = *fuse;
            hipLaunchKernelGGL(affnet_finish_kernel, dim3(aff_cdiv(row_count, 256), B), dim3(256), 0, st, scratch, packed + L.head_b, count, n_max, out,
                               row_begin, row_begin + row_count, skip_cnt, skip_n, sf);
        } else {
            DenormSel ds;
            memset(&ds, 0, sizeof(ds));
            if (denorm && rot_lafs) ds = *denorm;
            hipLaunchKernelGGL(orinet_finish_kernel, dim3(aff_cdiv(row_count, 4), B), dim3(256), 0, st, scratch, packed + L.head_b, count, n_max, out,
                               row_begin, row_begin + row_count, rot_lafs, ds);
        }
        AFF_LAUNCH_CHECK(ctx);
    }
    if (mark_head) aff_prof_mark(ctx, 7, st);
    if (kind == AFFNET_NET_HARDNET && dbg_layer < 0) {
        float* partial = scratch + (size_t)B * n_max * HEAD_K;   // [HEAD_KSPLIT][B * n_max][128] behind the trunk output
        // patches per workgroup: the 64-patch shape once it gives every CU a workgroup, else 32 / 16 (same sums, more workgroups)
        int mp = (aff_cdiv(n_max, 64) * HEAD_KSPLIT * B >= 256) ? 64 : ((aff_cdiv(n_max, 32) * HEAD_KSPLIT * B >= 256) ? 32 : 16);
        if (const char* e = getenv("AFFNET_HEAD_MP")) { const int v = atoi(e); if (v == 16 || v == 32 || v == 64) mp = v; }   // tuning aid
        if (split) {              // AFFNET_ARITH_FP32_SPLIT3 / _SPLIT2H: the head GEMM on split operands as well
            const float* hw = packed + (h2 ? L.head_h2 : L.head_s3);
#define HEAD_LAUNCH(MPV) do { if (h2) hipLaunchKernelGGL((hardnet_head_s3_kernel<MPV, 2>), dim3(aff_cdiv(n_max, MPV), HEAD_KSPLIT, B), dim3(256), 0, st, scratch, hw, count, n_max, partial); \
                              else hipLaunchKernelGGL((hardnet_head_s3_kernel<MPV, 3>), dim3(aff_cdiv(n_max, MPV), HEAD_KSPLIT, B), dim3(256), 0, st, scratch, hw, count, n_max, partial); } while (0)
            if (mp == 64) HEAD_LAUNCH(64);
            else if (mp == 32) HEAD_LAUNCH(32);
            else HEAD_LAUNCH(16);
#undef HEAD_LAUNCH
        } else if (mp == 64)
            hipLaunchKernelGGL(hardnet_head_kernel<64>, dim3(aff_cdiv(n_max, 64), HEAD_KSPLIT, B), dim3(256), 0, st, scratch, packed + L.head_w, count, n_max, partial);
        else if (mp == 32)
            hipLaunchKernelGGL(hardnet_head_kernel<32>, dim3(aff_cdiv(n_max, 32), HEAD_KSPLIT, B), dim3(256), 0, st, scratch, packed + L.head_w, count, n_max, partial);
        else
            hipLaunchKernelGGL(hardnet_head_kernel<16>, dim3(aff_cdiv(n_max, 16), HEAD_KSPLIT, B), dim3(256), 0, st, scratch, packed + L.head_w, count, n_max, partial);
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

// Rows [row_begin, row_begin + row_count) of every image only, optionally under the lazy-evaluation predicate (see CnnArgs).
int aff_cnn_forward_pyr_rows(affnet_ctx* ctx, int kind, const float* packed, const float* lafs, const int32_t* ids, const int32_t* count, int n_max,
                             float* out, float* scratch, int row_begin, int row_count, const int32_t* skip_cnt, int skip_n, hipStream_t st) {
    if (kind == AFFNET_NET_HARDNET) return aff_fail(ctx, AFFNET_ERR_INVALID, "cnn32: row windows are for the AffNet / OriNet trunks");
    return cnn_launch(ctx, kind, packed, nullptr, lafs, ids, count, n_max, out, scratch, -1, nullptr, st, false, row_begin, row_count, skip_cnt, skip_n);
}

// AffNet on a row window with the shape filter of every evaluated row fused into the finish kernel (key / good / survivor count in
// the context's stage buffers) and the shape-stage counter bookkeeping done by the trunk launch (shape_op: see CnnArgs).
int aff_affnet_filter_rows(affnet_ctx* ctx, const float* packed, const float* resp, const float* lafs, const int32_t* ids, const int32_t* count,
                           float* out, float* scratch, int row_begin, int row_count, bool lazy, int shape_op, hipStream_t st) {
    ShapeFuse sf;
    sf.resp = resp; sf.lafs = lafs; sf.key = ctx->st_key; sf.good = ctx->st_good; sf.cnt = ctx->cnt;
    return cnn_launch(ctx, AFFNET_NET_AFFNET, packed, nullptr, lafs, ids, count, ctx->cap_pre, out, scratch, -1, nullptr, st, false, row_begin, row_count,
                      lazy ? ctx->cnt : nullptr, ctx->cfg.num_features, &sf, shape_op);
}

// OriNet with LAF <- LAF * R applied by the finish kernel (d_lafs rotated in place).
// denorm != NULL: the finish kernel also denormalises the rotated frame, chooses its pyramid level and writes the re-normalised frame (aff_denorm_level_select's
// work, one launch less per call).
int aff_orinet_rotate(affnet_ctx* ctx, const float* packed, float* lafs, const int32_t* ids, const int32_t* count, int n_max, float* out, float* scratch,
                      hipStream_t st, const DenormSel* denorm) {
    return cnn_launch(ctx, AFFNET_NET_ORINET, packed, nullptr, lafs, ids, count, n_max, out, scratch, -1, nullptr, st, false, 0, -1, nullptr, 0, nullptr, 0,
                      lafs, denorm);
}

int aff_hardnet_forward_pyr_marked(affnet_ctx* ctx, const float* packed, const float* lafs, const int32_t* ids, const int32_t* count,
                                   int n_max, float* out, float* scratch, hipStream_t st) {
    return cnn_launch(ctx, AFFNET_NET_HARDNET, packed, nullptr, lafs, ids, count, n_max, out, scratch, -1, nullptr, st, true);
}

extern "C" int affnet_debug_split3_variant(affnet_ctx* ctx, int bits) {
    if (!ctx) return AFFNET_ERR_INVALID;
    ctx->split3_variant = bits;             // bit 0: HardNet loops with alternating wave priorities (A/B aid; default off since round 4)
    return AFFNET_OK;
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

#ifdef AFFNET_PROBES   // libaffnet_hip_probes.so only (include/affnet_hip_probes.h)
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

#endif  // AFFNET_PROBES

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
