// Numerics / rate probes of the split-operand arithmetic (include/affnet_hip_probes.h, libaffnet_hip_probes.so only; the product kernels of AFFNET_ARITH_FP32_SPLIT3 are in cnn32.hip / fullconv.hip).  fp32 = three bf16 terms: x = x0 + x1 + x2 with every term
// rounded to bf16 captures the 24-bit significand exactly, every bf16 x bf16 product is exact in the fp32 accumulator of
// v_mfma_f32_16x16x32_bf16, and six of the nine term products (i + j <= 2) reproduce an fp32 product to 2^-25 relative.  At 16x the
// fp32 matrix rate that is a 2.67x higher ceiling for the CNN stages.  Two probes:
//   affnet_split3_gemm   C = A B^T on fp32 operands: exact-fp32 MFMA chain / 6-term / 9-term / plain bf16 - numerics on the hardware
//   affnet_split3_rate   the inner-loop shape a trunk layer would have (operand fragments from LDS, 4 pixel tiles x 1 channel tile x
//                        6 terms = 24 MFMAs per 32-k step): sustained fp32-equivalent rate
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf16_rne(float x) {           // fp32 -> nearest-even bf16, as fp32
    unsigned u = __float_as_uint(x);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    return __uint_as_float(u);
}

struct Split8 { bf16x8 t[3]; };

// 8 consecutive fp32 -> three bf16x8 fragments (x0, x1, x2)
__device__ __forceinline__ Split8 split8(const float* __restrict__ p) {
    Split8 s;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = p[i];
        const float x0 = bf16_rne(x);
        const float r1 = x - x0;
        const float x1 = bf16_rne(r1);
        const float x2 = bf16_rne(r1 - x1);
        s.t[0][i] = (__bf16)x0; s.t[1][i] = (__bf16)x1; s.t[2][i] = (__bf16)x2;      // exact conversions: the low 16 bits are zero
    }
    return s;
}

// One wave = one 16 x 16 tile of C = A (M x K, row-major) x Bt^T (Bt: N x K, row-major).  mode 0: v_mfma_f32_16x16x4_f32 chain,
// 1: six split terms, 2: nine, 3: the leading bf16 term only.
__global__ __launch_bounds__(64) void split3_gemm_kernel(const float* __restrict__ A, const float* __restrict__ Bt, int M, int N, int K, int mode,
                                                         float* __restrict__ C) {
    const int lane = threadIdx.x, m = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.y * 16, col0 = blockIdx.x * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* ap = A + (size_t)(row0 + m) * K;
    const float* bp = Bt + (size_t)(col0 + m) * K;
    if (mode == 0) {
        for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[k + kq], bp[k + kq], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 32) {
            const Split8 a = split8(ap + k + kq * 8), b = split8(bp + k + kq * 8);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const bool use = mode == 2 || (mode == 1 && i + j <= 2) || (mode == 3 && i + j == 0);
                    if (use) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.t[i], b.t[j], acc, 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) C[(size_t)(row0 + 4 * kq + r) * N + col0 + m] = acc[r];
}

extern "C" int affnet_split3_gemm(const float* d_A, const float* d_Bt, int M, int N, int K, int mode, float* d_C, void* stream) {
    if (!d_A || !d_Bt || !d_C || M % 16 || N % 16 || K % 32 || mode < 0 || mode > 3) return AFFNET_ERR_INVALID;
    hipLaunchKernelGGL(split3_gemm_kernel, dim3(N / 16, M / 16), dim3(64), 0, (hipStream_t)stream, d_A, d_Bt, M, N, K, mode, d_C);
    return hipGetLastError() == hipSuccess ? AFFNET_OK : AFFNET_ERR_HIP;
}

// Inner-loop shape of a trunk layer on split operands: per 32-k step a wave reads 4 activation tiles x 3 terms from LDS (12 ds_read_b128)
// and 1 weight fragment x 3 terms (held in registers here: in a trunk they stream from L2 like today's fp32 fragments), then issues
// 4 tiles x 6 terms = 24 MFMAs.  terms = 6 or 9 (9: 36 MFMAs); terms = 1: the fp32 16x16x4 loop of the same 4 tiles (8 MFMAs per 32 k...
// i.e. 32 MFMAs of k = 4) for the same-kernel comparison.  512 threads, LDS sized like the HardNet trunk (1 workgroup per CU).
__global__ __launch_bounds__(512, 2) void split3_rate_kernel(int reps, int terms, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[38 * 1024];            // 152 KB: one workgroup per CU, as the HardNet trunk
    for (int i = threadIdx.x; i < 38 * 1024; i += 512) lds[i] = 0.001f * (float)((i * 7) & 255);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8* lb = reinterpret_cast<const bf16x8*>(lds);
    bf16x8 w[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) w[i] = lb[(wave * 64 + lane + i * 512) & 2047];
    if (terms == 1) {
        const float wf = lds[lane];
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {                                     // 32 k = 2 groups of 16 k: one ds_read_b128 = 4 k-steps per tile
                f32x4 a[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const f32x4*>(&lds[((r * 8 + g * 4 + t) * 256 + wave * 1024 + lane * 4) & (38 * 1024 - 4)]);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf, a[t][s4], acc[t], 0, 0, 0);
            }
        }
    } else {
        for (int r = 0; r < reps; ++r) {
            bf16x8 a[4][3];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 3; ++i) a[t][i] = lb[((r * 12 + t * 3 + i) * 64 + wave * 512 + lane) & 8191];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (terms == 6 && i + j > 2) continue;
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i], a[t][j], acc[t], 0, 0, 0);
                }
        }
    }
    float sink = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) sink += acc[t][0] + acc[t][3];
    if (sink == 12345.678f) out[0] = sink;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = sink;
}

extern "C" int affnet_split3_rate(int reps, int terms, int n_blocks, float* d_out, void* stream) {
    if (!d_out || (terms != 1 && terms != 6 && terms != 9) || reps < 1 || n_blocks < 1) return AFFNET_ERR_INVALID;
    hipLaunchKernelGGL(split3_rate_kernel, dim3(n_blocks), dim3(512), 0, (hipStream_t)stream, reps, terms, d_out);
    return hipGetLastError() == hipSuccess ? AFFNET_OK : AFFNET_ERR_HIP;
}
