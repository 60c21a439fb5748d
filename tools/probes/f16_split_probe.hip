// Tuning / numerics aid (not product code) for the two-term fp16 split arithmetic (AFFNET_ARITH_FP32_SPLIT2H):
//   1. does v_mfma_f32_16x16x32_f16 take SUBNORMAL fp16 inputs un-flushed (the low terms of small activations are subnormal)?
//   2. is the device split  x -> (hi = f16(x), lo = f16(x - hi))  bit-identical to the host's round-to-nearest-even split?
//   3. rate of a pure three-product f16 MFMA stream next to the six-product bf16 stream on random operands (same socket power cap).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/f16_split_probe tools/probes/f16_split_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../affnet_amd/csrc/cnn_mfma.h"      // split_h2_pair: the routine the product kernels use

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// D[i][j] = sum_k First[i][k] * Second[j][k]; lane (i or j = lane & 15, k = 8 (lane >> 4) .. + 7); D: j = lane & 15, i = 4 (lane >> 4) + reg
__global__ void subnormal_kernel(const uint16_t* __restrict__ first, const uint16_t* __restrict__ second, float* __restrict__ d) {
    const int lane = threadIdx.x, r = lane & 15, kq = lane >> 4;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = __builtin_bit_cast(_Float16, first[r * 32 + kq * 8 + e]);
        b[e] = __builtin_bit_cast(_Float16, second[r * 32 + kq * 8 + e]);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    for (int e = 0; e < 4; ++e) d[(4 * kq + e) * 16 + r] = acc[e];
}

__global__ void split_kernel(const float* __restrict__ x, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, int n) {
    const int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (i + 1 >= n) return;
    const f32x2 v = {x[i], x[i + 1]};
    unsigned h, l;
    split_h2_pair(v, h, l);                       // cnn_mfma.h (v_cvt_pk_f16_f32 + v_fma_mix_f32 in assembly; the compiler's vector conversion is wrong)
    hi[i] = (uint16_t)(h & 0xffffu); hi[i + 1] = (uint16_t)(h >> 16);
    lo[i] = (uint16_t)(l & 0xffffu); lo[i + 1] = (uint16_t)(l >> 16);
}

// 8 accumulators per wave, NPROD MFMAs per accumulator and iteration on register operands
template <bool HALF>
__global__ __launch_bounds__(512) void stream_kernel(const uint32_t* __restrict__ seed, int iters, float* __restrict__ out) {
    uint32_t s = seed[threadIdx.x & 63] + threadIdx.x * 2654435761u;
    auto next = [&]() { s = s * 1664525u + 1013904223u; return s; };
    uint32_t wv[3][4], av[3][4];
    for (int t = 0; t < 3; ++t)
        for (int e = 0; e < 4; ++e) {         // finite, moderate operands: exponent bits from a narrow range
            wv[t][e] = HALF ? ((next() & 0x83ff83ffu) | 0x34003400u) : ((next() & 0x807f807fu) | 0x3e003e00u);
            av[t][e] = HALF ? ((next() & 0x83ff83ffu) | 0x34003400u) : ((next() & 0x807f807fu) | 0x3e003e00u);
        }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (HALF) {
            f16x8 w[2], a[2];
            for (int t = 0; t < 2; ++t) { memcpy(&w[t], wv[t], 16); memcpy(&a[t], av[t], 16); }
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], a[0], acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[1], a[0], acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], a[1], acc[i], 0, 0, 0);
            }
        } else {
            bf16x8 w[3], a[3];
            for (int t = 0; t < 3; ++t) { memcpy(&w[t], wv[t], 16); memcpy(&a[t], av[t], 16); }
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                const int tw = p < 3 ? p : (p < 5 ? p - 3 : 0), ta = p < 3 ? 0 : (p < 5 ? 1 : 2);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[tw], a[ta], acc[i], 0, 0, 0);
            }
        }
        asm volatile("" : "+v"(wv[0][0]));
    }
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (r == 12345.678f) out[0] = r;
}

static uint16_t f16_bits(float x) { _Float16 h = (_Float16)x; uint16_t u; memcpy(&u, &h, 2); return u; }
static float f16_val(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    // 1. subnormal inputs
    {
        std::vector<uint16_t> first(16 * 32, 0), second(16 * 32, 0);
        for (int i = 0; i < 16; ++i) first[i * 32 + 0] = (uint16_t)(1 + i);           // subnormal: (1 + i) * 2^-24
        for (int j = 0; j < 16; ++j) second[j * 32 + 0] = f16_bits(1024.0f);          // 2^10
        for (int i = 0; i < 16; ++i) first[i * 32 + 9] = f16_bits(3.0f);              // a normal product as the control
        for (int j = 0; j < 16; ++j) second[j * 32 + 9] = (uint16_t)(0x0200);         // subnormal 2^-15 on the SECOND operand
        uint16_t *df, *ds; float* dd;
        CK(hipMalloc(&df, 1024)); CK(hipMalloc(&ds, 1024)); CK(hipMalloc(&dd, 1024));
        CK(hipMemcpy(df, first.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(ds, second.data(), 1024, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(subnormal_kernel, dim3(1), dim3(64), 0, 0, df, ds, dd);
        std::vector<float> d(256);
        CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                const float want = (float)(1 + i) * ldexpf(1.0f, -14) + 3.0f * ldexpf(1.0f, -15);
                if (d[i * 16 + j] != want) { if (bad < 4) printf("  D[%d][%d] = %.9g, want %.9g\n", i, j, d[i * 16 + j], want); ++bad; }
            }
        printf("subnormal fp16 inputs of v_mfma_f32_16x16x32_f16: %s (%d of 256 outputs differ)\n", bad ? "FLUSHED or wrong" : "honoured, exact", bad);
    }
    // 2. the device split against the host's
    {
        const int n = 1 << 22;
        std::vector<float> x(n);
        uint32_t s = 12345;
        for (int i = 0; i < n; ++i) {
            s = s * 1664525u + 1013904223u;
            const uint32_t mant = (s >> 8) & 0x007fffffu, sign = s & 0x80000000u;      // high bits: the low bits of an LCG have short periods (bit 0 alternates)
            s = s * 1664525u + 1013904223u;
            const uint32_t ex = 127 - 30 + (s >> 8) % 46;                // 2^-30 .. 2^15
            const uint32_t u = sign | (ex << 23) | mant;
            memcpy(&x[i], &u, 4);
        }
        float* dx; uint16_t *dh, *dl;
        CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dh, n * 2)); CK(hipMalloc(&dl, n * 2));
        CK(hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(split_kernel, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dh, dl, n);
        std::vector<uint16_t> h(n), l(n);
        CK(hipMemcpy(h.data(), dh, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(l.data(), dl, n * 2, hipMemcpyDeviceToHost));
        long bad = 0; double worst_rel = 0, worst_abs = 0;
        for (int i = 0; i < n; ++i) {
            const uint16_t hh = f16_bits(x[i]);
            const float r = x[i] - f16_val(hh);
            const uint16_t ll = f16_bits(r);
            if (hh != h[i] || ll != l[i]) { if (bad < 4) printf("  x = %.9g: device (%04x, %04x) host (%04x, %04x)\n", x[i], h[i], l[i], hh, ll); ++bad; }
            const double err = fabs((double)f16_val(h[i]) + (double)f16_val(l[i]) - (double)x[i]);
            if (fabs(x[i]) >= 0.25) { if (err / fabs((double)x[i]) > worst_rel) worst_rel = err / fabs((double)x[i]); }      // both terms in fp16's normal range
            else if (err > worst_abs) worst_abs = err;
        }
        printf("device split vs host round-to-nearest-even split: %ld of %d differ; worst |hi + lo - x| / |x| over |x| >= 2^-2: %.3g (2^-23 = %.3g); worst |hi + lo - x| "
               "over |x| < 2^-2: %.3g (2^-25 = %.3g)\n", bad, n, worst_rel, ldexp(1.0, -23), worst_abs, ldexp(1.0, -25));
    }
    // 3. stream rates
    {
        uint32_t* dseed; float* dout;
        std::vector<uint32_t> seed(64);
        for (int i = 0; i < 64; ++i) seed[i] = 777u * (i + 1);
        CK(hipMalloc(&dseed, 256)); CK(hipMalloc(&dout, 16));
        CK(hipMemcpy(dseed, seed.data(), 256, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int half = 0; half < 2; ++half) {
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (half) hipLaunchKernelGGL(stream_kernel<true>, dim3(1024), dim3(512), 0, 0, dseed, iters, dout);
                else hipLaunchKernelGGL(stream_kernel<false>, dim3(1024), dim3(512), 0, 0, dseed, iters, dout);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double mfma = 1024.0 * 8 * iters * 48;
                printf("%s stream: %.2f ms, %.1f TF/s of MFMA work (%.1f %% of 2516.8), fp32-equivalent products at %.1f TF/s\n", half ? "f16 x2 (3 products)" : "bf16 x3 (6 products)", ms,
                       mfma * 16384 / ms * 1e-9, mfma * 16384 / ms * 1e-9 / 2516.8 * 100, mfma * 16384 / ms * 1e-9 / (half ? 3 : 6));
            }
        }
    }
    return 0;
}
