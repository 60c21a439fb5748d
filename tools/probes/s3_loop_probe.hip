// Tuning aid (not product code): the split-operand MFMA loops of the trunks in isolation, old (LayB planes, tile-major MFMA order:
// conv3x3_mfma_s3p) against new (LayQ term-interleaved cells, term-major order: conv3x3_mfma_s3q), on the real layer shapes with the
// real LDS footprint (1 workgroup per CU for the HardNet shapes, 2 for the AffNet ones) and weights streaming from L2.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/probes/s3_loop_probe tools/probes/s3_loop_probe.hip
//   tools/probes/s3_loop_probe [reps] [blocks]
// Prints per (shape, variant): cycles per loop call (s_memtime, slowest wave of the sampled workgroups), the matrix-pipe floor of the
// call and the wall-clock rate of the launch against the dense bf16 MFMA peak.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../../affnet_amd/csrc/cnn_mfma.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// shapes: 0 HardNet conv1 (half patch) 32->32 @16x32 | 1 HardNet conv2 (half) 32->64 s2 | 2 HardNet conv3 64->64 @16x16
//         3 HardNet conv4 64->128 s2 | 4 HardNet conv5 128->128 @8x8 | 5 AffNet conv3 32->32 @16x16 | 6 AffNet conv5 64->64 @8x8
template <int SHAPE> struct Shape;
template <> struct Shape<0> { static constexpr int CB = 32, CIN = 32, COUT = 32, STRIDE = 1, TM = 4, TN = 2; typedef LayB<16, 32, 34, 32> LB; typedef LayQ<16, 32, 34, 32> LQ; static const char* name() { return "HardNet conv1 (half patch) 32->32 @16x32"; } };
template <> struct Shape<1> { static constexpr int CB = 32, CIN = 32, COUT = 64, STRIDE = 2, TM = 2, TN = 2; typedef LayB<16, 32, 34, 32, 16> LB; typedef LayQ<16, 32, 34, 32, 16> LQ; static const char* name() { return "HardNet conv2 (half patch) 32->64 stride 2"; } };
template <> struct Shape<2> { static constexpr int CB = 32, CIN = 64, COUT = 64, STRIDE = 1, TM = 4, TN = 2; typedef LayB<16, 16, 18, 64> LB; typedef LayQ<16, 16, 18, 64> LQ; static const char* name() { return "HardNet conv3 64->64 @16x16"; } };
template <> struct Shape<3> { static constexpr int CB = 32, CIN = 64, COUT = 128, STRIDE = 2, TM = 2, TN = 2; typedef LayB<16, 16, 18, 64> LB; typedef LayQ<16, 16, 18, 64> LQ; static const char* name() { return "HardNet conv4 64->128 stride 2"; } };
template <> struct Shape<4> { static constexpr int CB = 32, CIN = 128, COUT = 128, STRIDE = 1, TM = 2, TN = 2; typedef LayB<8, 8, 16, 128, 128> LB; typedef LayQ<8, 8, 16, 128, 128> LQ; static const char* name() { return "HardNet conv5 128->128 @8x8"; } };
template <> struct Shape<5> { static constexpr int CB = 16, CIN = 32, COUT = 32, STRIDE = 1, TM = 2, TN = 2; typedef LayB<16, 16, 18, 32> LB; typedef LayQ<16, 16, 18, 32> LQ; static const char* name() { return "AffNet conv3 32->32 @16x16 (2 workgroups / CU)"; } };
template <> struct Shape<6> { static constexpr int CB = 16, CIN = 64, COUT = 64, STRIDE = 1, TM = 2, TN = 1; typedef LayB<8, 8, 16, 64, 128> LB; typedef LayQ<8, 8, 16, 64, 128> LQ; static const char* name() { return "AffNet conv5 64->64 @8x8 (2 workgroups / CU)"; } };

// NW = 4 variants (ONE wave per SIMD, 16 tiles per wave, 512 registers): tiles per wave for the HardNet shapes
template <int SHAPE> struct Tile4;
template <> struct Tile4<0> { static constexpr int TM = 8, TN = 2; };
template <> struct Tile4<1> { static constexpr int TM = 2, TN = 4; };
template <> struct Tile4<2> { static constexpr int TM = 4, TN = 4; };
template <> struct Tile4<3> { static constexpr int TM = 2, TN = 4; };
template <> struct Tile4<4> { static constexpr int TM = 2, TN = 4; };

// alternative register blockings of the HardNet shapes (8 waves): more pixel tiles per weight fragment
template <> struct Shape<7> { static constexpr int CB = 32, CIN = 32, COUT = 32, STRIDE = 1, TM = 8, TN = 1; typedef LayB<16, 32, 34, 32> LB; typedef LayQ<16, 32, 34, 32> LQ; static const char* name() { return "HardNet conv1 (half) TM 8 x TN 1"; } };
template <> struct Shape<8> { static constexpr int CB = 32, CIN = 64, COUT = 64, STRIDE = 1, TM = 8, TN = 1; typedef LayB<16, 16, 18, 64> LB; typedef LayQ<16, 16, 18, 64> LQ; static const char* name() { return "HardNet conv3 TM 8 x TN 1"; } };
template <> struct Shape<9> { static constexpr int CB = 32, CIN = 64, COUT = 128, STRIDE = 2, TM = 4, TN = 1; typedef LayB<16, 16, 18, 64> LB; typedef LayQ<16, 16, 18, 64> LQ; static const char* name() { return "HardNet conv4 TM 4 x TN 1"; } };
template <> struct Shape<10> { static constexpr int CB = 32, CIN = 128, COUT = 128, STRIDE = 1, TM = 4, TN = 1; typedef LayB<8, 8, 16, 128, 128> LB; typedef LayQ<8, 8, 16, 128, 128> LQ; static const char* name() { return "HardNet conv5 TM 4 x TN 1"; } };
template <> struct Shape<11> { static constexpr int CB = 32, CIN = 128, COUT = 128, STRIDE = 1, TM = 1, TN = 4; typedef LayB<8, 8, 16, 128, 128> LB; typedef LayQ<8, 8, 16, 128, 128> LQ; static const char* name() { return "HardNet conv5 TM 1 x TN 4"; } };
template <> struct Shape<12> { static constexpr int CB = 32, CIN = 32, COUT = 64, STRIDE = 2, TM = 4, TN = 1; typedef LayB<16, 32, 34, 32, 16> LB; typedef LayQ<16, 32, 34, 32, 16> LQ; static const char* name() { return "HardNet conv2 (half) TM 4 x TN 1"; } };

template <> struct Shape<13> { static constexpr int CB = 16, CIN = 32, COUT = 32, STRIDE = 1, TM = 4, TN = 1; typedef LayB<16, 16, 18, 32> LB; typedef LayQ<16, 16, 18, 32> LQ; static const char* name() { return "AffNet conv3 TM 4 x TN 1"; } };
template <> struct Shape<14> { static constexpr int CB = 16, CIN = 32, COUT = 64, STRIDE = 2, TM = 1, TN = 2; typedef LayB<16, 16, 18, 32> LB; typedef LayQ<16, 16, 18, 32> LQ; static const char* name() { return "AffNet conv4 32->64 stride 2, TM 1 x TN 2 (trunk)"; } };
template <> struct Shape<15> { static constexpr int CB = 16, CIN = 32, COUT = 64, STRIDE = 2, TM = 2, TN = 1; typedef LayB<16, 16, 18, 32> LB; typedef LayQ<16, 16, 18, 32> LQ; static const char* name() { return "AffNet conv4 TM 2 x TN 1"; } };
template <> struct Shape<16> { static constexpr int CB = 16, CIN = 64, COUT = 64, STRIDE = 1, TM = 1, TN = 2; typedef LayB<8, 8, 16, 64, 128> LB; typedef LayQ<8, 8, 16, 64, 128> LQ; static const char* name() { return "AffNet conv5 TM 1 x TN 2"; } };

template <int SHAPE, int VAR> struct PB { static constexpr int v = 0; };
// VAR 4: new loop, 4 waves per workgroup (one per SIMD), 5: the other tile split; 6 / 7 / 8: VAR 4 without weight loads / without fragment
// reloads / without either (pure MFMA stream)
template <int SHAPE, int VAR>
__global__ __launch_bounds__(256, 1) void probe4_kernel(const float* __restrict__ Ws, int reps, float* __restrict__ out, unsigned long long* __restrict__ cyc) {
    typedef Shape<SHAPE> S;
    constexpr int NW = 4;
    constexpr int TM = (VAR != 5) ? Tile4<SHAPE>::TM : Tile4<SHAPE>::TM * 2, TN = (VAR != 5) ? Tile4<SHAPE>::TN : Tile4<SHAPE>::TN / 2;
    constexpr int PBITS = VAR == 6 ? 1 : (VAR == 7 ? 2 : (VAR == 8 ? 3 : 0));
    __shared__ __attribute__((aligned(16))) float lds[(S::CB / 4) * LayC0::PSG + WP32 * WP32 + 256];
    for (int i = threadIdx.x; i < (S::CB / 4) * LayC0::PSG + WP32 * WP32 + 256; i += 256) {
        const unsigned h = (unsigned)i * 2654435761u;
        const unsigned lo = 0x3C00u | ((h >> 3) & 0x3FFu), hi = 0x3C00u | ((h >> 17) & 0x3FFu);
        lds[i] = __uint_as_float(lo | (hi << 16));
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float sink = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        f32x4 acc[TM][TN];
        constexpr int MT = (S::LQ::H / S::STRIDE) * (S::LQ::W / S::STRIDE) / 16;
        S3W<TN> w0;
        s3_prefetch_w0<NW, S::CIN, S::COUT, MT, TM, TN>(Ws, w0, wave, lane);
        conv3x3_mfma_s3q<NW, S::CIN, S::COUT, typename S::LQ, S::STRIDE, TM, TN, PBITS>(lds, Ws, w0, acc, wave, lane, false);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) sink += acc[i][j][0] + acc[i][j][3];
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x < 64) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if (sink == 12345.678f) out[0] = sink;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = sink;
}

template <int SHAPE, int VAR>      // VAR 0 = old loop (LayB), 1 = new loop (LayQ), 2 = new loop without alternating priorities, 3 = old without, 15 = VAR 2 on TWO fp16 terms
__global__ __launch_bounds__(512, (Shape<SHAPE>::CB == 32) ? 2 : 4) void probe_kernel(const float* __restrict__ Ws, int reps, float* __restrict__ out,
                                                                                     unsigned long long* __restrict__ cyc) {
    typedef Shape<SHAPE> S;
    constexpr int NW = 8;
    __shared__ __attribute__((aligned(16))) float lds[(S::CB / 4) * LayC0::PSG + WP32 * WP32 + 256];      // = TrunkLds<CB>::TOTAL
    for (int i = threadIdx.x; i < (S::CB / 4) * LayC0::PSG + WP32 * WP32 + 256; i += 512) {
        // bf16 pairs of moderate magnitude (exponent ~ 2^-3 .. 2^0), different per address
        const unsigned h = (unsigned)i * 2654435761u;
        const unsigned lo = 0x3C00u | ((h >> 3) & 0x3FFu), hi = 0x3C00u | ((h >> 17) & 0x3FFu);
        lds[i] = __uint_as_float(lo | (hi << 16));
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool alt = (VAR == 0 || VAR == 1) && S::CB == 32;      // VAR >= 9: diagnostics on the new loop without alternating priorities
    float sink = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        f32x4 acc[S::TM][S::TN];
        if constexpr (VAR == 0 || VAR == 3) {
            conv3x3_mfma_s3p<NW, S::CIN, S::COUT, typename S::LB, S::STRIDE, S::TM, S::TN>(lds, Ws, acc, wave, lane, alt);
        } else {
            constexpr int MT = (S::LQ::H / S::STRIDE) * (S::LQ::W / S::STRIDE) / 16;
            S3W<S::TN> w0;
            s3_prefetch_w0<NW, S::CIN, S::COUT, MT, S::TM, S::TN>(Ws, w0, wave, lane);
            constexpr int PBITS = VAR == 9 ? 1 : (VAR == 10 ? 2 : (VAR == 11 ? 3 : (VAR == 12 ? (1 << 4) : (VAR == 13 ? (3 << 4) : (VAR == 14 ? (6 << 4) : 0)))));
            if constexpr (VAR == 15) {          // AFFNET_ARITH_FP32_SPLIT2H: the same cells and loop with two fp16 terms, three products
                // the trunks' two-term layouts (LayR: 16-byte pixels; row pitch / group stride per reader as in cnn32.hip)
                typedef LayR<S::LQ::H, S::LQ::W, (S::LQ::W == 16 ? 20 : (S::LQ::W == 8 ? 12 : 34)), S::LQ::C, ((S::STRIDE == 2 || (S::LQ::W == 16 && S::CB == 32)) ? 16 : 0)> LQ2T;
                s3_prefetch_w0<NW, S::CIN, S::COUT, MT, S::TM, S::TN, 2>(Ws, w0, wave, lane);
                conv3x3_mfma_s3q<NW, S::CIN, S::COUT, LQ2T, S::STRIDE, S::TM, S::TN, 0>(lds, Ws, w0, acc, wave, lane, alt);
            } else
            conv3x3_mfma_s3q<NW, S::CIN, S::COUT, typename S::LQ, S::STRIDE, S::TM, S::TN, PBITS>(lds, Ws, w0, acc, wave, lane, alt);
        }
#pragma unroll
        for (int i = 0; i < S::TM; ++i)
#pragma unroll
            for (int j = 0; j < S::TN; ++j) sink += acc[i][j][0] + acc[i][j][3];
        __syncthreads();          // a layer ends at a barrier: the next call starts with all waves together, like in the trunk
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x < 64) cyc[blockIdx.x * NW + wave] = t1 - t0;
    if (sink == 12345.678f) out[0] = sink;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = sink;
}

template <int SHAPE, int VAR>
static void run(const float* dW, float* dout, unsigned long long* dcyc, int reps, int blocks, double clock_ghz) {
    typedef Shape<SHAPE> S;
    constexpr int NS = 9 * (S::CIN / 32);
    const int nwaves = (VAR >= 4 && VAR <= 8) ? 4 : 8;
    const double mfma_per_wave = (double)NS * (VAR >= 15 ? 3 : 6) * S::TM * S::TN * (8 / nwaves);
    const double floor_cyc = mfma_per_wave * 16.0 * (nwaves / 4);         // the waves of a SIMD share its matrix pipe
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(dcyc, 0, 64 * 8 * 8));
    auto launch = [&](int rp) {
        if constexpr (VAR >= 4 && VAR <= 8) hipLaunchKernelGGL((probe4_kernel<SHAPE, VAR>), dim3(blocks), dim3(256), 0, 0, dW, rp, dout, dcyc);
        else hipLaunchKernelGGL((probe_kernel<SHAPE, VAR>), dim3(blocks), dim3(512), 0, 0, dW, rp, dout, dcyc);
    };
    launch(2);     // warm-up
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int k = 0; k < 3; ++k) {
        CK(hipEventRecord(e0));
        launch(reps);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    std::vector<unsigned long long> c(64 * 8);
    CK(hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0, mx = 0;
    const int nb = blocks < 64 ? blocks : 64;
    for (int b = 0; b < nb; ++b) { unsigned long long m = 0; for (int w = 0; w < 8; ++w) m = c[b * 8 + w] > m ? c[b * 8 + w] : m; mean += (double)m / reps; if ((double)m / reps > mx) mx = (double)m / reps; }
    mean /= nb;
    const double flops = (double)blocks * reps * nwaves * mfma_per_wave * 2.0 * 16 * 16 * 32;
    const double tf = flops / (best * 1e-3) / 1e12;
    static const char* vn[17] = {"old LayB tile-major", "NEW LayQ term-major", "NEW, no alt prio", "old, no alt prio", "NEW 4 waves (1/SIMD) A", "NEW 4 waves (1/SIMD) B",
                                "4w A, no weight loads", "4w A, no frag reloads", "4w A, pure MFMA", "8w no w loads", "8w no frag reloads", "8w pure MFMA",
                                "8w pace s_nop 0", "8w pace s_nop 2", "8w pace s_nop 5", "NEW, two fp16 terms", "(unused)"};
    printf("%-52s %-22s cycles/call %8.0f (max %8.0f) floor %7.0f -> %5.1f %% | launch %7.3f ms %7.1f TFLOP/s bf16 = %5.1f %% of 2516.8\n", S::name(), vn[VAR], mean, mx,
           floor_cyc, 100.0 * floor_cyc / mean, best, tf, 100.0 * tf / 2516.8);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20, blocks = argc > 2 ? atoi(argv[2]) : 1024;
    const size_t wf = 9 * 4 * 3 * 4 * 128 * 4 + 1024;                     // largest split weight block (conv5 128 -> 128), floats
    std::vector<unsigned> hw(wf);
    unsigned st = 12345u;
    for (size_t i = 0; i < wf; ++i) {       // bf16 pairs, magnitudes ~2^-6 .. 2^-3, random signs
        st = st * 1664525u + 1013904223u;
        const unsigned lo = ((st >> 8) & 0x83FFu) | 0x3C00u, hi = ((st >> 16) & 0x83FFu) | 0x3C00u;
        hw[i] = ((lo & 0xBFFFu) - 0x0300u) | (((hi & 0xBFFFu) - 0x0300u) << 16);
    }
    float *dW, *dout; unsigned long long* dcyc;
    CK(hipMalloc(&dW, wf * 4)); CK(hipMalloc(&dout, 64)); CK(hipMalloc(&dcyc, 64 * 8 * 8));
    CK(hipMemcpy(dW, hw.data(), wf * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dcyc, 0, 64 * 8 * 8));
    printf("reps %d, workgroups %d (512 threads)\n", reps, blocks);
#define BOTH(SH) run<SH, 0>(dW, dout, dcyc, reps, blocks, 2.4); run<SH, 1>(dW, dout, dcyc, reps, blocks, 2.4); run<SH, 2>(dW, dout, dcyc, reps, blocks, 2.4);
#define FOUR(SH) run<SH, 4>(dW, dout, dcyc, reps, blocks, 2.4); run<SH, 5>(dW, dout, dcyc, reps, blocks, 2.4);
#define DIAG(SH) run<SH, 6>(dW, dout, dcyc, reps, blocks, 2.4); run<SH, 7>(dW, dout, dcyc, reps, blocks, 2.4); run<SH, 8>(dW, dout, dcyc, reps, blocks, 2.4); \
    run<SH, 9>(dW, dout, dcyc, reps, blocks, 2.4); run<SH, 10>(dW, dout, dcyc, reps, blocks, 2.4); run<SH, 11>(dW, dout, dcyc, reps, blocks, 2.4); \
    run<SH, 12>(dW, dout, dcyc, reps, blocks, 2.4); run<SH, 13>(dW, dout, dcyc, reps, blocks, 2.4); run<SH, 14>(dW, dout, dcyc, reps, blocks, 2.4);
    if (argc > 3 && argv[3][0] == 't') {       // alternative tilings (new loop, no alternating priorities) next to the trunk's
#define ONE(SH) run<SH, 2>(dW, dout, dcyc, reps, blocks, 2.4);
        ONE(0) ONE(7) ONE(1) ONE(12) ONE(2) ONE(8) ONE(3) ONE(9) ONE(4) ONE(10) ONE(11)
        ONE(5) ONE(13) ONE(14) ONE(15) ONE(6) ONE(16)
        return 0;
    }
    if (argc > 3 && argv[3][0] == 'h') {       // two fp16 terms (AFFNET_ARITH_FP32_SPLIT2H) next to three bf16 terms: the trunks' blockings and the 2 x 2 alternatives
#define TWO(SH) run<SH, 2>(dW, dout, dcyc, reps, blocks, 2.4); run<SH, 15>(dW, dout, dcyc, reps, blocks, 2.4);
        TWO(0) TWO(7) TWO(12) TWO(1) TWO(2) TWO(8) TWO(9) TWO(3) TWO(10) TWO(4) TWO(13) TWO(5) TWO(15) TWO(14) TWO(6) TWO(16)
        return 0;
    }
    if (argc > 3) { BOTH(2) FOUR(2) DIAG(2) BOTH(4) FOUR(4) DIAG(4) return 0; }
    BOTH(0) FOUR(0) BOTH(1) FOUR(1) BOTH(2) FOUR(2) BOTH(3) FOUR(3) BOTH(4) FOUR(4) BOTH(5) BOTH(6)
    return 0;
}
