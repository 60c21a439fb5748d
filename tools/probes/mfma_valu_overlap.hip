// Does VALU work hide under bf16 MFMAs on gfx950?  One workgroup per CU, W waves per SIMD; every wave runs ITER x { NM MFMAs on 4
// independent accumulators, NV VALU instructions of one kind on independent registers }.  Prints cycles per iteration (s_memtime is
// not used: wall clock via hipEvents and the measured shader clock are enough for ratios).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NM, int NV, int KIND>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(threadIdx.x * 0.001f + k); b[k] = (__bf16)(1.0f + k * 0.01f); }
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 0.37f + k;
    unsigned c0;
    asm volatile("s_mov_b32 %0, 0xbf80" : "=s"(c0));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV / (NM > 0 ? NM : 1); ++q) {
                const int r = (m * (NV / (NM > 0 ? NM : 1)) + q) & 7;
                if (KIND == 0) asm volatile("v_fmac_f32 %0, %0, %0" : "+v"(v[r]));
                if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(v[r]));
                if (KIND == 2) asm volatile("v_dot2c_f32_bf16 %0, %1, %0" : "+v"(v[r]) : "s"(c0));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(*(f32x2*)&v[r & 6]));
                if (KIND == 4) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(v[r]));
            }
        }
        if (NM == 0) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int r = q & 7;
                if (KIND == 0) asm volatile("v_fmac_f32 %0, %0, %0" : "+v"(v[r]));
                if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(v[r]));
                if (KIND == 2) asm volatile("v_dot2c_f32_bf16 %0, %1, %0" : "+v"(v[r]) : "s"(c0));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(*(f32x2*)&v[r & 6]));
                if (KIND == 4) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(v[r]));
            }
        }
    }
    float s = 0;
    for (int k = 0; k < 4; ++k) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
    for (int k = 0; k < 8; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, int KIND>
static double run(int waves_per_simd, float* d, const char* name) {
    const int iters = 20000, threads = 64 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<NM, NV, KIND><<<256, threads>>>(d, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<NM, NV, KIND><<<256, threads>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns_it = ms * 1e6 / iters;
    printf("%-34s waves/SIMD %d: %7.1f ns / iteration  (%5.1f cycles @2.4 GHz)\n", name, waves_per_simd, ns_it, ns_it * 2.4);
    return ns_it;
}

int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 512 * 4);
    for (int w = 1; w <= 2; ++w) {
        run<12, 0, 0>(w, d, "12 MFMA");
        run<0, 36, 0>(w, d, "36 v_fmac_f32");
        run<12, 36, 0>(w, d, "12 MFMA + 36 v_fmac_f32");
        run<0, 36, 1>(w, d, "36 v_cvt_pk_bf16_f32");
        run<12, 36, 1>(w, d, "12 MFMA + 36 v_cvt_pk_bf16_f32");
        run<0, 36, 2>(w, d, "36 v_dot2c_f32_bf16");
        run<12, 36, 2>(w, d, "12 MFMA + 36 v_dot2c_f32_bf16");
        run<0, 36, 3>(w, d, "36 v_pk_add_f32");
        run<12, 36, 3>(w, d, "12 MFMA + 36 v_pk_add_f32");
        run<0, 36, 4>(w, d, "36 v_lshlrev_b32");
        run<12, 36, 4>(w, d, "12 MFMA + 36 v_lshlrev_b32");
        run<12, 12, 1>(w, d, "12 MFMA + 12 v_cvt_pk_bf16_f32");
        run<12, 24, 2>(w, d, "12 MFMA + 24 v_dot2c_f32_bf16");
    }
    return 0;
}
