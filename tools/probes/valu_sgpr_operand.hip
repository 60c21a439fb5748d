// VALU rate with SGPR vs VGPR operands (wave64, W waves per SIMD): cycles per instruction
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(1024) void probe(float* out, int iters, f32x2 sv) {
    float v[16];
    for (int k = 0; k < 16; ++k) v[k] = threadIdx.x * 0.37f + k;
    f32x2 sp = sv;
    asm volatile("" : "+s"(sp));
    const float s0 = sp.x;
    float w0 = threadIdx.x * 1e-3f, w1 = w0 + 1.f;
    f32x2 wp = {w0, w1};
    asm volatile("" : "+v"(wp));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int r = (q * 2) & 15;
            if (KIND == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[r]) : "v"(w0), "v"(v[(r + 5) & 15]));
            if (KIND == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[r]) : "s"(s0), "v"(v[(r + 5) & 15]));
            if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(f32x2*)&v[r]) : "v"(*(f32x2*)&v[(r + 4) & 14]), "v"(*(f32x2*)&v[(r + 8) & 14]));
            if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(*(f32x2*)&v[r]) : "v"(*(f32x2*)&v[(r + 4) & 14]), "s"(sp));
            if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(*(f32x2*)&v[r]) : "v"(*(f32x2*)&v[(r + 4) & 14]), "v"(wp));
        }
    }
    float s = 0;
    for (int k = 0; k < 16; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + w1 + sp.y + wp.y;
}
template <int KIND>
static void run(int wps, float* d, const char* name) {
    const int iters = 20000, blocks = wps > 4 ? 512 : 256, threads = 256 * (wps > 4 ? wps / 2 : wps);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<KIND><<<blocks, threads>>>(d, 100, (f32x2){1.0f, 1.5f}); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); probe<KIND><<<blocks, threads>>>(d, iters, (f32x2){1.0f, 1.5f}); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s waves/SIMD %d: %.2f ns per instruction per wave-slot (x2.4 = %.2f cycles)\n", name, wps, ms * 1e6 / iters / 32 / wps, ms * 1e6 / iters / 32 / wps * 2.4);
}
int main() {
    float* d; (void)hipMalloc(&d, 512 * 1024 * 4);
    for (int w : {1, 2, 4, 6, 8}) {
        run<0>(w, d, "v_fmac_f32 v, v, v");
        run<1>(w, d, "v_fmac_f32 v, s, v");
        run<2>(w, d, "v_pk_fma_f32 v, v, v");
        run<3>(w, d, "v_pk_fma_f32 v, v, s (op_sel_hi broadcast)");
        run<4>(w, d, "v_pk_fma_f32 v, v, v (op_sel_hi broadcast)");
    }
    return 0;
}
