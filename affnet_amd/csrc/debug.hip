// Counter-calibration kernels (include/affnet_hip_probes.h): known-byte-count streaming reads / writes / tile loads, so
// that rocprofv3's FETCH_SIZE / WRITE_SIZE can be turned into bytes for the access widths this library actually uses
// (MI355X_MICROARCH.md, HBM section: only 16 B/lane streaming reads are calibrated there).  Not part of the product path.
#include "common.h"

#ifdef AFFNET_PROBES   // libaffnet_hip_probes.so only (include/affnet_hip_probes.h)

template <typename T>
__global__ __launch_bounds__(256) void stream_read_kernel(const T* __restrict__ src, float* __restrict__ dst, size_t n_items) {
    // grid-stride, fully coalesced: lane i of a wave reads item base + i
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_items; i += (size_t)gridDim.x * 256) {
        const T v = src[i];
        const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
        for (int k = 0; k < (int)(sizeof(T) / 4); ++k) acc += f[k];
    }
    if (acc == 123.456f) dst[blockIdx.x] = acc;       // never true for the calibration pattern: keeps the loads alive
    if (threadIdx.x == 0 && blockIdx.x == 0) dst[0] = acc;
}

template <typename T>
__global__ __launch_bounds__(256) void stream_write_kernel(T* __restrict__ dst, size_t n_items) {
    T v;
    float* f = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 4); ++k) f[k] = (float)(threadIdx.x + k);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_items; i += (size_t)gridDim.x * 256) dst[i] = v;
}

// 64 x 64 output tile + `halo` apron loaded with 4-byte loads (one float per thread per load), the access pattern of
// blur2d_kernel / hessian_nms_kernel's tile loaders.
__global__ __launch_bounds__(256) void tile_read_kernel(const float* __restrict__ img, float* __restrict__ dst, int h, int w, int halo) {
    const int LW = 64 + 2 * halo, n_el = LW * LW;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 64;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_el; i += 256) {
        const int ty = i / LW, tx = i - ty * LW;
        int gy = y0 + ty - halo, gx = x0 + tx - halo;
        gy = gy < 0 ? 0 : (gy >= h ? h - 1 : gy);
        gx = gx < 0 ? 0 : (gx >= w ? w - 1 : gx);
        acc += img[(size_t)gy * w + gx];
    }
    if (acc == 123.456f) dst[blockIdx.x] = acc;
}

struct f32x2s { float a, b; };

extern "C" int affnet_debug_stream(const void* d_src, void* d_dst, size_t n_bytes, int width, int mode, int halo, void* stream) {
    if (!d_dst || n_bytes == 0 || (n_bytes & 65535) != 0) return AFFNET_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = 256 * 16;     // 16 workgroups per CU
    if (mode == 0) {
        if (!d_src) return AFFNET_ERR_INVALID;
        if (width == 4) hipLaunchKernelGGL(stream_read_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)d_src, (float*)d_dst, n_bytes / 4);
        else if (width == 8) hipLaunchKernelGGL(stream_read_kernel<f32x2s>, dim3(blocks), dim3(256), 0, st, (const f32x2s*)d_src, (float*)d_dst, n_bytes / 8);
        else if (width == 16) hipLaunchKernelGGL(stream_read_kernel<float4>, dim3(blocks), dim3(256), 0, st, (const float4*)d_src, (float*)d_dst, n_bytes / 16);
        else return AFFNET_ERR_INVALID;
    } else if (mode == 1) {
        if (width == 4) hipLaunchKernelGGL(stream_write_kernel<float>, dim3(blocks), dim3(256), 0, st, (float*)d_dst, n_bytes / 4);
        else if (width == 8) hipLaunchKernelGGL(stream_write_kernel<f32x2s>, dim3(blocks), dim3(256), 0, st, (f32x2s*)d_dst, n_bytes / 8);
        else if (width == 16) hipLaunchKernelGGL(stream_write_kernel<float4>, dim3(blocks), dim3(256), 0, st, (float4*)d_dst, n_bytes / 16);
        else return AFFNET_ERR_INVALID;
    } else if (mode == 2) {
        if (!d_src || halo < 0 || halo > 16) return AFFNET_ERR_INVALID;
        const int w = 4096, h = (int)(n_bytes / 4 / w);
        hipLaunchKernelGGL(tile_read_kernel, dim3(w / 64, (h + 63) / 64), dim3(256), 0, st, (const float*)d_src, (float*)d_dst, h, w, halo);
    } else {
        return AFFNET_ERR_INVALID;
    }
    return hipGetLastError() == hipSuccess ? AFFNET_OK : AFFNET_ERR_HIP;
}
#endif  // AFFNET_PROBES
