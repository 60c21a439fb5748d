// Shared host/device declarations of libaffnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/affnet_hip.h"
#include "../../include/affnet_hip_debug.h"
#ifdef AFFNET_PROBES
#include "../../include/affnet_hip_probes.h"
#endif

#define AFF_WAVE 64

// Counter block kept at the start of the workspace "lists" area (int32 each).
enum {
    CNT_RAW0 = 0,                            // CNT_RAW0 + o : raw maxima of octave o
    CNT_CAND = AFFNET_MAX_OCTAVES,           // accepted candidates (all octaves)
    CNT_OVERFLOW,                            // != 0 : some fixed-capacity list overflowed
    CNT_SEL,                                 // rows selected by the global top-k
    CNT_EQ_TAKEN,                            // ties at the threshold already taken
    CNT_SEL_MODE,                            // 1 = top-k (descending response), 0 = keep all (key order)
    CNT_SEL_THRESH,                          // order-preserving uint key of the C-th largest response
    CNT_SEL_NEED_EQ,                         // how many keys == threshold to take
    CNT_DET,                                 // rows emitted by the detector
    CNT_SHAPED,                              // rows after the shape filter
    CNT_SURVIVED,                            // survivors of the shape filter before top-N
    CNT_CAND2,                               // OnePassSIR: candidates that passed the per-level top-k and the boundary test
    CNT_AFF_EVAL,                            // candidates the shape CNN was actually evaluated on (lazy second pass, pipeline.hip)
    CNT_SURVIVED1,                           // survivors of the first (lazy) shape pass, frozen before the second pass starts
    CNT_SEL_EQ_TOTAL,                        // candidates whose response equals the top-k threshold (ties: taken in key order)
    CNT_SEL_TIE_LO, CNT_SEL_TIE_HI,          // ties at the threshold are taken up to this (octave, level, pixel) key (64 bits, two halves)
    CNT_POS0 = AFFNET_MAX_OCTAVES + 24,      // CNT_POS0 + (level-1)*AFFNET_MAX_OCTAVES + o : positive maxima of (octave, level)
    CNT_HYP0 = AFFNET_MAX_OCTAVES + 24 + (AFFNET_MAX_LEVELS - 2) * AFFNET_MAX_OCTAVES,   // CNT_HYP0 + 8 * o + k: positives of (octave o, level,
                                             // hypothesis) k = 0: level 1; 1 + h1: level 2 given level 1 applied (h1); 3 + h1 + 2 h2: level 3
    CNT_TOTAL = AFFNET_MAX_OCTAVES + 24 + (AFFNET_MAX_LEVELS - 2) * AFFNET_MAX_OCTAVES + 8 * AFFNET_MAX_OCTAVES
};

#define SEL_HIST_BINS 2048      // first digit (11 bits) of the global top-k's radix select, histogrammed by many workgroups

struct RawMax {            // one 3-D local maximum found by hessian_nms_kernel
    int32_t pix;           // flat pixel index y*w+x in the octave
    int32_t lvl;           // detection level 1..nLevels
    float val;             // NMS'ed (border-zeroed) response
    float s, y, x;         // normalised centroid scale / row / column
    float prev[2];         // NMS'ed responses of the SAME pixel at detection levels lvl-1 and lvl-2 when it is a raw maximum there too (else 0):
                           // what the octaveMap replay of this pixel needs (detect.hip, resolve_*_kernel); used for <= 3 detection levels
};

struct OctaveGeom {
    int32_t h, w;
    int64_t pyr_off;       // float offset of level 0 inside the pyramid area
    int64_t map_off;       // byte offset of the uint8 octaveMap
    int64_t raw_off;       // RawMax offset of this octave's raw list
    int32_t raw_cap;
};

struct affnet_ctx {
    int device = 0;
    affnet_config cfg;
    // Image batch: every workspace area holds B images back to back ([image][...]); kernels take the image
    // index from blockIdx.y / .z and offset their pointers by the per-image strides below.
    int B = 1;
    size_t pyr_stride = 0;             // floats between the pyramids of consecutive images
    size_t map_stride = 0;             // bytes between octaveMaps
    size_t raw_stride = 0;             // RawMax entries between raw lists
    std::string err;
    OctaveGeom oct[AFFNET_MAX_OCTAVES];
    // workspace layout (byte offsets from the workspace base)
    size_t off_pyr = 0, off_map = 0, off_raw = 0, off_cnt = 0, off_hist = 0, off_cand = 0, off_sel = 0, off_stage = 0;
    // OnePassSIR extras (cfg.onepass != 0): dense affine-shape maps (4, h_o, w_o) per octave, the dense net's scratch, a second
    // candidate list and the per-(octave, level) top-k table
    size_t off_affmap = 0, off_dense = 0, off_cand2 = 0, off_lvltab = 0;
    size_t aff_stride = 0;             // floats between the affine maps of consecutive images
    size_t aff_off[AFFNET_MAX_OCTAVES] = {0};   // float offset of octave o's (4, h, w) map inside one image's block
    size_t dense_stride = 0;           // floats of dense-net scratch per image
    float* affmap = nullptr; float* dense = nullptr;
    float* cand2_resp = nullptr; float* cand2_syx = nullptr; int32_t* cand2_ids = nullptr;
    int32_t* lvltab = nullptr;
    size_t ws_bytes = 0;
    size_t pyr_floats = 0, map_bytes = 0, raw_total = 0, cand_cap = 0;
    int cap_pre = 0, cap_final = 0;
    char* ws = nullptr;
    // convenience pointers into the workspace (valid after bind)
    float* pyr = nullptr;
    uint8_t* omap = nullptr;
    RawMax* raw = nullptr;
    int32_t* cnt = nullptr;
    uint32_t* sel_hist = nullptr;        // B x SEL_HIST_BINS: histogram of the top 11 key bits of the candidate responses
    float* cand_resp = nullptr; float* cand_syx = nullptr; int32_t* cand_ids = nullptr;
    float* sel_resp = nullptr; float* sel_syx = nullptr; int32_t* sel_ids = nullptr;
    // pipeline stage buffers
    float* st_det_resp = nullptr; float* st_det_lafs = nullptr; int32_t* st_det_ids = nullptr;
    int32_t* st_det_count = nullptr;     // B contiguous detector row counts (kernels index count[image])
    float* st_A = nullptr; float* st_key = nullptr; int32_t* st_good = nullptr;
    float* st_A2 = nullptr; float* st_lafs_iter = nullptr;   // AffNet iterations > 1: current A, re-extraction LAFs
    float* st_R = nullptr; float* st_lafs_norm = nullptr; int32_t* st_lvl_ids = nullptr;
    float* st_hard_scratch = nullptr;
    float* st_lafs_shaped = nullptr;
    int32_t* st_rank = nullptr;          // partial ranks / positions of the two selection stages (cap_pre ints)
    // stage profiling (HIP events on the caller's stream)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;   // ring: PROF_RING calls x PROF_EVENTS events
    int prof_calls = 0;
    // tuning aid (include/affnet_hip_debug.h): s_memtime stamp buffer of THIS context's CNN launches, or NULL
    unsigned long long* dbg_time = nullptr;
    int arith = AFFNET_ARITH_FP32_MFMA;   // arithmetic of the CNN contractions (cfg.arith / affnet_set_arith): exact fp32 MFMA or fp32 = 3 x bf16 split operands
    int split3_variant = 0;            // tuning aid (affnet_debug_split3_variant): bit 0 = alternating wave priorities in the split HardNet loops (round 3's
                                       // tile-major loops gained 2.5 % from it, the term-major loops of round 4 lose 1 %: off)
    // the whole path captured as one HIP graph (affnet_graph_capture_extract): one launch instead of ~45 for latency-bound callers
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    ~affnet_ctx() {
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        if (graph) (void)hipGraphDestroy(graph);
        for (auto& e : prof_ev) (void)hipEventDestroy(e);
    }
};

// Every entry point that launches work makes the context's device current for the duration of the call and restores
// the caller's device afterwards (a context is bound to ONE device, affnet_ctx_create; the caller's current device
// may be another one in a multi-GPU process).  Tolerates a host without a GPU (CPU-side argument checks still run).
struct AffDeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit AffDeviceGuard(const affnet_ctx* ctx) {
        if (!ctx) return;
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; return; }
        if (prev != ctx->device) switched = hipSetDevice(ctx->device) == hipSuccess;
    }
    ~AffDeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    AffDeviceGuard(const AffDeviceGuard&) = delete;
    AffDeviceGuard& operator=(const AffDeviceGuard&) = delete;
};
#define AFF_DEVICE(ctx) AffDeviceGuard aff_device_guard_(ctx)

#define PROF_RING 256
#define PROF_EVENTS (AFFNET_PROFILE_STAGES + 2)   // 9 stage boundaries + end-of-detector (index 9)
// pipeline.hip / cnn32.hip: record stage boundary `idx` of the current call (no-op when disabled)
void aff_prof_mark(affnet_ctx* ctx, int idx, hipStream_t st);

int aff_fail(affnet_ctx* ctx, int code, const char* fmt, ...);

#define AFF_HIP(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return aff_fail(ctx, AFFNET_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                                   \
    } while (0)

#define AFF_LAUNCH_CHECK(ctx)                                                                      \
    do {                                                                                           \
        hipError_t e_ = hipGetLastError();                                                         \
        if (e_ != hipSuccess)                                                                      \
            return aff_fail(ctx, AFFNET_ERR_HIP, "kernel launch failed: %s (%s:%d)",                \
                            hipGetErrorString(e_), __FILE__, __LINE__);                            \
    } while (0)

// Device-side fills and copies as plain kernels of this library (context.hip) instead of hipMemsetAsync / hipMemcpyAsync: the
// runtime's blit path shares per-queue state with captured graph nodes (replaying a captured graph after eager null-stream
// memsets faulted on ROCm 7.2), and a kernel of our own is also what a stream capture records most cheaply.
int aff_zero_async(affnet_ctx* ctx, void* dst, size_t bytes, hipStream_t st);
// up to 8 fills in one launch
struct AffZeroSegs {
    unsigned char* p[8];
    size_t bytes[8];
    int n = 0;
    bool overflow = false;   // more than 8 segments were added: aff_zero_multi_async refuses the launch instead of leaving an area uncleared
    void add(void* ptr, size_t nbytes) {
        if (!ptr || !nbytes) return;
        if (n >= 8) { overflow = true; return; }
        p[n] = (unsigned char*)ptr; bytes[n] = nbytes; ++n;
    }
};
int aff_zero_multi_async(affnet_ctx* ctx, const AffZeroSegs& z, hipStream_t st);
int aff_copy_async(affnet_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t st);
// rows x width_bytes, row r at dst + r * dpitch / src + r * spitch (all multiples of 4 bytes)
int aff_copy2d_async(affnet_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, size_t rows, hipStream_t st);

static inline size_t aff_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int aff_cdiv(int a, int b) { return (a + b - 1) / b; }

// Level pointers of the pyramid of image 0 in the workspace (+ img_stride floats per further image of the batch).
struct PyrTable {
    const float* lvl[AFFNET_MAX_OCTAVES][AFFNET_MAX_LEVELS];
    int h[AFFNET_MAX_OCTAVES], w[AFFNET_MAX_OCTAVES];
    int n_octaves, n_levels;
    size_t img_stride;
};
void aff_fill_pyr_table(const affnet_ctx* ctx, PyrTable* t);

// ---- sampler math shared by sampler.hip and cnn32.hip -------------------------------------------
// Host: fills base[ps] = (linspace(-1,1,ps) * (ps-1)) / ps exactly as torch does on CPU
// (linspace = fma(step, i, start) / fma(-step, ps-1-i, end); verified bit-for-bit in
// tests/test_host_mirror.py).
void aff_base_grid(int ps, float* base);

// LDS-staged footprint of one patch (north_star: "coalesced HBM reads and LDS-staged image tiles"): the affine frame of a PS x PS
// patch covers an axis-aligned box of the level image; when that box is small its rows are loaded once with coalesced row-segment
// loads (zeros outside the image = grid_sample's zero padding) and the four bilinear taps of every sample come from LDS.  The
// arithmetic of a sample is untouched (same weights, same fmaf chain), so staged and direct sampling are bit-identical.
struct AffTile {
    int x0, y0, tw, th;      // box origin in level pixels (may be negative), width / height; tw == 0: not staged
};

// Box that contains every tap of the patch (1-px margin for the fp32 round trip of the coordinates); staged iff it fits `cap` floats
// and is small enough that loading it costs fewer requests than the 4 * ps * ps gathers it replaces.
__device__ __forceinline__ AffTile aff_tile_box(float t00, float t01, float t02, float t10, float t11, float t12, int ps, int cap) {
    const float um = (float)(ps - 1) / (float)ps;
    const float hx = (fabsf(t00) + fabsf(t01)) * um, hy = (fabsf(t10) + fabsf(t11)) * um;
    const float cx = t02 - 0.5f, cy = t12 - 0.5f;
    AffTile t;
    t.tw = 0; t.th = 0; t.x0 = 0; t.y0 = 0;
    if (!(hx < 4096.0f && hy < 4096.0f && fabsf(cx) < 1.0e6f && fabsf(cy) < 1.0e6f)) return t;     // also rejects NaN / inf frames
    const int x0 = (int)floorf(cx - hx) - 1, x1 = (int)floorf(cx + hx) + 2;
    const int y0 = (int)floorf(cy - hy) - 1, y1 = (int)floorf(cy + hy) + 2;
    const int tw = x1 - x0 + 1, th = y1 - y0 + 1;
    if (tw * th > cap || tw * th > 2 * ps * ps) return t;
    t.x0 = x0; t.y0 = y0; t.tw = tw; t.th = th;
    return t;
}

// Cooperative load of the box into LDS (NTHR threads; the caller synchronises afterwards).
template <int NTHR>
__device__ __forceinline__ void aff_tile_load(float* tile, const AffTile t, const float* __restrict__ img, int h, int w, int tid) {
    const int n = t.tw * t.th;
    for (int i = tid; i < n; i += NTHR) {
        const int ty = i / t.tw, tx = i - ty * t.tw;
        const int gy = t.y0 + ty, gx = t.x0 + tx;
        tile[i] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? img[(size_t)gy * w + gx] : 0.0f;
    }
}

// Device: one bilinear sample.  Follows LAF.py:313-324 + F.affine_grid + F.grid_sample
// (align_corners=False, zeros padding) operation by operation in fp32:
//   theta = LAF * [[m,m,w],[m,m,h]];  g = fma(1,t02, fma(v,t01, u*t00));
//   gn = 2*g/size - 1;  i = fma(gn+1, size/2, -0.5);  out = fma chain nw,ne,sw,se.
__device__ __forceinline__ float aff_sample_bilinear(const float* __restrict__ img, int h, int w,
                                                     float t00, float t01, float t02, float t10,
                                                     float t11, float t12, float u, float v) {
    float gx = fmaf(1.0f, t02, fmaf(v, t01, u * t00));
    float gy = fmaf(1.0f, t12, fmaf(v, t11, u * t10));
    const float fw = (float)w, fh = (float)h;
    gx = 2.0f * gx / fw - 1.0f;
    gy = 2.0f * gy / fh - 1.0f;
    const float ix = fmaf(gx + 1.0f, fw * 0.5f, -0.5f);
    const float iy = fmaf(gy + 1.0f, fh * 0.5f, -0.5f);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
    const float wx1 = ix - x0f, wx0 = x1f - ix, wy1 = iy - y0f, wy0 = y1f - iy;
    const float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
    // float -> int after range clamping so that huge coordinates cannot overflow
    const float cx0 = fminf(fmaxf(x0f, -2.0f), fw + 1.0f), cy0 = fminf(fmaxf(y0f, -2.0f), fh + 1.0f);
    const int x0 = (int)cx0, y0 = (int)cy0, x1 = x0 + 1, y1 = y0 + 1;
    const bool xin0 = (x0 >= 0) & (x0 < w), xin1 = (x1 >= 0) & (x1 < w);
    const bool yin0 = (y0 >= 0) & (y0 < h), yin1 = (y1 >= 0) & (y1 < h);
    const float vnw = (xin0 & yin0) ? img[(size_t)y0 * w + x0] : 0.0f;
    const float vne = (xin1 & yin0) ? img[(size_t)y0 * w + x1] : 0.0f;
    const float vsw = (xin0 & yin1) ? img[(size_t)y1 * w + x0] : 0.0f;
    const float vse = (xin1 & yin1) ? img[(size_t)y1 * w + x1] : 0.0f;
    return fmaf(vse, se, fmaf(vsw, sw, fmaf(vne, ne, vnw * nw)));
}

// Same sample with the four taps taken from a staged box (zeros are already in the tile for pixels outside the image).  A tap
// outside the box (cannot happen for a box from aff_tile_box; guarded anyway) falls back to the predicated global load.
__device__ __forceinline__ float aff_sample_bilinear_tile(const float* tile, const AffTile tl, const float* __restrict__ img, int h, int w,
                                                          float t00, float t01, float t02, float t10, float t11, float t12, float u, float v) {
    float gx = fmaf(1.0f, t02, fmaf(v, t01, u * t00));
    float gy = fmaf(1.0f, t12, fmaf(v, t11, u * t10));
    const float fw = (float)w, fh = (float)h;
    gx = 2.0f * gx / fw - 1.0f;
    gy = 2.0f * gy / fh - 1.0f;
    const float ix = fmaf(gx + 1.0f, fw * 0.5f, -0.5f);
    const float iy = fmaf(gy + 1.0f, fh * 0.5f, -0.5f);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
    const float wx1 = ix - x0f, wx0 = x1f - ix, wy1 = iy - y0f, wy0 = y1f - iy;
    const float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
    const float cx0 = fminf(fmaxf(x0f, -2.0f), fw + 1.0f), cy0 = fminf(fmaxf(y0f, -2.0f), fh + 1.0f);
    const int x0 = (int)cx0, y0 = (int)cy0;
    const int lx = x0 - tl.x0, ly = y0 - tl.y0;
    float vnw, vne, vsw, vse;
    if (lx >= 0 && ly >= 0 && lx + 1 < tl.tw && ly + 1 < tl.th) {
        const float* p = tile + ly * tl.tw + lx;
        vnw = p[0]; vne = p[1]; vsw = p[tl.tw]; vse = p[tl.tw + 1];
    } else {
        const int x1 = x0 + 1, y1 = y0 + 1;
        const bool xin0 = (x0 >= 0) & (x0 < w), xin1 = (x1 >= 0) & (x1 < w);
        const bool yin0 = (y0 >= 0) & (y0 < h), yin1 = (y1 >= 0) & (y1 < h);
        vnw = (xin0 & yin0) ? img[(size_t)y0 * w + x0] : 0.0f;
        vne = (xin1 & yin0) ? img[(size_t)y0 * w + x1] : 0.0f;
        vsw = (xin0 & yin1) ? img[(size_t)y1 * w + x0] : 0.0f;
        vse = (xin1 & yin1) ? img[(size_t)y1 * w + x1] : 0.0f;
    }
    return fmaf(vse, se, fmaf(vsw, sw, fmaf(vne, ne, vnw * nw)));
}
