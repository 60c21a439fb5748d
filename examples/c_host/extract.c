/*
 * extract.c - a plain C99 host program that drives libaffnet_hip.so end to end: no Python, no torch in the process.
 *
 *   extract IMAGE.f32 H W N AFFNET.afnw ORINET.afnw HARDNET.afnw OUT_PREFIX [arith]
 *
 * What the reference's hesaffnet.py:35-60 + get_geometry_and_descriptors (train_OriNet_test_on_graffity.py:293-298) do for one image:
 * ScaleSpaceAffinePatchExtractor(mrSize=5.192, num_features=N, border=5, num_Baum_iters=1, AffNet, OriNet)(img, do_ori=True), HardNet
 * descriptors of the returned frames, Oxford ellipse text file.
 *
 *   IMAGE.f32    H x W float32, row-major, 0..255 (hesaffnet.py:35-39: channel mean of the RGB image)
 *   *.afnw       flat weight files (affnet_amd.engine.save_flat_weights / tools/export_weights.py):
 *                char magic[8] = "AFNW0001"; int32 kind (AFFNET_NET_*); int32 n_floats; then the float32 tensors of the state dict in the
 *                order conv weights features.{0,3,6,9,12,15}, BN running_mean features.{1,4,7,10,13,16}, BN running_var (same), head
 *                features.19.weight, [features.19.bias], [features.20.running_mean, features.20.running_var] (HardNet)
 *   OUT_PREFIX   writes OUT_PREFIX.txt (Oxford format: "1.0", count, rows `x y a b c`), OUT_PREFIX.lafs.f32 (count x 2 x 3, px),
 *                OUT_PREFIX.resp.f32, OUT_PREFIX.ids.i32 (count x 3: octave, level - 1, pixel), OUT_PREFIX.desc.f32 (count x 128)
 *   arith        0 (default, exact fp32 MFMA), 1 (fp32_split3) or 2 (fp32_split2h): AFFNET_ARITH_*
 *
 * Build (examples/c_host/build.sh): gcc -std=c99 extract.c -I include -I /opt/rocm/include -L affnet_amd -laffnet_hip -lamdhip64
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "affnet_hip.h"

#define DIE(...) do { fprintf(stderr, "extract: " __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } while (0)
#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) DIE("%s: %s", #call, hipGetErrorString(e_)); } while (0)
#define AFF(ctx, call) do { int rc_ = (call); if (rc_ != AFFNET_OK) DIE("%s failed (%d): %s", #call, rc_, (ctx) ? affnet_last_error(ctx) : ""); } while (0)

static void* read_file(const char* path, size_t expect_bytes, size_t* got_bytes) {
    FILE* f = fopen(path, "rb");
    if (!f) DIE("cannot open %s", path);
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (expect_bytes && (size_t)n != expect_bytes) DIE("%s: %ld bytes, expected %zu", path, n, expect_bytes);
    void* p = malloc((size_t)n);
    if (!p || fread(p, 1, (size_t)n, f) != (size_t)n) DIE("cannot read %s", path);
    fclose(f);
    if (got_bytes) *got_bytes = (size_t)n;
    return p;
}

static void write_file(const char* prefix, const char* suffix, const void* p, size_t bytes) {
    char path[4096];
    snprintf(path, sizeof(path), "%s%s", prefix, suffix);
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(p, 1, bytes, f) != bytes) DIE("cannot write %s", path);
    fclose(f);
}

/* flat weight file -> affnet_cnn32_pack_weights -> device blob */
static float* load_net(const char* path, int kind) {
    static const int aff_w[6] = {16, 16, 32, 32, 64, 64}, hard_w[6] = {32, 32, 64, 64, 128, 128};
    const int* width = kind == AFFNET_NET_HARDNET ? hard_w : aff_w;
    const int head_out = kind == AFFNET_NET_AFFNET ? 3 : kind == AFFNET_NET_ORINET ? 2 : 128;
    size_t bytes = 0;
    char* raw = (char*)read_file(path, 0, &bytes);
    int32_t hdr[2];
    if (bytes < 16 || memcmp(raw, "AFNW0001", 8) != 0) DIE("%s is not an AFNW0001 weight file", path);
    memcpy(hdr, raw + 8, 8);
    if (hdr[0] != kind) DIE("%s holds network kind %d, expected %d", path, hdr[0], kind);
    const float* p = (const float*)(raw + 16);
    const float *conv[6], *mean[6], *var[6];
    size_t used = 0;
    int cin = 1;
    for (int i = 0; i < 6; ++i) { conv[i] = p + used; used += (size_t)width[i] * cin * 9; cin = width[i]; }
    for (int i = 0; i < 6; ++i) { mean[i] = p + used; used += width[i]; }
    for (int i = 0; i < 6; ++i) { var[i] = p + used; used += width[i]; }
    const float* head_w = p + used; used += (size_t)head_out * width[5] * 64;
    const float *head_b = NULL, *hbm = NULL, *hbv = NULL;
    if (kind == AFFNET_NET_HARDNET) { hbm = p + used; used += 128; hbv = p + used; used += 128; }
    else { head_b = p + used; used += head_out; }
    if ((size_t)hdr[1] != used || bytes != 16 + used * 4) DIE("%s: %d floats in the header, %zu in the file, %zu expected", path, hdr[1], (bytes - 16) / 4, used);
    const size_t n = affnet_cnn32_packed_floats(kind);
    float* packed = (float*)malloc(n * sizeof(float));
    AFF((affnet_ctx*)NULL, affnet_cnn32_pack_weights(kind, conv, mean, var, head_w, head_b, hbm, hbv, packed));
    float* d = NULL;
    HIP(hipMalloc((void**)&d, n * sizeof(float)));
    HIP(hipMemcpy(d, packed, n * sizeof(float), hipMemcpyHostToDevice));
    free(packed);
    free(raw);
    return d;
}

int main(int argc, char** argv) {
    if (argc != 9 && argc != 10) {
        fprintf(stderr, "usage: extract IMAGE.f32 H W N AFFNET.afnw ORINET.afnw HARDNET.afnw OUT_PREFIX [arith 0|1|2]\n");
        return 1;                                   /* hesaffnet.py:21-23: wrong input format -> exit code 1 */
    }
    const int H = atoi(argv[2]), W = atoi(argv[3]), N = atoi(argv[4]);
    const int arith = argc == 10 ? atoi(argv[9]) : AFFNET_ARITH_FP32_MFMA;
    if (H < 1 || W < 1 || N < 1) DIE("H, W, N must be positive");
    float* img = (float*)read_file(argv[1], (size_t)H * W * sizeof(float), NULL);
    HIP(hipSetDevice(0));
    hipStream_t st;
    HIP(hipStreamCreate(&st));

    /* the extractor's configuration: ScaleSpaceAffinePatchExtractor(mrSize = 5.192, num_features = N, border = 5, num_Baum_iters = 1)
     * with the class defaults nlevels = 3, init_sigma = 1.6, th = None (SparseImgRepresenter.py:15-24) */
    affnet_config* cfg = (affnet_config*)malloc(sizeof(affnet_config));
    AFF((affnet_ctx*)NULL, affnet_config_fill(cfg, H, W, 3, 1.6, 5, 5.192, 0.0, N, (int)(1.5 * N), 1, 1));
    cfg->arith = arith;
    affnet_ctx* ctx = NULL;
    AFF(ctx, affnet_ctx_create(&ctx, 0, cfg));
    const size_t ws_bytes = affnet_workspace_bytes(ctx);
    void* d_ws = NULL;
    HIP(hipMalloc(&d_ws, ws_bytes));
    AFF(ctx, affnet_bind_workspace(ctx, d_ws, ws_bytes));

    affnet_nets nets;
    memset(&nets, 0, sizeof(nets));
    nets.d_affnet = load_net(argv[5], AFFNET_NET_AFFNET);
    nets.d_orinet = load_net(argv[6], AFFNET_NET_ORINET);
    nets.d_hardnet = load_net(argv[7], AFFNET_NET_HARDNET);

    const int cap = affnet_capacity_final(ctx);
    float *d_img, *d_lafs, *d_resp, *d_desc, *d_ell;
    int32_t *d_ids, *d_count;
    HIP(hipMalloc((void**)&d_img, (size_t)H * W * 4));
    HIP(hipMalloc((void**)&d_lafs, (size_t)cap * 6 * 4));
    HIP(hipMalloc((void**)&d_resp, (size_t)cap * 4));
    HIP(hipMalloc((void**)&d_ids, (size_t)cap * 3 * 4));
    HIP(hipMalloc((void**)&d_desc, (size_t)cap * 128 * 4));
    HIP(hipMalloc((void**)&d_ell, (size_t)cap * 5 * 4));
    HIP(hipMalloc((void**)&d_count, 4));
    HIP(hipMemsetAsync(d_count, 0, 4, st));
    HIP(hipMemcpyAsync(d_img, img, (size_t)H * W * 4, hipMemcpyHostToDevice, st));

    AFF(ctx, affnet_extract_features(ctx, &nets, d_img, 1, d_lafs, d_resp, d_ids, d_desc, d_count, st));
    AFF(ctx, affnet_lafs_to_ellipses(ctx, d_lafs, d_count, cap, d_ell, st));            /* LAF.py:35-51 */
    int32_t counts[4];
    AFF(ctx, affnet_read_counts(ctx, counts, st));      /* the one read-back: overflow -> error, no detections -> AFFNET_ERR_EMPTY */
    const int n = counts[1];

    float* lafs = (float*)malloc((size_t)cap * 6 * 4);
    float* resp = (float*)malloc((size_t)cap * 4);
    float* desc = (float*)malloc((size_t)cap * 128 * 4);
    float* ell = (float*)malloc((size_t)cap * 5 * 4);
    int32_t* ids = (int32_t*)malloc((size_t)cap * 3 * 4);
    HIP(hipMemcpy(lafs, d_lafs, (size_t)n * 6 * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(resp, d_resp, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(desc, d_desc, (size_t)n * 128 * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(ell, d_ell, (size_t)n * 5 * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(ids, d_ids, (size_t)n * 3 * 4, hipMemcpyDeviceToHost));

    write_file(argv[8], ".lafs.f32", lafs, (size_t)n * 6 * 4);
    write_file(argv[8], ".resp.f32", resp, (size_t)n * 4);
    write_file(argv[8], ".desc.f32", desc, (size_t)n * 128 * 4);
    write_file(argv[8], ".ids.i32", ids, (size_t)n * 3 * 4);
    char path[4096];
    snprintf(path, sizeof(path), "%s.txt", argv[8]);
    FILE* f = fopen(path, "w");
    if (!f) DIE("cannot write %s", path);
    fprintf(f, "1.0\n%d\n", n);                                                          /* Utils.py:177-182 line_prepender */
    for (int i = 0; i < n; ++i)                                                          /* hesaffnet.py:58: fmt '%10.10f' */
        fprintf(f, "%10.10f %10.10f %10.10f %10.10f %10.10f\n", ell[5 * i], ell[5 * i + 1], ell[5 * i + 2], ell[5 * i + 3], ell[5 * i + 4]);
    fclose(f);
    printf("%s: %d keypoints (%d detector candidates), %s, workspace %.1f MB\n", affnet_version(), n, counts[0],
           arith == 0 ? "exact fp32" : arith == 1 ? "fp32_split3" : "fp32_split2h", ws_bytes / 1048576.0);

    affnet_ctx_destroy(ctx);
    hipFree(d_ws); hipFree(d_img); hipFree(d_lafs); hipFree(d_resp); hipFree(d_ids); hipFree(d_desc); hipFree(d_ell); hipFree(d_count);
    hipFree((void*)nets.d_affnet); hipFree((void*)nets.d_orinet); hipFree((void*)nets.d_hardnet);
    free(img); free(cfg); free(lafs); free(resp); free(desc); free(ell); free(ids);
    return 0;
}
