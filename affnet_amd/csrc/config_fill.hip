// affnet_config_fill: the reference's pyramid bookkeeping and Gaussian tap tables computed by the library itself, so that a caller
// of include/affnet_hip.h that is not Python can create a context (examples/c_host/extract.c).  Host code only.
//
// Follows HandCraftedModules.py:14-56 (ScalePyramid: stop rule minSize = 2 border + 3, level sigmas, incremental blur sigmas, curSigma
// restarting at init_sigma after every octave) and Utils.py:92-114,155-161 (CircularGaussKernel / GaussianBlur.calculate_weights) under
// Python-3 semantics (`kernlen / 2` is a true division) - the behaviour the parity oracle pins.  Every intermediate is a double computed
// with the same operations, in the same order, as the numpy / Python-float expressions of affnet_amd/host_plan.py (which restates the
// reference for the Python mirror): pow() where Python writes `**`, numpy's linspace (i * step + start, last element = stop), numpy's
// pairwise summation for np.sum, one rounding to float32 at the end.  tests/test_host_mirror.py compares the two structs byte for byte.
#include <math.h>
#include <string.h>

#include "../../include/affnet_hip.h"

namespace {

// numpy/_core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum (contiguous input): what np.sum does on a float64 array
double np_pairwise_sum(const double* a, long n) {
    if (n < 8) {
        double res = 0.0;
        for (long i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        long i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

// Utils.py:156-161 with :92-114: k = int(6 sigma + 1) | 1, taps exp(-(x^2 + y^2) / (2 sigma^2)) at linspace(-k/2, k/2, k), / sum, -> float32.
// Returns k, or -1 if k exceeds AFFNET_MAX_TAPS.
int gaussian_taps(double sigma, float* out) {
    int k = (int)(2.0 * 3.0 * sigma + 1.0);
    if (k % 2 == 0) k += 1;
    if (k > AFFNET_MAX_TAPS) return -1;
    const double half = (double)k / 2.0;
    double ax[AFFNET_MAX_TAPS];
    if (k == 1) {
        ax[0] = -half;
    } else {
        const double step = (half - (-half)) / (double)(k - 1);          // numpy.linspace: delta / div
        for (int i = 0; i < k; ++i) ax[i] = (double)i * step + (-half);
        ax[k - 1] = half;
    }
    static thread_local double ker[AFFNET_MAX_TAPS * AFFNET_MAX_TAPS];
    const double two_s2 = 2.0 * sigma * sigma;
    for (int r = 0; r < k; ++r)
        for (int c = 0; c < k; ++c) ker[r * k + c] = exp(-((ax[c] * ax[c] + ax[r] * ax[r]) / two_s2));
    const double sum = np_pairwise_sum(ker, (long)k * k);
    for (int i = 0; i < k * k; ++i) out[i] = (float)(ker[i] / sum);
    return k;
}

}  // namespace

extern "C" int affnet_config_fill(affnet_config* cfg, int height, int width, int n_levels, double init_sigma, int border, double mr_size,
                                  double threshold, int num_features, int num_prefilter, int batch, int baum_iters) {
    if (!cfg || height < 1 || width < 1 || n_levels < 1 || n_levels + 2 > AFFNET_MAX_LEVELS || border < 0) return AFFNET_ERR_INVALID;
    memset(cfg, 0, sizeof(*cfg));
    const double step = pow(2.0, 1.0 / (double)n_levels);                 // HandCraftedModules.py:17
    const int min_size = 2 * border + 2 + 1;                              // :20-21
    double cur = 0.5;
    double first_blur_sigma = -1.0;
    if (init_sigma > cur) {                                               // :25-31
        first_blur_sigma = sqrt(pow(init_sigma, 2.0) - pow(cur, 2.0));
        cur = init_sigma;
    }
    double blur0[AFFNET_MAX_LEVELS] = {0}, blur_later[AFFNET_MAX_LEVELS] = {0};
    int h = height, w = width, n_oct = 0;
    double pix = 1.0;
    for (;;) {
        if (n_oct >= AFFNET_MAX_OCTAVES) return AFFNET_ERR_INVALID;       // pyramid too deep for the library limits
        double* blur = n_oct == 0 ? blur0 : blur_later;
        cfg->oct_h[n_oct] = h;
        cfg->oct_w[n_oct] = w;
        double lev = cur;
        for (int l = 0; l < n_levels + 2; ++l) {                          // :39-47
            if (l > 0) {
                blur[l] = cur * sqrt(step * step - 1.0);
                cur *= step;
                lev = cur;
            }
            cfg->level_sigma[n_oct][l] = (float)lev;
            cfg->level_sigma4[n_oct][l] = (float)pow(lev, 4.0);           // HessianResp: tensor * sigma**4 (:78)
            cfg->level_sigma_px[n_oct][l] = lev * pix;                    // LAF.py:459
        }
        ++n_oct;
        const int nh = (h - 1) / 2 + 1, nw = (w - 1) / 2 + 1;             // avg_pool2d(kernel 1, stride 2) (:48)
        pix *= 2.0;
        cur = init_sigma;                                                 // :49
        if (nh <= min_size || nw <= min_size) break;                      // :50
        h = nh;
        w = nw;
    }
    cfg->height = height;
    cfg->width = width;
    cfg->n_octaves = n_oct;
    cfg->levels_per_octave = n_levels + 2;
    if (first_blur_sigma >= 0.0) {
        const int k = gaussian_taps(first_blur_sigma, cfg->first_blur);
        if (k < 0) return AFFNET_ERR_INVALID;
        cfg->first_blur_taps = k;
    }
    const double* later = n_oct > 1 ? blur_later : blur0;
    bool differ = false;
    for (int l = 1; l < n_levels + 2; ++l) {
        const int k = gaussian_taps(later[l], cfg->level_blur[l]);
        if (k < 0) return AFFNET_ERR_INVALID;
        cfg->level_blur_taps[l] = k;
        differ |= blur0[l] != later[l];
    }
    if (differ)                                                           // init_sigma <= 0.5: octave 0 blurs with its own kernels
        for (int l = 1; l < n_levels + 2; ++l) {
            const int k = gaussian_taps(blur0[l], cfg->level_blur0[l]);
            if (k < 0) return AFFNET_ERR_INVALID;
            cfg->level_blur0_taps[l] = k;
        }
    cfg->mr_size = (float)mr_size;
    cfg->threshold = (float)threshold;
    cfg->num_features = num_features;
    cfg->num_prefilter = num_prefilter;
    cfg->max_raw_per_octave_div = 4;
    cfg->max_keep = 16384;
    cfg->batch = batch;
    cfg->baum_iters = baum_iters;
    cfg->onepass = 0;
    cfg->lazy_shape_rows = -1;
    cfg->arith = AFFNET_ARITH_FP32_MFMA;
    return AFFNET_OK;
}
