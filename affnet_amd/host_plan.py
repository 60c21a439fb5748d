"""Host-side mirror of the reference's pyramid bookkeeping: octave sizes, level sigmas, Gaussian
tap tables.  Fills the affnet_config struct of include/affnet_hip.h.

Follows HandCraftedModules.py:14-56 (ScalePyramid) and Utils.py:92-114,155-161
(CircularGaussKernel / GaussianBlur.calculate_weights) under Python-3 semantics (`kernlen / 2`
is a true division), which is the behaviour the parity oracle pins.  Keeping these formulas on
the host means device code never hard-codes a Gaussian.
"""
import numpy as np

from . import _lib


def gaussian_taps(sigma):
    """(k, k) float32 taps, k = int(6 sigma + 1) | 1, sampled at linspace(-k/2, k/2, k)."""
    k = int(2.0 * 3.0 * sigma + 1.0)
    k += (k % 2 == 0)
    half = k / 2
    ax = np.linspace(-half, half, k)
    xv, yv = np.meshgrid(ax, ax, sparse=False, indexing="xy")
    ker = np.exp(-((xv ** 2 + yv ** 2) / (2.0 * sigma * sigma)))
    ker /= np.sum(ker)
    return ker.astype(np.float32)


class PyramidPlan(object):
    def __init__(self, height, width, n_levels=3, init_sigma=1.6, border=5):
        self.height, self.width = int(height), int(width)
        self.n_levels, self.init_sigma, self.border = n_levels, init_sigma, border
        step = 2 ** (1.0 / float(n_levels))
        min_size = 2 * border + 2 + 1
        cur = 0.5
        self.first_blur_sigma = None
        if init_sigma > cur:
            self.first_blur_sigma = float(np.sqrt(init_sigma ** 2 - cur ** 2))
            cur = init_sigma
        self.sizes, self.sigmas, self.pix_dists = [], [], []
        self.blur_sigmas_per_octave = []
        h, w, pix = self.height, self.width, 1.0
        while True:
            lev, blur = [cur], []
            for _ in range(1, n_levels + 2):
                blur.append(float(cur * np.sqrt(step * step - 1.0)))
                cur *= step
                lev.append(cur)
            # curSigma restarts at init_sigma after every octave (HandCraftedModules.py:49): octaves >= 1 all share one blur
            # sequence; octave 0 has the same one unless init_sigma <= 0.5 (it then starts at curSigma = 0.5, :25-31)
            self.blur_sigmas_per_octave.append(blur)
            self.sizes.append((h, w))
            self.sigmas.append(lev)
            self.pix_dists.append([pix] * len(lev))
            nh, nw = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            pix *= 2.0
            cur = init_sigma
            if nh <= min_size or nw <= min_size:
                break
            h, w = nh, nw
        self.n_octaves = len(self.sizes)
        self.blur_sigmas = self.blur_sigmas_per_octave[0]
        self.levels_per_octave = n_levels + 2

    def fill_config(self, mr_size, threshold, num_features, num_prefilter, max_keep=16384, raw_div=4, batch=1, baum_iters=0, onepass=False, lazy_shape_rows=-1, arith=0):
        if self.n_octaves > _lib.MAX_OCTAVES or self.levels_per_octave > _lib.MAX_LEVELS:
            raise ValueError("pyramid too deep for the library limits")
        c = _lib.Config()
        c.height, c.width = self.height, self.width
        c.n_octaves, c.levels_per_octave = self.n_octaves, self.levels_per_octave
        for o, (h, w) in enumerate(self.sizes):
            c.oct_h[o], c.oct_w[o] = h, w
            for l, s in enumerate(self.sigmas[o]):
                c.level_sigma[o][l] = np.float32(s)
                c.level_sigma4[o][l] = np.float32(s ** 4)   # tensor * python float -> float32 scalar
                c.level_sigma_px[o][l] = float(np.array(s) * np.array(self.pix_dists[o][l]))
        if self.first_blur_sigma is not None:
            t = gaussian_taps(self.first_blur_sigma)
            if t.shape[0] > _lib.MAX_TAPS:
                raise ValueError("initial Gaussian of %d taps not supported (max %d)" % (t.shape[0], _lib.MAX_TAPS))
            c.first_blur_taps = t.shape[0]
            flat = t.reshape(-1)
            c.first_blur[:flat.size] = flat.tolist()
        def put(sigmas, taps_field, table_field):
            for l, s in enumerate(sigmas, start=1):
                t = gaussian_taps(s)
                if t.shape[0] > _lib.MAX_TAPS:
                    raise ValueError("Gaussian of %d taps not supported by the blur kernel (max %d)" % (t.shape[0], _lib.MAX_TAPS))
                taps_field[l] = t.shape[0]
                flat = t.reshape(-1)
                table_field[l][:flat.size] = flat.tolist()

        later = self.blur_sigmas_per_octave[1] if self.n_octaves > 1 else self.blur_sigmas_per_octave[0]
        put(later, c.level_blur_taps, c.level_blur)
        if self.blur_sigmas_per_octave[0] != later:          # init_sigma <= 0.5: octave 0 blurs with its own kernels
            put(self.blur_sigmas_per_octave[0], c.level_blur0_taps, c.level_blur0)
        c.mr_size, c.threshold = float(mr_size), float(threshold)
        c.num_features, c.num_prefilter = int(num_features), int(num_prefilter)
        c.max_raw_per_octave_div, c.max_keep = int(raw_div), int(max_keep)
        c.batch = int(batch)
        c.baum_iters = int(baum_iters)
        c.onepass = 1 if onepass else 0
        c.lazy_shape_rows = int(lazy_shape_rows)
        c.arith = _lib.arith_code(arith)
        return c


def circular_gauss_kernel(kernlen=None, circ_zeros=False, sigma=None, norm=True):
    """Utils.py:92-114 (CircularGaussKernel) under Python-3 semantics: the window of the hand-crafted
    OrientationDetector / AffineShapeEstimator slots, computed on the host and handed to the kernels as a table."""
    if kernlen is None:
        kernlen = int(2.0 * 3.0 * sigma + 1.0)
        if kernlen % 2 == 0:
            kernlen = kernlen + 1
    half = kernlen / 2
    r2 = float(half * half)
    sigma2 = 0.9 * r2 if sigma is None else 2.0 * sigma * sigma
    x = np.linspace(-half, half, kernlen)
    xv, yv = np.meshgrid(x, x, sparse=False, indexing="xy")
    distsq = xv ** 2 + yv ** 2
    kernel = np.exp(-(distsq / sigma2))
    if circ_zeros:
        kernel *= (distsq <= r2).astype(np.float32)
    if norm:
        kernel /= np.sum(kernel)
    return kernel
