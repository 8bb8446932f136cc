"""ORACLE (test infrastructure only) - the skimage SSIM that comparing-baseline.py:25 calls:
`skimage.measure.compare_ssim(x, y, data_range=255, multichannel=True)` (= `skimage.metrics.structural_similarity`).

skimage is absent from this image, so PARITY WITH THE PACKAGE IS UNPINNED: this restates its published algorithm
(Wang et al. 2004 as implemented in skimage/metrics/_structural_similarity.py: win_size 7, uniform filter, K1 = 0.01, K2 = 0.03,
use_sample_covariance=True -> cov_norm = NP/(NP-1), the map cropped by (win_size-1)//2 before the mean, channels averaged) on
scipy.ndimage.uniform_filter - the very filter skimage uses - in float64."""
import numpy as np
from scipy.ndimage import uniform_filter


def skimage_ssim(x, y, data_range=255.0, win_size=7, K1=0.01, K2=0.03):
    """x, y: [H,W,C] arrays on the data_range scale."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    vals = []
    for c in range(x.shape[2]):
        X, Y = x[..., c], y[..., c]
        NP = win_size ** 2
        cov_norm = NP / (NP - 1.0)
        ux, uy = uniform_filter(X, size=win_size), uniform_filter(Y, size=win_size)
        uxx, uyy, uxy = uniform_filter(X * X, size=win_size), uniform_filter(Y * Y, size=win_size), uniform_filter(X * Y, size=win_size)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
        pad = (win_size - 1) // 2
        vals.append(S[pad:-pad, pad:-pad].mean())
    return float(np.mean(vals))
