"""ORACLE (test infrastructure, NOT product code): cv2.GaussianBlur on uint8 images as Hybridgl_main.py:99 calls it
(`cv2.GaussianBlur(sam_img, (15, 15), 0)`), restated from the published OpenCV 4.x sources.

PARITY UNPINNED: opencv-python==4.10.0.84 (environment.yaml) is an external package, absent from the reference tree
and from this image, and no reference test holds a vector of its output.  Restated: modules/imgproc/src/
smooth.dispatch.cpp `getGaussianKernelBitExact` + `getGaussianKernelFixedPoint_ED` (the 8.8 fixed-point taps) and
smooth.simd.hpp `fixedSmoothInvoker` (row pass in 16 bits, column pass in 32 bits, one rounding), which is the path
OpenCV takes for CV_8U input (its IPP variant is compiled out because it is not bit-exact).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import math
from fractions import Fraction

import numpy as np

_SMALL = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
          7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}


def _round_half_even(x):
    return int(round(x))      # Python's round() is to-nearest-even: cvRound


def gaussian_kernel_bitexact(n, sigma=0.0):
    """getGaussianKernelBitExact: double-precision taps (OpenCV evaluates them in softdouble)."""
    if n <= 7 and sigma <= 0:
        return list(_SMALL[n])
    sx = float(sigma) if sigma > 0 else float(Fraction(n) * Fraction(0.15) + Fraction(0.35))   # mulAdd: one rounding
    scale2x = -0.125 / (sx * sx)
    n2 = (n - 1) // 2
    v = [math.exp(float((2 * i + 1 - n) ** 2) * scale2x) for i in range(n2)]
    s = 0.0
    for t in v:
        s += t
    s *= 2.0
    s += 1.0
    mul1 = 1.0 / s
    k = [0.0] * n
    for i in range(n2):
        k[i] = k[n - 1 - i] = v[i] * mul1
    k[n2] = 1.0 * mul1
    return k


def gaussian_kernel_q8(n, sigma=0.0):
    """getGaussianKernelFixedPoint_ED with 8 fractional bits -> n integers that sum to 256."""
    k = gaussian_kernel_bitexact(n, sigma)
    out = [0] * n
    err, total = 0.0, 0
    for i in range(n // 2):
        adj = k[i] * 256.0 + err
        v0 = _round_half_even(adj)
        err = adj - float(v0)
        out[i] = out[n - 1 - i] = v0
        total += 2 * v0
    out[n // 2] = 256 - total
    return out


def gaussian_blur_u8(img, k=15, sigma=0.0):
    """cv2.GaussianBlur(img, (k, k), sigma) for uint8 [H, W, C], BORDER_REFLECT_101."""
    taps = np.asarray(gaussian_kernel_q8(k, sigma), dtype=np.int64)
    r = k // 2
    a = img.astype(np.int64)
    pad = np.pad(a, ((0, 0), (r, r), (0, 0)), mode="reflect")
    rows = sum(taps[i] * pad[:, i:i + img.shape[1]] for i in range(k))          # <= 255 * 256: ufixedpoint16
    assert rows.max() <= 65535
    pad = np.pad(rows, ((r, r), (0, 0), (0, 0)), mode="reflect")
    cols = sum(taps[i] * pad[i:i + img.shape[0]] for i in range(k))              # ufixedpoint32
    return np.minimum((cols + 32768) >> 16, 255).astype(np.uint8)
