// 8-bit fixed-point Gaussian kernel of cv2.GaussianBlur on uint8 images (Hybridgl_main.py:99
// `cv2.GaussianBlur(sam_img, (15, 15), 0)`), host side.
//
// opencv-python 4.10.0.84 is an external dependency of the reference (environment.yaml) and absent here; this
// restates the published OpenCV 4.x algorithm (modules/imgproc/src/smooth.dispatch.cpp: getGaussianKernelBitExact
// and getGaussianKernelFixedPoint_ED): sigma = 0.15 n + 0.35 when not given, e^(-x^2 / (2 sigma^2)) normalised in
// double precision, then rounded to 8 fractional bits from the border inwards with the rounding error carried to
// the next tap, the centre tap taking whatever makes the sum exactly 256.  Parity with the package is unpinned
// (OpenCV evaluates this in its own soft-float `softdouble`; libm's exp agrees to the last place or one ulp, which
// only matters on an exact rounding tie).
#include "hgl_common.h"

#include <cmath>
#include <vector>

extern "C" int hgl_cv_gaussian_kernel_q8(int n, double sigma, uint16_t* taps) {
  HGL_REQUIRE(taps && n >= 1 && n <= 31 && (n & 1), "cv_gaussian_kernel_q8: n must be odd, 1..31 (got %d)", n);
  // OpenCV tabulates the kernels of size <= 7 for sigma <= 0 (smooth.dispatch.cpp small_gaussian_tab)
  static const double small_tab[4][7] = {{1.0},
                                         {0.25, 0.5, 0.25},
                                         {0.0625, 0.25, 0.375, 0.25, 0.0625},
                                         {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125}};
  std::vector<double> k(n);
  if (n <= 7 && sigma <= 0) {
    for (int i = 0; i < n; ++i) k[i] = small_tab[n >> 1][i];
  } else {
    const double sx = sigma > 0 ? sigma : std::fma((double)n, 0.15, 0.35);
    const double scale2x = -0.125 / (sx * sx);
    const int n2 = (n - 1) / 2;
    std::vector<double> v(n2 + 1);
    double sum = 0.0;
    for (int i = 0, x = 1 - n; i < n2; ++i, x += 2) {   // x = 2 * (i - (n-1)/2)
      v[i] = std::exp((double)(x * x) * scale2x);
      sum += v[i];
    }
    sum *= 2.0;
    sum += 1.0;
    const double mul1 = 1.0 / sum;
    for (int i = 0; i < n2; ++i) k[i] = k[n - 1 - i] = v[i] * mul1;
    k[n2] = 1.0 * mul1;
  }
  // 8 fractional bits with error diffusion (getGaussianKernelFixedPoint_ED)
  const int n2 = n / 2;
  double err = 0.0;
  long long sum = 0;
  for (int i = 0; i < n2; ++i) {
    const double adj = k[i] * 256.0 + err;
    const long long v0 = (long long)std::nearbyint(adj);   // cvRound: to nearest, ties to even
    err = adj - (double)v0;
    taps[i] = taps[n - 1 - i] = (uint16_t)v0;
    sum += 2 * v0;
  }
  taps[n2] = (uint16_t)(256 - sum);
  return HGL_OK;
}
