// Ground-truth masks of the REFER annotations, host side (no device work).
//
// The reference forms a ref's target mask on the CPU inside its DataLoader workers:
// refer/refer.py:277-291 getMask -> pycocotools-style mask.frPyObjects / mask.decode, i.e.
// refer/external/maskApi.c rleFrPoly (:161-201), rleDecode (:43-47), rleFrString (:217-230); the dataset then
// keeps the pixels whose polygon count is exactly one (data/dataset_refer_bert.py:118-121).  These entry
// points restate that arithmetic (5x super-sampled boundary walk, column crossings, even-odd fill in
// column-major order) and write the row-major uint8 image that hgl_iou consumes, without materialising
// the intermediate RLE objects.  Bit-exact against the reference's C file (oracle/_ref, tests/golden/gtmask.npz).
#include "hgl_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace {

// Column-major positions (x*H + y) at which the polygon's fill toggles (maskApi.c:161-196).
void polygon_crossings(const double* xy, int k, int H, int W, std::vector<unsigned>& pos) {
  const double scale = 5.0;
  std::vector<int> px(k + 1), py(k + 1);
  for (int j = 0; j < k; ++j) {
    px[j] = (int)(scale * xy[2 * j] + 0.5);
    py[j] = (int)(scale * xy[2 * j + 1] + 0.5);
  }
  px[k] = px[0];
  py[k] = py[0];
  // dense integer walk along every edge, major axis one step at a time, in the edge's own direction
  std::vector<int> u, v;
  for (int j = 0; j < k; ++j) {
    int xs = px[j], xe = px[j + 1], ys = py[j], ye = py[j + 1];
    const int dx = std::abs(xe - xs), dy = std::abs(ys - ye);
    const bool x_major = dx >= dy;
    const bool flip = (x_major && xs > xe) || (!x_major && ys > ye);
    if (flip) { std::swap(xs, xe); std::swap(ys, ye); }
    const double slope = x_major ? (double)(ye - ys) / dx : (double)(xe - xs) / dy;   // 0/0 -> NaN, used for one point only
    const int n = x_major ? dx : dy;
    for (int d = 0; d <= n; ++d) {
      const int t = flip ? n - d : d;
      if (x_major) { u.push_back(t + xs); v.push_back((int)(ys + slope * t + 0.5)); }
      else { v.push_back(t + ys); u.push_back((int)(xs + slope * t + 0.5)); }
    }
  }
  // where the walk moves to another super-sampled column: the crossing, back at pixel resolution
  for (size_t j = 1; j < u.size(); ++j) {
    if (u[j] == u[j - 1]) continue;
    double xd = (double)(u[j] < u[j - 1] ? u[j] : u[j] - 1);
    xd = (xd + 0.5) / scale - 0.5;
    if (std::floor(xd) != xd || xd < 0 || xd > W - 1) continue;
    double yd = (double)(v[j] < v[j - 1] ? v[j] : v[j - 1]);
    yd = (yd + 0.5) / scale - 0.5;
    if (yd < 0) yd = 0; else if (yd > H) yd = H;
    yd = std::ceil(yd);
    pos.push_back((unsigned)((int)xd * H + (int)yd));
  }
}

// even-odd fill of sorted toggle positions, column-major scan, added into the row-major image
long long fill_toggles(std::vector<unsigned>& pos, int H, int W, uint8_t* mask) {
  std::sort(pos.begin(), pos.end());
  long long area = 0;
  const unsigned total = (unsigned)H * (unsigned)W;
  size_t i = 0;
  while (i < pos.size()) {
    // a position listed an even number of times does not toggle (maskApi.c:192-196 merges the empty runs)
    size_t j = i;
    while (j < pos.size() && pos[j] == pos[i]) ++j;
    const bool toggles = ((j - i) & 1) != 0;
    const unsigned start = pos[i];
    i = j;
    if (!toggles) continue;
    // find the next toggling position
    unsigned end = total;
    while (i < pos.size()) {
      size_t j2 = i;
      while (j2 < pos.size() && pos[j2] == pos[i]) ++j2;
      const bool t2 = ((j2 - i) & 1) != 0;
      const unsigned p2 = pos[i];
      i = j2;
      if (t2) { end = p2; break; }
    }
    {
      unsigned x = start / (unsigned)H, y = start - x * (unsigned)H;
      for (unsigned p = start; p < end && p < total; ++p) {
        mask[(size_t)y * W + x] += 1;
        if (++y == (unsigned)H) { y = 0; ++x; }
      }
    }
    area += (long long)(std::min(end, total) - std::min(start, total));
  }
  return area;
}

long long fill_counts(const unsigned* cnts, long long m, int H, int W, uint8_t* mask) {
  const unsigned long long total = (unsigned long long)H * W;
  unsigned long long p = 0;
  long long area = 0;
  for (long long j = 0; j < m; ++j) {
    const unsigned long long c = cnts[j];
    if (j & 1) {
      unsigned long long x = p / (unsigned)H, y = p - x * (unsigned)H;
      for (unsigned long long q = p; q < p + c && q < total; ++q) {
        mask[(size_t)y * W + x] += 1;
        if (++y == (unsigned long long)H) { y = 0; ++x; }
      }
      area += (long long)c;
    }
    p += c;
  }
  return area;
}

}  // namespace

extern "C" {

int hgl_gt_mask_from_polygons(const double* xy, const int32_t* n_points, int n_polys, int H, int W, uint8_t* mask,
                              int64_t* area) {
  HGL_REQUIRE(xy && n_points && mask && n_polys >= 0 && H > 0 && W > 0, "gt_mask_from_polygons: bad arguments");
  HGL_REQUIRE((long long)H * W < (1ll << 31), "gt_mask_from_polygons: image too large");
  std::fill(mask, mask + (size_t)H * W, (uint8_t)0);
  long long total = 0;
  const double* p = xy;
  std::vector<unsigned> pos;
  for (int i = 0; i < n_polys; ++i) {
    const int k = n_points[i];
    HGL_REQUIRE(k >= 1, "gt_mask_from_polygons: polygon %d has %d points", i, k);
    for (int j = 0; j < 2 * k; ++j)      // the boundary walk is 5 steps per pixel: keep it finite and the int casts defined
      HGL_REQUIRE(p[j] == p[j] && p[j] > -1.0e5 && p[j] < 1.0e5, "gt_mask_from_polygons: coordinate %d of polygon %d out of range", j, i);
    pos.clear();
    polygon_crossings(p, k, H, W, pos);
    total += fill_toggles(pos, H, W, mask);
    p += 2 * (size_t)k;
  }
  if (area) *area = total;
  return HGL_OK;
}

int hgl_gt_mask_from_rle_counts(const uint32_t* counts, int m, int H, int W, uint8_t* mask, int64_t* area) {
  HGL_REQUIRE(counts && mask && m >= 0 && H > 0 && W > 0, "gt_mask_from_rle_counts: bad arguments");
  std::fill(mask, mask + (size_t)H * W, (uint8_t)0);
  const long long a = fill_counts(counts, m, H, W, mask);
  if (area) *area = a;
  return HGL_OK;
}

int hgl_gt_mask_from_rle_string(const char* s, int H, int W, uint8_t* mask, int64_t* area) {
  HGL_REQUIRE(s && mask && H > 0 && W > 0, "gt_mask_from_rle_string: bad arguments");
  // LEB128-like, 5 payload bits + continuation bit per character (offset 48), every count after the third stored
  // as a difference to the count two places earlier (maskApi.c:217-230)
  // The string comes from an annotation file: every shift below is on unsigned 64-bit values with a bounded count
  // (a count takes at most 7 characters = 35 bits; maskApi.c itself shifts 32-bit ints, so longer groups never occur in
  // valid data), a group cut off by the end of the string or longer than that is an error, never undefined behaviour.
  std::vector<unsigned> cnts;
  size_t p = 0;
  while (s[p]) {
    unsigned long long ux = 0;
    int k = 0;
    bool more = true;
    while (more) {
      HGL_REQUIRE(s[p] != 0, "gt_mask_from_rle_string: truncated string");
      HGL_REQUIRE(k < 7, "gt_mask_from_rle_string: malformed count (more than 7 characters)");
      const unsigned c = (unsigned)(unsigned char)(s[p] - 48);
      ux |= (unsigned long long)(c & 0x1fu) << (5 * k);
      more = (c & 0x20u) != 0;
      ++p;
      ++k;
      if (!more && (c & 0x10u)) ux |= ~0ULL << (5 * k);      // sign extension (maskApi.c: x |= -1 << 5*k)
    }
    long long x = (long long)ux;
    if (cnts.size() > 2) x += (long long)cnts[cnts.size() - 2];
    cnts.push_back((unsigned)(unsigned long long)x);
  }
  std::fill(mask, mask + (size_t)H * W, (uint8_t)0);
  const long long a = fill_counts(cnts.data(), (long long)cnts.size(), H, W, mask);
  if (area) *area = a;
  return HGL_OK;
}

// ---- the other direction: what SamAutomaticMaskGenerator's output_mode "uncompressed_rle" / "coco_rle" hands out
// (automatic_mask_generator.py:176-182, utils/amg.py:107-153,294-300 -> pycocotools frPyObjects -> maskApi.c rleToString)

// Column-major run lengths of a host mask [H,W] (any non-zero byte = foreground), first count = leading zeros (0 when the
// mask starts with foreground): utils/amg.py:107-136 mask_to_rle_pytorch.  *m receives the number of counts; with
// counts == nullptr or cap too small nothing is written beyond cap and the needed length is still reported.
int hgl_rle_encode_mask(const uint8_t* mask, int H, int W, uint32_t* counts, long long cap, long long* m) {
  HGL_REQUIRE(mask && m && H > 0 && W > 0 && cap >= 0, "rle_encode_mask: bad arguments");
  long long n = 0;
  unsigned run = 0;
  bool cur = false;      // the first run counts zeros
  for (int x = 0; x < W; ++x)
    for (int y = 0; y < H; ++y) {
      const bool v = mask[(size_t)y * W + x] != 0;
      if (v != cur) {
        if (counts && n < cap) counts[n] = run;
        ++n;
        run = 0;
        cur = v;
      }
      ++run;
    }
  if (counts && n < cap) counts[n] = run;
  ++n;
  *m = n;
  return HGL_OK;
}

// maskApi.c:203-216 rleToString: every count (from the fourth on: its difference to the count two places earlier) as a
// little-endian group of 5-bit digits + continuation bit, one ASCII character (offset 48) per digit.  out needs
// 6 * m + 1 bytes at most (a 32-bit value takes 7 characters only when negative differences need the sign digit; the
// bound checked here is the exact one).  *len receives strlen(out).
int hgl_rle_to_string(const uint32_t* counts, long long m, char* out, size_t cap, size_t* len) {
  HGL_REQUIRE(counts && out && len && m >= 0 && cap >= 1, "rle_to_string: bad arguments");
  size_t p = 0;
  for (long long i = 0; i < m; ++i) {
    long long x = (long long)counts[i];
    if (i > 2) x -= (long long)counts[i - 2];
    bool more = true;
    while (more) {
      unsigned c = (unsigned)(x & 0x1f);
      x >>= 5;                                        // arithmetic shift, as the C `long` of the original
      more = (c & 0x10) ? x != -1 : x != 0;
      if (more) c |= 0x20;
      HGL_REQUIRE(p + 1 < cap, "rle_to_string: output buffer too small (%zu bytes)", cap);
      out[p++] = (char)(c + 48);
    }
  }
  out[p] = 0;
  *len = p;
  return HGL_OK;
}

}  // extern "C"
