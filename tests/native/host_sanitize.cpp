// Sanitizer harness for the host-side C++ of libhybridgl (gtmask.cpp, cvkernel.cpp), built by tests/test_sanitize.py with
// g++ -fsanitize=address,undefined.  Reads a case file written by the test:
//   P H W npolys  k_0 .. k_{n-1}  x y x y ...      polygons
//   S H W <string>                                  compressed RLE string (may be malformed on purpose)
//   K n sigma                                       Gaussian kernel taps
// and prints one line per case: the return code and an FNV-1a hash of the mask / taps (compared with the regular build).
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
int hgl_gt_mask_from_polygons(const double* xy, const int32_t* n_points, int n_polys, int H, int W, uint8_t* mask, int64_t* area);
int hgl_gt_mask_from_rle_string(const char* s, int H, int W, uint8_t* mask, int64_t* area);
int hgl_cv_gaussian_kernel_q8(int n, double sigma, uint16_t* taps);
}
void hgl_set_error(const char*, ...) {}

static unsigned long long fnv(const void* p, size_t n) {
  const unsigned char* b = (const unsigned char*)p;
  unsigned long long h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "r");
  if (!f) return 2;
  char kind[8];
  while (fscanf(f, "%7s", kind) == 1) {
    if (kind[0] == 'P') {
      int H, W, n;
      if (fscanf(f, "%d %d %d", &H, &W, &n) != 3) return 3;
      std::vector<int32_t> k(n);
      size_t tot = 0;
      for (int i = 0; i < n; ++i) { if (fscanf(f, "%d", &k[i]) != 1) return 3; tot += k[i] > 0 ? 2 * (size_t)k[i] : 0; }
      std::vector<double> xy(tot);
      for (size_t i = 0; i < tot; ++i) if (fscanf(f, "%lf", &xy[i]) != 1) return 3;
      std::vector<uint8_t> m((size_t)H * W);
      int64_t area = -1;
      const int rc = hgl_gt_mask_from_polygons(xy.data(), k.data(), n, H, W, m.data(), &area);
      printf("P %d %lld %llu\n", rc, rc ? -1ll : (long long)area, rc ? 0ull : fnv(m.data(), m.size()));
    } else if (kind[0] == 'S') {
      int H, W;
      char buf[4096];
      if (fscanf(f, "%d %d %4095s", &H, &W, buf) != 3) return 3;
      std::string s(buf);
      for (char& c : s) if (c == '\x01') c = ' ';
      if (s == "<empty>") s.clear();
      // exact-size heap copy: a read past the terminator is a heap-buffer-overflow for ASan
      char* heap = new char[s.size() + 1];
      memcpy(heap, s.c_str(), s.size() + 1);
      std::vector<uint8_t> m((size_t)H * W);
      int64_t area = -1;
      const int rc = hgl_gt_mask_from_rle_string(heap, H, W, m.data(), &area);
      delete[] heap;
      printf("S %d %lld %llu\n", rc, rc ? -1ll : (long long)area, rc ? 0ull : fnv(m.data(), m.size()));
    } else if (kind[0] == 'K') {
      int n;
      double sigma;
      if (fscanf(f, "%d %lf", &n, &sigma) != 2) return 3;
      std::vector<uint16_t> taps(n > 0 ? n : 1);
      const int rc = hgl_cv_gaussian_kernel_q8(n, sigma, taps.data());
      printf("K %d 0 %llu\n", rc, rc ? 0ull : fnv(taps.data(), taps.size() * 2));
    } else {
      return 3;
    }
  }
  fclose(f);
  return 0;
}
