// Error plumbing, device probing and the extern "C" wrappers of the primitive operators.
#include "hgl_common.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

// the ONLY reads of the environment in the library (see hgl_common.h for the list of switches)
const char* hgl_env_str(const char* name) { return getenv(name); }
int hgl_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

static thread_local char g_err[512] = "";

void hgl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hgl_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    hgl_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return HGL_ELAUNCH;
  }
  return HGL_OK;
}

int hgl_require_device() {
  static int cached = -1;
  if (cached < 0) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    cached = (e == hipSuccess && n > 0) ? 1 : 0;
    if (e != hipSuccess) (void)hipGetLastError();
  }
  if (!cached) {
    hgl_set_error("no HIP device visible: libhybridgl has no CPU path");
    return HGL_ENODEVICE;
  }
  return HGL_OK;
}

// ---- profiler: event pairs recorded around launches while enabled ----
#include <vector>
namespace {
struct ProfRec { hipEvent_t a, b; int cls; double flops, bytes; };
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
bool g_prof_on = false;
hipEvent_t take_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}
}  // namespace

HglProfScope::HglProfScope(int cls, double flops, double bytes, hipStream_t s) : slot(-1), st(s) {
  if (!g_prof_on) return;
  ProfRec r{take_event(), take_event(), cls, flops, bytes};
  (void)hipEventRecord(r.a, st);
  slot = (int)g_recs.size();
  g_recs.push_back(r);
}
HglProfScope::~HglProfScope() {
  if (slot >= 0) (void)hipEventRecord(g_recs[slot].b, st);
}

extern "C" {

int hgl_prof_enable(int on) {
  HGL_TRY(hgl_require_device());
  g_prof_on = on != 0;
  return HGL_OK;
}

// Synchronises, sums and clears the records of class `cls`: launches, total ms, total
// algorithmic flops and bytes the launchers declared.
int hgl_prof_read(int cls, long long* launches, double* ms, double* flops, double* bytes) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(cls >= 0 && cls < HGL_PROF_NCLASS && launches && ms && flops && bytes, "prof_read: bad arguments");
  *launches = 0; *ms = 0; *flops = 0; *bytes = 0;
  std::vector<ProfRec> keep;
  for (auto& r : g_recs) {
    if (r.cls != cls) { keep.push_back(r); continue; }
    (void)hipEventSynchronize(r.b);
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { *ms += t; *flops += r.flops; *bytes += r.bytes; ++*launches; }
    g_pool.push_back(r.a); g_pool.push_back(r.b);
  }
  g_recs.swap(keep);
  return HGL_OK;
}

// Number of GPU threads of the split-fp16 (f16x3) path that met a value beyond the fp16 range since the last reset
// (GEMM split epilogues / operand splits, attention staging).  Blocking device reads: synchronise the streams that ran the work first.
int hgl_split_overflow_count(int reset, unsigned long long* count) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(count != nullptr, "split_overflow_count: null argument");
  *count = hgl_split_overflow_gemm(reset) + hgl_split_overflow_attention(reset) + hgl_split_overflow_decoder(reset);
  return HGL_OK;
}

// The same counters copied to `host2` (two words of PINNED host memory: GEMM, attention) in stream order, without waiting:
// the value is what the device had counted when the stream reached this point.  Lets a caller that already reads something
// back per batch (the evaluator's proposal counts) notice an overflow within that batch instead of at the end of the run.
int hgl_split_overflow_peek_async(unsigned int* host2, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(host2 != nullptr, "split_overflow_peek_async: null argument");
  if (hgl_split_overflow_gemm_peek(host2, (hipStream_t)stream) != 0 ||
      hgl_split_overflow_attention_peek(host2 + 1, (hipStream_t)stream) != 0) {
    hgl_set_error("split_overflow_peek_async: hipMemcpyFromSymbolAsync failed");
    return HGL_ELAUNCH;
  }
  return HGL_OK;
}

int hgl_abi_version(void) { return HGL_ABI_VERSION; }
const char* hgl_last_error(void) { return g_err; }

int hgl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int hgl_gemm_f32(const float* A, const float* W, const float* bias, const float* R, float* C, int M,
                 int N, int K, int lda, int ldw, int ldr, int ldc, int batch, long long sA,
                 long long sW, long long sR, long long sC, int act, void* stream) {
  HGL_TRY(hgl_require_device());
  return hgl_launch_gemm(A, W, bias, R, C, M, N, K, lda, ldw, ldr, ldc, batch, sA, sW, sR, sC, act,
                         (hipStream_t)stream);
}

int hgl_layernorm_f32(const float* x, const float* w, const float* b, float* y, int rows, int D,
                      float eps, void* stream) {
  HGL_TRY(hgl_require_device());
  return hgl_launch_layernorm(x, w, b, y, rows, D, eps, (hipStream_t)stream);
}

int hgl_attention_f32(const float* q, const float* k, const float* v, float* out, int B, int H, int Sq,
                      int Sk, int hd, int ldq, int ldk, int ldv, int ldo, long long sqb,
                      long long skb, long long svb, long long sob, float scale, int mask_kind,
                      const uint8_t* keep, int keep_b0, int keep_n, const float* rel_h,
                      const float* rel_w, int kh, int kw, void* stream) {
  HGL_TRY(hgl_require_device());
  return hgl_launch_attention(q, k, v, out, B, H, Sq, Sk, hd, ldq, ldk, ldv, ldo, sqb, skb, svb, sob,
                              scale, mask_kind, keep, keep_b0, keep_n, rel_h, rel_w, kh, kw,
                              (hipStream_t)stream);
}

int hgl_mask_resize(const uint8_t* masks, int N, int Hm, int Wm, int g, float* pm, void* stream) {
  HGL_TRY(hgl_require_device());
  return hgl_launch_mask_resize(masks, N, Hm, Wm, g, pm, nullptr, (hipStream_t)stream);
}

}  // extern "C"
