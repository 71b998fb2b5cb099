// Internal declarations shared by the HIP translation units of libhybridgl.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>
#include "../../include/hybridgl.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef __HIPCC__
// fp32 -> fp16 (hi, lo) with hi + lo == x up to the rounding of lo.  x must reach BOTH conversions as the same,
// already rounded fp32 value: when x is the result of a multiply / fma the compiler may otherwise fold that
// arithmetic into one of the conversions (single rounding from the exact product) and not the other, so that on
// ties the stored hi and the hi subtracted for lo are different fp16 neighbours and hi + lo misses x by 2*|lo|
// (measured: 0.1 % of attention rows off by up to 8e-5 relative).  The empty asm makes x opaque.
//
// Range: |x| > 65504 does not fit fp16 (hi = inf, lo = x - inf = -inf, and the MFMA sum is NaN).  The splits of values
// that are not bounded by construction -- GEMM outputs (epilogues with a split output, the attention's Q / K / V staging,
// the skinny kernel's A operand, the generic split kernel) -- go through the 4-argument form, which also folds |x| into a
// per-thread running maximum (one v_max per element, no branch); hgl_split_commit() at the end of the thread counts the
// thread in a per-translation-unit device counter when that maximum left the range.  hgl_split_overflow_count() sums the
// counters: the run then contains inf / NaN AND the host is told (hybridgl_amd raises and points at HYBRIDGL_PRECISION=f32)
// -- never silently.  LayerNorm outputs (|y| <= sqrt(D) |w| + |b|), normalised pixels, probabilities and convex
// combinations of already-checked values use the 3-argument form.  Weights are pre-scaled by a power of two and cannot
// overflow; activations of the trained CLIP / SAM models stay two to three orders of magnitude below the limit.
static __device__ __attribute__((unused)) unsigned int hgl_tu_split_overflow;
__device__ __forceinline__ void hgl_split_hi_lo(float x, _Float16& hi, _Float16& lo) {
  asm volatile("" : "+v"(x));
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}
__device__ __forceinline__ void hgl_split_hi_lo(float x, _Float16& hi, _Float16& lo, float& amax) {
  asm volatile("" : "+v"(x));
  amax = fmaxf(amax, fabsf(x));      // a NaN is ignored here (maxNum) and stays a NaN in hi / lo
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}
__device__ __forceinline__ void hgl_split_commit(float amax) {
  if (__builtin_expect(amax > 65504.0f, 0)) atomicAdd(&hgl_tu_split_overflow, 1u);
}
// host-side reader of this translation unit's counter (define once in every .hip that commits)
#define HGL_DEFINE_SPLIT_OVERFLOW_READER(name)                                                        \
  unsigned long long name(int reset) {                                                                \
    unsigned int v = 0;                                                                               \
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(hgl_tu_split_overflow), sizeof(v)) != hipSuccess) return 0; \
    if (reset && v) {                                                                                 \
      const unsigned int z = 0;                                                                       \
      (void)hipMemcpyToSymbol(HIP_SYMBOL(hgl_tu_split_overflow), &z, sizeof(z));                      \
    }                                                                                                 \
    return v;                                                                                         \
  }                                                                                                   \
  int name##_peek(unsigned int* dst, hipStream_t st) {                                                \
    return hipMemcpyFromSymbolAsync(dst, HIP_SYMBOL(hgl_tu_split_overflow), sizeof(unsigned int), 0,  \
                                    hipMemcpyDeviceToHost, st) == hipSuccess ? 0 : -1;                \
  }
// ---- activations of the GEMM epilogues ------------------------------------------------------------------------------
// nn.GELU (erf form; segment_anything/modeling/common.py:13-24 MLPBlock, mask_decoder.py:76-80): 0.5 x (1 + erf(x / sqrt 2)).
// The library erff costs ~40 VALU instructions (branches on |x|); in a GEMM write-out -- where the matrix pipe of that
// workgroup idles -- and in the fused decoder kernels that is the dominant cost (SAM's mlp.lin1: 168 M evaluations per
// block and group of images).  Here: erfc(t) = 2^(-t P(t)) on t = min(|x| / sqrt 2, 4) with a degree-7 minimax fit of P
// (|erf error| <= 1.0e-7 over the whole axis, the size of one float rounding of erf itself; one v_exp_f32, 8 fma); for x < 0
// 1 + erf(x) is erfc(|x|) itself, no subtraction.  |GELU error| <= 2.5e-7 max(1, |x|) (tests/test_gpu_primitives.py).
__device__ __forceinline__ float hgl_gelu_erf(float x) {
  const float t = fminf(fabsf(x) * 0.70710678118654752440f, 4.0f);
  float p = 4.582141628e-05f;
  p = fmaf(p, t, -4.491848231e-04f);
  p = fmaf(p, t, 1.500873244e-03f);
  p = fmaf(p, t, 7.568532601e-04f);
  p = fmaf(p, t, -2.823902667e-02f);
  p = fmaf(p, t, 1.484753788e-01f);
  p = fmaf(p, t, 9.184176326e-01f);
  p = fmaf(p, t, 1.627908468e+00f);
  float q = __builtin_amdgcn_exp2f(-(p * t));   // erfc(t)
  // beyond the clamp erfc(4) = 1.5e-8 would stay: on the negative side that is an error of 7.7e-9 |x| that grows without bound
  // (x = -1e4 -> -7.7e-5 where nn.GELU gives -0); erfc(t) < 2^-24 for t > 4, so 0 is the correctly rounded factor there
  q = fabsf(x) > 5.65685424949238f ? 0.f : q;
  return 0.5f * x * (x < 0.f ? q : 2.0f - q);
}
// QuickGELU (clip/model.py:198-200): x * sigmoid(1.702 x) with the hardware exponential and reciprocal (1 ulp each)
// instead of an IEEE division sequence
__device__ __forceinline__ float hgl_quick_gelu(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
}
#endif  // __HIPCC__ (device helpers: the host-only sources of the library include this header too)

unsigned long long hgl_split_overflow_gemm(int reset);
int hgl_split_overflow_gemm_peek(unsigned int* dst, hipStream_t st);
int hgl_split_overflow_attention_peek(unsigned int* dst, hipStream_t st);
unsigned long long hgl_split_overflow_attention(int reset);

void hgl_set_error(const char* fmt, ...);
int hgl_check_launch(const char* what);  // hipGetLastError -> HGL_ELAUNCH
int hgl_require_device();                // HGL_ENODEVICE when no GPU is visible

#define HGL_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      hgl_set_error(__VA_ARGS__);       \
      return HGL_EINVAL;                \
    }                                   \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the (function, device) pair: remembered per device at every call
// site, the result checked (a process that uses a second GPU would otherwise launch there with the default 64 KiB limit and
// fail in the launch).  KERNEL: a parenthesised function expression; use inside a function that returns an hgl status.
#define HGL_RESERVE_LDS(KERNEL, BYTES, WHAT)                                                                        \
  do {                                                                                                              \
    static std::atomic<bool> hgl_lds_set_[64];       /* zero-initialised; set after the attribute call returned */       \
    int hgl_dev_ = 0;                                                                                               \
    if (hipGetDevice(&hgl_dev_) != hipSuccess || hgl_dev_ < 0 || hgl_dev_ >= 64) hgl_dev_ = 0;                      \
    if (!hgl_lds_set_[hgl_dev_].load(std::memory_order_acquire)) {                                                  \
      if (hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES)) != hipSuccess) { \
        hgl_set_error("%s: cannot reserve %d bytes of LDS", WHAT, (int)(BYTES));                                    \
        return HGL_ELAUNCH;                                                                                         \
      }                                                                                                             \
      hgl_lds_set_[hgl_dev_].store(true, std::memory_order_release);                                                \
    }                                                                                                               \
  } while (0)

#define HGL_TRY(expr)            \
  do {                           \
    int _rc = (expr);            \
    if (_rc != HGL_OK) return _rc; \
  } while (0)

// ---- environment switches
// The product library reads exactly four HGL_* variables, each an A/B switch between two paths that give the same bits and
// each flipped by a -m gpu test (tests/test_abi.py checks the library's strings against this list):
//   HGL_ATTN_PP=0            long unmasked sequences on the tile kernel instead of the ping-pong kernel
//   HGL_X3_TERMS=3           the third split product kept for fp16-valued weights
//   HGL_SAM_POST_SEP=0       the per-pixel post-processing kernel instead of the shared-table one
//   HGL_ATTN_PS_CLIPBLOCKS   which CLIP residual blocks take the pre-split attention (0 never, 1 default, 2 also 197 tokens)
// Everything else that used to be an environment variable (tile-shape experiments, knock-outs for timing) exists only in the
// diagnostic twin `make diag` builds with -DHGL_DIAG (libhybridgl_diag.so, never loaded by the package): there
// HGL_DIAG_SWITCH reads the variable, here it IS its default and the compiler drops the other branch.
const char* hgl_env_str(const char* name);
int hgl_env_int(const char* name, int dflt);
#ifdef HGL_DIAG
#define HGL_DIAG_SWITCH(NAME, DFLT) hgl_env_int(NAME, DFLT)
#else
#define HGL_DIAG_SWITCH(NAME, DFLT) (DFLT)
#endif

static inline size_t hgl_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over the caller-provided workspace.
struct HglArena {
  char* base;
  size_t cap;
  size_t off;
  bool dry;  // size query: no base
  HglArena(void* b, size_t c) : base((char*)b), cap(c), off(0), dry(b == nullptr) {}
  template <typename T>
  T* take(size_t n) {
    size_t bytes = hgl_align_up(n * sizeof(T), 256);
    size_t o = off;
    off += bytes;
    if (dry) return nullptr;
    return (T*)(base + o);
  }
  bool ok() const { return dry || off <= cap; }
};

// ---- internal launchers (each returns HGL_*) -------------------------------
int hgl_launch_gemm(const float* A, const float* W, const float* bias, const float* R, float* C,
                    int M, int N, int K, int lda, int ldw, int ldr, int ldc, int batch,
                    long long sA, long long sW, long long sR, long long sC, int act,
                    hipStream_t st);
int hgl_launch_layernorm(const float* x, const float* w, const float* b, float* y, int rows, int D,
                         float eps, hipStream_t st);
int hgl_launch_attention_split(const float* q, const float* k, const float* v, float* out, void* out_hi, void* out_lo, int B,
                               int H, int Sq, int Sk, int hd, int ldq, int ldk, int ldv, int ldo, long long sqb,
                               long long skb, long long svb, long long sob, float scale, int mask_kind,
                               const uint8_t* keep, int keep_b0, int keep_n, const float* rel_h,
                               const float* rel_w, int kh, int kw, hipStream_t st);
int hgl_launch_attention_win14(const float* q, const float* k, const float* v, void* out_hi, void* out_lo, int B, int H, int hd,
                               int ldq, int ldk, int ldv, int ldo, long long sqb, long long skb, long long svb, long long sob,
                               float scale, const float* Rh, const float* Rw, hipStream_t st);
// attention on the split qkv planes the in-projection GEMM emits (attention_ps.hip); returns 1 when the shape is not served
bool hgl_attention_ps_enabled();
// would hgl_launch_attention_ps serve this call (0 = no)?  For callers that choose the in-projection's output form first
int hgl_attention_ps_serves(long long plane_delta, int ld, int B, int H, int S, int hd, int mask_kind, int rel_kh, int rel_kw,
                            const float* tab_h, const float* tab_w);
int hgl_launch_attention_ps(const void* qkv_hi, const void* qkv_lo, int ld, int qcol, int kcol, int vcol, long long sb, int B,
                            int H, int S, int hd, float* out, void* out_hi, void* out_lo, int ldo, long long sob, float scale,
                            int mask_kind, const uint8_t* keep, int keep_b0, int keep_n, const float* rel_h, const float* rel_w,
                            int kh, int kw, const float* tab_h, const float* tab_w, hipStream_t st);
int hgl_launch_attention_smallk(const float* q, const float* k, const float* v, float* out, void* out_hi, void* out_lo,
                                int B, int H, int Sq, int Sk, int hd, int ldq, int ldk, int ldv, int ldo, long long sqb,
                                long long skb, long long svb, long long sob, float scale, hipStream_t st);
size_t hgl_attention_fewq_part_bytes(int B, int Sk);
int hgl_launch_attention_fewq_chunked(const float* q, const float* k, const float* v, float* out, int B, int H, int Sq, int Sk,
                                      int hd, int ldq, int ldk, int ldv, int ldo, long long sqb, long long skb, long long svb,
                                      long long sob, float scale, float* part, size_t part_bytes, hipStream_t st);
int hgl_launch_attention(const float* q, const float* k, const float* v, float* out, int B, int H,
                         int Sq, int Sk, int hd, int ldq, int ldk, int ldv, int ldo, long long sqb,
                         long long skb, long long svb, long long sob, float scale, int mask_kind,
                         const uint8_t* keep, int keep_b0, int keep_n, const float* rel_h,
                         const float* rel_w, int kh, int kw, hipStream_t st);

// CLIP glue (clip_glue.hip)
int hgl_launch_im2col_patch(const float* img, int N, int res, int patch, float* cols, hipStream_t st);
int hgl_launch_im2col_patch_split(const float* img, int N, int res, int patch, void* hi, void* lo, hipStream_t st);
int hgl_launch_assemble_lnpre(const float* tok, const float* cls, const float* pos, const float* lw,
                              const float* lb, float* x, int B, int S, int D, hipStream_t st);
int hgl_launch_mask_resize(const uint8_t* masks, int N, int Hm, int Wm, int g, float* pm,
                           uint8_t* keep, hipStream_t st);
// out[n,s,:] = a[n,s,:]*ca + cb * b[n,s,:] * (s==0 ? 1 : pm[n,s-1]) ; pm may be null, a may be null
int hgl_launch_mix(float* out, const float* a, float ca, const float* b, float cb, const float* pm,
                   int N, int S, int D, hipStream_t st);
int hgl_launch_gather_rows(const float* x, long long row_stride, int rows, int D, float* y,
                           hipStream_t st);
int hgl_launch_add_inplace(float* y, const float* x, long long n, hipStream_t st);
int hgl_launch_text_embed(const int32_t* tokens, const float* emb, const float* pos, float* x, int B,
                          int S, int ctx, int D, int vocab, int32_t* eot, hipStream_t st);
int hgl_launch_gather_eot(const float* x, const int32_t* eot, int B, int S, int D, float* y,
                          hipStream_t st);
int hgl_launch_zero_positions(float* x, int B, int S, int D, const int32_t* pos, int n, hipStream_t st);

// CLIP transformer pieces shared by clip_api.hip and gem_api.hip
struct HglBlockBufs {
  float* H;    // [M, D]   LN output / attention output
  float* QKV;  // [M, 3D]
  float* F;    // [M, 4D]
};
int hgl_clip_run_block(const HglResBlockW& w, float* X, int B, int S, int D, int heads, const HglBlockBufs& bf,
                       int mask_kind, const uint8_t* keep, int keep_b0, int keep_n, hipStream_t st);
bool hgl_clip_block_uses_x3(const HglResBlockW& w, int M, int D);
bool hgl_clip_block_presplit(const HglResBlockW& w, int B, int S, int D, int heads, int mask_kind);
int hgl_clip_block_qkv(const HglResBlockW& w, const float* X, int M, int D, const HglBlockBufs& bf, hipStream_t st,
                       bool split_out = false);
int hgl_clip_block_rest(const HglResBlockW& w, float* X, int B, int S, int D, int heads, const HglBlockBufs& bf,
                        int mask_kind, const uint8_t* keep, int keep_b0, int keep_n, hipStream_t st, bool qkv_split = false);
int hgl_clip_embed_images(const HglClipVisionW* w, const float* imgs, int n_img, float* X, float* cols, float* tok,
                          hipStream_t st);

// ---- optional per-kernel-class timing with HIP events on the launch stream (bench roofline) ----
enum HglProfClass { HGL_PROF_GEMM = 0, HGL_PROF_ATTN = 1, HGL_PROF_OTHER = 2, HGL_PROF_GEMM_X3 = 3, HGL_PROF_GEMM_X3G = 4, HGL_PROF_GEMM_X3_FEW = 5, HGL_PROF_NCLASS = 6 };
struct HglProfScope {
  int slot;
  hipStream_t st;
  HglProfScope(int cls, double flops, double bytes, hipStream_t s);
  ~HglProfScope();
};

// SAM glue (sam_glue.hip)
int hgl_launch_sam_preprocess(const uint8_t* img, int h, int w, int S, float* out, hipStream_t st);
int hgl_launch_win_partition(const float* H, int g, int ws, int nw, int D, float* Hw, hipStream_t st);
int hgl_launch_win_unpartition_add(float* X, int g, int ws, int nw, int D, const float* P, hipStream_t st);
int hgl_launch_relpos_gather(const float* T, int B, int heads, int S, int size, int L, int use_w,
                             float* rel, hipStream_t st);
int hgl_launch_relpos_direct(const float* qkv, int ldq, int B, int heads, int S, int size, int hd,
                             const float* Rh, const float* Rw, float* rel_h, float* rel_w, hipStream_t st);
int hgl_launch_relpos_split(const void* q_hi, const void* q_lo, int ldq, int B, int heads, int S, int size, int hd,
                            const float* Rh, const float* Rw, float* rel_h, float* rel_w, hipStream_t st);
int hgl_launch_im2col3x3(const float* in, int g, int C, float* cols, hipStream_t st);
int hgl_launch_add_rows_bcast(const float* a, long long a_bstride, const float* pe, long long rows_elems,
                              int B, float* out, hipStream_t st);
int hgl_launch_pe(const float* coords01, const float* G, int n, int F, int mode, const float* pos_embed,
                  const float* not_a_point, float* out, hipStream_t st);
int hgl_launch_build_tokens(const float* iou_tok, const float* mask_tok, const float* sparse, int P, int C, int T,
                            float* tokens, hipStream_t st);
int hgl_launch_mask_downscaling(const float* in, int P, int g, const float* c1w, const float* c1b, const float* n1w, const float* n1b,
                                const float* c2w, const float* c2b, const float* n2w, const float* n2b, const float* c3w,
                                const float* c3b, float* out, hipStream_t st);
int hgl_launch_win_maps(int g, int ws, int nw, int nb, int* pad_of, int* tok_of, int* pad_list, int* pad_count, hipStream_t st);
int hgl_launch_fill_rows(float* dst, int ld, const int* rows, const int* nrows, int max_rows, const float* v, int N,
                         hipStream_t st);
int hgl_launch_fill_rows_split(void* hi, void* lo, int ld, const int* rows, const int* nrows, int max_rows, const float* v,
                               int N, hipStream_t st);
int hgl_launch_ln_gelu64(float* x, const float* w, const float* b, long long rows, float eps, void* hi, void* lo,
                         hipStream_t st);
int hgl_launch_ln256_pe_split(float* x, const float* w, const float* b, const float* pe, int pe_rows, long long rows,
                              float eps, int write_f32, void* kh, void* kl, void* ph, void* pl, hipStream_t st);
int hgl_launch_hyper_logits(const float* u2, const float* hyper, int P, int g, int row0, float* low_res, hipStream_t st);
int hgl_launch_pe_labeled(const float* coords01, const int32_t* labels, const float* G, int n, int F, const float* not_a_point,
                          const float* const* point_embed, float* out, hipStream_t st);
// fused decoder stages (sam_decoder_fused.hip)
// token -> image attention of the mask decoder on the raw image-token planes (sam_decoder_t2i.hip)
int hgl_launch_t2i_fold_q(const float* q1, const float* Wk, float scale, float* Qk, int P, hipStream_t st);
// ns = hgl_t2i_key_ranges(P, HW): 1 -> out = the attended rows [P*56, 256]; 8 (prompt batches of <= 128) -> out = key-range partials
// of hgl_t2i_part_bytes(P, HW) bytes, which hgl_launch_t2i_unfold_v (same ns) puts together
int hgl_t2i_key_ranges(int P, int HW);
size_t hgl_t2i_part_bytes(int P, int HW);
int hgl_launch_t2i_raw_attn(const void* Qh, const void* Ql, const float* bias, const void* Kh, const void* Kl, int P, int HW,
                            float* out, int ns, hipStream_t st);
int hgl_launch_i2t_prep(const float* k1, const float* v1, const float* Wq, const float* bq, const float* Wo, float scale, void* Kh,
                        void* Kl, float* cb, void* Uh, void* Ul, int P, hipStream_t st);
unsigned long long hgl_split_overflow_decoder(int reset);
int hgl_launch_dec_i2t_fold(const void* Xh, const void* Xl, const void* Kh, const void* Kl, const float* pek, const float* cb,
                            const void* Uh, const void* Ul, const float* bo, const float* ln_w, const float* ln_b, float eps, int P,
                            int HW, void* out_hi, void* out_lo, hipStream_t st);
int hgl_launch_t2i_unfold_v(const float* A, int ns, const float* Wv, const float* bv, float* att, int P, hipStream_t st);
int hgl_launch_dec_tail(const void* src_hi, const void* src_lo, const float* up0_w, const float* up0_b, const float* ln_w,
                        const float* ln_b, const float* up3_w, const float* up3_b, const float* hyper, int row0, int P, int g,
                        float eps, float* low_res, const uint8_t* skip, hipStream_t st);
int hgl_launch_dec_i2t(const float* q, int ldq, long long q_bstride, const float* k1, const float* v1, const float* out_w,
                       const float* out_b, const float* R, long long r_bstride, const float* ln_w, const float* ln_b, float eps,
                       float scale, int P, int HW, float* out32, void* out_hi, void* out_lo, hipStream_t st);

// split-fp16 GEMM path (gemm_f16x3.hip)
int hgl_precision();
bool hgl_has_split_weight(const float* W);
bool hgl_get_split_weight(const float* W, const void** hi, const void** lo, int* scale_log2, int* N, int* K);
int hgl_launch_split_f16(const float* x, float scale, void* hi, void* lo, long long n, hipStream_t st);
int hgl_launch_layernorm_split(const float* x, const float* w, const float* b, void* hi, void* lo, int rows, int D,
                               float eps, hipStream_t st);
int hgl_launch_layernorm_split_maps(const float* x, const float* w, const float* b, void* hi, void* lo, int rows, int D,
                                    float eps, const int* smap, const int* dmap, hipStream_t st);
bool hgl_gemm_skinny_applicable(const float* W32, int M, int N, int K, int lda, int ldw, int batch, int max_m = 1024);
int hgl_launch_gemm_x3_skinny(const float* A, int lda, const float* W32, const float* bias, const float* R, int ldr,
                              float* C, int ldc, int M, int N, int K, int act, hipStream_t st);
int hgl_gemm_f16x3_splitk_factor(int M, int N, int K);
int hgl_launch_gemm_f16x3_splitk(const void* Ah, const void* Al, int lda, const int* amap, const float* W32, const float* bias,
                                 const float* R, int ldr, const int* cmap, float* C, int ldc, int M, int N, int K, int act,
                                 int ksplit, float* part, size_t part_bytes, hipStream_t st);
// row-balanced launch: whole rounds of the persistent tiling + a split-K tail (gemm_f16x3.hip); `part`: scratch for the tail's partial sums
int hgl_launch_gemm_f16x3_balanced(const void* Ah, const void* Al, int lda, const int* amap, const float* W32, const float* bias,
                                   const float* R, int ldr, const int* cmap, float* C, int ldc, int M, int N, int K, int act,
                                   float* part, size_t part_bytes, hipStream_t st);
int hgl_launch_gemm_f16x3_maps(const void* Ah, const void* Al, int lda, const int* amap, const float* W32, const float* bias,
                               const float* R, int ldr, int rmod, const int* cmap, float* C, void* Ch, void* Cl, int ldc,
                               int M, int N, int K, int act, hipStream_t st);
int hgl_launch_gemm_f16x3_rmod(const void* Ah, const void* Al, int lda, const float* W32, const float* bias, const float* R,
                               int ldr, int rmod, float* C, void* Ch, void* Cl, int ldc, int M, int N, int K, int act, hipStream_t st);
int hgl_launch_gemm_f16x3(const void* Ah, const void* Al, int lda, const float* W32, const float* bias, const float* R,
                          int ldr, float* C, void* Ch, void* Cl, int ldc, int M, int N, int K, int act, hipStream_t st);
int hgl_launch_win_partition_split(const float* H, int g, int ws, int nw, int D, void* hi, void* lo, hipStream_t st);
// true when the split-fp16 path applies to a GEMM with this weight and reduction length
static inline bool hgl_use_x3(const float* W, int K) { return hgl_precision() == HGL_PREC_F16X3 && (K % 64) == 0 && hgl_has_split_weight(W); }
