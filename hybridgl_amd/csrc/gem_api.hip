// Text-conditioned heat-map (the "GEM" call of Hybridgl_main.py:36-39,200-201) behind the C ABI.
//
//   hgl_gem_image_features == GEMViT.forward of gem_torch 1.0.1 (external package: the published algorithm is
//                             restated, see oracle/gem_oracle.py -- parity unpinned)
//   hgl_gem_heatmap        == GEMWrapper.forward after the two encoders: cosine matching, bilinear up-sampling,
//                             min-max
//   hgl_resize_bilinear_aa == T.Resize((h, w), antialias=True) on a float tensor (Hybridgl_main.py:201)
//
// The image tower is the CLIP ViT of clip_api.hip at a 28x28 grid; the last gem_blocks blocks carry a second
// residual stream fed by self-self attention: softmax(n(x) n(x)^T * t) applied to x in {v, k, q} (n = per-head
// L2 normalisation, t = mean token norm of ln_1(x) / sqrt(head_dim)), iterated, the final assignment taken on v,
// the three results averaged and passed through the block's out-projection.  Each self-self attention is the
// library's flash attention kernel with q = n(x)*t, k = n(x) and the three projections stacked along the batch.
#include "hgl_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ||row|| of H [S, D]: one wave per row
__global__ __launch_bounds__(256) void row_norm_kernel(const float* __restrict__ H, int S, int D, float* __restrict__ norms) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= S) return;
  const float* r = H + (long long)row * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += r[i] * r[i];
  s = wave_sum(s);
  if (lane == 0) norms[row] = sqrtf(s);
}

// the same from the fp16 (hi, lo) pair of the split path: x = hi + lo
__global__ __launch_bounds__(256) void row_norm_split_kernel(const _Float16* __restrict__ Hh, const _Float16* __restrict__ Hl, int S,
                                                             int D, float* __restrict__ norms) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= S) return;
  const _Float16* h = Hh + (long long)row * D;
  const _Float16* l = Hl + (long long)row * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) {
    const float x = (float)h[i] + (float)l[i];
    s += x * x;
  }
  s = wave_sum(s);
  if (lane == 0) norms[row] = sqrtf(s);
}

// t[img] = mean(norms of image img) * scale  (fixed order: one work-group per image, strided partials, tree in LDS)
__global__ __launch_bounds__(256) void temp_kernel(const float* __restrict__ norms, int S, float scale, float* __restrict__ t) {
  __shared__ float part[256];
  const float* nr = norms + (long long)blockIdx.x * S;
  float s = 0.f;
  for (int i = threadIdx.x; i < S; i += 256) s += nr[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) t[blockIdx.x] = part[0] / (float)S * scale;
}

__global__ void set_scalar_kernel(float* t, float v, int n) { if ((int)threadIdx.x < n) t[threadIdx.x] = v; }

// per (set, row, head): n = x / max(||x||, 1e-12) -> nrm, n * t[row / S] -> nrm_s.   src set j at src + j*set_stride,
// row stride ld; `rows` = images * S token rows; outputs [3, rows, heads*hd] contiguous.  One 16-lane group per item.
__global__ __launch_bounds__(256) void normalize_heads_kernel(const float* __restrict__ src, long long set_stride, int ld,
                                                              int rows, int S, int heads, int hd, const float* __restrict__ t,
                                                              float* __restrict__ nrm, float* __restrict__ nrm_s) {
  const int sub = threadIdx.x & 15;
  const long long item = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const long long total = 3LL * rows * heads;
  if (item >= total) return;
  const int h = (int)(item % heads);
  const long long rs = item / heads;
  const int s = (int)(rs % rows);
  const int set = (int)(rs / rows);
  const float* x = src + set * set_stride + (long long)s * ld + h * hd;
  float q = 0.f;
  for (int i = sub; i < hd; i += 16) q += x[i] * x[i];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float inv = 1.0f / fmaxf(sqrtf(q), 1e-12f);
  const float tt = t[s / S];
  const long long D = (long long)heads * hd;
  float* o1 = nrm + ((long long)set * rows + s) * D + h * hd;
  float* o2 = nrm_s + ((long long)set * rows + s) * D + h * hd;
  for (int i = sub; i < hd; i += 16) {
    const float n = x[i] * inv;
    o1[i] = n;
    o2[i] = n * tt;
  }
}

// out = (a0 + a1 + a2) / 3 over [n] with the three sets `stride` apart
__global__ __launch_bounds__(256) void mean3_kernel(const float* __restrict__ a, long long stride, long long n, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (a[i] + a[i + stride] + a[i + 2 * stride]) / 3.0f;
}

// the same, written as the fp16 (hi, lo) pair the split out-projection reads
__global__ __launch_bounds__(256) void mean3_split_kernel(const float* __restrict__ a, long long stride, long long n,
                                                          _Float16* __restrict__ hi, _Float16* __restrict__ lo) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float m = (a[i] + a[i + stride] + a[i + 2 * stride]) / 3.0f;
    _Float16 h, l;
    hgl_split_hi_lo(m, h, l);
    hi[i] = h;
    lo[i] = l;
  }
}

// logits[t, p] = 100 * <f_p / |f_p|, x_t / |x_t|>; feat rows 1..P (row 0 = CLS). One wave per (p, t).
__global__ __launch_bounds__(256) void gem_logits_kernel(const float* __restrict__ feat, const float* __restrict__ text, int P,
                                                         int E, int T, float* __restrict__ logits) {
  const int lane = threadIdx.x & 63;
  const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (long long)P * T) return;
  const int p = (int)(item % P), t = (int)(item / P);
  const float* f = feat + (long long)(p + 1) * E;
  const float* x = text + (long long)t * E;
  float ff = 0.f, xx = 0.f;
  for (int i = lane; i < E; i += 64) {
    ff += f[i] * f[i];
    xx += x[i] * x[i];
  }
  const float fi = 1.0f / fmaxf(sqrtf(wave_sum(ff)), 1e-12f);
  const float xi = 1.0f / fmaxf(sqrtf(wave_sum(xx)), 1e-12f);
  float d = 0.f;
  for (int i = lane; i < E; i += 64) d += (f[i] * fi) * (x[i] * xi);
  d = wave_sum(d);
  if (lane == 0) logits[(long long)t * P + p] = 100.0f * d;
}

// F.interpolate(bilinear, align_corners=False) of [T, g, g] -> [T, R, R] + per-block min / max
__global__ __launch_bounds__(256) void upsample_minmax_kernel(const float* __restrict__ lo, int g, int R, float scale,
                                                              float* __restrict__ up, float* __restrict__ part) {
  const int t = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float v = 0.f;
  const bool live = i < R * R;
  if (live) {
    const int Y = i / R, X = i % R;
    const float sy = fmaxf(fmaf(scale, (float)Y + 0.5f, -0.5f), 0.f);
    const float sx = fmaxf(fmaf(scale, (float)X + 0.5f, -0.5f), 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < g - 1), x1 = x0 + (x0 < g - 1);
    const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
    const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const float* m = lo + (long long)t * g * g;
    v = ly0 * (lx0 * m[y0 * g + x0] + lx1 * m[y0 * g + x1]) + ly1 * (lx0 * m[y1 * g + x0] + lx1 * m[y1 * g + x1]);
    up[(long long)t * R * R + i] = v;
  }
  float mn = live ? v : INFINITY, mx = live ? v : -INFINITY;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o));
    mx = fmaxf(mx, __shfl_xor(mx, o));
  }
  __shared__ float smn[4], smx[4];
  if ((threadIdx.x & 63) == 0) {
    smn[threadIdx.x >> 6] = mn;
    smx[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* p = part + ((long long)t * gridDim.x + blockIdx.x) * 2;
    p[0] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    p[1] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
  }
}

// (v - min) / (max - min) in place; every block folds the per-block partials of its map (min / max are exact in any order)
__global__ __launch_bounds__(256) void minmax_apply_kernel(float* __restrict__ up, int RR, const float* __restrict__ part, int nblk) {
  const int t = blockIdx.y;
  float mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < nblk; i += 256) {
    mn = fminf(mn, part[((long long)t * nblk + i) * 2]);
    mx = fmaxf(mx, part[((long long)t * nblk + i) * 2 + 1]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o));
    mx = fmaxf(mx, __shfl_xor(mx, o));
  }
  __shared__ float smn[4], smx[4];
  if ((threadIdx.x & 63) == 0) {
    smn[threadIdx.x >> 6] = mn;
    smx[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
  mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < RR) {
    float* p = up + (long long)t * RR + i;
    *p = (*p - mn) / (mx - mn);
  }
}

// F.interpolate(bilinear, align_corners=False, antialias=False) of [C, h, w] -> [C, H, W]: what T.Resize does to a TENSOR
// in torchvision 0.15 (Hybridgl_main_PhraseCut.py:69-70 resizes the normalised image to the annotation's size this way).
// Source index as ATen's single fma, the four products and three sums rounded one by one in the order of ATen's CPU
// kernel (rows first, then columns) -- no contraction into fmas.
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, int h, int w, float* __restrict__ out,
                                                              int H, int W, float sy_scale, float sx_scale) {
  const int c = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  const int Y = i / W, X = i % W;
  const float sy = fmaxf(fmaf(sy_scale, (float)Y + 0.5f, -0.5f), 0.f);
  const float sx = fmaxf(fmaf(sx_scale, (float)X + 0.5f, -0.5f), 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < h - 1), x1 = x0 + (x0 < w - 1);
  const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
  const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
  const float* m = in + (long long)c * h * w;
  const float t = __fadd_rn(__fmul_rn(m[(long long)y0 * w + x0], lx0), __fmul_rn(m[(long long)y0 * w + x1], lx1));
  const float b = __fadd_rn(__fmul_rn(m[(long long)y1 * w + x0], lx0), __fmul_rn(m[(long long)y1 * w + x1], lx1));
  out[(long long)c * H * W + i] = __fadd_rn(__fmul_rn(t, ly0), __fmul_rn(b, ly1));
}

// ATen _upsample_bilinear2d_aa (align_corners=False): triangle filter of support max(scale, 1), weights
// normalised per output index; out[c, Y, X] = sum_y wy * (sum_x wx * in[c, y, x])
struct AaSpan {
  int lo, n;
  float center, invscale, total;
};
__device__ __forceinline__ AaSpan aa_span(int o, float scale, int in_size) {
  AaSpan s;
  const float support = scale >= 1.0f ? scale : 1.0f;
  s.invscale = scale >= 1.0f ? 1.0f / scale : 1.0f;
  s.center = scale * ((float)o + 0.5f);
  s.lo = max((int)(s.center - support + 0.5f), 0);
  s.n = min((int)(s.center + support + 0.5f), in_size) - s.lo;
  float tot = 0.f;
  for (int j = 0; j < s.n; ++j) tot += fmaxf(0.f, 1.0f - fabsf(((float)(j + s.lo) - s.center + 0.5f) * s.invscale));
  s.total = tot;
  return s;
}
__device__ __forceinline__ float aa_w(const AaSpan& s, int j) {
  const float w = fmaxf(0.f, 1.0f - fabsf(((float)(j + s.lo) - s.center + 0.5f) * s.invscale));
  return s.total != 0.f ? w / s.total : w;
}

__global__ __launch_bounds__(256) void resize_aa_kernel(const float* __restrict__ in, int h, int w, float* __restrict__ out, int H,
                                                        int W, float sh, float sw) {
  const int c = blockIdx.y;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)H * W) return;
  const int Y = (int)(i / W), X = (int)(i % W);
  const AaSpan ys = aa_span(Y, sh, h), xs = aa_span(X, sw, w);
  const float* src = in + (long long)c * h * w;
  float acc = 0.f;
  for (int a = 0; a < ys.n; ++a) {
    const float* row = src + (long long)(ys.lo + a) * w + xs.lo;
    float r = 0.f;
    for (int b = 0; b < xs.n; ++b) r += aa_w(xs, b) * row[b];
    acc += aa_w(ys, a) * r;
  }
  out[(long long)c * H * W + i] = acc;
}

struct GemPlan {
  float *X, *Xg, *H, *QKV, *F, *N3, *N3s, *X1, *norms, *t;
  int nb;   // images stacked along the token rows
};

bool carve(HglArena& ar, const HglClipVisionW* w, int nb, GemPlan& p) {
  const size_t S = (size_t)w->grid * w->grid + 1, D = w->width;
  const size_t act = (size_t)nb * S * D;
  p.nb = nb;
  p.X = ar.take<float>(act);
  p.Xg = ar.take<float>(act);
  p.H = ar.take<float>(act);
  p.QKV = ar.take<float>(3 * act);
  const size_t cols = (size_t)nb * (S - 1) * 3 * w->patch * w->patch;
  p.F = ar.take<float>(4 * act > cols ? 4 * act : cols);
  p.N3 = ar.take<float>(3 * act);
  p.N3s = ar.take<float>(3 * act);
  p.X1 = ar.take<float>(3 * act);
  p.norms = ar.take<float>((size_t)nb * S);
  p.t = ar.take<float>(64);
  return ar.ok();
}

int normalize_heads(const float* src, long long set_stride, int ld, int rows, int S, int heads, int hd, const GemPlan& p,
                    hipStream_t st) {
  const long long items = 3LL * rows * heads;
  hipLaunchKernelGGL(normalize_heads_kernel, dim3((unsigned)((items + 15) / 16)), dim3(256), 0, st, src, set_stride, ld, rows, S,
                     heads, hd, p.t, p.N3, p.N3s);
  return hgl_check_launch("gem_normalize_heads");
}

// One GEM block over the stacked token rows of p.nb images: the gem stream gets proj(self-self attention(ln_1 x)); the
// original stream is the plain block.
int gem_block(const HglResBlockW& w, const GemPlan& p, int S, int D, int heads, int ss_iter, float ss_temp, bool need_ori,
              hipStream_t st) {
  const int hd = D / heads, nb = p.nb, M = nb * S;
  const float scale = 1.0f / sqrtf((float)hd);
  const long long SD = (long long)S * D, MD = (long long)M * D;
  const HglBlockBufs bf{p.H, p.QKV, p.F};
  const bool x3 = hgl_clip_block_uses_x3(w, M, D);
  HGL_TRY(hgl_clip_block_qkv(w, p.X, M, D, bf, st));          // p.H = ln_1(x) (fp32 or the hi+lo pair), p.QKV
  if (ss_temp > 0.f) {
    hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(64), 0, st, p.t, ss_temp, nb);
  } else {
    if (x3)
      hipLaunchKernelGGL(row_norm_split_kernel, dim3((M + 3) / 4), dim3(256), 0, st, (const _Float16*)p.H,
                         (const _Float16*)p.H + MD, M, D, p.norms);
    else
      hipLaunchKernelGGL(row_norm_kernel, dim3((M + 3) / 4), dim3(256), 0, st, p.H, M, D, p.norms);
    hipLaunchKernelGGL(temp_kernel, dim3(nb), dim3(256), 0, st, p.norms, S, scale, p.t);   // one temperature per image
  }
  HGL_TRY(hgl_check_launch("gem_temperature"));
  // sets in the order (v, k, q): QKV + 2D, + D, + 0.  Attention batches = (set, image): 3 * nb sequences of S tokens.
  const float* V = p.QKV + 2 * D;
  HGL_TRY(normalize_heads(V, -(long long)D, 3 * D, M, S, heads, hd, p, st));
  for (int it = 0; it < ss_iter; ++it) {
    HGL_TRY(hgl_launch_attention(p.N3s, p.N3, p.N3, p.X1, 3 * nb, heads, S, S, hd, D, D, D, D, SD, SD, SD, SD, 1.0f,
                                 HGL_MASK_NONE, nullptr, 0, 0, nullptr, nullptr, 0, 0, st));
    HGL_TRY(normalize_heads(p.X1, MD, D, M, S, heads, hd, p, st));
  }
  // assignment to v: the value operand is the block's v for all three sets -- one launch when a batch stride of 0
  // expresses that (one image), else one launch per set over the images
  if (nb == 1) {
    HGL_TRY(hgl_launch_attention(p.N3s, p.N3, V, p.X1, 3, heads, S, S, hd, D, D, 3 * D, D, SD, SD, 0, SD, 1.0f, HGL_MASK_NONE,
                                 nullptr, 0, 0, nullptr, nullptr, 0, 0, st));
  } else {
    for (int set = 0; set < 3; ++set)
      HGL_TRY(hgl_launch_attention(p.N3s + set * MD, p.N3 + set * MD, V, p.X1 + set * MD, nb, heads, S, S, hd, D, D, 3 * D, D,
                                   SD, SD, 3 * SD, SD, 1.0f, HGL_MASK_NONE, nullptr, 0, 0, nullptr, nullptr, 0, 0, st));
  }
  if (x3) {
    _Float16* Mh = (_Float16*)p.N3;
    _Float16* Ml = Mh + MD;
    hipLaunchKernelGGL(mean3_split_kernel, dim3((unsigned)((MD + 255) / 256)), dim3(256), 0, st, p.X1, MD, MD, Mh, Ml);
    HGL_TRY(hgl_check_launch("gem_mean3"));
    HGL_TRY(hgl_launch_gemm_f16x3(Mh, Ml, D, w.out_proj_w, w.out_proj_b, p.Xg, D, p.Xg, nullptr, nullptr, D, M, D, D,
                                  HGL_ACT_NONE, st));
  } else {
    hipLaunchKernelGGL(mean3_kernel, dim3((unsigned)((MD + 255) / 256)), dim3(256), 0, st, p.X1, MD, MD, p.N3);
    HGL_TRY(hgl_check_launch("gem_mean3"));
    HGL_TRY(hgl_launch_gemm(p.N3, w.out_proj_w, w.out_proj_b, p.Xg, p.Xg, M, D, D, D, D, D, D, 1, 0, 0, 0, 0, HGL_ACT_NONE, st));
  }
  if (!need_ori) return HGL_OK;
  // original stream (clip/model.py:244-257): the plain block on the same QKV
  return hgl_clip_block_rest(w, p.X, nb, S, D, heads, bf, HGL_MASK_NONE, nullptr, 0, 0, st);
}

bool valid_vision(const HglClipVisionW* w) {
  return w && w->width > 0 && w->layers > 0 && w->heads > 0 && w->patch > 0 && w->grid > 0 && w->embed > 0 &&
         w->width % w->heads == 0 && (w->width & 3) == 0 && (w->embed & 3) == 0 && w->conv1_w && w->class_embedding &&
         w->positional_embedding && w->ln_pre_w && w->ln_pre_b && w->blocks && w->ln_post_w && w->ln_post_b && w->proj_t;
}

}  // namespace

extern "C" {

size_t hgl_gem_batch_workspace_bytes(const HglClipVisionW* w, int nb) {
  if (!valid_vision(w) || nb < 1) return 0;
  HglArena ar(nullptr, 0);
  GemPlan p;
  carve(ar, w, nb, p);
  return ar.off;
}

size_t hgl_gem_workspace_bytes(const HglClipVisionW* w) { return hgl_gem_batch_workspace_bytes(w, 1); }

int hgl_gem_image_features_batch(const HglClipVisionW* w, const float* imgs, int nb, int gem_blocks, int ss_attn_iter,
                                 float ss_attn_temp, float* feat_gem, float* feat_ori, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(valid_vision(w), "gem_image_features: invalid weight struct");
  HGL_REQUIRE(imgs && feat_gem && nb >= 1 && nb <= 64, "gem_image_features: bad arguments (nb %d)", nb);
  HGL_REQUIRE(gem_blocks >= 0 && gem_blocks <= w->layers, "gem_image_features: gem_blocks %d outside 0..%d", gem_blocks, w->layers);
  HGL_REQUIRE(ss_attn_iter >= 0 && ss_attn_iter <= 16, "gem_image_features: bad ss_attn_iter %d", ss_attn_iter);
  HglArena ar(workspace, workspace_bytes);
  GemPlan p;
  if (!workspace || !carve(ar, w, nb, p)) {
    hgl_set_error("gem_image_features: workspace too small (%zu bytes given)", workspace_bytes);
    return HGL_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int D = w->width, S = w->grid * w->grid + 1, heads = w->heads, E = w->embed, M = nb * S;
  HglBlockBufs bf{p.H, p.QKV, p.F};
  HGL_TRY(hgl_clip_embed_images(w, imgs, nb, p.X, p.F, p.QKV, st));
  const int first_gem = w->layers - gem_blocks;
  for (int l = 0; l < first_gem; ++l)
    HGL_TRY(hgl_clip_run_block(w->blocks[l], p.X, nb, S, D, heads, bf, HGL_MASK_NONE, nullptr, 0, 0, st));
  const float* gem_stream = p.X;
  if (gem_blocks > 0) {
    (void)hipMemcpyAsync(p.Xg, p.X, sizeof(float) * (size_t)M * D, hipMemcpyDeviceToDevice, st);
    gem_stream = p.Xg;
    for (int l = first_gem; l < w->layers; ++l) {
      // the original stream of the last block only feeds feat_ori
      const bool need_ori = l + 1 < w->layers || feat_ori != nullptr;
      HGL_TRY(gem_block(w->blocks[l], p, S, D, heads, ss_attn_iter, ss_attn_temp, need_ori, st));
    }
  }
  // ln_post + proj on every token of the stream(s)
  HGL_TRY(hgl_launch_layernorm(gem_stream, w->ln_post_w, w->ln_post_b, p.H, M, D, 1e-5f, st));
  HGL_TRY(hgl_launch_gemm(p.H, w->proj_t, nullptr, nullptr, feat_gem, M, E, D, D, D, 0, E, 1, 0, 0, 0, 0, HGL_ACT_NONE, st));
  if (feat_ori) {
    HGL_TRY(hgl_launch_layernorm(p.X, w->ln_post_w, w->ln_post_b, p.H, M, D, 1e-5f, st));
    HGL_TRY(hgl_launch_gemm(p.H, w->proj_t, nullptr, nullptr, feat_ori, M, E, D, D, D, 0, E, 1, 0, 0, 0, 0, HGL_ACT_NONE, st));
  }
  return HGL_OK;
}

int hgl_gem_image_features(const HglClipVisionW* w, const float* img, int gem_blocks, int ss_attn_iter, float ss_attn_temp,
                           float* feat_gem, float* feat_ori, void* workspace, size_t workspace_bytes, void* stream) {
  return hgl_gem_image_features_batch(w, img, 1, gem_blocks, ss_attn_iter, ss_attn_temp, feat_gem, feat_ori, workspace,
                                      workspace_bytes, stream);
}

size_t hgl_gem_heatmap_workspace_bytes(int grid, int T, int res) {
  if (grid <= 0 || T <= 0 || res <= 0) return 0;
  const size_t nblk = ((size_t)res * res + 255) / 256;
  return hgl_align_up((size_t)T * grid * grid * sizeof(float), 256) + hgl_align_up((size_t)T * nblk * 2 * sizeof(float), 256);
}

int hgl_gem_heatmap(const float* feat, int grid, int E, const float* text, int T, int res, int normalize, float* heat,
                    void* workspace, size_t workspace_bytes, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(feat && text && heat, "gem_heatmap: null operand");
  HGL_REQUIRE(grid > 0 && E > 0 && T > 0 && T <= 65535 && res > 0, "gem_heatmap: bad shape (grid %d, E %d, T %d, res %d)", grid, E, T, res);
  if (!workspace || workspace_bytes < hgl_gem_heatmap_workspace_bytes(grid, T, res)) {
    hgl_set_error("gem_heatmap: workspace too small (%zu bytes given)", workspace_bytes);
    return HGL_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int P = grid * grid;
  float* logits = (float*)workspace;
  float* part = (float*)((char*)workspace + hgl_align_up((size_t)T * P * sizeof(float), 256));
  const long long items = (long long)P * T;
  hipLaunchKernelGGL(gem_logits_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, feat, text, P, E, T, logits);
  HGL_TRY(hgl_check_launch("gem_logits"));
  const int nblk = (res * res + 255) / 256;
  hipLaunchKernelGGL(upsample_minmax_kernel, dim3(nblk, T), dim3(256), 0, st, logits, grid, res, (float)grid / (float)res, heat,
                     part);
  HGL_TRY(hgl_check_launch("gem_upsample"));
  if (normalize) {
    hipLaunchKernelGGL(minmax_apply_kernel, dim3(nblk, T), dim3(256), 0, st, heat, res * res, part, nblk);
    HGL_TRY(hgl_check_launch("gem_minmax"));
  }
  return HGL_OK;
}

int hgl_resize_bilinear(const float* in, int C, int h, int w, float* out, int H, int W, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(in && out, "resize_bilinear: null operand");
  HGL_REQUIRE(C > 0 && C <= 65535 && h > 0 && w > 0 && H > 0 && W > 0, "resize_bilinear: bad shape");
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)(((long long)H * W + 255) / 256), (unsigned)C), dim3(256), 0,
                     (hipStream_t)stream, in, h, w, out, H, W, (float)h / (float)H, (float)w / (float)W);
  return hgl_check_launch("resize_bilinear");
}

int hgl_resize_bilinear_aa(const float* in, int C, int h, int w, float* out, int H, int W, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(in && out, "resize_bilinear_aa: null operand");
  HGL_REQUIRE(C > 0 && C <= 65535 && h > 0 && w > 0 && H > 0 && W > 0, "resize_bilinear_aa: bad shape");
  const long long n = (long long)H * W;
  hipLaunchKernelGGL(resize_aa_kernel, dim3((unsigned)((n + 255) / 256), C), dim3(256), 0, (hipStream_t)stream, in, h, w, out, H,
                     W, (float)h / (float)H, (float)w / (float)W);
  return hgl_check_launch("resize_bilinear_aa");
}

}  // extern "C"
