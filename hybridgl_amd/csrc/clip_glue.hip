// Row-wise / elementwise kernels around the CLIP transformer blocks.
// All are HBM-bound: one wave per token row, 16-byte accesses, no re-reads.
//   layernorm          clip/model.py:189-195 (nn.LayerNorm in fp32)
//   im2col_patch       vit.conv1 as a GEMM operand (model/backbone.py:130)
//   assemble_lnpre     cat(cls, tokens) + positional_embedding -> ln_pre (model/backbone.py:132-139)
//   mask_resize        TF.resize(masks.float(), (g,g)) + make_attn_mask keep bits (model/backbone.py:160,108-115)
//   mix                stream mixing + token masking (model/backbone.py:213-217,235-249,278-291)
//   text_embed         token_embedding + positional_embedding (clip/model.py:415-416)
//   gather_eot         x[arange, text.argmax(-1)] (clip/model.py:429)
#include "hgl_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- LayerNorm: one wave per row. VEC = D/256 float4 per lane (register resident) ----
template <int VEC>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ b,
                                                            float* __restrict__ y, int rows, float eps) {
  constexpr int D = VEC * 256;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const f32x4* xr = (const f32x4*)(x + (long long)row * D);
  f32x4 v[VEC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    v[i] = xr[lane + 64 * i];
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mean = wave_sum(s) * (1.0f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = v[i][e] - mean;
      q += d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
  f32x4* yr = (f32x4*)(y + (long long)row * D);
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const f32x4 wv = ((const f32x4*)w)[lane + 64 * i];
    const f32x4 bv = ((const f32x4*)b)[lane + 64 * i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * wv[e] + bv[e];
    yr[lane + 64 * i] = o;
  }
}

// generic D (small test geometries): one wave per row, three cached passes
__global__ __launch_bounds__(256) void layernorm_any_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ b,
                                                            float* __restrict__ y, int rows, int D,
                                                            float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long long)row * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += xr[i];
  const float mean = wave_sum(s) / D;
  float q = 0.f;
  for (int i = lane; i < D; i += 64) {
    const float d = xr[i] - mean;
    q += d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) / D + eps);
  float* yr = y + (long long)row * D;
  for (int i = lane; i < D; i += 64) yr[i] = (xr[i] - mean) * rstd * w[i] + b[i];
}

// ---- im2col for the stride==kernel patch convolution ----
// cols[(n*g*g + py*g + px), c*p*p + ky*p + kx] = img[n, c, py*p+ky, px*p+kx]
// V = elements per thread: 4 when the patch side is a multiple of 4 (ViT-B/16, B/32), 2 for even sides (ViT-L/14), 1 else
template <int V>
__global__ __launch_bounds__(256) void im2col_patch_kernel(const float* __restrict__ img,
                                                           float* __restrict__ cols, int N, int res,
                                                           int p, long long total4) {
  typedef float vec_t __attribute__((ext_vector_type(V)));
  const int g = res / p;
  const int kdim = 3 * p * p;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4;
       i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * V;
    const int col = (int)(e % kdim);
    const long long rowi = e / kdim;
    const int kx = col % p, ky = (col / p) % p, c = col / (p * p);
    const int px = (int)(rowi % g), py = (int)((rowi / g) % g);
    const int n = (int)(rowi / ((long long)g * g));
    const float* src = img + (((long long)n * 3 + c) * res + (py * p + ky)) * res + px * p + kx;
    if constexpr (V == 1) cols[e] = *src;
    else *(vec_t*)(cols + e) = *(const vec_t*)src;
  }
}

// the same, written as the fp16 (hi, lo) pair the split-fp16 patch-embedding GEMM reads (patch side % 4 == 0)
typedef _Float16 h16x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void im2col_patch_split_kernel(const float* __restrict__ img, _Float16* __restrict__ hi,
                                                                 _Float16* __restrict__ lo, int N, int res, int p,
                                                                 long long total4) {
  const int g = res / p;
  const int kdim = 3 * p * p;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4;
       i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 4;
    const int col = (int)(e % kdim);
    const long long rowi = e / kdim;
    const int kx = col % p, ky = (col / p) % p, c = col / (p * p);
    const int px = (int)(rowi % g), py = (int)((rowi / g) % g);
    const int n = (int)(rowi / ((long long)g * g));
    const f32x4 v = *(const f32x4*)(img + (((long long)n * 3 + c) * res + (py * p + ky)) * res + px * p + kx);
    h16x4_t a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      _Float16 h0, l0;
      hgl_split_hi_lo(v[j], h0, l0);
      a[j] = h0;
      b[j] = l0;
    }
    *(h16x4_t*)(hi + e) = a;
    *(h16x4_t*)(lo + e) = b;
  }
}

// ---- assemble tokens + LN (one wave per row; generic D via loops over LDS-free passes) ----
__global__ __launch_bounds__(256) void assemble_lnpre_kernel(
    const float* __restrict__ tok, const float* __restrict__ cls, const float* __restrict__ pos,
    const float* __restrict__ lw, const float* __restrict__ lb, float* __restrict__ x, int B, int S,
    int D) {
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
  if (row >= (long long)B * S) return;
  const int s = (int)(row % S);
  const long long bidx = row / S;
  const float* src = (s == 0) ? cls : tok + (bidx * (S - 1) + (s - 1)) * D;
  const float* pr = pos + (long long)s * D;
  float* xr = x + row * D;
  float sum = 0.f;
  for (int i = lane; i < D; i += 64) {
    const float v = src[i] + pr[i];
    xr[i] = v;  // stash pre-norm value (same lane re-reads it below)
    sum += v;
  }
  const float mean = wave_sum(sum) / D;
  float q = 0.f;
  for (int i = lane; i < D; i += 64) {
    const float d = xr[i] - mean;
    q += d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) / D + 1e-5f);
  for (int i = lane; i < D; i += 64) xr[i] = (xr[i] - mean) * rstd * lw[i] + lb[i];
}

// ---- bilinear (align_corners=False, no antialias) mask down-sample: 4 taps per output ----
__global__ __launch_bounds__(256) void mask_resize_kernel(const uint8_t* __restrict__ masks, int N,
                                                          int Hm, int Wm, int g,
                                                          float* __restrict__ pm,
                                                          uint8_t* __restrict__ keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * g * g) return;
  const int ox = i % g, oy = (i / g) % g, n = i / (g * g);
  // ATen area_pixel_compute_source_index (UpSample.h), scale = in/out in float
  const float sy = (float)Hm / (float)g, sx = (float)Wm / (float)g;
  float fy = fmaf(sy, oy + 0.5f, -0.5f);  // ATen contracts this into one fma
  float fx = fmaf(sx, ox + 0.5f, -0.5f);
  fy = fy < 0.f ? 0.f : fy;
  fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < Hm - 1 ? 1 : 0), x1 = x0 + (x0 < Wm - 1 ? 1 : 0);
  const float ly1 = fy - y0, lx1 = fx - x0;
  const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const uint8_t* m = masks + (long long)n * Hm * Wm;
  const float p00 = m[(long long)y0 * Wm + x0], p01 = m[(long long)y0 * Wm + x1];
  const float p10 = m[(long long)y1 * Wm + x0], p11 = m[(long long)y1 * Wm + x1];
  const float v = ly0 * (lx0 * p00 + lx1 * p01) + ly1 * (lx0 * p10 + lx1 * p11);
  pm[i] = v;
  if (keep) keep[i] = v != 0.f ? 1 : 0;
}

// ---- out = ca*a + cb*b*(s==0 ? 1 : pm[n,s-1]) ----
__global__ __launch_bounds__(256) void mix_kernel(float* __restrict__ out, const float* __restrict__ a,
                                                  float ca, const float* __restrict__ b, float cb,
                                                  const float* __restrict__ pm, int S, int D4,
                                                  long long total4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / D4;
    const int s = (int)(row % S);
    const long long n = row / S;
    float m = 1.f;
    if (pm && s > 0) m = pm[n * (S - 1) + (s - 1)];
    const f32x4 bv = ((const f32x4*)b)[i];
    f32x4 o;
    if (a) {
      const f32x4 av = ((const f32x4*)a)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bv[e] * m) * cb + av[e] * ca;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bv[e] * m) * cb;
    }
    ((f32x4*)out)[i] = o;
  }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x,
                                                          long long row_stride, int rows, int D,
                                                          float* __restrict__ y) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)rows * D) return;
  const long long r = i / D;
  const int d = (int)(i % D);
  y[i] = x[r * row_stride + d];
}

__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ y,
                                                          const float* __restrict__ x, long long n) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n) y[i] += x[i];
}

// one wave per (b,s) row; also finds the EOT position = first argmax of token ids per sequence
// S <= ctx positions of every token row are embedded (row stride ctx); the EOT index is the arg-max over the whole row
__global__ __launch_bounds__(256) void text_embed_kernel(const int32_t* __restrict__ tokens,
                                                         const float* __restrict__ emb,
                                                         const float* __restrict__ pos,
                                                         float* __restrict__ x, int B, int S, int ctx, int D,
                                                         int vocab, int32_t* __restrict__ eot) {
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
  if (row >= (long long)B * S) return;
  const int s = (int)(row % S);
  const int bidx = (int)(row / S);
  int tk = tokens[(long long)bidx * ctx + s];
  tk = tk < 0 ? 0 : (tk >= vocab ? vocab - 1 : tk);
  const float* er = emb + (long long)tk * D;
  const float* pr = pos + (long long)s * D;
  float* xr = x + row * D;
  for (int i = lane; i < D; i += 64) xr[i] = er[i] + pr[i];
  if (s == 0 && lane == 0) {
    int best = 0, bv = tokens[(long long)bidx * ctx];
    for (int j = 1; j < ctx; ++j) {
      const int v = tokens[(long long)bidx * ctx + j];
      if (v > bv) { bv = v; best = j; }
    }
    eot[bidx] = best;
  }
}

// y[b] = x[b, eot[b]]; an EOT beyond the computed prefix (a caller that passed too short a prefix) gives NaN rows
__global__ __launch_bounds__(256) void gather_eot_kernel(const float* __restrict__ x,
                                                         const int32_t* __restrict__ eot, int B,
                                                         int S, int D, float* __restrict__ y) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)B * D) return;
  const int bidx = (int)(i / D), d = (int)(i % D);
  const int e = eot[bidx];
  y[i] = (e >= 0 && e < S) ? x[((long long)bidx * S + e) * D + d] : __int_as_float(0x7fc00000);   // a negative caller-chosen position is outside too
}

// x[b, pos[z], :] = 0 for every string b and listed position (model/backbone.py:44-46: x[masking_index] = 0 on [S,B,D])
__global__ __launch_bounds__(256) void zero_positions_kernel(float* __restrict__ x, int B, int S, int D,
                                                             const int32_t* __restrict__ pos, int n) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)B * n * D) return;
  const int d = (int)(i % D);
  const int z = (int)((i / D) % n);
  const int b = (int)(i / ((long long)D * n));
  const int p = pos[z];
  if (p >= 0 && p < S) x[((long long)b * S + p) * D + d] = 0.0f;
}

inline unsigned grid_for(long long n, int block = 256, unsigned cap = 256u * 16u) {
  long long g = (n + block - 1) / block;
  if (g < 1) g = 1;
  return (unsigned)(g > cap ? cap : g);
}

}  // namespace

int hgl_launch_layernorm(const float* x, const float* w, const float* b, float* y, int rows, int D,
                         float eps, hipStream_t st) {
  HGL_REQUIRE(x && w && b && y && rows > 0 && D > 0, "layernorm: bad arguments");
  const unsigned grid = (unsigned)((rows + 3) / 4);
  const bool al = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)b) & 15) == 0;
  if (al && D == 256) hipLaunchKernelGGL(layernorm_vec_kernel<1>, dim3(grid), dim3(256), 0, st, x, w, b, y, rows, eps);
  else if (al && D == 512) hipLaunchKernelGGL(layernorm_vec_kernel<2>, dim3(grid), dim3(256), 0, st, x, w, b, y, rows, eps);
  else if (al && D == 768) hipLaunchKernelGGL(layernorm_vec_kernel<3>, dim3(grid), dim3(256), 0, st, x, w, b, y, rows, eps);
  else if (al && D == 1024) hipLaunchKernelGGL(layernorm_vec_kernel<4>, dim3(grid), dim3(256), 0, st, x, w, b, y, rows, eps);
  else if (al && D == 1280) hipLaunchKernelGGL(layernorm_vec_kernel<5>, dim3(grid), dim3(256), 0, st, x, w, b, y, rows, eps);
  else hipLaunchKernelGGL(layernorm_any_kernel, dim3(grid), dim3(256), 0, st, x, w, b, y, rows, D, eps);
  return hgl_check_launch("layernorm");
}

int hgl_launch_im2col_patch(const float* img, int N, int res, int patch, float* cols, hipStream_t st) {
  HGL_REQUIRE(img && cols && N > 0 && res > 0 && patch > 0 && res % patch == 0, "im2col: bad arguments (res=%d patch=%d)", res, patch);
  const long long total = (long long)N * 3 * res * res;
  if (patch % 4 == 0)
    hipLaunchKernelGGL(im2col_patch_kernel<4>, dim3(grid_for(total / 4)), dim3(256), 0, st, img, cols, N, res, patch, total / 4);
  else if (patch % 2 == 0)
    hipLaunchKernelGGL(im2col_patch_kernel<2>, dim3(grid_for(total / 2)), dim3(256), 0, st, img, cols, N, res, patch, total / 2);
  else
    hipLaunchKernelGGL(im2col_patch_kernel<1>, dim3(grid_for(total)), dim3(256), 0, st, img, cols, N, res, patch, total);
  return hgl_check_launch("im2col_patch");
}

int hgl_launch_im2col_patch_split(const float* img, int N, int res, int patch, void* hi, void* lo, hipStream_t st) {
  HGL_REQUIRE(img && hi && lo && N > 0 && res > 0 && patch > 0 && res % patch == 0 && patch % 4 == 0,
              "im2col_split: bad arguments (res=%d patch=%d)", res, patch);
  const long long total = (long long)N * 3 * res * res;
  hipLaunchKernelGGL(im2col_patch_split_kernel, dim3(grid_for(total / 4)), dim3(256), 0, st, img, (_Float16*)hi, (_Float16*)lo, N,
                     res, patch, total / 4);
  return hgl_check_launch("im2col_patch_split");
}

int hgl_launch_assemble_lnpre(const float* tok, const float* cls, const float* pos, const float* lw,
                              const float* lb, float* x, int B, int S, int D, hipStream_t st) {
  const long long rows = (long long)B * S;
  hipLaunchKernelGGL(assemble_lnpre_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, tok, cls, pos, lw, lb, x, B, S, D);
  return hgl_check_launch("assemble_lnpre");
}

int hgl_launch_mask_resize(const uint8_t* masks, int N, int Hm, int Wm, int g, float* pm,
                           uint8_t* keep, hipStream_t st) {
  HGL_REQUIRE(masks && pm && N > 0 && Hm > 0 && Wm > 0 && g > 0, "mask_resize: bad arguments");
  const int total = N * g * g;
  hipLaunchKernelGGL(mask_resize_kernel, dim3((total + 255) / 256), dim3(256), 0, st, masks, N, Hm, Wm, g, pm, keep);
  return hgl_check_launch("mask_resize");
}

int hgl_launch_mix(float* out, const float* a, float ca, const float* b, float cb, const float* pm,
                   int N, int S, int D, hipStream_t st) {
  HGL_REQUIRE(out && b && (D & 3) == 0, "mix: bad arguments");
  const long long total4 = (long long)N * S * D / 4;
  hipLaunchKernelGGL(mix_kernel, dim3(grid_for(total4)), dim3(256), 0, st, out, a, ca, b, cb, pm, S, D / 4, total4);
  return hgl_check_launch("mix");
}

int hgl_launch_gather_rows(const float* x, long long row_stride, int rows, int D, float* y,
                           hipStream_t st) {
  const long long n = (long long)rows * D;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, row_stride, rows, D, y);
  return hgl_check_launch("gather_rows");
}

int hgl_launch_add_inplace(float* y, const float* x, long long n, hipStream_t st) {
  hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y, x, n);
  return hgl_check_launch("add_inplace");
}

int hgl_launch_text_embed(const int32_t* tokens, const float* emb, const float* pos, float* x, int B,
                          int S, int ctx, int D, int vocab, int32_t* eot, hipStream_t st) {
  const long long rows = (long long)B * S;
  hipLaunchKernelGGL(text_embed_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, tokens, emb, pos, x, B, S, ctx, D, vocab, eot);
  return hgl_check_launch("text_embed");
}

int hgl_launch_zero_positions(float* x, int B, int S, int D, const int32_t* pos, int n, hipStream_t st) {
  const long long total = (long long)B * n * D;
  hipLaunchKernelGGL(zero_positions_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, B, S, D, pos, n);
  return hgl_check_launch("zero_positions");
}

int hgl_launch_gather_eot(const float* x, const int32_t* eot, int B, int S, int D, float* y,
                          hipStream_t st) {
  const long long n = (long long)B * D;
  hipLaunchKernelGGL(gather_eot_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, eot, B, S, D, y);
  return hgl_check_launch("gather_eot");
}
