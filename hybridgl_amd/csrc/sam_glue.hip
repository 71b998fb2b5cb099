// HBM-bound glue kernels of the SAM path (third_party/segment-anything/segment_anything/*).
//   sam_preprocess        modeling/sam.py:164-174 (normalise + zero-pad the resized image)
//   win_partition / win_unpartition_add   modeling/image_encoder.py:243-290 (+ residual add, :176-179)
//   relpos_gather         modeling/image_encoder.py:325-361 (tables for the fused attention bias)
//   im2col3x3_nhwc        neck conv3x3, modeling/image_encoder.py:95-101
//   add_rows_bcast        keys + key_pe / queries + query_pe, modeling/transformer.py:160-179
//   pe_points / build_tokens  modeling/prompt_encoder.py:73-91,185-214, modeling/mask_decoder.py:120-123
//   ln_gelu_rows          LayerNorm2d + GELU of output_upscaling, modeling/mask_decoder.py:53-59
//   hyper_logits          hyper-network x upscaled embedding, written in the [P,3,4g,4g] pixel order
//   sam_postprocess       modeling/sam.py:133-162 + utils/amg.py:156-176,303-346 fused
//   sam_select / nms      automatic_mask_generator.py:251-257,287-319
#include "hgl_common.h"
#include <math.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4g __attribute__((ext_vector_type(4)));
typedef unsigned u32x4g __attribute__((ext_vector_type(4)));

namespace {

inline unsigned grid1(long long n, int block = 256) { return (unsigned)((n + block - 1) / block); }

__global__ __launch_bounds__(256) void sam_preprocess_kernel(const uint8_t* __restrict__ img, int h, int w,
                                                             int S, float* __restrict__ out) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= 3ll * S * S) return;
  const int x = (int)(i % S), y = (int)((i / S) % S), c = (int)(i / ((long long)S * S));
  const float mean[3] = {123.675f, 116.28f, 103.53f};
  const float stdv[3] = {58.395f, 57.12f, 57.375f};
  float v = 0.f;
  if (y < h && x < w) v = ((float)img[((long long)y * w + x) * 3 + c] - mean[c]) / stdv[c];
  out[i] = v;
}

// One pass of Pillow's 8-bit resampler (Resample.c ImagingResampleHorizontal/Vertical_8bpc) along one
// axis: out[o][..] = clip8((2^21 + sum_x in[first+x] * k[x]) >> 22), weights/bounds precomputed on the host
// exactly as precompute_coeffs/normalize_coeffs_8bpc do.  stride_in/out: element strides of the resampled axis;
// `inner` contiguous bytes (other axis x channels) are handled by consecutive threads.
__global__ __launch_bounds__(256) void pil_resample_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                           const int* __restrict__ kk, const int* __restrict__ bounds,
                                                           int ksize, int n_out, long long outer, long long inner,
                                                           long long in_outer_stride, long long in_axis_stride,
                                                           long long out_outer_stride, long long out_axis_stride) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  const long long total = outer * n_out * inner;
  if (i >= total) return;
  const long long c = i % inner;
  const int o = (int)((i / inner) % n_out);
  const long long r = i / (inner * n_out);
  const int first = bounds[2 * o], cnt = bounds[2 * o + 1];
  const int* k = kk + (long long)o * ksize;
  int acc = 1 << 21;
  const uint8_t* p = in + r * in_outer_stride + (long long)first * in_axis_stride + c;
  for (int x = 0; x < cnt; ++x) acc += (int)p[(long long)x * in_axis_stride] * k[x];
  acc >>= 22;
  out[r * out_outer_stride + (long long)o * out_axis_stride + c] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
}

// ToTensor + Normalize of a uint8 HWC image as a table look-up: out[c, y, x] = lut[c * 256 + img[y, x, c]].  The 3 x 256
// table is computed on the host with the reference's own fp32 arithmetic ((v / 255 - mean) / std), so the result is the
// reference's bit for bit whatever the device's division does.  Thread = 4 consecutive pixels of one channel plane.
__global__ __launch_bounds__(256) void u8_to_chw_lut_kernel(const uint8_t* __restrict__ img, const float* __restrict__ lut,
                                                            float* __restrict__ out, long long HW, int C) {
  const long long q = blockIdx.x * 256ll + threadIdx.x;     // quad of pixels
  const int c = blockIdx.y;
  const long long p0 = q * 4;
  if (p0 >= HW) return;
  const float* t = lut + c * 256;
  if (p0 + 4 <= HW) {
    f32x4 v;
    v[0] = t[img[(p0 + 0) * C + c]];
    v[1] = t[img[(p0 + 1) * C + c]];
    v[2] = t[img[(p0 + 2) * C + c]];
    v[3] = t[img[(p0 + 3) * C + c]];
    if ((((size_t)(out + (long long)c * HW + p0)) & 15) == 0) {
      *(f32x4*)(out + (long long)c * HW + p0) = v;
    } else {
      for (int j = 0; j < 4; ++j) out[(long long)c * HW + p0 + j] = v[j];
    }
  } else {
    for (long long p = p0; p < HW; ++p) out[(long long)c * HW + p] = t[img[p * C + c]];
  }
}

// Hw[(wy*nwx + wx)*ws*ws + py*ws + px, :] = (y<g && x<g) ? H[y*g+x, :] : 0
__global__ __launch_bounds__(256) void win_partition_kernel(const float* __restrict__ H, int g, int ws,
                                                            int nw, int D4, float* __restrict__ Hw,
                                                            long long total4) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total4) return;
  const int d = (int)(i % D4);
  const long long row = i / D4;
  const int p = (int)(row % (ws * ws)), win = (int)(row / (ws * ws));
  const int y = (win / nw) * ws + p / ws, x = (win % nw) * ws + p % ws;
  f32x4 v = {0, 0, 0, 0};
  if (y < g && x < g) v = ((const f32x4*)H)[((long long)y * g + x) * D4 + d];
  ((f32x4*)Hw)[i] = v;
}

// X[y*g+x, :] += P[window row, :]
__global__ __launch_bounds__(256) void win_unpartition_add_kernel(float* __restrict__ X, int g, int ws,
                                                                  int nw, int D4,
                                                                  const float* __restrict__ P,
                                                                  long long total4) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total4) return;
  const int d = (int)(i % D4);
  const long long tok = i / D4;
  const int y = (int)(tok / g), x = (int)(tok % g);
  const long long prow = ((long long)(y / ws) * nw + x / ws) * ws * ws + (y % ws) * ws + x % ws;
  f32x4 a = ((f32x4*)X)[i];
  const f32x4 b = ((const f32x4*)P)[prow * D4 + d];
#pragma unroll
  for (int e = 0; e < 4; ++e) a[e] += b[e];
  ((f32x4*)X)[i] = a;
}

// T: [heads][B*S][L] (q . rel_pos[r]);  rel: [(b*heads+h)][S][size], rel[q][k] = T[h][b*S+q][qc - k + size-1]
// use_w: qc = q % size (width axis) else q / size (height axis)
__global__ __launch_bounds__(256) void relpos_gather_kernel(const float* __restrict__ T, int B, int heads,
                                                            int S, int size, int L, int use_w,
                                                            float* __restrict__ rel, long long total) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i % size);
  const int q = (int)((i / size) % S);
  const int bh = (int)(i / ((long long)size * S));
  const int b = bh / heads, h = bh % heads;
  const int qc = use_w ? q % size : q / size;
  rel[i] = T[((long long)h * B * S + (long long)b * S + q) * L + (qc - k + size - 1)];
}

// Decomposed rel-pos tables computed directly (image_encoder.py:325-361):
//   rel_h[bh][q][k] = q_vec . rel_pos_h[qy - k + size-1],  rel_w[bh][q][k] = q_vec . rel_pos_w[qx - k + size-1]
// One workgroup = (bh, 256 queries, axis).  The axis' (2*size-1) x HD table sits in LDS with rows padded to HD+4
// floats (row stride 84 / 68 dwords: the 16 lanes of a ds_read_b128 group, which read consecutive rows on the w
// axis, fall on distinct banks; on the h axis a row of the image shares one table row = broadcast).  Each thread
// keeps its query vector in registers and walks the `size` table rows it needs.  Replaces two padded GEMMs + two
// gathers; the per-output version of this kernel was load-instruction bound (64 us per windowed block, now ~10).
template <int HD>
__global__ __launch_bounds__(256) void relpos_direct_kernel(const float* __restrict__ qkv, int ldq, int heads, int S,
                                                            int size, const float* __restrict__ Rh,
                                                            const float* __restrict__ Rw,
                                                            float* __restrict__ rel_h, float* __restrict__ rel_w) {
  extern __shared__ __attribute__((aligned(16))) float rp_tab[];   // [(2*size-1)][HD+4]
  constexpr int LD = HD + 4;
  const int axis = blockIdx.z, bh = blockIdx.y;
  const int b = bh / heads, h = bh % heads;
  const float* R = axis ? Rw : Rh;
  const int nrow = 2 * size - 1;
  for (int i = threadIdx.x; i < nrow * (HD / 4); i += 256) {
    const int rr = i / (HD / 4), c = i % (HD / 4);
    *(f32x4*)(rp_tab + rr * LD + 4 * c) = *(const f32x4*)(R + (long long)rr * HD + 4 * c);
  }
  __syncthreads();
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= S) return;
  f32x4 qv[HD / 4];
  const float* qp = qkv + ((long long)b * S + q) * ldq + h * HD;
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) qv[c] = *(const f32x4*)(qp + 4 * c);
  const int qc = axis ? q % size : q / size;
  float* out = (axis ? rel_w : rel_h) + ((long long)bh * S + q) * size;
  const float* row = rp_tab + (qc + size - 1) * LD;   // k = 0; row index falls by one per k
  for (int k = 0; k < size; k += 2) {                 // size is even (14 or 64): two outputs per 8-byte store
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      const f32x4 r0 = *(const f32x4*)(row + 4 * c), r1 = *(const f32x4*)(row - LD + 4 * c);
      // same summation order as a sequential dot product over the head dim
      a0 += qv[c][0] * r0[0]; a0 += qv[c][1] * r0[1]; a0 += qv[c][2] * r0[2]; a0 += qv[c][3] * r0[3];
      a1 += qv[c][0] * r1[0]; a1 += qv[c][1] * r1[1]; a1 += qv[c][2] * r1[2]; a1 += qv[c][3] * r1[3];
    }
    f32x2 o; o[0] = a0; o[1] = a1;
    *(f32x2*)(out + k) = o;
    row -= 2 * LD;
  }
}

// The same tables on the matrix cores (f16x3 mode): T^T = R . Q^T per (batch, head, axis) with the split-fp16 scheme
// of the GEMMs -- A operand = the (2*size-1) table rows (zero-padded to NT*32, split to fp16 hi | lo in LDS once per
// workgroup), B operand = the wave's 32 query vectors (split in registers), 3 MFMAs per 16-wide k step.  The
// accumulator (lane = query, 16 of 32 table indices per lane) goes through an LDS patch from which every query reads
// its `size` consecutive entries T[q][qc + size-1 - k].  The VALU version above is bound by its LDS reads (one
// 16-byte read per 4 fmas): 30 us per windowed block against ~10 for the bytes this one moves.
typedef _Float16 rp_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 rp_h4 __attribute__((ext_vector_type(4)));
// PS: q arrives as the fp16 hi / lo planes of the in-projection's split write-out (q_hi, q_lo; ldq in halfs; qkv unused)
template <int HD, int NT, bool PS = false>
__global__ __launch_bounds__(256) void relpos_mfma_kernel(const float* __restrict__ qkv, int ldq, int heads, int S, int size,
                                                          const float* __restrict__ Rh, const float* __restrict__ Rw,
                                                          float* __restrict__ rel_h, float* __restrict__ rel_w,
                                                          const _Float16* __restrict__ q_hi = nullptr,
                                                          const _Float16* __restrict__ q_lo = nullptr) {
  constexpr int KS = HD / 16;          // k steps
  constexpr int TP = HD + 8;           // halfs per staged table row (16-byte aligned, odd multiple of 16 B)
  constexpr int GP = NT * 32 + 1;      // floats per query row of the gather patch
  extern __shared__ __attribute__((aligned(16))) unsigned char rp_smem[];
  _Float16* Th = (_Float16*)rp_smem;                       // [NT*32][TP] hi
  _Float16* Tl = Th + NT * 32 * TP;                        // [NT*32][TP] lo
  float* G = (float*)(Tl + NT * 32 * TP);                  // [4 waves][32 queries][GP]
  const int bh = blockIdx.y;
  const int b = bh / heads, hh = bh % heads;
  const int nrow = 2 * size - 1;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  // Query blocks per workgroup: the global blocks' tables (127 x HD, split to fp16 hi / lo here) are staged ONCE per axis for
  // QB blocks of 128 queries (round 5: one block per workgroup spent as long splitting the table as multiplying)
  constexpr int QB = (PS && NT == 4) ? 4 : 1;
  float* Gw = G + (wave * 32 + r) * GP;
  for (int axis = 0; axis < 2; ++axis) {
    const float* R = axis ? Rw : Rh;
    if (axis) __syncthreads();                             // every wave has gathered from its patch / read the table
    for (int i = t; i < NT * 32 * (HD / 4); i += 256) {
      const int rr = i / (HD / 4), c = i % (HD / 4);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (rr < nrow) v = *(const f32x4*)(R + (long long)rr * HD + 4 * c);
      rp_h4 a, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        _Float16 h0, l0;
        hgl_split_hi_lo(v[e], h0, l0);
        a[e] = h0;
        l[e] = l0;
      }
      *(rp_h4*)(Th + rr * TP + 4 * c) = a;
      *(rp_h4*)(Tl + rr * TP + 4 * c) = l;
    }
    __syncthreads();
    for (int qb = 0; qb < QB; ++qb) {
    const int q = ((blockIdx.x * QB + qb) * 4 + wave) * 32 + r;
    const bool qvalid = q < S;
    rp_h8 qh[KS], ql[KS];
  if constexpr (PS) {
    const long long qo = ((long long)b * S + (qvalid ? q : 0)) * ldq + hh * HD + 8 * h;
#pragma unroll
    for (int c = 0; c < KS; ++c) {
      qh[c] = *(const rp_h8*)(q_hi + qo + 16 * c);
      ql[c] = *(const rp_h8*)(q_lo + qo + 16 * c);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!qvalid) {
#pragma unroll
      for (int c = 0; c < KS; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) { qh[c][e] = (_Float16)0.f; ql[c][e] = (_Float16)0.f; }
    }
  } else {
  const float* qp = qkv + ((long long)b * S + (qvalid ? q : 0)) * ldq + hh * HD;
  // one batch of loads, then the splits: a split right behind its load (hgl_split_hi_lo's opaque asm) is one serial memory
  // round trip per 16-byte piece
  f32x4 qraw[KS][2];
#pragma unroll
  for (int c = 0; c < KS; ++c)
#pragma unroll
    for (int half = 0; half < 2; ++half) qraw[c][half] = *(const f32x4*)(qp + 16 * c + 8 * h + 4 * half);   // qp: clamped row
  __builtin_amdgcn_sched_barrier(0);
  if (!qvalid) {
#pragma unroll
    for (int c = 0; c < KS; ++c) qraw[c][0] = qraw[c][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int c = 0; c < KS; ++c) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const f32x4 v = qraw[c][half];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        _Float16 h0, l0;
        hgl_split_hi_lo(v[e], h0, l0);
        qh[c][4 * half + e] = h0;
        ql[c][4 * half + e] = l0;
      }
    }
  }
  }
    if (QB > 1) {      // the previous block's gather reads of this wave's patch are done (LDS operations of a wave are in order)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      const _Float16* trow = Th + (tt * 32 + r) * TP + 8 * h;
      const _Float16* lrow = Tl + (tt * 32 + r) * TP + 8 * h;
#pragma unroll
      for (int c = 0; c < KS; ++c) {
        const rp_h8 ah = *(const rp_h8*)(trow + 16 * c);
        const rp_h8 al = *(const rp_h8*)(lrow + 16 * c);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[c], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[c], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[c], acc, 0, 0, 0);
      }
      // acc[e] = T[table index tt*32 + (e&3) + 8*(e>>2) + 4*h][query r]
#pragma unroll
      for (int e = 0; e < 16; ++e) Gw[tt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] = acc[e];
    }
    if (QB > 1) {      // the patch is private to the wave
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
      __syncthreads();
    }
    if ((size & 63) == 0) {
      // rows of `size` floats, written as rows: 16 consecutive lanes store the 256 contiguous bytes of 64 entries of ONE query
      // (the wave's patch holds T[query][table index]; entry k of query qq is index qc + size - 1 - k).  The first version
      // stored 8 bytes per lane at the stride of a row -- 64 separate 8-byte writes per instruction: 0.6 TB/s for the
      // 537 MB a global block writes.
      const int q0w = ((blockIdx.x * QB + qb) * 4 + wave) * 32;
      const float* Gq = G + wave * 32 * GP;
      const int per = size >> 2;                            // 16-byte pieces per row
      for (int i = lane; i < 32 * per; i += 64) {
        const int ql = i / per, k4 = 4 * (i - ql * per);
        const int qq = q0w + ql;
        if (qq < S) {
          const int qc = axis ? qq % size : qq / size;
          const float* src = Gq + ql * GP + qc + size - 1 - k4;
          const f32x4 o = {src[0], src[-1], src[-2], src[-3]};
          *(f32x4*)((axis ? rel_w : rel_h) + ((long long)bh * S + qq) * size + k4) = o;
        }
      }
    } else if (qvalid) {
      const int qc = axis ? q % size : q / size;
      float* out = (axis ? rel_w : rel_h) + ((long long)bh * S + q) * size;
      // the two lanes of a query share its row: k in [0, sp) and [sp, size), sp even so that both write aligned pairs
      const int sp = (size / 2 + 1) & ~1;
      const int k0 = h ? sp : 0, k1 = h ? size : sp;
      const float* src = Gw + qc + size - 1;                // entry of k = 0; the index falls by one per k
      for (int k = k0; k < k1; k += 2) {
        f32x2 o;
        o[0] = src[-k];
        o[1] = src[-k - 1];
        *(f32x2*)(out + k) = o;
      }
    }
    }   // query blocks
  }
}

// Row maps of the zero-padded window partition (image_encoder.py:244-283): the r-th REAL token in window order lives
// at padded row pad_of[r] = window*ws*ws + position and at token-order row tok_of[r] = y*g + x; pad_list collects
// the padded rows that hold no token.  The f16x3 GEMMs of a windowed block run over the real tokens only: the pad
// rows of qkv are the bias (filled by fill_rows_kernel) and the pad rows of the projection are never needed.
__global__ __launch_bounds__(256) void win_maps_kernel(int g, int ws, int nw, int nb, int* __restrict__ pad_of,
                                                       int* __restrict__ tok_of, int* __restrict__ pad_list,
                                                       int* __restrict__ pad_count) {
  // nb images stacked along the rows: image i owns token rows [i*g*g, (i+1)*g*g) and padded rows [i*Tw, (i+1)*Tw)
  const int Tw = nw * nw * ws * ws;
  const int gidx = blockIdx.x * 256 + threadIdx.x;
  if (gidx >= nb * Tw) return;
  const int img = gidx / Tw, idx = gidx - img * Tw;
  const int w = idx / (ws * ws), p = idx - w * ws * ws;
  const int wy = w / nw, wx = w - wy * nw, py = p / ws, px = p - py * ws;
  const int y = wy * ws + py, x = wx * ws + px;
  if (y < g && x < g) {
    const int rh = min(ws, g - wy * ws), rw = min(ws, g - wx * ws);
    const int r = img * g * g + wy * ws * g + rh * (wx * ws) + py * rw + px;
    pad_of[r] = gidx;
    tok_of[r] = img * g * g + y * g + x;
  } else {
    pad_list[atomicAdd(pad_count, 1)] = gidx;
  }
}

// dst[rows[i], :] = v[:] for i < *nrows   (N % 4 == 0)
__global__ __launch_bounds__(256) void fill_rows_kernel(float* __restrict__ dst, int ld, const int* __restrict__ rows,
                                                        const int* __restrict__ nrows, const float* __restrict__ v, int N4) {
  const int i = blockIdx.y;
  if (i >= *nrows) return;
  f32x4* d = (f32x4*)(dst + (long long)rows[i] * ld);
  for (int c = blockIdx.x * 256 + threadIdx.x; c < N4; c += gridDim.x * 256) d[c] = ((const f32x4*)v)[c];
}

// the same rows of a tensor kept as fp16 hi | lo planes (the split qkv output of the windowed blocks): hi[rows[i], :] | lo[rows[i], :]
// = the split of v[:]
__global__ __launch_bounds__(256) void fill_rows_split_kernel(_Float16* __restrict__ hi, _Float16* __restrict__ lo, int ld,
                                                              const int* __restrict__ rows, const int* __restrict__ nrows,
                                                              const float* __restrict__ v, int N4) {
  typedef _Float16 fr_h4 __attribute__((ext_vector_type(4)));
  const int i = blockIdx.y;
  if (i >= *nrows) return;
  const long long ro = (long long)rows[i] * ld;
  for (int c = blockIdx.x * 256 + threadIdx.x; c < N4; c += gridDim.x * 256) {
    const f32x4 x = ((const f32x4*)v)[c];
    fr_h4 a, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 h0, l0;
      hgl_split_hi_lo(x[e], h0, l0);
      a[e] = h0;
      l[e] = l0;
    }
    *(fr_h4*)(hi + ro + 4 * c) = a;
    *(fr_h4*)(lo + ro + 4 * c) = l;
  }
}

// cols[(y*g+x), c*9 + ky*3+kx] = in[(y+ky-1), (x+kx-1), c] (zero padded), NHWC input
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* __restrict__ in, int g, int C,
                                                        float* __restrict__ cols, long long total) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total) return;
  const int col = (int)(i % (C * 9));
  const long long tok = i / (C * 9);
  const int c = col / 9, kk = col % 9;
  const int y = (int)(tok / g) + kk / 3 - 1, x = (int)(tok % g) + kk % 3 - 1;
  cols[i] = (y >= 0 && y < g && x >= 0 && x < g) ? in[((long long)y * g + x) * C + c] : 0.f;
}

// out[b, r, :] = a[b (or shared), r, :] + pe[r, :]
__global__ __launch_bounds__(256) void add_rows_bcast_kernel(const float* __restrict__ a, long long a_bstride4,
                                                             const float* __restrict__ pe, long long rows4,
                                                             float* __restrict__ out, long long total4) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total4) return;
  const long long b = i / rows4, r = i - b * rows4;
  const f32x4 x = ((const f32x4*)a)[b * a_bstride4 + r];
  const f32x4 p = ((const f32x4*)pe)[r];
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = x[e] + p[e];
  ((f32x4*)out)[i] = o;
}

// random-Fourier positional encoding: out[n, 0:F] = sin(2*pi*((2c-1) @ G)), out[n, F:2F] = cos(...)
// coords01: [n,2] fp32 already normalised to [0,1]; then optional label embedding / replacement.
// mode 0: dense grid rows (no label); mode 1: prompt rows [P,2,C]: row 0 += pos_embed, row 1 = not_a_point
__global__ __launch_bounds__(256) void pe_kernel(const float* __restrict__ coords01,
                                                 const float* __restrict__ G, int n, int F, int mode,
                                                 const float* __restrict__ pos_embed,
                                                 const float* __restrict__ not_a_point,
                                                 float* __restrict__ out) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= (long long)n * F) return;
  const int f = (int)(i % F);
  const int r = (int)(i / F);
  const int C = 2 * F;
  float* o = out + (long long)r * C;
  if (mode == 1 && (r & 1)) {  // padding point: label -1 -> embedding replaced (prompt_encoder.py:86-87)
    o[f] = not_a_point[f];
    o[f + F] = not_a_point[f + F];
    return;
  }
  const int pt = mode == 1 ? (r >> 1) : r;  // prompts: rows (point, padding) share one coordinate
  const float cx = 2.f * coords01[2 * pt] - 1.f, cy = 2.f * coords01[2 * pt + 1] - 1.f;
  const float v = 6.283185307179586f * (cx * G[f] + cy * G[F + f]);
  float s = sinf(v), c = cosf(v);
  if (mode == 1) { s += pos_embed[f]; c += pos_embed[f + F]; }
  o[f] = s;
  o[f + F] = c;
}

// Prompt rows with labels (PromptEncoder._embed_points / _embed_boxes, prompt_encoder.py:73-101): row r = the positional
// encoding of coords01[r] (zeroed for label -1, the padding point) + the label's embedding: -1 not_a_point, 0 / 1
// point_embeddings[0 / 1] (background / foreground), 2 / 3 point_embeddings[2 / 3] (box corners)
struct PeLabelTab { const float* e[5]; };
__global__ __launch_bounds__(256) void pe_labeled_kernel(const float* __restrict__ coords01, const int32_t* __restrict__ labels,
                                                         const float* __restrict__ G, int n, int F, PeLabelTab tab,
                                                         float* __restrict__ out) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= (long long)n * F) return;
  const int f = (int)(i % F);
  const int r = (int)(i / F);
  int lab = labels[r];
  if (lab < -1 || lab > 3) lab = -1;   // labels outside [-1, 3] (prompt_encoder.py:73-86 knows no other) read as padding, never past the table
  const float* e = tab.e[lab + 1];
  float s = 0.f, c = 0.f;
  if (lab >= 0) {
    const float cx = 2.f * coords01[2 * r] - 1.f, cy = 2.f * coords01[2 * r + 1] - 1.f;
    const float v = 6.283185307179586f * (cx * G[f] + cy * G[F + f]);
    s = sinf(v);
    c = cosf(v);
  }
  out[(long long)r * 2 * F + f] = s + e[f];
  out[(long long)r * 2 * F + f + F] = c + e[f + F];
}

// tokens[p, 0] = iou_token; tokens[p, 1..4] = mask_tokens; tokens[p, 5..T-1] = sparse[p, 0..T-6]
__global__ __launch_bounds__(256) void build_tokens_kernel(const float* __restrict__ iou_tok,
                                                           const float* __restrict__ mask_tok,
                                                           const float* __restrict__ sparse, int P, int C, int T,
                                                           float* __restrict__ tokens) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= (long long)P * T * C) return;
  const int c = (int)(i % C), t = (int)((i / C) % T), p = (int)(i / ((long long)T * C));
  float v;
  if (t == 0) v = iou_tok[c];
  else if (t < 5) v = mask_tok[(t - 1) * C + c];
  else v = sparse[((long long)p * (T - 5) + (t - 5)) * C + c];
  tokens[i] = v;
}

// PromptEncoder.mask_downscaling (prompt_encoder.py:57-66, :103-106) on mask inputs [P,1,4g,4g] -> dense rows [P, g*g, 256]:
// Conv2d(1,4,k2,s2) + LayerNorm2d(4) + GELU + Conv2d(4,16,k2,s2) + LayerNorm2d(16) + GELU + Conv2d(16,256,k1).  64 threads
// per output pixel: each evaluates the pixel's 16 mid channels from its 4x4 input block (a few hundred flops), then writes
// four of the 256 output channels.  Off the hot path (SamPredictor with mask_input).
__global__ __launch_bounds__(256) void mask_downscaling_kernel(const float* __restrict__ in, int g, const float* __restrict__ c1w,
                                                               const float* __restrict__ c1b, const float* __restrict__ n1w,
                                                               const float* __restrict__ n1b, const float* __restrict__ c2w,
                                                               const float* __restrict__ c2b, const float* __restrict__ n2w,
                                                               const float* __restrict__ n2b, const float* __restrict__ c3w,
                                                               const float* __restrict__ c3b, float* __restrict__ out) {
  const int p = blockIdx.y, pix = blockIdx.x * 4 + (threadIdx.x >> 6), sub = threadIdx.x & 63;
  if (pix >= g * g) return;
  const int y = pix / g, x = pix - y * g, S = 4 * g;
  const float* src = in + ((long long)p * S + 4 * y) * S + 4 * x;
  float a1[4][4];                                    // [position dy*2+dx][channel]
#pragma unroll
  for (int pos = 0; pos < 4; ++pos) {
    const int dy = pos >> 1, dx = pos & 1;
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float acc = c1b[c];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc += c1w[c * 4 + k] * src[(long long)(2 * dy + (k >> 1)) * S + 2 * dx + (k & 1)];
      v[c] = acc;
    }
    const float mean = ((v[0] + v[1]) + (v[2] + v[3])) * 0.25f;
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) var += (v[c] - mean) * (v[c] - mean);
    const float rs = 1.0f / sqrtf(var * 0.25f + 1e-6f);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float u = n1w[c] * ((v[c] - mean) * rs) + n1b[c];
      a1[pos][c] = 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f));
    }
  }
  float a2[16];
  float mean = 0.f;
#pragma unroll
  for (int c2 = 0; c2 < 16; ++c2) {
    float acc = c2b[c2];
#pragma unroll
    for (int c1 = 0; c1 < 4; ++c1)
#pragma unroll
      for (int pos = 0; pos < 4; ++pos) acc += c2w[(c2 * 4 + c1) * 4 + pos] * a1[pos][c1];
    a2[c2] = acc;
    mean += acc;
  }
  mean *= (1.0f / 16.0f);
  float var = 0.f;
#pragma unroll
  for (int c2 = 0; c2 < 16; ++c2) var += (a2[c2] - mean) * (a2[c2] - mean);
  const float rs = 1.0f / sqrtf(var * (1.0f / 16.0f) + 1e-6f);
#pragma unroll
  for (int c2 = 0; c2 < 16; ++c2) {
    const float u = n2w[c2] * ((a2[c2] - mean) * rs) + n2b[c2];
    a2[c2] = 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f));
  }
  float* dst = out + ((long long)p * g * g + pix) * 256;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = sub + 64 * j;
    float acc = c3b[c];
#pragma unroll
    for (int c2 = 0; c2 < 16; ++c2) acc += c3w[c * 16 + c2] * a2[c2];
    dst[c] = acc;
  }
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// LayerNorm over rows of 64 channels followed by erf-GELU: one wave per row; in place (hi == nullptr) or into the
// fp16 hi+lo pair that the following f16x3 GEMM reads
__global__ __launch_bounds__(256) void ln_gelu64_kernel(float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, long long rows,
                                                        float eps, _Float16* __restrict__ hi, _Float16* __restrict__ lo) {
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v = x[row * 64 + lane];
  const float mean = wsum(v) * (1.f / 64.f);
  const float d = v - mean;
  const float var = wsum(d * d) * (1.f / 64.f);
  v = d * rsqrtf(var + eps) * w[lane] + b[lane];
  const float o = hgl_gelu_erf(v);
  if (hi) {
    _Float16 h, l;
    hgl_split_hi_lo(o, h, l);
    hi[row * 64 + lane] = h;
    lo[row * 64 + lane] = l;
  } else {
    x[row * 64 + lane] = o;
  }
}

// LayerNorm of the decoder's image tokens (rows of 256) emitting what the next GEMMs read: optionally the fp32 row
// (in place; the residual of the next layer), the row as fp16 hi+lo, and row + positional encoding (pe[row % pe_rows])
// as hi+lo (TwoWayAttentionBlock: k = keys + key_pe, transformer.py:139-150).  One wave per row.
__global__ __launch_bounds__(256) void ln256_pe_split_kernel(float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, const float* __restrict__ pe,
                                                             int pe_rows, long long rows, float eps, int write_f32,
                                                             _Float16* __restrict__ kh, _Float16* __restrict__ kl,
                                                             _Float16* __restrict__ ph, _Float16* __restrict__ pl) {
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 v = ((const f32x4*)(x + row * 256))[lane];
  const float mean = wsum((v[0] + v[1]) + (v[2] + v[3])) * (1.f / 256.f);
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
  const float rstd = rsqrtf(wsum(q) * (1.f / 256.f) + eps);
  const f32x4 wv = ((const f32x4*)w)[lane], bv = ((const f32x4*)b)[lane];
  f32x4 pv = {0.f, 0.f, 0.f, 0.f};
  if (ph) pv = ((const f32x4*)(pe + (row % pe_rows) * 256))[lane];   // (uniform: ph == nullptr = no "+ pe" planes wanted)
  f32x4 o;
  f16x4g a, c, a2, c2;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o[e] = (v[e] - mean) * rstd * wv[e] + bv[e];
    _Float16 h0, l0, h1, l1;
    hgl_split_hi_lo(o[e], h0, l0);
    hgl_split_hi_lo(o[e] + pv[e], h1, l1);
    a[e] = h0; c[e] = l0;
    a2[e] = h1; c2[e] = l1;
  }
  if (write_f32) ((f32x4*)(x + row * 256))[lane] = o;
  ((f16x4g*)(kh + row * 256))[lane] = a;
  ((f16x4g*)(kl + row * 256))[lane] = c;
  if (ph) {
    ((f16x4g*)(ph + row * 256))[lane] = a2;
    ((f16x4g*)(pl + row * 256))[lane] = c2;
  }
}

// masks[p, t, Y, X] = hyper[p, t+1, :] . upscaled[p, pixel(Y,X), :]  (mask_decoder.py:146-151, multimask rows 1..3),
// written straight into the [P,3,4g,4g] low-res layout.  upscaled = u2 [P*g*g*16, 32]: one 128-byte row per output
// pixel, ordered (y, x, ky, kx, ky2, kx2) by the two stride-2 transposed convolutions.  Eight lanes share a row (one
// 16-byte load each: every 128-byte line is fetched by one coalesced access) and reduce with three shuffle steps;
// a wave covers 8 consecutive X, so the three plane writes are 32-byte segments.  Replaces a [.., 4]-column GEMM
// (a 128-wide MFMA tile 97 % empty, 431 us) plus the un-shuffle pass (46 us); HBM-bound: 537 MB read, 50 MB written.
__global__ __launch_bounds__(256) void hyper_logits_kernel(const float* __restrict__ u2, const float* __restrict__ hyper,
                                                           int g, int row0, float* __restrict__ out) {
  const int S4 = 4 * g;
  const int p = blockIdx.y, Y = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int sub = lane & 7, grp = lane >> 3;
  const float* hp = hyper + (long long)p * 4 * 32 + 32 * row0;       // row0 = 1: the multimask tokens 1..3; 0: tokens 0..2
  const f32x4 h1 = *(const f32x4*)(hp + 4 * sub), h2 = *(const f32x4*)(hp + 32 + 4 * sub),
              h3 = *(const f32x4*)(hp + 64 + 4 * sub);
  const int y = Y >> 2, ky = (Y >> 1) & 1, ky2 = Y & 1;
  const float* base = u2 + (long long)p * g * g * 16 * 32;
  float* o = out + ((long long)p * 3 * S4 + Y) * S4;
  const long long plane = (long long)S4 * S4;
  for (int X0 = wave * 8; X0 < S4; X0 += 32) {
    const int X = X0 + grp;
    const int x = X >> 2, kx = (X >> 1) & 1, kx2 = X & 1;
    const long long row = (((long long)y * g + x) * 4 + (ky * 2 + kx)) * 4 + (ky2 * 2 + kx2);
    const f32x4 v = *(const f32x4*)(base + row * 32 + 4 * sub);
    float a1 = v[0] * h1[0], a2 = v[0] * h2[0], a3 = v[0] * h3[0];
#pragma unroll
    for (int e = 1; e < 4; ++e) { a1 += v[e] * h1[e]; a2 += v[e] * h2[e]; a3 += v[e] * h3[e]; }
#pragma unroll
    for (int s2 = 1; s2 < 8; s2 <<= 1) {
      a1 += __shfl_xor(a1, s2); a2 += __shfl_xor(a2, s2); a3 += __shfl_xor(a3, s2);
    }
    if (sub == 0) { o[X] = a1; o[plane + X] = a2; o[2 * plane + X] = a3; }
  }
}

// ---------------------------------------------------------------------------------------
// Fused post-processing of the mask logits (never writes a full-resolution fp32 tensor):
// low-res logits [K][hl][wl] -> bilinear to S x S -> crop [hi, wi] -> bilinear to H x W
// (both align_corners=False, ATen source-index fma) -> mask byte (> thr), stability counters
// (> thr+off, > thr-off) and the inclusive XYXY box, per candidate.
struct PostArgs {
  const float* low;      // [K, hl, wl]
  const float* iou;      // [K] predicted IoU (candidates <= iou_thresh are skipped) or null
  float iou_thresh;
  int K, hl, wl, S, hi, wi, H, W;
  float thr, off;
  uint8_t* masks;        // [K, H, W]
  unsigned* counters;    // [K, 6]: inter, union, minx, miny, maxx, maxy  (pre-initialised)
  float* full_logits;    // optional [K, H, W] (tests only) or null
};

__device__ __forceinline__ void src_idx(float scale, int dst, int in_size, int& i0, int& i1, float& l0,
                                        float& l1) {
  float f = fmaf(scale, dst + 0.5f, -0.5f);
  f = f < 0.f ? 0.f : f;
  i0 = (int)f;
  i0 = i0 < in_size - 1 ? i0 : in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = f - i0;
  l0 = 1.f - l1;
}

// One workgroup = a 64x64 tile of output pixels of one candidate; a thread computes 16 consecutive pixels of one
// row and stores them as one 16-byte word.  The low-res logits the tile touches (a <= 48x48 patch for every
// realistic size ratio) are staged in LDS once, so the 16 taps per pixel are LDS reads instead of scattered global
// loads.  (The first version used 16x16 tiles with one pixel per thread: 300k workgroups whose fixed cost -- bounds,
// staging, two barriers, the reduction, atomics -- was 95 % of the 3.3 ms the kernel took.)
constexpr int PTW = 64, PTH = 64;   // output tile
constexpr int PPX = 16;             // pixels per thread (one row segment)
constexpr int PR = 48;              // max staged low-res patch side

__global__ __launch_bounds__(256) void sam_postprocess_kernel(PostArgs a) {
  __shared__ float patch[PR * PR];
  __shared__ unsigned red[6 * 4];
  // XCD-aware tile map.  Workgroups go to the eight XCDs round-robin in linear order (x fastest), so the ~100 tiles of ONE
  // candidate -- whose low-res patches overlap by their bilinear halos and share 128-byte lines -- sat on all eight L2s and
  // every L2 fetched the lines for itself: 120.5 MB fetched for 50.3 MB of low-res logits (profiles/r04i_pmc_traffic.json).
  // Remapped so that an XCD works through whole candidates: consecutive slots of one XCD are the tiles of one candidate.
  int bx = blockIdx.x, by = blockIdx.y, k = blockIdx.z;
  if ((gridDim.z & 7) == 0) {
    const unsigned nt = gridDim.x * gridDim.y;
    const unsigned Lb = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned c = Lb & 7u, j = Lb >> 3;
    const unsigned t = j % nt;
    k = (int)((j / nt) * 8u + c);
    by = (int)(t / gridDim.x);
    bx = (int)(t - (unsigned)by * gridDim.x);
  }
  const int tx = threadIdx.x & 3, ty = threadIdx.x >> 2;
  const int X0 = bx * PTW, Y0 = by * PTH;
  const int Xb = X0 + tx * PPX, Y = Y0 + ty;
  if (a.iou && a.iou_thresh > 0.f && !(a.iou[k] > a.iou_thresh)) {   // the filter exists only for thresholds > 0 (:287)
    // filtered before any pixel work (uniform): the candidate keeps an all-zero mask
    if (Xb < a.W && Y < a.H) {
      const int npx = min(PPX, a.W - Xb);
      const long long pix0 = (long long)k * a.H * a.W + (long long)Y * a.W + Xb;
      if (npx == PPX && (pix0 & 15) == 0) {
        const u32x4g z = {0, 0, 0, 0};
        *(u32x4g*)(a.masks + pix0) = z;
      } else {
        for (int p = 0; p < npx; ++p) a.masks[pix0 + p] = 0;
      }
    }
    return;
  }
  const int Xl = min(X0 + PTW - 1, a.W - 1), Yl = min(Y0 + PTH - 1, a.H - 1);
  const float* L = a.low + (long long)k * a.hl * a.wl;
  const float sy1 = (float)a.hi / (float)a.H, sx1 = (float)a.wi / (float)a.W;
  const float s2y = (float)a.hl / (float)a.S, s2x = (float)a.wl / (float)a.S;
  // low-res patch bounds of the tile: taps are monotone in the output coordinate
  int i0, i1, j0, j1;
  float f0, f1;
  src_idx(sy1, Y0, a.hi, i0, i1, f0, f1);
  int vb, vdummy;
  src_idx(s2y, i0, a.hl, vb, vdummy, f0, f1);
  src_idx(sy1, Yl, a.hi, i0, i1, f0, f1);
  int ve0, ve;
  src_idx(s2y, i1, a.hl, ve0, ve, f0, f1);
  src_idx(sx1, X0, a.wi, j0, j1, f0, f1);
  int ub, udummy;
  src_idx(s2x, j0, a.wl, ub, udummy, f0, f1);
  src_idx(sx1, Xl, a.wi, j0, j1, f0, f1);
  int ue0, ue;
  src_idx(s2x, j1, a.wl, ue0, ue, f0, f1);
  const int ph = ve - vb + 1, pw = ue - ub + 1;
  const bool staged = ph <= PR && pw <= PR;      // uniform
  if (staged) {
    for (int v = threadIdx.x >> 6; v < ph; v += 4) {       // (row, column) by shifts: no integer division per element
      const int u = threadIdx.x & 63;
      if (u < pw) patch[v * PR + u] = L[(long long)(vb + v) * a.wl + (ub + u)];
    }
  }
  __syncthreads();
  auto ld = [&](int v, int u) -> float {
    return staged ? patch[(v - vb) * PR + (u - ub)] : L[(long long)v * a.wl + u];
  };

  const long long HW = (long long)a.H * a.W;
  unsigned inter = 0, uni = 0, minx = 0x7fffffff, miny = 0x7fffffff, maxx = 0, maxy = 0, any = 0;
  if (Xb < a.W && Y < a.H) {
    // row-dependent part, shared by the thread's pixels
    int y0, y1;
    float ly0, ly1;
    src_idx(sy1, Y, a.hi, y0, y1, ly0, ly1);
    const int tyv[2] = {y0, y1};
    int v0[2], v1[2];
    float m0[2], m1[2];
#pragma unroll
    for (int iy = 0; iy < 2; ++iy) src_idx(s2y, tyv[iy], a.hl, v0[iy], v1[iy], m0[iy], m1[iy]);
    unsigned bytes[PPX / 4] = {0, 0, 0, 0};
    const int npx = min(PPX, a.W - Xb);
    const long long pix0 = (long long)k * HW + (long long)Y * a.W + Xb;
#pragma unroll
    for (int p = 0; p < PPX; ++p) {
      if (p < npx) {
        const int X = Xb + p;
        int x0, x1;
        float lx0, lx1;
        src_idx(sx1, X, a.wi, x0, x1, lx0, lx1);
        float tap[2][2];
        const int txv[2] = {x0, x1};
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
          int u0, u1;
          float n0, n1;
          src_idx(s2x, txv[ix], a.wl, u0, u1, n0, n1);
#pragma unroll
          for (int iy = 0; iy < 2; ++iy) {
            const float top = ld(v0[iy], u0) * n0 + ld(v0[iy], u1) * n1;
            const float bot = ld(v1[iy], u0) * n0 + ld(v1[iy], u1) * n1;
            tap[iy][ix] = top * m0[iy] + bot * m1[iy];
          }
        }
        const float top = tap[0][0] * lx0 + tap[0][1] * lx1;
        const float bot = tap[1][0] * lx0 + tap[1][1] * lx1;
        const float v = top * ly0 + bot * ly1;
        const bool on = v > a.thr;
        bytes[p >> 2] |= (on ? 1u : 0u) << (8 * (p & 3));
        if (a.full_logits) a.full_logits[pix0 + p] = v;
        inter += v > a.thr + a.off;
        uni += v > a.thr - a.off;
        if (on) { minx = min(minx, (unsigned)X); maxx = max(maxx, (unsigned)X); any = 1; }
      }
    }
    if (any) miny = maxy = Y;
    if (npx == PPX && (pix0 & 15) == 0) {
      u32x4g w; w[0] = bytes[0]; w[1] = bytes[1]; w[2] = bytes[2]; w[3] = bytes[3];
      *(u32x4g*)(a.masks + pix0) = w;
    } else {
      for (int p = 0; p < npx; ++p) a.masks[pix0 + p] = (uint8_t)((bytes[p >> 2] >> (8 * (p & 3))) & 1u);
    }
  }
  // wave reduction, then one set of atomics per workgroup
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    inter += __shfl_xor(inter, o);
    uni += __shfl_xor(uni, o);
    minx = min(minx, (unsigned)__shfl_xor(minx, o));
    miny = min(miny, (unsigned)__shfl_xor(miny, o));
    maxx = max(maxx, (unsigned)__shfl_xor(maxx, o));
    maxy = max(maxy, (unsigned)__shfl_xor(maxy, o));
    any |= __shfl_xor(any, o);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[wave * 6 + 0] = inter; red[wave * 6 + 1] = uni;
    red[wave * 6 + 2] = any ? minx : 0x7fffffffu; red[wave * 6 + 3] = any ? miny : 0x7fffffffu;
    red[wave * 6 + 4] = any ? maxx : 0u; red[wave * 6 + 5] = any ? (maxy | 0x80000000u) : 0u;  // top bit: "has pixels"
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned I = 0, U = 0, mnx = 0x7fffffff, mny = 0x7fffffff, mxx = 0, mxy = 0, has = 0;
    for (int w = 0; w < 4; ++w) {
      I += red[w * 6]; U += red[w * 6 + 1];
      mnx = min(mnx, red[w * 6 + 2]); mny = min(mny, red[w * 6 + 3]);
      if (red[w * 6 + 5] & 0x80000000u) { has = 1; mxx = max(mxx, red[w * 6 + 4]); mxy = max(mxy, red[w * 6 + 5] & 0x7fffffffu); }
    }
    unsigned* c = a.counters + (long long)k * 6;
    if (I) atomicAdd(&c[0], I);
    if (U) atomicAdd(&c[1], U);
    if (has) {
      atomicMin(&c[2], mnx); atomicMin(&c[3], mny);
      atomicMax(&c[4], mxx); atomicMax(&c[5], mxy);
    }
  }
}

// The same arithmetic with the work shared inside a tile.  Stage 1 (low-res -> S x S) interpolates horizontally first:
// h(v, x1) = L[v][u0] n0 + L[v][u1] n1 depends only on the low-res row v and the stage-1 column x1, and every output pixel
// of a tile column uses the same two stage-1 columns -- the kernel above evaluates it 8 times per output pixel, with three
// source-index computations in front.  Here a tile first builds the table of its output columns (x0, x1, weights), the
// table of its stage-1 columns (u0, u1, weights) and h over (low-res rows of the patch) x (stage-1 columns of the tile) in
// LDS; an output pixel is then 8 LDS reads and the two vertical / one horizontal blends, expression for expression those
// of the kernel above (same products, same order, same contraction: the logits are bit-identical, tests/test_gpu_sam.py).
// ~40 instead of ~90 instructions per pixel.  The host picks this kernel when every tile's patch / column count fits.
constexpr int PX1 = 112;            // max stage-1 columns of a tile (64 output columns at a 1.6 : 1 size ratio: 104).  Wider
                                    // tables (192 columns = 47 KB of LDS, for the 2.7 : 1 crops of a crop layer) measured SLOWER
                                    // than the per-pixel kernel there (3.1 against 2.0 ms per crop): three workgroups per CU
                                    // do not hide the load -> build -> interpolate chain of a tile.  Round 6: 32-column tiles
                                    // (88 stage-1 columns: these tables, 128 threads) on the same crops: 3.1 ms again -- the
                                    // table of a tile costs what its 2048 pixels save

__global__ __launch_bounds__(256) void sam_postprocess_sep_kernel(PostArgs a) {
  __shared__ float patch[PR * PR];
  __shared__ float H1[PR * PX1];
  __shared__ int xc0[PTW], xc1[PTW];
  __shared__ float xl0[PTW], xl1[PTW];
  __shared__ unsigned red[6 * 4];
  // XCD-aware tile map (see sam_postprocess_kernel): an XCD works through whole candidates
  int bx = blockIdx.x, by = blockIdx.y, k = blockIdx.z;
  if ((gridDim.z & 7) == 0) {
    const unsigned nt = gridDim.x * gridDim.y;
    const unsigned Lb = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned c = Lb & 7u, j = Lb >> 3;
    const unsigned t = j % nt;
    k = (int)((j / nt) * 8u + c);
    by = (int)(t / gridDim.x);
    bx = (int)(t - (unsigned)by * gridDim.x);
  }
  const int tx = threadIdx.x & 3, ty = threadIdx.x >> 2;
  const int X0 = bx * PTW, Y0 = by * PTH;
  const int Xb = X0 + tx * PPX, Y = Y0 + ty;
  if (a.iou && a.iou_thresh > 0.f && !(a.iou[k] > a.iou_thresh)) {
    if (Xb < a.W && Y < a.H) {
      const int npx = min(PPX, a.W - Xb);
      const long long pix0 = (long long)k * a.H * a.W + (long long)Y * a.W + Xb;
      if (npx == PPX && (pix0 & 15) == 0) {
        const u32x4g z = {0, 0, 0, 0};
        *(u32x4g*)(a.masks + pix0) = z;
      } else {
        for (int p = 0; p < npx; ++p) a.masks[pix0 + p] = 0;
      }
    }
    return;
  }
  const int Xl = min(X0 + PTW - 1, a.W - 1), Yl = min(Y0 + PTH - 1, a.H - 1);
  const float* L = a.low + (long long)k * a.hl * a.wl;
  const float sy1 = (float)a.hi / (float)a.H, sx1 = (float)a.wi / (float)a.W;
  const float s2y = (float)a.hl / (float)a.S, s2x = (float)a.wl / (float)a.S;
  int i0, i1, j0, j1;
  float f0, f1;
  src_idx(sy1, Y0, a.hi, i0, i1, f0, f1);
  int vb, vdummy;
  src_idx(s2y, i0, a.hl, vb, vdummy, f0, f1);
  src_idx(sy1, Yl, a.hi, i0, i1, f0, f1);
  int ve0, ve;
  src_idx(s2y, i1, a.hl, ve0, ve, f0, f1);
  src_idx(sx1, X0, a.wi, j0, j1, f0, f1);
  const int X1b = j0;                                    // first stage-1 column of the tile
  int ub, udummy;
  src_idx(s2x, j0, a.wl, ub, udummy, f0, f1);
  src_idx(sx1, Xl, a.wi, j0, j1, f0, f1);
  // (the host launches this kernel only when these fit, with the same IEEE arithmetic; the clamps keep a disagreement -- which
  // cannot happen -- inside the tables)
  const int R1w = min(j1 - X1b + 1, PX1);                // stage-1 columns of the tile
  int ue0, ue;
  src_idx(s2x, j1, a.wl, ue0, ue, f0, f1);
  const int ph = min(ve - vb + 1, PR), pw = min(ue - ub + 1, PR);   // low-res patch
  // (thread -> (row, column) by shifts: an integer division per element cost more than the element)
  for (int v = threadIdx.x >> 6; v < ph; v += 4) {
    const int u = threadIdx.x & 63;
    if (u < pw) patch[v * PR + u] = L[(long long)(vb + v) * a.wl + (ub + u)];
  }
  if (threadIdx.x < PTW) {
    int x0, x1;
    float lx0, lx1;
    src_idx(sx1, min(X0 + (int)threadIdx.x, a.W - 1), a.wi, x0, x1, lx0, lx1);
    xc0[threadIdx.x] = x0 - X1b; xc1[threadIdx.x] = x1 - X1b;
    xl0[threadIdx.x] = lx0; xl1[threadIdx.x] = lx1;
  }
  __syncthreads();
  {
    // h(v, c) for the patch rows x the tile's stage-1 columns: a thread owns one column (its source indices and weights in
    // registers) and walks the rows
    const int cshift = R1w <= 64 ? 6 : (R1w <= 128 ? 7 : 8);
    const int c = threadIdx.x & ((1 << cshift) - 1), vstep = 256 >> cshift;
    if (c < R1w) {
      int u0, u1;
      float n0, n1;
      src_idx(s2x, X1b + c, a.wl, u0, u1, n0, n1);
      u0 -= ub; u1 -= ub;
      for (int v = threadIdx.x >> cshift; v < ph; v += vstep) {
        const float h1 = patch[v * PR + u0] * n0 + patch[v * PR + u1] * n1;
        H1[v * PX1 + c] = h1;
      }
    }
  }
  __syncthreads();

  const long long HW = (long long)a.H * a.W;
  unsigned inter = 0, uni = 0, minx = 0x7fffffff, miny = 0x7fffffff, maxx = 0, maxy = 0, any = 0;
  if (Xb < a.W && Y < a.H) {
    int y0, y1;
    float ly0, ly1;
    src_idx(sy1, Y, a.hi, y0, y1, ly0, ly1);
    const int tyv[2] = {y0, y1};
    float m0[2], m1[2];
    const float* r0[2];
    const float* r1[2];
#pragma unroll
    for (int iy = 0; iy < 2; ++iy) {
      int v0, v1;
      src_idx(s2y, tyv[iy], a.hl, v0, v1, m0[iy], m1[iy]);
      r0[iy] = H1 + (v0 - vb) * PX1;
      r1[iy] = H1 + (v1 - vb) * PX1;
    }
    unsigned bytes[PPX / 4] = {0, 0, 0, 0};
    const int npx = min(PPX, a.W - Xb);
    const long long pix0 = (long long)k * HW + (long long)Y * a.W + Xb;
#pragma unroll
    for (int p = 0; p < PPX; ++p) {
      if (p < npx) {
        const int X = Xb + p, xi = tx * PPX + p;
        const float lx0 = xl0[xi], lx1 = xl1[xi];
        const int txc[2] = {xc0[xi], xc1[xi]};
        float tap[2][2];
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
#pragma unroll
          for (int iy = 0; iy < 2; ++iy) {
            const float top = r0[iy][txc[ix]];
            const float bot = r1[iy][txc[ix]];
            tap[iy][ix] = top * m0[iy] + bot * m1[iy];
          }
        }
        const float top = tap[0][0] * lx0 + tap[0][1] * lx1;
        const float bot = tap[1][0] * lx0 + tap[1][1] * lx1;
        const float v = top * ly0 + bot * ly1;
        const bool on = v > a.thr;
        bytes[p >> 2] |= (on ? 1u : 0u) << (8 * (p & 3));
        if (a.full_logits) a.full_logits[pix0 + p] = v;
        inter += v > a.thr + a.off;
        uni += v > a.thr - a.off;
        if (on) { minx = min(minx, (unsigned)X); maxx = max(maxx, (unsigned)X); any = 1; }
      }
    }
    if (any) miny = maxy = Y;
    if (npx == PPX && (pix0 & 15) == 0) {
      u32x4g w; w[0] = bytes[0]; w[1] = bytes[1]; w[2] = bytes[2]; w[3] = bytes[3];
      *(u32x4g*)(a.masks + pix0) = w;
    } else {
      for (int p = 0; p < npx; ++p) a.masks[pix0 + p] = (uint8_t)((bytes[p >> 2] >> (8 * (p & 3))) & 1u);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    inter += __shfl_xor(inter, o);
    uni += __shfl_xor(uni, o);
    minx = min(minx, (unsigned)__shfl_xor(minx, o));
    miny = min(miny, (unsigned)__shfl_xor(miny, o));
    maxx = max(maxx, (unsigned)__shfl_xor(maxx, o));
    maxy = max(maxy, (unsigned)__shfl_xor(maxy, o));
    any |= __shfl_xor(any, o);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[wave * 6 + 0] = inter; red[wave * 6 + 1] = uni;
    red[wave * 6 + 2] = any ? minx : 0x7fffffffu; red[wave * 6 + 3] = any ? miny : 0x7fffffffu;
    red[wave * 6 + 4] = any ? maxx : 0u; red[wave * 6 + 5] = any ? (maxy | 0x80000000u) : 0u;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned I = 0, U = 0, mnx = 0x7fffffff, mny = 0x7fffffff, mxx = 0, mxy = 0, has = 0;
    for (int w = 0; w < 4; ++w) {
      I += red[w * 6]; U += red[w * 6 + 1];
      mnx = min(mnx, red[w * 6 + 2]); mny = min(mny, red[w * 6 + 3]);
      if (red[w * 6 + 5] & 0x80000000u) { has = 1; mxx = max(mxx, red[w * 6 + 4]); mxy = max(mxy, red[w * 6 + 5] & 0x7fffffffu); }
    }
    unsigned* c = a.counters + (long long)k * 6;
    if (I) atomicAdd(&c[0], I);
    if (U) atomicAdd(&c[1], U);
    if (has) {
      atomicMin(&c[2], mnx); atomicMin(&c[3], mny);
      atomicMax(&c[4], mxx); atomicMax(&c[5], mxy);
    }
  }
}

// host twin of src_idx (the same IEEE operations) and the test "does every tile of this geometry fit the shared tables"
static void src_idx_host(float scale, int dst, int in_size, int& i0, int& i1) {
  float f = fmaf(scale, dst + 0.5f, -0.5f);
  f = f < 0.f ? 0.f : f;
  i0 = (int)f;
  i0 = i0 < in_size - 1 ? i0 : in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
}
static bool postprocess_sep_fits(int out, int in1, int low, int S) {
  const float s1 = (float)in1 / (float)out, s2 = (float)low / (float)S;
  for (int o0 = 0; o0 < out; o0 += PTW) {
    const int ol = (o0 + PTW - 1 < out ? o0 + PTW - 1 : out - 1);
    int a0, a1, b0, b1, u0, u1, e0, e1;
    src_idx_host(s1, o0, in1, a0, a1);
    src_idx_host(s1, ol, in1, b0, b1);
    if (b1 - a0 + 1 > PX1) return false;
    src_idx_host(s2, a0, low, u0, u1);
    src_idx_host(s2, b1, low, e0, e1);
    if (e1 - u0 + 1 > PR) return false;
  }
  return true;
}

__global__ void init_counters_kernel(unsigned* c, int K) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  c[k * 6 + 0] = 0; c[k * 6 + 1] = 0;
  c[k * 6 + 2] = 0x7fffffff; c[k * 6 + 3] = 0x7fffffff;
  c[k * 6 + 4] = 0; c[k * 6 + 5] = 0;
}

// per candidate: stability = inter/union, XYXY box ([0,0,0,0] when empty), keep flag
__global__ void sam_finalize_kernel(const unsigned* __restrict__ c, const float* __restrict__ iou, int K,
                                    float iou_thresh, float stab_thresh, float* __restrict__ stab,
                                    int* __restrict__ boxes, uint8_t* __restrict__ keep) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  // automatic_mask_generator.py:287-298: each filter is applied only when its threshold is > 0 (so a NaN
  // stability score -- an empty mask -- survives a zero threshold, as in the reference)
  const bool pass_iou = !iou || !(iou_thresh > 0.f) || iou[k] > iou_thresh;
  const float s = (float)c[k * 6 + 0] / (float)c[k * 6 + 1];  // 0/0 -> NaN as in the reference
  stab[k] = pass_iou ? s : 0.f;
  const bool empty = c[k * 6 + 2] == 0x7fffffff;
  boxes[k * 4 + 0] = empty ? 0 : (int)c[k * 6 + 2];
  boxes[k * 4 + 1] = empty ? 0 : (int)c[k * 6 + 3];
  boxes[k * 4 + 2] = empty ? 0 : (int)c[k * 6 + 4];
  boxes[k * 4 + 3] = empty ? 0 : (int)c[k * 6 + 5];
  keep[k] = (pass_iou && (!(stab_thresh > 0.f) || s >= stab_thresh)) ? 1 : 0;  // NaN >= x is false
}

// The ranking order of every NMS kernel below: descending score, the original index breaks ties, and a NaN score ranks ABOVE
// every number (where torch.sort(descending=True) -- batched_nms's argsort, automatic_mask_generator.py:251-257 -- puts it).
// A total order: ranks by counting never collide, order[] has no holes below the number of valid candidates.  (With a plain
// `sj > sc || (sj == sc && j < i)` two NaN scores -- an f16x3 overflow with the thresholds open -- both got rank 0.)
__device__ __forceinline__ bool nms_before(float sj, int j, float si, int i) {
  const bool nj = sj != sj, ni = si != si;
  if (nj || ni) return nj && (!ni || j < i);
  return sj > si || (sj == si && j < i);
}

// Greedy NMS in one workgroup (K <= 1024): candidates with keep[k]!=0, descending score with the
// original index as tie-break (stable sort), suppress IoU > thr.  out_idx[0..n) in kept order.
__global__ __launch_bounds__(1024) void nms_kernel(const int* __restrict__ boxes,
                                                   const float* __restrict__ scores,
                                                   const uint8_t* __restrict__ keep, int K, float thr,
                                                   int* __restrict__ out_idx, int* __restrict__ out_n) {
  __shared__ int order[1024];
  __shared__ unsigned char alive[1024];
  __shared__ int cur, nkept;
  const int t = threadIdx.x;
  int valid = 0;
  float sc = 0.f;
  if (t < K && keep[t]) { valid = 1; sc = scores[t]; }
  // rank by counting (K small): position among valid candidates
  if (t < 1024) order[t] = -1;
  __syncthreads();
  if (valid) {
    int rank = 0;
    for (int j = 0; j < K; ++j) {
      if (!keep[j]) continue;
      const float sj = scores[j];
      if (nms_before(sj, j, sc, t)) ++rank;
    }
    order[rank] = t;
  }
  if (t < 1024) alive[t] = 1;
  if (t == 0) nkept = 0;
  __syncthreads();
  int nvalid = 0;
  for (int j = 0; j < K; ++j) nvalid += keep[j] ? 1 : 0;  // uniform
  for (int r = 0; r < nvalid; ++r) {
    if (t == 0) cur = alive[r] ? order[r] : -1;
    __syncthreads();
    const int ci = cur;
    if (ci >= 0) {
      if (t == 0) { out_idx[nkept] = ci; ++nkept; }
      // suppress lower-ranked boxes overlapping ci
      const int me = (t > r && t < nvalid) ? order[t] : -1;
      if (me >= 0 && alive[t]) {
        const float ax0 = boxes[ci * 4], ay0 = boxes[ci * 4 + 1], ax1 = boxes[ci * 4 + 2], ay1 = boxes[ci * 4 + 3];
        const float bx0 = boxes[me * 4], by0 = boxes[me * 4 + 1], bx1 = boxes[me * 4 + 2], by1 = boxes[me * 4 + 3];
        const float iw = fmaxf(fminf(ax1, bx1) - fmaxf(ax0, bx0), 0.f);
        const float ih = fmaxf(fminf(ay1, by1) - fmaxf(ay0, by0), 0.f);
        const float inter = iw * ih;
        const float iou = inter / ((ax1 - ax0) * (ay1 - ay0) + (bx1 - bx0) * (by1 - by0) - inter);
        if (iou > thr) alive[t] = 0;
      }
    }
    __syncthreads();
  }
  if (t == 0) *out_n = nkept;
}

// ---- NMS for any K (crop layers / dense point grids, automatic_mask_generator.py:209-220,259-266) ----------
// Same semantics as nms_kernel (descending score, original index breaks ties, suppress IoU > thr), in three passes:
// rank by counting -> 64x64-bit suppression words of the sorted boxes -> one workgroup walks the row blocks,
// resolving each 64-row block with wave shuffles and OR-ing the kept rows into the running "removed" bit set.
__global__ __launch_bounds__(256) void nms_rank_kernel(const float* __restrict__ scores, const uint8_t* __restrict__ keep,
                                                       int K, int* __restrict__ order, int* __restrict__ nvalid) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= K || !keep[i]) return;
  const float sc = scores[i];
  int rank = 0;
  for (int j = 0; j < K; ++j) {
    if (!keep[j]) continue;
    const float sj = scores[j];
    rank += nms_before(sj, j, sc, i) ? 1 : 0;
  }
  order[rank] = i;
  atomicAdd(nvalid, 1);
}

__device__ __forceinline__ bool nms_overlap(const int4 a, const int4 b, float thr) {
  const float ax0 = (float)a.x, ay0 = (float)a.y, ax1 = (float)a.z, ay1 = (float)a.w;
  const float bx0 = (float)b.x, by0 = (float)b.y, bx1 = (float)b.z, by1 = (float)b.w;
  const float iw = fmaxf(fminf(ax1, bx1) - fmaxf(ax0, bx0), 0.f);
  const float ih = fmaxf(fminf(ay1, by1) - fmaxf(ay0, by0), 0.f);
  const float inter = iw * ih;
  const float iou = inter / ((ax1 - ax0) * (ay1 - ay0) + (bx1 - bx0) * (by1 - by0) - inter);
  return iou > thr;
}

// nms_kernel's semantics for K <= 512 without its serial loop of K iterations x (two barriers + global loads of the current
// box: 82 us for the 192 candidates of an image): ranks by counting, the boxes in rank order in LDS, every thread the
// suppression words of its row (the same IoU expression: nms_overlap), then ONE thread walks the rows OR-ing the kept ones
// into the removed set -- the scheme of the any-K path below in one workgroup.
__global__ __launch_bounds__(512) void nms_bits_kernel(const int* __restrict__ boxes, const float* __restrict__ scores,
                                                       const uint8_t* __restrict__ keep, int K, float thr,
                                                       int* __restrict__ out_idx, int* __restrict__ out_n) {
  __shared__ int4 sbox[512];
  __shared__ int order[512];
  __shared__ float ssc[512];
  __shared__ unsigned char skeep[512];
  __shared__ unsigned long long mask[512][8];
  __shared__ unsigned long long removed[8];
  __shared__ unsigned long long kept_word;
  __shared__ int nv, nkept_s;
  const int t = threadIdx.x, lane = t & 63;
  const bool valid = t < K && keep[t];
  const float sc = valid ? scores[t] : 0.f;
  ssc[t] = sc;
  skeep[t] = valid ? 1 : 0;
  order[t] = -1;
  if (t < 8) removed[t] = 0;
  if (t == 0) { nv = 0; nkept_s = 0; }
  __syncthreads();
  if (valid) {      // rank by counting: position among the valid candidates (descending score, the index breaks ties)
    int rank = 0;
#pragma unroll 8
    for (int j = 0; j < K; ++j) {
      const float sj = ssc[j];
      rank += (skeep[j] && nms_before(sj, j, sc, t)) ? 1 : 0;
    }
    order[rank] = t;
    atomicAdd(&nv, 1);
  }
  __syncthreads();
  const int n = nv;
  if (t < n) sbox[t] = ((const int4*)boxes)[order[t]];
  __syncthreads();
  const int nw = (n + 63) >> 6;
  // word w is needed of rows 0 .. min(n, 64 (w + 1)) - 1 (words left of a row's diagonal block are never read): the (word, row)
  // pairs are dealt to all 512 threads -- 384 pairs of 64 IoUs for 192 boxes, not three words for each of 192 threads
  for (int w = 0, q0 = 0; w < nw; ++w) {
    const int rows = min(n, 64 * (w + 1));
    for (int q = t - (q0 & 511); q < rows; q += 512) {
      if (q < 0) continue;
      const int4 me = sbox[q];
      unsigned long long word = 0;
      const int b0 = 64 * w, jn = min(64, n - b0);
#pragma unroll 4
      for (int j = 0; j < jn; ++j)
        if (b0 + j > q && nms_overlap(me, sbox[b0 + j], thr)) word |= 1ull << j;
      mask[q][w] = word;
    }
    q0 += rows;
  }
  __syncthreads();
  // nms_scan_kernel's walk on the LDS-resident words: wave 0 resolves a block of 64 rows with shuffles, then every later word
  // takes the OR of the kept rows
  for (int rb = 0; rb < nw; ++rb) {
    if (t < 64) {
      const int a = rb * 64 + lane;
      const unsigned long long d = a < n ? mask[a][rb] : 0ull;
      unsigned long long rem = removed[rb], km = 0;
      for (int s2 = 0; s2 < 64; ++s2) {
        const unsigned lo = __shfl((unsigned)(d & 0xffffffffull), s2), hi = __shfl((unsigned)(d >> 32), s2);
        if (rb * 64 + s2 < n && !((rem >> s2) & 1ull)) {
          km |= 1ull << s2;
          rem |= ((unsigned long long)hi << 32) | lo;
        }
      }
      const int base = nkept_s;
      if ((km >> lane) & 1ull) out_idx[base + __popcll(km & ((1ull << lane) - 1ull))] = order[a];
      if (lane == 0) { kept_word = km; nkept_s = base + __popcll(km); }
    }
    __syncthreads();
    const unsigned long long km = kept_word;
    if (t > rb && t < nw) {
      unsigned long long acc = removed[t];
      unsigned long long bits = km;
      while (bits) {
        const int s2 = __ffsll((long long)bits) - 1;
        bits &= bits - 1;
        acc |= mask[rb * 64 + s2][t];
      }
      removed[t] = acc;
    }
    __syncthreads();
  }
  if (t == 0) *out_n = nkept_s;
}

// grid (W, W), 64 threads: word (row a = 64*by + t, column block bx) of the upper triangle
__global__ __launch_bounds__(64) void nms_mask_kernel(const int* __restrict__ boxes, const int* __restrict__ order,
                                                      const int* __restrict__ nvalid, int W, float thr,
                                                      unsigned long long* __restrict__ mask) {
  const int cb = blockIdx.x, rb = blockIdx.y, t = threadIdx.x;
  const int n = *nvalid;
  if (cb < rb || rb * 64 >= n) return;
  __shared__ int4 colbox[64];
  const int b = cb * 64 + t;
  colbox[t] = b < n ? ((const int4*)boxes)[order[b]] : make_int4(0, 0, 0, 0);
  __syncthreads();
  const int a = rb * 64 + t;
  unsigned long long w = 0;
  if (a < n) {
    const int4 me = ((const int4*)boxes)[order[a]];
    const int jmax = min(64, n - cb * 64);
    for (int j = 0; j < jmax; ++j) {
      if (cb * 64 + j > a && nms_overlap(me, colbox[j], thr)) w |= 1ull << j;
    }
  }
  mask[(long long)a * W + cb] = w;
}

__global__ __launch_bounds__(256) void nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                       const int* __restrict__ order, const int* __restrict__ nvalid,
                                                       int W, int* __restrict__ out_idx, int* __restrict__ out_n) {
  __shared__ unsigned long long removed[1024];
  __shared__ unsigned long long kept_word;
  __shared__ int nkept_s;
  const int t = threadIdx.x, lane = t & 63;
  const int n = *nvalid;
  for (int w = t; w < W; w += 256) removed[w] = 0;
  if (t == 0) nkept_s = 0;
  __syncthreads();
  const int nblk = (n + 63) / 64;
  for (int rb = 0; rb < nblk; ++rb) {
    if (t < 64) {
      const int a = rb * 64 + lane;
      const unsigned long long d = a < n ? mask[(long long)a * W + rb] : 0ull;
      unsigned long long rem = removed[rb], km = 0;
      for (int s2 = 0; s2 < 64; ++s2) {
        const unsigned lo = __shfl((unsigned)(d & 0xffffffffull), s2), hi = __shfl((unsigned)(d >> 32), s2);
        if (rb * 64 + s2 < n && !((rem >> s2) & 1ull)) {
          km |= 1ull << s2;
          rem |= ((unsigned long long)hi << 32) | lo;
        }
      }
      const int base = nkept_s;
      if ((km >> lane) & 1ull) out_idx[base + __popcll(km & ((1ull << lane) - 1ull))] = order[a];
      if (lane == 0) { kept_word = km; nkept_s = base + __popcll(km); }
    }
    __syncthreads();
    const unsigned long long km = kept_word;
    for (int w = rb + 1 + t; w < W; w += 256) {
      unsigned long long acc = removed[w];
      unsigned long long bits = km;
      while (bits) {
        const int s2 = __ffsll((long long)bits) - 1;
        bits &= bits - 1;
        acc |= mask[(long long)(rb * 64 + s2) * W + w];
      }
      removed[w] = acc;
    }
    __syncthreads();
  }
  if (t == 0) *out_n = nkept_s;
}

// is_box_near_crop_edge (utils/amg.py:78-88) applied to keep flags: boxes are in crop coordinates
__global__ __launch_bounds__(256) void crop_edge_kernel(const int* __restrict__ boxes, int K, int cx0, int cy0, int cx1,
                                                        int cy1, int ox0, int oy0, int ox1, int oy1, float atol,
                                                        uint8_t* __restrict__ keep) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= K) return;
  const float b[4] = {(float)(boxes[i * 4] + cx0), (float)(boxes[i * 4 + 1] + cy0), (float)(boxes[i * 4 + 2] + cx0),
                      (float)(boxes[i * 4 + 3] + cy0)};
  const float c[4] = {(float)cx0, (float)cy0, (float)cx1, (float)cy1};
  const float o[4] = {(float)ox0, (float)oy0, (float)ox1, (float)oy1};
  bool near = false;
#pragma unroll
  for (int e = 0; e < 4; ++e) near |= (fabsf(b[e] - c[e]) <= atol) && !(fabsf(b[e] - o[e]) <= atol);
  if (near) keep[i] = 0;
}

// dst[i] = src[idx[i]] for i < *n (rows of row_bytes bytes, 16-byte multiples)
__global__ __launch_bounds__(256) void gather_masks_kernel(const uint8_t* __restrict__ src,
                                                           const int* __restrict__ idx,
                                                           const int* __restrict__ n, long long row16,
                                                           uint8_t* __restrict__ dst) {
  const int i = blockIdx.y;
  if (i >= *n) return;
  const uint4* s = (const uint4*)(src + (long long)idx[i] * row16 * 16);
  uint4* d = (uint4*)(dst + (long long)i * row16 * 16);
  for (long long j = blockIdx.x * 256ll + threadIdx.x; j < row16; j += (long long)gridDim.x * 256) d[j] = s[j];
}

}  // namespace

// ------------------------------------------------------------------------------ launchers
int hgl_launch_sam_preprocess(const uint8_t* img, int h, int w, int S, float* out, hipStream_t st) {
  hipLaunchKernelGGL(sam_preprocess_kernel, dim3(grid1(3ll * S * S)), dim3(256), 0, st, img, h, w, S, out);
  return hgl_check_launch("sam_preprocess");
}
int hgl_launch_win_partition(const float* H, int g, int ws, int nw, int D, float* Hw, hipStream_t st) {
  const long long total4 = (long long)nw * nw * ws * ws * (D / 4);
  hipLaunchKernelGGL(win_partition_kernel, dim3(grid1(total4)), dim3(256), 0, st, H, g, ws, nw, D / 4, Hw, total4);
  return hgl_check_launch("win_partition");
}
int hgl_launch_win_unpartition_add(float* X, int g, int ws, int nw, int D, const float* P, hipStream_t st) {
  const long long total4 = (long long)g * g * (D / 4);
  hipLaunchKernelGGL(win_unpartition_add_kernel, dim3(grid1(total4)), dim3(256), 0, st, X, g, ws, nw, D / 4, P, total4);
  return hgl_check_launch("win_unpartition_add");
}
int hgl_launch_relpos_gather(const float* T, int B, int heads, int S, int size, int L, int use_w,
                             float* rel, hipStream_t st) {
  const long long total = (long long)B * heads * S * size;
  hipLaunchKernelGGL(relpos_gather_kernel, dim3(grid1(total)), dim3(256), 0, st, T, B, heads, S, size, L, use_w, rel, total);
  return hgl_check_launch("relpos_gather");
}
int hgl_launch_relpos_direct(const float* qkv, int ldq, int B, int heads, int S, int size, int hd,
                             const float* Rh, const float* Rw, float* rel_h, float* rel_w, hipStream_t st) {
  HGL_REQUIRE(hd == 80 || hd == 64, "relpos_direct: head dim %d unsupported", hd);
  HGL_REQUIRE((size & 1) == 0 && size > 0 && size <= 64, "relpos_direct: window size %d unsupported (even, <= 64)", size);
  // f16x3 mode: the matrix-core version (table of 2*size-1 rows padded to 32 / 128)
  if (hgl_precision() == HGL_PREC_F16X3 && (size == 14 || size == 64)) {
    const int NT = size == 14 ? 1 : 4;
    const dim3 gridm((unsigned)((S + 127) / 128), (unsigned)(B * heads), 1);   // both axes in one workgroup: q is read once
    const size_t ldsm = (size_t)2 * NT * 32 * (hd + 8) * sizeof(_Float16) + (size_t)4 * 32 * (NT * 32 + 1) * sizeof(float);
#define HGL_RP_LAUNCH(HD_, NT_)                                                                                          \
  do {                                                                                                                 \
    HGL_RESERVE_LDS((relpos_mfma_kernel<HD_, NT_>), ldsm, "relpos");                                                    \
    hipLaunchKernelGGL((relpos_mfma_kernel<HD_, NT_>), gridm, dim3(256), ldsm, st, qkv, ldq, heads, S, size, Rh, Rw, rel_h, rel_w); \
  } while (0)
    if (hd == 80 && NT == 1) HGL_RP_LAUNCH(80, 1);
    else if (hd == 80) HGL_RP_LAUNCH(80, 4);
    else if (NT == 1) HGL_RP_LAUNCH(64, 1);
    else HGL_RP_LAUNCH(64, 4);
#undef HGL_RP_LAUNCH
    return hgl_check_launch("relpos_mfma");
  }
  const dim3 grid((unsigned)((S + 255) / 256), (unsigned)(B * heads), 2);
  const size_t lds = (size_t)(2 * size - 1) * (hd + 4) * sizeof(float);
  if (hd == 80) hipLaunchKernelGGL(relpos_direct_kernel<80>, grid, dim3(256), lds, st, qkv, ldq, heads, S, size, Rh, Rw, rel_h, rel_w);
  else hipLaunchKernelGGL(relpos_direct_kernel<64>, grid, dim3(256), lds, st, qkv, ldq, heads, S, size, Rh, Rw, rel_h, rel_w);
  return hgl_check_launch("relpos_direct");
}
// the same tables from q given as split planes (f16x3 mode, size 14 or 64); ldq in halfs
int hgl_launch_relpos_split(const void* q_hi, const void* q_lo, int ldq, int B, int heads, int S, int size, int hd,
                            const float* Rh, const float* Rw, float* rel_h, float* rel_w, hipStream_t st) {
  HGL_REQUIRE((hd == 80 || hd == 64) && (size == 14 || size == 64) && q_hi && q_lo && (ldq & 7) == 0,
              "relpos_split: unsupported shape (hd %d, size %d)", hd, size);
  const int NT = size == 14 ? 1 : 4;
  const int qpw = NT == 4 ? 512 : 128;       // queries per workgroup (relpos_mfma_kernel: QB)
  const dim3 gridm((unsigned)((S + qpw - 1) / qpw), (unsigned)(B * heads), 1);
  const size_t ldsm = (size_t)2 * NT * 32 * (hd + 8) * sizeof(_Float16) + (size_t)4 * 32 * (NT * 32 + 1) * sizeof(float);
#define HGL_RPS_LAUNCH(HD_, NT_)                                                                                         \
  do {                                                                                                                 \
    HGL_RESERVE_LDS((relpos_mfma_kernel<HD_, NT_, true>), ldsm, "relpos_split");                                        \
    hipLaunchKernelGGL((relpos_mfma_kernel<HD_, NT_, true>), gridm, dim3(256), ldsm, st, (const float*)nullptr, ldq, heads, S, size, Rh, \
                       Rw, rel_h, rel_w, (const _Float16*)q_hi, (const _Float16*)q_lo);                                \
  } while (0)
  if (hd == 80 && NT == 1) HGL_RPS_LAUNCH(80, 1);
  else if (hd == 80) HGL_RPS_LAUNCH(80, 4);
  else if (NT == 1) HGL_RPS_LAUNCH(64, 1);
  else HGL_RPS_LAUNCH(64, 4);
#undef HGL_RPS_LAUNCH
  return hgl_check_launch("relpos_split");
}
int hgl_launch_win_maps(int g, int ws, int nw, int nb, int* pad_of, int* tok_of, int* pad_list, int* pad_count, hipStream_t st) {
  if (hipMemsetAsync(pad_count, 0, sizeof(int), st) != hipSuccess) {
    hgl_set_error("win_maps: memset failed");
    return HGL_ELAUNCH;
  }
  hipLaunchKernelGGL(win_maps_kernel, dim3(grid1((long long)nb * nw * nw * ws * ws)), dim3(256), 0, st, g, ws, nw, nb, pad_of,
                     tok_of, pad_list, pad_count);
  return hgl_check_launch("win_maps");
}
int hgl_launch_fill_rows(float* dst, int ld, const int* rows, const int* nrows, int max_rows, const float* v, int N,
                         hipStream_t st) {
  HGL_REQUIRE((N & 3) == 0 && max_rows > 0, "fill_rows: bad arguments");
  hipLaunchKernelGGL(fill_rows_kernel, dim3(4, (unsigned)max_rows), dim3(256), 0, st, dst, ld, rows, nrows, v, N / 4);
  return hgl_check_launch("fill_rows");
}
int hgl_launch_fill_rows_split(void* hi, void* lo, int ld, const int* rows, const int* nrows, int max_rows, const float* v,
                               int N, hipStream_t st) {
  HGL_REQUIRE((N & 3) == 0 && (ld & 3) == 0 && max_rows > 0 && hi && lo, "fill_rows_split: bad arguments");
  hipLaunchKernelGGL(fill_rows_split_kernel, dim3(4, (unsigned)max_rows), dim3(256), 0, st, (_Float16*)hi, (_Float16*)lo, ld, rows,
                     nrows, v, N / 4);
  return hgl_check_launch("fill_rows_split");
}
int hgl_launch_im2col3x3(const float* in, int g, int C, float* cols, hipStream_t st) {
  const long long total = (long long)g * g * C * 9;
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid1(total)), dim3(256), 0, st, in, g, C, cols, total);
  return hgl_check_launch("im2col3x3");
}
int hgl_launch_add_rows_bcast(const float* a, long long a_bstride, const float* pe, long long rows_elems,
                              int B, float* out, hipStream_t st) {
  const long long total4 = (long long)B * rows_elems / 4;
  hipLaunchKernelGGL(add_rows_bcast_kernel, dim3(grid1(total4)), dim3(256), 0, st, a, a_bstride / 4, pe, rows_elems / 4, out, total4);
  return hgl_check_launch("add_rows_bcast");
}
int hgl_launch_pe(const float* coords01, const float* G, int n, int F, int mode, const float* pos_embed,
                  const float* not_a_point, float* out, hipStream_t st) {
  hipLaunchKernelGGL(pe_kernel, dim3(grid1((long long)n * F)), dim3(256), 0, st, coords01, G, n, F, mode, pos_embed, not_a_point, out);
  return hgl_check_launch("pe");
}
int hgl_launch_pe_labeled(const float* coords01, const int32_t* labels, const float* G, int n, int F, const float* not_a_point,
                          const float* const* point_embed, float* out, hipStream_t st) {
  PeLabelTab tab;
  tab.e[0] = not_a_point;
  for (int i = 0; i < 4; ++i) tab.e[i + 1] = point_embed[i];
  hipLaunchKernelGGL(pe_labeled_kernel, dim3(grid1((long long)n * F)), dim3(256), 0, st, coords01, labels, G, n, F, tab, out);
  return hgl_check_launch("pe_labeled");
}
int hgl_launch_build_tokens(const float* iou_tok, const float* mask_tok, const float* sparse, int P, int C, int T,
                            float* tokens, hipStream_t st) {
  hipLaunchKernelGGL(build_tokens_kernel, dim3(grid1((long long)P * T * C)), dim3(256), 0, st, iou_tok, mask_tok, sparse, P, C, T, tokens);
  return hgl_check_launch("build_tokens");
}
int hgl_launch_mask_downscaling(const float* in, int P, int g, const float* c1w, const float* c1b, const float* n1w, const float* n1b,
                                const float* c2w, const float* c2b, const float* n2w, const float* n2b, const float* c3w,
                                const float* c3b, float* out, hipStream_t st) {
  hipLaunchKernelGGL(mask_downscaling_kernel, dim3((unsigned)((g * g + 3) / 4), (unsigned)P), dim3(256), 0, st, in, g, c1w, c1b, n1w,
                     n1b, c2w, c2b, n2w, n2b, c3w, c3b, out);
  return hgl_check_launch("mask_downscaling");
}
int hgl_launch_ln_gelu64(float* x, const float* w, const float* b, long long rows, float eps, void* hi, void* lo,
                         hipStream_t st) {
  hipLaunchKernelGGL(ln_gelu64_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, w, b, rows, eps,
                     (_Float16*)hi, (_Float16*)lo);
  return hgl_check_launch("ln_gelu64");
}
int hgl_launch_ln256_pe_split(float* x, const float* w, const float* b, const float* pe, int pe_rows, long long rows,
                              float eps, int write_f32, void* kh, void* kl, void* ph, void* pl, hipStream_t st) {
  hipLaunchKernelGGL(ln256_pe_split_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, w, b, pe, pe_rows, rows,
                     eps, write_f32, (_Float16*)kh, (_Float16*)kl, (_Float16*)ph, (_Float16*)pl);
  return hgl_check_launch("ln256_pe_split");
}
int hgl_launch_hyper_logits(const float* u2, const float* hyper, int P, int g, int row0, float* low_res, hipStream_t st) {
  HGL_REQUIRE((4 * g) % 32 == 0, "hyper_logits: 4*grid must be a multiple of 32 (grid %d)", g);
  hipLaunchKernelGGL(hyper_logits_kernel, dim3(4 * g, P), dim3(256), 0, st, u2, hyper, g, row0, low_res);
  return hgl_check_launch("hyper_logits");
}

extern "C" {

size_t hgl_resize_pil_bilinear_workspace_bytes(int H, int out_w, int C) { return hgl_align_up((size_t)H * out_w * C, 256); }

int hgl_resize_pil_bilinear(const uint8_t* img, int H, int W, int C, int out_h, int out_w, const int32_t* kx,
                            const int32_t* bx, int ksize_x, const int32_t* ky, const int32_t* by, int ksize_y,
                            uint8_t* out, void* workspace, size_t workspace_bytes, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(img && out && kx && bx && ky && by && H > 0 && W > 0 && C > 0 && out_h > 0 && out_w > 0 && ksize_x > 0 && ksize_y > 0,
              "resize_pil_bilinear: bad arguments");
  if (!workspace || workspace_bytes < hgl_resize_pil_bilinear_workspace_bytes(H, out_w, C)) {
    hgl_set_error("resize_pil_bilinear: workspace too small");
    return HGL_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  uint8_t* tmp = (uint8_t*)workspace;   // [H, out_w, C] after the horizontal pass
  // horizontal: outer = rows, axis = x (stride C), inner = C
  {
    const long long total = (long long)H * out_w * C;
    hipLaunchKernelGGL(pil_resample_kernel, dim3(grid1(total)), dim3(256), 0, st, img, tmp, (const int*)kx, (const int*)bx,
                       ksize_x, out_w, (long long)H, (long long)C, (long long)W * C, (long long)C, (long long)out_w * C,
                       (long long)C);
  }
  // vertical: outer = 1, axis = y (stride out_w*C), inner = out_w*C
  {
    const long long total = (long long)out_h * out_w * C;
    hipLaunchKernelGGL(pil_resample_kernel, dim3(grid1(total)), dim3(256), 0, st, tmp, out, (const int*)ky, (const int*)by,
                       ksize_y, out_h, 1ll, (long long)out_w * C, 0ll, (long long)out_w * C, 0ll, (long long)out_w * C);
  }
  return hgl_check_launch("resize_pil_bilinear");
}

int hgl_u8_to_chw_lut(const uint8_t* img, int H, int W, int C, const float* lut, float* out, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(img && lut && out && H > 0 && W > 0 && C > 0 && C <= 4, "u8_to_chw_lut: bad arguments");
  const long long HW = (long long)H * W;
  hipLaunchKernelGGL(u8_to_chw_lut_kernel, dim3(grid1((HW + 3) / 4), C), dim3(256), 0, (hipStream_t)stream, img, lut, out, HW, C);
  return hgl_check_launch("u8_to_chw_lut");
}

size_t hgl_sam_postprocess_workspace_bytes(int K) { return hgl_align_up((size_t)K * 6 * sizeof(unsigned), 256); }

int hgl_sam_postprocess(const float* low_res, const float* iou_pred, int K, int hl, int wl, int img_size,
                        int in_h, int in_w, int H, int W, float mask_threshold, float stability_offset,
                        float pred_iou_thresh, float stability_thresh, uint8_t* masks, int32_t* boxes_xyxy,
                        float* stability, uint8_t* keep, float* full_logits, void* workspace,
                        size_t workspace_bytes, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(low_res && masks && boxes_xyxy && stability && keep, "sam_postprocess: null argument");
  HGL_REQUIRE(K > 0 && K <= 65535 && hl > 0 && wl > 0 && img_size > 0 && in_h > 0 && in_w > 0 && H > 0 && W > 0 &&
              in_h <= img_size && in_w <= img_size, "sam_postprocess: bad shape");
  if (!workspace || workspace_bytes < hgl_sam_postprocess_workspace_bytes(K)) {
    hgl_set_error("sam_postprocess: workspace too small");
    return HGL_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  unsigned* counters = (unsigned*)workspace;
  hipLaunchKernelGGL(init_counters_kernel, dim3((K + 255) / 256), dim3(256), 0, st, counters, K);
  PostArgs a;
  a.low = low_res; a.iou = iou_pred; a.iou_thresh = pred_iou_thresh;
  a.K = K; a.hl = hl; a.wl = wl; a.S = img_size; a.hi = in_h; a.wi = in_w; a.H = H; a.W = W;
  a.thr = mask_threshold; a.off = stability_offset;
  a.masks = masks; a.counters = counters; a.full_logits = full_logits;
  const char* sep_env = hgl_env_str("HGL_SAM_POST_SEP");     // "0": the per-pixel kernel (A/B in tests: bit-identical outputs)
  const bool sep_on = !(sep_env && sep_env[0] == '0');
  const dim3 grid((W + PTW - 1) / PTW, (H + PTH - 1) / PTH, K);
  if (sep_on && postprocess_sep_fits(W, in_w, wl, img_size) && postprocess_sep_fits(H, in_h, hl, img_size))
    hipLaunchKernelGGL(sam_postprocess_sep_kernel, grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(sam_postprocess_kernel, grid, dim3(256), 0, st, a);
  hipLaunchKernelGGL(sam_finalize_kernel, dim3((K + 255) / 256), dim3(256), 0, st, counters, iou_pred, K,
                     pred_iou_thresh, stability_thresh, stability, (int*)boxes_xyxy, keep);
  return hgl_check_launch("sam_postprocess");
}

int hgl_nms(const int32_t* boxes_xyxy, const float* scores, const uint8_t* keep, int K, float iou_threshold,
            int32_t* out_idx, int32_t* out_n, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(boxes_xyxy && scores && keep && out_idx && out_n, "nms: null argument");
  HGL_REQUIRE(K > 0 && K <= 1024, "nms: K must be in [1,1024] (got %d)", K);
  if (K <= 512 && ((uintptr_t)boxes_xyxy & 15) == 0)
    hipLaunchKernelGGL(nms_bits_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, (const int*)boxes_xyxy, scores, keep, K, iou_threshold, (int*)out_idx, (int*)out_n);
  else
    hipLaunchKernelGGL(nms_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const int*)boxes_xyxy, scores, keep, K, iou_threshold, (int*)out_idx, (int*)out_n);
  return hgl_check_launch("nms");
}

size_t hgl_nms_large_workspace_bytes(int K) {
  const size_t W = ((size_t)K + 63) / 64;
  return hgl_align_up((size_t)K * sizeof(int), 256) + hgl_align_up(sizeof(int), 256) + hgl_align_up((size_t)K * W * 8, 256);
}

int hgl_nms_large(const int32_t* boxes_xyxy, const float* scores, const uint8_t* keep, int K, float iou_threshold,
                  int32_t* out_idx, int32_t* out_n, void* workspace, size_t workspace_bytes, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(boxes_xyxy && scores && keep && out_idx && out_n, "nms_large: null argument");
  HGL_REQUIRE(K > 0 && K <= 32768, "nms_large: K must be in [1,32768] (got %d)", K);
  HGL_REQUIRE(((uintptr_t)boxes_xyxy & 15) == 0, "nms_large: boxes must be 16-byte aligned");
  if (!workspace || workspace_bytes < hgl_nms_large_workspace_bytes(K)) {
    hgl_set_error("nms_large: workspace too small");
    return HGL_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int W = (K + 63) / 64;
  HglArena ar(workspace, workspace_bytes);
  int* order = ar.take<int>((size_t)K);
  int* nvalid = ar.take<int>(1);
  unsigned long long* mask = ar.take<unsigned long long>((size_t)K * W);
  if (hipMemsetAsync(nvalid, 0, sizeof(int), st) != hipSuccess) {
    hgl_set_error("nms_large: memset failed");
    return HGL_ELAUNCH;
  }
  hipLaunchKernelGGL(nms_rank_kernel, dim3((K + 255) / 256), dim3(256), 0, st, scores, keep, K, order, nvalid);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(W, W), dim3(64), 0, st, (const int*)boxes_xyxy, order, nvalid, W, iou_threshold, mask);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(256), 0, st, mask, order, nvalid, W, (int*)out_idx, (int*)out_n);
  return hgl_check_launch("nms_large");
}

int hgl_box_near_crop_edge(const int32_t* boxes_xyxy, int K, const int32_t* crop_box_xyxy, const int32_t* orig_box_xyxy,
                           float atol, uint8_t* keep, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(boxes_xyxy && crop_box_xyxy && orig_box_xyxy && keep && K > 0, "box_near_crop_edge: bad arguments");
  hipLaunchKernelGGL(crop_edge_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const int*)boxes_xyxy, K,
                     crop_box_xyxy[0], crop_box_xyxy[1], crop_box_xyxy[2], crop_box_xyxy[3], orig_box_xyxy[0],
                     orig_box_xyxy[1], orig_box_xyxy[2], orig_box_xyxy[3], atol, keep);
  return hgl_check_launch("box_near_crop_edge");
}

int hgl_gather_masks(const uint8_t* masks, const int32_t* idx, const int32_t* n, int max_n, long long HW,
                     uint8_t* out, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(masks && idx && n && out && max_n > 0 && HW > 0 && (HW & 15) == 0, "gather_masks: bad arguments (HW must be a multiple of 16)");
  hipLaunchKernelGGL(gather_masks_kernel, dim3(64, max_n), dim3(256), 0, (hipStream_t)stream, masks, (const int*)idx, (const int*)n, HW / 16, out);
  return hgl_check_launch("gather_masks");
}

}  // extern "C"
