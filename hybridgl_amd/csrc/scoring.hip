// Scoring tail + image synthesis kernels (all HBM / latency bound, no matrix work).
//   calculate_score      model/backbone.py:74-87
//   coherence scores     Hybridgl_main.py:201-223 (+ gen_dir_mask utils.py:135-161)
//   IoU                  utils.py:365-384
//   score_sentence       Hybridgl_main.py:153-196,225-228 (+ relation_boxes utils.py:240-268)
//   synthesize_views     Hybridgl_main.py:93-125
#include "hgl_common.h"
#include <mutex>
#include <condition_variable>
#include <math.h>
#include <string.h>

namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- calculate_score: one wave per (n,t) ----
__global__ __launch_bounds__(256) void calc_score_kernel(const float* __restrict__ img,
                                                         const float* __restrict__ txt, int N, int T,
                                                         int E, float logit_scale,
                                                         float* __restrict__ logits) {
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= N * T) return;
  const int n = pair / T, tt = pair - n * T;
  const float* ir = img + (long long)n * E;
  const float* tr = txt + (long long)tt * E;
  float si = 0.f, stx = 0.f;
  for (int i = lane; i < E; i += 64) {
    si += ir[i] * ir[i];
    stx += tr[i] * tr[i];
  }
  const float ni = sqrtf(wave_sum_f(si)), nt = sqrtf(wave_sum_f(stx));
  float d = 0.f;
  for (int i = lane; i < E; i += 64) d += (logit_scale * (ir[i] / ni)) * (tr[i] / nt);
  d = wave_sum_f(d);
  if (lane == 0) logits[pair] = d;
}

// ---- heat-map statistics ----
// stats[0]=min, stats[1]=max (as ordered-int encodings), reduced with atomics on monotone keys
__device__ __forceinline__ int f2ord(float f) {
  int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ a, long long n,
                                                     int* __restrict__ stats) {
  float mn = INFINITY, mx = -INFINITY;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float v = a[i];
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o));
    mx = fmaxf(mx, __shfl_xor(mx, o));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&stats[0], f2ord(mn));
    atomicMax(&stats[1], f2ord(mx));
  }
}

__global__ void init_stats_kernel(int* stats) {
  stats[0] = 0x7fffffff;          // +max ordered key
  stats[1] = (int)0x80000000;     // -max ordered key
}

// torch.linspace(start,end,steps)[i] in fp32 (ATen RangeFactories: symmetric evaluation)
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
  if (steps == 1) return start;
  const float step = (end - start) / (float)(steps - 1);
  const int half = steps / 2;
  return i < half ? fmaf(step, (float)i, start) : fmaf(-step, (float)(steps - i - 1), end);  // fmas, as ATen
}

// gen_dir_mask column weight (utils.py:135-161): 0 none, 1 left, 2 right, 3 middle
__device__ __forceinline__ float dir_weight(int dirflag, int x, int W) {
  switch (dirflag) {
    case 1: return linspace_at(1.f, 0.f, W, x);
    case 2: return linspace_at(0.f, 1.f, W, x);
    case 3: {
      const int w1 = W / 2;
      return x < w1 ? linspace_at(0.f, 1.f, w1, x) : linspace_at(1.f, 0.f, W - w1, x - w1);
    }
    default: return 1.f;
  }
}

// Masked pooling.  One block = PIX_PER_BLOCK consecutive pixels x all masks.
// The normalised heat-map tile lives in registers (16 px per lane); each mask's bytes
// are read exactly once with 16-byte loads; partial sums (double) go to part[blk][n].
constexpr int PX_LANE = 16;
constexpr int PIX_PER_BLOCK = 256 * PX_LANE;  // 4096

constexpr int MASK_GROUP = 8;  // masks per blockIdx.y
// the per-group pooling (three sentences of up to sixteen refs per launch: enough workgroups without splitting the masks
// eight ways; a single ref's launch keeps eight: 139 against 130 us with 32): the normalised heat-map values of a block's pixels -- a division and a direction weight per pixel -- are
// computed once per 32 masks instead of once per 8.  The partial sums per (mask, part) do not depend on the grouping.
constexpr int REF_MASK_GROUP = 32;

// ---- the pooling of a block, shared by the three pooling kernels (one heat-map / the maps of a ref / of a group of refs) -----
// One workgroup = PIX_PER_BLOCK consecutive pixels x a group of masks x UP TO FOUR heat-maps at once: the normalised values of
// the maps at the lane's 16 pixels live in registers (16 VGPRs per map), and a mask's 16 bytes are loaded ONCE, turned into
// sixteen 0.0 / 1.0 floats once, and multiplied into all the maps: cvt + one fma per (pixel, map) instead of a 64-bit select
// and a double-precision add per (pixel, map) and a re-read of the bytes per map.  (Until round 5 every (block, group, map) was
// its own workgroup with a double-precision sum per pixel: 130 vector instructions per KiB of mask bytes and map, the bytes
// fetched S times -- 1.37 TB/s on the algorithmic bytes, instruction-bound.)
// Arithmetic of one (map, mask, wave) partial -- THE definition all three kernels share, so their results are the same bits:
//   lane: a = 0; for e = 0..15 in order: a = fma(m_e, v_e, a) in fp32 (m_e is 0 or 1: the products are exact, the 16-term sum
//   carries at most a few 1e-8 of relative error); the 64 lane sums folded in fp32 by pool_sum_f's fixed DPP tree; the wave's
//   sum widened to double; partials over a mask's waves are added in double by the scoring kernels as before.
struct PoolMap {
  const float* attn;   // the heat-map
  int dirflag, sidx;   // gen_dir_mask weight; index of the map in the partial arrays
  float mn, range;     // min and (max - min) of the map
};

// dir_weight(dirflag, x, W) with everything that does not depend on x computed once per map: the weight of a column is
// linspace_at on one of at most two segments (dirflag 3: the two ramps of `middle`; 0: the constant 1 as a ramp of step 0).
// Same operations on the same values as dir_weight / linspace_at: same bits.
struct DirRamp {
  int xsplit;                       // columns < xsplit lie on segment A, the others on segment B (index x - xsplit)
  float stepA, startA, endA, stepB, startB, endB;
  int stepsA, halfA, stepsB, halfB;
  __device__ __forceinline__ void seg(float start, float end, int steps, float& st, float& s0, float& e0, int& n, int& h) {
    if (steps <= 1) { st = 0.f; s0 = start; e0 = start; n = 1; h = 1; return; }     // linspace_at: steps == 1 -> start
    st = (end - start) / (float)(steps - 1);
    s0 = start; e0 = end; n = steps; h = steps / 2;
  }
  __device__ __forceinline__ void init(int dirflag, int W) {
    xsplit = W;
    seg(1.f, 1.f, W, stepB, startB, endB, stepsB, halfB);
    if (dirflag == 1) seg(1.f, 0.f, W, stepA, startA, endA, stepsA, halfA);
    else if (dirflag == 2) seg(0.f, 1.f, W, stepA, startA, endA, stepsA, halfA);
    else if (dirflag == 3) {
      const int w1 = W / 2;
      xsplit = w1;
      seg(0.f, 1.f, w1, stepA, startA, endA, stepsA, halfA);
      seg(1.f, 0.f, W - w1, stepB, startB, endB, stepsB, halfB);
    } else {
      stepA = 0.f; startA = 1.f; endA = 1.f; stepsA = W; halfA = W;      // fma(0, i, 1) == 1
    }
  }
  __device__ __forceinline__ float at(int x) const {
    const bool a = x < xsplit;
    const int i = a ? x : x - xsplit;
    const float st = a ? stepA : stepB, s0 = a ? startA : startB, e0 = a ? endA : endB;
    const int n = a ? stepsA : stepsB, h = a ? halfA : halfB;
    return i < h ? fmaf(st, (float)i, s0) : fmaf(-st, (float)(n - i - 1), e0);
  }
};

// Sum over the 64 lanes for the pooling loop (EXEC all ones there), on the VALU alone, in fp32: six adds whose second operand
// comes through DPP (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: every lane then holds its row's sum; row_bcast:15 into
// rows 1 and 3, row_bcast:31 into rows 2 and 3) -- the total, ((r2 + r3) + (r0 + r1)), ends in lanes 48..63.  One instruction
// per step (v_add_f32 with a DPP operand).  (__shfl_xor compiles to ds_bpermute_b32: a double-precision butterfly was twelve
// LDS-crossbar operations and six adds per (mask, map), as much of the loop as its arithmetic; the DPP form in double costs three
// instructions per step.)  The wave's 1024 products are thus summed in fp32 -- as the reference sums ALL its products
// (Hybridgl_main.py:221, torch.sum of an fp32 tensor) -- in a fixed tree: deterministic, and the same function in all three
// pooling kernels; partials of different waves are added in double by the scoring kernels.
template <int CTRL, int ROWS>
__device__ __forceinline__ float pool_dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xf, false));
}
__device__ __forceinline__ float pool_sum_f(float v) {      // valid in lanes 48..63 (read it in lane 63)
  v += pool_dpp_f<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
  v += pool_dpp_f<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
  v += pool_dpp_f<0x141, 0xf>(v);     // row_half_mirror
  v += pool_dpp_f<0x140, 0xf>(v);     // row_mirror
  v += pool_dpp_f<0x142, 0xa>(v);     // row_bcast:15 -> rows 1, 3 (the other rows add 0)
  v += pool_dpp_f<0x143, 0xc>(v);     // row_bcast:31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ unsigned pool_sum_u(unsigned v) {   // valid in lanes 48..63
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
  return v;
}

typedef unsigned pool_u32x4 __attribute__((ext_vector_type(4)));
typedef pool_u32x4 pool_u32x4_u __attribute__((aligned(1)));     // 16 bytes at ANY address: one global_load_dwordx4 (the hardware takes
typedef f32x4 pool_f32x4_u __attribute__((aligned(4)));          // unaligned vector loads; a plane of an odd-sized image starts anywhere)

// FULL: the whole block of PIX_PER_BLOCK pixels lies inside the plane (wave-uniform; every block but the last one of a plane):
// no per-pixel predicate anywhere.  !FULL: the last block, pixel by pixel with predicates -- instantiated for one map only.
template <int NS, bool FULL>
__device__ __forceinline__ void pool_block(int blk, const PoolMap (&pm)[NS], int n0, int n1, const uint8_t* __restrict__ masks, int N, int H,
                                           int W, bool write_tot, double* __restrict__ part_sum, unsigned* __restrict__ part_cnt,
                                           double* __restrict__ part_tot, int nparts) {
  const long long HW = (long long)H * W;
  const long long p0 = (long long)blk * PIX_PER_BLOCK + threadIdx.x * PX_LANE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long part = (long long)blk * 4 + wave;      // one partial per wave: no block sync
  float v[NS][PX_LANE];
  const int xcol0 = (int)(p0 % W);      // ONE 64-bit division per lane; the pixels' columns follow by increment and wrap
  int cnt_map = -1;                     // the map (if any of these) that files the pixel counts: sidx == 0
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const float* attn = pm[j].attn;
    float raw[PX_LANE];      // four 16-byte loads issued together (sixteen conditional scalar loads are sixteen round trips)
    if (FULL) {
      const f32x4 r0 = *(const pool_f32x4_u*)(attn + p0), r1 = *(const pool_f32x4_u*)(attn + p0 + 4);
      const f32x4 r2 = *(const pool_f32x4_u*)(attn + p0 + 8), r3 = *(const pool_f32x4_u*)(attn + p0 + 12);
#pragma unroll
      for (int e = 0; e < 4; ++e) { raw[e] = r0[e]; raw[4 + e] = r1[e]; raw[8 + e] = r2[e]; raw[12 + e] = r3[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < PX_LANE; ++e) raw[e] = attn[min(p0 + e, HW - 1)];
    }
    int xcol = xcol0;
    DirRamp ramp;
    ramp.init(pm[j].dirflag, W);
    // the map's total over the wave's pixels in EXACTLY the arithmetic of a mask that covers them all (fma by 1.0 in order, the
    // same fold over the lanes): for a full mask  total - sum  is then 0 and the "outside" mean 0 / 0 = NaN, as in the reference
    // (Hybridgl_main.py:221: every term of its sum is x * 0 / 0) -- not +-inf from two roundings of the same sum
    float ta = 0.f;
#pragma unroll
    for (int e = 0; e < PX_LANE; ++e) {
      float val = ((raw[e] - pm[j].mn) / pm[j].range) * ramp.at(xcol);
      if (!FULL && p0 + e >= HW) val = 0.f;
      v[j][e] = val;
      ta = __builtin_fmaf(1.0f, val, ta);
      if (++xcol >= W) xcol = 0;
    }
    if (write_tot) {
      const float tot = pool_sum_f(ta);
      if (lane == 63) part_tot[(long long)pm[j].sidx * nparts + part] = (double)tot;
    }
    if (pm[j].sidx == 0) cnt_map = j;
  }
  for (int n = n0; n < n1; ++n) {
    const uint8_t* m = masks + (long long)n * HW + p0;
    unsigned w4[4] = {0u, 0u, 0u, 0u};
    if (FULL) {
      const pool_u32x4 mv = *(const pool_u32x4_u*)m;
      w4[0] = mv.x; w4[1] = mv.y; w4[2] = mv.z; w4[3] = mv.w;
    } else {      // byte loads at clamped addresses, pixels beyond the plane count as 0
#pragma unroll
      for (int e = 0; e < PX_LANE; ++e) {
        const long long pe = min(p0 + e, HW - 1) - p0;
        const unsigned byte = p0 + e < HW ? (unsigned)m[pe] : 0u;
        w4[e >> 2] |= byte << (8 * (e & 3));
      }
    }
    unsigned c = 0;
    float mf[PX_LANE];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // every byte -> (byte != 0): bit 7 of ((b & 0x7f) + 0x7f) | b is set exactly for b != 0 (no carry crosses a byte)
      const unsigned w = w4[q];
      const unsigned nz = ((((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) >> 7) & 0x01010101u;
      c += (unsigned)__popc(nz);
      mf[4 * q + 0] = (float)(nz & 0xffu);
      mf[4 * q + 1] = (float)((nz >> 8) & 0xffu);
      mf[4 * q + 2] = (float)((nz >> 16) & 0xffu);
      mf[4 * q + 3] = (float)(nz >> 24);
    }
    float sum[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < PX_LANE; ++e) a = __builtin_fmaf(mf[e], v[j][e], a);
      sum[j] = a;
    }
    // the folds of the maps and of the count in ONE basic block (their DPP steps interleave and fill each other's wait states),
    // then all the stores of lane 63 behind one branch
#pragma unroll
    for (int j = 0; j < NS; ++j) sum[j] = pool_sum_f(sum[j]);
    c = pool_sum_u(c);
    if (lane == 63) {
#pragma unroll
      for (int j = 0; j < NS; ++j) part_sum[((long long)pm[j].sidx * nparts + part) * N + n] = (double)sum[j];
      if (cnt_map >= 0) part_cnt[part * N + n] = c;
    }
  }
}

// The kernels below come in pairs: the main launch covers the blocks that lie wholly inside the plane (FULL: HW / PIX_PER_BLOCK
// of them -- all 100 of a 640 x 640 plane), a second, small launch the partial last block when there is one (!FULL, one map at a
// time; its 16 byte loads per lane and predicates would otherwise set the register count of the main kernel).
template <bool FULL>
__global__ __launch_bounds__(256) void masked_pool_kernel(const float* __restrict__ attn,
                                                          const uint8_t* __restrict__ masks, int N,
                                                          int H, int W, int dirflag,
                                                          const int* __restrict__ stats,
                                                          double* __restrict__ part_sum,
                                                          unsigned* __restrict__ part_cnt,
                                                          double* __restrict__ part_tot, int blk0) {
  const float mn = ord2f(stats[0]), mx = ord2f(stats[1]);
  const PoolMap pm[1] = {{attn, dirflag, 0, mn, mx - mn}};
  const int n0 = blockIdx.y * MASK_GROUP;
  pool_block<1, FULL>(blk0 + blockIdx.x, pm, n0, min(N, n0 + MASK_GROUP), masks, N, H, W, blockIdx.y == 0, part_sum, part_cnt, part_tot, 0);
}

// one wave per mask: lanes stride over the per-wave partials in a fixed pattern, then a fixed-order
// butterfly -- deterministic run to run
__global__ __launch_bounds__(256) void coherence_final_kernel(const double* __restrict__ part_sum,
                                                              const unsigned* __restrict__ part_cnt,
                                                              const double* __restrict__ part_tot,
                                                              int nblk, int N, long long HW,
                                                              float black, float* __restrict__ score) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  double tot = 0.0, s = 0.0;
  unsigned long long c = 0;
  for (int b = lane; b < nblk; b += 64) {
    tot += part_tot[b];
    s += part_sum[(long long)b * N + n];
    c += part_cnt[(long long)b * N + n];
  }
  tot = wave_sum_d(tot);
  s = wave_sum_d(s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if (lane != 0) return;
  const double mean = tot / (double)HW;
  const double in_sum = s / mean, out_sum = (tot - s) / mean;
  // (imgattn*(2-black)*m/m.sum()).sum() - (imgattn*black*(1-m)/(1-m).sum()).sum()
  const double a = (double)(2.f - black) * in_sum / (double)c;
  const double bterm = (double)black * out_sum / (double)(HW - (long long)c);
  score[n] = (float)(a - bterm);
}

// ---- IoU ----
__global__ __launch_bounds__(256) void iou_kernel(const uint8_t* __restrict__ p,
                                                  const uint8_t* __restrict__ g, long long n,
                                                  unsigned long long* __restrict__ out) {
  unsigned I = 0, U = 0;
  const long long n16 = n / 16;
  const bool al = (((uintptr_t)p | (uintptr_t)g) & 15) == 0;
  if (al) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n16;
         i += (long long)gridDim.x * blockDim.x) {
      const uint4 a = ((const uint4*)p)[i], b = ((const uint4*)g)[i];
      const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        // nonzero byte -> 0x01 per byte
        unsigned x = aw[w], y = bw[w];
        x |= x >> 4; x |= x >> 2; x |= x >> 1; x &= 0x01010101u;
        y |= y >> 4; y |= y >> 2; y |= y >> 1; y &= 0x01010101u;
        I += __popc(x & y);
        U += __popc(x | y);
      }
    }
  }
  const long long tail0 = al ? n16 * 16 : 0;
  for (long long i = tail0 + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const bool a = p[i] != 0, b = g[i] != 0;
    I += (a && b);
    U += (a || b);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    I += __shfl_xor(I, o);
    U += __shfl_xor(U, o);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&out[0], (unsigned long long)I);
    atomicAdd(&out[1], (unsigned long long)U);
  }
}

// same with pred = masks[idx[which]] resolved on the device
__global__ __launch_bounds__(256) void iou_select_kernel(const uint8_t* __restrict__ masks,
                                                         const int* __restrict__ idx, int which,
                                                         const uint8_t* __restrict__ g, long long n,
                                                         unsigned long long* __restrict__ out) {
  const uint8_t* p = masks + (long long)idx[which] * n;
  unsigned I = 0, U = 0;
  const bool al = ((((uintptr_t)p) | ((uintptr_t)g)) & 15) == 0;
  const long long n16 = al ? n / 16 : 0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n16;
       i += (long long)gridDim.x * blockDim.x) {
    const uint4 a = ((const uint4*)p)[i], b = ((const uint4*)g)[i];
    const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      unsigned x = aw[w], y = bw[w];
      x |= x >> 4; x |= x >> 2; x |= x >> 1; x &= 0x01010101u;
      y |= y >> 4; y |= y >> 2; y |= y >> 1; y &= 0x01010101u;
      I += __popc(x & y);
      U += __popc(x | y);
    }
  }
  for (long long i = n16 * 16 + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const bool a = p[i] != 0, b = g[i] != 0;
    I += (a && b);
    U += (a || b);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    I += __shfl_xor(I, o);
    U += __shfl_xor(U, o);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&out[0], (unsigned long long)I);
    atomicAdd(&out[1], (unsigned long long)U);
  }
}

// ---- per-sentence tail: single workgroup ----
constexpr int MAXK = 16;

__device__ float relation_boxes_dev(const long long* bi, const long long* bj, float si, float sj,
                                    int rela) {
  // utils.py:240-268 ; integer tensors /2 -> float32 true division
  switch (rela) {
    case 1: return si * sj * (float)(((float)bi[0] + (float)bi[2] / 2.f) < ((float)bj[0] + (float)bj[2] / 2.f));
    case 2: return si * sj * (float)(((float)bi[0] + (float)bi[2] / 2.f) > ((float)bj[0] + (float)bj[2] / 2.f));
    case 3: return si * sj * (float)(((float)bi[1] + (float)bi[3] / 2.f) < ((float)bj[1] + (float)bj[3] / 2.f));
    case 4: return si * sj * (float)(((float)bi[1] + (float)bi[3] / 2.f) > ((float)bj[1] + (float)bj[3] / 2.f));
    case 5: return si * sj * (float)((bi[2] * bi[3]) > (bj[2] * bj[3]));
    case 6: return si * sj * (float)((bi[2] * bi[3]) < (bj[2] * bj[3]));
    case 7: {
      const long long x1 = max(bi[0], bj[0]);
      const long long x2 = max(x1, min(bi[0] + bi[2], bj[0] + bj[2]));
      const long long y1 = max(bi[1], bj[1]);
      const long long y2 = max(y1, min(bi[1] + bi[3], bj[1] + bj[3]));
      return si * sj * (float)(x2 - x1) * (float)(y2 - y1) / (float)(bi[2] * bi[3]);
    }
    default: return si;
  }
}

__device__ __forceinline__ void score_sentence_body(
    const float* __restrict__ hybrid, const float* __restrict__ sent, const float* __restrict__ nphr,
    const float* __restrict__ others, int n_other, float r_mix,
    const long long* __restrict__ boxes, const float* __restrict__ gem, int N, int E,
    float logit_scale, int k1, int k2, float alpha, int rela, int has_other, int* __restrict__ idx,
    float* __restrict__ score_clip, float* __restrict__ score_neg, float* __restrict__ soft_scratch) {
  // soft_scratch: [2*N] softmax values + [2*E] text vectors (global scratch so N, E are unbounded)
  float* tpos = soft_scratch + 2 * N;
  float* tneg = tpos + E;
  // text_ensemble = r*sentence + (1-r)*noun_phrase ; neg = mean of "a photo of <other noun>"
  // features accumulated in order (Hybridgl_main.py:153,157-164); zeros when there is none
  for (int i = threadIdx.x; i < E; i += 256) {
    tpos[i] = r_mix * sent[i] + (1.f - r_mix) * nphr[i];
    float a = 0.f;
    for (int k = 0; k < n_other; ++k) a += others[(long long)k * E + i];
    tneg[i] = n_other > 0 ? a / (float)n_other : 0.f;
  }
  __syncthreads();
  __threadfence_block();
  __shared__ float red[8];
  __shared__ int redi[8];
  __shared__ int top1[MAXK], top2[MAXK];
  __shared__ float nrm[2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

  // text norms
  for (int which = wave; which < 2; which += 4) {
    const float* tv = which == 0 ? tpos : tneg;
    float s = 0.f;
    for (int i = lane; i < E; i += 64) s += tv[i] * tv[i];
    s = wave_sum_f(s);
    if (lane == 0) nrm[which] = sqrtf(s);
  }
  __syncthreads();
  // logits: one wave per mask (strided).  A lane's elements of a feature row (and of the two text vectors) are fetched in
  // batches of eight and four masks are in flight per wave: `for (i = lane; i < E; i += 64) s += ir[i] * ir[i]` is one memory
  // round trip per trip, twice per mask, sixteen masks per wave -- most of this kernel's 80-130 us.  Same operations in the
  // same order per lane: identical logits.
  {
    constexpr int UE = 8, MQ = 4;
    for (int nb = wave; nb < N; nb += 4 * MQ) {
      float sq[MQ], d0[MQ], d1[MQ], ni[MQ];
#pragma unroll
      for (int j = 0; j < MQ; ++j) { sq[j] = 0.f; d0[j] = 0.f; d1[j] = 0.f; }
      for (int i0 = lane; i0 < E; i0 += 64 * UE) {
        float x[MQ][UE];
#pragma unroll
        for (int j = 0; j < MQ; ++j) {
          const float* ir = hybrid + (long long)min(nb + 4 * j, N - 1) * E;
#pragma unroll
          for (int u = 0; u < UE; ++u) x[j][u] = ir[min(i0 + 64 * u, E - 1)];
        }
#pragma unroll
        for (int j = 0; j < MQ; ++j)
#pragma unroll
          for (int u = 0; u < UE; ++u)
            if (i0 + 64 * u < E) sq[j] += x[j][u] * x[j][u];
      }
#pragma unroll
      for (int j = 0; j < MQ; ++j) ni[j] = sqrtf(wave_sum_f(sq[j]));
      for (int i0 = lane; i0 < E; i0 += 64 * UE) {
        float x[MQ][UE], tp[UE], tn[UE];
#pragma unroll
        for (int u = 0; u < UE; ++u) {
          tp[u] = tpos[min(i0 + 64 * u, E - 1)];
          tn[u] = tneg[min(i0 + 64 * u, E - 1)];
        }
#pragma unroll
        for (int j = 0; j < MQ; ++j) {
          const float* ir = hybrid + (long long)min(nb + 4 * j, N - 1) * E;
#pragma unroll
          for (int u = 0; u < UE; ++u) x[j][u] = ir[min(i0 + 64 * u, E - 1)];
        }
#pragma unroll
        for (int j = 0; j < MQ; ++j)
#pragma unroll
          for (int u = 0; u < UE; ++u)
            if (i0 + 64 * u < E) {
              const float v = logit_scale * (x[j][u] / ni[j]);
              d0[j] += v * (tp[u] / nrm[0]);
              d1[j] += v * (tn[u] / nrm[1]);
            }
      }
#pragma unroll
      for (int j = 0; j < MQ; ++j) {
        const float r0 = wave_sum_f(d0[j]), r1 = wave_sum_f(d1[j]);
        if (lane == 0 && nb + 4 * j < N) {
          score_clip[nb + 4 * j] = r0;
          score_neg[nb + 4 * j] = r1;
        }
      }
    }
  }
  __syncthreads();
  __threadfence_block();

  // softmax over masks of both logit vectors + argmax of score_clip.  torch.argmax's order: a NaN is larger than every
  // number, the first of equal maxima (or NaNs) wins; a NaN maximum then makes the whole soft-max NaN, as torch's does.
  auto better = [](float v, int n, float mx, int mi) {
    const bool vn = v != v, mn = mx != mx;
    if (vn != mn) return vn;
    if (vn) return n < mi;
    return v > mx || (v == mx && n < mi);
  };
  for (int which = 0; which < 2; ++which) {
    const float* lg = which == 0 ? score_clip : score_neg;
    float mx = -INFINITY;
    int mi = 0x7fffffff;
    for (int n = t; n < N; n += 256) {
      const float v = lg[n];
      if (better(v, n, mx, mi)) { mx = v; mi = n; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(mx, o);
      const int oi = __shfl_xor(mi, o);
      if (better(ov, oi, mx, mi)) { mx = ov; mi = oi; }
    }
    if (lane == 0) { red[wave] = mx; redi[wave] = mi; }
    __syncthreads();
    float gmx = red[0];
    int gmi = redi[0];
    for (int w = 1; w < 4; ++w)
      if (better(red[w], redi[w], gmx, gmi)) { gmx = red[w]; gmi = redi[w]; }
    __syncthreads();
    if (which == 0 && t == 0) idx[0] = gmi == 0x7fffffff ? 0 : gmi;   // all-NaN logits (a NaN text feature): a valid index, NaN scores
    float se = 0.f;
    for (int n = t; n < N; n += 256) se += expf(lg[n] - gmx);
    se = wave_sum_f(se);
    if (lane == 0) red[wave] = se;
    __syncthreads();
    const float denom = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    for (int n = t; n < N; n += 256) soft_scratch[which * N + n] = expf(lg[n] - gmx) / denom;
  }
  __syncthreads();
  __threadfence_block();

  // top-k (descending, lowest index on ties) by repeated argmax -- wave 0 for list 1, wave 1 for list 2
  if (wave < 2) {
    const float* sv = soft_scratch + wave * N;
    int* top = wave == 0 ? top1 : top2;
    const int kk = wave == 0 ? k1 : k2;
    for (int j = 0; j < kk; ++j) {
      float mx = -INFINITY;
      int mi = 0x7fffffff;
      for (int n = lane; n < N; n += 64) {
        bool used = false;
        for (int u = 0; u < j; ++u) used |= top[u] == n;
        const float v = sv[n];
        if (!used && (v > mx || (v == mx && n < mi))) { mx = v; mi = n; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mx, o);
        const int oi = __shfl_xor(mi, o);
        if (ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; }
      }
      if (mi == 0x7fffffff) mi = j < N ? j : 0;  // all-NaN scores (no other nouns): any valid index
      if (lane == 0) top[j] = mi;
      __builtin_amdgcn_wave_barrier();
      __threadfence_block();
    }
  }
  __syncthreads();

  if (t == 0) {
    // Hybridgl_main.py:183-196 relation sums, softmax, blend with the coherence score, argmax
    const float* sc = soft_scratch;
    const float* sn = soft_scratch + N;
    // partial sums are fp32 in the reference (np.float64 + Tensor -> Tensor.__radd__ -> fp32)
    float ts[MAXK];
    for (int i = 0; i < k1; ++i) {
      float acc = 0.f;
      const int ii = top1[i];
      if (!has_other) {
        for (int j = 0; j < k1; ++j) {
          const int jj = top1[j];
          acc += relation_boxes_dev(boxes + 4 * ii, boxes + 4 * jj, sc[ii], sc[jj], rela);
        }
      } else {
        for (int j = 0; j < k2; ++j) {
          const int jj = top2[j];
          acc += relation_boxes_dev(boxes + 4 * ii, boxes + 4 * jj, sc[ii], sn[jj], rela);
        }
      }
      ts[i] = acc;
    }
    float tf[MAXK];
    float mx = -INFINITY;
    for (int i = 0; i < k1; ++i) { tf[i] = ts[i]; mx = fmaxf(mx, tf[i]); }
    float den = 0.f;
    for (int i = 0; i < k1; ++i) { tf[i] = expf(tf[i] - mx); den += tf[i]; }
    // torch.argmax's order: a NaN is larger than every number and the FIRST of equal maxima (or NaNs) wins -- a NaN coherence
    // score (an empty or full proposal mask, a constant heat-map: 0/0 in Hybridgl_main.py:203-223) among the top-k decides
    // the reference's answer (tests/golden/scoring_nan.npz)
    int best = 0;
    float bv = 0.f;
    for (int i = 0; i < k1; ++i) {
      const float v = (tf[i] / den) * (1.f - alpha) + alpha * gem[top1[i]];
      if (i == 0 || v > bv || (v != v && bv == bv)) { bv = v; best = i; }
    }
    idx[1] = top1[best];
  }
}

__global__ __launch_bounds__(256) void score_sentence_kernel(
    const float* __restrict__ hybrid, const float* __restrict__ sent, const float* __restrict__ nphr,
    const float* __restrict__ others, int n_other, float r_mix,
    const long long* __restrict__ boxes, const float* __restrict__ gem, int N, int E,
    float logit_scale, int k1, int k2, float alpha, int rela, int has_other, int* __restrict__ idx,
    float* __restrict__ score_clip, float* __restrict__ score_neg, float* __restrict__ soft_scratch) {
  score_sentence_body(hybrid, sent, nphr, others, n_other, r_mix, boxes, gem, N, E, logit_scale, k1, k2, alpha, rela, has_other,
                      idx, score_clip, score_neg, soft_scratch);
}

// ---- the tail of a whole REF (all its sentences) in four launches: Hybridgl_main.py:153-230 -------------------------------
// The per-sentence launches above re-read the N mask planes for every sentence (83.6 MB per ref at N = 64, 640 x 640, three
// sentences) and cost ~15 launches per sentence with torch glue between them.  Here the pooling of ALL sentences' heat-maps is
// one launch (the mask planes come from HBM once: the other maps' workgroups find them in L2 / the Infinity Cache), the
// scoring of the sentences runs as one workgroup each in one launch, the IoU of
// both winners of every sentence in one launch, and the four accumulators of Hybridgl_main.py:52-55 are updated by the
// last block of that launch.  Arithmetic and reduction orders are those of the per-sentence kernels (bit-identical results).
constexpr int REF_MAXS = 16;       // sentences per launch (the host loops over chunks)
constexpr int REF_MM_BLOCKS = 64;  // min / max partials per heat-map

struct RefSentences {
  const float* attn[REF_MAXS];
  const uint8_t* target[REF_MAXS];
  int sent_row[REF_MAXS], nphr_row[REF_MAXS], other_row0[REF_MAXS], n_other[REF_MAXS];
  int dirflag[REF_MAXS], rela[REF_MAXS], has_other[REF_MAXS];
  float black[REF_MAXS];
};

__device__ __forceinline__ void ref_minmax_body(const float* __restrict__ a, int s, long long n, float* __restrict__ part_mm) {
  float mn = INFINITY, mx = -INFINITY;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = a[i];
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
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
  if (threadIdx.x == 0) {
    float* p = part_mm + ((long long)s * REF_MM_BLOCKS + blockIdx.x) * 2;
    p[0] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    p[1] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
  }
}
__global__ __launch_bounds__(256) void ref_minmax_kernel(RefSentences rs, long long n, float* __restrict__ part_mm) {
  ref_minmax_body(rs.attn[blockIdx.y], blockIdx.y, n, part_mm);
}

// min / max are exact in any order: every wave folds the 64 partial pairs of a map itself (one pair per lane)
__device__ __forceinline__ void ref_fold_minmax(const float* __restrict__ part_mm, int s, int lane, float& mn, float& mx) {
  const float* p = part_mm + ((long long)s * REF_MM_BLOCKS + lane) * 2;
  mn = p[0];
  mx = p[1];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o));
    mx = fmaxf(mx, __shfl_xor(mx, o));
  }
}

// masked_pool_kernel for the S heat-maps of a ref in ONE launch: grid (pixel blocks, mask groups); the maps are looped INSIDE
// the workgroup (pool_block, four at a time).  History: a first version (round 3) kept four maps' tiles in registers with a
// double-precision sum per pixel and map -- 260 VGPRs, 8.5 k instructions, slower than three launches; rounds 4-5 made every
// (block, group, map) its own small workgroup and re-read the mask bytes per map from L2; the fp32 fma form of pool_block needs
// 16 VGPRs per map and fetches every mask byte once.  Partials in the layout of masked_pool_kernel.
// the S heat-maps of a ref over one FULL pixel block and one group of MG masks: the maps go through pool_block four at a time
// (the mask bytes of the block are fetched once per four maps: once for the three sentences of a RefCOCO ref)
// CH = maps per pass: 4, or 3 for the launches whose refs have at most three sentences (RefCOCO: 16 VGPRs fewer per map held
// -> 4 instead of 3 waves per SIMD)
template <int MG, int CH>
__device__ __forceinline__ void ref_masked_pool_body(int blk, const float* const* __restrict__ attn, const int* __restrict__ dirflag, int S, int mask_group,
                                                     const uint8_t* __restrict__ masks, int N, int H, int W,
                                                     const float* __restrict__ part_mm, double* __restrict__ part_sum,
                                                     unsigned* __restrict__ part_cnt, double* __restrict__ part_tot, int nparts) {
  const int lane = threadIdx.x & 63;
  const int n0 = mask_group * MG, n1 = min(N, n0 + MG);
  const bool wt = mask_group == 0;
  for (int s0 = 0; s0 < S; s0 += CH) {
    const int ns = min(CH, S - s0);
    PoolMap pm[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int sj = min(s0 + j, S - 1);
      float mn, mx;
      ref_fold_minmax(part_mm, sj, lane, mn, mx);
      pm[j].attn = attn[sj]; pm[j].dirflag = dirflag[sj]; pm[j].sidx = sj; pm[j].mn = mn; pm[j].range = mx - mn;
    }
    if (CH == 4 && ns == 4) {
      if constexpr (CH == 4) pool_block<4, true>(blk, pm, n0, n1, masks, N, H, W, wt, part_sum, part_cnt, part_tot, nparts);
    } else if (ns == 3) {
      const PoolMap p3[3] = {pm[0], pm[1], pm[2]};
      pool_block<3, true>(blk, p3, n0, n1, masks, N, H, W, wt, part_sum, part_cnt, part_tot, nparts);
    } else if (ns == 2) {
      const PoolMap p2[2] = {pm[0], pm[1]};
      pool_block<2, true>(blk, p2, n0, n1, masks, N, H, W, wt, part_sum, part_cnt, part_tot, nparts);
    } else {
      const PoolMap p1[1] = {pm[0]};
      pool_block<1, true>(blk, p1, n0, n1, masks, N, H, W, wt, part_sum, part_cnt, part_tot, nparts);
    }
  }
}
// ... and over the partial LAST block of the plane, one map at a time
template <int MG>
__device__ __forceinline__ void ref_masked_pool_last(int blk, const float* const* __restrict__ attn, const int* __restrict__ dirflag, int S, int mask_group,
                                                     const uint8_t* __restrict__ masks, int N, int H, int W,
                                                     const float* __restrict__ part_mm, double* __restrict__ part_sum,
                                                     unsigned* __restrict__ part_cnt, double* __restrict__ part_tot, int nparts) {
  const int lane = threadIdx.x & 63;
  const int n0 = mask_group * MG, n1 = min(N, n0 + MG);
  for (int sj = 0; sj < S; ++sj) {
    float mn, mx;
    ref_fold_minmax(part_mm, sj, lane, mn, mx);
    const PoolMap pm[1] = {{attn[sj], dirflag[sj], sj, mn, mx - mn}};
    pool_block<1, false>(blk, pm, n0, n1, masks, N, H, W, mask_group == 0, part_sum, part_cnt, part_tot, nparts);
  }
}

template <int CH>
__global__ __launch_bounds__(256) void ref_masked_pool_kernel(RefSentences rs, int S, const uint8_t* __restrict__ masks, int N, int H,
                                                              int W, const float* __restrict__ part_mm, double* __restrict__ part_sum,
                                                              unsigned* __restrict__ part_cnt, double* __restrict__ part_tot,
                                                              int nparts) {
  ref_masked_pool_body<MASK_GROUP, CH>(blockIdx.x, rs.attn, rs.dirflag, S, blockIdx.y, masks, N, H, W, part_mm, part_sum, part_cnt, part_tot, nparts);
}
__global__ __launch_bounds__(256) void ref_masked_pool_last_kernel(RefSentences rs, int S, const uint8_t* __restrict__ masks, int N, int H,
                                                                   int W, const float* __restrict__ part_mm, double* __restrict__ part_sum,
                                                                   unsigned* __restrict__ part_cnt, double* __restrict__ part_tot,
                                                                   int nparts, int blk) {
  ref_masked_pool_last<MASK_GROUP>(blk, rs.attn, rs.dirflag, S, blockIdx.y, masks, N, H, W, part_mm, part_sum, part_cnt, part_tot, nparts);
}

// one workgroup per sentence: coherence_final_kernel's reduction for its N masks, then the sentence's scoring
__device__ __forceinline__ void ref_score_body(const RefSentences& rs, int s, bool first, const float* __restrict__ hybrid,
                                               const float* __restrict__ text, const long long* __restrict__ boxes, int N, int E, int H,
                                               int W, float logit_scale, float r_mix, int k1, int k2, float alpha,
                                               const double* __restrict__ part_sum, const unsigned* __restrict__ part_cnt,
                                               const double* __restrict__ part_tot, int nparts, float* __restrict__ gem_all,
                                               float* __restrict__ clip_all, float* __restrict__ neg_all, float* __restrict__ soft_all,
                                               int* __restrict__ idx_all, unsigned long long* __restrict__ iu_all,
                                               unsigned* __restrict__ done) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long HW = (long long)H * W;
  float* gem = gem_all + (long long)s * N;
  const float black = rs.black[s];
  // coherence_final_kernel's reduction for this sentence's N masks.  One wave per mask, lanes stride over the per-wave
  // partials -- but a plain `for (b = lane; b < nparts; b += 64)` is one memory round trip per trip, and sixteen masks per
  // wave one after the other made this workgroup the longest kernel of the tail (130 us).  The total is summed ONCE per wave
  // (it is the same for every mask), and four masks' partials are fetched in batches of up to eight trips before they are
  // added -- in the same order as before: identical sums.
  constexpr int MR = 4, UB = 8;
  double tot = 0.0;
  for (int b0 = lane; b0 < nparts; b0 += 64 * UB) {
    double tv[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) tv[u] = part_tot[(long long)s * nparts + min(b0 + 64 * u, nparts - 1)];
#pragma unroll
    for (int u = 0; u < UB; ++u)
      if (b0 + 64 * u < nparts) tot += tv[u];
  }
  tot = wave_sum_d(tot);
  for (int nb = wave * MR; nb < N; nb += 4 * MR) {
    double sm[MR];
    unsigned long long c[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) { sm[i] = 0.0; c[i] = 0; }
    for (int b0 = lane; b0 < nparts; b0 += 64 * UB) {
      double sv[MR][UB];
      unsigned cv[MR][UB];
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        const int n = min(nb + i, N - 1);
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const long long b = min(b0 + 64 * u, nparts - 1);
          sv[i][u] = part_sum[((long long)s * nparts + b) * N + n];
          cv[i][u] = part_cnt[b * N + n];
        }
      }
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int u = 0; u < UB; ++u)
          if (b0 + 64 * u < nparts) { sm[i] += sv[i][u]; c[i] += cv[i][u]; }
    }
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      double smr = wave_sum_d(sm[i]);
      unsigned long long cr = c[i];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) cr += __shfl_xor(cr, o);
      if (lane == 0 && nb + i < N) {
        const double mean = tot / (double)HW;
        const double in_sum = smr / mean, out_sum = (tot - smr) / mean;
        const double a = (double)(2.f - black) * in_sum / (double)cr;
        const double bterm = (double)black * out_sum / (double)(HW - (long long)cr);
        gem[nb + i] = (float)(a - bterm);
      }
    }
  }
  if (threadIdx.x < 4) iu_all[4 * s + threadIdx.x] = 0ull;
  if (first && threadIdx.x == 0) *done = 0u;
  __syncthreads();
  __threadfence_block();
  score_sentence_body(hybrid, text + (long long)rs.sent_row[s] * E, text + (long long)rs.nphr_row[s] * E,
                      text + (long long)rs.other_row0[s] * E, rs.n_other[s], r_mix, boxes, gem, N, E, logit_scale, k1, k2, alpha,
                      rs.rela[s], rs.has_other[s], idx_all + 2 * s, clip_all + (long long)s * N, neg_all + (long long)s * N,
                      soft_all + (long long)s * (2 * N + 2 * E));
}

__global__ __launch_bounds__(256) void ref_score_kernel(RefSentences rs, const float* __restrict__ hybrid, const float* __restrict__ text,
                                                        const long long* __restrict__ boxes, int N, int E, int H, int W,
                                                        float logit_scale, float r_mix, int k1, int k2, float alpha,
                                                        const double* __restrict__ part_sum, const unsigned* __restrict__ part_cnt,
                                                        const double* __restrict__ part_tot, int nparts, float* __restrict__ gem_all,
                                                        float* __restrict__ clip_all, float* __restrict__ neg_all,
                                                        float* __restrict__ soft_all, int* __restrict__ idx_all,
                                                        unsigned long long* __restrict__ iu_all, unsigned* __restrict__ done) {
  ref_score_body(rs, blockIdx.x, blockIdx.x == 0, hybrid, text, boxes, N, E, H, W, logit_scale, r_mix, k1, k2, alpha, part_sum, part_cnt,
                 part_tot, nparts, gem_all, clip_all, neg_all, soft_all, idx_all, iu_all, done);
}

// Compute_IoU of both winners of every sentence: grid (blocks, 2 S); the last block to finish adds the ref's counts to
// the running accumulators cum_I, cum_U, cum_I_final, cum_U_final (integers: order-free)
// I / U of one block's share of a (mask, target) pair, reduced over the block: valid in thread 0
__device__ __forceinline__ void ref_iou_block(const uint8_t* __restrict__ p, const uint8_t* __restrict__ g, long long n, unsigned& Iout,
                                              unsigned& Uout) {
  unsigned I = 0, U = 0;
  const bool al = ((((uintptr_t)p) | ((uintptr_t)g)) & 15) == 0;
  const long long n16 = al ? n / 16 : 0;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
    const uint4 a = ((const uint4*)p)[i], b = ((const uint4*)g)[i];
    const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      unsigned x = aw[w], y = bw[w];
      x |= x >> 4; x |= x >> 2; x |= x >> 1; x &= 0x01010101u;
      y |= y >> 4; y |= y >> 2; y |= y >> 1; y &= 0x01010101u;
      I += __popc(x & y);
      U += __popc(x | y);
    }
  }
  for (long long i = n16 * 16 + blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const bool a = p[i] != 0, b = g[i] != 0;
    I += (a && b);
    U += (a || b);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    I += __shfl_xor(I, o);
    U += __shfl_xor(U, o);
  }
  __shared__ unsigned sI[4], sU[4];
  if ((threadIdx.x & 63) == 0) {
    sI[threadIdx.x >> 6] = I;
    sU[threadIdx.x >> 6] = U;
  }
  __syncthreads();
  Iout = (sI[0] + sI[1]) + (sI[2] + sI[3]);
  Uout = (sU[0] + sU[1]) + (sU[2] + sU[3]);
}

__global__ __launch_bounds__(256) void ref_iou_kernel(RefSentences rs, int S, const uint8_t* __restrict__ masks, long long n,
                                                      const int* __restrict__ idx_all, unsigned long long* __restrict__ iu_all,
                                                      unsigned long long* __restrict__ cum, unsigned* __restrict__ done) {
  const int s = blockIdx.y >> 1, which = blockIdx.y & 1;
  unsigned I, U;
  ref_iou_block(masks + (long long)idx_all[2 * s + which] * n, rs.target[s], n, I, U);
  // one pair of atomics per BLOCK (thousands of 64-bit atomics on two dozen addresses serialise in L2: 68 us for six IoUs)
  __shared__ int last;
  if (threadIdx.x == 0) {
    atomicAdd(&iu_all[4 * s + 2 * which], (unsigned long long)I);
    atomicAdd(&iu_all[4 * s + 2 * which + 1], (unsigned long long)U);
    __threadfence();
    last = atomicAdd(done, 1u) == gridDim.x * gridDim.y - 1;
  }
  __syncthreads();
  if (last && cum != nullptr && threadIdx.x < 4) {
    unsigned long long a = 0;
    for (int j = 0; j < S; ++j) a += atomicAdd(&iu_all[4 * j + threadIdx.x], 0ull);   // coherent read of the other blocks' sums
    cum[threadIdx.x] += a;
  }
}

// ---- the tail of a whole GROUP of refs in the same four launches (hgl_score_group) -------------------------------------------
// hgl_score_ref is four launches per ref of ~31 MB each: latency-bound by construction (pooling 33 us per ref = 0.9 TB/s on
// its algorithmic bytes).  The grouped loop scores the refs of a group (16 images) back to back, so the same kernels run over
// ALL of them at once: a table of per-ref descriptors in device memory (shapes differ from ref to ref), grids sized by the
// largest ref with the others' surplus blocks leaving at once, ~0.5 GB per pooling launch.  A block executes the per-ref
// kernels' bodies on its ref's operands: identical arithmetic and reduction order, identical rows.
constexpr int GRP_MAXR = 16;       // refs per launch (the host loops over chunks)
struct GroupRefDev {
  RefSentences rs;
  int S, N, H, W, k1, k2, nparts, nblk;
  const float *hybrid, *text;
  const long long* boxes;
  const uint8_t* masks;
  float* part_mm;
  double* part_sum;
  unsigned* part_cnt;
  double* part_tot;
  float *gem, *clip, *neg, *soft;
  int* idx;
  unsigned long long* iu;
};

__global__ __launch_bounds__(256) void grp_minmax_kernel(const GroupRefDev* __restrict__ tab) {
  const GroupRefDev& g = tab[blockIdx.z];
  const int s = blockIdx.y;
  if (s >= g.S) return;
  ref_minmax_body(g.rs.attn[s], s, (long long)g.H * g.W, g.part_mm);
}

template <int CH>
__global__ __launch_bounds__(256) void grp_masked_pool_kernel(const GroupRefDev* __restrict__ tab) {
  const GroupRefDev& g = tab[blockIdx.z];
  const int nfull = (int)(((long long)g.H * g.W) / PIX_PER_BLOCK);
  if ((int)blockIdx.x >= nfull || (int)blockIdx.y * REF_MASK_GROUP >= g.N) return;
  ref_masked_pool_body<REF_MASK_GROUP, CH>(blockIdx.x, g.rs.attn, g.rs.dirflag, g.S, blockIdx.y, g.masks, g.N, g.H, g.W, g.part_mm, g.part_sum,
                                       g.part_cnt, g.part_tot, g.nparts);
}
__global__ __launch_bounds__(256) void grp_masked_pool_last_kernel(const GroupRefDev* __restrict__ tab) {
  const GroupRefDev& g = tab[blockIdx.z];
  const long long HW = (long long)g.H * g.W;
  if (HW % PIX_PER_BLOCK == 0 || (int)blockIdx.y * REF_MASK_GROUP >= g.N) return;
  ref_masked_pool_last<REF_MASK_GROUP>((int)(HW / PIX_PER_BLOCK), g.rs.attn, g.rs.dirflag, g.S, blockIdx.y, g.masks, g.N, g.H, g.W, g.part_mm,
                                       g.part_sum, g.part_cnt, g.part_tot, g.nparts);
}

__global__ __launch_bounds__(256) void grp_score_kernel(const GroupRefDev* __restrict__ tab, int E, float logit_scale, float r_mix,
                                                        float alpha, unsigned* __restrict__ done) {
  const GroupRefDev& g = tab[blockIdx.y];
  const int s = blockIdx.x;
  if (s >= g.S) return;
  ref_score_body(g.rs, s, blockIdx.x == 0 && blockIdx.y == 0, g.hybrid, g.text, g.boxes, g.N, E, g.H, g.W, logit_scale, r_mix, g.k1, g.k2,
                 alpha, g.part_sum, g.part_cnt, g.part_tot, g.nparts, g.gem, g.clip, g.neg, g.soft, g.idx, g.iu, done);
}

// grid (blocks, 2 maxS, R): every block counts on `done` (those beyond their ref's sentences do nothing else); the last one adds
// all sentences' sums of all refs to the running accumulators (integers: order-free)
__global__ __launch_bounds__(256) void grp_iou_kernel(const GroupRefDev* __restrict__ tab, int R, unsigned long long* __restrict__ cum,
                                                      unsigned* __restrict__ done) {
  const GroupRefDev& g = tab[blockIdx.z];
  const int s = blockIdx.y >> 1, which = blockIdx.y & 1;
  unsigned I = 0, U = 0;
  const bool live = s < g.S;
  if (live) ref_iou_block(g.masks + (long long)g.idx[2 * s + which] * g.H * g.W, g.rs.target[s], (long long)g.H * g.W, I, U);
  __shared__ int last;
  if (threadIdx.x == 0) {
    if (live) {
      atomicAdd(&g.iu[4 * s + 2 * which], (unsigned long long)I);
      atomicAdd(&g.iu[4 * s + 2 * which + 1], (unsigned long long)U);
    }
    __threadfence();
    last = atomicAdd(done, 1u) == gridDim.x * gridDim.y * gridDim.z - 1;
  }
  __syncthreads();
  // the last block folds every (ref, sentence) counter into the four accumulators: one counter per THREAD, read with an
  // atomic (the other blocks' adds are device-scope atomics), summed through LDS -- integers: any order.  (Four threads walking
  // the R x S counters one after the other were 48 dependent atomic round trips: 80 of this launch's 102 us.)
  __shared__ unsigned long long acc[4];
  if (last && cum != nullptr) {
    if (threadIdx.x < 4) acc[threadIdx.x] = 0ull;
    __syncthreads();
    const int maxS = (int)gridDim.y >> 1;
    for (int e = threadIdx.x; e < R * maxS * 4; e += 256) {
      const int r = e / (maxS * 4), j = (e >> 2) % maxS, k = e & 3;
      if (j < tab[r].S) atomicAdd(&acc[k], atomicAdd(&tab[r].iu[4 * j + k], 0ull));
    }
    __syncthreads();
    if (threadIdx.x < 4) cum[threadIdx.x] += acc[threadIdx.x];
  }
}

// ---- image synthesis: local (mean-filled) and global (blurred background) views ----
__global__ __launch_bounds__(256) void synth_views_kernel(
    const uint8_t* __restrict__ sam_img, const uint8_t* __restrict__ blurred,
    const float* __restrict__ image_norm, const uint8_t* __restrict__ masks, int N, int H, int W,
    int res, float* __restrict__ local_imgs, float* __restrict__ global_imgs) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long total = (long long)N * res * res;
  if (i >= total) return;
  const int ox = (int)(i % res), oy = (int)((i / res) % res), n = (int)(i / ((long long)res * res));
  const float sy = (float)H / (float)res, sx = (float)W / (float)res;
  float fy = fmaf(sy, oy + 0.5f, -0.5f), fx = fmaf(sx, ox + 0.5f, -0.5f);  // one fma, as ATen
  fy = fy < 0.f ? 0.f : fy;
  fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly1 = fy - y0, lx1 = fx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const long long HW = (long long)H * W;
  const uint8_t* m = masks + (long long)n * HW;
  const long long t00 = (long long)y0 * W + x0, t01 = (long long)y0 * W + x1;
  const long long t10 = (long long)y1 * W + x0, t11 = (long long)y1 * W + x1;
  const bool m00 = m[t00] != 0, m01 = m[t01] != 0, m10 = m[t10] != 0, m11 = m[t11] != 0;
  const float clip_mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};   // Hybridgl_main.py:93
  const float in_mean[3] = {0.485f, 0.456f, 0.406f};                   // Hybridgl_main.py:117
  const float in_std[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* im = image_norm + (long long)c * HW;
    const float l00 = m00 ? im[t00] : clip_mean[c], l01 = m01 ? im[t01] : clip_mean[c];
    const float l10 = m10 ? im[t10] : clip_mean[c], l11 = m11 ? im[t11] : clip_mean[c];
    const float lv = ly0 * (lx0 * l00 + lx1 * l01) + ly1 * (lx0 * l10 + lx1 * l11);
    const float g00 = (float)(m00 ? sam_img[t00 * 3 + c] : blurred[t00 * 3 + c]) / 255.f;
    const float g01 = (float)(m01 ? sam_img[t01 * 3 + c] : blurred[t01 * 3 + c]) / 255.f;
    const float g10 = (float)(m10 ? sam_img[t10 * 3 + c] : blurred[t10 * 3 + c]) / 255.f;
    const float g11 = (float)(m11 ? sam_img[t11 * 3 + c] : blurred[t11 * 3 + c]) / 255.f;
    const float gv = ly0 * (lx0 * g00 + lx1 * g01) + ly1 * (lx0 * g10 + lx1 * g11);
    const long long o = (((long long)n * 3 + c) * res + oy) * res + ox;
    local_imgs[o] = lv;
    global_imgs[o] = (gv - in_mean[c]) / in_std[c];
  }
}

// ---- the reference's free helpers as standalone kernels (utils.py:135-161 gen_dir_mask, :240-268 relation_boxes) ----
// The fused tail (coherence_scores, score_sentence) does not call these; they exist so that code written against
// the reference's utils.py finds the same functions, evaluated by the same device arithmetic.
__global__ __launch_bounds__(256) void gen_dir_mask_kernel(int dirflag, int H, int W, float* __restrict__ out) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= (long long)H * W) return;
  out[i] = dir_weight(dirflag, (int)(i % W), W);
}

__global__ __launch_bounds__(256) void relation_boxes_kernel(const long long* __restrict__ bi, const long long* __restrict__ bj,
                                                             const float* __restrict__ si, const float* __restrict__ sj, int n,
                                                             int rela, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = relation_boxes_dev(bi + 4 * i, bj + 4 * i, si[i], sj[i], rela);
}

// ---- blurred background of the global views (Hybridgl_main.py:99 cv2.GaussianBlur(img, (15,15), 0)) ----------
// OpenCV is a third-party package that is absent offline and its 8-bit fixed-point kernel is unpinned
// (SURVEY.md 8f-2); this is the package's own definition (hybridgl_amd/synth.py box_blur_u8: separable Gaussian with
// OpenCV's sigma rule, reflect-101 borders, double accumulation in tap order, round half up), evaluated with the same
// double operations in the same order so that device and host agree bit for bit.
struct BlurTaps { double w[31]; int k; };

__device__ __forceinline__ int reflect101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

__global__ __launch_bounds__(256) void blur_v_kernel(const uint8_t* __restrict__ img, int H, int W, int C, BlurTaps t,
                                                     double* __restrict__ tmp) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  const long long total = (long long)H * W * C;
  if (i >= total) return;
  const long long wc = (long long)W * C;
  const int y = (int)(i / wc);
  const long long rest = i - (long long)y * wc;
  const int r = t.k / 2;
  double a = 0.0;
  for (int j = 0; j < t.k; ++j) {
    const int yy = reflect101(y + j - r, H);
    const double p = (double)img[(long long)yy * wc + rest];
    a = j == 0 ? __dmul_rn(t.w[0], p) : __dadd_rn(a, __dmul_rn(t.w[j], p));
  }
  tmp[i] = a;
}

__global__ __launch_bounds__(256) void blur_h_kernel(const double* __restrict__ tmp, int H, int W, int C, BlurTaps t,
                                                     uint8_t* __restrict__ out) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  const long long total = (long long)H * W * C;
  if (i >= total) return;
  const int c = (int)(i % C);
  const int x = (int)((i / C) % W);
  const long long y = i / ((long long)W * C);
  const int r = t.k / 2;
  double a = 0.0;
  for (int j = 0; j < t.k; ++j) {
    const int xx = reflect101(x + j - r, W);
    const double p = tmp[(y * W + xx) * C + c];
    a = j == 0 ? __dmul_rn(t.w[0], p) : __dadd_rn(a, __dmul_rn(t.w[j], p));
  }
  double v = floor(__dadd_rn(a, 0.5));
  v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
  out[i] = (uint8_t)v;
}

// ---- cv2.GaussianBlur on uint8 (OpenCV 4.x bit-exact path, smooth.simd.hpp fixedSmoothInvoker): both passes in
// integer arithmetic with 8.8 fixed-point taps that sum to 256 -- the row pass gives sum(src * kx) <= 255 * 256 in 16
// bits, the column pass sum(ky * row) in 32 bits, rounded once: (v + 2^15) >> 16.  Reflect-101 borders.
struct BlurTapsQ8 { uint16_t x[31], y[31]; int k; };

__global__ __launch_bounds__(256) void blur_q8_h_kernel(const uint8_t* __restrict__ img, int H, int W, int C, BlurTapsQ8 t,
                                                        uint16_t* __restrict__ tmp) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  const long long total = (long long)H * W * C;
  if (i >= total) return;
  const int c = (int)(i % C);
  const int x = (int)((i / C) % W);
  const long long y = i / ((long long)W * C);
  const int r = t.k / 2;
  unsigned a = 0;
  for (int j = 0; j < t.k; ++j) a += (unsigned)t.x[j] * img[(y * W + reflect101(x + j - r, W)) * C + c];
  tmp[i] = (uint16_t)a;
}

__global__ __launch_bounds__(256) void blur_q8_v_kernel(const uint16_t* __restrict__ tmp, int H, int W, int C, BlurTapsQ8 t,
                                                        uint8_t* __restrict__ out) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  const long long total = (long long)H * W * C;
  if (i >= total) return;
  const long long wc = (long long)W * C;
  const int y = (int)(i / wc);
  const long long rest = i - (long long)y * wc;
  const int r = t.k / 2;
  unsigned a = 0;
  for (int j = 0; j < t.k; ++j) a += (unsigned)t.y[j] * tmp[(long long)reflect101(y + j - r, H) * wc + rest];
  const unsigned v = (a + 32768u) >> 16;
  out[i] = (uint8_t)(v > 255u ? 255u : v);
}

// Both passes in one launch: a workgroup owns a BLUR_TH x BLUR_TW pixel tile, stages the tile and its halo (k / 2 rows and
// columns each side, reflect-101 at the image border) in LDS as bytes, runs the row pass into 16-bit LDS sums and the column pass
// out of them: the image is fetched from HBM once (the halos of neighbouring tiles hit in L2) and no 16-bit intermediate image
// is written and read back (the two-launch form above: every tap of every pixel through the cache, 5.6x the image fetched).
// Same integer arithmetic, same bits.  A thread owns one byte column of the tile in every phase and walks the rows (no
// division inside the loops; the loads of consecutive rows are independent); the taps sit in LDS (indexed from the kernel
// arguments every tap of every row was a scalar load).  Small tiles on purpose: the work of an image is tiny and a thread's
// chain of rows x taps is what the launch takes -- 8 x 32 tiles give 1200 workgroups for 480 x 640.
constexpr int BLUR_TH = 8, BLUR_TW = 32, BLUR_RMAX = 15, BLUR_CMAX = 4, BLUR_NT = 128;
__global__ __launch_bounds__(BLUR_NT) void blur_q8_tile_kernel(const uint8_t* __restrict__ img, int H, int W, int C, BlurTapsQ8 t,
                                                               uint8_t* __restrict__ out) {
  __shared__ uint8_t src[(BLUR_TH + 2 * BLUR_RMAX) * (BLUR_TW + 2 * BLUR_RMAX) * BLUR_CMAX];
  __shared__ uint16_t rsum[(BLUR_TH + 2 * BLUR_RMAX) * BLUR_TW * BLUR_CMAX];
  __shared__ unsigned tapx[32], tapy[32];
  // XCD-aware tile map: workgroups go to the eight XCDs round-robin, each XCD has its own L2 -- with the plain 2-D grid the
  // neighbours of a tile sat on other XCDs and every L2 fetched the shared halo rows for itself (7.6 MB moved per 2.5 MB
  // image).  Here XCD x works through one contiguous run of the row-major tile list: a horizontal band of the image.
  const int gx = (W + BLUR_TW - 1) / BLUR_TW, ntiles = gx * ((H + BLUR_TH - 1) / BLUR_TH), per = (ntiles + 7) >> 3;
  const int tile = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || tile >= ntiles) return;      // (uniform; before any barrier)
  const int ty = tile / gx;
  const int r = t.k / 2, x0 = (tile - ty * gx) * BLUR_TW, y0 = ty * BLUR_TH;
  const int rows = BLUR_TH + 2 * r, cols = BLUR_TW + 2 * r, rowb = cols * C, outb = BLUR_TW * C;
  if (threadIdx.x < 31) { tapx[threadIdx.x] = t.x[threadIdx.x]; tapy[threadIdx.x] = t.y[threadIdx.x]; }
  for (int cb = threadIdx.x; cb < rowb; cb += BLUR_NT) {
    const int xx = cb / C, c = cb - xx * C;
    const uint8_t* const col = img + (long long)reflect101(min(x0 - r + xx, W - 1 + r), W) * C + c;
#pragma unroll 8
    for (int yy = 0; yy < rows; ++yy)
      src[yy * rowb + cb] = col[(long long)reflect101(min(y0 - r + yy, H - 1 + r), H) * W * C];
  }
  __syncthreads();
  for (int ob = threadIdx.x; ob < outb; ob += BLUR_NT) {
#pragma unroll 2
    for (int yy = 0; yy < rows; ++yy) {
      const uint8_t* s0 = src + yy * rowb + ob;
      unsigned a = 0;
      for (int j = 0; j < t.k; ++j) a += tapx[j] * s0[j * C];
      rsum[yy * outb + ob] = (uint16_t)a;
    }
  }
  __syncthreads();
  for (int ob = threadIdx.x; ob < outb; ob += BLUR_NT) {
    if (x0 + ob / C >= W) continue;
    const int ny = min(BLUR_TH, H - y0);
#pragma unroll 2
    for (int yy = 0; yy < ny; ++yy) {
      const uint16_t* s0 = rsum + yy * outb + ob;
      unsigned a = 0;
      for (int j = 0; j < t.k; ++j) a += tapy[j] * s0[j * outb];
      const unsigned v = (a + 32768u) >> 16;
      out[((long long)(y0 + yy) * W + x0) * C + ob] = (uint8_t)(v > 255u ? 255u : v);
    }
  }
}

}  // namespace

extern "C" {

int hgl_calculate_score(const float* img, const float* txt, int N, int T, int E, float logit_scale,
                        float* logits, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(img && txt && logits && N > 0 && T > 0 && E > 0, "calculate_score: bad arguments");
  const int pairs = N * T;
  hipLaunchKernelGGL(calc_score_kernel, dim3((pairs + 3) / 4), dim3(256), 0, (hipStream_t)stream, img, txt, N, T, E, logit_scale, logits);
  return hgl_check_launch("calculate_score");
}

static inline int coh_nblk(int H, int W) {
  return (int)(((long long)H * W + PIX_PER_BLOCK - 1) / PIX_PER_BLOCK);
}
static inline int coh_nparts(int H, int W) { return coh_nblk(H, W) * 4; }

size_t hgl_coherence_workspace_bytes(int N, int H, int W) {
  const size_t nb = coh_nparts(H, W);
  return 256 + hgl_align_up(nb * N * sizeof(double), 256) + hgl_align_up(nb * N * sizeof(unsigned), 256) +
         hgl_align_up(nb * sizeof(double), 256);
}

int hgl_coherence_scores(const float* imgattn, const uint8_t* masks, int N, int H, int W, int dirflag,
                         float black, float* score, void* workspace, size_t workspace_bytes,
                         void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(imgattn && masks && score && N > 0 && H > 0 && W > 0, "coherence: bad arguments");
  HGL_REQUIRE(dirflag >= 0 && dirflag <= 3, "coherence: bad dirflag %d", dirflag);
  if (!workspace || workspace_bytes < hgl_coherence_workspace_bytes(N, H, W)) {
    hgl_set_error("coherence: workspace too small (%zu < %zu)", workspace_bytes, hgl_coherence_workspace_bytes(N, H, W));
    return HGL_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int nb = coh_nparts(H, W);
  HglArena ar(workspace, workspace_bytes);
  int* stats = ar.take<int>(64);
  double* psum = ar.take<double>((size_t)nb * N);
  unsigned* pcnt = ar.take<unsigned>((size_t)nb * N);
  double* ptot = ar.take<double>(nb);
  const long long HW = (long long)H * W;
  hipLaunchKernelGGL(init_stats_kernel, dim3(1), dim3(1), 0, st, stats);
  hipLaunchKernelGGL(minmax_kernel, dim3(256), dim3(256), 0, st, imgattn, HW, stats);
  {
    const int nfull = (int)(((long long)H * W) / PIX_PER_BLOCK), ngrp = (N + MASK_GROUP - 1) / MASK_GROUP;
    if (nfull > 0)
      hipLaunchKernelGGL(masked_pool_kernel<true>, dim3(nfull, ngrp), dim3(256), 0, st, imgattn, masks, N, H, W, dirflag, stats, psum, pcnt, ptot, 0);
    if (coh_nblk(H, W) > nfull)
      hipLaunchKernelGGL(masked_pool_kernel<false>, dim3(1, ngrp), dim3(256), 0, st, imgattn, masks, N, H, W, dirflag, stats, psum, pcnt, ptot, nfull);
  }
  hipLaunchKernelGGL(coherence_final_kernel, dim3((N + 3) / 4), dim3(256), 0, st, psum, pcnt, ptot, nb, N, HW, black, score);
  return hgl_check_launch("coherence_scores");
}

int hgl_iou(const uint8_t* pred, const uint8_t* gt, long long HW, int64_t* out_IU, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(pred && gt && out_IU && HW > 0, "iou: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(out_IU, 0, 2 * sizeof(int64_t), st) != hipSuccess) {
    hgl_set_error("iou: memset failed");
    return HGL_ELAUNCH;
  }
  long long blocks = (HW / 16 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(iou_kernel, dim3((unsigned)blocks), dim3(256), 0, st, pred, gt, HW, (unsigned long long*)out_IU);
  return hgl_check_launch("iou");
}

int hgl_iou_select(const uint8_t* masks, const int32_t* idx, int which, const uint8_t* gt,
                   long long HW, int64_t* out_IU, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(masks && idx && gt && out_IU && HW > 0 && which >= 0, "iou_select: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(out_IU, 0, 2 * sizeof(int64_t), st) != hipSuccess) {
    hgl_set_error("iou_select: memset failed");
    return HGL_ELAUNCH;
  }
  long long blocks = (HW / 16 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(iou_select_kernel, dim3((unsigned)blocks), dim3(256), 0, st, masks, (const int*)idx, which, gt, HW, (unsigned long long*)out_IU);
  return hgl_check_launch("iou_select");
}

size_t hgl_score_sentence_workspace_bytes(int N, int E) {
  return hgl_align_up(((size_t)2 * N + 2 * (size_t)E) * sizeof(float), 256);
}

int hgl_score_sentence(const float* hybrid, const float* sentence_feat, const float* noun_phrase_feat,
                       const float* other_noun_feats, int n_other, float r,
                       const int64_t* boxes, const float* gem_score, int N, int E, float logit_scale,
                       int k1, int k2, float alpha, int relaword, int has_other_nouns, int32_t* idx,
                       float* score_clip, float* score_neg, void* workspace, size_t workspace_bytes,
                       void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(hybrid && sentence_feat && noun_phrase_feat && boxes && gem_score && idx && score_clip && score_neg, "score_sentence: null argument");
  HGL_REQUIRE(n_other >= 0 && (n_other == 0 || other_noun_feats), "score_sentence: other_noun_feats missing");
  HGL_REQUIRE(N > 0 && E > 0, "score_sentence: bad shape");
  // Hybridgl_main.py:178-181: k clamps to the number of masks
  if (k1 > N) k1 = N;
  if (k2 > N) k2 = N;
  HGL_REQUIRE(k1 >= 1 && k1 <= MAXK && k2 >= 1 && k2 <= MAXK, "score_sentence: k1,k2 must be in [1,%d]", MAXK);
  HGL_REQUIRE(relaword >= 0 && relaword <= 7, "score_sentence: bad relaword %d", relaword);
  if (!workspace || workspace_bytes < hgl_score_sentence_workspace_bytes(N, E)) {
    hgl_set_error("score_sentence: workspace too small");
    return HGL_EWORKSPACE;
  }
  float* pool = (float*)workspace;
  hipLaunchKernelGGL(score_sentence_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, hybrid, sentence_feat, noun_phrase_feat, other_noun_feats, n_other, r, (const long long*)boxes, gem_score, N, E, logit_scale, k1, k2, alpha, relaword, has_other_nouns, (int*)idx, score_clip, score_neg, pool);
  return hgl_check_launch("score_sentence");
}

static size_t ref_ws_layout(int S, int N, int E, int H, int W, size_t* off) {
  const size_t nparts = coh_nparts(H, W);
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += hgl_align_up(bytes, 256); return at; };
  off[0] = take((size_t)S * REF_MM_BLOCKS * 2 * sizeof(float));          // part_mm
  off[1] = take((size_t)S * nparts * N * sizeof(double));                 // part_sum
  off[2] = take(nparts * (size_t)N * sizeof(unsigned));                   // part_cnt
  off[3] = take((size_t)S * nparts * sizeof(double));                     // part_tot
  off[4] = take((size_t)S * (2 * (size_t)N + 2 * (size_t)E) * sizeof(float));   // soft-max + text scratch per sentence
  off[5] = take((size_t)3 * S * N * sizeof(float));                       // gem / clip / neg when the caller does not want them
  off[6] = take(256);                                                     // done counter
  return o;
}

size_t hgl_score_ref_workspace_bytes(int S, int N, int E, int H, int W) {
  size_t off[7];
  return ref_ws_layout(S < REF_MAXS ? S : REF_MAXS, N, E, H, W, off);
}

int hgl_score_ref(const float* hybrid, const float* text, int T, const int64_t* boxes, const uint8_t* masks, int N, int E,
                  int H, int W, const HglSentence* sentences, int S, float logit_scale, float r, int k1, int k2, float alpha,
                  int32_t* idx, int64_t* iu, int64_t* cum, float* score_clip, float* score_neg, float* gem_score,
                  void* workspace, size_t workspace_bytes, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(hybrid && text && boxes && masks && sentences && idx && iu, "score_ref: null argument");
  HGL_REQUIRE(N > 0 && E > 0 && H > 0 && W > 0 && S > 0 && T > 0, "score_ref: bad shape");
  if (k1 > N) k1 = N;       // Hybridgl_main.py:178-181
  if (k2 > N) k2 = N;
  HGL_REQUIRE(k1 >= 1 && k1 <= MAXK && k2 >= 1 && k2 <= MAXK, "score_ref: k1,k2 must be in [1,%d]", MAXK);
  for (int s = 0; s < S; ++s) {
    const HglSentence& q = sentences[s];
    HGL_REQUIRE(q.imgattn && q.target, "score_ref: sentence %d has no heat-map / target", s);
    HGL_REQUIRE(q.sentence_row >= 0 && q.sentence_row < T && q.noun_phrase_row >= 0 && q.noun_phrase_row < T, "score_ref: sentence %d: text row out of range", s);
    HGL_REQUIRE(q.n_other >= 0 && (q.n_other == 0 || (q.other_row0 >= 0 && q.other_row0 + q.n_other <= T)), "score_ref: sentence %d: other-noun rows out of range", s);
    HGL_REQUIRE(q.dirflag >= 0 && q.dirflag <= 3 && q.relaword >= 0 && q.relaword <= 7, "score_ref: sentence %d: bad dirflag / relaword", s);
  }
  if (!workspace || workspace_bytes < hgl_score_ref_workspace_bytes(S, N, E, H, W)) {
    hgl_set_error("score_ref: workspace too small (%zu < %zu)", workspace_bytes, hgl_score_ref_workspace_bytes(S, N, E, H, W));
    return HGL_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const long long HW = (long long)H * W;
  const int nparts = coh_nparts(H, W);
  for (int s0 = 0; s0 < S; s0 += REF_MAXS) {
    const int sc = S - s0 < REF_MAXS ? S - s0 : REF_MAXS;
    size_t off[7];
    ref_ws_layout(sc, N, E, H, W, off);
    char* base = (char*)workspace;
    RefSentences rs;
    memset(&rs, 0, sizeof(rs));
    for (int j = 0; j < sc; ++j) {
      const HglSentence& q = sentences[s0 + j];
      rs.attn[j] = q.imgattn; rs.target[j] = q.target;
      rs.sent_row[j] = q.sentence_row; rs.nphr_row[j] = q.noun_phrase_row;
      rs.other_row0[j] = q.n_other > 0 ? q.other_row0 : 0; rs.n_other[j] = q.n_other;
      rs.dirflag[j] = q.dirflag; rs.rela[j] = q.relaword; rs.has_other[j] = q.has_other_nouns; rs.black[j] = q.black;
    }
    float* part_mm = (float*)(base + off[0]);
    double* psum = (double*)(base + off[1]);
    unsigned* pcnt = (unsigned*)(base + off[2]);
    double* ptot = (double*)(base + off[3]);
    float* soft = (float*)(base + off[4]);
    float* spare = (float*)(base + off[5]);
    unsigned* done = (unsigned*)(base + off[6]);
    float* gem = gem_score ? gem_score + (long long)s0 * N : spare;
    float* clip = score_clip ? score_clip + (long long)s0 * N : spare + (long long)sc * N;
    float* neg = score_neg ? score_neg + (long long)s0 * N : spare + 2ll * sc * N;
    hipLaunchKernelGGL(ref_minmax_kernel, dim3(REF_MM_BLOCKS, sc), dim3(256), 0, st, rs, HW, part_mm);
    {
      const int nfull = (int)(HW / PIX_PER_BLOCK), ngrp = (N + MASK_GROUP - 1) / MASK_GROUP;
      if (nfull > 0 && sc <= 3)
        hipLaunchKernelGGL(ref_masked_pool_kernel<3>, dim3(nfull, ngrp), dim3(256), 0, st, rs, sc, masks, N, H, W, part_mm, psum, pcnt, ptot, nparts);
      else if (nfull > 0)
        hipLaunchKernelGGL(ref_masked_pool_kernel<4>, dim3(nfull, ngrp), dim3(256), 0, st, rs, sc, masks, N, H, W, part_mm, psum, pcnt, ptot, nparts);
      if (coh_nblk(H, W) > nfull)
        hipLaunchKernelGGL(ref_masked_pool_last_kernel, dim3(1, ngrp), dim3(256), 0, st, rs, sc, masks, N, H, W, part_mm, psum, pcnt, ptot, nparts, nfull);
    }
    hipLaunchKernelGGL(ref_score_kernel, dim3(sc), dim3(256), 0, st, rs, hybrid, text, (const long long*)boxes, N, E, H, W, logit_scale,
                       r, k1, k2, alpha, psum, pcnt, ptot, nparts, gem, clip, neg, soft, (int*)idx + 2 * s0,
                       (unsigned long long*)iu + 4 * s0, done);
    long long blocks = (HW / 16 + 1023) / 1024;      // four 16-byte words per thread
    if (blocks < 1) blocks = 1;
    if (blocks > 32) blocks = 32;
    hipLaunchKernelGGL(ref_iou_kernel, dim3((unsigned)blocks, 2 * sc), dim3(256), 0, st, rs, sc, masks, HW, (const int*)idx + 2 * s0,
                       (unsigned long long*)iu + 4 * s0, (unsigned long long*)cum, done);
  }
  return hgl_check_launch("score_ref");
}

// ---- hgl_score_group: the refs of a group through ONE set of launches -------------------------------------------------------
namespace {
// workspace of one ref inside the group's workspace (the layout of hgl_score_ref's) + the descriptor table in front
size_t grp_ws_layout(const HglGroupRef* refs, int R, int E, size_t* ref_off) {
  size_t o = hgl_align_up((size_t)GRP_MAXR * sizeof(GroupRefDev), 256) + 256;     // table + the done counter
  for (int i = 0; i < R; ++i) {
    size_t off[7];
    if (ref_off) ref_off[i] = o;
    o += ref_ws_layout(refs[i].S, refs[i].N, E, refs[i].H, refs[i].W, off);
  }
  return o;
}
// pinned staging of the descriptor table: a ring of slots, a slot is reused only after the copy out of it has completed
// Pinned staging slots of hgl_score_group's descriptor table: one ring PER DEVICE (an event belongs to the device it was
// created on), a slot is held (busy) by one host thread from the moment it is picked until its event has been recorded behind
// the upload, and every event call is checked -- a failed record must not leave a slot looking reusable.
struct GrpStage {
  static constexpr int SLOTS = 8, DEVS = 64;
  struct Ring {
    void* buf[SLOTS] = {nullptr};
    hipEvent_t ev[SLOTS];
    bool used[SLOTS] = {false};     // the event has been recorded at least once
    bool busy[SLOTS] = {false};     // picked by a thread that has not recorded yet
    int next = 0;
  };
  Ring ring[DEVS];
  std::mutex mu;
  std::condition_variable freed;
};
GrpStage g_grp_stage;

// picks a slot of the current device's ring (waits while all eight are held by other threads); < 0 on error
int grp_stage_acquire(int* dev_out, void** host) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GrpStage::DEVS) {
    hgl_set_error("score_group: cannot tell the current device");
    return -1;
  }
  GrpStage::Ring& rg = g_grp_stage.ring[dev];
  int slot = -1;
  {
    std::unique_lock<std::mutex> lk(g_grp_stage.mu);
    for (;;) {
      for (int i = 0; i < GrpStage::SLOTS && slot < 0; ++i) {
        const int c = (rg.next + i) % GrpStage::SLOTS;
        if (!rg.busy[c]) slot = c;
      }
      if (slot >= 0) break;
      g_grp_stage.freed.wait(lk);
    }
    rg.busy[slot] = true;
    rg.next = (slot + 1) % GrpStage::SLOTS;
  }
  bool ok = true;
  if (!rg.buf[slot]) {       // only the holder of a busy slot touches its buffer and event
    ok = hipHostMalloc(&rg.buf[slot], GRP_MAXR * sizeof(GroupRefDev), hipHostMallocDefault) == hipSuccess;
    if (ok && hipEventCreateWithFlags(&rg.ev[slot], hipEventDisableTiming) != hipSuccess) {
      (void)hipHostFree(rg.buf[slot]);
      rg.buf[slot] = nullptr;
      ok = false;
    }
    if (!ok) hgl_set_error("score_group: cannot allocate the pinned descriptor slot");
  } else if (rg.used[slot]) {
    ok = hipEventSynchronize(rg.ev[slot]) == hipSuccess;      // the upload that last read this slot: eight calls ago
    if (!ok) hgl_set_error("score_group: waiting for a descriptor slot failed");
  }
  if (!ok) {
    (void)hipGetLastError();
    std::lock_guard<std::mutex> lk(g_grp_stage.mu);
    rg.busy[slot] = false;
    g_grp_stage.freed.notify_one();
    return -1;
  }
  *dev_out = dev;
  *host = rg.buf[slot];
  return slot;
}

// records the slot's event behind the upload on `st` and hands the slot back; false when the record failed (the caller then
// synchronises the stream itself before the slot can be rewritten)
bool grp_stage_release(int dev, int slot, hipStream_t st) {
  GrpStage::Ring& rg = g_grp_stage.ring[dev];
  bool ok = hipEventRecord(rg.ev[slot], st) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    ok = hipStreamSynchronize(st) == hipSuccess;      // no event to wait on later: make the upload complete now
  }
  std::lock_guard<std::mutex> lk(g_grp_stage.mu);
  rg.used[slot] = ok ? true : rg.used[slot];     // a failed record left the earlier event (if any) in place, and the stream was drained
  rg.busy[slot] = false;
  g_grp_stage.freed.notify_one();
  return ok;
}
}  // namespace

size_t hgl_score_group_workspace_bytes(const HglGroupRef* refs, int R, int E) {
  if (!refs || R <= 0 || E <= 0) return 0;
  size_t total = 0;
  for (int r0 = 0; r0 < R; r0 += GRP_MAXR) {
    const size_t n = grp_ws_layout(refs + r0, R - r0 < GRP_MAXR ? R - r0 : GRP_MAXR, E, nullptr);
    total = n > total ? n : total;
  }
  return total;
}

int hgl_score_group(const HglGroupRef* refs, int R, int E, float logit_scale, float r, float alpha, int64_t* cum, void* workspace,
                    size_t workspace_bytes, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(refs && R > 0 && E > 0, "score_group: bad arguments");
  for (int i = 0; i < R; ++i) {
    const HglGroupRef& q = refs[i];
    HGL_REQUIRE(q.hybrid && q.text && q.boxes && q.masks && q.sentences && q.idx && q.iu, "score_group: ref %d: null argument", i);
    HGL_REQUIRE(q.N > 0 && q.H > 0 && q.W > 0 && q.T > 0 && q.S > 0 && q.S <= REF_MAXS, "score_group: ref %d: bad shape (S %d: 1 .. %d)", i,
                q.S, REF_MAXS);
    HGL_REQUIRE((q.k1 < q.N ? q.k1 : q.N) >= 1 && q.k1 <= MAXK && (q.k2 < q.N ? q.k2 : q.N) >= 1 && q.k2 <= MAXK,
                "score_group: ref %d: k1,k2 must be in [1,%d]", i, MAXK);
    for (int s = 0; s < q.S; ++s) {
      const HglSentence& t = q.sentences[s];
      HGL_REQUIRE(t.imgattn && t.target, "score_group: ref %d sentence %d has no heat-map / target", i, s);
      HGL_REQUIRE(t.sentence_row >= 0 && t.sentence_row < q.T && t.noun_phrase_row >= 0 && t.noun_phrase_row < q.T,
                  "score_group: ref %d sentence %d: text row out of range", i, s);
      HGL_REQUIRE(t.n_other >= 0 && (t.n_other == 0 || (t.other_row0 >= 0 && t.other_row0 + t.n_other <= q.T)),
                  "score_group: ref %d sentence %d: other-noun rows out of range", i, s);
      HGL_REQUIRE(t.dirflag >= 0 && t.dirflag <= 3 && t.relaword >= 0 && t.relaword <= 7, "score_group: ref %d sentence %d: bad dirflag / relaword", i, s);
    }
  }
  if (!workspace || workspace_bytes < hgl_score_group_workspace_bytes(refs, R, E)) {
    hgl_set_error("score_group: workspace too small (%zu < %zu)", workspace_bytes, hgl_score_group_workspace_bytes(refs, R, E));
    return HGL_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)workspace;
  for (int r0 = 0; r0 < R; r0 += GRP_MAXR) {
    const int rc = R - r0 < GRP_MAXR ? R - r0 : GRP_MAXR;
    size_t ref_off[GRP_MAXR];
    grp_ws_layout(refs + r0, rc, E, ref_off);
    // the descriptor table, built in a pinned slot and copied in front of the workspace on the stream
    void* host_v = nullptr;
    int stage_dev = 0;
    const int slot = grp_stage_acquire(&stage_dev, &host_v);
    if (slot < 0) return HGL_ELAUNCH;
    GroupRefDev* host = (GroupRefDev*)host_v;
    int maxS = 0, max_nblk = 0, max_groups = 0;
    bool any_partial = false;      // a ref whose plane does not end on a block boundary: the pooling's second, small launch
    long long max_iou_blocks = 1;
    for (int i = 0; i < rc; ++i) {
      const HglGroupRef& q = refs[r0 + i];
      GroupRefDev& d = host[i];
      memset(&d, 0, sizeof(d));
      for (int j = 0; j < q.S; ++j) {
        const HglSentence& t = q.sentences[j];
        d.rs.attn[j] = t.imgattn; d.rs.target[j] = t.target;
        d.rs.sent_row[j] = t.sentence_row; d.rs.nphr_row[j] = t.noun_phrase_row;
        d.rs.other_row0[j] = t.n_other > 0 ? t.other_row0 : 0; d.rs.n_other[j] = t.n_other;
        d.rs.dirflag[j] = t.dirflag; d.rs.rela[j] = t.relaword; d.rs.has_other[j] = t.has_other_nouns; d.rs.black[j] = t.black;
      }
      d.S = q.S; d.N = q.N; d.H = q.H; d.W = q.W;
      d.k1 = q.k1 < q.N ? q.k1 : q.N;       // Hybridgl_main.py:178-181
      d.k2 = q.k2 < q.N ? q.k2 : q.N;
      d.nparts = coh_nparts(q.H, q.W);
      d.nblk = coh_nblk(q.H, q.W);
      d.hybrid = q.hybrid; d.text = q.text; d.boxes = (const long long*)q.boxes; d.masks = q.masks;
      size_t off[7];
      ref_ws_layout(q.S, q.N, E, q.H, q.W, off);
      char* rb = base + ref_off[i];
      d.part_mm = (float*)(rb + off[0]);
      d.part_sum = (double*)(rb + off[1]);
      d.part_cnt = (unsigned*)(rb + off[2]);
      d.part_tot = (double*)(rb + off[3]);
      d.soft = (float*)(rb + off[4]);
      float* spare = (float*)(rb + off[5]);
      d.gem = q.gem_score ? q.gem_score : spare;
      d.clip = q.score_clip ? q.score_clip : spare + (long long)q.S * q.N;
      d.neg = q.score_neg ? q.score_neg : spare + 2ll * q.S * q.N;
      d.idx = (int*)q.idx;
      d.iu = (unsigned long long*)q.iu;
      maxS = q.S > maxS ? q.S : maxS;
      max_nblk = d.nblk > max_nblk ? d.nblk : max_nblk;
      any_partial = any_partial || ((long long)q.H * q.W) % PIX_PER_BLOCK != 0;
      const int groups = (q.N + REF_MASK_GROUP - 1) / REF_MASK_GROUP;
      max_groups = groups > max_groups ? groups : max_groups;
      long long blocks = ((long long)q.H * q.W / 16 + 1023) / 1024;
      blocks = blocks < 1 ? 1 : (blocks > 32 ? 32 : blocks);
      max_iou_blocks = blocks > max_iou_blocks ? blocks : max_iou_blocks;
    }
    GroupRefDev* tab = (GroupRefDev*)base;
    unsigned* done = (unsigned*)(base + hgl_align_up((size_t)GRP_MAXR * sizeof(GroupRefDev), 256));
    const bool uploaded = hipMemcpyAsync(tab, host, (size_t)rc * sizeof(GroupRefDev), hipMemcpyHostToDevice, st) == hipSuccess;
    if (!grp_stage_release(stage_dev, slot, st) || !uploaded) {
      hgl_set_error("score_group: descriptor upload failed");
      return HGL_ELAUNCH;
    }
    hipLaunchKernelGGL(grp_minmax_kernel, dim3(REF_MM_BLOCKS, maxS, rc), dim3(256), 0, st, (const GroupRefDev*)tab);
    if (maxS <= 3)
      hipLaunchKernelGGL(grp_masked_pool_kernel<3>, dim3(max_nblk, max_groups, rc), dim3(256), 0, st, (const GroupRefDev*)tab);
    else
      hipLaunchKernelGGL(grp_masked_pool_kernel<4>, dim3(max_nblk, max_groups, rc), dim3(256), 0, st, (const GroupRefDev*)tab);
    if (any_partial)
      hipLaunchKernelGGL(grp_masked_pool_last_kernel, dim3(1, max_groups, rc), dim3(256), 0, st, (const GroupRefDev*)tab);
    hipLaunchKernelGGL(grp_score_kernel, dim3(maxS, rc), dim3(256), 0, st, (const GroupRefDev*)tab, E, logit_scale, r, alpha, done);
    hipLaunchKernelGGL(grp_iou_kernel, dim3((unsigned)max_iou_blocks, 2 * maxS, rc), dim3(256), 0, st, (const GroupRefDev*)tab, rc,
                       (unsigned long long*)cum, done);
  }
  return hgl_check_launch("score_group");
}

int hgl_gen_dir_mask(int dirflag, int H, int W, float* out, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(out && H > 0 && W > 0 && dirflag >= 0 && dirflag <= 3, "gen_dir_mask: bad arguments");
  hipLaunchKernelGGL(gen_dir_mask_kernel, dim3((unsigned)(((long long)H * W + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     dirflag, H, W, out);
  return hgl_check_launch("gen_dir_mask");
}

int hgl_relation_boxes(const int64_t* boxes_i, const int64_t* boxes_j, const float* score_i, const float* score_j, int n,
                       int relaword, float* out, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(boxes_i && boxes_j && score_i && score_j && out && n > 0, "relation_boxes: bad arguments");
  HGL_REQUIRE(relaword >= 0 && relaword <= 7, "relation_boxes: bad relaword %d", relaword);
  hipLaunchKernelGGL(relation_boxes_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const long long*)boxes_i,
                     (const long long*)boxes_j, score_i, score_j, n, relaword, out);
  return hgl_check_launch("relation_boxes");
}

size_t hgl_gaussian_blur_u8_workspace_bytes(int H, int W, int C) { return hgl_align_up((size_t)H * W * C * sizeof(double), 256); }

int hgl_gaussian_blur_u8(const uint8_t* img, int H, int W, int C, const double* taps, int k, uint8_t* out, void* workspace,
                         size_t workspace_bytes, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(img && out && taps && H > 0 && W > 0 && C > 0, "gaussian_blur_u8: bad arguments");
  HGL_REQUIRE(k >= 1 && k <= 31 && (k & 1) && k / 2 < H && k / 2 < W, "gaussian_blur_u8: kernel size %d unsupported for %dx%d", k, H, W);
  if (!workspace || workspace_bytes < hgl_gaussian_blur_u8_workspace_bytes(H, W, C)) {
    hgl_set_error("gaussian_blur_u8: workspace too small");
    return HGL_EWORKSPACE;
  }
  BlurTaps t;
  t.k = k;
  for (int i = 0; i < 31; ++i) t.w[i] = i < k ? taps[i] : 0.0;
  const long long total = (long long)H * W * C;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(blur_v_kernel, dim3(blocks), dim3(256), 0, st, img, H, W, C, t, (double*)workspace);
  hipLaunchKernelGGL(blur_h_kernel, dim3(blocks), dim3(256), 0, st, (const double*)workspace, H, W, C, t, out);
  return hgl_check_launch("gaussian_blur_u8");
}

int hgl_gaussian_blur_u8_q8(const uint8_t* img, int H, int W, int C, const uint16_t* taps_x, const uint16_t* taps_y, int k,
                            uint8_t* out, void* workspace, size_t workspace_bytes, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(img && out && taps_x && taps_y && H > 0 && W > 0 && C > 0, "gaussian_blur_u8_q8: bad arguments");
  HGL_REQUIRE(k >= 1 && k <= 31 && (k & 1) && k / 2 < H && k / 2 < W, "gaussian_blur_u8_q8: kernel size %d unsupported for %dx%d", k, H, W);
  if (!workspace || workspace_bytes < (size_t)H * W * C * sizeof(uint16_t)) {
    hgl_set_error("gaussian_blur_u8_q8: workspace too small");
    return HGL_EWORKSPACE;
  }
  BlurTapsQ8 t;
  t.k = k;
  unsigned sx = 0, sy = 0;
  for (int i = 0; i < 31; ++i) {
    t.x[i] = i < k ? taps_x[i] : 0;
    t.y[i] = i < k ? taps_y[i] : 0;
    sx += t.x[i];
    sy += t.y[i];
  }
  HGL_REQUIRE(sx == 256 && sy == 256, "gaussian_blur_u8_q8: the 8.8 fixed-point taps must sum to 256 (got %u, %u)", sx, sy);
  const long long total = (long long)H * W * C;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (C <= BLUR_CMAX) {      // (the workspace stays part of the signature: wider pixels take the two-launch form)
    const int ntiles = ((W + BLUR_TW - 1) / BLUR_TW) * ((H + BLUR_TH - 1) / BLUR_TH);
    hipLaunchKernelGGL(blur_q8_tile_kernel, dim3((unsigned)(8 * ((ntiles + 7) / 8))), dim3(BLUR_NT), 0, st, img, H, W, C, t, out);
    return hgl_check_launch("gaussian_blur_u8_q8");
  }
  hipLaunchKernelGGL(blur_q8_h_kernel, dim3(blocks), dim3(256), 0, st, img, H, W, C, t, (uint16_t*)workspace);
  hipLaunchKernelGGL(blur_q8_v_kernel, dim3(blocks), dim3(256), 0, st, (const uint16_t*)workspace, H, W, C, t, out);
  return hgl_check_launch("gaussian_blur_u8_q8");
}

int hgl_synthesize_views(const uint8_t* sam_img, const uint8_t* blurred, const float* image_norm,
                         const uint8_t* masks, int N, int H, int W, int res, float* local_imgs,
                         float* global_imgs, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(sam_img && blurred && image_norm && masks && local_imgs && global_imgs, "synthesize_views: null argument");
  HGL_REQUIRE(N > 0 && H > 0 && W > 0 && res > 0, "synthesize_views: bad shape");
  const long long total = (long long)N * res * res;
  hipLaunchKernelGGL(synth_views_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sam_img, blurred, image_norm, masks, N, H, W, res, local_imgs, global_imgs);
  return hgl_check_launch("synthesize_views");
}

}  // extern "C"
