// fp32-accurate GEMM on the fp16 matrix cores: C = act(A @ W^T + bias) + R with every fp32 operand
// split into two fp16 halves, x = hi + lo (hi = fp16(x), lo = fp16(x - hi)), and
//     a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi        (the dropped a_lo*b_lo term is 2^-22 relative)
// evaluated by three v_mfma_f32_32x32x16_f16 per tile step into ONE fp32 accumulator.  Each
// fp16 x fp16 product is exact in fp32 and the representation error of hi+lo is 2^-23, so the
// result carries ~2^-22 relative error per term -- within 4x of the fp32 MFMA path (gemm.hip) at
// 16/3 = 5.3x its matrix-core rate.  Weights are scaled by a power of two before splitting so
// that their lo halves stay in the fp16 normal range; the scale is undone in the epilogue.
//
// Same geometry as gemm.hip: 128x128 block tile, 4 waves x (2x2) 32x32 MFMA tiles, K-contiguous
// operands staged global -> registers -> LDS with 16-byte accesses.  An LDS row holds the hi and
// the lo halves of one operand row for BK = 64 (128 B + 128 B) plus one 16-byte pad, so the
// ds_read_b128 fragment reads are conflict free (row stride 272 B = 17 x 16 B).
#include "hgl_common.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <unordered_map>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // 16-byte staging register (HIP's uint4 struct defeats SROA)

namespace {

constexpr int BM = 128, BN = 128;
constexpr int NTHREADS = 256;

struct SplitW {
  const _Float16 *hi, *lo;
  int scale_log2;
  int N, K;
};
std::unordered_map<const void*, SplitW> g_split;   // fp32 weight pointer -> its fp16 split
int g_precision = HGL_PREC_F32;
enum { HGL_X3_V1 = 0, HGL_X3_L = 1, HGL_X3_M = 2, HGL_X3_S = 3, HGL_X3_N = 4, HGL_X3_Q = 5, HGL_X3_P = 6, HGL_X3_D = 7, HGL_X3_P16 = 8 };

struct Args {
  const _Float16 *Ah, *Al, *Wh, *Wl;
  const float *bias, *R;
  float* C;
  _Float16 *Ch, *Cl;   // split output (when C == nullptr)
  int M, N, K, lda, ldw, ldr, ldc;
  float out_scale;
  int tiles_m, tiles_n;
  int gm;      // M-tiles per tile group (L2 blocking of the resident tile set)
  float* part;   // split-K (LDS-DMA kernels, grid.y = ksplit): raw partial sums [ksplit][M][N]; bias / act / residual are
  int ksplit;    // applied by splitk_reduce_kernel, which sums the parts in a fixed order
  const int *amap, *cmap;   // optional row maps: A row m is read from row amap[m]; output / residual row m lives at cmap[m]
  int rmod;    // residual row = row % rmod when > 0 (a residual shared by every batch of rows), else row
  int dbg;     // timing experiments only (HGL_X3_DBG): bit 0 = skip the write-out
  int stg_mode, stg_ticks;   // start stagger of the persistent kernel: workgroup class and 10-ns ticks per class
};

int g_x3_kernel = -2;   // -2: read HGL_X3_KERNEL on first use; -1: cost model; >= 0: forced

// Cost model on 256 CUs, calibrated on MI355X with cold caches between launches (tools/x3_bench.py with
// X3_COLD=1, the regime of the pipeline; us): one round of tiles costs a + b * (K / 32), and a partial last round
// costs nearly a full one.  The 256x256 LDS-DMA tiling wins where there are many tiles and a wide N (the CLIP
// qkv / fc1 GEMMs, SAM's global-attention qkv); the 128x128 tilings at two workgroups per CU win on the rest (their
// epilogues overlap the other workgroup's K loop and they quantise better).  Between the two 128x128 kernels the
// register-staged one keeps two K tiles in flight per workgroup and tolerates HBM-latency weights slightly better,
// so it is the default there; a 128x160 LDS-DMA tiling takes the SAM shapes whose N it divides evenly (fewer, fuller
// rounds).  The LDS-DMA 256x128 / 128x128 / 160x160 tilings stay selectable (hgl_gemm_f16x3_select).
int pick_x3_kernel(int M, int N, int K) {
  struct Cfg { int kind, bm, bn, slots; double a, b, d; };
  static const Cfg cfgs[3] = {
      {HGL_X3_L, 256, 256, 256, 18.0, 2.30, 0.10},
      {HGL_X3_V1, 128, 128, 512, 13.8, 1.37, 0.25},
      {HGL_X3_N, 128, 160, 512, 10.4, 2.10, 0.25},   // 128x160 LDS-DMA: only where 160 divides N (1280 / 3840 / 5120)
  };
  const double nk = K / 32.0;
  int best = HGL_X3_V1;
  double best_t = 1e30;
  for (const Cfg& c : cfgs) {
    if (c.kind == HGL_X3_N && (N % 160) != 0) continue;
    const double tiles = (double)((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
    const double x = tiles / c.slots, up = ceil(x);
    // two-workgroup-per-CU tilings: a CU that holds a single workgroup finishes it in ~0.62 (register-staged) /
    // ~0.70 (4-wave LDS-DMA) of the pair's time
    const double rounds = (c.slots == 512 && x <= 0.5) ? (c.kind == HGL_X3_V1 ? 0.62 : 0.70) : up - c.d * (up - x);
    const double t = rounds * (c.a + c.b * nk);
    if (t < best_t) { best_t = t; best = c.kind; }
  }
  return best;
}

// grid of the persistent ping-pong kernel: one workgroup per CU (HGL_X3_PERSIST=0: one per tile)
long long x3p_grid(long long tiles) {
  static int ncu = 0, persist = -1;
  if (persist < 0) {
    const char* v = getenv("HGL_X3_PERSIST");
    persist = v ? atoi(v) : 1;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
  }
  return persist && tiles > ncu ? ncu : tiles;
}

template <int ACT>
__device__ __forceinline__ float act_apply(float x) {
  if constexpr (ACT == HGL_ACT_QUICKGELU) return x / (1.0f + __expf(-1.702f * x));
  if constexpr (ACT == HGL_ACT_GELU) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
  if constexpr (ACT == HGL_ACT_RELU) return x > 0.0f ? x : 0.0f;
  return x;
}

template <int ACT, int BK, int OCC>
__global__ __launch_bounds__(NTHREADS, OCC) void gemm_f16x3_kernel(Args g) {
  extern __shared__ __attribute__((aligned(16))) _Float16 smem[];  // [A: BM rows | W: BN rows] x ROW_H
  constexpr int ROW_H = 2 * BK + 8;   // halfs per LDS row: hi | lo | pad
  constexpr int CH = BK / 8;          // 16-byte chunks per row per half
  constexpr int RSTEP = NTHREADS / CH; // rows covered per pass
  constexpr int NLD = BM / RSTEP;      // loads per thread per array per K tile
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int GM = g.gm;
  const int group = bid / (GM * g.tiles_n);
  const int first_m = group * GM;
  const int gm = min(g.tiles_m - first_m, GM);
  const int rem = bid - group * GM * g.tiles_n;
  const int tile_m = first_m + rem % gm;
  const int tile_n = rem / gm;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int row0 = tile_m * BM, col0 = tile_n * BN;
  const int ld_c = t % CH;     // 16-byte chunk (8 halfs) within the K tile
  const int ld_row = t / CH;   // (+RSTEP*i)
  const int mclamp = g.M - 1, nclamp = g.N - 1;

  // Staging loads carry NO arithmetic on the loaded registers (anything that touches them would
  // make the compiler wait for the loads before the MFMA section): out-of-range rows read a
  // clamped (valid) row -- those accumulator rows/columns are never stored -- and K is a
  // multiple of BK (checked by the launcher), so there is no K tail.
  u32x4 pah[NLD], pal[NLD], pwh[NLD], pwl[NLD];
  const _Float16 *pa_h[NLD], *pa_l[NLD], *pw_h[NLD], *pw_l[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int arow = min(row0 + ld_row + RSTEP * i, mclamp);
    if (g.amap) arow = g.amap[arow];
    const long long oa = (long long)arow * g.lda + ld_c * 8;
    const long long ow = (long long)min(col0 + ld_row + RSTEP * i, nclamp) * g.ldw + ld_c * 8;
    pa_h[i] = g.Ah + oa; pa_l[i] = g.Al + oa; pw_h[i] = g.Wh + ow; pw_l[i] = g.Wl + ow;
  }
  auto load_tile = [&](int kt) {
    const int k = kt * BK;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      pah[i] = *(const u32x4*)(pa_h[i] + k); pal[i] = *(const u32x4*)(pa_l[i] + k);
      pwh[i] = *(const u32x4*)(pw_h[i] + k); pwl[i] = *(const u32x4*)(pw_l[i] + k);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      _Float16* ar = smem + (ld_row + RSTEP * i) * ROW_H + ld_c * 8;
      _Float16* wr = smem + (BM + ld_row + RSTEP * i) * ROW_H + ld_c * 8;
      *(u32x4*)ar = pah[i]; *(u32x4*)(ar + BK) = pal[i];
      *(u32x4*)wr = pwh[i]; *(u32x4*)(wr + BK) = pwl[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  const int nk = (g.K + BK - 1) / BK;
  load_tile(0);
  store_tile();
  __syncthreads();
  const _Float16* As = smem + (wm * 64 + r) * ROW_H + 8 * h;
  const _Float16* Ws = smem + (BM + wn * 64 + r) * ROW_H + 8 * h;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      const f16x8 ah0 = *(const f16x8*)(As + 16 * s), al0 = *(const f16x8*)(As + BK + 16 * s);
      const f16x8 ah1 = *(const f16x8*)(As + 32 * ROW_H + 16 * s), al1 = *(const f16x8*)(As + 32 * ROW_H + BK + 16 * s);
      const f16x8 bh0 = *(const f16x8*)(Ws + 16 * s), bl0 = *(const f16x8*)(Ws + BK + 16 * s);
      const f16x8 bh1 = *(const f16x8*)(Ws + 32 * ROW_H + 16 * s), bl1 = *(const f16x8*)(Ws + 32 * ROW_H + BK + 16 * s);
      // small cross terms first, then the hi*hi term
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh1, acc[1][1], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl1, acc[1][1], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
    if (kt + 1 < nk) store_tile();
    __syncthreads();
  }

  // ---- epilogue (same element map as gemm.hip) ----
  const bool full_tile = (row0 + BM <= g.M) && (col0 + BN <= g.N);
  // output rows of this lane (through the optional row map), fetched in one batch ahead of the dependent loads
  int crow[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rr = min(row0 + wm * 64 + i * 32 + 4 * h + (e & 3) + 8 * (e >> 2), mclamp);
      crow[i][e] = g.cmap ? g.cmap[rr] : rr;
    }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = col0 + wn * 64 + j * 32 + r;
    const bool cok = col < g.N;
    const int colc = cok ? col : nclamp;
    const float bv = g.bias ? g.bias[colc] : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rbase = row0 + wm * 64 + i * 32 + 4 * h;
      float rv[16];
      if (g.R) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          rv[e] = g.R[(long long)(g.rmod > 0 ? crow[i][e] % g.rmod : crow[i][e]) * g.ldr + colc];
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) rv[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = rbase + (e & 3) + 8 * (e >> 2);
        const float v = act_apply<ACT>(acc[i][j][e] * g.out_scale + bv) + rv[e];
        if (full_tile || (cok && row < g.M)) {
          const long long o = (long long)crow[i][e] * g.ldc + col;
          if (g.C) {
            g.C[o] = v;
          } else {
            _Float16 hi, lo;
            hgl_split_hi_lo(v, hi, lo);
            g.Ch[o] = hi;
            g.Cl[o] = lo;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Direct-to-LDS variant (global_load_lds_dwordx4): no VGPR staging and no ds_write pass -- on the
// register-staged kernel above the ds_write_b128 stream (13 cycles per wave-instruction) costs
// about half of the MFMA time of a K tile.  BK = 32; one stage holds four planes
// [A_hi | A_lo | W_hi | W_lo], each rows x 64 B with no padding (an LDS-DMA wave-instruction
// writes 1 KiB = 16 rows contiguously).  Bank conflicts are avoided by an XOR swizzle of the
// 16-byte chunk index with (row >> 2) & 3, applied to the per-lane GLOBAL source address on the way
// in and to the ds_read_b128 address on the way out.  Two stages: the DMA for tile t+1 is in flight
// while tile t is multiplied; one barrier per K tile.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

// One LDS-DMA wave-instruction: lane i copies 16 B from sbase + voff(i) to LDS byte lds_addr + 16*i.  Written as
// inline asm for the SGPR-base + 32-bit-VGPR-offset addressing form (the builtin keeps a 64-bit VGPR address
// per piece, which on the 256x256 tile pushes the K loop into scratch).  M0 is compiler-reserved: saved and
// restored around the instruction.  The compiler does not count these on vmcnt -- every consumer below waits
// with an explicit s_waitcnt vmcnt(0).
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_addr)
               : "memory");
}

template <int ACT, int TBM, int TBN, int WGM, int WGN, int OCC>
__global__ __launch_bounds__(WGM * WGN * 64, OCC) void gemm_x3g_kernel(Args g) {
  constexpr int NW = WGM * WGN;
  constexpr int WTM = TBM / WGM, WTN = TBN / WGN, MI = WTM / 32, NI = WTN / 32;
  // 1-KiB staging pieces (16 rows x 64 B of one plane).  When both operands' pieces divide evenly over the waves
  // each wave takes PA (A) + PW (W) row groups and their hi and lo planes share one per-lane offset; otherwise
  // (e.g. 128x160 on 4 waves) the flat list [A_hi | A_lo | W_hi | W_lo] is cut into NW equal runs.
  constexpr bool FLAT = (TBM / 16) % NW != 0 || (TBN / 16) % NW != 0;
  constexpr int PA = FLAT ? 0 : TBM / 16 / NW, PW = FLAT ? 0 : TBN / 16 / NW;
  constexpr int NPT = 2 * (TBM + TBN) / 16;               // pieces per stage
  constexpr int NPIECE = NPT / NW;                        // per wave
  constexpr int A_BYTES = TBM * 64, W_BYTES = TBN * 64;    // one plane of one stage
  constexpr int STAGE = 2 * (A_BYTES + W_BYTES);
  static_assert(NPT % NW == 0 && TBM % (32 * WGM) == 0 && TBN % (32 * WGN) == 0, "tile/wave shape");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_g[];

  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  }
  const int GM = g.gm;
  const int group = bid / (GM * g.tiles_n);
  const int first_m = group * GM;
  const int gmn = min(g.tiles_m - first_m, GM);
  const int rem = bid - group * GM * g.tiles_n;
  const int tile_m = first_m + rem % gmn;
  const int tile_n = rem / gmn;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave / WGN, wn = wave % WGN;
  const int row0 = tile_m * TBM, col0 = tile_n * TBN;
  const int mclamp = g.M - 1, nclamp = g.N - 1;
  // K range of this workgroup: everything, or the blockIdx.y-th of ksplit even-length slices (split-K)
  int kbeg = 0, nk = g.K / 32;   // K tiles of 32; even and >= 2 (K % 64 == 0, checked by the launcher)
  if (g.ksplit > 1) {
    const int chunk = (nk / g.ksplit) & ~1;
    kbeg = (int)blockIdx.y * chunk;
    nk = (int)blockIdx.y == g.ksplit - 1 ? nk - kbeg : chunk;
  }

  // staging: lane -> (row within the 16-row piece, swizzled source chunk).  Per-lane state is one 32-bit byte
  // offset per piece row; the plane bases are wave-uniform (SGPR) and advance by 64 B per K tile.
  const int prow = lane >> 2, pchunk = (lane & 3) ^ ((lane >> 4) & 3);
  const unsigned char *bAh = (const unsigned char*)g.Ah, *bAl = (const unsigned char*)g.Al;
  const unsigned char *bWh = (const unsigned char*)g.Wh, *bWl = (const unsigned char*)g.Wl;
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem_g;
  unsigned oa[FLAT ? 1 : PA], ow[FLAT ? 1 : PW];        // paired mode: per-lane byte offsets of the row groups
  unsigned pvoff[FLAT ? NPIECE : 1], plds[FLAT ? NPIECE : 1];   // flat mode: per piece offset (VGPR), LDS offset (SGPR)
  const unsigned char* pbase[FLAT ? NPIECE : 1];         //            and plane base (SGPR)
  if constexpr (!FLAT) {
#pragma unroll
    for (int j = 0; j < PA; ++j)
    {
      int arow = min(row0 + (wave * PA + j) * 16 + prow, mclamp);
      if (g.amap) arow = g.amap[arow];
      oa[j] = (unsigned)(arow * g.lda + pchunk * 8) * 2u;
    }
#pragma unroll
    for (int j = 0; j < PW; ++j)
      ow[j] = (unsigned)(min(col0 + (wave * PW + j) * 16 + prow, nclamp) * g.ldw + pchunk * 8) * 2u;
  } else {
    constexpr int PAh = TBM / 16, PWh = TBN / 16;
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const int gidx = wave * NPIECE + i;                  // wave-uniform
      if (gidx < 2 * PAh) {
        const int lo = gidx >= PAh, rp = gidx - lo * PAh;
        pbase[i] = lo ? bAl : bAh;
        plds[i] = lo * A_BYTES + rp * 1024;
        int arow = min(row0 + rp * 16 + prow, mclamp);
        if (g.amap) arow = g.amap[arow];
        pvoff[i] = (unsigned)(arow * g.lda + pchunk * 8) * 2u;
      } else {
        const int g2 = gidx - 2 * PAh;
        const int lo = g2 >= PWh, rp = g2 - lo * PWh;
        pbase[i] = lo ? bWl : bWh;
        plds[i] = 2 * A_BYTES + lo * W_BYTES + rp * 1024;
        pvoff[i] = (unsigned)(min(col0 + rp * 16 + prow, nclamp) * g.ldw + pchunk * 8) * 2u;
      }
    }
  }

  // one 1-KiB piece q (0 .. NPIECE-1) of this wave's share of K tile kt; the pieces are spread between the
  // MFMAs of a k-step so that their issue cost hides under the other waves' MFMAs
  auto issue_piece = [&](int kt, int stage, int q) {
    const unsigned sb = lds0 + stage * STAGE;
    const long long ko = (long long)(kbeg + kt) * 64;
    if constexpr (FLAT) {
      glds16(pbase[q] + ko, pvoff[q], sb + plds[q]);
    } else if (q < 2 * PA) {
      const int j = q >> 1, lo = q & 1;
      glds16((lo ? bAl : bAh) + ko, oa[j], sb + lo * A_BYTES + (wave * PA + j) * 1024);
    } else {
      const int j = (q - 2 * PA) >> 1, lo = q & 1;
      glds16((lo ? bWl : bWh) + ko, ow[j], sb + 2 * A_BYTES + lo * W_BYTES + (wave * PW + j) * 1024);
    }
  };
  auto issue = [&](int kt, int stage) {
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) issue_piece(kt, stage, q);
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  const int sw = (r >> 2) & 3;
  const int co0 = ((0 + h) ^ sw) * 16, co1 = ((2 + h) ^ sw) * 16;
  const unsigned char* Ab = smem_g + (wm * WTM + r) * 64;
  const unsigned char* Wb = smem_g + 2 * A_BYTES + (wn * WTN + r) * 64;
  // Register-double-buffered schedule.  F0 / F1 hold the fragments of k-step 0 / 1 of a stage; the reads of
  // (t, 0) are issued right after the barrier that publishes stage t and are covered by the MFMAs of (t-1, 1);
  // the reads of (t, 1) are covered by the MFMAs of (t, 0).  The DMA for stage t+1 is issued between the MFMAs
  // of (t-1, 1) -- after the barrier, so every wave has finished reading that buffer -- and has a whole K tile
  // of MFMAs to land.
  struct Frag { f16x8 ah[MI], al[MI], bh[NI], bl[NI]; };
  auto read_frag = [&](Frag& f, int stage, int s) {
    const int co = stage * STAGE + (s ? co1 : co0);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      f.bh[j] = *(const f16x8*)(Wb + j * 2048 + co);
      f.bl[j] = *(const f16x8*)(Wb + W_BYTES + j * 2048 + co);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      f.ah[i] = *(const f16x8*)(Ab + i * 2048 + co);
      f.al[i] = *(const f16x8*)(Ab + A_BYTES + i * 2048 + co);
    }
  };
  // the 3*MI*NI MFMAs of one k-step (small cross terms first, then hi*hi); with ISSUE also the DMA pieces of
  // K tile nxt into buffer nstage, spread between the MFMAs
  auto mfma_step = [&](const Frag& f, int nxt, int nstage, auto issue_tag) {
    constexpr bool ISSUE = decltype(issue_tag)::value;
    constexpr int NMF = 3 * MI * NI;
    constexpr int EVERY = NMF / NPIECE > 0 ? NMF / NPIECE : 1;
    int q = 0, n = 0;
#pragma unroll
    for (int term = 0; term < 3; ++term)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const f16x8 a = term == 0 ? f.al[i] : f.ah[i];
          const f16x8 b = term == 1 ? f.bl[j] : f.bh[j];
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i][j], 0, 0, 0);
          ++n;
          if (ISSUE && n % EVERY == 0 && q < NPIECE) {
            __builtin_amdgcn_sched_barrier(0);
            issue_piece(nxt, nstage, q++);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    if (ISSUE)
      for (; q < NPIECE; ++q) issue_piece(nxt, nstage, q);
  };
  const std::true_type yes;
  const std::false_type no;
  Frag F0, F1;
  // stage landed (own DMA pieces) + own LDS reads retired, then the workgroup barrier
  auto publish = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto iter = [&](int stage, int t, auto issue_tag) {
    publish();
    read_frag(F0, stage, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(F1, t + 1, stage ^ 1, issue_tag);
    __builtin_amdgcn_sched_barrier(0);
    read_frag(F1, stage, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(F0, 0, 0, no);
  };

  issue(0, 0);
  publish();
  read_frag(F0, 0, 0);
  issue(1, 1);
  read_frag(F1, 0, 1);
  __builtin_amdgcn_sched_barrier(0);
  mfma_step(F0, 0, 0, no);
  for (int t = 1; t + 1 < nk; t += 2) {
    iter(1, t, yes);
    iter(0, t + 1, yes);
  }
  iter(1, nk - 1, no);
  mfma_step(F1, 0, 0, no);

  // ---- epilogue ----
  const bool full_tile = (row0 + TBM <= g.M) && (col0 + TBN <= g.N);
  if (g.ksplit > 1) {   // raw partial sums; splitk_reduce_kernel finishes
    float* pp = g.part + (long long)blockIdx.y * g.M * g.N;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int col = col0 + wn * WTN + j * 32 + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row0 + wm * WTM + i * 32 + 4 * h + (e & 3) + 8 * (e >> 2);
          if (full_tile || (col < g.N && row < g.M)) pp[(long long)row * g.N + col] = acc[i][j][e] * g.out_scale;
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int rbase = row0 + wm * WTM + i * 32 + 4 * h;
    int crow[16];   // output rows of this lane (through the optional row map), fetched in one batch
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rr = min(rbase + (e & 3) + 8 * (e >> 2), mclamp);
      crow[e] = g.cmap ? g.cmap[rr] : rr;
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = col0 + wn * WTN + j * 32 + r;
      const bool cok = col < g.N;
      const int colc = cok ? col : nclamp;
      const float bv = g.bias ? g.bias[colc] : 0.0f;
      float rv[16];
      if (g.R) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          rv[e] = g.R[(long long)(g.rmod > 0 ? crow[e] % g.rmod : crow[e]) * g.ldr + colc];
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) rv[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = rbase + (e & 3) + 8 * (e >> 2);
        const float v = act_apply<ACT>(acc[i][j][e] * g.out_scale + bv) + rv[e];
        if (full_tile || (cok && row < g.M)) {
          const long long o = (long long)crow[e] * g.ldc + col;
          if (g.C) {
            g.C[o] = v;
          } else {
            _Float16 hi, lo;
            hgl_split_hi_lo(v, hi, lo);
            g.Ch[o] = hi;
            g.Cl[o] = lo;
          }
        }
      }
    }
  }
}

// Shared write-out of a wave's MI x NI accumulator tiles (32x32 each): bias / activation / residual / optional
// row maps / split (fp16 hi+lo) output, or the raw partial sums of a split-K slice.
template <int ACT, int MI, int NI>
__device__ __forceinline__ void x3_epilogue(const Args& g, f32x16 (&acc)[MI][NI], int row0, int col0, int wrow, int wcol,
                                            int r, int h, int TBM, int TBN) {
  const int mclamp = g.M - 1, nclamp = g.N - 1;
  const bool full_tile = (row0 + TBM <= g.M) && (col0 + TBN <= g.N);
  if (g.ksplit > 1) {   // raw partial sums; splitk_reduce_kernel finishes
    float* pp = g.part + (long long)blockIdx.y * g.M * g.N;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int col = col0 + wcol + j * 32 + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row0 + wrow + i * 32 + 4 * h + (e & 3) + 8 * (e >> 2);
          if (full_tile || (col < g.N && row < g.M)) pp[(long long)row * g.N + col] = acc[i][j][e] * g.out_scale;
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int rbase = row0 + wrow + i * 32 + 4 * h;
    int crow[16];   // output rows of this lane (through the optional row map), fetched in one batch
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rr = min(rbase + (e & 3) + 8 * (e >> 2), mclamp);
      crow[e] = g.cmap ? g.cmap[rr] : rr;
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = col0 + wcol + j * 32 + r;
      const bool cok = col < g.N;
      const int colc = cok ? col : nclamp;
      const float bv = g.bias ? g.bias[colc] : 0.0f;
      float rv[16];
      if (g.R) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          rv[e] = g.R[(long long)(g.rmod > 0 ? crow[e] % g.rmod : crow[e]) * g.ldr + colc];
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) rv[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = rbase + (e & 3) + 8 * (e >> 2);
        const float v = act_apply<ACT>(acc[i][j][e] * g.out_scale + bv) + rv[e];
        if (full_tile || (cok && row < g.M)) {
          const long long o = (long long)crow[e] * g.ldc + col;
          if (g.C) {
            g.C[o] = v;
          } else {
            _Float16 hi, lo;
            hgl_split_hi_lo(v, hi, lo);
            g.Ch[o] = hi;
            g.Cl[o] = lo;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Ping-pong variant of the 256x256 LDS-DMA tiling (kind P).  Same tile, same LDS image (planes [A_hi|A_lo|W_hi|W_lo]
// of 256 rows x 64 B per stage, XOR-swizzled chunks, two stages) and the same per-accumulator order of the MFMAs as
// gemm_x3g_kernel<256,256,2,4> -- so the outputs are bit-identical -- but a different schedule.  In the kernel
// above the eight waves run the same program in the same phase: the two waves of a SIMD want the matrix pipe
// together and reach the one barrier per K tile together, where the pipe drains (53-56 % MFMA-busy measured).
// Here the waves form two groups of four (waves 0-3 = the upper 128 rows, waves 4-7 = the lower 128 rows: one wave
// of each group per SIMD) that run ONE BARRIER APART: a K tile is cut into four phases (a 64 x 32 quadrant of the
// wave's 128 x 64 tile = 12 MFMAs = 384 matrix-pipe cycles each); a phase is a load segment (the quadrant's
// ds_read_b128 fragment reads + two LDS-DMA pieces + the counted waits), a barrier, an MFMA-only segment under
// s_setprio 1, a barrier.  Because group 1 executed one extra barrier at the start, every barrier interval has one
// group issuing nothing but MFMAs while the other group's loads, DMA issue and waits run beside it on the same SIMDs.
//
// Staging is in UNITS of 16 KiB (16 pieces, two per wave) in the order the phases need them: U1 = A rows {0-63} of
// each group's half, U2 = W columns {0-31} of each wave column's 64, U3 = W columns {32-63}, U4 = A rows {64-127};
// phase 1 reads U1 + U2, phase 2 U3, phase 3 U4, phase 4 nothing.  A unit's LDS slot is free again one barrier after
// its reading phase (the fragment reads are retired by lgkmcnt(0) BEFORE the load segment's barrier), so the DMA runs
// seven units ahead of the reads: the load segment of (tile t, phase 1) issues U4(t+1), phases 2-4 issue U1-U3(t+2).
// Each load segment ends with s_waitcnt vmcnt(10): all but the wave's five youngest units have landed, i.e. every
// unit read by ANY wave in the next segment -- published by the barrier that follows.  A unit has five phases
// (about 3800 matrix-pipe cycles) to land against one K tile (3072) in the kernel above.  The last two K tiles use
// the exact smaller counts (8, 6, 4, 2, 0).
template <int N>
__device__ __forceinline__ void x3p_wait() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

// The accumulators are kept TRANSPOSED (the W fragment is the MFMA's first operand): lane (r, h) of acc[i][j] holds
// output row i*32 + r and the columns j*32 + 8*q + 4*h + {0..3} (q = e >> 2) -- four consecutive columns per
// quad of registers, so bias / residual / output move as 16-byte vectors (4 stores per 32x32 tile instead of 16).
// The products and their order per accumulator are unchanged (a*b = b*a, same k order): bit-identical results.
// Requires N, ldc, ldr multiples of 4 and 16-byte aligned bases (checked by the launcher, which otherwise takes the L tiling).
template <int ACT>
__global__ __launch_bounds__(512, 1) void gemm_x3p_kernel(Args g) {
  constexpr int TBM = 256, TBN = 256, MI = 4, NI = 2, WTM = 128, WTN = 64;
  constexpr int PLANE = 256 * 64;    // bytes of one plane of one stage
  constexpr int STAGE = 4 * PLANE;   // [A_hi | A_lo | W_hi | W_lo]
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_p[];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;   // wm = the phase group
  const int mclamp = g.M - 1, nclamp = g.N - 1;
  int kbeg = 0, nk = g.K / 32;   // K tiles of 32; even and >= 2 (K % 64 == 0, checked by the launcher)
  if (g.ksplit > 1) {
    const int chunk = (nk / g.ksplit) & ~1;
    kbeg = (int)blockIdx.y * chunk;
    nk = (int)blockIdx.y == g.ksplit - 1 ? nk - kbeg : chunk;
  }

  // Tile schedule.  The tile list (group of GM row tiles x all column tiles, row-fastest) is cut into eight contiguous
  // chunks, one per XCD (blockIdx.x & 7 names the workgroups that share an XCD); the workgroups of an XCD walk their
  // chunk with a stride of their count, so the tiles in flight on one L2 are neighbours.  gridDim.x == number of tiles
  // gives one tile per workgroup (the mapping of the other tilings); a smaller grid makes the kernel persistent.
  const int ntiles = g.tiles_m * g.tiles_n;
  const int xcd = blockIdx.x & 7;
  int vstart, vlen;
  {
    const int q = ntiles >> 3, rr = ntiles & 7;
    vstart = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
    vlen = q + (xcd < rr ? 1 : 0);
  }
  const int vstride = ((int)gridDim.x - xcd + 7) >> 3;
  int v = blockIdx.x >> 3;
  if (v >= vlen) return;

  // staging: this wave's two 16-row groups of A (rows 0-63 / 64-127 of its group's half) and of W
  const int prow = lane >> 2, pchunk = (lane & 3) ^ ((lane >> 4) & 3);
  const unsigned char *bAh = (const unsigned char*)g.Ah, *bAl = (const unsigned char*)g.Al;
  const unsigned char *bWh = (const unsigned char*)g.Wh, *bWl = (const unsigned char*)g.Wl;
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem_p;
  const int rgA0 = wm * 8 + (wave & 3), rgA1 = rgA0 + 4;
  const int cgW0 = 4 * (wave >> 1) + (wave & 1), cgW1 = cgW0 + 2;
  struct TileOff { int row0, col0; unsigned oa0, oa1, ow0, ow1; };
  auto tile_offsets = [&](int vid) {
    const int bid = vstart + vid;
    const int GM = g.gm;
    const int group = bid / (GM * g.tiles_n);
    const int first_m = group * GM;
    const int gmn = min(g.tiles_m - first_m, GM);
    const int rem = bid - group * GM * g.tiles_n;
    TileOff o;
    o.row0 = (first_m + rem % gmn) * TBM;
    o.col0 = (rem / gmn) * TBN;
    int a0 = min(o.row0 + rgA0 * 16 + prow, mclamp), a1 = min(o.row0 + rgA1 * 16 + prow, mclamp);
    if (g.amap) { a0 = g.amap[a0]; a1 = g.amap[a1]; }
    o.oa0 = (unsigned)(a0 * g.lda + pchunk * 8) * 2u;
    o.oa1 = (unsigned)(a1 * g.lda + pchunk * 8) * 2u;
    o.ow0 = (unsigned)(min(o.col0 + cgW0 * 16 + prow, nclamp) * g.ldw + pchunk * 8) * 2u;
    o.ow1 = (unsigned)(min(o.col0 + cgW1 * 16 + prow, nclamp) * g.ldw + pchunk * 8) * 2u;
    return o;
  };
  TileOff cur = tile_offsets(v);
  if (g.stg_ticks > 0) {   // spread the workgroups' phases so that their write-out bursts do not coincide
    const int j = blockIdx.x >> 3;
    const int cls = g.stg_mode == 1 ? j : g.stg_mode == 2 ? xcd : g.stg_mode == 3 ? (j & 3) : g.stg_mode == 4 ? (j >> 3) : (j & 7);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long want = (unsigned long long)cls * g.stg_ticks;
    while (__builtin_amdgcn_s_memrealtime() - t0 < want) __builtin_amdgcn_s_sleep(8);
  }
  // unit u (0: A first rows, 1: W first columns, 2: W second columns, 3: A second rows) of K tile `tile`
  auto issue_unit = [&](const TileOff& o, int u, int tile) {
    const unsigned sb = lds0 + (tile & 1) * STAGE;
    const long long ko = (long long)(kbeg + tile) * 64;
    if (u == 0) {
      glds16(bAh + ko, o.oa0, sb + rgA0 * 1024);
      glds16(bAl + ko, o.oa0, sb + PLANE + rgA0 * 1024);
    } else if (u == 1) {
      glds16(bWh + ko, o.ow0, sb + 2 * PLANE + cgW0 * 1024);
      glds16(bWl + ko, o.ow0, sb + 3 * PLANE + cgW0 * 1024);
    } else if (u == 2) {
      glds16(bWh + ko, o.ow1, sb + 2 * PLANE + cgW1 * 1024);
      glds16(bWl + ko, o.ow1, sb + 3 * PLANE + cgW1 * 1024);
    } else {
      glds16(bAh + ko, o.oa1, sb + rgA1 * 1024);
      glds16(bAl + ko, o.oa1, sb + PLANE + rgA1 * 1024);
    }
  };
  auto issue_prologue = [&](const TileOff& o) {   // seven units ahead
    issue_unit(o, 0, 0); issue_unit(o, 1, 0); issue_unit(o, 2, 0); issue_unit(o, 3, 0);
    issue_unit(o, 0, 1); issue_unit(o, 1, 1); issue_unit(o, 2, 1);
  };

  const int sw = (r >> 2) & 3;
  const int co0 = ((0 + h) ^ sw) * 16, co1 = ((2 + h) ^ sw) * 16;
  const unsigned char* Ab = smem_p + (wm * WTM + r) * 64;
  const unsigned char* Wb = smem_p + 2 * PLANE + (wn * WTN + r) * 64;
  f32x16 acc[MI][NI];
  f16x8 ah[2][2], al[2][2];      // [row block of the half][k-step]
  f16x8 bh[2][2], bl[2][2];      // [column block][k-step]
  auto read_A = [&](int ih, int stage) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const unsigned char* p = Ab + stage * STAGE + (2 * ih + ii) * 2048;
      ah[ii][0] = *(const f16x8*)(p + co0); ah[ii][1] = *(const f16x8*)(p + co1);
      al[ii][0] = *(const f16x8*)(p + PLANE + co0); al[ii][1] = *(const f16x8*)(p + PLANE + co1);
    }
  };
  auto read_B = [&](int j, int stage) {
    const unsigned char* p = Wb + stage * STAGE + j * 2048;
    bh[j][0] = *(const f16x8*)(p + co0); bh[j][1] = *(const f16x8*)(p + co1);
    bl[j][0] = *(const f16x8*)(p + PLANE + co0); bl[j][1] = *(const f16x8*)(p + PLANE + co1);
  };
  // the 12 MFMAs of quadrant (ih, j): per accumulator k-step 0 then 1, each lo*hi, hi*lo, hi*hi (the order of
  // every other f16x3 kernel); W fragment first = transposed accumulator
  auto quad = [&](int ih, int j) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const f16x8 a = term == 0 ? al[ii][s] : ah[ii][s];
          const f16x8 b = term == 1 ? bl[j][s] : bh[j][s];
          acc[2 * ih + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[2 * ih + ii][j], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // one K tile: C1..C4 = the vmcnt left outstanding at the end of the four load segments; I1 / I2: issue the unit
  // of phase 1 (tile + 1) / of phases 2-4 (tile + 2); LAST drops group 1's unpaired final barrier
  auto tile_body = [&](int tile, auto c1, auto c2, auto c3, auto c4, auto i1, auto i2, auto last) {
    constexpr int C1 = decltype(c1)::value, C2 = decltype(c2)::value, C3 = decltype(c3)::value, C4 = decltype(c4)::value;
    constexpr bool I1 = decltype(i1)::value, I2 = decltype(i2)::value, LAST = decltype(last)::value;
    const int stage = tile & 1;
    // phase 1
    read_A(0, stage);
    read_B(0, stage);
    if (I1) issue_unit(cur, 3, tile + 1);
    __builtin_amdgcn_sched_barrier(0);
    x3p_wait<C1>();
    bar();
    quad(0, 0);
    bar();
    // phase 2
    read_B(1, stage);
    if (I2) issue_unit(cur, 0, tile + 2);
    __builtin_amdgcn_sched_barrier(0);
    x3p_wait<C2>();
    bar();
    quad(0, 1);
    bar();
    // phase 3
    read_A(1, stage);
    if (I2) issue_unit(cur, 1, tile + 2);
    __builtin_amdgcn_sched_barrier(0);
    x3p_wait<C3>();
    bar();
    quad(1, 0);
    bar();
    // phase 4
    if (I2) issue_unit(cur, 2, tile + 2);
    __builtin_amdgcn_sched_barrier(0);
    x3p_wait<C4>();
    bar();
    quad(1, 1);
    if (!(LAST && wm == 1)) bar();
  };
  using I0 = std::integral_constant<int, 0>;
  using I2_ = std::integral_constant<int, 2>;
  using I4 = std::integral_constant<int, 4>;
  using I6 = std::integral_constant<int, 6>;
  using I8 = std::integral_constant<int, 8>;
  using I10 = std::integral_constant<int, 10>;
  const std::true_type yes;
  const std::false_type no;

  issue_prologue(cur);
  for (;;) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    // Every wait below is correct whatever else the wave still has in flight (the stores of the previous tile's
    // write-out are YOUNGER than this tile's prologue pieces, so they can only make a counted wait stricter).
    x3p_wait<10>();
    bar();
    if (wm == 1) bar();   // group 1 runs one barrier behind group 0
    for (int tile = 0; tile + 2 < nk; ++tile) tile_body(tile, I10(), I10(), I10(), I10(), yes, yes, no);
    tile_body(nk - 2, I10(), I8(), I6(), I4(), yes, no, no);
    tile_body(nk - 1, I2_(), I0(), I0(), I0(), no, no, yes);
    // Both groups have retired every LDS read of this tile when either leaves its last barrier (group 1's phase 4
    // has none, group 0 finishes after group 1's last read segment): the next tile's prologue DMA goes out BEFORE
    // the write-out, which it overlaps.
    const TileOff done = cur;
    v += vstride;
    const bool more = v < vlen;
    if (more) {
      cur = tile_offsets(v);
      issue_prologue(cur);
    }

    // ---- write-out (transposed accumulators: 16-byte vectors along the output row) ----
    if ((g.dbg & 1) && acc[0][0][0] != 12345.678f) {
    } else {
      const int row0 = done.row0, col0 = done.col0;
      const bool full_tile = (row0 + TBM <= g.M) && (col0 + TBN <= g.N);
      const int cb = col0 + wn * WTN + 4 * h;   // + j*32 + 8*q
      if (g.ksplit > 1) {   // raw partial sums; splitk_reduce_kernel finishes
        float* pp = g.part + (long long)blockIdx.y * g.M * g.N;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int row = row0 + wm * WTM + i * 32 + r;
#pragma unroll
          for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int col = cb + j * 32 + 8 * q;
              if (full_tile || (row < g.M && col < g.N)) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[i][j][4 * q + e] * g.out_scale;
                *(f32x4*)(pp + (long long)row * g.N + col) = o;
              }
            }
        }
      } else {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          f32x4 bv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int colc = min(cb + j * 32 + 8 * q, g.N - 4);
            bv[q] = g.bias ? *(const f32x4*)(g.bias + colc) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int row = row0 + wm * WTM + i * 32 + r;
            const int rc = min(row, mclamp);
            const int crow = g.cmap ? g.cmap[rc] : rc;
            const long long rrow = g.rmod > 0 ? crow % g.rmod : crow;
            f32x4 rv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int colc = min(cb + j * 32 + 8 * q, g.N - 4);
              rv[q] = g.R ? *(const f32x4*)(g.R + rrow * g.ldr + colc) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int col = cb + j * 32 + 8 * q;
              if (full_tile || (row < g.M && col < g.N)) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = act_apply<ACT>(acc[i][j][4 * q + e] * g.out_scale + bv[q][e]) + rv[q][e];
                const long long off = (long long)crow * g.ldc + col;
                if (g.C) {
                  *(f32x4*)(g.C + off) = o;
                } else {
                  f16x4 hi4, lo4;
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    _Float16 hh, ll;
                    hgl_split_hi_lo(o[e], hh, ll);
                    hi4[e] = hh;
                    lo4[e] = ll;
                  }
                  *(f16x4*)(g.Ch + off) = hi4;
                  *(f16x4*)(g.Cl + off) = lo4;
                }
              }
            }
          }
        }
      }
    }
    if (!more) break;
  }
}

// The same kernel on v_mfma_f32_16x16x32_f16 (kind P16): one MFMA per term covers the whole K tile of 32; same LDS
// image with the swizzle of the 16-row fragment reads, same staging order, same counts.  The K sum inside an
// instruction is ordered differently, so the results differ from the 32x32x16 tilings in the last bits.
template <int ACT>
__global__ __launch_bounds__(512, 1) void gemm_x3p16_kernel(Args g) {
  constexpr int TBM = 256, TBN = 256, MB = 8, NB = 4, WTM = 128, WTN = 64;   // 16x16 accumulator blocks per wave
  constexpr int PLANE = 256 * 64;    // bytes of one plane of one stage
  constexpr int STAGE = 4 * PLANE;   // [A_hi | A_lo | W_hi | W_lo]
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_p[];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 15, h = lane >> 4;   // fragment row / 8-wide k group; accumulator: row r, columns 4*h..4*h+3
  const int wm = wave >> 2, wn = wave & 3;   // wm = the phase group
  const int mclamp = g.M - 1, nclamp = g.N - 1;
  int kbeg = 0, nk = g.K / 32;   // K tiles of 32; even and >= 2 (K % 64 == 0, checked by the launcher)
  if (g.ksplit > 1) {
    const int chunk = (nk / g.ksplit) & ~1;
    kbeg = (int)blockIdx.y * chunk;
    nk = (int)blockIdx.y == g.ksplit - 1 ? nk - kbeg : chunk;
  }

  // Tile schedule.  The tile list (group of GM row tiles x all column tiles, row-fastest) is cut into eight contiguous
  // chunks, one per XCD (blockIdx.x & 7 names the workgroups that share an XCD); the workgroups of an XCD walk their
  // chunk with a stride of their count, so the tiles in flight on one L2 are neighbours.  gridDim.x == number of tiles
  // gives one tile per workgroup (the mapping of the other tilings); a smaller grid makes the kernel persistent.
  const int ntiles = g.tiles_m * g.tiles_n;
  const int xcd = blockIdx.x & 7;
  int vstart, vlen;
  {
    const int q = ntiles >> 3, rr = ntiles & 7;
    vstart = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
    vlen = q + (xcd < rr ? 1 : 0);
  }
  const int vstride = ((int)gridDim.x - xcd + 7) >> 3;
  int v = blockIdx.x >> 3;
  if (v >= vlen) return;

  // staging: this wave's two 16-row groups of A (rows 0-63 / 64-127 of its group's half) and of W
  const int prow = lane >> 2, pchunk = (lane & 3) ^ ((0x78 >> (2 * ((lane >> 4) & 3))) & 3);   // swizzle {0,2,3,1}[(row >> 2) & 3]
  const unsigned char *bAh = (const unsigned char*)g.Ah, *bAl = (const unsigned char*)g.Al;
  const unsigned char *bWh = (const unsigned char*)g.Wh, *bWl = (const unsigned char*)g.Wl;
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem_p;
  const int rgA0 = wm * 8 + (wave & 3), rgA1 = rgA0 + 4;
  const int cgW0 = 4 * (wave >> 1) + (wave & 1), cgW1 = cgW0 + 2;
  struct TileOff { int row0, col0; unsigned oa0, oa1, ow0, ow1; };
  auto tile_offsets = [&](int vid) {
    const int bid = vstart + vid;
    const int GM = g.gm;
    const int group = bid / (GM * g.tiles_n);
    const int first_m = group * GM;
    const int gmn = min(g.tiles_m - first_m, GM);
    const int rem = bid - group * GM * g.tiles_n;
    TileOff o;
    o.row0 = (first_m + rem % gmn) * TBM;
    o.col0 = (rem / gmn) * TBN;
    int a0 = min(o.row0 + rgA0 * 16 + prow, mclamp), a1 = min(o.row0 + rgA1 * 16 + prow, mclamp);
    if (g.amap) { a0 = g.amap[a0]; a1 = g.amap[a1]; }
    o.oa0 = (unsigned)(a0 * g.lda + pchunk * 8) * 2u;
    o.oa1 = (unsigned)(a1 * g.lda + pchunk * 8) * 2u;
    o.ow0 = (unsigned)(min(o.col0 + cgW0 * 16 + prow, nclamp) * g.ldw + pchunk * 8) * 2u;
    o.ow1 = (unsigned)(min(o.col0 + cgW1 * 16 + prow, nclamp) * g.ldw + pchunk * 8) * 2u;
    return o;
  };
  TileOff cur = tile_offsets(v);
  if (g.stg_ticks > 0) {   // spread the workgroups' phases so that their write-out bursts do not coincide
    const int j = blockIdx.x >> 3;
    const int cls = g.stg_mode == 1 ? j : g.stg_mode == 2 ? xcd : g.stg_mode == 3 ? (j & 3) : g.stg_mode == 4 ? (j >> 3) : (j & 7);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long want = (unsigned long long)cls * g.stg_ticks;
    while (__builtin_amdgcn_s_memrealtime() - t0 < want) __builtin_amdgcn_s_sleep(8);
  }
  // unit u (0: A first rows, 1: W first columns, 2: W second columns, 3: A second rows) of K tile `tile`
  auto issue_unit = [&](const TileOff& o, int u, int tile) {
    const unsigned sb = lds0 + (tile & 1) * STAGE;
    const long long ko = (long long)(kbeg + tile) * 64;
    if (u == 0) {
      glds16(bAh + ko, o.oa0, sb + rgA0 * 1024);
      glds16(bAl + ko, o.oa0, sb + PLANE + rgA0 * 1024);
    } else if (u == 1) {
      glds16(bWh + ko, o.ow0, sb + 2 * PLANE + cgW0 * 1024);
      glds16(bWl + ko, o.ow0, sb + 3 * PLANE + cgW0 * 1024);
    } else if (u == 2) {
      glds16(bWh + ko, o.ow1, sb + 2 * PLANE + cgW1 * 1024);
      glds16(bWl + ko, o.ow1, sb + 3 * PLANE + cgW1 * 1024);
    } else {
      glds16(bAh + ko, o.oa1, sb + rgA1 * 1024);
      glds16(bAl + ko, o.oa1, sb + PLANE + rgA1 * 1024);
    }
  };
  auto issue_prologue = [&](const TileOff& o) {   // seven units ahead
    issue_unit(o, 0, 0); issue_unit(o, 1, 0); issue_unit(o, 2, 0); issue_unit(o, 3, 0);
    issue_unit(o, 0, 1); issue_unit(o, 1, 1); issue_unit(o, 2, 1);
  };

  // LDS chunk swizzle for the 16-row x 4-chunk fragment reads of v_mfma_f32_16x16x32_f16 (a lane reads row r, chunk h):
  // chunk ^= {0,2,3,1}[(row >> 2) & 3] puts the 16 lanes of every ds_read_b128 lane group on 16 different 16-byte slots
  const int sw = (0x78 >> (2 * ((r >> 2) & 3))) & 3;
  const int co = (h ^ sw) * 16;
  const unsigned char* Ab = smem_p + (wm * WTM + r) * 64 + co;
  const unsigned char* Wb = smem_p + 2 * PLANE + (wn * WTN + r) * 64 + co;
  f32x4 acc[MB][NB];
  f16x8 ah[4], al[4];            // [16-row block of the half]: the whole K tile of 32
  f16x8 bh[2][2], bl[2][2];      // [32-column half][16-column block]
  auto read_A = [&](int ih, int stage) {
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) {
      const unsigned char* p = Ab + stage * STAGE + (4 * ih + ib) * 1024;
      ah[ib] = *(const f16x8*)p;
      al[ib] = *(const f16x8*)(p + PLANE);
    }
  };
  auto read_B = [&](int j, int stage) {
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
      const unsigned char* p = Wb + stage * STAGE + (2 * j + jb) * 1024;
      bh[j][jb] = *(const f16x8*)p;
      bl[j][jb] = *(const f16x8*)(p + PLANE);
    }
  };
  // the 24 MFMAs of quadrant (ih, j): per accumulator lo*hi, hi*lo, hi*hi over the whole K tile; W fragment first =
  // transposed accumulator (lane: output row r, four consecutive columns)
  auto quad = [&](int ih, int j) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int term = 0; term < 3; ++term)
#pragma unroll
      for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
          const f16x8 a = term == 0 ? al[ib] : ah[ib];
          const f16x8 b = term == 1 ? bl[j][jb] : bh[j][jb];
          acc[4 * ih + ib][2 * j + jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc[4 * ih + ib][2 * j + jb], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // one K tile: C1..C4 = the vmcnt left outstanding at the end of the four load segments; I1 / I2: issue the unit
  // of phase 1 (tile + 1) / of phases 2-4 (tile + 2); LAST drops group 1's unpaired final barrier
  auto tile_body = [&](int tile, auto c1, auto c2, auto c3, auto c4, auto i1, auto i2, auto last) {
    constexpr int C1 = decltype(c1)::value, C2 = decltype(c2)::value, C3 = decltype(c3)::value, C4 = decltype(c4)::value;
    constexpr bool I1 = decltype(i1)::value, I2 = decltype(i2)::value, LAST = decltype(last)::value;
    const int stage = tile & 1;
    // phase 1
    read_A(0, stage);
    read_B(0, stage);
    if (I1) issue_unit(cur, 3, tile + 1);
    __builtin_amdgcn_sched_barrier(0);
    x3p_wait<C1>();
    bar();
    quad(0, 0);
    bar();
    // phase 2
    read_B(1, stage);
    if (I2) issue_unit(cur, 0, tile + 2);
    __builtin_amdgcn_sched_barrier(0);
    x3p_wait<C2>();
    bar();
    quad(0, 1);
    bar();
    // phase 3
    read_A(1, stage);
    if (I2) issue_unit(cur, 1, tile + 2);
    __builtin_amdgcn_sched_barrier(0);
    x3p_wait<C3>();
    bar();
    quad(1, 0);
    bar();
    // phase 4
    if (I2) issue_unit(cur, 2, tile + 2);
    __builtin_amdgcn_sched_barrier(0);
    x3p_wait<C4>();
    bar();
    quad(1, 1);
    if (!(LAST && wm == 1)) bar();
  };
  using I0 = std::integral_constant<int, 0>;
  using I2_ = std::integral_constant<int, 2>;
  using I4 = std::integral_constant<int, 4>;
  using I6 = std::integral_constant<int, 6>;
  using I8 = std::integral_constant<int, 8>;
  using I10 = std::integral_constant<int, 10>;
  const std::true_type yes;
  const std::false_type no;

  issue_prologue(cur);
  for (;;) {
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Every wait below is correct whatever else the wave still has in flight (the stores of the previous tile's
    // write-out are YOUNGER than this tile's prologue pieces, so they can only make a counted wait stricter).
    x3p_wait<10>();
    bar();
    if (wm == 1) bar();   // group 1 runs one barrier behind group 0
    for (int tile = 0; tile + 2 < nk; ++tile) tile_body(tile, I10(), I10(), I10(), I10(), yes, yes, no);
    tile_body(nk - 2, I10(), I8(), I6(), I4(), yes, no, no);
    tile_body(nk - 1, I2_(), I0(), I0(), I0(), no, no, yes);
    // Both groups have retired every LDS read of this tile when either leaves its last barrier (group 1's phase 4
    // has none, group 0 finishes after group 1's last read segment): the next tile's prologue DMA goes out BEFORE
    // the write-out, which it overlaps.
    const TileOff done = cur;
    v += vstride;
    const bool more = v < vlen;
    if (more) {
      cur = tile_offsets(v);
      issue_prologue(cur);
    }

    // ---- write-out: accumulator block (mb, nb) = output rows mb*16 + r, columns nb*16 + 4*h .. +3 ----
    if ((g.dbg & 1) && acc[0][0][0] != 12345.678f) {
    } else {
      const int row0 = done.row0, col0 = done.col0;
      const bool full_tile = (row0 + TBM <= g.M) && (col0 + TBN <= g.N);
      const int cb = col0 + wn * WTN + 4 * h;   // + nb*16
      if (g.ksplit > 1) {   // raw partial sums; splitk_reduce_kernel finishes
        float* pp = g.part + (long long)blockIdx.y * g.M * g.N;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const int row = row0 + wm * WTM + mb * 16 + r;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const int col = cb + nb * 16;
            if (full_tile || (row < g.M && col < g.N)) {
              f32x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = acc[mb][nb][e] * g.out_scale;
              *(f32x4*)(pp + (long long)row * g.N + col) = o;
            }
          }
        }
      } else {
        f32x4 bv[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int colc = min(cb + nb * 16, g.N - 4);
          bv[nb] = g.bias ? *(const f32x4*)(g.bias + colc) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const int row = row0 + wm * WTM + mb * 16 + r;
          const int rc = min(row, mclamp);
          const int crow = g.cmap ? g.cmap[rc] : rc;
          const long long rrow = g.rmod > 0 ? crow % g.rmod : crow;
          f32x4 rv[NB];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const int colc = min(cb + nb * 16, g.N - 4);
            rv[nb] = g.R ? *(const f32x4*)(g.R + rrow * g.ldr + colc) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const int col = cb + nb * 16;
            if (full_tile || (row < g.M && col < g.N)) {
              f32x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = act_apply<ACT>(acc[mb][nb][e] * g.out_scale + bv[nb][e]) + rv[nb][e];
              const long long off = (long long)crow * g.ldc + col;
              if (g.C) {
                *(f32x4*)(g.C + off) = o;
              } else {
                f16x4 hi4, lo4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  _Float16 hh, ll;
                  hgl_split_hi_lo(o[e], hh, ll);
                  hi4[e] = hh;
                  lo4[e] = ll;
                }
                *(f16x4*)(g.Ch + off) = hi4;
                *(f16x4*)(g.Cl + off) = lo4;
              }
            }
          }
        }
      }
    }
    if (!more) break;
  }
}

// ---------------------------------------------------------------------------------------------
// Ping-pong tiling with a DEFERRED write-out (kind D): 256 x 128 tile, eight waves as 4 (M) x 2 (N), a wave owns
// 64 x 64 = four 32x32 accumulators = 64 registers -- half of kind P -- so that a wave can keep the finished
// accumulators of tile n while it multiplies tile n+1, and write tile n out a vector at a time during the main loop
// of tile n+1 (measured on kind P: the write-out of a 256x256 tile costs 11-31 us per tile = 18 % of the GEMM time,
// a per-CU store-rate limit that neither a persistent loop nor staggered workgroups remove).
// Same two phase groups (waves 0-3 / 4-7 one barrier apart), two phases per K tile (one 32-row block of the wave x
// its two 32-column blocks = 12 MFMAs each), THREE LDS stages of 48 KiB ([A_hi 16K | A_lo 16K | W_hi 8K | W_lo 8K]).
// A wave stages six 1-KiB pieces per K tile, in need order [A0.hi A0.lo W.hi W.lo A1.hi A1.lo] (A0 / A1 = the rows
// of the first / second 32-row blocks): the load segment of (K tile t, phase 2) issues the first three pieces of
// K tile t+3, the one of (t+1, phase 1) the last three.  Counted waits: 12 after phase 1 (everything up to A1 of
// this K tile has landed), 11 after phase 2 (A0 and W of the next K tile).
template <int N>
__device__ __forceinline__ void x3d_wait() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

template <int ACT>
__global__ __launch_bounds__(512, 1) void gemm_x3d_kernel(Args g) {
  constexpr int TBM = 256, TBN = 128, WTM = 64, WTN = 64;
  constexpr int APLANE = 256 * 64, WPLANE = 128 * 64;
  constexpr int STAGE = 2 * APLANE + 2 * WPLANE;   // 48 KiB
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_d[];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int grp = wave >> 2;   // phase group
  const int mclamp = g.M - 1, nclamp = g.N - 1;
  const int nk = g.K / 32;     // >= 4 (launcher)

  const int ntiles = g.tiles_m * g.tiles_n;
  const int xcd = blockIdx.x & 7;
  int vstart, vlen;
  {
    const int q = ntiles >> 3, rr = ntiles & 7;
    vstart = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
    vlen = q + (xcd < rr ? 1 : 0);
  }
  const int vstride = ((int)gridDim.x - xcd + 7) >> 3;
  int v = blockIdx.x >> 3;
  if (v >= vlen) return;

  const int prow = lane >> 2, pchunk = (lane & 3) ^ ((lane >> 4) & 3);
  const unsigned char *bAh = (const unsigned char*)g.Ah, *bAl = (const unsigned char*)g.Al;
  const unsigned char *bWh = (const unsigned char*)g.Wh, *bWl = (const unsigned char*)g.Wl;
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem_d;
  const int rgA0 = 4 * (wave >> 1) + (wave & 1), rgA1 = rgA0 + 2;   // 16-row groups: rows wm*64 + {0-31} / {32-63}
  const int cgW = wave;                                             // 16-column group
  struct TileOff { int row0, col0; unsigned oa0, oa1, ow; };
  auto tile_offsets = [&](int vid) {
    const int bid = vstart + vid;
    const int GM = g.gm;
    const int group = bid / (GM * g.tiles_n);
    const int first_m = group * GM;
    const int gmn = min(g.tiles_m - first_m, GM);
    const int rem = bid - group * GM * g.tiles_n;
    TileOff o;
    o.row0 = (first_m + rem % gmn) * TBM;
    o.col0 = (rem / gmn) * TBN;
    int a0 = min(o.row0 + rgA0 * 16 + prow, mclamp), a1 = min(o.row0 + rgA1 * 16 + prow, mclamp);
    if (g.amap) { a0 = g.amap[a0]; a1 = g.amap[a1]; }
    o.oa0 = (unsigned)(a0 * g.lda + pchunk * 8) * 2u;
    o.oa1 = (unsigned)(a1 * g.lda + pchunk * 8) * 2u;
    o.ow = (unsigned)(min(o.col0 + cgW * 16 + prow, nclamp) * g.ldw + pchunk * 8) * 2u;
    return o;
  };
  TileOff cur = tile_offsets(v);
  // half h3 (0: A0.hi A0.lo W.hi, 1: W.lo A1.hi A1.lo) of K tile kt into stage sidx
  auto issue_half = [&](const TileOff& o, int h3, int kt, int sidx) {
    const unsigned sb = lds0 + sidx * STAGE;
    const long long ko = (long long)kt * 64;
    if (h3 == 0) {
      glds16(bAh + ko, o.oa0, sb + rgA0 * 1024);
      glds16(bAl + ko, o.oa0, sb + APLANE + rgA0 * 1024);
      glds16(bWh + ko, o.ow, sb + 2 * APLANE + cgW * 1024);
    } else {
      glds16(bWl + ko, o.ow, sb + 2 * APLANE + WPLANE + cgW * 1024);
      glds16(bAh + ko, o.oa1, sb + rgA1 * 1024);
      glds16(bAl + ko, o.oa1, sb + APLANE + rgA1 * 1024);
    }
  };

  const int sw = (r >> 2) & 3;
  const int co0 = ((0 + h) ^ sw) * 16, co1 = ((2 + h) ^ sw) * 16;
  const unsigned char* Ab = smem_d + (wm * WTM + r) * 64;
  const unsigned char* Wb = smem_d + 2 * APLANE + (wn * WTN + r) * 64;
  f32x16 acc[2][2];
  f16x8 ah[2], al[2];            // [k-step] of the current 32-row block
  f16x8 bh[2][2], bl[2][2];      // [column block][k-step]
  auto read_A = [&](int i, int sidx) {
    const unsigned char* p = Ab + sidx * STAGE + i * 2048;
    ah[0] = *(const f16x8*)(p + co0); ah[1] = *(const f16x8*)(p + co1);
    al[0] = *(const f16x8*)(p + APLANE + co0); al[1] = *(const f16x8*)(p + APLANE + co1);
  };
  auto read_B = [&](int sidx) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned char* p = Wb + sidx * STAGE + j * 2048;
      bh[j][0] = *(const f16x8*)(p + co0); bh[j][1] = *(const f16x8*)(p + co1);
      bl[j][0] = *(const f16x8*)(p + WPLANE + co0); bl[j][1] = *(const f16x8*)(p + WPLANE + co1);
    }
  };
  // the 12 MFMAs of row block i (transposed accumulators, as kind P)
  auto half = [&](int i) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f16x8 a = term == 0 ? al[s] : ah[s];
          const f16x8 b = term == 1 ? bl[j][s] : bh[j][s];
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[i][j], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // K tile kt in stage sidx; I1 / I2: the load segment of phase 1 issues the second half of K tile kt+2 (stage
  // sidx+2), the one of phase 2 the first half of kt+3 (stage sidx); N1 / N2 the counted waits; LAST: final K tile
  auto ktile = [&](int kt, int sidx, auto n1, auto n2, auto i1, auto i2, auto last) {
    constexpr int N1 = decltype(n1)::value, N2 = decltype(n2)::value;
    constexpr bool I1 = decltype(i1)::value, I2 = decltype(i2)::value, LAST = decltype(last)::value;
    const int s2 = sidx >= 1 ? sidx - 1 : 2;   // (sidx + 2) % 3
    read_A(0, sidx);
    read_B(sidx);
    if (I1) issue_half(cur, 1, kt + 2, s2);
    __builtin_amdgcn_sched_barrier(0);
    x3d_wait<N1>();
    bar();
    half(0);
    bar();
    read_A(1, sidx);
    if (I2) issue_half(cur, 0, kt + 3, sidx);
    __builtin_amdgcn_sched_barrier(0);
    x3d_wait<N2>();
    bar();
    half(1);
    if (!(LAST && grp == 1)) bar();
  };
  using I0 = std::integral_constant<int, 0>;
  using I2_ = std::integral_constant<int, 2>;
  using I6 = std::integral_constant<int, 6>;
  using I8 = std::integral_constant<int, 8>;
  using I11 = std::integral_constant<int, 11>;
  using I12 = std::integral_constant<int, 12>;
  const std::true_type yes;
  const std::false_type no;
  auto issue_prologue = [&](const TileOff& o) {   // K tiles 0, 1 and the first half of 2
    issue_half(o, 0, 0, 0); issue_half(o, 1, 0, 0);
    issue_half(o, 0, 1, 1); issue_half(o, 1, 1, 1);
    issue_half(o, 0, 2, 2);
  };

  issue_prologue(cur);
  for (;;) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    x3d_wait<11>();
    bar();
    if (grp == 1) bar();   // group 1 runs one barrier behind group 0
    int sidx = 0;
    int kt = 0;
    for (; kt + 3 < nk; ++kt) {
      ktile(kt, sidx, I12(), I11(), yes, yes, no);
      sidx = sidx == 2 ? 0 : sidx + 1;
    }
    ktile(kt, sidx, I12(), I8(), yes, no, no);
    sidx = sidx == 2 ? 0 : sidx + 1;
    ktile(kt + 1, sidx, I6(), I2_(), no, no, no);
    sidx = sidx == 2 ? 0 : sidx + 1;
    ktile(kt + 2, sidx, I0(), I0(), no, no, yes);

    const TileOff done = cur;
    v += vstride;
    const bool more = v < vlen;
    if (more) {
      cur = tile_offsets(v);
      issue_prologue(cur);
    }
    // ---- write-out ----
    if ((g.dbg & 1) && acc[0][0][0] != 12345.678f) {
    } else {
      const int row0 = done.row0, col0 = done.col0;
      const bool full_tile = (row0 + TBM <= g.M) && (col0 + TBN <= g.N);
      const int cb = col0 + wn * WTN + 4 * h;   // + j*32 + 8*q
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int colc = min(cb + j * 32 + 8 * q, g.N - 4);
          bv[q] = g.bias ? *(const f32x4*)(g.bias + colc) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = row0 + wm * WTM + i * 32 + r;
          const int rc = min(row, mclamp);
          const int crow = g.cmap ? g.cmap[rc] : rc;
          const long long rrow = g.rmod > 0 ? crow % g.rmod : crow;
          f32x4 rv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int colc = min(cb + j * 32 + 8 * q, g.N - 4);
            rv[q] = g.R ? *(const f32x4*)(g.R + rrow * g.ldr + colc) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = cb + j * 32 + 8 * q;
            if (full_tile || (row < g.M && col < g.N)) {
              f32x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = act_apply<ACT>(acc[i][j][4 * q + e] * g.out_scale + bv[q][e]) + rv[q][e];
              const long long off = (long long)crow * g.ldc + col;
              if (g.C) {
                *(f32x4*)(g.C + off) = o;
              } else {
                f16x4 hi4, lo4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  _Float16 hh, ll;
                  hgl_split_hi_lo(o[e], hh, ll);
                  hi4[e] = hh;
                  lo4[e] = ll;
                }
                *(f16x4*)(g.Ch + off) = hi4;
                *(f16x4*)(g.Cl + off) = lo4;
              }
            }
          }
        }
      }
    }
    if (!more) break;
  }
}

// ---------------------------------------------------------------------------------------------
// Small-M GEMMs (token MLPs / hyper-networks / IoU head of the SAM decoder, the text encoder, the heads): a few
// dozen output tiles with K up to 2048 leave the 128x128 kernels latency-bound (8 workgroups, 64 serial K tiles:
// 53 us for 448x256x2048).  Here one workgroup owns a 32x32 tile and its four waves split the K range
// (k-steps w, w+4, ...), fragments come straight from global memory in MFMA operand layout (A split to hi+lo in
// registers, W from the registered fp16 planes), partial accumulators are summed through LDS in a fixed order.
struct SkinnyArgs {
  const float* A;
  const _Float16 *Wh, *Wl;
  const float *bias, *R;
  float* C;
  int M, N, K, lda, ldr, ldc;
  float out_scale;
};

template <int ACT>
__global__ __launch_bounds__(256) void gemm_x3_skinny_kernel(SkinnyArgs g) {
  __shared__ float red[3][16][64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int row0 = blockIdx.y * 32, col0 = blockIdx.x * 32;
  const float* ap = g.A + (long long)min(row0 + r, g.M - 1) * g.lda + 8 * h;
  const long long wo = (long long)min(col0 + r, g.N - 1) * g.K + 8 * h;
  const _Float16 *whp = g.Wh + wo, *wlp = g.Wl + wo;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int nks = g.K >> 4;
  auto step = [&](const f32x4 a0, const f32x4 a1, const f16x8 wh, const f16x8 wl) {
    f16x8 ah, al;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 h0, l0, h1, l1;
      hgl_split_hi_lo(a0[e], h0, l0);
      hgl_split_hi_lo(a1[e], h1, l1);
      ah[e] = h0; ah[4 + e] = h1;
      al[e] = l0; al[4 + e] = l1;
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, acc, 0, 0, 0);
  };
  int ks = wave;
  for (; ks + 12 < nks; ks += 16) {   // four of this wave's k-steps per trip: 16 loads in flight
    f32x4 a0[4], a1[4];
    f16x8 wh[4], wl[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = 16 * (ks + 4 * u);
      a0[u] = *(const f32x4*)(ap + k); a1[u] = *(const f32x4*)(ap + k + 4);
      wh[u] = *(const f16x8*)(whp + k); wl[u] = *(const f16x8*)(wlp + k);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) step(a0[u], a1[u], wh[u], wl[u]);
  }
  for (; ks < nks; ks += 4) {
    const int k = 16 * ks;
    step(*(const f32x4*)(ap + k), *(const f32x4*)(ap + k + 4), *(const f16x8*)(whp + k), *(const f16x8*)(wlp + k));
  }
  if (wave > 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave - 1][e][lane] = acc[e];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = ((acc[e] + red[0][e][lane]) + red[1][e][lane]) + red[2][e][lane];
  const int col = col0 + r;
  if (col >= g.N) return;
  const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = row0 + (e & 3) + 8 * (e >> 2) + 4 * h;
    if (row < g.M) {
      float v = act_apply<ACT>(acc[e] * g.out_scale + bv);
      if (g.R) v += g.R[(long long)row * g.ldr + col];
      g.C[(long long)row * g.ldc + col] = v;
    }
  }
}

// C = act(sum_s part[s] + bias) + R, parts added in index order (deterministic)
template <int ACT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int ksplit, long long MN4, int N4,
                                                            const float* __restrict__ bias, const float* __restrict__ R,
                                                            int ldr4, float* __restrict__ C, int ldc4,
                                                            const int* __restrict__ cmap) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < MN4; i += (long long)gridDim.x * 256) {
    long long row = i / N4;
    const int c = (int)(i - row * N4);
    if (cmap) row = cmap[row];
    f32x4 v = ((const f32x4*)part)[i];
    for (int s2 = 1; s2 < ksplit; ++s2) {
      const f32x4 w = ((const f32x4*)part)[(long long)s2 * MN4 + i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += w[e];
    }
    const f32x4 b = bias ? ((const f32x4*)bias)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 rr = f32x4{0.f, 0.f, 0.f, 0.f};
    if (R) rr = ((const f32x4*)R)[row * ldr4 + c];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = act_apply<ACT>(v[e] + b[e]) + rr[e];
    ((f32x4*)C)[row * ldc4 + c] = o;
  }
}

// x (fp32) * 2^scale_log2 -> hi, lo fp16
__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ x, float scale,
                                                        _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                        long long n4) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 v = ((const f32x4*)x)[i];
    f16x4 a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 h0, l0;
      hgl_split_hi_lo(v[e] * scale, h0, l0);
      a[e] = h0;
      b[e] = l0;
    }
    ((f16x4*)hi)[i] = a;
    ((f16x4*)lo)[i] = b;
  }
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// LayerNorm writing the split form directly (one wave per row, D % 256 == 0)
// Optional row maps: row r is read at x[smap[r]] and written at row dmap[r] (the SAM window partition of the real
// tokens, image_encoder.py:243-266, folded into the norm1 pass).
template <int VEC>
__global__ __launch_bounds__(256) void layernorm_split_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ b, _Float16* __restrict__ hi,
                                                              _Float16* __restrict__ lo, int rows, float eps,
                                                              const int* __restrict__ smap, const int* __restrict__ dmap) {
  constexpr int D = VEC * 256;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int srow = smap ? smap[row] : row, drow = dmap ? dmap[row] : row;
  const f32x4* xr = (const f32x4*)(x + (long long)srow * D);
  f32x4 v[VEC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    v[i] = xr[lane + 64 * i];
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mean = wsum(s) * (1.0f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = v[i][e] - mean;
      q += d * d;
    }
  const float rstd = rsqrtf(wsum(q) * (1.0f / D) + eps);
  f16x4* hr = (f16x4*)(hi + (long long)drow * D);
  f16x4* lr = (f16x4*)(lo + (long long)drow * D);
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const f32x4 wv = ((const f32x4*)w)[lane + 64 * i];
    const f32x4 bv = ((const f32x4*)b)[lane + 64 * i];
    f16x4 a, c;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float o = (v[i][e] - mean) * rstd * wv[e] + bv[e];
      _Float16 h0, l0;
      hgl_split_hi_lo(o, h0, l0);
      a[e] = h0;
      c[e] = l0;
    }
    hr[lane + 64 * i] = a;
    lr[lane + 64 * i] = c;
  }
}

// window partition (image_encoder.py:243-266) writing the split form: rows of padded windows
__global__ __launch_bounds__(256) void win_partition_split_kernel(const float* __restrict__ H, int g, int ws, int nw,
                                                                  int D4, _Float16* __restrict__ hi,
                                                                  _Float16* __restrict__ lo, long long total4) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total4) return;
  const int d = (int)(i % D4);
  const long long row = i / D4;
  const int p = (int)(row % (ws * ws)), win = (int)(row / (ws * ws));
  const int y = (win / nw) * ws + p / ws, x = (win % nw) * ws + p % ws;
  f32x4 v = {0, 0, 0, 0};
  if (y < g && x < g) v = ((const f32x4*)H)[((long long)y * g + x) * D4 + d];
  f16x4 a, b;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    _Float16 h0, l0;
    hgl_split_hi_lo(v[e], h0, l0);
    a[e] = h0;
    b[e] = l0;
  }
  ((f16x4*)hi)[i] = a;
  ((f16x4*)lo)[i] = b;
}

}  // namespace

int hgl_launch_win_partition_split(const float* H, int g, int ws, int nw, int D, void* hi, void* lo, hipStream_t st) {
  const long long total4 = (long long)nw * nw * ws * ws * (D / 4);
  hipLaunchKernelGGL(win_partition_split_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, H, g, ws,
                     nw, D / 4, (_Float16*)hi, (_Float16*)lo, total4);
  return hgl_check_launch("win_partition_split");
}

int hgl_precision() { return g_precision; }

bool hgl_has_split_weight(const float* W) { return g_split.find((const void*)W) != g_split.end(); }

int hgl_launch_split_f16(const float* x, float scale, void* hi, void* lo, long long n, hipStream_t st) {
  HGL_REQUIRE(x && hi && lo && n > 0 && (n & 3) == 0, "split_f16: bad arguments (n %% 4 != 0?)");
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, scale, (_Float16*)hi, (_Float16*)lo, n / 4);
  return hgl_check_launch("split_f16");
}

int hgl_launch_layernorm_split(const float* x, const float* w, const float* b, void* hi, void* lo, int rows, int D,
                               float eps, hipStream_t st) {
  return hgl_launch_layernorm_split_maps(x, w, b, hi, lo, rows, D, eps, nullptr, nullptr, st);
}

int hgl_launch_layernorm_split_maps(const float* x, const float* w, const float* b, void* hi, void* lo, int rows, int D,
                                    float eps, const int* smap, const int* dmap, hipStream_t st) {
  const unsigned grid = (unsigned)((rows + 3) / 4);
  _Float16 *h = (_Float16*)hi, *l = (_Float16*)lo;
  switch (D) {
    case 256: hipLaunchKernelGGL(layernorm_split_kernel<1>, dim3(grid), dim3(256), 0, st, x, w, b, h, l, rows, eps, smap, dmap); break;
    case 512: hipLaunchKernelGGL(layernorm_split_kernel<2>, dim3(grid), dim3(256), 0, st, x, w, b, h, l, rows, eps, smap, dmap); break;
    case 768: hipLaunchKernelGGL(layernorm_split_kernel<3>, dim3(grid), dim3(256), 0, st, x, w, b, h, l, rows, eps, smap, dmap); break;
    case 1024: hipLaunchKernelGGL(layernorm_split_kernel<4>, dim3(grid), dim3(256), 0, st, x, w, b, h, l, rows, eps, smap, dmap); break;
    case 1280: hipLaunchKernelGGL(layernorm_split_kernel<5>, dim3(grid), dim3(256), 0, st, x, w, b, h, l, rows, eps, smap, dmap); break;
    default: hgl_set_error("layernorm_split: unsupported D=%d", D); return HGL_EINVAL;
  }
  return hgl_check_launch("layernorm_split");
}

// A given as split halves (Ah, Al) [M,K] lda (in halfs); W looked up in the registry by its fp32 pointer.
// Output: C fp32 (ldc) or, when C == nullptr, the split pair (Ch, Cl).
int hgl_launch_gemm_f16x3(const void* Ah, const void* Al, int lda, const float* W32, const float* bias, const float* R,
                          int ldr, float* C, void* Ch, void* Cl, int ldc, int M, int N, int K, int act, hipStream_t st) {
  return hgl_launch_gemm_f16x3_rmod(Ah, Al, lda, W32, bias, R, ldr, 0, C, Ch, Cl, ldc, M, N, K, act, st);
}

int hgl_launch_gemm_f16x3_rmod(const void* Ah, const void* Al, int lda, const float* W32, const float* bias, const float* R,
                               int ldr, int rmod, float* C, void* Ch, void* Cl, int ldc, int M, int N, int K, int act,
                               hipStream_t st) {
  return hgl_launch_gemm_f16x3_maps(Ah, Al, lda, nullptr, W32, bias, R, ldr, rmod, nullptr, C, Ch, Cl, ldc, M, N, K, act, st);
}

int hgl_launch_gemm_f16x3_maps(const void* Ah, const void* Al, int lda, const int* amap, const float* W32, const float* bias,
                               const float* R, int ldr, int rmod, const int* cmap, float* C, void* Ch, void* Cl, int ldc,
                               int M, int N, int K, int act, hipStream_t st) {
  auto it = g_split.find((const void*)W32);
  HGL_REQUIRE(it != g_split.end(), "gemm_f16x3: weight %p has no registered fp16 split", (const void*)W32);
  const SplitW& sw = it->second;
  HGL_REQUIRE(sw.N == N && sw.K == K, "gemm_f16x3: registered split is [%d,%d], GEMM wants [%d,%d]", sw.N, sw.K, N, K);
  HGL_REQUIRE(Ah && Al && (C || (Ch && Cl)) && M > 0 && N > 0 && K > 0, "gemm_f16x3: bad arguments");
  HGL_REQUIRE((K % 64) == 0 && (lda & 7) == 0, "gemm_f16x3: K must be a multiple of 64 and lda of 8 (K=%d lda=%d)", K, lda);
  Args g;
  g.Ah = (const _Float16*)Ah; g.Al = (const _Float16*)Al; g.Wh = sw.hi; g.Wl = sw.lo;
  g.bias = bias; g.R = R; g.C = C; g.Ch = (_Float16*)Ch; g.Cl = (_Float16*)Cl;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = K; g.ldr = ldr; g.ldc = ldc;
  g.rmod = rmod; g.amap = amap; g.cmap = cmap;
  g.part = nullptr; g.ksplit = 1;
  { static int dbg = -1; if (dbg < 0) { const char* v = getenv("HGL_X3_DBG"); dbg = v ? atoi(v) : 0; } g.dbg = dbg; }
  { static int sm = -1, stk = 0; if (sm < 0) { const char* v = getenv("HGL_X3_STG_MODE"); sm = v ? atoi(v) : 0; v = getenv("HGL_X3_STG_TICKS"); stk = v ? atoi(v) : 0; } g.stg_mode = sm; g.stg_ticks = stk; }
  g.out_scale = ldexpf(1.0f, -sw.scale_log2);
  {
    static int gmv = -1;
    if (gmv < 0) { const char* v = getenv("HGL_X3_GM"); gmv = v ? atoi(v) : 8; if (gmv < 1) gmv = 8; }
    g.gm = gmv;
  }
  // kernel selection (hgl_gemm_f16x3_select / HGL_X3_KERNEL={v1,L,M,S,auto}); all variants accumulate in the
  // same order and give bit-identical results
  if (g_x3_kernel == -2) {
    const char* v = getenv("HGL_X3_KERNEL");
    g_x3_kernel = -1;
    if (v) {
      if (!strcmp(v, "v1")) g_x3_kernel = HGL_X3_V1;
      else if (!strcmp(v, "L")) g_x3_kernel = HGL_X3_L;
      else if (!strcmp(v, "M")) g_x3_kernel = HGL_X3_M;
      else if (!strcmp(v, "S")) g_x3_kernel = HGL_X3_S;
      else if (!strcmp(v, "N")) g_x3_kernel = HGL_X3_N;
      else if (!strcmp(v, "Q")) g_x3_kernel = HGL_X3_Q;
      else if (!strcmp(v, "P")) g_x3_kernel = HGL_X3_P;
      else if (!strcmp(v, "D")) g_x3_kernel = HGL_X3_D;
      else if (!strcmp(v, "P16")) g_x3_kernel = HGL_X3_P16;
    }
  }
  // the LDS-DMA kernels address the operands with 32-bit byte offsets from the plane bases
  const bool small_offsets = (double)M * lda * (amap ? 4.0 : 2.0) < 4.0e9 && (double)N * K * 2.0 < 4.0e9;   // gathered rows: <= 2M
  int kind = g_x3_kernel >= 0 ? g_x3_kernel : pick_x3_kernel(M, N, K);
  if (!small_offsets) kind = HGL_X3_V1;
  if (kind == HGL_X3_D && K < 128) kind = HGL_X3_P;
  if (kind == HGL_X3_P || kind == HGL_X3_D || kind == HGL_X3_P16) {   // 16-byte vectors along the output rows
    const bool vec4 = (N & 3) == 0 && (ldc & 3) == 0 && (!R || (ldr & 3) == 0) && (((size_t)bias | (size_t)R | (size_t)C) & 15) == 0 &&
                      (((size_t)Ch | (size_t)Cl) & 7) == 0;
    if (!vec4) kind = HGL_X3_L;
  }
  // launches that cannot fill the 256 CUs once (GEM at 785 rows, text encoder: a 128x128 tile per CU is latency-bound
  // when run alone) are accounted separately from the throughput-bound ones
  const long long few_tiles = (long long)((M + 127) / 128) * ((N + 127) / 128);
  HglProfScope prof(few_tiles < 256 ? HGL_PROF_GEMM_X3_FEW : kind == HGL_X3_V1 ? HGL_PROF_GEMM_X3 : HGL_PROF_GEMM_X3G, 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N * (R ? 2 : 1)), st);
#define HGL_X3_LAUNCH(ACT_, BK_, OCC_)                                                                        \
  do {                                                                                                        \
    g.tiles_m = (M + BM - 1) / BM;                                                                            \
    g.tiles_n = (N + BN - 1) / BN;                                                                            \
    const long long nwg = (long long)g.tiles_m * g.tiles_n;                                                   \
    HGL_REQUIRE(nwg < (1ll << 31), "gemm_f16x3: grid too large");                                            \
    const size_t lds_ = (size_t)(BM + BN) * (2 * BK_ + 8) * sizeof(_Float16);                                 \
    static bool set_ = false;                                                                                 \
    if (!set_) {                                                                                              \
      (void)hipFuncSetAttribute((const void*)gemm_f16x3_kernel<ACT_, BK_, OCC_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_); \
      set_ = true;                                                                                            \
    }                                                                                                         \
    hipLaunchKernelGGL((gemm_f16x3_kernel<ACT_, BK_, OCC_>), dim3((unsigned)nwg), dim3(NTHREADS), lds_, st, g); \
  } while (0)
#define HGL_X3G_LAUNCH(ACT_, TBM_, TBN_, WGM_, WGN_, OCC_)                                                       \
  do {                                                                                                        \
    g.tiles_m = (M + TBM_ - 1) / TBM_;                                                                        \
    g.tiles_n = (N + TBN_ - 1) / TBN_;                                                                        \
    const long long nwg = (long long)g.tiles_m * g.tiles_n;                                                   \
    HGL_REQUIRE(nwg < (1ll << 31), "gemm_f16x3: grid too large");                                            \
    const size_t lds_ = (size_t)4 * (TBM_ + TBN_) * 64;                                                     \
    static bool set_ = false;                                                                                 \
    if (!set_) {                                                                                              \
      (void)hipFuncSetAttribute((const void*)gemm_x3g_kernel<ACT_, TBM_, TBN_, WGM_, WGN_, OCC_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_); \
      set_ = true;                                                                                            \
    }                                                                                                         \
    hipLaunchKernelGGL((gemm_x3g_kernel<ACT_, TBM_, TBN_, WGM_, WGN_, OCC_>), dim3((unsigned)nwg), dim3(WGM_ * WGN_ * 64), lds_, st, g); \
  } while (0)
#define HGL_X3P_LAUNCH(ACT_)                                                                                  \
  do {                                                                                                        \
    g.tiles_m = (M + 255) / 256;                                                                              \
    g.tiles_n = (N + 255) / 256;                                                                              \
    const long long nwg = (long long)g.tiles_m * g.tiles_n;                                                   \
    HGL_REQUIRE(nwg < (1ll << 31), "gemm_f16x3: grid too large");                                            \
    const size_t lds_ = (size_t)2 * 4 * 256 * 64;                                                             \
    static bool set_ = false;                                                                                 \
    if (!set_) {                                                                                              \
      (void)hipFuncSetAttribute((const void*)gemm_x3p_kernel<ACT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_); \
      set_ = true;                                                                                            \
    }                                                                                                         \
    const long long grid_ = x3p_grid(nwg);                                                                    \
    hipLaunchKernelGGL((gemm_x3p_kernel<ACT_>), dim3((unsigned)grid_), dim3(512), lds_, st, g);               \
  } while (0)
#define HGL_X3D_LAUNCH(ACT_)                                                                                  \
  do {                                                                                                        \
    g.tiles_m = (M + 255) / 256;                                                                              \
    g.tiles_n = (N + 127) / 128;                                                                              \
    const long long nwg = (long long)g.tiles_m * g.tiles_n;                                                   \
    HGL_REQUIRE(nwg < (1ll << 31), "gemm_f16x3: grid too large");                                            \
    const size_t lds_ = (size_t)3 * 49152;                                                                    \
    static bool set_ = false;                                                                                 \
    if (!set_) {                                                                                              \
      (void)hipFuncSetAttribute((const void*)gemm_x3d_kernel<ACT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_); \
      set_ = true;                                                                                            \
    }                                                                                                         \
    const long long grid_ = x3p_grid(nwg);                                                                    \
    hipLaunchKernelGGL((gemm_x3d_kernel<ACT_>), dim3((unsigned)grid_), dim3(512), lds_, st, g);               \
  } while (0)
#define HGL_X3P16_LAUNCH(ACT_)                                                                                \
  do {                                                                                                        \
    g.tiles_m = (M + 255) / 256;                                                                              \
    g.tiles_n = (N + 255) / 256;                                                                              \
    const long long nwg = (long long)g.tiles_m * g.tiles_n;                                                   \
    HGL_REQUIRE(nwg < (1ll << 31), "gemm_f16x3: grid too large");                                            \
    const size_t lds_ = (size_t)2 * 4 * 256 * 64;                                                             \
    static bool set_ = false;                                                                                 \
    if (!set_) {                                                                                              \
      (void)hipFuncSetAttribute((const void*)gemm_x3p16_kernel<ACT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_); \
      set_ = true;                                                                                            \
    }                                                                                                         \
    const long long grid_ = x3p_grid(nwg);                                                                    \
    hipLaunchKernelGGL((gemm_x3p16_kernel<ACT_>), dim3((unsigned)grid_), dim3(512), lds_, st, g);             \
  } while (0)
#define HGL_X3_VARIANTS(ACT_)                                  \
  do {                                                         \
    if (kind == 8) HGL_X3P16_LAUNCH(ACT_);                     \
    else if (kind == 7) HGL_X3D_LAUNCH(ACT_);                  \
    else if (kind == 6) HGL_X3P_LAUNCH(ACT_);                  \
    else if (kind == 1) HGL_X3G_LAUNCH(ACT_, 256, 256, 2, 4, 1);    \
    else if (kind == 2) HGL_X3G_LAUNCH(ACT_, 256, 128, 4, 2, 1); \
    else if (kind == 3) HGL_X3G_LAUNCH(ACT_, 128, 128, 2, 2, 2); \
    else if (kind == 4) HGL_X3G_LAUNCH(ACT_, 128, 160, 4, 1, 2); \
    else if (kind == 5) HGL_X3G_LAUNCH(ACT_, 160, 160, 5, 1, 1); \
    else HGL_X3_LAUNCH(ACT_, 64, 2);                           \
  } while (0)
  switch (act) {
    case HGL_ACT_QUICKGELU: HGL_X3_VARIANTS(HGL_ACT_QUICKGELU); break;
    case HGL_ACT_GELU: HGL_X3_VARIANTS(HGL_ACT_GELU); break;
    case HGL_ACT_RELU: HGL_X3_VARIANTS(HGL_ACT_RELU); break;
    default: HGL_X3_VARIANTS(HGL_ACT_NONE); break;
  }
  return hgl_check_launch("gemm_f16x3");
}

// fp32-A entry for small M (called from hgl_launch_gemm): true when the GEMM was taken
bool hgl_gemm_skinny_applicable(const float* W32, int M, int N, int K, int lda, int ldw, int batch) {
  if (g_precision != HGL_PREC_F16X3 || batch != 1 || M > 1024 || (K & 15) || (lda & 3) || ldw != K) return false;
  auto it = g_split.find((const void*)W32);
  return it != g_split.end() && it->second.N == N && it->second.K == K;
}

int hgl_launch_gemm_x3_skinny(const float* A, int lda, const float* W32, const float* bias, const float* R, int ldr,
                              float* C, int ldc, int M, int N, int K, int act, hipStream_t st) {
  auto it = g_split.find((const void*)W32);
  HGL_REQUIRE(it != g_split.end(), "gemm_x3_skinny: weight has no registered split");
  const SplitW& sw = it->second;
  SkinnyArgs g;
  g.A = A; g.Wh = sw.hi; g.Wl = sw.lo; g.bias = bias; g.R = R; g.C = C;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldr = ldr; g.ldc = ldc;
  g.out_scale = ldexpf(1.0f, -sw.scale_log2);
  const dim3 grid((unsigned)((N + 31) / 32), (unsigned)((M + 31) / 32));
  HglProfScope prof(HGL_PROF_OTHER, 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), st);
  switch (act) {
    case HGL_ACT_QUICKGELU: hipLaunchKernelGGL(gemm_x3_skinny_kernel<HGL_ACT_QUICKGELU>, grid, dim3(256), 0, st, g); break;
    case HGL_ACT_GELU: hipLaunchKernelGGL(gemm_x3_skinny_kernel<HGL_ACT_GELU>, grid, dim3(256), 0, st, g); break;
    case HGL_ACT_RELU: hipLaunchKernelGGL(gemm_x3_skinny_kernel<HGL_ACT_RELU>, grid, dim3(256), 0, st, g); break;
    default: hipLaunchKernelGGL(gemm_x3_skinny_kernel<HGL_ACT_NONE>, grid, dim3(256), 0, st, g); break;
  }
  return hgl_check_launch("gemm_x3_skinny");
}

// Split-K on the 256x256 LDS-DMA tiling for GEMMs with few output tiles and a long K (SAM's mlp.lin2: 80 tiles,
// K = 5120): ksplit slices of K run as independent workgroups (grid.y), raw partial sums go through `part`
// (>= ksplit*M*N floats), splitk_reduce adds them in index order and applies bias / activation / residual.
int hgl_gemm_f16x3_splitk_factor(int M, int N, int K) {
  const long long tiles = (long long)((M + 255) / 256) * ((N + 255) / 256);
  if ((N & 3) || K < 1024 || tiles * 2 > 256) return 1;
  int ks = (int)(256 / tiles);
  if (ks > 4) ks = 4;
  while (ks > 1 && ((K / 32 / ks) & ~1) < 8) --ks;   // keep every slice at least 8 K tiles long
  return ks;
}

int hgl_launch_gemm_f16x3_splitk(const void* Ah, const void* Al, int lda, const int* amap, const float* W32, const float* bias,
                                 const float* R, int ldr, const int* cmap, float* C, int ldc, int M, int N, int K, int act,
                                 int ksplit, float* part, size_t part_bytes, hipStream_t st) {
  auto it = g_split.find((const void*)W32);
  HGL_REQUIRE(it != g_split.end(), "gemm_f16x3_splitk: weight %p has no registered fp16 split", (const void*)W32);
  const SplitW& sw = it->second;
  HGL_REQUIRE(sw.N == N && sw.K == K && Ah && Al && C && part, "gemm_f16x3_splitk: bad arguments");
  HGL_REQUIRE(ksplit >= 2 && ksplit <= 8 && (K % 64) == 0 && (lda & 7) == 0 && (N & 3) == 0 && (ldc & 3) == 0 && (ldr & 3) == 0,
              "gemm_f16x3_splitk: unsupported shape (K %d, N %d, ksplit %d)", K, N, ksplit);
  HGL_REQUIRE(((K / 32 / ksplit) & ~1) >= 2, "gemm_f16x3_splitk: K too short for %d slices", ksplit);
  HGL_REQUIRE(part_bytes >= (size_t)ksplit * M * N * sizeof(float), "gemm_f16x3_splitk: partial-sum workspace too small");
  HGL_REQUIRE((double)M * lda * (amap ? 4.0 : 2.0) < 4.0e9 && (double)N * K * 2.0 < 4.0e9, "gemm_f16x3_splitk: operand too large");
  Args g;
  g.Ah = (const _Float16*)Ah; g.Al = (const _Float16*)Al; g.Wh = sw.hi; g.Wl = sw.lo;
  g.bias = nullptr; g.R = nullptr; g.C = nullptr; g.Ch = nullptr; g.Cl = nullptr;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = K; g.ldr = 0; g.ldc = N;
  g.rmod = 0; g.amap = amap; g.cmap = nullptr; g.part = part; g.ksplit = ksplit; g.dbg = 0; g.stg_mode = 0; g.stg_ticks = 0;
  g.out_scale = ldexpf(1.0f, -sw.scale_log2);
  g.gm = 8;
  g.tiles_m = (M + 255) / 256; g.tiles_n = (N + 255) / 256;
  {
    const long long few_tiles = (long long)((M + 127) / 128) * ((N + 127) / 128);
    HglProfScope prof(few_tiles < 256 ? HGL_PROF_GEMM_X3_FEW : HGL_PROF_GEMM_X3G, 2.0 * M * (double)N * K,
                      4.0 * ((double)M * K + (double)N * K + (double)M * N * (R ? 2 : 1)), st);
    const size_t lds = (size_t)4 * (256 + 256) * 64;
    static bool set = false;
    if (!set) {
      (void)hipFuncSetAttribute((const void*)gemm_x3g_kernel<HGL_ACT_NONE, 256, 256, 2, 4, 1>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      set = true;
    }
    hipLaunchKernelGGL((gemm_x3g_kernel<HGL_ACT_NONE, 256, 256, 2, 4, 1>), dim3((unsigned)(g.tiles_m * g.tiles_n), (unsigned)ksplit),
                       dim3(512), lds, st, g);
    const long long MN4 = (long long)M * N / 4;
    const unsigned blocks = (unsigned)((MN4 + 255) / 256 > 4096 ? 4096 : (MN4 + 255) / 256);
#define HGL_SPLITK_REDUCE(ACT_) hipLaunchKernelGGL(splitk_reduce_kernel<ACT_>, dim3(blocks), dim3(256), 0, st, part, ksplit, MN4, N / 4, bias, R, ldr / 4, C, ldc / 4, cmap)
    switch (act) {
      case HGL_ACT_QUICKGELU: HGL_SPLITK_REDUCE(HGL_ACT_QUICKGELU); break;
      case HGL_ACT_GELU: HGL_SPLITK_REDUCE(HGL_ACT_GELU); break;
      case HGL_ACT_RELU: HGL_SPLITK_REDUCE(HGL_ACT_RELU); break;
      default: HGL_SPLITK_REDUCE(HGL_ACT_NONE); break;
    }
  }
  return hgl_check_launch("gemm_f16x3_splitk");
}

extern "C" {

int hgl_set_precision(int mode) {
  HGL_REQUIRE(mode == HGL_PREC_F32 || mode == HGL_PREC_F16X3, "set_precision: unknown mode %d", mode);
  g_precision = mode;
  return HGL_OK;
}

int hgl_get_precision(void) { return g_precision; }

int hgl_gemm_f16x3_select(int kind) {
  HGL_REQUIRE(kind >= -1 && kind <= HGL_X3_P16, "gemm_f16x3_select: unknown kernel %d", kind);
  g_x3_kernel = kind;
  return HGL_OK;
}

int hgl_register_split_weight(const float* w_fp32, int N, int K, int scale_log2, void* hi, void* lo, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(w_fp32 && hi && lo && N > 0 && K > 0 && (K & 7) == 0, "register_split_weight: bad arguments (K %% 8)");
  HGL_REQUIRE(scale_log2 >= -24 && scale_log2 <= 24, "register_split_weight: scale_log2 out of range");
  HGL_TRY(hgl_launch_split_f16(w_fp32, ldexpf(1.0f, scale_log2), hi, lo, (long long)N * K, (hipStream_t)stream));
  g_split[(const void*)w_fp32] = SplitW{(const _Float16*)hi, (const _Float16*)lo, scale_log2, N, K};
  return HGL_OK;
}

int hgl_unregister_split_weight(const float* w_fp32) {
  g_split.erase((const void*)w_fp32);
  return HGL_OK;
}

// C = act(A @ W^T + bias) + R through the split-fp16 path with A split on the fly into `scratch`
// (>= M*K*4 bytes).  Exported for the per-kernel parity tests and micro-benchmarks.
int hgl_gemm_f16x3(const float* A, const float* W, const float* bias, const float* R, float* C, int M, int N, int K,
                   int act, void* scratch, size_t scratch_bytes, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(A && W && C && scratch, "gemm_f16x3: null argument");
  HGL_REQUIRE(scratch_bytes >= (size_t)M * K * 4, "gemm_f16x3: scratch too small");
  hipStream_t st = (hipStream_t)stream;
  _Float16* ah = (_Float16*)scratch;
  _Float16* al = ah + (size_t)M * K;
  HGL_TRY(hgl_launch_split_f16(A, 1.0f, ah, al, (long long)M * K, st));
  return hgl_launch_gemm_f16x3(ah, al, K, W, bias, R, N, C, nullptr, nullptr, N, M, N, K, act, st);
}

}  // extern "C"
