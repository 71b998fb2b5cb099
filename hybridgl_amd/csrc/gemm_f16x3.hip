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
enum { HGL_X3_V1 = 0, HGL_X3_L = 1, HGL_X3_M = 2, HGL_X3_S = 3, HGL_X3_N = 4, HGL_X3_Q = 5 };

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
    }
  }
  // the LDS-DMA kernels address the operands with 32-bit byte offsets from the plane bases
  const bool small_offsets = (double)M * lda * (amap ? 4.0 : 2.0) < 4.0e9 && (double)N * K * 2.0 < 4.0e9;   // gathered rows: <= 2M
  int kind = g_x3_kernel >= 0 ? g_x3_kernel : pick_x3_kernel(M, N, K);
  if (!small_offsets) kind = HGL_X3_V1;
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
#define HGL_X3_VARIANTS(ACT_)                                  \
  do {                                                         \
    if (kind == 1) HGL_X3G_LAUNCH(ACT_, 256, 256, 2, 4, 1);    \
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
  g.rmod = 0; g.amap = amap; g.cmap = nullptr; g.part = part; g.ksplit = ksplit;
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
  HGL_REQUIRE(kind >= -1 && kind <= HGL_X3_Q, "gemm_f16x3_select: unknown kernel %d", kind);
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
