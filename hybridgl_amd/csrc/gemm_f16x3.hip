// fp32-accurate GEMM on the fp16 matrix cores: C = act(A @ W^T + bias) + R with every fp32 operand
// split into two fp16 halves, x = hi + lo (hi = fp16(x), lo = fp16(x - hi)), and
//     a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi        (the dropped a_lo*b_lo term is 2^-22 relative)
// evaluated by three v_mfma_f32_16x16x32_f16 per 32-deep K step into ONE fp32 accumulator.  Each
// fp16 x fp16 product is exact in fp32 and the representation error of hi+lo is 2^-23, so the
// result carries ~2^-22 relative error per term -- within 4x of the fp32 MFMA path (gemm.hip) at
// 16/3 = 5.3x its matrix-core rate.  Weights are scaled by a power of two before splitting so
// that their lo halves stay in the fp16 normal range; the scale is undone in the epilogue.
//
// Two tilings, same instruction (16x16x32: the chip is power-limited under this load and holds a higher
// clock on it than on 32x32x16 -- +9 % measured at equal cycles), same order of the products per accumulator
// (K steps of 32 ascending; lo*hi, hi*lo, hi*hi) => bit-identical outputs (tested):
//   * gemm_f16x3_kernel: 128x128 tile, 4 waves x 64x64, register-staged through a padded LDS image, two
//     workgroups per CU: launches with few tiles, any N / leading dimensions;
//   * gemm_x3p_kernel: 256x256 tile, LDS-DMA, two wave groups one barrier apart (ping-pong), persistent:
//     everything that fills the chip.
// The accumulators are kept TRANSPOSED (the W fragment is the MFMA's first operand): a lane holds one output row
// and four consecutive columns per accumulator block, so bias / residual / output move as 16-byte vectors.
#include "hgl_common.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
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
  bool lo_zero;   // every lo half is zero: the weight is fp16-valued (OpenAI's CLIP archives), W = W_hi exactly
};
std::unordered_map<const void*, SplitW> g_split;   // fp32 weight pointer -> its fp16 split
std::mutex g_split_mu;                             // the registry may be touched from several host threads
int g_precision = HGL_PREC_F32;
enum { HGL_X3_V1 = 0, HGL_X3_P = 1 };

bool find_split(const void* w, SplitW* out) {
  std::lock_guard<std::mutex> lk(g_split_mu);
  auto it = g_split.find(w);
  if (it == g_split.end()) return false;
  if (out) *out = it->second;
  return true;
}

struct Args {
  const _Float16 *Ah, *Al, *Wh, *Wl;
  const float *bias, *R;
  float* C;
  _Float16 *Ch, *Cl;   // split output (when C == nullptr)
  int M, N, K, lda, ldw, ldr, ldc;
  float out_scale;
  int lo_zero;                // the W_lo plane is all zero: the A_hi * W_lo products are skipped (they add exact zeros)
  int tiles_m, tiles_n;
  int gm;      // M-tiles per tile group (L2 blocking of the resident tile set)
  float* part;   // split-K (ping-pong kernel, grid.y = ksplit): raw partial sums [ksplit][M][N]; bias / act / residual are
  int ksplit;    // applied by splitk_reduce_kernel, which sums the parts in a fixed order
  const int *amap, *cmap;   // optional row maps: A row m is read from row amap[m]; output / residual row m lives at cmap[m]
  int rmod;    // residual row = row % rmod when > 0 (a residual shared by every batch of rows), else row
  int rp_p, rp_t;   // ping-pong kernel, rmod > 0: schedule slot s of the row tiles works on row tile (s % rp_p) * rp_t + s / rp_p --
                    // consecutive slots are the SAME rows of the residual table in consecutive batches, so the 256 KiB of
                    // table rows a tile adds stay in the XCD's L2 across the batches (rp_t = 0: identity)
  int vec4;    // N, ldc, ldr multiples of 4 and 16-byte aligned bases: the write-out moves 16-byte vectors
};

int g_x3_kernel = -2;   // -2: not yet chosen (-> -1); -1: cost model; >= 0: forced by hgl_gemm_f16x3_select

// grid of the persistent ping-pong kernel: one workgroup per CU (HGL_X3_PERSIST=0: one per tile)
int x3_num_cus() {
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
  }
  return ncu;
}
long long x3p_grid(long long tiles) {
  static int persist = -1;
  if (persist < 0) {
    persist = HGL_DIAG_SWITCH("HGL_X3_PERSIST", 1);
  }
  const int ncu = x3_num_cus();
  return persist && tiles > ncu ? ncu : tiles;
}

// Cost model (us; calibrated on MI355X with cold caches between launches, tools/x3_bench.py X3_COLD=1).  Ping-pong
// kernel: a workgroup takes ceil(tiles / CUs) tiles of 256x256, each 14 + 2.08 * (K / 32) (+12 with a residual: the
// write-out is not overlapped).  Register-staged kernel: 128x128 tiles at two workgroups per CU (512 slots), a round
// costs 13 + 1.52 * (K / 32); a partial last round costs nearly a full one, a CU that holds a single workgroup
// finishes it in ~0.62 of the pair's time.
int pick_x3_kernel(int M, int N, int K, bool has_r) {
  const double nk = K / 32.0;
  const int ncu = x3_num_cus();
  const double tp = (double)((M + 255) / 256) * ((N + 255) / 256);
  const double t_p = ceil(tp / ncu) * (14.0 + 2.08 * nk + (has_r ? 12.0 : 0.0)) + 5.0;
  const double tv = (double)((M + 127) / 128) * ((N + 127) / 128);
  const double x = tv / (2.0 * ncu), up = ceil(x);
  const double rounds = x <= 0.5 ? 0.62 : up - 0.25 * (up - x);
  const double t_v = rounds * (13.0 + 1.52 * nk);
  return t_p < t_v ? HGL_X3_P : HGL_X3_V1;
}

template <int ACT>
__device__ __forceinline__ float act_apply(float x) {
  if constexpr (ACT == HGL_ACT_QUICKGELU) return hgl_quick_gelu(x);
  if constexpr (ACT == HGL_ACT_GELU) return hgl_gelu_erf(x);
  if constexpr (ACT == HGL_ACT_RELU) return x > 0.0f ? x : 0.0f;
  return x;
}

// Write-out of one transposed 16x16 accumulator block: this lane holds output row `row` and the four columns
// col .. col+3.  vec: 16-byte accesses (g.vec4); otherwise element by element with bounds checks.
template <int ACT>
__device__ __forceinline__ void x3_store_block(const Args& g, const f32x4 acc, int row, int crow, long long rrow, int col,
                                               bool inside, float& amax) {
  if (g.vec4) {
    if (!(inside || (row < g.M && col < g.N))) return;
    const f32x4 bv = g.bias ? *(const f32x4*)(g.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 rv = g.R ? *(const f32x4*)(g.R + rrow * g.ldr + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = act_apply<ACT>(acc[e] * g.out_scale + bv[e]) + rv[e];
    const long long off = (long long)crow * g.ldc + col;
    if (g.C) {
      *(f32x4*)(g.C + off) = o;
    } else {
      f16x4 hi4, lo4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        _Float16 hh, ll;
        hgl_split_hi_lo(o[e], hh, ll, amax);
        hi4[e] = hh;
        lo4[e] = ll;
      }
      *(f16x4*)(g.Ch + off) = hi4;
      *(f16x4*)(g.Cl + off) = lo4;
    }
  } else {
    if (row >= g.M) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = col + e;
      if (c < g.N) {
        const float bv = g.bias ? g.bias[c] : 0.0f;
        const float rv = g.R ? g.R[rrow * g.ldr + c] : 0.0f;
        const float v = act_apply<ACT>(acc[e] * g.out_scale + bv) + rv;
        const long long off = (long long)crow * g.ldc + c;
        if (g.C) {
          g.C[off] = v;
        } else {
          _Float16 hh, ll;
          hgl_split_hi_lo(v, hh, ll, amax);
          g.Ch[off] = hh;
          g.Cl[off] = ll;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Register-staged tiling: 128x128 block tile, 4 waves x (4x4) 16x16 accumulator blocks, K-contiguous operands staged
// global -> registers -> LDS with 16-byte accesses.  An LDS row holds the hi and the lo halves of one operand row for
// BK = 64 (128 B + 128 B) plus one 16-byte pad (row stride 272 B = 17 x 16 B: at most one two-way conflict per
// ds_read_b128 lane group).
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
  const int r = lane & 15, h = lane >> 4;   // fragment row / 8-wide k group; accumulator: row r, columns 4*h..4*h+3
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

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (g.K + BK - 1) / BK;
  load_tile(0);
  store_tile();
  __syncthreads();
  const _Float16* As = smem + (wm * 64 + r) * ROW_H + 8 * h;
  const _Float16* Ws = smem + (BM + wn * 64 + r) * ROW_H + 8 * h;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int s = 0; s < BK / 32; ++s) {
      f16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        ah[b] = *(const f16x8*)(As + b * 16 * ROW_H + 32 * s); al[b] = *(const f16x8*)(As + b * 16 * ROW_H + BK + 32 * s);
        bh[b] = *(const f16x8*)(Ws + b * 16 * ROW_H + 32 * s); bl[b] = *(const f16x8*)(Ws + b * 16 * ROW_H + BK + 32 * s);
      }
      // small cross terms first, then the hi*hi term; W fragment first = transposed accumulator
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f16x8 a = term == 0 ? al[i] : ah[i];
            const f16x8 b = term == 1 ? bl[j] : bh[j];
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc[i][j], 0, 0, 0);
          }
    }
    __syncthreads();
    if (kt + 1 < nk) store_tile();
    __syncthreads();
  }

  // ---- write-out ----
  const bool full_tile = (row0 + BM <= g.M) && (col0 + BN <= g.N);
  float amax = 0.f;   // of the values split in the write-out
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = row0 + wm * 64 + i * 16 + r;
    const int rc = min(row, mclamp);
    const int crow = g.cmap ? g.cmap[rc] : rc;
    const long long rrow = g.rmod > 0 ? crow % g.rmod : crow;
#pragma unroll
    for (int j = 0; j < 4; ++j) x3_store_block<ACT>(g, acc[i][j], row, crow, rrow, col0 + wn * 64 + j * 16 + 4 * h, full_tile, amax);
  }
  hgl_split_commit(amax);
}

// ---------------------------------------------------------------------------------------------
// Ping-pong tiling: 256x256 tile, eight waves as 2 (M) x 4 (N), a wave owns 128 x 64 = 8 x 4 accumulator blocks.
// Direct-to-LDS staging (global_load_lds_dwordx4: no VGPR staging, no ds_write pass); BK = 32; one stage holds four
// planes [A_hi | A_lo | W_hi | W_lo] of 256 rows x 64 B, unpadded (an LDS-DMA wave-instruction writes 1 KiB = 16 rows
// contiguously); two stages.  Bank conflicts are avoided by an XOR swizzle of the 16-byte chunk index,
// chunk ^= {0,2,3,1}[(row >> 2) & 3], applied to the per-lane GLOBAL source address on the way in and to the
// ds_read_b128 address on the way out: the 16 lanes of every ds_read_b128 lane group (a lane reads fragment row
// lane & 15, chunk lane >> 4) land on 16 different 16-byte slots.
//
// Schedule.  The waves form two groups of four (waves 0-3 = the upper 128 rows, waves 4-7 = the lower 128 rows: one
// wave of each group per SIMD) that run ONE BARRIER APART.  A K tile is cut into four phases (a 64 x 32 quadrant of
// the wave's tile = 24 MFMAs = 384 matrix-pipe cycles each); a phase is a load segment (the quadrant's fragment
// reads + two LDS-DMA pieces + the counted waits), a barrier, an MFMA-only segment under s_setprio 1, a barrier.
// Because group 1 executed one extra barrier at the start, every barrier interval has one group issuing nothing but
// MFMAs while the other group's loads, DMA issue and waits run beside it on the same SIMDs.  (The earlier schedule
// -- eight waves in the same phase, one barrier per K tile -- left the matrix pipe 53-56 % busy; this one measures
// 68 % at the clock the chip then holds, with the remainder in the write-out.)
//
// Staging is in UNITS of 16 KiB (16 pieces, two per wave) in the order the phases need them: U1 = A rows {0-63} of
// each group's half, U2 = W columns {0-31} of each wave column's 64, U3 = W columns {32-63}, U4 = A rows {64-127};
// phase 1 reads U1 + U2, phase 2 U3, phase 3 U4, phase 4 nothing.  A unit's LDS slot is free again one barrier after
// its reading phase (the fragment reads are retired by lgkmcnt(0) BEFORE the load segment's barrier), so the DMA runs
// seven units ahead of the reads: the load segment of (tile t, phase 1) issues U4(t+1), phases 2-4 issue U1-U3(t+2).
// Each load segment ends with s_waitcnt vmcnt(10): all but the wave's five youngest units have landed, i.e. every
// unit read by ANY wave in the next segment -- published by the barrier that follows.  A unit has five phases
// (about 3800 matrix-pipe cycles) to land.  The last two K tiles use the exact smaller counts (8, 6, 4, 2, 0).
//
// Persistent: a workgroup walks its share of the tile list and issues the next tile's seven-unit prologue before it
// writes the finished tile out.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

// The lane number, computed where it is used.  In the ping-pong kernel (256 VGPRs, the limit at two waves per SIMD) every
// lane-derived address that lives from the prologue to its one use per output tile is a spill candidate, and a spilled
// value comes back through a scratch load the compiler waits for with vmcnt(0) -- in the middle of the hand-counted DMA
// stream (seven units in flight at a tile boundary).  volatile: not merged with the prologue's copy, not hoisted.
__device__ __forceinline__ int x3p_fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// One LDS-DMA wave-instruction: lane i copies 16 B from sbase + voff(i) to LDS byte lds_addr + 16*i.  Written as
// inline asm for the SGPR-base + 32-bit-VGPR-offset addressing form (the builtin keeps a 64-bit VGPR address
// per piece).  M0 is compiler-reserved: saved and restored around the instruction.  The compiler does not count
// these on vmcnt -- the consumers wait with explicit counted s_waitcnt.
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_addr)
               : "memory");
}

// one dword per lane: LDS[lds_addr + 4*lane] = *(sbase + voff)
__device__ __forceinline__ void glds4(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_addr)
               : "memory");
}

template <int N>
__device__ __forceinline__ void x3p_wait() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
// The counts below are written for NT = 3, where every staging unit is TWO DMA pieces per wave (hi and lo plane).  With an
// fp16-valued weight (NT = 2) the W_lo plane is all zero and is NOT STAGED: the two W units of a K tile are one piece each
// (U1 = 2, U2 = 1, U3 = 1, U4 = 2 pieces).  "All but the five youngest units" is then 8, 8, 7, 7 pieces in the four phases
// (7 everywhere is stricter, hence safe); the drained tail 8, 6, 4, 2, 0 becomes 6, 4, 3, 2, 0; the 32 stores of a full
// write-out are added as before; the four units a K tile issues are 6 pieces.
constexpr int x3p_cnt(int nt, int c) {
  return nt == 3 ? c : (c == 42 ? 39 : c == 10 ? 7 : c == 8 ? 6 : c == 6 ? 4 : c == 4 ? 3 : c);
}

template <int ACT, int WLOADS, int NT>
__global__ __launch_bounds__(512, 1) void gemm_x3p_kernel(Args g) {
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
  // rows of A this lane stages (a0, a1: clamped, before the row map) and its two W offsets
  auto tile_rows = [&](int vid, int& a0, int& a1) {
    const int bid = vstart + vid;
    const int GM = g.gm;
    const int group = bid / (GM * g.tiles_n);
    const int first_m = group * GM;
    const int gmn = min(g.tiles_m - first_m, GM);
    const int rem = bid - group * GM * g.tiles_n;
    TileOff o;
    int rt = first_m + rem % gmn;
    if (g.rp_t) rt = (rt % g.rp_p) * g.rp_t + rt / g.rp_p;
    o.row0 = rt * TBM;
    o.col0 = (rem / gmn) * TBN;
    a0 = min(o.row0 + rgA0 * 16 + prow, mclamp);
    a1 = min(o.row0 + rgA1 * 16 + prow, mclamp);
    o.oa0 = o.oa1 = 0;
    o.ow0 = (unsigned)(min(o.col0 + cgW0 * 16 + prow, nclamp) * g.ldw + pchunk * 8) * 2u;
    o.ow1 = (unsigned)(min(o.col0 + cgW1 * 16 + prow, nclamp) * g.ldw + pchunk * 8) * 2u;
    return o;
  };
  auto a_offset = [&](int arow) { return (unsigned)(arow * g.lda + pchunk * 8) * 2u; };
  auto tile_offsets = [&](int vid) {
    int a0, a1;
    TileOff o = tile_rows(vid, a0, a1);
    if (g.amap) { a0 = g.amap[a0]; a1 = g.amap[a1]; }
    o.oa0 = a_offset(a0);
    o.oa1 = a_offset(a1);
    return o;
  };
  TileOff cur = tile_offsets(v);
  // unit u (0: A first rows, 1: W first columns, 2: W second columns, 3: A second rows) of K tile `tile`; a K tile index
  // >= nk means K tile (tile - nk) of the NEXT output tile of this workgroup (nk is even, so the stage parity carries on):
  // the DMA stream runs seven units ahead straight across the boundary between two output tiles
  auto issue_unit = [&](const TileOff& o0, int u, int tile, bool wrap = false) {
    const unsigned sb = lds0 + (tile & 1) * STAGE;
    const long long ko = (long long)(kbeg + (wrap ? tile - nk : tile)) * 64;
    // the next tile's four per-lane offsets wait in the wave's write-out patch (idle during the main loop), not in VGPRs
    const unsigned* nx = (const unsigned*)(smem_p + 2 * STAGE + wave * 4096) + lane * 4;
    if (u == 0) {
      const unsigned off = wrap ? nx[0] : o0.oa0;
      glds16(bAh + ko, off, sb + rgA0 * 1024);
      glds16(bAl + ko, off, sb + PLANE + rgA0 * 1024);
    } else if (u == 1) {
      const unsigned off = wrap ? nx[2] : o0.ow0;
      glds16(bWh + ko, off, sb + 2 * PLANE + cgW0 * 1024);
      if (NT == 3) glds16(bWl + ko, off, sb + 3 * PLANE + cgW0 * 1024);     // NT = 2: the plane is all zero and never read
    } else if (u == 2) {
      const unsigned off = wrap ? nx[3] : o0.ow1;
      glds16(bWh + ko, off, sb + 2 * PLANE + cgW1 * 1024);
      if (NT == 3) glds16(bWl + ko, off, sb + 3 * PLANE + cgW1 * 1024);
    } else {
      const unsigned off = wrap ? nx[1] : o0.oa1;
      glds16(bAh + ko, off, sb + rgA1 * 1024);
      glds16(bAl + ko, off, sb + PLANE + rgA1 * 1024);
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
  // transposed accumulator (lane: output row r, four consecutive columns).  NT == 2: the weight is fp16-valued (W_lo == 0),
  // the hi*lo products would add exact zeros and are not issued (16 MFMAs; same accumulators bit for bit)
  auto quad = [&](int ih, int j) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int term = 0; term < 3; ++term)
#pragma unroll
      for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
          if (NT == 2 && term == 1) continue;
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
  // `go` (runtime, uniform): the workgroup has another output tile after this one -- the K tile then issues whatever I1 / I2
  // say and waits with the steady count of 10 (ONE code path for the accumulators: two instantiations behind an
  // if / else made the register allocator rename the accumulators and spill)
  auto tile_body = [&](int tile, auto c1, auto c2, auto c3, auto c4, auto i1, auto i2, auto last, bool go) {
    constexpr bool I1 = decltype(i1)::value, I2 = decltype(i2)::value, LAST = decltype(last)::value;
    const int stage = tile & 1;
    auto wait = [&](auto c) {
      constexpr int CC = decltype(c)::value;
      if constexpr (CC == 10 || CC == 42) x3p_wait<x3p_cnt(NT, CC)>();
      else if (go) x3p_wait<x3p_cnt(NT, 10)>();
      else x3p_wait<x3p_cnt(NT, CC)>();
    };
    // phase 1
    read_A(0, stage);
    read_B(0, stage);
    if (I1 || go) issue_unit(cur, 3, tile + 1, go && tile + 1 >= nk);
    __builtin_amdgcn_sched_barrier(0);
    wait(c1);
    bar();
    quad(0, 0);
    bar();
    // phase 2
    read_B(1, stage);
    if (I2 || go) issue_unit(cur, 0, tile + 2, go);
    __builtin_amdgcn_sched_barrier(0);
    wait(c2);
    bar();
    quad(0, 1);
    bar();
    // phase 3
    read_A(1, stage);
    if (I2 || go) issue_unit(cur, 1, tile + 2, go);
    __builtin_amdgcn_sched_barrier(0);
    wait(c3);
    bar();
    quad(1, 0);
    bar();
    // phase 4
    if (I2 || go) issue_unit(cur, 2, tile + 2, go);
    __builtin_amdgcn_sched_barrier(0);
    wait(c4);
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
  using I42 = std::integral_constant<int, 42>;
  const std::true_type yes;
  const std::false_type no;

  float amax = 0.f;   // of the values split in the write-outs
  bool prev_full = false;   // the previous tile of this workgroup was written out in full (see the waits below)
  issue_prologue(cur);
  for (;;) {
    const bool more = v + vstride < vlen;
    int nrow0 = 0, ncol0 = 0;
    unsigned char* const nxp = smem_p + 2 * STAGE + wave * 4096;   // the wave's write-out patch, idle until the write-out
    if (more) {
      // the next tile's offsets, parked in the patch.  With a row map the two A rows are map LOADS: a compiler-counted
      // wait for them here would be vmcnt(0) in the middle of the DMA stream and of the previous tile's stores, so
      // they go to the patch by DMA too and are turned into offsets before the last two K tiles
      int a0, a1;
      const TileOff nn = tile_rows(v + vstride, a0, a1);
      nrow0 = nn.row0; ncol0 = nn.col0;
      *(u32x4*)(nxp + lane * 16) = u32x4{a_offset(a0), a_offset(a1), nn.ow0, nn.ow1};
      if (g.amap) {
        const unsigned pl = lds0 + 2 * STAGE + wave * 4096 + 1024;
        glds4(g.amap, (unsigned)a0 * 4u, pl);
        glds4(g.amap, (unsigned)a1 * 4u, pl + 256);
      }
    }
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The write-out of the previous tile was issued AFTER the seven units that were in flight at the tile boundary, so its
    // loads / stores are YOUNGER than those units on the wave's in-order vmcnt.  A counted wait that ignores them is
    // always correct (stricter) but drains the stores before the tile can go on.  When the previous tile was a full one
    // its write-out issued at least 32 vector-memory operations (one store per row group, never predicated off), so the
    // waits that only need units older than the write-out may leave 32 more operations outstanding: the stores then
    // drain under the first five phases of this tile.  From phase 2 of K tile 1 on the needed units are younger than the
    // stores (count 10).
    if (prev_full && nk >= 4) {
      x3p_wait<x3p_cnt(NT, 42)>();
      bar();
      if (wm == 1) bar();   // group 1 runs one barrier behind group 0
      tile_body(0, I42(), I42(), I42(), I42(), yes, yes, no, false);
      tile_body(1, I42(), I10(), I10(), I10(), yes, yes, no, false);
      for (int tile = 2; tile + 2 < nk; ++tile) tile_body(tile, I10(), I10(), I10(), I10(), yes, yes, no, false);
    } else {
      x3p_wait<x3p_cnt(NT, 10)>();
      bar();
      if (wm == 1) bar();   // group 1 runs one barrier behind group 0
      for (int tile = 0; tile + 2 < nk; ++tile) tile_body(tile, I10(), I10(), I10(), I10(), yes, yes, no, false);
    }
    // the tile's bias vector (row-contiguous write-out: ONE 16-byte vector per lane) is fetched under the last K tile and
    // waited for BEFORE the write-out: a compiler-counted wait in the middle of the DMA stream would have to be vmcnt(0)
    // (loaded by asm so that the compiler does not count it; vmcnt(8) = the four units K tile nk-1 issues when the stream
    // goes on, and the drained tail otherwise)
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (more && g.amap) {
      // the two map DMAs are older than the 16 operations of the last two K tiles issued so far (none if nk == 2)
      if (nk >= 4) x3p_wait<x3p_cnt(NT, 10)>(); else x3p_wait<0>();
      const int ln = x3p_fresh_lane();
      const int m0 = *(const int*)(nxp + 1024 + ln * 4), m1 = *(const int*)(nxp + 1280 + ln * 4);
      *(unsigned*)(nxp + ln * 16) = a_offset(m0);
      *(unsigned*)(nxp + ln * 16 + 4) = a_offset(m1);
    }
    tile_body(nk - 2, I10(), I8(), I6(), I4(), yes, no, no, more);
    const bool has_bias = g.bias && g.ksplit <= 1;
    if (has_bias) {
      const float* bp = g.bias + min(cur.col0 + wn * WTN + 4 * (lane & 15), g.N - 4);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(bv) : "v"(bp) : "memory");
    }
    tile_body(nk - 1, I2_(), I0(), I0(), I0(), no, no, yes, more);
    if (has_bias) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(bv) : "n"(x3p_cnt(NT, 8)) : "memory");
    const TileOff done = cur;
    v += vstride;
    if (more) {
      const unsigned* nx = (const unsigned*)(smem_p + 2 * STAGE + wave * 4096) + lane * 4;
      cur.row0 = nrow0; cur.col0 = ncol0; cur.oa0 = nx[0]; cur.oa1 = nx[1]; cur.ow0 = nx[2]; cur.ow1 = nx[3];
    }
    // ---- write-out: accumulator block (mb, nb) = output rows mb*16 + r, columns nb*16 + 4*h .. +3 ----
    {
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
        // Row-contiguous write-out.  In the accumulator layout adjacent lanes hold DIFFERENT rows (lane = 16*h + r), and a
        // store instruction whose adjacent lanes are not contiguous costs ~260 cycles to issue whatever else the chip
        // does (tools/micro/store_rate.hip: 8.4k cycles per 256 KB tile for a LONE workgroup; 2.4k with 4 rows x 256 B per
        // instruction).  So every 16-row block goes through a wave-private 4 KiB patch of the 32 KiB of LDS the two
        // stages leave free: written as it lies in the registers (element (row r, 16-byte chunk c) at r*256 +
        // ((c ^ r) & 15)*16: both directions conflict-free), read back with lane = 16*(row & 3) + chunk.  Residual rows
        // are then loaded in the same row-contiguous form and a lane needs ONE bias vector.  LDS operations of a wave
        // execute in order, the patch is private: no barrier, no wait between the writes and the reads.
        unsigned char* const tp = smem_p + 2 * STAGE + wave * 4096;
        const int lw = x3p_fresh_lane();
        const int rl = lw >> 4, cl = lw & 15;
        const int col = col0 + wn * WTN + 4 * cl;
        const bool cok = col < g.N;
        const int colc = min(col, g.N - 4);
        // WLOADS = what the write-out reads: 0 nothing, 1 residual rows, 2 a row map (+ residual rows if any).  The flavours
        // are template instantiations because a load behind a RUN-TIME condition makes the compiler put `s_waitcnt
        // vmcnt(0)` at the join, taken or not: with `if (g.cmap)` inside the residual flavour every row block of every
        // residual GEMM drained all earlier stores and the next tile's DMA units before it went on.
        // Without loads nothing in the loop waits on
        // vmcnt and the stores stream out back to back; with a runtime `R ? load : 0` the compiler put an s_waitcnt
        // vmcnt(0) into every row block -- every block then waited for all earlier stores AND for the next tile's
        // prologue DMA (in-kernel stamps: ~500 cycles per store instruction).  With loads, the rows of block mb+1 are
        // fetched before block mb is finished (compiler-counted partial waits).
        int crow[2][4];
        f32x4 rv[2][4];
        auto fetch = [&](int mb, int (&cr)[4], f32x4 (&rr)[4]) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int rc = min(row0 + wm * WTM + mb * 16 + 4 * i + rl, mclamp);
            cr[i] = rc;
            if constexpr (WLOADS == 2) cr[i] = g.cmap[rc];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            rr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (WLOADS != 0) {
              const long long rrow = g.rmod > 0 ? cr[i] % g.rmod : cr[i];
              if (WLOADS == 1 || g.R) rr[i] = *(const f32x4*)(g.R + rrow * g.ldr + colc);
            }
          }
        };
        fetch(0, crow[0], rv[0]);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) *(f32x4*)(tp + r * 256 + (((4 * nb + h) ^ r) & 15) * 16) = acc[mb][nb];
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (no instruction: tells the compiler that the lanes
          __builtin_amdgcn_wave_barrier();                            //  of the wave exchange data through the patch)
          f32x4 v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int rho = 4 * i + rl;
            v[i] = *(const f32x4*)(tp + rho * 256 + ((cl ^ rho) & 15) * 16);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if (mb + 1 < MB) fetch(mb + 1, crow[(mb + 1) & 1], rv[(mb + 1) & 1]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = row0 + wm * WTM + mb * 16 + 4 * i + rl;
            if (full_tile || (row < g.M && cok)) {
              f32x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = act_apply<ACT>(v[i][e] * g.out_scale + bv[e]) + rv[mb & 1][i][e];
              const long long off = (long long)crow[mb & 1][i] * g.ldc + col;
              if (g.C) {
                *(f32x4*)(g.C + off) = o;
              } else {
                f16x4 hi4, lo4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  _Float16 hh, ll;
                  hgl_split_hi_lo(o[e], hh, ll, amax);
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
    prev_full = (done.row0 + TBM <= g.M) && (done.col0 + TBN <= g.N) && g.ksplit <= 1;
  }
  hgl_split_commit(amax);
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
  float amax = 0.f;
  auto step = [&](const f32x4 a0, const f32x4 a1, const f16x8 wh, const f16x8 wl) {
    f16x8 ah, al;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 h0, l0, h1, l1;
      hgl_split_hi_lo(a0[e], h0, l0, amax);
      hgl_split_hi_lo(a1[e], h1, l1, amax);
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
  hgl_split_commit(amax);
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
  // the 16 residual values in one batch of loads (a load behind `if (row < M)` inside the store loop is waited for before
  // the next one is issued: 16 serial round trips per tile)
  float rv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) rv[e] = 0.f;
  if (g.R) {
#pragma unroll
    for (int e = 0; e < 16; ++e) rv[e] = g.R[(long long)min(row0 + (e & 3) + 8 * (e >> 2) + 4 * h, g.M - 1) * g.ldr + col];
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = row0 + (e & 3) + 8 * (e >> 2) + 4 * h;
    if (row < g.M) g.C[(long long)row * g.ldc + col] = act_apply<ACT>(acc[e] * g.out_scale + bv) + rv[e];
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
__device__ unsigned g_lo_nonzero;
__global__ __launch_bounds__(256) void lo_nonzero_kernel(const u32x4* __restrict__ lo, long long n8) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= n8) return;
  const u32x4 v = lo[i];
  // (-0.0 halves count as zero: a product with them adds zeros as well)
  if (((v[0] | v[1] | v[2] | v[3]) & 0x7fff7fffu) != 0) atomicOr(&g_lo_nonzero, 1u);
}

__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ x, float scale,
                                                        _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                        long long n4) {
  float amax = 0.f;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 v = ((const f32x4*)x)[i];
    f16x4 a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 h0, l0;
      hgl_split_hi_lo(v[e] * scale, h0, l0, amax);
      a[e] = h0;
      b[e] = l0;
    }
    ((f16x4*)hi)[i] = a;
    ((f16x4*)lo)[i] = b;
  }
  hgl_split_commit(amax);
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

bool hgl_has_split_weight(const float* W) { return find_split((const void*)W, nullptr); }

// the registered fp16 halves of a weight, for fused kernels outside this file that multiply with them directly
bool hgl_get_split_weight(const float* W, const void** hi, const void** lo, int* scale_log2, int* N, int* K) {
  SplitW sw;
  if (!find_split((const void*)W, &sw)) return false;
  *hi = sw.hi; *lo = sw.lo; *scale_log2 = sw.scale_log2; *N = sw.N; *K = sw.K;
  return true;
}

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

namespace {

bool x3_vec4_ok(const float* bias, const float* R, int ldr, const float* C, const void* Ch, const void* Cl, int ldc, int N) {
  return (N & 3) == 0 && (ldc & 3) == 0 && (!R || (ldr & 3) == 0) && (((size_t)bias | (size_t)R | (size_t)C) & 15) == 0 &&
         (((size_t)Ch | (size_t)Cl) & 7) == 0;
}

template <int ACT>
int launch_x3_v1(Args& g, hipStream_t st) {
  constexpr int BK = 64;
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  const long long nwg = (long long)g.tiles_m * g.tiles_n;
  HGL_REQUIRE(nwg < (1ll << 31), "gemm_f16x3: grid too large");
  const size_t lds = (size_t)(BM + BN) * (2 * BK + 8) * sizeof(_Float16);
  HGL_RESERVE_LDS((gemm_f16x3_kernel<ACT, BK, 2>), lds, "gemm_f16x3");
  hipLaunchKernelGGL((gemm_f16x3_kernel<ACT, BK, 2>), dim3((unsigned)nwg), dim3(NTHREADS), lds, st, g);
  return HGL_OK;
}

template <int ACT, int WLOADS, int NT>
int launch_x3_p3(Args& g, hipStream_t st) {
  g.tiles_m = (g.M + 255) / 256;
  g.tiles_n = (g.N + 255) / 256;
  const long long nwg = (long long)g.tiles_m * g.tiles_n;
  HGL_REQUIRE(nwg < (1ll << 31), "gemm_f16x3: grid too large");
  const size_t lds = (size_t)2 * 4 * 256 * 64 + 8 * 4096;   // two stages + the write-out patches: all 160 KiB
  HGL_RESERVE_LDS((gemm_x3p_kernel<ACT, WLOADS, NT>), lds, "gemm_f16x3 (ping-pong)");
  const long long grid = g.ksplit > 1 ? nwg : x3p_grid(nwg);
  hipLaunchKernelGGL((gemm_x3p_kernel<ACT, WLOADS, NT>), dim3((unsigned)grid, (unsigned)(g.ksplit > 1 ? g.ksplit : 1)), dim3(512), lds, st, g);
  return HGL_OK;
}

template <int ACT, int WLOADS>
int launch_x3_p2(Args& g, hipStream_t st) {
  return g.lo_zero ? launch_x3_p3<ACT, WLOADS, 2>(g, st) : launch_x3_p3<ACT, WLOADS, 3>(g, st);
}

template <int ACT>
int launch_x3_p(Args& g, hipStream_t st) {
  // write-out flavour: 0 = reads nothing, 1 = residual rows only (every launch of the CLIP blocks, SAM's MLP), 2 = a row map
  // (SAM's window un-partition), with or without a residual
  return g.cmap ? launch_x3_p2<ACT, 2>(g, st) : g.R ? launch_x3_p2<ACT, 1>(g, st) : launch_x3_p2<ACT, 0>(g, st);
}

template <int ACT>
int launch_x3(int kind, Args& g, hipStream_t st) {
  return kind == HGL_X3_P ? launch_x3_p<ACT>(g, st) : launch_x3_v1<ACT>(g, st);
}

// HGL_X3_TERMS=3 keeps the third product for fp16-valued weights too (A/B; the results are the same bit for bit)
bool x3_two_terms(const SplitW& sw) {
  const char* v = hgl_env_str("HGL_X3_TERMS");      // read per launch: tests flip it inside one process
  return sw.lo_zero && !(v && v[0] == '3');
}

int x3_gm() {
  static int gmv = -1;
  if (gmv < 0) { gmv = HGL_DIAG_SWITCH("HGL_X3_GM", 8); if (gmv < 1) gmv = 8; }
  return gmv;
}

}  // namespace

int hgl_launch_gemm_f16x3_maps(const void* Ah, const void* Al, int lda, const int* amap, const float* W32, const float* bias,
                               const float* R, int ldr, int rmod, const int* cmap, float* C, void* Ch, void* Cl, int ldc,
                               int M, int N, int K, int act, hipStream_t st) {
  SplitW sw;
  HGL_REQUIRE(find_split((const void*)W32, &sw), "gemm_f16x3: weight %p has no registered fp16 split", (const void*)W32);
  HGL_REQUIRE(sw.N == N && sw.K == K, "gemm_f16x3: registered split is [%d,%d], GEMM wants [%d,%d]", sw.N, sw.K, N, K);
  HGL_REQUIRE(Ah && Al && (C || (Ch && Cl)) && M > 0 && N > 0 && K > 0, "gemm_f16x3: bad arguments");
  HGL_REQUIRE((K % 64) == 0 && (lda & 7) == 0, "gemm_f16x3: K must be a multiple of 64 and lda of 8 (K=%d lda=%d)", K, lda);
  // The LDS-DMA kernel addresses an operand plane with 32-bit byte offsets: a plane of 4 GB or more (CLIP ViT-L/14 fc2 over a
  // group of 16 refs: 526 336 rows x 4096 halves) would fall back to the register-staged kernel at half the rate.  Rows are
  // independent: run it as row chunks that fit (same tiles, same sums per row).
  if (!amap && !cmap && rmod == 0 && (double)M * lda * 2.0 >= 4.0e9 && (double)256 * lda * 2.0 < 2.0e9) {
    const int chunk = (int)(3.9e9 / ((double)lda * 2.0)) / 256 * 256;
    for (int m0 = 0; m0 < M; m0 += chunk) {
      const int mc = M - m0 < chunk ? M - m0 : chunk;
      HGL_TRY(hgl_launch_gemm_f16x3_maps((const _Float16*)Ah + (long long)m0 * lda, (const _Float16*)Al + (long long)m0 * lda, lda,
                                         nullptr, W32, bias, R ? R + (long long)m0 * ldr : nullptr, ldr, 0, nullptr,
                                         C ? C + (long long)m0 * ldc : nullptr, Ch ? (_Float16*)Ch + (long long)m0 * ldc : nullptr,
                                         Cl ? (_Float16*)Cl + (long long)m0 * ldc : nullptr, ldc, mc, N, K, act, st));
    }
    return HGL_OK;
  }
  Args g;
  g.Ah = (const _Float16*)Ah; g.Al = (const _Float16*)Al; g.Wh = sw.hi; g.Wl = sw.lo;
  g.bias = bias; g.R = R; g.C = C; g.Ch = (_Float16*)Ch; g.Cl = (_Float16*)Cl;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = K; g.ldr = ldr; g.ldc = ldc;
  g.rmod = rmod; g.amap = amap; g.cmap = cmap;
  g.rp_p = g.rp_t = 0;
  {
    // The decoder's k | v | q projections add a positional table of rmod rows to P batches of rmod rows: in row order an
    // XCD walks the WHOLE table (4 - 6 MB, more than its L2) once per batch and the table rows were re-fetched for every
    // tile -- as many bytes as the A operand itself (profiles/r05b_decoder_traffic.json: 18.8 MB fetched per prompt for
    // 8.4 MB of operands).  HGL_X3_RPERM=0 keeps the row order.
    static const int rperm = HGL_DIAG_SWITCH("HGL_X3_RPERM", 1);
    if (rperm && R && rmod > 0 && (rmod % 256) == 0 && (M % rmod) == 0 && M / rmod > 1 && !amap && !cmap) {
      g.rp_p = M / rmod;
      g.rp_t = rmod / 256;
    }
  }
  g.part = nullptr; g.ksplit = 1;
  g.out_scale = ldexpf(1.0f, -sw.scale_log2);
  g.lo_zero = x3_two_terms(sw) ? 1 : 0;
  g.gm = x3_gm();
  g.vec4 = x3_vec4_ok(bias, R, ldr, C, Ch, Cl, ldc, N) ? 1 : 0;
  // kernel selection (hgl_gemm_f16x3_select); both tilings accumulate in the same order
  // and give bit-identical results
  if (g_x3_kernel == -2) {
    g_x3_kernel = -1;      // automatic (the cost model); hgl_gemm_f16x3_select() pins a tiling
  }
  // the LDS-DMA kernel addresses the operands with 32-bit byte offsets from the plane bases and writes 16-byte vectors
  const bool small_offsets = (double)M * lda * (amap ? 4.0 : 2.0) < 4.0e9 && (double)N * K * 2.0 < 4.0e9;   // gathered rows: <= 2M
  // a row-modulo residual (positional table) of a few MB stays in L2: it does not cost what a streamed residual costs
  // (decoder k|v|q projection, 262144 x 384 x 256 with a 6 MB table: ping-pong 233 us, register-staged 308 -- the model said 346 / 302)
  const bool streamed_r = R != nullptr && (rmod == 0 || (double)rmod * N * 4.0 > 8.0e6);
  int kind = g_x3_kernel >= 0 ? g_x3_kernel : pick_x3_kernel(M, N, K, streamed_r);
  if (!small_offsets || !g.vec4) kind = HGL_X3_V1;
  // launches that cannot fill the 256 CUs once (GEM at 785 rows, text encoder: a 128x128 tile per CU is latency-bound
  // when run alone) are accounted separately from the throughput-bound ones
  const long long few_tiles = (long long)((M + 127) / 128) * ((N + 127) / 128);
  HglProfScope prof(few_tiles < 256 ? HGL_PROF_GEMM_X3_FEW : kind == HGL_X3_V1 ? HGL_PROF_GEMM_X3 : HGL_PROF_GEMM_X3G, 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N * (R ? 2 : 1)), st);
  switch (act) {
    case HGL_ACT_QUICKGELU: HGL_TRY(launch_x3<HGL_ACT_QUICKGELU>(kind, g, st)); break;
    case HGL_ACT_GELU: HGL_TRY(launch_x3<HGL_ACT_GELU>(kind, g, st)); break;
    case HGL_ACT_RELU: HGL_TRY(launch_x3<HGL_ACT_RELU>(kind, g, st)); break;
    default: HGL_TRY(launch_x3<HGL_ACT_NONE>(kind, g, st)); break;
  }
  return hgl_check_launch("gemm_f16x3");
}

// fp32-A entry for small M (called from hgl_launch_gemm): true when the GEMM was taken
bool hgl_gemm_skinny_applicable(const float* W32, int M, int N, int K, int lda, int ldw, int batch, int max_m) {
  if (g_precision != HGL_PREC_F16X3 || batch != 1 || M > max_m || (K & 15) || (lda & 3) || ldw != K) return false;
  SplitW sw;
  return find_split((const void*)W32, &sw) && sw.N == N && sw.K == K;
}

int hgl_launch_gemm_x3_skinny(const float* A, int lda, const float* W32, const float* bias, const float* R, int ldr,
                              float* C, int ldc, int M, int N, int K, int act, hipStream_t st) {
  SplitW sw;
  HGL_REQUIRE(find_split((const void*)W32, &sw), "gemm_x3_skinny: weight has no registered split");
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

// Split-K on the ping-pong tiling for GEMMs with few output tiles and a long K (SAM's mlp.lin2: 80 tiles,
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
  SplitW sw;
  HGL_REQUIRE(find_split((const void*)W32, &sw), "gemm_f16x3_splitk: weight %p has no registered fp16 split", (const void*)W32);
  HGL_REQUIRE(sw.N == N && sw.K == K && Ah && Al && C && part, "gemm_f16x3_splitk: bad arguments");
  HGL_REQUIRE(ksplit >= 2 && ksplit <= 8 && (K % 64) == 0 && (lda & 7) == 0 && (N & 3) == 0 && (ldc & 3) == 0 && (ldr & 3) == 0,
              "gemm_f16x3_splitk: unsupported shape (K %d, N %d, ksplit %d)", K, N, ksplit);
  HGL_REQUIRE(((K / 32 / ksplit) & ~1) >= 2, "gemm_f16x3_splitk: K too short for %d slices", ksplit);
  HGL_REQUIRE(part_bytes >= (size_t)ksplit * M * N * sizeof(float), "gemm_f16x3_splitk: partial-sum workspace too small");
  HGL_REQUIRE(((size_t)part & 15) == 0, "gemm_f16x3_splitk: partial-sum workspace must be 16-byte aligned");
  HGL_REQUIRE((double)M * lda * (amap ? 4.0 : 2.0) < 4.0e9 && (double)N * K * 2.0 < 4.0e9, "gemm_f16x3_splitk: operand too large");
  Args g;
  g.Ah = (const _Float16*)Ah; g.Al = (const _Float16*)Al; g.Wh = sw.hi; g.Wl = sw.lo;
  g.bias = nullptr; g.R = nullptr; g.C = nullptr; g.Ch = nullptr; g.Cl = nullptr;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = K; g.ldr = 0; g.ldc = N;
  g.rmod = 0; g.rp_p = g.rp_t = 0; g.amap = amap; g.cmap = nullptr; g.part = part; g.ksplit = ksplit; g.vec4 = 1;
  g.out_scale = ldexpf(1.0f, -sw.scale_log2);
  g.lo_zero = x3_two_terms(sw) ? 1 : 0;
  g.gm = 8;
  {
    const long long few_tiles = (long long)((M + 127) / 128) * ((N + 127) / 128);
    HglProfScope prof(few_tiles < 256 ? HGL_PROF_GEMM_X3_FEW : HGL_PROF_GEMM_X3G, 2.0 * M * (double)N * K,
                      4.0 * ((double)M * K + (double)N * K + (double)M * N * (R ? 2 : 1)), st);
    HGL_TRY(launch_x3_p<HGL_ACT_NONE>(g, st));
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

// A GEMM whose LAST round of the persistent 256 x 256 tiling is mostly empty (CLIP's out / fc2 over a group of 16 refs: 2364
// tiles on 256 CUs = 9.23 rounds, i.e. ten, the last one with 60 tiles; SAM's proj / lin2 over 10 images: 3.125 -> four) pays a
// whole round for the remainder.  Rows are independent, so the GEMM is cut BY ROWS into a main part that fills whole rounds and
// a tail of a few row tiles that runs split-K over all CUs (the existing split-K path: K slices as grid.y, partial sums through
// `part`, summed in index order with bias / activation / residual by splitk_reduce_kernel: deterministic).  A tail row's sum
// is associated differently from a main row's (K slices), as any split-K row's is: equal to fp32 rounding, not bit for bit
// -- which rows form the tail depends on M alone.  Falls through to the plain launch when the plan does not pay.
bool x3_tail_plan(int M, int N, int K, bool has_r, size_t part_bytes, int* m_main, int* ksplit) {
  if ((N & 3) || K < 1024 || (K % 64)) return false;      // (K = 768, CLIP's out-projection: measured no gain -- six-K-tile slices cost what the round costs)
  const int ncu = x3_num_cus();
  const long long tiles_n = (N + 255) / 256, tiles_m = (M + 255) / 256, tiles = tiles_m * tiles_n;
  const long long rounds = (tiles + ncu - 1) / ncu;
  if (rounds < 3) return false;
  const long long rem = tiles - (rounds - 1) * ncu;
  if (rem * 2 > ncu) return false;                       // the last round is at least half full: leave it
  const long long main_rows = ((rounds - 1) * ncu) / tiles_n;      // row tiles that fit in rounds - 1 rounds
  const long long tail_tiles = (tiles_m - main_rows) * tiles_n;
  if (main_rows <= 0 || tail_tiles <= 0 || tail_tiles * 2 > ncu) return false;
  const int nk = K / 32;
  int ks = (int)(ncu / tail_tiles);
  if (ks > 8) ks = 8;
  while (ks > 1 && ((nk / ks) & ~1) < 12) --ks;          // every slice at least twelve K tiles (the DMA prologue alone is seven units)
  if (ks < 2) return false;
  const long long m_tail = (long long)M - main_rows * 256;
  if ((size_t)ks * (size_t)m_tail * (size_t)N * sizeof(float) > part_bytes) return false;
  // cost model of pick_x3_kernel: one more round against a split-K launch (its slices, the reduce pass, two launch gaps)
  const double t_round = 14.0 + 2.08 * nk + (has_r ? 12.0 : 0.0);
  const double t_tail = 14.0 + 2.08 * ((nk / ks) & ~1) + 22.0;
  if (t_tail > 0.8 * t_round) return false;
  *m_main = (int)(main_rows * 256);
  *ksplit = ks;
  return true;
}

int hgl_launch_gemm_f16x3_balanced(const void* Ah, const void* Al, int lda, const int* amap, const float* W32, const float* bias,
                                   const float* R, int ldr, const int* cmap, float* C, int ldc, int M, int N, int K, int act,
                                   float* part, size_t part_bytes, hipStream_t st) {
  int m_main = 0, ks = 1;
  static const int on = HGL_DIAG_SWITCH("HGL_X3_TAIL", 1);
  const bool p_kernel = g_x3_kernel < 0 ? pick_x3_kernel(M, N, K, R != nullptr) == HGL_X3_P : g_x3_kernel == HGL_X3_P;
  if (!on || !C || !part || !p_kernel || (((size_t)part) & 15) || !x3_tail_plan(M, N, K, R != nullptr, part_bytes, &m_main, &ks))
    return hgl_launch_gemm_f16x3_maps(Ah, Al, lda, amap, W32, bias, R, ldr, 0, cmap, C, nullptr, nullptr, ldc, M, N, K, act, st);
  HGL_TRY(hgl_launch_gemm_f16x3_maps(Ah, Al, lda, amap, W32, bias, R, ldr, 0, cmap, C, nullptr, nullptr, ldc, m_main, N, K, act, st));
  const int m_tail = M - m_main;
  // the tail's rows: through the maps when there are any (their entries are absolute rows), by pointer offset otherwise
  const _Float16* th = (const _Float16*)Ah + (amap ? 0 : (long long)m_main * lda);
  const _Float16* tl = (const _Float16*)Al + (amap ? 0 : (long long)m_main * lda);
  const float* tr = R ? R + (cmap ? 0 : (long long)m_main * ldr) : nullptr;
  float* tc = C + (cmap ? 0 : (long long)m_main * ldc);
  return hgl_launch_gemm_f16x3_splitk(th, tl, lda, amap ? amap + m_main : nullptr, W32, bias, tr, ldr, cmap ? cmap + m_main : nullptr,
                                      tc, ldc, m_tail, N, K, act, ks, part, part_bytes, st);
}

extern "C" {

int hgl_set_precision(int mode) {
  HGL_REQUIRE(mode == HGL_PREC_F32 || mode == HGL_PREC_F16X3, "set_precision: unknown mode %d", mode);
  g_precision = mode;
  return HGL_OK;
}

int hgl_get_precision(void) { return g_precision; }

int hgl_gemm_f16x3_select(int kind) {
  HGL_REQUIRE(kind >= -1 && kind <= HGL_X3_P, "gemm_f16x3_select: unknown kernel %d", kind);
  g_x3_kernel = kind;
  return HGL_OK;
}

int hgl_register_split_weight(const float* w_fp32, int N, int K, int scale_log2, void* hi, void* lo, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(w_fp32 && hi && lo && N > 0 && K > 0 && (K & 7) == 0, "register_split_weight: bad arguments (K %% 8)");
  HGL_REQUIRE(scale_log2 >= -24 && scale_log2 <= 24, "register_split_weight: scale_log2 out of range");
  HGL_TRY(hgl_launch_split_f16(w_fp32, ldexpf(1.0f, scale_log2), hi, lo, (long long)N * K, (hipStream_t)stream));
  // fp16-valued weights (the OpenAI CLIP archives store fp16; clip/model.py:509 is commented out in the reference, so the
  // model holds those values as fp32): every lo half is zero and the GEMMs drop the A_hi * W_lo products.  One read-back
  // per weight at model construction.
  unsigned nz = 1;
  {
    // ONE device-global flag: reset, kernel and read-back of two registrations (two models built on two host threads /
    // streams) must not interleave -- A's flag zeroed by B between A's kernel and A's read-back would record a genuine fp32
    // weight as fp16-valued and silently drop its A_hi * W_lo products.  Serialised here (construction time only).
    static std::mutex flag_mu;
    std::lock_guard<std::mutex> flag_lk(flag_mu);
    const unsigned zero = 0;
    HGL_REQUIRE(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_lo_nonzero), &zero, sizeof(zero), 0, hipMemcpyHostToDevice, (hipStream_t)stream) == hipSuccess,
                "register_split_weight: flag reset failed");
    const long long n8 = (long long)N * K / 8;
    hipLaunchKernelGGL(lo_nonzero_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)lo, n8);
    HGL_REQUIRE(hipMemcpyFromSymbolAsync(&nz, HIP_SYMBOL(g_lo_nonzero), sizeof(nz), 0, hipMemcpyDeviceToHost, (hipStream_t)stream) == hipSuccess &&
                hipStreamSynchronize((hipStream_t)stream) == hipSuccess, "register_split_weight: flag read-back failed");
  }
  std::lock_guard<std::mutex> lk(g_split_mu);
  g_split[(const void*)w_fp32] = SplitW{(const _Float16*)hi, (const _Float16*)lo, scale_log2, N, K, nz == 0};
  return HGL_OK;
}

int hgl_split_weight_is_fp16_valued(const float* w_fp32) {
  SplitW sw;
  if (!find_split((const void*)w_fp32, &sw)) return -1;
  return sw.lo_zero ? 1 : 0;
}

int hgl_unregister_split_weight(const float* w_fp32) {
  std::lock_guard<std::mutex> lk(g_split_mu);
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
  // scratch beyond the split A: room for the partial sums of the row-balanced launch (whole rounds + split-K tail), which the
  // model code uses for its residual GEMMs; without it, the plain launch
  const size_t a_bytes = hgl_align_up((size_t)M * K * 4, 256);
  if (scratch_bytes > a_bytes + 256)
    return hgl_launch_gemm_f16x3_balanced(ah, al, K, nullptr, W, bias, R, N, nullptr, C, N, M, N, K, act,
                                          (float*)((char*)scratch + a_bytes), scratch_bytes - a_bytes, st);
  return hgl_launch_gemm_f16x3(ah, al, K, W, bias, R, N, C, nullptr, nullptr, N, M, N, K, act, st);
}

}  // extern "C"

HGL_DEFINE_SPLIT_OVERFLOW_READER(hgl_split_overflow_gemm)
