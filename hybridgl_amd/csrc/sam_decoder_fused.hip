// Fused stages of the SAM mask decoder (split-fp16 matrix-core mode).
//
// The decoder's image side works on [P prompts x 4096 tokens x 256 channels] (268 MB in fp32 for 64 prompts): as
// separate GEMM / LayerNorm / activation launches every stage streams that tensor (or a wider one) through HBM, and the
// decoder runs at the memory rate (2.9 ms per 64 prompts for 0.23 TFLOP).  Every stage of the OUTPUT UPSCALING is
// row-wise -- a ConvTranspose2d(k=2, s=2) is a per-pixel GEMM -- so a tile of rows can go through the whole chain on chip:
//
//   dec_tail_kernel      upscaled = output_upscaling(src)          mask_decoder.py:132-134, :75-81
//                          ConvTranspose2d(256 -> 64, k2 s2)   = rows x [4 pos x 64]    (MFMA, K = 256)
//                          LayerNorm2d(64) + GELU                 per (row, pos)        (common.py:33-43)
//                          ConvTranspose2d(64 -> 32, k2 s2) + GELU = rows x [4 sub x 32] per pos (MFMA, K = 64)
//                        masks = hyper_in @ upscaled               mask_decoder.py:146-151 (multimask rows 1..3)
//                      reads the 256-channel rows once (fp16 hi + lo planes), writes the [P,3,4g,4g] logits: 0.32 GB instead
//                      of 2.6 GB through four launches.
//
//   dec_i2t_kernel       keys' = norm4(keys + out_proj(attention of the image tokens over the 7 prompt tokens))
//                                                                   transformer.py:139-150; described at the kernel
//
// The matrix products are those of the unfused path, operation for operation (the same split-fp16 products in the same
// order: K steps of 32; per step A_lo*W_hi, A_hi*W_lo, A_hi*W_hi into one fp32 accumulator) and so are the epilogue
// expressions; the LayerNorm sums and the hyper-network dot products are associated differently (per lane first, then
// across the four lanes that share a row: a first version reproduced the unfused kernels' butterfly order bit for bit and
// spent a quarter of its time in 256 dependent ds_bpermute + wait pairs per wave).  Fused and unfused logits agree to
// fp32 rounding (tests/test_gpu_sam.py).  Round 5: eight waves per tile instead of four (see dec_tail_kernel), the epilogue
// vectors through LDS, no register spilled: 2.93 -> 2.36 ms per 529 prompts; the vector pipe is then ~55 % and the matrix pipe
// ~23 % busy per SIMD with 3.8 waves resident (profiles/r05c_sq_counters_dec_tail_*.json): the rest is the six barriers of a tile
// and the L2 round trips of the weight fragments.  (The GELU polynomial two values per instruction -- v_pk_fma_f32 -- was built and
// measured: a quarter fewer vector instructions, 1.5 % SLOWER: the packed fp32 forms issue at half rate on gfx950.)
#include "hgl_common.h"

bool hgl_get_split_weight(const float* W, const void** hi, const void** lo, int* scale_log2, int* N, int* K);

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct TailArgs {
  const _Float16 *Ah, *Al;     // src rows [P*HW, 256] as fp16 hi / lo planes
  const _Float16 *W0h, *W0l;   // up0 weight [256 = (ky,kx,c64), 256] split
  const _Float16 *W3h, *W3l;   // up3 weight [128 = (ky2,kx2,c32), 64] split
  const float *b0, *ln_w, *ln_b, *b3;
  const float* hyper;          // [P, 4, 32]
  float s0, s3, eps;           // weight scales undone in the epilogues; LayerNorm2d eps
  int g, HW;
  int hrow0;                   // first of the three hyper-network rows multiplied (1: multimask tokens 1..3; 0: tokens 0..2)
  float* out;                  // [P, 3, 4g, 4g]
  const uint8_t* skip;         // optional [P]: prompts whose rows of `out` nobody will read (IoU gate): their workgroups leave at once
};

// LDS image of an operand tile for the 16-row x 4-chunk fragment reads of v_mfma_f32_16x16x32_f16 (a lane reads row r,
// 16-byte chunk h of a 32-deep K step): blocks of [16 rows x 64 B] per (K step, row block), chunk ^= {0,2,3,1}[(row >> 2) & 3]
// (the swizzle of the ping-pong GEMM: the 16 lanes of every ds_read_b128 lane group land on 16 different 16-byte slots).
__device__ __forceinline__ unsigned frag_off(int kstep, int nmb, int mb, int r, int chunk) {
  const int sw = (0x78 >> (2 * ((r >> 2) & 3))) & 3;
  return (unsigned)((((kstep * nmb + mb) * 16 + r) * 64) + ((chunk ^ sw) * 16));
}

__device__ __forceinline__ float gelu_erf(float x) { return hgl_gelu_erf(x); }

constexpr int TAIL_ROWS = 64;
constexpr int TAIL_A_PLANE = TAIL_ROWS * 256 * 2;        // bytes of one plane of the source tile (32 KiB)
constexpr int TAIL_G_PLANE = TAIL_ROWS * 64 * 2;         // bytes of one plane of a position's 64 x 64 patch (8 KiB)
constexpr int TAIL_STAGE = 2 * TAIL_A_PLANE;             // output staging [3][64 pixels][16] floats behind the tile
constexpr int TAIL_RED = TAIL_STAGE;                     // LayerNorm partial sums [2 passes][4 positions][2 halves][64 rows] (4 KiB,
                                                         // dead before the staging area that overlays them is written)
constexpr int TAIL_CST = TAIL_STAGE + 3 * TAIL_ROWS * 16 * 4;   // b0[256] ln_w[64] ln_b[64] b3[128] hyper[3][32]: 608 floats
constexpr int TAIL_LDS = TAIL_CST + 608 * 4;             // 78.4 KiB: two workgroups per CU

// Round 5: EIGHT waves per 64-row tile, <= 128 registers, so that sixteen waves (four per SIMD) are resident per CU.  The
// four-wave version (one wave per position, 64 channels each, 247 registers, two waves per SIMD) spent 47 % of its wave
// cycles waiting and kept the vector pipe 53 % busy (profiles/r05c_sq_counters_dec_tail.json): two waves per SIMD cannot
// cover the L2 round trips of the weight fragments, the tile's HBM load and the barriers.  Wave (pos, half):
//   first product   position pos, channels 32 half .. 32 half + 31 of its 64 (half of that position's weight rows: the
//                   weight traffic from L2 stays 256 KiB per tile); the LayerNorm sums of a (row, position) are completed
//                   through LDS between the two waves that share it (two passes: mean, then centred squares, as before)
//   second product  position pos, sub-positions 2 half and 2 half + 1 (64 of the 128 output columns), all 64 rows
__global__ __launch_bounds__(512, 4) void dec_tail_kernel(TailArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 15, h = lane >> 4;
  const int pos = wave & 3, half = wave >> 2;
  const int p = blockIdx.y, tile = blockIdx.x;
  if (a.skip != nullptr && a.skip[p]) return;      // (uniform over the workgroup, before any barrier)
  const long long row0 = (long long)p * a.HW + (long long)tile * TAIL_ROWS;

  // ---- the tile's 64 rows x 256 channels, both planes, into the fragment image ----
  // the first two K steps of this wave's W fragments are requested before the tile itself
  const _Float16* const wh = a.W0h + (long long)(pos * 64 + 32 * half + r) * 256 + 8 * h;
  const _Float16* const wl = a.W0l + (long long)(pos * 64 + 32 * half + r) * 256 + 8 * h;
  f16x8 bh[3][2], bl[3][2];
#pragma unroll
  for (int pre = 0; pre < 2; ++pre)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bh[pre][j] = *(const f16x8*)(wh + j * 16 * 256 + pre * 32);
      bl[pre][j] = *(const f16x8*)(wl + j * 16 * 256 + pre * 32);
    }
  // the epilogues' vectors go through LDS with the tile (one exposed round trip to L2 instead of four, and no registers
  // held across the products)
  float* const cst = (float*)(smem + TAIL_CST);
  if (t < 152) {
    const float* src = t < 64 ? a.b0 + 4 * t : t < 80 ? a.ln_w + 4 * (t - 64) : t < 96 ? a.ln_b + 4 * (t - 80)
                     : t < 128 ? a.b3 + 4 * (t - 96) : a.hyper + (long long)p * 4 * 32 + 32 * a.hrow0 + 4 * (t - 128);
    *(f32x4*)(cst + 4 * t) = *(const f32x4*)src;
  }
  // four consecutive threads fetch the 64 contiguous bytes of one (row, K step), the next four the next ROW: consecutive
  // K steps of a row lie 4 KiB apart in the image (same banks: an 8-way conflict on the 16-byte stores when threads walk along a row)
#pragma unroll
  for (int i = t; i < TAIL_ROWS * 32; i += 512) {
    const int row = (i >> 2) & 63, kc = 4 * (i >> 8) + (i & 3);
    const unsigned off = frag_off(kc >> 2, 4, row >> 4, row & 15, kc & 3);
    *(u32x4*)(smem + off) = *(const u32x4*)(a.Ah + (row0 + row) * 256 + kc * 8);
    *(u32x4*)(smem + TAIL_A_PLANE + off) = *(const u32x4*)(a.Al + (row0 + row) * 256 + kc * 8);
  }
  __syncthreads();

  // ---- ConvTranspose2d(256 -> 64, k2 s2), position pos = (ky, kx): rows x this wave's 32 channels ----
  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    // W fragments come straight from L2 (the 256 KiB of up0's halves stay resident), two K steps ahead of their use
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int cb = ks % 3, nb = (ks + 2) % 3;
      if (ks + 2 < 8) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bh[nb][j] = *(const f16x8*)(wh + j * 16 * 256 + (ks + 2) * 32);
          bl[nb][j] = *(const f16x8*)(wl + j * 16 * 256 + (ks + 2) * 32);
        }
      }
      f16x8 ah[4], al[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned off = frag_off(ks, 4, i, r, h);
        ah[i] = *(const f16x8*)(smem + off);
        al[i] = *(const f16x8*)(smem + TAIL_A_PLANE + off);
      }
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const f16x8 av = term == 0 ? al[i] : ah[i];
            const f16x8 bv = term == 1 ? bl[cb][j] : bh[cb][j];
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bv, av, acc[i][j], 0, 0, 0);
          }
    }
  }
  __syncthreads();   // every wave has read the source tile: its LDS becomes the positions' 64 x 64 patches

  // ---- + bias, LayerNorm2d over the 64 channels of (row, position), GELU, split, into the position's patch ----
  unsigned char* const patch = smem + pos * 2 * TAIL_G_PLANE;
  float* const red = (float*)(smem + TAIL_RED);
  {
    f32x4 b0v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) b0v[j] = *(const f32x4*)(cst + pos * 64 + 32 * half + 16 * j + 4 * h);
    // sum over this wave's 32 channels of a row: this lane's 8, then the four lanes (h) that share the row; the other 32
    // come from the partner wave through LDS (both waves add the two partial sums in the same order)
    auto tree = [&](const f32x4 (&x)[2]) {
      float t = ((x[0][0] + x[0][1]) + (x[0][2] + x[0][3])) + ((x[1][0] + x[1][1]) + (x[1][2] + x[1][3]));
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      return t;
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = acc[i][j][e] * a.s0 + b0v[j][e];
      const float s = tree(acc[i]);
      if (h == 0) red[((0 * 4 + pos) * 2 + half) * TAIL_ROWS + 16 * i + r] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* rp = red + (0 * 4 + pos) * 2 * TAIL_ROWS + 16 * i + r;
      const float mean = (rp[0] + rp[TAIL_ROWS]) * (1.f / 64.f);
      f32x4 q[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[i][j][e] -= mean; q[j][e] = acc[i][j][e] * acc[i][j][e]; }
      const float s = tree(q);
      if (h == 0) red[((1 * 4 + pos) * 2 + half) * TAIL_ROWS + 16 * i + r] = s;
    }
    __syncthreads();
    f32x4 lw[2], lb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      lw[j] = *(const f32x4*)(cst + 256 + 32 * half + 16 * j + 4 * h);
      lb[j] = *(const f32x4*)(cst + 320 + 32 * half + 16 * j + 4 * h);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* rp = red + (1 * 4 + pos) * 2 * TAIL_ROWS + 16 * i + r;
      const float var = (rp[0] + rp[TAIL_ROWS]) * (1.f / 64.f);
      const float rs = rsqrtf(var + a.eps);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f16x4 hi4, lo4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float o = gelu_erf(acc[i][j][e] * rs * lw[j][e] + lb[j][e]);
          _Float16 hh, ll;
          hgl_split_hi_lo(o, hh, ll);
          hi4[e] = hh;
          lo4[e] = ll;
        }
        // channel c0 = 32 half + 16 j + 4 h: K step c0 / 32 (= half), chunk (c0 % 32) / 8, half-chunk (c0 % 8) / 4
        const int c0 = 32 * half + 16 * j + 4 * h;
        const unsigned off = frag_off(c0 >> 5, 4, i, r, (c0 & 31) >> 3) + (unsigned)((c0 & 7) * 2);
        *(f16x4*)(patch + off) = hi4;
        *(f16x4*)(patch + TAIL_G_PLANE + off) = lo4;
      }
    }
  }
  __syncthreads();   // both K steps of every position's patch are written

  // ---- ConvTranspose2d(64 -> 32, k2 s2) + GELU, then the three hyper-network dot products: this wave's pair of
  //      sub-positions, one at a time (32 output columns: the weight fragments of the next one are in flight during the
  //      epilogue of the current one) ----
  float* const stage = (float*)(smem + TAIL_STAGE);
  {
    const int pass = half;
    // the weight fragments of the first sub-position (requested here: held across the LayerNorm they cost 15 spilled registers,
    // whose scratch traffic showed as 2.4 MB written per prompt)
    const _Float16* const w3h = a.W3h + (long long)(half * 64 + r) * 64 + 8 * h;
    const _Float16* const w3l = a.W3l + (long long)(half * 64 + r) * 64 + 8 * h;
    f16x8 qh[2][2][2], ql[2][2][2];      // [sub-position][K step][column block]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        qh[0][ks][j] = *(const f16x8*)(w3h + j * 16 * 64 + ks * 32);
        ql[0][ks][j] = *(const f16x8*)(w3l + j * 16 * 64 + ks * 32);
      }
    // columns of this pass: n = 64 pass + 32 sp + 16 jj + 4 h + e = sub-position 2 pass + sp, channel 16 jj + 4 h + e.
    // part[sp][i][m]: this lane's share (8 of the 32 channels) of the dot product of row 16 i + r, sub-position 2 pass + sp
    float part[2][4][3];
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      f32x4 c2[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) c2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (sp == 1) {      // (held across the first sub-position's epilogue these fragments spill: 16 - 128 B of scratch per lane;
                          //  the fence keeps the compiler from hoisting the loads there by itself)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            qh[1][ks][j] = *(const f16x8*)(w3h + (32 + j * 16) * 64 + ks * 32);
            ql[1][ks][j] = *(const f16x8*)(w3l + (32 + j * 16) * 64 + ks * 32);
          }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned off = frag_off(ks, 4, i, r, h);
          const f16x8 gh = *(const f16x8*)(patch + off);
          const f16x8 gl = *(const f16x8*)(patch + TAIL_G_PLANE + off);
#pragma unroll
          for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const f16x8 av = term == 0 ? gl : gh;
              const f16x8 bv = term == 1 ? ql[sp][ks][j] : qh[sp][ks][j];
              c2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bv, av, c2[i][j], 0, 0, 0);
            }
        }
      }
      // (the hyper-network rows are re-read from LDS per row block: holding them would not fit beside the fragments in flight)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float d3[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const f32x4 c = c2[i][jj];
          const f32x4 b3v1 = *(const f32x4*)(cst + 384 + pass * 64 + 32 * sp + 16 * jj + 4 * h);
          f32x4 hv1[3];
#pragma unroll
          for (int m = 0; m < 3; ++m) hv1[m] = *(const f32x4*)(cst + 512 + 32 * m + 16 * jj + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float u = gelu_erf(c[e] * a.s3 + b3v1[e]);
#pragma unroll
            for (int m = 0; m < 3; ++m) d3[m] = fmaf(u, hv1[m][e], d3[m]);
          }
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) part[sp][i][m] = d3[m];
      }
    }
    // Sum over the four lanes (h) of a row as a reduce-scatter: lanes h and h ^ 2 trade the sub-position they do not keep
    // (12 values), then lanes h and h ^ 1 trade the two row blocks they do not keep (6 values): 18 shuffles instead of 48,
    // and a lane ends up with six finished logits: sub-position h >> 1, row blocks 2 (h & 1) and 2 (h & 1) + 1, three masks.
    const bool up2 = (h & 2) != 0, up1 = (h & 1) != 0;
    float k12[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const float send = up2 ? part[0][i][m] : part[1][i][m];
        const float keep = up2 ? part[1][i][m] : part[0][i][m];
        k12[i][m] = keep + __shfl_xor(send, 32);
      }
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const float send = up1 ? k12[ii][m] : k12[2 + ii][m];
        const float keep = up1 ? k12[2 + ii][m] : k12[ii][m];
        const float res = keep + __shfl_xor(send, 16);
        const int i = 2 * (h & 1) + ii;
        const int sub = 2 * pass + (h >> 1);                   // (ky2, kx2)
        const int dy = 2 * (pos >> 1) + (sub >> 1), dx = 2 * (pos & 1) + (sub & 1);
        stage[(m * TAIL_ROWS + 16 * i + r) * 16 + dy * 4 + dx] = res;
      }
  }
  __syncthreads();

  // ---- the tile's 3 x 64 x 16 logits to the [P, 3, 4g, 4g] layout, rows of 4 * min(g, 64) contiguous floats ----
  // (the launcher admits g = a power of two <= 64 or a multiple of 64: the row length and the rows per tile are powers of two)
  const int g = a.g, S4 = 4 * g;
  const int gw = g < TAIL_ROWS ? g : TAIL_ROWS;          // pixels of one grid row inside the tile
  const int lrow = 31 - __clz(4 * gw);                   // log2 of the floats of one output row of the tile
  const int lnr = 31 - __clz(TAIL_ROWS / gw);            // log2 of the grid rows the tile covers
  const int pix0 = tile * TAIL_ROWS, y0 = pix0 / g, x0 = pix0 - y0 * g;
#pragma unroll
  for (int o = t; o < 3 * TAIL_ROWS * 16; o += 512) {
    const int xx = o & ((1 << lrow) - 1);
    int q = o >> lrow;
    const int dy = q & 3;
    q >>= 2;
    const int yl = q & ((1 << lnr) - 1), m = q >> lnr;
    const int xl = xx >> 2, dx = xx & 3;
    const float v = stage[(m * TAIL_ROWS + yl * gw + xl) * 16 + dy * 4 + dx];
    a.out[(((long long)p * 3 + m) * S4 + 4 * (y0 + yl) + dy) * S4 + 4 * (x0 + xl) + dx] = v;
  }
}


// ---------------------------------------------------------------------------------------------------------------------
//   dec_i2t_kernel       keys' = norm4(keys + out_proj(softmax(q k^T / 4) v))     transformer.py:139-150 (step 4 of a
//                        TwoWayAttentionBlock: the image tokens attend to the 7 prompt tokens), per tile of 64 image tokens:
//                          attention of (token, head) against 7 keys of 16 channels          VALU, the arithmetic of attn_smallk
//                          out-projection 128 -> 256 (+ bias, + residual)                   MFMA, K = 128, W fragments from L2
//                          LayerNorm over the 256 channels                                   lane sums + one LDS exchange per pass
//                        unfused: attention (reads q, writes 134 MB), GEMM (reads them, reads / writes 2 x 268 MB), LayerNorm
//                        (reads 268, writes 268 / 536 MB).  Fused: reads q (and the per-prompt residual), writes the split planes
//                        (and, while a later layer needs them as a residual, the fp32 rows).
struct I2TArgs {
  const float* q;             // image-side queries: row (p, n) at q + p * sqb + n * ldq, 128 floats (8 heads x 16)
  long long sqb;
  int ldq;
  const float *k1, *v1;       // projected prompt tokens [P, 7, 128]
  const _Float16 *Wh, *Wl;    // out_proj weight [256, 128] split
  const float* bo;
  const float* R;             // residual rows of 256 floats: R + p * srb + n * 256   (srb = 0: the same rows for every prompt)
  long long srb;
  const float *ln_w, *ln_b;
  float so, eps, scale;
  float* out32;               // [P*HW, 256] fp32 rows or nullptr
  _Float16 *oh, *ol;          // [P*HW, 256] fp16 hi / lo planes
  int HW;
};

constexpr int I2T_ROWS = 64;
constexpr int I2T_TOK = 7;
constexpr int I2T_HS = 20;                                  // floats between the heads of a token in LDS (16 + 4 of padding)
constexpr int I2T_A_PLANE = I2T_ROWS * 128 * 2;          // bytes of one plane of the attention output tile (16 KiB)

__global__ __launch_bounds__(512, 2) void dec_i2t_kernel(I2TArgs a) {
  // Eight waves per 64-token tile (a wave owns 32 of the 256 output columns): 101 registers, 16 waves per CU.  (Four waves of 64
  // columns each needed 198 registers, 8 waves per CU, and took the same time: the launch is bound by its ~1100 VALU
  // instructions per wave and by the store pattern, see the staging at the end, not by latency.)
  __shared__ __attribute__((aligned(16))) unsigned char img[2 * I2T_A_PLANE];
  __shared__ __attribute__((aligned(16))) float kv_s[2 * I2T_TOK * 8 * I2T_HS];
  __shared__ __attribute__((aligned(16))) float red[2][I2T_ROWS][8];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 15, h = lane >> 4;
  const int p = blockIdx.y, tile = blockIdx.x;
  const int n0 = tile * I2T_ROWS;
  const long long row0 = (long long)p * a.HW + n0;

  // the first two K steps of this wave's W fragments are requested before anything else
  const _Float16* const wh = a.Wh + (long long)(wave * 32 + r) * 128 + 8 * h;
  const _Float16* const wl = a.Wl + (long long)(wave * 32 + r) * 128 + 8 * h;
  f16x8 bh[2][2], bl[2][2];
#pragma unroll
  for (int pre = 0; pre < 2; ++pre)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bh[pre][j] = *(const f16x8*)(wh + j * 16 * 128 + pre * 32);
      bl[pre][j] = *(const f16x8*)(wl + j * 16 * 128 + pre * 32);
    }

  // ---- the prompt's 7 projected tokens (keys, values): a head's 16 floats at a stride of 20, so that the eight heads a
  //      ds_read_b128 of the attention touches lie in eight different bank groups ----
  if (t < I2T_TOK * 32) {
    const int j = t >> 5, c = t & 31;                    // token, 16-byte piece of its 128 floats
    const int dst = (j * 8 + (c >> 2)) * I2T_HS + 4 * (c & 3);
    *(f32x4*)(kv_s + dst) = ((const f32x4*)(a.k1 + (long long)p * I2T_TOK * 128))[t];
    *(f32x4*)(kv_s + I2T_TOK * 8 * I2T_HS + dst) = ((const f32x4*)(a.v1 + (long long)p * I2T_TOK * 128))[t];
  }
  // q of this thread's (token, head): eight consecutive threads read the 512 contiguous bytes of one token
  const int hh = t & 7, arow = t >> 3;
  f32x4 qv[4];
  {
    const float* qp = a.q + p * a.sqb + (long long)(n0 + arow) * a.ldq + hh * 16;
#pragma unroll
    for (int c = 0; c < 4; ++c) qv[c] = *(const f32x4*)(qp + 4 * c);
  }
  __syncthreads();

  // ---- softmax(q k^T * scale) v over the 7 tokens (the arithmetic of attn_smallk_kernel); the result as fp16 hi + lo
  //      into the fragment image ----
  {
    float sc[I2T_TOK];
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < I2T_TOK; ++j) {
      const float* kr = kv_s + (j * 8 + hh) * I2T_HS;
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 kk = *(const f32x4*)(kr + 4 * c);
        d += qv[c][0] * kk[0]; d += qv[c][1] * kk[1]; d += qv[c][2] * kk[2]; d += qv[c][3] * kk[3];
      }
      sc[j] = d * a.scale;
      mx = fmaxf(mx, sc[j]);
    }
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < I2T_TOK; ++j) {
      sc[j] = expf(sc[j] - mx);
      l += sc[j];
    }
    const float inv = 1.0f / l;
    f32x4 o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < I2T_TOK; ++j) {
      const float* vr = kv_s + I2T_TOK * 8 * I2T_HS + (j * 8 + hh) * I2T_HS;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 vv = *(const f32x4*)(vr + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[c][e] += sc[j] * vv[e];
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f16x4 hi4, lo4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        _Float16 x, y;
        hgl_split_hi_lo(o[c][e] * inv, x, y);
        hi4[e] = x;
        lo4[e] = y;
      }
      const int c0 = 16 * hh + 4 * c;       // channel: K step c0 / 32, chunk (c0 % 32) / 8, half-chunk (c0 % 8) / 4
      const unsigned off = frag_off(c0 >> 5, 4, arow >> 4, arow & 15, (c0 & 31) >> 3) + (unsigned)((c0 & 7) * 2);
      *(f16x4*)(img + off) = hi4;
      *(f16x4*)(img + I2T_A_PLANE + off) = lo4;
    }
  }
  __syncthreads();

  // ---- out-projection: wave w computes columns 32 w .. 32 w + 31 of the tile's 64 rows ----
  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int cb = ks & 1;
    f16x8 ah[4], al[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned off = frag_off(ks, 4, i, r, h);
      ah[i] = *(const f16x8*)(img + off);
      al[i] = *(const f16x8*)(img + I2T_A_PLANE + off);
    }
#pragma unroll
    for (int term = 0; term < 3; ++term)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f16x8 av = term == 0 ? al[i] : ah[i];
          const f16x8 bv = term == 1 ? bl[cb][j] : bh[cb][j];
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bv, av, acc[i][j], 0, 0, 0);
        }
    if (ks + 2 < 4) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[cb][j] = *(const f16x8*)(wh + j * 16 * 128 + (ks + 2) * 32);
        bl[cb][j] = *(const f16x8*)(wl + j * 16 * 128 + (ks + 2) * 32);
      }
    }
  }

  // ---- + bias + residual; LayerNorm over the 256 channels of a row: lane sums (8 channels), the four lanes of the row,
  //      then the eight waves through LDS ----
  const int colw = wave * 32 + 4 * h;           // + 16 j + e
  f32x4 bov[2], lw[2], lb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    bov[j] = *(const f32x4*)(a.bo + colw + 16 * j);
    lw[j] = *(const f32x4*)(a.ln_w + colw + 16 * j);
    lb[j] = *(const f32x4*)(a.ln_b + colw + 16 * j);
  }
  const float* Rp = a.R + p * a.srb + (long long)n0 * 256 + colw;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 rv = *(const f32x4*)(Rp + (long long)(16 * i + r) * 256 + 16 * j);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = (acc[i][j][e] * a.so + bov[j][e]) + rv[e];
    }
  auto row_sum = [&](const f32x4 (&x)[2]) {
    float u = ((x[0][0] + x[0][1]) + (x[0][2] + x[0][3])) + ((x[1][0] + x[1][1]) + (x[1][2] + x[1][3]));
    u += __shfl_xor(u, 16);
    u += __shfl_xor(u, 32);
    return u;
  };
  auto tile_sum = [&](const float* s8) {
    const f32x4 s = *(const f32x4*)s8, u = *(const f32x4*)(s8 + 4);
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((u[0] + u[1]) + (u[2] + u[3]));
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float u = row_sum(acc[i]);
    if (h == 0) red[0][16 * i + r][wave] = u;
  }
  __syncthreads();
  float mean[4], rstd[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    mean[i] = tile_sum(red[0][16 * i + r]) * (1.f / 256.f);
    f32x4 q[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = acc[i][j][e] - mean[i]; q[j][e] = d * d; }
    const float u = row_sum(q);
    if (h == 0) red[1][16 * i + r][wave] = u;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) rstd[i] = rsqrtf(tile_sum(red[1][16 * i + r]) * (1.f / 256.f) + a.eps);
  // ---- normalise, split, and write 16 rows at a time through LDS (the fragment image is free): a lane holds 4-channel
  //      pieces of 16 different rows -- stored directly, an instruction scatters 16 x 32 / 64 bytes and the launch ran at
  //      the rate of those partial lines (226 us in layer 0; with whole rows per instruction 150).  16-byte chunks of a
  //      staged row are XOR-swizzled with the row so that neither side of the exchange has bank conflicts ----
  float* const st32 = (float*)img;                       // [16][256] floats
  unsigned char* const sth = img + 16 * 256 * 4;         // [16][256] halfs, hi
  unsigned char* const stl = sth + 16 * 256 * 2;         // lo
  const int orow = t >> 5, occ = t & 31;                 // read-back of the planes: 16 bytes = 8 channels per thread
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i) __syncthreads();                              // the previous 16 rows have left the staging area
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 y;
      f16x4 hi4, lo4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        y[e] = (acc[i][j][e] - mean[i]) * rstd[i] * lw[j][e] + lb[j][e];
        _Float16 x, z;
        hgl_split_hi_lo(y[e], x, z);
        hi4[e] = x;
        lo4[e] = z;
      }
      if (a.out32) *(f32x4*)(st32 + r * 256 + (((wave * 8 + 4 * j + h) ^ r) * 4)) = y;
      const unsigned off = (unsigned)(r * 512 + (((wave * 4 + 2 * j + (h >> 1)) ^ r) * 16) + (h & 1) * 8);
      *(f16x4*)(sth + off) = hi4;
      *(f16x4*)(stl + off) = lo4;
    }
    __syncthreads();
    const long long ob = (row0 + 16 * i) * 256;
    const unsigned roff = (unsigned)(orow * 512 + ((occ ^ orow) * 16));
    *(u32x4*)(a.oh + ob + orow * 256 + occ * 8) = *(const u32x4*)(sth + roff);
    *(u32x4*)(a.ol + ob + orow * 256 + occ * 8) = *(const u32x4*)(stl + roff);
    if (a.out32) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int idx = t + 512 * k, row = idx >> 6, c4 = idx & 63;
        *(f32x4*)(a.out32 + ob + row * 256 + c4 * 4) = *(const f32x4*)(st32 + row * 256 + ((c4 ^ row) * 4));
      }
    }
  }
}

}  // namespace

// upscaling + hyper-network products of the mask decoder in one launch.  src_hi / src_lo: the decoder's final image
// tokens [P*HW, 256] as fp16 hi / lo planes; hyper [P, 4, 32]; low_res [P, 3, 4g, 4g].  Returns HGL_EINVAL (and leaves the
// error string) when the geometry or the weights do not fit the kernel: the caller then takes the unfused launches.
int hgl_launch_dec_tail(const void* src_hi, const void* src_lo, const float* up0_w, const float* up0_b, const float* ln_w,
                        const float* ln_b, const float* up3_w, const float* up3_b, const float* hyper, int row0, int P, int g,
                        float eps, float* low_res, const uint8_t* skip, hipStream_t st) {
  const void *w0h, *w0l, *w3h, *w3l;
  int s0 = 0, s3 = 0, n0 = 0, k0 = 0, n3 = 0, k3 = 0;
  const int HW = g * g;
  HGL_REQUIRE(hgl_get_split_weight(up0_w, &w0h, &w0l, &s0, &n0, &k0) && hgl_get_split_weight(up3_w, &w3h, &w3l, &s3, &n3, &k3),
              "dec_tail: the upscaling weights have no registered fp16 split");
  HGL_REQUIRE(n0 == 256 && k0 == 256 && n3 == 128 && k3 == 64, "dec_tail: upscaling geometry [%d,%d] / [%d,%d] unsupported", n0, k0, n3, k3);
  HGL_REQUIRE(HW % TAIL_ROWS == 0 && (g % TAIL_ROWS == 0 || TAIL_ROWS % g == 0) && P > 0 && P <= 65535,
              "dec_tail: grid %d / %d prompts unsupported", g, P);
  TailArgs a;
  a.Ah = (const _Float16*)src_hi; a.Al = (const _Float16*)src_lo;
  a.W0h = (const _Float16*)w0h; a.W0l = (const _Float16*)w0l; a.W3h = (const _Float16*)w3h; a.W3l = (const _Float16*)w3l;
  a.b0 = up0_b; a.ln_w = ln_w; a.ln_b = ln_b; a.b3 = up3_b; a.hyper = hyper;
  a.s0 = ldexpf(1.0f, -s0); a.s3 = ldexpf(1.0f, -s3); a.eps = eps;
  a.g = g; a.HW = HW; a.hrow0 = row0; a.out = low_res; a.skip = skip;
  HGL_RESERVE_LDS((dec_tail_kernel), TAIL_LDS, "dec_tail");
  HglProfScope prof(HGL_PROF_OTHER, 2.0 * P * HW * (256.0 * 256 + 4 * 64.0 * 128), 0.0, st);
  hipLaunchKernelGGL(dec_tail_kernel, dim3((unsigned)(HW / TAIL_ROWS), (unsigned)P), dim3(512), TAIL_LDS, st, a);
  return hgl_check_launch("dec_tail");
}

// Image -> token attention + out-projection + residual + norm4 of a TwoWayAttentionBlock in one launch (dec_i2t_kernel).
// q: image-side queries (row (p, n) at q + p * q_bstride + n * ldq; q_bstride = 0: one set for every prompt), k1 / v1 the
// projected prompt tokens [P, 7, 128], R the residual rows (r_bstride = 0: shared), out32 (may be null) / out_hi / out_lo the
// normalised rows.  HGL_EINVAL when the geometry or the weight does not fit.
int hgl_launch_dec_i2t(const float* q, int ldq, long long q_bstride, const float* k1, const float* v1, const float* out_w,
                       const float* out_b, const float* R, long long r_bstride, const float* ln_w, const float* ln_b, float eps,
                       float scale, int P, int HW, float* out32, void* out_hi, void* out_lo, hipStream_t st) {
  const void *wh, *wl;
  int s = 0, n = 0, k = 0;
  HGL_REQUIRE(hgl_get_split_weight(out_w, &wh, &wl, &s, &n, &k), "dec_i2t: the out-projection weight has no registered fp16 split");
  HGL_REQUIRE(n == 256 && k == 128, "dec_i2t: out-projection geometry [%d,%d] unsupported", n, k);
  HGL_REQUIRE(q && k1 && v1 && R && out_hi && out_lo, "dec_i2t: null operand");
  HGL_REQUIRE(HW % I2T_ROWS == 0 && P > 0 && P <= 65535 && (ldq & 3) == 0 && (q_bstride & 3) == 0 && (r_bstride & 3) == 0,
              "dec_i2t: %d tokens / %d prompts / strides unsupported", HW, P);
  I2TArgs a;
  a.q = q; a.sqb = q_bstride; a.ldq = ldq; a.k1 = k1; a.v1 = v1;
  a.Wh = (const _Float16*)wh; a.Wl = (const _Float16*)wl; a.bo = out_b; a.R = R; a.srb = r_bstride;
  a.ln_w = ln_w; a.ln_b = ln_b; a.so = ldexpf(1.0f, -s); a.eps = eps; a.scale = scale;
  a.out32 = out32; a.oh = (_Float16*)out_hi; a.ol = (_Float16*)out_lo; a.HW = HW;
  HglProfScope prof(HGL_PROF_OTHER, 2.0 * P * HW * (128.0 * 256 + 2 * 7 * 128), 0.0, st);
  hipLaunchKernelGGL(dec_i2t_kernel, dim3((unsigned)(HW / I2T_ROWS), (unsigned)P), dim3(512), 0, st, a);
  return hgl_check_launch("dec_i2t");
}
