// fp32 GEMM on the CDNA4 matrix cores:  C = act(A @ W^T + bias) + R
//
// Replaces every nn.Linear / conv-as-matmul call of the reference hot path
// (clip/model.py:209-218 in_proj/out_proj/c_fc/c_proj, image_encoder.py:213-214,
// common.py:21-27 MLPBlock, transformer.py q/k/v/out_proj, ...).
//
// Design (gfx950):
//   * v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate (the
//     reference runs pure fp32, clip/model.py:509) -> roofline = fp32 matrix peak.
//   * 128x128 block tile, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles
//     (64 accumulator registers), BK = 32.
//   * Both operands are K-contiguous ("NT" GEMM), staged global -> registers ->
//     LDS with 16-byte accesses; LDS rows padded by one float4 (36 floats) so the
//     ds_read_b128 fragment reads are bank-conflict free.
//   * A lane's fragment is 4 consecutive k (one ds_read_b128) feeding 4 MFMA
//     k-steps; the k-permutation is the same for A and W so the sum is unchanged.
//   * register prefetch of tile t+1 is issued before the MFMAs of tile t and written to LDS
//     afterwards; ONE LDS buffer (two barriers per K-tile) so that three workgroups fit a CU:
//     the extra wave per SIMD hides the staging/barrier gaps better than double buffering did.
//   * 1-D grid with a bijective XCD-aware remap so that blocks sharing A/W panels
//     run on the same XCD (private L2).
#include "hgl_common.h"
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int BM = 128, BN = 128;
constexpr int NTHREADS = 256;

struct GemmArgs {
  const float* A;
  const float* W;
  const float* bias;
  const float* R;
  float* C;
  int M, N, K;
  int lda, ldw, ldr, ldc;
  long long sA, sW, sR, sC;
  int act;
  int tiles_m, tiles_n;
};

template <int ACT>
__device__ __forceinline__ float act_apply(float x) {
  if constexpr (ACT == HGL_ACT_QUICKGELU) return hgl_quick_gelu(x);
  if constexpr (ACT == HGL_ACT_GELU) return hgl_gelu_erf(x);
  if constexpr (ACT == HGL_ACT_RELU) return x > 0.0f ? x : 0.0f;
  return x;
}

// ACT is a template parameter so that the epilogue carries no per-element switch.
template <int ACT, int BK, int NBUF, int OCC>
__global__ __launch_bounds__(NTHREADS, OCC) void gemm_f32_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // layout: [NBUF buffers][A: BM x LDS_LD | W: BN x LDS_LD]
  constexpr int LDS_LD = BK + 4;  // floats per LDS row (padded by one float4: conflict-free b128 reads)
  constexpr int TILE_F = (BM + BN) * LDS_LD;
  constexpr int K4 = BK / 4;            // float4 per row of a K tile
  constexpr int RSTEP = NTHREADS / K4;  // rows covered by one pass of the block
  constexpr int NLD = BM / RSTEP;       // float4 loads per thread per operand per K tile

  // ---- XCD-aware bijective block remap (blocks b, b+8 share an XCD) ----
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tiles_per_batch = g.tiles_m * g.tiles_n;
  const int batch = bid / tiles_per_batch;
  const int tid_in_batch = bid - batch * tiles_per_batch;
  // group 8 M-tiles per column sweep so that a resident set covers a compact 2-D patch
  constexpr int GM = 8;
  const int group = tid_in_batch / (GM * g.tiles_n);
  const int first_m = group * GM;
  const int gm = min(g.tiles_m - first_m, GM);
  const int rem = tid_in_batch - group * GM * g.tiles_n;
  const int tile_m = first_m + rem % gm;
  const int tile_n = rem / gm;

  const float* __restrict__ A = g.A + batch * g.sA;
  const float* __restrict__ W = g.W + batch * g.sW;
  float* __restrict__ C = g.C + batch * g.sC;
  const float* R = g.R ? g.R + batch * g.sR : nullptr;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  // global staging coordinates: 4 float4 of A and 4 of W per thread per K-tile
  const int ld_k4 = t % K4;   // float4 index within the K tile
  const int ld_row = t / K4;  // + RSTEP*i
  const int row0 = tile_m * BM, col0 = tile_n * BN;

  f32x4 pa[NLD], pw[NLD];
  // Staging loads carry no arithmetic on the loaded registers (that would pull the wait for the
  // loads in front of the MFMA section): out-of-range rows read a clamped (valid) row whose
  // accumulators are never stored; only a partial last K tile is zero-masked, at store time.
  const int mclamp = g.M - 1, nclamp = g.N - 1, kclamp = g.K - 4;
  const float *pa_p[NLD], *pw_p[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    pa_p[i] = A + (long long)min(row0 + ld_row + RSTEP * i, mclamp) * g.lda;
    pw_p[i] = W + (long long)min(col0 + ld_row + RSTEP * i, nclamp) * g.ldw;
  }
  auto load_tile = [&](int kt) {
    const int kk = min(kt * BK + ld_k4 * 4, kclamp);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      pa[i] = *(const f32x4*)(pa_p[i] + kk);
      pw[i] = *(const f32x4*)(pw_p[i] + kk);
    }
  };
  const bool k_partial = (g.K % BK) != 0;
  auto mask_tail = [&](int kt) {  // only for the last, partial K tile
    if (k_partial && kt * BK + ld_k4 * 4 >= g.K) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) { pa[i] = f32x4{0, 0, 0, 0}; pw[i] = f32x4{0, 0, 0, 0}; }
    }
  };
  auto store_tile = [&](int buf) {
    float* As = smem + buf * TILE_F;
    float* Ws = As + BM * LDS_LD;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      *(f32x4*)(As + (ld_row + RSTEP * i) * LDS_LD + ld_k4 * 4) = pa[i];
      *(f32x4*)(Ws + (ld_row + RSTEP * i) * LDS_LD + ld_k4 * 4) = pw[i];
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
  mask_tail(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = NBUF == 2 ? (kt & 1) : 0;
    if (kt + 1 < nk) load_tile(kt + 1);
    const float* As = smem + buf * TILE_F + (wm * 64 + r) * LDS_LD + 4 * h;
    const float* Ws = smem + buf * TILE_F + BM * LDS_LD + (wn * 64 + r) * LDS_LD + 4 * h;
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
      f32x4 a0 = *(const f32x4*)(As + c * 8);
      f32x4 a1 = *(const f32x4*)(As + 32 * LDS_LD + c * 8);
      f32x4 b0 = *(const f32x4*)(Ws + c * 8);
      f32x4 b1 = *(const f32x4*)(Ws + 32 * LDS_LD + c * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
      }
    }
    if (NBUF == 1) __syncthreads();  // single buffer: everyone done reading before it is overwritten
    if (kt + 1 < nk) {
      if (kt + 2 == nk) mask_tail(kt + 1);
      store_tile(NBUF == 2 ? (buf ^ 1) : 0);
    }
    __syncthreads();
  }

  // ---- epilogue: bias, activation, residual, store ----
  // acc[i][j][e]: row = (e&3) + 8*(e>>2) + 4*h, col = r within the 32x32 tile.
  // Residual values of a whole 32x32 tile are fetched first (16 loads in flight, clamped
  // addresses: no per-element wait), then combined and stored.
  const bool full_tile = (row0 + BM <= g.M) && (col0 + BN <= g.N);
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
      if (R) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = min(rbase + (e & 3) + 8 * (e >> 2), mclamp);
          rv[e] = R[(long long)row * g.ldr + colc];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) rv[e] = 0.f;
      }
      if (full_tile) {  // uniform: interior tiles store without per-element predicates
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = rbase + (e & 3) + 8 * (e >> 2);
          C[(long long)row * g.ldc + col] = act_apply<ACT>(acc[i][j][e] + bv) + rv[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = rbase + (e & 3) + 8 * (e >> 2);
          const float v = act_apply<ACT>(acc[i][j][e] + bv) + rv[e];
          if (cok && row < g.M) C[(long long)row * g.ldc + col] = v;
        }
      }
    }
  }
}

}  // namespace

int hgl_launch_gemm(const float* A, const float* W, const float* bias, const float* R, float* C,
                    int M, int N, int K, int lda, int ldw, int ldr, int ldc, int batch,
                    long long sA, long long sW, long long sR, long long sC, int act,
                    hipStream_t st) {
  HGL_REQUIRE(A && W && C, "gemm: null operand");
  HGL_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, "gemm: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
  HGL_REQUIRE((K & 3) == 0 && (lda & 3) == 0 && (ldw & 3) == 0, "gemm: K, lda, ldw must be multiples of 4 (K=%d lda=%d ldw=%d)", K, lda, ldw);
  HGL_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0, "gemm: A and W must be 16-byte aligned");
  HGL_REQUIRE((sA & 3) == 0 && (sW & 3) == 0, "gemm: batch strides of A and W must be multiples of 4");
  HGL_REQUIRE(act >= 0 && act <= 3, "gemm: bad activation %d", act);
  // f16x3 mode, small M, registered weight: the split-fp16 small-tile kernel (gemm_f16x3.hip)
  if (hgl_gemm_skinny_applicable(W, M, N, K, lda, ldw, batch))
    return hgl_launch_gemm_x3_skinny(A, lda, W, bias, R, ldr, C, ldc, M, N, K, act, st);
  GemmArgs g;
  g.A = A; g.W = W; g.bias = bias; g.R = R; g.C = C;
  g.M = M; g.N = N; g.K = K;
  g.lda = lda; g.ldw = ldw; g.ldr = ldr; g.ldc = ldc;
  g.sA = sA; g.sW = sW; g.sR = sR; g.sC = sC;
  g.act = act;
  g.tiles_m = (M + BM - 1) / BM;
  g.tiles_n = (N + BN - 1) / BN;
  const long long nwg = (long long)g.tiles_m * g.tiles_n * batch;
  HGL_REQUIRE(nwg < (1ll << 31), "gemm: grid too large");
  // default: single LDS buffer (36.9 KB) -> 3 workgroups per CU (3 waves/SIMD); measured on MI355X
  // 104-121 TF/s on the CLIP shapes vs 87-108 for the double-buffered 2-workgroup variant, which
  // HGL_GEMM_BK32X2=1 keeps selectable in the diagnostic build (make diag).
  static int variant = -1;
  if (variant < 0) {
    variant = HGL_DIAG_SWITCH("HGL_GEMM_BK32X2", 0) ? 1 : 0;
  }
  HglProfScope prof(HGL_PROF_GEMM, 2.0 * M * (double)N * K * batch,
                    4.0 * batch * ((double)M * K + (double)N * K + (double)M * N * (R ? 2 : 1)), st);
#define HGL_GEMM_LAUNCH(ACT_, BK_, NBUF_, OCC_)                                                            \
  do {                                                                                                  \
    const size_t lds_ = (size_t)NBUF_ * (BM + BN) * (BK_ + 4) * sizeof(float);                          \
    HGL_RESERVE_LDS((gemm_f32_kernel<ACT_, BK_, NBUF_, OCC_>), lds_, "gemm_f32");                        \
    hipLaunchKernelGGL((gemm_f32_kernel<ACT_, BK_, NBUF_, OCC_>), dim3((unsigned)nwg), dim3(NTHREADS), lds_,  \
                       st, g);                                                                          \
  } while (0)
#define HGL_GEMM_VARIANTS(ACT_)                               \
  do {                                                        \
    if (variant == 1) HGL_GEMM_LAUNCH(ACT_, 32, 2, 2);        \
    else HGL_GEMM_LAUNCH(ACT_, 32, 1, 3);                     \
  } while (0)
  switch (act) {
    case HGL_ACT_QUICKGELU: HGL_GEMM_VARIANTS(HGL_ACT_QUICKGELU); break;
    case HGL_ACT_GELU: HGL_GEMM_VARIANTS(HGL_ACT_GELU); break;
    case HGL_ACT_RELU: HGL_GEMM_VARIANTS(HGL_ACT_RELU); break;
    default: HGL_GEMM_VARIANTS(HGL_ACT_NONE); break;
  }
  return hgl_check_launch("gemm_f32");
}
