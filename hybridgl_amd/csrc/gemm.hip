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
//   * register prefetch of tile t+1 is issued before the MFMAs of tile t and
//     written to the other LDS buffer afterwards: one barrier per K-tile.
//   * 1-D grid with a bijective XCD-aware remap so that blocks sharing A/W panels
//     run on the same XCD (private L2).
#include "hgl_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_LD = BK + 4;  // floats per LDS row (padded)
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

__device__ __forceinline__ float act_apply(float x, int act) {
  switch (act) {
    case HGL_ACT_QUICKGELU:
      return x / (1.0f + __expf(-1.702f * x));
    case HGL_ACT_GELU:
      return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    case HGL_ACT_RELU:
      return x > 0.0f ? x : 0.0f;
    default:
      return x;
  }
}

__global__ __launch_bounds__(NTHREADS, 2) void gemm_f32_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // layout: [2 buffers][A: BM x LDS_LD | W: BN x LDS_LD]
  constexpr int TILE_F = (BM + BN) * LDS_LD;

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
  const int ld_k4 = t & 7;   // float4 index within the 32-wide K tile
  const int ld_row = t >> 3; // 0..31, +32*i
  const int row0 = tile_m * BM, col0 = tile_n * BN;

  f32x4 pa[4], pw[4];
  auto load_tile = [&](int kt) {
    const int k = kt * BK + ld_k4 * 4;
    const bool kok = k < g.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ra = row0 + ld_row + 32 * i;
      const int rw = col0 + ld_row + 32 * i;
      pa[i] = (kok && ra < g.M) ? *(const f32x4*)(A + (long long)ra * g.lda + k) : f32x4{0, 0, 0, 0};
      pw[i] = (kok && rw < g.N) ? *(const f32x4*)(W + (long long)rw * g.ldw + k) : f32x4{0, 0, 0, 0};
    }
  };
  auto store_tile = [&](int buf) {
    float* As = smem + buf * TILE_F;
    float* Ws = As + BM * LDS_LD;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(f32x4*)(As + (ld_row + 32 * i) * LDS_LD + ld_k4 * 4) = pa[i];
      *(f32x4*)(Ws + (ld_row + 32 * i) * LDS_LD + ld_k4 * 4) = pw[i];
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
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
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
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias, activation, residual, store ----
  // acc[i][j][e]: row = (e&3) + 8*(e>>2) + 4*h, col = r within the 32x32 tile
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = col0 + wn * 64 + j * 32 + r;
    if (col >= g.N) continue;
    const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = row0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < g.M) {
          float v = act_apply(acc[i][j][e] + bv, g.act);
          if (R) v += R[(long long)row * g.ldr + col];
          C[(long long)row * g.ldc + col] = v;
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
  const size_t lds = 2 * (BM + BN) * LDS_LD * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  HglProfScope prof(HGL_PROF_GEMM, 2.0 * M * (double)N * K * batch,
                    4.0 * batch * ((double)M * K + (double)N * K + (double)M * N * (R ? 2 : 1)), st);
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)nwg), dim3(NTHREADS), lds, st, g);
  return hgl_check_launch("gemm_f32");
}
