// Micro-benchmark: sustained rate of the two fp16 MFMA shapes with every CU busy (the chip is power-limited under dense
// MFMA work: which shape buys more flops per watt?).  Registers only, no memory traffic: 8 waves per CU, each wave issues
// chains of independent MFMAs.   Build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate ; run: ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
  h8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) {
      // pseudo-random operands in [-1, 1): realistic bit toggling (power), no overflow (products sum to ~sqrt(K) per step; the
      // accumulators are rescaled by nothing: they grow like sqrt(iters), far below fp32 range)
      unsigned h1 = (threadIdx.x * 2654435761u) ^ ((i * 8 + e) * 40503u) ^ (blockIdx.x * 97u);
      h1 ^= h1 >> 13; h1 *= 0x5bd1e995u; h1 ^= h1 >> 15;
      unsigned h2 = h1 * 0x27d4eb2du; h2 ^= h2 >> 15;
      a[i][e] = (_Float16)(seed * ((int)(h1 & 0xffff) - 32768) / 32768.0f);
      b[i][e] = (_Float16)(seed * ((int)(h2 & 0xffff) - 32768) / 32768.0f);
    }
  float s = 0.f;
  if (SHAPE == 16) {
    f4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)          // 8 MFMAs of 32x32x16 = the flops of 16 MFMAs of 16x16x32
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i + 2 * r], b[j + 2 * r], acc[i * 2 + j], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  }
  if (s == 12345.678f) out[0] = s;
}

int main() {
  float* d; hipMalloc(&d, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, wgs = 256 * 1;
  for (float seedv : {1.0f, 0.0f})
  for (int shape : {16, 32, 16, 32}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      for (int l = 0; l < 20; ++l) {
        if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(wgs), dim3(512), 0, 0, d, iters, seedv);
        else hipLaunchKernelGGL(k<32>, dim3(wgs), dim3(512), 0, 0, d, iters, seedv);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = 20.0 * wgs * 8 * (double)iters * 16 * (16.0 * 16 * 32 * 2);
      if (rep) printf("operands %s, shape %dx%d: %.1f ms for 20 launches, %.0f TF/s issued\n", seedv != 0.f ? "random" : "zero", shape, shape, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
