// Micro-benchmark: how fast does ONE workgroup (512 threads, the write-out pattern of the ping-pong GEMM: a lane stores 16 B
// of a row, 32 stores per lane, 256 KB per workgroup) drain its stores, as a function of how many CUs do it at once?
// Build: hipcc --offload-arch=gfx950 -O3 store_rate.hip -o store_rate ; run: ./store_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ __launch_bounds__(512) void store_kernel(float* out, long long ld, int active, long long* t_issue, long long* t_done, int reps) {
  const int wg = blockIdx.x;
  if (wg >= active) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 15, h = lane >> 4;
  float* base = out + (long long)wg * 256 * ld;
  f32x4 v = {1.f * t, 2.f, 3.f, 4.f};
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int rep = 0; rep < reps; ++rep) {
    float* b2 = base + (long long)rep * 256 * 4096 * ld;   // fresh lines every repetition
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        int row = (wave >> 2) * 128 + mb * 16 + r, col = (wave & 3) * 64 + nb * 16 + 4 * h;
        if (PATTERN == 1) {   // full 128-byte lines: 8 rows x 8 chunks per instruction (pairs of column blocks)
          const int rr = (nb & 1) * 8 + (r & 7), ch = 4 * h + (r >> 3) * 16;
          row = (wave >> 2) * 128 + mb * 16 + rr; col = (wave & 3) * 64 + (nb >> 1) * 32 + ch;
        }
        if (PATTERN == 2) {   // 4 rows x 256 B per instruction
          const int rr = (nb) * 4 + (lane >> 4), ch = 4 * (lane & 15);
          row = (wave >> 2) * 128 + mb * 16 + rr; col = (wave & 3) * 64 + ch;
        }
        *(f32x4*)(b2 + (long long)row * ld + col) = v;
      }
  }
  const long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t2 = __builtin_readcyclecounter();
  if (t == 0) { t_issue[wg] = t1 - t0; t_done[wg] = t2 - t0; }
}

int main() {
  const long long ld = 2304;
  const int reps = 4;
  float* out;
  const size_t bytes = (size_t)reps * 256 * 4096 * ld * 4 / 16 + (size_t)256 * 256 * ld * 4;   // generous
  hipMalloc(&out, (size_t)reps * 256ull * 4096 * ld * 4);
  long long *ti, *td;
  hipMalloc(&ti, 256 * 8); hipMalloc(&td, 256 * 8);
  (void)bytes;
  for (int pattern = 0; pattern < 3; ++pattern)
  for (int active : {1, 256}) {
    for (int it = 0; it < 3; ++it) {
      hipMemset(ti, 0, 256 * 8); hipMemset(td, 0, 256 * 8);
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      if (pattern == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(256), dim3(512), 0, 0, out, ld, active, ti, td, reps);
      else if (pattern == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(256), dim3(512), 0, 0, out, ld, active, ti, td, reps);
      else hipLaunchKernelGGL(store_kernel<2>, dim3(256), dim3(512), 0, 0, out, ld, active, ti, td, reps);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      std::vector<long long> hi(256), hd(256);
      hipMemcpy(hi.data(), ti, 256 * 8, hipMemcpyDeviceToHost); hipMemcpy(hd.data(), td, 256 * 8, hipMemcpyDeviceToHost);
      double si = 0, sd = 0; for (int i = 0; i < active; ++i) { si += hi[i]; sd += hd[i]; }
      if (it == 2) printf("pattern %d active CUs %3d: kernel %.1f us; per WG (256 KB x %d): issue %.0f cyc, issue+drain %.0f cyc per 256 KB (counter ticks)\n", pattern, active, ms * 1e3, reps, si / active / reps, sd / active / reps);
    }
  }
  return 0;
}
