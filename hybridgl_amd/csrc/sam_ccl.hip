// Connected-component clean-up of mask proposals on the device:
//   remove_small_regions (utils/amg.py:267-291) -- the reference runs cv2.connectedComponentsWithStats
//   (8-connectivity) on the CPU for every mask, twice (holes, islands); here every mask of the batch
//   is labelled in parallel with a lock-free union-find (atomicMin on parent links), component
//   areas are histogrammed with atomics and the fill rule is applied in one more pass.
//   hgl_mask_boxes == batched_mask_to_box (utils/amg.py:303-346).
// Roots are the smallest pixel index of a component, i.e. components are ordered by their first
// pixel in raster order exactly like the reference's label numbering (needed for its
// "keep the largest, first on ties" rule).
#include "hgl_common.h"

namespace {

// Parent links only ever decrease and the value returned by atomicMin is authoritative, so a
// stale read (another CU's L1 / another XCD's L2) costs extra iterations, never correctness.
// Relaxed agent-scope loads keep the compiler from caching a link in a register.
__device__ __forceinline__ int uf_load(const int* L, int x) {
  return __hip_atomic_load(L + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int uf_find(const int* L, int x) {
  int p = uf_load(L, x);
  while (p != x) { x = p; p = uf_load(L, x); }
  return x;
}

__device__ __forceinline__ void uf_union(int* L, int a, int b) {
  while (true) {
    a = uf_find(L, a);
    b = uf_find(L, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }   // a > b: hang a under b
    const int old = atomicMin(&L[a], b);
    if (old == a) return;
    a = old;  // somebody re-parented a meanwhile: retry from its new parent
  }
}

// working pixel = (mask != 0) XOR holes
__global__ __launch_bounds__(256) void ccl_init_kernel(const uint8_t* __restrict__ masks, int holes,
                                                       long long total, int* __restrict__ L,
                                                       int* __restrict__ area) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total) return;
  const bool w = (masks[i] != 0) != (holes != 0);
  L[i] = w ? (int)i : -1;   // indices are global over the batch (< 2^31, checked by the launcher)
  area[i] = 0;
}

__global__ __launch_bounds__(256) void ccl_merge_kernel(int* __restrict__ L, int H, int W, long long total) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total || L[i] < 0) return;
  const long long HW = (long long)H * W;
  const int p = (int)(i % HW);
  const int y = p / W, x = p % W;
  // forward neighbours of the 8-neighbourhood: E, SW, S, SE (the others are covered symmetrically)
  if (x + 1 < W && L[i + 1] >= 0) uf_union(L, (int)i, (int)i + 1);
  if (y + 1 < H) {
    if (x > 0 && L[i + W - 1] >= 0) uf_union(L, (int)i, (int)(i + W - 1));
    if (L[i + W] >= 0) uf_union(L, (int)i, (int)(i + W));
    if (x + 1 < W && L[i + W + 1] >= 0) uf_union(L, (int)i, (int)(i + W + 1));
  }
}

__global__ __launch_bounds__(256) void ccl_count_kernel(int* __restrict__ L, int* __restrict__ area, long long total) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total || L[i] < 0) return;
  const int r = uf_find(L, (int)i);
  L[i] = r;   // full compression: later passes read the root directly
  atomicAdd(&area[r], 1);
}

// per mask: stats[n*4+0] = number of small components, [1] = number of large ones,
// [2] = max area, [3] = smallest root among the components of max area
__global__ __launch_bounds__(256) void ccl_stats_kernel(const int* __restrict__ L, const int* __restrict__ area,
                                                        long long HW, long long total, int thresh,
                                                        int* __restrict__ stats) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total || L[i] != (int)i) return;   // one thread per root
  const int n = (int)(i / HW), a = area[i];
  atomicAdd(&stats[n * 4 + (a < thresh ? 0 : 1)], 1);
  atomicMax(&stats[n * 4 + 2], a);
}
__global__ __launch_bounds__(256) void ccl_argmax_kernel(const int* __restrict__ L, const int* __restrict__ area,
                                                         long long HW, long long total, int* __restrict__ stats) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total || L[i] != (int)i) return;
  const int n = (int)(i / HW);
  if (area[i] == stats[n * 4 + 2]) atomicMin(&stats[n * 4 + 3], (int)i);
}

__global__ void ccl_stats_init_kernel(int* stats, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  stats[n * 4 + 0] = 0; stats[n * 4 + 1] = 0; stats[n * 4 + 2] = 0; stats[n * 4 + 3] = 0x7fffffff;
}

// holes:   out = mask | (working && area < thresh)                      (fill small holes)
// islands: out = working && area >= thresh ; if no component is large, keep the (first) largest
__global__ __launch_bounds__(256) void ccl_apply_kernel(const uint8_t* __restrict__ masks, const int* __restrict__ L,
                                                        const int* __restrict__ area, const int* __restrict__ stats,
                                                        int holes, long long HW, long long total, int thresh,
                                                        uint8_t* __restrict__ out, uint8_t* __restrict__ changed) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total) return;
  const int n = (int)(i / HW);
  const bool m = masks[i] != 0;
  const int n_small = stats[n * 4 + 0];
  if (n_small == 0) {            // nothing below the threshold: mask unchanged (utils/amg.py:281-282)
    out[i] = m ? 1 : 0;
    if (i % HW == 0) changed[n] = 0;
    return;
  }
  if (i % HW == 0) changed[n] = 1;
  const int r = L[i];
  bool o;
  if (holes) {
    o = m || (r >= 0 && area[r] < thresh);
  } else {
    const bool any_large = stats[n * 4 + 1] > 0;
    o = r >= 0 && (any_large ? area[r] >= thresh : r == stats[n * 4 + 3]);
  }
  out[i] = o ? 1 : 0;
}

// batched_mask_to_box: counters [N,4] = minx, miny, maxx, maxy
__global__ void box_init_kernel(int* b, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  b[n * 4 + 0] = 0x7fffffff; b[n * 4 + 1] = 0x7fffffff; b[n * 4 + 2] = -1; b[n * 4 + 3] = -1;
}
__global__ __launch_bounds__(256) void box_kernel(const uint8_t* __restrict__ masks, int W, long long HW,
                                                  long long total, int* __restrict__ b) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  int minx = 0x7fffffff, miny = 0x7fffffff, maxx = -1, maxy = -1;
  const int n = i < total ? (int)(i / HW) : -1;
  if (i < total && masks[i]) {
    const int p = (int)(i % HW);
    minx = maxx = p % W;
    miny = maxy = p / W;
  }
  // a wave may straddle two masks only at a boundary: fall back to per-lane atomics there
  const int n0 = __shfl(n, 0), n63 = __shfl(n, 63);
  if (n0 == n63 && n0 >= 0) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      minx = min(minx, __shfl_xor(minx, o)); miny = min(miny, __shfl_xor(miny, o));
      maxx = max(maxx, __shfl_xor(maxx, o)); maxy = max(maxy, __shfl_xor(maxy, o));
    }
    if ((threadIdx.x & 63) == 0 && maxx >= 0) {
      atomicMin(&b[n0 * 4 + 0], minx); atomicMin(&b[n0 * 4 + 1], miny);
      atomicMax(&b[n0 * 4 + 2], maxx); atomicMax(&b[n0 * 4 + 3], maxy);
    }
  } else if (n >= 0 && maxx >= 0) {
    atomicMin(&b[n * 4 + 0], minx); atomicMin(&b[n * 4 + 1], miny);
    atomicMax(&b[n * 4 + 2], maxx); atomicMax(&b[n * 4 + 3], maxy);
  }
}
__global__ void box_final_kernel(int* b, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  if (b[n * 4 + 2] < 0) { b[n * 4 + 0] = 0; b[n * 4 + 1] = 0; b[n * 4 + 2] = 0; b[n * 4 + 3] = 0; }
}

inline unsigned g1(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" {

size_t hgl_remove_small_regions_workspace_bytes(int N, int H, int W) {
  const size_t px = (size_t)N * H * W;
  return hgl_align_up(px * sizeof(int), 256) * 2 + hgl_align_up((size_t)N * 4 * sizeof(int), 256);
}

int hgl_remove_small_regions(const uint8_t* masks, int N, int H, int W, int area_thresh, int holes, uint8_t* out,
                             uint8_t* changed, void* workspace, size_t workspace_bytes, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(masks && out && changed && N > 0 && H > 0 && W > 0, "remove_small_regions: bad arguments");
  const long long total = (long long)N * H * W;
  HGL_REQUIRE(total < (1ll << 31), "remove_small_regions: batch too large (N*H*W must be < 2^31)");
  if (!workspace || workspace_bytes < hgl_remove_small_regions_workspace_bytes(N, H, W)) {
    hgl_set_error("remove_small_regions: workspace too small");
    return HGL_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  HglArena ar(workspace, workspace_bytes);
  int* L = ar.take<int>((size_t)total);
  int* area = ar.take<int>((size_t)total);
  int* stats = ar.take<int>((size_t)N * 4);
  const long long HW = (long long)H * W;
  hipLaunchKernelGGL(ccl_init_kernel, dim3(g1(total)), dim3(256), 0, st, masks, holes, total, L, area);
  hipLaunchKernelGGL(ccl_stats_init_kernel, dim3((N + 255) / 256), dim3(256), 0, st, stats, N);
  hipLaunchKernelGGL(ccl_merge_kernel, dim3(g1(total)), dim3(256), 0, st, L, H, W, total);
  hipLaunchKernelGGL(ccl_count_kernel, dim3(g1(total)), dim3(256), 0, st, L, area, total);
  hipLaunchKernelGGL(ccl_stats_kernel, dim3(g1(total)), dim3(256), 0, st, L, area, HW, total, area_thresh, stats);
  hipLaunchKernelGGL(ccl_argmax_kernel, dim3(g1(total)), dim3(256), 0, st, L, area, HW, total, stats);
  hipLaunchKernelGGL(ccl_apply_kernel, dim3(g1(total)), dim3(256), 0, st, masks, L, area, stats, holes, HW, total,
                     area_thresh, out, changed);
  return hgl_check_launch("remove_small_regions");
}

int hgl_mask_boxes(const uint8_t* masks, int N, int H, int W, int32_t* boxes_xyxy, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(masks && boxes_xyxy && N > 0 && H > 0 && W > 0, "mask_boxes: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const long long HW = (long long)H * W, total = HW * N;
  hipLaunchKernelGGL(box_init_kernel, dim3((N + 255) / 256), dim3(256), 0, st, (int*)boxes_xyxy, N);
  hipLaunchKernelGGL(box_kernel, dim3(g1(total)), dim3(256), 0, st, masks, W, HW, total, (int*)boxes_xyxy);
  hipLaunchKernelGGL(box_final_kernel, dim3((N + 255) / 256), dim3(256), 0, st, (int*)boxes_xyxy, N);
  return hgl_check_launch("mask_boxes");
}

}  // extern "C"
