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

// find with path halving: every visited node is re-linked to its grandparent (atomicMin keeps the
// links monotone under concurrent unions), so the row-by-row chains of a tall component collapse.
__device__ __forceinline__ int uf_find_compress(int* L, int x) {
  int p = uf_load(L, x);
  while (p != x) {
    const int gp = uf_load(L, p);
    if (gp != p) atomicMin(&L[x], gp);
    x = p;
    p = gp;
  }
  return x;
}

__device__ __forceinline__ void uf_union(int* L, int a, int b) {
  while (true) {
    a = uf_find_compress(L, a);
    b = uf_find_compress(L, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }   // a > b: hang a under b
    const int old = atomicMin(&L[a], b);
    if (old == a) return;
    a = old;  // somebody re-parented a meanwhile: retry from its new parent
  }
}

// Pass A -- horizontal runs, one wave per image row: L[p] = index of the first pixel of p's run
// (no atomics: a prefix-max of "last non-working column" across the row), -1 outside the working
// set (working pixel = (mask != 0) XOR holes).  area[] is zeroed.
__global__ __launch_bounds__(256) void ccl_rows_kernel(const uint8_t* __restrict__ masks, int holes, int W,
                                                       long long rows, int* __restrict__ L,
                                                       int* __restrict__ area) {
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long long base = row * W;
  const int ppl = (W + 63) / 64;                 // pixels per lane
  const int x0 = lane * ppl;
  int last = -1;                                 // last non-working column inside this lane's span
  for (int i = 0; i < ppl; ++i) {
    const int x = x0 + i;
    if (x < W && ((masks[base + x] != 0) == (holes != 0))) last = x;   // non-working pixel
  }
  int pre = last;                                // inclusive prefix max over lanes
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(pre, o);
    if (lane >= o) pre = max(pre, v);
  }
  int cur = __shfl_up(pre, 1);                   // exclusive: everything left of this lane
  if (lane == 0) cur = -1;
  for (int i = 0; i < ppl; ++i) {
    const int x = x0 + i;
    if (x >= W) break;
    const bool work = (masks[base + x] != 0) != (holes != 0);
    if (!work) cur = x;
    L[base + x] = work ? (int)(base + cur + 1) : -1;
    area[base + x] = 0;
  }
}

// Pass B -- vertical / diagonal links between runs of adjacent rows.  A link is issued only where it
// is not implied by a link one column to the left or by run membership, so a blob costs O(1) unions
// per row instead of one per pixel.
__global__ __launch_bounds__(256) void ccl_merge_kernel(int* __restrict__ L, int H, int W, long long total) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total) return;
  const int me = L[i];
  if (me < 0) return;
  const long long HW = (long long)H * W;
  const int p = (int)(i % HW);
  const int y = p / W, x = p % W;
  if (y + 1 >= H) return;
  const int s = L[i + W];                                   // below
  const int sw = x > 0 ? L[i + W - 1] : -1;                 // below-left
  const int se = x + 1 < W ? L[i + W + 1] : -1;             // below-right
  const int w = x > 0 ? L[i - 1] : -1;                      // left (same run when >= 0)
  const int e = x + 1 < W ? L[i + 1] : -1;                  // right (same run when >= 0)
  if (s >= 0 && !(w >= 0 && sw >= 0)) uf_union(L, me, s);   // first column where the two runs touch
  if (s < 0) {
    if (sw >= 0 && w < 0) uf_union(L, me, sw);              // isolated diagonal contacts
    if (se >= 0 && e < 0) uf_union(L, me, se);
  }
}

// Pass C -- per run: compress the run start's link to its root and add the run length to the
// component's area (one atomic per run, issued by the run's last pixel).
__global__ __launch_bounds__(256) void ccl_count_kernel(int* __restrict__ L, int* __restrict__ area, int W,
                                                        long long total) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total) return;
  const int me = L[i];
  if (me < 0) return;
  const int x = (int)(i % W);
  const bool is_end = (x + 1 == W) || L[i + 1] < 0;
  if (!is_end) return;
  const bool is_start = (x == 0) || L[i - 1] < 0;
  const int start = is_start ? (int)i : me;                 // non-start pixels still hold their run start
  const int r = uf_find(L, start);
  atomicAdd(&area[r], (int)i - start + 1);
}
__global__ __launch_bounds__(256) void ccl_compress_kernel(int* __restrict__ L, int W, long long total) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total || L[i] < 0) return;
  const int x = (int)(i % W);
  if ((x == 0) || L[i - 1] < 0) L[i] = uf_find(L, (int)i);  // run starts point at the root
}

// root of the component of working pixel i after ccl_compress_kernel
__device__ __forceinline__ int ccl_root(const int* __restrict__ L, long long i, int W) {
  const int me = L[i];
  const int x = (int)(i % W);
  const bool is_start = (x == 0) || L[i - 1] < 0;
  return is_start ? me : L[me];
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
                                                        int holes, int W, long long HW, long long total, int thresh,
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
  const int r = L[i] >= 0 ? ccl_root(L, i, W) : -1;
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
  // grid.y = mask; each thread scans 16 pixels (one 16-byte load when aligned)
  const int n = blockIdx.y;
  const uint8_t* m = masks + (long long)n * HW;
  int minx = 0x7fffffff, miny = 0x7fffffff, maxx = -1, maxy = -1;
  const bool al = (((uintptr_t)m) & 15) == 0;
  for (long long p0 = (blockIdx.x * 256ll + threadIdx.x) * 16; p0 < HW; p0 += (long long)gridDim.x * 256 * 16) {
    unsigned w4[4] = {0, 0, 0, 0};
    if (al && p0 + 16 <= HW) {
      const uint4 v = *(const uint4*)(m + p0);
      w4[0] = v.x; w4[1] = v.y; w4[2] = v.z; w4[3] = v.w;
    } else {
      for (int e = 0; e < 16 && p0 + e < HW; ++e) w4[e >> 2] |= (unsigned)(m[p0 + e] != 0) << (8 * (e & 3));
    }
    if ((w4[0] | w4[1] | w4[2] | w4[3]) == 0) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if ((w4[e >> 2] >> (8 * (e & 3))) & 0xff) {
        const long long p = p0 + e;
        const int x = (int)(p % W), y = (int)(p / W);
        minx = min(minx, x); maxx = max(maxx, x); miny = min(miny, y); maxy = max(maxy, y);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    minx = min(minx, __shfl_xor(minx, o)); miny = min(miny, __shfl_xor(miny, o));
    maxx = max(maxx, __shfl_xor(maxx, o)); maxy = max(maxy, __shfl_xor(maxy, o));
  }
  if ((threadIdx.x & 63) == 0 && maxx >= 0) {
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
  hipLaunchKernelGGL(ccl_rows_kernel, dim3((unsigned)(((long long)N * H + 3) / 4)), dim3(256), 0, st, masks, holes, W,
                     (long long)N * H, L, area);
  hipLaunchKernelGGL(ccl_stats_init_kernel, dim3((N + 255) / 256), dim3(256), 0, st, stats, N);
  hipLaunchKernelGGL(ccl_merge_kernel, dim3(g1(total)), dim3(256), 0, st, L, H, W, total);
  hipLaunchKernelGGL(ccl_count_kernel, dim3(g1(total)), dim3(256), 0, st, L, area, W, total);
  hipLaunchKernelGGL(ccl_compress_kernel, dim3(g1(total)), dim3(256), 0, st, L, W, total);
  hipLaunchKernelGGL(ccl_stats_kernel, dim3(g1(total)), dim3(256), 0, st, L, area, HW, total, area_thresh, stats);
  hipLaunchKernelGGL(ccl_argmax_kernel, dim3(g1(total)), dim3(256), 0, st, L, area, HW, total, stats);
  hipLaunchKernelGGL(ccl_apply_kernel, dim3(g1(total)), dim3(256), 0, st, masks, L, area, stats, holes, W, HW, total,
                     area_thresh, out, changed);
  return hgl_check_launch("remove_small_regions");
}

int hgl_mask_boxes(const uint8_t* masks, int N, int H, int W, int32_t* boxes_xyxy, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(masks && boxes_xyxy && N > 0 && H > 0 && W > 0, "mask_boxes: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const long long HW = (long long)H * W, total = HW * N;
  hipLaunchKernelGGL(box_init_kernel, dim3((N + 255) / 256), dim3(256), 0, st, (int*)boxes_xyxy, N);
  hipLaunchKernelGGL(box_kernel, dim3((unsigned)min((HW / 16 + 255) / 256 + 1, 64ll), N), dim3(256), 0, st, masks, W, HW, total, (int*)boxes_xyxy);
  hipLaunchKernelGGL(box_final_kernel, dim3((N + 255) / 256), dim3(256), 0, st, (int*)boxes_xyxy, N);
  return hgl_check_launch("mask_boxes");
}

}  // extern "C"
