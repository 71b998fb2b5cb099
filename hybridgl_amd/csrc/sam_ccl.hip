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

// ---- All passes walk the MASK BYTES, one wave per image row and 64 consecutive pixels per step (coalesced); the
// run structure of a row is recovered on the fly from wave ballots (working pixel = (mask != 0) XOR holes), so the
// int arrays are touched only at run starts: L[start] = parent link of the run (a union-find over runs, not over
// pixels), area[root] = pixels of the component.  (A first version kept a label per pixel and spent ~700 us per call
// at 64 x 640 x 640 moving 105 MB int arrays through six passes; the byte passes take a fifth of that.)
constexpr int CCL_STEPS = 16;     // steps of 64 pixels whose bytes are fetched up front (independent loads: one latency)
struct RowScan {
  const uint8_t* row;     // mask bytes of this image row
  int W, holes;
  int carry;              // last non-working column of the steps done so far
  bool prev_last;         // working bit of the column just left of the current step
  int prev_last_start;    // its run start (valid when prev_last)
  uint8_t v[CCL_STEPS + 1];   // working flags of this lane's pixel in each step of the group (+ the first of the next group)
  unsigned lo_mask, hi_mask;  // bits of the lanes below this one, per 32-bit half of a ballot
  // per step
  bool work;              // this lane's pixel is a working pixel
  unsigned long long wb;  // ballot of `work`
  int start;              // run start column of this lane's pixel (valid when work)
  bool left, right;       // working bits of the columns just left / right of this lane's pixel

  __device__ __forceinline__ void init(const uint8_t* r, int W_, int holes_, int lane) {
    row = r; W = W_; holes = holes_; carry = -1; prev_last = false; prev_last_start = 0;
    lo_mask = lane < 32 ? (1u << lane) - 1u : 0xffffffffu;
    hi_mask = lane < 32 ? 0u : (1u << (lane - 32)) - 1u;
  }
  // fetch the group of steps that begins at column g0
  __device__ __forceinline__ void load(int g0, int lane) {
#pragma unroll
    for (int k = 0; k <= CCL_STEPS; ++k) {
      const int x = g0 + 64 * k + (k < CCL_STEPS ? lane : 0);
      v[k] = x < W ? row[x] : (uint8_t)(holes ? 1 : 0);       // beyond the row: a non-working value
    }
  }
  // step k of the group (columns x0 .. x0+63).  Needs the whole wave (ballot, neighbour exchange): 32-bit VALU
  // work only -- the first version shifted 64-bit masks per lane and was bound by exactly that.
  __device__ __forceinline__ void step(int k, int x0, int lane) {
    work = (v[k] != 0) != (holes != 0);
    wb = __ballot(work);
    const unsigned blo = ~(unsigned)wb & lo_mask, bhi = ~(unsigned)(wb >> 32) & hi_mask;
    int last = carry;
    if (blo) last = x0 + 31 - __clz(blo);
    if (bhi) last = x0 + 63 - __clz(bhi);
    start = last + 1;
    const int wi = work ? 1 : 0;
    const int up = __shfl_up(wi, 1), dn = __shfl_down(wi, 1);
    const bool next0 = (__shfl((int)v[k + 1], 0) != 0) != (holes != 0);   // lane 0 of the next step (or the look-ahead byte)
    left = lane > 0 ? up != 0 : prev_last;
    right = lane < 63 ? dn != 0 : next0;
  }
  __device__ __forceinline__ void advance(int x0) {
    const unsigned long long nb = ~wb;
    prev_last = (wb >> 63) != 0;
    prev_last_start = __shfl(start, 63);
    if (nb) carry = x0 + 63 - __clzll(nb);                 // (all 64 working: the carry stays)
  }
};
#define CCL_FOR_STEPS(scan_load, ...)                                    \
  for (int g0 = 0; g0 < W; g0 += 64 * CCL_STEPS) {                        \
    scan_load;                                                            \
    _Pragma("unroll") for (int k = 0; k < CCL_STEPS; ++k) {               \
      const int x0 = g0 + 64 * k;                                         \
      if (x0 >= W) break;                                                 \
      __VA_ARGS__                                                         \
    }                                                                     \
  }

// The same scan over a row of WORKING BITS (64 pixels per 8-byte word, written once by pass A): the passes after the first read
// an eighth of the bytes, the wave fetches a step's word with one scalar load, and a lane's neighbours come out of the word
// by shifts -- no ballot, no wave shuffles.  Bits beyond the row's width are 0 (not working).  Same fields as RowScan.
struct BitScan {
  const unsigned long long* bits;   // the row's words (wave-uniform pointer)
  int W, carry;
  bool prev_last;
  int prev_last_start;
  unsigned lo_mask, hi_mask;
  bool work;
  unsigned long long wb;
  int start;
  bool left, right;
  __device__ __forceinline__ void init(const unsigned long long* b, int W_, int lane) {
    bits = b; W = W_; carry = -1; prev_last = false; prev_last_start = 0;
    lo_mask = lane < 32 ? (1u << lane) - 1u : 0xffffffffu;
    hi_mask = lane < 32 ? 0u : (1u << (lane - 32)) - 1u;
  }
  __device__ __forceinline__ void load(int, int) {}
  __device__ __forceinline__ void step(int, int x0, int lane) {
    const int wi = x0 >> 6;
    wb = bits[wi];
    const unsigned long long nextw = x0 + 64 < W ? bits[wi + 1] : 0ull;
    work = ((wb >> lane) & 1ull) != 0;
    const unsigned blo = ~(unsigned)wb & lo_mask, bhi = ~(unsigned)(wb >> 32) & hi_mask;
    int last = carry;
    if (blo) last = x0 + 31 - __clz(blo);
    if (bhi) last = x0 + 63 - __clz(bhi);
    start = last + 1;
    left = lane > 0 ? ((wb >> (lane - 1)) & 1ull) != 0 : prev_last;
    right = lane < 63 ? ((wb >> (lane + 1)) & 1ull) != 0 : (nextw & 1ull) != 0;
  }
  __device__ __forceinline__ void advance(int x0) {
    const unsigned long long nb = ~wb;
    prev_last = (wb >> 63) != 0;
    prev_last_start = __builtin_amdgcn_readlane(start, 63);
    if (nb) carry = x0 + 63 - __clzll(nb);
  }
};
__device__ __forceinline__ long long ccl_words(int W) { return (W + 63) >> 6; }

// Pass A -- every run start becomes its own parent and gets a zero area; the row's working bits go to the bit plane.
__global__ __launch_bounds__(256) void ccl_rows_kernel(const uint8_t* __restrict__ masks, int holes, int W,
                                                       long long rows, int* __restrict__ L,
                                                       int* __restrict__ area, unsigned long long* __restrict__ bits) {
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long long base = row * W;
  unsigned long long* brow = bits + row * ccl_words(W);
  RowScan c;
  c.init(masks + base, W, holes, lane);
  CCL_FOR_STEPS(c.load(g0, lane), {
    c.step(k, x0, lane);
    if (L != nullptr && c.work && !c.left) {      // (nullptr: ccl_strip_kernel files the run starts)
      L[base + x0 + lane] = (int)(base + x0 + lane);
      area[base + x0 + lane] = 0;
    }
    if (lane == 0) brow[x0 >> 6] = c.wb;      // (pixels beyond the row are non-working: RowScan::load)
    c.advance(x0);
  })
}

// Pass A' -- a STRIP of up to CCL_STRIP rows of one mask per workgroup: the runs of the strip are linked among themselves in
// LDS (the same lock-free union-find, on an int array indexed by the pixel's offset in the strip), then every run start gets
// its strip-local root as its parent in the global plane.  What is left for the global pass B are the row pairs that straddle
// two strips: a sixteenth of the unions, and the chains the later finds walk start flattened.  (Pass B over ALL row pairs was
// the longest kernel of the clean-up: 185 us per call on speckle masks, every union a chain of device-scope atomic round trips.)
constexpr int CCL_STRIP = 16;
constexpr int CCL_STRIP_PIX = 12288;      // LDS ints (48 KiB): strips of rows up to 768 pixels hold 16 rows, wider rows fewer
__global__ __launch_bounds__(256) void ccl_strip_kernel(const unsigned long long* __restrict__ bits, int H, int W, int strip_rows,
                                                        int strips_per_mask, int* __restrict__ L, int* __restrict__ area) {
  __shared__ int Ls[CCL_STRIP_PIX];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n = blockIdx.x / strips_per_mask, sidx = blockIdx.x - n * strips_per_mask;
  const int r0 = sidx * strip_rows, nr = min(strip_rows, H - r0);
  const long long srow = (long long)n * H + r0;             // the strip's first row
  const long long sbase = srow * W;                         // ... and first pixel
  const long long wq = ccl_words(W);
  for (int i = threadIdx.x; i < nr * W; i += 256) Ls[i] = i;      // every pixel its own parent (only run starts are ever read)
  __syncthreads();
  // the row pairs inside the strip (pass B's contact rules)
  for (int r = wave; r + 1 < nr; r += 4) {
    const int base = r * W;
    BitScan c, d;
    c.init(bits + (srow + r) * wq, W, lane);
    d.init(bits + (srow + r + 1) * wq, W, lane);
    CCL_FOR_STEPS(c.load(g0, lane); d.load(g0, lane), {
      c.step(k, x0, lane);
      d.step(k, x0, lane);
      const int d_up = __shfl_up(d.start, 1);
      const int d_left_start = lane > 0 ? d_up : d.prev_last_start;
      if (c.work) {
        const int me = base + c.start;
        const bool s = d.work, sw = d.left, se = d.right, w = c.left, e = c.right;
        if (s && !(w && sw)) uf_union(Ls, me, base + W + d.start);
        if (!s) {
          if (sw && !w) uf_union(Ls, me, base + W + d_left_start);
          if (se && !e) uf_union(Ls, me, base + W + x0 + lane + 1);
        }
      }
      c.advance(x0);
      d.advance(x0);
    })
  }
  __syncthreads();
  // run start -> its strip-local root, as a global pixel index; zero areas
  for (int r = wave; r < nr; r += 4) {
    BitScan c;
    c.init(bits + (srow + r) * wq, W, lane);
    CCL_FOR_STEPS(c.load(g0, lane), {
      c.step(k, x0, lane);
      if (c.work && !c.left) {
        const int i = r * W + x0 + lane;
        L[sbase + i] = (int)(sbase + uf_find(Ls, i));
        area[sbase + i] = 0;
      }
      c.advance(x0);
    })
  }
}

// Pass B -- links between the runs of adjacent rows (8-connectivity).  A link is issued only at the first column
// where two runs touch (not implied by a contact one column to the left), so a blob costs O(1) unions per row.
// strip_rows > 0: the row pairs inside a strip were linked by ccl_strip_kernel; wave w takes the w-th pair that straddles two
// strips (rows strip_rows - 1 | strip_rows, 2 strip_rows - 1 | 2 strip_rows, ... of every mask).
__global__ __launch_bounds__(256) void ccl_merge_kernel(const unsigned long long* __restrict__ bits, int* __restrict__ L, int H,
                                                        int W, long long rows, int strip_rows) {
  const int lane = threadIdx.x & 63;
  long long row = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (strip_rows > 0) {
    const int per_mask = (H - 1) / strip_rows;             // strip boundaries inside one mask
    if (per_mask <= 0) return;
    const long long n = row / per_mask;
    const int b = (int)(row - n * per_mask);
    if (n * H >= rows) return;
    row = n * H + (long long)(b + 1) * strip_rows - 1;
  }
  if (row >= rows || (int)(row % H) + 1 >= H) return;
  const long long base = row * W;
  BitScan c, d;
  c.init(bits + row * ccl_words(W), W, lane);
  d.init(bits + (row + 1) * ccl_words(W), W, lane);
  CCL_FOR_STEPS(c.load(g0, lane); d.load(g0, lane), {
    c.step(k, x0, lane);
    d.step(k, x0, lane);
    const int d_up = __shfl_up(d.start, 1);                                // (every lane takes part in the shuffle)
    const int d_left_start = lane > 0 ? d_up : d.prev_last_start;           // run start of the below-left pixel
    if (c.work) {
      const int me = (int)(base + c.start);
      const bool s = d.work, sw = d.left, se = d.right, w = c.left, e = c.right;
      if (s && !(w && sw)) uf_union(L, me, (int)(base + W + d.start));
      if (!s) {
        if (sw && !w) uf_union(L, me, (int)(base + W + d_left_start));
        if (se && !e) uf_union(L, me, (int)(base + W + x0 + lane + 1));   // below is not working: the run starts there
      }
    }
    c.advance(x0);
    d.advance(x0);
  })
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Pass D -- per run (its last pixel): add the run length to the component's area.  Speckled masks (thousands of runs per
// row, most of them in ONE percolating component) would send all those adds to one address; the wave therefore carries
// a (root, sum) pair across the steps of its row: lanes whose run belongs to the carried root are summed with shuffles,
// the pair is flushed with one atomic when another root takes over or the row ends.  Two candidate roots per step at most;
// what is left adds on its own.
// The pass also points every run start at its root (find with path halving: the chains the unions left behind collapse as the
// pass goes; the passes after it read a run's component with ONE load).  Until round 5 that was a pass of its own ("C"): one
// scan of the mask bytes and one launch more per call, 0.96 against 0.86 ms per ref for the six / five passes.  The unions are
// over when this pass runs, so roots no longer move; concurrent halving by other waves only shortens chains.
__global__ __launch_bounds__(256) void ccl_count_kernel(const unsigned long long* __restrict__ bits, int* __restrict__ L,
                                                        int* __restrict__ area, int W, long long rows) {
  const int lane = threadIdx.x & 63;
  const long long row = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (row >= rows) return;
  const long long base = row * W;
  BitScan c;
  c.init(bits + row * ccl_words(W), W, lane);
  int acc_root = -1, acc_sum = 0;      // wave-uniform
  CCL_FOR_STEPS(c.load(g0, lane), {
    c.step(k, x0, lane);
    bool act = c.work && !c.right;
    int r = -1;
    if (act) {
      const int i = (int)(base + c.start);
      r = uf_find_compress(L, i);
      if (r != i) L[i] = r;      // (a plain store: links only ever move towards the root, and r IS the root now)
    }
    const int len = x0 + lane - c.start + 1;
    unsigned long long todo = __ballot(act);
    if (todo) {
      if (acc_root >= 0) {
        const bool same = act && r == acc_root;
        const unsigned long long m = __ballot(same);
        if (m) { acc_sum += wave_sum(same ? len : 0); act = act && !same; todo &= ~m; }
      }
      if (todo) {     // runs of other components: the first one's root becomes the carried one if it brings company
        const int rr = __shfl(r, __ffsll((long long)todo) - 1);
        const bool same = act && r == rr;
        const unsigned long long m = __ballot(same);
        if (__popcll(m) > 1 || acc_root < 0) {
          if (acc_root >= 0 && lane == 0) atomicAdd(&area[acc_root], acc_sum);
          acc_root = rr;
          acc_sum = wave_sum(same ? len : 0);
          act = act && !same;
        }
      }
      if (act) atomicAdd(&area[r], len);
    }
    c.advance(x0);
  })
  if (acc_root >= 0 && lane == 0) atomicAdd(&area[acc_root], acc_sum);
}

// Pass E -- the roots (L[i] == i) file their component into the mask's statistics: stats[n*4+0] = number of small
// components, [1] = number of large ones, ([3]:[2]) = one 64-bit key (area << 32 | 0x7fffffff - root) maximised, i.e. the
// largest area and, among equals, the smallest root -- the reference's "first largest" component (label order = raster
// order of first pixels).  A wave's row belongs to ONE mask, so the counts and the key are reduced over the row first:
// three atomics per row, not three per component (a speckled mask has tens of thousands).
__global__ __launch_bounds__(256) void ccl_stats_kernel(const unsigned long long* __restrict__ bits, const int* __restrict__ L,
                                                        const int* __restrict__ area, int W, long long HW, long long rows,
                                                        int thresh, int* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const long long row = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (row >= rows) return;
  const long long base = row * W;
  BitScan c;
  c.init(bits + row * ccl_words(W), W, lane);
  int n_small = 0, n_large = 0;
  unsigned long long key = 0;
  CCL_FOR_STEPS(c.load(g0, lane), {
    c.step(k, x0, lane);
    if (c.work && !c.left) {
      const int i = (int)(base + x0 + lane);
      if (L[i] == i) {
        const int a = area[i];
        if (a < thresh) ++n_small; else ++n_large;
        const unsigned long long kk = ((unsigned long long)(unsigned)a << 32) | (unsigned)(0x7fffffff - i);
        key = kk > key ? kk : key;
      }
    }
    c.advance(x0);
  })
  n_small = wave_sum(n_small);
  n_large = wave_sum(n_large);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = ((unsigned long long)(unsigned)__shfl_xor((int)(key >> 32), o) << 32) |
                                     (unsigned)__shfl_xor((int)(unsigned)key, o);
    key = other > key ? other : key;
  }
  if (lane == 0) {
    const int n = (int)(base / HW);
    if (n_small) atomicAdd(&stats[n * 4 + 0], n_small);
    if (n_large) atomicAdd(&stats[n * 4 + 1], n_large);
    if (key) atomicMax((unsigned long long*)(stats + n * 4 + 2), key);
  }
}

__global__ void ccl_stats_init_kernel(int* stats, int N, int* boxes) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  stats[n * 4 + 0] = 0; stats[n * 4 + 1] = 0; stats[n * 4 + 2] = 0; stats[n * 4 + 3] = 0;
  (void)boxes;
}

// Pass F -- holes:   out = mask | (working && area < thresh)                      (fill small holes)
//           islands: out = working && area >= thresh ; if no component is large, keep the (first) largest
__global__ __launch_bounds__(256) void ccl_apply_kernel(const unsigned long long* __restrict__ bits, const int* __restrict__ L,
                                                        const int* __restrict__ area, const int* __restrict__ stats,
                                                        int holes, int H, int W, long long rows, int thresh,
                                                        uint8_t* __restrict__ out, uint8_t* __restrict__ changed,
                                                        int* __restrict__ boxes) {
  const int lane = threadIdx.x & 63;
  const long long row = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (row >= rows) return;
  const long long base = row * W;
  const unsigned long long* brow = bits + row * ccl_words(W);
  const int n = (int)(row / H);
  const int n_small = stats[n * 4 + 0];
  if (row % H == 0 && lane == 0) changed[n] = n_small != 0;
  // boxes != nullptr: the written mask's box (batched_mask_to_box, utils/amg.py:303-346) comes out of this pass instead of
  // another pass over the N*H*W bytes just written (box_kernel: 96 us per ref): the row's first / last set pixel from the
  // wave's ballots, stored as ONE word per row (min x | max x << 16; 0xffff | 0 for an empty row) that box_rows_kernel folds
  // per mask.  (Atomics per row on the mask's four box words were tried first: every row improves max y when the rows
  // arrive in order -- 164 k contended atomics per ref, 31 -> 360 us.)
  int rminx = 0xffff, rmaxx = 0;
  bool rany = false;
  auto row_bits = [&](bool o, int x0) {
    const unsigned long long b = __builtin_amdgcn_ballot_w64(o);
    if (b) {
      rminx = min(rminx, x0 + (int)__builtin_ctzll(b));
      rmaxx = max(rmaxx, x0 + 63 - (int)__builtin_clzll(b));
      rany = true;
    }
  };
  auto row_done = [&]() {
    if (boxes && lane == 0) ((unsigned*)boxes)[row] = rany ? ((unsigned)rminx | ((unsigned)rmaxx << 16)) : 0x0000ffffu;
  };
  if (n_small == 0) {            // nothing below the threshold: mask unchanged (utils/amg.py:281-282)
    for (int x0 = 0; x0 < W; x0 += 64) {
      const int x = x0 + lane;
      const bool wk = ((brow[x0 >> 6] >> lane) & 1ull) != 0;      // the mask bit is the working bit (islands) or its complement (holes)
      const bool o = x < W && (wk != (holes != 0));
      if (x < W) out[base + x] = o ? 1 : 0;
      if (boxes) row_bits(o, x0);
    }
    row_done();
    return;
  }
  const bool any_large = stats[n * 4 + 1] > 0;
  const int best = 0x7fffffff - stats[n * 4 + 2];
  BitScan c;
  c.init(brow, W, lane);
  CCL_FOR_STEPS(c.load(g0, lane), {
    c.step(k, x0, lane);
    const int x = x0 + lane;
    bool o = false;
    if (x < W) {
      const bool m = c.work != (holes != 0);
      int r = -1;
      if (c.work) r = L[base + c.start];     // the run start holds the root (or is the root)
      if (holes) o = m || (r >= 0 && area[r] < thresh);
      else o = r >= 0 && (any_large ? area[r] >= thresh : r == best);
      out[base + x] = o ? 1 : 0;
    }
    if (boxes) row_bits(o, x0);
    c.advance(x0);
  })
  row_done();
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
    const int y0 = (int)(p0 / W), xs = (int)(p0 - (long long)y0 * W);   // one division per 16 pixels
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if ((w4[e >> 2] >> (8 * (e & 3))) & 0xff) {
        int x = xs + e, y = y0;
        while (x >= W) { x -= W; ++y; }
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
// per-row words of ccl_apply_kernel -> XYXY box per mask (zeros for an empty mask): one wave per mask
__global__ __launch_bounds__(64) void box_rows_kernel(const unsigned* __restrict__ rowbox, int H, int* __restrict__ b) {
  const int n = blockIdx.x, lane = threadIdx.x;
  int minx = 0x7fffffff, miny = 0x7fffffff, maxx = -1, maxy = -1;
  for (int y = lane; y < H; y += 64) {
    const unsigned w = rowbox[(long long)n * H + y];
    const int lo = (int)(w & 0xffffu), hi = (int)(w >> 16);
    if (lo <= hi && w != 0x0000ffffu) {
      minx = min(minx, lo); maxx = max(maxx, hi);
      miny = min(miny, y); maxy = max(maxy, y);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    minx = min(minx, __shfl_xor(minx, o)); miny = min(miny, __shfl_xor(miny, o));
    maxx = max(maxx, __shfl_xor(maxx, o)); maxy = max(maxy, __shfl_xor(maxy, o));
  }
  if (lane == 0) {
    const bool any = maxx >= 0;
    b[n * 4 + 0] = any ? minx : 0; b[n * 4 + 1] = any ? miny : 0; b[n * 4 + 2] = any ? maxx : 0; b[n * 4 + 3] = any ? maxy : 0;
  }
}
__global__ void box_final_kernel(int* b, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  if (b[n * 4 + 2] < 0) { b[n * 4 + 0] = 0; b[n * 4 + 1] = 0; b[n * 4 + 2] = 0; b[n * 4 + 3] = 0; }
}

}  // namespace

extern "C" {

size_t hgl_remove_small_regions_workspace_bytes(int N, int H, int W) {
  const size_t px = (size_t)N * H * W;
  return hgl_align_up(px * sizeof(int), 256) * 2 + hgl_align_up((size_t)N * 4 * sizeof(int), 256) +
         hgl_align_up((size_t)N * H * sizeof(unsigned), 256) +                        // L, area, stats, the per-row box words,
         hgl_align_up((size_t)N * H * ((W + 63) / 64) * sizeof(unsigned long long), 256);   // the working-bit plane
}

static int remove_small_regions_impl(const uint8_t* masks, int N, int H, int W, int area_thresh, int holes, uint8_t* out,
                                     uint8_t* changed, int32_t* boxes_xyxy, void* workspace, size_t workspace_bytes, void* stream);

int hgl_remove_small_regions(const uint8_t* masks, int N, int H, int W, int area_thresh, int holes, uint8_t* out,
                             uint8_t* changed, void* workspace, size_t workspace_bytes, void* stream) {
  return remove_small_regions_impl(masks, N, H, W, area_thresh, holes, out, changed, nullptr, workspace, workspace_bytes, stream);
}

int hgl_remove_small_regions_boxes(const uint8_t* masks, int N, int H, int W, int area_thresh, int holes, uint8_t* out,
                                   uint8_t* changed, int32_t* boxes_xyxy, void* workspace, size_t workspace_bytes, void* stream) {
  HGL_REQUIRE(boxes_xyxy, "remove_small_regions_boxes: null boxes");
  HGL_REQUIRE(W <= 65535, "remove_small_regions_boxes: rows of more than 65535 pixels (the per-row box words hold 16-bit columns)");
  return remove_small_regions_impl(masks, N, H, W, area_thresh, holes, out, changed, boxes_xyxy, workspace, workspace_bytes, stream);
}

static int remove_small_regions_impl(const uint8_t* masks, int N, int H, int W, int area_thresh, int holes, uint8_t* out,
                                     uint8_t* changed, int32_t* boxes_xyxy, void* workspace, size_t workspace_bytes, void* stream) {
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
  unsigned* rowbox = ar.take<unsigned>((size_t)N * H);
  unsigned long long* bits = ar.take<unsigned long long>((size_t)N * H * ((W + 63) / 64));
  const long long HW = (long long)H * W;
  const long long rows = (long long)N * H;
  const dim3 grid((unsigned)((rows + 3) / 4));
  const unsigned long long* cbits = bits;
  // strips of rows linked in LDS, then only the pairs between strips globally (rows too wide for the LDS plane: the global pass alone)
  const int strip_rows = W <= CCL_STRIP_PIX / 2 ? (CCL_STRIP_PIX / W < CCL_STRIP ? CCL_STRIP_PIX / W : CCL_STRIP) : 0;
  hipLaunchKernelGGL(ccl_stats_init_kernel, dim3((N + 255) / 256), dim3(256), 0, st, stats, N, (int*)nullptr);
  if (strip_rows >= 2) {
    const int strips = (H + strip_rows - 1) / strip_rows;
    hipLaunchKernelGGL(ccl_rows_kernel, grid, dim3(256), 0, st, masks, holes, W, rows, (int*)nullptr, (int*)nullptr, bits);
    hipLaunchKernelGGL(ccl_strip_kernel, dim3((unsigned)(N * strips)), dim3(256), 0, st, cbits, H, W, strip_rows, strips, L, area);
    const long long pairs = (long long)N * ((H - 1) / strip_rows);
    if (pairs > 0)
      hipLaunchKernelGGL(ccl_merge_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, st, cbits, L, H, W, rows, strip_rows);
  } else {
    hipLaunchKernelGGL(ccl_rows_kernel, grid, dim3(256), 0, st, masks, holes, W, rows, L, area, bits);
    hipLaunchKernelGGL(ccl_merge_kernel, grid, dim3(256), 0, st, cbits, L, H, W, rows, 0);
  }
  hipLaunchKernelGGL(ccl_count_kernel, grid, dim3(256), 0, st, cbits, L, area, W, rows);
  hipLaunchKernelGGL(ccl_stats_kernel, grid, dim3(256), 0, st, cbits, (const int*)L, (const int*)area, W, HW, rows,
                     area_thresh, stats);
  hipLaunchKernelGGL(ccl_apply_kernel, grid, dim3(256), 0, st, cbits, (const int*)L, (const int*)area, (const int*)stats, holes,
                     H, W, rows, area_thresh, out, changed, boxes_xyxy ? (int*)rowbox : nullptr);
  if (boxes_xyxy) hipLaunchKernelGGL(box_rows_kernel, dim3(N), dim3(64), 0, st, (const unsigned*)rowbox, H, (int*)boxes_xyxy);
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
