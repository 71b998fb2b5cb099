// Token -> image attention of the SAM mask decoder ON THE RAW IMAGE TOKENS (split-fp16 matrix-core mode).
//
// Attention.forward (modeling/transformer.py:218-240) for the decoder's cross_attn_token_to_image / final_attn_token_to_image
// (transformer.py:126-131, :97-105): 7 prompt tokens attend to the 4096 image tokens of THEIR prompt, 8 heads of 16 channels.
// As written, k = (keys + pe) W_k and v = keys W_v are projections of all 4096 x 256 image tokens of every prompt: a GEMM that
// reads the per-prompt token planes (4.2 MB per prompt) and writes k | v (4.2 MB), which the attention then reads (4.2 MB) --
// 12.6 MB of HBM traffic per prompt and attention, at the 4 TB/s these kernels reach the time of the whole step.
//
// The algebra lets the 7 tokens be projected instead of the 4096.  Per head h (W_k[h] = rows 16h .. 16h+15 of W_k, [16, 256]):
//   q_h k_h^T = q_h (W_k[h] (keys + pe)^T + b_k[h] 1^T) = (q_h W_k[h]) keys^T + (q_h W_k[h]) pe^T + const      (const: the same for
//   every key -> cancels in the soft-max), and   P_h v_h = P_h (keys W_v[h]^T + 1 b_v[h]) = (P_h keys) W_v[h]^T + b_v[h]   (rows of P sum to 1).
// So:  Qk[p, h*7 + t, :] = scale * q[p, t, 16h:16h+16] W_k[h]                     [P, 56, 256]   t2i_fold_q_kernel
//      bias[p, r, key]   = Qk[p, r, :] . pe[key, :]                               [P, 56, HW]    one split-fp16 GEMM (pe as the weight)
//      A[p, r, :]        = soft-max_key(Qk[p, r, :] . keys[p, key, :] + bias) . keys[p, :, :]     t2i_raw_attn_kernel: the planes read ONCE
//      att[p, t, 16h+i]  = A[p, h*7 + t, :] . W_v[16h + i, :] + b_v[16h + i]      [P, 7, 128]    t2i_unfold_v_kernel
// and the k | v projection GEMM, its 4.2 MB of output per prompt and their re-read disappear: 4.2 + 0.9 (bias) x 2 instead
// of 12.6 MB per prompt and attention.  Exact in real arithmetic; in floating point every product is the split-fp16 triple
// (hi*hi + hi*lo + lo*hi into fp32) as everywhere else: equal to the projected path to fp32 rounding (tests/test_gpu_sam.py).
//
// t2i_raw_attn_kernel.  One workgroup of four waves per prompt (per eighth of a prompt's keys in launches of <= 128 prompts, the
// pieces joined by t2i_unfold_v_kernel); wave w owns query rows 16w .. 16w+15 (of the 56) and ALL 256
// channels: S^T = K Q'^T (v_mfma_f32_16x16x32_f16: first operand rows of the staged key chunk, second operand the wave's 16
// query rows, register-resident for the whole prompt: 8 K steps x hi / lo = 64 VGPRs) -- the accumulator then holds, per lane,
// query (lane & 15) and keys 4g .. 4g+3 (g = lane >> 4) of each 16-key tile, which is exactly the B operand of
// O'^T = K^T P^T when the K step's eight key slots of lane group g are {4g..4g+3, 16+4g..16+4g+3}; the first operand K^T comes
// out of the same LDS image through two transposing reads (ds_read_b64_tr_b16) per fragment with those rows.  No lane movement,
// no second copy of the chunk, no exchange between the waves.  Key chunks of 32 go global -> LDS by LDS-DMA into two stages, one
// barrier per chunk; two workgroups per CU (68 KB of LDS each) cover each other's waits.  Online soft-max in base 2 with the
// rescaling skipped while no row's maximum moves.
//
// The image -> token step of layer 1 with W_q / W_o folded into the same 7 tokens (i2t_prep_kernel, dec_i2t_fold_kernel) is
// described at those kernels, further down.
#include "hgl_common.h"

bool hgl_get_split_weight(const float* W, const void** hi, const void** lo, int* scale_log2, int* N, int* K);

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4v __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int T2I_ROWS = 56;                       // 8 heads x 7 tokens
constexpr int T2I_C = 256;                         // channels of an image token
constexpr int T2I_CHUNK = 32;                      // keys per staged chunk
constexpr int T2I_SLOTS = 33;                      // 16-byte slots per staged key row: 32 + 1 pad (row pitch 528 B: the 16 rows of a
constexpr int T2I_PITCH = T2I_SLOTS * 16;          //   ds_read_b128 lane group land in 16 different bank quads)
constexpr int T2I_PLANE_NI = 17;                   // LDS-DMA wave-instructions per plane: 32 x 33 = 1056 slots -> 17 x 64
constexpr int T2I_PLANE = T2I_PLANE_NI * 1024;     // bytes of one staged plane
constexpr int T2I_STAGE = 2 * T2I_PLANE;           // hi | lo
constexpr int T2I_NI = 2 * T2I_PLANE_NI;           // wave-instructions per chunk
constexpr int T2I_NJ = (T2I_NI + 3) / 4;           // per wave
constexpr size_t T2I_LDS = 2 * (size_t)T2I_STAGE;  // two stages: 69 632 B
constexpr float T2I_LOG2E = 1.4426950408889634f;

// ds_read_b64_tr_b16 (EXEC must be all ones at the call)
__device__ __forceinline__ f16x4 t2i_tr4(const unsigned char* p) {
  typedef __attribute__((address_space(3))) fp16x4v lds_v;
  const fp16x4v v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_v*)p);
  f16x4 r;
  __builtin_memcpy(&r, &v, 8);
  return r;
}

// One LDS-DMA wave-instruction: lane i copies 16 B from sbase + voff(i) to LDS byte lds_addr + 16 * i (see gemm_f16x3.hip).
// Not counted by the compiler on vmcnt: the consumer waits with an explicit s_waitcnt.
__device__ __forceinline__ void t2i_glds16(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_addr)
               : "memory");
}

// Qk[p, h*T + t, c] = scale * sum_j q[p, t, 16h + j] * Wk[16h + j, c]   [P*56, 256] fp32 (the caller splits it into fp16 hi / lo
// planes with the guarded split of gemm_f16x3.hip: its range counter is the one the evaluator watches)
__global__ __launch_bounds__(256) void t2i_fold_q_kernel(const float* __restrict__ q, const float* __restrict__ Wk, float scale,
                                                         float* __restrict__ Qk) {
  __shared__ float qs[7 * 128];
  const int p = blockIdx.x, c = threadIdx.x;
  for (int i = threadIdx.x; i < 7 * 128; i += 256) qs[i] = q[(long long)p * 7 * 128 + i];
  __syncthreads();
  for (int h = 0; h < 8; ++h) {
    float w[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) w[j] = Wk[(16 * h + j) * T2I_C + c];
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) a = fmaf(qs[t * 128 + 16 * h + j], w[j], a);
      Qk[((long long)p * T2I_ROWS + h * 7 + t) * T2I_C + c] = a * scale;
    }
  }
}

// att[p, t, 16h + i] = A[p, h*7 + t, :] . Wv[16h + i, :] + bv[16h + i]
// ns > 1: A is not the attended rows but ns partial results per prompt, one per key range (small prompt batches, see
// t2i_raw_attn_kernel): part = [P][ns] x { O[56][256] un-normalised, ml[56][2] = (row maximum, row sum) }; the rows are put
// together here, on the way into LDS: A[q, :] = sum_s O_s[q, :] 2^((m_s - M) log2 e) / sum_s l_s 2^((m_s - M) log2 e).
constexpr int T2I_PART = T2I_ROWS * T2I_C + T2I_ROWS * 2;      // floats of one partial
__global__ __launch_bounds__(256) void t2i_unfold_v_kernel(const float* __restrict__ A, int ns, const float* __restrict__ Wv,
                                                           const float* __restrict__ bv, float* __restrict__ att) {
  __shared__ float As[T2I_ROWS * T2I_C];      // 57 344 B
  __shared__ float wgt[8 * T2I_ROWS];         // per (key range, row): 2^((m_s - M) log2 e) / L
  const int p = blockIdx.x;
  if (ns > 1) {
    const float* const part = A + (long long)p * ns * T2I_PART;
    if (threadIdx.x < T2I_ROWS) {
      const int q = threadIdx.x;
      float M = -INFINITY;
      for (int s = 0; s < ns; ++s) M = fmaxf(M, part[(long long)s * T2I_PART + T2I_ROWS * T2I_C + 2 * q]);
      float L = 0.f;
      for (int s = 0; s < ns; ++s) {
        const float* ml = part + (long long)s * T2I_PART + T2I_ROWS * T2I_C + 2 * q;
        const float e = exp2f((ml[0] - M) * T2I_LOG2E);
        wgt[s * T2I_ROWS + q] = e;
        L += ml[1] * e;
      }
      const float inv = 1.0f / L;
      for (int s = 0; s < ns; ++s) wgt[s * T2I_ROWS + q] *= inv;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T2I_ROWS * T2I_C / 4; i += 256) {
      const int q = i >> 6;      // 64 four-float pieces per row
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < ns; ++s) {
        const f32x4 o = ((const f32x4*)(part + (long long)s * T2I_PART))[i];
        const float w = wgt[s * T2I_ROWS + q];
        acc[0] = fmaf(o[0], w, acc[0]); acc[1] = fmaf(o[1], w, acc[1]); acc[2] = fmaf(o[2], w, acc[2]); acc[3] = fmaf(o[3], w, acc[3]);
      }
      ((f32x4*)As)[i] = acc;
    }
  } else {
    for (int i = threadIdx.x; i < T2I_ROWS * T2I_C / 4; i += 256)
      ((f32x4*)As)[i] = ((const f32x4*)(A + (long long)p * T2I_ROWS * T2I_C))[i];
  }
  __syncthreads();
  const int col = threadIdx.x & 127, h = col >> 4;
  const float* wrow = Wv + (long long)col * T2I_C;
  for (int t = threadIdx.x >> 7; t < 7; t += 2) {
    const float* arow = As + (h * 7 + t) * T2I_C;
    float a = 0.f;
    for (int c0 = 0; c0 < T2I_C; c0 += 4) {
      const f32x4 wv = *(const f32x4*)(wrow + c0);
      const f32x4 av = *(const f32x4*)(arow + c0);
      a = fmaf(av[0], wv[0], a); a = fmaf(av[1], wv[1], a); a = fmaf(av[2], wv[2], a); a = fmaf(av[3], wv[3], a);
    }
    att[((long long)p * 7 + t) * 128 + col] = a + bv[col];
  }
}

struct T2IArgs {
  const _Float16 *Qh, *Ql;     // [P*56, 256] scaled folded queries
  const float* bias;           // [P*56, HW]  Qk . pe^T (scaled)
  const _Float16 *Kh, *Kl;     // [P*HW, 256] image-token planes
  float* out;                  // [P*56, 256]; with ns > 1: the partials [P][ns] x T2I_PART floats (see t2i_unfold_v_kernel)
  int HW, ns;                  // ns key ranges per prompt, one workgroup each (grid.y)
};

__global__ __launch_bounds__(256, 2) void t2i_raw_attn_kernel(T2IArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char t2i_smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int p = blockIdx.x;
  // Small prompt batches (the RefCOCO configuration decodes 64 prompts per image): one workgroup per prompt leaves three
  // quarters of the CUs without work and each of the others alone with its chunk loop (290 us per launch of 64 prompts, 63 us
  // per 64 at 1024).  The key range is then cut into ns = 8 pieces, a workgroup each; the pieces leave their un-normalised
  // rows with (maximum, sum) and t2i_unfold_v_kernel puts them together.
  const int nchunks = a.HW / T2I_CHUNK / a.ns, chunk0 = (int)blockIdx.y * nchunks;
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)t2i_smem;

  // ---- this wave's DMA pieces: instruction i (0 .. 33) = plane i / 17, slots 64 (i % 17) .. +63 of that plane ----
  const unsigned char* const kh_p = (const unsigned char*)(a.Kh + (long long)p * a.HW * T2I_C);
  const unsigned char* const kl_p = (const unsigned char*)(a.Kl + (long long)p * a.HW * T2I_C);
  unsigned voff[T2I_NJ];
#pragma unroll
  for (int j = 0; j < T2I_NJ; ++j) {
    const int i = wave + 4 * j;
    const int slot = (i % T2I_PLANE_NI) * 64 + lane;
    const int key = min(slot / T2I_SLOTS, T2I_CHUNK - 1), col = min(slot % T2I_SLOTS, 31);      // pad slots re-read a neighbour
    voff[j] = (unsigned)(key * T2I_C + col * 8) * 2u;
  }
  auto issue = [&](int chunk, int stage) {
#pragma unroll
    for (int j = 0; j < T2I_NJ; ++j) {
      const int i = wave + 4 * j;
      if (i < T2I_NI) {
        const int plane = i / T2I_PLANE_NI;
        t2i_glds16(plane ? kl_p : kh_p, voff[j] + (unsigned)chunk * (T2I_CHUNK * T2I_C * 2),
                   lds0 + stage * T2I_STAGE + plane * T2I_PLANE + (i % T2I_PLANE_NI) * 1024);
      }
    }
  };
  issue(chunk0, 0);

  // ---- the wave's 16 query rows: second operand of S^T = K Q'^T, lane (query r, group g) holds channels 32 ks + 8 g .. +7 ----
  const int qrow = min(16 * wave + r, T2I_ROWS - 1);
  const long long qbase = ((long long)p * T2I_ROWS + qrow) * T2I_C + 8 * g;
  f16x8 qh[8], ql[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    qh[ks] = *(const f16x8*)(a.Qh + qbase + 32 * ks);
    ql[ks] = *(const f16x8*)(a.Ql + qbase + 32 * ks);
  }
  const float* const brow = a.bias + ((long long)p * T2I_ROWS + qrow) * a.HW + chunk0 * T2I_CHUNK + 4 * g;
  f32x4 bnext[2];
  bnext[0] = *(const f32x4*)(brow);
  bnext[1] = *(const f32x4*)(brow + 16);

  f32x4 oacc[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) oacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, lsum = 0.f;

  // LDS offsets of this lane's fragment reads inside a stage's hi plane
  const unsigned row_off = (unsigned)(r * T2I_PITCH + 16 * g);                                   // + 16-row tile, + 64 ks
  const unsigned tr_off = (unsigned)((4 * g + (r >> 2)) * T2I_PITCH + 8 * (r & 3));              // + 16 rows (second read), + 32 dt

  for (int c = 0; c < nchunks; ++c) {
    const int stage = c & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // chunk c has landed (and the bias vectors requested a chunk ago)
    __syncthreads();                                      // ... for every wave; and nobody reads the other stage any more
    if (c + 1 < nchunks) issue(chunk0 + c + 1, stage ^ 1);
    const f32x4 b0 = bnext[0], b1 = bnext[1];
    if (c + 1 < nchunks) {
      bnext[0] = *(const f32x4*)(brow + (c + 1) * T2I_CHUNK);
      bnext[1] = *(const f32x4*)(brow + (c + 1) * T2I_CHUNK + 16);
    }
    const unsigned char* const sh = t2i_smem + stage * T2I_STAGE;
    const unsigned char* const sl = sh + T2I_PLANE;

    // ---- S^T tiles (keys 0-15 / 16-31 of the chunk) x (the wave's 16 queries) ----
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const f16x8 k0h = *(const f16x8*)(sh + row_off + 64 * ks), k0l = *(const f16x8*)(sl + row_off + 64 * ks);
      const f16x8 k1h = *(const f16x8*)(sh + row_off + 16 * T2I_PITCH + 64 * ks);
      const f16x8 k1l = *(const f16x8*)(sl + row_off + 16 * T2I_PITCH + 64 * ks);
      s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0l, qh[ks], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1l, qh[ks], s1, 0, 0, 0);
      s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, ql[ks], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1h, ql[ks], s1, 0, 0, 0);
      s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, qh[ks], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1h, qh[ks], s1, 0, 0, 0);
    }
    // ---- soft-max of the lane's query over the chunk: its keys 4g .. 4g+3 and 16+4g .. 16+4g+3; the other 24 keys of the
    // query sit in the lanes r + 16, r + 32, r + 48 ----
    float x[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[i] = s0[i] + b0[i]; x[4 + i] = s1[i] + b1[i]; }
    float mx = fmaxf(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])), fmaxf(fmaxf(x[4], x[5]), fmaxf(x[6], x[7])));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mnew = fmaxf(m, mx);
    if (__any(mnew > m)) {      // (wave-uniform: the branch is skipped once the maxima have settled)
      const float alpha = exp2f((m - mnew) * T2I_LOG2E);
      lsum *= alpha;
#pragma unroll
      for (int d = 0; d < 16; ++d)
#pragma unroll
        for (int i = 0; i < 4; ++i) oacc[d][i] *= alpha;
      m = mnew;
    }
    f16x8 ph, pl;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float e = exp2f((x[i] - m) * T2I_LOG2E);
      lsum += e;
      _Float16 hh, ll;
      hgl_split_hi_lo(e, hh, ll);
      ph[i] = hh;
      pl[i] = ll;
    }
    // ---- O'^T (channels x queries) += K^T P^T: first operand from the same image through the transposing read ----
#pragma unroll
    for (int d = 0; d < 16; ++d) {
      const f16x4 a0h = t2i_tr4(sh + tr_off + 32 * d), a1h = t2i_tr4(sh + tr_off + 16 * T2I_PITCH + 32 * d);
      const f16x4 a0l = t2i_tr4(sl + tr_off + 32 * d), a1l = t2i_tr4(sl + tr_off + 16 * T2I_PITCH + 32 * d);
      const f16x8 kth = {a0h[0], a0h[1], a0h[2], a0h[3], a1h[0], a1h[1], a1h[2], a1h[3]};
      const f16x8 ktl = {a0l[0], a0l[1], a0l[2], a0l[3], a1l[0], a1l[1], a1l[2], a1l[3]};
      oacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ktl, ph, oacc[d], 0, 0, 0);
      oacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kth, pl, oacc[d], 0, 0, 0);
      oacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kth, ph, oacc[d], 0, 0, 0);
    }
  }
  // ---- normalise and store: lane (query r, group g) holds channels 16 d + 4 g .. +3 of its query ----
  lsum += __shfl_xor(lsum, 16);
  lsum += __shfl_xor(lsum, 32);
  if (a.ns > 1) {      // this key range's share: un-normalised rows + (maximum, sum)
    if (16 * wave + r < T2I_ROWS) {
      float* const part = a.out + ((long long)p * a.ns + blockIdx.y) * T2I_PART;
      float* const orow = part + (16 * wave + r) * T2I_C + 4 * g;
#pragma unroll
      for (int d = 0; d < 16; ++d) *(f32x4*)(orow + 16 * d) = oacc[d];
      if (g == 0) {
        part[T2I_ROWS * T2I_C + 2 * (16 * wave + r)] = m;
        part[T2I_ROWS * T2I_C + 2 * (16 * wave + r) + 1] = lsum;
      }
    }
    return;
  }
  const float inv = 1.0f / lsum;
  if (16 * wave + r < T2I_ROWS) {
    float* const orow = a.out + ((long long)p * T2I_ROWS + 16 * wave + r) * T2I_C + 4 * g;
#pragma unroll
    for (int d = 0; d < 16; ++d) {
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = oacc[d][i] * inv;
      *(f32x4*)(orow + 16 * d) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Image -> token attention of layer 1 with the SAME fold (transformer.py:139-150): per image token, 8 heads attend to the 7
// prompt tokens; q = (keys + pe) W_q projected all 4096 x 256 image tokens of every prompt (a GEMM: 4.2 MB read + 2.1 MB written
// + 2.1 MB re-read per prompt).  With n = j*8 + h (token j, head h):
//   score[token, n] = (keys[token] + pe[token]) . K'[n] + cb[n],   K'[n, :] = scale * k_tok[j, 16h:16h+16] W_q[16h:16h+16, :],
//                                                                   cb[n]    = scale * k_tok[j, 16h:16h+16] . b_q[16h:16h+16]
//   out[token, :]   = sum_n P[token, n] U[n, :] + b_o,              U[n, :]  = W_o[:, 16h:16h+16] v_tok[j, 16h:16h+16]
// (P = soft-max over j within a head).  Both products are [64 tokens x 256] x [256 x 56] / [64 x 56] x [56 x 256] per tile
// against per-PROMPT matrices of 56 x 256: the q projection, its output and the 16-channel dot products disappear, the token
// planes are read once (as the MFMA operand and, reconstructed as hi + lo, as the residual), nothing else per-prompt but the
// positional term pek = K' pe^T (one GEMM, 0.9 MB per prompt).
// i2t_prep_kernel: K' (rows n), cb, and U in the fragment order of the second product's first operand, K' and U as fp16 hi / lo
// planes (guarded split: this file's range counter is part of hgl_split_overflow_count).
__global__ __launch_bounds__(256) void i2t_prep_kernel(const float* __restrict__ k1, const float* __restrict__ v1,
                                                       const float* __restrict__ Wq, const float* __restrict__ bq,
                                                       const float* __restrict__ Wo, float scale, _Float16* __restrict__ Kh,
                                                       _Float16* __restrict__ Kl, float* __restrict__ cb,
                                                       _Float16* __restrict__ Uh, _Float16* __restrict__ Ul) {
  __shared__ float ks[7 * 128], vs[7 * 128];
  const int p = blockIdx.x, c = threadIdx.x;
  for (int i = threadIdx.x; i < 7 * 128; i += 256) {
    ks[i] = k1[(long long)p * 7 * 128 + i];
    vs[i] = v1[(long long)p * 7 * 128 + i];
  }
  __syncthreads();
  if (c < 56) {      // cb[n], n = j*8 + h
    const int j = c >> 3, h = c & 7;
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a = fmaf(ks[j * 128 + 16 * h + i], bq[16 * h + i], a);
    cb[(long long)p * 56 + c] = a * scale;
  }
  const long long ub = (long long)p * 16 * 2 * 64 * 8;
  float amax = 0.f;
  for (int h = 0; h < 8; ++h) {
    float wq[16], wo[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) wq[i] = Wq[(16 * h + i) * T2I_C + c];
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const f32x4 w4 = *(const f32x4*)(Wo + (long long)c * 128 + 16 * h + 4 * i4);
      wo[4 * i4] = w4[0]; wo[4 * i4 + 1] = w4[1]; wo[4 * i4 + 2] = w4[2]; wo[4 * i4 + 3] = w4[3];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int n = j * 8 + h;
      float a = 0.f, u = 0.f;
      if (j < 7) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          a = fmaf(ks[j * 128 + 16 * h + i], wq[i], a);
          u = fmaf(vs[j * 128 + 16 * h + i], wo[i], u);
        }
        _Float16 x, y;
        hgl_split_hi_lo(a * scale, x, y, amax);
        Kh[((long long)p * 56 + n) * T2I_C + c] = x;
        Kl[((long long)p * 56 + n) * T2I_C + c] = y;
      }
      // U[n, c] at its place in the fragment of (channel tile c >> 4, K step n >> 5): lane 16 g + (c & 15), slot jj, where
      // the K step's 32 values of n sit as {4g .. 4g+3 | 16+4g .. 16+4g+3} in lane group g (the accumulator order of the
      // first product); n >= 56 (j == 7) is padding: zeros
      const int m = n & 31, hi16 = m >> 4, g = (m & 15) >> 2, jj = 4 * hi16 + (m & 3);
      _Float16 x, y;
      hgl_split_hi_lo(u, x, y, amax);
      const long long o = ub + (((c >> 4) * 2 + (n >> 5)) * 64 + 16 * g + (c & 15)) * 8 + jj;
      Uh[o] = x;
      Ul[o] = y;
    }
  }
  hgl_split_commit(amax);
}

struct I2TFArgs {
  const _Float16 *Xh, *Xl;     // [P*HW, 256] image-token planes (input of the layer)
  const _Float16 *Kh, *Kl;     // [P*56, 256] K' planes, rows n = j*8 + h
  const float* pek;            // [P*56, HW]  K' pe^T
  const float* cb;             // [P*56]
  const _Float16 *Uh, *Ul;     // [P][16][2][64][8] U fragments
  const float *bo, *ln_w, *ln_b;
  float eps;
  _Float16 *oh, *ol;           // [P*HW, 256] normalised rows as planes
  int HW;
};

constexpr int I2TF_WAVES = 8;
constexpr int I2TF_ROWS = 16 * I2TF_WAVES;          // tokens per workgroup: 16 per wave
constexpr size_t I2TF_LDS = 65536;                  // 64 KiB, used three times over: K' (56 rows x 528 B x 2 planes), then the U
                                                    // fragments (2 x 32 KiB), then per wave 8 output rows x 256 halfs, hi and lo
constexpr int I2TF_KNI = 29;                        // LDS-DMA wave-instructions per K' plane: 56 x 33 = 1848 slots -> 29 x 64
constexpr int I2TF_KPLANE = I2TF_KNI * 1024;

// (Registers: 123 as written, two workgroups per CU.  Asking for four waves per SIMD in the launch bounds gives 108 and a
// schedule that takes 1.57 ms where this one takes 1.33: the bound stays at two, the count is checked by tools/isa_stats.py.)
__global__ __launch_bounds__(64 * I2TF_WAVES, 2) void dec_i2t_fold_kernel(I2TFArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char i2tf_smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int p = blockIdx.y;
  const long long row0 = (long long)p * a.HW + (long long)blockIdx.x * I2TF_ROWS + 16 * wave;      // the wave's first token row
  const int tok0 = blockIdx.x * I2TF_ROWS + 16 * wave;

  // Every wave needs ALL of K' (first product) and ALL of U (second product): read from L2 by each wave of each tile that was
  // 512 KB per 64 tokens and the launch ran at the L2's rate (3.4 ms per 529 prompts).  Both go through the one 64 KiB LDS
  // region instead, one after the other (K' by per-lane gathers into rows of 33 slots, U linearly: it is stored in fragment
  // order), each requested under work that does not need it: 121 KB per tile.
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)i2tf_smem;
  {
    const unsigned char* const kh_p = (const unsigned char*)(a.Kh + (long long)p * 56 * T2I_C);
    const unsigned char* const kl_p = (const unsigned char*)(a.Kl + (long long)p * 56 * T2I_C);
#pragma unroll
    for (int j = 0; j < (2 * I2TF_KNI + I2TF_WAVES - 1) / I2TF_WAVES; ++j) {
      const int i = wave + I2TF_WAVES * j;
      if (i < 2 * I2TF_KNI) {
        const int plane = i / I2TF_KNI, slot = (i % I2TF_KNI) * 64 + lane;
        const int row = min(slot / T2I_SLOTS, 55), col = min(slot % T2I_SLOTS, 31);
        t2i_glds16(plane ? kl_p : kh_p, (unsigned)(row * T2I_C + col * 8) * 2u, lds0 + plane * I2TF_KPLANE + (i % I2TF_KNI) * 1024);
      }
    }
  }
  // ---- first product: S^T[n, token] = K' X^T; second operand = the wave's 16 token rows ----
  f16x8 xh[8], xl[8];
  {
    const long long xb = (row0 + r) * T2I_C + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      xh[ks] = *(const f16x8*)(a.Xh + xb + 32 * ks);
      xl[ks] = *(const f16x8*)(a.Xl + xb + 32 * ks);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x4 s[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    s[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* const kb = i2tf_smem + min(16 * nt + r, 55) * T2I_PITCH + 16 * g;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const f16x8 kh = *(const f16x8*)(kb + 64 * ks), kl = *(const f16x8*)(kb + I2TF_KPLANE + 64 * ks);
      s[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, xh[ks], s[nt], 0, 0, 0);
      s[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, xl[ks], s[nt], 0, 0, 0);
      s[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, xh[ks], s[nt], 0, 0, 0);
    }
  }
  // The residual (the token's own row, hi + lo) enters the accumulators of the second product here, while the row's
  // fragments of the first product are still in registers: D^T[channel, token] = E X^T with E the 16 x 32 selection matrix
  // of channel tile d inside K step d >> 1 (1.0 x fp16 into fp32: exact).  Loading the rows again in the accumulators'
  // layout was 32 eight-byte loads per lane at a 512-byte stride.
  f32x4 acc[16];
  {
    f16x8 sel[2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int j = 0; j < 8; ++j) sel[e][j] = (8 * g + j == 16 * e + r) ? (_Float16)1.0f : (_Float16)0.0f;
#pragma unroll
    for (int d = 0; d < 16; ++d) {
      acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sel[d & 1], xl[d >> 1], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sel[d & 1], xh[d >> 1], acc[d], 0, 0, 0);
    }
  }
  __syncthreads();      // every wave is done with K': the region takes the U fragments (requested under the soft-max)
  {
    const unsigned char* const uh_p = (const unsigned char*)(a.Uh + (long long)p * 16384);
    const unsigned char* const ul_p = (const unsigned char*)(a.Ul + (long long)p * 16384);
#pragma unroll
    for (int j = 0; j < 64 / I2TF_WAVES; ++j) {
      const int i = wave + I2TF_WAVES * j;      // 64 pieces of 1 KiB: plane i / 32, fragment i % 32
      t2i_glds16(i < 32 ? uh_p : ul_p, (unsigned)((i & 31) * 1024 + lane * 16), lds0 + i * 1024);
    }
  }
  // ---- + positional term + bias term; soft-max over the 7 tokens of a head.  The lane holds, for token r of the wave,
  // n = 16 nt + 4 g + i: head 4 (g & 1) + i, token j = 2 nt + (g >> 1) -- the other parity of j sits 32 lanes away ----
  f16x8 ph[2], pl[2];
  {
    float x[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = 16 * nt + 4 * g + i, nc = min(n, 55);
        const float v = s[nt][i] + a.pek[((long long)p * 56 + nc) * a.HW + tok0 + r] + a.cb[(long long)p * 56 + nc];
        x[nt][i] = n < 56 ? v : -INFINITY;
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float m = fmaxf(fmaxf(x[0][i], x[1][i]), fmaxf(x[2][i], x[3][i]));
      m = fmaxf(m, __shfl_xor(m, 32));
      float l = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        x[nt][i] = exp2f((x[nt][i] - m) * T2I_LOG2E);
        l += x[nt][i];
      }
      l += __shfl_xor(l, 32);
      const float inv = 1.0f / l;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        _Float16 hh, ll;
        hgl_split_hi_lo(x[nt][i] * inv, hh, ll);
        ph[nt >> 1][4 * (nt & 1) + i] = hh;      // K step nt >> 1: slots 0-3 from the even tile, 4-7 from the odd one
        pl[nt >> 1][4 * (nt & 1) + i] = ll;
      }
    }
  }
  // ---- second product: D^T[channel, token] = U^T P^T; first operand from the fragment-ordered U planes ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    const unsigned char* const uf = i2tf_smem + lane * 16;
#pragma unroll
    for (int d = 0; d < 16; ++d) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const f16x8 fh = *(const f16x8*)(uf + (d * 2 + ks) * 1024), fl = *(const f16x8*)(uf + 32768 + (d * 2 + ks) * 1024);
        acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl, ph[ks], acc[d], 0, 0, 0);
        acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, pl[ks], acc[d], 0, 0, 0);
        acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, ph[ks], acc[d], 0, 0, 0);
      }
    }
  }
  // ---- + b_o; LayerNorm over the token's 256 channels: this lane holds channels 16 d + 4 g + i, the other three quarters
  // sit 16 / 32 / 48 lanes away ----
  float sum = 0.f;
#pragma unroll
  for (int d = 0; d < 16; ++d) {
    const f32x4 bo = *(const f32x4*)(a.bo + 16 * d + 4 * g);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[d][i] += bo[i];
      sum += acc[d][i];
    }
  }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  const float mean = sum * (1.f / 256.f);
  float var = 0.f;
#pragma unroll
  for (int d = 0; d < 16; ++d)
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float dd = acc[d][i] - mean; var += dd * dd; }
  var += __shfl_xor(var, 16);
  var += __shfl_xor(var, 32);
  const float rstd = rsqrtf(var * (1.f / 256.f) + a.eps);
  // ---- normalise, split, and leave through LDS so that a store instruction writes whole rows: the wave's 16 rows x 512 B
  // per plane, 8-byte chunks XOR-ed with the row ----
  __syncthreads();      // every wave is done with U: the region becomes the four waves' staging areas
  unsigned char* const sth = i2tf_smem + wave * (2 * 8 * 512);      // 8 rows x 512 B, hi; then lo
  unsigned char* const stl = sth + 8 * 512;
#pragma unroll
  for (int half = 0; half < 2; ++half) {      // rows 0-7, then 8-15 of the wave's 16 (the staging area holds eight)
    if ((r >> 3) == half) {
#pragma unroll
      for (int d = 0; d < 16; ++d) {
        const int ch = 16 * d + 4 * g;
        const f32x4 lw = *(const f32x4*)(a.ln_w + ch), lb = *(const f32x4*)(a.ln_b + ch);
        f16x4 hi4, lo4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float y = (acc[d][i] - mean) * rstd * lw[i] + lb[i];
          _Float16 hh, ll;
          hgl_split_hi_lo(y, hh, ll);
          hi4[i] = hh;
          lo4[i] = ll;
        }
        const unsigned off = (unsigned)((r & 7) * 512 + (((4 * d + g) ^ (4 * (r & 7))) & 63) * 8);
        *(f16x4*)(sth + off) = hi4;
        *(f16x4*)(stl + off) = lo4;
      }
    }
    // The staging area is private to the wave and the LDS operations of a wave execute in order: no s_barrier.  The fence is
    // for the compiler: without it the lanes' exchange through LDS is a race in its memory model (it moved the second pass's
    // reads under the first pass's mask, "this lane has not written since").
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int row = 2 * k + (lane >> 5), c16 = lane & 31;
      const unsigned roff = (unsigned)(row * 512 + ((c16 ^ (2 * row)) & 31) * 16);
      *(f16x8*)(a.oh + (row0 + 8 * half + row) * T2I_C + c16 * 8) = *(const f16x8*)(sth + roff);      // (read with the element
      *(f16x8*)(a.ol + (row0 + 8 * half + row) * T2I_C + c16 * 8) = *(const f16x8*)(stl + roff);      // type it was written with)
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

// The three launches around the bias GEMM (which the caller issues: it owns the GEMM entry and pe's registered split).
int hgl_launch_t2i_fold_q(const float* q1, const float* Wk, float scale, float* Qk, int P, hipStream_t st) {
  hipLaunchKernelGGL(t2i_fold_q_kernel, dim3((unsigned)P), dim3(256), 0, st, q1, Wk, scale, Qk);
  return hgl_check_launch("t2i_fold_q");
}

int hgl_t2i_key_ranges(int P, int HW) { return (P <= 128 && HW % (8 * T2I_CHUNK) == 0) ? 8 : 1; }
size_t hgl_t2i_part_bytes(int P, int HW) { return (size_t)P * hgl_t2i_key_ranges(P, HW) * T2I_PART * sizeof(float); }

int hgl_launch_t2i_raw_attn(const void* Qh, const void* Ql, const float* bias, const void* Kh, const void* Kl, int P, int HW,
                            float* out, int ns, hipStream_t st) {
  HGL_REQUIRE(HW % T2I_CHUNK == 0 && HW >= T2I_CHUNK && P > 0, "t2i_raw_attn: %d image tokens unsupported", HW);
  HGL_REQUIRE(ns == 1 || (ns == 8 && HW % (8 * T2I_CHUNK) == 0), "t2i_raw_attn: %d key ranges over %d image tokens", ns, HW);
  HGL_REQUIRE((((uintptr_t)Qh | (uintptr_t)Ql | (uintptr_t)bias | (uintptr_t)Kh | (uintptr_t)Kl | (uintptr_t)out) & 15) == 0,
              "t2i_raw_attn: operands must be 16-byte aligned");
  T2IArgs a;
  a.Qh = (const _Float16*)Qh; a.Ql = (const _Float16*)Ql; a.bias = bias;
  a.Kh = (const _Float16*)Kh; a.Kl = (const _Float16*)Kl; a.out = out; a.HW = HW; a.ns = ns;
  HGL_RESERVE_LDS((t2i_raw_attn_kernel), T2I_LDS, "t2i_raw_attn");
  HglProfScope prof(HGL_PROF_ATTN, 4.0 * P * T2I_ROWS * (double)HW * T2I_C, 0.0, st);
  hipLaunchKernelGGL(t2i_raw_attn_kernel, dim3((unsigned)P, (unsigned)ns), dim3(256), T2I_LDS, st, a);
  return hgl_check_launch("t2i_raw_attn");
}

int hgl_launch_t2i_unfold_v(const float* A, int ns, const float* Wv, const float* bv, float* att, int P, hipStream_t st) {
  HGL_REQUIRE(ns >= 1 && ns <= 8, "t2i_unfold_v: %d key ranges", ns);
  hipLaunchKernelGGL(t2i_unfold_v_kernel, dim3((unsigned)P), dim3(256), 0, st, A, ns, Wv, bv, att);
  return hgl_check_launch("t2i_unfold_v");
}

int hgl_launch_i2t_prep(const float* k1, const float* v1, const float* Wq, const float* bq, const float* Wo, float scale, void* Kh,
                        void* Kl, float* cb, void* Uh, void* Ul, int P, hipStream_t st) {
  hipLaunchKernelGGL(i2t_prep_kernel, dim3((unsigned)P), dim3(256), 0, st, k1, v1, Wq, bq, Wo, scale, (_Float16*)Kh, (_Float16*)Kl, cb,
                     (_Float16*)Uh, (_Float16*)Ul);
  return hgl_check_launch("i2t_prep");
}

HGL_DEFINE_SPLIT_OVERFLOW_READER(hgl_split_overflow_decoder)

int hgl_launch_dec_i2t_fold(const void* Xh, const void* Xl, const void* Kh, const void* Kl, const float* pek, const float* cb,
                            const void* Uh, const void* Ul, const float* bo, const float* ln_w, const float* ln_b, float eps, int P,
                            int HW, void* out_hi, void* out_lo, hipStream_t st) {
  HGL_REQUIRE(HW % I2TF_ROWS == 0 && P > 0 && P <= 65535, "dec_i2t_fold: %d image tokens / %d prompts unsupported", HW, P);
  I2TFArgs a;
  a.Xh = (const _Float16*)Xh; a.Xl = (const _Float16*)Xl; a.Kh = (const _Float16*)Kh; a.Kl = (const _Float16*)Kl;
  a.pek = pek; a.cb = cb; a.Uh = (const _Float16*)Uh; a.Ul = (const _Float16*)Ul;
  a.bo = bo; a.ln_w = ln_w; a.ln_b = ln_b; a.eps = eps;
  a.oh = (_Float16*)out_hi; a.ol = (_Float16*)out_lo; a.HW = HW;
  HGL_RESERVE_LDS((dec_i2t_fold_kernel), I2TF_LDS, "dec_i2t_fold");
  HglProfScope prof(HGL_PROF_OTHER, 2.0 * P * (double)HW * 256.0 * 56 * 2, 0.0, st);
  hipLaunchKernelGGL(dec_i2t_fold_kernel, dim3((unsigned)(HW / I2TF_ROWS), (unsigned)P), dim3(64 * I2TF_WAVES), I2TF_LDS, st, a);
  return hgl_check_launch("dec_i2t_fold");
}
