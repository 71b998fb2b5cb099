// Fused fp32 attention on the CDNA4 matrix cores (flash-style, online softmax).
//
// Replaces nn.MultiheadAttention inside ResidualAttentionBlock (clip/model.py:209,
// 220-229) including the per-call CLS-row mask built by make_attn_mask
// (model/backbone.py:108-115, never materialised here: [N,196] keep bytes instead of
// a [N*12,197,197] tensor), the causal text mask (clip/model.py:396-402), SAM's
// Attention with decomposed relative position bias (image_encoder.py:224-240,
// 325-361) and the decoder's Attention (transformer.py:185-240).
//
// Structure (one workgroup = 4 waves = 128 queries of one (batch, head)):
//   * "swapped" QK^T: S^T[key][query] = K Q^T with v_mfma_f32_32x32x2_f32, so each
//     lane owns ONE query column: row-max / row-sum are in-register reductions plus a
//     single cross-half exchange (lanes l and l+32 hold the other 16 keys of the tile).
//   * the P^T accumulator registers are directly the B operand of the PV product
//     O^T[d][query] += V^T[d][key] P^T[key][query]: no LDS round trip for P.
//   * K and V tiles of 64 keys are staged in LDS (K rows padded by one float4 so the
//     ds_read_b128 fragment reads are conflict free; V read as ds_read_b32 rows).
//   * everything fp32: exact products, fp32 accumulation -> matches the fp32 reference.
#include "hgl_common.h"
#include <stdlib.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int KV_CHUNK = 64;
constexpr int NEG_BIG_BITS = 0xff800000;  // -inf

struct AttnArgs {
  const float *q, *k, *v;
  float* out;
  int B, H, Sq, Sk;
  int ldq, ldk, ldv, ldo;
  long long sqb, skb, svb, sob;
  float scale;
  int mask_kind;
  const uint8_t* keep;
  int keep_b0, keep_n;
  const float *rel_h, *rel_w;
  int kh, kw;
  const float *tab_h, *tab_w;  // RELW kernels: the rel-pos TABLES [2*RELW-1, HD]; rel_h / rel_w are then computed in the kernel
  // the same tables split into fp16 hi / lo ONCE per model (hgl_register_split_weight with scale 2^0: the values hgl_split_hi_lo
  // gives in the kernel, bit for bit); nullptr: the kernel splits its fragments itself, per wave and item
  const _Float16 *tabh_hi = nullptr, *tabh_lo = nullptr, *tabw_hi = nullptr, *tabw_lo = nullptr;
  _Float16 *out_hi, *out_lo;   // f16x3 kernels: when out == nullptr the result is written as the fp16 hi+lo pair
};

template <int HD>
__global__ __launch_bounds__(256, 2) void attn_f32_kernel(AttnArgs a) {
  constexpr int HDP = HD + 4;              // padded K row (floats)
  constexpr int DT = (HD + 31) / 32;       // 32-wide d tiles of the output
  constexpr int VLD = DT * 32;             // V row in LDS (zero padded to a multiple of 32)
  constexpr int QC = HD / 8;               // 8-wide d chunks (HD % 8 == 0)
  __shared__ __attribute__((aligned(16))) float Ks[KV_CHUNK * HDP];
  __shared__ __attribute__((aligned(16))) float Vs[KV_CHUNK * VLD];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y;
  const int b = bh / a.H, hh = bh - b * a.H;
  const int q0 = (blockIdx.x * 4 + wave) * 32;
  const int qi = q0 + r;  // this lane's query
  const bool qvalid = qi < a.Sq;

  const float* qp = a.q + b * a.sqb + (long long)(qvalid ? qi : 0) * a.ldq + hh * HD;
  const float* kp = a.k + b * a.skb + hh * HD;
  const float* vp = a.v + b * a.svb + hh * HD;

  // Q fragment: lane (r,h) holds Q[q][8c+4h+j] (pre-scaled)
  f32x4 qf[QC];
#pragma unroll
  for (int c = 0; c < QC; ++c) {
    f32x4 v = qvalid ? *(const f32x4*)(qp + 8 * c + 4 * h) : f32x4{0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= a.scale;
    qf[c] = v;
  }

  f32x16 o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  const float NEG_INF = __int_as_float(NEG_BIG_BITS);
  float m_run = NEG_INF, l_run = 0.f;

  const uint8_t* keep_row = nullptr;
  if (a.mask_kind == HGL_MASK_CLS_KEEP && b >= a.keep_b0)
    keep_row = a.keep + (long long)((b - a.keep_b0) % a.keep_n) * (a.Sk - 1);
  const float* relh = a.rel_h ? a.rel_h + ((long long)bh * a.Sq + (qvalid ? qi : 0)) * a.kh : nullptr;
  const float* relw = a.rel_w ? a.rel_w + ((long long)bh * a.Sq + (qvalid ? qi : 0)) * a.kw : nullptr;

  // causal: keys beyond the last query of this workgroup are never needed
  int sk_eff = a.Sk;
  if (a.mask_kind == HGL_MASK_CAUSAL) sk_eff = min(a.Sk, (int)(blockIdx.x * 4 + 4) * 32);

  // K/V staging is software pipelined: the next 64-key chunk is fetched into registers while the
  // current one is consumed from LDS (rows beyond Sk read a clamped valid row: their scores are
  // masked to -inf, so P = 0 multiplies finite values).
  constexpr int F4_PER_ROW = HD / 4;
  constexpr int NLD = KV_CHUNK * F4_PER_ROW / 256;   // float4 per thread per matrix (exact for hd 16/32/64/80)
  static_assert(KV_CHUNK * F4_PER_ROW % 256 == 0, "chunk must divide evenly over the workgroup");
  f32x4 pk[NLD], pv[NLD];
  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = t + 256 * i;
      const int row = idx / F4_PER_ROW, c4 = idx - row * F4_PER_ROW;
      const int kg = min(kc + row, a.Sk - 1);
      pk[i] = *(const f32x4*)(kp + (long long)kg * a.ldk + c4 * 4);
      pv[i] = *(const f32x4*)(vp + (long long)kg * a.ldv + c4 * 4);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = t + 256 * i;
      const int row = idx / F4_PER_ROW, c4 = idx - row * F4_PER_ROW;
      *(f32x4*)(Ks + row * HDP + c4 * 4) = pk[i];
      *(f32x4*)(Vs + row * VLD + c4 * 4) = pv[i];
    }
  };
  if (VLD > HD) {  // zero the d padding of V once (the staging stores never touch it)
    constexpr int PADW = (VLD - HD) > 0 ? (VLD - HD) : 1;
    for (int i = t; i < KV_CHUNK * PADW; i += 256) {
      const int row = i / PADW, c = i - row * PADW;
      Vs[row * VLD + HD + c] = 0.f;
    }
  }
  load_chunk(0);
  store_chunk();
  __syncthreads();

  for (int kc = 0; kc < sk_eff; kc += KV_CHUNK) {
    const bool has_next = kc + KV_CHUNK < sk_eff;
    if (has_next) load_chunk(kc + KV_CHUNK);

#pragma unroll
    for (int kt = 0; kt < KV_CHUNK / 32; ++kt) {
      const int kbase = kc + kt * 32;
      if (kbase >= sk_eff) break;  // uniform
      // ---- S^T tile: [32 keys x 32 queries] ----
      f32x16 s;
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = 0.f;
      const float* krow = Ks + (kt * 32 + r) * HDP + 4 * h;
#pragma unroll
      for (int c = 0; c < QC; ++c) {
        const f32x4 kf = *(const f32x4*)(krow + 8 * c);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[c][j], s, 0, 0, 0);
      }
      // s[e] = S^T[key = kbase + (e&3) + 8*(e>>2) + 4*h][query qi]
      if (relh) {
        if ((a.kw & 31) == 0) {
          // a 32-key tile lies inside one key row: one rel_h value, rel_w as four aligned float4
          const float rh = relh[kbase / a.kw];
          const float* rw = relw + (kbase % a.kw) + 4 * h;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 w4 = *(const f32x4*)(rw + 8 * g4);
#pragma unroll
            for (int i = 0; i < 4; ++i) s[4 * g4 + i] += rh + w4[i];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int kg = kbase + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (kg < a.Sk) s[e] += relh[kg / a.kw] + relw[kg % a.kw];
          }
        }
      }
      float mx = NEG_INF;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int kg = kbase + (e & 3) + 8 * (e >> 2) + 4 * h;
        float sv = s[e];
        bool masked = kg >= a.Sk;
        if (a.mask_kind == HGL_MASK_CAUSAL) masked |= kg > qi;
        if (keep_row && qi == 0 && kg >= 1 && kg < a.Sk) masked |= keep_row[kg - 1] == 0;
        sv = masked ? NEG_INF : sv;
        s[e] = sv;
        mx = fmaxf(mx, sv);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx);
      // all keys so far masked -> keep everything at zero
      const float m_use = (m_new == NEG_INF) ? 0.f : m_new;
      const float alpha = expf(m_run - m_use);  // m_run=-inf -> 0
      float rs = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = expf(s[e] - m_use);
        s[e] = p;
        rs += p;
      }
      l_run = l_run * alpha + rs;
      m_run = m_new;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
      // ---- O^T += V^T P^T : A = V^T (lane (d=r,h): V[key(e,h)][d]), B = P^T = s[e] ----
      const float* vbase = Vs + (kt * 32 + 4 * h) * VLD + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float* vr = vbase + ((e & 3) + 8 * (e >> 2)) * VLD;
#pragma unroll
        for (int d = 0; d < DT; ++d)
          o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[d * 32], s[e], o[d], 0, 0, 0);
      }
    }
    __syncthreads();              // chunk fully consumed by every wave
    if (has_next) {
      store_chunk();
      __syncthreads();
    }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (qvalid) {
    float* op = a.out + b * a.sob + (long long)qi * a.ldo + hh * HD;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = d * 32 + 8 * g + 4 * h;  // o[d][4g..4g+3] = O^T[dd..dd+3][qi]
        if (dd < HD) {
          f32x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = o[d][4 * g + e] * inv;
          *(f32x4*)(op + dd) = w;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Same algorithm on the fp16 matrix cores with fp32-class accuracy (HGL_PREC_F16X3): Q, K, V and the
// probabilities P are split into fp16 hi+lo halves and every product is evaluated as
// hi*hi + hi*lo + lo*hi with v_mfma_f32_32x32x16_f16 into fp32 accumulators (see gemm_f16x3.hip
// for the error analysis).  K is staged as rows [hi | lo]; V is staged TRANSPOSED ([d][key], hi
// and lo planes) because the P^T accumulator registers 8s..8s+7 of a lane form the B operand of
// k-step s with the key order 16s + 8(j>>2) + 4h + (j&3): the matching A operand is then two
// 8-byte reads of consecutive keys from a V^T row.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split4(const f32x4 v, h16x4& hi, h16x4& lo, float& amax) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    _Float16 a, c;
    hgl_split_hi_lo(v[e], a, c, amax);
    hi[e] = a;
    lo[e] = c;
  }
}

// ds_read_b64_tr_b16: see attn_x3_kernel (EXEC must be all ones at the call)
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x4v __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ h16x4 lds_tr4(const _Float16* p) {
  typedef __attribute__((address_space(3))) fp16x4v lds_v;
  const fp16x4v v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_v*)p);
  h16x4 r;
  __builtin_memcpy(&r, &v, 8);
  return r;
}

// RELW > 0: decomposed rel-pos bias of a RELW x RELW window (Sk == RELW*RELW, 2*RELW <= 32, e.g. SAM's 14 x 14)
// evaluated ON THE MATRIX CORES: bias[q][key] = rel_h[q][key / RELW] + rel_w[q][key % RELW] = R[q][:] . E[key][:]
// with R[q] = [rel_h row | rel_w row | 0] (32 wide, split hi+lo like every other operand) and E[key] the 0/1
// indicator of (key / RELW, RELW + key % RELW).  E is appended to the staged K rows, so the bias costs two more
// k-steps (4 MFMAs) per key tile and no index arithmetic.  (The generic path divides and issues two dependent
// global loads per score; on the windowed blocks that made the kernel 3x slower than its MFMA work.)
// NW waves per workgroup, 32 queries each: 4 (two workgroups per CU) or 8 (one workgroup per CU, 256 queries: a
// 14 x 14 window or a 197-token CLIP sequence is then ONE workgroup that stages the keys and values once instead of
// two workgroups that each stage all of them).
template <int HD, int RELW, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void attn_x3_kernel(AttnArgs a) {
  constexpr int NT = NW * 64;
  constexpr int KS = HD / 16;                 // k-steps of the QK^T contraction
  constexpr int EW = RELW > 0 ? 32 : 0;       // indicator columns appended to a K row
  constexpr int KROW = 2 * HD + EW + 8;       // halfs per staged K row: hi | lo | E | pad (odd multiple of 16 B)
  static_assert(2 * RELW <= 32, "window side too large for the MFMA bias");
  constexpr int DT = (HD + 31) / 32;          // 32-wide d tiles of the output
  // V stays ROW-major in LDS (8-byte stores, like K) and the P^T V product reads it with the transposing
  // ds_read_b64_tr_b16: a 16-lane group fetches a 4-key x 16-d block and each lane receives its d column of the 4
  // keys -- exactly the 4 consecutive keys an A-operand half wants.  Row pitch 192 B (64 B for narrow heads): the
  // four rows of a block then fall on four different 64-byte bank groups.
  constexpr int VP = HD <= 32 ? 32 : 96;      // halfs per staged V row (>= DT*32)
  constexpr int F4 = HD / 4;
  constexpr int NLK = (KV_CHUNK * F4 + NT - 1) / NT;    // K float4 per thread per chunk (the last one guarded)
  constexpr int NLV = NLK;                    // V float4 per thread per chunk (same row-major walk as K)
  static_assert(HD % 16 == 0, "unsupported head dim");
  constexpr bool EXACT = KV_CHUNK * F4 % NT == 0;   // every thread has NLK elements
  __shared__ __attribute__((aligned(16))) _Float16 Ks[KV_CHUNK * KROW];
  static_assert(VP >= DT * 32, "V row pitch");
  __shared__ __attribute__((aligned(16))) _Float16 Vh[KV_CHUNK * VP];
  __shared__ __attribute__((aligned(16))) _Float16 Vl[KV_CHUNK * VP];
  // per wave: T[query][table index] of the two axes when the decomposed rel-pos terms are computed here
  constexpr int RPP = 33;
  __shared__ float RPatch[(RELW > 0 && NW == 8) ? NW * 2 * 32 * RPP : 1];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware tile map: workgroups go to the eight XCDs round-robin in linear order (x fastest), so the query blocks of one
  // (batch, head) -- which all stream the SAME keys and values -- landed on all eight L2s and each L2 fetched every head's K / V
  // (global blocks of a group of 16 images: 14 GB fetched for ~1.5 GB of q / k / v / rel-pos, profiles/r04b_sq_counters_attn_x3_global).
  // Remapped so that an XCD works through whole heads: consecutive slots of one XCD are the query blocks of one head.
  int bx = blockIdx.x, bh = blockIdx.y;
  if (gridDim.x > 1 && (gridDim.y & 7) == 0) {
    const unsigned G = gridDim.x, L = blockIdx.y * G + blockIdx.x;
    const unsigned c = L & 7, j = L >> 3;
    bh = (int)((j / G) * 8 + c);
    bx = (int)(j % G);
  }
  const int b = bh / a.H, hh = bh - b * a.H;
  const int q0 = (bx * NW + wave) * 32;
  const int qi = q0 + r;
  const bool qvalid = qi < a.Sq;
  // a wave whose 32 queries all lie beyond the sequence (e.g. the 4th wave of the second 128-query block at S = 197)
  // only helps staging K/V: it skips its MFMA / softmax work and leaves its SIMD to the co-resident workgroup
  const bool wave_active = q0 < a.Sq;
  const float* qp = a.q + b * a.sqb + (long long)(qvalid ? qi : 0) * a.ldq + hh * HD;
  const float* kp = a.k + b * a.skb + hh * HD;
  const float* vp = a.v + b * a.svb + hh * HD;

  // chunk 0 of K / V is requested FIRST: its round trip overlaps the Q loads, the Q split and the rel-pos table products
  f32x4 pk[NLK], pv[NLV];
#pragma unroll
  for (int i = 0; i < NLK; ++i) {
    const int idx = t + NT * i;
    if (!EXACT && idx >= KV_CHUNK * F4) break;
    const int row = idx / F4, c4 = idx - row * F4;
    pk[i] = *(const f32x4*)(kp + (long long)min(row, a.Sk - 1) * a.ldk + c4 * 4);
    pv[i] = *(const f32x4*)(vp + (long long)min(row, a.Sk - 1) * a.ldv + c4 * 4);
  }
  // Q fragments (pre-scaled in fp32, then split -- through hgl_split_hi_lo, see its comment: this product is where
  // the inconsistent-rounding problem was found): lane (r,h) element j of step s = Q[q][16s + 8h + j].
  // The scores are moved to log2 units AFTER the QK^T product (one multiply per score) so that the softmax is one
  // v_exp_f32 per element: the precise expf made this kernel VALU-bound (~10 instructions per exponential
  // against 33 MFMAs per key tile).
  // The soft-max scale lives in the EXPONENT's constant: exp(scale * (q . k) + bias - m) = exp2(fma(s, scale * log2 e, -m')) with
  // s = q . k + bias / scale accumulated on the matrix cores.  Q is then split ONCE, unscaled -- the same fragments feed the
  // rel-pos table products (image_encoder.py:325-361 uses the unscaled q) -- instead of twice (240 VALU instructions and 40
  // VGPRs per wave and item at head dim 80); the bias terms are divided by the scale where they enter the accumulators.
  constexpr float LOG2E = 1.4426950408889634f;
  const float sl2e = a.scale * LOG2E;
  const float inv_scale = 1.0f / a.scale;
  const float rescale_thr = 5.5f * inv_scale;     // 5.5 nats ~ 2^8, in units of the unscaled scores
  float amax = 0.f;   // of the Q / K / V values this thread splits (fp16 range guard, hgl_common.h)
  h16x8 qh[KS], ql[KS];
  // ALL of the query's raw fragments are requested before the first one is split: hgl_split_hi_lo's opaque asm keeps the
  // compiler from moving a load across it, so `load; split; load; split ...` compiled to one global round trip PER
  // 16-byte piece (10 here, 10 more for the unscaled copy, 20 for the rel-pos tables: 40 serial memory latencies at the
  // head of every 196-token item -- 56 % of the windowed kernel's wave cycles were spent waiting, tools/isa_serial_loads.py)
  f32x4 qraw[KS][2];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int half = 0; half < 2; ++half)
      qraw[s][half] = *(const f32x4*)(qp + 16 * s + 8 * h + 4 * half);   // qp is clamped to a valid row: unconditional, no branch
  __builtin_amdgcn_sched_barrier(0);
  if (!qvalid) {
#pragma unroll
    for (int s = 0; s < KS; ++s) qraw[s][0] = qraw[s][1] = f32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int s = 0; s < KS; ++s) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const f32x4 v = qraw[s][half];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        _Float16 hi, lo;
        hgl_split_hi_lo(v[e], hi, lo, amax);
        qh[s][4 * half + e] = hi;
        ql[s][4 * half + e] = lo;
      }
    }
  }

  f32x16 o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  const float NEG_INF = __int_as_float(NEG_BIG_BITS);
  float m_run = NEG_INF, l_run = 0.f;

  const uint8_t* keep_row = nullptr;
  if (a.mask_kind == HGL_MASK_CLS_KEEP && b >= a.keep_b0)
    keep_row = a.keep + (long long)((b - a.keep_b0) % a.keep_n) * (a.Sk - 1);
  const float* relh = a.rel_h ? a.rel_h + ((long long)bh * a.Sq + (qvalid ? qi : 0)) * a.kh : nullptr;
  const float* relw = a.rel_w ? a.rel_w + ((long long)bh * a.Sq + (qvalid ? qi : 0)) * a.kw : nullptr;
  int sk_eff = a.Sk;
  if (a.mask_kind == HGL_MASK_CAUSAL) sk_eff = min(a.Sk, (int)(bx * NW + NW) * 32);
  // R fragments of the MFMA bias: lane (r,h) element j of step c = R[q][16c + 8h + j]
  h16x8 rbh[2], rbl[2];
  if constexpr (RELW > 0) {
    bool from_tables = false;
    if constexpr (NW == 8) from_tables = a.tab_h != nullptr;
    if constexpr (NW == 8) {
      if (from_tables && wave_active) {
        // rel_h[q][k] = q . Rh[qy - k + RELW-1] (image_encoder.py:325-361, UNSCALED q): T^T = R . Q^T on the matrix cores
        // with the same split-fp16 scheme and summation order as relpos_mfma_kernel, per wave, through an LDS patch
        // the unscaled Q fragments are qh / ql themselves (see the exponent's constant above)
        float* P0 = RPatch + wave * 2 * 32 * RPP;
#pragma unroll
        for (int axis = 0; axis < 2; ++axis) {
          const float* Rt = axis ? a.tab_w : a.tab_h;
          f32x16 acc;
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[e] = 0.f;
          if (a.tabh_hi != nullptr) {
            // tables split once per model: the fragments are 16-byte loads of fp16 pairs, no conversion work here
            const _Float16* Th = axis ? a.tabw_hi : a.tabh_hi;
            const _Float16* Tl = axis ? a.tabw_lo : a.tabh_lo;
            const long long to = (long long)min(r, 2 * RELW - 2) * HD + 8 * h;
            h16x8 thr[KS], tlr[KS];
#pragma unroll
            for (int sx = 0; sx < KS; ++sx) {
              thr[sx] = *(const h16x8*)(Th + to + 16 * sx);
              tlr[sx] = *(const h16x8*)(Tl + to + 16 * sx);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (r >= 2 * RELW - 1) {
#pragma unroll
              for (int sx = 0; sx < KS; ++sx)
#pragma unroll
                for (int e = 0; e < 8; ++e) { thr[sx][e] = (_Float16)0.f; tlr[sx][e] = (_Float16)0.f; }
            }
#pragma unroll
            for (int sx = 0; sx < KS; ++sx) {
              acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(tlr[sx], qh[sx], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(thr[sx], ql[sx], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(thr[sx], qh[sx], acc, 0, 0, 0);
            }
          } else {
            f32x4 traw[KS][2];       // the axis' table fragments in one batch of loads (see the Q fragments)
#pragma unroll
            for (int sx = 0; sx < KS; ++sx)
#pragma unroll
              for (int half = 0; half < 2; ++half) {
                traw[sx][half] = *(const f32x4*)(Rt + (long long)min(r, 2 * RELW - 2) * HD + 16 * sx + 8 * h + 4 * half);
              }
            __builtin_amdgcn_sched_barrier(0);
            if (r >= 2 * RELW - 1) {
#pragma unroll
              for (int sx = 0; sx < KS; ++sx) traw[sx][0] = traw[sx][1] = f32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int sx = 0; sx < KS; ++sx) {
              h16x8 th, tl;
#pragma unroll
              for (int half = 0; half < 2; ++half) {
                const f32x4 v = traw[sx][half];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  _Float16 hi, lo;
                  hgl_split_hi_lo(v[e], hi, lo);
                  th[4 * half + e] = hi;
                  tl[4 * half + e] = lo;
                }
              }
              acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl, qh[sx], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(th, ql[sx], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(th, qh[sx], acc, 0, 0, 0);
            }
          }
          // acc[e] = T[table index (e&3) + 8*(e>>2) + 4*h][query r]
          float* Pw = P0 + axis * 32 * RPP + r * RPP;
#pragma unroll
          for (int e = 0; e < 16; ++e) Pw[(e & 3) + 8 * (e >> 2) + 4 * h] = acc[e];
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        const int qq = qvalid ? qi : 0;
        const int qy = qq / RELW, qx = qq - qy * RELW;
        const float* Ph = P0 + r * RPP + qy + RELW - 1;               // entry of k = 0; the index falls by one per k
        const float* Pv = P0 + 32 * RPP + r * RPP + qx + RELW - 1;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int idx = 16 * c + 8 * h + j;
            float x = 0.f;
            if (idx < RELW) x = Ph[-idx];
            else if (idx < 2 * RELW) x = Pv[-(idx - RELW)];
            x *= inv_scale;
            _Float16 hi, lo;
            hgl_split_hi_lo(x, hi, lo);
            rbh[c][j] = hi;
            rbl[c][j] = lo;
          }
      }
    }
    if (!from_tables) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int idx = 16 * c + 8 * h + j;
          float x = 0.f;
          if (idx < RELW) x = relh[idx];
          else if (idx < 2 * RELW) x = relw[idx - RELW];
          x *= inv_scale;
          _Float16 hi, lo;
          hgl_split_hi_lo(x, hi, lo);
          rbh[c][j] = hi;
          rbl[c][j] = lo;
        }
    }
  }

  // ---- staging (software pipelined through registers) ----
  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int i = 0; i < NLK; ++i) {
      const int idx = t + NT * i;
      if (!EXACT && idx >= KV_CHUNK * F4) break;
      const int row = idx / F4, c4 = idx - row * F4;
      pk[i] = *(const f32x4*)(kp + (long long)min(kc + row, a.Sk - 1) * a.ldk + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < NLV; ++i) {
      const int idx = t + NT * i;
      if (!EXACT && idx >= KV_CHUNK * F4) break;
      const int row = idx / F4, c4 = idx - row * F4;
      pv[i] = *(const f32x4*)(vp + (long long)min(kc + row, a.Sk - 1) * a.ldv + c4 * 4);
    }
  };
  auto store_chunk = [&](int kc) {
#pragma unroll
    for (int i = 0; i < NLK; ++i) {
      const int idx = t + NT * i;
      if (!EXACT && idx >= KV_CHUNK * F4) break;
      const int row = idx / F4, c4 = idx - row * F4;
      h16x4 hi, lo;
      split4(pk[i], hi, lo, amax);
      *(h16x4*)(Ks + row * KROW + c4 * 4) = hi;
      *(h16x4*)(Ks + row * KROW + HD + c4 * 4) = lo;
    }
    if (RELW > 0 && t < 256) {   // indicator columns: thread -> (key row t/4, 8 of the 32 columns)
      const int row = t >> 2, j0 = 8 * (t & 3), kg = kc + row;
      const int ih = kg / RELW, iw = RELW + kg - ih * RELW;
      h16x8 e;
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = (j0 + j == ih || j0 + j == iw) ? (_Float16)1.f : (_Float16)0.f;
      *(h16x8*)(Ks + row * KROW + 2 * HD + j0) = e;
    }
#pragma unroll
    for (int i = 0; i < NLV; ++i) {
      const int idx = t + NT * i;
      if (!EXACT && idx >= KV_CHUNK * F4) break;
      const int row = idx / F4, c4 = idx - row * F4;
      h16x4 hi, lo;
      split4(pv[i], hi, lo, amax);
      *(h16x4*)(Vh + row * VP + c4 * 4) = hi;
      *(h16x4*)(Vl + row * VP + c4 * 4) = lo;
    }
  };
  if (VP > HD) {  // zero the d padding of every row once (it feeds output rows that are never stored)
    constexpr int PADW = VP - HD > 0 ? VP - HD : 1;
    for (int i = t; i < KV_CHUNK * PADW; i += NT) {
      const int row = i / PADW, c = i - row * PADW;
      Vh[row * VP + HD + c] = (_Float16)0.f;
      Vl[row * VP + HD + c] = (_Float16)0.f;
    }
  }
  // transposed-read addressing: lane = 16*grp + 4*q + p supplies row q, columns 4p..4p+3 of its group's block;
  // groups 0/1 carry d columns 0-15 / 16-31 for h = 0, groups 2/3 the same for h = 1
  const int tr_off = (((lane >> 2) & 3) + 4 * h) * VP + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  store_chunk(0);      // chunk 0 was requested at the top of the kernel
  __syncthreads();
  // Decomposed rel-pos terms given as tensors (SAM's global blocks: 64 x 64 keys): the 17 values a lane adds to the scores
  // of a key tile are fetched ONE TILE AHEAD.  Requested behind the tile's QK^T products and used at once they cost a
  // full memory round trip per 32 keys (the tables of a group of images are 2 x 268 MB: L2 misses), with nothing of the
  // wave's own work beside it.
  const bool relfast = RELW == 0 && relh != nullptr && (a.kw & 31) == 0;
  float rh_next = 0.f;
  f32x4 rw_next[4];
  // uniform base + 32-bit lane offset (a 64-bit per-lane pointer kept across the tile spilled, and its reload waited for
  // the loads just issued): (batch * head, query) rows of kh / kw floats -- 268 MB per tensor at sixteen images, < 2^32 bytes
  // The lane's row is RECOMPUTED from a fresh lane id at every use: kept across the tile, the two offsets were spilled, and
  // the scratch reload's vmcnt(0) waited for the loads just issued.
  const int rel_q0 = (bx * NW + wave) * 32;
  auto rel_prefetch = [&](int kb) {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const unsigned row = (unsigned)bh * (unsigned)a.Sq + (unsigned)min(rel_q0 + (ln & 31), a.Sq - 1);
    const unsigned relh_off = row * (unsigned)a.kh;
    const unsigned relw_off = row * (unsigned)a.kw + 4u * (unsigned)(ln >> 5);
    kb = min(kb, a.Sk - 32);
    rh_next = a.rel_h[relh_off + (unsigned)(kb / a.kw)];
    const unsigned o = relw_off + (unsigned)(kb % a.kw);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) rw_next[g4] = *(const f32x4*)(a.rel_w + o + 8 * g4);
  };
  if (relfast && wave_active) rel_prefetch(0);

  for (int kc = 0; kc < sk_eff; kc += KV_CHUNK) {
    const bool has_next = kc + KV_CHUNK < sk_eff;
    if (has_next) load_chunk(kc + KV_CHUNK);
#pragma unroll
    for (int kt = 0; kt < KV_CHUNK / 32; ++kt) {
      const int kbase = kc + kt * 32;
      if (kbase >= sk_eff || !wave_active) break;  // uniform
      f32x16 s;
      if (relfast) {     // the score accumulators START at the rel-pos terms (fetched during the previous tile's P V products)
        const float rhs = rh_next * inv_scale;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = fmaf(rw_next[e >> 2][e & 3], inv_scale, rhs);
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
      }
      const _Float16* krow = Ks + (kt * 32 + r) * KROW + 8 * h;
#pragma unroll
      for (int c = 0; c < KS; ++c) {
        const h16x8 kh8 = *(const h16x8*)(krow + 16 * c);
        const h16x8 kl8 = *(const h16x8*)(krow + HD + 16 * c);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl8, qh[c], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, ql[c], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, qh[c], s, 0, 0, 0);
      }
      if constexpr (RELW > 0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const h16x8 e8 = *(const h16x8*)(krow + 2 * HD + 16 * c);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(e8, rbl[c], s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(e8, rbh[c], s, 0, 0, 0);
        }
      }
      // s[e] = S^T[key = kbase + (e&3) + 8*(e>>2) + 4*h][query qi], in natural-log units; log2(e) is folded into the
      // exponent's fma below
      if constexpr (RELW > 0) {
        // bias already accumulated by the MFMAs above
      } else if (relh) {
        if ((a.kw & 31) == 0) {
          // already in the accumulators (see the tile's start)
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int kg = kbase + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (kg < a.Sk) s[e] += (relh[kg / a.kw] + relw[kg % a.kw]) * inv_scale;
          }
        }
      }
      float mx = NEG_INF;
      // only the tile that crosses the end of the sequence, causal tiles and CLS-keep batches need masking (uniform)
      if (kbase + 32 > a.Sk || a.mask_kind == HGL_MASK_CAUSAL || keep_row) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int kg = kbase + (e & 3) + 8 * (e >> 2) + 4 * h;
          float sv = s[e];
          bool masked = kg >= a.Sk;
          if (a.mask_kind == HGL_MASK_CAUSAL) masked |= kg > qi;
          if (keep_row && qi == 0 && kg >= 1 && kg < a.Sk) masked |= keep_row[kg - 1] == 0;
          sv = masked ? NEG_INF : sv;
          s[e] = sv;
          mx = fmaxf(mx, sv);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[e]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      // Lazy rescaling: the running maximum follows the scores only when some query of the wave would otherwise see
      // probabilities above 2^8 (uniform branch).  Until then alpha == 1 and the 16*DT multiplies of O are skipped;
      // p <= 256 keeps its full relative precision in fp32 and in the fp16 hi+lo pair.
      const float m_cand = fmaxf(m_run, mx);
      float m_new = m_run;
      if (__builtin_amdgcn_ballot_w64(m_cand > m_run + rescale_thr)) {
        m_new = m_cand;
        const float m_use0 = (m_new == NEG_INF) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_use0) * sl2e);
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
        m_run = m_new;
      }
      const float mneg = -((m_new == NEG_INF) ? 0.f : m_new) * sl2e;
      float rs = 0.f;
      h16x8 ph[2], pl[2];
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const float p0 = __builtin_amdgcn_exp2f(fmaf(s[e], sl2e, mneg));
        const float p1 = __builtin_amdgcn_exp2f(fmaf(s[e + 1], sl2e, mneg));
        rs += p0;
        rs += p1;
        // hi by one packed round-toward-zero conversion (any rounding works: lo is the exact remainder, rounded to nearest)
        const h16x2 hi2 = __builtin_bit_cast(h16x2, __builtin_amdgcn_cvt_pkrtz(p0, p1));
        ph[e >> 3][e & 7] = hi2[0]; ph[e >> 3][(e & 7) + 1] = hi2[1];
        pl[e >> 3][e & 7] = (_Float16)(p0 - (float)hi2[0]);
        pl[e >> 3][(e & 7) + 1] = (_Float16)(p1 - (float)hi2[1]);
      }
      l_run += rs;
      // the scores are dead from here on: the next tile's rel-pos terms are requested now and travel under the P V products
      if (relfast) rel_prefetch(kbase + 32);
      // O^T += V^T P^T ; A operand element j of lane (d, h) = V^T[d][kt*32 + 16*s2 + 8*(j>>2) + 4*h + (j&3)]
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if (kbase + 16 * s2 >= sk_eff) break;   // uniform: the keys of this k-step are all beyond the sequence (P = 0)
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const int off = (kt * 32 + 16 * s2) * VP + d * 32 + tr_off;
          const h16x4 vh0 = lds_tr4(Vh + off), vh1 = lds_tr4(Vh + off + 8 * VP);
          const h16x4 vl0 = lds_tr4(Vl + off), vl1 = lds_tr4(Vl + off + 8 * VP);
          h16x8 vh8, vl8;
#pragma unroll
          for (int e = 0; e < 4; ++e) { vh8[e] = vh0[e]; vh8[4 + e] = vh1[e]; vl8[e] = vl0[e]; vl8[4 + e] = vl1[e]; }
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl8, ph[s2], o[d], 0, 0, 0);
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, pl[s2], o[d], 0, 0, 0);
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, ph[s2], o[d], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (has_next) {
      store_chunk(kc + KV_CHUNK);
      __syncthreads();
    }
  }

  hgl_split_commit(amax);
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (qvalid) {
    const long long oo = b * a.sob + (long long)qi * a.ldo + hh * HD;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = d * 32 + 8 * g + 4 * h;
        if (dd < HD) {
          f32x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = o[d][4 * g + e] * inv;
          if (a.out) {
            *(f32x4*)(a.out + oo + dd) = w;
          } else {   // the operand form of the following f16x3 projection
            h16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              _Float16 a2, c2;
              hgl_split_hi_lo(w[e], a2, c2);
              hi[e] = a2;
              lo[e] = c2;
            }
            *(h16x4*)(a.out_hi + oo + dd) = hi;
            *(h16x4*)(a.out_lo + oo + dd) = lo;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// PING-PONG variant for long sequences without a mask (SAM's global blocks: 4096 x 4096 keys with the rel-pos terms given
// as tensors; GEM's 785-token self-self attention).  Counters of attn_x3_kernel on the global blocks
// (profiles/r04c_sq_counters_attn_x3_global_4096x80.json): matrix pipe busy 37 % of the cycles, VALU issue 42 %, and only 22 %
// of the matrix-busy cycles had a vector instruction executing beside them -- the two waves of a SIMD ran the same phase
// at the same time (two independent 4-wave workgroups drift into lock-step at the chunk barriers), so the soft-max of
// one never filled the issue gaps of the other's MFMAs.  Here ONE 8-wave workgroup (256 queries) is two groups of four
// waves, one wave of each per SIMD, that run ONE BARRIER INTERVAL APART (the schedule of gemm_x3p_kernel):
//
//     interval   2t          2t+1            2t+2           2t+3
//     group A    M(t)        V(t)            M(t+1)         V(t+1)          M(t) = P V of tile t-1, then Q K^T of tile t
//     group B    V(t-1)      M(t)            V(t)           M(t+1)          V(t) = soft-max of tile t (+ staging duty)
//
// so that in every interval one wave of a SIMD issues matrix instructions and the other vector instructions.  K / V chunks
// of 64 keys are double-buffered in LDS; chunk c is written by group B in interval 4c-2 and by group A in interval 4c-1
// (each thread its own pieces, fetched four intervals earlier), after the last reader of the buffer's previous chunk
// (group B's P V of tile 2c-3, interval 4c-3) and before its first reader (group A's Q K^T of tile 2c, interval 4c).
// The arithmetic of a (query tile, key tile) pair is attn_x3_kernel's, instruction for instruction: identical results.
template <int HD>
__global__ __launch_bounds__(512, 1) void attn_x3pp_kernel(AttnArgs a) {
  constexpr int NT = 512;
  constexpr int KS = HD / 16;
  constexpr int KROW = 2 * HD + 8;
  constexpr int DT = (HD + 31) / 32;
  constexpr int VP = HD <= 32 ? 32 : 96;
  constexpr int F4 = HD / 4;
  constexpr int HALF = KV_CHUNK * F4;            // float4 pieces of K (and of V) per chunk
  constexpr int NP = 2 * HALF / NT;              // pieces per thread and chunk
  static_assert(2 * HALF % NT == 0 && HD % 16 == 0, "unsupported head dim");
  constexpr int KBUF = KV_CHUNK * KROW, VBUF = KV_CHUNK * VP;
  extern __shared__ __attribute__((aligned(16))) _Float16 pp_smem[];
  _Float16* const Ks = pp_smem;                  // [2][KBUF]
  _Float16* const Vh = Ks + 2 * KBUF;            // [2][VBUF]
  _Float16* const Vl = Vh + 2 * VBUF;            // [2][VBUF]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = wave >> 2;
  const int r = lane & 31, h = lane >> 5;
  int bx = blockIdx.x, bh = blockIdx.y;
  if (gridDim.x > 1 && (gridDim.y & 7) == 0) {   // an XCD works through whole heads (see attn_x3_kernel)
    const unsigned G = gridDim.x, L = blockIdx.y * G + blockIdx.x;
    const unsigned c = L & 7, j = L >> 3;
    bh = (int)((j / G) * 8 + c);
    bx = (int)(j % G);
  }
  const int b = bh / a.H, hh = bh - b * a.H;
  const int q0 = (bx * 8 + wave) * 32;
  const int qi = q0 + r;
  const bool qvalid = qi < a.Sq;
  const bool wave_active = q0 < a.Sq;
  const float* qp = a.q + b * a.sqb + (long long)(qvalid ? qi : 0) * a.ldq + hh * HD;
  const float* kp = a.k + b * a.skb + hh * HD;
  const float* vp = a.v + b * a.svb + hh * HD;
  const int ntile = (a.Sk + 31) / 32, nchunk = (a.Sk + KV_CHUNK - 1) / KV_CHUNK;

  constexpr float LOG2E = 1.4426950408889634f;
  const float sl2e = a.scale * LOG2E;
  const float inv_scale = 1.0f / a.scale;
  const float rescale_thr = 5.5f * inv_scale;
  float amax = 0.f;

  // ---- staging pieces of this thread: piece p = t + NT * i; p < HALF: K row p / F4, else V ----
  f32x4 pc[NP];
  auto load_pieces = [&](int c) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = t + NT * i;
      const int q = p < HALF ? p : p - HALF;
      const int row = q / F4, c4 = q - row * F4;
      const long long gr = min(c * KV_CHUNK + row, a.Sk - 1);
      pc[i] = p < HALF ? *(const f32x4*)(kp + gr * a.ldk + c4 * 4) : *(const f32x4*)(vp + gr * a.ldv + c4 * 4);
    }
  };
  auto store_pieces = [&](int c) {
    const int buf = c & 1;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = t + NT * i;
      const int q = p < HALF ? p : p - HALF;
      const int row = q / F4, c4 = q - row * F4;
      h16x4 hi, lo;
      split4(pc[i], hi, lo, amax);
      if (p < HALF) {
        *(h16x4*)(Ks + buf * KBUF + row * KROW + c4 * 4) = hi;
        *(h16x4*)(Ks + buf * KBUF + row * KROW + HD + c4 * 4) = lo;
      } else {
        *(h16x4*)(Vh + buf * VBUF + row * VP + c4 * 4) = hi;
        *(h16x4*)(Vl + buf * VBUF + row * VP + c4 * 4) = lo;
      }
    }
  };
  load_pieces(0);
  // Q fragments, unscaled (the scale lives in the exponent's constant): one batch of loads, then the splits
  h16x8 qh[KS], ql[KS];
  {
    f32x4 qraw[KS][2];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int half = 0; half < 2; ++half) qraw[s][half] = *(const f32x4*)(qp + 16 * s + 8 * h + 4 * half);
    __builtin_amdgcn_sched_barrier(0);
    if (!qvalid) {
#pragma unroll
      for (int s = 0; s < KS; ++s) qraw[s][0] = qraw[s][1] = f32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          _Float16 hi, lo;
          hgl_split_hi_lo(qraw[s][half][e], hi, lo, amax);
          qh[s][4 * half + e] = hi;
          ql[s][4 * half + e] = lo;
        }
  }
  if (VP > HD) {   // zero the d padding of both V buffers once
    constexpr int PADW = VP - HD > 0 ? VP - HD : 1;
    for (int i = t; i < 2 * KV_CHUNK * PADW; i += NT) {
      const int row = i / PADW, c = i - row * PADW;
      Vh[row * VP + HD + c] = (_Float16)0.f;     // rows 0..127 = both buffers (VBUF = 64 * VP)
      Vl[row * VP + HD + c] = (_Float16)0.f;
    }
  }
  store_pieces(0);
  if (nchunk > 1) {
    load_pieces(1);
    store_pieces(1);
  }
  if (nchunk > 2) load_pieces(2);
  __syncthreads();

  f32x16 o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  const float NEG_INF = __int_as_float(NEG_BIG_BITS);
  float m_run = NEG_INF, l_run = 0.f;
  const int tr_off = (((lane >> 2) & 3) + 4 * h) * VP + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  // rel-pos terms of the NEXT tile (tensors [B*H, Sq, kh] / [B*H, Sq, kw], kw a multiple of 32)
  const bool rel = a.rel_h != nullptr;
  float rh_next = 0.f;
  f32x4 rw_next[4];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) rw_next[g4] = f32x4{0, 0, 0, 0};
  const unsigned rel_row = (unsigned)bh * (unsigned)a.Sq + (unsigned)min(qi, a.Sq - 1);
  auto rel_prefetch = [&](int kb) {
    kb = min(kb, a.Sk - 32);
    rh_next = a.rel_h[rel_row * (unsigned)a.kh + (unsigned)(kb / a.kw)];
    const unsigned o2 = rel_row * (unsigned)a.kw + 4u * (unsigned)h + (unsigned)(kb % a.kw);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) rw_next[g4] = *(const f32x4*)(a.rel_w + o2 + 8 * g4);
  };
  if (rel && wave_active) rel_prefetch(0);

  auto bar = [&]() {     // this wave's LDS writes (staging) and reads (fragments) retired, then the workgroup barrier
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  f32x16 s;
  h16x8 ph[2], pl[2];
  auto pv_step = [&](int tp) {      // O^T += V^T P^T for tile tp (its probabilities are in ph / pl)
    const int buf = (tp >> 1) & 1, kt = tp & 1, kbase = tp * 32;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      if (kbase + 16 * s2 >= a.Sk) break;
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const int off = buf * VBUF + (kt * 32 + 16 * s2) * VP + d * 32 + tr_off;
        const h16x4 vh0 = lds_tr4(Vh + off), vh1 = lds_tr4(Vh + off + 8 * VP);
        const h16x4 vl0 = lds_tr4(Vl + off), vl1 = lds_tr4(Vl + off + 8 * VP);
        h16x8 vh8, vl8;
#pragma unroll
        for (int e = 0; e < 4; ++e) { vh8[e] = vh0[e]; vh8[4 + e] = vh1[e]; vl8[e] = vl0[e]; vl8[4 + e] = vl1[e]; }
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl8, ph[s2], o[d], 0, 0, 0);
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, pl[s2], o[d], 0, 0, 0);
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, ph[s2], o[d], 0, 0, 0);
      }
    }
  };

  if (grp == 1) bar();               // group B runs one interval behind group A
  for (int tt = 0; tt < ntile; ++tt) {
    const int kbase = tt * 32;
    // ---------------- M step: P V of the previous tile, Q K^T of this one ----------------
    if (wave_active) {
      __builtin_amdgcn_s_setprio(1);
      if (tt > 0) pv_step(tt - 1);
      if (rel) {
        const float rhs = rh_next * inv_scale;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = fmaf(rw_next[e >> 2][e & 3], inv_scale, rhs);
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
      }
      const _Float16* krow = Ks + ((tt >> 1) & 1) * KBUF + ((tt & 1) * 32 + r) * KROW + 8 * h;
#pragma unroll
      for (int c = 0; c < KS; ++c) {
        const h16x8 kh8 = *(const h16x8*)(krow + 16 * c);
        const h16x8 kl8 = *(const h16x8*)(krow + HD + 16 * c);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl8, qh[c], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, ql[c], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, qh[c], s, 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    }
    bar();
    // ---------------- V step: staging duty, soft-max of this tile ----------------
    // staging duty FIRST (the next pieces then travel under the soft-max, a barrier and the next P V products): group B writes
    // its pieces of chunk c in its V step of tile 2c-2, group A in its V step of tile 2c-1
    {
      const int c = grp == 1 ? (tt + 2) >> 1 : (tt + 1) >> 1;
      const bool mine = grp == 1 ? (tt & 1) == 0 : (tt & 1) == 1;
      if (mine && c >= 2 && c < nchunk) {
        store_pieces(c);
        if (c + 1 < nchunk) load_pieces(c + 1);
      }
    }
    if (wave_active) {
      if (rel) rel_prefetch(kbase + 32);       // consumed at the next M step's start: a whole V step + a barrier away
      float mx = NEG_INF;
      if (kbase + 32 > a.Sk) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int kg = kbase + (e & 3) + 8 * (e >> 2) + 4 * h;
          const float sv = kg >= a.Sk ? NEG_INF : s[e];
          s[e] = sv;
          mx = fmaxf(mx, sv);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[e]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_cand = fmaxf(m_run, mx);
      float m_new = m_run;
      if (__builtin_amdgcn_ballot_w64(m_cand > m_run + rescale_thr)) {
        m_new = m_cand;
        const float m_use0 = (m_new == NEG_INF) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_use0) * sl2e);
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
        m_run = m_new;
      }
      const float mneg = -((m_new == NEG_INF) ? 0.f : m_new) * sl2e;
      float rs = 0.f;
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const float p0 = __builtin_amdgcn_exp2f(fmaf(s[e], sl2e, mneg));
        const float p1 = __builtin_amdgcn_exp2f(fmaf(s[e + 1], sl2e, mneg));
        rs += p0;
        rs += p1;
        const h16x2 hi2 = __builtin_bit_cast(h16x2, __builtin_amdgcn_cvt_pkrtz(p0, p1));
        ph[e >> 3][e & 7] = hi2[0]; ph[e >> 3][(e & 7) + 1] = hi2[1];
        pl[e >> 3][e & 7] = (_Float16)(p0 - (float)hi2[0]);
        pl[e >> 3][(e & 7) + 1] = (_Float16)(p1 - (float)hi2[1]);
      }
      l_run += rs;
    }
    bar();
  }
  if (wave_active) {
    __builtin_amdgcn_s_setprio(1);
    pv_step(ntile - 1);
    __builtin_amdgcn_s_setprio(0);
  }
  if (grp == 0) bar();               // group A's last interval: group B is still one behind

  hgl_split_commit(amax);
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (qvalid) {
    const long long oo = b * a.sob + (long long)qi * a.ldo + hh * HD;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = d * 32 + 8 * g + 4 * h;
        if (dd < HD) {
          f32x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = o[d][4 * g + e] * inv;
          if (a.out) {
            *(f32x4*)(a.out + oo + dd) = w;
          } else {
            h16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              _Float16 a2, c2;
              hgl_split_hi_lo(w[e], a2, c2);
              hi[e] = a2;
              lo[e] = c2;
            }
            *(h16x4*)(a.out_hi + oo + dd) = hi;
            *(h16x4*)(a.out_lo + oo + dd) = lo;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// "Two items per CU" variant for sequences of 129..256 tokens without rel-pos / causal mask (the CLIP sequences): ONE
// 4-wave workgroup per (batch, head) item, a wave owns TWO 32-query tiles (64 queries), so the item's keys and values are
// staged once (as in the 8-wave kernels) but a CU holds two workgroups = two items whose phases are independent: while one
// waits for its loads, splits a chunk or stores its output rows, the other multiplies.  (The 8-wave workgroups put both
// waves of a SIMD on the SAME item: they meet at every chunk barrier and wait for the same loads.)  Arithmetic per
// (query tile, key tile) is that of attn_x3_kernel, operation for operation.
template <int HD, int QT>
__global__ __launch_bounds__(256, 2) void attn_x3q_kernel(AttnArgs a) {
  constexpr int NW = 4, NT = NW * 64;
  constexpr int KS = HD / 16;
  constexpr int KROW = 2 * HD + 8;
  constexpr int DT = (HD + 31) / 32;
  constexpr int VP = HD <= 32 ? 32 : 96;
  constexpr int F4 = HD / 4;
  constexpr int NLK = (KV_CHUNK * F4 + NT - 1) / NT;
  static_assert(KV_CHUNK * F4 % NT == 0 && HD % 16 == 0, "unsupported head dim");
  __shared__ __attribute__((aligned(16))) _Float16 Ks[KV_CHUNK * KROW];
  __shared__ __attribute__((aligned(16))) _Float16 Vh[KV_CHUNK * VP];
  __shared__ __attribute__((aligned(16))) _Float16 Vl[KV_CHUNK * VP];
  __shared__ uint8_t keepL[256];   // keepL[key - 1] = the item's CLS-keep row (make_attn_mask: only query 0 is restricted)

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.x;
  const int b = bh / a.H, hh = bh - b * a.H;
  const float* kp = a.k + b * a.skb + hh * HD;
  const float* vp = a.v + b * a.svb + hh * HD;
  constexpr float LOG2E = 1.4426950408889634f;
  const float NEG_INF = __int_as_float(NEG_BIG_BITS);
  float amax = 0.f;

  const uint8_t* keep_row = nullptr;
  if (a.mask_kind == HGL_MASK_CLS_KEEP && b >= a.keep_b0)
    keep_row = a.keep + (long long)((b - a.keep_b0) % a.keep_n) * (a.Sk - 1);
  if (keep_row && t < a.Sk - 1) keepL[t] = keep_row[t];

  // ---- Q fragments of the wave's QT query tiles (pre-scaled in fp32, then split) ----
  int qi[QT];
  bool qvalid[QT], tile_active[QT];
  h16x8 qh[QT][KS], ql[QT][KS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q0 = (wave * QT + qt) * 32;
    qi[qt] = q0 + r;
    qvalid[qt] = qi[qt] < a.Sq;
    tile_active[qt] = q0 < a.Sq;
    const float* qp = a.q + b * a.sqb + (long long)(qvalid[qt] ? qi[qt] : 0) * a.ldq + hh * HD;
    f32x4 qraw[KS][2];      // one batch of loads, then the splits (see attn_x3_kernel: a split right behind its load serialises them)
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int half = 0; half < 2; ++half)
        qraw[s][half] = *(const f32x4*)(qp + 16 * s + 8 * h + 4 * half);     // qp is clamped to a valid row
    __builtin_amdgcn_sched_barrier(0);
    if (!qvalid[qt]) {
#pragma unroll
      for (int s = 0; s < KS; ++s) qraw[s][0] = qraw[s][1] = f32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const f32x4 v = qraw[s][half];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          _Float16 hi, lo;
          hgl_split_hi_lo(v[e] * a.scale, hi, lo, amax);
          qh[qt][s][4 * half + e] = hi;
          ql[qt][s][4 * half + e] = lo;
        }
      }
  }
  f32x16 o[QT][DT];
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_run[qt] = NEG_INF;
    l_run[qt] = 0.f;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[qt][d][e] = 0.f;
  }

  // ---- staging (software pipelined through registers) ----
  f32x4 pk[NLK], pv[NLK];
  // Rows of a chunk beyond the sequence's last 16-key step are neither fetched nor converted: their scores are masked and
  // their probabilities never multiplied (the LDS rows keep the finite values of the previous chunk).  The fourth chunk of
  // a 197-token sequence holds 5 keys: without this a quarter of all staging work went into clamped copies of key 196.
  const int sk16 = (a.Sk + 15) & ~15;
  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int i = 0; i < NLK; ++i) {
      const int idx = t + NT * i;
      const int row = idx / F4, c4 = idx - row * F4;
      if (kc + row < sk16) {
        pk[i] = *(const f32x4*)(kp + (long long)min(kc + row, a.Sk - 1) * a.ldk + c4 * 4);
        pv[i] = *(const f32x4*)(vp + (long long)min(kc + row, a.Sk - 1) * a.ldv + c4 * 4);
      }
    }
  };
  auto store_chunk = [&](int kc) {
#pragma unroll
    for (int i = 0; i < NLK; ++i) {
      const int idx = t + NT * i;
      const int row = idx / F4, c4 = idx - row * F4;
      if (kc + row < sk16) {
        h16x4 hi, lo;
        split4(pk[i], hi, lo, amax);
        *(h16x4*)(Ks + row * KROW + c4 * 4) = hi;
        *(h16x4*)(Ks + row * KROW + HD + c4 * 4) = lo;
        split4(pv[i], hi, lo, amax);
        *(h16x4*)(Vh + row * VP + c4 * 4) = hi;
        *(h16x4*)(Vl + row * VP + c4 * 4) = lo;
      }
    }
  };
  if (VP > HD) {
    constexpr int PADW = VP - HD > 0 ? VP - HD : 1;
    for (int i = t; i < KV_CHUNK * PADW; i += NT) {
      const int row = i / PADW, c = i - row * PADW;
      Vh[row * VP + HD + c] = (_Float16)0.f;
      Vl[row * VP + HD + c] = (_Float16)0.f;
    }
  }
  const int tr_off = (((lane >> 2) & 3) + 4 * h) * VP + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  for (int kc = 0; kc < a.Sk; kc += KV_CHUNK) {
    const bool has_next = kc + KV_CHUNK < a.Sk;
    if (has_next) load_chunk(kc + KV_CHUNK);
#pragma unroll
    for (int kt = 0; kt < KV_CHUNK / 32; ++kt) {
      const int kbase = kc + kt * 32;
      if (kbase >= a.Sk) break;  // uniform
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        if (!tile_active[qt]) continue;   // uniform per wave
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
        const _Float16* krow = Ks + (kt * 32 + r) * KROW + 8 * h;
#pragma unroll
        for (int c = 0; c < KS; ++c) {
          const h16x8 kh8 = *(const h16x8*)(krow + 16 * c);
          const h16x8 kl8 = *(const h16x8*)(krow + HD + 16 * c);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl8, qh[qt][c], s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, ql[qt][c], s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, qh[qt][c], s, 0, 0, 0);
        }
        if (kbase + 32 > a.Sk) {   // uniform: the tile that crosses the end of the sequence
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int kg = kbase + (e & 3) + 8 * (e >> 2) + 4 * h;
            s[e] = kg >= a.Sk ? NEG_INF : s[e];
          }
        }
        if (keep_row && wave == 0 && qt == 0) {   // uniform: the wave and tile that own query 0 of a CLS-keep batch
          const int kk = kbase + (lane & 31);
          const unsigned kb = kk >= 1 && kk < a.Sk ? keepL[kk - 1] : 1u;
          const unsigned bits = (unsigned)__builtin_amdgcn_ballot_w64(kb != 0) >> (4 * h);
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const bool kept = (bits >> ((e & 3) + 8 * (e >> 2))) & 1u;
            s[e] = (qi[0] == 0 && !kept) ? NEG_INF : s[e];
          }
        }
        float mx = NEG_INF;
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_cand = fmaxf(m_run[qt], mx);
        float m_new = m_run[qt];
        if (__builtin_amdgcn_ballot_w64(m_cand > m_run[qt] + 5.5f)) {   // lazy rescaling (attn_x3_kernel)
          m_new = m_cand;
          const float m_use0 = (m_new == NEG_INF) ? 0.f : m_new;
          const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_use0) * LOG2E);
          l_run[qt] *= alpha;
#pragma unroll
          for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[qt][d][e] *= alpha;
          m_run[qt] = m_new;
        }
        const float mneg = -((m_new == NEG_INF) ? 0.f : m_new) * LOG2E;
        float rs = 0.f;
        h16x8 ph[2], pl[2];
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          const float p0 = __builtin_amdgcn_exp2f(fmaf(s[e], LOG2E, mneg));
          const float p1 = __builtin_amdgcn_exp2f(fmaf(s[e + 1], LOG2E, mneg));
          rs += p0;
          rs += p1;
          const h16x2 hi2 = __builtin_bit_cast(h16x2, __builtin_amdgcn_cvt_pkrtz(p0, p1));
          ph[e >> 3][e & 7] = hi2[0]; ph[e >> 3][(e & 7) + 1] = hi2[1];
          pl[e >> 3][e & 7] = (_Float16)(p0 - (float)hi2[0]);
          pl[e >> 3][(e & 7) + 1] = (_Float16)(p1 - (float)hi2[1]);
        }
        l_run[qt] += rs;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          if (kbase + 16 * s2 >= a.Sk) break;   // uniform
#pragma unroll
          for (int d = 0; d < DT; ++d) {
            const int off = (kt * 32 + 16 * s2) * VP + d * 32 + tr_off;
            const h16x4 vh0 = lds_tr4(Vh + off), vh1 = lds_tr4(Vh + off + 8 * VP);
            const h16x4 vl0 = lds_tr4(Vl + off), vl1 = lds_tr4(Vl + off + 8 * VP);
            h16x8 vh8, vl8;
#pragma unroll
            for (int e = 0; e < 4; ++e) { vh8[e] = vh0[e]; vh8[4 + e] = vh1[e]; vl8[e] = vl0[e]; vl8[4 + e] = vl1[e]; }
            o[qt][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl8, ph[s2], o[qt][d], 0, 0, 0);
            o[qt][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, pl[s2], o[qt][d], 0, 0, 0);
            o[qt][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, ph[s2], o[qt][d], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
    if (has_next) {
      store_chunk(kc + KV_CHUNK);
      __syncthreads();
    }
  }

  hgl_split_commit(amax);
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const float l_tot = l_run[qt] + __shfl_xor(l_run[qt], 32);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qvalid[qt]) {
      const long long oo = b * a.sob + (long long)qi[qt] * a.ldo + hh * HD;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dd = d * 32 + 8 * g + 4 * h;
          if (dd < HD) {
            f32x4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = o[qt][d][4 * g + e] * inv;
            if (a.out) {
              *(f32x4*)(a.out + oo + dd) = w;
            } else {
              h16x4 hi, lo;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                _Float16 a2, c2;
                hgl_split_hi_lo(w[e], a2, c2);
                hi[e] = a2;
                lo[e] = c2;
              }
              *(h16x4*)(a.out_hi + oo + dd) = hi;
              *(h16x4*)(a.out_lo + oo + dd) = lo;
            }
          }
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Attention with a handful of keys (the mask decoder's image -> token direction, transformer.py:139-150: 4096
// image queries per prompt against the 7 prompt tokens, 8 heads of 16).  An MFMA tile would be 78 % padding;
// here one thread owns one (query, head): 16 q values in registers, the prompt's K and V (Sk x H*16 floats each)
// in LDS, scores / softmax / PV on the VALU.  HBM-bound: reads q once, writes the output once -- as fp32 or
// directly as the fp16 hi+lo pair the following f16x3 out-projection consumes.
struct SmallKArgs {
  const float *q, *k, *v;
  float* out;
  _Float16 *out_hi, *out_lo;
  int B, H, Sq, Sk, ldq, ldk, ldv, ldo;
  long long sqb, skb, svb, sob;
  float scale;
};

constexpr int SMALLK_MAX = 8;

__global__ __launch_bounds__(256) void attn_smallk_kernel(SmallKArgs a) {
  extern __shared__ __attribute__((aligned(16))) float kv_s[];   // [2][Sk][H*16]
  const int HD = 16, W = a.H * HD;
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < a.Sk * (W / 4); i += 256) {
    const int j = i / (W / 4), c = i - j * (W / 4);
    ((f32x4*)kv_s)[i] = *(const f32x4*)(a.k + b * a.skb + (long long)j * a.ldk + 4 * c);
    ((f32x4*)kv_s)[a.Sk * (W / 4) + i] = *(const f32x4*)(a.v + b * a.svb + (long long)j * a.ldv + 4 * c);
  }
  __syncthreads();
  const int qpb = 256 / a.H;                       // queries per workgroup
  const int hh = threadIdx.x % a.H, ql = threadIdx.x / a.H;
  const int qi = blockIdx.x * qpb + ql;
  if (ql >= qpb || qi >= a.Sq) return;
  const float* qp = a.q + b * a.sqb + (long long)qi * a.ldq + hh * HD;
  f32x4 qv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) qv[c] = *(const f32x4*)(qp + 4 * c);
  float sc[SMALLK_MAX];
  float mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < SMALLK_MAX; ++j) {
    sc[j] = -3.0e38f;
    if (j < a.Sk) {
      const float* kr = kv_s + j * W + hh * HD;
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 kk = *(const f32x4*)(kr + 4 * c);
        d += qv[c][0] * kk[0]; d += qv[c][1] * kk[1]; d += qv[c][2] * kk[2]; d += qv[c][3] * kk[3];
      }
      sc[j] = d * a.scale;
      mx = fmaxf(mx, sc[j]);
    }
  }
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < SMALLK_MAX; ++j) {
    sc[j] = j < a.Sk ? expf(sc[j] - mx) : 0.f;
    l += sc[j];
  }
  const float inv = 1.0f / l;
  f32x4 o[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < SMALLK_MAX; ++j) {
    if (j < a.Sk) {
      const float* vr = kv_s + (a.Sk + j) * W + hh * HD;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 vv = *(const f32x4*)(vr + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[c][e] += sc[j] * vv[e];
      }
    }
  }
  const long long oo = b * a.sob + (long long)qi * a.ldo + hh * HD;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = o[c][e] * inv;
    if (a.out) {
      *(f32x4*)(a.out + oo + 4 * c) = r;
    } else {
      h16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        _Float16 a2, c2;
        hgl_split_hi_lo(r[e], a2, c2);
        hi[e] = a2;
        lo[e] = c2;
      }
      *(h16x4*)(a.out_hi + oo + 4 * c) = hi;
      *(h16x4*)(a.out_lo + oo + 4 * c) = lo;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Attention with a handful of QUERIES over many keys (the mask decoder's token -> image direction,
// transformer.py:126-131: 7 prompt tokens against 4096 image tokens, 8 heads of 16).  One workgroup per (batch, head);
// a thread walks keys t, t+256, ... with the running (max, sum, 16-wide output) of all queries in registers, the
// query vectors broadcast from LDS; the per-thread states are merged with the usual (m, l, o) rule by wave
// shuffles, then across the four waves through LDS.  HBM-bound (K and V read once); the MFMA kernel spent 78 % of
// its tile on padding queries.  Workgroups are numbered so that the heads of one batch element share an XCD (they
// read complementary 64-byte halves of the same 128-byte lines).
struct FewQArgs {
  const float *q, *k, *v;
  float* out;
  int B, H, Sq, Sk, ldq, ldk, ldv, ldo;
  long long sqb, skb, svb, sob;
  float scale;
};

constexpr int FEWQ_MAX = 8;

__global__ __launch_bounds__(256) void attn_fewq_kernel(FewQArgs a) {
  constexpr int HD = 16;
  __shared__ __attribute__((aligned(16))) float q_s[FEWQ_MAX * HD];
  __shared__ float part[4][FEWQ_MAX][HD + 2];
  // (b, h) of this workgroup: consecutive workgroups round-robin over the 8 XCDs
  int bh;
  {
    const int bid = blockIdx.x, n = gridDim.x;
    const int per = a.H * 8;                     // workgroups of 8 batch elements
    const int full = (n / per) * per;
    if (bid < full) {
      const int grp = bid / per, r = bid - grp * per;
      const int xcd = r & 7, j = r >> 3;          // j-th workgroup of this XCD inside the group
      bh = (grp * 8 + xcd) * a.H + j;
    } else {
      bh = bid;
    }
  }
  const int b = bh / a.H, hh = bh - b * a.H;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t < a.Sq * HD) q_s[t] = a.q[b * a.sqb + (long long)(t / HD) * a.ldq + hh * HD + (t % HD)] * a.scale;
  else if (t < FEWQ_MAX * HD) q_s[t] = 0.f;
  __syncthreads();
  float m[FEWQ_MAX], l[FEWQ_MAX];
  f32x4 o[FEWQ_MAX][4];
#pragma unroll
  for (int qi = 0; qi < FEWQ_MAX; ++qi) {
    m[qi] = -3.0e38f; l[qi] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[qi][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float* kp = a.k + b * a.skb + hh * HD;
  const float* vp = a.v + b * a.svb + hh * HD;
  for (int key = t; key < a.Sk; key += 256) {
    f32x4 kk[4], vv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      kk[c] = *(const f32x4*)(kp + (long long)key * a.ldk + 4 * c);
      vv[c] = *(const f32x4*)(vp + (long long)key * a.ldv + 4 * c);
    }
#pragma unroll
    for (int qi = 0; qi < FEWQ_MAX; ++qi) {
      if (qi < a.Sq) {     // uniform
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 qq = *(const f32x4*)(q_s + qi * HD + 4 * c);
          s += qq[0] * kk[c][0]; s += qq[1] * kk[c][1]; s += qq[2] * kk[c][2]; s += qq[3] * kk[c][3];
        }
        const float mn = fmaxf(m[qi], s);
        const float al = expf(m[qi] - mn), p = expf(s - mn);
        l[qi] = l[qi] * al + p;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[qi][c][e] = o[qi][c][e] * al + p * vv[c][e];
        m[qi] = mn;
      }
    }
  }
  // merge the 64 lanes of a wave
#pragma unroll
  for (int qi = 0; qi < FEWQ_MAX; ++qi) {
    if (qi < a.Sq) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float m2 = __shfl_xor(m[qi], off), l2 = __shfl_xor(l[qi], off);
        const float mn = fmaxf(m[qi], m2);
        const float a1 = expf(m[qi] - mn), a2 = expf(m2 - mn);
        l[qi] = l[qi] * a1 + l2 * a2;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[qi][c][e] = o[qi][c][e] * a1 + __shfl_xor(o[qi][c][e], off) * a2;
        m[qi] = mn;
      }
      if (lane == 0) {
        part[wave][qi][HD] = m[qi];
        part[wave][qi][HD + 1] = l[qi];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) part[wave][qi][4 * c + e] = o[qi][c][e];
      }
    }
  }
  __syncthreads();
  // the four waves: thread (qi, d) combines in wave order
  if (t < a.Sq * HD) {
    const int qi = t / HD, d = t % HD;
    float mm = part[0][qi][HD];
#pragma unroll
    for (int w2 = 1; w2 < 4; ++w2) mm = fmaxf(mm, part[w2][qi][HD]);
    float ll = 0.f, oo = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
      const float sc = expf(part[w2][qi][HD] - mm);
      ll += part[w2][qi][HD + 1] * sc;
      oo += part[w2][qi][d] * sc;
    }
    a.out[b * a.sob + (long long)qi * a.ldo + hh * HD + d] = oo / ll;
  }
}

// Few queries over many keys, chunked (the decoder's token -> image attention at its real size: 64..512 prompts x 4096 image
// tokens, 8 heads of 16; transformer.py:126-131).  attn_fewq_kernel keeps the state of all queries (7 x 18 floats) per thread,
// runs 8 waves per CU and reads 64-byte pieces at the row stride: 140 us for 268 MB (1.9 TB/s).  Here
//   * a workgroup owns 256 consecutive keys of one prompt and ALL heads: the 16 lanes (head, channel half) of a key read its
//     512-byte K row and V row contiguously;
//   * a lane owns 8 of a head's 16 channels (the two halves of a score meet through one DPP add): 7 x 10 floats of state,
//     ~110 registers, 16 waves per CU, the next key's 64 bytes in flight while the current one is multiplied;
//   * exponentials in base 2 (log2 e folded into the query scale): one v_exp_f32 each;
//   * the (max, sum, output) partials of a workgroup go to `part`; attn_fewq_combine_kernel merges the chunks of a prompt.
constexpr int FQC_KEYS = 256;                 // keys per workgroup
constexpr int FQC_QS = 7 * 16 + 4;            // floats between the heads of the query block in LDS (distinct bank groups)
constexpr int FQC_PART = 10;                  // floats of a partial: 8 channels, max, sum

__global__ __launch_bounds__(256, 4) void attn_fewq_chunk_kernel(FewQArgs a, float* __restrict__ part) {
  constexpr int NQ = 7;
  __shared__ __attribute__((aligned(16))) float q_s[8 * FQC_QS];
  __shared__ __attribute__((aligned(16))) float mrg[4][16][NQ][FQC_PART + 2];
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l16 = t & 15, c2 = t & 1, hh = (t >> 1) & 7, ks = t >> 4;
  constexpr float LOG2E = 1.4426950408889634f;
  for (int i = t; i < 8 * NQ * 16; i += 256) {
    const int head = i / (NQ * 16), rem = i - head * (NQ * 16), qi = rem >> 4, d = rem & 15;
    q_s[head * FQC_QS + rem] = qi < a.Sq ? a.q[b * a.sqb + (long long)qi * a.ldq + head * 16 + d] * (a.scale * LOG2E) : 0.f;
  }
  __syncthreads();
  float m[NQ], l[NQ];
  f32x4 o[NQ][2];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    m[qi] = -3.0e38f; l[qi] = 0.f;
    o[qi][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    o[qi][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float* kp = a.k + b * a.skb + hh * 16 + c2 * 8;
  const float* vp = a.v + b * a.svb + hh * 16 + c2 * 8;
  const float* qs = q_s + hh * FQC_QS + c2 * 8;
  const int k0 = chunk * FQC_KEYS + ks;
  f32x4 kn[2], vn[2];
  {
    const int key = k0 < a.Sk ? k0 : a.Sk - 1;
    kn[0] = *(const f32x4*)(kp + (long long)key * a.ldk); kn[1] = *(const f32x4*)(kp + (long long)key * a.ldk + 4);
    vn[0] = *(const f32x4*)(vp + (long long)key * a.ldv); vn[1] = *(const f32x4*)(vp + (long long)key * a.ldv + 4);
  }
#pragma unroll 2
  for (int pass = 0; pass < FQC_KEYS / 16; ++pass) {
    const int key = k0 + 16 * pass;
    const f32x4 kk0 = kn[0], kk1 = kn[1], vv0 = vn[0], vv1 = vn[1];
    {
      const int nk = key + 16 < a.Sk ? key + 16 : a.Sk - 1;
      kn[0] = *(const f32x4*)(kp + (long long)nk * a.ldk); kn[1] = *(const f32x4*)(kp + (long long)nk * a.ldk + 4);
      vn[0] = *(const f32x4*)(vp + (long long)nk * a.ldv); vn[1] = *(const f32x4*)(vp + (long long)nk * a.ldv + 4);
    }
    if (key < a.Sk) {
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        const f32x4 q0 = *(const f32x4*)(qs + qi * 16), q1 = *(const f32x4*)(qs + qi * 16 + 4);
        float s = q0[0] * kk0[0];
        s = fmaf(q0[1], kk0[1], s); s = fmaf(q0[2], kk0[2], s); s = fmaf(q0[3], kk0[3], s);
        s = fmaf(q1[0], kk1[0], s); s = fmaf(q1[1], kk1[1], s); s = fmaf(q1[2], kk1[2], s); s = fmaf(q1[3], kk1[3], s);
        s += __shfl_xor(s, 1);                      // the other half of the head's 16 channels
        const float mn = fmaxf(m[qi], s);
        const float al = __builtin_amdgcn_exp2f(m[qi] - mn), p = __builtin_amdgcn_exp2f(s - mn);
        l[qi] = fmaf(l[qi], al, p);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[qi][0][e] = fmaf(o[qi][0][e], al, p * vv0[e]);
          o[qi][1][e] = fmaf(o[qi][1][e], al, p * vv1[e]);
        }
        m[qi] = mn;
      }
    }
  }
  // merge the four keys a wave holds per (head, half) (lanes 16 and 32 apart), then the four waves through LDS
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      const float m2 = __shfl_xor(m[qi], off), l2 = __shfl_xor(l[qi], off);
      const float mn = fmaxf(m[qi], m2);
      const float a1 = __builtin_amdgcn_exp2f(m[qi] - mn), a2 = __builtin_amdgcn_exp2f(m2 - mn);
      l[qi] = l[qi] * a1 + l2 * a2;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[qi][c][e] = o[qi][c][e] * a1 + __shfl_xor(o[qi][c][e], off) * a2;
      m[qi] = mn;
    }
    if (lane < 16) {
      float* d = mrg[wave][l16][qi];
      *(f32x4*)d = o[qi][0];
      *(f32x4*)(d + 4) = o[qi][1];
      d[8] = m[qi];
      d[9] = l[qi];
    }
  }
  __syncthreads();
  if (t < 16 * NQ) {
    const int ll = t / NQ, qi = t - ll * NQ;
    float mm = mrg[0][ll][qi][8];
#pragma unroll
    for (int w2 = 1; w2 < 4; ++w2) mm = fmaxf(mm, mrg[w2][ll][qi][8]);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float lsum = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
      const float sc = __builtin_amdgcn_exp2f(mrg[w2][ll][qi][8] - mm);
      lsum += mrg[w2][ll][qi][9] * sc;
#pragma unroll
      for (int d = 0; d < 8; ++d) acc[d] += mrg[w2][ll][qi][d] * sc;
    }
    float* dst = part + ((((long long)b * nchunk + chunk) * 16 + ll) * NQ + qi) * FQC_PART;
#pragma unroll
    for (int d = 0; d < 8; ++d) dst[d] = acc[d];
    dst[8] = mm;
    dst[9] = lsum;
  }
}

// out[b, q, head * 16 + half * 8 + d] = sum_c o_c 2^(m_c - M) / sum_c l_c 2^(m_c - M) over the chunks of prompt b: one thread
// per output element, the chunk partials fetched sixteen at a time (independent loads: one round trip, not 2 x nchunk)
__global__ __launch_bounds__(128) void attn_fewq_combine_kernel(const float* __restrict__ part, int nchunk, int Sq, float* __restrict__ out,
                                                                 int ldo, long long sob) {
  constexpr int NQ = 7;
  const int b = blockIdx.y, i = blockIdx.x * 128 + threadIdx.x;
  if (i >= 16 * NQ * 8) return;
  const int d = i & 7, qi = (i >> 3) % NQ, ll = i / (8 * NQ);
  if (qi >= Sq) return;
  const float* src = part + (((long long)b * nchunk * 16 + ll) * NQ + qi) * FQC_PART;
  const long long cs = (long long)16 * NQ * FQC_PART;
  float mm = -3.0e38f, lsum = 0.f, acc = 0.f;
  for (int c0 = 0; c0 < nchunk; c0 += 16) {
    float mc[16], lc[16], oc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const bool in = c0 + c < nchunk;
      const float* s = src + (in ? c0 + c : c0) * cs;
      mc[c] = in ? s[8] : -3.0e38f;
      lc[c] = in ? s[9] : 0.f;
      oc[c] = in ? s[d] : 0.f;
    }
    float mn = mm;
#pragma unroll
    for (int c = 0; c < 16; ++c) mn = fmaxf(mn, mc[c]);
    const float a0 = __builtin_amdgcn_exp2f(mm - mn);
    lsum *= a0;
    acc *= a0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float sc = __builtin_amdgcn_exp2f(mc[c] - mn);
      lsum += lc[c] * sc;
      acc += oc[c] * sc;
    }
    mm = mn;
  }
  out[b * sob + (long long)qi * ldo + ll * 8 + d] = acc / lsum;
}

template <int HD>
int launch_hd(const AttnArgs& a, hipStream_t st) {
  dim3 grid((a.Sq + 127) / 128, a.B * a.H);
  HglProfScope prof(HGL_PROF_ATTN, 4.0 * a.B * a.H * (double)a.Sq * a.Sk * HD,
                    4.0 * a.B * a.H * HD * (2.0 * a.Sq + 2.0 * a.Sk), st);
  if (hgl_precision() == HGL_PREC_F16X3) {
    // sequences of 129..256 queries (a 14 x 14 window, a 197-token CLIP sequence): one 8-wave workgroup per (batch, head)
    static const int wide = HGL_DIAG_SWITCH("HGL_ATTN_WIDE", 1);
    static const int dual = HGL_DIAG_SWITCH("HGL_ATTN_DUAL", 1);
    // (longer sequences measured neutral for 785 queries, slower for the 4096-query global blocks: two independent
    // 4-wave workgroups per CU interleave their phases, one 8-wave workgroup meets at every barrier)
    // the persistent kernel parks the CLS-keep row of an item in a 256-byte LDS tail: keys beyond 257 do not fit there
    const bool w8 = wide && HD >= 64 && a.Sq > 128 && a.Sq <= 256 && a.mask_kind != HGL_MASK_CAUSAL &&
                    (a.mask_kind != HGL_MASK_CLS_KEEP || a.Sk <= 257);
    // long unmasked sequences (SAM's global blocks, GEM's 785 tokens): the ping-pong kernel, one 8-wave workgroup per 256 queries
    static const int pp = hgl_env_int("HGL_ATTN_PP", 1);
    if constexpr (HD == 64 || HD == 80) {
      // (measured, tools/attn_pp_ab.py: 4096 x 4096 x 80 with rel-pos 1690 against 1745 us; 785- and 1000-token sequences
      // 15-25 % SLOWER than two 4-wave workgroups per CU -- few query blocks per head, and the co-execution the schedule
      // is built for did not appear: SQ_VALU_MFMA_COEXEC stayed at 23 % of the matrix-busy cycles)
      if (pp && a.Sq >= 2048 && a.Sk >= 2048 && a.mask_kind == HGL_MASK_NONE && (!a.rel_h || (a.kw & 31) == 0) &&
          ((a.Sk & 31) == 0 || !a.rel_h)) {
        constexpr int KROW_ = 2 * HD + 8, VP_ = 96;
        constexpr size_t lds = (size_t)2 * KV_CHUNK * (KROW_ + 2 * VP_) * sizeof(_Float16);
        HGL_RESERVE_LDS((attn_x3pp_kernel<HD>), lds, "attention (ping-pong kernel)");
        hipLaunchKernelGGL((attn_x3pp_kernel<HD>), dim3((a.Sq + 255) / 256, a.B * a.H), dim3(512), lds, st, a);
        return hgl_check_launch("attention");
      }
    }
    if (HD == 80 && a.rel_h && a.kh == 14 && a.kw == 14 && a.Sk == 196 && a.mask_kind == HGL_MASK_NONE) {
      hipLaunchKernelGGL((attn_x3_kernel<HD, HD == 80 ? 14 : 0>), grid, dim3(256), 0, st, a);   // rel_h / rel_w given as tensors
    } else if (w8 && !a.rel_h && dual && HD == 64 && (a.mask_kind != HGL_MASK_CLS_KEEP || a.Sk <= 257)) {
      // head dim 64 (the CLIP sequences): two items per CU -- one 4-wave workgroup per item, two query tiles per wave.
      // 868 against 976 us on 1024 x 12 x 197 x 64 (141 against 125 TF/s), -0.4 ms per benchmark step (HGL_ATTN_DUAL=0: the
      // persistent 8-wave kernel)
      hipLaunchKernelGGL((attn_x3q_kernel<HD == 64 ? 64 : 16, 2>), dim3((unsigned)(a.B * a.H)), dim3(256), 0, st, a);
    } else if (w8 && !a.rel_h) {   // one 8-wave workgroup per item (K / V staged once)
      hipLaunchKernelGGL((attn_x3_kernel<HD, 0, 8>), dim3((a.Sq + 255) / 256, a.B * a.H), dim3(512), 0, st, a);
    } else {
      hipLaunchKernelGGL((attn_x3_kernel<HD, 0>), grid, dim3(256), 0, st, a);
    }
  }
  else hipLaunchKernelGGL(attn_f32_kernel<HD>, grid, dim3(256), 0, st, a);
  return hgl_check_launch("attention");
}

}  // namespace


// Windowed attention of the SAM encoder (14 x 14 windows, head dim 80, f16x3 mode) with the decomposed rel-pos terms
// computed INSIDE the kernel from the tables Rh / Rw [27, 80] (no rel_h / rel_w tensors, no separate table kernel).
// Returns HGL_EINVAL-free "not applicable" (1) when the shape is not the one this path serves.
int hgl_launch_attention_win14(const float* q, const float* k, const float* v, void* out_hi, void* out_lo, int B, int H, int hd,
                               int ldq, int ldk, int ldv, int ldo, long long sqb, long long skb, long long svb, long long sob,
                               float scale, const float* Rh, const float* Rw, hipStream_t st) {
  static const int wide = HGL_DIAG_SWITCH("HGL_ATTN_WIDE", 1);
  static const int fused = HGL_DIAG_SWITCH("HGL_ATTN_RELPOS_FUSED", 1);
  if (!wide || !fused || hd != 80 || hgl_precision() != HGL_PREC_F16X3 || !out_hi || !out_lo || !Rh || !Rw) return 1;
  HGL_REQUIRE(q && k && v && B > 0 && H > 0 && (long long)B * H <= 65535, "attention_win14: bad arguments");
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.out = nullptr;
  a.B = B; a.H = H; a.Sq = 196; a.Sk = 196;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.sqb = sqb; a.skb = skb; a.svb = svb; a.sob = sob;
  a.scale = scale; a.mask_kind = HGL_MASK_NONE; a.keep = nullptr; a.keep_b0 = 0; a.keep_n = B;
  a.rel_h = nullptr; a.rel_w = nullptr; a.kh = 14; a.kw = 14;
  a.tab_h = Rh; a.tab_w = Rw;
  {   // the tables' fp16 halves when the model registered them (unscaled: scale 2^0)
    const void *hh = nullptr, *hl = nullptr, *wh = nullptr, *wl = nullptr;
    int sh = 1, sw = 1, n1 = 0, k1 = 0, n2 = 0, k2 = 0;
    if (hgl_get_split_weight(Rh, &hh, &hl, &sh, &n1, &k1) && hgl_get_split_weight(Rw, &wh, &wl, &sw, &n2, &k2) && sh == 0 && sw == 0 &&
        n1 == 27 && n2 == 27 && k1 == 80 && k2 == 80) {
      a.tabh_hi = (const _Float16*)hh; a.tabh_lo = (const _Float16*)hl;
      a.tabw_hi = (const _Float16*)wh; a.tabw_lo = (const _Float16*)wl;
    }
  }
  a.out_hi = (_Float16*)out_hi; a.out_lo = (_Float16*)out_lo;
  HglProfScope prof(HGL_PROF_ATTN, 4.0 * B * H * 196.0 * 196.0 * 80, 0.0, st);
  hipLaunchKernelGGL((attn_x3_kernel<80, 14, 8>), dim3(1, (unsigned)(B * H)), dim3(512), 0, st, a);
  return hgl_check_launch("attention_win14");
}

// few-key attention (Sk <= 8, head dim 16): output fp32 (out) or the fp16 split pair (out == nullptr)
int hgl_launch_attention_smallk(const float* q, const float* k, const float* v, float* out, void* out_hi, void* out_lo,
                                int B, int H, int Sq, int Sk, int hd, int ldq, int ldk, int ldv, int ldo, long long sqb,
                                long long skb, long long svb, long long sob, float scale, hipStream_t st) {
  HGL_REQUIRE(q && k && v && (out || (out_hi && out_lo)), "attention_smallk: null operand");
  HGL_REQUIRE(hd == 16 && Sk >= 1 && Sk <= SMALLK_MAX && H >= 1 && 256 % H == 0, "attention_smallk: unsupported shape (hd %d, Sk %d, H %d)", hd, Sk, H);
  HGL_REQUIRE(((ldq | ldk | ldv | ldo) & 3) == 0 && ((sqb | skb | svb | sob) & 3) == 0, "attention_smallk: strides must be multiples of 4");
  HGL_REQUIRE(B <= 65535, "attention_smallk: B too large");
  SmallKArgs a;
  a.q = q; a.k = k; a.v = v; a.out = out; a.out_hi = (_Float16*)out_hi; a.out_lo = (_Float16*)out_lo;
  a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.sqb = sqb; a.skb = skb; a.svb = svb; a.sob = sob; a.scale = scale;
  const int qpb = 256 / H;
  HglProfScope prof(HGL_PROF_ATTN, 4.0 * B * (double)H * Sq * Sk * hd, 0.0, st);
  hipLaunchKernelGGL(attn_smallk_kernel, dim3((unsigned)((Sq + qpb - 1) / qpb), (unsigned)B), dim3(256),
                     (size_t)2 * Sk * H * 16 * sizeof(float), st, a);
  return hgl_check_launch("attention_smallk");
}

int hgl_launch_attention_fewq(const float* q, const float* k, const float* v, float* out, int B, int H, int Sq, int Sk,
                              int hd, int ldq, int ldk, int ldv, int ldo, long long sqb, long long skb, long long svb,
                              long long sob, float scale, hipStream_t st) {
  HGL_REQUIRE(q && k && v && out, "attention_fewq: null operand");
  HGL_REQUIRE(hd == 16 && Sq >= 1 && Sq <= FEWQ_MAX, "attention_fewq: unsupported shape (hd %d, Sq %d)", hd, Sq);
  HGL_REQUIRE(((ldk | ldv) & 3) == 0 && ((skb | svb) & 3) == 0, "attention_fewq: K/V strides must be multiples of 4");
  FewQArgs a;
  a.q = q; a.k = k; a.v = v; a.out = out; a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.sqb = sqb; a.skb = skb; a.svb = svb; a.sob = sob; a.scale = scale;
  HglProfScope prof(HGL_PROF_ATTN, 4.0 * B * (double)H * Sq * Sk * hd, 0.0, st);
  hipLaunchKernelGGL(attn_fewq_kernel, dim3((unsigned)(B * H)), dim3(256), 0, st, a);
  return hgl_check_launch("attention_fewq");
}

// chunked form for the decoder's sizes (8 heads of 16, up to 7 queries): `part` >= hgl_attention_fewq_part_bytes(B, Sk) bytes
size_t hgl_attention_fewq_part_bytes(int B, int Sk) {
  return (size_t)B * ((Sk + FQC_KEYS - 1) / FQC_KEYS) * 16 * 7 * FQC_PART * sizeof(float);
}

int hgl_launch_attention_fewq_chunked(const float* q, const float* k, const float* v, float* out, int B, int H, int Sq, int Sk,
                                      int hd, int ldq, int ldk, int ldv, int ldo, long long sqb, long long skb, long long svb,
                                      long long sob, float scale, float* part, size_t part_bytes, hipStream_t st) {
  HGL_REQUIRE(q && k && v && out && part, "attention_fewq_chunked: null operand");
  HGL_REQUIRE(hd == 16 && H == 8 && Sq >= 1 && Sq <= 7 && Sk >= 1 && B >= 1 && B <= 65535,
              "attention_fewq_chunked: unsupported shape (hd %d, H %d, Sq %d)", hd, H, Sq);
  HGL_REQUIRE(((ldk | ldv) & 3) == 0 && ((skb | svb) & 3) == 0 && (((uintptr_t)k | (uintptr_t)v) & 15) == 0,
              "attention_fewq_chunked: K/V strides must be multiples of 4");
  HGL_REQUIRE(part_bytes >= hgl_attention_fewq_part_bytes(B, Sk), "attention_fewq_chunked: partial buffer too small");
  FewQArgs a;
  a.q = q; a.k = k; a.v = v; a.out = out; a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.sqb = sqb; a.skb = skb; a.svb = svb; a.sob = sob; a.scale = scale;
  const int nchunk = (Sk + FQC_KEYS - 1) / FQC_KEYS;
  HglProfScope prof(HGL_PROF_ATTN, 4.0 * B * (double)H * Sq * Sk * hd, 0.0, st);
  hipLaunchKernelGGL(attn_fewq_chunk_kernel, dim3((unsigned)nchunk, (unsigned)B), dim3(256), 0, st, a, part);
  hipLaunchKernelGGL(attn_fewq_combine_kernel, dim3(7, (unsigned)B), dim3(128), 0, st, (const float*)part, nchunk, Sq, out, ldo, sob);
  return hgl_check_launch("attention_fewq_chunked");
}

int hgl_launch_attention(const float* q, const float* k, const float* v, float* out, int B, int H,
                         int Sq, int Sk, int hd, int ldq, int ldk, int ldv, int ldo, long long sqb,
                         long long skb, long long svb, long long sob, float scale, int mask_kind,
                         const uint8_t* keep, int keep_b0, int keep_n, const float* rel_h,
                         const float* rel_w, int kh, int kw, hipStream_t st) {
  return hgl_launch_attention_split(q, k, v, out, nullptr, nullptr, B, H, Sq, Sk, hd, ldq, ldk, ldv, ldo, sqb, skb, svb, sob,
                                    scale, mask_kind, keep, keep_b0, keep_n, rel_h, rel_w, kh, kw, st);
}

// out != nullptr: fp32 output; out == nullptr (f16x3 mode only): the fp16 hi+lo pair (out_hi, out_lo), same strides
int hgl_launch_attention_split(const float* q, const float* k, const float* v, float* out, void* out_hi, void* out_lo, int B,
                               int H, int Sq, int Sk, int hd, int ldq, int ldk, int ldv, int ldo, long long sqb,
                               long long skb, long long svb, long long sob, float scale, int mask_kind,
                               const uint8_t* keep, int keep_b0, int keep_n, const float* rel_h,
                               const float* rel_w, int kh, int kw, hipStream_t st) {
  HGL_REQUIRE(q && k && v && (out || (out_hi && out_lo)), "attention: null operand");
  HGL_REQUIRE(out || hgl_precision() == HGL_PREC_F16X3, "attention: split output exists in f16x3 mode only");
  HGL_REQUIRE(B > 0 && H > 0 && Sq > 0 && Sk > 0, "attention: bad shape");
  HGL_REQUIRE((ldq & 3) == 0 && (ldk & 3) == 0 && (ldv & 3) == 0 && (ldo & 3) == 0, "attention: leading dims must be multiples of 4");
  HGL_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)out_hi | (uintptr_t)out_lo) & 15) == 0,
              "attention: operands must be 16-byte aligned");
  HGL_REQUIRE(((sqb | skb | svb | sob) & 3) == 0, "attention: batch strides must be multiples of 4");
  HGL_REQUIRE(mask_kind >= 0 && mask_kind <= 2, "attention: bad mask kind %d", mask_kind);
  HGL_REQUIRE(mask_kind != HGL_MASK_CLS_KEEP || keep, "attention: HGL_MASK_CLS_KEEP needs keep bytes");
  HGL_REQUIRE((rel_h == nullptr) == (rel_w == nullptr), "attention: rel_h and rel_w go together");
  HGL_REQUIRE(!rel_h || (kh > 0 && kw > 0 && kh * kw == Sk), "attention: kh*kw must equal Sk");
  HGL_REQUIRE((long long)B * H <= 65535, "attention: B*H too large for grid.y");
  if (out && hd == 16 && Sq <= FEWQ_MAX && Sk >= 1024 && mask_kind == HGL_MASK_NONE && !rel_h)
    return hgl_launch_attention_fewq(q, k, v, out, B, H, Sq, Sk, hd, ldq, ldk, ldv, ldo, sqb, skb, svb, sob, scale, st);
  if (hd == 16 && Sk <= SMALLK_MAX && Sq >= 256 && mask_kind == HGL_MASK_NONE && !rel_h && 256 % H == 0)
    return hgl_launch_attention_smallk(q, k, v, out, out_hi, out_lo, B, H, Sq, Sk, hd, ldq, ldk, ldv, ldo, sqb, skb, svb,
                                       sob, scale, st);
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.out = out;
  a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.sqb = sqb; a.skb = skb; a.svb = svb; a.sob = sob;
  a.scale = scale; a.mask_kind = mask_kind; a.keep = keep; a.keep_b0 = keep_b0; a.keep_n = keep_n > 0 ? keep_n : B;
  a.rel_h = rel_h; a.rel_w = rel_w; a.kh = kh; a.kw = kw;
  a.tab_h = nullptr; a.tab_w = nullptr;
  a.out_hi = (_Float16*)out_hi; a.out_lo = (_Float16*)out_lo;
  switch (hd) {
    case 16: return launch_hd<16>(a, st);
    case 32: return launch_hd<32>(a, st);
    case 64: return launch_hd<64>(a, st);
    case 80: return launch_hd<80>(a, st);
    default:
      hgl_set_error("attention: unsupported head dim %d (16,32,64,80)", hd);
      return HGL_EINVAL;
  }
}

HGL_DEFINE_SPLIT_OVERFLOW_READER(hgl_split_overflow_attention)
