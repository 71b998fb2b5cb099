// Fused attention on PRE-SPLIT operands (f16x3 mode): q, k, v arrive as the fp16 hi / lo planes that the in-projection GEMM's
// write-out emits (gemm_f16x3.hip: C == nullptr -> Ch / Cl), so this kernel converts nothing on the way in.
//
// Replaces, for the shapes that carry the step, attn_x3_kernel<80,14,8> (SAM's 14 x 14 windows, image_encoder.py:224-240 with
// the decomposed rel-pos terms of :325-361 computed in the kernel), attn_x3pp_kernel<80> (SAM's global blocks, rel-pos terms
// given as tensors) and attn_x3q_kernel<64,2> (CLIP's 197-token sequences, clip/model.py:209, with the CLS-row keep mask of
// model/backbone.py:108-115).  Counters of the windowed kernel (profiles/r04h_sq_counters_attn_x3_win14_6400items.json):
// 9.4 vector instructions per matrix instruction, matrix pipe 24 % busy, waves parked 40 % of their cycles, ONE 8-wave
// workgroup per CU (238 VGPRs, 118 KB of LDS) -- so nothing covered an item's head (Q / K / V round trips, 1300 of the 2300
// vector instructions of a wave and item were fp32 -> fp16 hi / lo conversions of Q, K and V) or its eight chunk barriers.
//
// Here:
//   * no conversion work: Q fragments are 16-byte loads of the planes in MFMA operand layout; K / V tiles go global -> LDS by
//     LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write pass, no vector instruction but the address);
//   * 4-wave workgroups, a wave owns QT tiles of 32 queries, 32-key chunks double-buffered by the DMA with ONE barrier per
//     chunk; <= 66 KB of LDS and <= 256 VGPRs: TWO workgroups per CU whose phases are independent (one's prologue, barrier
//     waits and write-out run beside the other's MFMAs);
//   * windows: two workgroups per (window, head) (query tiles 0-3 / 4-6), placed on the same XCD so that the second finds the
//     item's K / V planes in L2; staging twice costs DMA bytes only.
// The arithmetic of a (query tile, key tile) pair is attn_x3_kernel's, operation for operation (swapped QK^T, scale in the
// exponent, lazy rescaling, P split by a packed round-toward-zero conversion, P^T accumulators as the B operand of P V through
// the transposing LDS read): results are bit-identical to that kernel on the same hi / lo planes.
#include "hgl_common.h"
#include <stdlib.h>
#include <mutex>
#include <type_traits>

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x4v __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int PS_NEG_BIG_BITS = 0xff800000;  // -inf
constexpr int PS_CHUNK = 32;                 // keys per staged chunk = one key tile

enum { PS_PLAIN = 0, PS_WIN14 = 1, PS_RELT = 2 };

struct PsArgs {
  const _Float16 *hi, *lo;      // the qkv planes: row (b * sb + s), column xcol + head * HD; leading dimension ld (halfs)
  int ld, qcol, kcol, vcol;
  long long sb;                 // rows between two batch elements
  int B, H, S;
  float* out;                   // fp32 output, or nullptr: the fp16 pair below (the operand form of the projection GEMM)
  _Float16 *out_hi, *out_lo;
  int ldo;
  long long sob;
  float scale;
  int mask_kind;                // HGL_MASK_NONE / HGL_MASK_CLS_KEEP
  const uint8_t* keep;
  int keep_b0, keep_n;
  const float *rel_h, *rel_w;   // PS_RELT: [B*H, S, kh] / [B*H, S, kw] fp32
  int kh, kw;
  const _Float16 *tabh_hi, *tabh_lo, *tabw_hi, *tabw_lo;   // PS_WIN14: the [27, 80] tables split once per model (scale 2^0)
  int nqb;                      // workgroups (blocks of 128 * QT queries) per item
  int dbg;                      // diagnostic build only (HGL_ATTN_PS_DBG; bit 0: no QK^T products, 1: no P V products, 2: no K / V requests, 3: no stores, 4: no rel-pos prologue)
};
// The knock-outs for timing experiments are compiled into the diagnostic twin only (make diag): in the product PS_DBG is the
// constant 0, the kernels carry no such branch and no environment variable can make them skip work.
#ifdef HGL_DIAG
#define PS_DBG(a) ((a).dbg)
#else
#define PS_DBG(a) 0
#endif

// ds_read_b64_tr_b16 (EXEC must be all ones at the call)
__device__ __forceinline__ h16x4 ps_tr4(const _Float16* p) {
  typedef __attribute__((address_space(3))) fp16x4v lds_v;
  const fp16x4v v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_v*)p);
  h16x4 r;
  __builtin_memcpy(&r, &v, 8);
  return r;
}

// One LDS-DMA wave-instruction: lane i copies 16 B from sbase + voff(i) to LDS byte lds_addr + 16 * i (see gemm_f16x3.hip).
// Not counted by the compiler on vmcnt: the consumer waits with an explicit s_waitcnt.
__device__ __forceinline__ void ps_glds16(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_addr)
               : "memory");
}

// In-kernel cycle stamps (tools/attn_ps_stamps.py; compiled in with -DHGL_PS_STAMPS only): one wave of one workgroup
#ifdef HGL_PS_STAMPS
__device__ unsigned long long g_ps_stamps[1024];
__device__ int g_ps_stamp_sel[2] = {0, 0};   // workgroup, wave
#define PS_STAMP(id) do { if ((int)blockIdx.x == g_ps_stamp_sel[0] && t == 64 * g_ps_stamp_sel[1] && nst < 1000) { g_ps_stamps[nst++] = ((unsigned long long)(id) << 48) | ((unsigned long long)clock64() & 0xffffffffffffull); } } while (0)
#else
#define PS_STAMP(id) do {} while (0)
#endif

template <int HD>
struct PsGeom {
  static constexpr int KS = HD / 16;                    // k-steps of the QK^T contraction
  static constexpr int DT = (HD + 31) / 32;             // 32-wide d tiles of the output
  static constexpr int KSL = HD / 4 + 1;                // 16-byte slots of a staged K row: hi | lo | pad (odd: conflict-free b128 reads)
  static constexpr int KP = KSL * 8;                    // halfs per staged K row
  static constexpr int VSL = 12;                        // slots of a staged V row (192 B: the transposing reads' pitch)
  static constexpr int VP = VSL * 8;                    // halfs
  static constexpr int NSLOT = PS_CHUNK * (KSL + 2 * VSL);
  static constexpr int NI = (NSLOT + 63) / 64;          // DMA wave-instructions per chunk
  static constexpr int NJ = (NI + 3) / 4;               // per wave
  static constexpr int STAGE = NI * 1024;               // bytes
  static constexpr int K_OFF = 0, VH_OFF = PS_CHUNK * KSL * 16, VL_OFF = VH_OFF + PS_CHUNK * VSL * 16;
  static_assert(DT * 32 <= VP, "V row pitch");
};
constexpr int PS_EP = 40;            // halfs per row of the indicator table (80 B: conflict-free b128 reads)
constexpr int PS_EROWS = 224;        // 7 key tiles

template <int HD, int MODE>
constexpr size_t ps_lds_bytes() {
  return (size_t)2 * PsGeom<HD>::STAGE + (MODE == PS_WIN14 ? PS_EROWS * PS_EP * 2 : 0) + 256;
}

template <int HD, int MODE, int QT>
__global__ __launch_bounds__(256, 2) void attn_ps_kernel(PsArgs a) {
  using G = PsGeom<HD>;
  constexpr int KS = G::KS, DT = G::DT, KP = G::KP, VP = G::VP;
  extern __shared__ __attribute__((aligned(1024))) unsigned char ps_smem[];
  static_assert(MODE == PS_PLAIN, "the un-pipelined kernel serves the plain / CLS-keep shapes only (rel-pos: attn_psp_kernel)");
  uint8_t* const keepL = ps_smem + 2 * G::STAGE;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, h = lane >> 5;
#ifdef HGL_PS_STAMPS
  int nst = 0;
#endif
  PS_STAMP(1);
  // XCD-aware item map: workgroups go to the eight XCDs round-robin in linear order; consecutive slots of ONE XCD are the
  // query blocks of one item (they stream the same K / V), items are dealt to the XCDs round-robin.
  int item, bx;
  {
    const unsigned L = blockIdx.x, nqb = (unsigned)a.nqb, items = (unsigned)(a.B * a.H);
    if ((items & 7u) == 0) {
      const unsigned c = L & 7u, j = L >> 3;
      item = (int)((j / nqb) * 8u + c);
      bx = (int)(j % nqb);
    } else {
      item = (int)(L / nqb);
      bx = (int)(L % nqb);
    }
  }
  const int b = item / a.H, hh = item - b * a.H;
  const float NEG_INF = __int_as_float(PS_NEG_BIG_BITS);
  constexpr float LOG2E = 1.4426950408889634f;
  const float sl2e = a.scale * LOG2E;
  const float inv_scale = 1.0f / a.scale;
  const float rescale_thr = 5.5f * inv_scale;     // 5.5 nats ~ 2^8, in units of the unscaled scores

  // ---- DMA plan: slot n = 64 * i + lane of instruction i = wave + 4 * j -> (plane, key row of the chunk, 16-byte column) ----
  const unsigned ldb = (unsigned)a.ld * 2u;
  const long long lo_delta_ll = (const char*)a.lo - (const char*)a.hi;   // checked by the launcher to fit 32 bits with the offsets
  const unsigned lo_delta = (unsigned)lo_delta_ll;
  const char* const kv_base = (const char*)a.hi + (long long)b * a.sb * (long long)ldb;
  unsigned d_row[G::NJ], d_col[G::NJ];
#pragma unroll
  for (int j = 0; j < G::NJ; ++j) {
    const int n = 64 * (wave + 4 * j) + lane;
    int row = 0;
    unsigned col = (unsigned)(a.kcol + hh * HD) * 2u;    // pad slots and slots beyond the image fetch a valid, unused piece
    if (n < PS_CHUNK * G::KSL) {
      row = n / G::KSL;
      const int c = n - row * G::KSL;
      if (c < HD / 8) col = (unsigned)(a.kcol + hh * HD + 8 * c) * 2u;
      else if (c < HD / 4) col = (unsigned)(a.kcol + hh * HD + 8 * (c - HD / 8)) * 2u + lo_delta;
    } else if (n < G::NSLOT) {
      const int n2 = n - PS_CHUNK * G::KSL;
      const int pl = n2 / (PS_CHUNK * G::VSL), n3 = n2 - pl * (PS_CHUNK * G::VSL);
      row = n3 / G::VSL;
      const int c = n3 - row * G::VSL;
      if (c < HD / 8) col = (unsigned)(a.vcol + hh * HD + 8 * c) * 2u + (pl ? lo_delta : 0u);
    }
    d_row[j] = (unsigned)row;
    d_col[j] = col;
  }
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)ps_smem;
  auto issue_chunk = [&](int ci) {
    const unsigned sbase = lds0 + (unsigned)(ci & 1) * G::STAGE;
#pragma unroll
    for (int j = 0; j < G::NJ; ++j) {
      const int i = wave + 4 * j;
      if (i < G::NI) {   // uniform
        const unsigned krow = min((unsigned)(ci * PS_CHUNK) + d_row[j], (unsigned)(a.S - 1));
        ps_glds16(kv_base, krow * ldb + d_col[j], sbase + (unsigned)i * 1024u);
      }
    }
  };
  issue_chunk(0);      // requested first: its round trip overlaps the Q loads and the rel-pos table products
  PS_STAMP(2);

  // ---- CLS keep row / indicator table ----
  const uint8_t* keep_row = nullptr;
  if (MODE == PS_PLAIN && a.mask_kind == HGL_MASK_CLS_KEEP && b >= a.keep_b0)
    keep_row = a.keep + (long long)((b - a.keep_b0) % a.keep_n) * (a.S - 1);
  if (keep_row && t < a.S - 1) keepL[t] = keep_row[t];

  // ---- Q fragments: lane (r, h) element j of k-step s = Q[q][16 s + 8 h + j], straight from the planes ----
  int qi[QT];
  bool qvalid[QT], tile_active[QT];
  h16x8 qh[QT][KS], ql[QT][KS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q0 = ((bx * 4 + wave) * QT + qt) * 32;
    qi[qt] = q0 + r;
    qvalid[qt] = qi[qt] < a.S;
    tile_active[qt] = q0 < a.S;
    const long long qo = ((long long)b * a.sb + (qvalid[qt] ? qi[qt] : 0)) * a.ld + a.qcol + hh * HD + 8 * h;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      qh[qt][s] = *(const h16x8*)(a.hi + qo + 16 * s);
      ql[qt][s] = *(const h16x8*)(a.lo + qo + 16 * s);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    if (!qvalid[qt]) {
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) { qh[qt][s][e] = (_Float16)0.f; ql[qt][s][e] = (_Float16)0.f; }
    }
  }
  const bool wave_active = tile_active[0];
  PS_STAMP(3);

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

  PS_STAMP(4);

  // The compiler does not see the DMA pieces on vmcnt: wherever IT waits for one of its own loads inside the loop it waits
  // with a count that ignores them, i.e. in effect for the chunk just requested (first version: `s_waitcnt vmcnt(0)` in the
  // middle of every tile's QK^T products -- for Q fragments that had landed long before -- exposed every chunk's round trip).
  // So every register a compiler-counted load fills is "used" here, before the loop: its wait lands here.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int sx = 0; sx < KS; ++sx) asm volatile("" : "+v"(qh[qt][sx]), "+v"(ql[qt][sx]));

  // transposed-read addressing (attn_x3_kernel): lane = 16*grp + 4*q + p supplies row q, columns 4p..4p+3 of its group's block
  const int tr_off = (((lane >> 2) & 3) + 4 * h) * VP + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  const int nchunk = (a.S + PS_CHUNK - 1) / PS_CHUNK;
  for (int ci = 0; ci < nchunk; ++ci) {
    // this wave's pieces of chunk ci have landed; behind the barrier so have everyone's, and every wave has finished reading
    // chunk ci - 1, whose stage the next DMA overwrites (PS_WIN14, ci == 0: every wave has finished with its rel-pos patch)
    PS_STAMP(10);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PS_STAMP(11);
    __syncthreads();
    PS_STAMP(12);
    if (ci + 1 < nchunk) issue_chunk(ci + 1);
    PS_STAMP(13);
    if (!wave_active) continue;   // uniform: a wave beyond the sequence only helps staging
    const unsigned char* const st = ps_smem + (ci & 1) * G::STAGE;
    const _Float16* const Ks = (const _Float16*)(st + G::K_OFF);
    const _Float16* const Vh = (const _Float16*)(st + G::VH_OFF);
    const _Float16* const Vl = (const _Float16*)(st + G::VL_OFF);
    const int kbase = ci * PS_CHUNK;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      if (!tile_active[qt]) continue;   // uniform per wave
      f32x16 s;
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = 0.f;
      const _Float16* krow = Ks + r * KP + 8 * h;
#pragma unroll
      for (int c = 0; c < KS; ++c) {
        const h16x8 kh8 = *(const h16x8*)(krow + 16 * c);
        const h16x8 kl8 = *(const h16x8*)(krow + HD + 16 * c);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl8, qh[qt][c], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, ql[qt][c], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, qh[qt][c], s, 0, 0, 0);
      }
      // s[e] = S^T[key = kbase + (e&3) + 8*(e>>2) + 4*h][query], unscaled; scale and log2(e) are folded into the exponent's fma
      PS_STAMP(14);
      float mx = NEG_INF;
      if (kbase + 32 > a.S) {   // uniform: the tile that crosses the end of the sequence
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int kg = kbase + (e & 3) + 8 * (e >> 2) + 4 * h;
          s[e] = kg >= a.S ? NEG_INF : s[e];
        }
      }
      if (MODE == PS_PLAIN && keep_row && bx == 0 && wave == 0 && qt == 0) {   // uniform: the tile that owns query 0
        const int kk = kbase + (lane & 31);
        const unsigned kb = kk >= 1 && kk < a.S ? keepL[kk - 1] : 1u;
        const unsigned bits = (unsigned)__builtin_amdgcn_ballot_w64(kb != 0) >> (4 * h);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const bool kept = (bits >> ((e & 3) + 8 * (e >> 2))) & 1u;
          s[e] = (qi[0] == 0 && !kept) ? NEG_INF : s[e];
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[e]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      // lazy rescaling (attn_x3_kernel): the running maximum follows the scores only when some query of the wave would
      // otherwise see probabilities above 2^8 (uniform branch)
      const float m_cand = fmaxf(m_run[qt], mx);
      float m_new = m_run[qt];
      if (__builtin_amdgcn_ballot_w64(m_cand > m_run[qt] + rescale_thr)) {
        m_new = m_cand;
        const float m_use0 = (m_new == NEG_INF) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_use0) * sl2e);
        l_run[qt] *= alpha;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[qt][d][e] *= alpha;
        m_run[qt] = m_new;
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
      l_run[qt] += rs;
      PS_STAMP(15);
      // the scores are dead from here on: the next tile's rel-pos terms travel under the P V products
      // O^T += V^T P^T ; A operand element j of lane (d, h) = V^T[d][16*s2 + 8*(j>>2) + 4*h + (j&3)]
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if (kbase + 16 * s2 >= a.S) break;   // uniform: the keys of this k-step are all beyond the sequence (P = 0)
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const int off = (16 * s2) * VP + d * 32 + tr_off;
          const h16x4 vh0 = ps_tr4(Vh + off), vh1 = ps_tr4(Vh + off + 8 * VP);
          const h16x4 vl0 = ps_tr4(Vl + off), vl1 = ps_tr4(Vl + off + 8 * VP);
          h16x8 vh8, vl8;
#pragma unroll
          for (int e = 0; e < 4; ++e) { vh8[e] = vh0[e]; vh8[4 + e] = vh1[e]; vl8[e] = vl0[e]; vl8[4 + e] = vl1[e]; }
          o[qt][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl8, ph[s2], o[qt][d], 0, 0, 0);
          o[qt][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, pl[s2], o[qt][d], 0, 0, 0);
          o[qt][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, ph[s2], o[qt][d], 0, 0, 0);
        }
      }
      PS_STAMP(16);
    }
  }
  PS_STAMP(20);

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
  PS_STAMP(21);
}

// ---------------------------------------------------------------------------------------------
// SOFTWARE-PIPELINED form (one query tile per wave).  Counters and stamps of attn_ps_kernel on the windows
// (profiles/r05a_*): two independent workgroups per CU do NOT interleave their phases -- the two waves of a SIMD run the same
// program and fall into step (both in their MFMA chains, then both in their soft-max), the matrix pipe was 27 % busy and a
// wave's 2 x 600 cycles of dependent MFMAs per tile had nothing of its own beside them.  An in-order wave overlaps vector
// and matrix work only when they ALTERNATE in its instruction stream, so the loop is rotated by one tile:
//
//     iteration t :   [ QK^T(t+1) MFMAs  ||  soft-max(t): max, exponentials, hi / lo split of P ]   ->   P V (t)
//
// both halves of the first segment sit in ONE basic block (the score accumulators exist twice), the rare rescaling of O
// (uniform branch) is applied between the two segments.  K chunks live in a ring of three stages (chunk t+2 is requested
// while chunk t+1 is multiplied), V chunks in a ring of two; one barrier per chunk as before.
// the other half-wave's value of x (lanes l and l ^ 32 hold the two key halves of one query): one v_permlane32_swap instead of
// a ds_bpermute round trip
__device__ __forceinline__ float ps_max_halves(float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // {lower half's values everywhere, upper half's everywhere}
  return fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
}
__device__ __forceinline__ float ps_add_halves(float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
}
template <int HD>
struct PspGeom {
  static constexpr int KS = HD / 16, DT = (HD + 31) / 32;
  static constexpr int KSL = HD / 4 + 1, KP = KSL * 8;          // slots / halfs of a staged K row: hi | lo | pad
  static constexpr int VSL = 12, VP = VSL * 8;                  // slots / halfs of a staged V row
  static constexpr int NIK = (PS_CHUNK * KSL + 63) / 64;        // DMA wave-instructions of a K chunk
  static constexpr int NJK = (NIK + 3) / 4;
  static constexpr int KSTAGE = NIK * 1024;
  static constexpr int NIV = 2 * PS_CHUNK * VSL / 64;           // of a V chunk (hi plane, lo plane): 12
  static constexpr int NJV = NIV / 4;
  static constexpr int VSTAGE = NIV * 1024;
  static constexpr int VL_OFF = PS_CHUNK * VSL * 16;
  static constexpr int K_RING = 0, V_RING = 3 * KSTAGE, E_OFF = V_RING + 2 * VSTAGE;
  static_assert(DT * 32 <= VP && NIV % 4 == 0, "geometry");
};
constexpr int PSP_EBYTES = PS_EROWS * 64;    // the indicator table, 64-byte rows, 16-byte chunks XOR-swizzled with (row >> 2) & 3

template <int HD, int MODE>
constexpr size_t psp_lds_bytes() {
  return (size_t)PspGeom<HD>::E_OFF + (MODE == PS_WIN14 ? PSP_EBYTES : 0) + 256;
}

template <int HD, int MODE, int QT = 1>
__global__ __launch_bounds__(256, 2) void attn_psp_kernel(PsArgs a, const unsigned char* __restrict__ etab) {
  static_assert(QT == 1 || (QT == 2 && MODE == PS_PLAIN), "two query tiles per wave: the plain mode only");
  using G = PspGeom<HD>;
  constexpr int KS = G::KS, DT = G::DT, KP = G::KP, VP = G::VP;
  extern __shared__ __attribute__((aligned(1024))) unsigned char ps_smem[];
  uint8_t* const keepL = ps_smem + G::E_OFF + (MODE == PS_WIN14 ? PSP_EBYTES : 0);

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, h = lane >> 5;
  const float NEG_INF = __int_as_float(PS_NEG_BIG_BITS);
  constexpr float LOG2E = 1.4426950408889634f;
  const float sl2e = a.scale * LOG2E;
  const float inv_scale = 1.0f / a.scale;
  const float rescale_thr = 5.5f * inv_scale;

  // ---- DMA plan.  K image: 32 rows of KSL slots (hi | lo | pad); V image: hi plane, lo plane of 32 rows x 12 slots ----
  const unsigned ldb = (unsigned)a.ld * 2u;
  const unsigned lo_delta = (unsigned)((const char*)a.lo - (const char*)a.hi);
  const unsigned vrel = (unsigned)(a.vcol - a.kcol) * 2u;    // V columns relative to the K columns of the same head (vcol >= kcol)
  unsigned k_row[G::NJK], k_col[G::NJK], v_row[G::NJV], v_col[G::NJV];
#pragma unroll
  for (int j = 0; j < G::NJK; ++j) {
    const unsigned n = 64u * (unsigned)(wave + 4 * j) + (unsigned)lane;
    unsigned row = n / (unsigned)G::KSL;
    const unsigned c = n - row * (unsigned)G::KSL;
    unsigned col = 0;                                       // pad slot / beyond the image: a valid, unused piece
    if (c < (unsigned)(HD / 8)) col = 16u * c;
    else if (c < (unsigned)(HD / 4)) col = 16u * (c - (unsigned)(HD / 8)) + lo_delta;
    if (row >= (unsigned)PS_CHUNK) row = 0;
    k_row[j] = row;
    k_col[j] = col;
  }
#pragma unroll
  for (int j = 0; j < G::NJV; ++j) {
    const unsigned n = 64u * (unsigned)(wave + 4 * j) + (unsigned)lane;
    const unsigned pl = n / (unsigned)(PS_CHUNK * G::VSL), n3 = n - pl * (unsigned)(PS_CHUNK * G::VSL);
    const unsigned row = n3 / (unsigned)G::VSL, c = n3 - row * (unsigned)G::VSL;
    v_row[j] = row;
    v_col[j] = vrel + 16u * (c < (unsigned)(HD / 8) ? c : 0u) + (pl ? lo_delta : 0u);
  }
  const char* kv_base = nullptr;     // of the current item: the hi plane at its first key row, K columns of its head
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)ps_smem;
  auto issue_k = [&](int ci) {
    const unsigned sbase = lds0 + G::K_RING + (unsigned)(ci % 3) * G::KSTAGE;
#pragma unroll
    for (int j = 0; j < G::NJK; ++j) {
      const int i = wave + 4 * j;
      if (i < G::NIK) {   // uniform
        const unsigned krow = min((unsigned)(ci * PS_CHUNK) + k_row[j], (unsigned)(a.S - 1));
        ps_glds16(kv_base, krow * ldb + k_col[j], sbase + (unsigned)i * 1024u);
      }
    }
  };
  auto issue_v = [&](int ci) {
    const unsigned sbase = lds0 + G::V_RING + (unsigned)(ci & 1) * G::VSTAGE;
#pragma unroll
    for (int j = 0; j < G::NJV; ++j) {
      const unsigned krow = min((unsigned)(ci * PS_CHUNK) + v_row[j], (unsigned)(a.S - 1));
      ps_glds16(kv_base, krow * ldb + v_col[j], sbase + (unsigned)(wave + 4 * j) * 1024u);
    }
  };
  const int nchunk = (a.S + PS_CHUNK - 1) / PS_CHUNK;
  if constexpr (MODE == PS_WIN14) {   // the indicator table, once per workgroup: a linear copy of the (already swizzled) global image
    for (int i = wave; i < PSP_EBYTES / 1024; i += 4) ps_glds16(etab, (unsigned)(i * 1024 + lane * 16), lds0 + G::E_OFF + (unsigned)i * 1024u);
  }
  const int tr_off = (((lane >> 2) & 3) + 4 * h) * VP + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int k_off = r * KP + 8 * h;                                   // halfs, within a K stage
  int e_off[2];                                                       // bytes, within the table, of key row r (kbase % 32 == 0)
#pragma unroll
  for (int c = 0; c < 2; ++c) e_off[c] = r * 64 + (((2 * c + h) ^ ((r >> 2) & 3)) * 16);

  // PERSISTENT: the workgroup walks the work units (item, query block) L = blockIdx.x, + gridDim.x, ... (gridDim.x % 8 == 0, so
  // a unit's XCD is the workgroup's: consecutive slots of one XCD are the query blocks of one item, see attn_ps_kernel).
  // One workgroup per unit left 18 % of the wave slots empty between a workgroup's end and its successor's first instruction
  // (profiles/r05a_sq_counters_attn_psp_win14.json: SQ_WAVE_CYCLES against the launch's cycles), and paid the DMA plan and the
  // table once per unit.
  const unsigned nunits = (unsigned)(a.B * a.H) * (unsigned)a.nqb;
  for (unsigned L = blockIdx.x; L < nunits; L += gridDim.x) {
  // Unit -> (batch element, head, query block).  The heads of one batch element are ADJACENT 2*HD-byte pieces of the same
  // rows of the planes: dealt to different XCDs (first version: item = batch * H + head round-robin over the XCDs) every L2
  // fetches whole 128-byte lines for its two heads' 160-byte pieces (FETCH_SIZE 2.1 GB for 1.2 GB of q | k | v on the
  // windows).  So an XCD owns whole batch elements: XCD c works through elements c, c + 8, ...; consecutive slots are the
  // query blocks of one head, then the next head.  (The launch time did not move with it: the kernel is not bound by HBM.)
  int item, bx;
  {
    const unsigned nqb = (unsigned)a.nqb;
    if ((a.B & 7) == 0) {
      const unsigned c = L & 7u, j = L >> 3;
      const unsigned per_b = (unsigned)a.H * nqb;
      const unsigned bl = j / per_b, rem = j - bl * per_b;
      item = (int)((bl * 8u + c) * (unsigned)a.H + rem / nqb);
      bx = (int)(rem % nqb);
    } else {
      item = (int)(L / nqb);
      bx = (int)(L % nqb);
    }
  }
  const int b = item / a.H, hh = item - b * a.H;
  kv_base = (const char*)a.hi + ((long long)b * a.sb * (long long)a.ld + a.kcol + hh * HD) * 2;
  if (L != blockIdx.x) __syncthreads();   // every wave has left the previous unit: its V stage and the rel-pos patches are free
  if (!(PS_DBG(a) & 4)) {
  issue_k(0);
  issue_v(0);
  if (nchunk > 1) issue_k(1);
  }

  const uint8_t* keep_row = nullptr;
  if (MODE == PS_PLAIN && a.mask_kind == HGL_MASK_CLS_KEEP && b >= a.keep_b0)
    keep_row = a.keep + (long long)((b - a.keep_b0) % a.keep_n) * (a.S - 1);
  if (keep_row && t < a.S - 1) keepL[t] = keep_row[t];

  // ---- Q fragments of the wave's QT query tiles ----
  const int q0 = (bx * 4 + wave) * QT * 32;
  int qi[QT];
  bool qvalid[QT], tile_active[QT];
  h16x8 qh[QT][KS], ql[QT][KS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    qi[qt] = q0 + 32 * qt + r;
    qvalid[qt] = qi[qt] < a.S;
    tile_active[qt] = q0 + 32 * qt < a.S;
    const long long qo = ((long long)b * a.sb + (qvalid[qt] ? qi[qt] : 0)) * a.ld + a.qcol + hh * HD + 8 * h;
#pragma unroll
    for (int sx = 0; sx < KS; ++sx) {
      qh[qt][sx] = *(const h16x8*)(a.hi + qo + 16 * sx);
      ql[qt][sx] = *(const h16x8*)(a.lo + qo + 16 * sx);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    if (!qvalid[qt]) {
#pragma unroll
      for (int sx = 0; sx < KS; ++sx)
#pragma unroll
        for (int e = 0; e < 8; ++e) { qh[qt][sx][e] = (_Float16)0.f; ql[qt][sx][e] = (_Float16)0.f; }
    }
  }
  const bool wave_active = tile_active[0];

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

  // ---- PS_WIN14: R fragments of the MFMA bias (see attn_ps_kernel); the per-wave patch aliases ring stages that receive
  // their first DMA behind the loop's first barrier (waves 0-1: K stage 2, waves 2-3: V stage 1)
  h16x8 rbh[2], rbl[2];
  if constexpr (MODE == PS_WIN14) {
    float xs[2][8];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) xs[c][j] = 0.f;
    if (wave_active && !(PS_DBG(a) & 16)) {
      float* const P0 = (float*)(ps_smem + (wave < 2 ? G::K_RING + 2 * G::KSTAGE : G::V_RING + G::VSTAGE)) + (wave & 1) * 32 * 32;
      const int qq = qvalid[0] ? qi[0] : 0;
      const int qy = qq / 14, qx = qq - qy * 14;
#pragma unroll
      for (int axis = 0; axis < 2; ++axis) {
        const _Float16* Th = axis ? a.tabw_hi : a.tabh_hi;
        const _Float16* Tl = axis ? a.tabw_lo : a.tabh_lo;
        const long long to = (long long)min(r, 26) * HD + 8 * h;   // rows 27..31 of the product are never gathered
        h16x8 thr[KS], tlr[KS];
#pragma unroll
        for (int sx = 0; sx < KS; ++sx) {
          thr[sx] = *(const h16x8*)(Th + to + 16 * sx);
          tlr[sx] = *(const h16x8*)(Tl + to + 16 * sx);
        }
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int sx = 0; sx < KS; ++sx) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(tlr[sx], qh[0][sx], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(thr[sx], ql[0][sx], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(thr[sx], qh[0][sx], acc, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) P0[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[e];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        const int qc = axis ? qx : qy;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int k = 16 * c + 8 * h + j - 14 * axis;
            if (k >= 0 && k < 14) xs[c][j] = P0[(qc + 13 - k) * 32 + r];
          }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x = xs[c][j] * inv_scale;
        _Float16 hi, lo;
        hgl_split_hi_lo(x, hi, lo);
        rbh[c][j] = hi;
        rbl[c][j] = lo;
      }
  }

  float rh_next = 0.f;
  f32x4 rw_next[4];
  auto rel_prefetch = [&](int kb) {
    if constexpr (MODE == PS_RELT) {
      int ln;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      const unsigned row = (unsigned)item * (unsigned)a.S + (unsigned)min(q0 + (ln & 31), a.S - 1);
      const unsigned relh_off = row * (unsigned)a.kh;
      const unsigned relw_off = row * (unsigned)a.kw + 4u * (unsigned)(ln >> 5);
      kb = min(kb, a.S - 32);
      rh_next = a.rel_h[relh_off + (unsigned)(kb / a.kw)];
      const unsigned o2 = relw_off + (unsigned)(kb % a.kw);
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) rw_next[g4] = *(const f32x4*)(a.rel_w + o2 + 8 * g4);
    }
  };
  if (MODE == PS_RELT && wave_active) rel_prefetch(0);

  // every compiler-counted load is waited for HERE (see attn_ps_kernel): chunk 0 of K and V, chunk 1 of K, the table and Q
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int sx = 0; sx < KS; ++sx) asm volatile("" : "+v"(qh[qt][sx]), "+v"(ql[qt][sx]));
  if constexpr (MODE == PS_RELT) {
    asm volatile("" : "+v"(rh_next), "+v"(rw_next[0]), "+v"(rw_next[1]), "+v"(rw_next[2]), "+v"(rw_next[3]));
  }
  __syncthreads();

  // QK^T of key chunk ci into a fresh accumulator (PS_RELT: started at the prefetched rel-pos terms)
  auto qk = [&](int ci, f32x16& s, const h16x8 (&QH)[KS], const h16x8 (&QL)[KS]) {
    const _Float16* krow = (const _Float16*)(ps_smem + G::K_RING + (ci % 3) * G::KSTAGE) + k_off;
#pragma unroll
    for (int c = 0; c < KS; ++c) {
      const h16x8 kh8 = *(const h16x8*)(krow + 16 * c);
      const h16x8 kl8 = *(const h16x8*)(krow + HD + 16 * c);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl8, QH[c], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, QL[c], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, QH[c], s, 0, 0, 0);
    }
    if constexpr (MODE == PS_WIN14) {
      const unsigned char* eb = ps_smem + G::E_OFF + ci * (PS_CHUNK * 64);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const h16x8 e8 = *(const h16x8*)(eb + e_off[c]);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(e8, rbl[c], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(e8, rbh[c], s, 0, 0, 0);
      }
    }
  };
  auto s_init = [&](f32x16& s) {
    if constexpr (MODE == PS_RELT) {
      const float rhs = rh_next * inv_scale;
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = fmaf(rw_next[e >> 2][e & 3], inv_scale, rhs);
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = 0.f;
    }
  };

  f32x16 s_cur;
  if (wave_active) {
    s_init(s_cur);
    qk(0, s_cur, qh[0], ql[0]);
    if (MODE == PS_RELT && nchunk > 1) rel_prefetch(32);
  }

  for (int ci = 0; ci < nchunk; ++ci) {
    if (ci > 0) {
      // K (ci + 1) and V (ci), requested one iteration ago, have landed (everyone's, behind the barrier); every wave has left
      // iteration ci - 1, whose K (ci - 1 ... read one iteration earlier still) and V (ci - 1) stages the requests below overwrite
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (MODE == PS_RELT) {
        asm volatile("" : "+v"(rh_next), "+v"(rw_next[0]), "+v"(rw_next[1]), "+v"(rw_next[2]), "+v"(rw_next[3]));
      }
      __syncthreads();
    }
    if (!(PS_DBG(a) & 4)) {
    if (ci + 2 < nchunk) issue_k(ci + 2);
    if (ci + 1 < nchunk) issue_v(ci + 1);
    }
    if (!wave_active) continue;   // uniform
    const int kbase = ci * PS_CHUNK;
    const bool has_next = ci + 1 < nchunk;
    // One (query tile, key tile) pair: masks, segment A = its soft-max beside the QK^T chain of the wave's NEXT pair (the other
    // query tile on this chunk, or the first on the next chunk), the rare rescaling of O, segment B = its P V products.
    //   cur      : the tile (a compile-time index: its O, running maximum and sum are registers)
    //   nxt      : the query tile of the next pair
    //   with_next: there is a next pair; nci its key chunk
    auto tile = [&](auto cur_c, auto nxt_c, bool wn, int nci) {
      constexpr int CUR = decltype(cur_c)::value, NXT = decltype(nxt_c)::value;
      if (kbase + 32 > a.S) {   // uniform: the tile that crosses the end of the sequence
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int kg = kbase + (e & 3) + 8 * (e >> 2) + 4 * h;
          s_cur[e] = kg >= a.S ? NEG_INF : s_cur[e];
        }
      }
      if (MODE == PS_PLAIN && CUR == 0 && keep_row && bx == 0 && wave == 0) {   // uniform: the tile that owns query 0
        const int kk = kbase + (lane & 31);
        const unsigned kb = kk >= 1 && kk < a.S ? keepL[kk - 1] : 1u;
        const unsigned bits = (unsigned)__builtin_amdgcn_ballot_w64(kb != 0) >> (4 * h);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const bool kept = (bits >> ((e & 3) + 8 * (e >> 2))) & 1u;
          s_cur[e] = (qi[0] == 0 && !kept) ? NEG_INF : s_cur[e];
        }
      }
      // ---- segment A (one basic block).  The interleave is written out by hand: 19 groups of soft-max work (U: the max chain,
      // the rescaling decision; E_k: two exponentials; F_k / G_k: the hi / lo split of a pair), each behind its share of the
      // QK^T chain's matrix instructions, with __builtin_amdgcn_sched_barrier(0) between the groups -- sched_group_barrier
      // pipelines are not honoured by this compiler for this block (it emits the dependent MFMAs first and the soft-max behind
      // them, round 2 found the same), and an in-order wave overlaps the two pipes only where they alternate in its stream.
      f32x16 s_next;
      h16x8 ph[2], pl[2];
      float alpha = 1.0f;
      bool need;
      auto seg_a = [&](auto with_next) {
        constexpr bool WN = decltype(with_next)::value;
        constexpr int NM = WN ? 3 * KS + (MODE == PS_WIN14 ? 4 : 0) : 0;
        constexpr int NG = 19;
        const _Float16* krow = (const _Float16*)(ps_smem + G::K_RING + (nci % 3) * G::KSTAGE) + k_off;
        const unsigned char* eb = ps_smem + G::E_OFF + nci * (PS_CHUNK * 64);
        h16x8 fa, fb, fa2, fb2;       // fragments of the current k-step (hi, lo) and of the next
        if constexpr (WN) {
          s_init(s_next);
          fa = *(const h16x8*)(krow);
          fb = *(const h16x8*)(krow + HD);
        }
        // matrix instruction i of the chain: k-step c = i / 3 (lo*hi, hi*lo, hi*hi), then the two indicator steps (lo, hi of R)
        auto mfma_i = [&](int i) {
          if constexpr (WN) {
            if (i < 3 * KS) {
              const int c = i / 3, term = i - 3 * c;
              if (term == 0) {
                // the next step's fragments travel under this step's three products
                if (c + 1 < KS) {
                  fa2 = *(const h16x8*)(krow + 16 * (c + 1));
                  fb2 = *(const h16x8*)(krow + HD + 16 * (c + 1));
                } else if (MODE == PS_WIN14) {
                  fa2 = *(const h16x8*)(eb + e_off[0]);
                  fb2 = *(const h16x8*)(eb + e_off[1]);
                }
                s_next = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, qh[NXT][c], s_next, 0, 0, 0);
              } else if (term == 1) {
                s_next = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, ql[NXT][c], s_next, 0, 0, 0);
              } else {
                s_next = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, qh[NXT][c], s_next, 0, 0, 0);
                fa = fa2;
                fb = fb2;
              }
            } else if (MODE == PS_WIN14) {
              const int j = i - 3 * KS;      // 0, 1: E step 0 with R lo, hi; 2, 3: E step 1
              const h16x8 e8 = j < 2 ? fa : fb;
              s_next = __builtin_amdgcn_mfma_f32_32x32x16_f16(e8, (j & 1) ? rbh[j >> 1] : rbl[j >> 1], s_next, 0, 0, 0);
            }
          }
        };
        float mxa = NEG_INF, mxb = NEG_INF, m_use0 = 0.f, mneg = 0.f, rs = 0.f;
        float pe[16];
        h16x2 hi2[8];
        auto valu_g = [&](int g) {
          if (g == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) mxa = fmaxf(mxa, s_cur[e]);
          } else if (g == 1) {
#pragma unroll
            for (int e = 8; e < 16; ++e) mxb = fmaxf(mxb, s_cur[e]);
            mxa = ps_max_halves(fmaxf(mxa, mxb));
          } else if (g == 2) {
            const float m_cand = fmaxf(m_run[CUR], mxa);
            need = __builtin_amdgcn_ballot_w64(m_cand > m_run[CUR] + rescale_thr) != 0;
            const float m_new = need ? m_cand : m_run[CUR];
            m_use0 = (m_new == NEG_INF) ? 0.f : m_new;
            alpha = __builtin_amdgcn_exp2f((m_run[CUR] - m_use0) * sl2e);   // used only when `need`
            m_run[CUR] = m_new;
            mneg = -m_use0 * sl2e;
            l_run[CUR] = need ? l_run[CUR] * alpha : l_run[CUR];
          } else {
            // pair k of the scores: E_k in group ge[k], F_k in gf[k], G_k in gg[k]
            constexpr int ge[8] = {3, 4, 6, 8, 10, 12, 14, 16};
            constexpr int gf[8] = {4, 6, 8, 10, 12, 14, 16, 17};
            constexpr int gg[8] = {5, 7, 9, 11, 13, 15, 17, 18};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              if (g == ge[k]) {
                pe[2 * k] = __builtin_amdgcn_exp2f(fmaf(s_cur[2 * k], sl2e, mneg));
                pe[2 * k + 1] = __builtin_amdgcn_exp2f(fmaf(s_cur[2 * k + 1], sl2e, mneg));
              }
              if (g == gf[k]) {
                rs += pe[2 * k];
                rs += pe[2 * k + 1];
                hi2[k] = __builtin_bit_cast(h16x2, __builtin_amdgcn_cvt_pkrtz(pe[2 * k], pe[2 * k + 1]));
              }
              if (g == gg[k]) {
                const int e = 2 * k;
                ph[e >> 3][e & 7] = hi2[k][0]; ph[e >> 3][(e & 7) + 1] = hi2[k][1];
                pl[e >> 3][e & 7] = (_Float16)(pe[e] - (float)hi2[k][0]);
                pl[e >> 3][(e & 7) + 1] = (_Float16)(pe[e + 1] - (float)hi2[k][1]);
              }
            }
          }
        };
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
          for (int i = g * NM / NG; i < (g + 1) * NM / NG; ++i) mfma_i(i);
          valu_g(g);
          __builtin_amdgcn_sched_barrier(0);
        }
        l_run[CUR] += rs;
      };
      if (wn) seg_a(std::true_type());
      else seg_a(std::false_type());
      if (need) {   // uniform, rare: O follows the new maximum before this tile's products are added
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[CUR][d][e] *= alpha;
      }
      if (MODE == PS_RELT && ci + 2 < nchunk) rel_prefetch(kbase + 64);   // the terms of tile ci + 2 travel under P V
      // ---- segment B: O^T += V^T P^T ----
      if (!(PS_DBG(a) & 2)) {
        const _Float16* Vh = (const _Float16*)(ps_smem + G::V_RING + (ci & 1) * G::VSTAGE);
        const _Float16* Vl = (const _Float16*)(ps_smem + G::V_RING + (ci & 1) * G::VSTAGE + G::VL_OFF);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          if (kbase + 16 * s2 >= a.S) break;   // uniform
#pragma unroll
          for (int d = 0; d < DT; ++d) {
            const int off = (16 * s2) * VP + d * 32 + tr_off;
            const h16x4 vh0 = ps_tr4(Vh + off), vh1 = ps_tr4(Vh + off + 8 * VP);
            const h16x4 vl0 = ps_tr4(Vl + off), vl1 = ps_tr4(Vl + off + 8 * VP);
            h16x8 vh8, vl8;
#pragma unroll
            for (int e = 0; e < 4; ++e) { vh8[e] = vh0[e]; vh8[4 + e] = vh1[e]; vl8[e] = vl0[e]; vl8[4 + e] = vl1[e]; }
            o[CUR][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl8, ph[s2], o[CUR][d], 0, 0, 0);
            o[CUR][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, pl[s2], o[CUR][d], 0, 0, 0);
            o[CUR][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, ph[s2], o[CUR][d], 0, 0, 0);
          }
        }
      }
      if (wn) s_cur = s_next;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, QT - 1>;
    const bool chain = !(PS_DBG(a) & 1);
    if constexpr (QT == 2) {
      // tile 0 beside the chain of tile 1 on THIS chunk; tile 1 beside the chain of tile 0 on the next.  (A wave whose second
      // tile lies beyond the sequence -- the last wave of a 197-token item -- computes it on zero queries and stores nothing:
      // its workgroup waits for the other waves' two tiles anyway.)
      tile(I0(), I1(), true, ci);
      tile(I1(), I0(), has_next, ci + 1);
    } else {
      tile(I0(), I0(), has_next && chain, ci + 1);
    }
  }

#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
  const float l_tot = ps_add_halves(l_run[qt]);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (tile_active[qt] && !(PS_DBG(a) & 8)) {
    // Write-out.  A lane holds, per accumulator group g, four consecutive d of ITS query row (8 bytes as fp16); the two lanes
    // of a row hold alternating groups.  Stored as they lie that is 20 instructions of 8 bytes per lane, each to 32 different
    // rows -- and a store instruction whose adjacent lanes are not contiguous costs ~280 cycles of issue (knock-out timing:
    // the stores were 148 of the windowed launch's 768 us).  One v_permlane32_swap per dword pairs the groups (g, g + 1) of
    // the two lanes into 16 contiguous bytes per lane (cdna_hip_programming.md T21): half the instructions, 32 bytes per row.
    const long long orow = b * a.sob + (long long)(qvalid[qt] ? qi[qt] : 0) * a.ldo + hh * HD;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        if (d * 32 + 8 * g >= HD) continue;      // HD % 16 == 0: a pair of groups is inside the head or beyond it
        f32x4 w0, w1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { w0[e] = o[qt][d][4 * g + e] * inv; w1[e] = o[qt][d][4 * (g + 1) + e] * inv; }
        if (a.out) {
          // fp32 rows: 16 bytes per lane and group as they lie
          if (qvalid[qt]) {
            *(f32x4*)(a.out + orow + d * 32 + 8 * g + 4 * h) = w0;
            *(f32x4*)(a.out + orow + d * 32 + 8 * (g + 1) + 4 * h) = w1;
          }
        } else {
          h16x4 hi0, lo0, hi1, lo1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            _Float16 a2, c2;
            hgl_split_hi_lo(w0[e], a2, c2);
            hi0[e] = a2; lo0[e] = c2;
            hgl_split_hi_lo(w1[e], a2, c2);
            hi1[e] = a2; lo1[e] = c2;
          }
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
          auto pair16 = [&](const h16x4 ga, const h16x4 gb) {
            u32x2 x = __builtin_bit_cast(u32x2, ga), y = __builtin_bit_cast(u32x2, gb);
            const auto s0 = __builtin_amdgcn_permlane32_swap(x[0], y[0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(x[1], y[1], false, false);
            return u32x4s{(unsigned)s0[0], (unsigned)s1[0], (unsigned)s0[1], (unsigned)s1[1]};
          };
          const u32x4s vh = pair16(hi0, hi1), vl = pair16(lo0, lo1);
          if (qvalid[qt]) {
            const long long off = orow + d * 32 + 8 * g + 8 * h;     // lower lanes: d 8g .. 8g+7 of the row, upper lanes: the next 8
            *(u32x4s*)(a.out_hi + off) = vh;
            *(u32x4s*)(a.out_lo + off) = vl;
          }
        }
      }
  }
  }   // query tiles
  }   // units
}

// the indicator table of the 14 x 14 window (E[key][j] = 1 at j = key / 14 and j = 14 + key % 14, 32 fp16 columns, 224 rows),
// chunk-swizzled as the kernel reads it; built once per device
const unsigned char* psp_etab() {
  static const unsigned char* tab[64] = {nullptr};
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!tab[dev]) {
    static unsigned short host[PS_EROWS * 32];
    for (int row = 0; row < PS_EROWS; ++row)
      for (int j = 0; j < 32; ++j) {
        const int ih = row / 14, iw = 14 + row % 14;
        const int chunk = j >> 3, phys = chunk ^ ((row >> 2) & 3);
        host[row * 32 + phys * 8 + (j & 7)] = (j == ih || j == iw) ? 0x3c00 : 0;
      }
    void* d = nullptr;
    if (hipMalloc(&d, sizeof(host)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, host, sizeof(host), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    tab[dev] = (const unsigned char*)d;
  }
  return tab[dev];
}

template <int HD, int MODE, int QT = 1>
int psp_launch(const PsArgs& a, hipStream_t st) {
  constexpr size_t lds = psp_lds_bytes<HD, MODE>();
  HGL_RESERVE_LDS((attn_psp_kernel<HD, MODE, QT>), lds, "attention_ps");
  const unsigned char* et = nullptr;
  if (MODE == PS_WIN14) {
    et = psp_etab();
    if (!et) { hgl_set_error("attention_ps: cannot allocate the indicator table"); return HGL_ELAUNCH; }
  }
  // persistent: two workgroups per CU (a multiple of 8, so that a workgroup's units stay on its XCD)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  static int ncu[64] = {0};
  if (ncu[dev] == 0) {
    hipDeviceProp_t prop;
    ncu[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  static const int persist = HGL_DIAG_SWITCH("HGL_ATTN_PS_PERSIST", 2);
  long long wgs = (long long)a.B * a.H * a.nqb;
  if (persist > 0) {
    const long long cap = ((long long)persist * ncu[dev]) & ~7ll;
    if (cap >= 8 && wgs > cap) wgs = cap;
  }
  hipLaunchKernelGGL((attn_psp_kernel<HD, MODE, QT>), dim3((unsigned)wgs), dim3(256), lds, st, a, et);
  return hgl_check_launch("attention_ps");
}

template <int HD, int MODE, int QT>
int ps_launch(const PsArgs& a, hipStream_t st) {
  constexpr size_t lds = ps_lds_bytes<HD, MODE>();
  HGL_RESERVE_LDS((attn_ps_kernel<HD, MODE, QT>), lds, "attention_ps");
  const long long wgs = (long long)a.B * a.H * a.nqb;
  hipLaunchKernelGGL((attn_ps_kernel<HD, MODE, QT>), dim3((unsigned)wgs), dim3(256), lds, st, a);
  return hgl_check_launch("attention_ps");
}

}  // namespace

static int g_attn_ps = -1;   // -1: default (on) at first use
static int attn_ps_flag() {
  if (g_attn_ps < 0) g_attn_ps = HGL_DIAG_SWITCH("HGL_ATTN_PS", 1) != 0;   // product: hgl_attention_presplit() is the switch
  return g_attn_ps;
}
#ifdef HGL_PS_STAMPS
extern "C" int hgl_debug_ps_stamps(unsigned long long* out, int n, int block, int wave) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ps_stamps), sizeof(unsigned long long) * (n < 1024 ? n : 1024));
  unsigned long long z[1024] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ps_stamps), z, sizeof(z));
  int sel[2] = {block, wave};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ps_stamp_sel), sel, sizeof(sel));
  return 0;
}
#endif

bool hgl_attention_ps_enabled() { return attn_ps_flag() != 0 && hgl_precision() == HGL_PREC_F16X3; }

extern "C" int hgl_attention_presplit(int on) {
  const int prev = attn_ps_flag();
  if (on >= 0) g_attn_ps = on != 0;
  return prev;
}

// THE predicate of this file: does hgl_launch_attention_ps serve this call?  The callers that must decide BEFORE the
// in-projection whether it writes split planes (sam_api.hip, clip_api.hip: afterwards there is no fp32 qkv to fall back to) ask
// exactly what the launch asks.  plane_delta = byte distance from the hi to the lo plane; rel_kh / rel_kw != 0: rel-pos terms
// given as tensors; tab_h / tab_w: the windowed blocks' tables.  Returns the kernel kind (> 0) or 0.
enum { PS_K_NONE = 0, PS_K_WIN, PS_K_RELT80, PS_K_RELT64, PS_K_CLIP, PS_K_PLAIN80, PS_K_PLAIN64 };
int hgl_attention_ps_serves(long long plane_delta, int ld, int B, int H, int S, int hd, int mask_kind, int rel_kh, int rel_kw,
                            const float* tab_h, const float* tab_w) {
  if (!hgl_attention_ps_enabled()) return PS_K_NONE;
  if (!(hd == 80 || hd == 64) || mask_kind == HGL_MASK_CAUSAL || B <= 0 || H <= 0 || S <= 0) return PS_K_NONE;
  // the DMA addresses one item's K / V rows with a 32-bit offset from the item's base in the hi plane, lo plane included
  const long long span = (long long)S * ld * 2 + (long long)ld * 2;
  if (plane_delta < 0 || plane_delta + span >= (1ll << 32)) return PS_K_NONE;
  // the 197-token CLIP sequences: 1 = one workgroup per item with two query tiles per wave (K / V staged once),
  // 2 = the pipelined persistent kernel with two workgroups per item.  (The pipelined kernel with two query tiles per wave
  // -- attn_psp_kernel<64, PS_PLAIN, 2>, which the template still admits -- needs 256 VGPRs + 19 spilled dwords whose scratch
  // reloads wait on vmcnt in the middle of the DMA stream: 855 us against 742 for (1) on 1024 x 12 x 197 x 64; not instantiated.)
  static const int clip_kernel = HGL_DIAG_SWITCH("HGL_ATTN_PS_CLIP", 1);
  if (tab_h || tab_w) {
    const void *hh = nullptr, *hl = nullptr;
    int sh = 1, n1 = 0, k1 = 0;
    if (tab_h && tab_w && hd == 80 && S == 196 && mask_kind == HGL_MASK_NONE && !rel_kh &&
        hgl_get_split_weight(tab_h, &hh, &hl, &sh, &n1, &k1) && sh == 0 && n1 == 27 && k1 == 80 &&
        hgl_get_split_weight(tab_w, &hh, &hl, &sh, &n1, &k1) && sh == 0 && n1 == 27 && k1 == 80)
      return PS_K_WIN;
    return PS_K_NONE;
  }
  if (rel_kh || rel_kw) {
    // 32-bit element offsets into the rel-pos tensors
    if (mask_kind == HGL_MASK_NONE && (rel_kw & 31) == 0 && (S & 31) == 0 && rel_kh * rel_kw == S &&
        (long long)B * H * S * (long long)(rel_kh > rel_kw ? rel_kh : rel_kw) < (1ll << 32))
      return hd == 80 ? PS_K_RELT80 : PS_K_RELT64;
    return PS_K_NONE;
  }
  if (hd == 64 && S > 128 && S <= 256 && clip_kernel == 1 && (mask_kind != HGL_MASK_CLS_KEEP || S <= 257)) return PS_K_CLIP;
  if (mask_kind == HGL_MASK_NONE || (mask_kind == HGL_MASK_CLS_KEEP && S <= 257)) return hd == 80 ? PS_K_PLAIN80 : PS_K_PLAIN64;
  return PS_K_NONE;
}

// Attention on the split qkv planes.  Returns 1 ("not applicable": the caller takes the fp32-input path) when the shape is
// not one this kernel serves, 0 on success, < 0 on error.
//   qkv_hi / qkv_lo : fp16 planes, row (b * sb + s), columns qcol / kcol / vcol + head * hd, leading dimension ld (halfs)
//   tab_h / tab_w   : SAM's windowed blocks (S == 196, hd == 80): the rel-pos tables (fp32 pointers registered with
//                     hgl_register_split_weight at scale 2^0); rel_h / rel_w: the terms as tensors (global blocks)
int hgl_launch_attention_ps(const void* qkv_hi, const void* qkv_lo, int ld, int qcol, int kcol, int vcol, long long sb, int B,
                            int H, int S, int hd, float* out, void* out_hi, void* out_lo, int ldo, long long sob, float scale,
                            int mask_kind, const uint8_t* keep, int keep_b0, int keep_n, const float* rel_h, const float* rel_w,
                            int kh, int kw, const float* tab_h, const float* tab_w, hipStream_t st) {
  const long long delta = (const char*)qkv_lo - (const char*)qkv_hi;
  const int kind = hgl_attention_ps_serves(delta, ld, B, H, S, hd, mask_kind, rel_h ? kh : 0, rel_h ? kw : 0, tab_h, tab_w);
  if (kind == PS_K_NONE) return 1;
  HGL_REQUIRE(qkv_hi && qkv_lo && (out || (out_hi && out_lo)) && B > 0 && H > 0 && S > 0, "attention_ps: bad arguments");
  HGL_REQUIRE((ld & 7) == 0 && (qcol & 7) == 0 && (kcol & 7) == 0 && (vcol & 7) == 0 && (ldo & 3) == 0 && (sob & 3) == 0 &&
                  (((uintptr_t)qkv_hi | (uintptr_t)qkv_lo | (uintptr_t)out | (uintptr_t)out_hi | (uintptr_t)out_lo) & 15) == 0,
              "attention_ps: operands must be 16-byte aligned");
  HGL_REQUIRE(mask_kind != HGL_MASK_CLS_KEEP || keep, "attention_ps: HGL_MASK_CLS_KEEP needs keep bytes");
  HGL_REQUIRE(vcol >= kcol, "attention_ps: the V columns must not lie before the K columns (32-bit offsets from the K base)");
  HGL_REQUIRE((rel_h == nullptr) == (rel_w == nullptr) && (tab_h == nullptr) == (tab_w == nullptr), "attention_ps: rel-pos operands go in pairs");
  PsArgs a;
  a.hi = (const _Float16*)qkv_hi; a.lo = (const _Float16*)qkv_lo;
  a.ld = ld; a.qcol = qcol; a.kcol = kcol; a.vcol = vcol; a.sb = sb;
  a.B = B; a.H = H; a.S = S;
  a.out = out; a.out_hi = (_Float16*)out_hi; a.out_lo = (_Float16*)out_lo; a.ldo = ldo; a.sob = sob;
  a.scale = scale; a.mask_kind = mask_kind; a.keep = keep; a.keep_b0 = keep_b0; a.keep_n = keep_n > 0 ? keep_n : B;
  a.rel_h = rel_h; a.rel_w = rel_w; a.kh = kh; a.kw = kw;
  a.tabh_hi = a.tabh_lo = a.tabw_hi = a.tabw_lo = nullptr;
  a.nqb = 1;
  a.dbg = HGL_DIAG_SWITCH("HGL_ATTN_PS_DBG", 0);
  if (kind == PS_K_WIN) {
    const void *hh = nullptr, *hl = nullptr, *wh = nullptr, *wl = nullptr;
    int sh = 1, sw = 1, n1 = 0, k1 = 0, n2 = 0, k2 = 0;
    (void)hgl_get_split_weight(tab_h, &hh, &hl, &sh, &n1, &k1);
    (void)hgl_get_split_weight(tab_w, &wh, &wl, &sw, &n2, &k2);
    a.tabh_hi = (const _Float16*)hh; a.tabh_lo = (const _Float16*)hl;
    a.tabw_hi = (const _Float16*)wh; a.tabw_lo = (const _Float16*)wl;
    a.nqb = 2;
  } else if (kind != PS_K_CLIP) {
    a.nqb = (S + 127) / 128;   // blocks of 128 queries (the CLS keep row of an item sits in 256 bytes of LDS)
  }
  HglProfScope prof(HGL_PROF_ATTN, 4.0 * B * H * (double)S * S * hd, 0.0, st);
  switch (kind) {
    case PS_K_WIN: return psp_launch<80, PS_WIN14>(a, st);
    case PS_K_RELT80: return psp_launch<80, PS_RELT>(a, st);
    case PS_K_RELT64: return psp_launch<64, PS_RELT>(a, st);
    case PS_K_PLAIN80: return psp_launch<80, PS_PLAIN>(a, st);
    case PS_K_PLAIN64: return psp_launch<64, PS_PLAIN>(a, st);
    default: return ps_launch<64, PS_PLAIN, 2>(a, st);     // PS_K_CLIP: the un-pipelined kernel, two query tiles per wave
  }
}

// The same kernels behind the C ABI for tests and micro-benchmarks: qkv [B*S, ld] fp32 with q | k | v at columns 0, H*hd, 2*H*hd is
// split into fp16 hi / lo planes in `scratch` (>= B*S*ld*4 bytes) the way the in-projection's write-out splits it, then
// multiplied.  HGL_EINVAL when the shape is not one the pre-split kernels serve.
extern "C" int hgl_attention_presplit_f32(const float* qkv, int ld, int B, int H, int S, int hd, float* out, int ldo, float scale,
                                          int mask_kind, const uint8_t* keep, int keep_b0, int keep_n, void* scratch,
                                          size_t scratch_bytes, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(qkv && out && scratch && B > 0 && H > 0 && S > 0 && ld >= 3 * H * hd && (ld & 7) == 0, "attention_presplit: bad arguments");
  const size_t n = (size_t)B * S * ld;
  HGL_REQUIRE(scratch_bytes >= n * 4, "attention_presplit: scratch too small (%zu bytes for %zu)", scratch_bytes, n * 4);
  HGL_REQUIRE(hgl_precision() == HGL_PREC_F16X3, "attention_presplit: split-fp16 mode only");
  hipStream_t st = (hipStream_t)stream;
  uint16_t* hi = (uint16_t*)scratch;
  uint16_t* lo = hi + n;
  HGL_TRY(hgl_launch_split_f16(qkv, 1.0f, hi, lo, (long long)n, st));
  const int prev = hgl_attention_presplit(1);
  const int rc = hgl_launch_attention_ps(hi, lo, ld, 0, H * hd, 2 * H * hd, S, B, H, S, hd, out, nullptr, nullptr, ldo, (long long)S * ldo,
                                         scale, mask_kind, keep, keep_b0, keep_n, nullptr, nullptr, 0, 0, nullptr, nullptr, st);
  hgl_attention_presplit(prev);
  HGL_REQUIRE(rc <= 0, "attention_presplit: shape not served (B %d, H %d, S %d, hd %d, mask %d)", B, H, S, hd, mask_kind);
  return rc;
}
