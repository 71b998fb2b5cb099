// Host orchestration of the SAM image encoder and prompt/mask decoder behind the C ABI.
//   hgl_sam_encode         == Sam.preprocess + ImageEncoderViT.forward
//                             (modeling/sam.py:164-174, modeling/image_encoder.py:106-116)
//   hgl_sam_decode_points  == PromptEncoder(points) + MaskDecoder(multimask_output=True)
//                             (predictor.py:222-235, modeling/mask_decoder.py:71-149)
// Every contraction is the fp32 MFMA GEMM / fused attention of gemm.hip / attention.hip.
// Tokens stay NHWC ([g*g, C] rows) end to end; the reference's NCHW permutes disappear.
#include "hgl_common.h"
#include <stdlib.h>
#include <math.h>

namespace {

struct EncPlan {
  float *img, *cols, *X, *H, *Hw, *QKV, *O, *P, *F, *Th, *Tw, *relh, *relw, *neckA, *neckB, *cols3;
  int *pad_of, *tok_of, *pad_list, *pad_count;   // row maps of the padded window partition (win_maps_kernel)
  int n_pad_max;
  int nb;                                         // images stacked along the token rows
};

bool carve_enc(HglArena& ar, const HglSamEncoderW* w, int nb, EncPlan& p) {
  const int D = w->embed_dim, S = w->img_size, g = S / w->patch, C = w->out_chans;
  const size_t T = (size_t)nb * g * g;
  // windowed blocks pad the grid up to a multiple of the window size
  size_t Tw_max = T;
  int rl_max = 0;
  for (int i = 0; i < w->depth; ++i) {
    const int ws = w->blocks[i].window;
    if (ws > 0) {
      const size_t nw = (g + ws - 1) / ws;
      Tw_max = Tw_max > nb * nw * nw * ws * ws ? Tw_max : nb * nw * nw * ws * ws;
    }
    rl_max = rl_max > w->blocks[i].rel_len ? rl_max : w->blocks[i].rel_len;
  }
  p.nb = nb;
  p.img = ar.take<float>((size_t)nb * 3 * S * S);
  p.cols = ar.take<float>(T * 3 * w->patch * w->patch);
  p.X = ar.take<float>(T * D);
  p.H = ar.take<float>(T * D);
  p.Hw = ar.take<float>(Tw_max * D);
  p.QKV = ar.take<float>(Tw_max * 3 * D);
  p.O = ar.take<float>(Tw_max * D);
  p.P = ar.take<float>(Tw_max * D);
  p.F = ar.take<float>(T * 4 * D);
  p.Th = ar.take<float>((size_t)w->heads * Tw_max * rl_max);
  p.Tw = ar.take<float>((size_t)w->heads * Tw_max * rl_max);
  p.relh = ar.take<float>((size_t)w->heads * Tw_max * g);
  p.relw = ar.take<float>((size_t)w->heads * Tw_max * g);
  p.neckA = ar.take<float>(T * C);
  p.neckB = ar.take<float>(T * C);
  p.cols3 = ar.take<float>(T * C * 9);
  p.pad_of = ar.take<int>(T);
  p.tok_of = ar.take<int>(T);
  p.n_pad_max = (int)(Tw_max - T);
  p.pad_list = ar.take<int>(Tw_max - T + 1);
  p.pad_count = ar.take<int>(1);
  return ar.ok();
}

bool valid_enc(const HglSamEncoderW* w) {
  return w && w->embed_dim > 0 && w->depth > 0 && w->heads > 0 && w->img_size > 0 && w->patch > 0 &&
         w->out_chans > 0 && w->img_size % w->patch == 0 && w->embed_dim % w->heads == 0 &&
         (w->embed_dim & 3) == 0 && w->patch_w && w->patch_b && w->pos_embed && w->blocks && w->neck0_w &&
         w->neck1_w && w->neck1_b && w->neck2_w && w->neck3_w && w->neck3_b;
}

// Block.forward (modeling/image_encoder.py:166-182)
int enc_block(const HglSamEncoderW* w, const HglSamBlockW& b, const EncPlan& p, hipStream_t st) {
  const int D = w->embed_dim, g = w->img_size / w->patch, heads = w->heads, hd = D / heads;
  const int T1 = g * g;                   // tokens of one image
  const int T = p.nb * T1;                // token rows of the batch (images stacked along the rows)
  const int ws = b.window;
  const int size = ws > 0 ? ws : g;       // attention grid side
  const int nw = ws > 0 ? (g + ws - 1) / ws : 1;
  const int B1 = nw * nw;                 // windows of one image (1 for global attention)
  const int B = p.nb * B1;
  const int S = size * size;              // tokens per window
  const int M1 = B1 * S;                  // padded rows of one image
  const int M = B * S;
  const int L = b.rel_len;
  HGL_REQUIRE(L == 2 * size - 1, "sam_encode: rel_pos length %d does not match attention size %d", L, size);

  const bool x3 = hgl_use_x3(b.qkv_w, D) && hgl_use_x3(b.proj_w, D) && hgl_use_x3(b.lin1_w, D) &&
                  hgl_use_x3(b.lin2_w, 4 * D) && (D % 256) == 0;
  static int padskip = -1;   // HGL_SAM_PADSKIP=0: run the windowed GEMMs over the padded rows as well (A/B timing)
  if (padskip < 0) padskip = HGL_DIAG_SWITCH("HGL_SAM_PADSKIP", 1) ? 1 : 0;
  uint16_t* Ah = (uint16_t*)p.Hw;                   // split GEMM input (aliases the window buffer)
  uint16_t* Al = Ah + (size_t)M * D;
  uint16_t* Hh = (uint16_t*)p.H;
  uint16_t* Hl = Hh + (size_t)T * D;
  uint16_t* Fh = (uint16_t*)p.F;
  uint16_t* Fl = Fh + (size_t)T * 4 * D;
  uint16_t* Qh = (uint16_t*)p.QKV;                  // split qkv (aliases the fp32 tensor)
  uint16_t* Ql = Qh + (size_t)M * 3 * D;
  // global blocks (the whole 64 x 64 grid, rel-pos terms as tensors): the same split planes, the terms from the split q
  static int ps_glob_on = -1;     // HGL_ATTN_PS_GLOBAL=0: the global blocks keep the fp32-input kernels (A/B timing)
  if (ps_glob_on < 0) ps_glob_on = HGL_DIAG_SWITCH("HGL_ATTN_PS_GLOBAL", 1) ? 1 : 0;
  // both decisions are hgl_attention_ps_serves' (the predicate the launch itself applies: plane distance + one item's rows
  // within 32 bits, shapes, registered tables), taken here because the in-projection below writes split planes only
  const long long plane_delta = (long long)M * 3 * D * 2;
  const bool ps_glob = x3 && ws == 0 && size == 64 && ps_glob_on &&
                       hgl_attention_ps_serves(plane_delta, 3 * D, B, heads, S, hd, HGL_MASK_NONE, size, size, nullptr, nullptr) != 0;
  const bool ps_win = x3 && ws == 14 &&
                      hgl_attention_ps_serves(plane_delta, 3 * D, B, heads, S, hd, HGL_MASK_NONE, 0, 0, b.rel_pos_h, b.rel_pos_w) != 0;
  if (x3) {
    if (ws > 0 && M > T && padskip) {
      // norm1 of the real tokens written straight to their rows of the padded window layout (the pad rows are never
      // read: the GEMMs below gather the real tokens only)
      HGL_TRY(hgl_launch_layernorm_split_maps(p.X, b.norm1_w, b.norm1_b, Ah, Al, T, D, 1e-6f, p.tok_of, p.pad_of, st));
    } else if (ws > 0) {
      HGL_TRY(hgl_launch_layernorm(p.X, b.norm1_w, b.norm1_b, p.H, T, D, 1e-6f, st));
      for (int i = 0; i < p.nb; ++i)
        HGL_TRY(hgl_launch_win_partition_split(p.H + (size_t)i * T1 * D, g, ws, nw, D, Ah + (size_t)i * M1 * D,
                                               Al + (size_t)i * M1 * D, st));
    } else {
      HGL_TRY(hgl_launch_layernorm_split(p.X, b.norm1_w, b.norm1_b, Ah, Al, T, D, 1e-6f, st));
    }
    // q | k | v as fp16 hi / lo planes (the in-projection's write-out splits; same bytes as the fp32 tensor) for the attention
    // kernel that stages them by LDS-DMA without converting (attention_ps.hip): the 14 x 14 windows at head dim 80
    if (ps_glob) {
      HGL_TRY(hgl_launch_gemm_f16x3(Ah, Al, D, b.qkv_w, b.qkv_b, nullptr, 0, nullptr, Qh, Ql, 3 * D, M, 3 * D, D, HGL_ACT_NONE, st));
    } else if (ps_win) {
      if (M > T && padskip) {
        HGL_TRY(hgl_launch_fill_rows_split(Qh, Ql, 3 * D, p.pad_list, p.pad_count, p.n_pad_max, b.qkv_b, 3 * D, st));
        HGL_TRY(hgl_launch_gemm_f16x3_maps(Ah, Al, D, p.pad_of, b.qkv_w, b.qkv_b, nullptr, 0, 0, p.pad_of, nullptr, Qh, Ql, 3 * D,
                                           T, 3 * D, D, HGL_ACT_NONE, st));
      } else {
        HGL_TRY(hgl_launch_gemm_f16x3(Ah, Al, D, b.qkv_w, b.qkv_b, nullptr, 0, nullptr, Qh, Ql, 3 * D, M, 3 * D, D, HGL_ACT_NONE, st));
      }
    } else if (ws > 0 && M > T && padskip) {
      // only the real tokens go through the GEMM (16 % fewer rows at 64x64 / 14x14); a padded row of qkv is the bias
      HGL_TRY(hgl_launch_fill_rows(p.QKV, 3 * D, p.pad_list, p.pad_count, p.n_pad_max, b.qkv_b, 3 * D, st));
      HGL_TRY(hgl_launch_gemm_f16x3_maps(Ah, Al, D, p.pad_of, b.qkv_w, b.qkv_b, nullptr, 0, 0, p.pad_of, p.QKV, nullptr,
                                         nullptr, 3 * D, T, 3 * D, D, HGL_ACT_NONE, st));
    } else {
      HGL_TRY(hgl_launch_gemm_f16x3(Ah, Al, D, b.qkv_w, b.qkv_b, nullptr, 0, p.QKV, nullptr, nullptr, 3 * D, M, 3 * D, D,
                                    HGL_ACT_NONE, st));
    }
  } else {
    HGL_TRY(hgl_launch_layernorm(p.X, b.norm1_w, b.norm1_b, p.H, T, D, 1e-6f, st));
    const float* A = p.H;
    if (ws > 0) {
      for (int i = 0; i < p.nb; ++i)
        HGL_TRY(hgl_launch_win_partition(p.H + (size_t)i * T1 * D, g, ws, nw, D, p.Hw + (size_t)i * M1 * D, st));
      A = p.Hw;
    }
    HGL_TRY(hgl_launch_gemm(A, b.qkv_w, b.qkv_b, nullptr, p.QKV, M, 3 * D, D, D, D, 0, 3 * D, 1, 0, 0, 0, 0,
                            HGL_ACT_NONE, st));
  }
  if (ps_win) {
    const int rc = hgl_launch_attention_ps(Qh, Ql, 3 * D, 0, D, 2 * D, S, B, heads, S, hd, nullptr, Ah, Al, D, (long long)S * D,
                                           1.0f / sqrtf((float)hd), HGL_MASK_NONE, nullptr, 0, 0, nullptr, nullptr, 0, 0,
                                           b.rel_pos_h, b.rel_pos_w, st);
    if (rc < 0) return rc;
    HGL_REQUIRE(rc == 0, "sam_encode: the pre-split attention refused a shape its caller had checked");
    goto attention_done;
  }
  if (ps_glob) {
    HGL_TRY(hgl_launch_relpos_split(Qh, Ql, 3 * D, B, heads, S, size, hd, b.rel_pos_h, b.rel_pos_w, p.relh, p.relw, st));
    const int rc = hgl_launch_attention_ps(Qh, Ql, 3 * D, 0, D, 2 * D, S, B, heads, S, hd, nullptr, Ah, Al, D, (long long)S * D,
                                           1.0f / sqrtf((float)hd), HGL_MASK_NONE, nullptr, 0, 0, p.relh, p.relw, size, size,
                                           nullptr, nullptr, st);
    if (rc < 0) return rc;
    HGL_REQUIRE(rc == 0, "sam_encode: the pre-split attention refused a shape its caller had checked");
    goto attention_done;
  }
  // 14 x 14 windows at head dim 80 in f16x3 mode: the attention kernel computes the decomposed rel-pos terms itself
  if (x3 && ws == 14 && hd == 80) {
    const int rc = hgl_launch_attention_win14(p.QKV, p.QKV + D, p.QKV + 2 * D, Ah, Al, B, heads, hd, 3 * D, 3 * D, 3 * D, D,
                                              (long long)S * 3 * D, (long long)S * 3 * D, (long long)S * 3 * D, (long long)S * D,
                                              1.0f / sqrtf((float)hd), b.rel_pos_h, b.rel_pos_w, st);
    if (rc < 0) return rc;
    if (rc == 0) goto attention_done;
  }
  // decomposed rel-pos tables rel_h/rel_w [B*heads, S, size] from the UNSCALED q (image_encoder.py:351-354)
  if (hd == 80 || hd == 64) {
    HGL_TRY(hgl_launch_relpos_direct(p.QKV, 3 * D, B, heads, S, size, hd, b.rel_pos_h, b.rel_pos_w, p.relh, p.relw, st));
  } else {  // generic head dims: q . rel_pos[r] for every r as a batched GEMM, then gathered per (q,k)
    HGL_TRY(hgl_launch_gemm(p.QKV, b.rel_pos_h, nullptr, nullptr, p.Th, M, L, hd, 3 * D, hd, 0, L, heads, hd, 0,
                            0, (long long)M * L, HGL_ACT_NONE, st));
    HGL_TRY(hgl_launch_gemm(p.QKV, b.rel_pos_w, nullptr, nullptr, p.Tw, M, L, hd, 3 * D, hd, 0, L, heads, hd, 0,
                            0, (long long)M * L, HGL_ACT_NONE, st));
    HGL_TRY(hgl_launch_relpos_gather(p.Th, B, heads, S, size, L, 0, p.relh, st));
    HGL_TRY(hgl_launch_relpos_gather(p.Tw, B, heads, S, size, L, 1, p.relw, st));
  }
  // f16x3: the attention writes its output as the fp16 hi+lo pair the projection reads
  HGL_TRY(hgl_launch_attention_split(p.QKV, p.QKV + D, p.QKV + 2 * D, x3 ? nullptr : p.O, Ah, Al, B, heads, S, S, hd, 3 * D,
                                     3 * D, 3 * D, D, (long long)S * 3 * D, (long long)S * 3 * D, (long long)S * 3 * D,
                                     (long long)S * D, 1.0f / sqrtf((float)hd), HGL_MASK_NONE, nullptr, 0, 0, p.relh,
                                     p.relw, size, size, st));
attention_done:
  if (x3) {
    static int splitk_proj = -1;   // HGL_SAM_SPLITK_PROJ=1 enables split-K for the projection too (measured neutral: K is short)
    if (splitk_proj < 0) splitk_proj = HGL_DIAG_SWITCH("HGL_SAM_SPLITK_PROJ", 0) ? 1 : 0;
    const int ksp = splitk_proj ? hgl_gemm_f16x3_splitk_factor(T, D, D) : 1;
    const size_t qkv_cap = (size_t)M * 3 * D * sizeof(float);     // q, k, v are dead after the attention
    const bool proj_splitk = ksp > 1 && (size_t)ksp * T * D * sizeof(float) <= qkv_cap && (ws == 0 || (M > T && padskip));
    if (proj_splitk) {
      // few 256x256 tiles: split-K over the idle CUs; real tokens only, written to token order with the residual
      HGL_TRY(hgl_launch_gemm_f16x3_splitk(Ah, Al, D, ws > 0 ? p.pad_of : nullptr, b.proj_w, b.proj_b, p.X, D,
                                           ws > 0 ? p.tok_of : nullptr, p.X, D, T, D, D, HGL_ACT_NONE, ksp, p.QKV, qkv_cap, st));
    } else if (ws > 0 && M > T && padskip) {
      // projection of the real tokens only, written straight back to token order with the residual added
      // (window_unpartition + shortcut, image_encoder.py:178-180)
      HGL_TRY(hgl_launch_gemm_f16x3_balanced(Ah, Al, D, p.pad_of, b.proj_w, b.proj_b, p.X, D, p.tok_of, p.X, D, T, D, D, HGL_ACT_NONE,
                                             p.QKV, qkv_cap, st));
    } else if (ws > 0) {
      HGL_TRY(hgl_launch_gemm_f16x3(Ah, Al, D, b.proj_w, b.proj_b, nullptr, 0, p.P, nullptr, nullptr, D, M, D, D,
                                    HGL_ACT_NONE, st));
      for (int i = 0; i < p.nb; ++i)
        HGL_TRY(hgl_launch_win_unpartition_add(p.X + (size_t)i * T1 * D, g, ws, nw, D, p.P + (size_t)i * M1 * D, st));
    } else {
      HGL_TRY(hgl_launch_gemm_f16x3_balanced(Ah, Al, D, nullptr, b.proj_w, b.proj_b, p.X, D, nullptr, p.X, D, T, D, D, HGL_ACT_NONE,
                                             p.QKV, qkv_cap, st));
    }
    HGL_TRY(hgl_launch_layernorm_split(p.X, b.norm2_w, b.norm2_b, Hh, Hl, T, D, 1e-6f, st));
    HGL_TRY(hgl_launch_gemm_f16x3(Hh, Hl, D, b.lin1_w, b.lin1_b, nullptr, 0, nullptr, Fh, Fl, 4 * D, T, 4 * D, D,
                                  HGL_ACT_GELU, st));
    // mlp.lin2: few output tiles, K = 4D -> split-K over the idle CUs; the partial sums borrow the qkv buffer
    static int splitk_on = -1;   // HGL_SAM_SPLITK=0 disables (A/B timing)
    if (splitk_on < 0) splitk_on = HGL_DIAG_SWITCH("HGL_SAM_SPLITK", 1) ? 1 : 0;
    const int ks = splitk_on ? hgl_gemm_f16x3_splitk_factor(T, D, 4 * D) : 1;
    const size_t qkv_bytes = (size_t)M * 3 * D * sizeof(float);
    if (ks > 1 && (size_t)ks * T * D * sizeof(float) <= qkv_bytes) {
      HGL_TRY(hgl_launch_gemm_f16x3_splitk(Fh, Fl, 4 * D, nullptr, b.lin2_w, b.lin2_b, p.X, D, nullptr, p.X, D, T, D, 4 * D,
                                           HGL_ACT_NONE, ks, p.QKV, qkv_bytes, st));
    } else {
      HGL_TRY(hgl_launch_gemm_f16x3_balanced(Fh, Fl, 4 * D, nullptr, b.lin2_w, b.lin2_b, p.X, D, nullptr, p.X, D, T, D, 4 * D, HGL_ACT_NONE,
                                             p.QKV, qkv_bytes, st));
    }
    return HGL_OK;
  }
  if (ws > 0) {
    HGL_TRY(hgl_launch_gemm(p.O, b.proj_w, b.proj_b, nullptr, p.P, M, D, D, D, D, 0, D, 1, 0, 0, 0, 0,
                            HGL_ACT_NONE, st));
    for (int i = 0; i < p.nb; ++i)
      HGL_TRY(hgl_launch_win_unpartition_add(p.X + (size_t)i * T1 * D, g, ws, nw, D, p.P + (size_t)i * M1 * D, st));
  } else {
    HGL_TRY(hgl_launch_gemm(p.O, b.proj_w, b.proj_b, p.X, p.X, T, D, D, D, D, D, D, 1, 0, 0, 0, 0,
                            HGL_ACT_NONE, st));
  }
  HGL_TRY(hgl_launch_layernorm(p.X, b.norm2_w, b.norm2_b, p.H, T, D, 1e-6f, st));
  HGL_TRY(hgl_launch_gemm(p.H, b.lin1_w, b.lin1_b, nullptr, p.F, T, 4 * D, D, D, D, 0, 4 * D, 1, 0, 0, 0, 0,
                          HGL_ACT_GELU, st));
  HGL_TRY(hgl_launch_gemm(p.F, b.lin2_w, b.lin2_b, p.X, p.X, T, D, 4 * D, 4 * D, 4 * D, D, D, 1, 0, 0, 0, 0,
                          HGL_ACT_NONE, st));
  return HGL_OK;
}

// ------------------------------------------------------------------------------ decoder
struct DecPlan {
  float *coords, *sparse, *tokens, *queries, *qpe, *q1, *k1, *v1, *att, *keys0, *kpe0, *keys, *kpe, *kp, *vp,
      *qi, *atti, *mlp, *u1, *u2, *hy_a, *hy_b, *hyper, *iou_a, *iou_b, *keysS;
  uint8_t* skip;      // IoU gate: prompts whose upscaling is skipped
};

bool carve_dec(HglArena& ar, const HglSamDecoderW* w, int P, DecPlan& p) {
  const size_t C = w->C, HW = (size_t)w->grid * w->grid, T = 16;   // 5 output tokens + up to 11 sparse prompt tokens
  p.sparse = ar.take<float>((size_t)P * (T - 5) * C);
  p.tokens = ar.take<float>(P * T * C);
  p.queries = ar.take<float>(P * T * C);
  p.qpe = ar.take<float>(P * T * C);
  p.q1 = ar.take<float>(P * T * C);
  p.k1 = ar.take<float>(P * T * C);
  p.v1 = ar.take<float>(P * T * C);
  p.att = ar.take<float>(P * T * C);
  p.keys0 = ar.take<float>(HW * C);
  p.kpe0 = ar.take<float>(HW * C);
  p.keys = ar.take<float>(P * HW * C);
  p.kpe = ar.take<float>(P * HW * C);
  // k / v / q projections of the image tokens: three [P*HW, C/2] matrices, or -- merged projections -- ONE [P*HW, 3C/2]
  p.kp = ar.take<float>(3 * (P * HW * C / 2));
  p.vp = p.kp ? p.kp + P * HW * C / 2 : nullptr;
  p.qi = p.kp ? p.kp + 2 * (P * HW * C / 2) : nullptr;
  p.atti = ar.take<float>(P * HW * C / 2);
  p.mlp = ar.take<float>(P * T * w->mlp_dim);
  p.u1 = ar.take<float>(P * HW * C);           // [P*HW*4, C/4]
  p.u2 = ar.take<float>(P * HW * 16 * (C / 8)); // [P*HW*16, C/8]
  p.hy_a = ar.take<float>((size_t)P * C);
  p.hy_b = ar.take<float>((size_t)P * C);
  p.hyper = ar.take<float>((size_t)P * 4 * (C / 8));
  p.keysS = ar.take<float>(P * HW * C);        // f16x3 mode: the normalised image tokens as fp16 hi | lo planes
  p.iou_a = ar.take<float>((size_t)P * C);
  p.iou_b = ar.take<float>((size_t)P * C);
  p.skip = ar.take<uint8_t>((size_t)P);
  return ar.ok();
}

bool valid_lin(const HglLinearW& l) { return l.w && l.b; }
bool valid_attn(const HglSamAttnW& a) {
  return valid_lin(a.q) && valid_lin(a.k) && valid_lin(a.v) && valid_lin(a.out) && a.internal > 0;
}
bool valid_dec(const HglSamDecoderW* w) {
  if (!w || w->C <= 0 || w->grid <= 0 || w->heads <= 0 || w->mlp_dim <= 0 || (w->C & 31)) return false;
  if (!w->pe_gauss || !w->point_embed_pos || !w->not_a_point || !w->no_mask || !w->iou_token || !w->mask_tokens) return false;
  for (int i = 0; i < 2; ++i) {
    const auto& l = w->layer[i];
    if (!valid_attn(l.self_attn) || !valid_attn(l.t2i) || !valid_attn(l.i2t) || !valid_lin(l.lin1) || !valid_lin(l.lin2) ||
        !l.n1.w || !l.n2.w || !l.n3.w || !l.n4.w) return false;
  }
  if (!valid_attn(w->final_t2i) || !w->norm_final.w || !w->up0_w || !w->up0_b || !w->up1.w || !w->up3_w || !w->up3_b) return false;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) if (!valid_lin(w->hyper[i][j])) return false;
  for (int j = 0; j < 3; ++j) if (!valid_lin(w->iou_head[j])) return false;
  return true;
}

inline int lin(const float* A, int lda, const HglLinearW& l, const float* R, int ldr, float* Cc, int ldc, int M,
               int N, int K, int act, hipStream_t st) {
  // the token side of a large prompt batch (512 prompts: 3584 rows) and the image-side projections shared by all prompts
  // (4096 rows) are still small-tile work: 8 x 116 workgroups of 32 x 32 instead of the fp32 kernel (35 -> 12 us each,
  // 23 launches per decoder call)
  if (M > 1024 && hgl_gemm_skinny_applicable(l.w, M, N, K, lda, K, 1, 8192))
    return hgl_launch_gemm_x3_skinny(A, lda, l.w, l.b, R, ldr, Cc, ldc, M, N, K, act, st);
  return hgl_launch_gemm(A, l.w, l.b, R, Cc, M, N, K, lda, K, ldr, ldc, 1, 0, 0, 0, 0, act, st);
}

// which fused decoder stages are in use (bit 0: upscaling + hyper-network products, bit 1: merged image-side projections,
// bit 2: image -> token attention + out-projection + norm4, bit 3: unused, bit 4: chunked token -> image attention);
// default: all stages fused
int g_dec_fusion = -1;
int dec_fusion_mask() {
  if (g_dec_fusion < 0) g_dec_fusion = HGL_DIAG_SWITCH("HGL_SAM_DEC_FUSED", 0x7fffffff);   // product: hgl_sam_decoder_fusion()
  return g_dec_fusion;
}

// token -> image attention (7 queries, thousands of keys): the chunked kernel when the shape fits (fusion bit 4), with the
// partials in `scratch` (the image -> token buffer, idle at that point)
int dec_fewq(const float* q, const float* k, const float* v, float* att, int B, int heads, int Nq, int Nk, int hd, int ldq,
             int ldk, int ldv, int ldo, long long sqb, long long skb, long long svb, long long sob, float* scratch,
             size_t scratch_bytes, hipStream_t st) {
  const float scale = 1.0f / sqrtf((float)hd);
  if ((dec_fusion_mask() & 16) && scratch && heads == 8 && hd == 16 && Nk >= 256 && B <= 65535 &&
      scratch_bytes >= hgl_attention_fewq_part_bytes(B, Nk)) {
    // the chunked kernel holds up to 7 queries: longer prompts (more sparse tokens) go through it 7 queries at a time
    for (int q0 = 0; q0 < Nq; q0 += 7)
      HGL_TRY(hgl_launch_attention_fewq_chunked(q + (long long)q0 * ldq, k, v, att + (long long)q0 * ldo, B, heads,
                                                Nq - q0 < 7 ? Nq - q0 : 7, Nk, hd, ldq, ldk, ldv, ldo, sqb, skb, svb, sob, scale,
                                                scratch, scratch_bytes, st));
    return HGL_OK;
  }
  return hgl_launch_attention(q, k, v, att, B, heads, Nq, Nk, hd, ldq, ldk, ldv, ldo, sqb, skb, svb, sob, scale, HGL_MASK_NONE,
                              nullptr, 0, 0, nullptr, nullptr, 0, 0, st);
}

// Attention.forward (modeling/transformer.py:218-240).  q: [Bq? , Nq, C] rows; when q_shared the same
// Nq rows serve every batch (batch stride 0).  out: [B, Nq, C] (+ residual R, may alias out).
int dec_attn(const HglSamDecoderW* w, const HglSamAttnW& a, const float* q, bool q_shared, int Nq, const float* k,
             const float* v, bool kv_shared, int Nk, int B, float* qp, float* kp, float* vp, float* att,
             const float* R, long long sR, float* out, hipStream_t st, float* scratch = nullptr, size_t scratch_bytes = 0) {
  const int C = w->C, I = a.internal, heads = w->heads, hd = I / heads;
  const int Bq = q_shared ? 1 : B, Bk = kv_shared ? 1 : B;
  HGL_TRY(lin(q, C, a.q, nullptr, 0, qp, I, Bq * Nq, I, C, HGL_ACT_NONE, st));
  HGL_TRY(lin(k, C, a.k, nullptr, 0, kp, I, Bk * Nk, I, C, HGL_ACT_NONE, st));
  HGL_TRY(lin(v, C, a.v, nullptr, 0, vp, I, Bk * Nk, I, C, HGL_ACT_NONE, st));
  HGL_TRY(dec_fewq(qp, kp, vp, att, B, heads, Nq, Nk, hd, I, I, I, I, q_shared ? 0 : (long long)Nq * I,
                   kv_shared ? 0 : (long long)Nk * I, kv_shared ? 0 : (long long)Nk * I, (long long)Nq * I, scratch,
                   scratch_bytes, st));
  // out_proj (+ residual): one GEMM over all B*Nq rows when the residual is laid out like the output (small row
  // counts then take the small-tile kernel); batched when a shared residual (stride 0) has to be broadcast
  if (!R || sR == (long long)Nq * C)
    return lin(att, I, a.out, R, C, out, C, B * Nq, C, I, HGL_ACT_NONE, st);
  return hgl_launch_gemm(att, a.out.w, a.out.b, R, out, Nq, C, I, I, I, C, C, B, (long long)Nq * I, 0, sR,
                         (long long)Nq * C, HGL_ACT_NONE, st);
}

// ---- f16x3 path of the decoder's image-token side (M = P*HW rows) -------------------------------------------
// The normalised image tokens exist as fp16 hi|lo planes (keysS) and, with the positional encoding added, as a
// second pair (kpeS): ln256_pe_split emits both, the few-key attention emits its output split, so every large
// GEMM below reads split operands and nothing is converted in a separate pass.
struct SplitPair { uint16_t *hi, *lo; };
inline SplitPair split_view(float* buf, size_t elems) { return SplitPair{(uint16_t*)buf, (uint16_t*)buf + elems}; }

bool dec_x3_ready(const HglSamDecoderW* w) {
  if (hgl_precision() != HGL_PREC_F16X3 || w->C != 256) return false;
  const float* need[] = {w->layer[0].i2t.out.w, w->layer[1].i2t.out.w, w->layer[1].i2t.q.w, w->layer[1].t2i.k.w,
                         w->layer[1].t2i.v.w, w->final_t2i.k.w, w->final_t2i.v.w, w->up0_w, w->up3_w};
  for (const float* x : need) if (!hgl_has_split_weight(x)) return false;
  return true;
}

// image -> token attention of one layer (transformer.py:139-150): q = (keys + pe) Wq, 7 token keys / values,
// keys' = keys + out_proj(attn)   (LayerNorm follows in the caller)
int dec_i2t_x3(const HglSamDecoderW* w, const HglSamAttnW& a, bool shared, const float* kpe0, const SplitPair& kpeS,
               const float* tok_k, const float* tok_v, int P, int HW, int T, float* qi, float* k1, float* v1, float* atti,
               const float* R, int rmod, float* keys_out, hipStream_t st) {
  const int C = w->C, I = a.internal, heads = w->heads, hd = I / heads;
  if (shared) {
    HGL_TRY(lin(kpe0, C, a.q, nullptr, 0, qi, I, HW, I, C, HGL_ACT_NONE, st));
  } else {
    HGL_TRY(hgl_launch_gemm_f16x3(kpeS.hi, kpeS.lo, C, a.q.w, a.q.b, nullptr, 0, qi, nullptr, nullptr, I, P * HW, I, C,
                                  HGL_ACT_NONE, st));
  }
  HGL_TRY(lin(tok_k, C, a.k, nullptr, 0, k1, I, P * T, I, C, HGL_ACT_NONE, st));
  HGL_TRY(lin(tok_v, C, a.v, nullptr, 0, v1, I, P * T, I, C, HGL_ACT_NONE, st));
  const SplitPair at = split_view(atti, (size_t)P * HW * I);
  // (the few-key kernel for up to 8 tokens, the general one beyond)
  HGL_TRY(hgl_launch_attention_split(qi, k1, v1, nullptr, at.hi, at.lo, P, heads, HW, T, hd, I, I, I, I,
                                     shared ? 0 : (long long)HW * I, (long long)T * I, (long long)T * I, (long long)HW * I,
                                     1.0f / sqrtf((float)hd), HGL_MASK_NONE, nullptr, 0, 0, nullptr, nullptr, 0, 0, st));
  return hgl_launch_gemm_f16x3_rmod(at.hi, at.lo, I, a.out.w, a.out.b, R, C, rmod, keys_out, nullptr, nullptr, C, P * HW, C,
                                    I, HGL_ACT_NONE, st);
}

// token -> image attention with per-prompt image tokens (transformer.py:126-131): K/V projections read the split planes
int dec_t2i_x3(const HglSamDecoderW* w, const HglSamAttnW& a, const float* qpe, const SplitPair& kpeS, const SplitPair& keysS,
               int P, int HW, int T, float* q1, float* kp, float* vp, float* att, float* queries, float* scratch,
               size_t scratch_bytes, hipStream_t st) {
  const int C = w->C, I = a.internal, heads = w->heads, hd = I / heads;
  HGL_TRY(lin(qpe, C, a.q, nullptr, 0, q1, I, P * T, I, C, HGL_ACT_NONE, st));
  HGL_TRY(hgl_launch_gemm_f16x3(kpeS.hi, kpeS.lo, C, a.k.w, a.k.b, nullptr, 0, kp, nullptr, nullptr, I, P * HW, I, C,
                                HGL_ACT_NONE, st));
  HGL_TRY(hgl_launch_gemm_f16x3(keysS.hi, keysS.lo, C, a.v.w, a.v.b, nullptr, 0, vp, nullptr, nullptr, I, P * HW, I, C,
                                HGL_ACT_NONE, st));
  HGL_TRY(dec_fewq(q1, kp, vp, att, P, heads, T, HW, hd, I, I, I, I, (long long)T * I, (long long)HW * I, (long long)HW * I,
                   (long long)T * I, scratch, scratch_bytes, st));
  return lin(att, I, a.out, queries, C, queries, C, P * T, C, I, HGL_ACT_NONE, st);
}

// ---- merged image-side projections (HglSamDecoderW.kvq1 / kvf) ------------------------------------------------------
bool dec_merged_ready(const HglSamDecoderW* w) {
  return (dec_fusion_mask() & 2) && w->kvq1_w && w->kvq1_b && w->kvq1_pe && w->kvf_w && w->kvf_b && w->kvf_pe &&
         hgl_has_split_weight(w->kvq1_w) && hgl_has_split_weight(w->kvf_w);
}

// kvq [P*HW, N] = keys W^T + b + pe_table[row % HW]   (N = 3I: k | v | q of layer 1;  N = 2I: k | v of the final attention)
int dec_project_merged(const SplitPair& keysS, const float* W, const float* b, const float* pe_tab, int P, int HW, int C, int N,
                       float* kvq, hipStream_t st) {
  return hgl_launch_gemm_f16x3_rmod(keysS.hi, keysS.lo, C, W, b, pe_tab, N, HW, kvq, nullptr, nullptr, N, P * HW, N, C,
                                    HGL_ACT_NONE, st);
}

// token -> image attention on merged projections: k = kvq[:, 0:I], v = kvq[:, I:2I], row stride ld
int dec_t2i_merged(const HglSamDecoderW* w, const HglSamAttnW& a, const float* qpe, const float* kvq, int ld, int P, int HW,
                   int T, float* q1, float* att, float* queries, float* scratch, size_t scratch_bytes, hipStream_t st) {
  const int C = w->C, I = a.internal, heads = w->heads, hd = I / heads;
  HGL_TRY(lin(qpe, C, a.q, nullptr, 0, q1, I, P * T, I, C, HGL_ACT_NONE, st));
  HGL_TRY(dec_fewq(q1, kvq, kvq + I, att, P, heads, T, HW, hd, I, ld, ld, I, (long long)T * I, (long long)HW * ld,
                   (long long)HW * ld, (long long)T * I, scratch, scratch_bytes, st));
  return lin(att, I, a.out, queries, C, queries, C, P * T, C, I, HGL_ACT_NONE, st);
}

// token -> image attention on the RAW image-token planes (sam_decoder_t2i.hip): the 7 tokens are projected through W_k / W_v
// instead of the HW image tokens.  Scratch: `small` (>= P * 56 * 256 * 8 bytes: the folded queries' planes + the attended
// rows), `bias` (P * 56 * HW floats).
int dec_t2i_raw(const HglSamDecoderW* w, const HglSamAttnW& a, const float* qpe, const SplitPair& keysS, int P, int HW, int T,
                float* q1, float* small, float* bias, float* att, float* queries, hipStream_t st) {
  const int C = w->C, I = a.internal, heads = w->heads, hd = I / heads;
  HGL_TRY(lin(qpe, C, a.q, nullptr, 0, q1, I, P * T, I, C, HGL_ACT_NONE, st));
  uint16_t* Qh = (uint16_t*)small;
  uint16_t* Ql = Qh + (size_t)P * 56 * C;
  float* A = (float*)(Ql + (size_t)P * 56 * C);      // first the folded queries in fp32, then the attended rows
  HGL_TRY(hgl_launch_t2i_fold_q(q1, a.k.w, 1.0f / sqrtf((float)hd), A, P, st));
  HGL_TRY(hgl_launch_split_f16(A, 1.0f, Qh, Ql, (long long)P * 56 * C, st));
  // bias[p*56 + r, key] = Qk[p*56 + r, :] . pe[key, :]: pe is the "weight" [HW, C] of a split-fp16 GEMM
  HGL_TRY(hgl_launch_gemm_f16x3(Qh, Ql, C, w->dense_pe, nullptr, nullptr, 0, bias, nullptr, nullptr, HW, P * 56, HW, C,
                                HGL_ACT_NONE, st));
  // (prompt batches of <= 128: eight key ranges per prompt, their partial rows behind the folded queries; see the kernel)
  const int ns = hgl_t2i_key_ranges(P, HW);
  float* const rows = ns > 1 ? A + (size_t)P * 56 * C : A;
  HGL_TRY(hgl_launch_t2i_raw_attn(Qh, Ql, bias, keysS.hi, keysS.lo, P, HW, rows, ns, st));
  HGL_TRY(hgl_launch_t2i_unfold_v(rows, ns, a.v.w, a.v.b, att, P, st));
  return lin(att, I, a.out, queries, C, queries, C, P * T, C, I, HGL_ACT_NONE, st);
}

// image -> token attention on merged projections: q = kvq[:, 2I:3I]
int dec_i2t_merged(const HglSamDecoderW* w, const HglSamAttnW& a, const float* kvq, int ld, const float* tok_k, const float* tok_v,
                   int P, int HW, int T, float* k1, float* v1, float* atti, const float* R, float* keys_out, hipStream_t st) {
  const int C = w->C, I = a.internal, heads = w->heads, hd = I / heads;
  HGL_TRY(lin(tok_k, C, a.k, nullptr, 0, k1, I, P * T, I, C, HGL_ACT_NONE, st));
  HGL_TRY(lin(tok_v, C, a.v, nullptr, 0, v1, I, P * T, I, C, HGL_ACT_NONE, st));
  const SplitPair at = split_view(atti, (size_t)P * HW * I);
  HGL_TRY(hgl_launch_attention_split(kvq + 2 * I, k1, v1, nullptr, at.hi, at.lo, P, heads, HW, T, hd, ld, I, I, I,
                                     (long long)HW * ld, (long long)T * I, (long long)T * I, (long long)HW * I,
                                     1.0f / sqrtf((float)hd), HGL_MASK_NONE, nullptr, 0, 0, nullptr, nullptr, 0, 0, st));
  return hgl_launch_gemm_f16x3_rmod(at.hi, at.lo, I, a.out.w, a.out.b, R, C, 0, keys_out, nullptr, nullptr, C, P * HW, C, I,
                                    HGL_ACT_NONE, st);
}

}  // namespace

extern "C" {

size_t hgl_sam_encode_batch_workspace_bytes(const HglSamEncoderW* w, int nb) {
  if (!valid_enc(w) || nb < 1) return 0;
  HglArena ar(nullptr, 0);
  EncPlan p;
  carve_enc(ar, w, nb, p);
  return ar.off;
}

size_t hgl_sam_encode_workspace_bytes(const HglSamEncoderW* w) { return hgl_sam_encode_batch_workspace_bytes(w, 1); }

// nb images through the encoder at once: the token rows of the images are stacked, so every GEMM / LayerNorm / window
// attention launch covers all of them (weights read once, M = nb * 4096: the tilings fill the chip better and mlp.lin2
// needs no split-K).  Each image's result is what the single-image call gives up to the summation order of split-K.
int hgl_sam_encode_batch(const HglSamEncoderW* w, const uint8_t* const* resized_imgs, const int* in_h, const int* in_w, int nb,
                         float* emb, void* workspace, size_t workspace_bytes, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(valid_enc(w), "sam_encode: invalid weight struct");
  HGL_REQUIRE(resized_imgs && in_h && in_w && emb && nb >= 1 && nb <= 64, "sam_encode: bad arguments (nb %d)", nb);
  for (int i = 0; i < nb; ++i)
    HGL_REQUIRE(resized_imgs[i] && in_h[i] > 0 && in_w[i] > 0 && in_h[i] <= w->img_size && in_w[i] <= w->img_size,
                "sam_encode: bad image %d (%dx%d for img_size %d)", i, in_h[i], in_w[i], w->img_size);
  HglArena ar(workspace, workspace_bytes);
  EncPlan p;
  if (!workspace || !carve_enc(ar, w, nb, p)) {
    hgl_set_error("sam_encode: workspace too small (%zu bytes given)", workspace_bytes);
    return HGL_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int D = w->embed_dim, S = w->img_size, g = S / w->patch, C = w->out_chans, T1 = g * g, T = nb * T1;
  const int kd = 3 * w->patch * w->patch;
  for (int i = 0; i < nb; ++i)
    HGL_TRY(hgl_launch_sam_preprocess(resized_imgs[i], in_h[i], in_w[i], S, p.img + (size_t)i * 3 * S * S, st));
  // patch embedding + bias + absolute position embedding (image_encoder.py:107-109); the position rows repeat per image
  if ((w->patch & 3) == 0 && hgl_use_x3(w->patch_w, kd)) {
    uint16_t* ch = (uint16_t*)p.cols;            // im2col written as fp16 hi | lo planes (same bytes as fp32)
    uint16_t* cl = ch + (size_t)T * kd;
    HGL_TRY(hgl_launch_im2col_patch_split(p.img, nb, S, w->patch, ch, cl, st));
    HGL_TRY(hgl_launch_gemm_f16x3_rmod(ch, cl, kd, w->patch_w, w->patch_b, w->pos_embed, D, nb > 1 ? T1 : 0, p.X, nullptr,
                                       nullptr, D, T, D, kd, HGL_ACT_NONE, st));
  } else {
    HGL_TRY(hgl_launch_im2col_patch(p.img, nb, S, w->patch, p.cols, st));
    HGL_TRY(hgl_launch_gemm(p.cols, w->patch_w, w->patch_b, w->pos_embed, p.X, T1, D, kd, kd, kd, D, D, nb,
                            (long long)T1 * kd, 0, 0, (long long)T1 * D, HGL_ACT_NONE, st));
  }
  for (int i = 0; i < w->depth; ++i) {
    const int ws = w->blocks[i].window;
    if (ws > 0) {   // every windowed block shares one window size (build_sam.py:55-101)
      HGL_TRY(hgl_launch_win_maps(g, ws, (g + ws - 1) / ws, nb, p.pad_of, p.tok_of, p.pad_list, p.pad_count, st));
      break;
    }
  }
  for (int i = 0; i < w->depth; ++i) HGL_TRY(enc_block(w, w->blocks[i], p, st));
  // neck: conv1x1 -> LayerNorm2d -> conv3x3(pad 1) -> LayerNorm2d, all on NHWC rows
  const bool neck_x3 = hgl_use_x3(w->neck0_w, D) && hgl_use_x3(w->neck2_w, C * 9);
  if (neck_x3) {   // operands split into the (dead) MLP buffers: H holds T*D, F holds T*4D floats
    uint16_t* xh = (uint16_t*)p.H;
    uint16_t* xl = xh + (size_t)T * D;
    HGL_TRY(hgl_launch_split_f16(p.X, 1.0f, xh, xl, (long long)T * D, st));
    HGL_TRY(hgl_launch_gemm_f16x3(xh, xl, D, w->neck0_w, nullptr, nullptr, 0, p.neckA, nullptr, nullptr, C, T, C, D,
                                  HGL_ACT_NONE, st));
  } else {
    HGL_TRY(hgl_launch_gemm(p.X, w->neck0_w, nullptr, nullptr, p.neckA, T, C, D, D, D, 0, C, 1, 0, 0, 0, 0,
                            HGL_ACT_NONE, st));
  }
  HGL_TRY(hgl_launch_layernorm(p.neckA, w->neck1_w, w->neck1_b, p.neckB, T, C, 1e-6f, st));
  for (int i = 0; i < nb; ++i)
    HGL_TRY(hgl_launch_im2col3x3(p.neckB + (size_t)i * T1 * C, g, C, p.cols3 + (size_t)i * T1 * C * 9, st));
  if (neck_x3 && (size_t)C * 9 <= (size_t)4 * D) {
    uint16_t* ch = (uint16_t*)p.F;
    uint16_t* cl = ch + (size_t)T * C * 9;
    HGL_TRY(hgl_launch_split_f16(p.cols3, 1.0f, ch, cl, (long long)T * C * 9, st));
    HGL_TRY(hgl_launch_gemm_f16x3(ch, cl, C * 9, w->neck2_w, nullptr, nullptr, 0, p.neckA, nullptr, nullptr, C, T, C, C * 9,
                                  HGL_ACT_NONE, st));
  } else {
    HGL_TRY(hgl_launch_gemm(p.cols3, w->neck2_w, nullptr, nullptr, p.neckA, T, C, C * 9, C * 9, C * 9, 0, C, 1, 0, 0,
                            0, 0, HGL_ACT_NONE, st));
  }
  HGL_TRY(hgl_launch_layernorm(p.neckA, w->neck3_w, w->neck3_b, emb, T, C, 1e-6f, st));
  return HGL_OK;
}

int hgl_sam_encode(const HglSamEncoderW* w, const uint8_t* resized_img, int in_h, int in_w, float* emb,
                   void* workspace, size_t workspace_bytes, void* stream) {
  return hgl_sam_encode_batch(w, &resized_img, &in_h, &in_w, 1, emb, workspace, workspace_bytes, stream);
}

int hgl_sam_dense_pe(const HglSamDecoderW* w, const float* grid_coords01, float* dense_pe, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(w && w->pe_gauss && grid_coords01 && dense_pe && w->grid > 0 && w->C > 0, "sam_dense_pe: bad arguments");
  return hgl_launch_pe(grid_coords01, w->pe_gauss, w->grid * w->grid, w->C / 2, 0, nullptr, nullptr, dense_pe,
                       (hipStream_t)stream);
}

size_t hgl_sam_decode_workspace_bytes(const HglSamDecoderW* w, int P) {
  if (!valid_dec(w) || P <= 0) return 0;
  HglArena ar(nullptr, 0);
  DecPlan p;
  carve_dec(ar, w, P, p);
  return ar.off;
}

int hgl_sam_decoder_fusion(int mask) {
  const int old = dec_fusion_mask();
  if (mask >= 0) g_dec_fusion = mask;
  return old;
}

// MaskDecoder.predict_masks.  points01 != null: one foreground point + the padding point per prompt (what
// SamAutomaticMaskGenerator issues); else coords01 [P,n_sparse,2] / labels [P,n_sparse] with n_sparse = 2 .. 11 and, optionally,
// dense [P,HW,C]: per-prompt dense embeddings (mask inputs) instead of no_mask_embed.  first_mask = 1: the three multimask
// outputs (mask tokens 1..3); 0: tokens 0..2 (token 0 is the single-mask output, mask_decoder.py:99-105).
// IoU gate (hgl_sam_decode_points_gated): skip[p] = none of prompt p's three quality predictions exceeds `gate` (NaN counts as
// failing, as `iou_preds > thresh` does in automatic_mask_generator.py:287-288)
__global__ void iou_gate_kernel(const float* __restrict__ iou, int P, float gate, uint8_t* __restrict__ skip) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  skip[p] = (iou[3 * p] > gate || iou[3 * p + 1] > gate || iou[3 * p + 2] > gate) ? 0 : 1;
}

static int decode_impl(const HglSamDecoderW* w, const float* emb, const float* points01, const float* coords01,
                       const int32_t* labels, int n_sparse, const float* dense, int first_mask, int P, float* low_res,
                       float* iou_pred, void* workspace, size_t workspace_bytes, void* stream, bool gated = false,
                       float iou_gate = 0.f) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(valid_dec(w) && w->dense_pe, "sam_decode: invalid weight struct (dense_pe missing?)");
  HGL_REQUIRE(emb && (points01 || (coords01 && labels)) && low_res && iou_pred && P > 0, "sam_decode: null input");
  HGL_REQUIRE(first_mask == 0 || first_mask == 1, "sam_decode: first_mask must be 0 or 1");
  HGL_REQUIRE(n_sparse >= 2 && n_sparse <= 11, "sam_decode: %d sparse tokens per prompt (2 .. 11 supported)", n_sparse);
  HGL_REQUIRE(n_sparse <= 3 || (long long)P * w->heads <= 65535, "sam_decode: %d prompts of more than 3 sparse tokens in one call", P);
  HglArena ar(workspace, workspace_bytes);
  DecPlan p;
  if (!workspace || !carve_dec(ar, w, P, p)) {
    hgl_set_error("sam_decode: workspace too small (%zu bytes given)", workspace_bytes);
    return HGL_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int C = w->C, g = w->grid, HW = g * g, T = 5 + n_sparse;
  const bool perprompt = dense != nullptr;     // the image tokens differ from prompt to prompt already in layer 0
  const long long sQ = (long long)T * C, sK = (long long)HW * C;
  const size_t atti_bytes = (size_t)P * HW * (C / 2) * sizeof(float);

  // ---- prompt encoder + token assembly ----
  if (points01) {
    HGL_TRY(hgl_launch_pe(points01, w->pe_gauss, 2 * P, C / 2, 1, w->point_embed_pos, w->not_a_point, p.sparse, st));
  } else {
    const float* pe4[4] = {w->point_embed_neg, w->point_embed_pos, w->point_embed_box0, w->point_embed_box1};
    HGL_REQUIRE(pe4[0] && pe4[2] && pe4[3], "sam_decode_prompts: the weight struct lacks point_embeddings 0 / 2 / 3");
    HGL_TRY(hgl_launch_pe_labeled(coords01, labels, w->pe_gauss, n_sparse * P, C / 2, w->not_a_point, pe4, p.sparse, st));
  }
  HGL_TRY(hgl_launch_build_tokens(w->iou_token, w->mask_tokens, p.sparse, P, C, T, p.tokens, st));
  if (perprompt) {
    // src[p] = image_embedding + dense[p] (mask_decoder.py:121-123 with dense_prompt_embeddings from mask inputs)
    HGL_TRY(hgl_launch_add_rows_bcast(dense, sK, emb, sK, P, p.keys, st));
    HGL_TRY(hgl_launch_add_rows_bcast(p.keys, sK, w->dense_pe, sK, P, p.kpe, st));
  } else {
    // src = image_embedding + no_mask_embed (dense prompt) ; shared by all prompts until the first update
    HGL_TRY(hgl_launch_add_rows_bcast(emb, C, w->no_mask, C, HW, p.keys0, st));   // rows of C, "pe" = no_mask [C]
    HGL_TRY(hgl_launch_add_rows_bcast(p.keys0, 0, w->dense_pe, (long long)HW * C, 1, p.kpe0, st));
  }
  (void)hipMemcpyAsync(p.queries, p.tokens, sizeof(float) * P * sQ, hipMemcpyDeviceToDevice, st);

  const bool x3 = dec_x3_ready(w);
  const bool merged = x3 && dec_merged_ready(w) && w->layer[1].t2i.internal == w->layer[1].i2t.internal &&
                      w->layer[1].t2i.internal == w->final_t2i.internal && 2 * w->final_t2i.internal == C;
  const int I1 = w->layer[1].t2i.internal;
  const SplitPair keysS = split_view(p.keysS, (size_t)P * HW * C), kpeS = split_view(p.kpe, (size_t)P * HW * C);
  // fusion bit 5: the token -> image attention of layer 1 and the final one on the raw image-token planes (the 7 tokens go
  // through W_k / W_v instead of the HW image tokens: no k | v projection GEMM; layer 1 projects q alone for its step 4)
  const bool raw_t2i = merged && (dec_fusion_mask() & 32) && (dec_fusion_mask() & 4) && T == 7 && I1 == 128 && w->heads == 8 &&
                       C == 256 && HW % 128 == 0 && P <= 65535 && !perprompt && hgl_has_split_weight(w->dense_pe) &&
                       (size_t)P * (56 * C * 8) + hgl_t2i_part_bytes(P, HW) <= atti_bytes &&   // dec_t2i_raw: Q' planes + attended rows (+ partials)
                       (size_t)P * (56 * C * 4 + 16384 * 4 + 256) <= atti_bytes;      // step (4): K' and U planes + cb
  for (int li = 0; li < 2; ++li) {
    const auto& L = w->layer[li];
    const bool shared = li == 0 && !perprompt;   // keys identical for every prompt in layer 0
    // layer 0 on per-prompt image tokens: the plain fp32 launches (the split planes / merged weights belong to layer 1's input)
    const bool plain0 = li == 0 && perprompt;
    const float* keys = shared ? p.keys0 : p.keys;
    const float* kpe = shared ? p.kpe0 : p.kpe;
    // (1) self attention of the tokens
    if (li == 0) {  // skip_first_layer_pe: queries = self_attn(q,q,q), no residual
      HGL_TRY(dec_attn(w, L.self_attn, p.queries, false, T, p.queries, p.queries, false, T, P, p.q1, p.k1, p.v1, p.att,
                       nullptr, 0, p.qpe, st));
      (void)hipMemcpyAsync(p.queries, p.qpe, sizeof(float) * P * sQ, hipMemcpyDeviceToDevice, st);
    } else {
      HGL_TRY(hgl_launch_add_rows_bcast(p.queries, P * sQ, p.tokens, P * sQ, 1, p.qpe, st));
      HGL_TRY(dec_attn(w, L.self_attn, p.qpe, false, T, p.qpe, p.queries, false, T, P, p.q1, p.k1, p.v1, p.att,
                       p.queries, sQ, p.queries, st));
    }
    HGL_TRY(hgl_launch_layernorm(p.queries, L.n1.w, L.n1.b, p.queries, P * T, C, 1e-5f, st));
    // (2) tokens attend to the image
    HGL_TRY(hgl_launch_add_rows_bcast(p.queries, P * sQ, p.tokens, P * sQ, 1, p.qpe, st));
    if (raw_t2i && !shared && !plain0) {
      // no projection of the image tokens at all: this attention reads the raw planes (its positional term in p.kp), and
      // step (4) below folds W_q into the 7 token keys as well
      HGL_TRY(dec_t2i_raw(w, L.t2i, p.qpe, keysS, P, HW, T, p.q1, p.atti, p.kp, p.att, p.queries, st));
    } else if (merged && !shared && !plain0) {
      // k, v of this step and q of step (4) read the same rows: one GEMM, the positional encoding as a per-position table
      HGL_TRY(dec_project_merged(keysS, w->kvq1_w, w->kvq1_b, w->kvq1_pe, P, HW, C, 3 * I1, p.kp, st));
      HGL_TRY(dec_t2i_merged(w, L.t2i, p.qpe, p.kp, 3 * I1, P, HW, T, p.q1, p.att, p.queries, p.atti, atti_bytes, st));
    } else if (x3 && !shared && !plain0) {
      HGL_TRY(dec_t2i_x3(w, L.t2i, p.qpe, kpeS, keysS, P, HW, T, p.q1, p.kp, p.vp, p.att, p.queries, p.atti, atti_bytes, st));
    } else {
      HGL_TRY(dec_attn(w, L.t2i, p.qpe, false, T, kpe, keys, shared, HW, P, p.q1, p.kp, p.vp, p.att, p.queries, sQ,
                       p.queries, st, p.atti, atti_bytes));
    }
    HGL_TRY(hgl_launch_layernorm(p.queries, L.n2.w, L.n2.b, p.queries, P * T, C, 1e-5f, st));
    // (3) MLP on the tokens
    HGL_TRY(lin(p.queries, C, L.lin1, nullptr, 0, p.mlp, w->mlp_dim, P * T, w->mlp_dim, C, HGL_ACT_RELU, st));
    HGL_TRY(lin(p.mlp, w->mlp_dim, L.lin2, p.queries, C, p.queries, C, P * T, C, w->mlp_dim, HGL_ACT_NONE, st));
    HGL_TRY(hgl_launch_layernorm(p.queries, L.n3.w, L.n3.b, p.queries, P * T, C, 1e-5f, st));
    // (4) image attends to the tokens: q = keys+pe, k = queries+pe, v = queries ; keys += out
    HGL_TRY(hgl_launch_add_rows_bcast(p.queries, P * sQ, p.tokens, P * sQ, 1, p.qpe, st));
    const bool fuse_i2t = merged && (dec_fusion_mask() & 4) && L.i2t.internal == I1 && I1 == 128 && w->heads == 8 &&
                          HW % 64 == 0 && P <= 65535 && T == 7 && !plain0;
    if (fuse_i2t && raw_t2i && !shared) {
      // layer 1 on per-prompt image tokens: scores = (keys + pe) . (W_q^T k_tok) and update = P (W_o v_tok) by MFMA against
      // per-prompt 56 x 256 matrices (sam_decoder_t2i.hip: dec_i2t_fold_kernel); the planes are updated in place
      HGL_TRY(lin(p.qpe, C, L.i2t.k, nullptr, 0, p.k1, I1, P * T, I1, C, HGL_ACT_NONE, st));
      HGL_TRY(lin(p.queries, C, L.i2t.v, nullptr, 0, p.v1, I1, P * T, I1, C, HGL_ACT_NONE, st));
      uint16_t* Kh = (uint16_t*)p.atti;                            // [P*56, 256] K' hi / lo
      uint16_t* Kl = Kh + (size_t)P * 56 * C;
      uint16_t* Uh = Kl + (size_t)P * 56 * C;                      // [P, 16, 2, 64, 8] U fragments hi / lo
      uint16_t* Ul = Uh + (size_t)P * 16384;
      float* cbv = (float*)(Ul + (size_t)P * 16384);               // [P*56]
      HGL_TRY(hgl_launch_i2t_prep(p.k1, p.v1, L.i2t.q.w, L.i2t.q.b, L.i2t.out.w, 1.0f / sqrtf((float)(I1 / w->heads)), Kh, Kl, cbv, Uh,
                                  Ul, P, st));
      HGL_TRY(hgl_launch_gemm_f16x3(Kh, Kl, C, w->dense_pe, nullptr, nullptr, 0, p.kp, nullptr, nullptr, HW, P * 56, HW, C,
                                    HGL_ACT_NONE, st));
      HGL_TRY(hgl_launch_dec_i2t_fold(keysS.hi, keysS.lo, Kh, Kl, p.kp, cbv, Uh, Ul, L.i2t.out.b, L.n4.w, L.n4.b, 1e-5f, P, HW,
                                      keysS.hi, keysS.lo, st));
    } else if (fuse_i2t) {
      // attention over the 7 tokens, out-projection, residual and norm4 in one launch: the image tokens leave it as the
      // split planes the next projections read (and, in layer 0, as the fp32 rows layer 1 adds its update to)
      HGL_TRY(lin(p.qpe, C, L.i2t.k, nullptr, 0, p.k1, I1, P * T, I1, C, HGL_ACT_NONE, st));
      HGL_TRY(lin(p.queries, C, L.i2t.v, nullptr, 0, p.v1, I1, P * T, I1, C, HGL_ACT_NONE, st));
      if (shared) HGL_TRY(lin(p.kpe0, C, L.i2t.q, nullptr, 0, p.qi, I1, HW, I1, C, HGL_ACT_NONE, st));
      HGL_TRY(hgl_launch_dec_i2t(shared ? p.qi : p.kp + 2 * I1, shared ? I1 : 3 * I1, shared ? 0 : (long long)HW * 3 * I1, p.k1,
                                 p.v1, L.i2t.out.w, L.i2t.out.b, keys, shared ? 0 : sK, L.n4.w, L.n4.b, 1e-5f,
                                 1.0f / sqrtf((float)(I1 / w->heads)), P, HW, (li == 0 && !raw_t2i) ? p.keys : nullptr, keysS.hi,
                                 keysS.lo, st));
    } else if (x3 && plain0) {
      HGL_TRY(dec_attn(w, L.i2t, kpe, false, HW, p.qpe, p.queries, false, T, P, p.qi, p.k1, p.v1, p.atti, keys, sK, p.keys, st));
      HGL_TRY(hgl_launch_ln256_pe_split(p.keys, L.n4.w, L.n4.b, w->dense_pe, HW, (long long)P * HW, 1e-5f, 1, keysS.hi, keysS.lo,
                                        merged ? nullptr : kpeS.hi, merged ? nullptr : kpeS.lo, st));
    } else if (x3) {
      if (merged && !shared) {
        HGL_TRY(dec_i2t_merged(w, L.i2t, p.kp, 3 * I1, p.qpe, p.queries, P, HW, T, p.k1, p.v1, p.atti, keys, p.keys, st));
      } else {
        HGL_TRY(dec_i2t_x3(w, L.i2t, shared, p.kpe0, kpeS, p.qpe, p.queries, P, HW, T, p.qi, p.k1, p.v1, p.atti, keys,
                           shared ? HW : 0, p.keys, st));
      }
      // norm4, then keys (and, unmerged, keys + dense_pe) as split planes; the fp32 rows are kept only while a later layer
      // needs them as a residual
      HGL_TRY(hgl_launch_ln256_pe_split(p.keys, L.n4.w, L.n4.b, w->dense_pe, HW, (long long)P * HW, 1e-5f, li == 0 ? 1 : 0,
                                        keysS.hi, keysS.lo, merged ? nullptr : kpeS.hi, merged ? nullptr : kpeS.lo, st));
    } else {
      HGL_TRY(dec_attn(w, L.i2t, kpe, shared, HW, p.qpe, p.queries, false, T, P, p.qi, p.k1, p.v1, p.atti, keys,
                       shared ? 0 : sK, p.keys, st));
      HGL_TRY(hgl_launch_layernorm(p.keys, L.n4.w, L.n4.b, p.keys, P * HW, C, 1e-5f, st));
      HGL_TRY(hgl_launch_add_rows_bcast(p.keys, sK, w->dense_pe, sK, P, p.kpe, st));
    }
  }
  // final token -> image attention
  HGL_TRY(hgl_launch_add_rows_bcast(p.queries, P * sQ, p.tokens, P * sQ, 1, p.qpe, st));
  if (raw_t2i) {
    HGL_TRY(dec_t2i_raw(w, w->final_t2i, p.qpe, keysS, P, HW, T, p.q1, p.atti, p.kp, p.att, p.queries, st));
  } else if (merged) {
    HGL_TRY(dec_project_merged(keysS, w->kvf_w, w->kvf_b, w->kvf_pe, P, HW, C, 2 * I1, p.kp, st));
    HGL_TRY(dec_t2i_merged(w, w->final_t2i, p.qpe, p.kp, 2 * I1, P, HW, T, p.q1, p.att, p.queries, p.atti, atti_bytes, st));
  } else if (x3) {
    HGL_TRY(dec_t2i_x3(w, w->final_t2i, p.qpe, kpeS, keysS, P, HW, T, p.q1, p.kp, p.vp, p.att, p.queries, p.atti, atti_bytes, st));
  } else {
    HGL_TRY(dec_attn(w, w->final_t2i, p.qpe, false, T, p.kpe, p.keys, false, HW, P, p.q1, p.kp, p.vp, p.att, p.queries,
                     sQ, p.queries, st, p.atti, atti_bytes));
  }
  HGL_TRY(hgl_launch_layernorm(p.queries, w->norm_final.w, w->norm_final.b, p.queries, P * T, C, 1e-5f, st));

  // ---- IoU head on the iou token (row 0); multimask output = columns 1..3.  BEFORE the upscaling: the predictions depend on
  // the token outputs only (mask_decoder.py:132-149), and the automatic generator drops every mask whose prediction does not
  // exceed pred_iou_thresh (automatic_mask_generator.py:287-291) -- a prompt whose three predictions all fail needs no
  // upscaling at all (the gate of hgl_sam_decode_points_gated) ----
  HGL_TRY(lin(p.queries, T * C, w->iou_head[0], nullptr, 0, p.iou_a, C, P, C, C, HGL_ACT_RELU, st));
  HGL_TRY(lin(p.iou_a, C, w->iou_head[1], nullptr, 0, p.iou_b, C, P, C, C, HGL_ACT_RELU, st));
  HGL_TRY(lin(p.iou_b, C, w->iou_head[2], nullptr, 0, p.iou_a, 4, P, 4, C, HGL_ACT_NONE, st));
  HGL_TRY(hgl_launch_gather_rows(p.iou_a + first_mask, 4, P, 3, iou_pred, st));
  const uint8_t* skip = nullptr;
  if (gated) {
    hipLaunchKernelGGL(iou_gate_kernel, dim3((P + 255) / 256), dim3(256), 0, st, (const float*)iou_pred, P, iou_gate, p.skip);
    skip = p.skip;
  }

  // ---- output upscaling: two ConvTranspose2d(k=2,s=2) as GEMMs, columns ordered (pos, out_channel) ----
  const int C4 = C / 4, C8 = C / 8;
  HGL_REQUIRE(C4 == 64, "sam_decode: LayerNorm2d width %d unsupported (64 expected)", C4);
  HGL_REQUIRE(C8 == 32, "sam_decode: hyper-network width %d unsupported (32 expected)", C8);
  // ---- hyper-networks on the mask tokens (rows 1..4 of each prompt's 7 tokens) ----
  for (int i = 0; i < 4; ++i) {
    HGL_TRY(lin(p.queries + (1 + i) * C, T * C, w->hyper[i][0], nullptr, 0, p.hy_a, C, P, C, C, HGL_ACT_RELU, st));
    HGL_TRY(lin(p.hy_a, C, w->hyper[i][1], nullptr, 0, p.hy_b, C, P, C, C, HGL_ACT_RELU, st));
    HGL_TRY(lin(p.hy_b, C, w->hyper[i][2], nullptr, 0, p.hyper + i * C8, 4 * C8, P, C8, C, HGL_ACT_NONE, st));
  }
  // fused upscaling + hyper-network products (one launch, the 256-channel rows read once); hgl_sam_decoder_fusion(0) /
  // HGL_SAM_DEC_FUSED=0 keep the four launches below (same products, sums associated differently: tests compare the two)
  const bool fused_tail = x3 && (dec_fusion_mask() & 1) && (HW % 64) == 0 && (g % 64 == 0 || 64 % g == 0) && P <= 65535;
  if (fused_tail) {
    HGL_TRY(hgl_launch_dec_tail(keysS.hi, keysS.lo, w->up0_w, w->up0_b, w->up1.w, w->up1.b, w->up3_w, w->up3_b, p.hyper, first_mask,
                                P, g, 1e-6f, low_res, skip, st));
  } else {
    if (x3) {
      HGL_TRY(hgl_launch_gemm_f16x3(keysS.hi, keysS.lo, C, w->up0_w, w->up0_b, nullptr, 0, p.u1, nullptr, nullptr, 4 * C4,
                                    P * HW, 4 * C4, C, HGL_ACT_NONE, st));
      const SplitPair u1S = split_view(p.kpe, (size_t)P * HW * 4 * C4);   // kpeS is dead from here on
      HGL_TRY(hgl_launch_ln_gelu64(p.u1, w->up1.w, w->up1.b, (long long)P * HW * 4, 1e-6f, u1S.hi, u1S.lo, st));
      HGL_TRY(hgl_launch_gemm_f16x3(u1S.hi, u1S.lo, C4, w->up3_w, w->up3_b, nullptr, 0, p.u2, nullptr, nullptr, 4 * C8,
                                    P * HW * 4, 4 * C8, C4, HGL_ACT_GELU, st));
    } else {
      HGL_TRY(hgl_launch_gemm(p.keys, w->up0_w, w->up0_b, nullptr, p.u1, P * HW, 4 * C4, C, C, C, 0, 4 * C4, 1, 0, 0, 0, 0,
                              HGL_ACT_NONE, st));
      HGL_TRY(hgl_launch_ln_gelu64(p.u1, w->up1.w, w->up1.b, (long long)P * HW * 4, 1e-6f, nullptr, nullptr, st));
      HGL_TRY(hgl_launch_gemm(p.u1, w->up3_w, w->up3_b, nullptr, p.u2, P * HW * 4, 4 * C8, C4, C4, C4, 0, 4 * C8, 1, 0, 0,
                              0, 0, HGL_ACT_GELU, st));
    }
    // masks[p, t, pix] = hyper[p, t, :] . upscaled[p, pix, :] for the three multimask tokens, un-shuffled into
    // [P,3,4g,4g] by the same kernel
    HGL_TRY(hgl_launch_hyper_logits(p.u2, p.hyper, P, g, first_mask, low_res, st));
  }
  return HGL_OK;
}

int hgl_sam_decode_points(const HglSamDecoderW* w, const float* emb, const float* points01, int P, float* low_res,
                          float* iou_pred, void* workspace, size_t workspace_bytes, void* stream) {
  HGL_REQUIRE(points01, "sam_decode: null input");
  return decode_impl(w, emb, points01, nullptr, nullptr, 2, nullptr, 1, P, low_res, iou_pred, workspace, workspace_bytes, stream);
}

int hgl_sam_decode_points_gated(const HglSamDecoderW* w, const float* emb, const float* points01, int P, float iou_gate,
                                float* low_res, float* iou_pred, void* workspace, size_t workspace_bytes, void* stream) {
  HGL_REQUIRE(points01, "sam_decode: null input");
  return decode_impl(w, emb, points01, nullptr, nullptr, 2, nullptr, 1, P, low_res, iou_pred, workspace, workspace_bytes, stream, true,
                     iou_gate);
}

int hgl_sam_decode_prompts(const HglSamDecoderW* w, const float* emb, const float* coords01, const int32_t* labels, int n_sparse,
                           const float* dense, int first_mask, int P, float* low_res, float* iou_pred, void* workspace,
                           size_t workspace_bytes, void* stream) {
  HGL_REQUIRE(coords01 && labels, "sam_decode_prompts: null input");
  return decode_impl(w, emb, nullptr, coords01, labels, n_sparse, dense, first_mask, P, low_res, iou_pred, workspace,
                     workspace_bytes, stream);
}

int hgl_sam_embed_masks(const HglSamDecoderW* w, const float* mask_input, int P, float* dense, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(w && mask_input && dense && P > 0 && P <= 65535, "sam_embed_masks: bad arguments");
  HGL_REQUIRE(w->md_c1_w && w->md_c1_b && w->md_n1_w && w->md_n1_b && w->md_c2_w && w->md_c2_b && w->md_n2_w && w->md_n2_b &&
              w->md_c3_w && w->md_c3_b, "sam_embed_masks: the weight struct lacks mask_downscaling");
  HGL_REQUIRE(w->C == 256, "sam_embed_masks: embedding width %d unsupported", w->C);
  return hgl_launch_mask_downscaling(mask_input, P, w->grid, w->md_c1_w, w->md_c1_b, w->md_n1_w, w->md_n1_b, w->md_c2_w, w->md_c2_b,
                                     w->md_n2_w, w->md_n2_b, w->md_c3_w, w->md_c3_b, dense, (hipStream_t)stream);
}

}  // extern "C"
