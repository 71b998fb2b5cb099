// Host orchestration of the CLIP hybrid encoder and text encoder behind the C ABI.
//
//   hgl_clip_hybrid_forward  == CLIPViTFM.forward          (model/backbone.py:117-309)
//   hgl_clip_encode_text     == CLIP.encode_text           (clip/model.py:414-431)
//
// Activations are kept batch-major [B, S, D] (the reference uses [S, B, D]; the math is
// identical).  The local and global streams (and the two hybrid streams of G2L&L2G) are
// stacked along B so every transformer block is ONE sequence of large GEMMs.  Streams whose
// final-block output is never consumed (SURVEY.md section 3.3 "dead work") are not computed.
#include "hgl_common.h"
#include <math.h>
#include <stdlib.h>

typedef HglBlockBufs BlockBufs;

// The split-fp16 matrix-core path of a block: every GEMM operand activation exists only as fp16 (hi, lo) halves.
// Rows <= 512 (tiny batches) go through hgl_launch_gemm instead, which hands GEMMs with a registered weight to the
// small-tile f16x3 kernel.  Between 512 and ~1500 rows the large tilings are latency-bound when run alone (a 128x128
// tile per CU), but they cost a third of the small-tile kernel's CU-time, which is what matters for work that runs
// on a side stream underneath the SAM / CLIP kernels (text encoder: 12 x 77 rows, GEM: 785 rows): measured
// 49.05 -> 47.8 ms per benchmark step.
bool hgl_clip_block_uses_x3(const HglResBlockW& w, int M, int D) {
  static const int min_m = HGL_DIAG_SWITCH("HGL_X3_MIN_M", 512);
  return M > min_m && hgl_use_x3(w.in_proj_w, D) && hgl_use_x3(w.out_proj_w, D) && hgl_use_x3(w.fc_w, D) &&
         hgl_use_x3(w.proj_w, 4 * D) && (D % 256) == 0;
}

// the attention of a block runs on q | k | v as fp16 hi / lo planes (attention_ps.hip): head dim 64, no causal mask -- GEM's
// 785-token blocks.  CLIP's 197-token sequences stay on attn_x3q_kernel: alone (1024 sequences) the pre-split kernel takes 742
// against 868 us, but in the pipeline's launches (2048 sequences of a group) it is no faster (1771 against 1761 us) while the
// in-projection's split write-out costs 3 % of that GEMM (tools/clip_ps_ab.sh: 31.77 against 31.79 ms of kernel time per ref);
// HGL_ATTN_PS_CLIPBLOCKS=2 routes them here as well (A/B timing, the parity test)
bool hgl_clip_block_presplit(const HglResBlockW& w, int B, int S, int D, int heads, int mask_kind) {
  const int hd = D / heads;
  static const int on = hgl_env_int("HGL_ATTN_PS_CLIPBLOCKS", 1);   // 0: never, 2: also S <= 256
  return on && (S > 256 || on == 2) && hgl_clip_block_uses_x3(w, B * S, D) && hd == 64 && S > 128 &&
         hgl_attention_ps_serves((long long)B * S * 3 * D * 2, 3 * D, B, heads, S, hd, mask_kind, 0, 0, nullptr, nullptr) != 0;
}

// first half of a block: H = ln_1(X) (fp32, or the fp16 hi+lo pair aliasing bf.H on the split path), QKV = H W_in + b
// (split_out: q | k | v as fp16 hi / lo planes aliasing bf.QKV, for hgl_clip_block_rest(..., qkv_split = true))
int hgl_clip_block_qkv(const HglResBlockW& w, const float* X, int M, int D, const HglBlockBufs& bf, hipStream_t st, bool split_out) {
  if (hgl_clip_block_uses_x3(w, M, D)) {
    uint16_t* Hh = (uint16_t*)bf.H;
    uint16_t* Hl = Hh + (size_t)M * D;
    HGL_TRY(hgl_launch_layernorm_split(X, w.ln1_w, w.ln1_b, Hh, Hl, M, D, 1e-5f, st));
    if (split_out) {
      uint16_t* Qh = (uint16_t*)bf.QKV;
      return hgl_launch_gemm_f16x3(Hh, Hl, D, w.in_proj_w, w.in_proj_b, nullptr, 0, nullptr, Qh, Qh + (size_t)M * 3 * D, 3 * D, M,
                                   3 * D, D, HGL_ACT_NONE, st);
    }
    return hgl_launch_gemm_f16x3(Hh, Hl, D, w.in_proj_w, w.in_proj_b, nullptr, 0, bf.QKV, nullptr, nullptr, 3 * D, M, 3 * D, D,
                                 HGL_ACT_NONE, st);
  }
  HGL_REQUIRE(!split_out, "clip_block_qkv: split output exists on the split-fp16 path only");
  HGL_TRY(hgl_launch_layernorm(X, w.ln1_w, w.ln1_b, bf.H, M, D, 1e-5f, st));
  return hgl_launch_gemm(bf.H, w.in_proj_w, w.in_proj_b, nullptr, bf.QKV, M, 3 * D, D, D, D, 0, 3 * D, 1, 0, 0, 0, 0,
                         HGL_ACT_NONE, st);
}

// second half: x <- x + out_proj(attention(QKV)) ; x <- x + mlp(ln_2 x)
int hgl_clip_block_rest(const HglResBlockW& w, float* X, int B, int S, int D, int heads, const HglBlockBufs& bf,
                        int mask_kind, const uint8_t* keep, int keep_b0, int keep_n, hipStream_t st, bool qkv_split) {
  const int M = B * S;
  const int hd = D / heads;
  const long long sQKV = (long long)S * 3 * D;
  if (hgl_clip_block_uses_x3(w, M, D)) {
    // activations feeding a GEMM alias the fp32 scratch buffers (same byte size)
    uint16_t* Hh = (uint16_t*)bf.H;
    uint16_t* Hl = Hh + (size_t)M * D;
    uint16_t* Fh = (uint16_t*)bf.F;
    uint16_t* Fl = Fh + (size_t)M * 4 * D;
    // the attention writes its output as the fp16 hi+lo pair the out-projection reads
    if (qkv_split) {
      const uint16_t* Qh = (const uint16_t*)bf.QKV;
      const int rc = hgl_launch_attention_ps(Qh, Qh + (size_t)M * 3 * D, 3 * D, 0, D, 2 * D, S, B, heads, S, hd, nullptr, Hh, Hl, D,
                                             (long long)S * D, 1.0f / sqrtf((float)hd), mask_kind, keep, keep_b0, keep_n, nullptr,
                                             nullptr, 0, 0, nullptr, nullptr, st);
      if (rc < 0) return rc;
      HGL_REQUIRE(rc == 0, "clip_block: the pre-split attention refused a shape its caller had checked");
    } else {
    HGL_TRY(hgl_launch_attention_split(bf.QKV, bf.QKV + D, bf.QKV + 2 * D, nullptr, Hh, Hl, B, heads, S, S, hd, 3 * D, 3 * D,
                                       3 * D, D, sQKV, sQKV, sQKV, (long long)S * D, 1.0f / sqrtf((float)hd), mask_kind, keep,
                                       keep_b0, keep_n, nullptr, nullptr, 0, 0, st));
    }
    // (balanced: whole rounds of the persistent tiling + a split-K tail when the last round would be mostly empty; the
    // partial sums borrow the qkv buffer, dead after the attention)
    HGL_TRY(hgl_launch_gemm_f16x3_balanced(Hh, Hl, D, nullptr, w.out_proj_w, w.out_proj_b, X, D, nullptr, X, D, M, D, D, HGL_ACT_NONE,
                                           bf.QKV, (size_t)M * 3 * D * sizeof(float), st));
    HGL_TRY(hgl_launch_layernorm_split(X, w.ln2_w, w.ln2_b, Hh, Hl, M, D, 1e-5f, st));
    HGL_TRY(hgl_launch_gemm_f16x3(Hh, Hl, D, w.fc_w, w.fc_b, nullptr, 0, nullptr, Fh, Fl, 4 * D, M, 4 * D, D,
                                  HGL_ACT_QUICKGELU, st));
    // mlp.c_proj with few output tiles and K = 4D (GEM at 785 rows, the text encoder, small batches): split-K over the
    // idle CUs; the partial sums borrow the qkv buffer (dead after the attention; it holds three slices)
    static const int splitk_on = HGL_DIAG_SWITCH("HGL_CLIP_SPLITK", 1);
    int ks = splitk_on ? hgl_gemm_f16x3_splitk_factor(M, D, 4 * D) : 1;
    if (ks > 3) ks = 3;
    if (ks > 1) {
      HGL_TRY(hgl_launch_gemm_f16x3_splitk(Fh, Fl, 4 * D, nullptr, w.proj_w, w.proj_b, X, D, nullptr, X, D, M, D, 4 * D,
                                           HGL_ACT_NONE, ks, bf.QKV, (size_t)M * 3 * D * sizeof(float), st));
    } else {
      HGL_TRY(hgl_launch_gemm_f16x3_balanced(Fh, Fl, 4 * D, nullptr, w.proj_w, w.proj_b, X, D, nullptr, X, D, M, D, 4 * D, HGL_ACT_NONE,
                                             bf.QKV, (size_t)M * 3 * D * sizeof(float), st));
    }
    return HGL_OK;
  }
  HGL_REQUIRE(!qkv_split, "clip_block_rest: split qkv exists on the split-fp16 path only");
  HGL_TRY(hgl_launch_attention(bf.QKV, bf.QKV + D, bf.QKV + 2 * D, bf.H, B, heads, S, S, hd, 3 * D,
                               3 * D, 3 * D, D, sQKV, sQKV, sQKV, (long long)S * D, 1.0f / sqrtf((float)hd),
                               mask_kind, keep, keep_b0, keep_n, nullptr, nullptr, 0, 0, st));
  HGL_TRY(hgl_launch_gemm(bf.H, w.out_proj_w, w.out_proj_b, X, X, M, D, D, D, D, D, D, 1, 0, 0, 0, 0,
                          HGL_ACT_NONE, st));
  HGL_TRY(hgl_launch_layernorm(X, w.ln2_w, w.ln2_b, bf.H, M, D, 1e-5f, st));
  HGL_TRY(hgl_launch_gemm(bf.H, w.fc_w, w.fc_b, nullptr, bf.F, M, 4 * D, D, D, D, 0, 4 * D, 1, 0, 0,
                          0, 0, HGL_ACT_QUICKGELU, st));
  HGL_TRY(hgl_launch_gemm(bf.F, w.proj_w, w.proj_b, X, X, M, D, 4 * D, 4 * D, 4 * D, D, D, 1, 0, 0,
                          0, 0, HGL_ACT_NONE, st));
  return HGL_OK;
}

// x <- x + attn(ln_1 x) ; x <- x + mlp(ln_2 x)      (clip/model.py:244-257)
int hgl_clip_run_block(const HglResBlockW& w, float* X, int B, int S, int D, int heads, const HglBlockBufs& bf,
                       int mask_kind, const uint8_t* keep, int keep_b0, int keep_n, hipStream_t st) {
  const bool ps = hgl_clip_block_presplit(w, B, S, D, heads, mask_kind);
  HGL_TRY(hgl_clip_block_qkv(w, X, B * S, D, bf, st, ps));
  return hgl_clip_block_rest(w, X, B, S, D, heads, bf, mask_kind, keep, keep_b0, keep_n, st, ps);
}

// patch embedding + cls + pos + ln_pre for `n_img` images -> X [n_img, S, D]; cols holds the im2col matrix
// (n_img*P*3p^2 floats), tok the patch tokens (n_img*P*D floats)
int hgl_clip_embed_images(const HglClipVisionW* w, const float* imgs, int n_img, float* X, float* cols, float* tok,
                          hipStream_t st) {
  const int D = w->width, g = w->grid, P = g * g, S = P + 1, kd = 3 * w->patch * w->patch;
  if (n_img * P > 512 && (w->patch & 3) == 0 && hgl_use_x3(w->conv1_w, kd)) {
    // split-fp16 path: the im2col matrix is written directly as fp16 hi | lo planes (same bytes as the fp32 matrix)
    uint16_t* ch = (uint16_t*)cols;
    uint16_t* cl = ch + (size_t)n_img * P * kd;
    HGL_TRY(hgl_launch_im2col_patch_split(imgs, n_img, g * w->patch, w->patch, ch, cl, st));
    HGL_TRY(hgl_launch_gemm_f16x3(ch, cl, kd, w->conv1_w, nullptr, nullptr, 0, tok, nullptr, nullptr, D, n_img * P, D, kd,
                                  HGL_ACT_NONE, st));
  } else {
    HGL_TRY(hgl_launch_im2col_patch(imgs, n_img, g * w->patch, w->patch, cols, st));
    HGL_TRY(hgl_launch_gemm(cols, w->conv1_w, nullptr, nullptr, tok, n_img * P, D, kd, kd, kd, 0, D, 1,
                            0, 0, 0, 0, HGL_ACT_NONE, st));
  }
  HGL_TRY(hgl_launch_assemble_lnpre(tok, w->class_embedding, w->positional_embedding, w->ln_pre_w,
                                    w->ln_pre_b, X, n_img, S, D, st));
  return HGL_OK;
}

namespace {

inline int run_block(const HglResBlockW& w, float* X, int B, int S, int D, int heads, const BlockBufs& bf,
                     int mask_kind, const uint8_t* keep, int keep_b0, int keep_n, hipStream_t st) {
  return hgl_clip_run_block(w, X, B, S, D, heads, bf, mask_kind, keep, keep_b0, keep_n, st);
}

int n_streams(int mode) {
  switch (mode) {
    case HGL_FUSION_G2L:
    case HGL_FUSION_L2G: return 2;
    case HGL_FUSION_G2L_L2G: return 4;
    default: return 1;
  }
}

struct ClipPlan {
  float *X, *Y, *H, *QKV, *F, *pm, *cls_rows, *cls_ln;
  uint8_t* keep;
};

// identical carve for the size query (dry) and the real run
bool carve(HglArena& ar, const HglClipVisionW* w, int N, int mode, ClipPlan& p) {
  const int D = w->width, S = w->grid * w->grid + 1;
  const size_t ns = n_streams(mode);
  const size_t act = ns * N * S * (size_t)D;
  p.X = ar.take<float>(act);
  p.Y = ar.take<float>(act);
  p.H = ar.take<float>(act);
  p.QKV = ar.take<float>(3 * act);
  // F also holds the im2col matrix of one stream (N*P*3p^2 floats)
  const size_t cols = (size_t)N * (S - 1) * 3 * w->patch * w->patch;
  p.F = ar.take<float>(4 * act > cols ? 4 * act : cols);
  p.pm = ar.take<float>((size_t)N * (S - 1));
  p.keep = ar.take<uint8_t>((size_t)N * (S - 1));
  p.cls_rows = ar.take<float>((size_t)2 * N * D);
  p.cls_ln = ar.take<float>((size_t)2 * N * D);
  return ar.ok();
}

int embed_images(const HglClipVisionW* w, const float* imgs, int n_img, float* X, const ClipPlan& p,
                 hipStream_t st) {
  return hgl_clip_embed_images(w, imgs, n_img, X, p.F, p.QKV, st);
}

// The RETURNING block: only the CLS row of its output is consumed (ln_post(x[:, 0]) @ proj, model/backbone.py:254-260),
// so only that row is computed past the attention: ln_1 + the qkv projection run on every token (the keys and values
// of all tokens are needed), the attention on one query per sequence, the out-projection and the MLP on one row per
// sequence.  2.2 of the block's 2.9 GFLOP per sequence are dead rows.  Leaves the CLS rows [B, D] in p.cls_rows.
int run_block_cls(const HglResBlockW& w, const float* X, int B, int S, int D, int heads, const BlockBufs& bf, int mask_kind,
                  const uint8_t* keep, int keep_b0, int keep_n, const ClipPlan& p, hipStream_t st) {
  const int M = B * S, hd = D / heads;
  const long long sQKV = (long long)S * 3 * D;
  HGL_TRY(hgl_clip_block_qkv(w, X, M, D, bf, st));
  float* att = bf.H;            // [B, D]: the ln_1 output is dead after the qkv projection
  float* xcls = p.cls_rows;     // [B, D]
  float* h = p.cls_ln;          // [B, D]
  HGL_TRY(hgl_launch_attention(bf.QKV, bf.QKV + D, bf.QKV + 2 * D, att, B, heads, 1, S, hd, 3 * D, 3 * D, 3 * D, D, sQKV, sQKV,
                               sQKV, (long long)D, 1.0f / sqrtf((float)hd), mask_kind, keep, keep_b0, keep_n, nullptr, nullptr,
                               0, 0, st));
  HGL_TRY(hgl_launch_gather_rows(X, (long long)S * D, B, D, xcls, st));
  HGL_TRY(hgl_launch_gemm(att, w.out_proj_w, w.out_proj_b, xcls, xcls, B, D, D, D, D, D, D, 1, 0, 0, 0, 0, HGL_ACT_NONE, st));
  HGL_TRY(hgl_launch_layernorm(xcls, w.ln2_w, w.ln2_b, h, B, D, 1e-5f, st));
  HGL_TRY(hgl_launch_gemm(h, w.fc_w, w.fc_b, nullptr, bf.F, B, 4 * D, D, D, D, 0, 4 * D, 1, 0, 0, 0, 0, HGL_ACT_QUICKGELU, st));
  HGL_TRY(hgl_launch_gemm(bf.F, w.proj_w, w.proj_b, xcls, xcls, B, D, 4 * D, 4 * D, 4 * D, D, D, 1, 0, 0, 0, 0, HGL_ACT_NONE,
                          st));
  return HGL_OK;
}

// ln_post(rows) @ proj (+ R) on CLS rows that are already gathered (p.cls_rows)
int head_rows(const HglClipVisionW* w, int N, float* out, const float* R, const ClipPlan& p, hipStream_t st) {
  const int D = w->width, E = w->embed;
  HGL_TRY(hgl_launch_layernorm(p.cls_rows, w->ln_post_w, w->ln_post_b, p.cls_ln, N, D, 1e-5f, st));
  HGL_TRY(hgl_launch_gemm(p.cls_ln, w->proj_t, nullptr, R, out, N, E, D, D, D, E, E, 1, 0, 0, 0, 0, HGL_ACT_NONE, st));
  return HGL_OK;
}

bool valid_vision(const HglClipVisionW* w) {
  return w && w->width > 0 && w->layers > 0 && w->heads > 0 && w->patch > 0 && w->grid > 0 &&
         w->embed > 0 && w->width % w->heads == 0 && (w->width & 3) == 0 && (w->embed & 3) == 0 &&
         w->conv1_w && w->class_embedding && w->positional_embedding && w->ln_pre_w && w->ln_pre_b &&
         w->blocks && w->ln_post_w && w->ln_post_b && w->proj_t;
}

}  // namespace

extern "C" {

size_t hgl_clip_hybrid_workspace_bytes(const HglClipVisionW* w, int N, int Hm, int Wm, int fusion_mode) {
  (void)Hm; (void)Wm;
  if (!valid_vision(w) || N <= 0) return 0;
  HglArena ar(nullptr, 0);
  ClipPlan p;
  carve(ar, w, N, fusion_mode, p);
  return ar.off;
}

// The masks of one hybrid forward: n_seg runs of rows, run s = seg_n[s] masks of seg_h[s] x seg_w[s] pixels at seg_masks[s]
// (one run = the proposals of one image; a group of refs brings images of different sizes).
struct MaskSegs {
  const uint8_t* const* masks;
  const int *n, *h, *w;
  int n_seg;
};

static int hybrid_forward_impl(const HglClipVisionW* w, const float* local_imgs, const float* global_imgs,
                               const MaskSegs& segs, int N, int fusion_mode,
                               int masking_block, int last_layer, float* out, void* workspace,
                               size_t workspace_bytes, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(valid_vision(w), "clip_hybrid_forward: invalid weight struct");
  HGL_REQUIRE(local_imgs && out && N > 0, "clip_hybrid_forward: null input");
  HGL_REQUIRE(fusion_mode >= 0 && fusion_mode <= 5, "clip_hybrid_forward: bad fusion mode %d", fusion_mode);
  const bool two_stream = n_streams(fusion_mode) >= 2;
  HGL_REQUIRE(!two_stream || global_imgs, "clip_hybrid_forward: global_imgs required for this fusion mode");
  const bool need_masks = fusion_mode != HGL_FUSION_CROP;
  if (need_masks) {
    HGL_REQUIRE(segs.masks && segs.n && segs.h && segs.w && segs.n_seg > 0, "clip_hybrid_forward: masks required");
    long long rows = 0;
    for (int s = 0; s < segs.n_seg; ++s) {
      HGL_REQUIRE(segs.masks[s] && segs.n[s] > 0 && segs.h[s] > 0 && segs.w[s] > 0,
                  "clip_hybrid_forward: mask run %d is empty (n %d, %d x %d)", s, segs.n[s], segs.h[s], segs.w[s]);
      rows += segs.n[s];
    }
    HGL_REQUIRE(rows == N, "clip_hybrid_forward: the mask runs hold %lld masks for %d image rows", rows, N);
  }
  if (masking_block < 0) masking_block = last_layer;
  HGL_REQUIRE(last_layer >= 0 && last_layer + 1 < w->layers + 1, "clip_hybrid_forward: bad last_layer %d", last_layer);
  // the reference's return sits inside the per-block loop: the returning block must exist and be
  // at or after masking_block (model/backbone.py:178,197,220,254,294)
  const int ret_block = fusion_mode == HGL_FUSION_ATTN_MASKING ? last_layer : last_layer + 1;
  HGL_REQUIRE(fusion_mode == HGL_FUSION_CROP || (ret_block < w->layers && masking_block <= ret_block && masking_block >= 0),
              "clip_hybrid_forward: masking_block %d / return block %d outside the %d-layer transformer", masking_block, ret_block, w->layers);

  HglArena ar(workspace, workspace_bytes);
  ClipPlan p;
  if (!workspace || !carve(ar, w, N, fusion_mode, p)) {
    hgl_set_error("clip_hybrid_forward: workspace too small (%zu bytes given)", workspace_bytes);
    return HGL_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int D = w->width, g = w->grid, S = g * g + 1, heads = w->heads;
  const long long sN = (long long)N * S * D;  // elements per stream
  BlockBufs bf{p.H, p.QKV, p.F};

  // ---- patch embedding of the streams ----
  HGL_TRY(embed_images(w, local_imgs, N, p.X, p, st));
  if (two_stream) HGL_TRY(embed_images(w, global_imgs, N, p.X + sN, p, st));
  if (need_masks) {
    size_t r0 = 0;
    for (int s = 0; s < segs.n_seg; ++s) {
      HGL_TRY(hgl_launch_mask_resize(segs.masks[s], segs.n[s], segs.h[s], segs.w[s], g, p.pm + r0 * (S - 1), p.keep + r0 * (S - 1), st));
      r0 += segs.n[s];
    }
  }

  if (fusion_mode == HGL_FUSION_CROP) {
    for (int l = 0; l + 1 < w->layers; ++l)
      HGL_TRY(run_block(w->blocks[l], p.X, N, S, D, heads, bf, HGL_MASK_NONE, nullptr, 0, 0, st));
    HGL_TRY(run_block_cls(w->blocks[w->layers - 1], p.X, N, S, D, heads, bf, HGL_MASK_NONE, nullptr, 0, 0, p, st));
    return head_rows(w, N, out, nullptr, p, st);
  }

  const int nb0 = two_stream ? 2 * N : N;
  for (int l = 0; l < masking_block; ++l)
    HGL_TRY(run_block(w->blocks[l], p.X, nb0, S, D, heads, bf, HGL_MASK_NONE, nullptr, 0, 0, st));

  float* X = p.X;
  float* Y = p.Y;
  switch (fusion_mode) {
    case HGL_FUSION_TOKEN_MASKING: {
      for (int l = masking_block; l <= ret_block; ++l) {
        // x = cat(cls, x*pm)  (model/backbone.py:163-176)
        HGL_TRY(hgl_launch_mix(X, nullptr, 0.f, X, 1.f, p.pm, N, S, D, st));
        if (l < ret_block) HGL_TRY(run_block(w->blocks[l], X, N, S, D, heads, bf, HGL_MASK_NONE, nullptr, 0, 0, st));
        else HGL_TRY(run_block_cls(w->blocks[l], X, N, S, D, heads, bf, HGL_MASK_NONE, nullptr, 0, 0, p, st));
      }
      return head_rows(w, N, out, nullptr, p, st);
    }
    case HGL_FUSION_ATTN_MASKING: {
      for (int l = masking_block; l < ret_block; ++l)
        HGL_TRY(run_block(w->blocks[l], X, N, S, D, heads, bf, HGL_MASK_CLS_KEEP, p.keep, 0, N, st));
      HGL_TRY(run_block_cls(w->blocks[ret_block], X, N, S, D, heads, bf, HGL_MASK_CLS_KEEP, p.keep, 0, N, p, st));
      return head_rows(w, N, out, nullptr, p, st);
    }
    case HGL_FUSION_G2L: {
      // streams: [local | global].  local' = blk(local + 2*tokmask(global)); global' = blk(global, keep)
      for (int l = masking_block; l <= ret_block; ++l) {
        HGL_TRY(hgl_launch_mix(Y, X, 1.f, X + sN, 2.f, p.pm, N, S, D, st));
        if (l < ret_block) {
          (void)hipMemcpyAsync(Y + sN, X + sN, sizeof(float) * sN, hipMemcpyDeviceToDevice, st);
          HGL_TRY(run_block(w->blocks[l], Y, 2 * N, S, D, heads, bf, HGL_MASK_CLS_KEEP, p.keep, N, N, st));
        } else {  // the global stream of the returning block is dead, and so are the non-CLS rows of the local one
          HGL_TRY(run_block_cls(w->blocks[l], Y, N, S, D, heads, bf, HGL_MASK_NONE, nullptr, 0, 0, p, st));
        }
        float* t = X; X = Y; Y = t;
      }
      return head_rows(w, N, out, nullptr, p, st);
    }
    case HGL_FUSION_L2G: {
      // local' = blk(local); global' = blk(local + 2*global, keep)
      for (int l = masking_block; l <= ret_block; ++l) {
        HGL_TRY(hgl_launch_mix(Y + sN, X, 1.f, X + sN, 2.f, nullptr, N, S, D, st));
        if (l < ret_block) {
          (void)hipMemcpyAsync(Y, X, sizeof(float) * sN, hipMemcpyDeviceToDevice, st);
          HGL_TRY(run_block(w->blocks[l], Y, 2 * N, S, D, heads, bf, HGL_MASK_CLS_KEEP, p.keep, N, N, st));
        } else {  // the local stream of the returning block is dead, and so are the non-CLS rows of the global one
          HGL_TRY(run_block_cls(w->blocks[l], Y + sN, N, S, D, heads, bf, HGL_MASK_CLS_KEEP, p.keep, 0, N, p, st));
        }
        float* t = X; X = Y; Y = t;
      }
      return head_rows(w, N, out, nullptr, p, st);
    }
    case HGL_FUSION_G2L_L2G: {
      // stream order here: [xl | hl | xg | hg]  (keep applies to the last two)
      // init: xg currently at X+sN -> move to slot 2; hl = xl, hg = xg (model/backbone.py:272-276)
      (void)hipMemcpyAsync(X + 2 * sN, X + sN, sizeof(float) * sN, hipMemcpyDeviceToDevice, st);
      (void)hipMemcpyAsync(X + sN, X, sizeof(float) * sN, hipMemcpyDeviceToDevice, st);
      (void)hipMemcpyAsync(X + 3 * sN, X + 2 * sN, sizeof(float) * sN, hipMemcpyDeviceToDevice, st);
      for (int l = masking_block; l <= ret_block; ++l) {
        // hl_in = hl + 2*tokmask(xg) ; hg_in = xl + 2*hg   (pre-block values of xl, xg)
        HGL_TRY(hgl_launch_mix(Y + sN, X + sN, 1.f, X + 2 * sN, 2.f, p.pm, N, S, D, st));
        HGL_TRY(hgl_launch_mix(Y + 3 * sN, X, 1.f, X + 3 * sN, 2.f, nullptr, N, S, D, st));
        if (l < ret_block) {
          (void)hipMemcpyAsync(Y, X, sizeof(float) * sN, hipMemcpyDeviceToDevice, st);
          (void)hipMemcpyAsync(Y + 2 * sN, X + 2 * sN, sizeof(float) * sN, hipMemcpyDeviceToDevice, st);
          HGL_TRY(run_block(w->blocks[l], Y, 4 * N, S, D, heads, bf, HGL_MASK_CLS_KEEP, p.keep, 2 * N, N, st));
        } else {  // plain xl / xg streams of the returning block are dead; of hl / hg only the CLS rows are consumed
          HGL_TRY(run_block_cls(w->blocks[l], Y + sN, N, S, D, heads, bf, HGL_MASK_NONE, nullptr, 0, 0, p, st));
          HGL_TRY(head_rows(w, N, out, nullptr, p, st));
          HGL_TRY(run_block_cls(w->blocks[l], Y + 3 * sN, N, S, D, heads, bf, HGL_MASK_CLS_KEEP, p.keep, 0, N, p, st));
          // out = head(hl) + head(hg): the second projection accumulates onto the first
          return head_rows(w, N, out, out, p, st);
        }
        float* t = X; X = Y; Y = t;
      }
      break;   // not reached: the loop returns at the returning block
    }
  }
  hgl_set_error("clip_hybrid_forward: unreachable");
  return HGL_EINVAL;
}

int hgl_clip_hybrid_forward(const HglClipVisionW* w, const float* local_imgs, const float* global_imgs,
                            const uint8_t* masks, int N, int Hm, int Wm, int fusion_mode,
                            int masking_block, int last_layer, float* out, void* workspace,
                            size_t workspace_bytes, void* stream) {
  MaskSegs one{&masks, &N, &Hm, &Wm, masks ? 1 : 0};
  return hybrid_forward_impl(w, local_imgs, global_imgs, one, N, fusion_mode, masking_block, last_layer, out, workspace,
                             workspace_bytes, stream);
}

int hgl_clip_hybrid_forward_segments(const HglClipVisionW* w, const float* local_imgs, const float* global_imgs,
                                     const uint8_t* const* seg_masks, const int* seg_n, const int* seg_h, const int* seg_w,
                                     int n_seg, int N, int fusion_mode, int masking_block, int last_layer, float* out,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  MaskSegs segs{seg_masks, seg_n, seg_h, seg_w, n_seg};
  return hybrid_forward_impl(w, local_imgs, global_imgs, segs, N, fusion_mode, masking_block, last_layer, out, workspace,
                             workspace_bytes, stream);
}

static bool valid_text(const HglClipTextW* w) {
  return w && w->width > 0 && w->layers > 0 && w->heads > 0 && w->context > 0 && w->vocab > 0 &&
         w->embed > 0 && w->width % w->heads == 0 && (w->width & 3) == 0 && w->token_embedding &&
         w->positional_embedding && w->blocks && w->ln_final_w && w->ln_final_b && w->text_projection_t;
}

struct TextPlan {
  float *X, *H, *QKV, *F, *rows, *rows_ln;
  int32_t* eot;
};
static bool carve_text(HglArena& ar, const HglClipTextW* w, int B, TextPlan& p) {
  const size_t act = (size_t)B * w->context * w->width;
  p.X = ar.take<float>(act);
  p.H = ar.take<float>(act);
  p.QKV = ar.take<float>(3 * act);
  p.F = ar.take<float>(4 * act);
  p.rows = ar.take<float>((size_t)B * w->width);
  p.rows_ln = ar.take<float>((size_t)B * w->width);
  p.eot = ar.take<int32_t>(B);
  return ar.ok();
}

size_t hgl_clip_text_workspace_bytes(const HglClipTextW* w, int B) {
  if (!valid_text(w) || B <= 0) return 0;
  HglArena ar(nullptr, 0);
  TextPlan p;
  carve_text(ar, w, B, p);
  return ar.off;
}

// seq_len <= context: only the first seq_len positions of every string are computed.  Under the causal mask a
// position never sees later ones, so the pooled EOT feature is unchanged as long as every EOT lies inside the prefix
// (the caller's promise; an EOT beyond it yields NaN rows rather than a wrong feature).
int hgl_clip_encode_text_ex(const HglClipTextW* w, const int32_t* tokens, int B, int seq_len, const int32_t* pool_pos,
                            const int32_t* zero_pos, int n_zero, int masking_block, float* out,
                            void* workspace, size_t workspace_bytes, void* stream) {
  HGL_TRY(hgl_require_device());
  HGL_REQUIRE(valid_text(w), "clip_encode_text: invalid weight struct");
  HGL_REQUIRE(tokens && out && B > 0, "clip_encode_text: null input");
  HGL_REQUIRE(seq_len >= 1 && seq_len <= w->context, "clip_encode_text: prefix length %d outside 1..%d", seq_len, w->context);
  HGL_REQUIRE(n_zero >= 0 && (n_zero == 0 || zero_pos), "clip_encode_text: %d masked positions without a list", n_zero);
  HglArena ar(workspace, workspace_bytes);
  TextPlan p;
  if (!workspace || !carve_text(ar, w, B, p)) {
    hgl_set_error("clip_encode_text: workspace too small (%zu bytes given)", workspace_bytes);
    return HGL_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int D = w->width, S = seq_len;
  HGL_TRY(hgl_launch_text_embed(tokens, w->token_embedding, w->positional_embedding, p.X, B, S, w->context, D,
                                w->vocab, p.eot, st));
  if (pool_pos)   // clip/model.py:426-428: the pooled row is chosen by the caller
    if (hipMemcpyAsync(p.eot, pool_pos, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, st) != hipSuccess) {
      hgl_set_error("clip_encode_text: copying pool_pos failed");
      return HGL_ELAUNCH;
    }
  BlockBufs bf{p.H, p.QKV, p.F};
  for (int l = 0; l < w->layers; ++l) {
    if (n_zero > 0 && l >= masking_block) HGL_TRY(hgl_launch_zero_positions(p.X, B, S, D, zero_pos, n_zero, st));   // backbone.py:44-46
    HGL_TRY(run_block(w->blocks[l], p.X, B, S, D, w->heads, bf, HGL_MASK_CAUSAL, nullptr, 0, 0, st));
  }
  // ln_final is row-wise: normalise only the pooled rows, then project (clip/model.py:424-429)
  HGL_TRY(hgl_launch_gather_eot(p.X, p.eot, B, S, D, p.rows, st));
  HGL_TRY(hgl_launch_layernorm(p.rows, w->ln_final_w, w->ln_final_b, p.rows_ln, B, D, 1e-5f, st));
  HGL_TRY(hgl_launch_gemm(p.rows_ln, w->text_projection_t, nullptr, nullptr, out, B, w->embed, D, D, D,
                          0, w->embed, 1, 0, 0, 0, 0, HGL_ACT_NONE, st));
  return HGL_OK;
}

int hgl_clip_encode_text_prefix(const HglClipTextW* w, const int32_t* tokens, int B, int seq_len, float* out,
                                void* workspace, size_t workspace_bytes, void* stream) {
  return hgl_clip_encode_text_ex(w, tokens, B, seq_len, nullptr, nullptr, 0, 0, out, workspace, workspace_bytes, stream);
}

int hgl_clip_encode_text(const HglClipTextW* w, const int32_t* tokens, int B, float* out,
                         void* workspace, size_t workspace_bytes, void* stream) {
  return hgl_clip_encode_text_prefix(w, tokens, B, w ? w->context : 0, out, workspace, workspace_bytes, stream);
}

}  // extern "C"
