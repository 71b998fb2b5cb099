"""Drop-in for the segment-anything surface Hybridgl_main.py uses (Hybridgl_main.py:66-74,85):
`sam_model_registry[...]`, `SamAutomaticMaskGenerator(model, ...).generate(image)`.

The ViT-H image encoder, the prompt encoder, the two-way mask decoder and the fused mask
post-processing (upsampling, thresholds, stability score, boxes, NMS, connected-component clean-up)
and the Pillow-exact ResizeLongestSide resize run in libhybridgl.so (hand-written HIP).  Host work
kept here: the point grid, the two proposal counts read back, and the list-of-dict packaging.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib, ops, weights
from ._lib import check


from ._lib import (HglLinearW, HglNormW, HglSamAttnW, HglSamBlockW, HglSamDecoderW,  # noqa: E402
                   HglSamEncoderW)


def _dev(a, device):
    if isinstance(a, torch.Tensor):
        t = a.detach().to(torch.float32)
    else:
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return t.contiguous().to(device)


def load_sam_state_dict(path):
    """`torch.load` of the released checkpoint (build_sam.py:103-106)."""
    sd = torch.load(path, map_location="cpu")
    return {k: v.float().numpy() for k, v in sd.items()}


class Sam:
    """Weights of segment_anything.modeling.Sam on the device + the C weight structs."""

    mask_threshold = 0.0  # modeling/sam.py:19
    image_format = "RGB"

    def __init__(self, state_dict, cfg, device="cuda", precision=None):
        _lib.load()
        import weakref
        precision = precision or ops.default_precision()
        ops.use_precision(precision)
        self.precision = precision
        _before = set(ops._split_cache)
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        self._t = []  # keep every device tensor alive
        sd = state_dict
        D, L, H = cfg["embed_dim"], cfg["depth"], cfg["num_heads"]
        S, ps, Cc = cfg["img_size"], cfg["patch_size"], cfg["out_chans"]
        g = S // ps
        self.img_size, self.grid = S, g

        def t(a):
            x = _dev(a, self.device)
            self._t.append(x)
            return x.data_ptr()

        e = "image_encoder"
        self._blocks = (HglSamBlockW * L)()
        for i in range(L):
            b, p = self._blocks[i], f"{e}.blocks.{i}"
            b.window = 0 if i in cfg["global_attn_indexes"] else cfg["window_size"]
            b.rel_len = sd[f"{p}.attn.rel_pos_h"].shape[0]
            for f, k in [("norm1_w", "norm1.weight"), ("norm1_b", "norm1.bias"), ("qkv_w", "attn.qkv.weight"),
                         ("qkv_b", "attn.qkv.bias"), ("proj_w", "attn.proj.weight"), ("proj_b", "attn.proj.bias"),
                         ("rel_pos_h", "attn.rel_pos_h"), ("rel_pos_w", "attn.rel_pos_w"),
                         ("norm2_w", "norm2.weight"), ("norm2_b", "norm2.bias"), ("lin1_w", "mlp.lin1.weight"),
                         ("lin1_b", "mlp.lin1.bias"), ("lin2_w", "mlp.lin2.weight"), ("lin2_b", "mlp.lin2.bias")]:
                setattr(b, f, t(sd[f"{p}.{k}"]))
                if precision == "f16x3" and f in ("qkv_w", "proj_w", "lin1_w", "lin2_w"):
                    ops.register_split_weight(self._t[-1])
                if precision == "f16x3" and f in ("rel_pos_h", "rel_pos_w") and b.window == 14 and self._t[-1].shape == (27, 80):
                    # the windowed attention multiplies with these tables on the matrix cores: split once here (unscaled)
                    # instead of per wave and item in the kernel
                    ops.register_split_weight(self._t[-1], scale_log2=0)
        enc = HglSamEncoderW()
        enc.embed_dim, enc.depth, enc.heads, enc.img_size, enc.patch, enc.out_chans = D, L, H, S, ps, Cc
        enc.patch_w = t(np.asarray(sd[f"{e}.patch_embed.proj.weight"]).reshape(D, -1))
        enc.patch_b = t(sd[f"{e}.patch_embed.proj.bias"])
        enc.pos_embed = t(np.asarray(sd[f"{e}.pos_embed"]).reshape(g * g, D))
        enc.blocks = C.cast(self._blocks, C.POINTER(HglSamBlockW))
        enc.neck0_w = t(np.asarray(sd[f"{e}.neck.0.weight"]).reshape(Cc, D))
        enc.neck1_w, enc.neck1_b = t(sd[f"{e}.neck.1.weight"]), t(sd[f"{e}.neck.1.bias"])
        enc.neck2_w = t(np.asarray(sd[f"{e}.neck.2.weight"]).reshape(Cc, Cc * 9))
        enc.neck3_w, enc.neck3_b = t(sd[f"{e}.neck.3.weight"]), t(sd[f"{e}.neck.3.bias"])
        if precision == "f16x3":     # patch embedding and neck convolutions as split-fp16 GEMMs (K % 64 == 0 permitting)
            for x in self._t:
                if x.data_ptr() in (enc.patch_w, enc.neck0_w, enc.neck2_w) and x.shape[1] % 64 == 0:
                    ops.register_split_weight(x)
        self.enc_w = enc

        m, pe = "mask_decoder", "prompt_encoder"
        dec = HglSamDecoderW()
        dec.C, dec.grid, dec.heads, dec.mlp_dim = Cc, g, 8, sd[f"{m}.transformer.layers.0.mlp.lin1.weight"].shape[0]
        dec.pe_gauss = t(sd[f"{pe}.pe_layer.positional_encoding_gaussian_matrix"])
        dec.point_embed_pos = t(np.asarray(sd[f"{pe}.point_embeddings.1.weight"]).reshape(-1))
        dec.not_a_point = t(np.asarray(sd[f"{pe}.not_a_point_embed.weight"]).reshape(-1))
        dec.point_embed_neg = t(np.asarray(sd[f"{pe}.point_embeddings.0.weight"]).reshape(-1))
        dec.point_embed_box0 = t(np.asarray(sd[f"{pe}.point_embeddings.2.weight"]).reshape(-1))
        dec.point_embed_box1 = t(np.asarray(sd[f"{pe}.point_embeddings.3.weight"]).reshape(-1))
        if f"{pe}.mask_downscaling.0.weight" in sd and tuple(np.asarray(sd[f"{pe}.mask_downscaling.3.weight"]).shape) == (16, 4, 2, 2):
            md = f"{pe}.mask_downscaling"                      # prompt_encoder.py:57-66 (mask_in_chans = 16)
            for dst, key, shape in (("md_c1_w", "0.weight", (4, 4)), ("md_c1_b", "0.bias", (4,)), ("md_n1_w", "1.weight", (4,)),
                                    ("md_n1_b", "1.bias", (4,)), ("md_c2_w", "3.weight", (16, 16)), ("md_c2_b", "3.bias", (16,)),
                                    ("md_n2_w", "4.weight", (16,)), ("md_n2_b", "4.bias", (16,)),
                                    ("md_c3_w", "6.weight", (Cc, 16)), ("md_c3_b", "6.bias", (Cc,))):
                setattr(dec, dst, t(np.asarray(sd[f"{md}.{key}"]).reshape(shape)))
        dec.no_mask = t(np.asarray(sd[f"{pe}.no_mask_embed.weight"]).reshape(-1))
        dec.iou_token = t(np.asarray(sd[f"{m}.iou_token.weight"]).reshape(-1))
        dec.mask_tokens = t(sd[f"{m}.mask_tokens.weight"])

        def lin(dst, key):
            dst.w, dst.b = t(sd[f"{key}.weight"]), t(sd[f"{key}.bias"])
            wshape = np.asarray(sd[f"{key}.weight"]).shape
            if precision == "f16x3" and len(wshape) == 2 and wshape[1] % 16 == 0:
                ops.register_split_weight(self._t[-2])     # small-M GEMMs of the decoder (token MLPs, hyper-nets, IoU head)

        def attn(dst, key):
            for nm in ("q", "k", "v", "out"):
                lin(getattr(dst, nm), f"{key}.{nm}_proj")
            dst.internal = sd[f"{key}.q_proj.weight"].shape[0]

        for i in range(2):
            l, lay = f"{m}.transformer.layers.{i}", dec.layer[i]
            attn(lay.self_attn, f"{l}.self_attn")
            attn(lay.t2i, f"{l}.cross_attn_token_to_image")
            attn(lay.i2t, f"{l}.cross_attn_image_to_token")
            for j, nm in enumerate(("n1", "n2", "n3", "n4")):
                lin(getattr(lay, nm), f"{l}.norm{j + 1}")
            lin(lay.lin1, f"{l}.mlp.lin1")
            lin(lay.lin2, f"{l}.mlp.lin2")
        attn(dec.final_t2i, f"{m}.transformer.final_attn_token_to_image")
        lin(dec.norm_final, f"{m}.transformer.norm_final_attn")
        # ConvTranspose2d(k=2,s=2) weights [Cin,Cout,2,2] -> GEMM rows ordered (ky,kx,cout)
        w0 = np.asarray(sd[f"{m}.output_upscaling.0.weight"])
        dec.up0_w = t(np.transpose(w0, (2, 3, 1, 0)).reshape(-1, w0.shape[0]))
        if precision == "f16x3":
            ops.register_split_weight(self._t[-1])
        dec.up0_b = t(np.tile(np.asarray(sd[f"{m}.output_upscaling.0.bias"]), 4))
        lin(dec.up1, f"{m}.output_upscaling.1")
        w3 = np.asarray(sd[f"{m}.output_upscaling.3.weight"])
        dec.up3_w = t(np.transpose(w3, (2, 3, 1, 0)).reshape(-1, w3.shape[0]))
        if precision == "f16x3":
            ops.register_split_weight(self._t[-1])
        dec.up3_b = t(np.tile(np.asarray(sd[f"{m}.output_upscaling.3.bias"]), 4))
        for i in range(4):
            for j in range(3):
                lin(dec.hyper[i][j], f"{m}.output_hypernetworks_mlps.{i}.layers.{j}")
        for j in range(3):
            lin(dec.iou_head[j], f"{m}.iou_prediction_head.layers.{j}")
        # dense positional encoding of the embedding grid, once (prompt_encoder.py:194-205)
        ax = ((np.arange(g, dtype=np.float32) + np.float32(1)) - np.float32(0.5)) / np.float32(g)
        coords = np.stack(np.broadcast_arrays(ax[None, :], ax[:, None]), axis=-1).reshape(-1, 2)
        self.dense_pe = torch.empty((g * g, Cc), dtype=torch.float32, device=self.device)
        dec.dense_pe = self.dense_pe.data_ptr()
        cd = _dev(coords, self.device)
        check(_lib.load().hgl_sam_dense_pe(C.byref(dec), cd.data_ptr(), self.dense_pe.data_ptr(), ops._stream()),
              "hgl_sam_dense_pe")
        torch.cuda.current_stream().synchronize()
        if precision == "f16x3" and Cc == 256:
            # merged image-side projections of the decoder (HglSamDecoderW.kvq1 / kvf): concatenated weights, and the
            # positional-encoding part of k and q as per-position tables: (keys + pe) W^T = keys W^T + pe W^T
            def cat_w(keys):
                w = torch.cat([_dev(sd[f"{k}.weight"], self.device) for k in keys], 0).contiguous()
                b = torch.cat([_dev(sd[f"{k}.bias"], self.device) for k in keys], 0).contiguous()
                self._t += [w, b]
                ops.register_split_weight(w)
                return w, b

            def pe_table(keys, with_pe):
                cols = []
                for k, use in zip(keys, with_pe):
                    wk = _dev(sd[f"{k}.weight"], self.device)
                    cols.append(ops.gemm(self.dense_pe, wk) if use else
                                torch.zeros((self.dense_pe.shape[0], wk.shape[0]), dtype=torch.float32, device=self.device))
                tab = torch.cat(cols, 1).contiguous()
                self._t.append(tab)
                return tab
            l1 = f"{m}.transformer.layers.1"
            k1 = [f"{l1}.cross_attn_token_to_image.k_proj", f"{l1}.cross_attn_token_to_image.v_proj",
                  f"{l1}.cross_attn_image_to_token.q_proj"]
            kf = [f"{m}.transformer.final_attn_token_to_image.k_proj", f"{m}.transformer.final_attn_token_to_image.v_proj"]
            w1, b1 = cat_w(k1)
            wf, bf = cat_w(kf)
            dec.kvq1_w, dec.kvq1_b, dec.kvq1_pe = w1.data_ptr(), b1.data_ptr(), pe_table(k1, (True, False, True)).data_ptr()
            dec.kvf_w, dec.kvf_b, dec.kvf_pe = wf.data_ptr(), bf.data_ptr(), pe_table(kf, (True, False)).data_ptr()
            # the token -> image attention on the raw image tokens (csrc/sam_decoder_t2i.hip) multiplies its folded queries with
            # the positional encoding as a GEMM "weight" [HW, C]
            ops.register_split_weight(self.dense_pe)
            torch.cuda.current_stream().synchronize()
        self.dec_w = dec
        # the fp16 splits registered above die with this model (library registry + hi/lo tensors)
        self._split_keys = ops.split_weight_keys_since(_before)
        weakref.finalize(self, ops.release_split_weights, list(self._split_keys))

    def to(self, device):
        if torch.device(device).type != "cuda":
            raise _lib.HybridGLError("Sam runs on the GPU only (no CPU path exists)")
        return self

    def eval(self):
        return self

    # ---- the three device stages -------------------------------------------------------
    def encode(self, resized_u8):
        """resized_u8: [h,w,3] uint8 device tensor (long side == img_size) -> emb [g*g, C]."""
        lib = _lib.load()
        ops.use_precision(self.precision)
        h, w = resized_u8.shape[:2]
        need = lib.hgl_sam_encode_workspace_bytes(C.byref(self.enc_w))
        ws = ops.workspace(need, self.device, "sam_encode")
        emb = torch.empty((self.grid * self.grid, self.cfg["out_chans"]), dtype=torch.float32, device=self.device)
        check(lib.hgl_sam_encode(C.byref(self.enc_w), ops._dev(resized_u8, torch.uint8, "resized_img"), h, w,
                                 emb.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()), "hgl_sam_encode")
        return emb

    def encode_batch(self, resized_list):
        """several images through the encoder at once (token rows stacked: weights read once, better-filled GEMMs).
        resized_list: [h_i,w_i,3] uint8 device tensors -> emb [nb, g*g, C]; each slice equals encode() of that image
        up to the summation order of split-K."""
        lib = _lib.load()
        ops.use_precision(self.precision)
        nb = len(resized_list)
        imgs = [r.contiguous() for r in resized_list]
        ptrs = (C.c_void_p * nb)(*[ops._dev(r, torch.uint8, "resized_img") for r in imgs])
        hs = (C.c_int * nb)(*[int(r.shape[0]) for r in imgs])
        wsz = (C.c_int * nb)(*[int(r.shape[1]) for r in imgs])
        need = lib.hgl_sam_encode_batch_workspace_bytes(C.byref(self.enc_w), nb)
        ws = ops.workspace(need, self.device, "sam_encode")
        emb = torch.empty((nb, self.grid * self.grid, self.cfg["out_chans"]), dtype=torch.float32, device=self.device)
        check(lib.hgl_sam_encode_batch(C.byref(self.enc_w), ptrs, hs, wsz, nb, emb.data_ptr(), ws.data_ptr(), ws.numel(),
                                       ops._stream()), "hgl_sam_encode_batch")
        return emb

    def decode_points(self, emb, points01, iou_gate=None):
        """points01: [P,2] fp32 device ((point+0.5)/img_size) -> (low_res [P,3,4g,4g], iou [P,3]).
        iou_gate: a pred_iou_thresh the caller will filter with (automatic_mask_generator.py:287-291): prompts none of whose
        three predictions exceeds it skip the output upscaling -- their rows of low_res are unwritten memory, which the caller's
        filter never reads (hgl_sam_decode_points_gated)."""
        lib = _lib.load()
        ops.use_precision(self.precision)
        P = points01.shape[0]
        need = lib.hgl_sam_decode_workspace_bytes(C.byref(self.dec_w), P)
        ws = ops.workspace(need, self.device, "sam_decode")
        g4 = 4 * self.grid
        low = torch.empty((P, 3, g4, g4), dtype=torch.float32, device=self.device)
        iou = torch.empty((P, 3), dtype=torch.float32, device=self.device)
        if iou_gate is not None:
            check(lib.hgl_sam_decode_points_gated(C.byref(self.dec_w), ops._dev(emb, torch.float32, "emb"),
                                                  ops._dev(points01, torch.float32, "points01"), P, float(iou_gate), low.data_ptr(),
                                                  iou.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()),
                  "hgl_sam_decode_points_gated")
            return low, iou
        check(lib.hgl_sam_decode_points(C.byref(self.dec_w), ops._dev(emb, torch.float32, "emb"),
                                        ops._dev(points01, torch.float32, "points01"), P, low.data_ptr(),
                                        iou.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()),
              "hgl_sam_decode_points")
        return low, iou

    def decode_prompts(self, emb, coords01, labels, first_mask=1, dense=None):
        """prompts of two or three sparse tokens (prompt_encoder.py:73-101): coords01 [P,n,2] fp32 device
        ((coordinate + 0.5) / img_size), labels [P,n] int32 (-1 padding, 0 / 1 background / foreground point, 2 / 3 box
        corners); dense: None or [P, g*g, C] from embed_masks; first_mask 1 -> mask tokens 1..3 (multimask), 0 -> tokens 0..2
        (column 0 = the single-mask output) -> (low_res [P,3,4g,4g], iou [P,3])."""
        lib = _lib.load()
        ops.use_precision(self.precision)
        P, n = int(coords01.shape[0]), int(coords01.shape[1])
        need = lib.hgl_sam_decode_workspace_bytes(C.byref(self.dec_w), P)
        ws = ops.workspace(need, self.device, "sam_decode")
        g4 = 4 * self.grid
        low = torch.empty((P, 3, g4, g4), dtype=torch.float32, device=self.device)
        iou = torch.empty((P, 3), dtype=torch.float32, device=self.device)
        check(lib.hgl_sam_decode_prompts(C.byref(self.dec_w), ops._dev(emb, torch.float32, "emb"),
                                         ops._dev(coords01, torch.float32, "coords01"), ops._dev(labels, torch.int32, "labels"), n,
                                         None if dense is None else ops._dev(dense, torch.float32, "dense"),
                                         int(first_mask), P, low.data_ptr(), iou.data_ptr(), ws.data_ptr(), ws.numel(),
                                         ops._stream()), "hgl_sam_decode_prompts")
        return low, iou

    def embed_masks(self, mask_input):
        """PromptEncoder._embed_masks (prompt_encoder.py:103-106): [P,1,4g,4g] fp32 device -> dense rows [P, g*g, C]"""
        lib = _lib.load()
        P = int(mask_input.shape[0])
        g4 = 4 * self.grid
        if tuple(mask_input.shape) != (P, 1, g4, g4):
            raise ValueError(f"mask_input must be [P,1,{g4},{g4}], got {tuple(mask_input.shape)}")
        dense = torch.empty((P, self.grid * self.grid, self.dec_w.C), dtype=torch.float32, device=self.device)
        check(lib.hgl_sam_embed_masks(C.byref(self.dec_w), ops._dev(mask_input, torch.float32, "mask_input"), P, dense.data_ptr(),
                                      ops._stream()), "hgl_sam_embed_masks")
        return dense

    def postprocess(self, low_res, iou_pred, input_size, original_size, pred_iou_thresh=-1e30,
                    stability_thresh=0.0, stability_offset=1.0, return_logits=False):
        """low_res [K,hl,wl], iou_pred [K] -> masks [K,H,W] u8, boxes XYXY [K,4] i32, stability [K], keep [K]."""
        lib = _lib.load()
        K, hl, wl = low_res.shape
        H, W = original_size
        dev = self.device
        masks = torch.empty((K, H, W), dtype=torch.uint8, device=dev)
        boxes = torch.empty((K, 4), dtype=torch.int32, device=dev)
        stab = torch.empty((K,), dtype=torch.float32, device=dev)
        keep = torch.empty((K,), dtype=torch.uint8, device=dev)
        full = torch.empty((K, H, W), dtype=torch.float32, device=dev) if return_logits else None
        need = lib.hgl_sam_postprocess_workspace_bytes(K)
        ws = ops.workspace(need, dev, "sam_post")
        check(lib.hgl_sam_postprocess(ops._dev(low_res, torch.float32, "low_res"),
                                      ops._dev(iou_pred, torch.float32, "iou_pred"), K, hl, wl, self.img_size,
                                      int(input_size[0]), int(input_size[1]), H, W, float(self.mask_threshold),
                                      float(stability_offset), float(pred_iou_thresh), float(stability_thresh),
                                      masks.data_ptr(), boxes.data_ptr(), stab.data_ptr(), keep.data_ptr(),
                                      full.data_ptr() if full is not None else None, ws.data_ptr(), ws.numel(),
                                      ops._stream()), "hgl_sam_postprocess")
        return masks, boxes, stab, keep, full


def nms(boxes_xyxy, scores, keep, iou_threshold):
    """Device NMS -> (idx [K] int32, n [1] int32), both on the device.  K <= 1024: one workgroup; larger K
    (dense grids, crop layers): the three-pass bit-matrix kernels, same semantics."""
    lib = _lib.load()
    K = boxes_xyxy.shape[0]
    idx = torch.empty((K,), dtype=torch.int32, device=boxes_xyxy.device)
    n = torch.empty((1,), dtype=torch.int32, device=boxes_xyxy.device)
    if K <= 1024:
        check(lib.hgl_nms(ops._dev(boxes_xyxy, torch.int32, "boxes"), ops._dev(scores, torch.float32, "scores"),
                          ops._dev(keep, torch.uint8, "keep"), K, float(iou_threshold), idx.data_ptr(), n.data_ptr(),
                          ops._stream()), "hgl_nms")
    else:
        need = lib.hgl_nms_large_workspace_bytes(K)
        ws = ops.workspace(need, boxes_xyxy.device, "nms_large")
        check(lib.hgl_nms_large(ops._dev(boxes_xyxy, torch.int32, "boxes"), ops._dev(scores, torch.float32, "scores"),
                                ops._dev(keep, torch.uint8, "keep"), K, float(iou_threshold), idx.data_ptr(),
                                n.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()), "hgl_nms_large")
    return idx, n


def nms_large(boxes_xyxy, scores, keep, iou_threshold):
    """hgl_nms_large regardless of K (tests)."""
    lib = _lib.load()
    K = boxes_xyxy.shape[0]
    idx = torch.empty((K,), dtype=torch.int32, device=boxes_xyxy.device)
    n = torch.empty((1,), dtype=torch.int32, device=boxes_xyxy.device)
    ws = ops.workspace(lib.hgl_nms_large_workspace_bytes(K), boxes_xyxy.device, "nms_large")
    check(lib.hgl_nms_large(ops._dev(boxes_xyxy, torch.int32, "boxes"), ops._dev(scores, torch.float32, "scores"),
                            ops._dev(keep, torch.uint8, "keep"), K, float(iou_threshold), idx.data_ptr(), n.data_ptr(),
                            ws.data_ptr(), ws.numel(), ops._stream()), "hgl_nms_large")
    return idx, n


def box_near_crop_edge(boxes_xyxy, keep, crop_box, orig_box, atol=20.0):
    """keep[i] = 0 where box i (crop coordinates) touches a crop edge that is not an image edge (amg.py:78-88)."""
    import ctypes as C
    lib = _lib.load()
    cb = (C.c_int32 * 4)(*[int(v) for v in crop_box])
    ob = (C.c_int32 * 4)(*[int(v) for v in orig_box])
    check(lib.hgl_box_near_crop_edge(ops._dev(boxes_xyxy, torch.int32, "boxes"), boxes_xyxy.shape[0], cb, ob, float(atol),
                                     ops._dev(keep, torch.uint8, "keep"), ops._stream()), "hgl_box_near_crop_edge")
    return keep


def _build(cfg_name, checkpoint=None, state_dict=None, seed=0, device="cuda", precision=None):
    checkpoint = checkpoint or os.environ.get("HYBRIDGL_SAM_CHECKPOINT")
    if state_dict is None and checkpoint:
        state_dict = load_sam_state_dict(checkpoint)
    if state_dict is None:
        state_dict = weights.sam_state_dict(cfg_name, seed)
    return Sam(state_dict, weights.SAM_CONFIGS[cfg_name], device, precision)


def build_sam_vit_h(checkpoint=None, **kw):
    """build_sam.py:14-21."""
    return _build("vit_h", checkpoint, **kw)


# build_sam.py:47-52 (Hybridgl_main.py:66 uses 'default' = vit_h)
sam_model_registry = {"default": build_sam_vit_h, "vit_h": build_sam_vit_h,
                      "vit_l": lambda checkpoint=None, **kw: _build("vit_l", checkpoint, **kw),
                      "vit_b": lambda checkpoint=None, **kw: _build("vit_b", checkpoint, **kw),
                      "tiny": lambda checkpoint=None, **kw: _build("tiny", checkpoint, **kw)}


_PIL_PB = 22
_coeff_cache = {}


def pil_bilinear_coeffs(in_size, out_size):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter (host, double precision):
    int32 weights [out, ksize] and (first, count) bounds."""
    import math
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = 1.0 * fs
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        cnt = min(int(center + support + 0.5), in_size) - xmin
        w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) / fs)) for x in range(cnt)]
        ww = sum(w)
        for x in range(cnt):
            kk[xx, x] = int(0.5 + (w[x] / ww if ww != 0.0 else w[x]) * (1 << _PIL_PB))
        bounds[xx] = (xmin, cnt)
    return kk, bounds


def resize_longest_side(img_u8, long_side):
    """ResizeLongestSide.apply_image on the device, bit-exact with Pillow: img_u8 [H,W,3] uint8 device tensor."""
    lib = _lib.load()
    H, W, Cc = img_u8.shape
    nh, nw = get_preprocess_shape(H, W, long_side)
    key = (H, W, nh, nw, str(img_u8.device))
    tabs = _coeff_cache.get(key)
    if tabs is None:
        kx, bx = pil_bilinear_coeffs(W, nw)
        ky, by = pil_bilinear_coeffs(H, nh)
        tabs = tuple(torch.from_numpy(a).to(img_u8.device) for a in (kx, bx, ky, by))
        _coeff_cache[key] = tabs
    kx, bx, ky, by = tabs
    out = torch.empty((nh, nw, Cc), dtype=torch.uint8, device=img_u8.device)
    need = lib.hgl_resize_pil_bilinear_workspace_bytes(H, nw, Cc)
    ws = ops.workspace(need, img_u8.device, "pil_resize")
    check(lib.hgl_resize_pil_bilinear(ops._dev(img_u8, torch.uint8, "img"), H, W, Cc, nh, nw, kx.data_ptr(), bx.data_ptr(),
                                      kx.shape[1], ky.data_ptr(), by.data_ptr(), ky.shape[1], out.data_ptr(),
                                      ws.data_ptr(), ws.numel(), ops._stream()), "hgl_resize_pil_bilinear")
    return out


def build_point_grid(n):
    """utils/amg.py:179-186."""
    off = 1 / (2 * n)
    p = np.linspace(off, 1 - off, n)
    return np.stack([np.tile(p[None, :], (n, 1)), np.tile(p[:, None], (1, n))], axis=-1).reshape(-1, 2)


def build_all_layer_point_grids(n_per_side, n_layers, scale_per_layer):
    """utils/amg.py:189-198."""
    return [build_point_grid(int(n_per_side / (scale_per_layer ** i))) for i in range(n_layers + 1)]


def generate_crop_boxes(im_size, n_layers, overlap_ratio):
    """Crop windows of the crop-layer generator (what utils/amg.py:201-238 produces; pinned by the known answers in
    tests/golden/sam_crops.npz) -> (XYXY boxes, layer index of each).  Layer 0 is the whole image; layer L tiles the
    image with a 2^L x 2^L grid of equal windows that overlap by int(overlap_ratio * short_side * 2 / 2^L) pixels,
    window size = ceil((overlap * (n - 1) + side) / n), origins = int((size - overlap) * k), enumerated x-major,
    clipped to the image."""
    import math
    height, width = im_size
    boxes, layers = [[0, 0, width, height]], [0]
    for layer in range(1, n_layers + 1):
        n = 1 << layer
        overlap = int(overlap_ratio * min(height, width) * (2 / n))
        win_w = int(math.ceil((overlap * (n - 1) + width) / n))
        win_h = int(math.ceil((overlap * (n - 1) + height) / n))
        for kx in range(n):
            x0 = int((win_w - overlap) * kx)
            for ky in range(n):
                y0 = int((win_h - overlap) * ky)
                boxes.append([x0, y0, min(x0 + win_w, width), min(y0 + win_h, height)])
                layers.append(layer)
    return boxes, layers


def get_preprocess_shape(oldh, oldw, long_side):
    """utils/transforms.py:93-102."""
    scale = long_side * 1.0 / max(oldh, oldw)
    return int(oldh * scale + 0.5), int(oldw * scale + 0.5)


class ResizeLongestSide:
    """utils/transforms.py:16-102 (the pieces the predictor uses)."""

    def __init__(self, target_length):
        self.target_length = target_length

    def apply_image(self, image):
        dev = image if isinstance(image, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(image)).cuda()
        return resize_longest_side(dev.contiguous(), self.target_length)

    def apply_coords(self, coords, original_size):
        """utils/transforms.py:33-45 (float64, as numpy)."""
        old_h, old_w = original_size
        new_h, new_w = get_preprocess_shape(old_h, old_w, self.target_length)
        c = np.array(coords, dtype=np.float64, copy=True)
        c[..., 0] = c[..., 0] * (new_w / old_w)
        c[..., 1] = c[..., 1] * (new_h / old_h)
        return c

    def apply_boxes(self, boxes, original_size):
        """utils/transforms.py:47-53: [B,4] XYXY -> the resized frame"""
        return self.apply_coords(np.asarray(boxes, dtype=np.float64).reshape(-1, 2, 2), original_size).reshape(-1, 4)


class SamPredictor:
    """predictor.py:17-269: set_image, predict_torch, predict.  Points (foreground / background; the padding point follows
    them), a box (its two corners), or points and a box -- up to eleven sparse tokens per prompt --, optionally a mask
    input per prompt (per-prompt dense embeddings); multimask_output True or False.  One point per prompt (what
    automatic_mask_generator.py:269-285 issues) takes the fused decoder stages."""

    def __init__(self, sam_model):
        self.model = sam_model
        self.transform = ResizeLongestSide(sam_model.img_size)
        self.reset_image()

    @property
    def device(self):
        return self.model.device

    def reset_image(self):
        self.is_image_set = False
        self.features = None
        self.original_size = self.input_size = None

    def set_image(self, image, image_format="RGB"):
        assert image_format in ("RGB", "BGR"), f"image_format must be in ['RGB', 'BGR'], is {image_format}."
        img = image if isinstance(image, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(image)).to(self.device)
        if image_format != self.model.image_format:
            img = img.flip(-1)
        self.reset_image()
        self.original_size = tuple(int(v) for v in img.shape[:2])
        resized = resize_longest_side(img.contiguous(), self.model.img_size)
        self.input_size = tuple(int(v) for v in resized.shape[:2])
        self.features = self.model.encode(resized)
        self.is_image_set = True

    def _coords01(self, xy):
        """(coordinate + 0.5) / img_size in the dtype the caller handed over (prompt_encoder.py:79,95,207-214: float64 from
        the automatic generator, float32 from predict()), then float32"""
        xy = torch.as_tensor(xy, device=self.device)
        xy = xy if xy.dtype == torch.float64 else xy.to(torch.float32)
        return ((xy + 0.5) / float(self.model.img_size)).to(torch.float32)

    def predict_torch(self, point_coords, point_labels, boxes=None, mask_input=None, multimask_output=True,
                      return_logits=False):
        """predictor.py:169-243.  point_coords [P,N,2] in the resized frame (transform.apply_coords) with point_labels [P,N]
        in {0, 1}, and / or boxes [P,4] XYXY in the resized frame (transform.apply_boxes), and / or mask_input [P,1,4g,4g]
        (low-resolution logits of an earlier call).  Up to eleven sparse tokens per prompt: points (the padding point
        follows them when there is no box), a box, or points and a box.  Returns (masks [P,C,H,W] bool or logits,
        iou_predictions [P,C], low_res_masks [P,C,4g,4g]) with C = 3 (multimask_output) or 1."""
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")   # predictor.py:214
        toks, labs = [], []
        P = None
        if point_coords is not None:
            pc = torch.as_tensor(point_coords, device=self.device)
            pl = torch.as_tensor(point_labels, device=self.device)
            if pc.dim() != 3 or pc.shape[2] != 2 or pl.shape != pc.shape[:2]:
                raise ValueError(f"point_coords must be [P,N,2] with point_labels [P,N], got {tuple(pc.shape)} / {tuple(pl.shape)}")
            if not bool(((pl == 0) | (pl == 1) | (pl == -1)).all()):
                raise ValueError("point_labels must be 1 (foreground), 0 (background) or -1 (padding)")
            P = int(pc.shape[0])
            toks.append(self._coords01(pc))
            labs.append(pl.to(torch.int32))
            if boxes is None:                                                            # prompt_encoder.py:80-84: padding point
                toks.append(torch.zeros((P, 1, 2), dtype=torch.float32, device=self.device))
                labs.append(torch.full((P, 1), -1, dtype=torch.int32, device=self.device))
        if boxes is not None:
            b = torch.as_tensor(boxes, device=self.device)
            if b.dim() != 2 or b.shape[1] != 4 or (P is not None and b.shape[0] != P):
                raise ValueError(f"boxes must be [P,4] (XYXY), got {tuple(b.shape)}")
            P = int(b.shape[0])
            toks.append(self._coords01(b.reshape(-1, 2, 2)))                              # prompt_encoder.py:93-101
            labs.append(torch.tensor([2, 3], dtype=torch.int32, device=self.device).repeat(P, 1))
        if P is None:
            raise NotImplementedError("a prompt needs points and / or a box (a mask input alone has no sparse tokens)")
        c01, labels = torch.cat(toks, dim=1).contiguous(), torch.cat(labs, dim=1).contiguous()
        if c01.shape[1] > 11:
            raise NotImplementedError(f"{c01.shape[1]} sparse tokens per prompt: up to eleven are supported (ten points, or nine "
                                      "points and a box)")
        dense = None
        if mask_input is not None:
            mi = torch.as_tensor(mask_input, device=self.device).to(torch.float32)
            if mi.shape[0] != P:
                raise ValueError(f"mask_input must have one mask per prompt ({P}), got {tuple(mi.shape)}")
            dense = self.model.embed_masks(mi.contiguous())
        fast = (dense is None and boxes is None and multimask_output and c01.shape[1] == 2 and bool((labels[:, 0] == 1).all()))
        if fast:                                                                          # the automatic generator's prompts
            low, iou = self.model.decode_points(self.features, c01[:, 0, :].contiguous())
        else:
            low, iou = self.model.decode_prompts(self.features, c01, labels, first_mask=1 if multimask_output else 0, dense=dense)
            if not multimask_output:
                low, iou = low[:, :1].contiguous(), iou[:, :1].contiguous()
        Cm = low.shape[1]
        H, W = self.original_size
        _, _, _, _, full = self.model.postprocess(low.flatten(0, 1), iou.flatten(), self.input_size, (H, W), -1e30, 0.0, 1.0,
                                                  return_logits=True)
        full = full.reshape(P, Cm, H, W)
        masks = full if return_logits else full > self.model.mask_threshold
        return masks, iou, low

    def predict(self, point_coords=None, point_labels=None, box=None, mask_input=None, multimask_output=True,
                return_logits=False):
        """predictor.py:90-167 for one prompt: points [N,2] with labels [N], a box [4] (XYXY), both in the original frame,
        mask_input [1,4g,4g]: numpy in, numpy out ([C,H,W], [C], [C,4g,4g])."""
        pc = pl = bx = mi = None
        if point_coords is not None:
            assert point_labels is not None, "point_labels must be supplied if point_coords is supplied."
            pts = self.transform.apply_coords(np.asarray(point_coords, dtype=np.float64), self.original_size)
            pc = torch.as_tensor(pts, dtype=torch.float32)[None, :, :]                   # predictor.py:141-143: float32
            pl = torch.as_tensor(np.asarray(point_labels), dtype=torch.int32)[None, :]
        if box is not None:
            bx = torch.as_tensor(self.transform.apply_boxes(np.asarray(box, dtype=np.float64), self.original_size),
                                 dtype=torch.float32).reshape(1, 4)
        if mask_input is not None:
            mi = torch.as_tensor(np.asarray(mask_input), dtype=torch.float32)[None, :, :, :]
        m, iou, low = self.predict_torch(pc, pl, bx, mi, multimask_output=multimask_output, return_logits=return_logits)
        return m[0].cpu().numpy(), iou[0].cpu().numpy(), low[0].cpu().numpy()


class _GroupState:
    """a group of images on its way through SamAutomaticMaskGenerator.group_begin / group_cleanup / group_finish"""
    __slots__ = ("sizes", "cap", "cand", "n1_dev", "n1", "ev1", "n1_list", "stage", "n2", "ev2", "ovf", "overflow")


class SamAutomaticMaskGenerator:
    """automatic_mask_generator.py:35-372: the Hybridgl_main.py:67-73 configuration (one crop, 8x8 points) runs
    entirely on the device with two host syncs; crop layers / dense grids (Hybridgl_main_PhraseCut.py) add one
    sync per crop (its survivor count) and the cross-crop NMS."""

    def __init__(self, model, points_per_side=32, points_per_batch=64, pred_iou_thresh=0.88,
                 stability_score_thresh=0.95, stability_score_offset=1.0, box_nms_thresh=0.7, crop_n_layers=0,
                 crop_nms_thresh=0.7, crop_overlap_ratio=512 / 1500, crop_n_points_downscale_factor=1,
                 point_grids=None, min_mask_region_area=0, output_mode="binary_mask"):
        assert (points_per_side is None) != (point_grids is None), \
            "Exactly one of points_per_side or point_grid must be provided."
        assert output_mode in ["binary_mask", "uncompressed_rle", "coco_rle"], f"Unknown output_mode {output_mode}."
        self.output_mode = output_mode      # automatic_mask_generator.py:103-112 (coco_rle needs no pycocotools here: native codec)
        self.model = model
        self.predictor = SamPredictor(model)
        if point_grids is None:
            self.point_grids = build_all_layer_point_grids(points_per_side, crop_n_layers, crop_n_points_downscale_factor)
        else:
            self.point_grids = point_grids
        assert len(self.point_grids) >= crop_n_layers + 1, "one point grid per crop layer"
        self.crop_n_layers = crop_n_layers
        self.crop_overlap_ratio = crop_overlap_ratio
        self.points_per_batch = points_per_batch
        self.pred_iou_thresh = pred_iou_thresh
        self.stability_score_thresh = stability_score_thresh
        self.stability_score_offset = stability_score_offset
        self.box_nms_thresh = box_nms_thresh
        self.crop_nms_thresh = crop_nms_thresh
        self.min_mask_region_area = min_mask_region_area

    # ---- device part: everything up to and including the first NMS, no host sync -----------
    def propose(self, image, resized=None, layer_idx=0, crop_box=None, orig_size=None):
        """image: uint8 [H,W,3] numpy or device tensor: the crop (automatic_mask_generator.py:222-267; the whole
        image in the reference's own configuration).  crop_box / orig_size (H, W) of the full image enable the
        crop-edge filter (:305-307).
        Returns device tensors (masks [K,H,W] u8, boxes_xyxy [K,4] i32, iou [K], stab [K], order [K] i32,
        n [1] i32, points [K,2] float64 numpy): candidates order[:n] survive the filters + NMS; boxes, masks and
        points are in crop coordinates."""
        m = self.model
        H, W = image.shape[:2]
        nh, nw = get_preprocess_shape(H, W, m.img_size)
        if resized is None:
            # ResizeLongestSide.apply_image (utils/transforms.py:26-31): Pillow's bilinear resampler, bit-exact, on the device
            dev_img = image if isinstance(image, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(image)).to(m.device)
            resized = resize_longest_side(dev_img.contiguous(), m.img_size)
        emb = m.encode(resized)
        return self._propose_from_embedding(emb, H, W, nh, nw, layer_idx, crop_box, orig_size)

    def propose_batch(self, images, encoded_event=None):
        """propose() for several whole images with ONE encoder pass over all of them (Sam.encode_batch); decoder,
        post-processing and NMS per image.  Returns the list of propose() tuples.  encoded_event: a torch.cuda.Event that
        is recorded behind the encoder pass (the boundary between the GEMM-bound and the latency-bound part of the stage)."""
        m = self.model
        sizes, resized = [], []
        for image in images:
            H, W = image.shape[:2]
            dev_img = image if isinstance(image, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(image)).to(m.device)
            sizes.append((H, W) + get_preprocess_shape(H, W, m.img_size))
            resized.append(resize_longest_side(dev_img.contiguous(), m.img_size))
        emb = m.encode_batch(resized)
        if encoded_event is not None:
            encoded_event.record(torch.cuda.current_stream(m.device))
        return [self._propose_from_embedding(emb[i], *sizes[i], 0, None, None) for i in range(len(images))]

    def _propose_from_embedding(self, emb, H, W, nh, nw, layer_idx, crop_box, orig_size):
        m = self.model
        pts = self.point_grids[layer_idx] * np.array([[W, H]], dtype=np.float64)   # automatic_mask_generator.py:240-241
        tp = pts.copy()
        tp[:, 0] *= nw / W                                                       # apply_coords, utils/transforms.py:33-45
        tp[:, 1] *= nh / H
        key = (H, W, layer_idx)
        cache = self.__dict__.setdefault("_p01_cache", {})
        p01 = cache.get(key)
        if p01 is None:      # the prompt grid depends on the crop size only: upload it once (a crop layer alternates between sizes)
            if len(cache) >= 64:
                cache.clear()
            p01 = cache[key] = torch.from_numpy(((tp + 0.5) / float(m.img_size)).astype(np.float32)).to(m.device)
        lows, ious = [], []
        # the post-processing below drops every candidate whose prediction does not exceed pred_iou_thresh (when that is > 0:
        # automatic_mask_generator.py:287-291): prompts that fail with all three masks skip the decoder's upscaling
        gate = float(self.pred_iou_thresh) if self.pred_iou_thresh > 0 and getattr(self, "iou_gate", True) else None
        for s in range(0, len(pts), self.points_per_batch):
            low, iou = m.decode_points(emb, p01[s:s + self.points_per_batch].contiguous(), iou_gate=gate)
            lows.append(low.flatten(0, 1))
            ious.append(iou.flatten())
        low = lows[0] if len(lows) == 1 else torch.cat(lows)
        iou = ious[0] if len(ious) == 1 else torch.cat(ious)
        masks, boxes, stab, keep, _ = m.postprocess(low, iou, (nh, nw), (H, W), self.pred_iou_thresh,
                                                    self.stability_score_thresh, self.stability_score_offset)
        if crop_box is not None and orig_size is not None and list(crop_box) != [0, 0, orig_size[1], orig_size[0]]:
            box_near_crop_edge(boxes, keep, crop_box, [0, 0, orig_size[1], orig_size[0]])
        order, n = nms(boxes, iou, keep, self.box_nms_thresh)
        return masks, boxes, iou, stab, order, n, np.repeat(pts, 3, axis=0)

    def generate_device_crops(self, image):
        """_generate_masks with crop layers (automatic_mask_generator.py:197-220) + postprocess_small_regions.
        Returns device tensors (masks [n,H,W] u8, boxes_xywh [n,4] i64, iou [n], stability [n]) and host arrays
        (points [n,2] f64, crop_boxes [n,4] i64), in the reference's output order."""
        m = self.model
        dev_img = image if isinstance(image, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(image)).to(m.device)
        H, W = dev_img.shape[:2]
        crop_boxes, layer_idxs = generate_crop_boxes((H, W), self.crop_n_layers, self.crop_overlap_ratio)
        all_m, all_b, all_iou, all_stab, all_pts, all_cb = [], [], [], [], [], []
        # the crops are known from the image size alone: their encoder passes run as batches of up to 8 (better-filled GEMMs,
        # weights read once: 13.5 -> 11 ms per crop at ViT-H), the decoder / filters / NMS then crop by crop as in the reference
        embs = []
        for c0 in range(0, len(crop_boxes), 8):
            res = [resize_longest_side(dev_img[y0:y1, x0:x1, :].contiguous(), m.img_size) for x0, y0, x1, y1 in crop_boxes[c0:c0 + 8]]
            e = m.encode_batch(res) if len(res) > 1 else [m.encode(res[0])]
            embs.extend(e[i] for i in range(len(res)))
        for ci, (crop_box, layer_idx) in enumerate(zip(crop_boxes, layer_idxs)):
            x0, y0, x1, y1 = crop_box
            ch, cw = y1 - y0, x1 - x0
            masks, boxes, iou, stab, order, n, pts = self._propose_from_embedding(embs[ci], ch, cw, *get_preprocess_shape(ch, cw, m.img_size),
                                                                                   layer_idx, crop_box, (H, W))
            n = int(n.item())                                    # host sync: survivors of this crop
            if n == 0:
                continue
            idx = order[:n].long()
            full = torch.zeros((n, H, W), dtype=torch.uint8, device=m.device)     # uncrop_masks (amg.py:241-252)
            full[:, y0:y1, x0:x1] = masks.index_select(0, idx)
            off = torch.tensor([x0, y0, x0, y0], dtype=torch.int32, device=m.device)
            all_m.append(full)
            all_b.append(boxes.index_select(0, idx) + off)       # uncrop_boxes_xyxy (amg.py:225-231)
            all_iou.append(iou.index_select(0, idx))
            all_stab.append(stab.index_select(0, idx))
            all_pts.append(pts[idx.cpu().numpy()] + np.array([[x0, y0]], dtype=np.float64))   # uncrop_points
            all_cb.append(np.tile(np.array([crop_box], dtype=np.int64), (n, 1)))
        if not all_m:
            e = torch.empty
            return (e((0, H, W), dtype=torch.uint8, device=m.device), e((0, 4), dtype=torch.int64, device=m.device),
                    e((0,), device=m.device), e((0,), device=m.device), np.zeros((0, 2)), np.zeros((0, 4), np.int64))
        mk, bx = torch.cat(all_m), torch.cat(all_b).contiguous()
        iou, stab = torch.cat(all_iou), torch.cat(all_stab)
        pts, cbs = np.concatenate(all_pts), np.concatenate(all_cb)
        if len(crop_boxes) > 1:
            # duplicates between crops: prefer masks from smaller crops (automatic_mask_generator.py:209-220)
            area = (cbs[:, 2] - cbs[:, 0]) * (cbs[:, 3] - cbs[:, 1])
            scores = torch.from_numpy((1.0 / torch.from_numpy(area)).to(torch.float32).numpy()).to(m.device)
            order, n = nms(bx, scores, torch.ones(len(bx), dtype=torch.uint8, device=m.device), self.crop_nms_thresh)
            k = order[: int(n.item())].long()
            kc = k.cpu().numpy()
            mk, bx, iou, stab, pts, cbs = mk.index_select(0, k), bx.index_select(0, k).contiguous(), iou[k], stab[k], pts[kc], cbs[kc]
        if self.min_mask_region_area > 0 and len(bx) > 0:
            m1, c1 = remove_small_regions(mk.contiguous(), self.min_mask_region_area, "holes")
            m2, c2, nb = remove_small_regions_boxes(m1, self.min_mask_region_area, "islands")
            unchanged = ((c1 | c2) == 0).to(torch.float32)
            order2, n2 = nms(nb, unchanged, torch.ones(len(nb), dtype=torch.uint8, device=m.device),
                             max(self.box_nms_thresh, self.crop_nms_thresh))
            k = order2[: int(n2.item())].long()
            kc = k.cpu().numpy()
            mk, bx, iou, stab, pts, cbs = m2.index_select(0, k), nb.index_select(0, k), iou[k], stab[k], pts[kc], cbs[kc]
        b = bx.long()
        xywh = torch.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1)
        return mk, xywh, iou, stab, pts, cbs

    # ---- crop layers for a GROUP of images: three host syncs for the whole group instead of 1 per crop + 2 per image ----
    def crops_begin(self, images, encoded_event=None):
        """Stage A of generate_device_crops for several images on the current stream, nothing waited for: the crops of all
        images through the encoder in batches of up to 16, then decoder + fused post-processing + crop-edge filter + first
        NMS crop by crop; the survivor counts of ALL crops leave in one pinned copy behind an event (with the fp16 range
        counters).  The caller enqueues other work (the CLIP stage of the previous group) before crops_mid()."""
        m = self.model
        st = _GroupState()
        st.sizes, st.cand = [], []
        res, owner = [], []
        imgs = []
        for image in images:
            dev_img = image if isinstance(image, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(image)).to(m.device)
            H, W = dev_img.shape[:2]
            crop_boxes, layer_idxs = generate_crop_boxes((H, W), self.crop_n_layers, self.crop_overlap_ratio)
            st.sizes.append((int(H), int(W), crop_boxes, layer_idxs))
            imgs.append(dev_img)
            for x0, y0, x1, y1 in crop_boxes:
                res.append(resize_longest_side(dev_img[y0:y1, x0:x1, :].contiguous(), m.img_size))
                owner.append(len(imgs) - 1)
        embs = []
        for c0 in range(0, len(res), 16):
            chunk = res[c0:c0 + 16]
            e = m.encode_batch(chunk) if len(chunk) > 1 else [m.encode(chunk[0])]
            embs.extend(e[i] for i in range(len(chunk)))
        if encoded_event is not None:
            encoded_event.record(torch.cuda.current_stream(m.device))
        k, counts = 0, []
        for (H, W, crop_boxes, layer_idxs) in st.sizes:
            per = []
            for crop_box, layer_idx in zip(crop_boxes, layer_idxs):
                x0, y0, x1, y1 = crop_box
                ch, cw = y1 - y0, x1 - x0
                c = self._propose_from_embedding(embs[k], ch, cw, *get_preprocess_shape(ch, cw, m.img_size), layer_idx, crop_box, (H, W))
                k += 1
                per.append(c[:5])
                counts.append(c[5])
            st.cand.append(per)
        st.n1_dev = torch.cat(counts)
        st.n1 = torch.empty(len(counts), dtype=torch.int32).pin_memory()
        st.n1.copy_(st.n1_dev, non_blocking=True)
        st.overflow = 0
        st.ovf = torch.zeros(2, dtype=torch.int32).pin_memory()
        ops.split_overflow_peek(st.ovf)     # (the counters are process-wide: the CLIP / GEM models of the loop may be f16x3 when this one is not)
        st.ev1 = torch.cuda.Event()
        st.ev1.record(torch.cuda.current_stream(m.device))
        return st

    def crops_mid(self, st):
        """Stage B (host sync 1 of 3): every crop's survivors pasted into full-size masks (uncrop_masks / uncrop_boxes_xyxy,
        amg.py:225-252), the crops of an image concatenated in the reference's order, the cross-crop NMS that prefers
        masks of smaller crops (automatic_mask_generator.py:209-220); its counts leave in one copy."""
        st.ev1.synchronize()
        st.overflow = int(st.ovf[0]) + int(st.ovf[1])
        m = self.model
        dev = m.device
        n1 = [int(v) for v in st.n1.tolist()]
        k = 0
        st.stage, n_dev = [], []
        for (H, W, crop_boxes, layer_idxs), per in zip(st.sizes, st.cand):
            all_m, all_b, all_iou, all_stab, areas = [], [], [], [], []
            for crop_box, (masks, boxes, iou, stab, order) in zip(crop_boxes, per):
                n = n1[k]
                k += 1
                if n == 0:
                    continue
                x0, y0, x1, y1 = crop_box
                idx = order[:n].long()
                if (x0, y0, x1, y1) == (0, 0, W, H):
                    full = masks.index_select(0, idx)
                else:
                    full = torch.zeros((n, H, W), dtype=torch.uint8, device=dev)
                    full[:, y0:y1, x0:x1] = masks.index_select(0, idx)
                all_m.append(full)
                all_b.append(boxes.index_select(0, idx) + torch.tensor([x0, y0, x0, y0], dtype=torch.int32, device=dev))
                all_iou.append(iou.index_select(0, idx))
                all_stab.append(stab.index_select(0, idx))
                areas.append(np.full(n, (x1 - x0) * (y1 - y0), dtype=np.int64))
            if not all_m:
                st.stage.append(None)
                n_dev.append(torch.zeros(1, dtype=torch.int32, device=dev))
                continue
            mk, bx = torch.cat(all_m), torch.cat(all_b).contiguous()
            iou, stab = torch.cat(all_iou), torch.cat(all_stab)
            order, n = None, torch.full((1,), len(bx), dtype=torch.int32, device=dev)
            if len(crop_boxes) > 1:
                area = np.concatenate(areas)
                scores = torch.from_numpy((1.0 / torch.from_numpy(area)).to(torch.float32).numpy()).to(dev)
                order, n = nms(bx, scores, torch.ones(len(bx), dtype=torch.uint8, device=dev), self.crop_nms_thresh)
            st.stage.append((mk, bx, iou, stab, order))
            n_dev.append(n.reshape(1))
        st.cand = None
        st.n2 = torch.empty(len(n_dev), dtype=torch.int32).pin_memory()
        st.n2.copy_(torch.cat(n_dev), non_blocking=True)
        st.ev2 = torch.cuda.Event()
        st.ev2.record(torch.cuda.current_stream(dev))
        return st

    def crops_post(self, st):
        """Stage C (host sync 2 of 3): the survivors of the cross-crop NMS through postprocess_small_regions' kernels (holes,
        islands, boxes, the NMS that prefers untouched masks; automatic_mask_generator.py:324-372); counts in one copy."""
        st.ev2.synchronize()
        dev = self.model.device
        n2 = [int(v) for v in st.n2.tolist()]
        stage, n_dev = [], []
        for stg, n in zip(st.stage, n2):
            if stg is None or n == 0:
                stage.append(None)
                n_dev.append(torch.zeros(1, dtype=torch.int32, device=dev))
                continue
            mk, bx, iou, stab, order = stg
            if order is not None:
                k = order[:n].long()
                mk, bx, iou, stab = mk.index_select(0, k), bx.index_select(0, k).contiguous(), iou[k], stab[k]
            if self.min_mask_region_area > 0:
                m1, c1 = remove_small_regions(mk.contiguous(), self.min_mask_region_area, "holes")
                m2, c2, nb = remove_small_regions_boxes(m1, self.min_mask_region_area, "islands")
                unchanged = ((c1 | c2) == 0).to(torch.float32)
                order2, nn = nms(nb, unchanged, torch.ones(len(nb), dtype=torch.uint8, device=dev),
                                 max(self.box_nms_thresh, self.crop_nms_thresh))
                stage.append((m2, nb, iou, stab, order2))
                n_dev.append(nn.reshape(1))
            else:
                stage.append((mk, bx, iou, stab, None))
                n_dev.append(torch.full((1,), n, dtype=torch.int32, device=dev))
        st.stage = stage
        st.n1 = torch.empty(len(n_dev), dtype=torch.int32).pin_memory()
        st.n1.copy_(torch.cat(n_dev), non_blocking=True)
        st.ev1 = torch.cuda.Event()
        st.ev1.record(torch.cuda.current_stream(dev))
        return st

    def crops_finish(self, st):
        """Stage D (host sync 3 of 3): the final gathers.  Per image (masks [n,H,W] u8, boxes_xywh [n,4] i64, iou [n],
        stability [n]) -- the first four outputs of generate_device_crops, same order; n may be 0."""
        st.ev1.synchronize()
        dev = self.model.device
        n3 = [int(v) for v in st.n1.tolist()]
        out = []
        for stg, n, (H, W, _cb, _li) in zip(st.stage, n3, st.sizes):
            if stg is None or n == 0:
                e = torch.empty
                out.append((e((0, H, W), dtype=torch.uint8, device=dev), e((0, 4), dtype=torch.int64, device=dev),
                            e((0,), device=dev), e((0,), device=dev)))
                continue
            mk, bx, iou, stab, order2 = stg
            if order2 is not None:
                k = order2[:n].long()
                mk, bx, iou, stab = mk.index_select(0, k), bx.index_select(0, k), iou[k], stab[k]
            b = bx.long()
            out.append((mk, torch.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1), iou, stab))
        st.stage = None
        return out

    def generate_crops_group(self, images):
        """generate_device_crops (masks, boxes, iou, stability) for several images with three host syncs in all"""
        return self.crops_finish(self.crops_post(self.crops_mid(self.crops_begin(images))))

    def generate_device(self, image, resized=None, fixed_n=None):
        """Whole `generate` on the device.  Returns (masks [n,H,W] uint8, boxes_xywh [n,4] int64,
        iou [n], stability [n], cand [n] int64 indices into the 3*points candidates), all device
        tensors, in the reference's output order.  Two host syncs (the two proposal counts).
        fixed_n (benchmark only): take the first fixed_n survivors of the first NMS without reading
        the count back (the caller guarantees that many survive), run the clean-up kernels on them
        and skip the final gather -- no host sync at all."""
        masks, boxes, iou, stab, order, n, points = self.propose(image, resized)
        if fixed_n is not None:
            idx = order[:fixed_n].long()
            m = masks.index_select(0, idx).contiguous()
            m, nb = self.cleanup_fixed(m)
            return m, nb, iou.index_select(0, idx), stab.index_select(0, idx), idx
        n = int(n.item())                       # host sync: number of survivors of the first NMS
        idx = order[:n].long()
        m = masks.index_select(0, idx).contiguous()
        bx = boxes.index_select(0, idx).contiguous()
        if self.min_mask_region_area > 0 and n > 0:
            # postprocess_small_regions (automatic_mask_generator.py:324-372) on the device
            m1, c1 = remove_small_regions(m, self.min_mask_region_area, "holes")
            m2, c2, nb = remove_small_regions_boxes(m1, self.min_mask_region_area, "islands")
            unchanged = ((c1 | c2) == 0).to(torch.float32)           # score 1 for untouched masks
            keep_all = torch.ones(n, dtype=torch.uint8, device=m.device)
            order2, n2 = nms(nb, unchanged, keep_all, max(self.box_nms_thresh, self.crop_nms_thresh))
            k = order2[: int(n2.item())].long()
            m, bx, idx = m2.index_select(0, k), nb.index_select(0, k), idx.index_select(0, k)
        b = bx.long()
        xywh = torch.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1) if n > 0 else b
        return m, xywh, iou.index_select(0, idx), stab.index_select(0, idx), idx

    # ---- a GROUP of images through the generator: one encoder pass, two host syncs for the whole group ------------
    def group_begin(self, images, cap=None, encoded_event=None):
        """Stage A of generate() for several whole images (no crop layers) on the current stream: Pillow-exact resize,
        ONE encoder pass over all of them (Sam.encode_batch), then per image decoder + fused post-processing + first NMS.
        The survivor counts of all images leave in ONE device->host copy (pinned), marked by an event: nothing is waited for
        here, so the caller can enqueue other work (the CLIP stage of the previous group) before group_cleanup().
        cap: keep at most this many survivors per image (first NMS order; not in the reference -- synthetic benchmark:
        'AMG forced to keep a fixed 64')."""
        assert self.crop_n_layers == 0, "crop layers take generate_device_crops (one crop at a time)"
        st = _GroupState()
        st.cap = cap
        st.sizes = [tuple(int(v) for v in im.shape[:2]) for im in images]
        st.cand = self.propose_batch(images, encoded_event)
        dev = self.model.device
        st.n1_dev = torch.cat([c[5] for c in st.cand])
        st.n1 = torch.empty(len(images), dtype=torch.int32).pin_memory()
        st.n1.copy_(st.n1_dev, non_blocking=True)
        # the fp16 range guard of the f16x3 mode rides on the same read-back: what the device had counted (any stream) when
        # this stream got here -- the caller stops at this group instead of finding out at the end of the dataset
        st.overflow = 0
        st.ovf = torch.zeros(2, dtype=torch.int32).pin_memory()
        ops.split_overflow_peek(st.ovf)     # (the counters are process-wide: the CLIP / GEM models of the loop may be f16x3 when this one is not)
        st.ev1 = torch.cuda.Event()
        st.ev1.record(torch.cuda.current_stream(dev))
        return st

    def group_cleanup(self, st):
        """Stage B: waits for the counts of group_begin (host sync 1 of 2), gathers each image's survivors and runs
        postprocess_small_regions' kernels on them (holes, islands, boxes, second NMS; automatic_mask_generator.py:324-372).
        The second NMS counts leave in one copy again."""
        st.ev1.synchronize()
        if st.ovf is not None:
            st.overflow = int(st.ovf[0]) + int(st.ovf[1])
        dev = self.model.device
        n1 = [int(v) for v in st.n1.tolist()]
        if st.cap is not None:
            n1 = [min(v, st.cap) for v in n1]
        st.n1_list = n1
        st.stage = []
        n2_dev = []
        for (masks, boxes, iou, stab, order, _n, _pts), n in zip(st.cand, n1):
            if n == 0:
                st.stage.append(None)
                n2_dev.append(torch.zeros(1, dtype=torch.int32, device=dev))
                continue
            idx = order[:n].long()
            m = masks.index_select(0, idx)
            bx = boxes.index_select(0, idx)
            if self.min_mask_region_area > 0:
                m1, c1 = remove_small_regions(m, self.min_mask_region_area, "holes")
                m2, c2, nb = remove_small_regions_boxes(m1, self.min_mask_region_area, "islands")
                unchanged = ((c1 | c2) == 0).to(torch.float32)           # score 1 for untouched masks
                order2, n2 = nms(nb, unchanged, torch.ones(n, dtype=torch.uint8, device=dev),
                                 max(self.box_nms_thresh, self.crop_nms_thresh))
                st.stage.append((m2, nb, idx, order2, iou, stab))
                n2_dev.append(n2)
            else:
                st.stage.append((m, bx, idx, None, iou, stab))
                n2_dev.append(torch.full((1,), n, dtype=torch.int32, device=dev))
        st.cand = None     # the candidate tensors (192 full-size masks per image) can go back to the allocator
        st.n2 = None
        if self.min_mask_region_area > 0:
            st.n2 = torch.empty(len(n1), dtype=torch.int32).pin_memory()
            st.n2.copy_(torch.cat(n2_dev), non_blocking=True)
            st.ev2 = torch.cuda.Event()
            st.ev2.record(torch.cuda.current_stream(dev))
        return st

    def group_finish(self, st):
        """Stage C: host sync 2 of 2 (none without the small-region clean-up), the final gathers.  Returns per image what
        generate_device returns: (masks [n,H,W] uint8, boxes_xywh [n,4] int64, iou [n], stability [n], cand [n] int64);
        n may be 0."""
        if st.n2 is not None:
            st.ev2.synchronize()
            n2 = [int(v) for v in st.n2.tolist()]
        else:
            n2 = list(st.n1_list)
        out = []
        dev = self.model.device
        for stg, n, hw in zip(st.stage, n2, st.sizes):
            if stg is None or n == 0:
                e = torch.empty
                out.append((e((0,) + hw, dtype=torch.uint8, device=dev), e((0, 4), dtype=torch.int64, device=dev),
                            e((0,), device=dev), e((0,), device=dev), e((0,), dtype=torch.int64, device=dev)))
                continue
            m, bx, idx, order2, iou, stab = stg
            if order2 is not None:
                k = order2[:n].long()
                m, bx, idx = m.index_select(0, k), bx.index_select(0, k), idx.index_select(0, k)
            b = bx.long()
            xywh = torch.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1)
            out.append((m, xywh, iou.index_select(0, idx), stab.index_select(0, idx), idx))
        st.stage = None
        return out

    def generate_group(self, images, cap=None):
        """generate_device() for several images with one encoder pass and two host syncs in all (one without clean-up)."""
        return self.group_finish(self.group_cleanup(self.group_begin(images, cap)))

    def cleanup_fixed(self, m):
        """postprocess_small_regions kernels on a fixed batch of masks [n,H,W] uint8 without reading any
        count back: holes, islands, boxes, second NMS (its order is left on the device)."""
        if self.min_mask_region_area <= 0:
            return m, mask_boxes(m)
        m1, c1 = remove_small_regions(m, self.min_mask_region_area, "holes")
        m2, c2, nb = remove_small_regions_boxes(m1, self.min_mask_region_area, "islands")
        nms(nb, ((c1 | c2) == 0).to(torch.float32), torch.ones(m.shape[0], dtype=torch.uint8, device=m.device),
            max(self.box_nms_thresh, self.crop_nms_thresh))
        return m2, nb

    def _segmentation(self, mask):
        """automatic_mask_generator.py:176-182: the record's `segmentation` in the configured output mode (binary mask,
        uncompressed RLE dict, COCO RLE dict with the compressed counts string)."""
        if self.output_mode == "binary_mask":
            return mask
        rle = mask_to_rle(mask)
        return coco_encode_rle(rle) if self.output_mode == "coco_rle" else rle

    def generate(self, image):
        """automatic_mask_generator.py:137-195 -> list of records."""
        if self.crop_n_layers > 0:
            m, xywh, iou, stab, pts, cbs = self.generate_device_crops(image)
            masks = m.bool().cpu().numpy()
            xywh, iou, stab = xywh.cpu().numpy(), iou.cpu().numpy(), stab.cpu().numpy()
            out = []
            for i in range(len(masks)):
                cb = cbs[i]
                out.append({"segmentation": self._segmentation(masks[i]), "area": int(masks[i].sum()), "bbox": [int(v) for v in xywh[i]],
                            "predicted_iou": float(iou[i]), "point_coords": [pts[i].tolist()],
                            "stability_score": float(stab[i]),
                            "crop_box": [int(cb[0]), int(cb[1]), int(cb[2] - cb[0]), int(cb[3] - cb[1])]})   # XYWH
            return out
        m, xywh, iou, stab, idx = self.generate_device(image)
        masks = m.bool().cpu().numpy()
        xywh, iou, stab, idx = xywh.cpu().numpy(), iou.cpu().numpy(), stab.cpu().numpy(), idx.cpu().numpy()
        H, W = image.shape[:2]
        points = np.repeat(self.point_grids[0] * np.array([[W, H]], dtype=np.float64), 3, axis=0)
        out = []
        for i in range(len(masks)):
            out.append({"segmentation": self._segmentation(masks[i]), "area": int(masks[i].sum()),
                        "bbox": [int(v) for v in xywh[i]], "predicted_iou": float(iou[i]),
                        "point_coords": [points[idx[i]].tolist()], "stability_score": float(stab[i]),
                        "crop_box": [0, 0, W, H]})
        return out


def mask_to_rle(mask):
    """utils/amg.py:107-136 mask_to_rle_pytorch for one host mask [H,W]: {"size": [h, w], "counts": [...]} (column-major
    runs, the first one counts zeros) through the native codec."""
    lib = _lib.load()
    mk = np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
    H, W = mk.shape
    m = C.c_longlong(0)
    check(lib.hgl_rle_encode_mask(mk.ctypes.data, H, W, None, 0, C.byref(m)), "hgl_rle_encode_mask")
    counts = np.empty(m.value, dtype=np.uint32)
    check(lib.hgl_rle_encode_mask(mk.ctypes.data, H, W, counts.ctypes.data, m.value, C.byref(m)), "hgl_rle_encode_mask")
    return {"size": [H, W], "counts": [int(v) for v in counts]}


def rle_to_mask(rle):
    """utils/amg.py:139-151."""
    lib = _lib.load()
    h, w = rle["size"]
    counts = np.ascontiguousarray(np.asarray(rle["counts"], dtype=np.uint32))
    out = np.empty((h, w), dtype=np.uint8)
    check(lib.hgl_gt_mask_from_rle_counts(counts.ctypes.data, len(counts), h, w, out.ctypes.data, None), "hgl_gt_mask_from_rle_counts")
    return out.astype(bool)


def area_from_rle(rle):
    """utils/amg.py:154-155."""
    return sum(rle["counts"][1::2])


def coco_encode_rle(uncompressed_rle):
    """utils/amg.py:294-300: pycocotools' frPyObjects(uncompressed_rle) with the counts as a str -- the compressed string
    comes from the native restatement of maskApi.c:203-216 rleToString."""
    lib = _lib.load()
    h, w = uncompressed_rle["size"]
    counts = np.ascontiguousarray(np.asarray(uncompressed_rle["counts"], dtype=np.uint32))
    buf = C.create_string_buffer(7 * len(counts) + 1)
    n = C.c_size_t(0)
    check(lib.hgl_rle_to_string(counts.ctypes.data, len(counts), buf, len(buf), C.byref(n)), "hgl_rle_to_string")
    return {"size": [h, w], "counts": buf.raw[:n.value].decode("utf-8")}


def remove_small_regions(masks, area_thresh, mode):
    """utils/amg.py:267-291 for a batch [n,H,W] uint8 on the device -> (new masks, changed [n] uint8)."""
    lib = _lib.load()
    n, H, W = masks.shape
    out = torch.empty_like(masks)
    changed = torch.empty((n,), dtype=torch.uint8, device=masks.device)
    need = lib.hgl_remove_small_regions_workspace_bytes(n, H, W)
    ws = ops.workspace(need, masks.device, "sam_ccl")
    check(lib.hgl_remove_small_regions(ops._dev(masks, torch.uint8, "masks"), n, H, W, int(area_thresh),
                                       1 if mode == "holes" else 0, out.data_ptr(), changed.data_ptr(),
                                       ws.data_ptr(), ws.numel(), ops._stream()), "hgl_remove_small_regions")
    return out, changed


def remove_small_regions_boxes(masks, area_thresh, mode):
    """remove_small_regions with the boxes of the masks it writes (batched_mask_to_box, utils/amg.py:303-346) out of the
    same pass -> (new masks, changed [n] uint8, int32 XYXY [n,4])."""
    lib = _lib.load()
    n, H, W = masks.shape
    out = torch.empty_like(masks)
    changed = torch.empty((n,), dtype=torch.uint8, device=masks.device)
    boxes = torch.empty((n, 4), dtype=torch.int32, device=masks.device)
    need = lib.hgl_remove_small_regions_workspace_bytes(n, H, W)
    ws = ops.workspace(need, masks.device, "sam_ccl")
    check(lib.hgl_remove_small_regions_boxes(ops._dev(masks, torch.uint8, "masks"), n, H, W, int(area_thresh),
                                             1 if mode == "holes" else 0, out.data_ptr(), changed.data_ptr(), boxes.data_ptr(),
                                             ws.data_ptr(), ws.numel(), ops._stream()), "hgl_remove_small_regions_boxes")
    return out, changed, boxes


def mask_boxes(masks):
    """batched_mask_to_box (utils/amg.py:303-346) -> int32 XYXY [n,4]."""
    lib = _lib.load()
    n, H, W = masks.shape
    boxes = torch.empty((n, 4), dtype=torch.int32, device=masks.device)
    check(lib.hgl_mask_boxes(ops._dev(masks, torch.uint8, "masks"), n, H, W, boxes.data_ptr(), ops._stream()),
          "hgl_mask_boxes")
    return boxes
