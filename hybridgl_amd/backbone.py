"""Drop-in for the reference's `model.backbone.CLIPViTFM` (model/backbone.py:12-309).

Same constructor, attributes and methods; the arithmetic runs in libhybridgl.so
(hand-written HIP for gfx950) through the C ABI of include/hybridgl.h.  Weights live in
torch CUDA tensors owned by this object (torch = device memory + streams only).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib, ops, weights
from ._lib import HglClipTextW, HglClipVisionW, HglResBlockW, check

FUSION = {"G2L": 0, "L2G": 1, "G2L&L2G": 2, "token_masking": 3, "attn_masking": 4, "crop": 5}


def load_clip_state_dict(path):
    """OpenAI CLIP checkpoint (JIT archive or plain state_dict), as clip/clip.py:119-142 does."""
    try:
        sd = torch.jit.load(path, map_location="cpu").state_dict()
    except RuntimeError:
        sd = torch.load(path, map_location="cpu")
        if "state_dict" in sd:
            sd = sd["state_dict"]
    return {k: v.float().numpy() for k, v in sd.items()
            if k not in ("input_resolution", "context_length", "vocab_size")}


def _infer_config(sd):
    """clip/model.py:474-503 build_model geometry inference (ViT only)."""
    vw = sd["visual.conv1.weight"].shape[0]
    vl = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    vp = sd["visual.conv1.weight"].shape[-1]
    grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    tw = sd["ln_final.weight"].shape[0]
    tl = len(set(k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")))
    return dict(embed_dim=sd["text_projection"].shape[1], image_resolution=vp * grid, vision_layers=vl,
                vision_width=vw, vision_patch_size=vp, context_length=sd["positional_embedding"].shape[0],
                vocab_size=sd["token_embedding.weight"].shape[0], transformer_width=tw,
                transformer_heads=tw // 64, transformer_layers=tl)


class _Blocks:
    """Device tensors + the C array of HglResBlockW for one transformer."""

    def __init__(self, sd, prefix, layers, device, precision="f32"):
        self.t = []  # keep tensors alive
        self.arr = (HglResBlockW * layers)()
        names = [("ln1_w", "ln_1.weight"), ("ln1_b", "ln_1.bias"),
                 ("in_proj_w", "attn.in_proj_weight"), ("in_proj_b", "attn.in_proj_bias"),
                 ("out_proj_w", "attn.out_proj.weight"), ("out_proj_b", "attn.out_proj.bias"),
                 ("ln2_w", "ln_2.weight"), ("ln2_b", "ln_2.bias"),
                 ("fc_w", "mlp.c_fc.weight"), ("fc_b", "mlp.c_fc.bias"),
                 ("proj_w", "mlp.c_proj.weight"), ("proj_b", "mlp.c_proj.bias")]
        for i in range(layers):
            for field, key in names:
                t = _to_dev(sd[f"{prefix}.resblocks.{i}.{key}"], device)
                self.t.append(t)
                setattr(self.arr[i], field, t.data_ptr())
                if precision == "f16x3" and field in ("in_proj_w", "out_proj_w", "fc_w", "proj_w"):
                    ops.register_split_weight(t)   # fp16 hi/lo halves for the split matrix-core path


def _to_dev(a, device):
    if isinstance(a, torch.Tensor):
        t = a.detach().to(torch.float32)
    else:
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return t.contiguous().to(device)


class _ClipModel:
    """The `.model` attribute: what the reference reaches as Model.model.* (clip/model.py:340-431)."""

    def __init__(self, sd, cfg, device, precision="f32"):
        import weakref
        self.cfg = cfg
        self.precision = precision
        self.device = torch.device(device)
        _before = set(ops._split_cache)
        self.dtype = torch.float32  # convert_weights is disabled in the reference (clip/model.py:509)
        vw, vl = cfg["vision_width"], cfg["vision_layers"]
        p = cfg["vision_patch_size"]
        grid = cfg["image_resolution"] // p
        self._vb = _Blocks(sd, "visual.transformer", vl, device, precision)
        self._vt = {
            "conv1": _to_dev(np.asarray(sd["visual.conv1.weight"]).reshape(vw, -1), device),
            "cls": _to_dev(sd["visual.class_embedding"], device),
            "pos": _to_dev(sd["visual.positional_embedding"], device),
            "ln_pre_w": _to_dev(sd["visual.ln_pre.weight"], device),
            "ln_pre_b": _to_dev(sd["visual.ln_pre.bias"], device),
            "ln_post_w": _to_dev(sd["visual.ln_post.weight"], device),
            "ln_post_b": _to_dev(sd["visual.ln_post.bias"], device),
            "proj_t": _to_dev(np.ascontiguousarray(np.asarray(sd["visual.proj"]).T), device),
        }
        if precision == "f16x3" and self._vt["conv1"].shape[1] % 64 == 0:
            ops.register_split_weight(self._vt["conv1"])     # patch embedding as a split-fp16 GEMM
        v = HglClipVisionW()
        v.width, v.layers, v.heads, v.patch, v.grid, v.embed = vw, vl, vw // 64, p, grid, cfg["embed_dim"]
        v.conv1_w = self._vt["conv1"].data_ptr()
        v.class_embedding = self._vt["cls"].data_ptr()
        v.positional_embedding = self._vt["pos"].data_ptr()
        v.ln_pre_w, v.ln_pre_b = self._vt["ln_pre_w"].data_ptr(), self._vt["ln_pre_b"].data_ptr()
        v.blocks = C.cast(self._vb.arr, C.POINTER(HglResBlockW))
        v.ln_post_w, v.ln_post_b = self._vt["ln_post_w"].data_ptr(), self._vt["ln_post_b"].data_ptr()
        v.proj_t = self._vt["proj_t"].data_ptr()
        self.visual_w = v

        tw, tl = cfg["transformer_width"], cfg["transformer_layers"]
        self._tb = _Blocks(sd, "transformer", tl, device, precision)
        self._tt = {
            "emb": _to_dev(sd["token_embedding.weight"], device),
            "pos": _to_dev(sd["positional_embedding"], device),
            "ln_w": _to_dev(sd["ln_final.weight"], device),
            "ln_b": _to_dev(sd["ln_final.bias"], device),
            "proj_t": _to_dev(np.ascontiguousarray(np.asarray(sd["text_projection"]).T), device),
        }
        t = HglClipTextW()
        t.width, t.layers, t.heads = tw, tl, cfg["transformer_heads"]
        t.context, t.vocab, t.embed = cfg["context_length"], cfg["vocab_size"], cfg["embed_dim"]
        t.token_embedding = self._tt["emb"].data_ptr()
        t.positional_embedding = self._tt["pos"].data_ptr()
        t.blocks = C.cast(self._tb.arr, C.POINTER(HglResBlockW))
        t.ln_final_w, t.ln_final_b = self._tt["ln_w"].data_ptr(), self._tt["ln_b"].data_ptr()
        t.text_projection_t = self._tt["proj_t"].data_ptr()
        self.text_w = t
        self.logit_scale = _to_dev(np.asarray(sd["logit_scale"], dtype=np.float32).reshape(()), device)
        self._logit_scale_exp = float(np.exp(np.float32(np.asarray(sd["logit_scale"]))))
        self.context_length = cfg["context_length"]
        # the fp16 splits registered above die with this model (library registry + hi/lo tensors)
        self._split_keys = ops.split_weight_keys_since(_before)
        weakref.finalize(self, ops.release_split_weights, list(self._split_keys))

    def encode_text(self, text, target_noun_index=None, seq_len=None, masking_index=(), masking_block=None):
        """CLIP.encode_text (clip/model.py:414-431). text: [B, context] integer tokens.
        target_noun_index: the projected row is position target_noun_index + 1 of every string instead of its EOT
        (clip/model.py:426-428); like the reference's `if target_noun_index:` a None or 0 pools the EOT.  A tensor /
        sequence of one index per string is accepted as well (the reference's truth test only admits one element).
        seq_len (not in the reference): compute only the first seq_len positions -- exact under the causal mask when
        every pooled position lies inside that prefix (the caller's promise; the tokenizer knows the lengths).
        masking_index / masking_block: CLIPViTFM.text_masking_feature (see there)."""
        lib = _lib.load()
        ops.use_precision(self.precision)
        if not text.is_cuda:
            raise _lib.HybridGLError("encode_text: tokens must be on the GPU (no CPU path exists)")
        tok = text.to(torch.int32).contiguous()
        B = tok.shape[0]
        assert tok.shape[1] == self.context_length
        pool = None
        if target_noun_index is not None:
            if isinstance(target_noun_index, torch.Tensor):
                tni = target_noun_index.reshape(-1).to(torch.int64).cpu()
            else:
                tni = torch.as_tensor(target_noun_index, dtype=torch.int64).reshape(-1)
            if tni.numel() not in (1, B):
                raise ValueError(f"target_noun_index: expected 1 or {B} indices, got {tni.numel()}")
            if bool((tni != 0).any()) or tni.numel() > 1:      # a lone 0 is falsy in the reference -> EOT pooling
                pos = (tni + 1).expand(B) if tni.numel() == 1 else tni + 1
                if int(pos.min()) < 0 or int(pos.max()) >= self.context_length:
                    raise IndexError("target_noun_index + 1 outside the context")
                pool = pos.to(torch.int32).contiguous().to(tok.device)
        zero = None
        if len(masking_index):
            zero = torch.as_tensor([int(i) + 1 for i in masking_index], dtype=torch.int32).to(tok.device)   # + start token
        need = lib.hgl_clip_text_workspace_bytes(C.byref(self.text_w), B)
        ws = ops.workspace(need, tok.device, "clip_text")
        out = torch.empty((B, self.cfg["embed_dim"]), dtype=torch.float32, device=tok.device)
        S = self.context_length if seq_len is None else max(1, min(int(seq_len), self.context_length))
        check(lib.hgl_clip_encode_text_ex(C.byref(self.text_w), tok.data_ptr(), B, S,
                                          pool.data_ptr() if pool is not None else None,
                                          zero.data_ptr() if zero is not None else None, 0 if zero is None else zero.numel(),
                                          0 if masking_block is None else int(masking_block), out.data_ptr(),
                                          ws.data_ptr(), ws.numel(), ops._stream()), "hgl_clip_encode_text")
        return out


class CLIPViTFM:
    """model/backbone.py:12 -- `CLIPViTFM(model_name='ViT-B/16', size=224)`.

    Extra keyword arguments (not in the reference): `state_dict` / `checkpoint` to supply
    weights (the reference downloads them, clip/clip.py:94-142; there is no network here),
    `seed` for the synthetic weights used when neither is given, `device`.
    """

    def __init__(self, model_name="ViT-B/16", size=224, state_dict=None, checkpoint=None, seed=0,
                 device="cuda", precision=None):
        """precision: 'f32' (exact fp32 MFMA) or 'f16x3' (split-fp16 MFMA, fp32-class accuracy, ~2.4x
        faster GEMMs); default from HYBRIDGL_PRECISION (ops.default_precision)."""
        _lib.load()
        precision = precision or ops.default_precision()
        ops.use_precision(precision)
        # model/backbone.py:16-21 (+ the ViT-L/14 extension of SURVEY.md note 2)
        if model_name in ("ViT-B/32", "ViT-B/16", "tiny"):
            self.last_layer, self.num_heads = 10, 12
        elif model_name == "ViT-L/14":
            self.last_layer, self.num_heads = 22, 16
        else:
            raise ValueError(f"unsupported model_name {model_name!r}")
        checkpoint = checkpoint or os.environ.get("HYBRIDGL_CLIP_CHECKPOINT")
        if state_dict is None and checkpoint:
            state_dict = load_clip_state_dict(checkpoint)
        if state_dict is None:
            state_dict = weights.clip_state_dict(model_name, seed)
        cfg = _infer_config(state_dict)
        self.model_name = model_name
        self.model = _ClipModel(state_dict, cfg, device, precision)

    # nn.Module-compatible no-ops used by Hybridgl_main.py:47-48
    def to(self, device):
        if torch.device(device).type != "cuda":
            raise _lib.HybridGLError("CLIPViTFM runs on the GPU only (no CPU path exists)")
        return self

    def eval(self):
        return self

    @property
    def device(self):
        return self.model.device

    @property
    def dtype(self):
        return self.model.dtype

    def text_feature(self, text):
        """model/backbone.py:58-70: the text tower with its causal mask, EOT pooling and projection -- the same
        computation as CLIP.encode_text."""
        return self.model.encode_text(text)

    def text_masking_feature(self, text, masking_index=[], masking_block=11):   # noqa: B006 -- reference signature
        """model/backbone.py:34-56: the text tower with the token positions masking_index (+1 for the start token) of
        every string zeroed before each block >= masking_block; EOT pooling and projection as text_feature."""
        return self.model.encode_text(text, masking_index=list(masking_index), masking_block=masking_block)

    def calculate_score(self, image_features, text_features, visual_norm_dim=1):
        """model/backbone.py:74-87 -> [N, T] logits."""
        assert visual_norm_dim == 1
        return ops.calculate_score(image_features.contiguous(), text_features.contiguous(),
                                   self.model._logit_scale_exp)

    def forward(self, local_imgs, global_imgs, pred_masks, masking_block=None, fusion_mode="G2L"):
        """model/backbone.py:117 -> [N, embed_dim]."""
        lib = _lib.load()
        ops.use_precision(self.model.precision)
        if fusion_mode not in FUSION:
            raise ValueError(f"unknown fusion_mode {fusion_mode!r}")
        mode = FUSION[fusion_mode]
        N = local_imgs.shape[0]
        res = self.model.cfg["image_resolution"]
        assert tuple(local_imgs.shape[1:]) == (3, res, res), "local_imgs must be [N,3,res,res]"
        local_imgs = local_imgs.to(torch.float32).contiguous()      # x.type(self.model.dtype), model/backbone.py:123
        lp = ops._dev(local_imgs, torch.float32, "local_imgs")
        gp = None
        if global_imgs is not None:
            assert global_imgs.shape == local_imgs.shape, "global_imgs must have the shape of local_imgs"
            global_imgs = global_imgs.to(torch.float32).contiguous()
            gp = ops._dev(global_imgs, torch.float32, "global_imgs")
        v = self.model.visual_w
        out = torch.empty((N, self.model.cfg["embed_dim"]), dtype=torch.float32, device=local_imgs.device)
        mb = -1 if masking_block is None else int(masking_block)
        if isinstance(pred_masks, (list, tuple)):
            # (not in the reference) the proposals of several images, each [n_i, H_i, W_i] with its own size: one forward
            # over all rows (a group of dataset items; every row is independent)
            segs = [pm if pm.dtype in (torch.bool, torch.uint8) else (pm != 0) for pm in pred_masks]
            segs = [ops._u8(pm.contiguous(), "pred_masks")[1] for pm in segs]
            ns = len(segs)       # (the library checks that the runs hold N masks in all)
            ptrs = (C.c_void_p * ns)(*[pm.data_ptr() for pm in segs])
            sn = (C.c_int * ns)(*[int(pm.shape[0]) for pm in segs])
            sh = (C.c_int * ns)(*[int(pm.shape[1]) for pm in segs])
            sw = (C.c_int * ns)(*[int(pm.shape[2]) for pm in segs])
            need = lib.hgl_clip_hybrid_workspace_bytes(C.byref(v), N, 0, 0, mode)
            ws = ops.workspace(need, local_imgs.device, "clip_hybrid")
            check(lib.hgl_clip_hybrid_forward_segments(C.byref(v), lp, gp, ptrs, sn, sh, sw, ns, N, mode, mb, self.last_layer,
                                                       out.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()),
                  "hgl_clip_hybrid_forward_segments")
            return out
        mp, Hm, Wm = None, 0, 0
        if pred_masks is not None:
            assert pred_masks.shape[0] == N, "one mask per image row"
            Hm, Wm = pred_masks.shape[1:]
            pm = pred_masks if pred_masks.dtype in (torch.bool, torch.uint8) else (pred_masks != 0)
            mp, pm = ops._u8(pm.contiguous(), "pred_masks")
        need = lib.hgl_clip_hybrid_workspace_bytes(C.byref(v), N, Hm, Wm, mode)
        ws = ops.workspace(need, local_imgs.device, "clip_hybrid")
        check(lib.hgl_clip_hybrid_forward(C.byref(v), lp, gp, mp, N, Hm, Wm, mode, mb,
                                          self.last_layer, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                          ops._stream()), "hgl_clip_hybrid_forward")
        return out

    __call__ = forward
