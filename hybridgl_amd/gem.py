"""Drop-in for the `gem` package as Hybridgl_main.py uses it (`import gem`, :16; `gem.create_gem_model(
model_name='ViT-B/16', pretrained='openai', device=device)`, :36-38; `gem.get_gem_img_transform()`, :39;
`gem_model(image['tensor_img'].to(device), [noun_phrase])[0]`, :200).

gem_torch 1.0.1 is an external dependency of the reference (environment.yaml:206), absent here: what runs is the
published algorithm (self-self attention over q/k/v in the last gem_depth-1 blocks, adaptive temperature, second
residual stream, cosine matching with "a photo of a {phrase}.", bilinear up-sampling, min-max) in libhybridgl.so
(csrc/gem_api.hip).  Parity with the package is unpinned (oracle/gem_oracle.py says why); the torch resampling
operators it relies on are pinned against torch in tests/.

The ViT is the OpenAI CLIP ViT-B/16 -- the same checkpoint CLIPViTFM loads -- so `create_gem_model(clip=model)`
shares the device weights of an existing CLIPViTFM and only adds the interpolated positional embedding.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib, ops
from ._lib import HglClipVisionW, check

OPENAI_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_STD = (0.26862954, 0.26130258, 0.27577711)


def _cubic_taps(out_size, in_size, scale_factor):
    """source rows and the four cubic-convolution weights (A = -0.75) of F.interpolate(mode='bicubic',
    align_corners=False, scale_factor=s): the kernel maps with 1/s, border taps are clamped."""
    A = np.float32(-0.75)
    scale = np.float32(1.0 / scale_factor)
    src = scale * (np.arange(out_size, dtype=np.float32) + np.float32(0.5)) - np.float32(0.5)
    i = np.floor(src).astype(np.int64)
    t = (src - i).astype(np.float32)
    inner = lambda x: ((A + 2) * x - (A + 3)) * x * x + 1
    outer = lambda x: ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    w = np.stack([outer(t + 1), inner(t), inner(1 - t), outer(2 - t)], axis=1).astype(np.float32)
    idx = np.clip(i[:, None] + np.arange(-1, 3)[None, :], 0, in_size - 1)
    return idx, w


def interpolate_pos_encoding(pos, grid_h, grid_w):
    """GEMViT.interpolate_pos_encoding: bicubic resampling of the patch part of positional_embedding
    [1 + n*n, D] to grid_h x grid_w with scale factors (grid + 0.1) / n (host, once per geometry)."""
    pos = np.asarray(pos, dtype=np.float32)
    n = int(round(math.sqrt(pos.shape[0] - 1)))
    if (grid_h, grid_w) == (n, n):
        return pos
    D = pos.shape[1]
    patch = pos[1:].reshape(n, n, D)
    ix, wx = _cubic_taps(grid_w, n, (grid_w + 0.1) / n)
    iy, wy = _cubic_taps(grid_h, n, (grid_h + 0.1) / n)
    rows = (patch[:, ix, :] * wx[None, :, :, None]).sum(axis=2, dtype=np.float32)        # [n, gw, D]
    up = (rows[iy] * wy[:, :, None, None]).sum(axis=1, dtype=np.float32)                  # [gh, gw, D]
    return np.concatenate([pos[:1], up.reshape(grid_h * grid_w, D)], axis=0).astype(np.float32)


def get_gem_img_transform(img_size=448):
    """gem.get_gem_img_transform: Resize((s, s), bicubic) -> RGB -> ToTensor -> Normalize(OpenAI mean/std).
    Host-side (PIL), as in the reference's data loader (data/dataset_refer_bert.py:108-109)."""
    from PIL import Image
    mean = np.asarray(OPENAI_MEAN, dtype=np.float32).reshape(3, 1, 1)
    std = np.asarray(OPENAI_STD, dtype=np.float32).reshape(3, 1, 1)

    def transform(img):
        if not isinstance(img, Image.Image):
            img = Image.fromarray(np.asarray(img))
        img = img.resize((img_size, img_size), Image.BICUBIC).convert("RGB")
        a = np.asarray(img, dtype=np.float32).transpose(2, 0, 1) / np.float32(255.0)
        return torch.from_numpy(((a - mean) / std).astype(np.float32))

    return transform


def resize_antialias(x, size):
    """T.Resize(size, antialias=True) on a float tensor [..., h, w] (Hybridgl_main.py:201)."""
    lib = _lib.load()
    H, W = int(size[0]), int(size[1])
    h, w = x.shape[-2:]
    xc = x.contiguous()
    Cn = xc.numel() // (h * w)
    out = torch.empty(tuple(x.shape[:-2]) + (H, W), dtype=torch.float32, device=x.device)
    check(lib.hgl_resize_bilinear_aa(ops._dev(xc, torch.float32, "x"), Cn, h, w, out.data_ptr(), H, W, ops._stream()),
          "hgl_resize_bilinear_aa")
    return out


class GEMWrapper:
    """gem.gem_wrapper.GEMWrapper: callable (image [B,3,R,R], [texts]) -> [B, len(texts), R, R]."""

    def __init__(self, clip_model, tokenizer=None, depth=7, ss_attn_iter=1, ss_attn_temp=None):
        self.model = clip_model                    # backbone._ClipModel: vision + text towers on the device
        self.tokenizer = tokenizer
        self.depth = int(depth)
        self.ss_attn_iter = int(ss_attn_iter)
        self.ss_attn_temp = ss_attn_temp
        self.patch_size = clip_model.cfg["vision_patch_size"]
        self.device = clip_model.device
        self._pos_host = clip_model._vt["pos"].cpu().numpy()
        self._geom = {}                            # grid -> (pos tensor, HglClipVisionW)

    def _vision(self, grid):
        if grid not in self._geom:
            base = self.model.visual_w
            pos = torch.from_numpy(interpolate_pos_encoding(self._pos_host, grid, grid)).to(self.device)
            v = HglClipVisionW()
            for name, _ in HglClipVisionW._fields_:
                setattr(v, name, getattr(base, name))
            v.grid = grid
            v.positional_embedding = pos.data_ptr()
            self._geom[grid] = (pos, v)
        return self._geom[grid][1]

    @property
    def gem_blocks(self):
        """gem_wrapper swaps resblocks[-i] for i in range(1, depth): the last depth-1 blocks"""
        return max(0, min(self.model.cfg["vision_layers"], self.depth - 1))

    def image_features(self, image, return_ori=False):
        """GEMViT.forward on one image [3, R, R] -> [1 + g*g, embed] (ln_post + proj of every token)."""
        lib = _lib.load()
        ops.use_precision(self.model.precision)
        assert image.dim() == 3 and image.shape[0] == 3 and image.shape[1] == image.shape[2], "image must be [3,R,R]"
        R = image.shape[-1]
        assert R % self.patch_size == 0, "image side must be a multiple of the patch size"
        grid = R // self.patch_size
        v = self._vision(grid)
        img = image.contiguous()
        ip = ops._dev(img, torch.float32, "image")
        E = self.model.cfg["embed_dim"]
        S = grid * grid + 1
        feat = torch.empty((S, E), dtype=torch.float32, device=img.device)
        ori = torch.empty((S, E), dtype=torch.float32, device=img.device) if return_ori else None
        need = lib.hgl_gem_workspace_bytes(C.byref(v))
        ws = ops.workspace(need, img.device, "gem")
        check(lib.hgl_gem_image_features(C.byref(v), ip, self.gem_blocks, self.ss_attn_iter,
                                         float(self.ss_attn_temp) if self.ss_attn_temp else 0.0, feat.data_ptr(),
                                         ori.data_ptr() if return_ori else None, ws.data_ptr(), ws.numel(), ops._stream()),
              "hgl_gem_image_features")
        return ori if return_ori else feat

    def image_features_batch(self, images, return_ori=False):
        """GEMViT.forward on several images at once [B, 3, R, R] -> [B, 1 + g*g, embed]: the token rows of the images are
        stacked, so every GEMM / LayerNorm launch covers all of them (one self-self temperature per image)."""
        lib = _lib.load()
        ops.use_precision(self.model.precision)
        assert images.dim() == 4 and images.shape[1] == 3 and images.shape[2] == images.shape[3], "images must be [B,3,R,R]"
        nb, R = images.shape[0], images.shape[-1]
        assert R % self.patch_size == 0, "image side must be a multiple of the patch size"
        grid = R // self.patch_size
        v = self._vision(grid)
        imgs = images.contiguous()
        E = self.model.cfg["embed_dim"]
        S = grid * grid + 1
        feat = torch.empty((nb, S, E), dtype=torch.float32, device=imgs.device)
        ori = torch.empty((nb, S, E), dtype=torch.float32, device=imgs.device) if return_ori else None
        need = lib.hgl_gem_batch_workspace_bytes(C.byref(v), nb)
        ws = ops.workspace(need, imgs.device, "gem")
        check(lib.hgl_gem_image_features_batch(C.byref(v), ops._dev(imgs, torch.float32, "images"), nb, self.gem_blocks,
                                               self.ss_attn_iter, float(self.ss_attn_temp) if self.ss_attn_temp else 0.0,
                                               feat.data_ptr(), ori.data_ptr() if return_ori else None, ws.data_ptr(),
                                               ws.numel(), ops._stream()), "hgl_gem_image_features_batch")
        return ori if return_ori else feat

    def heatmap(self, feat, text_feats, res, normalize=True):
        """feat [1+g*g, E], text_feats [T, E] -> [T, res, res]"""
        lib = _lib.load()
        ops.use_precision(self.model.precision)
        T, E = text_feats.shape
        grid = int(round(math.sqrt(feat.shape[0] - 1)))
        tf = text_feats.contiguous()
        heat = torch.empty((T, res, res), dtype=torch.float32, device=feat.device)
        need = lib.hgl_gem_heatmap_workspace_bytes(grid, T, res)
        ws = ops.workspace(need, feat.device, "gem_heat")
        check(lib.hgl_gem_heatmap(ops._dev(feat, torch.float32, "feat"), grid, E, ops._dev(tf, torch.float32, "text"), T, res,
                                  1 if normalize else 0, heat.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()),
              "hgl_gem_heatmap")
        return heat

    @staticmethod
    def prompts(text):
        return [f"a photo of a {cls}." for cls in text]

    def tokenize(self, text):
        from .tokenizer import tokenize
        return tokenize(self.prompts(text), self.model.context_length, tokenizer=self.tokenizer)

    def encode_text(self, text):
        """-> [1, T, E]: embeddings of "a photo of a {cls}." (normalised inside hgl_gem_heatmap)"""
        tok = torch.from_numpy(self.tokenize(text)).to(self.device)
        return self.model.encode_text(tok).unsqueeze(0)

    def forward(self, image, text, normalize=True, return_ori=False):
        assert image.dim() == 4, "image must be [B,3,W,H]"
        txt = self.encode_text(text)[0]
        feats = self.image_features_batch(image, return_ori)
        maps = [self.heatmap(feats[b], txt, image.shape[-1], normalize) for b in range(image.shape[0])]
        return torch.stack(maps, dim=0)

    __call__ = forward

    def batched_forward(self, image, text, normalize=True, return_ori=False):
        """one list of prompts per image -> list of [T_b, W, H]"""
        assert image.shape[0] == len(text)
        return [self.forward(image[b:b + 1], text[b], normalize, return_ori)[0] for b in range(len(text))]

    def to(self, device):
        if torch.device(device).type != "cuda":
            raise _lib.HybridGLError("GEM runs on the GPU only (no CPU path exists)")
        return self

    def eval(self):
        return self


def create_gem_model(model_name="ViT-B/16", pretrained="openai", gem_depth=7, ss_attn_iter=1, ss_attn_temp=None,
                     device="cuda", clip=None, state_dict=None, checkpoint=None, seed=0, precision=None, tokenizer=None):
    """gem.create_gem_model.  Extra keywords (not in the package): `clip` = an existing CLIPViTFM whose device
    weights are shared; `state_dict` / `checkpoint` / `seed` as for CLIPViTFM."""
    if pretrained not in ("openai", None):
        raise ValueError("only the OpenAI CLIP checkpoints are supported (Hybridgl_main.py:37)")
    if clip is None:
        from .backbone import CLIPViTFM
        clip = CLIPViTFM(model_name, state_dict=state_dict, checkpoint=checkpoint, seed=seed, device=device,
                         precision=precision)
    return GEMWrapper(clip.model, tokenizer, gem_depth, ss_attn_iter, ss_attn_temp)
