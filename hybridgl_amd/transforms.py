"""The per-item transforms of the reference's DataLoader workers, on the device, bit-identical to the host versions.

The reference prepares every dataset item on CPU worker processes (Hybridgl_main.py:40-45):

    image['image']       T.ToTensor() + T.Normalize(ImageNet)            data/dataset_refer_bert.py:155-158
    image['tensor_img']  gem.get_gem_img_transform(): Resize((448, 448), bicubic) + ToTensor + Normalize(OpenAI)
                                                                           Hybridgl_main.py:39, dataset_refer_bert.py:108-109

On a 640 x 480 image those two cost 15 + 9 ms of host float work per item (numpy / PIL) against 3 ms for the JPEG decode;
eight ranks on one node would need them for ~50 items a second.  Here the loader threads upload the decoded uint8 image
once and run

    hgl_u8_to_chw_lut        out[c, y, x] = lut[c, img[y, x, c]]; the 3 x 256 table holds the host transform's own fp32
                             results ((v / 255 - mean) / std), so the output is the host transform's bit for bit
    hgl_resize_pil_bilinear  Pillow's 8-bit two-pass resampler with 22-bit fixed-point weights; the weight tables are the
                             caller's, so the same kernel does BICUBIC (pil_coeffs(..., "bicubic")) -- bit-exact with
                             Image.resize (tests/test_gpu_transforms.py compares against Pillow itself)

on their own copy stream.  Temporaries are plain torch allocations (stream-aware), not the shared named workspaces of
`ops.workspace`, because several loader threads run these at the same time.
"""
import math
import threading

import numpy as np
import torch

from . import _lib, ops
from ._lib import check

_PIL_PB = 22          # Resample.c PRECISION_BITS = 32 - 8 - 2
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


_FILTERS = {"bilinear": (_bilinear, 1.0), "bicubic": (_bicubic, 2.0)}


def pil_coeffs(in_size, out_size, filt="bilinear"):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (Resample.c; double precision, sums in index order):
    int32 weights [out, ksize] and (first, count) bounds per output index."""
    f, fsupport = _FILTERS[filt]
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = fsupport * fs
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / fs
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        cnt = min(int(center + support + 0.5), in_size) - xmin
        w = [f((x + xmin - center + 0.5) * ss) for x in range(cnt)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(cnt):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << _PIL_PB)) if k < 0 else int(0.5 + k * (1 << _PIL_PB))
        bounds[xx] = (xmin, cnt)
    return kk, bounds


def pil_resample_numpy(img, out_h, out_w, filt="bilinear"):
    """The two passes of ImagingResample on a uint8 HWC array in numpy (host checker of the tables; the device kernel
    does the same integer arithmetic)."""
    def one(a, n_out, axis, kk, bounds):
        a = np.moveaxis(a, axis, 0).astype(np.int64)
        out = np.empty((n_out,) + a.shape[1:], dtype=np.uint8)
        for o in range(n_out):
            first, cnt = bounds[o]
            acc = np.tensordot(kk[o, :cnt].astype(np.int64), a[first:first + cnt], axes=(0, 0)) + (1 << (_PIL_PB - 1))
            out[o] = np.clip(acc >> _PIL_PB, 0, 255)
        return np.moveaxis(out, 0, axis)
    H, W = img.shape[:2]
    tmp = one(img, out_w, 1, *pil_coeffs(W, out_w, filt)) if out_w != W else img
    return one(tmp, out_h, 0, *pil_coeffs(H, out_h, filt)) if out_h != H else tmp


_tab_lock = threading.Lock()
_tab_cache = {}


def _tables(H, W, out_h, out_w, filt, device):
    key = (H, W, out_h, out_w, filt, str(device))
    with _tab_lock:
        t = _tab_cache.get(key)
    if t is None:
        kx, bx = pil_coeffs(W, out_w, filt)
        ky, by = pil_coeffs(H, out_h, filt)
        t = tuple(torch.from_numpy(a).to(device) for a in (kx, bx, ky, by))
        torch.cuda.current_stream(device).synchronize()      # built once per geometry; later users may sit on other streams
        with _tab_lock:
            _tab_cache[key] = t
    return t


def pil_resize_u8(img_u8, out_h, out_w, filt="bilinear"):
    """PIL.Image.resize((out_w, out_h), filt) of a uint8 [H, W, C] device tensor, bit-exact; current stream."""
    lib = _lib.load()
    H, W, Cc = img_u8.shape
    kx, bx, ky, by = _tables(H, W, out_h, out_w, filt, img_u8.device)
    out = torch.empty((out_h, out_w, Cc), dtype=torch.uint8, device=img_u8.device)
    need = lib.hgl_resize_pil_bilinear_workspace_bytes(H, out_w, Cc)
    ws = torch.empty((need,), dtype=torch.uint8, device=img_u8.device)
    check(lib.hgl_resize_pil_bilinear(ops._dev(img_u8, torch.uint8, "img"), H, W, Cc, out_h, out_w, kx.data_ptr(), bx.data_ptr(),
                                      kx.shape[1], ky.data_ptr(), by.data_ptr(), ky.shape[1], out.data_ptr(),
                                      ws.data_ptr(), ws.numel(), ops._stream()), "hgl_resize_pil_bilinear")
    return out


def normalize_lut(mean, std):
    """[3, 256] fp32: (v / 255 - mean[c]) / std[c] with the host transform's own operations (ToTensor's division by 255 in
    fp32, then the subtraction and the division of Normalize in fp32)."""
    v = np.arange(256, dtype=np.float32) / np.float32(255)
    m = np.asarray(mean, dtype=np.float32).reshape(-1, 1)
    s = np.asarray(std, dtype=np.float32).reshape(-1, 1)
    return ((v[None, :] - m) / s).astype(np.float32)


_lut_cache = {}


def _lut(mean, std, device):
    key = (tuple(mean), tuple(std), str(device))
    with _tab_lock:
        t = _lut_cache.get(key)
    if t is None:
        t = torch.from_numpy(normalize_lut(mean, std)).to(device)
        torch.cuda.current_stream(device).synchronize()
        with _tab_lock:
            _lut_cache[key] = t
    return t


def to_tensor_normalize(img_u8, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """T.ToTensor() + T.Normalize(mean, std) of a uint8 [H, W, 3] device tensor -> fp32 [3, H, W]; current stream."""
    lib = _lib.load()
    H, W, Cc = img_u8.shape
    out = torch.empty((Cc, H, W), dtype=torch.float32, device=img_u8.device)
    check(lib.hgl_u8_to_chw_lut(ops._dev(img_u8, torch.uint8, "img"), H, W, Cc, _lut(mean, std, img_u8.device).data_ptr(),
                                out.data_ptr(), ops._stream()), "hgl_u8_to_chw_lut")
    return out


def gem_img_transform(img_u8, img_size=448):
    """gem.get_gem_img_transform() on the device: Resize((s, s), bicubic) on uint8 -> ToTensor -> Normalize(OpenAI)."""
    from .gem import OPENAI_MEAN, OPENAI_STD
    return to_tensor_normalize(pil_resize_u8(img_u8, img_size, img_size, "bicubic"), OPENAI_MEAN, OPENAI_STD)


def resize_bilinear(x, size):
    """T.Resize(size) on a float TENSOR [C, h, w] as torchvision 0.15 does it: F.interpolate(bilinear, align_corners=False,
    antialias=False); current stream."""
    lib = _lib.load()
    Cc, h, w = x.shape
    H, W = int(size[0]), int(size[1])
    if (H, W) == (h, w):
        return x
    out = torch.empty((Cc, H, W), dtype=torch.float32, device=x.device)
    check(lib.hgl_resize_bilinear(ops._dev(x, torch.float32, "x"), Cc, h, w, out.data_ptr(), H, W, ops._stream()), "hgl_resize_bilinear")
    return out


def resize_shorter_side(h, w, size):
    """output (h, w) of T.Resize(int) (torchvision _compute_resized_output_size): the shorter side becomes `size`, the
    longer one int(size * long / short)"""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def phrasecut_image_norm(file_img_u8, height, width):
    """image['image'] of the PhraseCut evaluator as its loop uses it: T.Resize(800) on the PIL image (shorter side to 800,
    Pillow bilinear) -> ToTensor -> Normalize(ImageNet) (data/dataset_phrasecut.py:49-51), then T.Resize((height, width)) on
    that tensor (Hybridgl_main_PhraseCut.py:69-70: plain bilinear, no antialias) -> fp32 [3, height, width]."""
    H, W = file_img_u8.shape[:2]
    nh, nw = resize_shorter_side(H, W, 800)
    r = file_img_u8 if (nh, nw) == (H, W) else pil_resize_u8(file_img_u8, nh, nw, "bilinear")
    return resize_bilinear(to_tensor_normalize(r), (height, width))
