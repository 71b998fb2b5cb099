"""Synthetic, seeded inputs of the benchmark / parity workload.

Pure numpy, deterministic for a given seed on any machine: images (smooth noise), mask
proposals (ellipses / rectangles covering 1-40 % of the image), boxes (the reference's
inclusive XYWH rule, utils/amg.py:303-346), a smooth positive heat-map standing in for the
external GEM output, and token id rows.
"""
import numpy as np


def synth_image(H, W, seed):
    """uint8 [H,W,3]: sum of 8 low-frequency sinusoids per channel + U(0,16) jitter."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.zeros((H, W, 3), dtype=np.float32)
    for c in range(3):
        for _ in range(8):
            fy, fx = rng.uniform(0.5, 4.0, size=2)
            ph = rng.uniform(0, 2 * np.pi)
            amp = rng.uniform(8, 24)
            img[..., c] += amp * np.sin(2 * np.pi * (fy * yy / H + fx * xx / W) + ph)
    img += 128.0
    img += rng.uniform(0, 16, size=img.shape).astype(np.float32)
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_masks(N, H, W, seed):
    """bool [N,H,W]: axis-aligned ellipses (even index) / rectangles (odd), area U(1%,40%)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    out = np.zeros((N, H, W), dtype=bool)
    for n in range(N):
        area = rng.uniform(0.01, 0.40) * H * W
        aspect = rng.uniform(0.5, 2.0)
        if n % 2 == 0:  # ellipse: pi*a*b = area
            a = np.sqrt(area * aspect / np.pi)
            b = area / (np.pi * a)
        else:           # rectangle: (2a)*(2b) = area
            a = np.sqrt(area * aspect) / 2
            b = area / (4 * a)
        a, b = min(a, W / 2 - 1), min(b, H / 2 - 1)
        cx = rng.uniform(a, W - a)
        cy = rng.uniform(b, H - b)
        if n % 2 == 0:
            out[n] = ((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1.0
        else:
            out[n] = (np.abs(xx - cx) <= a) & (np.abs(yy - cy) <= b)
        if not out[n].any():
            out[n, int(cy), int(cx)] = True
    return out


def boxes_from_masks(masks):
    """int64 [N,4] XYWH with w = x1-x0, h = y1-y0 of inclusive pixel coordinates
    (batched_mask_to_box + box_xyxy_to_xywh, utils/amg.py:303-346,91-95)."""
    out = np.zeros((len(masks), 4), dtype=np.int64)
    for i, m in enumerate(masks):
        ys, xs = np.nonzero(m)
        if len(ys):
            out[i] = [xs.min(), ys.min(), xs.max() - xs.min(), ys.max() - ys.min()]
    return out


def synth_heatmap(H, W, seed):
    """fp32 [H,W] smooth positive field (stand-in for the GEM heat-map, Hybridgl_main.py:200)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    f = np.zeros((H, W), dtype=np.float32)
    for _ in range(6):
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        s = rng.uniform(0.08, 0.3) * max(H, W)
        f += rng.uniform(0.3, 1.0) * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
    f += rng.uniform(0, 0.05, size=f.shape).astype(np.float32)
    return f.astype(np.float32)


def synth_tokens(B, context, vocab, seed):
    """int32 [B,context]: SOT, 3..12 random ids, EOT (= vocab-1, the arg-max pooled token), zeros."""
    rng = np.random.default_rng(seed)
    tok = np.zeros((B, context), dtype=np.int32)
    for b in range(B):
        n = int(rng.integers(3, min(13, context - 2)))
        tok[b, 0] = vocab - 2
        tok[b, 1:1 + n] = rng.integers(1, vocab - 2, size=n)
        tok[b, 1 + n] = vocab - 1
    return tok


# parse records cycled by the synthetic workload: (dirflag, relaflag, n_other_nouns)
PARSE_RECORDS = [("left", "left", 1), ("none", "big", 0), ("middle", "none", 2)]


def imagenet_normalize(img_u8):
    """dataset transform (data/dataset_refer_bert.py:155): ToTensor + Normalize(ImageNet) -> [3,H,W]."""
    x = img_u8.astype(np.float32) / np.float32(255)
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32)
    return ((x - mean) / std).transpose(2, 0, 1).astype(np.float32).copy()


def box_blur_u8(img, k=15):
    """Deterministic stand-in for cv2.GaussianBlur(img,(15,15),0) (OpenCV is absent offline and
    its fixed-point kernel is unpinned): separable Gaussian, sigma per OpenCV's
    rule 0.3*((k-1)*0.5-1)+0.8, reflect-101 borders, round-half-up to uint8."""
    sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    r = k // 2
    x = np.arange(-r, r + 1, dtype=np.float64)
    g = np.exp(-(x * x) / (2 * sigma * sigma))
    g /= g.sum()
    a = img.astype(np.float64)
    pad = np.pad(a, ((r, r), (0, 0), (0, 0)), mode="reflect")
    a = sum(g[i] * pad[i:i + img.shape[0]] for i in range(k))
    pad = np.pad(a, ((0, 0), (r, r), (0, 0)), mode="reflect")
    a = sum(g[i] * pad[:, i:i + img.shape[1]] for i in range(k))
    return np.clip(np.floor(a + 0.5), 0, 255).astype(np.uint8)
