"""TEST INFRASTRUCTURE (not shipped, not imported by the product): CPU oracle of the ground-truth mask codec.

* `poly_to_mask` / `rle_string_to_counts`: a plain-Python restatement of refer/external/maskApi.c
  (rleFrPoly :161-201, rleDecode :43-47, rleFrString :217-230) for small cases;
* `RefMaskApi`: ctypes binding of the reference's own C file compiled by oracle/Makefile into oracle/_ref
  (present wherever `make -C oracle` ran with /root/reference available, and on the GPU box as a shipped .so).
Pinned: the restatement agrees with the compiled reference on every vector of tests/golden/gtmask.npz, which
was produced by the compiled reference (oracle/gen_gtmask_golden.py).
"""
import ctypes as C
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(HERE, "_ref", "libmaskapi_ref.so")


def _crossings(xy, H, W):
    """toggle positions x*H + y of one polygon (maskApi.c:161-190)."""
    k = len(xy) // 2
    scale = 5.0
    px = [int(scale * xy[2 * j] + 0.5) for j in range(k)]
    py = [int(scale * xy[2 * j + 1] + 0.5) for j in range(k)]
    px.append(px[0])
    py.append(py[0])
    u, v = [], []
    for j in range(k):
        xs, xe, ys, ye = px[j], px[j + 1], py[j], py[j + 1]
        dx, dy = abs(xe - xs), abs(ys - ye)
        flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
        if flip:
            xs, xe, ys, ye = xe, xs, ye, ys
        if dx >= dy:
            s = (ye - ys) / dx if dx else float("nan")
            for d in range(dx + 1):
                t = dx - d if flip else d
                u.append(t + xs)
                v.append(int(ys + s * t + 0.5) if dx else ys)   # dx == dy == 0: C evaluates (int)NaN; one point, never a crossing
        else:
            s = (xe - xs) / dy
            for d in range(dy + 1):
                t = dy - d if flip else d
                v.append(t + ys)
                u.append(int(xs + s * t + 0.5))
    pos = []
    for j in range(1, len(u)):
        if u[j] == u[j - 1]:
            continue
        xd = float(u[j] if u[j] < u[j - 1] else u[j] - 1)
        xd = (xd + 0.5) / scale - 0.5
        if math.floor(xd) != xd or xd < 0 or xd > W - 1:
            continue
        yd = float(v[j] if v[j] < v[j - 1] else v[j - 1])
        yd = (yd + 0.5) / scale - 0.5
        yd = 0.0 if yd < 0 else (float(H) if yd > H else yd)
        pos.append(int(xd) * H + int(math.ceil(yd)))
    return pos


def poly_to_mask(polys, H, W):
    """list of flat [x0,y0,x1,y1,...] polygons -> (count image [H,W] uint8, summed area), as REFER.getMask."""
    out = np.zeros((H, W), np.uint8)
    area = 0
    for xy in polys:
        pos = sorted(_crossings(list(xy), H, W))
        col = np.zeros(H * W + 1, np.int64)
        for p in pos:
            col[min(p, H * W)] ^= 1
        fill = (np.cumsum(col[:H * W]) & 1).astype(np.uint8)
        area += int(fill.sum())
        out += fill.reshape(W, H).T
    return out, area


def rle_string_to_counts(s):
    """maskApi.c:217-230."""
    cnts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x & 0xFFFFFFFF)
    return cnts


def counts_to_mask(cnts, H, W):
    flat = np.zeros(H * W, np.uint8)
    p = 0
    for j, c in enumerate(cnts):
        if j & 1:
            flat[p:p + c] = 1
        p += c
    return flat.reshape(W, H).T.copy()


class _RLE(C.Structure):
    _fields_ = [("h", C.c_ulong), ("w", C.c_ulong), ("m", C.c_ulong), ("cnts", C.POINTER(C.c_uint))]


class RefMaskApi:
    """The reference's compiled maskApi.c (oracle/_ref)."""

    def __init__(self, path=REF_SO):
        self.lib = C.CDLL(path)
        self.lib.rleToString.restype = C.c_void_p
        self.libc = C.CDLL(None)
        self.libc.free.argtypes = [C.c_void_p]

    def poly_to_mask(self, polys, H, W):
        out = np.zeros((H, W), np.uint8)
        area = 0
        for xy in polys:
            arr = np.ascontiguousarray(xy, dtype=np.float64)
            R = _RLE()
            self.lib.rleFrPoly(C.byref(R), arr.ctypes.data_as(C.POINTER(C.c_double)), C.c_ulong(len(arr) // 2),
                               C.c_ulong(H), C.c_ulong(W))
            buf = np.zeros(H * W, np.uint8)
            self.lib.rleDecode(C.byref(R), buf.ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_ulong(1))
            a = C.c_uint(0)
            self.lib.rleArea(C.byref(R), C.c_ulong(1), C.byref(a))
            area += int(a.value)
            self.lib.rleFree(C.byref(R))
            out += buf.reshape(W, H).T
        return out, area

    def encode_to_string(self, mask):
        """[H,W] 0/1 mask -> compressed RLE string (rleEncode + rleToString)."""
        H, W = mask.shape
        col = np.ascontiguousarray(mask.T, dtype=np.uint8).ravel()
        R = _RLE()
        self.lib.rleEncode(C.byref(R), col.ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_ulong(H), C.c_ulong(W), C.c_ulong(1))
        ptr = self.lib.rleToString(C.byref(R))
        s = C.string_at(ptr).decode("ascii")
        cnts = [int(R.cnts[i]) for i in range(R.m)]
        self.libc.free(ptr)
        self.lib.rleFree(C.byref(R))
        return s, cnts

    def string_to_mask(self, s, H, W):
        R = _RLE()
        self.lib.rleFrString(C.byref(R), C.c_char_p(s.encode("ascii")), C.c_ulong(H), C.c_ulong(W))
        buf = np.zeros(H * W, np.uint8)
        self.lib.rleDecode(C.byref(R), buf.ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_ulong(1))
        self.lib.rleFree(C.byref(R))
        return buf.reshape(W, H).T.copy()


def have_ref():
    return os.path.exists(REF_SO)
