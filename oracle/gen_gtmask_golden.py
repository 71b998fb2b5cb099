"""Golden vectors of the ground-truth mask codec, produced by the reference's own C file compiled into
oracle/_ref (make -C oracle).  Run here only:  python -m oracle.gen_gtmask_golden"""
import os

import numpy as np

from oracle import gtmask_oracle as G

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def cases():
    rng = np.random.default_rng(11)
    out = []
    # hand-written: square, degenerate point / segment, triangle with fractional vertices, bow-tie, out of bounds
    out.append((12, 16, [[2, 2, 10, 2, 10, 8, 2, 8]]))
    out.append((8, 8, [[3.0, 3.0]]))
    out.append((8, 8, [[1.0, 1.0, 6.0, 6.0]]))
    out.append((20, 30, [[1.5, 2.25, 25.75, 4.5, 12.1, 18.9]]))
    out.append((24, 24, [[2, 2, 20, 20, 20, 2, 2, 20]]))
    out.append((16, 16, [[-5, -5, 30, 3, 8, 40]]))
    out.append((30, 40, [[2, 2, 20, 2, 20, 15, 2, 15], [10, 8, 35, 8, 35, 28, 10, 28]]))       # overlapping pair -> 2s
    for t in range(40):
        H, W = int(rng.integers(6, 120)), int(rng.integers(6, 160))
        polys = []
        for _ in range(int(rng.integers(1, 4))):
            k = int(rng.integers(3, 12))
            xy = rng.random(2 * k) * np.tile([W + 12, H + 12], k) - 6
            if t % 4 == 0:
                xy = np.round(xy)
            polys.append(np.round(xy, 2).tolist())
        out.append((H, W, polys))
    # COCO-like: a 480x640 image with a smooth 40-gon and a small blob
    th = np.linspace(0, 2 * np.pi, 40, endpoint=False)
    poly = np.stack([320 + 180 * np.cos(th) * (1 + 0.2 * np.sin(5 * th)), 240 + 150 * np.sin(th)], 1).ravel()
    out.append((480, 640, [np.round(poly, 2).tolist(), [600.5, 10.2, 630.1, 12.0, 615.3, 40.7]]))
    return out


def main():
    ref = G.RefMaskApi()
    rec = {}
    cs = cases()
    rec["n_cases"] = np.array([len(cs)])
    for i, (H, W, polys) in enumerate(cs):
        m, area = ref.poly_to_mask(polys, H, W)
        rec[f"c{i}_size"] = np.array([H, W])
        rec[f"c{i}_xy"] = np.concatenate([np.asarray(p, np.float64) for p in polys])
        rec[f"c{i}_npts"] = np.array([len(p) // 2 for p in polys], np.int32)
        rec[f"c{i}_mask"] = m
        rec[f"c{i}_area"] = np.array([area])
    # RLE strings / counts of random blobs
    rng = np.random.default_rng(12)
    strs = []
    for j in range(12):
        H, W = int(rng.integers(4, 90)), int(rng.integers(4, 90))
        yy, xx = np.mgrid[0:H, 0:W]
        m = (((yy - H * rng.random()) ** 2 / (H * 0.3) ** 2 + (xx - W * rng.random()) ** 2 / (W * 0.3) ** 2) < 1).astype(np.uint8)
        if j % 3 == 0:
            m ^= (rng.random((H, W)) < 0.1).astype(np.uint8)
        if j == 5:
            m[:] = 0
        if j == 6:
            m[:] = 1
        s, cnts = ref.encode_to_string(m)
        assert np.array_equal(ref.string_to_mask(s, H, W), m)
        strs.append(s)
        rec[f"r{j}_size"] = np.array([H, W])
        rec[f"r{j}_counts"] = np.array(cnts, np.uint32)
        rec[f"r{j}_mask"] = m
    rec["r_strings"] = np.array(strs)
    rec["n_rle"] = np.array([len(strs)])
    np.savez_compressed(os.path.join(GOLD, "gtmask.npz"), **rec)
    print("gtmask golden:", len(cs), "polygon cases,", len(strs), "RLE cases")


if __name__ == "__main__":
    main()
