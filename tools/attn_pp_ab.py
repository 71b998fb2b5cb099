#!/usr/bin/env python3
"""A/B of the ping-pong attention kernel (attn_x3pp_kernel) against attn_x3_kernel on the long-sequence shapes: same inputs in
two child processes (HGL_ATTN_PP=1 / 0), outputs compared bit for bit, launches timed.  usage: attn_pp_ab.py"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [("global 4x16 S4096 hd80 rel", 4, 16, 4096, 80, True), ("global 2x16 S1024 hd80 rel", 2, 16, 1024, 80, True),
          ("gem 24x12 S785 hd64", 24, 12, 785, 64, False), ("gem 8x12 S785 hd64", 8, 12, 785, 64, False),
          ("plain 4x8 S1000 hd80", 4, 8, 1000, 80, False), ("plain 3x8 S513 hd64", 3, 8, 513, 64, False)]


def child(out):
    import numpy as np
    import torch
    from hybridgl_amd import ops
    ops.set_precision("f16x3")
    dev = torch.device("cuda:0")
    res = {}
    for name, B, H, S, hd, rel in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(S * 131 + hd)
        q, k, v = (torch.randn(B, S, H * hd, generator=g).to(dev) for _ in range(3))
        kw = {}
        if rel:
            side = int(round(S ** 0.5))
            kw = dict(rel_h=torch.randn(B * H, S, side, generator=g).to(dev), rel_w=torch.randn(B * H, S, side, generator=g).to(dev))
        y = ops.attention(q, k, v, H, **kw)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            ops.attention(q, k, v, H, **kw)
        b.record()
        torch.cuda.synchronize()
        res[name] = (y.cpu().numpy(), a.elapsed_time(b) / 5 * 1e3, 4.0 * B * H * S * S * hd)
    np.savez(out, **{k: v[0] for k, v in res.items()})
    for k, v in res.items():
        print(f"CHILD {k:30s} {v[1]:9.1f} us {v[2] / v[1] / 1e6:7.1f} TF/s", flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        return child(sys.argv[2])
    import numpy as np
    outs = {}
    for pp in ("1", "0"):
        path = f"/tmp/attn_pp_{pp}.npz"
        env = dict(os.environ, HGL_ATTN_PP=pp)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", path], env=env, capture_output=True, text=True, timeout=900)
        print(f"---- HGL_ATTN_PP={pp} (rc {r.returncode})")
        print("\n".join(l for l in r.stdout.splitlines() if l.startswith("CHILD")))
        if r.returncode:
            print(r.stderr[-1500:])
            return 1
        outs[pp] = np.load(path)
    bad = 0
    for name in outs["1"].files:
        a, b = outs["1"][name], outs["0"][name]
        same = np.array_equal(a, b)
        print(f"{name:30s} identical={same} max|diff|={np.abs(a - b).max():.3e} finite={np.isfinite(a).all()}")
        bad += not same
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
