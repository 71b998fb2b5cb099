"""GPU parity of every primitive kernel against the numpy oracle, through the C ABI."""
import os

import numpy as np
import pytest
import torch

from hybridgl_amd import ops
from oracle import clip_oracle as O
from oracle.cases import RESIZE_CASES, TAIL_CASES, resize_case, tail_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture(params=["f32", "f16x3"])
def precision(request, cuda):
    """run a test under both matrix-core modes (fp32 MFMA / split-fp16 MFMA) and restore the mode"""
    ops.set_precision(request.param)
    yield request.param
    ops.set_precision(ops.default_precision())


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (197, 2304, 768), (64, 512, 768), (1, 4, 256),
                                   (300, 96, 100), (1000, 136, 3072), (77, 32, 64)])
@pytest.mark.parametrize("act", ["none", "quickgelu", "gelu", "relu"])
def test_gemm(cuda, M, N, K, act):
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    r = rng.standard_normal((M, N)).astype(np.float32)
    y = ops.gemm(T(a, cuda), T(w, cuda), T(b, cuda), T(r, cuda), act).cpu().numpy()
    z = a.astype(np.float64) @ w.astype(np.float64).T + b
    if act == "quickgelu":
        z = z / (1 + np.exp(-1.702 * z))
    elif act == "gelu":
        from scipy.special import erf
        z = 0.5 * z * (1 + erf(z / np.sqrt(2)))
    elif act == "relu":
        z = np.maximum(z, 0)
    z = z + r
    np.testing.assert_allclose(y, z, rtol=0, atol=2e-5)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (197, 2304, 768), (300, 96, 128), (1000, 136, 3072), (77, 32, 64)])
@pytest.mark.parametrize("act", ["none", "quickgelu", "gelu", "relu"])
def test_gemm_f16x3(cuda, M, N, K, act):
    """split-fp16 matrix-core GEMM: fp32-class accuracy (tolerance 1.5x the fp32 kernel's)."""
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    a = (rng.standard_normal((M, K)) * rng.choice([0.01, 1.0, 30.0], size=(M, 1))).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    r = rng.standard_normal((M, N)).astype(np.float32)
    y = ops.gemm_f16x3(T(a, cuda), T(w, cuda), T(b, cuda), T(r, cuda), act).cpu().numpy()
    z = a.astype(np.float64) @ w.astype(np.float64).T + b
    if act == "quickgelu":
        z = z / (1 + np.exp(-1.702 * z))
    elif act == "gelu":
        from scipy.special import erf
        z = 0.5 * z * (1 + erf(z / np.sqrt(2)))
    elif act == "relu":
        z = np.maximum(z, 0)
    z = z + r
    # error budget relative to the magnitude of the row's terms (rows are scaled by 0.01 / 1 / 30)
    scale = np.maximum(1.0, np.abs(z)) * np.maximum(1.0, np.abs(a).max(axis=1, keepdims=True) / 4)
    assert (np.abs(y - z) / scale).max() < 3e-5


@pytest.mark.parametrize("M,N,K", [(300, 200, 64), (777, 1000, 192), (1031, 520, 1280), (64, 3000, 128)])
def test_gemm_f16x3_tilings_bit_identical(cuda, M, N, K):
    """Every tiling of the f16x3 GEMM (register-staged 128x128, LDS-DMA ping-pong 256x256) accumulates in the
    same order: outputs must be bit-identical, ragged edges included, and the cost model must pick one of them."""
    rng = np.random.default_rng(M + N + K)
    a = T(rng.standard_normal((M, K)).astype(np.float32), cuda)
    w = T((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32), cuda)
    b = T(rng.standard_normal(N).astype(np.float32), cuda)
    r = T(rng.standard_normal((M, N)).astype(np.float32), cuda)
    try:
        outs = {}
        for kind in ("v1", "P", "auto"):
            ops.select_x3_kernel(kind)
            outs[kind] = ops.gemm_f16x3(a, w, b, r, "gelu").cpu().numpy()
    finally:
        ops.select_x3_kernel("auto")
    for kind in ("P", "auto"):
        assert np.array_equal(outs[kind], outs["v1"]), kind
    z = a.cpu().numpy().astype(np.float64) @ w.cpu().numpy().astype(np.float64).T + b.cpu().numpy()
    from scipy.special import erf
    z = 0.5 * z * (1 + erf(z / np.sqrt(2))) + r.cpu().numpy()
    assert np.abs(outs["v1"] - z).max() < 3e-5


def test_gemm_f16x3_tilings_fuzz_many_tiles(cuda):
    """The ping-pong kernel counts its waits by hand (LDS-DMA units that run across output-tile boundaries, write-out
    stores, the bias load -- all on one in-order counter); the register-staged kernel's waits are the compiler's.  An
    under-count would read stale data: random shapes with MANY tiles per persistent workgroup, each repeated, must
    match bit for bit (tools/x3_fuzz.py is the long form: 2 500 cases)."""
    rng = np.random.default_rng(11)
    try:
        for c in range(40):
            K = 64 * int(rng.integers(1, 33))
            N = 4 * int(rng.integers(16, 700))
            M = int(rng.integers(2000, 120000))
            if M * (N + K) > 200e6:
                M = int(200e6 // (N + K))
            act = ("none", "quickgelu", "gelu", "relu")[int(rng.integers(0, 4))]
            a = torch.randn(M, K, device=cuda)
            w = torch.randn(N, K, device=cuda) / K ** 0.5
            b = torch.randn(N, device=cuda) if rng.random() < 0.7 else None
            r = torch.randn(M, N, device=cuda) if rng.random() < 0.5 else None
            ops.select_x3_kernel("v1")
            ref = ops.gemm_f16x3(a, w, b, r, act)
            ops.select_x3_kernel("P")
            for rep in range(2):
                assert torch.equal(ops.gemm_f16x3(a, w, b, r, act), ref), (c, rep, M, N, K, act)
            torch.cuda.synchronize()
            ops.release_split_weights([w.data_ptr()])
    finally:
        ops.select_x3_kernel("auto")


def test_gemm_f16x3_inplace_residual(cuda):
    """residual aliasing the output (x += proj(...)) on every tiling"""
    rng = np.random.default_rng(5)
    M, N, K = 520, 384, 128
    a = T(rng.standard_normal((M, K)).astype(np.float32), cuda)
    w = T((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32), cuda)
    c0 = rng.standard_normal((M, N)).astype(np.float32)
    try:
        ref = None
        for kind in ("v1", "P"):
            ops.select_x3_kernel(kind)
            c = T(c0, cuda)
            ops.gemm_f16x3(a, w, None, c, "none", out=c)
            y = c.cpu().numpy()
            ref = y if ref is None else ref
            assert np.array_equal(y, ref), kind
    finally:
        ops.select_x3_kernel("auto")
    z = a.cpu().numpy().astype(np.float64) @ w.cpu().numpy().astype(np.float64).T + c0
    assert np.abs(ref - z).max() < 3e-5


@pytest.mark.parametrize("M,N,K,act", [(2364 // 3 * 256 - 128, 768, 768, "none"),       # CLIP out over 16 refs: 9.23 rounds, but K < 1024: plain
                                        (2364 // 3 * 256 - 128, 768, 3072, "none"),      # CLIP fc2 over 16 refs: 788 x 3 tiles = 9.23 rounds
                                        (160 * 256, 1280, 1280, "none"),               # SAM proj over 10 images: 800 tiles = 3.125 rounds
                                        (160 * 256, 1280, 5120, "gelu"),               # lin2's shape (with an activation, for the reduce pass)
                                        (256 * 256, 1280, 1280, "none")])              # 1280 tiles = 5 rounds exactly: the plain launch
def test_gemm_f16x3_row_balanced_launch(cuda, M, N, K, act):
    """hgl_launch_gemm_f16x3_balanced (the residual GEMMs of the CLIP / SAM blocks): a GEMM whose last round of the persistent
    256 x 256 tiling would be mostly empty runs as whole rounds + a split-K tail over the last row tiles.  Against the plain
    launch: the main rows bit for bit, the tail rows (K slices summed in index order) to fp32 rounding; in place on the
    residual, as the blocks call it; and against a float64 product on a sample of rows."""
    if ops.default_precision() != "f16x3":
        pytest.skip("the split-fp16 GEMM")
    g = torch.Generator(device=cuda).manual_seed(M % 1000 + K)
    a = torch.randn((M, K), device=cuda, generator=g)
    w = torch.randn((N, K), device=cuda, generator=g) / K ** 0.5
    b = torch.randn((N,), device=cuda, generator=g)
    r = torch.randn((M, N), device=cuda, generator=g)
    try:
        plain = ops.gemm_f16x3(a, w, b, r, act)
        x = r.clone()
        ops.gemm_f16x3(a, w, b, x, act, out=x, balanced=True)           # x = act(a w^T + b) + x
        torch.cuda.synchronize()
        same = (x == plain).all(dim=1)
        n_diff = int((~same).sum())
        tiles_n, ncu = (N + 255) // 256, torch.cuda.get_device_properties(cuda).multi_processor_count
        tiles = (M + 255) // 256 * tiles_n
        rounds = -(-tiles // ncu)
        rem = tiles - (rounds - 1) * ncu
        if rem * 2 > ncu or rem == 0 or K < 1024 or rounds < 3:
            assert n_diff == 0                                           # no tail: the plain launch
        else:
            m_main = (rounds - 1) * ncu // tiles_n * 256
            assert same[:m_main].all()                                   # main rows: the same kernel on the same rows
            assert 0 < n_diff <= M - m_main                              # the tail went through K slices
            scale = float(plain.abs().max())
            assert float((x - plain).abs().max()) <= 4e-6 * scale
        rows = torch.tensor([0, 1, M // 2, M - 300, M - 2, M - 1], device=cuda)
        z = a[rows].double() @ w.double().T + b.double()
        if act == "gelu":
            z = torch.nn.functional.gelu(z)
        z = z + r[rows].double()
        assert float((x[rows].double() - z).abs().max()) < 3e-5 * max(1.0, float(z.abs().max()))
    finally:
        ops.release_split_weights([w.data_ptr()])


def test_gemm_f16x3_operand_planes_beyond_4gb(cuda):
    """an A plane of 4 GB or more (CLIP ViT-L/14 fc2 over a group of 16 refs) runs as row chunks on the LDS-DMA kernel, whose
    operand offsets are 32-bit: every row equals the row of a small GEMM over its neighbourhood, at the chunk borders too"""
    M, N, K = 1_050_000, 256, 2048                      # 4.3 GB per fp16 plane
    g = torch.Generator(device=cuda).manual_seed(7)
    a = torch.randn((M, K), device=cuda, generator=g)
    w = torch.randn((N, K), device=cuda, generator=g) / 45.0
    bias = torch.randn((N,), device=cuda, generator=g)
    r = torch.randn((M, N), device=cuda, generator=g)
    out = ops.gemm_f16x3(a, w, bias, r, "gelu")
    chunk = int(3.9e9 / (K * 2.0)) // 256 * 256
    for m0 in (0, chunk - 300, chunk, M - 700):
        sl = slice(m0, min(M, m0 + 700))
        small = ops.gemm_f16x3(a[sl].contiguous(), w, bias, r[sl].contiguous(), "gelu")
        assert torch.equal(small, out[sl]), m0
    ref = torch.nn.functional.gelu(a[:64].double() @ w.double().T + bias.double()) + r[:64].double()
    assert float((out[:64].double() - ref).abs().max()) < 2e-4
    del a, r, out
    ops.release_split_weights([w.data_ptr()])
    torch.cuda.empty_cache()


def test_gemm_inplace_residual_and_asymmetry(cuda):
    """A = I with an asymmetric W catches a transposed C write; residual aliasing C must work."""
    n = 160
    w = np.arange(n * n, dtype=np.float32).reshape(n, n) / (n * n)
    x = T(np.eye(n, dtype=np.float32), cuda)
    c = T(np.ones((n, n), np.float32), cuda)
    ops.gemm(x, T(w, cuda), None, c, "none", out=c)
    np.testing.assert_allclose(c.cpu().numpy(), w.T + 1, rtol=0, atol=1e-6)


@pytest.mark.parametrize("D", [256, 512, 768, 1280, 96, 128])
def test_layernorm(cuda, D):
    rng = np.random.default_rng(D)
    x = (rng.standard_normal((301, D)) * 3 + 1).astype(np.float32)
    w = rng.standard_normal(D).astype(np.float32)
    b = rng.standard_normal(D).astype(np.float32)
    for eps in (1e-5, 1e-6):
        y = ops.layernorm(T(x, cuda), T(w, cuda), T(b, cuda), eps).cpu().numpy()
        np.testing.assert_allclose(y, O.layer_norm(x, w, b, eps), rtol=0, atol=2e-5)


def _attn_ref(q, k, v, heads, scale, add_mask=None, bias=None):
    B, Sq, D = q.shape
    Sk = k.shape[1]
    hd = D // heads
    qh = q.reshape(B, Sq, heads, hd).transpose(0, 2, 1, 3).astype(np.float64) * scale
    kh = k.reshape(B, Sk, heads, hd).transpose(0, 2, 1, 3).astype(np.float64)
    vh = v.reshape(B, Sk, heads, hd).transpose(0, 2, 1, 3).astype(np.float64)
    s = qh @ kh.transpose(0, 1, 3, 2)
    if bias is not None:
        s = s + bias
    if add_mask is not None:
        s = s + add_mask
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    return (p @ vh).transpose(0, 2, 1, 3).reshape(B, Sq, D)


@pytest.mark.parametrize("B,heads,S,hd", [(3, 12, 197, 64), (2, 2, 17, 64), (2, 16, 196, 80), (5, 8, 7, 32),
                                          (2, 8, 300, 16), (1, 4, 64, 64), (1, 2, 129, 64)])
def test_attention_plain(cuda, precision, B, heads, S, hd):
    rng = np.random.default_rng(B * 100 + S)
    D = heads * hd
    q, k, v = (rng.standard_normal((B, S, D)).astype(np.float32) for _ in range(3))
    y = ops.attention(T(q, cuda), T(k, cuda), T(v, cuda), heads).cpu().numpy()
    np.testing.assert_allclose(y, _attn_ref(q, k, v, heads, hd ** -0.5), rtol=0, atol=2e-5)


def test_attention_cross_lengths(cuda, precision):
    """decoder shapes: few queries x many keys and the reverse (transformer.py:185-240)."""
    rng = np.random.default_rng(5)
    for Sq, Sk, heads, hd in [(7, 4096, 8, 16), (4096, 7, 8, 16), (33, 95, 4, 32)]:
        D = heads * hd
        q = rng.standard_normal((2, Sq, D)).astype(np.float32)
        k = rng.standard_normal((2, Sk, D)).astype(np.float32)
        v = rng.standard_normal((2, Sk, D)).astype(np.float32)
        y = ops.attention(T(q, cuda), T(k, cuda), T(v, cuda), heads).cpu().numpy()
        np.testing.assert_allclose(y, _attn_ref(q, k, v, heads, hd ** -0.5), rtol=0, atol=2e-5)


def test_attention_causal(cuda, precision):
    rng = np.random.default_rng(9)
    B, heads, S, hd = 4, 8, 77, 64
    D = heads * hd
    q, k, v = (rng.standard_normal((B, S, D)).astype(np.float32) for _ in range(3))
    mask = np.triu(np.full((S, S), -np.inf), 1)
    y = ops.attention(T(q, cuda), T(k, cuda), T(v, cuda), heads, mask="causal").cpu().numpy()
    np.testing.assert_allclose(y, _attn_ref(q, k, v, heads, hd ** -0.5, mask), rtol=0, atol=2e-5)


def test_attention_cls_keep_with_offsets(cuda, precision):
    """keep bytes apply to batches >= keep_b0, row (b-keep_b0)%keep_n; empty keep row -> CLS sees itself."""
    rng = np.random.default_rng(10)
    B, heads, S, hd, n = 6, 12, 197, 64, 2
    D = heads * hd
    q, k, v = (rng.standard_normal((B, S, D)).astype(np.float32) for _ in range(3))
    keep = rng.random((n, S - 1)) > 0.5
    keep[1] = False
    add = np.zeros((B, 1, S, S))
    for b in range(2, B):
        add[b, 0, 0, 1:] = np.where(keep[(b - 2) % n], 0, -np.inf)
    y = ops.attention(T(q, cuda), T(k, cuda), T(v, cuda), heads, mask="cls_keep", keep=T(keep, cuda),
                      keep_b0=2, keep_n=n).cpu().numpy()
    np.testing.assert_allclose(y, _attn_ref(q, k, v, heads, hd ** -0.5, add), rtol=0, atol=2e-5)


@pytest.mark.parametrize("path", ["f32", "f16x3"])
def test_epilogue_activations_accuracy(cuda, path):
    """hgl_gelu_erf / hgl_quick_gelu (hgl_common.h) through a GEMM with the identity as weight: erf-GELU within 2.5e-7 *
    max(1, |x|) of the double-precision function on the whole axis (one float rounding of erf alone is 6e-8; a float
    erff-based evaluation -- the reference's -- is no closer); QuickGELU within 3e-6 relative (the rounding of the
    exponential's argument, 1.702 |x| <= 15, is 1e-6 of that in any float evaluation)."""
    from scipy import special
    n = 256
    x = np.concatenate([np.linspace(-9, 9, n * n - 4096), np.random.default_rng(0).standard_normal(4096) * 2]).astype(np.float32)
    a = x.reshape(n, n)
    eye = np.eye(n, dtype=np.float32)
    fn = ops.gemm if path == "f32" else ops.gemm_f16x3
    xd = a.astype(np.float64)
    if path == "f16x3":      # the operand is split into fp16 hi + lo: representable only to 2^-22; compare on what the GEMM saw
        hi = a.astype(np.float16).astype(np.float32)
        xd = (hi + (a - hi).astype(np.float16).astype(np.float32)).astype(np.float64)
    y = fn(T(a, cuda), T(eye, cuda), act="gelu").cpu().numpy().astype(np.float64)
    ref = 0.5 * xd * special.erfc(-xd / np.sqrt(2.0))
    assert np.abs(y - ref).max() < 2e-7 * 9, np.abs(y - ref).max()
    assert (np.abs(y - ref) <= 2.5e-7 * np.maximum(1.0, np.abs(xd))).all()
    # the negative tail: nn.GELU tends to -0, not to a multiple of x (erfc is clamped at t = 4 inside the fit)
    far = np.zeros((n, n), dtype=np.float32)
    far[0, :6] = [-10.0, -100.0, -1.0e4, -6.0, 10.0, 1.0e4]
    yf = fn(T(far, cuda), T(eye, cuda), act="gelu").cpu().numpy()[0, :6].astype(np.float64)
    assert (np.abs(yf[:3]) == 0).all() and abs(yf[3] - (-6.0 * 0.5 * special.erfc(6.0 / np.sqrt(2.0)))) < 5e-8, yf
    assert yf[4] == 10.0 and yf[5] == 1.0e4
    q = fn(T(a, cuda), T(eye, cuda), act="quickgelu").cpu().numpy().astype(np.float64)
    qref = xd / (1.0 + np.exp(-1.702 * xd))
    big = np.abs(qref) > 1e-30
    assert (np.abs(q - qref)[big] <= 3e-6 * np.abs(qref[big]) + 1e-37).all(), (np.abs(q - qref)[big] / np.abs(qref[big])).max()


@pytest.mark.parametrize("Sk", [300, 600])
def test_attention_cls_keep_more_keys_than_queries(cuda, precision, Sk):
    """197 queries (the 8-wave kernels' range) against more than 257 keys under the CLS keep mask: the persistent wide
    kernel parks an item's keep row in a 256-byte LDS tail and must not be chosen here (reachable through hgl_attention_f32)."""
    rng = np.random.default_rng(14)
    B, heads, Sq, hd, n = 3, 2, 197, 64, 3
    D = heads * hd
    q = rng.standard_normal((B, Sq, D)).astype(np.float32)
    k, v = (rng.standard_normal((B, Sk, D)).astype(np.float32) for _ in range(2))
    keep = rng.random((n, Sk - 1)) > 0.5
    add = np.zeros((B, 1, Sq, Sk))
    for b in range(B):
        add[b, 0, 0, 1:] = np.where(keep[b % n], 0, -np.inf)
    y = ops.attention(T(q, cuda), T(k, cuda), T(v, cuda), heads, mask="cls_keep", keep=T(keep, cuda),
                      keep_b0=0, keep_n=n).cpu().numpy()
    np.testing.assert_allclose(y, _attn_ref(q, k, v, heads, hd ** -0.5, add), rtol=0, atol=2e-5)


def test_attention_spiked_scores_online_softmax(cuda, precision):
    """force the running max to jump at a late key tile (rescale branch of the online softmax)."""
    rng = np.random.default_rng(12)
    B, heads, S, hd = 1, 1, 200, 64
    q, k, v = (rng.standard_normal((B, S, hd)).astype(np.float32) for _ in range(3))
    k[0, 150] = q[0, 3] * 4
    k[0, 199] = q[0, 100] * 6
    y = ops.attention(T(q, cuda), T(k, cuda), T(v, cuda), heads).cpu().numpy()
    np.testing.assert_allclose(y, _attn_ref(q, k, v, heads, hd ** -0.5), rtol=0, atol=2e-5)


@pytest.mark.parametrize("B,heads,S,hd", [(2, 16, 196, 80), (1, 16, 1024, 80), (4, 12, 197, 64)])
def test_attention_no_rare_row_outliers(cuda, B, heads, S, hd):
    """Regression: the fp16 hi+lo split of a COMPUTED value (q * scale, LayerNorm outputs, GEMM epilogues) must use
    one rounded fp32 value for both halves.  When the compiler folded the multiply into only one of the two
    conversions, ties made hi + lo miss the value by 2*|lo| and 0.1 % of the attention rows were off by up to 8e-5
    (head dim 80, whose scale is not a power of two).  Every row must stay at the 1e-6 level."""
    from hybridgl_amd import ops
    ops.set_precision("f16x3")
    try:
        worst = 0.0
        for seed in range(3):
            rng = np.random.default_rng(1000 + seed)
            D = heads * hd
            q, k, v = (rng.standard_normal((B, S, D)).astype(np.float32) for _ in range(3))
            y = ops.attention(T(q, cuda), T(k, cuda), T(v, cuda), heads).cpu().numpy()
            worst = max(worst, float(np.abs(y - _attn_ref(q, k, v, heads, hd ** -0.5)).max()))
        assert worst < 3e-6, worst
    finally:
        ops.set_precision(ops.default_precision())


@pytest.mark.parametrize("kh,kw,hd", [(14, 14, 80), (6, 10, 64), (4, 32, 80), (5, 64, 32)])
def test_attention_rel_pos_bias(cuda, precision, kh, kw, hd):
    """decomposed relative position bias tables (image_encoder.py:325-361): the 14x14 / hd 80 window (bias on the
    matrix cores in f16x3 mode), a ragged window (generic path) and row widths that are multiples of 32 (vector path)."""
    rng = np.random.default_rng(13 + kh)
    B, heads = 2, 3
    S, D = kh * kw, heads * hd
    q, k, v = (rng.standard_normal((B, S, D)).astype(np.float32) for _ in range(3))
    rel_h = rng.standard_normal((B * heads, S, kh)).astype(np.float32)
    rel_w = rng.standard_normal((B * heads, S, kw)).astype(np.float32)
    bias = (rel_h[:, :, :, None] + rel_w[:, :, None, :]).reshape(B, heads, S, S)
    y = ops.attention(T(q, cuda), T(k, cuda), T(v, cuda), heads, rel_h=T(rel_h, cuda), rel_w=T(rel_w, cuda)).cpu().numpy()
    np.testing.assert_allclose(y, _attn_ref(q, k, v, heads, hd ** -0.5, None, bias), rtol=0, atol=3e-5)


@pytest.mark.parametrize("i", range(3))
def test_mask_resize_matches_golden(cuda, golden_dir, i):
    import os
    g = np.load(os.path.join(golden_dir, "resize.npz"))
    x, (H, W, oh, ow) = resize_case(i)
    pm = ops.mask_resize(T(x.astype(bool), cuda), oh).cpu().numpy().reshape(-1, oh, ow)
    np.testing.assert_allclose(pm, g[f"r{i}_out"], rtol=0, atol=1e-6)
    assert np.array_equal(pm != 0, g[f"r{i}_out"] != 0)


def test_calculate_score(cuda):
    rng = np.random.default_rng(14)
    img = rng.standard_normal((64, 512)).astype(np.float32)
    txt = rng.standard_normal((3, 512)).astype(np.float32)
    y = ops.calculate_score(T(img, cuda), T(txt, cuda), 100.0).cpu().numpy()
    np.testing.assert_allclose(y, O.calculate_score(img, txt, 100.0), rtol=0, atol=1e-4)


@pytest.mark.parametrize("ci", range(len(TAIL_CASES)))
def test_scoring_tail_vs_golden(cuda, golden_dir, ci):
    import os
    g = np.load(os.path.join(golden_dir, "scoring.npz"))
    H, W, N = (int(v) for v in g["tail_hw"])
    rela, dirflag, has_other = TAIL_CASES[ci]
    hybrid, t_pos, t_neg, masks, boxes, attn, gt = tail_case(ci, N, 32, H, W)
    black = 1.95 if rela == "big" else (1.5 if rela == "small" else 1.8)
    gem = ops.coherence_scores(T(attn, cuda), T(masks, cuda), dirflag, black)
    np.testing.assert_allclose(gem.cpu().numpy(), g[f"tail{ci}_gem"], rtol=2e-5, atol=2e-5)
    # sentence == noun phrase == t_pos makes the r-ensemble exact; one "other noun" row = t_neg
    idx, sc, sn = ops.score_sentence(T(hybrid, cuda), T(t_pos, cuda), T(t_pos, cuda), T(t_neg, cuda), T(boxes, cuda),
                                     gem, float(g["cs_logit_scale"]), 0.5, 3, 6, 0.6, rela, has_other)
    np.testing.assert_allclose(sc.cpu().numpy(), g[f"tail{ci}_sc"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(sn.cpu().numpy(), g[f"tail{ci}_sn"], rtol=0, atol=1e-3)
    assert [int(v) for v in idx.cpu()] == [int(v) for v in g[f"tail{ci}_idx"]]   # bit-exact indices
    iu = ops.iou_counts(T(masks[int(idx[1])], cuda), T(gt, cuda)).cpu().numpy()
    assert tuple(int(v) for v in iu) == tuple(int(v) for v in g[f"tail{ci}_IU"])
    iu2 = ops.iou_select(T(masks, cuda), idx, 1, T(gt, cuda)).cpu().numpy()   # device-side selection
    assert tuple(int(v) for v in iu2) == tuple(int(v) for v in iu)


def test_score_sentence_ensemble_and_no_other_nouns(cuda):
    """text ensemble r*s+(1-r)*np and the mean of K other-noun rows are formed in the kernel;
    with no other nouns the negative logits are NaN (as in the reference) and are unused."""
    rng = np.random.default_rng(21)
    N, E = 20, 512
    hybrid = rng.standard_normal((N, E)).astype(np.float32)
    s, p = rng.standard_normal((2, E)).astype(np.float32)
    oth = rng.standard_normal((3, E)).astype(np.float32)
    from hybridgl_amd.synth import boxes_from_masks, synth_masks
    boxes = boxes_from_masks(synth_masks(N, 64, 64, 1))
    gem = rng.standard_normal(N).astype(np.float32)
    for others, has in [(oth, True), (None, False)]:
        idx, sc, sn = ops.score_sentence(T(hybrid, cuda), T(s, cuda), T(p, cuda),
                                         T(others, cuda) if others is not None else None, T(boxes, cuda),
                                         T(gem, cuda), 100.0, 0.5, 3, 6, 0.6, "left", has)
        tpos = (np.float32(0.5) * s + np.float32(0.5) * p)[None]
        tneg = (others.sum(0, dtype=np.float32) / np.float32(3))[None] if others is not None else np.ones((1, E), np.float32)
        ip, ifin, rsc, rsn = O.score_sentence(hybrid, tpos, tneg, boxes, gem, 100.0, 3, 6, 0.6, "left", has)
        np.testing.assert_allclose(sc.cpu().numpy(), rsc, rtol=0, atol=1e-3)
        assert [int(v) for v in idx.cpu()] == [ip, ifin]
        if others is None:
            assert torch.isnan(sn).all()
        else:
            np.testing.assert_allclose(sn.cpu().numpy(), rsn, rtol=0, atol=1e-3)


def test_coherence_full_size_and_edges(cuda):
    """640x640 x 64 masks (BASELINE size) vs the oracle; odd sizes exercise the unaligned path."""
    from hybridgl_amd.synth import synth_heatmap, synth_masks
    for (N, H, W, flag) in [(64, 640, 640, "middle"), (5, 37, 53, "left"), (3, 427, 640, "right")]:
        masks = synth_masks(N, H, W, 1)
        attn = synth_heatmap(H, W, 2)
        y = ops.coherence_scores(T(attn, cuda), T(masks, cuda), flag, 1.8).cpu().numpy()
        np.testing.assert_allclose(y, O.coherence_scores(attn, masks, flag, 1.8), rtol=2e-5, atol=2e-5)


def test_iou_sizes_and_properties(cuda):
    rng = np.random.default_rng(15)
    for n in [1, 15, 16, 17, 640 * 640, 427 * 640 + 3]:
        p = rng.random(n) > 0.5
        g = rng.random(n) > 0.3
        iu = ops.iou_counts(T(p, cuda), T(g, cuda)).cpu().numpy()
        assert tuple(iu) == O.compute_iou(p, g)
        same = ops.iou_counts(T(p, cuda), T(p, cuda)).cpu().numpy()
        assert same[0] == same[1] == p.sum()
    z = np.zeros(100, bool)
    assert tuple(ops.iou_counts(T(z, cuda), T(z, cuda)).cpu().numpy()) == (0, 0)
    # nonzero bytes other than 1 count as True (target.type(torch.bool), utils.py:367-368)
    a = np.array([0, 2, 255, 1] * 8, np.uint8)
    b = np.array([1, 0, 7, 0] * 8, np.uint8)
    assert tuple(ops.iou_counts(T(a, cuda), T(b, cuda)).cpu().numpy()) == (8, 32)


def test_synthesize_views(cuda):
    from hybridgl_amd.synth import box_blur_u8, imagenet_normalize, synth_image, synth_masks
    for (N, H, W, res) in [(4, 97, 130, 64), (3, 640, 640, 224)]:
        img = synth_image(H, W, 3)
        blur = box_blur_u8(img)
        norm = imagenet_normalize(img)
        masks = synth_masks(N, H, W, 4)
        loc, glo = ops.synthesize_views(T(img, cuda), T(blur, cuda), T(norm, cuda), T(masks, cuda), res)
        rl, rg = O.synthesize_views(img, blur, norm, masks, res)
        np.testing.assert_allclose(loc.cpu().numpy(), rl, rtol=0, atol=2e-6)
        np.testing.assert_allclose(glo.cpu().numpy(), rg, rtol=0, atol=5e-6)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_synthesize_views_vs_reference_loop_golden(cuda, golden_dir, tag):
    """device view synthesis (with the device's own fixed-point blur in front) vs tests/golden/views.npz, the loop of
    Hybridgl_main.py:93-125 with its torch arithmetic pinned (ragged masks: single pixel, full, thin line, empty)"""
    import os
    from hybridgl_amd.synth import imagenet_normalize, synth_image
    from oracle.cases import edge_masks
    g = np.load(os.path.join(golden_dir, "views.npz"))
    H, W, N, res, s_img, s_mask = (int(v) for v in g[f"{tag}_meta"])
    img = synth_image(H, W, s_img)
    masks = edge_masks(N, H, W, s_mask)
    blur = ops.gaussian_blur_u8(T(img, cuda), 15)
    loc, glo = ops.synthesize_views(T(img, cuda), blur, T(imagenet_normalize(img), cuda), T(masks, cuda), res)
    np.testing.assert_allclose(glo.cpu().numpy(), g[f"{tag}_global"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(loc.cpu().numpy(), g[f"{tag}_local"], rtol=0, atol=2e-6)


def test_errors_are_loud(cuda):
    from hybridgl_amd._lib import HybridGLError
    with pytest.raises(HybridGLError):
        ops.gemm(torch.zeros(4, 6, device=cuda), torch.zeros(4, 6, device=cuda))  # K % 4 != 0
    with pytest.raises(HybridGLError):
        ops.layernorm(torch.zeros(2, 8), torch.ones(8), torch.zeros(8))  # CPU tensor
    with pytest.raises(HybridGLError):
        ops.attention(*(torch.zeros(1, 8, 24, device=cuda) for _ in range(3)), heads=1)  # hd=24


def test_split_overflow_is_counted_and_raised(cuda):
    """f16x3 mode: an activation beyond the fp16 range (|x| > 65504) cannot be split; the kernels fold |x| into a
    per-thread running maximum and count the thread (hgl_split_overflow_count), the Python layer raises -- inf / NaN
    never leaves the library silently.  A NaN input stays a NaN and is not an overflow."""
    from hybridgl_amd._lib import HybridGLError
    rng = np.random.default_rng(3)
    M, N, K = 256, 128, 128
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    ops.split_overflow_count(reset=True)
    y = ops.gemm_f16x3(T(a, cuda), T(w, cuda)).cpu().numpy()
    assert ops.split_overflow_count(reset=True) == 0 and np.isfinite(y).all()
    a2 = a.copy()
    a2[3, 5] = 1.0e6
    a2[7, 9] = -3.0e5
    y2 = ops.gemm_f16x3(T(a2, cuda), T(w, cuda)).cpu().numpy()
    assert not np.isfinite(y2[[3, 7]]).any()                      # inf - inf inside the split: the rows are lost ...
    assert np.array_equal(y2[[0, 1, 2, 4]], y[[0, 1, 2, 4]])      # ... the other rows untouched ...
    assert ops.split_overflow_count(reset=False) == 2             # ... and the two offending threads are counted
    with pytest.raises(HybridGLError, match="fp16 range"):
        ops.check_split_overflow()
    assert ops.split_overflow_count() == 0                        # the check resets
    a3 = a.copy()
    a3[2, 2] = np.nan
    y3 = ops.gemm_f16x3(T(a3, cuda), T(w, cuda)).cpu().numpy()
    assert np.isnan(y3[2]).all() and np.isfinite(y3[3]).all() and ops.split_overflow_count() == 0
    a4 = a.copy()
    a4[1, 1] = 65504.0                                            # the fp16 maximum itself is a member of the split
    assert np.isfinite(ops.gemm_f16x3(T(a4, cuda), T(w, cuda)).cpu().numpy()).all()
    assert ops.split_overflow_count() == 0
    q = torch.randn(1, 64, 64, device=cuda) * 1.0e5                # the attention's Q / K / V staging sees GEMM outputs: counted too
    ops.set_precision("f16x3")                                     # (the attention kernel follows the library mode)
    try:
        ops.attention(q, q, q, 1)
        assert ops.split_overflow_count() > 0
    finally:
        ops.set_precision(ops.default_precision())


def test_models_release_their_split_weights_and_keep_their_precision(cuda):
    """Each model carries its precision (re-asserted on entry) and its registered fp16 splits die with it."""
    import gc
    from hybridgl_amd import weights
    from hybridgl_amd.backbone import CLIPViTFM
    from oracle.cases import views_for_case
    sd = weights.clip_state_dict("tiny", 0)
    n0 = len(ops._split_cache)
    m16 = CLIPViTFM("tiny", state_dict=sd, device=cuda, precision="f16x3")
    n1 = len(ops._split_cache)
    assert n1 > n0
    m32 = CLIPViTFM("tiny", state_dict=sd, device=cuda, precision="f32")     # registers nothing, flips the library mode
    assert len(ops._split_cache) == n1
    loc, glo, masks = views_for_case(5, 64, 97, 130)
    args = (T(loc, cuda), T(glo, cuda), T(masks, cuda))
    y32 = m32(*args, masking_block=9, fusion_mode="G2L").cpu().numpy()
    y16 = m16(*args, masking_block=9, fusion_mode="G2L").cpu().numpy()           # must run in ITS mode again
    assert ops._precision_now == "f16x3"
    y32b = m32(*args, masking_block=9, fusion_mode="G2L").cpu().numpy()
    assert np.array_equal(y32, y32b) and np.abs(y16 - y32).max() < 1e-4
    solo = CLIPViTFM("tiny", state_dict=sd, device=cuda, precision="f16x3")
    assert np.array_equal(solo(*args, masking_block=9, fusion_mode="G2L").cpu().numpy(), y16)
    del m16, solo
    gc.collect()
    assert len(ops._split_cache) == n0


def test_gemm_f16x3_fp16_valued_weights_drop_the_zero_products(cuda):
    """a weight whose values are fp16 numbers (OpenAI's CLIP archives) has an all-zero lo plane: the registry says so and the
    ping-pong GEMM issues two products per step; the result is that of the three-product kernel (HGL_X3_TERMS=3) and of the
    register-staged tiling, bit for bit; a genuine fp32 weight keeps three"""
    import ctypes as C
    from hybridgl_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device=cuda).manual_seed(3)
    M, N, K = 4096, 768, 512
    a = torch.randn((M, K), device=cuda, generator=g)
    w32 = torch.randn((N, K), device=cuda, generator=g) / 22.0
    w16 = w32.half().float()
    b = torch.randn((N,), device=cuda, generator=g)
    r = torch.randn((M, N), device=cuda, generator=g)
    try:
        outs = {}
        for name, w in (("fp32", w32), ("fp16", w16)):
            for terms in ("auto", "3"):
                for kind in ("P", "v1"):
                    ops.select_x3_kernel(kind)
                    if terms == "3":
                        os.environ["HGL_X3_TERMS"] = "3"
                    else:
                        os.environ.pop("HGL_X3_TERMS", None)
                    outs[(name, terms, kind)] = ops.gemm_f16x3(a, w, b, r, "quickgelu")
        assert lib.hgl_split_weight_is_fp16_valued(C.c_void_p(w16.data_ptr())) == 1
        assert lib.hgl_split_weight_is_fp16_valued(C.c_void_p(w32.data_ptr())) == 0
        assert lib.hgl_split_weight_is_fp16_valued(C.c_void_p(a.data_ptr())) == -1
        for name in ("fp32", "fp16"):
            ref = outs[(name, "3", "v1")]
            for key, o in outs.items():
                if key[0] == name:
                    assert torch.equal(o, ref), key
        z = torch.nn.functional.linear(a.double(), w16.double(), b.double())
        z = z * torch.sigmoid(1.702 * z) + r.double()
        assert float((outs[("fp16", "auto", "P")].double() - z).abs().max()) < 3e-5
    finally:
        os.environ.pop("HGL_X3_TERMS", None)
        ops.select_x3_kernel("auto")
        ops.release_split_weights([w32.data_ptr(), w16.data_ptr()])


@pytest.mark.parametrize("B,H,S,hd,rel", [(2, 4, 2304, 80, True), (1, 8, 2048, 64, False), (1, 2, 2100, 80, False)])
def test_ping_pong_attention_is_bit_identical_to_the_tile_kernel(cuda, B, H, S, hd, rel):
    """attn_x3pp_kernel (one 8-wave workgroup per 256 queries, two wave groups one barrier interval apart, K / V chunks
    double-buffered) against attn_x3_kernel (HGL_ATTN_PP=0, a child process): same arithmetic per (query tile, key tile)
    pair -> identical bits; with the rel-pos tensors of SAM's global blocks, a 64-wide head, a ragged last key tile."""
    import subprocess
    import sys
    code = f"""
import sys, numpy as np, torch
sys.path.insert(0, {ROOT!r})
from hybridgl_amd import ops
ops.set_precision('f16x3')
g = torch.Generator().manual_seed({S + hd})
q, k, v = (torch.randn({B}, {S}, {H * hd}, generator=g).cuda() for _ in range(3))
kw = {{}}
if {rel}:
    side = int(round({S} ** 0.5))
    kw = dict(rel_h=torch.randn({B * H}, {S}, side, generator=g).cuda(), rel_w=torch.randn({B * H}, {S}, side, generator=g).cuda())
np.save(sys.argv[1], ops.attention(q, k, v, {H}, **kw).cpu().numpy())
"""
    outs = []
    for pp in ("1", "0"):
        path = f"/tmp/hgl_pp_{pp}_{S}_{hd}.npy"
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, HGL_ATTN_PP=pp), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(np.load(path))
    assert np.isfinite(outs[0]).all() and np.array_equal(outs[0], outs[1])
