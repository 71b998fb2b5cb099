"""The host-side C++ of the library (csrc/gtmask.cpp parses annotation strings, csrc/cvkernel.cpp) under
AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers on the host build only -- GPU ASan is not
available on this pool).  A harness executable (tests/native/host_sanitize.cpp) is built with
g++ -fsanitize=address,undefined -fno-sanitize-recover and fed 450 seeded polygon cases, well-formed and FUZZED RLE strings
and kernel sizes; it must finish without a sanitizer report and print the same results as the regular library."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hybridgl_amd", "csrc")


def _fnv(b):
    h = 1469598103934665603
    for v in bytes(b):
        h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    out = tmp_path_factory.mktemp("asan") / "host_sanitize"
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off",
           "-Wno-attributes", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
           os.path.join(ROOT, "tests", "native", "host_sanitize.cpp"), os.path.join(CSRC, "gtmask.cpp"),
           os.path.join(CSRC, "cvkernel.cpp"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(out)


def _rle_string(counts):
    """maskApi.c rleToString (:203-215): the encoder, used here to make well-formed inputs"""
    s = []
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1F
            x >>= 5
            more = not ((x == -1 and (ch & 0x10)) or (x == 0 and not (ch & 0x10)))
            if more:
                ch |= 0x20
            s.append(chr(ch + 48))
    return "".join(s)


def test_host_code_is_clean_under_asan_ubsan(harness, tmp_path):
    from hybridgl_amd import _lib, refer_io
    rng = np.random.default_rng(2024)
    lines, expect = [], []
    for _ in range(450):                                     # the fuzz of tests/test_gtmask.py, under the sanitizers
        H, W = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        polys = []
        for _p in range(int(rng.integers(1, 4))):
            k = int(rng.integers(1, 9))
            polys.append((rng.random(2 * k) * np.array([W, H] * k) * 1.4 - 0.2 * max(H, W)).round(int(rng.integers(0, 3))).tolist())
        lines.append(f"P {H} {W} {len(polys)} " + " ".join(str(len(p) // 2) for p in polys) + " " +
                     " ".join(repr(float(v)) for p in polys for v in p))
        m, area = refer_io.gt_mask_from_polygons(polys, H, W)
        expect.append(("P", 0, area, _fnv(np.ascontiguousarray(m))))
    lines.append("P 8 8 1 2 nan 1.0 2.0 3.0")                # rejected, not cast
    expect.append(("P", None, -1, 0))
    lines.append("P 8 8 1 2 1e30 1.0 2.0 3.0")
    expect.append(("P", None, -1, 0))
    n_ok = 0
    for i in range(300):
        H, W = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        if i % 3 == 0:                                       # well-formed
            cuts = np.sort(rng.integers(0, H * W + 1, size=int(rng.integers(0, 9))))
            counts = np.diff(np.concatenate([[0], cuts, [H * W]])).tolist()
            s = _rle_string(counts)
        else:                                                # fuzzed: random printable bytes, truncated groups, long chains
            s = "".join(chr(int(v)) for v in rng.integers(33, 127, size=int(rng.integers(0, 40))))
            if i % 3 == 2:
                s += "o" * int(rng.integers(0, 12))          # continuation bits with no end
        tok = s if s else "<empty>"
        lines.append(f"S {H} {W} {tok}")
        try:
            m, area = refer_io.gt_mask_from_rle({"size": [H, W], "counts": s})
            expect.append(("S", 0, area, _fnv(np.ascontiguousarray(m))))
            n_ok += 1
        except Exception:
            expect.append(("S", None, -1, 0))
    lib = _lib.load()
    for n, sigma in [(1, 0.0), (3, 0.0), (7, 0.0), (15, 0.0), (15, 2.6), (31, 5.0), (9, -1.0), (4, 0.0), (33, 0.0), (0, 0.0)]:
        lines.append(f"K {n} {sigma}")
        taps = (C.c_uint16 * max(n, 1))()
        rc = lib.hgl_cv_gaussian_kernel_q8(n, C.c_double(sigma), taps)
        expect.append(("K", 0 if rc == 0 else None, 0, _fnv(bytes(taps)[:2 * n]) if rc == 0 else 0))
    assert n_ok >= 100
    path = tmp_path / "cases.txt"
    path.write_text("\n".join(lines) + "\n")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([harness, str(path)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stderr[-4000:], r.stdout[-500:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
    got = [l.split() for l in r.stdout.strip().splitlines()]
    assert len(got) == len(expect)
    for g, (kind, rc, area, h) in zip(got, expect):
        assert g[0] == kind
        if rc is None:
            assert int(g[1]) != 0, g
        else:
            assert int(g[1]) == 0 and int(g[3]) == h, (g, kind, area, h)
            if kind != "K":
                assert int(g[2]) == area
