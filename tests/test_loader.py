"""hybridgl_amd/loader.py: the background preparation of dataset items (Hybridgl_main.py:45 DataLoader(num_workers=4,
shuffle=False)) -- ordering, bounded look-ahead, error propagation -- and the grouping rule of HybridGLPipeline.run.
Host logic only (no GPU)."""
import os
import sys
import threading
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hybridgl_amd.loader import Prefetcher   # noqa: E402


def test_items_come_out_in_job_order_whatever_the_worker_timing():
    def make(i):
        time.sleep(0.002 * ((i * 7) % 5))      # later jobs often finish first
        return i * i
    assert list(Prefetcher(range(40), make, workers=4, depth=8)) == [i * i for i in range(40)]
    assert list(Prefetcher([], make)) == []
    assert list(Prefetcher(iter(range(3)), make, workers=2, depth=100)) == [0, 1, 4]


def test_look_ahead_is_bounded_and_runs_beside_the_consumer():
    started, lock = [], threading.Lock()

    def make(i):
        with lock:
            started.append(i)
        return i
    pf = Prefetcher(range(100), make, workers=4, depth=6)
    it = iter(pf)
    assert next(it) == 0
    time.sleep(0.2)
    with lock:
        assert max(started) <= 6 and len(started) >= 6       # window of `depth` jobs (+ the refill for item 0), not the whole list
    rest = list(it)
    assert rest == list(range(1, 100)) and pf.max_in_flight <= 6


def test_a_failing_job_raises_at_its_position():
    def make(i):
        if i == 5:
            raise ValueError("bad item 5")
        return i
    got = []
    with pytest.raises(ValueError, match="bad item 5"):
        for v in Prefetcher(range(10), make, workers=3, depth=4):
            got.append(v)
    assert got == [0, 1, 2, 3, 4]
    with pytest.raises(ValueError):
        Prefetcher(range(3), make, workers=0)


def test_image_units_and_groups():
    """HybridGLPipeline._units: consecutive items of one image are ONE unit (proposals and hybrid features once per
    image); groups hold `group` units and are never split inside a unit; items without image_id stand alone."""
    from types import SimpleNamespace as NS
    from hybridgl_amd.pipeline import HybridGLPipeline
    ids = [7, 7, 7, 3, None, None, 9, 9, 7, 1, 1]
    refs = [NS(image_id=v, k=i) for i, v in enumerate(ids)]
    groups = list(HybridGLPipeline._units(iter(refs), 3))
    shape = [[[r.k for r in u] for u in g] for g in groups]
    assert shape == [[[0, 1, 2], [3], [4]], [[5], [6, 7], [8]], [[9, 10]]]
    assert list(HybridGLPipeline._units(iter([]), 4)) == []
    assert [[len(u) for u in g] for g in HybridGLPipeline._units(iter(refs), 100)] == [[3, 1, 1, 1, 2, 1, 2]]
