"""The N > 1 path of the product (hybridgl_amd/dist.py: sharding rule, metric-row exchange, report) on CPU over gloo,
world_size 2 -- the code bench.py and hybridgl_amd.main run, not a copy of it.  The oracle only plays the checker:
it produces per-sentence (I, U) rows for seeded tail cases; the product code shards, gathers and reports them."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hybridgl_amd import dist as D   # noqa: E402


def _rows_for(indices, k_of=None):
    """oracle-made metric rows of the given dataset positions (two sentences per ref)"""
    from oracle import clip_oracle as O
    from oracle.cases import tail_case
    rows = []
    for i in indices:
        hybrid, t_pos, t_neg, masks, boxes, attn, gt = tail_case(i % 8)
        for s, (d, rel) in enumerate((("none", "none"), ("left", "big"))):
            gem = O.coherence_scores(attn, masks, d, 1.8)
            ip, ifin, _, _ = O.score_sentence(hybrid, t_pos, t_neg, boxes, gem, 100.0, 3, 6, 0.6, rel, s == 1)
            rows.append([i, s, *O.compute_iou(masks[ip], gt), *O.compute_iou(masks[ifin], gt)])
    return np.asarray(rows, dtype=np.int64).reshape(-1, 6)


def _worker(rank, world, port, n_items, q):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    sys.path.insert(0, ROOT)
    from hybridgl_amd import dist as DD
    dist = DD.init_process_group("gloo")
    r, _, w = DD.env_rank()
    mine = DD.shard_indices(n_items, r, w)
    rows = _rows_for(mine)
    # rank 1 holds one row fewer than rank 0 when n_items is odd: the padded exchange must cope
    m = DD.gather_metrics(rows, dist)
    t = DD.max_over_ranks(float(rank + 1), dist)
    dist.barrier()
    q.put((rank, m, t, len(rows)))
    dist.destroy_process_group()


def test_sharding_is_a_partition():
    for n, w in [(10, 2), (7, 4), (64, 8), (3, 8), (0, 2)]:
        parts = [D.shard_indices(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert all(p == sorted(p) for p in parts)
    with pytest.raises(ValueError):
        D.shard_indices(4, 2, 2)
    assert [D.owned_index(j, 1, 4) for j in range(3)] == [1, 5, 9]


def test_group_sharding_keeps_an_image_on_one_rank():
    keys = [7, 7, 7, 3, 3, 9, 7, 7, 1]          # image ids in loader order (the second run of 7 is a new group)
    parts = [D.shard_by_groups(keys, r, 2) for r in range(2)]
    assert sorted(parts[0] + parts[1]) == list(range(len(keys)))
    assert parts[0] == [0, 1, 2, 5, 8] and parts[1] == [3, 4, 6, 7]
    assert D.shard_by_groups(keys, 0, 1) == list(range(len(keys)))


def test_report_matches_the_reference_formulas():
    """metrics_from_rows restates Hybridgl_main.py:240-247 / utils.py:365-384: float32 per-sentence IoU (0 where U == 0),
    torch.mean in the loader's order, cumulative I * 100 / U."""
    import torch
    rows = np.array([[1, 0, 10, 40, 0, 0], [0, 1, 3, 9, 9, 9], [0, 0, 5, 7, 1, 3]], dtype=np.int64)
    m = D.metrics_from_rows(rows)
    assert m["cum"] == [18, 56, 10, 12] and m["n_sentences"] == 3
    assert m["oIoU"] == 18 * 100.0 / 56 and m["oIoU_final"] == 10 * 100.0 / 12
    ordered = [(5, 7), (3, 9), (10, 40)]            # (ref 0, s0), (ref 0, s1), (ref 1, s0)
    ref = torch.mean(torch.tensor([torch.tensor(i) * 1.0 / torch.tensor(u) for i, u in ordered])) * 100.0
    assert m["mIoU"] == float(ref)
    ref_f = torch.mean(torch.tensor([torch.tensor(1) * 1.0 / torch.tensor(3), torch.tensor(1.0), torch.tensor(0.0)])) * 100.0
    assert m["mIoU_final"] == float(ref_f)
    assert D.metrics_from_rows(np.zeros((0, 6), np.int64))["oIoU"] == 0.0


def test_two_ranks_report_equals_single_process():
    n_items, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = D.free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = D.metrics_from_rows(_rows_for(range(n_items)))
    for rank, m, tmax, n_rows in got:
        assert m == single, (rank, m, single)       # every rank holds the job's report, bit for bit
        assert tmax == 2.0                           # MAX over ranks, as bench.py times the job
    assert [g[3] for g in got] == [6, 4]             # ragged row counts went through the padded all-gather


def test_spawn_local_ranks_sets_the_launcher_environment(tmp_path):
    """bench.py --gpus N without a launcher: N fresh interpreters with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*,
    a failing rank makes the job fail."""
    out = tmp_path / "r"
    code = ("import os, sys; open(sys.argv[1] + os.environ['RANK'], 'w').write(' '.join(os.environ[k] for k in "
            "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR'))); sys.exit(3 if os.environ['RANK'] == '1' and len(sys.argv) > 2 else 0)")
    assert D.spawn_local_ranks(3, [sys.executable, "-c", code, str(out)]) == 0
    assert [open(f"{out}{r}").read() for r in range(3)] == [f"{r} {r} 3 127.0.0.1" for r in range(3)]
    assert D.spawn_local_ranks(2, [sys.executable, "-c", code, str(out), "fail"]) == 3


def test_a_late_rank_that_dies_ends_the_job_at_once(tmp_path):
    """Rank 2 of 3 fails while ranks 0 and 1 sit in a (simulated) collective: the spawner polls ALL ranks, stops the
    others and returns rank 2's code -- it does not block on rank 0 first."""
    import time
    code = ("import os, sys, time\n"
            "if os.environ['RANK'] == '2':\n    time.sleep(0.3); sys.exit(7)\n"
            "time.sleep(600)\n")
    t0 = time.monotonic()
    assert D.spawn_local_ranks(3, [sys.executable, "-c", code]) == 7
    assert time.monotonic() - t0 < 30
    # a rank killed by a signal reports 128 + signal; a job that never finishes hits the timeout (124)
    code = "import os, signal, time\nif os.environ['RANK'] == '1':\n    os.kill(os.getpid(), signal.SIGKILL)\ntime.sleep(600)\n"
    assert D.spawn_local_ranks(2, [sys.executable, "-c", code]) == 128 + 9
    assert D.spawn_local_ranks(2, [sys.executable, "-c", "import time; time.sleep(600)"], timeout=1.0) == 124


def _worker8(rank, world, port, counts, q):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    sys.path.insert(0, ROOT)
    from hybridgl_amd import dist as DD
    dist = DD.init_process_group("gloo")
    # rank r holds counts[r] seeded rows (some ranks none at all); ref_index encodes the owner so the report order is checkable
    rng = np.random.default_rng(100 + rank)
    rows = np.stack([np.array([rank + world * j, 0, *sorted(rng.integers(1, 1000, 2)), *sorted(rng.integers(1, 1000, 2))])
                     for j in range(counts[rank])]).astype(np.int64) if counts[rank] else np.zeros((0, 6), np.int64)
    all_rows = DD.gather_rows(rows, dist)
    m = DD.metrics_from_rows(all_rows)
    q.put((rank, m, all_rows.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_with_ragged_and_empty_shares():
    """world 8 over gloo: row counts 3, 0, 1, 0, 0, 5, 2, 0 -- ranks WITHOUT a single row (a dataset tail shorter than the
    node, images without proposals) go through the padded all-gather; every rank ends with the same rows and report."""
    world, counts = 8, [3, 0, 1, 0, 0, 5, 2, 0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = D.free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, counts, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rows0 = got[0][2]
    assert len(rows0) == sum(counts)
    assert [r[0] % world for r in rows0] == [r for r, c in enumerate(counts) for _ in range(c)]   # rank-major, each rank in order
    for rank, m, rows in got:
        assert rows == rows0 and m == got[0][1]
    assert got[0][1] == D.metrics_from_rows(np.asarray(rows0))


def test_rank_core_shares_are_disjoint():
    cpus = list(range(3, 67))                      # 64 cores the process may use
    parts = [D.rank_cpu_affinity(r, 8, cpus) for r in range(8)]
    assert all(len(p) == 8 for p in parts) and sorted(sum(parts, [])) == cpus
    assert D.rank_cpu_affinity(0, 8, list(range(4))) == []        # fewer cores than ranks: no pinning
    assert D.rank_cpu_affinity(2, 3, list(range(8))) == [4, 5]


def test_bench_spawns_before_touching_the_gpu():
    """`python bench.py --gpus 2` with no launcher must reach spawn_local_ranks without a GPU call: in this GPU-less
    container the children then stop at bench.py's own 'needs a GPU' assertion -- not the parent."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_driver.py::test_bench_two_ranks")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--backend", "gloo"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "bench.py needs a GPU" in r.stderr and "spawn_local_ranks" not in r.stderr


def test_other_nouns_carry_the_reference_prefix():
    """Hybridgl_main.py:160 / Hybridgl_main_PhraseCut.py:147: clip.tokenize('a photo of ' + other_noun)"""
    from hybridgl_amd.main import sentence_strings
    rec = {"noun_phrase": "the cat", "other_nouns": ["a dog", "sofa"]}
    assert sentence_strings("the cat left of a dog on the sofa", rec) == [
        "the cat left of a dog on the sofa", "the cat", "a photo of a dog", "a photo of sofa"]
    assert sentence_strings("cat", {}) == ["cat", "cat"]


def _force_worker(port, q):
    os.environ.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    sys.path.insert(0, ROOT)
    from hybridgl_amd import dist as DD
    dist = DD.init_process_group("gloo")
    rows = np.arange(30, dtype=np.int64).reshape(5, 6)
    q.put((DD.gather_rows(rows, dist, force=True).tolist(), DD.gather_rows(rows, dist).tolist(),
           DD.max_over_ranks(2.5, dist, force=True)))
    dist.destroy_process_group()


def test_a_world_of_one_goes_through_the_collectives_when_forced():
    """gather_rows / max_over_ranks return early in a world of one; force=True sends the rows through the two all-gathers
    and the all-reduce anyway (what the RCCL self-check of a 1-GPU box does, dist.rccl_selfcheck; here over gloo)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_force_worker, args=(D.free_port(), q))
    p.start()
    forced, plain, mx = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert forced == plain == np.arange(30).reshape(5, 6).tolist() and mx == 2.5


def test_host_thread_pool_is_sized_for_the_rank_s_share():
    import torch
    old = torch.get_num_threads()
    try:
        assert D.size_host_threads([], 4) is None and torch.get_num_threads() == old
        assert D.size_host_threads(list(range(32)), 4) == 6 and torch.get_num_threads() == 6
        assert D.size_host_threads([0, 1], 4) == 1
    finally:
        torch.set_num_threads(old)
