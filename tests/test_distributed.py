"""CPU, world_size 2 over gloo: the sharding rule and the metric all-gather of the N>1 path
(bench.py / SURVEY.md 8e) -- refs i = r (mod R), one all_gather of [cum_I, cum_U, cum_I_f, cum_U_f, n]."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shard(n_items, rank, world):
    return list(range(rank, n_items, world))


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import clip_oracle as O
    from oracle.cases import tail_case
    cum = np.zeros(5, dtype=np.int64)
    for i in shard(n_items, rank, world):
        hybrid, t_pos, t_neg, masks, boxes, attn, gt = tail_case(i % 8)
        gem = O.coherence_scores(attn, masks, "none", 1.8)
        ip, ifin, _, _ = O.score_sentence(hybrid, t_pos, t_neg, boxes, gem, 100.0, 3, 6, 0.6, "none", False)
        I0, U0 = O.compute_iou(masks[ip], gt)
        I1, U1 = O.compute_iou(masks[ifin], gt)
        cum += np.array([I0, U0, I1, U1, 1])
    vec = torch.from_numpy(cum)
    out = [torch.zeros_like(vec) for _ in range(world)]
    dist.all_gather(out, vec)
    tmax = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.barrier()
    if rank == 0:
        q.put((torch.stack(out).numpy(), float(tmax.item())))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharding_is_a_partition():
    for n, w in [(10, 2), (7, 4), (64, 8), (3, 8)]:
        parts = [shard(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))


def test_two_ranks_gather_equals_single_process():
    sys.path.insert(0, ROOT)
    from oracle import clip_oracle as O
    from oracle.cases import tail_case
    n_items, world = 6, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, tmax = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = np.zeros(5, dtype=np.int64)
    for i in range(n_items):
        hybrid, t_pos, t_neg, masks, boxes, attn, gt = tail_case(i % 8)
        gem = O.coherence_scores(attn, masks, "none", 1.8)
        ip, ifin, _, _ = O.score_sentence(hybrid, t_pos, t_neg, boxes, gem, 100.0, 3, 6, 0.6, "none", False)
        ref += np.array([*O.compute_iou(masks[ip], gt), *O.compute_iou(masks[ifin], gt), 1])
    assert gathered.shape == (2, 5)
    assert np.array_equal(gathered.sum(0), ref)          # oIoU numerators/denominators are additive
    assert tmax == 2.0                                    # MAX over ranks, as bench.py times the job
