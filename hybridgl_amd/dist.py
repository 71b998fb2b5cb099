"""Image-parallel evaluation over the GPUs of one node.

The per-ref loop of Hybridgl_main.py:79-230 carries no state from one dataset item to the next except the four IoU
accumulators and the per-sentence IoU list (Hybridgl_main.py:52-55) -- and the k1 / k2 clamp quirk (:178-181, see
HybridGLPipeline.k_clamp).  So: one process per GPU, rank r owns the items i = r (mod R) of the loader's order, weights
are replicated, there is NO data-path collective, and the only exchange is one all-gather of the per-sentence
(ref, sentence, I, U, I_final, U_final) rows at the end (RCCL over xGMI on a GPU node; a few kB).  Every rank then
holds all rows and evaluates the reference's report on them in the reference's order, so oIoU and mIoU are identical
to the single-process run whatever the number of ranks.

This module is the ONE implementation of that rule: bench.py, hybridgl_amd.main and the tests import it.
Nothing here touches the GPU at import time.
"""
import os
import socket
import subprocess
import sys

import numpy as np

ROW_FIELDS = ("ref_index", "sentence", "I", "U", "I_final", "U_final")


def env_rank():
    """(rank, local_rank, world) as torch.distributed.run exports them; (0, 0, 1) for a plain process."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def shard_indices(n_items, rank, world):
    """dataset positions owned by `rank`: i = rank (mod world) of the shuffle=False order (Hybridgl_main.py:45)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, n_items, world))


def shard_by_groups(keys, rank, world):
    """Like shard_indices, but consecutive items with the same key (the COCO image id: one image backs several
    consecutive refs, Hybridgl_main.py:79) stay on one rank, so the per-image cache of proposals and hybrid features
    (HybridGLPipeline.step) keeps working under sharding: run g of equal keys goes to rank g (mod world)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    out, g, prev = [], -1, object()
    for i, k in enumerate(keys):
        if k != prev:
            g += 1
            prev = k
        if g % world == rank:
            out.append(i)
    return out


def owned_index(j, rank, world):
    """the j-th item of `rank` (an endless stream: the synthetic benchmark)."""
    return rank + world * j


def init_process_group(backend, device=None):
    """torch.distributed from the launcher's environment (MASTER_ADDR defaults to 127.0.0.1: one node)."""
    import torch.distributed as dist
    rank, _, world = env_rank()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":   # = RCCL on ROCm
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def _comm_device(dist, device):
    import torch
    if dist.get_backend() == "nccl":
        return device if device is not None else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_rows(rows, dist=None, device=None, force=False):
    """All ranks' per-sentence rows [n, 6] int64 (ROW_FIELDS) -> the rows of the whole job on every rank.
    Two all-gathers: the row counts, then the rows padded to the largest count.  A world of one returns its rows as they
    are unless force=True, which sends them through the collectives anyway (the RCCL self-check of a 1-GPU box)."""
    import torch
    rows = np.ascontiguousarray(np.asarray(rows, dtype=np.int64).reshape(-1, len(ROW_FIELDS)))
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return rows
    world = dist.get_world_size()
    cdev = _comm_device(dist, device)
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=cdev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    width = max(max(counts), 1)
    pad = torch.zeros((width, len(ROW_FIELDS)), dtype=torch.int64, device=cdev)
    if rows.shape[0]:
        pad[:rows.shape[0]] = torch.from_numpy(rows).to(cdev)
    parts = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return np.concatenate([p[:c].cpu().numpy() for p, c in zip(parts, counts)], axis=0)


def metrics_from_rows(rows):
    """The report of Hybridgl_main.py:240-247 from per-sentence rows, evaluated in the reference's order (dataset
    position, then sentence): oIoU = sum(I) * 100 / sum(U); mIoU = torch.mean over the float32 per-sentence
    I * 1.0 / U (0 where U == 0, utils.py:373-376) * 100."""
    import torch
    rows = np.asarray(rows, dtype=np.int64).reshape(-1, len(ROW_FIELDS))
    order = np.lexsort((rows[:, 1], rows[:, 0]))
    rows = rows[order]
    cum = rows[:, 2:6].sum(axis=0) if len(rows) else np.zeros(4, dtype=np.int64)

    def mean_iou(i_col, u_col):
        if not len(rows):
            return 0.0
        i = torch.from_numpy(rows[:, i_col].copy()).to(torch.float32)
        u = torch.from_numpy(rows[:, u_col].copy()).to(torch.float32)
        iou = torch.where(u == 0, torch.zeros_like(i), i / torch.where(u == 0, torch.ones_like(u), u))
        return float(torch.mean(iou) * 100.0)

    return {
        "cum": [int(v) for v in cum],
        "oIoU": float(cum[0]) * 100.0 / float(cum[1]) if cum[1] else 0.0,
        "mIoU": mean_iou(2, 3),
        "oIoU_final": float(cum[2]) * 100.0 / float(cum[3]) if cum[3] else 0.0,
        "mIoU_final": mean_iou(4, 5),
        "n_sentences": int(len(rows)),
    }


def gather_metrics(rows, dist=None, device=None):
    """metrics of the whole job from this rank's rows (one exchange; identical on every rank)."""
    return metrics_from_rows(gather_rows(rows, dist, device))


def max_over_ranks(value, dist=None, device=None, force=False):
    """the job's time = the slowest rank's"""
    import torch
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_comm_device(dist, device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def rccl_selfcheck(device, rows=None):
    """Initialise the `nccl` (= RCCL) backend with a world of ONE on `device` and push metric rows through the same
    collectives a multi-GPU job uses (gather_rows / max_over_ranks with force=True): librccl load, communicator creation and
    the device-tensor branch run on a 1-GPU box.  Returns {"backend", "world_size_seen", "rows_roundtrip_ok", "max_ok",
    "seconds"}; the process group is destroyed again.  Must not be called while another process group is alive."""
    import time
    import torch.distributed as dist
    t0 = time.perf_counter()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        if rows is None:
            rows = np.arange(42, dtype=np.int64).reshape(7, 6) * 1000003
        got = gather_rows(rows, dist, device, force=True)
        mx = max_over_ranks(3.25, dist, device, force=True)
        dist.barrier()
        return {"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(),
                "rows_roundtrip_ok": bool(np.array_equal(got, np.asarray(rows, dtype=np.int64).reshape(-1, len(ROW_FIELDS)))),
                "max_ok": mx == 3.25, "seconds": time.perf_counter() - t0}
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_cpu_affinity(local_rank, local_world, cpus=None):
    """The host cores of one rank: an even contiguous share of the cores this process may run on (each rank keeps a
    launch thread + loader threads busy; eight ranks that roam over all cores evict each other's caches).  Returns the
    sorted core list; [] when there are fewer cores than ranks (no pinning then)."""
    cpus = sorted(cpus if cpus is not None else os.sched_getaffinity(0))
    per = len(cpus) // max(local_world, 1)
    if per < 1:
        return []
    return cpus[local_rank * per:(local_rank + 1) * per]


def pin_rank_to_cores(local_rank, local_world, max_cores=32):
    """os.sched_setaffinity of this process to rank_cpu_affinity (HYBRIDGL_PIN_CORES=0 disables); returns the cores.
    A rank never takes more than `max_cores` of its share, and a single rank on a big host is confined as well: the launch
    thread and the four loader threads share the interpreter lock, and on 256 cores they migrate across sockets (measured
    on the evaluator fed from disk: 71 refs/s roaming over 256 cores, 77 on 32; loader CPU time per item 6.4 -> 3.0 ms)."""
    if os.environ.get("HYBRIDGL_PIN_CORES", "1") == "0":
        return []
    cores = rank_cpu_affinity(local_rank, max(local_world, 1))
    if local_world <= 1 and len(cores) <= max_cores:
        return []
    cores = cores[:max_cores]
    if cores:
        try:
            os.sched_setaffinity(0, cores)
        except OSError:
            return []
    return cores


def size_host_threads(cores, workers=4):
    """After pin_rank_to_cores: size torch's intra-op (OpenMP) pool for the share of cores this rank owns.  `import torch`
    sized it for ALL host cores; the loader threads' tensor ops would each fan out to that pool on the rank's small share
    and oversubscribe exactly the cores the launch thread needs.  Returns the thread count set (None: not pinned)."""
    if not cores:
        return None
    import torch
    n = max(1, len(cores) // (int(workers) + 1))
    torch.set_num_threads(n)
    return n


def spawn_local_ranks(n, argv, extra_env=None, timeout=None, poll_s=0.05):
    """Start `n` fresh processes of `argv` (one rank each) with the launcher's environment (RANK, LOCAL_RANK, WORLD_SIZE,
    MASTER_ADDR = 127.0.0.1, MASTER_PORT = a free port) and wait for them.  Must be called BEFORE the calling process
    touches the GPU (it only forks interpreters; nothing is exec'ed over an initialised device).  Rank 0 inherits
    stdout.  ALL ranks are polled: the first rank that exits non-zero -- whichever it is -- ends the job at once (the
    others may be parked in a collective waiting for it: they are terminated, then killed) and ITS exit code is returned;
    `timeout` seconds without completion kill every rank and return 124.  0 when every rank exits 0."""
    import time
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(argv, env=env, stdout=None if r == 0 else subprocess.DEVNULL))

    def stop_all():
        for q in procs:
            if q.poll() is None:
                q.terminate()
        t_end = time.monotonic() + 5.0
        for q in procs:
            try:
                q.wait(timeout=max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                q.kill()
                q.wait()

    deadline = None if timeout is None else time.monotonic() + timeout
    try:
        while True:
            running = 0
            for p in procs:
                rc = p.poll()
                if rc is None:
                    running += 1
                elif rc != 0:
                    stop_all()
                    return rc if rc > 0 else 128 - rc     # killed by signal s -> 128 + s, as a shell reports it
            if running == 0:
                return 0
            if deadline is not None and time.monotonic() > deadline:
                stop_all()
                return 124
            time.sleep(poll_s)
    except BaseException:
        stop_all()
        raise


def visible_gpu_count():
    """number of HIP devices without initialising one (torch.cuda.device_count() does not create a context)"""
    import torch
    return torch.cuda.device_count()
