"""Background preparation of dataset items: the counterpart of the reference's
`DataLoader(dataset, batch_size=1, num_workers=4, shuffle=False)` (Hybridgl_main.py:45).

The reference decodes, transforms and tokenises on four worker processes while the GPU works on the previous item.
Here the host work of an item (image decode, GEM transform, tokenisation, ground-truth rasterisation -- PIL, numpy and
the native codec all release the GIL) runs on `workers` threads, `depth` items ahead of the consumer, and the
host->device copies of an item are issued by the same thread on its own stream from pinned memory; the item carries
an event (`RefBatch.ready`) that the consuming streams wait for.  Items come out in the order of `jobs`
(shuffle=False); an exception raised while preparing an item is re-raised at that item's position.
"""
import collections
import threading
from concurrent.futures import ThreadPoolExecutor


class Prefetcher:
    def __init__(self, jobs, make, workers=4, depth=16, device=None):
        """jobs: iterable of job descriptions (dataset positions); make(job) -> item, called on a worker thread.
        device: a torch cuda device -- make() then runs with a per-thread side stream current (every `.to(device,
        non_blocking=True)` inside it lands there) and the item's `ready` attribute is set to an event recorded behind
        those copies.  None: plain host prefetching (tests)."""
        if workers < 1 or depth < 1:
            raise ValueError("Prefetcher: workers and depth must be >= 1")
        self.jobs = jobs
        self.make = make
        self.workers = workers
        self.depth = depth
        self.device = device
        self._tls = threading.local()
        self.max_in_flight = 0     # high-water mark of prepared-but-unconsumed items (tests: the look-ahead is bounded)
        # where the host time goes (read by hybridgl_amd.main / bench.py): seconds the workers spent preparing items, seconds
        # the consumer sat waiting for the next item, items delivered
        self.make_s = 0.0
        self.wait_s = 0.0
        self.items = 0
        self._stat_lock = threading.Lock()

    def _run(self, job):
        import time
        t0 = time.perf_counter()
        try:
            return self._run_inner(job)
        finally:
            with self._stat_lock:
                self.make_s += time.perf_counter() - t0

    def _run_inner(self, job):
        if self.device is None:
            return self.make(job)
        import torch
        st = getattr(self._tls, "stream", None)
        if st is None:
            st = self._tls.stream = torch.cuda.Stream(self.device)
        with torch.cuda.device(self.device), torch.cuda.stream(st):
            item = self.make(job)
            ev = torch.cuda.Event()
            ev.record(st)
        try:
            item.ready = ev
        except AttributeError:
            ev.synchronize()     # an item that cannot carry the event: finish its copies here
        return item

    def __iter__(self):
        import time
        it = iter(self.jobs)
        q = collections.deque()
        with ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="hgl-loader") as pool:
            try:
                for job in it:
                    q.append(pool.submit(self._run, job))
                    if len(q) >= self.depth:
                        break
                while q:
                    self.max_in_flight = max(self.max_in_flight, len(q))
                    t0 = time.perf_counter()
                    item = q.popleft().result()      # re-raises the worker's exception at this position
                    self.wait_s += time.perf_counter() - t0
                    self.items += 1
                    for job in it:                   # keep the window full: one new job per consumed item
                        q.append(pool.submit(self._run, job))
                        break
                    yield item
            finally:
                for f in q:
                    f.cancel()


def pin_upload(a, device):
    """numpy array / CPU tensor -> device tensor through pinned memory on the CURRENT stream (asynchronous)."""
    import numpy as np
    import torch
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    return t.contiguous().pin_memory().to(device, non_blocking=True)
