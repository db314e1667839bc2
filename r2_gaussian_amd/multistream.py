"""Several independent views in flight on one GPU (new functionality; the reference renders one view at a time).

One training view is half latency-bound bookkeeping (the binning chain: thirteen thin kernels with dependent memory round
trips between them) and half VALU-bound rendering (profiles/r02e_step_timeline.md).  Two independent views
overlap the one with the other when each has its own HIP stream AND its own host thread -- the library keeps its mutable state
per host thread, and the compiled torch boundary releases the GIL inside the calls: 4500 -> 6000 views/s at 300k Gaussians /
512^2 (``bench.py``: ``concurrent_streams``).

``StreamPool`` keeps ``n`` worker threads, each bound to its own ``torch.cuda.Stream``::

    pool = StreamPool(2, device)
    def one_view(cam, xyz, density, scaling, rotation):          # runs on a worker: its stream is the current stream
        img, radii = render(cam, xyz, density, scaling, rotation)
        loss_of(img, cam).backward()
        return radii
    # per worker: separate autograd leaves over the SAME storage, so that every worker accumulates into its own .grad
    leaves = [pool.leaves_like(params) for _ in range(pool.n)]
    out = pool.map(lambda i, cam: one_view(cam, *leaves[i % pool.n]), cameras[:2])
    grads = pool.sum_grads(leaves)                                  # what the optimiser step consumes

``map`` orders the workers' streams after the caller's current stream (inputs produced there are visible) and the caller's
stream after the workers' (results are ready for whatever the caller enqueues next) with events: no host synchronisation.
"""
import queue
import threading

import torch


class StreamPool:
    def __init__(self, n=2, device="cuda"):
        self.device = torch.device(device)
        self.n = int(n)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.n)]
        self._q = [queue.Queue() for _ in range(self.n)]
        self._threads = [threading.Thread(target=self._run, args=(i,), daemon=True) for i in range(self.n)]
        for t in self._threads:
            t.start()

    def _run(self, i):
        torch.cuda.set_device(self.device)
        with torch.cuda.stream(self.streams[i]):
            while True:
                job = self._q[i].get()
                if job is None:
                    return
                fn, args, slot, done = job
                try:
                    slot[0] = fn(*args)
                except BaseException as ex:   # noqa: BLE001  (handed to the caller)
                    slot[1] = ex
                done.set()

    def map(self, fn, items):
        """``fn(k, item)`` for every item, item k on worker k % n (in order per worker), concurrently -> list of results."""
        items = list(items)
        cur = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        for s in self.streams:
            s.wait_event(ready)
        jobs = []
        for k, it in enumerate(items):
            slot, done = [None, None], threading.Event()
            self._q[k % self.n].put((fn, (k, it), slot, done))
            jobs.append((slot, done))
        # every job is waited for and the caller's stream is ordered after the workers' BEFORE a failure is re-raised: no
        # worker may still be enqueueing or running on shared parameter storage when the caller carries on
        for _slot, done in jobs:
            done.wait()
        for s in self.streams:
            ev = torch.cuda.Event()
            ev.record(s)
            cur.wait_event(ev)
        for slot, _done in jobs:
            if slot[1] is not None:
                raise slot[1]
        return [slot[0] for slot, _done in jobs]

    @staticmethod
    def leaves_like(params):
        """Fresh autograd leaves over the storage of ``params`` (no copy): one set per worker keeps the .grad tensors apart."""
        return [p.detach().requires_grad_(True) for p in params]

    @staticmethod
    def sum_grads(leaf_sets):
        """Sum of the workers' gradients, parameter by parameter (None where no worker produced one); clears them."""
        out = []
        for ps in zip(*leaf_sets):
            gs = [p.grad for p in ps if p.grad is not None]
            out.append(None if not gs else (gs[0] if len(gs) == 1 else torch.stack(gs).sum(0)))
            for p in ps:
                p.grad = None
        return out

    def close(self):
        for q in self._q:
            q.put(None)
        for t in self._threads:
            t.join(timeout=5)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
