"""Host-side pipeline for training loops: window batches are *prepared* (plan, row maps, union of the cached
snapshot views, uploads -- temp_amd.dynamic_rgcn.DynamicRGCN.prepare) by background threads a few steps ahead of
the GPU, so the host work per batch overlaps the previous steps instead of preceding each one.

    for wb in BatchPrefetcher(model, batches, seq_len=model.train_seq_len, workers=2):
        loss = model.run_loss(wb)
        ...

The reference builds its batched DGL graphs inline at the top of every forward (models/DynamicRGCN.py:76-94);
this replaces that with a bounded, ordered pipeline.  About half of `prepare` runs inside the C++ planner library with
the GIL released; the rest shares the GIL with the training loop's launch code, so ONE worker delivers a batch every
(its own time + the main thread's share of the GIL); a second worker takes that off the critical path.
"""
import collections
import threading

import numpy as np
import torch


class BatchPrefetcher:
    """Yields the prepared batches IN ORDER.

    On a GPU every worker prepares under its OWN HIP stream: the uploads and the small device kernels of `prepare` then
    queue behind each other only, not behind the training step's kernels on the main stream.  The consumer's stream waits
    for the batch's `ready` event; a consumed batch is kept alive until the main stream has passed it, so the caching
    allocator cannot hand its memory (allocated on a worker's stream) to a later batch while kernels still read it.  Shared
    resident objects created on first use (a snapshot's views, the true-set store) are published only after the creating
    stream has drained (temp_amd._lib.publish), so another worker's stream may read them without an event.

    workers > 1 (or batch_seeds=True): batch i is prepared with its own random generator, seeded from `model.sample_rng`
    in batch order when the batch is handed to a worker -- the subsets drawn for a batch, and hence the whole run, do not
    depend on which worker took it or when.  With one worker and batch_seeds=False the model's generator is used directly,
    exactly as by an inline `model.prepare`."""

    def __init__(self, model, batches, seq_len=None, train=True, depth=2, workers=1, batch_seeds=None, cooperative=True):
        self.model, self.batches, self.train = model, batches, train
        # cooperative: the workers' Python pauses (between the stages of `prepare`, _lib.pause_point) while the consumer issues a
        # step, and runs while the consumer waits for a batch -- the two halves share ONE interpreter lock, and taking it from each
        # other every switch interval doubles the time of both.  The workers' C++ planner calls run regardless.
        self.cooperative = bool(cooperative)
        self._gate = threading.Event()
        self._gate.set()
        self.seq_len = seq_len if seq_len is not None else model.train_seq_len
        self.workers = max(1, int(workers))
        self.batch_seeds = (self.workers > 1) if batch_seeds is None else bool(batch_seeds)
        if self.workers > 1 and not self.batch_seeds:
            raise ValueError("several workers need per-batch seeds (batch_seeds=True)")
        self.window = max(1, depth) + self.workers - 1          # batches handed out and not yet consumed
        self.threads = []
        dev = next(model.parameters()).device
        self.device = dev if dev.type == "cuda" else None
        self._inflight = collections.deque()

    # ---- workers ---------------------------------------------------------------------------------------------
    def _take(self):
        """Next (index, batch, seed) in batch order, or None when the source is exhausted / the run is stopping."""
        with self._src_lock:
            if self._stop:
                return None
            try:
                t_list = next(self._src)
            except StopIteration:
                return None
            except BaseException as e:                          # an error of the batch source surfaces at this position
                i = self._n_taken
                self._n_taken += 1
                self._stop = True
                return (i, e, None)
            i = self._n_taken
            self._n_taken += 1
            try:
                seed = int(self.model.sample_rng.integers(1 << 62)) if self.batch_seeds else None
            except BaseException as e:                          # index i is already consumed: the error is THIS batch's result
                self._stop = True
                return (i, e, None)
            return (i, t_list, seed)

    def _work(self):
        """Worker thread.  Whatever ends it -- the normal end of the source, or an exception anywhere in the body (stream creation,
        the seed draw inside _take, the import) -- `_live` is decremented and the consumer woken in the `finally`; an unexpected
        failure is posted as the result of the next batch index, so the consumer raises it instead of waiting forever."""
        fatal = None
        try:
            from .tkg_module import TKG_Module
            from . import _lib
            stream = torch.cuda.Stream(self.device) if self.device is not None else None
            while True:
                self._slots.acquire()
                job = self._take()
                if job is None:
                    self._slots.release()
                    break
                i, t_list, seed = job
                try:
                    if isinstance(t_list, BaseException):
                        raise t_list
                    if seed is not None:
                        TKG_Module._rng_override.rng = np.random.default_rng(seed)
                    if self.cooperative:
                        _lib.coop_begin(self._gate)
                    if stream is None:
                        wb = self.model.prepare(t_list, self.seq_len, self.train)
                    else:
                        with torch.cuda.stream(stream):
                            wb = self.model.prepare(t_list, self.seq_len, self.train)
                            wb.ready = torch.cuda.Event()
                            wb.ready.record(stream)
                    res = ("ok", wb)
                except BaseException as e:                          # surfaced in the consumer, at this batch's position
                    res = ("err", e)
                    with self._src_lock:
                        self._stop = True
                finally:
                    TKG_Module._rng_override.rng = None
                    _lib.coop_end()
                with self._cv:
                    self._done[i] = res
                    self._cv.notify_all()
        except BaseException as e:                                  # outside a batch: no index of its own
            fatal = e
        finally:
            with self._src_lock:
                if fatal is not None:
                    self._stop = True
                    i = self._n_taken                               # the next index nobody will prepare now
                    self._n_taken += 1
            with self._cv:
                if fatal is not None:
                    self._done[i] = ("err", fatal)
                self._live -= 1
                self._cv.notify_all()

    def _retire(self, wb):
        """Called when the consumer is done issuing work for `wb`."""
        if self.device is None:
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._inflight.append((ev, wb))
        while self._inflight and self._inflight[0][0].query():
            self._inflight.popleft()

    # ---- consumer --------------------------------------------------------------------------------------------
    def __iter__(self):
        import sys
        old_interval = sys.getswitchinterval()
        import os
        sys.setswitchinterval(min(old_interval, float(os.environ.get("TEMP_SWITCH_INTERVAL", "5e-5"))))   # all threads issue many short calls: hand the GIL over quickly
                                                              # (50 us against 200 us: 5.35 against 5.7 ms per S-gdelt step, mean of 8 runs)
        try:
            yield from self._iterate()
        finally:
            with self._src_lock:
                self._stop = True
            for _ in self.threads:                              # wake workers blocked on the window
                self._slots.release()
            sys.setswitchinterval(old_interval)

    def _iterate(self):
        self._src = iter(self.batches)
        self._src_lock = threading.Lock()
        self._cv = threading.Condition()
        self._slots = threading.Semaphore(self.window)
        self._done, self._n_taken, self._stop, self._live = {}, 0, False, self.workers
        self.threads = [threading.Thread(target=self._work, daemon=True) for _ in range(self.workers)]
        for th in self.threads:
            th.start()
        i = 0
        while True:
            with self._cv:
                while i not in self._done and self._live > 0:
                    self._cv.wait()
                if i not in self._done:                         # every worker has ended and batch i was never taken: the end
                    pending = [v for _, v in sorted(self._done.items()) if v[0] == "err"]
                    if pending:                                 # (an error posted past the end must not be swallowed as "no more data")
                        raise pending[0][1]
                    break
                kind, item = self._done.pop(i)
            self._slots.release()
            if kind == "err":
                raise item
            if getattr(item, "ready", None) is not None:
                torch.cuda.current_stream(self.device).wait_event(item.ready)
            self._gate.clear()                                  # the consumer issues a step: the workers' Python parks at its next stage boundary
            try:
                yield item
            finally:
                self._gate.set()                                # ... and runs while the consumer waits for the next batch (or has left the loop)
            self._retire(item)
            item = None
            i += 1
        for th in self.threads:
            th.join()
        if self.device is not None:
            torch.cuda.current_stream(self.device).synchronize()
        self._inflight.clear()
