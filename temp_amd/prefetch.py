"""Host-side pipeline for training loops: window batches are *prepared* (plan, row maps, union of the cached
snapshot views, uploads -- temp_amd.dynamic_rgcn.DynamicRGCN.prepare) by a background thread a few steps ahead of
the GPU, so the ~30 ms of host work per batch overlaps the previous steps instead of preceding each one.

    for wb in BatchPrefetcher(model, batches, seq_len=model.train_seq_len):
        loss = model.run_loss(wb)
        ...

The reference builds its batched DGL graphs inline at the top of every forward (models/DynamicRGCN.py:76-94);
this replaces that with a bounded queue.  numpy releases the GIL in the sorting / concatenation calls that
dominate `prepare`, so one thread is enough.
"""
import queue
import threading


class BatchPrefetcher:
    def __init__(self, model, batches, seq_len=None, train=True, depth=2):
        self.model, self.batches, self.train = model, batches, train
        self.seq_len = seq_len if seq_len is not None else model.train_seq_len
        self.q = queue.Queue(maxsize=max(1, depth))
        self.thread = None

    def _work(self):
        try:
            for t_list in self.batches:
                self.q.put(("ok", self.model.prepare(t_list, self.seq_len, self.train)))
        except BaseException as e:          # surfaced in the consumer
            self.q.put(("err", e))
            return
        self.q.put(("end", None))

    def __iter__(self):
        self.thread = threading.Thread(target=self._work, daemon=True)
        self.thread.start()
        while True:
            kind, item = self.q.get()
            if kind == "end":
                break
            if kind == "err":
                raise item
            yield item
        self.thread.join()
