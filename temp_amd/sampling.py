"""Filtered negative sampling with the interface of the reference's CorruptTriples
(utils/CorrptTriples.py:7-106), vectorised with numpy instead of per-triple Python loops.

Semantics kept: per target graph, P = min(E, num_pos_facts) positives (random subset when E is
larger); for each positive (h, r, t) draw `negative_rate` corrupted tails (resp. heads) uniformly
over ALL entities (global ids), rejecting the graph's own true tails of (h, r) (resp. true heads
of (r, t)); column 0 of each row is the true entity's GLOBAL id; labels are all 0.
The random stream differs from the reference's (it uses unseeded np.random); parity
tests inject the reference's recorded samples instead.
"""
import numpy as np
import torch

from . import _lib


class CorruptTriples:
    def __init__(self, args, graph_dict_train, seed=None):
        self.args = args
        self.negative_rate = args.negative_rate
        self.num_pos_facts = args.num_pos_facts
        self.graph_dict_train = graph_dict_train
        self.rng = np.random.default_rng(seed)

    def single_graph_negative_sampling(self, t, g, num_ents):
        """-> (triples (P,3) int64 local ids, neg_tail (P,1+K), neg_head (P,1+K) global ids, labels (P,))"""
        trip = np.stack([g.src, g.rel, g.dst], axis=1)
        P = min(trip.shape[0], self.num_pos_facts)
        if self.num_pos_facts < trip.shape[0]:
            trip = trip[self.rng.permutation(trip.shape[0])[:P]]
        K = self.negative_rate
        gid = g.gids
        neg_tail = np.empty((P, 1 + K), dtype=np.int64)
        neg_head = np.empty((P, 1 + K), dtype=np.int64)
        neg_tail[:, 0] = gid[trip[:, 2]]
        neg_head[:, 0] = gid[trip[:, 0]]
        # true sets of this snapshot, keyed (h, r) / (r, t), as sorted composite keys over GLOBAL ids
        all_trip = np.stack([g.src, g.rel, g.dst], axis=1)
        R = int(all_trip[:, 1].max()) + 1 if all_trip.shape[0] else 1
        key_tail = (all_trip[:, 0] * R + all_trip[:, 1]) * num_ents + gid[all_trip[:, 2]]
        key_head = (all_trip[:, 2] * R + all_trip[:, 1]) * num_ents + gid[all_trip[:, 0]]
        key_tail.sort()
        key_head.sort()
        neg_tail[:, 1:] = self._draw(trip[:, 0] * R + trip[:, 1], key_tail, num_ents, K)
        neg_head[:, 1:] = self._draw(trip[:, 2] * R + trip[:, 1], key_head, num_ents, K)
        labels = np.zeros(P, dtype=np.int64)
        return (torch.from_numpy(trip), torch.from_numpy(neg_tail), torch.from_numpy(neg_head), torch.from_numpy(labels))

    def _draw(self, prefix, sorted_keys, num_ents, K):
        P = prefix.shape[0]
        out = self.rng.integers(0, num_ents, size=(P, K))
        for _ in range(64):
            keys = prefix[:, None] * num_ents + out
            pos = np.searchsorted(sorted_keys, keys)
            pos[pos >= sorted_keys.shape[0]] = 0
            bad = sorted_keys[pos] == keys if sorted_keys.shape[0] else np.zeros_like(keys, dtype=bool)
            nbad = int(bad.sum())
            if nbad == 0:
                break
            out[bad] = self.rng.integers(0, num_ents, size=nbad)
        return out


class DeviceCorruptTriples(CorruptTriples):
    """Same sampler with the draws and the true-triple filter on the model's device (torch ops: randint +
    searchsorted against the snapshot's sorted composite keys, cached per timestamp).  No per-step host work and no
    H2D copy of the (P, 1 + negative_rate) candidate lists (3000 x 501 int64 = 12 MB per direction per graph)."""

    def __init__(self, args, graph_dict_train, device, seed=None):
        super().__init__(args, graph_dict_train, seed)
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(0 if seed is None else int(seed))
        self._cache = {}

    def _graph_tensors(self, t, g, num_ents):
        c = self._cache.get((t, id(g)))
        if c is None:
            dev = self.device
            trip = torch.from_numpy(np.stack([g.src, g.rel, g.dst], axis=1)).to(dev)
            gid = torch.from_numpy(g.gids).to(dev)
            R = int(g.rel.max()) + 1 if g.rel.shape[0] else 1
            key_tail = torch.sort((trip[:, 0] * R + trip[:, 1]) * num_ents + gid[trip[:, 2]]).values
            key_head = torch.sort((trip[:, 2] * R + trip[:, 1]) * num_ents + gid[trip[:, 0]]).values
            c = self._cache[(t, id(g))] = (trip, gid, R, key_tail, key_head)
        return c

    def _draw_dev(self, prefix, sorted_keys, num_ents, K):
        P = prefix.shape[0]
        out = torch.randint(0, num_ents, (P, K), device=self.device, generator=self.gen)
        if sorted_keys.shape[0] == 0:
            return out
        for _ in range(16):                                   # rejection rounds; a round redraws only the collisions
            keys = prefix[:, None] * num_ents + out
            pos = torch.searchsorted(sorted_keys, keys).clamp_(max=sorted_keys.shape[0] - 1)
            bad = sorted_keys[pos] == keys
            redraw = torch.randint(0, num_ents, (P, K), device=self.device, generator=self.gen)
            out = torch.where(bad, redraw, out)
        return out

    def single_graph_negative_sampling(self, t, g, num_ents):
        trip_all, gid, R, key_tail, key_head = self._graph_tensors(t, g, num_ents)
        E = trip_all.shape[0]
        P = min(E, self.num_pos_facts)
        trip = trip_all
        if self.num_pos_facts < E:
            trip = trip_all[torch.randperm(E, device=self.device, generator=self.gen)[:P]]
        K = self.negative_rate
        neg_tail = torch.cat([gid[trip[:, 2]].view(-1, 1), self._draw_dev(trip[:, 0] * R + trip[:, 1], key_tail, num_ents, K)], dim=1)
        neg_head = torch.cat([gid[trip[:, 0]].view(-1, 1), self._draw_dev(trip[:, 2] * R + trip[:, 1], key_head, num_ents, K)], dim=1)
        return trip, neg_tail, neg_head, torch.zeros(P, dtype=torch.int64, device=self.device)


class TrueSetStore:
    """Known-true entity lists of the train snapshots, resident on the device for temp_corrupt_sample.

    Per snapshot (built on first use, appended to ONE growing device array): for every edge e = (h, r, t) the slice of
    global ids that are true tails of (h, r) and the slice that are true heads of (r, t) in that snapshot
    (get_true_head_and_tail_per_graph, utils/CorrptTriples.py:87-106), each ascending.  `rows(t, edge_idx)` returns the
    absolute [lo, hi) offsets of the chosen edges' slices; offsets stay valid when the array grows."""

    def __init__(self, graph_dict_train, num_ents, device):
        self.graphs, self.N, self.device = graph_dict_train, int(num_ents), torch.device(device)
        self._snap = {}
        self.ids = torch.zeros(1024, dtype=torch.int32, device=self.device)
        self.size = 0

    def _append(self, arr):
        n = arr.shape[0]
        if self.size + n > self.ids.shape[0]:
            grown = torch.zeros(max(2 * self.ids.shape[0], self.size + n), dtype=torch.int32, device=self.device)
            grown[:self.size] = self.ids[:self.size]
            self.ids = grown
        self.ids[self.size:self.size + n] = _lib.to_device(arr, self.device)
        base = self.size
        self.size += n
        return base

    def snapshot(self, t):
        s = self._snap.get(t)
        if s is None:
            with _lib.create_lock:
                s = self._snap.get(t)
                if s is None:
                    s = self._build_snapshot(t)
                    _lib.publish(self.device)            # the appended slice has landed before any other stream can be told of it
                    self._snap[t] = s
        return s

    def _build_snapshot(self, t):
        g, N = self.graphs[t], self.N
        gid = g.gids.astype(np.int64)
        src, rel, dst = g.src.astype(np.int64), g.rel.astype(np.int64), g.dst.astype(np.int64)
        R = int(rel.max()) + 1 if rel.shape[0] else 1
        pt, ph = src * R + rel, dst * R + rel
        kt = np.unique(pt * N + gid[dst]) if rel.shape[0] else np.zeros(0, np.int64)
        kh = np.unique(ph * N + gid[src]) if rel.shape[0] else np.zeros(0, np.int64)
        base = self._append(np.concatenate([kt % N, kh % N]).astype(np.int32))
        off_h = base + kt.shape[0]
        s = dict(
            tail_lo=(base + np.searchsorted(kt, pt * N)).astype(np.int32), tail_hi=(base + np.searchsorted(kt, (pt + 1) * N)).astype(np.int32),
            head_lo=(off_h + np.searchsorted(kh, ph * N)).astype(np.int32), head_hi=(off_h + np.searchsorted(kh, (ph + 1) * N)).astype(np.int32))
        # addresses of the per-edge arrays for the host planner (the arrays live as long as the snapshot / this store)
        s["ptrs"] = np.array([g.src.ctypes.data, g.rel.ctypes.data, g.dst.ctypes.data, g.gids.ctypes.data, s["tail_lo"].ctypes.data,
                              s["tail_hi"].ctypes.data, s["head_lo"].ctypes.data, s["head_hi"].ctypes.data], dtype=np.int64)
        return s


def plan_batch_loss(store, times, graphs, row_offsets, num_pos_facts, rng, n_rows, n_rel_rows, device):
    """Everything of a window batch's link-prediction loss that does not depend on the negative draws, built on the host
    (no device round trip) and uploaded once: per target graph the P = min(E, num_pos_facts) positives (a random subset
    when E is larger, utils/CorrptTriples.py:37-40), stacked as [tail-corruption rows ; head-corruption rows]:
      known / rel / is_tail   operands of the folded query      truth, lo, hi    inputs of temp_corrupt_sample
      weights (1 / P), splits, known_inv / rel_inv (static inverses for the deterministic backward), triples (host, per graph)."""
    from . import _hostlib
    from . import functional as TF
    ptrs, idxs = [], []
    for t, g in zip(times, graphs):
        E = g.number_of_edges()
        P = min(E, num_pos_facts)
        idxs.append(_hostlib.sample_subset(E, P, rng) if num_pos_facts < E else np.arange(E, dtype=np.int64))
        ptrs.append(store.snapshot(t)["ptrs"])
    packed, weights, trip_all, n_pos, block = _hostlib.plan_loss(np.stack(ptrs) if ptrs else np.zeros((0, 8), np.int64), idxs,
                                                                 np.asarray(row_offsets, dtype=np.int64))
    if packed.shape[1] == 0:
        return None
    _lib.pause_point()
    ends = np.cumsum(block)                                     # a graph's block = 2 P rows (+ weight-0 padding to a multiple of 4)
    splits = [(int(e - k), int(e)) for e, k in zip(ends, block)]
    tcut = np.cumsum(n_pos)
    triples = [trip_all[int(e - p):int(e)] for e, p in zip(tcut, n_pos)]
    dev_i = _lib.to_device(packed, device)                       # ONE upload for the six index vectors
    return dict(known=dev_i[0], rel=dev_i[1], is_tail=dev_i[2], truth=dev_i[3], lo=dev_i[4], hi=dev_i[5], ids=store.ids,
                weights=_lib.to_device(weights, device), splits=splits, triples=triples,
                known_inv=TF.gather_inverse(packed[0], n_rows, device), rel_inv=TF.gather_inverse(packed[1], n_rel_rows, device))
