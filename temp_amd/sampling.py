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
