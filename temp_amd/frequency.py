"""Per-timestamp frequency features behind the post-ensemble models' learned score-mixing weights.

The reference builds them in utils/DropEdge.py:34-82 (`DropEdge.count_frequency`) from utils/frequency.py:30-53
(`count_freq_per_time`, `calc_aggregated_statistics`): for a target timestamp t and every item present at t in the TRAINING
quadruples -- subject s, object o, relation r, pair (s, r), pair (o, r) -- the aggregated count

    agg[t][item] = sum over cur in [max(0, t - L + 1), upper), cur != t, of count[cur][item]
    upper = t                               (uni-directional modules)
          = min(T + 1, t + L)               (modules with "Bi" in the name; T = number of train timestamps)

and `calc_ensemble_ratio` (models/PostDynamicRGCN.py:425-461) reads, per triple (s, r, o) of the target graph,
    subject features = [agg_obj[t][o], agg_rel[t][r], agg_obj_rel[t][(o, r)]]
    object  features = [agg_sub[t][s], agg_rel[t][r], agg_sub_rel[t][(s, r)]]            (0 for an item unseen at t).

Here the counts come from the train snapshots themselves (the reference's train graphs hold exactly the train quadruples of
their timestamp, no inverse edges: utils/dataset.py:151-232), as sorted unique keys + counts per timestamp; the window
aggregate of a target timestamp is built on first use with searchsorted joins and cached.  Host-side integer work, not on
the encoder path.
"""
import numpy as np

STATS = ("sub", "obj", "rel", "sub_rel", "obj_rel")


class FrequencyTables:
    def __init__(self, graph_dict_train, seq_len, future, num_rel_rows=1):
        self.seq_len = int(seq_len)
        self.future = bool(future)
        self.n_times = len(graph_dict_train)               # DropEdge.max_time_step = len(train_times)
        self.rel_base = max(1, int(num_rel_rows))          # pair keys: entity * rel_base + relation
        for g in graph_dict_train.values():
            if g.number_of_edges():
                self.rel_base = max(self.rel_base, int(g.rel.max()) + 1)
        self._per_time = {}
        for t, g in graph_dict_train.items():
            if g.number_of_edges() == 0:
                continue
            s, o, r = g.gids[g.src], g.gids[g.dst], g.rel
            keys = dict(sub=s, obj=o, rel=r, sub_rel=s * self.rel_base + r, obj_rel=o * self.rel_base + r)
            self._per_time[int(t)] = {k: np.unique(v.astype(np.int64), return_counts=True) for k, v in keys.items()}
        self._agg = {}

    def _aggregate(self, t):
        """stat -> (sorted keys present at t, aggregated counts over the window around t)."""
        a = self._agg.get(t)
        if a is not None:
            return a
        empty = (np.zeros(0, np.int64), np.zeros(0, np.int64))
        here = self._per_time.get(t)
        if here is None:
            a = {k: empty for k in STATS}
        else:
            upper = t if not self.future else min(self.n_times + 1, t + self.seq_len)
            a = {}
            for k in STATS:
                keys = here[k][0]
                tot = np.zeros(keys.shape[0], dtype=np.int64)
                for cur in range(max(0, t - self.seq_len + 1), upper):
                    if cur == t or cur not in self._per_time:
                        continue
                    ck, cc = self._per_time[cur][k]
                    pos = np.minimum(np.searchsorted(ck, keys), ck.shape[0] - 1)
                    hit = ck[pos] == keys
                    tot[hit] += cc[pos[hit]]
                a[k] = (keys, tot)
        self._agg[t] = a
        return a

    def _lookup(self, t, stat, q):
        keys, tot = self._aggregate(t)[stat]
        q = np.asarray(q, dtype=np.int64)
        if keys.shape[0] == 0:
            return np.zeros(q.shape[0], dtype=np.float32)
        pos = np.minimum(np.searchsorted(keys, q), keys.shape[0] - 1)
        return np.where(keys[pos] == q, tot[pos], 0).astype(np.float32)

    def features(self, t, s, r, o):
        """Global ids of the triples of target timestamp t -> (subject features (n, 3), object features (n, 3)), float32."""
        t = int(t)
        s, r, o = (np.asarray(x, dtype=np.int64).reshape(-1) for x in (s, r, o))
        rel = self._lookup(t, "rel", r)
        sub_f = np.stack([self._lookup(t, "obj", o), rel, self._lookup(t, "obj_rel", o * self.rel_base + r)], axis=1)
        obj_f = np.stack([self._lookup(t, "sub", s), rel, self._lookup(t, "sub_rel", s * self.rel_base + r)], axis=1)
        return sub_f, obj_f
