"""Filtered link-prediction ranking with the interface of the reference's EvaluationFilter
(utils/evaluation.py:6-106): for every test triple, score the true subject (resp. object) against
ALL entities, mask the other entities known to be true for the same (relation, object) (resp.
(subject, relation)) at that timestamp in train+valid+test, and report the 1-indexed rank.

Vectorised: the known-true sets are sorted composite keys (searchsorted instead of per-triple Python
dict lookups); the rank is the target's position in a stable descending order, computed by counting
instead of sorting."""
import numpy as np
import torch


class EvaluationFilter:
    def __init__(self, args, calc_score, graph_dict_train, graph_dict_val, graph_dict_test):
        self.args = args
        self.calc_score = calc_score
        self.graph_dicts = (graph_dict_train, graph_dict_val, graph_dict_test)
        self._keys = {}

    def _true_keys(self, time, num_ent):
        """Sorted keys (h*R + r)*N + global(t) and (t*R + r)*N + global(h) over the three splits at `time`."""
        k = self._keys.get(time)
        if k is None:
            trip = [np.stack([g[time].src, g[time].rel, g[time].dst], axis=1) for g in self.graph_dicts if time in g]
            trip = np.concatenate(trip, axis=0) if trip else np.zeros((0, 3), np.int64)
            gid = next(g[time].gids for g in self.graph_dicts if time in g)
            R = int(trip[:, 1].max()) + 1 if trip.shape[0] else 1
            tails = np.unique((trip[:, 0] * R + trip[:, 1]) * num_ent + gid[trip[:, 2]])
            heads = np.unique((trip[:, 2] * R + trip[:, 1]) * num_ent + gid[trip[:, 0]])
            k = self._keys[time] = (R, tails, heads)
        return k

    @staticmethod
    def _mask(prefix, keys, num_ent, target_global):
        """Boolean (P, N) mask of the entities that form a known-true triple with `prefix`, target excluded."""
        P = prefix.shape[0]
        lo = np.searchsorted(keys, prefix * num_ent, side="left")
        hi = np.searchsorted(keys, (prefix + 1) * num_ent, side="left")
        cnt = hi - lo
        rows = np.repeat(np.arange(P), cnt)
        pos = np.concatenate([np.arange(a, b) for a, b in zip(lo, hi)]) if cnt.sum() else np.zeros(0, np.int64)
        mask = np.zeros((P, num_ent), dtype=bool)
        mask[rows, keys[pos] % num_ent] = True
        mask[np.arange(P), target_global] = False
        return mask

    def calc_metrics_single_graph(self, ent_mean, rel_enc_means, all_ent_embeds, samples, graph, time, eval_bz=100):
        """-> ranks (2P,) int64, subject-corruption ranks first, then object-corruption (reference order)."""
        with torch.no_grad():
            dev = all_ent_embeds.device
            num_ent = all_ent_embeds.shape[0]
            time = int(time)
            s_np = samples.detach().cpu().numpy().astype(np.int64)
            R, tails, heads = self._true_keys(time, num_ent)
            gid = graph.gids
            out = {}
            for mode in ("head", "tail"):
                if mode == "tail":
                    prefix, keys, tgt = s_np[:, 0] * R + s_np[:, 1], tails, gid[s_np[:, 2]]
                else:
                    prefix, keys, tgt = s_np[:, 2] * R + s_np[:, 1], heads, gid[s_np[:, 0]]
                mask = torch.from_numpy(self._mask(prefix, keys, num_ent, tgt)).to(dev)
                target = torch.from_numpy(tgt).to(dev)
                ent_index = torch.arange(num_ent, device=dev).view(1, -1)
                ranks = []
                for a in range(0, s_np.shape[0], eval_bz):
                    b = min(s_np.shape[0], a + eval_bz)
                    r = rel_enc_means[samples[a:b, 1]]
                    if mode == "tail":
                        score = self.calc_score(ent_mean[samples[a:b, 0]], r, all_ent_embeds, mode="tail")
                    else:
                        score = self.calc_score(all_ent_embeds, r, ent_mean[samples[a:b, 2]], mode="head")
                    score = torch.sigmoid(torch.where(mask[a:b], torch.full_like(score, -10e6), score))
                    ts = score.gather(1, target[a:b].view(-1, 1))
                    # position in a stable descending sort: strictly better candidates, then equal-scored
                    # candidates with a smaller entity id (sigmoid in fp32 produces real ties near 0.5)
                    ahead = (score > ts) | ((score == ts) & (ent_index < target[a:b].view(-1, 1)))
                    ranks.append(ahead.sum(dim=1) + 1)
                out[mode] = torch.cat(ranks) if ranks else torch.zeros(0, dtype=torch.int64, device=dev)
            return torch.cat([out["head"], out["tail"]])
