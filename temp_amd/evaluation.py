"""Filtered link-prediction ranking with the interface of the reference's EvaluationFilter
(utils/evaluation.py:6-106): for every test triple, score the true subject (resp. object) against
ALL entities, mask the other entities known to be true for the same (relation, object) (resp.
(subject, relation)) at that timestamp in train+valid+test, and report the 1-indexed rank.

Restructured for the device:
  * the known-true sets are sorted composite keys, the per-triple filter lists come out of two
    `searchsorted` calls (no per-triple Python dict lookups) and are cached per (timestamp, split, mode)
    on the device;
  * DistMult / ComplEx are bilinear, so the P x N score matrix is ONE GEMM of the folded query against the
    all-entity table (`temp_linear`), never a (P, N, D) broadcast;
  * the rank is the target's position in a stable descending order, computed by counting
    (`temp_filtered_rank`) instead of sorting."""
import numpy as np
import torch

from . import scores as S
from .backend import get_backend


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
    def filter_lists(prefix, keys, num_ent):
        """CSR lists of the global entity ids that form a known-true triple with every `prefix`:
        -> (ptr [P+1] int32, ids [ptr[-1]] int32), each row's ids unique and ascending."""
        lo = np.searchsorted(keys, prefix * num_ent, side="left")
        hi = np.searchsorted(keys, (prefix + 1) * num_ent, side="left")
        cnt = hi - lo
        ptr = np.zeros(prefix.shape[0] + 1, dtype=np.int64)
        np.cumsum(cnt, out=ptr[1:])
        pos = np.repeat(lo - ptr[:-1], cnt) + np.arange(int(ptr[-1]), dtype=np.int64)
        return ptr.astype(np.int32), (keys[pos] % num_ent).astype(np.int32)

    def _mode_inputs(self, mode, samples, graph, time, num_ent, dev):
        """(target [P], filt_ptr [P+1], filt_ids) on `dev` for one corruption mode.  Cached ON the graph object (so the entry
        dies with the graph: no id() reuse) together with the sample tensor it was built from; any other `samples` -- a
        subset, a re-ordering, different triples -- rebuilds the lists."""
        cache = graph.__dict__.setdefault("_filter_lists", {})
        key = (id(self), time, mode, str(dev))
        got = cache.get(key)
        if got is not None:
            ref = got[0]
            same = ref is samples or (ref.shape == samples.shape and ref.device == samples.device and bool(torch.equal(ref, samples)))
            if not same:
                got = None
        if got is None:
            samples_np = samples.detach().cpu().numpy().astype(np.int64)
            R, tails, heads = self._true_keys(time, num_ent)
            gid = graph.gids
            if mode == "tail":
                prefix, keys, tgt = samples_np[:, 0] * R + samples_np[:, 1], tails, gid[samples_np[:, 2]]
            else:
                prefix, keys, tgt = samples_np[:, 2] * R + samples_np[:, 1], heads, gid[samples_np[:, 0]]
            ptr, ids = self.filter_lists(prefix, keys, num_ent)
            got = cache[key] = (samples.detach().clone(), torch.from_numpy(tgt.astype(np.int32)).to(dev), torch.from_numpy(ptr).to(dev),
                                torch.from_numpy(ids).to(dev))
        target, ptr, ids = got[1:]
        assert target.shape[0] == samples.shape[0] and ptr.shape[0] == samples.shape[0] + 1, "filter lists do not match the samples"
        return target, ptr, ids

    def calc_metrics_single_graph(self, ent_mean, rel_enc_means, all_ent_embeds, samples, graph, time, eval_bz=100):
        """-> ranks (2P,) int64, subject-corruption ranks first, then object-corruption (reference order)."""
        with torch.no_grad():
            dev = all_ent_embeds.device
            num_ent = all_ent_embeds.shape[0]
            time = int(time)
            P = samples.shape[0]
            if P == 0:
                return torch.zeros(0, dtype=torch.int64, device=dev)
            name = getattr(self.args, "score_function", None)
            fused = name in ("distmult", "complex") and num_ent % 4 == 0 and all_ent_embeds.shape[1] % 4 == 0
            out = {}
            for mode in ("head", "tail"):
                target, ptr, ids = self._mode_inputs(mode, samples, graph, time, num_ent, dev)
                known = ent_mean[samples[:, 0] if mode == "tail" else samples[:, 2]]
                r = rel_enc_means[samples[:, 1]]
                if fused:
                    q = S.bilinear_query(name, known, r, mode).contiguous()
                    score = get_backend().linear(q, all_ent_embeds.contiguous(), True)
                    out[mode] = get_backend().filtered_rank(score, target, ptr, ids)
                    continue
                ranks = []
                for a in range(0, P, eval_bz):
                    b = min(P, a + eval_bz)
                    if mode == "tail":
                        score = self.calc_score(known[a:b], r[a:b], all_ent_embeds, mode="tail")
                    else:
                        score = self.calc_score(all_ent_embeds, r[a:b], known[a:b], mode="head")
                    lo, hi = int(ptr[a]), int(ptr[b])
                    ranks.append(get_backend().filtered_rank(_pad4(score), target[a:b], ptr[a:b + 1] - lo, ids[lo:hi]))
                out[mode] = torch.cat(ranks)
            return torch.cat([out["head"], out["tail"]])


def _w_col(w, P, like):
    """Per-triple mixing weight as a (P, 1) column (the reference passes (P, 1) tensors; scalars and (P,) vectors broadcast)."""
    w = torch.as_tensor(w, dtype=like.dtype, device=like.device)
    if w.dim() == 0:
        return w.view(1, 1).expand(P, 1)
    return w.reshape(P, 1)


class _MixedRankFilter(EvaluationFilter):
    """Shared part of the two post-ensemble filters: for one corruption mode, the score matrix of the mixed (local, temporal)
    representation, then the filtered rank (the filter list replaces the reference's -10e6 overwrite, as in the base class)."""

    def _score_matrix(self, known, r, all_embeds, mode, eval_bz):
        name = getattr(self.args, "score_function", None)
        P, num_ent = known.shape[0], all_embeds.shape[0]
        if name in ("distmult", "complex") and num_ent % 4 == 0 and all_embeds.shape[1] % 4 == 0:
            q = S.bilinear_query(name, known, r, mode).contiguous()
            return get_backend().linear(q, all_embeds.contiguous(), True)
        rows = []
        for a in range(0, P, eval_bz):
            b = min(P, a + eval_bz)
            rows.append(self.calc_score(known[a:b], r[a:b], all_embeds, mode="tail") if mode == "tail"
                        else self.calc_score(all_embeds, r[a:b], known[a:b], mode="head"))
        return torch.cat(rows)

    def _rank(self, score, mode, samples, graph, time, num_ent, dev):
        target, ptr, ids = self._mode_inputs(mode, samples, graph, time, num_ent, dev)
        return get_backend().filtered_rank(_pad4(score), target, ptr, ids)


class PostEnsembleEvaluationFilter(_MixedRankFilter):
    """utils/post_evaluation.py:63-134: SCORE-level ensemble.  score = w * score(local) + (1 - w) * score(temporal), per triple
    (the reference masks each score matrix before mixing; both masked values are -10e6, so filtering after the mix is the
    same).  As in the reference, the object-corruption ranks use `weight_subject`, the subject-corruption ranks `weight_object`."""

    def calc_metrics_single_graph(self, ent_embed_local, ent_embed_temporal, rel_enc_mean, all_embeds_g_local, all_embeds_g_temporal,
                                  weight_subject, weight_object, samples, graph, time, eval_bz=100):
        with torch.no_grad():
            dev, num_ent, time, P = all_embeds_g_local.device, all_embeds_g_local.shape[0], int(time), samples.shape[0]
            if P == 0:
                return torch.zeros(0, dtype=torch.int64, device=dev)
            r = rel_enc_mean[samples[:, 1]]
            out = {}
            for mode, w in (("head", weight_object), ("tail", weight_subject)):
                sel = samples[:, 0] if mode == "tail" else samples[:, 2]
                s_loc = self._score_matrix(ent_embed_local[sel], r, all_embeds_g_local, mode, eval_bz)
                s_tmp = self._score_matrix(ent_embed_temporal[sel], r, all_embeds_g_temporal, mode, eval_bz)
                wc = _w_col(w, P, s_loc)
                out[mode] = self._rank(wc * s_loc + (1 - wc) * s_tmp, mode, samples, graph, time, num_ent, dev)
            return torch.cat([out["head"], out["tail"]])


class PostEvaluationFilter(_MixedRankFilter):
    """utils/post_evaluation.py:7-60: EMBEDDING-level ensemble with four per-triple weights.  The known entity is mixed as
    w * local + (1 - w) * temporal; every candidate likewise with the other weight.  distmult / complex are linear in the
    candidate, so the candidate mix is applied to the two score matrices instead of materialising (P, N_ents, D) candidates;
    any other scorer takes the reference's literal route."""

    def calc_metrics_single_graph(self, ent_embed_loc, ent_embed_rec, rel_enc_means, all_embeds_g_loc, all_embeds_g_rec, samples,
                                  weight_subject_query_subject_embed, weight_subject_query_object_embed,
                                  weight_object_query_subject_embed, weight_object_query_object_embed, graph, time, eval_bz=100):
        with torch.no_grad():
            dev, num_ent, time, P = all_embeds_g_loc.device, all_embeds_g_loc.shape[0], int(time), samples.shape[0]
            if P == 0:
                return torch.zeros(0, dtype=torch.int64, device=dev)
            r = rel_enc_means[samples[:, 1]]
            name = getattr(self.args, "score_function", None)
            out = {}
            for mode, w_s, w_o in (("head", weight_subject_query_subject_embed, weight_subject_query_object_embed),
                                   ("tail", weight_object_query_subject_embed, weight_object_query_object_embed)):
                ws, wo = _w_col(w_s, P, ent_embed_loc), _w_col(w_o, P, ent_embed_loc)
                sel = samples[:, 0] if mode == "tail" else samples[:, 2]
                w_known, w_cand = (ws, wo) if mode == "tail" else (wo, ws)     # tail: subject known; head: object known
                known = w_known * ent_embed_loc[sel] + (1 - w_known) * ent_embed_rec[sel]
                if name in ("distmult", "complex"):
                    score = w_cand * self._score_matrix(known, r, all_embeds_g_loc, mode, eval_bz) \
                        + (1 - w_cand) * self._score_matrix(known, r, all_embeds_g_rec, mode, eval_bz)
                else:
                    rows = []
                    for a in range(0, P, eval_bz):
                        b = min(P, a + eval_bz)
                        cand = w_cand[a:b].unsqueeze(-1) * all_embeds_g_loc.unsqueeze(0) + (1 - w_cand[a:b]).unsqueeze(-1) * all_embeds_g_rec.unsqueeze(0)
                        rows.append(self.calc_score(known[a:b], r[a:b], cand, mode="tail") if mode == "tail"
                                    else self.calc_score(cand, r[a:b], known[a:b], mode="head"))
                    score = torch.cat(rows)
                out[mode] = self._rank(score, mode, samples, graph, time, num_ent, dev)
            return torch.cat([out["head"], out["tail"]])


def _pad4(score):
    """Score rows padded to a multiple of 4 columns with -inf (sigmoid 0 at ids above every target: never ahead)."""
    n = score.shape[1]
    if n % 4 == 0:
        return score.contiguous()
    pad = score.new_full((score.shape[0], 4 - n % 4), float("-inf"))
    return torch.cat([score, pad], dim=1).contiguous()
