"""Impute / post-ensemble window models -- the reference's models/PostDynamicRGCN.py and models/PostBiDynamicRGCN.py
(`--impute`, `--post-ensemble`; BASELINE config 3 = BiGRRGCN --rec-only-last-layer --post-ensemble on ICEWS05-15).

On top of the (Bi)DynamicRGCN window loop these models keep a THIRD history stream: the "local" layer-2 state of every
node, i.e. the layer-2 RGCN output BEFORE the GRU (models/PostDynamicRGCN.py:33-42, models/PostBiDynamicRGCN.py:77-101), and
the target position returns (local, temporal) embeddings (models/PostBiDynamicRGCN.py:53-75).  The all-entity pass is
RRGCN / BiRRGCN.forward_isolated_impute (impute models) or forward_post_ensemble_isolated (post-ensemble models).

Execution: with --rec-only-last-layer the batched step of the parent class is reused unchanged -- the local stream is
simply the GRU INPUT rows of the step (one launch per RGCN layer over all visited snapshots + the persistent chain kernels);
otherwise the reference-granular path walks the positions through forward_post_ensemble(_one_direction).  Histories are
row maps into the last executed position (window.ChainPlan), never dense (bsz, N_ents, D) tensors.

The learned score-mixing weights of PostEnsemble* (`calc_ensemble_ratio`, models/PostDynamicRGCN.py:425-461): two MLPs
`subject_linear` / `object_linear` (3 -> 3 -> 1, sigmoid; same parameter names as the reference, so its checkpoints load) over
per-timestamp frequency features of every triple (temp_amd/frequency.py restates the tables of utils/DropEdge.py:34-82).
`forward(..., ensemble_weights=...)` still accepts injected weights (parity tests that replay the reference's)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import functional as TF
from .bi_dynamic_rgcn import BiDynamicRGCN
from .birrgcn import BiGRRGCNLayer
from .dynamic_rgcn import DynamicRGCN
from .rrgcn import GRRGCNLayer


def _dev_cached(owner, key, device, make):
    """A host array uploaded ONCE per (owner, key, device): `owner` is an object that lives as long as the array is valid (the
    prepared batch's ChainPlan, a Snapshot).  The per-window index uploads of the all-entity pass used to be issued inside every
    step: a host-to-device copy per window made the step un-capturable as a HIP graph and host-bound (23 ms for 8 ms of kernels at
    the S-icews0515 shape)."""
    cache = owner.__dict__.setdefault("_dev_cache", {})
    k = (key, str(device))
    t = cache.get(k)
    if t is None:
        t = cache[k] = make().to(device)
    return t


def _gids_dev(g, device):
    return _dev_cached(g, "gids", device, lambda: torch.from_numpy(np.ascontiguousarray(g.gids)))


def _final_index(plan, b, device):
    """(row of every entity in the last history output of window b or -1, time gaps (N, 1)) of a plan, on the device."""
    def make_idx():
        row_of, _ = plan.final_all(b, plan.seq_len - 1)
        return torch.from_numpy(row_of.astype(np.int32))

    def make_dt():
        _, dt = plan.final_all(b, plan.seq_len - 1)
        return torch.from_numpy(np.ascontiguousarray(dt)).view(-1, 1)
    return _dev_cached(plan, ("final_idx", b), device, make_idx), _dev_cached(plan, ("final_dt", b), device, make_dt)


def _rows_or_zero(rows, idx_t, n, d, like):
    return like.new_zeros(n, d) if rows is None else TF.gather_rows(rows, idx_t)


def _impute_evaluate(self, t_list, val):
    from .evaluation import EvaluationFilter
    if not hasattr(self, "evaluater"):
        self.evaluater = EvaluationFilter(self.args, self.calc_score, self.graph_dict_train, self.graph_dict_val, self.graph_dict_test)
    graph_dict = self.graph_dict_val if val else self.graph_dict_test
    dev = self._device()
    with torch.no_grad():
        wb = self.prepare(t_list, self.test_seq_len, train=False)
        out, hist = self.run(wb)
        ranks, losses = [], []
        for i, ent_embed in enumerate(out.split(wb.target.sizes)):
            t = wb.rows[i][-1]
            g = graph_dict[t]
            if g.number_of_edges() == 0:
                continue
            all_embeds_g = self.get_all_embeds_Gt(ent_embed, g, t, wb.plan, i, hist, wb.hist_loc)
            index_sample = torch.from_numpy(np.stack([g.src, g.rel, g.dst], axis=1)).to(dev)
            label = torch.ones(index_sample.shape[0], device=dev)
            ranks.append(self.evaluater.calc_metrics_single_graph(ent_embed, self.rel_embeds, all_embeds_g, index_sample, g, t))
            losses.append(self.link_classification_loss(ent_embed, self.rel_embeds, index_sample, label))
    ranks = torch.cat(ranks) if ranks else torch.zeros(0, dtype=torch.int64, device=dev)
    return ranks, (float(torch.stack(losses).mean().item()) if losses else float("nan"))


def _post_ensemble_evaluate(self, t_list, val):
    from .evaluation import PostEnsembleEvaluationFilter
    if not isinstance(getattr(self, "evaluater", None), PostEnsembleEvaluationFilter):
        self.evaluater = PostEnsembleEvaluationFilter(self.args, self.calc_score, self.graph_dict_train, self.graph_dict_val, self.graph_dict_test)
    graph_dict = self.graph_dict_val if val else self.graph_dict_test
    dev = self._device()
    with torch.no_grad():
        wb = self.prepare(t_list, self.test_seq_len, train=False)
        out, hist = self.run(wb)
        ranks = []
        for i, (loc, rec) in enumerate(zip(wb.out_loc.split(wb.target.sizes), out.split(wb.target.sizes))):
            t = wb.rows[i][-1]
            g = graph_dict[t]
            if g.number_of_edges() == 0:
                continue
            all_loc, all_rec = self.get_all_embeds_Gt(loc, rec, g, t, wb.plan, i, hist, wb.hist_loc)
            index_sample = torch.from_numpy(np.stack([g.src, g.rel, g.dst], axis=1)).to(dev)
            w_subject, w_object = self.calc_ensemble_ratio(index_sample, t, g)
            ranks.append(self.evaluater.calc_metrics_single_graph(loc, rec, self.rel_embeds, all_loc, all_rec, w_subject, w_object,
                                                                  index_sample, g, t))
    ranks = torch.cat(ranks) if ranks else torch.zeros(0, dtype=torch.int64, device=dev)
    return ranks, float("nan")


class _PostWindowMixin:
    """What the uni- and bidirectional post models share: slicing the local stream out of a batched run and the
    score-level ensemble loss."""

    def _chain_input_rows(self, wb, inst_id=None, step=None):
        """GRU-input rows (= local layer-2 states) of one chain instance / step of a batched run."""
        x = wb.last_x
        if wb.program is not None:
            it = wb.program.inst[inst_id]
            return x[it.x0:it.x0 + it.n]
        return x[step.row0:step.row0 + step.n_rows]

    # Reference quirk kept for parity (pinned by golden G19_post_ratio_bi): PostEnsembleBiDynamicRGCN.train_link_prediction forwards
    # to the uni-directional class with `corrupt_tail=True` hard-coded (models/PostBiDynamicRGCN.py:294-295), so its "head" scores
    # are score(s, r, all_embeds[neg_head], mode='tail') -- the head-corruption candidates scored as tails of the true subject.
    head_scored_as_tail = False

    # -- score-level ensemble (models/PostDynamicRGCN.py:357-373, 399-406) ---------------------------------------------------
    def init_freq_mlp(self):
        """PostEnsembleDynamicRGCN.init_freq_mlp, models/PostDynamicRGCN.py:328-338 (same module names => same state_dict keys)."""
        self.subject_linear = nn.Sequential(nn.Linear(3, 3), nn.ReLU(), nn.Linear(3, 1))
        self.object_linear = nn.Sequential(nn.Linear(3, 3), nn.ReLU(), nn.Linear(3, 1))

    def frequency_tables(self):
        ft = getattr(self, "_freq_tables", None)
        if ft is None:
            from .frequency import FrequencyTables
            ft = self._freq_tables = FrequencyTables(self.graph_dict_train, self.train_seq_len, "Bi" in self.args.module,
                                                     2 * self.num_rels)
        return ft

    def ensemble_features(self, triples, t, g):
        """Frequency features of the (local-id) triples of graph g at timestamp t -> (subject (n, 3), object (n, 3)) on the device."""
        tr = triples.detach().cpu().numpy() if torch.is_tensor(triples) else np.asarray(triples)
        tr = tr.reshape(-1, 3)
        sub_f, obj_f = self.frequency_tables().features(int(t), g.gids[tr[:, 0]], tr[:, 1], g.gids[tr[:, 2]])
        dev = self._device()
        return torch.from_numpy(sub_f).to(dev), torch.from_numpy(obj_f).to(dev)

    def calc_ensemble_ratio(self, triples, t, g):
        """models/PostDynamicRGCN.py:425-461 -> (weight_subject (n, 1), weight_object (n, 1)); empty tensors for no triples."""
        if not hasattr(self, "subject_linear"):
            raise NotImplementedError("calc_ensemble_ratio needs the frequency MLPs of a PostEnsemble* model")
        if len(triples) == 0:
            e = torch.zeros(0, dtype=torch.int64, device=self._device())
            return e, e
        sub_f, obj_f = self.ensemble_features(triples, t, g)
        return torch.sigmoid(self.subject_linear(sub_f)), torch.sigmoid(self.object_linear(obj_f))

    def _scores(self, ent_embed, triplets, neg_samples, all_embeds_g, corrupt_tail):
        r = self.rel_embeds[triplets[:, 1]]
        if corrupt_tail:
            return self.calc_score(ent_embed[triplets[:, 0]], r, all_embeds_g[neg_samples], mode='tail')
        return self.calc_score(all_embeds_g[neg_samples], r, ent_embed[triplets[:, 2]], mode='head')

    def batched_all_embeds_post(self, wb, out, hist, base):
        """(all_loc, all_rec) of EVERY window in one pass, or None (per-window get_all_embeds_Gt then).  Without imputation the
        temporal all-entity matrix is the base model's (models/BiRRGCN.py:259-293 runs the same isolated trunk + GRUs as
        forward_isolated), so base.all_embeds_batched applies -- zero-state GRU rows once per entity, own rows only for the
        (window, entity) pairs that carry a state; the local one is the isolated trunk Iso2(Iso1(E)) -- the same N rows for every
        window -- with each window's target rows written over it (one static row map)."""
        enc = self.ent_encoder
        if getattr(enc, "impute", False) or not wb.batched or not base._fused_all_entity_ok(self, wb):
            return None
        dev = self._device()
        B, N, D = len(wb.graphs), self.num_ents, self.embed_size
        big_rec = base.all_embeds_batched(self, wb, out, hist)                          # (B, N, D)
        # while the self-loop dropout draws, the local stream too keeps one row per (window, entity): the reference's
        # forward_post_ensemble_isolated runs per window with its own mask (base._all_maps has set wb.all_rep / all_rep_ids)
        rep = bool(getattr(wb, "all_rep", False))
        m = getattr(wb, "_asm_loc", None)
        if m is None or m[2] != rep:
            sizes = [g.n for g in wb.graphs]
            n_out = int(sum(sizes))
            off = np.concatenate([[0], np.cumsum(sizes)])
            tab = np.arange(B * N, dtype=np.int64).reshape(B, N) if rep else np.broadcast_to(np.arange(N, dtype=np.int64)[None, :], (B, N))
            asm = (n_out + tab).copy()
            for b, g in enumerate(wb.graphs):
                asm[b, g.gids] = off[b] + np.arange(g.n)
            asm = asm.reshape(-1)
            m = wb._asm_loc = (_lib.to_device(asm.astype(np.int32), dev), TF.gather_inverse(asm, n_out + (B * N if rep else N), dev), rep)
        E = TF.gather_rows(self.ent_embeds, wb.all_rep_ids, wb.all_rep_inv) if rep else self.ent_embeds
        x = enc.layer_2.conv_isolated(enc.layer_1.conv_isolated(E))                     # local stream of an entity outside the graph
        big_loc = TF.gather_rows(torch.cat([wb.out_loc, x], dim=0), m[0], m[1]).view(B, N, D)
        return big_loc, big_rec

    def batched_ensemble_loss(self, wb, locs, recs, alls, samples, weights):
        """The ensemble loss of ALL windows as one fused node (functional.batched_ensemble_link_prediction), or None when the scorer
        / shapes need the per-window path.  locs / recs: per-window target rows of the two streams; alls: per window (all_loc,
        all_rec), or the pair of (B, N_ents, D) tensors of batched_all_embeds_post as they are (no per-window slices: every slice
        is a zero-filled (B, N_ents, D) gradient and an addition in the backward); weights: per window (weight_subject (P, 1),
        weight_object (P, 1))."""
        name = self.args.score_function
        D = self.embed_size
        if not (self.fused_loss and name in ("distmult", "complex") and self.num_ents % 4 == 0 and D % (8 if name == "complex" else 4) == 0):
            return None
        dev = self._device()
        cache = getattr(wb, "_ens_inputs", None)
        if cache is None or cache[0] is not samples:                 # index tensors are static for a given sample set
            offs = np.concatenate([[0], np.cumsum(wb.target.sizes)])[:-1]
            n_rows = int(sum(wb.target.sizes))
            cache = wb._ens_inputs = (samples, self.loss_inputs([int(o) for o in offs], samples, dev, n_rows, self.rel_embeds.shape[0],
                                                                head_as_tail=self.head_scored_as_tail))
        inp = cache[1]
        if inp is None:
            return torch.cat(locs).sum() * 0.0
        w = torch.cat([torch.cat([wo.reshape(-1, 1), ws.reshape(-1, 1)]) for (ws, wo), smp in zip(weights, samples) if smp[0].shape[0] > 0]).to(dev)
        if isinstance(alls, tuple):
            big_loc, big_rec = alls[0].reshape(-1, D), alls[1].reshape(-1, D)
        else:
            big_loc, big_rec = torch.cat([a for a, _ in alls], dim=0), torch.cat([a for _, a in alls], dim=0)
        return TF.batched_ensemble_link_prediction(torch.cat(locs), torch.cat(recs), self.rel_embeds, big_loc, big_rec, w, name, inp)

    def ensemble_loss(self, loc, rec, all_loc, all_rec, triplets, neg_tail, neg_head, w_subject, w_object):
        """loss_tail + loss_head of one target graph, models/PostDynamicRGCN.py:335-349 + combined_scores :404-406."""
        name = self.args.score_function
        P = triplets.shape[0]
        if self.fused_loss and name in ("distmult", "complex") and all_loc.shape[0] % 4 == 0 and P > 0:
            # both corruption directions stacked into one (2P)-row operand per stream; the two streams share the candidate lists and
            # the per-row mixing weights, so the mix happens on the (2P, N) score matrices (functional._MixedCandidateCEFn)
            from . import scores
            t32 = triplets.to(torch.int32)
            r = TF.gather_rows(self.rel_embeds, t32[:, 1].contiguous())
            head_known = t32[:, 0] if self.head_scored_as_tail else t32[:, 2]          # (the reference's quirk: see head_scored_as_tail)
            head_mode = "tail" if self.head_scored_as_tail else "head"
            idx = torch.cat([t32[:, 0], head_known]).contiguous()
            qs = []
            for rows in (loc, rec):
                known = TF.gather_rows(rows, idx)
                qs.append(torch.cat([scores.bilinear_query(name, known[:P], r, "tail"), scores.bilinear_query(name, known[P:], r, head_mode)], dim=0))
            w = torch.cat([w_object.reshape(-1, 1), w_subject.reshape(-1, 1)], dim=0)
            cand = torch.cat([neg_tail, neg_head], dim=0).to(torch.int32).contiguous()
            return 2.0 * TF.candidate_cross_entropy_mixed(qs[0], all_loc, qs[1], all_rec, w, cand)
        labels = torch.zeros(triplets.shape[0], dtype=torch.int64, device=triplets.device)
        out = 0
        for neg, tail, w in ((neg_tail, True, w_object), (neg_head, False, w_subject)):
            tail = tail or self.head_scored_as_tail
            local = self._scores(loc, triplets, neg, all_loc, tail)
            temporal = self._scores(rec, triplets, neg, all_rec, tail)
            out = out + F.cross_entropy(w * local + (1 - w) * temporal, labels)
        return out


# =====================================================================================================================
# unidirectional
# =====================================================================================================================
class ImputeDynamicRGCN(_PostWindowMixin, DynamicRGCN):
    """models/PostDynamicRGCN.py:20-128."""

    def _can_batch(self):
        enc = self.ent_encoder
        return self.use_batched_path and enc.rec_only_last_layer and isinstance(enc.layer_2, GRRGCNLayer)

    # -- reference-granular path ------------------------------------------------------------------------------------------
    def _encode_step_post(self, st, prev_first, prev_second):
        dev = self._device()
        ids, pidx, dt = st.tensors(dev)
        g = st.batched()
        g.ndata['h'] = TF.gather_rows(self.ent_embeds, ids)
        fp = self._gather_prev(prev_first, pidx, st.n_rows)
        sp = self._gather_prev(prev_second, pidx, st.n_rows)
        return self.ent_encoder.forward_post_ensemble(g, fp, sp, dt, st.times, st.sizes)

    def _run_generic(self, wb):
        loc = first = second = None
        for st in wb.plan.steps:
            loc, first, second = self._encode_step_post(st, first, second)
        out_loc, _, out = self._encode_step_post(wb.target, first, second)
        wb.out_loc, wb.hist_loc = out_loc, loc
        return out, (first, second)

    def _run_batched(self, wb):
        out, hist = super()._run_batched(wb)
        if wb.program is not None:
            wb.out_loc = self._chain_input_rows(wb, wb.out_inst[0])
            wb.hist_loc = self._chain_input_rows(wb, wb.hist_inst) if wb.hist_inst >= 0 else None
        else:
            wb.out_loc = self._chain_input_rows(wb, step=wb.target)
            wb.hist_loc = self._chain_input_rows(wb, step=wb.plan.steps[-1]) if wb.plan.steps else None
        return out, hist

    def _fused_all_entity_ok(self, wb):
        return False                     # the all-entity pass of these models mixes in the local history: per window, below

    def _plan_loss(self, wb):
        wb.loss_plan = None

    # -- all-entity pass ---------------------------------------------------------------------------------------------------
    def _final_prevs(self, plan, b, hist, loc):
        dev, N, D = self._device(), self.num_ents, self.embed_size
        idx, dt = _final_index(plan, b, dev)
        p1 = _rows_or_zero(hist[0], idx, N, D, self.ent_embeds)
        p2 = p1 if hist[1] is hist[0] else _rows_or_zero(hist[1], idx, N, D, self.ent_embeds)
        pl = _rows_or_zero(loc, idx, N, D, self.ent_embeds)
        return p1, p2, pl, dt

    def get_all_embeds_Gt(self, convoluted_embeds, g, t, plan, b, hist, hist_loc=None):
        """ImputeDynamicRGCN.get_all_embeds_Gt, models/PostDynamicRGCN.py:24-31."""
        p1, p2, pl, dt = self._final_prevs(plan, b, hist, hist_loc)
        all_embeds = self.ent_encoder.forward_isolated_impute(self.ent_embeds, p1, p2, dt, t, pl)
        return all_embeds.index_copy(0, _gids_dev(g, self._device()), convoluted_embeds)

    def run_loss(self, wb, samples=None):
        """ImputeDynamicRGCN.forward, models/PostDynamicRGCN.py:80-96."""
        dev = self._device()
        out, hist = self.run(wb)
        per_graph = list(out.split(wb.target.sizes))
        if samples is None:
            samples = self.draw_samples(wb)
        loss = 0
        for i, (g, ent_embed) in enumerate(zip(wb.graphs, per_graph)):
            t = wb.rows[i][-1]
            triplets, neg_tail, neg_head = (x.to(dev) for x in samples[i])
            labels = torch.zeros(triplets.shape[0], dtype=torch.int64, device=dev)
            all_embeds_g = self.get_all_embeds_Gt(ent_embed, g, t, wb.plan, i, hist, wb.hist_loc)
            loss = loss + self.train_link_prediction_both(ent_embed, triplets, neg_tail, neg_head, labels, all_embeds_g)
        return loss

    def encode_post(self, t_list, seq_len, train=True, target_edge_ids=None):
        """-> (per-window local embeddings, per-window temporal embeddings, prepared batch, final histories)."""
        wb = self.prepare(t_list, seq_len, train, target_edge_ids)
        out, hist = self.run(wb)
        return list(wb.out_loc.split(wb.target.sizes)), list(out.split(wb.target.sizes)), wb, hist

    def evaluate(self, t_list, val=True):
        """ImputeDynamicRGCN.evaluate / calc_metrics, models/PostDynamicRGCN.py:101-143 (bidirectional:
        models/PostBiDynamicRGCN.py:126-176): the window loop with the local history stream on the full train graphs, the
        IMPUTED all-entity matrix, then the standard filtered ranks (temp_amd.evaluation.EvaluationFilter) and classification
        loss of the valid (or test) triples of every target timestamp."""
        return _impute_evaluate(self, t_list, val)


class PostEnsembleDynamicRGCN(ImputeDynamicRGCN):
    """models/PostDynamicRGCN.py:323-461 (PostEnsembleDynamicRGCN: score-level ensemble with the frequency MLPs)."""

    def __init__(self, args, num_ents, num_rels, graph_dict_train, graph_dict_val, graph_dict_test, evaluater_type=None):
        super().__init__(args, num_ents, num_rels, graph_dict_train, graph_dict_val, graph_dict_test, evaluater_type)
        self.init_freq_mlp()

    def evaluate(self, t_list, val=True):
        """PostEnsemble(Bi)DynamicRGCN.evaluate / calc_metrics (models/PostDynamicRGCN.py:367-423, models/PostBiDynamicRGCN.py:297-360):
        window loop with the local stream on the full train graphs, (local, temporal) all-entity matrices, score-level ensemble
        ranks (PostEnsembleEvaluationFilter).  The mixing weights come from calc_ensemble_ratio(index_sample, t, g) (frequency MLPs).
        As in the reference no classification loss is computed (nan)."""
        return _post_ensemble_evaluate(self, t_list, val)

    def get_all_embeds_Gt(self, convoluted_loc, convoluted_rec, g, t, plan, b, hist, hist_loc=None):
        """PostDynamicRGCN.get_all_embeds_Gt, models/PostDynamicRGCN.py:160-174 -> (all_loc, all_rec)."""
        p1, p2, pl, dt = self._final_prevs(plan, b, hist, hist_loc)
        a_loc, a_rec = self.ent_encoder.forward_post_ensemble_isolated(self.ent_embeds, p1, p2, dt, t, pl)
        gid = _gids_dev(g, self._device())
        return a_loc.index_copy(0, gid, convoluted_loc), a_rec.index_copy(0, gid, convoluted_rec)

    def run_loss(self, wb, samples=None, ensemble_weights=None):
        """PostEnsembleDynamicRGCN.forward, models/PostDynamicRGCN.py:375-397."""
        dev = self._device()
        out, hist = self.run(wb)
        recs, locs = list(out.split(wb.target.sizes)), list(wb.out_loc.split(wb.target.sizes))
        if samples is None:
            samples = self.draw_samples(wb)
        loss = 0
        wts = [ensemble_weights[i] if ensemble_weights is not None else self.calc_ensemble_ratio(samples[i][0].to(dev), wb.rows[i][-1], g)
               for i, g in enumerate(wb.graphs)]
        both = self.batched_all_embeds_post(wb, out, hist, DynamicRGCN)
        if both is not None:
            fused = self.batched_ensemble_loss(wb, locs, recs, both, samples, wts)      # all windows' losses as one node, on the (B, N, D) tensors
            if fused is not None:
                return fused
        alls = [(both[0][i], both[1][i]) if both is not None else self.get_all_embeds_Gt(locs[i], recs[i], g, wb.rows[i][-1], wb.plan, i, hist, wb.hist_loc)
                for i, g in enumerate(wb.graphs)]
        if both is None:
            fused = self.batched_ensemble_loss(wb, locs, recs, alls, samples, wts)
            if fused is not None:
                return fused
        for i, g in enumerate(wb.graphs):
            triplets, neg_tail, neg_head = (x.to(dev) for x in samples[i])
            (a_loc, a_rec), (ws, wo) = alls[i], wts[i]
            loss = loss + self.ensemble_loss(locs[i], recs[i], a_loc, a_rec, triplets, neg_tail, neg_head, ws.to(dev), wo.to(dev))
        return loss

    def forward(self, t_list, reverse=False, target_edge_ids=None, samples=None, ensemble_weights=None):
        wb = self.prepare(t_list, self.train_seq_len, True, target_edge_ids)
        return self.run_loss(wb, samples, ensemble_weights)


# =====================================================================================================================
# bidirectional
# =====================================================================================================================
class ImputeBiDynamicRGCN(_PostWindowMixin, BiDynamicRGCN):
    """models/PostBiDynamicRGCN.py:22-167."""

    def _can_batch(self):
        enc = self.ent_encoder
        return self.use_batched_path and enc.rec_only_last_layer and isinstance(enc.layer_2, BiGRRGCNLayer)

    # -- reference-granular path ------------------------------------------------------------------------------------------
    def _pre_forward_post(self, plan, forward):
        dev = self._device()
        loc = first = second = None
        for st in plan.steps:
            ids, pidx, dt = st.tensors(dev)
            g = st.batched()
            g.ndata['h'] = TF.gather_rows(self.ent_embeds, ids)
            fp = self._gather_prev(first, pidx, st.n_rows)
            sp = self._gather_prev(second, pidx, st.n_rows)
            loc, first, second = self.ent_encoder.forward_post_ensemble_one_direction(g, fp, sp, dt, st.times, st.sizes, forward)
        return loc, (first, second)

    def _run_generic(self, wb):
        dev = self._device()
        plan_f, plan_b = wb.plan
        tf, tb = wb.target, wb.target_b
        loc_f, hf = self._pre_forward_post(plan_f, True)
        loc_b, hb = self._pre_forward_post(plan_b, False)
        ids, pf, dtf = tf.tensors(dev)
        _, pb, dtb = tb.tensors(dev)
        g = tf.batched()
        g.ndata['h'] = TF.gather_rows(self.ent_embeds, ids)
        n = tf.n_rows
        out_loc, out = self.ent_encoder.forward_post_ensemble(
            g, self._gather_prev(hf[0], pf, n), self._gather_prev(hf[1], pf, n), dtf,
            self._gather_prev(hb[0], pb, n), self._gather_prev(hb[1], pb, n), dtb, tf.times, tf.sizes)
        wb.out_loc, wb.hist_loc = out_loc, (loc_f, loc_b)
        return out, (hf, hb)

    def _run_batched(self, wb):
        out, hist = super()._run_batched(wb)
        plan_f, plan_b = wb.plan
        if wb.program is not None:
            wb.out_loc = self._chain_input_rows(wb, wb.out_inst[0])
            wb.hist_loc = tuple(self._chain_input_rows(wb, i) if i >= 0 else None for i in wb.hist_inst)
        else:
            wb.out_loc = self._chain_input_rows(wb, step=wb.target)
            wb.hist_loc = tuple(self._chain_input_rows(wb, step=p.steps[-1]) if p.steps else None for p in (plan_f, plan_b))
        return out, hist

    def _fused_all_entity_ok(self, wb):
        return False

    def _plan_loss(self, wb):
        wb.loss_plan = None

    # -- all-entity pass ---------------------------------------------------------------------------------------------------
    def _final_prevs(self, plans, b, hist, hist_loc):
        dev, N, D = self._device(), self.num_ents, self.embed_size
        out = []
        for plan, h, loc in zip(plans, hist, hist_loc if hist_loc is not None else (None, None)):
            idx, dt = _final_index(plan, b, dev)
            p1 = _rows_or_zero(h[0], idx, N, D, self.ent_embeds)
            p2 = p1 if h[1] is h[0] else _rows_or_zero(h[1], idx, N, D, self.ent_embeds)
            out.append((p1, p2, _rows_or_zero(loc, idx, N, D, self.ent_embeds), dt))
        return out

    def get_all_embeds_Gt(self, convoluted_embeds, g, t, plans, b, hist, hist_loc=None):
        """ImputeBiDynamicRGCN.get_all_embeds_Gt, models/PostBiDynamicRGCN.py:29-39."""
        (f1, f2, fl, dtf), (b1, b2, bl, dtb) = self._final_prevs(plans, b, hist, hist_loc)
        all_embeds = self.ent_encoder.forward_isolated_impute(self.ent_embeds, f1, f2, dtf, b1, b2, dtb, t, fl, bl)
        return all_embeds.index_copy(0, _gids_dev(g, self._device()), convoluted_embeds)

    def run_loss(self, wb, samples=None):
        """ImputeBiDynamicRGCN.forward, models/PostBiDynamicRGCN.py:103-124."""
        dev = self._device()
        out, hist = self.run(wb)
        per_graph = list(out.split(wb.target.sizes))
        if samples is None:
            samples = self.draw_samples(wb)
        loss = 0
        for i, (g, ent_embed) in enumerate(zip(wb.graphs, per_graph)):
            t = wb.rows[i][-1]
            triplets, neg_tail, neg_head = (x.to(dev) for x in samples[i])
            labels = torch.zeros(triplets.shape[0], dtype=torch.int64, device=dev)
            all_embeds_g = self.get_all_embeds_Gt(ent_embed, g, t, wb.plan, i, hist, wb.hist_loc)
            loss = loss + self.train_link_prediction_both(ent_embed, triplets, neg_tail, neg_head, labels, all_embeds_g)
        return loss

    def encode_post(self, t_list, seq_len, train=True, target_edge_ids=None):
        """-> (per-window local embeddings, per-window temporal embeddings, prepared batch, final histories)."""
        wb = self.prepare(t_list, seq_len, train, target_edge_ids)
        out, hist = self.run(wb)
        return list(wb.out_loc.split(wb.target.sizes)), list(out.split(wb.target.sizes)), wb, hist

    def evaluate(self, t_list, val=True):
        """ImputeDynamicRGCN.evaluate / calc_metrics, models/PostDynamicRGCN.py:101-143 (bidirectional:
        models/PostBiDynamicRGCN.py:126-176): the window loop with the local history stream on the full train graphs, the
        IMPUTED all-entity matrix, then the standard filtered ranks (temp_amd.evaluation.EvaluationFilter) and classification
        loss of the valid (or test) triples of every target timestamp."""
        return _impute_evaluate(self, t_list, val)


class PostEnsembleBiDynamicRGCN(ImputeBiDynamicRGCN):
    """models/PostBiDynamicRGCN.py:283-372 (score-level ensemble with the frequency MLPs) -- BASELINE config 3's model."""
    head_scored_as_tail = True            # models/PostBiDynamicRGCN.py:294-295, see _PostWindowMixin

    def __init__(self, args, num_ents, num_rels, graph_dict_train, graph_dict_val, graph_dict_test, evaluater_type=None):
        super().__init__(args, num_ents, num_rels, graph_dict_train, graph_dict_val, graph_dict_test, evaluater_type)
        self.init_freq_mlp()

    def evaluate(self, t_list, val=True):
        """PostEnsemble(Bi)DynamicRGCN.evaluate / calc_metrics (models/PostDynamicRGCN.py:367-423, models/PostBiDynamicRGCN.py:297-360):
        window loop with the local stream on the full train graphs, (local, temporal) all-entity matrices, score-level ensemble
        ranks (PostEnsembleEvaluationFilter).  The mixing weights come from calc_ensemble_ratio(index_sample, t, g) (frequency MLPs).
        As in the reference no classification loss is computed (nan)."""
        return _post_ensemble_evaluate(self, t_list, val)

    def get_all_embeds_Gt(self, convoluted_loc, convoluted_rec, g, t, plans, b, hist, hist_loc=None):
        """PostBiDynamicRGCN.get_all_embeds_Gt, models/PostBiDynamicRGCN.py:176-190 -> (all_loc, all_rec)."""
        (f1, f2, fl, dtf), (b1, b2, bl, dtb) = self._final_prevs(plans, b, hist, hist_loc)
        a_loc, a_rec = self.ent_encoder.forward_post_ensemble_isolated(self.ent_embeds, f1, f2, dtf, b1, b2, dtb, t, fl, bl)
        gid = _gids_dev(g, self._device())
        return a_loc.index_copy(0, gid, convoluted_loc), a_rec.index_copy(0, gid, convoluted_rec)

    def run_loss(self, wb, samples=None, ensemble_weights=None):
        """PostEnsembleBiDynamicRGCN.forward, models/PostBiDynamicRGCN.py:329-354."""
        dev = self._device()
        out, hist = self.run(wb)
        recs, locs = list(out.split(wb.target.sizes)), list(wb.out_loc.split(wb.target.sizes))
        if samples is None:
            samples = self.draw_samples(wb)
        loss = 0
        wts = [ensemble_weights[i] if ensemble_weights is not None else self.calc_ensemble_ratio(samples[i][0].to(dev), wb.rows[i][-1], g)
               for i, g in enumerate(wb.graphs)]
        both = self.batched_all_embeds_post(wb, out, hist, BiDynamicRGCN)
        if both is not None:
            fused = self.batched_ensemble_loss(wb, locs, recs, both, samples, wts)      # all windows' losses as one node, on the (B, N, D) tensors
            if fused is not None:
                return fused
        alls = [(both[0][i], both[1][i]) if both is not None else self.get_all_embeds_Gt(locs[i], recs[i], g, wb.rows[i][-1], wb.plan, i, hist, wb.hist_loc)
                for i, g in enumerate(wb.graphs)]
        if both is None:
            fused = self.batched_ensemble_loss(wb, locs, recs, alls, samples, wts)
            if fused is not None:
                return fused
        for i, g in enumerate(wb.graphs):
            triplets, neg_tail, neg_head = (x.to(dev) for x in samples[i])
            (a_loc, a_rec), (ws, wo) = alls[i], wts[i]
            loss = loss + self.ensemble_loss(locs[i], recs[i], a_loc, a_rec, triplets, neg_tail, neg_head, ws.to(dev), wo.to(dev))
        return loss

    def forward(self, t_list, reverse=False, target_edge_ids=None, samples=None, ensemble_weights=None):
        wb = self.prepare(t_list, self.train_seq_len, True, target_edge_ids)
        return self.run_loss(wb, samples, ensemble_weights)
