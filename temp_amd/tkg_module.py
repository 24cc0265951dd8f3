"""TKG_Module -- the task base class with the reference's constructor / method surface
(models/TKG_Module.py:20-274) minus the pytorch_lightning plumbing (Lightning 0.5.2 is a training
harness, not part of the hot path; the hooks a harness calls -- training_step, validation_step,
configure_optimizers -- are kept and need no Lightning import).
"""
import threading

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import scores
from .sampling import CorruptTriples


class TKG_Module(nn.Module):
    def __init__(self, args, num_ents, num_rels, graph_dict_train, graph_dict_val, graph_dict_test, evaluater_type=None):
        super().__init__()
        self.args = self.hparams = args
        self.graph_dict_train = graph_dict_train
        self.graph_dict_val = graph_dict_val
        self.graph_dict_test = graph_dict_test
        self.total_time = np.array(list(graph_dict_train.keys()))
        self.num_rels = num_rels
        self.num_ents = num_ents
        self.embed_size = args.embed_size
        self.hidden_size = args.hidden_size
        self.use_cuda = getattr(args, "use_cuda", False)
        self.num_pos_facts = args.num_pos_facts
        self.negative_rate = args.negative_rate
        self.calc_score = {'distmult': scores.distmult, 'complex': scores.complex, 'transE': scores.transE}[args.score_function]
        self.fused_loss = True
        self.build_model()
        if not getattr(args, "debug", False):
            self.corrupter = CorruptTriples(self.args, graph_dict_train)
            if evaluater_type is not None:
                self.evaluater = evaluater_type(args, self.calc_score, graph_dict_train, graph_dict_val, graph_dict_test)

    # `sample_rng` draws the training-time edge / positive subsets inside `prepare`.  A prefetch worker that prepares batches
    # out of order (temp_amd.prefetch.BatchPrefetcher with several workers) installs a generator of its own for the batch it
    # is working on -- seeded, in batch order, from the model's -- so that a run does not depend on the workers' timing.
    _rng_override = threading.local()

    @property
    def sample_rng(self):
        o = getattr(TKG_Module._rng_override, "rng", None)
        return o if o is not None else self._sample_rng

    @sample_rng.setter
    def sample_rng(self, rng):
        self._sample_rng = rng

    def build_model(self):
        raise NotImplementedError

    # -- harness hooks (models/TKG_Module.py:43-160) ------------------------------------------------
    def training_step(self, batch_time, batch_idx=0):
        loss = self.forward(batch_time)
        return {'loss': loss, 'progress_bar': {'train_loss': loss}, 'log': {'train_loss': loss}}

    def validation_step(self, batch_time, batch_idx=0):
        ranks, loss = self.evaluate(batch_time)
        return {'ranks': ranks, 'val_loss': loss}

    def get_metrics(self, ranks):
        r = ranks.float()
        return torch.mean(1.0 / r), torch.mean((ranks <= 1).float()), torch.mean((ranks <= 3).float()), torch.mean((ranks <= 10).float())

    def configure_optimizers(self):
        """models/TKG_Module.py:154-160: Adam(lr, weight_decay = 1e-4).  On the GPU torch's fused implementation of the same
        update (one launch over all parameters instead of ten multi-tensor ones: 0.6 ms less host time and 0.3 ms less device
        time per S-gdelt training step, tools/adam_probe.py)."""
        params = list(self.parameters())
        fused = bool(params) and all(p.is_cuda and p.dtype == torch.float32 for p in params)
        return torch.optim.Adam(params, lr=self.args.lr, weight_decay=0.0001, fused=fused)

    # -- loss (models/TKG_Module.py:202-221) ----------------------------------------------------------
    def train_link_prediction(self, ent_embed, triplets, neg_samples, labels, all_embeds_g, corrupt_tail=True):
        """models/TKG_Module.py:202-213.  DistMult / ComplEx take the fused path: ONE GEMM of the
        folded query against all entities + a candidate cross-entropy kernel, instead of gathering
        a (P, 1+neg, D) tensor; other scorers use the tensor-algebra path."""
        name = self.args.score_function
        if self.fused_loss and name in ("distmult", "complex") and all_embeds_g.shape[0] % 4 == 0 and triplets.shape[0] > 0:
            from . import functional as TF
            t32 = triplets.to(torch.int32)
            # row gathers through the HIP kernel: its backward is one atomic scatter-add instead of
            # torch's sort-based index_put (the single largest cost of the loss path otherwise)
            r = TF.gather_rows(self.rel_embeds, t32[:, 1].contiguous())
            known = TF.gather_rows(ent_embed, (t32[:, 0] if corrupt_tail else t32[:, 2]).contiguous())
            q = scores.bilinear_query(name, known, r, "tail" if corrupt_tail else "head")
            return TF.candidate_cross_entropy(q.contiguous(), all_embeds_g.contiguous(), neg_samples.to(torch.int32).contiguous())
        r = self.rel_embeds[triplets[:, 1]]
        if corrupt_tail:
            score = self.calc_score(ent_embed[triplets[:, 0]], r, all_embeds_g[neg_samples], mode='tail')
        else:
            score = self.calc_score(all_embeds_g[neg_samples], r, ent_embed[triplets[:, 2]], mode='head')
        return F.cross_entropy(score, labels)

    def train_link_prediction_both(self, ent_embed, triplets, neg_tail, neg_head, labels, all_embeds_g):
        """loss_tail + loss_head of one target graph (models/DynamicRGCN.py:189-192).  On the fused path the
        two directions share the score GEMM against all entities: their queries are stacked into one (2P, D)
        operand (both halves have P rows, so the sum of the two means is twice the mean over the stack)."""
        name = self.args.score_function
        P = triplets.shape[0]
        if self.fused_loss and name in ("distmult", "complex") and all_embeds_g.shape[0] % 4 == 0 and P > 0:
            from . import functional as TF
            t32 = triplets.to(torch.int32)
            r = TF.gather_rows(self.rel_embeds, t32[:, 1].contiguous())
            known = TF.gather_rows(ent_embed, torch.cat([t32[:, 0], t32[:, 2]]).contiguous())
            q = torch.cat([scores.bilinear_query(name, known[:P], r, "tail"), scores.bilinear_query(name, known[P:], r, "head")], dim=0)
            cand = torch.cat([neg_tail, neg_head], dim=0).to(torch.int32).contiguous()
            return 2.0 * TF.candidate_cross_entropy(q.contiguous(), all_embeds_g.contiguous(), cand)
        return (self.train_link_prediction(ent_embed, triplets, neg_tail, labels, all_embeds_g, corrupt_tail=True)
                + self.train_link_prediction(ent_embed, triplets, neg_head, labels, all_embeds_g, corrupt_tail=False))

    @staticmethod
    def loss_inputs(row_offsets, samples, dev, n_rows=None, n_rel_rows=None, head_as_tail=False):
        """Index tensors of the batched loss for one set of samples (static for a prepared batch, so callers cache it):
        per graph the stacked operand is [tail queries (P rows); head queries (P rows)].  `n_rows` / `n_rel_rows` (the row
        counts of the target-embedding stack and of rel_embeds) add the inverse maps the deterministic backward reduces over."""
        known, rel, tail, cand, splits, weights = [], [], [], [], [], []
        row = 0
        for b, (trip, neg_tail, neg_head) in enumerate(samples):
            P = trip.shape[0]
            if P == 0:
                splits.append((row, row))
                continue
            t = trip.to(dev)
            # head_as_tail: the reference's PostEnsembleBiDynamicRGCN scores the head-corruption candidates as tails of the true
            # subject (models/PostBiDynamicRGCN.py:294-295; post_dynamic_rgcn._PostWindowMixin.head_scored_as_tail)
            known.append(torch.cat([t[:, 0], t[:, 0] if head_as_tail else t[:, 2]]) + row_offsets[b])
            rel.append(torch.cat([t[:, 1], t[:, 1]]))
            tail.append(torch.cat([torch.ones(P, dtype=torch.int32, device=dev),
                                   (torch.ones if head_as_tail else torch.zeros)(P, dtype=torch.int32, device=dev)]))
            cand.append(neg_tail.to(dev)); cand.append(neg_head.to(dev))
            splits.append((row, row + 2 * P))
            weights.append(torch.full((2 * P,), 1.0 / P, dtype=torch.float32, device=dev))
            row += 2 * P
        if row == 0:
            return None
        out = dict(known=torch.cat(known).to(torch.int32).contiguous(), rel=torch.cat(rel).to(torch.int32).contiguous(),
                   is_tail=torch.cat(tail).contiguous(), cand=torch.cat(cand, dim=0).to(torch.int32).contiguous(), splits=splits,
                   weights=torch.cat(weights))
        if n_rows is not None:
            from . import functional as TF
            out["known_inv"] = TF.gather_inverse(out["known"].cpu().numpy(), n_rows, dev)
            out["rel_inv"] = TF.gather_inverse(out["rel"].cpu().numpy(), n_rel_rows, dev)
        return out

    def batched_link_prediction(self, ent_rows, inputs, all_embeds):
        """Sum over the target graphs of loss_tail + loss_head (models/DynamicRGCN.py:186-193) as one fused node
        (functional.batched_link_prediction): `ent_rows` is the concatenation of the per-graph target embeddings,
        `inputs` = loss_inputs(..., n_rows, n_rel_rows), `all_embeds` the (B * N_ents, D) stack of the windows' all-entity
        matrices (or a list of B (N_ents, D) matrices).  Returns None when the scorer is not bilinear or the shapes are
        outside the kernels' alignment (the caller takes the per-graph path)."""
        name = self.args.score_function
        D = ent_rows.shape[1]
        if not (self.fused_loss and name in ("distmult", "complex") and self.num_ents % 4 == 0 and D % (8 if name == "complex" else 4) == 0):
            return None
        if inputs is None:
            return ent_rows.sum() * 0.0
        from . import functional as TF
        big = all_embeds if torch.is_tensor(all_embeds) else torch.cat(list(all_embeds), dim=0)
        return TF.batched_link_prediction(ent_rows, self.rel_embeds, big.reshape(-1, D).contiguous(), name, inputs)

    def link_classification_loss(self, ent_embed, rel_embeds, triplets, labels):
        score = self.calc_score(ent_embed[triplets[:, 0]], rel_embeds[triplets[:, 1]], ent_embed[triplets[:, 2]])
        return F.binary_cross_entropy_with_logits(score, labels)

    # -- window construction (models/TKG_Module.py:232-250) --------------------------------------------
    def get_batch_graph_list(self, t_list, seq_len, graph_dict):
        from .window import window_times
        rows = window_times(t_list, seq_len, list(graph_dict.keys()))
        t_batched = [list(x) for x in zip(*rows)]
        g_batched = [[graph_dict[t] if t is not None else None for t in col] for col in t_batched]
        return g_batched, t_batched
