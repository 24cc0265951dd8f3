"""StaticRGCN -- the non-recurrent baseline (BASELINE config 1) with the reference's interface
(baselines/StaticRGCN.py:10-113, baselines/TKG_Non_Recurrent.py): per-timestamp 2-layer RGCN on a
50 % edge subsample of each target snapshot, no history."""
import numpy as np
import torch
import torch.nn as nn

from . import _hostlib
from . import _lib
from . import functional as TF
from . import snapshot as S
from .rgcn import RGCN
from .tkg_module import TKG_Module


class StaticRGCN(TKG_Module):
    def __init__(self, args, num_ents, num_rels, graph_dict_train, graph_dict_val, graph_dict_test, evaluater_type=None):
        super().__init__(args, num_ents, num_rels, graph_dict_train, graph_dict_val, graph_dict_test, evaluater_type)
        self.ent_embeds = nn.Parameter(torch.Tensor(self.num_ents, self.embed_size))
        self.rel_embeds = nn.Parameter(torch.Tensor(self.num_rels * 2, self.embed_size))
        nn.init.xavier_uniform_(self.ent_embeds, gain=nn.init.calculate_gain('relu'))
        nn.init.xavier_uniform_(self.rel_embeds, gain=nn.init.calculate_gain('relu'))
        self.sample_rng = np.random.default_rng(getattr(args, "seed", None))
        self.seed_rng = np.random.default_rng(None if getattr(args, "seed", None) is None else int(args.seed) + 1)
        self.use_device_sampler = True

    def build_model(self):
        self.train_seq_len = self.args.train_seq_len
        self.test_seq_len = self.args.train_seq_len
        self.ent_encoder = RGCN(self.args, self.hidden_size, self.embed_size, self.num_rels, self.total_time)

    def get_per_graph_ent_embeds(self, t_list, graph_train_list, val=False, edge_ids=None):
        """baselines/StaticRGCN.py:60-89."""
        if val:
            graphs = graph_train_list
        else:
            graphs = []
            for i, g in enumerate(graph_train_list):
                E = g.number_of_edges()
                idx = edge_ids[i] if edge_ids is not None else _hostlib.sample_subset(E, int(0.5 * E), self.sample_rng)
                graphs.append(g.edge_subgraph(idx))
        bg = S.batch(graphs)
        ids = torch.from_numpy(bg.gids.astype(np.int32)).to(self.ent_embeds.device)
        # (static ids: the gather's adjoint is a deterministic segment sum, not an atomic scatter -- entities repeat across graphs)
        bg.ndata['h'] = TF.gather_rows(self.ent_embeds, ids, TF.gather_inverse(bg.gids, self.num_ents, self.ent_embeds.device))
        sizes = [g.n for g in graph_train_list]
        out = self.ent_encoder(bg, [int(t) for t in t_list], sizes)
        self._last_rows = out.ndata['h']                      # the unsplit (sum n_b, D) rows (the fused loss consumes them whole)
        return out.ndata['h'].split(sizes)

    def get_all_embeds_Gt(self, t, g, convoluted_embeds):
        """baselines/StaticRGCN.py:48-58."""
        if getattr(self.args, "use_embed_for_non_active", False):
            all_embeds = self.ent_embeds
        else:
            all_embeds = self.ent_encoder.forward_isolated(self.ent_embeds, int(t))
        gid = torch.from_numpy(g.gids).to(self.ent_embeds.device)
        return all_embeds.index_copy(0, gid, convoluted_embeds)

    def evaluate(self, t_list, val=True):
        """baselines/StaticRGCN.py:20-34,91-113: full train graphs -> embeddings, filtered ranks of the
        valid/test triples + classification loss."""
        from .evaluation import EvaluationFilter
        if not hasattr(self, "evaluater"):
            self.evaluater = EvaluationFilter(self.args, self.calc_score, self.graph_dict_train, self.graph_dict_val, self.graph_dict_test)
        graph_dict = self.graph_dict_val if val else self.graph_dict_test
        dev = self.ent_embeds.device
        ts = [int(t) for t in t_list]
        with torch.no_grad():
            per_graph = self.get_per_graph_ent_embeds(ts, [self.graph_dict_train[t] for t in ts], val=True)
            ranks, losses = [], []
            for t, ent_embed in zip(ts, per_graph):
                g = graph_dict[t]
                if g.number_of_edges() == 0:
                    continue
                all_embeds_g = self.get_all_embeds_Gt(t, g, ent_embed)
                index_sample = torch.from_numpy(np.stack([g.src, g.rel, g.dst], axis=1)).to(dev)
                label = torch.ones(index_sample.shape[0], device=dev)
                ranks.append(self.evaluater.calc_metrics_single_graph(ent_embed, self.rel_embeds, all_embeds_g, index_sample, g, t))
                losses.append(self.link_classification_loss(ent_embed, self.rel_embeds, index_sample, label).item())
        ranks = torch.cat(ranks) if ranks else torch.zeros(0, dtype=torch.int64, device=dev)
        return ranks, (float(np.mean(losses)) if losses else float("nan"))

    def _fused_loss_ok(self):
        return (self.use_device_sampler and self.fused_loss and self.args.score_function in ("distmult", "complex")
                and not self.ent_encoder.use_time_embedding and not getattr(self.args, "use_embed_for_non_active", False)
                and self.num_ents % 4 == 0)

    def _fused_plan(self, ts, g_list):
        """Host half of the fused loss of a batch (static for the batch): the row map that assembles every window's all-entity
        matrix from [its graph's rows ; the isolated table], positives / operand indices / known-true slices."""
        from .sampling import TrueSetStore, plan_batch_loss
        dev = self.ent_embeds.device
        N, B = self.num_ents, len(g_list)
        sizes = [g.n for g in g_list]
        n_out = int(sum(sizes))
        off_out = np.concatenate([[0], np.cumsum(sizes)])
        asm = np.broadcast_to(n_out + np.arange(N, dtype=np.int64)[None, :], (B, N)).copy()
        for b, g in enumerate(g_list):
            asm[b, g.gids] = off_out[b] + np.arange(g.n)
        store = getattr(self, "_true_store", None)
        if store is None or store.device != dev:
            store = self._true_store = TrueSetStore(self.graph_dict_train, N, dev)
        plan = plan_batch_loss(store, ts, g_list, off_out[:-1], self.args.num_pos_facts, self.sample_rng, n_out, int(self.rel_embeds.shape[0]), dev)
        return dict(plan=plan, asm=_lib.to_device(asm.reshape(-1).astype(np.int32), dev), asm_inv=TF.gather_inverse(asm.reshape(-1), n_out + N, dev),
                    B=B, t0=ts[0])

    def _fused_loss(self, fp, out, cand=None):
        """All windows' losses as ONE fused node (functional.batched_link_prediction), like the recurrent models: the isolated
        pass RGCN.forward_isolated(ent_embeds) does not depend on the window (no time embedding), so it runs once; every
        window's all-entity matrix is [its graph's rows ; the table] through one static row map; the negatives of all graphs
        come from one launch (`cand`: a fixed draw to reuse, else a fresh one per call)."""
        from .backend import get_backend
        plan = fp["plan"]
        if plan is None:
            return out.sum() * 0.0
        N = self.num_ents
        table = self.ent_encoder.forward_isolated(self.ent_embeds, fp["t0"])
        big = TF.gather_rows(torch.cat([out, table], dim=0), fp["asm"], fp["asm_inv"])
        if cand is None:
            cand = get_backend().corrupt_sample(int(self.seed_rng.integers(1 << 62)), plan["truth"], plan["lo"], plan["hi"], plan["ids"],
                                                self.args.negative_rate, N)
        self._last_plan = (plan, cand)
        return self.batched_link_prediction(out, dict(plan, cand=cand), big.view(fp["B"], N, big.shape[1]))

    def _fused_forward(self, ts, g_list, out):
        return self._fused_loss(self._fused_plan(ts, g_list), out)

    # -- prepare / run split (what the recurrent models have: everything that depends only on the batch is planned and uploaded once,
    #    the device work of a step can then be replayed -- bench.py captures it as a HIP graph) --------------------------------------
    def prepare(self, t_list, target_edge_ids=None):
        """-> a prepared batch: the 50 % edge subsamples (baselines/StaticRGCN.py:60-89), their union graph on the device, the
        entity ids, and -- when the fused loss applies -- its host plan."""
        dev = self.ent_embeds.device
        wb = type("StaticBatch", (), {})()
        wb.ts = [int(t) for t in t_list]
        wb.g_list = [self.graph_dict_train[t] for t in wb.ts]
        graphs = []
        for i, g in enumerate(wb.g_list):
            E = g.number_of_edges()
            idx = target_edge_ids[i] if target_edge_ids is not None else _hostlib.sample_subset(E, int(0.5 * E), self.sample_rng)
            graphs.append(g.edge_subgraph(idx))
        wb.bg = S.batch(graphs)
        wb.bg.device_graph(dev, 2 * self.num_rels)
        wb.ids = torch.from_numpy(wb.bg.gids.astype(np.int32)).to(dev)
        wb.ids_inv = TF.gather_inverse(wb.bg.gids, self.num_ents, dev)      # deterministic adjoint of the embedding gather
        wb.sizes = [g.n for g in wb.g_list]
        wb.n_edge_visits = int(sum(g.number_of_edges() for g in graphs))
        wb.fused = self._fused_plan(wb.ts, wb.g_list) if self._fused_loss_ok() else None
        return wb

    def run(self, wb):
        """Device work of the encoder on a prepared batch -> the (sum n_b, D) target rows."""
        wb.bg.ndata['h'] = TF.gather_rows(self.ent_embeds, wb.ids, wb.ids_inv)
        out = self.ent_encoder(wb.bg, wb.ts, wb.sizes)
        self._last_rows = out.ndata['h']
        return out.ndata['h']

    def run_loss(self, wb, cand=None):
        """Encoder + fused loss on a prepared batch (cand: fixed negatives, e.g. the draw of an earlier call: self._last_plan[1])."""
        rows = self.run(wb)
        if wb.fused is None:
            raise NotImplementedError("run_loss needs the fused loss (bilinear scorer, no time embedding); use forward()")
        return self._fused_loss(wb.fused, rows, cand)

    def forward(self, t_list, target_edge_ids=None, samples=None):
        """baselines/StaticRGCN.py:36-46."""
        dev = self.ent_embeds.device
        ts = [int(t) for t in t_list]
        g_list = [self.graph_dict_train[t] for t in ts]
        per_graph = self.get_per_graph_ent_embeds(ts, g_list, edge_ids=target_edge_ids)
        if samples is None and self._fused_loss_ok():
            fused = self._fused_forward(ts, g_list, self._last_rows)
            if fused is not None:
                return fused
        loss = 0
        for i, (t, g, ent_embed) in enumerate(zip(ts, g_list, per_graph)):
            if samples is not None:
                triplets, neg_tail, neg_head = samples[i]
                labels = torch.zeros(triplets.shape[0], dtype=torch.int64)
            else:
                triplets, neg_tail, neg_head, labels = self.corrupter.single_graph_negative_sampling(t, g, self.num_ents)
            triplets, neg_tail, neg_head, labels = triplets.to(dev), neg_tail.to(dev), neg_head.to(dev), labels.to(dev)
            all_embeds_g = self.get_all_embeds_Gt(t, g, ent_embed)
            loss = loss + self.train_link_prediction_both(ent_embed, triplets, neg_tail, neg_head, labels, all_embeds_g)
        return loss
