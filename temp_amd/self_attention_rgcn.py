"""SelfAttentionRGCN / BiSelfAttentionRGCN -- the attention window models with the reference's
interface (models/SelfAttentionRGCN.py:13-180, models/BiSelfAttentionRGCN.py:10-95): same
constructor, `.forward(t_list) -> loss`, `.evaluate(t_list)`, same parameter names.

Reference data flow: every history position of every window runs the 2-layer RGCN again and writes
its states into a dense (L-1, bsz, 2, N_ents, D) tensor (+ a (L, bsz, N_ents) additive mask); the
target pass and the all-entity pass then attend over dense (n, T, D) gathers of it.

Here none of the history visits is recurrent, so the whole step is
  1. ONE 2-layer RGCN pass over the union of the DISTINCT snapshots the windows touch (history
     snapshots once each, however many windows / directions share them, then the subsampled targets),
  2. ONE K/V projection GEMM per attention layer over the history rows of that pass (the table),
  3. the sparse history-attention kernel for the target rows and for the bsz * N_ents all-entity rows,
     each row listing its active history rows through an int32 map built on the host (membership is
     static).
T = L positions (uni: time_diff L-1..0) or 2(L-1)+1 (bi: forward history, backward history, current;
time_diff L-1..1, L-1..1, 0).
"""
import numpy as np
import torch

from . import functional as TF
from . import _lib
from . import snapshot as S
from .dynamic_rgcn import DynamicRGCN, WindowBatch
from .sargcn import SARGCN, jk_max
from .window import window_times


class SelfAttentionRGCN(DynamicRGCN):
    bidirectional = False

    def __init__(self, args, num_ents, num_rels, graph_dict_train, graph_dict_val, graph_dict_test, evaluater_type=None):
        super().__init__(args, num_ents, num_rels, graph_dict_train, graph_dict_val, graph_dict_test, evaluater_type)
        self.EMA = getattr(args, "EMA", False)
        if self.EMA:
            raise NotImplementedError("--EMA (models/SARGCN.py:64-81 stops in pdb.set_trace()) is outside the hot-path scope")

    def build_model(self):
        self.ent_encoder = SARGCN(self.args, self.hidden_size, self.embed_size, self.num_rels, self.total_time)
        self.register_buffer("time_diff_train", self._time_diff(self.train_seq_len), persistent=False)
        self.register_buffer("time_diff_test", self._time_diff(self.test_seq_len), persistent=False)

    def _time_diff(self, L):
        """models/SelfAttentionRGCN.py:22-23 / models/BiSelfAttentionRGCN.py:19-20."""
        if self.bidirectional:
            r = list(range(L - 1, 0, -1))
            return torch.tensor(r + r + [0.], dtype=torch.float32)
        return torch.arange(L - 1, -1, -1, dtype=torch.float32)

    # ---------------------------------------------------------------------------------------------------
    def _history_times(self, t_list, seq_len):
        """rows_f (target last) and, per window, the timestamps (or None) of its history positions in
        the order the reference concatenates them."""
        rows_f = window_times(t_list, seq_len, self.total_time)
        hist = [list(r[:seq_len - 1]) for r in rows_f]
        if self.bidirectional:
            rows_b = window_times(t_list, seq_len, self.total_time, ascending=True)[::-1]      # flip -> forward batch order
            assert [r[-1] for r in rows_b] == [r[-1] for r in rows_f]
            hist = [h + list(r[:seq_len - 1]) for h, r in zip(hist, rows_b)]
        return rows_f, hist

    def prepare(self, t_list, seq_len, train=True, target_edge_ids=None):
        dev = self._device()
        N = self.num_ents
        wb = WindowBatch()
        wb.rows, wb.hist_times = self._history_times(t_list, seq_len)
        wb.seq_len = seq_len
        wb.graphs = [self.graph_dict_train[r[-1]] for r in wb.rows]
        wb.targets = self.sample_target_graphs(wb.graphs, 0.5, target_edge_ids) if train else wb.graphs
        wb.target_sizes = [g.n for g in wb.targets]
        wb.target_times = [r[-1] for r in wb.rows]
        # distinct history snapshots, in first-use order
        # (--random-dropout: every training visit is its own 80 % edge subsample, so visits are keyed per window)
        # (self-loop dropout drawing: the reference encodes every history visit on its own, models/SelfAttentionRGCN.py:97-120, so
        #  every visit has its own mask -- visits are keyed per window then, too)
        resample = train and self.random_dropout
        per_visit = resample or not self._share_visits(train)
        node_row, hist_graphs, hist_ts, off = {}, [], [], 0
        for b, times in enumerate(wb.hist_times):
            for t in times:
                key = (b, t) if per_visit else t
                if t is not None and key not in node_row:
                    g = self.graph_dict_train[t]
                    if resample:
                        g = self.sample_target_graphs([g], 0.8)[0]
                    m = np.full(N, -1, dtype=np.int32)
                    m[g.gids] = off + np.arange(g.n, dtype=np.int32)
                    node_row[key] = m
                    hist_graphs.append(g)
                    hist_ts.append(t)
                    off += g.n
        wb.n_hist_rows = off
        none_row = np.full(N, -1, dtype=np.int32)
        idx_all = [np.stack([node_row[(b, t) if per_visit else t] if t is not None else none_row for t in times], axis=1) if times
                   else np.zeros((N, 0), np.int32) for b, times in enumerate(wb.hist_times)]                     # bsz x (N, Th)
        idx_tgt = [idx_all[b][g.gids] for b, g in enumerate(wb.targets)]
        all_graphs = hist_graphs + list(wb.targets)
        wb.g_all = S.batch(all_graphs)
        wb.g_all.device_graph(dev, 2 * self.num_rels)
        as_dev = lambda a, dt: _lib.to_device(np.ascontiguousarray(a).astype(dt), dev)
        wb.ids_all = as_dev(wb.g_all.gids, np.int32)
        wb.ids_inv = TF.gather_inverse(wb.g_all.gids, N, dev)
        time_rows = np.repeat(np.array(hist_ts + wb.target_times, dtype=np.int64), [g.n for g in all_graphs])
        wb.time_rows = as_dev(time_rows, np.int32)
        wb.time_inv = TF.gather_inverse(time_rows, len(self.total_time), dev)
        # the isolated (all-entity) pass is only needed for the entities that are NOT nodes of a window's target graph
        inact = [np.setdiff1d(np.arange(N, dtype=np.int64), g.gids) for g in wb.graphs]
        idx_tgt_np = np.concatenate(idx_tgt, axis=0)
        idx_all_np = np.concatenate([idx_all[b][inact[b]] for b in range(len(idx_all))], axis=0)
        wb.idx_tgt = as_dev(idx_tgt_np, np.int32)
        wb.idx_all = as_dev(idx_all_np, np.int32)
        wb.inv_tgt = TF.attention_inverse(idx_tgt_np, off, dev) if off > 0 and idx_tgt_np.shape[1] > 0 else None
        wb.inv_all = TF.attention_inverse(idx_all_np, off, dev) if off > 0 and idx_all_np.shape[1] > 0 and idx_all_np.shape[0] > 0 else None
        all_time_rows = np.repeat(np.array(wb.target_times, dtype=np.int64), [len(x) for x in inact])
        wb.all_time_rows = as_dev(all_time_rows, np.int32)
        wb.all_time_inv = TF.gather_inverse(all_time_rows, len(self.total_time), dev)
        sizes = [g.n for g in wb.graphs]
        n_out = int(sum(sizes))
        off_out = np.concatenate([[0], np.cumsum(sizes)])
        off_in = np.concatenate([[0], np.cumsum([len(x) for x in inact])])
        asm = np.empty((len(wb.graphs), N), dtype=np.int64)
        for b, g in enumerate(wb.graphs):
            asm[b, g.gids] = off_out[b] + np.arange(g.n)
            asm[b, inact[b]] = n_out + off_in[b] + np.arange(len(inact[b]))
        wb.inactive_ent = as_dev(np.concatenate(inact), np.int32)
        wb.n_inactive = int(off_in[-1])
        wb.assemble = as_dev(asm.reshape(-1), np.int32)
        wb.assemble_inv = TF.gather_inverse(asm.reshape(-1), n_out + wb.n_inactive, dev)
        wb.inactive_inv = TF.gather_inverse(np.concatenate(inact), N, dev)
        wb.time_diff = self.time_diff_train if seq_len == self.train_seq_len else self._time_diff(seq_len).to(dev)
        wb.batched = True
        wb.n_edge_visits = int(sum(self.graph_dict_train[t].number_of_edges() for times in wb.hist_times for t in times if t is not None)
                               + sum(g.number_of_edges() for g in wb.targets))
        wb.n_edges_distinct = int(wb.g_all.number_of_edges())
        wb.n_nodes_distinct = int(wb.g_all.n)
        wb.n_node_visits = int(sum(self.graph_dict_train[t].n for times in wb.hist_times for t in times if t is not None)
                               + sum(wb.target_sizes))
        if train:
            self._plan_loss(wb)
        return wb

    def _target_sizes(self, wb):
        return wb.target_sizes

    def _all_maps(self, wb):
        return None                                   # this model's all-entity maps are built in prepare

    def _fused_all_entity_ok(self, wb):
        return False                                  # (its own run_loss always takes the batched all-entity pass)

    def run(self, wb):
        """-> (target rows (sum n_b, D), (layer-1 K/V table or None, layer-2 K/V table))."""
        enc = self.ent_encoder
        l1, l2 = enc.layer_1, enc.layer_2
        R = wb.n_hist_rows
        y1 = l1.conv_table(wb.g_all, self.ent_embeds, wb.ids_all, wb.ids_inv)
        y2 = l2.conv(wb.g_all, y1)
        s = y2 + TF.gather_rows(l2.time_embed, wb.time_rows, wb.time_inv)
        # (split, not two slices: the backward of a split is ONE concatenation of the two gradients; two slices are two zero-filled
        # full-size gradients, two copies and an addition)
        s_hist, s_tgt = s.split([R, s.shape[0] - R])
        kv2 = l2.project_kv(s_hist)
        second = l2.attend(s_tgt, kv2, wb.idx_tgt, wb.time_diff, wb.inv_tgt)
        if enc.rec_only_last_layer:
            return second, (None, kv2)
        f = y1 + TF.gather_rows(l1.time_embed, wb.time_rows, wb.time_inv)
        f_hist, f_tgt = f.split([R, f.shape[0] - R])
        kv1 = l1.project_kv(f_hist)
        first = l1.attend(f_tgt, kv1, wb.idx_tgt, wb.time_diff, wb.inv_tgt)
        return jk_max(first, second), (kv1, kv2)

    def all_embeds_batched(self, wb, out, tables):
        """get_all_embeds_Gt for every window at once (models/SelfAttentionRGCN.py:28-45 with
        SARGCN.forward_isolated, models/SARGCN.py:119-125) -> (B, N_ents, D).  `out` = the concatenated target rows;
        the isolated pass runs only over the entities that are inactive in their window's target graph."""
        enc = self.ent_encoder
        l1, l2 = enc.layer_1, enc.layer_2
        if wb.n_inactive == 0:
            return self._assemble_all(wb, out, None)
        if getattr(self.args, "use_embed_for_non_active", False):
            return self._assemble_all(wb, out, TF.gather_rows(self.ent_embeds, wb.inactive_ent, wb.inactive_inv))
        kv1, kv2 = tables
        y1 = l1.conv_isolated(self.ent_embeds)
        if enc.rec_only_last_layer:
            y2 = TF.gather_rows(l2.conv_isolated(y1), wb.inactive_ent, wb.inactive_inv)
            first = None
        else:
            cur1 = TF.gather_rows(y1, wb.inactive_ent, wb.inactive_inv) + TF.gather_rows(l1.time_embed, wb.all_time_rows, wb.all_time_inv)
            first = l1.attend(cur1, kv1, wb.idx_all, wb.time_diff, wb.inv_all)
            y2 = l2.conv_isolated(first)
        cur2 = y2 + TF.gather_rows(l2.time_embed, wb.all_time_rows, wb.all_time_inv)
        second = l2.attend(cur2, kv2, wb.idx_all, wb.time_diff, wb.inv_all)
        return self._assemble_all(wb, out, second if first is None else jk_max(first, second))

    def encode(self, t_list, seq_len, train=True, target_edge_ids=None):
        wb = self.prepare(t_list, seq_len, train, target_edge_ids)
        out, tables = self.run(wb)
        return list(out.split(wb.target_sizes)), wb, tables

    def run_loss(self, wb, samples=None):
        dev = self._device()
        out, tables = self.run(wb)
        per_graph = list(out.split(wb.target_sizes))
        all_list = self.all_embeds_batched(wb, out, tables)
        if samples is None and getattr(wb, "loss_plan", None) is not None:
            fused = self._sampled_loss(wb, out, all_list)
            if fused is not None:
                return fused
        if samples is None:
            samples = self._samples_from_plan(wb) if getattr(wb, "loss_plan", None) is not None else self.draw_samples(wb)
        cache = getattr(wb, "_loss_inputs", None)
        if cache is None or cache[0] is not samples:
            offs = np.concatenate([[0], np.cumsum(wb.target_sizes)])[:-1]
            cache = wb._loss_inputs = (samples, self.loss_inputs([int(o) for o in offs], samples, dev, out.shape[0], self.rel_embeds.shape[0]))
        fused = self.batched_link_prediction(out, cache[1], all_list)
        if fused is not None:
            return fused
        loss = 0
        for i, ent_embed in enumerate(per_graph):
            triplets, neg_tail, neg_head = (x.to(dev) for x in samples[i])
            labels = torch.zeros(triplets.shape[0], dtype=torch.int64, device=dev)
            loss = loss + self.train_link_prediction_both(ent_embed, triplets, neg_tail, neg_head, labels, all_list[i])
        return loss

    def evaluate(self, t_list, val=True):
        """models/SelfAttentionRGCN.py:142-180 (Bi: models/BiSelfAttentionRGCN.py:71-95)."""
        from .evaluation import EvaluationFilter
        if not hasattr(self, "evaluater"):
            self.evaluater = EvaluationFilter(self.args, self.calc_score, self.graph_dict_train, self.graph_dict_val, self.graph_dict_test)
        graph_dict = self.graph_dict_val if val else self.graph_dict_test
        dev = self._device()
        with torch.no_grad():
            per_graph, wb, tables = self.encode(t_list, self.test_seq_len, train=False)
            all_list = self.all_embeds_batched(wb, torch.cat(per_graph, dim=0), tables)
            ranks, losses = [], []
            for i, ent_embed in enumerate(per_graph):
                t = wb.rows[i][-1]
                g = graph_dict[t]
                if g.number_of_edges() == 0:
                    continue
                index_sample = torch.from_numpy(np.stack([g.src, g.rel, g.dst], axis=1)).to(dev)
                label = torch.ones(index_sample.shape[0], device=dev)
                ranks.append(self.evaluater.calc_metrics_single_graph(ent_embed, self.rel_embeds, all_list[i], index_sample, g, t))
                losses.append(self.link_classification_loss(ent_embed, self.rel_embeds, index_sample, label).item())
        ranks = torch.cat(ranks) if ranks else torch.zeros(0, dtype=torch.int64, device=dev)
        return ranks, (float(np.mean(losses)) if losses else float("nan"))


class BiSelfAttentionRGCN(SelfAttentionRGCN):
    bidirectional = True
