"""DynamicRGCN -- the uni-directional window model with the reference's interface
(models/DynamicRGCN.py:14-220): `DynamicRGCN(args, num_ents, num_rels, graph_dict_train,
graph_dict_val, graph_dict_test)`, `.forward(t_list) -> loss`, same parameter / state_dict names.

Two execution paths, same results:
  * reference-granular: one encoder call per window position through the drop-in `RRGCN` API,
    previous states fetched by row-gather from the previous position's output (window.ChainPlan)
    instead of the reference's dense re-zeroed history;
  * batched (GRU module + --rec-only-last-layer, the paper's recommended mode): the two RGCN layers
    of EVERY visit of the step run as ONE launch each over a block-diagonal union of all visited
    snapshots, and only the fused decay+GRU kernel walks the window positions, reading its
    previous state through `prev_idx`.
"""
import numpy as np
import torch
import torch.nn as nn

from . import functional as TF
from . import _lib
from . import _hostlib
from . import snapshot as S
from .rrgcn import RRGCN, GRRGCNLayer, run_rnn
from .tkg_module import TKG_Module
from .gru_chain import GruInstance, GruProgram, gru_chain, prepare_program, zero_state_program
from .gru_cell import GRUCell
from .window import ChainPlan, Step, concat_steps, concat_steps_dedup, window_times


from .backend import get_backend  # noqa: E402

class WindowBatch:
    """Everything `run` needs for one batch of windows (plans, batched graphs, device index tensors)."""
    pass


class DynamicRGCN(TKG_Module):
    def __init__(self, args, num_ents, num_rels, graph_dict_train, graph_dict_val, graph_dict_test, evaluater_type=None):
        self.num_layers = args.num_layers
        self.train_seq_len = args.train_seq_len
        self.test_seq_len = args.train_seq_len            # --test-seq-len is ignored by the reference too (F10)
        super().__init__(args, num_ents, num_rels, graph_dict_train, graph_dict_val, graph_dict_test, evaluater_type)
        self.ent_embeds = nn.Parameter(torch.Tensor(self.num_ents, self.embed_size))
        self.rel_embeds = nn.Parameter(torch.Tensor(self.num_rels * 2, self.embed_size))
        self.edge_dropout = getattr(args, "edge_dropout", False)
        self.post_aggregation = getattr(args, "post_aggregation", False)
        self.random_dropout = getattr(args, "random_dropout", False)
        if self.edge_dropout:
            raise NotImplementedError("--edge-dropout: the reference's DropEdge.sample_subgraph reads drop_rate_cache, which is never built "
                                      "(utils/DropEdge.py:31-32,126); not on the hot path (SURVEY section 2)")
        nn.init.xavier_uniform_(self.ent_embeds, gain=nn.init.calculate_gain('relu'))
        nn.init.xavier_uniform_(self.rel_embeds, gain=nn.init.calculate_gain('relu'))
        self.sample_rng = np.random.default_rng(getattr(args, "seed", None))
        self.seed_rng = np.random.default_rng(None if getattr(args, "seed", None) is None else int(args.seed) + 1)   # per-step sampler seeds (main thread)
        self.plan_loss_in_prepare = True
        self.use_batched_path = True
        self.use_gru_chain = True
        self.dedup_snapshots = True
        self.use_rec_stack = True             # reference-default flags (both layers recurrent): one autograd node (rec_stack.py)
        self.device_subsample = True          # training-time edge subsets drawn and applied on the GPU (host sampler when False)

    def build_model(self):
        self.ent_encoder = RRGCN(self.args, self.hidden_size, self.embed_size, self.num_rels, self.total_time)

    # ---------------------------------------------------------------------------------------------
    # shared helpers
    # ---------------------------------------------------------------------------------------------
    def _device(self):
        return self.ent_embeds.device

    def _dropout_active(self, train=True):
        """The self-loop dropout of the RGCN layers (models/RGCN.py:57-59) draws in this pass: training mode, p > 0."""
        return bool(train and self.training and float(getattr(self.args, "dropout", 0.0) or 0.0) > 0.0)

    def _share_visits(self, train=True):
        """May visits of one snapshot share ONE RGCN pass?  Not while dropout draws: the reference runs the encoder per window
        position (models/DynamicRGCN.py:156-174), so every (window, position) visit gets its own mask, and layer 2 aggregates
        layer-1 rows that differ per visit -- nothing of a visit is shared then.  (Round-4 verdict: de-duplication ignored it.)"""
        return self.dedup_snapshots and not self._dropout_active(train)

    def _can_batch(self):
        enc = self.ent_encoder
        return (self.use_batched_path and enc.rec_only_last_layer and isinstance(enc.layer_2, GRRGCNLayer)
                and not (enc.layer_2.post_aggregation or enc.layer_2.post_ensemble or enc.layer_2.impute))

    def _can_stack(self):
        """Both layers recurrent (the reference's default: no --rec-only-last-layer) on the plain uni-directional GRU model:
        the position loop runs as ONE autograd node with layer 1's RGCN and all GRU GEMMs that do not depend on the
        recurrence hoisted out of it (rec_stack.py).  Anything else keeps the reference-granular loop."""
        enc = self.ent_encoder
        l1, l2 = enc.layer_1, enc.layer_2
        return (type(self) is DynamicRGCN and self.use_rec_stack and self.use_batched_path and not enc.rec_only_last_layer
                and isinstance(l1, GRRGCNLayer) and isinstance(l2, GRRGCNLayer) and not l1._extra() and not l2._extra()
                and l1.decay_spec() is None and l2.decay_spec() is None and not enc.use_time_embedding
                and getattr(l1, "num_layers", 1) == 1 and l1._post_act is None and l2._post_act is None
                and l1.inv_temperature == l2.inv_temperature)

    def _gather_prev(self, prev_out, idx_t, n):
        if prev_out is None:
            return self.ent_embeds.new_zeros(n, self.embed_size)
        return TF.gather_rows(prev_out, idx_t)

    def sample_target_graphs(self, graphs, rate=0.5, edge_ids=None):
        """Random edge subsample of the target snapshots with recomputed norms
        (get_batch_graph_embeds(full=False), models/DynamicRGCN.py:76-90).  `edge_ids`
        injects the kept edge ids (tests replay the reference's recorded draws)."""
        dev = self._device()
        if edge_ids is None and dev.type == "cuda" and self.device_subsample and S.DEVICE_STORE == "kernel" and graphs:
            # on the device: the subgraphs' sorted views are derived from the snapshots' resident views (temp_subsample_views) --
            # no host rebuild, no edge upload; the draw depends only on the per-graph seed
            seeds = self.sample_rng.integers(0, 1 << 62, size=len(graphs))
            return S.device_subsample(graphs, [int(rate * g.number_of_edges()) for g in graphs], seeds, dev, 2 * self.num_rels, want_mask=True)
        out = []
        for i, g in enumerate(graphs):
            E = g.number_of_edges()
            idx = edge_ids[i] if edge_ids is not None else _hostlib.sample_subset(E, int(rate * E), self.sample_rng)
            out.append(g.edge_subgraph(idx))
        return out

    def sample_history_graphs(self, plan, rate=0.8):
        """--random-dropout: every training visit of a history snapshot runs on its own random 80 % of the edges, same
        nodes, norms recomputed (get_per_graph_ent_embeds(full=False, rate=0.8) -> get_batch_graph_embeds,
        models/DynamicRGCN.py:76-90,162-171).  Row maps depend only on the node sets, so the plan is unchanged."""
        for st in plan.steps:
            st.graphs = self.sample_target_graphs(st.graphs, rate)
            st.graph = None

    def _target_step(self, plan, rows, graphs):
        L = plan.seq_len
        st = Step(L - 1, list(range(len(graphs))), graphs, [r[-1] for r in rows])
        pidx, dts = [], []
        for b, g in enumerate(graphs):
            a, d = plan.final_prev(b, g.gids, L - 1)
            pidx.append(a)
            dts.append(d)
        st.prev_idx = np.concatenate(pidx) if pidx else np.zeros(0, np.int64)
        st.dt = np.concatenate(dts) if dts else np.zeros(0, np.float32)
        return st

    # ---------------------------------------------------------------------------------------------
    # reference-granular path
    # ---------------------------------------------------------------------------------------------
    def _encode_step(self, st, prev_first, prev_second):
        dev = self._device()
        ids, pidx, dt = st.tensors(dev)
        g = st.batched()
        g.ndata['h'] = TF.gather_rows(self.ent_embeds, ids)
        fp = self._gather_prev(prev_first, pidx, st.n_rows)
        sp = self._gather_prev(prev_second, pidx, st.n_rows)
        return self.ent_encoder(g, fp, sp, dt, st.times, st.sizes)

    def pre_forward(self, plan):
        """History positions 0..L-2 (models/DynamicRGCN.py:156-174) -> outputs of the last executed one."""
        first = second = None
        for st in plan.steps:
            first, second = self._encode_step(st, first, second)
        return first, second

    def _run_generic(self, wb):
        first, second = self.pre_forward(wb.plan)
        _, out = self._encode_step(wb.target, first, second)
        return out, (first, second)

    def _run_stack(self, wb):
        """The same loop as _run_generic as one autograd node (rec_stack.py): layer 1's RGCN over the union of the distinct
        snapshots, then per position cell 1 -> RGCN 2 -> cell 2; both cells read h2 of the position before."""
        from .rec_stack import rec_stack
        enc, dev = self.ent_encoder, self._device()
        y1 = enc.layer_1.conv_table(wb.g_all, self.ent_embeds, wb.ids_all, wb.ids_inv)
        if wb.visit_rows is not None:
            y1 = TF.gather_rows(y1, wb.visit_rows, wb.visit_inv)
        graphs = [st.batched().device_graph(dev, 2 * self.num_rels) for st in wb.steps]
        g_union = wb.g_visits.device_graph(dev, 2 * self.num_rels) if getattr(wb, "g_visits", None) is not None else None
        got = rec_stack(y1, wb.program, graphs, enc.layer_1.rnn, enc.layer_2, self._chain_want(wb), g_union)
        hist = got[1] if wb.hist_inst >= 0 else None
        return got[0], (hist, hist)

    # ---------------------------------------------------------------------------------------------
    # batched path
    # ---------------------------------------------------------------------------------------------
    def _can_chain(self):
        """The whole recurrence as one autograd node (gru_chain): fixed decay, single GRU layer,
        no per-position time embedding."""
        enc = self.ent_encoder
        l2 = enc.layer_2
        return (self.use_gru_chain and l2.decay_spec() is None and not enc.use_time_embedding and getattr(l2, "num_layers", 1) == 1)

    def _visit_rows_on_device(self):
        return True

    def _build_program(self, wb):
        inst = []
        for k, st in enumerate(wb.steps):
            has_prev = k > 0
            inst.append(GruInstance(st.n_rows, st.row0, 0, k - 1 if has_prev else -1, st.prev_idx, st.dt, st.next_idx))
        wb.program = GruProgram(inst)
        wb.out_inst = [len(inst) - 1]
        wb.hist_inst = len(inst) - 2 if len(inst) > 1 else -1

    def _chain_want(self, wb):
        """Instances whose states leave the chain: the target position and the last history position."""
        return [wb.out_inst[0]] + ([wb.hist_inst] if wb.hist_inst >= 0 else [])

    def _run_batched(self, wb):
        enc, dev = self.ent_encoder, self._device()
        y1 = enc.layer_1.conv_table(wb.g_all, self.ent_embeds, wb.ids_all, wb.ids_inv)
        y2 = enc.layer_2.conv(wb.g_all, y1)
        if wb.visit_rows is not None:                 # distinct-snapshot rows -> visit rows
            y2 = TF.gather_rows(y2, wb.visit_rows, getattr(wb, "visit_inv", None))
        l2 = enc.layer_2
        wb.last_x = y2                                # GRU input rows of the step (= the "local" states of the post models)
        if wb.program is not None:
            want = self._chain_want(wb)
            got = gru_chain(y2, wb.program, [l2.rnn], l2.inv_temperature, isinstance(l2.rnn, GRUCell), want=want)
            hist = got[1] if wb.hist_inst >= 0 else None
            return got[0], (hist, hist)
        H, hist = None, None
        for st, x in zip(wb.steps, TF.row_spans(y2, [(st.row0, st.n_rows) for st in wb.steps])):
            _, pidx, dt = st.tensors(dev)
            prev = H if H is not None else x.new_zeros(1, x.shape[1])
            H = run_rnn(l2.rnn, x, prev, dt, l2.inv_temperature, l2.decay_spec(), pidx)
            if enc.use_time_embedding:
                H = H + l2.get_time_embedding(st.times, st.sizes)
            if st is not wb.target:
                hist = H
        return H, (hist, hist)

    # ---------------------------------------------------------------------------------------------
    def prepare(self, t_list, seq_len, train=True, target_edge_ids=None):
        """Host-side planning + upload of everything that depends only on which snapshots are
        visited (window layout, row maps, sorted/chunked edge views).  Returns a WindowBatch that
        `run` can execute any number of times."""
        dev = self._device()
        wb = WindowBatch()
        wb.rows = window_times(t_list, seq_len, self.total_time)
        wb.plan = ChainPlan(wb.rows, self.graph_dict_train, self.num_ents, seq_len)
        _lib.pause_point()                            # (between the stages of `prepare`: a prefetch worker parks here while the loop issues a step)
        if train and self.random_dropout:
            self.sample_history_graphs(wb.plan)
        wb.graphs = [self.graph_dict_train[r[-1]] for r in wb.rows]
        tgt = self.sample_target_graphs(wb.graphs, 0.5, target_edge_ids) if train else wb.graphs
        _lib.pause_point()
        wb.target = self._target_step(wb.plan, wb.rows, tgt)
        wb.batched = self._can_batch()
        wb.stack = not wb.batched and self._can_stack()
        wb.steps = wb.plan.steps + [wb.target]
        _lib.pause_point()
        self._upload(wb, dev, train)
        _lib.pause_point()
        if train:
            self._plan_loss(wb)
        return wb

    def _upload(self, wb, dev, train=True):
        wb.program = None
        wb.visit_rows = wb.visit_rows_host = None
        wb.shared_visits = self._share_visits(train)
        if wb.batched or getattr(wb, "stack", False):
            if wb.shared_visits:
                wb.g_all, vr, wb.total_rows = concat_steps_dedup(wb.steps)
                wb.visit_rows_host = vr                  # the device copy + its inverse only where `run` gathers by them
                on_dev = vr is not None and self._visit_rows_on_device()
                wb.visit_rows = _lib.to_device(vr, dev) if on_dev else None
                wb.visit_inv = TF.gather_inverse(vr, int(wb.g_all.n), dev) if on_dev else None
            else:
                wb.g_all, wb.total_rows = concat_steps(wb.steps)
            _lib.pause_point()
            wb.ids_all = _lib.to_device(wb.g_all.gids.astype(np.int32), dev)
            wb.ids_inv = TF.gather_inverse(wb.g_all.gids, self.num_ents, dev)        # static ids: deterministic embedding gradient
            _lib.pause_point()
            wb.g_all.device_graph(dev, 2 * self.num_rels)
            _lib.pause_point()
            if getattr(wb, "stack", False):          # layer 2 runs per position: every position's own union graph as well
                for st in wb.steps:
                    st.batched().device_graph(dev, 2 * self.num_rels)
                # ... and the union of ALL visits in program row order (layer 2's weight gradients in one pass, rec_stack.py): the
                # layer-1 graph itself when no snapshot is shared between visits
                wb.g_visits = wb.g_all if wb.visit_rows_host is None else concat_steps(wb.steps)[0]
                wb.g_visits.device_graph(dev, 2 * self.num_rels)
                self._build_program(wb)
                wb.program.upload(dev)
                wb.program.constants(dev, self.embed_size)
            elif self._can_chain():
                self._build_program(wb)
                _lib.pause_point()
                prepare_program(wb.program, dev, self.embed_size, len(wb.out_inst), self._chain_want(wb))
        else:
            for st in wb.steps:
                st.batched().device_graph(dev, 2 * self.num_rels)
        if wb.program is None:                       # the per-step row maps are only read outside the chain program
            for st in wb.steps:
                st.tensors(dev)
        wb.n_edge_visits = int(sum(g.number_of_edges() for st in wb.steps for g in st.graphs))
        wb.n_edges_distinct = int(wb.g_all.number_of_edges()) if wb.batched else wb.n_edge_visits
        wb.n_nodes_distinct = int(wb.g_all.n) if wb.batched else 0
        wb.n_node_visits = int(sum(st.n_rows for st in wb.steps))

    def run(self, wb):
        """Device work of one encoder pass -> (target rows (sum n_b, D), final history outputs)."""
        if getattr(wb, "stack", False):
            return self._run_stack(wb)
        return (self._run_batched if wb.batched else self._run_generic)(wb)

    def encode(self, t_list, seq_len, train=True, target_edge_ids=None):
        """Window encoder: -> (per-window target embeddings, plan, rows, target graphs, final history)."""
        wb = self.prepare(t_list, seq_len, train, target_edge_ids)
        out, hist = self.run(wb)
        return list(out.split(wb.target.sizes)), wb.plan, wb.rows, wb.graphs, hist

    def get_all_embeds_Gt(self, convoluted_embeds, g, t, plan, b, hist):
        """Isolated pass over ALL entities, then the active rows overwritten
        (models/DynamicRGCN.py:56-64).  Previous states come from the last history output by row map."""
        dev = self._device()
        L = plan.seq_len
        if getattr(self.args, "use_embed_for_non_active", False):
            all_embeds = self.ent_embeds
        else:
            row_of, dt = plan.final_all(b, L - 1)
            idx = torch.from_numpy(row_of.astype(np.int32)).to(dev)
            dt_t = torch.from_numpy(dt).view(-1, 1).to(dev)
            p1 = self._gather_prev(hist[0], idx, self.num_ents)
            p2 = p1 if hist[1] is hist[0] else self._gather_prev(hist[1], idx, self.num_ents)
            all_embeds = self.ent_encoder.forward_isolated(self.ent_embeds, p1, p2, dt_t, t)
        gid = torch.from_numpy(g.gids).to(dev)
        return all_embeds.index_copy(0, gid, convoluted_embeds)

    def evaluate(self, t_list, val=True):
        """models/DynamicRGCN.py:118-144,196-220: window encoder on the full train graphs, then filtered
        ranks of the valid (or test) triples of every target timestamp + their classification loss.
        (The reference skips graphs without triples WITHOUT advancing its history index; here the index
        always follows the window.)"""
        from .evaluation import EvaluationFilter
        if not hasattr(self, "evaluater"):
            self.evaluater = EvaluationFilter(self.args, self.calc_score, self.graph_dict_train, self.graph_dict_val, self.graph_dict_test)
        graph_dict = self.graph_dict_val if val else self.graph_dict_test
        dev = self._device()
        with torch.no_grad():
            wb = self.prepare(t_list, self.test_seq_len, train=False)
            out, hist = self.run(wb)
            per_graph, plan, rows = list(out.split(wb.target.sizes)), wb.plan, wb.rows
            # all windows' all-entity matrices in one batched pass when the model allows it (same classes as the training loss)
            all_b = self.all_embeds_batched(wb, out, hist) if self._fused_all_entity_ok(wb) else None
            ranks, losses = [], []
            for i, ent_embed in enumerate(per_graph):
                t = rows[i][-1]
                g = graph_dict[t]
                if g.number_of_edges() == 0:
                    continue
                all_embeds_g = all_b[i] if all_b is not None else self.get_all_embeds_Gt(ent_embed, g, t, plan, i, hist)
                index_sample = torch.from_numpy(np.stack([g.src, g.rel, g.dst], axis=1)).to(dev)
                label = torch.ones(index_sample.shape[0], device=dev)
                ranks.append(self.evaluater.calc_metrics_single_graph(ent_embed, self.rel_embeds, all_embeds_g, index_sample, g, t))
                losses.append(self.link_classification_loss(ent_embed, self.rel_embeds, index_sample, label))
        ranks = torch.cat(ranks) if ranks else torch.zeros(0, dtype=torch.int64, device=dev)
        return ranks, (float(torch.stack(losses).mean().item()) if losses else float("nan"))     # ONE device->host sync per batch

    def forward(self, t_list, reverse=False, target_edge_ids=None, samples=None):
        """models/DynamicRGCN.py:176-194.  `target_edge_ids` / `samples` inject the random draws
        (SURVEY F11); by default they are sampled here."""
        wb = self.prepare(t_list, self.train_seq_len, True, target_edge_ids)
        return self.run_loss(wb, samples)

    def _plan_loss(self, wb):
        """Host half of the fused loss, done with the rest of `prepare` (i.e. by the prefetch thread): positives, operand
        index lists, known-true slices for the device sampler, static inverses, all-entity row maps."""
        wb.loss_plan = None
        if not (self.plan_loss_in_prepare and getattr(self, "use_device_sampler", True) and self.fused_loss
                and self.args.score_function in ("distmult", "complex")):
            return
        from .sampling import TrueSetStore, plan_batch_loss
        dev = self._device()
        with _lib.create_lock:
            store = getattr(self, "_true_store", None)
            if store is None or store.device != dev:
                store = self._true_store = TrueSetStore(self.graph_dict_train, self.num_ents, dev)
        sizes = self._target_sizes(wb)
        offs = np.concatenate([[0], np.cumsum(sizes)])[:-1]
        wb.loss_plan = plan_batch_loss(store, [r[-1] for r in wb.rows], wb.graphs, offs, self.args.num_pos_facts, self.sample_rng,
                                       int(sum(sizes)), int(self.rel_embeds.shape[0]), dev)
        _lib.pause_point()
        if self._fused_all_entity_ok(wb):
            self._all_maps(wb)

    def _target_sizes(self, wb):
        return wb.target.sizes

    def _sampled_loss(self, wb, out, all_embeds):
        """Fused loss on fresh negatives: ONE sampler launch (temp_corrupt_sample) + the fused link-prediction node."""
        plan = wb.loss_plan
        cand = get_backend().corrupt_sample(int(self.seed_rng.integers(1 << 62)), plan["truth"], plan["lo"], plan["hi"], plan["ids"],
                                            self.args.negative_rate, self.num_ents)
        return self.batched_link_prediction(out, dict(plan, cand=cand), all_embeds)

    def _samples_from_plan(self, wb):
        """Reference-shaped samples [(triples, neg_tail, neg_head)] per target graph from the batch's loss plan: the negatives
        of all graphs come from ONE temp_corrupt_sample launch (the per-graph torch-op sampler costs ~1.4 ms per graph)."""
        plan = wb.loss_plan
        cand = get_backend().corrupt_sample(int(self.seed_rng.integers(1 << 62)), plan["truth"], plan["lo"], plan["hi"], plan["ids"],
                                            self.args.negative_rate, self.num_ents)
        out = []
        for trip, (a0, _) in zip(plan["triples"], plan["splits"]):
            P = trip.shape[0]
            out.append((torch.from_numpy(trip), cand[a0:a0 + P], cand[a0 + P:a0 + 2 * P]))
        return out

    def draw_samples(self, wb):
        """Negative samples of every target graph of a prepared batch.  On a GPU the draws and the true-triple filter
        run on the device (sampling.DeviceCorruptTriples); `use_device_sampler = False` keeps the host sampler."""
        dev = self._device()
        c = self.corrupter
        if dev.type == "cuda" and getattr(self, "use_device_sampler", True):
            from .sampling import DeviceCorruptTriples
            dc = getattr(self, "_dev_corrupter", None)
            if dc is None or dc.device != dev:
                dc = self._dev_corrupter = DeviceCorruptTriples(self.args, self.graph_dict_train, dev, seed=getattr(self.args, "seed", None))
            c = dc
        return [c.single_graph_negative_sampling(wb.rows[i][-1], g, self.num_ents)[:3] for i, g in enumerate(wb.graphs)]

    def _all_maps(self, wb):
        """Row maps of the batched all-entity pass (get_all_embeds_Gt for every window at once), cached on the batch.
        An entity of window b falls in one of three classes:
          active in b's target graph      -> its row of the encoder output (models/DynamicRGCN.py:60-63);
          inactive, with a previous state -> one row of a GRU pass over exactly those (window, entity) pairs;
          inactive, never seen in the window's history (previous state 0, the vast majority at ICEWS scale)
                                          -> GRU(x_e, 0), which does not depend on the window: row e of ONE N_ents-row table.
        Per chain plan (direction): the pairs with a previous state (entity, history row, time gap) and the map that assembles
        the (B, N_ents) rows from [encoder output ; pair rows ; table]; for the second direction of a bidirectional encoder
        the active rows map to -1 (the encoder output already holds both directions' sum)."""
        rep = self._all_rep()
        if getattr(wb, "all_maps", None) is None or getattr(wb, "all_rep", False) != rep:
            dev = self._device()
            N, B = self.num_ents, len(wb.graphs)
            plans = wb.plan if isinstance(wb.plan, tuple) else (wb.plan,)
            L = plans[0].seq_len
            # rep (the self-loop dropout draws): the reference runs forward_isolated per window, each with its own mask
            # (models/DynamicRGCN.py:56-64, models/RGCN.py:78-89), so nothing of the isolated pass is shared between windows: the
            # once-per-entity table becomes a (window, entity) table of B * N_ents rows -- the kernels' masks are a function of
            # (seed, row, column), so ONE launch over those rows gives every window its own mask
            T = B * N if rep else N
            tab = (np.arange(B * N, dtype=np.int64).reshape(B, N) if rep else np.broadcast_to(np.arange(N, dtype=np.int64)[None, :], (B, N)))
            wb.all_rep = rep
            if rep:
                ids = np.tile(np.arange(N, dtype=np.int32), B)
                wb.all_rep_ids = _lib.to_device(ids, dev)
                wb.all_rep_inv = TF.gather_inverse(ids, N, dev)
            act = np.zeros((B, N), dtype=bool)
            sizes = [g.n for g in wb.graphs]
            n_out = int(sum(sizes))
            off_out = np.concatenate([[0], np.cumsum(sizes)])
            for b, g in enumerate(wb.graphs):
                act[b, g.gids] = True
            wb.n_inactive = int(B * N - act.sum())
            host, meta = {}, []
            for d, plan in enumerate(plans):
                _lib.pause_point()
                row_of = np.stack([plan.final_all(b, L - 1)[0] for b in range(B)])
                gap = np.stack([plan.final_all(b, L - 1)[1] for b in range(B)])
                has = ~act & (row_of >= 0)
                bb, ee = np.nonzero(has)                                   # window-major
                n_prev = int(bb.shape[0])
                if d == 0:
                    asm = (n_out + n_prev + tab).copy()
                    asm[has] = n_out + np.arange(n_prev)
                    for b, g in enumerate(wb.graphs):
                        asm[b, g.gids] = off_out[b] + np.arange(g.n)
                    n_src = n_out + n_prev + T
                else:
                    asm = (n_prev + tab).copy()
                    asm[has] = np.arange(n_prev)
                    asm[act] = -1
                    n_src = n_prev + T
                if rep:
                    ee = bb * N + ee                                           # the pair's row of the (window, entity) table
                host["ent%d" % d], host["idx%d" % d], host["asm%d" % d] = ee, row_of[bb, ee % N], asm.reshape(-1)
                host["dt%d" % d] = gap[bb, ee % N].astype(np.float32).view(np.int32)       # float bits ride in the int32 pack
                meta.append((n_prev, n_src, ee, asm.reshape(-1), row_of[bb, ee % N]))
            _lib.pause_point()
            dd = S.upload_packed(host, dev, np.int32)
            wb.all_maps = []
            for d, (n_prev, n_src, ee, asm, idx_host) in enumerate(meta):
                wb.all_maps.append(dict(n_prev=n_prev, ent=dd["ent%d" % d], idx=dd["idx%d" % d], dt=dd["dt%d" % d].view(torch.float32).view(-1, 1), idx_host=idx_host,
                                        asm=dd["asm%d" % d], ent_inv=TF.gather_inverse(ee, T, dev) if n_prev else None,
                                        asm_inv=TF.gather_inverse(asm, n_src if wb.n_inactive else n_out, dev)))
        return wb.all_maps

    @staticmethod
    def _pair_inverse(m, key, prev):
        """Inverse of the pairs' history-row map m[key] over the rows of `prev` (a (window, entity) pair continues from its OWN
        last history row, so the map is injective): the adjoint of the previous-state gather is then a gather
        (TF.gru_step prev_inv).  Built on first use, cached on the batch's map; None if the rows repeat after all."""
        c = m.setdefault("_inv", {})
        k = (key, int(prev.shape[0]))
        if k not in c:
            c[k] = TF.injective_inverse(m[key + "_host"], prev.shape[0], prev.device)
        return c[k]

    def _assemble_all(self, wb, out, isolated):
        """-> (B, N_ents, D): per window the active rows from `out` (concatenated target rows), the rest from the
        isolated pass (None when every entity is active in its window's target graph).  One gather through a static map."""
        src = out if isolated is None else torch.cat([out, isolated], dim=0)
        big = TF.gather_rows(src, wb.assemble, wb.assemble_inv)
        return big.view(len(wb.graphs), self.num_ents, big.shape[1])

    def _isolated_rnns(self, hist):
        """[(GRU, final history of its direction)] of the recurrent layer."""
        return [(self.ent_encoder.layer_2.rnn, hist[1])]

    def _fused_all_entity_ok(self, wb):
        """The batched all-entity pass + fused loss apply: always on the batched path; on the reference-granular path when the
        model is the unidirectional GRU encoder with BOTH layers recurrent (the reference's default flags), where the entity
        classes of _all_maps carry over to the first layer as well."""
        enc = self.ent_encoder
        # (while the self-loop dropout draws the isolated pass runs over (window, entity) rows, every window with its own mask:
        #  _all_rep / _all_maps -- the reference's default is dropout 0.1, utils/args.py:17)
        plain = (not enc.use_time_embedding and not getattr(self.args, "use_embed_for_non_active", False)
                 and getattr(enc.layer_2, "num_layers", 1) == 1)
        if wb.batched:
            return plain
        return (plain and self.use_batched_path and not isinstance(wb.plan, tuple) and isinstance(enc.layer_1, GRRGCNLayer)
                and isinstance(enc.layer_2, GRRGCNLayer) and not (enc.layer_1._extra() or enc.layer_2._extra()))

    def _all_rep(self):
        """The batched all-entity pass keeps one row per (window, entity) instead of one per entity: while the dropout draws
        (`_force_all_rep`: tests compare the two layouts without dropout)."""
        return bool(self._dropout_active() or getattr(self, "_force_all_rep", False))

    def _zero_state_rows(self, rnn, x, layer):
        """GRU(x, 0) over all rows of x: input-gate GEMM + pointwise cell (gru_chain.zero_state_program); the general
        single-step kernel would still run its recurrent GEMM and d_prev / d_W_hh products on zeros."""
        if getattr(rnn, "num_layers", 1) != 1:
            N = x.shape[0]
            return run_rnn(rnn, x, x.new_zeros(1, x.shape[1]), x.new_zeros(N, 1), layer.inv_temperature, layer.decay_spec(),
                           torch.full((N,), -1, dtype=torch.int32, device=x.device))
        progs = self.__dict__.setdefault("_zero_progs", {})
        prog = progs.get(x.shape[0])
        if prog is None:
            prog = progs[x.shape[0]] = zero_state_program(x.shape[0])
        return gru_chain(x, prog, [rnn], layer.inv_temperature, isinstance(rnn, GRUCell), want=[0])[0]

    def all_embeds_batched(self, wb, out, hist):
        """get_all_embeds_Gt for ALL windows at once (models/DynamicRGCN.py:56-64; Bi: models/BiDynamicRGCN.py:102-112,
        models/BiRRGCN.py:65-82).  The isolated RGCN trunk e -> Iso2(Iso1(e)) does not depend on the window, so it runs ONCE
        over the N_ents entities; so does every GRU from a zero state.  Only the (window, entity) pairs that carry a previous
        state get their own GRU rows (see _all_maps) -- through BOTH layers when the first one is recurrent too
        (rec_only_last_layer = False: y1 = GRU_1(Iso1(e), prev), y2 = GRU_2(Iso2(y1), prev), models/RRGCN.py:234-253).
        `out` = the concatenated target rows of the encoder.  Returns (B, N_ents, D)."""
        enc = self.ent_encoder
        l1, l2 = enc.layer_1, enc.layer_2
        maps = self._all_maps(wb)
        B, N = len(wb.graphs), self.num_ents
        if wb.n_inactive == 0:
            return TF.gather_rows(out, maps[0]["asm"], maps[0]["asm_inv"]).view(B, N, out.shape[1])
        l1_rec = isinstance(l1, GRRGCNLayer)
        E = TF.gather_rows(self.ent_embeds, wb.all_rep_ids, wb.all_rep_inv) if wb.all_rep else self.ent_embeds      # (window, entity) rows
        iso1 = l1.conv_isolated(E)
        t1 = self._zero_state_rows(l1.rnn, iso1, l1) if l1_rec else iso1
        x = l2.conv_isolated(t1)
        lam, dec = l2.inv_temperature, l2.decay_spec()
        big = None
        for d, (m, (rnn, H)) in enumerate(zip(maps, self._isolated_rnns(hist))):
            parts = [out] if d == 0 else []
            if m["n_prev"]:
                if l1_rec:                                    # the pair's own first-layer state, then the second layer's input
                    y1p = run_rnn(l1.rnn, TF.gather_rows(iso1, m["ent"], m["ent_inv"]), hist[0], m["dt"], l1.inv_temperature, l1.decay_spec(), m["idx"],
                                  self._pair_inverse(m, "idx", hist[0]))
                    xp = l2.conv_isolated(y1p)
                else:
                    xp = TF.gather_rows(x, m["ent"], m["ent_inv"])
                parts.append(run_rnn(rnn, xp, H, m["dt"], lam, dec, m["idx"], self._pair_inverse(m, "idx", H)))
            parts.append(self._zero_state_rows(rnn, x, l2))                      # GRU(x_e, 0): one row per entity, every window
            g = TF.gather_rows(torch.cat(parts, dim=0), m["asm"], m["asm_inv"])
            big = g if big is None else big + g
        return big.view(B, N, big.shape[1])

    def run_loss(self, wb, samples=None):
        """Encoder pass + the per-window link-prediction losses (summed, as the reference does)."""
        dev = self._device()
        out, hist = self.run(wb)
        per_graph = list(out.split(wb.target.sizes))
        batched = self._fused_all_entity_ok(wb)
        all_list = self.all_embeds_batched(wb, out, hist) if batched else None
        if samples is None and batched and getattr(wb, "loss_plan", None) is not None:
            fused = self._sampled_loss(wb, out, all_list)
            if fused is not None:
                return fused
        if samples is None:
            samples = self._samples_from_plan(wb) if getattr(wb, "loss_plan", None) is not None else self.draw_samples(wb)
        if batched:
            cache = getattr(wb, "_loss_inputs", None)
            if cache is None or cache[0] is not samples:          # index tensors are static for a given sample set
                offs = np.concatenate([[0], np.cumsum(wb.target.sizes)])[:-1]
                cache = wb._loss_inputs = (samples, self.loss_inputs([int(o) for o in offs], samples, dev, out.shape[0], self.rel_embeds.shape[0]))
            fused = self.batched_link_prediction(out, cache[1], all_list)
            if fused is not None:
                return fused
        loss = 0
        for i, (g, ent_embed) in enumerate(zip(wb.graphs, per_graph)):
            t = wb.rows[i][-1]
            triplets, neg_tail, neg_head = samples[i]
            triplets, neg_tail, neg_head = triplets.to(dev), neg_tail.to(dev), neg_head.to(dev)
            labels = torch.zeros(triplets.shape[0], dtype=torch.int64, device=dev)
            all_embeds_g = all_list[i] if batched else self.get_all_embeds_Gt(ent_embed, g, t, wb.plan, i, hist)
            loss = loss + self.train_link_prediction_both(ent_embed, triplets, neg_tail, neg_head, labels, all_embeds_g)
        return loss
