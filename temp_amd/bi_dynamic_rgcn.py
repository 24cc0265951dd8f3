"""BiDynamicRGCN -- the bi-directional window model with the reference's interface
(models/BiDynamicRGCN.py:10-208).  A forward chain over [t-L+1 .. t-1] (forward_rnn) and a backward
chain over [t+L-1 .. t+1] (backward_rnn) feed the centre step
    H = GRU_f(Y2, dec(S_f)) + GRU_b(Y2, dec(S_b))            (models/BiRRGCN.py:27-47).
The batched path runs the 2 RGCN layers of all 2(L-1)+1 positions x bsz windows as one launch
each; only the GRU cells walk the two chains.
"""
import numpy as np
import torch

from . import functional as TF
from . import _lib
from . import snapshot as S
from .birrgcn import BiGRRGCNLayer, BiRRGCN
from .dynamic_rgcn import DynamicRGCN, WindowBatch
from .gru_cell import GRUCell
from .gru_chain import GruInstance, GruProgram, gru_chain, chain_kernels_usable
from .rrgcn import run_rnn
from .window import ChainPlan, Step, window_times


class BiDynamicRGCN(DynamicRGCN):
    def build_model(self):
        self.ent_encoder = BiRRGCN(self.args, self.hidden_size, self.embed_size, self.num_rels, self.total_time)

    def _can_batch(self):
        enc = self.ent_encoder
        return (self.use_batched_path and enc.rec_only_last_layer and isinstance(enc.layer_2, BiGRRGCNLayer)
                and not (enc.layer_2.post_aggregation or enc.layer_2.post_ensemble or enc.layer_2.impute))

    @staticmethod
    def get_batch_graph_list(t_list, seq_len, graph_dict):
        """models/BiDynamicRGCN.py:17-49 -> (g_fwd[p][b], t_fwd[p][b], g_bwd[p][b], t_bwd[p][b])."""
        times = list(graph_dict.keys())
        out = []
        for asc in (False, True):
            rows = window_times(t_list, seq_len, times, ascending=asc)
            tb = [list(x) for x in zip(*rows)]
            gb = [[graph_dict[t] if t is not None else None for t in col] for col in tb]
            out += [gb, tb]
        return tuple(out)

    # -- reference-granular path ------------------------------------------------------------------------
    def _encode_step_one_direction(self, st, prev_first, prev_second, forward):
        dev = self._device()
        ids, pidx, dt = st.tensors(dev)
        g = st.batched()
        g.ndata['h'] = TF.gather_rows(self.ent_embeds, ids)
        fp = self._gather_prev(prev_first, pidx, st.n_rows)
        sp = self._gather_prev(prev_second, pidx, st.n_rows)
        return self.ent_encoder.forward_one_direction(g, fp, sp, dt, st.times, st.sizes, forward)

    def pre_forward(self, plan, forward=True):
        first = second = None
        for st in plan.steps:
            first, second = self._encode_step_one_direction(st, first, second, forward)
        return first, second

    def _bi_target(self, plan_f, plan_b, rows, graphs):
        L = plan_f.seq_len
        st = self._target_step(plan_f, rows, graphs)
        pidx, dts = [], []
        for b, g in enumerate(graphs):
            a, d = plan_b.final_prev(b, g.gids, L - 1)
            pidx.append(a)
            dts.append(d)
        st_b = Step(L - 1, st.windows, graphs, st.times)
        st_b.prev_idx = np.concatenate(pidx) if pidx else np.zeros(0, np.int64)
        st_b.dt = np.concatenate(dts) if dts else np.zeros(0, np.float32)
        st_b.graph = None
        return st, st_b

    def _run_generic(self, wb):
        dev = self._device()
        plan_f, plan_b = wb.plan
        tf, tb = wb.target, wb.target_b
        hf = self.pre_forward(plan_f, True)
        hb = self.pre_forward(plan_b, False)
        ids, pf, dtf = tf.tensors(dev)
        _, pb, dtb = tb.tensors(dev)
        g = tf.batched()
        g.ndata['h'] = TF.gather_rows(self.ent_embeds, ids)
        n = tf.n_rows
        out = self.ent_encoder(g, self._gather_prev(hf[0], pf, n), self._gather_prev(hf[1], pf, n), dtf,
                               self._gather_prev(hb[0], pb, n), self._gather_prev(hb[1], pb, n), dtb, tf.times, tf.sizes)
        return out, (hf, hb)

    # -- batched path --------------------------------------------------------------------------------------
    def _run_batched(self, wb):
        enc, dev = self.ent_encoder, self._device()
        plan_f, plan_b = wb.plan
        tf, tb = wb.target, wb.target_b
        y1 = enc.layer_1.conv_table(wb.g_all, self.ent_embeds, wb.ids_all, wb.ids_inv)
        l2 = enc.layer_2
        lam, dec = l2.inv_temperature, l2.decay_spec()
        if wb.program is not None:
            prog = wb.program
            want = self._chain_want(wb)
            # y2 = relu(.) (models/BiRRGCN.py:202-203) is read by ONE consumer here, the gather into chain order: the ReLU's
            # adjoint rides in that gather's backward (one pass less over the (n, d) gradient)
            # (support is decided BEFORE the conv: ONE flag drops the mask from the layer's backward and hands it to the gather's)
            fold = l2.relu_fused() and TF.relu_gather_supported(l2.out_feat, wb.chain_inv)
            y2 = l2.conv(wb.g_all, y1, grad_premasked=fold)
            # (the gather also hands out the magnitude keys of the rows it writes: the f16 products of the chain scale x by them)
            keyed = TF.gather_keys_supported(y2.shape[1]) and chain_kernels_usable(y2.shape[1], 2)
            x_keys = None
            if keyed:
                wb.last_x, x_keys = TF.gather_rows(y2, wb.chain_rows, wb.chain_inv, relu_table=fold, keys=True)
            else:
                wb.last_x = TF.gather_rows(y2, wb.chain_rows, wb.chain_inv, relu_table=fold)      # GRU input rows in chain order
            got = dict(zip(want, gru_chain(wb.last_x, prog, [l2.forward_rnn, l2.backward_rnn], lam,
                                           isinstance(l2.forward_rnn, GRUCell), want=want, x_keys=x_keys)))
            out = got[wb.out_inst[0]] + got[wb.out_inst[1]]
            Hf, Hb = got.get(wb.hist_inst[0]), got.get(wb.hist_inst[1])
            return out, ((Hf, Hf), (Hb, Hb))
        y2 = l2.conv(wb.g_all, y1)                            # ReLU fused (models/BiRRGCN.py:202-203)
        if wb.visit_rows is not None:                         # distinct-snapshot rows -> visit rows
            y2 = TF.gather_rows(y2, wb.visit_rows, getattr(wb, "visit_inv", None))
        wb.last_x = y2

        # every position's rows of y2 (forward steps, backward steps, target rows) from ONE split (TF.row_spans)
        order = sorted([(st.row0, st.n_rows) for st in list(plan_f.steps) + list(plan_b.steps)] + [(tf.row0, tf.n_rows)])
        x_of = dict(zip(order, TF.row_spans(y2, order)))

        def chain(plan, rnn):
            H = None
            for st in plan.steps:
                _, pidx, dt = st.tensors(dev)
                x = x_of[(st.row0, st.n_rows)]
                prev = H if H is not None else x.new_zeros(1, x.shape[1])
                H = run_rnn(rnn, x, prev, dt, lam, dec, pidx)
                if enc.use_time_embedding:
                    H = H + l2.get_time_embedding(st.times, st.sizes)
            return H

        Hf = chain(plan_f, l2.forward_rnn)
        Hb = chain(plan_b, l2.backward_rnn)
        x = x_of[(tf.row0, tf.n_rows)]
        _, pf, dtf = tf.tensors(dev)
        _, pb, dtb = tb.tensors(dev)
        zero = x.new_zeros(1, x.shape[1])
        out = run_rnn(l2.forward_rnn, x, Hf if Hf is not None else zero, dtf, lam, dec, pf) + \
            run_rnn(l2.backward_rnn, x, Hb if Hb is not None else zero, dtb, lam, dec, pb)
        if enc.use_time_embedding:
            out = out + l2.get_time_embedding(tf.times, tf.sizes)
        return out, ((Hf, Hf), (Hb, Hb))

    def _chain_want(self, wb):
        return [i for i in (wb.out_inst[0], wb.out_inst[1], wb.hist_inst[0], wb.hist_inst[1]) if i >= 0]

    def _visit_rows_on_device(self):
        return not self._can_chain()              # the chain program gathers by its own row list (chain_rows, _build_program)

    def _build_program(self, wb):
        """Chain program with ONE contiguous row range per direction: x rows are laid out
        [forward history | target | backward history | target (second copy)] and the instances follow the
        same order, so each direction's input-gate GEMM, d_x GEMM and weight-gradient GEMMs run once over
        all of its 15 positions (the copy costs one row gather; its gradient is summed by the gather's
        backward)."""
        plan_f, plan_b = wb.plan
        tf, tb = wb.target, wb.target_b
        nf = sum(st.n_rows for st in plan_f.steps)
        nb = sum(st.n_rows for st in plan_b.steps)
        nt = tf.n_rows
        total = nf + nb + nt
        base = wb.visit_rows_host.astype(np.int64) if wb.visit_rows_host is not None else np.arange(total, dtype=np.int64)
        assert base.shape[0] == total and tf.row0 == nf + nb
        f_rows = base[:nf]
        b_rows = base[nf:nf + nb]
        t_rows = base[tf.row0:tf.row0 + nt]
        chain = np.concatenate([f_rows, t_rows, b_rows, t_rows])
        wb.chain_rows = _lib.to_device(chain.astype(np.int32), self._device())
        wb.chain_inv = TF.gather_inverse(chain, int(wb.g_all.n), self._device())
        _lib.pause_point()
        inst = []
        last = -1
        for st in plan_f.steps:
            inst.append(GruInstance(st.n_rows, st.row0, 0, last, st.prev_idx, st.dt, st.next_idx))
            last = len(inst) - 1
        inst.append(GruInstance(nt, nf, 0, last, tf.prev_idx, tf.dt))
        hist_f, out_f = last, len(inst) - 1
        last = -1
        for st in plan_b.steps:
            inst.append(GruInstance(st.n_rows, st.row0 + nt, 1, last, st.prev_idx, st.dt, st.next_idx))      # backward rows sit after the target copy
            last = len(inst) - 1
        inst.append(GruInstance(nt, nf + nt + nb, 1, last, tb.prev_idx, tb.dt))
        hist_b, out_b = last, len(inst) - 1
        wb.program = GruProgram(inst)
        wb.program.x_src = chain                  # x row i is layer-output row chain[i]: the input gates are computed once per
                                                  # distinct row of a direction (GruProgram.gi_shared)
        assert len(wb.program.groups) <= 2
        wb.out_inst = [out_f, out_b]
        wb.hist_inst = [hist_f, hist_b]

    # ---------------------------------------------------------------------------------------------
    def prepare(self, t_list, seq_len, train=True, target_edge_ids=None):
        dev = self._device()
        wb = WindowBatch()
        wb.rows = window_times(t_list, seq_len, self.total_time)
        rows_b = window_times(t_list, seq_len, self.total_time, ascending=True)
        plan_f = ChainPlan(wb.rows, self.graph_dict_train, self.num_ents, seq_len)
        _lib.pause_point()                            # (between the stages of `prepare`: a prefetch worker parks here while the loop issues a step)
        plan_b = ChainPlan(rows_b, self.graph_dict_train, self.num_ents, seq_len).flipped()
        _lib.pause_point()
        wb.plan = (plan_f, plan_b)
        if train and self.random_dropout:
            self.sample_history_graphs(plan_f)
            self.sample_history_graphs(plan_b)
        wb.graphs = [self.graph_dict_train[r[-1]] for r in wb.rows]
        tgt = self.sample_target_graphs(wb.graphs, 0.5, target_edge_ids) if train else wb.graphs
        _lib.pause_point()
        wb.target, wb.target_b = self._bi_target(plan_f, plan_b, wb.rows, tgt)
        wb.batched = self._can_batch()
        wb.steps = plan_f.steps + plan_b.steps + [wb.target]
        _lib.pause_point()
        self._upload(wb, dev, train)
        _lib.pause_point()
        if wb.program is None:
            wb.target_b.tensors(dev)
        if train:
            self._plan_loss(wb)
        return wb

    def encode(self, t_list, seq_len, train=True, target_edge_ids=None):
        wb = self.prepare(t_list, seq_len, train, target_edge_ids)
        out, hist = self.run(wb)
        return list(out.split(wb.target.sizes)), wb.plan, wb.rows, wb.graphs, hist

    def _isolated_rnns(self, hist):
        l2 = self.ent_encoder.layer_2
        (Hf, _), (Hb, _) = hist
        return [(l2.forward_rnn, Hf), (l2.backward_rnn, Hb)]

    # -- reference-default flags (both layers recurrent): all-entity pass of all windows at once -------------------
    def _fused_all_entity_ok(self, wb):
        if wb.batched:
            return super()._fused_all_entity_ok(wb)
        enc = self.ent_encoder
        # (this variant keeps ONE isolated pass per entity: not while the self-loop dropout draws -- every window then needs its own mask)
        return (self.use_batched_path and not enc.use_time_embedding and not getattr(self.args, "use_embed_for_non_active", False)
                and isinstance(enc.layer_1, BiGRRGCNLayer) and isinstance(enc.layer_2, BiGRRGCNLayer) and not self._all_rep()
                and getattr(enc.layer_2, "num_layers", 1) == 1 and not (enc.layer_1._extra() or enc.layer_2._extra()))

    def _all_maps(self, wb):
        """Batched path: per-direction maps (DynamicRGCN._all_maps).  Reference-granular path (both layers recurrent): the
        first layer's state of a (window, entity) pair depends on BOTH directions' previous states, so the pairs are the union
        -- a previous state in either direction -- each with its two history rows (-1 = zero state in that direction); every
        other inactive entity takes its row of the once-per-entity table."""
        if wb.batched:
            return super()._all_maps(wb)
        if getattr(wb, "all_maps", None) is None:
            dev = self._device()
            N, B = self.num_ents, len(wb.graphs)
            plan_f, plan_b = wb.plan
            L = plan_f.seq_len
            act = np.zeros((B, N), dtype=bool)
            sizes = [g.n for g in wb.graphs]
            n_out = int(sum(sizes))
            off_out = np.concatenate([[0], np.cumsum(sizes)])
            for b, g in enumerate(wb.graphs):
                act[b, g.gids] = True
            wb.n_inactive = int(B * N - act.sum())
            rf = np.stack([plan_f.final_all(b, L - 1)[0] for b in range(B)])
            rb = np.stack([plan_b.final_all(b, L - 1)[0] for b in range(B)])
            gf = np.stack([plan_f.final_all(b, L - 1)[1] for b in range(B)])
            gb = np.stack([plan_b.final_all(b, L - 1)[1] for b in range(B)])
            has = ~act & ((rf >= 0) | (rb >= 0))
            bb, ee = np.nonzero(has)
            n_prev = int(bb.shape[0])
            asm = np.broadcast_to(n_out + n_prev + np.arange(N, dtype=np.int64)[None, :], (B, N)).copy()
            asm[has] = n_out + np.arange(n_prev)
            for b, g in enumerate(wb.graphs):
                asm[b, g.gids] = off_out[b] + np.arange(g.n)
            host = dict(ent=ee, idx_f=rf[bb, ee], idx_b=rb[bb, ee], asm=asm.reshape(-1),
                        dt_f=gf[bb, ee].astype(np.float32).view(np.int32), dt_b=gb[bb, ee].astype(np.float32).view(np.int32))
            d = S.upload_packed(host, dev, np.int32)
            f32 = lambda t: t.view(torch.float32).view(-1, 1)
            wb.all_maps = [dict(n_prev=n_prev, ent=d["ent"], idx_f=d["idx_f"], idx_b=d["idx_b"], dt_f=f32(d["dt_f"]), dt_b=f32(d["dt_b"]),
                                idx_f_host=host["idx_f"], idx_b_host=host["idx_b"],
                                asm=d["asm"], ent_inv=TF.gather_inverse(ee, N, dev) if n_prev else None,
                                asm_inv=TF.gather_inverse(asm.reshape(-1), n_out + n_prev + N if wb.n_inactive else n_out, dev))]
        return wb.all_maps

    def all_embeds_batched(self, wb, out, hist):
        if wb.batched:
            return super().all_embeds_batched(wb, out, hist)
        # both layers recurrent (models/BiRRGCN.py:242-257): zero-state rows once per entity through both layers' GRU pairs,
        # own rows for the union pairs
        enc = self.ent_encoder
        l1, l2 = enc.layer_1, enc.layer_2
        (m,) = self._all_maps(wb)
        B, N = len(wb.graphs), self.num_ents
        if wb.n_inactive == 0:
            return TF.gather_rows(out, m["asm"], m["asm_inv"]).view(B, N, out.shape[1])
        (f1, f2), (b1, b2) = hist
        iso1 = l1.conv_isolated(self.ent_embeds)
        t1 = self._zero_state_rows(l1.forward_rnn, iso1, l1) + self._zero_state_rows(l1.backward_rnn, iso1, l1)
        x2 = l2.conv_isolated(t1)
        parts = [out]
        if m["n_prev"]:
            zero = iso1.new_zeros(1, iso1.shape[1])

            def both(layer, x, pf, pb):
                lam, dec = layer.inv_temperature, layer.decay_spec()
                none = torch.full_like(m["idx_f"], -1)
                return run_rnn(layer.forward_rnn, x, pf if pf is not None else zero, m["dt_f"], lam, dec, m["idx_f"] if pf is not None else none,
                               self._pair_inverse(m, "idx_f", pf) if pf is not None else None) + \
                    run_rnn(layer.backward_rnn, x, pb if pb is not None else zero, m["dt_b"], lam, dec, m["idx_b"] if pb is not None else none,
                            self._pair_inverse(m, "idx_b", pb) if pb is not None else None)
            h1p = both(l1, TF.gather_rows(iso1, m["ent"], m["ent_inv"]), f1, b1)
            parts.append(both(l2, l2.conv_isolated(h1p), f2, b2))
        parts.append(self._zero_state_rows(l2.forward_rnn, x2, l2) + self._zero_state_rows(l2.backward_rnn, x2, l2))
        big = TF.gather_rows(torch.cat(parts, dim=0), m["asm"], m["asm_inv"])
        return big.view(B, N, big.shape[1])

    def get_all_embeds_Gt(self, convoluted_embeds, g, t, plans, b, hist):
        """models/BiDynamicRGCN.py:102-112."""
        dev = self._device()
        plan_f, plan_b = plans
        hf, hb = hist
        L = plan_f.seq_len
        if getattr(self.args, "use_embed_for_non_active", False):
            all_embeds = self.ent_embeds
        else:
            def prevs(plan, h):
                row_of, dt = plan.final_all(b, L - 1)
                idx = torch.from_numpy(row_of.astype(np.int32)).to(dev)
                p1 = self._gather_prev(h[0], idx, self.num_ents)
                p2 = p1 if h[1] is h[0] else self._gather_prev(h[1], idx, self.num_ents)
                return p1, p2, torch.from_numpy(dt).view(-1, 1).to(dev)
            f1, f2, dtf = prevs(plan_f, hf)
            b1, b2, dtb = prevs(plan_b, hb)
            all_embeds = self.ent_encoder.forward_isolated(self.ent_embeds, f1, f2, dtf, b1, b2, dtb, t)
        gid = torch.from_numpy(g.gids).to(dev)
        return all_embeds.index_copy(0, gid, convoluted_embeds)
