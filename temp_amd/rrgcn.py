"""GRRGCNLayer / RRGCNLayer / RRGCN -- mirror of the reference's models/RRGCN.py interface on the
HIP kernels.  The GRU layers reproduce the reference's aliasing quirk: the GRU output
is written into the CALLER's graph object, so RRGCN.forward returns the same tensor twice.
"""
import torch
import torch.nn as nn

from . import functional as TF
from .gru_cell import GRUCell
from .rgcn import RGCNLayer


def run_rnn(rnn, x, prev, dt, lam, decay, prev_idx=None, prev_inv=None):
    """`self.rnn(x[None], decayed_prev.expand(num_layers, ...))` -> hidden[-1]
    (models/RRGCN.py:84-85).  `rnn` is an nn.GRU used as a parameter container (never called) or
    the type-1 GRUCell; every stacked layer restarts from the same decayed previous state."""
    if isinstance(rnn, GRUCell):
        return TF.gru_step(x, prev, dt, rnn.weight_ih, rnn.weight_hh, rnn.bias_ih, rnn.bias_hh, lam, decay, prev_idx, type1=True, prev_inv=prev_inv)
    inp = x
    for k in range(rnn.num_layers):
        inp = TF.gru_step(inp, prev, dt, getattr(rnn, 'weight_ih_l%d' % k), getattr(rnn, 'weight_hh_l%d' % k),
                          getattr(rnn, 'bias_ih_l%d' % k), getattr(rnn, 'bias_hh_l%d' % k), lam, decay, prev_idx, prev_inv=prev_inv)
    return inp


class GRRGCNLayer(RGCNLayer):
    """models/RRGCN.py:64-116."""

    def __init__(self, args, in_feat, out_feat, num_rels, num_bases, total_times, bias=True, activation=None,
                 self_loop=True, dropout=0.0):
        super().__init__(args, in_feat, out_feat, num_rels, num_bases, total_times, bias, activation, self_loop, dropout)
        self.post_aggregation = args.post_aggregation
        self.post_ensemble = args.post_ensemble
        self.num_layers = args.num_layers
        if args.type1:
            self.rnn = GRUCell(input_size=in_feat, hidden_size=out_feat)
        else:
            self.rnn = nn.GRU(input_size=in_feat, hidden_size=out_feat, num_layers=self.num_layers)

    def _extra(self):
        return self.post_aggregation or self.post_ensemble or self.impute

    def forward(self, g, prev_graph_embeds, time_diff_tensor, time_batched_list_t, node_sizes):
        result_graph, time_embedding = super().forward(g, time_batched_list_t, node_sizes)
        hidden = run_rnn(self.rnn, result_graph.ndata['h'], prev_graph_embeds, time_diff_tensor, self.inv_temperature,
                         self.decay_spec())
        g.ndata['h'] = hidden                      # written into the caller's graph (F7)
        if self._extra():
            return result_graph, g, time_embedding
        return g, time_embedding

    def forward_isolated(self, node_repr, prev_graph_embeds, time_diff_tensor, time):
        node_repr, time_embedding = super().forward_isolated(node_repr, time)
        hidden = run_rnn(self.rnn, node_repr, prev_graph_embeds, time_diff_tensor, self.inv_temperature, self.decay_spec())
        if self._extra():
            return node_repr, hidden, time_embedding
        return hidden, time_embedding

    def forward_isolated_impute(self, node_repr, imputation_weight, prev_graph_embeds_loc, prev_graph_embeds_rec,
                                time_diff_tensor, time):
        node_repr, time_embedding = super().forward_isolated(node_repr, time)
        node_repr = imputation_weight * prev_graph_embeds_loc + (1 - imputation_weight) * node_repr
        hidden = run_rnn(self.rnn, node_repr, prev_graph_embeds_rec, time_diff_tensor, self.inv_temperature, self.decay_spec())
        return hidden, time_embedding


class RRGCNLayer(RGCNLayer):
    """Linear recurrence, models/RRGCN.py:120-167:
    out = act(prop + (prev @ W_time) * exp(-dt*lam) [+bias] + loop).  The propagate + self-loop part
    runs on the HIP layer kernel; the recurrent term is the MFMA panel GEMM (temp_linear) followed by the
    row-decay kernel (temp_decay_rows) -- no library GEMM, no ATen arithmetic."""

    def __init__(self, args, in_feat, out_feat, num_rels, num_bases, total_times, bias=True, activation=None,
                 self_loop=True, dropout=0.0):
        super().__init__(args, in_feat, out_feat, num_rels, num_bases, total_times, bias, activation, self_loop, dropout)
        self.num_layers = args.num_layers
        self.time_weight = nn.Parameter(torch.Tensor(in_feat, out_feat))
        nn.init.xavier_uniform_(self.time_weight, gain=nn.init.calculate_gain('relu'))

    def _finish(self, pre, rec):
        out = pre + rec
        if self.bias:
            out = out + self.h_bias
        if self.activation:
            out = self.activation(out)
        return out

    def _linear_core(self, g, h):
        dg = g.device_graph(h.device, self.num_rels)
        return TF.rgcn_layer(h, dg, self.weight, self.loop_weight, None, self.num_bases, None, self._drop())

    def forward(self, g, prev_graph_embeds, time_diff_tensor, time_batched_list_t, node_sizes):
        g = g.local_var()
        pre = self._linear_core(g, g.ndata['h'])
        rec = TF.decay_rows(TF.linear_nt(prev_graph_embeds, self.time_weight), time_diff_tensor, self.inv_temperature)
        g.ndata['h'] = self._finish(pre, rec)
        return g, self.get_time_embedding(time_batched_list_t, node_sizes)

    def forward_isolated(self, ent_embeds, prev_graph_embeds, time_diff_tensor, time):
        pre = TF.rgcn_isolated(ent_embeds, self.loop_weight, None, None, self._drop())
        rec = TF.decay_rows(TF.linear_nt(prev_graph_embeds, self.time_weight), time_diff_tensor, self.inv_temperature)
        return self._finish(pre, rec), (self.time_embed[int(time)] if self.compute_time_embedding else None)


class RRGCN(nn.Module):
    """Uni-directional 2-layer container with the interface of models/RRGCN.py:170-272.

    Entry points and what they return (p1/p2 = previous layer-1 / layer-2 state, dt = (n,1)):
      forward(g, p1, p2, dt, times, sizes)                  -> (first_h, second_h)
      forward_isolated(e, p1, p2, dt, time)                 -> second_h
      forward_post_ensemble(g, p1, p2, dt, times, sizes)    -> (local_h2, first_h, second_h)
      forward_post_ensemble_isolated(e, p1, p2, dt, time, pre_loc) -> (local_h2, second_h)
      forward_isolated_impute(e, p1, p2, dt, time, pre_loc) -> second_h
    Layer 1 is recurrent only when args.rec_only_last_layer is False (:179-187)."""

    def __init__(self, args, hidden_size, embed_size, num_rels, total_times):
        super().__init__()
        self.rec_only_last_layer = args.rec_only_last_layer
        self.use_time_embedding = args.use_time_embedding
        rec_cls = {'GRRGCN': GRRGCNLayer, 'RRGCN': RRGCNLayer}[args.module]
        first_cls = RGCNLayer if self.rec_only_last_layer else rec_cls
        common = dict(bias=False, activation=None, self_loop=True, dropout=args.dropout)
        self.layer_1 = first_cls(args, embed_size, hidden_size, 2 * num_rels, args.n_bases, total_times, **common)
        self.layer_2 = rec_cls(args, hidden_size, hidden_size, 2 * num_rels, args.n_bases, total_times, **common)
        self.impute = args.impute
        if self.impute:
            self.impute_weight = nn.Linear(1, 1)
        for layer in (self.layer_1, self.layer_2):
            layer.compute_time_embedding = bool(self.use_time_embedding)

    # -- layer 1, graph and isolated flavours ------------------------------------------------------
    def _first(self, g, p1, dt, times, sizes):
        if self.rec_only_last_layer:
            return self.layer_1(g, times, sizes)[0]
        res = self.layer_1(g, p1, dt, times, sizes)          # (.., graph, time_emb); GRU layers alias `g` (F7)
        g1, temb = res[-2], res[-1]
        if self.use_time_embedding:
            g1.ndata['h'] = g1.ndata['h'] + temb
        return g1

    def _first_isolated(self, e, p1, dt, time):
        if self.rec_only_last_layer:
            return self.layer_1.forward_isolated(e, time)[0]
        res = self.layer_1.forward_isolated(e, p1, dt, time)
        y1, temb = res[-2], res[-1]
        return y1 + temb if self.use_time_embedding else y1

    # -- reference API -------------------------------------------------------------------------------
    def forward(self, batched_graph, first_prev_graph_embeds, second_prev_graph_embeds, time_diff_tensor,
                time_batched_list_t, node_sizes):
        g1 = self._first(batched_graph, first_prev_graph_embeds, time_diff_tensor, time_batched_list_t, node_sizes)
        g2, temb = self.layer_2(g1, second_prev_graph_embeds, time_diff_tensor, time_batched_list_t, node_sizes)
        if self.use_time_embedding:
            g2.ndata['h'] = g2.ndata['h'] + temb
        return g1.ndata['h'], g2.ndata['h']

    def forward_isolated(self, ent_embeds, first_prev_graph_embeds, second_prev_graph_embeds, time_diff_tensor, time):
        y1 = self._first_isolated(ent_embeds, first_prev_graph_embeds, time_diff_tensor, time)
        y2, temb = self.layer_2.forward_isolated(y1, second_prev_graph_embeds, time_diff_tensor, time)
        return y2 + temb if self.use_time_embedding else y2

    def forward_post_ensemble(self, batched_graph, first_prev_graph_embeds, second_prev_graph_embeds, time_diff_tensor,
                              time_batched_list_t, node_sizes):
        g1 = self._first(batched_graph, first_prev_graph_embeds, time_diff_tensor, time_batched_list_t, node_sizes)
        g_loc, g2, temb = self.layer_2(g1, second_prev_graph_embeds, time_diff_tensor, time_batched_list_t, node_sizes)
        if self.use_time_embedding:
            g_loc.ndata['h'] = g_loc.ndata['h'] + temb
            g2.ndata['h'] = g2.ndata['h'] + temb
        return g_loc.ndata['h'], g1.ndata['h'], g2.ndata['h']

    def forward_post_ensemble_isolated(self, ent_embeds, first_prev_graph_embeds, second_prev_graph_embeds, time_diff_tensor,
                                       time, pre_embeds_loc):
        y1 = self._first_isolated(ent_embeds, first_prev_graph_embeds, time_diff_tensor, time)
        loc2, y2, temb = self.layer_2.forward_isolated(y1, second_prev_graph_embeds, time_diff_tensor, time)
        if self.impute:
            w = self.calc_impute_weight(time_diff_tensor)
            loc2 = w * pre_embeds_loc + (1 - w) * loc2
        if self.use_time_embedding:
            loc2, y2 = loc2 + temb, y2 + temb
        return loc2, y2

    def forward_isolated_impute(self, ent_embeds, first_prev_graph_embeds, second_prev_graph_embeds, time_diff_tensor, time,
                                pre_embeds_loc):
        y1 = self._first_isolated(ent_embeds, first_prev_graph_embeds, time_diff_tensor, time)
        w = self.calc_impute_weight(time_diff_tensor)
        y2, temb = self.layer_2.forward_isolated_impute(y1, w, pre_embeds_loc, second_prev_graph_embeds, time_diff_tensor, time)
        return y2 + temb if self.use_time_embedding else y2

    def calc_impute_weight(self, time_diff_tensor):
        """exp(-max(0, Linear(dt))), models/RRGCN.py:271-272."""
        return torch.exp(-torch.clamp(self.impute_weight(time_diff_tensor), min=0))
