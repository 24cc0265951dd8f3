"""BiGRRGCNLayer / BiRRGCNLayer / BiRRGCN -- mirror of the reference's models/BiRRGCN.py interface
on the HIP kernels.  Layer 2 applies ReLU before the two GRUs (models/BiRRGCN.py:202-203)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as TF
from .gru_cell import GRUCell
from .rgcn import RGCNLayer
from .rrgcn import run_rnn


class BiGRRGCNLayer(RGCNLayer):
    """models/BiRRGCN.py:9-100."""

    def __init__(self, args, in_feat, out_feat, num_rels, num_bases, total_times, bias=None, activation=None,
                 self_loop=True, dropout=0.0):
        super().__init__(args, in_feat, out_feat, num_rels, num_bases, total_times, bias, activation, self_loop, dropout)
        self.num_layers = args.num_layers
        self.post_aggregation = args.post_aggregation
        self.post_ensemble = args.post_ensemble
        if args.type1:
            self.forward_rnn = GRUCell(input_size=in_feat, hidden_size=out_feat)
            self.backward_rnn = GRUCell(input_size=in_feat, hidden_size=out_feat)
        else:
            self.forward_rnn = nn.GRU(input_size=in_feat, hidden_size=out_feat, num_layers=self.num_layers)
            self.backward_rnn = nn.GRU(input_size=in_feat, hidden_size=out_feat, num_layers=self.num_layers)

    def _extra(self):
        return self.post_aggregation or self.post_ensemble or self.impute

    def _rnn(self, rnn, x, prev, dt):
        return run_rnn(rnn, x, prev, dt, self.inv_temperature, self.decay_spec())

    def forward(self, g, prev_graph_embeds_forward, time_diff_tensor_forward, prev_graph_embeds_backward,
                time_diff_tensor_backward, time_batched_list_t, node_sizes):
        result_graph, time_embedding = super().forward(g, time_batched_list_t, node_sizes)
        x = result_graph.ndata['h']
        hidden_forward = self._rnn(self.forward_rnn, x, prev_graph_embeds_forward, time_diff_tensor_forward)
        hidden_backward = self._rnn(self.backward_rnn, x, prev_graph_embeds_backward, time_diff_tensor_backward)
        g.ndata['h'] = hidden_forward + hidden_backward
        if self._extra():
            return result_graph, g, time_embedding
        return g, time_embedding

    def forward_one_direction(self, g, prev_graph_embeds, time_diff_tensor, time_batched_list_t, node_sizes, forward):
        result_graph, time_embedding = super().forward(g, time_batched_list_t, node_sizes)
        model = self.forward_rnn if forward else self.backward_rnn
        g.ndata['h'] = self._rnn(model, result_graph.ndata['h'], prev_graph_embeds, time_diff_tensor)
        if self._extra():
            return result_graph, g, time_embedding
        return g, time_embedding

    def forward_isolated(self, node_repr, prev_graph_embeds_forward, prev_graph_embeds_backward, time_diff_tensor_forward,
                         time_diff_tensor_backward, time):
        node_repr, time_embedding = super().forward_isolated(node_repr, time)
        hidden = self._rnn(self.forward_rnn, node_repr, prev_graph_embeds_forward, time_diff_tensor_forward) + \
            self._rnn(self.backward_rnn, node_repr, prev_graph_embeds_backward, time_diff_tensor_backward)
        if self._extra():
            return node_repr, hidden, time_embedding
        return hidden, time_embedding

    def forward_isolated_impute(self, node_repr, imputation_weight_forward, imputation_weight_backward, second_embeds_forward_loc,
                                second_embeds_backward_loc, prev_graph_embeds_forward, prev_graph_embeds_backward,
                                time_diff_tensor_forward, time_diff_tensor_backward, time):
        node_repr, time_embedding = super().forward_isolated(node_repr, time)
        node_repr = imputation_weight_forward * second_embeds_forward_loc + imputation_weight_backward * second_embeds_backward_loc + \
            (1 - imputation_weight_forward - imputation_weight_backward) * node_repr
        hidden = self._rnn(self.forward_rnn, node_repr, prev_graph_embeds_forward, time_diff_tensor_forward) + \
            self._rnn(self.backward_rnn, node_repr, prev_graph_embeds_backward, time_diff_tensor_backward)
        return hidden, time_embedding


class BiRRGCNLayer(RGCNLayer):
    """Linear bidirectional recurrence, models/BiRRGCN.py:102-185; the recurrent terms run on temp_decay_rows + the MFMA
    panel GEMM (temp_linear), like everything else of the encoder."""

    def __init__(self, args, in_feat, out_feat, num_rels, num_bases, total_times, bias=None, activation=None,
                 self_loop=True, dropout=0.0):
        super().__init__(args, in_feat, out_feat, num_rels, num_bases, total_times, bias, activation, self_loop, dropout)
        self.num_layers = args.num_layers
        self.time_weight_forward = nn.Parameter(torch.Tensor(in_feat, out_feat))
        nn.init.xavier_uniform_(self.time_weight_forward, gain=nn.init.calculate_gain('relu'))
        self.time_weight_backward = nn.Parameter(torch.Tensor(in_feat, out_feat))
        nn.init.xavier_uniform_(self.time_weight_backward, gain=nn.init.calculate_gain('relu'))

    def _finish(self, out):
        if self.bias:
            out = out + self.h_bias
        if self.activation:
            out = self.activation(out)
        return out

    def _core(self, g, h):
        dg = g.device_graph(h.device, self.num_rels)
        return TF.rgcn_layer(h, dg, self.weight, self.loop_weight, None, self.num_bases, None, self._drop())

    def forward(self, g, prev_graph_embeds_forward, time_diff_tensor_forward, prev_graph_embeds_backward,
                time_diff_tensor_backward, time_batched_list_t, node_sizes):
        g = g.local_var()
        lam = self.inv_temperature
        pre = self._core(g, g.ndata['h'])
        pre = pre + TF.linear_nt(TF.decay_rows(prev_graph_embeds_forward, time_diff_tensor_forward, lam), self.time_weight_forward)
        pre = pre + TF.linear_nt(TF.decay_rows(prev_graph_embeds_backward, time_diff_tensor_backward, lam), self.time_weight_backward)
        g.ndata['h'] = self._finish(pre)
        return g, self.get_time_embedding(time_batched_list_t, node_sizes)

    def forward_one_direction(self, g, prev_graph_embeds, time_diff_tensor, time_batched_list_t, node_sizes, forward):
        g = g.local_var()
        pre = self._core(g, g.ndata['h'])
        weight = self.time_weight_forward if forward else self.time_weight_backward
        pre = pre + TF.decay_rows(TF.linear_nt(prev_graph_embeds, weight), time_diff_tensor, self.inv_temperature)
        g.ndata['h'] = self._finish(pre)
        return g, self.get_time_embedding(time_batched_list_t, node_sizes)

    def forward_isolated(self, node_repr, prev_graph_embeds_forward, prev_graph_embeds_backward, time_diff_tensor_forward,
                         time_diff_tensor_backward, time):
        lam = self.inv_temperature
        pre = TF.rgcn_isolated(node_repr, self.loop_weight, None, None, self._drop())
        pre = pre + TF.linear_nt(TF.decay_rows(prev_graph_embeds_forward, time_diff_tensor_forward, lam), self.time_weight_forward)
        pre = pre + TF.linear_nt(TF.decay_rows(prev_graph_embeds_backward, time_diff_tensor_backward, lam), self.time_weight_backward)
        return self._finish(pre), (self.time_embed[int(time)] if self.compute_time_embedding else None)


class BiRRGCN(nn.Module):
    """models/BiRRGCN.py:188-338."""

    def __init__(self, args, hidden_size, embed_size, num_rels, total_times):
        super().__init__()
        self.rec_only_last_layer = args.rec_only_last_layer
        self.use_time_embedding = args.use_time_embedding
        module = {'BiGRRGCN': BiGRRGCNLayer, 'BiRRGCN': BiRRGCNLayer}[args.module]
        if not self.rec_only_last_layer:
            self.layer_1 = module(args, embed_size, hidden_size, 2 * num_rels, args.n_bases, total_times,
                                  bias=False, activation=None, self_loop=True, dropout=args.dropout)
        else:
            self.layer_1 = RGCNLayer(args, embed_size, hidden_size, 2 * num_rels, args.n_bases, total_times,
                                     bias=False, activation=None, self_loop=True, dropout=args.dropout)
        self.layer_2 = module(args, hidden_size, hidden_size, 2 * num_rels, args.n_bases, total_times,
                              bias=False, activation=F.relu, self_loop=True, dropout=args.dropout)
        self.impute = args.impute
        if self.impute:
            self.impute_weight_forward = nn.Linear(1, 1)
            self.impute_weight_backward = nn.Linear(1, 1)
        for layer in (self.layer_1, self.layer_2):
            layer.compute_time_embedding = bool(self.use_time_embedding)

    # -- helpers shared by the entry points -----------------------------------------------------
    def _layer1(self, batched_graph, f1, dtf, b1, dtb, tl, sizes, post):
        if not self.rec_only_last_layer:
            res = self.layer_1(batched_graph, f1, dtf, b1, dtb, tl, sizes)
            first_batched_graph, first_temp_embed = res[-2], res[-1]
            if self.use_time_embedding:
                first_batched_graph.ndata['h'] = first_batched_graph.ndata['h'] + first_temp_embed
            return first_batched_graph
        first_batched_graph, _ = self.layer_1(batched_graph, tl, sizes)
        return first_batched_graph

    def _layer1_one(self, batched_graph, p1, dt, tl, sizes, forward):
        if not self.rec_only_last_layer:
            res = self.layer_1.forward_one_direction(batched_graph, p1, dt, tl, sizes, forward)
            first_batched_graph, first_temp_embed = res[-2], res[-1]
            if self.use_time_embedding:
                first_batched_graph.ndata['h'] = first_batched_graph.ndata['h'] + first_temp_embed
            return first_batched_graph
        first_batched_graph, _ = self.layer_1(batched_graph, tl, sizes)
        return first_batched_graph

    def _layer1_iso(self, ent_embeds, f1, b1, dtf, dtb, time):
        if not self.rec_only_last_layer:
            res = self.layer_1.forward_isolated(ent_embeds, f1, b1, dtf, dtb, time)
            first_ent_embeds, first_time_embedding = res[-2], res[-1]
            if self.use_time_embedding:
                first_ent_embeds = first_ent_embeds + first_time_embedding
            return first_ent_embeds
        first_ent_embeds, _ = self.layer_1.forward_isolated(ent_embeds, time)
        return first_ent_embeds

    # -- reference API ----------------------------------------------------------------------------
    def forward(self, batched_graph, first_prev_graph_embeds_forward, second_prev_graph_embeds_forward, time_diff_tensor_forward,
                first_prev_graph_embeds_backward, second_prev_graph_embeds_backward, time_diff_tensor_backward,
                time_batched_list_t, node_sizes):
        first = self._layer1(batched_graph, first_prev_graph_embeds_forward, time_diff_tensor_forward,
                             first_prev_graph_embeds_backward, time_diff_tensor_backward, time_batched_list_t, node_sizes, False)
        second_batched_graph, second_temp_embed = self.layer_2(first, second_prev_graph_embeds_forward, time_diff_tensor_forward,
                                                               second_prev_graph_embeds_backward, time_diff_tensor_backward,
                                                               time_batched_list_t, node_sizes)
        if self.use_time_embedding:
            second_batched_graph.ndata['h'] = second_batched_graph.ndata['h'] + second_temp_embed
        return second_batched_graph.ndata['h']

    def forward_one_direction(self, batched_graph, first_prev_graph_embeds, second_prev_graph_embeds, time_diff_tensor,
                              time_batched_list_t, node_sizes, forward):
        first = self._layer1_one(batched_graph, first_prev_graph_embeds, time_diff_tensor, time_batched_list_t, node_sizes, forward)
        second_batched_graph, second_temp_embed = self.layer_2.forward_one_direction(
            first, second_prev_graph_embeds, time_diff_tensor, time_batched_list_t, node_sizes, forward)
        if self.use_time_embedding:
            second_batched_graph.ndata['h'] = second_batched_graph.ndata['h'] + second_temp_embed
        return first.ndata['h'], second_batched_graph.ndata['h']

    def forward_isolated(self, ent_embeds, first_prev_graph_embeds_forward, second_prev_graph_embeds_forward, time_diff_tensor_forward,
                         first_prev_graph_embeds_backward, second_prev_graph_embeds_backward, time_diff_tensor_backward, time):
        first_ent_embeds = self._layer1_iso(ent_embeds, first_prev_graph_embeds_forward, first_prev_graph_embeds_backward,
                                            time_diff_tensor_forward, time_diff_tensor_backward, time)
        second_ent_embeds, second_time_embedding = self.layer_2.forward_isolated(
            first_ent_embeds, second_prev_graph_embeds_forward, second_prev_graph_embeds_backward, time_diff_tensor_forward,
            time_diff_tensor_backward, time)
        if self.use_time_embedding:
            second_ent_embeds = second_ent_embeds + second_time_embedding
        return second_ent_embeds

    def forward_post_ensemble(self, batched_graph, first_prev_graph_embeds_forward, second_prev_graph_embeds_forward,
                              time_diff_tensor_forward, first_prev_graph_embeds_backward, second_prev_graph_embeds_backward,
                              time_diff_tensor_backward, time_batched_list_t, node_sizes):
        first = self._layer1(batched_graph, first_prev_graph_embeds_forward, time_diff_tensor_forward,
                             first_prev_graph_embeds_backward, time_diff_tensor_backward, time_batched_list_t, node_sizes, True)
        second_local_graph, second_batched_graph, second_temp_embed = self.layer_2(
            first, second_prev_graph_embeds_forward, time_diff_tensor_forward, second_prev_graph_embeds_backward,
            time_diff_tensor_backward, time_batched_list_t, node_sizes)
        if self.use_time_embedding:
            second_local_graph.ndata['h'] = second_local_graph.ndata['h'] + second_temp_embed
            second_batched_graph.ndata['h'] = second_batched_graph.ndata['h'] + second_temp_embed
        return second_local_graph.ndata['h'], second_batched_graph.ndata['h']

    def forward_post_ensemble_one_direction(self, batched_graph, first_prev_graph_embeds, second_prev_graph_embeds, time_diff_tensor,
                                            time_batched_list_t, node_sizes, forward):
        first = self._layer1_one(batched_graph, first_prev_graph_embeds, time_diff_tensor, time_batched_list_t, node_sizes, forward)
        second_local_graph, second_batched_graph, second_temp_embed = self.layer_2.forward_one_direction(
            first, second_prev_graph_embeds, time_diff_tensor, time_batched_list_t, node_sizes, forward)
        if self.use_time_embedding:
            second_local_graph.ndata['h'] = second_local_graph.ndata['h'] + second_temp_embed
            second_batched_graph.ndata['h'] = second_batched_graph.ndata['h'] + second_temp_embed
        return second_local_graph.ndata['h'], first.ndata['h'], second_batched_graph.ndata['h']

    def _impute_weights(self, time_diff_tensor_forward, time_diff_tensor_backward):
        wf = torch.exp(-torch.clamp(self.impute_weight_forward(time_diff_tensor_forward), min=0)) / 2
        wb = torch.exp(-torch.clamp(self.impute_weight_backward(time_diff_tensor_backward), min=0)) / 2
        return wf, wb

    def forward_post_ensemble_isolated(self, ent_embeds, first_prev_graph_embeds_forward, second_prev_graph_embeds_forward,
                                       time_diff_tensor_forward, first_prev_graph_embeds_backward, second_prev_graph_embeds_backward,
                                       time_diff_tensor_backward, time, second_embeds_forward_loc, second_embeds_backward_loc):
        first_ent_embeds = self._layer1_iso(ent_embeds, first_prev_graph_embeds_forward, first_prev_graph_embeds_backward,
                                            time_diff_tensor_forward, time_diff_tensor_backward, time)
        second_local_embeds, second_ent_embeds, second_time_embedding = self.layer_2.forward_isolated(
            first_ent_embeds, second_prev_graph_embeds_forward, second_prev_graph_embeds_backward, time_diff_tensor_forward,
            time_diff_tensor_backward, time)
        if self.impute:
            wf, wb = self._impute_weights(time_diff_tensor_forward, time_diff_tensor_backward)
            second_local_embeds = wf * second_embeds_forward_loc + wb * second_embeds_backward_loc + (1 - wf - wb) * second_local_embeds
        if self.use_time_embedding:
            second_local_embeds = second_local_embeds + second_time_embedding
            second_ent_embeds = second_ent_embeds + second_time_embedding
        return second_local_embeds, second_ent_embeds

    def forward_isolated_impute(self, ent_embeds, first_prev_graph_embeds_forward, second_prev_graph_embeds_forward,
                                time_diff_tensor_forward, first_prev_graph_embeds_backward, second_prev_graph_embeds_backward,
                                time_diff_tensor_backward, time, second_embeds_forward_loc, second_embeds_backward_loc):
        first_ent_embeds = self._layer1_iso(ent_embeds, first_prev_graph_embeds_forward, first_prev_graph_embeds_backward,
                                            time_diff_tensor_forward, time_diff_tensor_backward, time)
        wf, wb = self._impute_weights(time_diff_tensor_forward, time_diff_tensor_backward)
        second_ent_embeds, second_time_embedding = self.layer_2.forward_isolated_impute(
            first_ent_embeds, wf, wb, second_embeds_forward_loc, second_embeds_backward_loc, second_prev_graph_embeds_forward,
            second_prev_graph_embeds_backward, time_diff_tensor_forward, time_diff_tensor_backward, time)
        if self.use_time_embedding:
            second_ent_embeds = second_ent_embeds + second_time_embedding
        return second_ent_embeds
