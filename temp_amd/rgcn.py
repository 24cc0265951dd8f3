"""RGCNLayer / RGCN -- host-side mirror of the reference's models/RGCN.py interface (same
constructor arguments, method names, return arity and state_dict keys), computing through the
HIP kernels in libtemp_amd.so.  `g` is a temp_amd.snapshot.Snapshot (the DGL-graph stand-in)
with `g.ndata['h']` set by the caller, exactly like the reference.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as TF


def _act_name(activation):
    if activation is None:
        return None, None
    if activation in (F.relu, torch.relu) or activation == "relu":
        return "relu", None
    return None, activation          # unknown callable: applied unfused after the kernel


class RGCNLayer(nn.Module):
    """models/RGCN.py:7-107.  Parameters: time_embed (T,in), weight (num_rels, B*si*so),
    h_bias (out) iff bias, loop_weight (in,out), exponential_decay iff args.learnable_lambda."""

    def __init__(self, args, in_feat, out_feat, num_rels, num_bases, total_times, bias=True,
                 activation=None, self_loop=False, dropout=0.0):
        super().__init__()
        self.bias = bias
        self.activation = activation
        self._act, self._post_act = _act_name(activation)
        self.self_loop = self_loop
        if not self_loop:
            raise NotImplementedError("temp_amd RGCNLayer fuses the self-loop; every TeMP encoder uses self_loop=True "
                                      "(models/RGCN.py:149-152, models/RRGCN.py:180-187)")
        self.time_embed = nn.Parameter(torch.Tensor(len(total_times), in_feat))
        nn.init.xavier_uniform_(self.time_embed, gain=nn.init.calculate_gain('relu'))
        self.num_rels = num_rels
        self.num_bases = num_bases
        assert self.num_bases > 0
        self.in_feat, self.out_feat = in_feat, out_feat
        self.submat_in = in_feat // self.num_bases
        self.submat_out = out_feat // self.num_bases
        self.weight = nn.Parameter(torch.Tensor(self.num_rels, self.num_bases * self.submat_in * self.submat_out))
        nn.init.xavier_uniform_(self.weight, gain=nn.init.calculate_gain('relu'))
        if self.bias:
            self.h_bias = nn.Parameter(torch.Tensor(out_feat))
            nn.init.zeros_(self.h_bias)
        self.loop_weight = nn.Parameter(torch.Tensor(in_feat, out_feat))
        nn.init.xavier_uniform_(self.loop_weight, gain=nn.init.calculate_gain('relu'))
        self.dropout_p = float(dropout) if dropout else 0.0
        self.inv_temperature = args.inv_temperature
        self.learnable_lambda = args.learnable_lambda
        if self.learnable_lambda:
            self.exponential_decay = nn.Linear(1, 1)
        self.impute = args.impute
        self.compute_time_embedding = True      # containers switch it off when use_time_embedding is False

    # -- helpers --------------------------------------------------------------------------------
    def _drop(self):
        """(p, seed) of the self-loop dropout for THIS call (models/RGCN.py:57-59, training mode only), else None.
        The kernels derive the keep mask from the seed, so the autograd node only remembers the pair."""
        if self.dropout_p > 0 and self.training:
            # The seed is drawn on the host and reaches the kernels by value: a HIP graph captured now would replay THIS mask on
            # every launch (dropout would stop regularising, silently).  Refuse: callers that capture (bench GraphStep,
            # dist.ShardedStep) fall back to eager launches for a model whose dropout draws.
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("temp_amd: the self-loop dropout (p > 0, training) draws a host seed per call and cannot be captured "
                                   "into a HIP graph -- run this step eagerly")
            return (self.dropout_p, int(torch.randint(0, 2 ** 62, (1,)).item()))
        return None

    def _bias(self):
        return self.h_bias if self.bias else None

    def conv(self, g, h, grad_premasked=False, out=None):
        """The fused layer on node features `h` of graph `g` (models/RGCN.py:53-70).  grad_premasked: see TF.rgcn_layer (only
        meaningful for a fused ReLU; the caller guarantees the single masking consumer)."""
        dg = g.device_graph(h.device, self.num_rels)
        out = TF.rgcn_layer(h, dg, self.weight, self.loop_weight, self._bias(), self.num_bases, self._act, self._drop(),
                            grad_premasked=grad_premasked and self._post_act is None, out=out if self._post_act is None else None)
        return self._post_act(out) if self._post_act is not None else out

    def relu_fused(self):
        """True when this layer's activation is the ReLU the kernels fuse (its adjoint can move into the consumer's backward)."""
        return self._act == "relu" and self._post_act is None

    def conv_table(self, g, table, ids, inverse):
        """conv(g, table[ids]) for a layer fed straight from an embedding table (layer 1: h = ent_embeds[id],
        models/DynamicRGCN.py:93) -- the gather is folded into the kernels."""
        if inverse is None or self.in_feat > 256 or self.out_feat > 256:
            return self.conv(g, TF.gather_rows(table, ids))
        dg = g.device_graph(table.device, self.num_rels)
        out = TF.rgcn_layer_table(table, ids, inverse, dg, self.weight, self.loop_weight, self._bias(), self.num_bases, self._act, self._drop())
        return self._post_act(out) if self._post_act is not None else out

    def conv_isolated(self, e):
        out = TF.rgcn_isolated(e, self.loop_weight, self._bias(), self._act, self._drop())
        return self._post_act(out) if self._post_act is not None else out

    def get_time_embedding(self, time_batched_list_t, node_sizes):
        """models/RGCN.py:47-51 (zip truncates to the shorter list)."""
        if not self.compute_time_embedding:
            return None
        pairs = list(zip(time_batched_list_t, node_sizes))
        rows = np.repeat(np.array([int(t) for t, _ in pairs], dtype=np.int64), [int(s) for _, s in pairs])
        return self.time_embed[torch.from_numpy(rows).to(self.time_embed.device)]

    # -- reference API ----------------------------------------------------------------------------
    def forward(self, g, time_batched_list_t, node_sizes):
        g = g.local_var()
        g.ndata['h'] = self.conv(g, g.ndata['h'])
        return g, self.get_time_embedding(time_batched_list_t, node_sizes)

    def forward_isolated(self, ent_embeds, time):
        out = self.conv_isolated(ent_embeds)
        return out, (self.time_embed[int(time)] if self.compute_time_embedding else None)

    def decay_spec(self):
        """None for the fixed exp(-dt*inv_temperature); (weight, bias) for --learnable-lambda
        (RGCNLayer.decay_hidden, models/RGCN.py:106-107)."""
        if self.learnable_lambda:
            return (self.exponential_decay.weight, self.exponential_decay.bias)
        return None


class RGCN(nn.Module):
    """Static 2-layer encoder, models/RGCN.py:145-164: L1 (bias, no act) -> L2 (bias, ReLU)."""

    def __init__(self, args, hidden_size, embed_size, num_rels, total_times):
        super().__init__()
        self.use_time_embedding = args.use_time_embedding
        self.layer_1 = RGCNLayer(args, embed_size, hidden_size, 2 * num_rels, args.n_bases, total_times,
                                 activation=None, self_loop=True, dropout=args.dropout)
        self.layer_2 = RGCNLayer(args, hidden_size, hidden_size, 2 * num_rels, args.n_bases, total_times,
                                 activation=F.relu, self_loop=True, dropout=args.dropout)
        self.layer_1.compute_time_embedding = False
        self.layer_2.compute_time_embedding = bool(self.use_time_embedding)

    def forward(self, batched_graph, time_batched_list_t, node_sizes):
        first, _ = self.layer_1(batched_graph, time_batched_list_t, node_sizes)
        second, second_time_embedding = self.layer_2(first, time_batched_list_t, node_sizes)
        if self.use_time_embedding:
            second.ndata['h'] = second.ndata['h'] + second_time_embedding
        return second

    def forward_isolated(self, ent_embeds, time):
        first, _ = self.layer_1.forward_isolated(ent_embeds, time)
        second, second_time_embedding = self.layer_2.forward_isolated(first, time)
        return second + second_time_embedding if self.use_time_embedding else second
