"""SARGCNLayer / SARGCN -- the self-attention snapshot encoder with the reference's interface
(models/SARGCN.py:10-125; same constructor arguments, method names and state_dict keys:
q_linear / k_linear / v_linear `.weight`).

The reference attends over a DENSE (n, T-1, D) history tensor that is mostly zero rows hidden by an
additive -10e9 mask, and pushes every one of those rows through k_linear and v_linear.  Here the K/V
projections run once over a TABLE of the distinct history rows (`project_kv`), and each query row
carries the int32 rows of the positions where its node was active (`attend`): same result (a masked
position has softmax weight exactly 0 in fp32), ~T/active times less GEMM work and HBM traffic.
The dense reference signatures (`calc_result`, `forward_final`, `forward_isolated`) are kept and
routed through the same kernel.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as TF
from .rgcn import RGCNLayer

MASKED = -1e8        # the reference's mask value is -10e9; anything below this counts as masked


class SARGCNLayer(RGCNLayer):
    def __init__(self, args, in_feat, out_feat, num_rels, num_bases, total_times, bias=True, activation=None,
                 self_loop=True, dropout=0.0):
        super().__init__(args, in_feat, out_feat, num_rels, num_bases, total_times, bias, activation, self_loop, dropout)
        self.num_layers = args.num_layers
        self.q_linear = nn.Linear(in_feat, in_feat, bias=False)
        self.v_linear = nn.Linear(in_feat, in_feat, bias=False)
        self.k_linear = nn.Linear(in_feat, in_feat, bias=False)
        self.h = 8
        self.d_k = in_feat // self.h
        if in_feat % self.h:
            raise ValueError("SARGCNLayer needs in_feat divisible by its 8 heads (models/SARGCN.py:21-22)")
        self.post_aggregation = getattr(args, "post_aggregation", False)
        self.post_ensemble = getattr(args, "post_ensemble", False)
        if self.post_aggregation or self.post_ensemble:
            raise NotImplementedError("--post-aggregation / --post-ensemble are outside the hot-path scope (SURVEY section 8)")

    # -- table formulation (what the window models call) ----------------------------------------------
    def project_kv(self, rows):
        """[k_linear(rows) | v_linear(rows)] -> (R, 2D): one MFMA GEMM over the distinct history rows."""
        return TF.linear(rows.contiguous(), torch.cat([self.k_linear.weight, self.v_linear.weight], dim=0))

    def project_qkv(self, cur):
        return TF.linear(cur.contiguous(), torch.cat([self.q_linear.weight, self.k_linear.weight, self.v_linear.weight], dim=0))

    def decay_bias(self, time_diff):
        """models/SARGCN.py:26-29: -clamp(Linear(1,1)(time_diff), min=0) per position, None when fixed."""
        if not self.learnable_lambda:
            return None
        return -torch.clamp(self.exponential_decay(time_diff.unsqueeze(1)), min=0).reshape(-1)

    def attend(self, cur, kv_hist, idx, time_diff, inverse=None):
        """Attention of every row of `cur` (n,D) over kv_hist[idx[n,:]] (idx (n,T-1) int32, -1 masked)
        and itself.  `inverse` = TF.attention_inverse(idx, R) (static maps) makes the backward deterministic."""
        if kv_hist.shape[0] == 0:
            kv_hist, inverse = cur.new_zeros(1, 2 * self.in_feat), None
        return TF.history_attention(self.project_qkv(cur), kv_hist, idx, self.decay_bias(time_diff), inverse)

    # -- reference (dense) API ----------------------------------------------------------------------------
    def calc_result(self, cur_embeddings, prev_embeddings, time_diff, local_attn_mask):
        """models/SARGCN.py:25-37.  prev (n,T-1,D), mask (n,T) additive (0 / -10e9)."""
        n, Th, D = prev_embeddings.shape
        rows = torch.arange(n * Th, device=cur_embeddings.device, dtype=torch.int32).view(n, Th)
        idx = torch.where(local_attn_mask[:, :Th] > MASKED, rows, torch.full_like(rows, -1))
        kv = self.project_kv(prev_embeddings.reshape(n * Th, D))
        return self.attend(cur_embeddings, kv, idx.contiguous(), time_diff)

    def forward_final(self, g, prev_embeddings, time_diff, local_attn_mask, time_batched_list_t, node_sizes):
        current_graph, time_embedding = self.forward(g, time_batched_list_t, node_sizes)
        cur = current_graph.ndata['h'] + time_embedding
        return current_graph, self.calc_result(cur, prev_embeddings, time_diff, local_attn_mask)

    def forward_isolated(self, node_repr, prev_embeddings, time_diff, local_attn_mask, time):
        cur, time_embedding = super().forward_isolated(node_repr, time)
        return self.calc_result(cur + time_embedding, prev_embeddings, time_diff, local_attn_mask)


def jk_max(first, second):
    """torch.max(torch.stack([first, second], -1), -1)[0] (models/SARGCN.py:117,125): ties go to `first`."""
    return torch.where(second > first, second, first)


class SARGCN(nn.Module):
    """models/SARGCN.py:83-125.  Forces args.use_time_embedding = True like the reference."""

    def __init__(self, args, hidden_size, embed_size, num_rels, total_time):
        super().__init__()
        self.rec_only_last_layer = args.rec_only_last_layer
        args.use_time_embedding = True
        first = RGCNLayer if self.rec_only_last_layer else SARGCNLayer
        self.layer_1 = first(args, embed_size, hidden_size, 2 * num_rels, args.n_bases, total_time,
                             activation=None, self_loop=True, dropout=args.dropout)
        self.layer_2 = SARGCNLayer(args, hidden_size, hidden_size, 2 * num_rels, args.n_bases, total_time,
                                   activation=F.relu, self_loop=True, dropout=args.dropout)

    def forward(self, batched_graph, time_batched_list_t, node_sizes):
        """-> (layer-1 states + time embedding, layer-2 states + time embedding); layer 2 consumes the
        PLAIN layer-1 states (models/SARGCN.py:103-107)."""
        first, first_temb = self.layer_1(batched_graph, time_batched_list_t, node_sizes)
        second, second_temb = self.layer_2(first, time_batched_list_t, node_sizes)
        return first.ndata['h'] + first_temb, second.ndata['h'] + second_temb

    def forward_final(self, batched_graph, first_prev, second_prev, time_diff, local_attn_mask, time_batched_list_t, node_sizes):
        if not self.rec_only_last_layer:
            first, first_attn = self.layer_1.forward_final(batched_graph, first_prev, time_diff, local_attn_mask, time_batched_list_t, node_sizes)
        else:
            first, _ = self.layer_1(batched_graph, time_batched_list_t, node_sizes)
        _, second_attn = self.layer_2.forward_final(first, second_prev, time_diff, local_attn_mask, time_batched_list_t, node_sizes)
        return second_attn if self.rec_only_last_layer else jk_max(first_attn, second_attn)

    def forward_isolated(self, ent_embeds, first_prev, second_prev, time_diff, local_attn_mask, time):
        if not self.rec_only_last_layer:
            first = self.layer_1.forward_isolated(ent_embeds, first_prev, time_diff, local_attn_mask, time)
        else:
            first, _ = self.layer_1.forward_isolated(ent_embeds, time)
        second = self.layer_2.forward_isolated(first, second_prev, time_diff, local_attn_mask, time)
        return second if self.rec_only_last_layer else jk_max(first, second)
