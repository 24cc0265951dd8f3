"""CPU oracle for the TeMP snapshot-encoder hot path (RGCN message passing + GRU/BiGRU window).

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this module; `temp_amd/` never does.

What it is: a PyTorch-CPU restatement of the *reference's op sequence* for this path (the
reference is 100 % Python on torch/DGL ops), written on plain tensors instead of
DGL graphs.  Every function cites the reference file:line it follows (paths relative to the
TeMP repository root).  Arithmetic is fp32 by default (`dtype=torch.float64` for tight checks).

Parity pinning: the reference ships NO tests, golden vectors or fixtures for this path
(SURVEY.md F2), so there is nothing reference-held to check against.  Instead this oracle is
pinned against outputs of the reference's OWN modules (`models/RGCN.py`, `RRGCN.py`,
`BiRRGCN.py`, `GRU_cell.py`, `DynamicRGCN.py`, `BiDynamicRGCN.py`, `utils/scores.py`) imported
and executed in the build container by `oracle/gen_golden.py`, whose outputs are committed under
`tests/golden/*.npz` and replayed by `tests/test_oracle_golden.py`.  The one piece of third-party
arithmetic that had to be restated to run them is DGL 0.4.1's builtin `fn.sum` reducer
(README.md:15 pins `dgl-cuda10.1==0.4.1`; not vendored, not installed): sum of messages over
in-edges == `index_add` by destination.  That piece is "parity unpinned" by any DGL artefact;
everything else is the reference's own code executed as-is.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# Graph container (the DGL-graph field contract of utils/dataset.py:210-231 as plain tensors)
# --------------------------------------------------------------------------------------
class SnapGraph:
    """One snapshot (or a disjoint union of snapshots).

    n      : number of local nodes
    src,dst: (E,) int64 local node ids               -- g.edges()
    rel    : (E,) int64 relation ids                  -- g.edata['type_s']
    ids    : (n,) int64 global entity ids             -- g.ndata['id'] / g.ids
    nnorm  : (n,) float, 1/in_deg (0 when in_deg==0)  -- g.ndata['norm']
    enorm  : (E,) float, nnorm[dst]                   -- g.edata['norm']
    """

    def __init__(self, n, src, dst, rel, ids, nnorm=None, enorm=None):
        self.n = int(n)
        self.src = torch.as_tensor(src, dtype=torch.int64).view(-1)
        self.dst = torch.as_tensor(dst, dtype=torch.int64).view(-1)
        self.rel = torch.as_tensor(rel, dtype=torch.int64).view(-1)
        self.ids = torch.as_tensor(ids, dtype=torch.int64).view(-1)
        if nnorm is None:
            nnorm = comp_deg_norm(self.n, self.dst)
        self.nnorm = torch.as_tensor(nnorm, dtype=torch.float32).view(-1)
        self.enorm = self.nnorm[self.dst] if enorm is None else torch.as_tensor(enorm, dtype=torch.float32).view(-1)

    @property
    def num_edges(self):
        return int(self.src.shape[0])


def comp_deg_norm(n, dst):
    """utils/utils.py:74-79 -- 1/in_degree in fp32, inf -> 0."""
    in_deg = torch.bincount(torch.as_tensor(dst, dtype=torch.int64), minlength=n).float().numpy()
    with np.errstate(divide='ignore'):
        norm = 1.0 / in_deg
    norm[np.isinf(norm)] = 0
    return torch.from_numpy(norm.astype(np.float32))


def edge_subgraph(g, idx):
    """models/DynamicRGCN.py:80-90 -- keep edges `idx` (in that order), same node set,
    recompute node norm from the subgraph's in-degrees and edge norm = dst node norm
    (utils/utils.py:23-28)."""
    idx = torch.as_tensor(idx, dtype=torch.int64)
    return SnapGraph(g.n, g.src[idx], g.dst[idx], g.rel[idx], g.ids)


def batch_graphs(graphs):
    """dgl.batch as used at models/DynamicRGCN.py:92 -- disjoint union with node-id offsets."""
    off, src, dst, rel, ids, nn_, en_ = 0, [], [], [], [], [], []
    for g in graphs:
        src.append(g.src + off)
        dst.append(g.dst + off)
        rel.append(g.rel)
        ids.append(g.ids)
        nn_.append(g.nnorm)
        en_.append(g.enorm)
        off += g.n
    cat = lambda xs, dt: torch.cat(xs) if xs else torch.zeros(0, dtype=dt)
    return SnapGraph(off, cat(src, torch.int64), cat(dst, torch.int64), cat(rel, torch.int64),
                     cat(ids, torch.int64), cat(nn_, torch.float32), cat(en_, torch.float32))


# --------------------------------------------------------------------------------------
# a1-a4: RGCN layer
# --------------------------------------------------------------------------------------
def rgcn_messages(h, g, weight, num_bases):
    """RGCNLayer.msg_func, models/RGCN.py:91-98: per-edge block-diagonal product
    (index_select of the relation row, bmm of (1,si)@(si,so) per block) times the edge norm."""
    in_feat = h.shape[1]
    si = in_feat // num_bases
    so = weight.shape[1] // (num_bases * si)
    out_feat = so * num_bases
    w = weight.index_select(0, g.rel).view(-1, si, so)
    node = h[g.src].view(-1, 1, si)
    msg = torch.bmm(node, w).view(-1, out_feat)
    return msg * g.enorm.to(h.dtype).view(-1, 1)


def rgcn_propagate(h, g, weight, num_bases):
    """RGCNLayer.propagate + apply_func, models/RGCN.py:100-104: fn.sum over in-edges
    (restated as index_add by dst -- the DGL builtin), then times the node norm again
    (double normalisation; zero-in-degree rows are exactly 0)."""
    so_total = weight.shape[1] // (h.shape[1] // num_bases)
    if g.num_edges == 0:
        # DGL runs only the apply step: ndata['h'] (the layer input) * norm (== 0 everywhere).
        return h * g.nnorm.to(h.dtype).view(-1, 1)
    msg = rgcn_messages(h, g, weight, num_bases)
    agg = h.new_zeros(g.n, so_total).index_add(0, g.dst, msg)
    return agg * g.nnorm.to(h.dtype).view(-1, 1)


def rgcn_layer(h, g, weight, loop_weight, num_bases, bias=None, act=None):
    """RGCNLayer.forward, models/RGCN.py:53-76 with dropout=0 (F12): self-loop mm, propagate,
    + bias, + loop message, activation (`act` is None or 'relu')."""
    loop = torch.mm(h, loop_weight)
    out = rgcn_propagate(h, g, weight, num_bases)
    if bias is not None:
        out = out + bias
    out = out + loop
    if act == 'relu':
        out = F.relu(out)
    return out


def rgcn_layer_isolated(e, loop_weight, bias=None, act=None):
    """RGCNLayer.forward_isolated, models/RGCN.py:78-89: e + e@W_loop [+bias][act] (F9)."""
    out = e + torch.mm(e, loop_weight)
    if bias is not None:
        out = out + bias
    if act == 'relu':
        out = F.relu(out)
    return out


def time_embedding_rows(time_embed, times, node_sizes):
    """RGCNLayer.get_time_embedding, models/RGCN.py:47-51 (zip stops at the shorter list, so
    trailing None in `times` are never touched)."""
    rows = [time_embed[int(t)].unsqueeze(0).expand(size, time_embed.shape[1]) for t, size in zip(times, node_sizes)]
    return torch.cat(rows, dim=0)


# --------------------------------------------------------------------------------------
# a5-a7: decay + GRU cells
# --------------------------------------------------------------------------------------
def decay_hidden(prev, dt, inv_temperature, learnable=None):
    """models/RRGCN.py:79-83 (fixed lambda) / models/RGCN.py:106-107 (learnable: Linear(1,1)
    then clamp(min=0)).  `dt` is (n,1)."""
    dt = dt.to(prev.dtype).view(-1, 1)
    if learnable is not None:
        w, b = learnable
        return prev * torch.exp(-torch.clamp(dt * w.view(1, 1) + b.view(1, 1), min=0))
    return prev * torch.exp(-dt * inv_temperature)


def gru_torch(x, h, w_ih, w_hh, b_ih, b_hh):
    """nn.GRU single step, one layer (models/RRGCN.py:72,84): PyTorch gate equations, rows of
    weight_ih_l0 / weight_hh_l0 ordered r, z, n."""
    gi = torch.mm(x, w_ih.t()) + b_ih
    gh = torch.mm(h, w_hh.t()) + b_hh
    i_r, i_z, i_n = gi.chunk(3, 1)
    h_r, h_z, h_n = gh.chunk(3, 1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (1 - z) * n + z * h


def gru_type1(x, h, w_ih, w_hh, b_ih, b_hh):
    """GRUCell ("type-1"), models/GRU_cell.py:18-30: gates r,z from the hidden state only,
    W_ih is (H,I) and feeds the new gate only; h' = n + z*(h-n)."""
    i_n = torch.mm(x, w_ih.t()) + b_ih
    gh = torch.mm(h, w_hh.t()) + b_hh
    h_r, h_i, h_n = gh.chunk(3, 1)
    r = torch.sigmoid(h_r)
    z = torch.sigmoid(h_i)
    n = torch.tanh(i_n + r * h_n)
    return n + z * (h - n)


def gru_stack(x, h0, rnn, type1=False):
    """`self.rnn(x[None], h0.expand(num_layers, ...))` -> hidden[-1] (models/RRGCN.py:84-85).
    `rnn` is a list of per-layer dicts {w_ih,w_hh,b_ih,b_hh}; layer k>0 consumes layer k-1's
    output as input and the SAME decayed h0 (the expand)."""
    if type1:
        p = rnn[0]
        return gru_type1(x, h0, p['w_ih'], p['w_hh'], p['b_ih'], p['b_hh'])
    inp = x
    for p in rnn:
        inp = gru_torch(inp, h0, p['w_ih'], p['w_hh'], p['b_ih'], p['b_hh'])
    return inp


# --------------------------------------------------------------------------------------
# Parameter containers.  A layer is a dict with keys:
#   weight (2R, B*si*so), loop_weight (D,D), h_bias (D,)|None, time_embed (T,D),
#   rnn / forward_rnn / backward_rnn : list of {w_ih,w_hh,b_ih,b_hh}
#   time_weight / time_weight_forward / time_weight_backward (D,D)   (linear-recurrence modules)
#   exponential_decay: (w,b) | None
# cfg is a dict: n_bases, inv_temperature, rec_only_last_layer, use_time_embedding, type1,
#   learnable_lambda, module ('GRRGCN'|'RRGCN'|'BiGRRGCN'|'BiRRGCN'|'SRGCN')
# --------------------------------------------------------------------------------------
def _decay(layer, cfg, prev, dt):
    return decay_hidden(prev, dt, cfg['inv_temperature'], layer.get('exponential_decay') if cfg.get('learnable_lambda') else None)


def grrgcn_layer(layer, cfg, g, h, prev, dt, bias=None, act=None):
    """GRRGCNLayer.forward, models/RRGCN.py:77-89 -> (pre-GRU RGCN output, GRU output)."""
    y = rgcn_layer(h, g, layer['weight'], layer['loop_weight'], cfg['n_bases'], bias, act)
    hid = gru_stack(y, _decay(layer, cfg, prev, dt), layer['rnn'], cfg.get('type1', False))
    return y, hid


def rrgcn_linear_layer(layer, cfg, g, h, prev, dt, bias=None, act=None):
    """RRGCNLayer.forward, models/RRGCN.py:130-150: out = act(prop + (prev@W_time)*exp(-dt*lam) [+bias] + loop)."""
    loop = torch.mm(h, layer['loop_weight'])
    out = rgcn_propagate(h, g, layer['weight'], cfg['n_bases'])
    out = out + torch.mm(prev, layer['time_weight']) * torch.exp(-dt.to(h.dtype).view(-1, 1) * cfg['inv_temperature'])
    if bias is not None:
        out = out + bias
    out = out + loop
    if act == 'relu':
        out = F.relu(out)
    return out


def static_rgcn_forward(enc, cfg, g, h0, times=None, node_sizes=None):
    """RGCN.forward, models/RGCN.py:154-159: L1 (bias, no act) -> L2 (bias, ReLU) [+time emb]."""
    l1, l2 = enc['layer_1'], enc['layer_2']
    y1 = rgcn_layer(h0, g, l1['weight'], l1['loop_weight'], cfg['n_bases'], l1.get('h_bias'), None)
    y2 = rgcn_layer(y1, g, l2['weight'], l2['loop_weight'], cfg['n_bases'], l2.get('h_bias'), 'relu')
    if cfg.get('use_time_embedding'):
        y2 = y2 + time_embedding_rows(l2['time_embed'], times, node_sizes)
    return y2


def static_rgcn_isolated(enc, cfg, e, t=None):
    """RGCN.forward_isolated, models/RGCN.py:161-164."""
    l1, l2 = enc['layer_1'], enc['layer_2']
    y1 = rgcn_layer_isolated(e, l1['loop_weight'], l1.get('h_bias'), None)
    y2 = rgcn_layer_isolated(y1, l2['loop_weight'], l2.get('h_bias'), 'relu')
    if cfg.get('use_time_embedding'):
        y2 = y2 + l2['time_embed'][int(t)]
    return y2


def rrgcn_forward(enc, cfg, g, h0, first_prev, second_prev, dt, times=None, node_sizes=None, post=False):
    """RRGCN.forward / forward_post_ensemble, models/RRGCN.py:192-204, 219-233.

    Returns (first_h, second_h) -- for the GRU module both are the layer-2 GRU output (the
    reference writes it into the shared graph object); `post=True` returns
    (local_h2, first_h, second_h) where local_h2 is the pre-GRU layer-2 RGCN output."""
    l1, l2 = enc['layer_1'], enc['layer_2']
    te = cfg.get('use_time_embedding', False)
    gru = cfg['module'] == 'GRRGCN'
    if cfg['rec_only_last_layer']:
        y1 = rgcn_layer(h0, g, l1['weight'], l1['loop_weight'], cfg['n_bases'])
    elif gru:
        _, y1 = grrgcn_layer(l1, cfg, g, h0, first_prev, dt)
        if te:
            y1 = y1 + time_embedding_rows(l1['time_embed'], times, node_sizes)
    else:
        y1 = rrgcn_linear_layer(l1, cfg, g, h0, first_prev, dt)
        if te:
            y1 = y1 + time_embedding_rows(l1['time_embed'], times, node_sizes)
    if gru:
        loc2, out = grrgcn_layer(l2, cfg, g, y1, second_prev, dt)
    else:
        loc2 = None
        out = rrgcn_linear_layer(l2, cfg, g, y1, second_prev, dt)
    if te:
        t2 = time_embedding_rows(l2['time_embed'], times, node_sizes)
        out = out + t2
        if loc2 is not None:
            loc2 = loc2 + t2
    first = out if gru else y1          # F7 aliasing only exists in the GRU layers
    if post:
        return loc2, first, out
    return first, out


def rrgcn_isolated(enc, cfg, e, first_prev, second_prev, dt, t=None, post=False):
    """RRGCN.forward_isolated / forward_post_ensemble_isolated (no impute),
    models/RRGCN.py:206-217, 235-253 with GRRGCNLayer.forward_isolated :91-103."""
    l1, l2 = enc['layer_1'], enc['layer_2']
    te = cfg.get('use_time_embedding', False)
    gru = cfg['module'] == 'GRRGCN'
    if cfg['rec_only_last_layer']:
        y1 = rgcn_layer_isolated(e, l1['loop_weight'])
    elif gru:
        y1 = rgcn_layer_isolated(e, l1['loop_weight'])
        y1 = gru_stack(y1, _decay(l1, cfg, first_prev, dt), l1['rnn'], cfg.get('type1', False))
        if te:
            y1 = y1 + l1['time_embed'][int(t)]
    else:
        y1 = rrgcn_linear_isolated(l1, cfg, e, first_prev, dt)
        if te:
            y1 = y1 + l1['time_embed'][int(t)]
    if gru:
        loc2 = rgcn_layer_isolated(y1, l2['loop_weight'])
        out = gru_stack(loc2, _decay(l2, cfg, second_prev, dt), l2['rnn'], cfg.get('type1', False))
    else:
        loc2 = None
        out = rrgcn_linear_isolated(l2, cfg, y1, second_prev, dt)
    if te:
        out = out + l2['time_embed'][int(t)]
        if loc2 is not None:
            loc2 = loc2 + l2['time_embed'][int(t)]
    return (loc2, out) if post else out


def rrgcn_linear_isolated(layer, cfg, e, prev, dt, bias=None, act=None):
    """RRGCNLayer.forward_isolated, models/RRGCN.py:152-167."""
    out = e + torch.mm(e, layer['loop_weight'])
    out = out + torch.mm(prev, layer['time_weight']) * torch.exp(-dt.to(e.dtype).view(-1, 1) * cfg['inv_temperature'])
    if bias is not None:
        out = out + bias
    if act == 'relu':
        out = F.relu(out)
    return out


# ---- bidirectional ---------------------------------------------------------------------
def bigrrgcn_layer(layer, cfg, g, h, prev_f, dt_f, prev_b, dt_b, act=None):
    """BiGRRGCNLayer.forward, models/BiRRGCN.py:27-47: h = GRU_f(x, dec(prev_f)) + GRU_b(x, dec(prev_b))."""
    y = rgcn_layer(h, g, layer['weight'], layer['loop_weight'], cfg['n_bases'], None, act)
    t1 = cfg.get('type1', False)
    hf = gru_stack(y, _decay(layer, cfg, prev_f, dt_f), layer['forward_rnn'], t1)
    hb = gru_stack(y, _decay(layer, cfg, prev_b, dt_b), layer['backward_rnn'], t1)
    return y, hf + hb


def bigrrgcn_layer_one_direction(layer, cfg, g, h, prev, dt, forward, act=None):
    """BiGRRGCNLayer.forward_one_direction, models/BiRRGCN.py:49-63."""
    y = rgcn_layer(h, g, layer['weight'], layer['loop_weight'], cfg['n_bases'], None, act)
    rnn = layer['forward_rnn'] if forward else layer['backward_rnn']
    return y, gru_stack(y, _decay(layer, cfg, prev, dt), rnn, cfg.get('type1', False))


def birrgcn_linear_layer(layer, cfg, g, h, prev_f, dt_f, prev_b, dt_b, act=None, one_direction=None):
    """BiRRGCNLayer.forward / forward_one_direction, models/BiRRGCN.py:114-162."""
    lam = cfg['inv_temperature']
    loop = torch.mm(h, layer['loop_weight'])
    out = rgcn_propagate(h, g, layer['weight'], cfg['n_bases'])
    if one_direction is None:
        out = out + torch.mm(prev_f * torch.exp(-dt_f.to(h.dtype).view(-1, 1) * lam), layer['time_weight_forward'])
        out = out + torch.mm(prev_b * torch.exp(-dt_b.to(h.dtype).view(-1, 1) * lam), layer['time_weight_backward'])
    else:
        w = layer['time_weight_forward'] if one_direction else layer['time_weight_backward']
        out = out + torch.mm(prev_f, w) * torch.exp(-dt_f.to(h.dtype).view(-1, 1) * lam)
    out = out + loop
    if act == 'relu':
        out = F.relu(out)
    return out


def birrgcn_forward(enc, cfg, g, h0, f1, f2, dt_f, b1, b2, dt_b, times=None, node_sizes=None, post=False):
    """BiRRGCN.forward / forward_post_ensemble, models/BiRRGCN.py:210-226, 259-277.
    Layer 2 has ReLU before the GRUs (:202-203).  Returns second_h
    (post: (local_h2, second_h))."""
    l1, l2 = enc['layer_1'], enc['layer_2']
    te = cfg.get('use_time_embedding', False)
    gru = cfg['module'] == 'BiGRRGCN'
    if cfg['rec_only_last_layer']:
        y1 = rgcn_layer(h0, g, l1['weight'], l1['loop_weight'], cfg['n_bases'])
    else:
        if gru:
            _, y1 = bigrrgcn_layer(l1, cfg, g, h0, f1, dt_f, b1, dt_b)
        else:
            y1 = birrgcn_linear_layer(l1, cfg, g, h0, f1, dt_f, b1, dt_b)
        if te:
            y1 = y1 + time_embedding_rows(l1['time_embed'], times, node_sizes)
    if gru:
        loc2, out = bigrrgcn_layer(l2, cfg, g, y1, f2, dt_f, b2, dt_b, act='relu')
    else:
        loc2, out = None, birrgcn_linear_layer(l2, cfg, g, y1, f2, dt_f, b2, dt_b, act='relu')
    if te:
        t2 = time_embedding_rows(l2['time_embed'], times, node_sizes)
        out = out + t2
        if loc2 is not None:
            loc2 = loc2 + t2
    return (loc2, out) if post else out


def birrgcn_forward_one_direction(enc, cfg, g, h0, p1, p2, dt, forward, times=None, node_sizes=None, post=False):
    """BiRRGCN.forward_one_direction / forward_post_ensemble_one_direction,
    models/BiRRGCN.py:228-240, 279-293.  Returns (first_h, second_h) (aliased for the GRU
    module, F7); post: (local_h2, first_h, second_h)."""
    l1, l2 = enc['layer_1'], enc['layer_2']
    te = cfg.get('use_time_embedding', False)
    gru = cfg['module'] == 'BiGRRGCN'
    if cfg['rec_only_last_layer']:
        y1 = rgcn_layer(h0, g, l1['weight'], l1['loop_weight'], cfg['n_bases'])
    else:
        if gru:
            _, y1 = bigrrgcn_layer_one_direction(l1, cfg, g, h0, p1, dt, forward)
        else:
            y1 = birrgcn_linear_layer(l1, cfg, g, h0, p1, dt, None, None, one_direction=forward)
        if te:
            y1 = y1 + time_embedding_rows(l1['time_embed'], times, node_sizes)
    if gru:
        loc2, out = bigrrgcn_layer_one_direction(l2, cfg, g, y1, p2, dt, forward, act='relu')
    else:
        loc2, out = None, birrgcn_linear_layer(l2, cfg, g, y1, p2, dt, None, None, act='relu', one_direction=forward)
    if te:
        t2 = time_embedding_rows(l2['time_embed'], times, node_sizes)
        out = out + t2
        if loc2 is not None:
            loc2 = loc2 + t2
    first = out if gru else y1
    return (loc2, first, out) if post else (first, out)


def birrgcn_isolated(enc, cfg, e, f1, f2, dt_f, b1, b2, dt_b, t=None, post=False):
    """BiRRGCN.forward_isolated, models/BiRRGCN.py:242-257 with BiGRRGCNLayer.forward_isolated
    :65-82 (layer 2: e + e@W_loop, ReLU, then both GRUs summed)."""
    l1, l2 = enc['layer_1'], enc['layer_2']
    te = cfg.get('use_time_embedding', False)
    gru = cfg['module'] == 'BiGRRGCN'
    t1 = cfg.get('type1', False)
    lam = cfg['inv_temperature']

    def lin_iso(layer, x, pf, pb, act):
        out = x + torch.mm(x, layer['loop_weight'])
        out = out + torch.mm(pf * torch.exp(-dt_f.to(x.dtype).view(-1, 1) * lam), layer['time_weight_forward'])
        out = out + torch.mm(pb * torch.exp(-dt_b.to(x.dtype).view(-1, 1) * lam), layer['time_weight_backward'])
        return F.relu(out) if act == 'relu' else out

    if cfg['rec_only_last_layer']:
        y1 = rgcn_layer_isolated(e, l1['loop_weight'])
    else:
        if gru:
            y1 = rgcn_layer_isolated(e, l1['loop_weight'])
            y1 = gru_stack(y1, _decay(l1, cfg, f1, dt_f), l1['forward_rnn'], t1) + \
                gru_stack(y1, _decay(l1, cfg, b1, dt_b), l1['backward_rnn'], t1)
        else:
            y1 = lin_iso(l1, e, f1, b1, None)
        if te:
            y1 = y1 + l1['time_embed'][int(t)]
    if gru:
        loc2 = rgcn_layer_isolated(y1, l2['loop_weight'], None, 'relu')
        out = gru_stack(loc2, _decay(l2, cfg, f2, dt_f), l2['forward_rnn'], t1) + \
            gru_stack(loc2, _decay(l2, cfg, b2, dt_b), l2['backward_rnn'], t1)
    else:
        loc2, out = None, lin_iso(l2, y1, f2, b2, 'relu')
    if te:
        out = out + l2['time_embed'][int(t)]
        if loc2 is not None:
            loc2 = loc2 + l2['time_embed'][int(t)]
    return (loc2, out) if post else out


# --------------------------------------------------------------------------------------
# a21: scorers (utils/scores.py:4-55) and the link-prediction loss
# --------------------------------------------------------------------------------------
def distmult(s, r, o, mode='single'):
    """utils/scores.py:4-11."""
    if mode == 'tail':
        return torch.sum((s * r).unsqueeze(1) * o, dim=-1)
    if mode == 'head':
        return torch.sum(s * (r * o).unsqueeze(1), dim=-1)
    return torch.sum(s * r * o, dim=-1)


def complex_score(head, relation, tail, mode='single'):
    """utils/scores.py:27-44."""
    re_h, im_h = torch.chunk(head, 2, dim=-1)
    re_r, im_r = torch.chunk(relation, 2, dim=-1)
    re_t, im_t = torch.chunk(tail, 2, dim=-1)
    if mode == 'tail':
        re_s = re_h * re_r - im_h * im_r
        im_s = re_h * im_r + im_h * re_r
        score = re_s.unsqueeze(1) * re_t + im_s.unsqueeze(1) * im_t
    elif mode == 'head':
        re_s = re_r * re_t + im_r * im_t
        im_s = re_r * im_t - im_r * re_t
        score = re_h * re_s.unsqueeze(1) + im_h * im_s.unsqueeze(1)
    else:
        re_s = re_h * re_r - im_h * im_r
        im_s = re_h * im_r + im_h * re_r
        score = re_s * re_t + im_s * im_t
    return score.sum(dim=-1)


def transE(head, relation, tail, mode='single'):
    """utils/scores.py:47-55."""
    if mode == 'tail':
        score = (head + relation).unsqueeze(1) - tail
    elif mode == 'head':
        score = head + (relation - tail).unsqueeze(1)
    else:
        score = head + relation - tail
    return -torch.norm(score, p=1, dim=-1)


SCORERS = {'distmult': distmult, 'complex': complex_score, 'transE': transE}


def train_link_prediction(score_fn, ent_embed, rel_embeds, triplets, neg_samples, all_embeds_g, corrupt_tail):
    """TKG_Module.train_link_prediction, models/TKG_Module.py:202-213 (labels are all 0)."""
    r = rel_embeds[triplets[:, 1]]
    if corrupt_tail:
        s = ent_embed[triplets[:, 0]]
        score = score_fn(s, r, all_embeds_g[neg_samples], mode='tail')
    else:
        o = ent_embed[triplets[:, 2]]
        score = score_fn(all_embeds_g[neg_samples], r, o, mode='head')
    labels = torch.zeros(score.shape[0], dtype=torch.int64)
    return F.cross_entropy(score, labels)


# --------------------------------------------------------------------------------------
# a15-a20: window models (dense-history bookkeeping kept exactly as the reference executes it)
# --------------------------------------------------------------------------------------
def get_batch_graph_list(t_list, seq_len, times):
    """TKG_Module.get_batch_graph_list, models/TKG_Module.py:232-250: windows sorted by target
    time descending, left-padded with None; returns time_batched_list[p][b]."""
    times = list(times)
    ts = sorted([int(t) for t in t_list], reverse=True)
    rows = []
    for tim in ts:
        length = times.index(tim) + 1
        seq = times[length - seq_len:length] if seq_len <= length else times[:length]
        rows.append([None] * (seq_len - len(seq)) + list(seq))
    return [list(x) for x in zip(*rows)]


def get_batch_graph_list_bi(t_list, seq_len, times):
    """BiDynamicRGCN.get_batch_graph_list, models/BiDynamicRGCN.py:17-49: forward windows
    (targets descending) and backward windows (targets ASCENDING, [t..t+L-1] reversed so the
    target is last), both left-padded with None."""
    times = list(times)
    fwd = get_batch_graph_list(t_list, seq_len, times)
    ts = sorted([int(t) for t in t_list])
    rows = []
    for tim in ts:
        k = times.index(tim)
        seq = times[k:k + seq_len] if seq_len <= len(times) - k else times[k:]
        seq = list(seq)
        seq.reverse()
        rows.append([None] * (seq_len - len(seq)) + seq)
    return fwd, [list(x) for x in zip(*rows)]


class DenseHistory:
    """hist_embeddings (bsz,2,N_ents,D) + start_time_tensor (bsz,N_ents) exactly as
    models/DynamicRGCN.py:35-54,160-161 keeps them (re-zeroed each step: SURVEY F8)."""

    def __init__(self, bsz, num_ents, dim, dtype):
        self.hist = torch.zeros(bsz, 2, num_ents, dim, dtype=dtype)
        self.start = torch.zeros(bsz, num_ents, dtype=dtype)
        self.bsz, self.num_ents, self.dim = bsz, num_ents, dim

    def get_prev(self, graphs, cur_t):
        """get_prev_embeddings, models/DynamicRGCN.py:35-45."""
        f, s, d = [], [], []
        for i, g in enumerate(graphs):
            f.append(self.hist[i][0][g.ids])
            s.append(self.hist[i][1][g.ids])
            d.append((cur_t - self.start[i][g.ids]).view(-1, 1))
        return torch.cat(f), torch.cat(s), torch.cat(d)

    def update(self, first_list, second_list, graphs, cur_t):
        """update_time_diff_hist_embeddings, models/DynamicRGCN.py:47-54."""
        res = self.hist.new_zeros(self.bsz, 2, self.num_ents, self.dim)
        for i in range(len(first_list)):
            idx = graphs[i].ids
            res[i][0][idx] = first_list[i]
            res[i][1][idx] = second_list[i]
            self.start[i][idx] = cur_t
        self.hist = res

    def flip(self):
        """models/BiDynamicRGCN.py:97-99."""
        self.hist = torch.flip(self.hist, [0])
        self.start = torch.flip(self.start, [0])


def _filter_none(xs):
    return [x for x in xs if x is not None]


def uni_pre_forward(model, cfg, graph_dict, time_batched_list, seq_len):
    """DynamicRGCN.pre_forward, models/DynamicRGCN.py:156-174 (full graphs; the
    --random-dropout / --edge-dropout history subsampling is not part of any BASELINE config)."""
    ent = model['ent_embeds']
    bsz = len(time_batched_list[0])
    H = DenseHistory(bsz, ent.shape[0], ent.shape[1], ent.dtype)
    for cur_t in range(seq_len - 1):
        ts = _filter_none(time_batched_list[cur_t])
        if len(ts) == 0:
            continue
        graphs = [graph_dict[t] for t in ts]
        sizes = [g.n for g in graphs]
        fp, sp, dt = H.get_prev(graphs, cur_t)
        bg = batch_graphs(graphs)
        first, second = rrgcn_forward(model['ent_encoder'], cfg, bg, ent[bg.ids], fp, sp, dt,
                                      time_batched_list[cur_t], sizes)
        H.update(first.split(sizes), second.split(sizes), graphs, cur_t)
    return H


def uni_target_embeds(model, cfg, H, target_graphs, target_times, seq_len):
    """DynamicRGCN.forward lines 181-184 (models/DynamicRGCN.py): encoder on the (possibly
    edge-subsampled, caller-provided) target graphs -> list of per-graph (n_b, D)."""
    ent = model['ent_embeds']
    fp, sp, dt = H.get_prev(target_graphs, seq_len - 1)
    sizes = [g.n for g in target_graphs]
    bg = batch_graphs(target_graphs)
    _, second = rrgcn_forward(model['ent_encoder'], cfg, bg, ent[bg.ids], fp, sp, dt, target_times, sizes)
    return list(second.split(sizes))


def uni_all_embeds(model, cfg, H, i, g, t, ent_embed, seq_len):
    """DynamicRGCN.get_all_embeds_Gt, models/DynamicRGCN.py:56-64 (+ time diff of :189):
    isolated pass over ALL entities, then rows of active nodes overwritten."""
    dt = (seq_len - 1 - H.start[i]).unsqueeze(-1)
    all_e = rrgcn_isolated(model['ent_encoder'], cfg, model['ent_embeds'], H.hist[i][0], H.hist[i][1], dt, t)
    return all_e.index_copy(0, g.ids, ent_embed)


def uni_forward_loss(model, cfg, graph_dict, t_list, times, seq_len, target_graphs, samples, score='complex'):
    """DynamicRGCN.forward, models/DynamicRGCN.py:176-194.  `target_graphs[b]` are the target
    snapshots already edge-subsampled (F13) and `samples[b]` = (triplets, neg_tail, neg_head)
    -- both injected because the reference draws them from unseeded np.random (F11)."""
    tbl = get_batch_graph_list(t_list, seq_len, times)
    H = uni_pre_forward(model, cfg, graph_dict, tbl, seq_len)
    per_graph = uni_target_embeds(model, cfg, H, target_graphs, tbl[-1], seq_len)
    fn = SCORERS[score]
    loss = 0
    for i, (t, g, emb) in enumerate(zip(tbl[-1], target_graphs, per_graph)):
        trip, neg_tail, neg_head = samples[i]
        all_e = uni_all_embeds(model, cfg, H, i, graph_dict[t], t, emb, seq_len)
        loss = loss + train_link_prediction(fn, emb, model['rel_embeds'], trip, neg_tail, all_e, True)
        loss = loss + train_link_prediction(fn, emb, model['rel_embeds'], trip, neg_head, all_e, False)
    return loss, per_graph


def bi_pre_forward(model, cfg, graph_dict, time_batched_list, seq_len, forward):
    """BiDynamicRGCN.pre_forward, models/BiDynamicRGCN.py:77-100."""
    ent = model['ent_embeds']
    bsz = len(time_batched_list[0])
    H = DenseHistory(bsz, ent.shape[0], ent.shape[1], ent.dtype)
    for cur_t in range(seq_len - 1):
        ts = _filter_none(time_batched_list[cur_t])
        if len(ts) == 0:
            continue
        graphs = [graph_dict[t] for t in ts]
        sizes = [g.n for g in graphs]
        fp, sp, dt = H.get_prev(graphs, cur_t)
        bg = batch_graphs(graphs)
        first, second = birrgcn_forward_one_direction(model['ent_encoder'], cfg, bg, ent[bg.ids], fp, sp, dt,
                                                      forward, time_batched_list[cur_t], sizes)
        H.update(first.split(sizes), second.split(sizes), graphs, cur_t)
    if not forward:
        H.flip()
    return H


def bi_target_embeds(model, cfg, Hf, Hb, target_graphs, target_times, seq_len):
    """BiDynamicRGCN.get_final_graph_embeds + get_graph_embeds_center,
    models/BiDynamicRGCN.py:114-121, 67-75."""
    ent = model['ent_embeds']
    f1, f2, dtf = Hf.get_prev(target_graphs, seq_len - 1)
    b1, b2, dtb = Hb.get_prev(target_graphs, seq_len - 1)
    sizes = [g.n for g in target_graphs]
    bg = batch_graphs(target_graphs)
    out = birrgcn_forward(model['ent_encoder'], cfg, bg, ent[bg.ids], f1, f2, dtf, b1, b2, dtb, target_times, sizes)
    return list(out.split(sizes))


def bi_all_embeds(model, cfg, Hf, Hb, i, g, t, ent_embed, seq_len):
    """BiDynamicRGCN.get_all_embeds_Gt, models/BiDynamicRGCN.py:102-112 (+ :134-135)."""
    dtf = (seq_len - 1 - Hf.start[i]).unsqueeze(-1)
    dtb = (seq_len - 1 - Hb.start[i]).unsqueeze(-1)
    all_e = birrgcn_isolated(model['ent_encoder'], cfg, model['ent_embeds'], Hf.hist[i][0], Hf.hist[i][1], dtf,
                             Hb.hist[i][0], Hb.hist[i][1], dtb, t)
    return all_e.index_copy(0, g.ids, ent_embed)


def bi_forward_loss(model, cfg, graph_dict, t_list, times, seq_len, target_graphs, samples, score='complex'):
    """BiDynamicRGCN.forward, models/BiDynamicRGCN.py:123-144 (targets/negatives injected)."""
    tf, tb = get_batch_graph_list_bi(t_list, seq_len, times)
    Hf = bi_pre_forward(model, cfg, graph_dict, tf, seq_len, True)
    Hb = bi_pre_forward(model, cfg, graph_dict, tb, seq_len, False)
    per_graph = bi_target_embeds(model, cfg, Hf, Hb, target_graphs, tf[-1], seq_len)
    fn = SCORERS[score]
    loss = 0
    for i, (t, g, emb) in enumerate(zip(tf[-1], target_graphs, per_graph)):
        trip, neg_tail, neg_head = samples[i]
        all_e = bi_all_embeds(model, cfg, Hf, Hb, i, graph_dict[t], t, emb, seq_len)
        loss = loss + train_link_prediction(fn, emb, model['rel_embeds'], trip, neg_tail, all_e, True)
        loss = loss + train_link_prediction(fn, emb, model['rel_embeds'], trip, neg_head, all_e, False)
    return loss, per_graph


# --------------------------------------------------------------------------------------
# a22: post-ensemble / impute window models (models/PostDynamicRGCN.py, models/PostBiDynamicRGCN.py)
# The encoder dict may carry the nn.Linear(1, 1) impute gates as (weight (1,1), bias (1,)) tuples:
#   'impute_weight' (uni, models/RRGCN.py:188-190) / 'impute_weight_forward', 'impute_weight_backward' (bi, models/BiRRGCN.py:204-207)
# --------------------------------------------------------------------------------------
class LocHistory(DenseHistory):
    """hist_embeddings_loc (bsz,N,D) next to the recurrent history, both re-zeroed every position
    (ImputeDynamicRGCN.update_time_diff_hist_embeddings, models/PostDynamicRGCN.py:33-42)."""

    def __init__(self, bsz, num_ents, dim, dtype):
        super().__init__(bsz, num_ents, dim, dtype)
        self.loc = torch.zeros(bsz, num_ents, dim, dtype=dtype)

    def update_loc(self, loc_list, first_list, second_list, graphs, cur_t):
        loc = self.loc.new_zeros(self.bsz, self.num_ents, self.dim)
        for i in range(len(loc_list)):
            loc[i][graphs[i].ids] = loc_list[i]
        self.loc = loc
        self.update(first_list, second_list, graphs, cur_t)

    def flip(self):
        super().flip()
        self.loc = torch.flip(self.loc, [0])


def impute_gate(lin, dt):
    """exp(-clamp(Linear(dt), min=0)), models/RRGCN.py:271-272 (the bi model halves it, models/BiRRGCN.py:301-302)."""
    w, b = lin
    return torch.exp(-torch.clamp(dt * w.view(1, 1) + b.view(1, 1), min=0))


def rrgcn_isolated_impute(enc, cfg, e, first_prev, second_prev, dt, t, pre_loc):
    """RRGCN.forward_isolated_impute, models/RRGCN.py:255-269 with GRRGCNLayer.forward_isolated_impute :105-116: the input of
    the layer-2 GRU is w * (last local state) + (1 - w) * (isolated layer-2 output)."""
    l1, l2 = enc['layer_1'], enc['layer_2']
    te = cfg.get('use_time_embedding', False)
    if cfg['rec_only_last_layer']:
        y1 = rgcn_layer_isolated(e, l1['loop_weight'])
    else:
        y1 = rgcn_layer_isolated(e, l1['loop_weight'])
        y1 = gru_stack(y1, _decay(l1, cfg, first_prev, dt), l1['rnn'], cfg.get('type1', False))
        if te:
            y1 = y1 + l1['time_embed'][int(t)]
    w = impute_gate(enc['impute_weight'], dt)
    x = rgcn_layer_isolated(y1, l2['loop_weight'])
    x = w * pre_loc + (1 - w) * x
    out = gru_stack(x, _decay(l2, cfg, second_prev, dt), l2['rnn'], cfg.get('type1', False))
    return out + l2['time_embed'][int(t)] if te else out


def birrgcn_isolated_impute(enc, cfg, e, f1, f2, dt_f, b1, b2, dt_b, t, f_loc, b_loc):
    """BiRRGCN.forward_isolated_impute, models/BiRRGCN.py:320-338 with BiGRRGCNLayer.forward_isolated_impute :84-100."""
    l1, l2 = enc['layer_1'], enc['layer_2']
    te = cfg.get('use_time_embedding', False)
    t1 = cfg.get('type1', False)
    y1 = rgcn_layer_isolated(e, l1['loop_weight'])
    if not cfg['rec_only_last_layer']:
        y1 = gru_stack(y1, _decay(l1, cfg, f1, dt_f), l1['forward_rnn'], t1) + gru_stack(y1, _decay(l1, cfg, b1, dt_b), l1['backward_rnn'], t1)
        if te:
            y1 = y1 + l1['time_embed'][int(t)]
    wf = impute_gate(enc['impute_weight_forward'], dt_f) / 2
    wb = impute_gate(enc['impute_weight_backward'], dt_b) / 2
    x = rgcn_layer_isolated(y1, l2['loop_weight'], None, 'relu')
    x = wf * f_loc + wb * b_loc + (1 - wf - wb) * x
    out = gru_stack(x, _decay(l2, cfg, f2, dt_f), l2['forward_rnn'], t1) + gru_stack(x, _decay(l2, cfg, b2, dt_b), l2['backward_rnn'], t1)
    return out + l2['time_embed'][int(t)] if te else out


def post_uni_pre_forward(model, cfg, graph_dict, time_batched_list, seq_len):
    """ImputeDynamicRGCN.pre_forward, models/PostDynamicRGCN.py:58-78 (full graphs)."""
    ent = model['ent_embeds']
    H = LocHistory(len(time_batched_list[0]), ent.shape[0], ent.shape[1], ent.dtype)
    for cur_t in range(seq_len - 1):
        ts = _filter_none(time_batched_list[cur_t])
        if len(ts) == 0:
            continue
        graphs = [graph_dict[t] for t in ts]
        sizes = [g.n for g in graphs]
        fp, sp, dt = H.get_prev(graphs, cur_t)
        bg = batch_graphs(graphs)
        loc, first, second = rrgcn_forward(model['ent_encoder'], cfg, bg, ent[bg.ids], fp, sp, dt, time_batched_list[cur_t], sizes, post=True)
        H.update_loc(loc.split(sizes), first.split(sizes), second.split(sizes), graphs, cur_t)
    return H


def post_bi_pre_forward(model, cfg, graph_dict, time_batched_list, seq_len, forward):
    """ImputeBiDynamicRGCN.pre_forward, models/PostBiDynamicRGCN.py:77-101."""
    ent = model['ent_embeds']
    H = LocHistory(len(time_batched_list[0]), ent.shape[0], ent.shape[1], ent.dtype)
    for cur_t in range(seq_len - 1):
        ts = _filter_none(time_batched_list[cur_t])
        if len(ts) == 0:
            continue
        graphs = [graph_dict[t] for t in ts]
        sizes = [g.n for g in graphs]
        fp, sp, dt = H.get_prev(graphs, cur_t)
        bg = batch_graphs(graphs)
        loc, first, second = birrgcn_forward_one_direction(model['ent_encoder'], cfg, bg, ent[bg.ids], fp, sp, dt, forward,
                                                           time_batched_list[cur_t], sizes, post=True)
        H.update_loc(loc.split(sizes), first.split(sizes), second.split(sizes), graphs, cur_t)
    if not forward:
        H.flip()
    return H


def post_bi_target_embeds(model, cfg, Hf, Hb, target_graphs, target_times, seq_len):
    """ImputeBiDynamicRGCN.get_final_graph_embeds -> get_graph_embeds_center, models/PostBiDynamicRGCN.py:53-75:
    (local, temporal) per-graph target embeddings."""
    ent = model['ent_embeds']
    f1, f2, dtf = Hf.get_prev(target_graphs, seq_len - 1)
    b1, b2, dtb = Hb.get_prev(target_graphs, seq_len - 1)
    sizes = [g.n for g in target_graphs]
    bg = batch_graphs(target_graphs)
    loc, out = birrgcn_forward(model['ent_encoder'], cfg, bg, ent[bg.ids], f1, f2, dtf, b1, b2, dtb, target_times, sizes, post=True)
    return list(loc.split(sizes)), list(out.split(sizes))


def post_bi_all_embeds(model, cfg, Hf, Hb, i, t, seq_len):
    """PostBiDynamicRGCN.get_all_embeds_Gt before the active rows are written over it, models/PostBiDynamicRGCN.py:182-186
    -> BiRRGCN.forward_post_ensemble_isolated (models/BiRRGCN.py:295-318; no impute mixing unless the gates exist)."""
    dtf = (seq_len - 1 - Hf.start[i]).unsqueeze(-1)
    dtb = (seq_len - 1 - Hb.start[i]).unsqueeze(-1)
    enc = model['ent_encoder']
    loc, rec = birrgcn_isolated(enc, cfg, model['ent_embeds'], Hf.hist[i][0], Hf.hist[i][1], dtf, Hb.hist[i][0], Hb.hist[i][1], dtb, t, post=True)
    if cfg.get('impute'):
        wf = impute_gate(enc['impute_weight_forward'], dtf) / 2
        wb = impute_gate(enc['impute_weight_backward'], dtb) / 2
        loc = wf * Hf.loc[i] + wb * Hb.loc[i] + (1 - wf - wb) * loc
    return loc, rec


def impute_bi_forward_loss(model, cfg, graph_dict, t_list, times, seq_len, target_graphs, samples, score='complex'):
    """ImputeBiDynamicRGCN.forward, models/PostBiDynamicRGCN.py:103-124."""
    tf, tb = get_batch_graph_list_bi(t_list, seq_len, times)
    Hf = post_bi_pre_forward(model, cfg, graph_dict, tf, seq_len, True)
    Hb = post_bi_pre_forward(model, cfg, graph_dict, tb, seq_len, False)
    _, per_graph = post_bi_target_embeds(model, cfg, Hf, Hb, target_graphs, tf[-1], seq_len)
    fn = SCORERS[score]
    loss = 0
    for i, (t, emb) in enumerate(zip(tf[-1], per_graph)):
        trip, neg_tail, neg_head = samples[i]
        dtf = (seq_len - 1 - Hf.start[i]).unsqueeze(-1)
        dtb = (seq_len - 1 - Hb.start[i]).unsqueeze(-1)
        all_e = birrgcn_isolated_impute(model['ent_encoder'], cfg, model['ent_embeds'], Hf.hist[i][0], Hf.hist[i][1], dtf,
                                        Hb.hist[i][0], Hb.hist[i][1], dtb, t, Hf.loc[i], Hb.loc[i])
        all_e = all_e.index_copy(0, torch.as_tensor(graph_dict[t].ids), emb)
        loss = loss + train_link_prediction(fn, emb, model['rel_embeds'], trip, neg_tail, all_e, True)
        loss = loss + train_link_prediction(fn, emb, model['rel_embeds'], trip, neg_head, all_e, False)
    return loss, per_graph


def impute_uni_forward_loss(model, cfg, graph_dict, t_list, times, seq_len, target_graphs, samples, score='complex'):
    """ImputeDynamicRGCN.forward, models/PostDynamicRGCN.py:80-96."""
    tbl = get_batch_graph_list(t_list, seq_len, times)
    H = post_uni_pre_forward(model, cfg, graph_dict, tbl, seq_len)
    ent = model['ent_embeds']
    fp, sp, dt = H.get_prev(target_graphs, seq_len - 1)
    sizes = [g.n for g in target_graphs]
    bg = batch_graphs(target_graphs)
    _, _, second = rrgcn_forward(model['ent_encoder'], cfg, bg, ent[bg.ids], fp, sp, dt, tbl[-1], sizes, post=True)
    per_graph = list(second.split(sizes))
    fn = SCORERS[score]
    loss = 0
    for i, (t, emb) in enumerate(zip(tbl[-1], per_graph)):
        trip, neg_tail, neg_head = samples[i]
        dti = (seq_len - 1 - H.start[i]).unsqueeze(-1)
        all_e = rrgcn_isolated_impute(model['ent_encoder'], cfg, ent, H.hist[i][0], H.hist[i][1], dti, t, H.loc[i])
        all_e = all_e.index_copy(0, torch.as_tensor(graph_dict[t].ids), emb)
        loss = loss + train_link_prediction(fn, emb, model['rel_embeds'], trip, neg_tail, all_e, True)
        loss = loss + train_link_prediction(fn, emb, model['rel_embeds'], trip, neg_head, all_e, False)
    return loss, per_graph


def static_forward_embeds(model, cfg, target_graphs, target_times):
    """StaticRGCN.get_per_graph_ent_embeds, baselines/StaticRGCN.py:60-89 (targets injected)."""
    ent = model['ent_embeds']
    sizes = [g.n for g in target_graphs]
    bg = batch_graphs(target_graphs)
    out = static_rgcn_forward(model['ent_encoder'], cfg, bg, ent[bg.ids], target_times, sizes)
    return list(out.split(sizes))


def static_forward_loss(model, cfg, graph_dict, t_list, target_graphs, samples, score='complex'):
    """StaticRGCN.forward, baselines/StaticRGCN.py:36-58: per target graph the encoder rows, the isolated pass over ALL
    entities with the graph's rows written over it, and loss_tail + loss_head (t_list order is kept: no window sorting)."""
    fn = SCORERS[score]
    per_graph = static_forward_embeds(model, cfg, target_graphs, t_list)
    loss = 0
    for i, (t, g, emb) in enumerate(zip(t_list, target_graphs, per_graph)):
        trip, neg_tail, neg_head = samples[i]
        all_e = static_rgcn_isolated(model['ent_encoder'], cfg, model['ent_embeds'], t).index_copy(0, torch.as_tensor(g.ids), emb)
        loss = loss + train_link_prediction(fn, emb, model['rel_embeds'], trip, neg_tail, all_e, True)
        loss = loss + train_link_prediction(fn, emb, model['rel_embeds'], trip, neg_head, all_e, False)
    return loss, per_graph


def filtered_ranks(score_fn, ent_embed, rel_embeds, all_embeds, triples, gids, true_tails, true_heads, batch=100):
    """EvaluationFilter.calc_metrics_single_graph, utils/evaluation.py:34-106: for every triple (local ids) score the true
    subject / object against ALL entities, overwrite the OTHER known-true entities with -10e6 (the target itself stays),
    sigmoid, sort descending (torch.sort: order inside a tie group is whatever it returns), rank = position of the target,
    1-indexed.  -> [subject-corruption ranks ; object-corruption ranks].
    true_tails[(h, r)] / true_heads[(r, t)]: local node ids known true at this timestamp over train+valid+test."""
    gids = torch.as_tensor(gids)
    P, N = triples.shape[0], all_embeds.shape[0]
    out = {}
    for mode in ('tail', 'head'):
        mask = torch.zeros(P, N, dtype=torch.bool)
        for i in range(P):
            h, r, t = (int(x) for x in triples[i])
            if mode == 'tail':
                mask[i, gids[torch.as_tensor(sorted(true_tails[(h, r)]))]] = True
                mask[i, gids[t]] = False
            else:
                mask[i, gids[torch.as_tensor(sorted(true_heads[(r, t)]))]] = True
                mask[i, gids[h]] = False
        ranks = []
        for a in range(0, P, batch):
            b = min(P, a + batch)
            r = rel_embeds[triples[a:b, 1]]
            if mode == 'tail':
                sc = score_fn(ent_embed[triples[a:b, 0]], r, all_embeds, mode='tail')
                target = gids[triples[a:b, 2]]
            else:
                sc = score_fn(all_embeds, r, ent_embed[triples[a:b, 2]], mode='head')
                target = gids[triples[a:b, 0]]
            ranks.append(rank_from_scores(sc, mask[a:b], target))
        out[mode] = torch.cat(ranks)
    return torch.cat([out['head'], out['tail']])


def rank_from_scores(score, mask, target):
    """perturb_and_get_rank + sort_and_rank, utils/evaluation.py:75-78,101-106: masked entries -> -10e6, sigmoid, sort
    descending, 1-indexed position of the target."""
    sc = torch.sigmoid(torch.where(mask, torch.full_like(score, -10e6), score))
    _, idx = torch.sort(sc, dim=1, descending=True)
    return torch.nonzero(idx == target.view(-1, 1))[:, 1].view(-1) + 1


def true_heads_and_tails(triple_sets):
    """CorruptTriples.get_true_head_and_tail_per_graph over the concatenated train/valid/test triples of one timestamp
    (utils/evaluation.py:16-32): {(h, r): {t}}, {(r, t): {h}} in LOCAL node ids."""
    tails, heads = {}, {}
    for trip in triple_sets:
        for h, r, t in trip.tolist():
            tails.setdefault((h, r), set()).add(t)
            heads.setdefault((r, t), set()).add(h)
    return heads, tails


# --------------------------------------------------------------------------------------
# Parameter init + state_dict conversion helpers
# --------------------------------------------------------------------------------------
def _xavier(rng, rows, cols, gain=math.sqrt(2.0)):
    a = gain * math.sqrt(6.0 / (rows + cols))
    return torch.from_numpy(rng.uniform(-a, a, size=(rows, cols)).astype(np.float32))


def _gru_params(rng, D, layers=1):
    k = 1.0 / math.sqrt(D)
    u = lambda *s: torch.from_numpy(rng.uniform(-k, k, size=s).astype(np.float32))
    return [dict(w_ih=u(3 * D, D), w_hh=u(3 * D, D), b_ih=u(3 * D), b_hh=u(3 * D)) for _ in range(layers)]


def init_model(cfg, num_ents, num_rels, num_times, D, seed=1, bias=False):
    """Random parameters with the reference's shapes/initialisers (models/DynamicRGCN.py:21-30,
    models/RGCN.py:15-38, nn.GRU default)."""
    rng = np.random.default_rng(seed)
    B = cfg['n_bases']
    s = D // B
    mod = cfg['module']

    def layer(recurrent):
        d = dict(weight=_xavier(rng, 2 * num_rels, B * s * s), loop_weight=_xavier(rng, D, D),
                 time_embed=_xavier(rng, num_times, D), h_bias=(torch.zeros(D) if bias else None))
        if recurrent:
            if mod == 'GRRGCN':
                d['rnn'] = _gru_params(rng, D)
            elif mod == 'BiGRRGCN':
                d['forward_rnn'] = _gru_params(rng, D)
                d['backward_rnn'] = _gru_params(rng, D)
            elif mod == 'RRGCN':
                d['time_weight'] = _xavier(rng, D, D)
            elif mod == 'BiRRGCN':
                d['time_weight_forward'] = _xavier(rng, D, D)
                d['time_weight_backward'] = _xavier(rng, D, D)
            elif mod in ('SARGCN', 'BiSARGCN'):
                k = 1.0 / math.sqrt(D)
                for nm in ('q_linear', 'v_linear', 'k_linear'):
                    d[nm] = torch.from_numpy(rng.uniform(-k, k, size=(D, D)).astype(np.float32))
        if mod in ('SARGCN', 'BiSARGCN'):
            d['h_bias'] = torch.from_numpy(rng.uniform(-0.3, 0.3, D).astype(np.float32))
        return d

    rec1 = (mod != 'SRGCN') and not cfg.get('rec_only_last_layer', False)
    return dict(ent_embeds=_xavier(rng, num_ents, D), rel_embeds=_xavier(rng, 2 * num_rels, D),
                ent_encoder=dict(layer_1=layer(rec1), layer_2=layer(mod != 'SRGCN')))


def model_from_state_dict(sd, cfg):
    """Build the oracle's parameter dict from a reference state_dict (SURVEY Appendix B keys)."""
    def rnn(prefix, type1):
        if type1:
            return [dict(w_ih=sd[prefix + 'weight_ih'], w_hh=sd[prefix + 'weight_hh'],
                         b_ih=sd[prefix + 'bias_ih'], b_hh=sd[prefix + 'bias_hh'])]
        out, k = [], 0
        while prefix + 'weight_ih_l%d' % k in sd:
            out.append(dict(w_ih=sd[prefix + 'weight_ih_l%d' % k], w_hh=sd[prefix + 'weight_hh_l%d' % k],
                            b_ih=sd[prefix + 'bias_ih_l%d' % k], b_hh=sd[prefix + 'bias_hh_l%d' % k]))
            k += 1
        return out

    enc = {}
    for ln in ('layer_1', 'layer_2'):
        p = 'ent_encoder.%s.' % ln
        d = dict(weight=sd[p + 'weight'], loop_weight=sd[p + 'loop_weight'], time_embed=sd[p + 'time_embed'],
                 h_bias=sd.get(p + 'h_bias'))
        t1 = cfg.get('type1', False)
        for name in ('rnn', 'forward_rnn', 'backward_rnn'):
            if any(k.startswith(p + name + '.') for k in sd):
                d[name] = rnn(p + name + '.', t1)
        for name in ('time_weight', 'time_weight_forward', 'time_weight_backward'):
            if p + name in sd:
                d[name] = sd[p + name]
        for name in ('q_linear', 'k_linear', 'v_linear'):
            if p + name + '.weight' in sd:
                d[name] = sd[p + name + '.weight']
        if p + 'exponential_decay.weight' in sd:
            d['exponential_decay'] = (sd[p + 'exponential_decay.weight'], sd[p + 'exponential_decay.bias'])
        enc[ln] = d
    return dict(ent_embeds=sd['ent_embeds'], rel_embeds=sd['rel_embeds'], ent_encoder=enc)


def map_params(model, fn):
    """Apply `fn` to every tensor of a (nested) parameter dict."""
    if isinstance(model, torch.Tensor):
        return fn(model)
    if isinstance(model, dict):
        return {k: map_params(v, fn) for k, v in model.items()}
    if isinstance(model, (list, tuple)):
        return type(model)(map_params(v, fn) for v in model)
    return model


def leaf_tensors(model, prefix=''):
    """Flatten a parameter dict to {dotted_name: tensor}."""
    out = {}
    if isinstance(model, torch.Tensor):
        out[prefix.rstrip('.')] = model
    elif isinstance(model, dict):
        for k, v in model.items():
            out.update(leaf_tensors(v, prefix + str(k) + '.'))
    elif isinstance(model, (list, tuple)):
        for i, v in enumerate(model):
            out.update(leaf_tensors(v, prefix + str(i) + '.'))
    return out


# --------------------------------------------------------------------------------------
# a23: self-attention encoder (models/SARGCN.py, models/SelfAttentionRGCN.py,
#      models/BiSelfAttentionRGCN.py) -- config 5
# Layer dict gains: q_linear, k_linear, v_linear (D,D) (nn.Linear weights, no bias).
# --------------------------------------------------------------------------------------
def sa_attention(layer, cfg, cur, prev, time_diff, mask, heads=8):
    """SARGCNLayer.calc_result + attention, models/SARGCN.py:25-53: q from the current state, K/V
    from [history..., current] (T positions), `heads` heads of d_k = D // heads,
    softmax(q.K^T / sqrt(d_k) + mask + decay) . V.   cur (n,D), prev (n,T-1,D), mask (n,T)."""
    n, D = cur.shape
    d_k = D // heads
    if cfg.get('learnable_lambda'):
        w, b = layer['exponential_decay']
        decay = -torch.clamp(time_diff.view(-1, 1) * w.view(1, 1) + b.view(1, 1), min=0).view(-1)
    else:
        decay = 0
    all_t = torch.cat([prev, cur.unsqueeze(1)], dim=1)
    q = torch.mm(cur, layer['q_linear'].t()).view(n, 1, heads, d_k).transpose(1, 2)
    k = torch.matmul(all_t, layer['k_linear'].t()).view(n, -1, heads, d_k).transpose(1, 2)
    v = torch.matmul(all_t, layer['v_linear'].t()).view(n, -1, heads, d_k).transpose(1, 2)
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(d_k)                 # (n,h,1,T)
    p = F.softmax(scores.squeeze(2) + mask.unsqueeze(1) + decay, dim=-1)          # (n,h,T)
    out = torch.matmul(p.unsqueeze(2), v).squeeze(2)                                # (n,h,d_k)
    # models/SARGCN.py:37: the squeeze() in `attention` dropped the length-1 query axis, so the
    # reference's transpose(1, 2) swaps (heads, d_k): output feature index = d * heads + head.
    return out.transpose(1, 2).contiguous().view(n, D)


def sargcn_forward(enc, cfg, g, h0, times, node_sizes):
    """SARGCN.forward, models/SARGCN.py:103-107: plain 2-layer RGCN (bias, L2 ReLU); the time
    embeddings are added to the RETURNED states only (layer 2 consumes layer 1's plain output)."""
    l1, l2 = enc['layer_1'], enc['layer_2']
    y1 = rgcn_layer(h0, g, l1['weight'], l1['loop_weight'], cfg['n_bases'], l1.get('h_bias'), None)
    y2 = rgcn_layer(y1, g, l2['weight'], l2['loop_weight'], cfg['n_bases'], l2.get('h_bias'), 'relu')
    return y1 + time_embedding_rows(l1['time_embed'], times, node_sizes), y2 + time_embedding_rows(l2['time_embed'], times, node_sizes), y1


def sargcn_forward_final(enc, cfg, g, h0, prev1, prev2, time_diff, mask, times, node_sizes):
    """SARGCN.forward_final, models/SARGCN.py:109-117."""
    l1, l2 = enc['layer_1'], enc['layer_2']
    f, s, y1 = sargcn_forward(enc, cfg, g, h0, times, node_sizes)
    second = sa_attention(l2, cfg, s, prev2, time_diff, mask)
    if cfg['rec_only_last_layer']:
        return second
    first = sa_attention(l1, cfg, f, prev1, time_diff, mask)
    return torch.max(torch.stack([first, second], dim=-1), dim=-1)[0]


def sargcn_isolated(enc, cfg, e, prev1, prev2, time_diff, mask, t):
    """SARGCN.forward_isolated, models/SARGCN.py:119-125 with SARGCNLayer.forward_isolated :55-62."""
    l1, l2 = enc['layer_1'], enc['layer_2']
    y1 = rgcn_layer_isolated(e, l1['loop_weight'], l1.get('h_bias'), None)
    if cfg['rec_only_last_layer']:
        first = y1
    else:
        first = sa_attention(l1, cfg, y1 + l1['time_embed'][int(t)], prev1, time_diff, mask)
    y2 = rgcn_layer_isolated(first, l2['loop_weight'], l2.get('h_bias'), 'relu')
    second = sa_attention(l2, cfg, y2 + l2['time_embed'][int(t)], prev2, time_diff, mask)
    return second if cfg['rec_only_last_layer'] else torch.max(torch.stack([first, second], dim=-1), dim=-1)[0]


def sa_pre_forward(model, cfg, graph_dict, time_batched_list, seq_len, with_current):
    """SelfAttentionRGCN.pre_forward (models/SelfAttentionRGCN.py:97-120) / the Bi variant
    (models/BiSelfAttentionRGCN.py:25-46): dense hist (L-1,bsz,2,N,D) and additive mask
    (L or L-1, bsz, N) = -10e9 except where a node was active."""
    ent = model['ent_embeds']
    bsz = len(time_batched_list[0])
    N, D = ent.shape
    hist = torch.zeros(seq_len - 1, bsz, 2, N, D, dtype=ent.dtype)
    mask = torch.zeros(seq_len if with_current else seq_len - 1, bsz, N, dtype=ent.dtype) - 10e9
    if with_current:
        mask[-1] = 0
    for cur_t in range(seq_len - 1):
        ts = _filter_none(time_batched_list[cur_t])
        if not ts:
            continue
        graphs = [graph_dict[t] for t in ts]
        sizes = [g.n for g in graphs]
        bg = batch_graphs(graphs)
        f, s, _ = sargcn_forward(model['ent_encoder'], cfg, bg, ent[bg.ids], ts, sizes)
        for i, (fi, si) in enumerate(zip(f.split(sizes), s.split(sizes))):
            idx = graphs[i].ids
            mask[cur_t][i][idx] = 0
            hist[cur_t][i][0][idx] = fi
            hist[cur_t][i][1][idx] = si
    return hist, mask


def sa_encode(model, cfg, graph_dict, t_list, times, seq_len, target_graphs, bi=False):
    """History passes + the attention over them for the target graphs (the part of
    SelfAttentionRGCN.forward / BiSelfAttentionRGCN.forward before the losses,
    models/SelfAttentionRGCN.py:122-129, models/BiSelfAttentionRGCN.py:48-59).
    -> (per-graph target embeddings, target times, hist, mask, time_diff)."""
    ent = model['ent_embeds']
    if not bi:
        tbl = get_batch_graph_list(t_list, seq_len, times)
        hist, mask = sa_pre_forward(model, cfg, graph_dict, tbl, seq_len, True)
        td = torch.arange(seq_len - 1, -1, -1, dtype=ent.dtype)
        target_times = tbl[-1]
    else:
        tf, tb = get_batch_graph_list_bi(t_list, seq_len, times)
        hf, mf = sa_pre_forward(model, cfg, graph_dict, tf, seq_len, False)
        hb, mb = sa_pre_forward(model, cfg, graph_dict, tb, seq_len, False)
        hb, mb = torch.flip(hb, [1]), torch.flip(mb, [1])
        hist = torch.cat([hf, hb], dim=0)
        mask = torch.cat([mf, mb, mf.new_zeros(1, *mf.shape[1:])], dim=0)
        r = list(range(seq_len - 1, 0, -1))
        td = torch.tensor(r + r + [0.], dtype=ent.dtype)
        target_times = tf[-1]
    sizes = [g.n for g in target_graphs]
    bg = batch_graphs(target_graphs)
    p1 = torch.cat([hist[:, i, 0][:, g.ids] for i, g in enumerate(target_graphs)], dim=1).transpose(0, 1)
    p2 = torch.cat([hist[:, i, 1][:, g.ids] for i, g in enumerate(target_graphs)], dim=1).transpose(0, 1)
    lm = torch.cat([mask[:, i][:, g.ids] for i, g in enumerate(target_graphs)], dim=1).transpose(0, 1)
    out = sargcn_forward_final(model['ent_encoder'], cfg, bg, ent[bg.ids], p1, p2, td, lm, target_times, sizes)
    return list(out.split(sizes)), target_times, hist, mask, td


def sa_all_embeds(model, cfg, graph_dict, i, t, emb, hist, mask, td):
    """SelfAttentionRGCN.get_all_embeds_Gt, models/SelfAttentionRGCN.py:28-45."""
    all_e = sargcn_isolated(model['ent_encoder'], cfg, model['ent_embeds'], hist[:, i, 0].transpose(0, 1),
                            hist[:, i, 1].transpose(0, 1), td, mask[:, i].transpose(0, 1), t)
    return all_e.index_copy(0, graph_dict[t].ids, emb)


def sa_forward_loss(model, cfg, graph_dict, t_list, times, seq_len, target_graphs, samples, bi=False, score='complex'):
    """SelfAttentionRGCN.forward / BiSelfAttentionRGCN.forward (models/SelfAttentionRGCN.py:122-140,
    models/BiSelfAttentionRGCN.py:48-69), targets / negatives injected."""
    per_graph, target_times, hist, mask, td = sa_encode(model, cfg, graph_dict, t_list, times, seq_len, target_graphs, bi)
    fn = SCORERS[score]
    loss = 0
    for i, (t, emb) in enumerate(zip(target_times, per_graph)):
        trip, neg_tail, neg_head = samples[i]
        all_e = sa_all_embeds(model, cfg, graph_dict, i, t, emb, hist, mask, td)
        loss = loss + train_link_prediction(fn, emb, model['rel_embeds'], trip, neg_tail, all_e, True)
        loss = loss + train_link_prediction(fn, emb, model['rel_embeds'], trip, neg_head, all_e, False)
    return loss, per_graph
