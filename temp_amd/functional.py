"""torch.autograd wrappers around the HIP kernels (through temp_amd.backend).

Each Function is one reference op with its hand-written backward:
  rgcn_layer      RGCNLayer.forward            models/RGCN.py:53-104
  rgcn_isolated   RGCNLayer.forward_isolated   models/RGCN.py:78-89
  gru_step        decay + single GRU step      models/RRGCN.py:79-85, models/GRU_cell.py:18-30
  gather_rows     ent_embeds[id] / history gather   models/DynamicRGCN.py:41-43,93
  linear          nn.Linear without bias (q/k/v projections)   models/SARGCN.py:16-18,32-34
  history_attention   SARGCNLayer.attention over the active history rows   models/SARGCN.py:25-53
"""
import numpy as np
import torch

from . import _lib
from .backend import get_backend

ACTS = {None: _lib.ACT_NONE, "relu": _lib.ACT_RELU}


class _RGCNLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, weight, loop_w, bias, dg, num_bases, act, drop, grad_premasked, dest):
        be = get_backend()
        out = be.rgcn_fwd(dg, h, None, weight, loop_w, bias, num_bases, act, drop, out=dest)
        ctx.save_for_backward(h, weight, loop_w, out)
        ctx.dg, ctx.num_bases, ctx.has_bias, ctx.drop = dg, num_bases, bias is not None, drop
        # grad_premasked: the ONLY consumer of `out` folds the activation's adjoint into the gradient it returns
        # (gather_rows(..., relu_table=True)), so the backward kernels take d_out as the pre-activation gradient
        ctx.act = _lib.ACT_NONE if grad_premasked else act
        return out

    @staticmethod
    def backward(ctx, d_out):
        h, weight, loop_w, out = ctx.saved_tensors
        d_h, d_w, d_loop, d_bias = get_backend().rgcn_bwd(ctx.dg, h, out, d_out.contiguous(), weight, loop_w, ctx.has_bias,
                                                          ctx.num_bases, ctx.act, ctx.drop)
        return d_h, d_w, d_loop, d_bias, None, None, None, None, None, None


def rgcn_layer(h, dg, weight, loop_w, bias, num_bases, act=None, drop=None, grad_premasked=False, out=None):
    """out = act(nnorm^2 * sum_in h_u BD(W_r) [+bias] + dropout(h W_loop)) on a device graph `dg`.
    drop = (p, seed) or None: dropout of the self-loop message (models/RGCN.py:57-59), mask = hash(seed, row, col).
    out: optional caller-owned contiguous (n_nodes, d_out) destination that no autograd graph knows (e.g. this rank's row range
    of the exchange buffer of temp_amd.dist): the layer writes its result there instead of allocating."""
    return _RGCNLayerFn.apply(h, weight, loop_w, bias, dg, num_bases, ACTS[act], drop, bool(grad_premasked and act == "relu"),
                              None if out is None else out.detach())


class _RGCNTableLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, weight, loop_w, bias, ids, inverse, dg, num_bases, act, drop):
        out = get_backend().rgcn_table_fwd(dg, table, ids, weight, loop_w, bias, num_bases, act, drop)
        ctx.save_for_backward(table, weight, loop_w, out, ids)
        ctx.dg, ctx.num_bases, ctx.act, ctx.has_bias, ctx.inverse, ctx.drop = dg, num_bases, act, bias is not None, inverse, drop
        return out

    @staticmethod
    def backward(ctx, d_out):
        table, weight, loop_w, out, ids = ctx.saved_tensors
        d_t, d_w, d_loop, d_bias = get_backend().rgcn_table_bwd(ctx.dg, table, ids, ctx.inverse, out, d_out.contiguous(), weight, loop_w,
                                                               ctx.has_bias, ctx.num_bases, ctx.act, ctx.drop)
        return d_t, d_w, d_loop, d_bias, None, None, None, None, None, None


def rgcn_layer_table(table, ids, inverse, dg, weight, loop_w, bias, num_bases, act=None, drop=None):
    """rgcn_layer on h = table[ids] (ids int32, static; inverse = gather_inverse(ids, rows)) without materialising h:
    the self-loop product and its gradients run over the table's rows, not over every node row."""
    return _RGCNTableLayerFn.apply(table, weight, loop_w, bias, ids, inverse, dg, num_bases, ACTS[act], drop)


class _RGCNIsolatedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e, loop_w, bias, act, drop):
        out = get_backend().rgcn_isolated_fwd(e, loop_w, bias, act, drop)
        ctx.save_for_backward(e, loop_w, out)
        ctx.act, ctx.has_bias, ctx.drop = act, bias is not None, drop
        return out

    @staticmethod
    def backward(ctx, d_out):
        e, loop_w, out = ctx.saved_tensors
        d_e, d_loop, d_bias = get_backend().rgcn_isolated_bwd(e, out, d_out.contiguous(), loop_w, ctx.has_bias, ctx.act, ctx.drop)
        return d_e, d_loop, d_bias, None, None


def rgcn_isolated(e, loop_w, bias, act=None, drop=None):
    """out = act(e + dropout(e W_loop) [+bias])."""
    return _RGCNIsolatedFn.apply(e, loop_w, bias, ACTS[act], drop)


class _GRUStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, prev, dt, w_ih, w_hh, b_ih, b_hh, decay_w, decay_b, prev_idx, lam, variant, prev_inv=None):
        be = get_backend()
        wb = None
        if decay_w is not None:
            wb = torch.cat([decay_w.detach().reshape(1), decay_b.detach().reshape(1)]).contiguous()
        dt = dt.reshape(-1).contiguous()
        h_out, saved = be.gru_fwd(x, prev, prev_idx, dt, lam, wb, w_ih, w_hh, b_ih, b_hh, variant)
        ctx.save_for_backward(x, prev, dt, w_ih, w_hh, saved, wb if wb is not None else x.new_zeros(0))
        ctx.prev_idx, ctx.lam, ctx.variant, ctx.learn = prev_idx, lam, variant, wb is not None
        ctx.prev_inv = prev_inv
        ctx.wshape = decay_w.shape if decay_w is not None else None
        ctx.bshape = decay_b.shape if decay_b is not None else None
        return h_out

    @staticmethod
    def backward(ctx, d_h):
        x, prev, dt, w_ih, w_hh, saved, wb = ctx.saved_tensors
        be = get_backend()
        d_x, d_prev_rows, d_w_ih, d_w_hh, d_b_ih, d_b_hh, d_wb = be.gru_bwd(
            x, prev, ctx.prev_idx, dt, ctx.lam, wb if ctx.learn else None, w_ih, w_hh, saved, d_h.contiguous(), ctx.variant)
        if ctx.prev_idx is None:
            d_prev = d_prev_rows
        elif ctx.prev_inv is not None:                        # injective row map: the adjoint of the gather is a gather through the
            d_prev = be.gather_rows(d_prev_rows, ctx.prev_inv)   # inverse map (-1 -> zero row): one pass, no atomics, no zero fill
        else:
            d_prev = torch.zeros_like(prev)
            be.scatter_add_rows(d_prev_rows, ctx.prev_idx, d_prev)
        d_dw = d_wb[0].reshape(ctx.wshape) if ctx.learn else None
        d_db = d_wb[1].reshape(ctx.bshape) if ctx.learn else None
        return d_x, d_prev, None, d_w_ih, d_w_hh, d_b_ih, d_b_hh, d_dw, d_db, None, None, None, None


def gru_step(x, prev, dt, w_ih, w_hh, b_ih, b_hh, lam, decay=None, prev_idx=None, type1=False, prev_inv=None):
    """h' = GRU(x, prev[prev_idx] * exp(-lam*dt))  (learnable decay when `decay` = (weight, bias)).
    prev_idx: optional int32 tensor, -1 => zero previous state.
    prev_inv: optional int32 tensor [rows of prev], the inverse of an INJECTIVE prev_idx (prev_inv[prev_idx[i]] = i, -1 for rows
    nobody continues from; injective_inverse()): the gradient of `prev` is then one gather instead of a zero fill + atomic scatter."""
    dw, db = decay if decay is not None else (None, None)
    return _GRUStepFn.apply(x, prev, dt, w_ih, w_hh, b_ih, b_hh, dw, db, prev_idx, float(lam),
                            _lib.GRU_TYPE1 if type1 else _lib.GRU_TORCH, prev_inv)


def injective_inverse(idx_np, n_rows, device):
    """int32 device tensor inv [n_rows] with inv[idx[i]] = i (-1 where no i maps) for a host index list whose non-negative
    entries are pairwise distinct, or None when they are not (the caller then keeps the scatter-add adjoint)."""
    idx = np.asarray(idx_np).reshape(-1)
    ok = idx >= 0
    inv = np.full(int(n_rows), -1, dtype=np.int32)
    inv[idx[ok]] = np.nonzero(ok)[0].astype(np.int32)
    if int((inv >= 0).sum()) != int(ok.sum()):
        return None
    return _lib.to_device(inv, device)


class _GatherRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, idx, inverse, relu_table):
        ctx.rows = table.shape[0]
        ctx.inverse = inverse
        ctx.relu_table = relu_table
        ctx.save_for_backward(idx, table.detach()) if relu_table else ctx.save_for_backward(idx)
        return get_backend().gather_rows(table, idx)

    @staticmethod
    def backward(ctx, d_out):
        idx = ctx.saved_tensors[0]
        be = get_backend()
        if ctx.relu_table:                                # (only offered with an inverse: see gather_rows)
            seg_ptr, order = ctx.inverse
            return be.segment_sum_rows(d_out.contiguous(), seg_ptr, order, ctx.rows, relu_of=ctx.saved_tensors[1]), None, None, None
        if ctx.inverse is not None and d_out.shape[1] % 4 == 0 and d_out.shape[1] <= 256:
            seg_ptr, order = ctx.inverse
            return be.segment_sum_rows(d_out.contiguous(), seg_ptr, order, ctx.rows), None, None, None
        d_table = torch.zeros(ctx.rows, d_out.shape[1], dtype=d_out.dtype, device=d_out.device)
        be.scatter_add_rows(d_out.contiguous(), idx, d_table)
        return d_table, None, None, None


class _GatherRowsKeysFn(torch.autograd.Function):
    """_GatherRowsFn that also returns the magnitude keys of its output (row keys [n], column keys [d]; int32, not differentiable):
    the f16 products that read the gathered rows scale them by these (include/temp_amd.h: temp_gather_rows_keys)."""

    @staticmethod
    def forward(ctx, table, idx, inverse, relu_table):
        ctx.rows = table.shape[0]
        ctx.inverse = inverse
        ctx.relu_table = relu_table
        ctx.save_for_backward(idx, table.detach()) if relu_table else ctx.save_for_backward(idx)
        out, rk, ck = get_backend().gather_rows_keys(table, idx)
        ctx.mark_non_differentiable(rk, ck)
        ctx.set_materialize_grads(False)                  # (else autograd zero-fills "gradients" for the two key tensors every step)
        return out, rk, ck

    @staticmethod
    def backward(ctx, d_out, _drk, _dck):
        if d_out is None:
            return None, None, None, None
        return _GatherRowsFn.backward(ctx, d_out)


def gather_keys_supported(d):
    """True when gather_rows(..., keys=True) is available and of use for width d on the installed backend."""
    be = get_backend()
    return hasattr(be, "gather_rows_keys") and be.keys_supported(d)


def relu_gather_supported(table, inverse):
    """True when gather_rows(table, ., inverse, relu_table=True) is available for this table (a tensor, or its width)."""
    d = table if isinstance(table, int) else table.shape[1]
    return inverse is not None and d % 4 == 0 and d <= 256


def gather_rows(table, idx, inverse=None, relu_table=False, keys=False):
    """out[i] = table[idx[i]] (idx int32, -1 => zero row).  Backward: atomic scatter-add, or -- when the caller
    supplies `inverse` = gather_inverse(idx, rows) for a static index list -- a deterministic segment sum.
    relu_table=True: `table` is the ReLU output of the layer before and this gather is its ONLY consumer: the backward returns
    the gradient with that ReLU's adjoint already applied (zero where table <= 0), in the same kernel; the producing layer must
    then run its own backward with grad_premasked=True (rgcn_layer)."""
    if relu_table and not relu_gather_supported(table, inverse):
        raise ValueError("relu_table needs a static inverse and a width the segment-sum kernels take")
    if keys:                                              # -> (out, (row_keys, col_keys)): see _GatherRowsKeysFn
        out, rk, ck = _GatherRowsKeysFn.apply(table, idx, inverse, bool(relu_table))
        return out, (rk, ck)
    return _GatherRowsFn.apply(table, idx, inverse, bool(relu_table))


def row_spans(y, spans):
    """Row ranges [(row0, n), ...] of `y` for several consumers.  When the ranges tile y in order they come from ONE split -- its
    adjoint is one concatenation of the consumers' gradients; separate slices each cost a zero-filled full-size gradient, a copy
    and an addition in the backward (the per-position loops hand every position its rows of one layer output)."""
    off = 0
    for r0, n in spans:
        if r0 != off:
            return [y[r0:r0 + n] for r0, n in spans]
        off += n
    if off != y.shape[0]:
        return [y[r0:r0 + n] for r0, n in spans]
    return list(y.split([n for _, n in spans]))


def gather_inverse(idx_np, n_rows, device):
    """(seg_ptr int32 [n_rows+1], order int32) grouping the positions of a gather index list by table row
    (positions of negative entries are left out): a stable counting sort in the host planner library
    (temp_host_gather_inverse), one upload."""
    from . import _hostlib
    both, _ = _hostlib.gather_inverse(idx_np, n_rows)
    dev = _lib.to_device(both, device)
    return dev[:n_rows + 1], dev[n_rows + 1:]


class _CandidateCEFn(torch.autograd.Function):
    """mean_p CE(query[p] . all_embeds[cand[p, :]]^T, label 0) without materialising (P, C, D)."""

    @staticmethod
    def forward(ctx, query, all_embeds, cand):
        be = get_backend()
        scores = be.linear(query, all_embeds, True)                  # (P, N): every positive against ALL entities
        loss_rows, lse = be.gather_ce_fwd(scores, cand)
        ctx.save_for_backward(query, all_embeds, scores, lse, cand)
        return loss_rows.mean()

    @staticmethod
    def backward(ctx, d_loss):
        query, all_embeds, scores, lse, cand = ctx.saved_tensors
        be = get_backend()
        d_scores = be.gather_ce_bwd(scores, cand, lse, d_loss.reshape(1).contiguous(), 1.0 / max(scores.shape[0], 1))
        d_query = be.linear(d_scores, all_embeds, False)             # (P,N) . (N,D)
        d_all = be.linear_tn(d_scores, query)                        # (N,P) . (P,D)
        return d_query, d_all, None


class _BatchedCandidateCEFn(torch.autograd.Function):
    """sum_b mean_{rows of window b} CE(query[p] . all_embeds_b[cand[p, :]]^T, label 0): the row-wise work (candidate CE
    forward / backward) runs once over ALL windows' rows; only the three GEMMs are per window, because every window scores
    against its own all-entity matrix."""

    @staticmethod
    def forward(ctx, query, cand, splits, row_w, *all_embeds):
        be = get_backend()
        scores = torch.cat([be.linear(query[a:b], e, True) for (a, b), e in zip(splits, all_embeds)], dim=0)    # (sum P, N)
        loss_rows, lse = be.gather_ce_fwd(scores, cand)
        ctx.save_for_backward(query, scores, lse, cand, row_w, *all_embeds)
        ctx.splits = splits
        return (loss_rows * row_w).sum()

    @staticmethod
    def backward(ctx, d_loss):
        query, scores, lse, cand, row_w = ctx.saved_tensors[:5]
        all_embeds = ctx.saved_tensors[5:]
        be = get_backend()
        d_scores = be.gather_ce_bwd(scores, cand, lse, d_loss.reshape(1).contiguous(), 1.0, row_w)
        d_query = torch.cat([be.linear(d_scores[a:b], e, False) for (a, b), e in zip(ctx.splits, all_embeds)], dim=0)
        d_all = tuple(be.linear_tn(d_scores[a:b], query[a:b]) for (a, b) in ctx.splits)
        return (d_query, None, None, None) + d_all


# Entity-major score GEMMs (see forward): 10 % less device time on ICEWS-shaped steps (3.51 -> 3.18 ms under graph replay), but
# ~30 more launches per step, which makes an eager, host-bound training loop SLOWER (5.7 -> 6.8 ms/step measured).  Off unless
# TEMP_LOSS_TALL=1.
_TALL_SCORES = __import__("os").environ.get("TEMP_LOSS_TALL", "0") == "1"


class _BatchedLinkPredictionFn(torch.autograd.Function):
    """The whole batched link-prediction loss as ONE autograd node:
        q      = bilinear_query(ent_rows[known], rel[rel_idx])                 (gathers fused, temp_bilinear_query_fwd)
        scores = q[rows of window b] . all_b^T   for every window b             (temp_linear_multi, 4 windows per launch)
        loss   = sum_rows w_row * CE(scores[row, cand[row, :]], label 0)        (temp_gather_ce_fwd)
    `big` is the (B * N, D) stack of the windows' all-entity matrices.  The backward mirrors it and reduces the per-row
    gradients of the gathered operands with deterministic segment sums over the static index lists."""

    @staticmethod
    def forward(ctx, ent_rows, rel, big, kind, inp):
        be = get_backend()
        N = big.shape[0] // len(inp["splits"])
        q = be.bilinear_query_fwd(kind, ent_rows, inp["known"], rel, inp["rel"], inp["is_tail"])
        live = [(b, a0, a1) for b, (a0, a1) in enumerate(inp["splits"]) if a1 > a0]
        # few positives against many entities (ICEWS-like: ~200 rows x 10 000 entities per window): the ENTITY axis is made the
        # tall one of every GEMM -- scores = (all_b . q_b^T)^T through a transposed-store epilogue, and the backward products
        # from d_scores^T.  (The planner pads every window's block to a multiple of 4 rows so it can be an N / K extent.)
        tall = _TALL_SCORES and all((a1 - a0) % 4 == 0 and 8 * (a1 - a0) <= N for _, a0, a1 in live)
        scores = torch.empty(q.shape[0], N, dtype=torch.float32, device=q.device)
        if tall:
            for b, a0, a1 in live:
                be.linear_t(big[b * N:(b + 1) * N], q[a0:a1], True, scores[a0:a1])
        else:
            be.linear_multi([q[a0:a1] for _, a0, a1 in live], [big[b * N:(b + 1) * N] for b, _, _ in live], True, scores)
        loss_rows, lse = be.gather_ce_fwd(scores, inp["cand"])
        ctx.save_for_backward(ent_rows, rel, big, q, scores, lse)
        ctx.kind, ctx.inp, ctx.live, ctx.N, ctx.tall = kind, inp, live, N, tall
        return (loss_rows * inp["weights"]).sum()

    @staticmethod
    def backward(ctx, d_loss):
        ent_rows, rel, big, q, scores, lse = ctx.saved_tensors
        inp, live, N = ctx.inp, ctx.live, ctx.N
        be = get_backend()
        d_scores = be.gather_ce_bwd(scores, inp["cand"], lse, d_loss.reshape(1).contiguous(), 1.0, inp["weights"])
        d_q = torch.empty_like(q)
        d_big = torch.empty_like(big) if len(live) == len(inp["splits"]) else torch.zeros_like(big)
        if ctx.tall:
            for b, a0, a1 in live:
                dst = d_scores[a0:a1].t().contiguous()                                        # (N, rows of window b)
                be.linear_tn(dst, big[b * N:(b + 1) * N], out=d_q[a0:a1])                     # d_q   = d_scores . all_b
                d_big[b * N:(b + 1) * N].copy_(be.linear(dst, q[a0:a1], False))               # d_all = d_scores^T . q
        else:
            be.linear_multi([d_scores[a0:a1] for _, a0, a1 in live], [big[b * N:(b + 1) * N] for b, _, _ in live], False, d_q)
            if hasattr(be, "linear_tn_multi"):                                                # d_all_b = d_scores_b^T . q_b, one launch
                be.linear_tn_multi([d_scores[a0:a1] for _, a0, a1 in live], [q[a0:a1] for _, a0, a1 in live],
                                   [d_big[b * N:(b + 1) * N] for b, _, _ in live])
            else:
                for b, a0, a1 in live:
                    be.linear_tn(d_scores[a0:a1], q[a0:a1], out=d_big[b * N:(b + 1) * N])
        dk, dr = be.bilinear_query_bwd(ctx.kind, ent_rows, inp["known"], rel, inp["rel"], inp["is_tail"], d_q)
        d_ent = be.segment_sum_rows(dk, inp["known_inv"][0], inp["known_inv"][1], ent_rows.shape[0])
        d_rel = be.segment_sum_rows(dr, inp["rel_inv"][0], inp["rel_inv"][1], rel.shape[0])
        return d_ent, d_rel, d_big, None, None


class _BatchedEnsembleLinkPredictionFn(torch.autograd.Function):
    """_BatchedLinkPredictionFn for the score-level ensemble of the post-ensemble models (combined_scores,
    models/PostDynamicRGCN.py:404-406, 425-428): TWO streams (local = pre-GRU, temporal = post-GRU) score against their own
    all-entity stacks, the scores are mixed row by row with w (local) and 1 - w (temporal) on the (rows, N) matrices, then the
    candidate cross-entropy.  Same launches per stream as the plain node (folded query, score GEMMs of all windows as
    multi-problem launches -- k-sliced where the rows are few against 10 000 entities -- one candidate CE), one mix pass."""

    @staticmethod
    def forward(ctx, loc_rows, rec_rows, rel, big_loc, big_rec, w, kind, inp):
        be = get_backend()
        N = big_loc.shape[0] // len(inp["splits"])
        live = [(b, a0, a1) for b, (a0, a1) in enumerate(inp["splits"]) if a1 > a0]
        qs, sc = [], []
        for rows, big in ((loc_rows, big_loc), (rec_rows, big_rec)):
            q = be.bilinear_query_fwd(kind, rows, inp["known"], rel, inp["rel"], inp["is_tail"])
            s = torch.empty(q.shape[0], N, dtype=torch.float32, device=q.device)
            be.linear_multi([q[a0:a1] for _, a0, a1 in live], [big[b * N:(b + 1) * N] for b, _, _ in live], True, s)
            qs.append(q)
            sc.append(s)
        mixed = torch.lerp(sc[1], sc[0], w)                          # w (rows, 1): w * local + (1 - w) * temporal
        loss_rows, lse = be.gather_ce_fwd(mixed, inp["cand"])
        ctx.save_for_backward(loc_rows, rec_rows, rel, big_loc, big_rec, w, qs[0], qs[1], sc[0], sc[1], mixed, lse)
        ctx.kind, ctx.inp, ctx.live, ctx.N = kind, inp, live, N
        return (loss_rows * inp["weights"]).sum()

    @staticmethod
    def backward(ctx, d_loss):
        loc_rows, rec_rows, rel, big_loc, big_rec, w, q_l, q_r, s_l, s_r, mixed, lse = ctx.saved_tensors
        inp, live, N = ctx.inp, ctx.live, ctx.N
        be = get_backend()
        d_m = be.gather_ce_bwd(mixed, inp["cand"], lse, d_loss.reshape(1).contiguous(), 1.0, inp["weights"])
        d_sl = d_m * w
        d_sr = d_m - d_sl
        d_w = (d_m * (s_l - s_r)).sum(dim=1, keepdim=True) if ctx.needs_input_grad[5] else None
        outs, d_rel = [], None
        for rows, big, q, d_s in ((loc_rows, big_loc, q_l, d_sl), (rec_rows, big_rec, q_r, d_sr)):
            d_q = torch.empty_like(q)
            d_big = torch.empty_like(big) if len(live) == len(inp["splits"]) else torch.zeros_like(big)
            be.linear_multi([d_s[a0:a1] for _, a0, a1 in live], [big[b * N:(b + 1) * N] for b, _, _ in live], False, d_q)
            if hasattr(be, "linear_tn_multi"):
                be.linear_tn_multi([d_s[a0:a1] for _, a0, a1 in live], [q[a0:a1] for _, a0, a1 in live], [d_big[b * N:(b + 1) * N] for b, _, _ in live])
            else:
                for b, a0, a1 in live:
                    be.linear_tn(d_s[a0:a1], q[a0:a1], out=d_big[b * N:(b + 1) * N])
            dk, dr = be.bilinear_query_bwd(ctx.kind, rows, inp["known"], rel, inp["rel"], inp["is_tail"], d_q)
            outs.append((be.segment_sum_rows(dk, inp["known_inv"][0], inp["known_inv"][1], rows.shape[0]), d_big))
            d_rel = dr if d_rel is None else d_rel + dr
        d_rel = be.segment_sum_rows(d_rel, inp["rel_inv"][0], inp["rel_inv"][1], rel.shape[0])
        return outs[0][0], outs[1][0], d_rel, outs[0][1], outs[1][1], d_w, None, None


def batched_ensemble_link_prediction(loc_rows, rec_rows, rel, big_loc, big_rec, w, kind, inputs):
    """sum over windows of the ensemble CE_tail + CE_head; `w` (rows, 1) = weight of the LOCAL stream of every stacked query row
    (per window: [tail rows: weight_object ; head rows: weight_subject]); `inputs` = TKG_Module.loss_inputs(...)."""
    return _BatchedEnsembleLinkPredictionFn.apply(loc_rows.contiguous(), rec_rows.contiguous(), rel, big_loc.contiguous(), big_rec.contiguous(),
                                                  w.reshape(-1, 1).to(loc_rows.dtype).contiguous(), kind, inputs)


def batched_link_prediction(ent_rows, rel, big, kind, inputs):
    """sum over windows of CE_tail + CE_head (models/DynamicRGCN.py:186-193) for a bilinear scorer `kind`
    ('distmult' | 'complex'); `inputs` = TKG_Module.loss_inputs(...)."""
    return _BatchedLinkPredictionFn.apply(ent_rows, rel, big, kind, inputs)


def candidate_cross_entropy_batched(query, cand, splits, row_w, all_embeds):
    """query (R, D), cand (R, C) int32, splits = [(row_begin, row_end)] per window, row_w (R,) = weight of every row's loss
    (1 / P_b for a mean per window and direction), all_embeds = list of (N, D) per window."""
    return _BatchedCandidateCEFn.apply(query, cand, splits, row_w, *all_embeds)


class _MixedCandidateCEFn(torch.autograd.Function):
    """mean_p CE( w_p * (q_loc[p] . A_loc[cand[p, :]]^T) + (1 - w_p) * (q_rec[p] . A_rec[cand[p, :]]^T), label 0 ): the score-level
    ensemble of the post-ensemble models (combined_scores, models/PostDynamicRGCN.py:404-406, 425-428) without the
    (P, 1 + neg, D) gathers: both streams score against ALL entities (two small GEMMs), the mix is one pass over the (P, N) score
    matrices, the candidate CE reads the mixed matrix."""

    @staticmethod
    def forward(ctx, q_loc, all_loc, q_rec, all_rec, w, cand):
        be = get_backend()
        s_loc = be.linear(q_loc, all_loc, True)                      # (P, N)
        s_rec = be.linear(q_rec, all_rec, True)
        mixed = torch.lerp(s_rec, s_loc, w)                          # w (P, 1): w * loc + (1 - w) * rec
        loss_rows, lse = be.gather_ce_fwd(mixed, cand)
        ctx.save_for_backward(q_loc, all_loc, q_rec, all_rec, w, cand, s_loc, s_rec, mixed, lse)
        return loss_rows.mean()

    @staticmethod
    def backward(ctx, d_loss):
        q_loc, all_loc, q_rec, all_rec, w, cand, s_loc, s_rec, mixed, lse = ctx.saved_tensors
        be = get_backend()
        d_mixed = be.gather_ce_bwd(mixed, cand, lse, d_loss.reshape(1).contiguous(), 1.0 / max(mixed.shape[0], 1))
        d_loc = d_mixed * w
        d_rec = d_mixed - d_loc
        d_w = (d_mixed * (s_loc - s_rec)).sum(dim=1, keepdim=True) if ctx.needs_input_grad[4] else None
        return (be.linear(d_loc, all_loc, False), be.linear_tn(d_loc, q_loc), be.linear(d_rec, all_rec, False), be.linear_tn(d_rec, q_rec),
                d_w, None)


def candidate_cross_entropy_mixed(q_loc, all_loc, q_rec, all_rec, w, cand):
    """Score-level ensemble CE (see _MixedCandidateCEFn).  w: (P, 1) mixing weight of the local stream; cand: int32 (P, C), column
    0 is the true entity."""
    return _MixedCandidateCEFn.apply(q_loc.contiguous(), all_loc.contiguous(), q_rec.contiguous(), all_rec.contiguous(),
                                     w.reshape(-1, 1).to(q_loc.dtype).contiguous(), cand)


def candidate_cross_entropy(query, all_embeds, cand):
    """F.cross_entropy(score(query, all_embeds[cand]), 0) for scorers that are bilinear in
    (query, candidate) -- DistMult and ComplEx (utils/scores.py:4-44).  cand: int32 (P, C), column 0
    is the true entity."""
    return _CandidateCEFn.apply(query, all_embeds, cand)


class _LinearFn(torch.autograd.Function):
    """y = x . W^T (W stored (out, in) like nn.Linear.weight) through the MFMA panel GEMM."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return get_backend().linear(x, w, True)

    @staticmethod
    def backward(ctx, d_y):
        x, w = ctx.saved_tensors
        be = get_backend()
        d_y = d_y.contiguous()
        d_x = be.linear(d_y, w, False) if ctx.needs_input_grad[0] else None
        d_w = be.linear_tn(d_y, x) if ctx.needs_input_grad[1] else None
        return d_x, d_w


def linear(x, w):
    return _LinearFn.apply(x, w)


class _LinearNTFn(torch.autograd.Function):
    """y = x . W with W stored (in, out) -- the `time_weight` matrices of the linear-recurrence layers."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return get_backend().linear(x, w, False)

    @staticmethod
    def backward(ctx, d_y):
        x, w = ctx.saved_tensors
        be = get_backend()
        d_y = d_y.contiguous()
        d_x = be.linear(d_y, w, True) if ctx.needs_input_grad[0] else None
        d_w = be.linear_tn(x, d_y) if ctx.needs_input_grad[1] else None
        return d_x, d_w


def linear_nt(x, w):
    return _LinearNTFn.apply(x.contiguous(), w)


class _DecayRowsFn(torch.autograd.Function):
    """x[r, :] * exp(-dt[r] * lam) (temp_decay_rows); the row scale is its own adjoint."""

    @staticmethod
    def forward(ctx, x, dt, lam):
        ctx.save_for_backward(dt)
        ctx.lam = lam
        return get_backend().decay_rows(x.contiguous(), dt.contiguous().view(-1), lam)

    @staticmethod
    def backward(ctx, d_y):
        (dt,) = ctx.saved_tensors
        return get_backend().decay_rows(d_y.contiguous(), dt.contiguous().view(-1), ctx.lam), None, None


def decay_rows(x, dt, lam):
    return _DecayRowsFn.apply(x, dt, float(lam))


class _HistoryAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, kv_hist, idx, decay, inverse):
        out, score, lse = get_backend().sa_attn_fwd(qkv, kv_hist, idx, decay)
        ctx.save_for_backward(qkv, kv_hist, idx, decay if decay is not None else qkv.new_zeros(0), out, score, lse)
        ctx.has_decay = decay is not None
        ctx.inverse = inverse
        return out

    @staticmethod
    def backward(ctx, d_out):
        qkv, kv_hist, idx, decay, out, score, lse = ctx.saved_tensors
        d_qkv, d_hist, d_decay = get_backend().sa_attn_bwd(qkv, kv_hist, idx, decay if ctx.has_decay else None, out, score, lse,
                                                           d_out.contiguous(), ctx.inverse)
        return d_qkv, d_hist, None, d_decay, None


def history_attention(qkv, kv_hist, idx, decay=None, inverse=None):
    """8-head attention of every query row over its active history rows + itself.
    qkv (n,3D) = [q | k | v] projections of the query rows' current states; kv_hist (R,2D) = [k | v]
    projections of the history table; idx (n,T-1) int32 rows of the table, -1 => masked;
    decay (T,) additive score bias (already negated / clamped) or None;
    inverse = attention_inverse(idx, R): static maps that make the backward deterministic (no atomics).
    Returns (n,D) in the reference's feature order (d * 8 + head)."""
    return _HistoryAttentionFn.apply(qkv, kv_hist, idx, decay, inverse)


def attention_inverse(idx_np, n_table, device):
    """(inv_ptr int32 [n_table+1], inv_ref int32): the (query row, position) pairs of idx (n, T-1) grouped by the table
    row they point at; ref = i * (T-1) + t.  Same stable counting sort as gather_inverse, over the flattened index matrix."""
    import numpy as np
    return gather_inverse(np.asarray(idx_np).reshape(-1), n_table, device)
