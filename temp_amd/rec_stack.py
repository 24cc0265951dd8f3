"""The reference's DEFAULT flags (no --rec-only-last-layer) as one autograd node.

With both layers recurrent (models/RRGCN.py:179-181: layer 1 is a GRRGCNLayer too) position p of a window computes
    y1 = RGCN_1(g_p, ent_embeds[ids])            h1 = GRU_1(y1, dec(prev))
    y2 = RGCN_2(g_p, h1)                         h2 = GRU_2(y2, dec(prev))
and -- the GRU layers write their output into the CALLER's graph (models/RRGCN.py:86) -- RRGCN.forward
returns h2 twice, so `prev` of BOTH layers at position p + 1 is h2 of position p (models/DynamicRGCN.py:156-174).  The
chain h2_p -> h1_{p+1} -> RGCN_2 -> h2_{p+1} is therefore sequential through a graph convolution and cannot be handed
to the window-chain kernels; what does not depend on the recurrence is hoisted out of the loop:
    * RGCN_1 of EVERY visit in one launch over the union of the distinct snapshots (the caller: conv_table on g_all),
    * GRU_1's input gates of all rows in one GEMM (temp_gru_input_gates),
    * both GRUs' weight / bias gradients and d_y1 in one call each over all rows (temp_gru_weight_grads),
and the loop itself is four library calls per position forward (cell 1, RGCN_2, its input gates, cell 2) and four
backward, issued from ONE autograd node: no per-position autograd graph (the reference-granular path builds ~8 nodes
per position and pays for them twice), previous states and their gradients through the plan's row maps.

rec_stack() returns exactly what the per-position RRGCN.forward loop returns for the same weights and draws.
"""
import torch

from . import _lib
from .backend import get_backend
from .functional import ACTS


class _RecStackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y1, prog, graphs, g_union, cfg, want, w2, loop2, bias2, *gru):
        be = get_backend()
        dev, d = y1.device, y1.shape[1]
        N = prog.n_total
        lam, variant, nb, act, drops = cfg["lam"], cfg["variant"], cfg["num_bases"], cfg["act"], cfg["drops"]
        y1 = y1.detach().contiguous()
        w2, loop2 = w2.detach().contiguous(), loop2.detach().contiguous()
        bias2 = bias2.detach().contiguous() if bias2 is not None else None
        (wi1, wh1, bi1, bh1), (wi2, wh2, bi2, bh2) = [tuple(w.detach().contiguous() for w in gru[4 * r:4 * r + 4]) for r in range(2)]
        G = wi1.shape[0]
        new = lambda *shape: torch.empty(*shape, dtype=y1.dtype, device=dev)
        gi1, gi2 = new(N, G), new(N, G)
        H1, Y2, H2 = new(N, d), new(N, d), new(N, d)
        saved1, saved2 = new(5, N, d), new(5, N, d)
        be.gru_input_gates(y1, wi1, bi1, variant, gi1)
        tens = prog.upload(dev)
        _, none_idx = prog.constants(dev, d)
        for i, it in enumerate(prog.inst):
            if it.n == 0:
                continue
            sl = slice(it.h0, it.h0 + it.n)
            pi, _, dt = tens[i]
            if it.prev >= 0:
                p = prog.inst[it.prev]
                prev, pidx = H2[p.h0:p.h0 + p.n], pi
            else:                                 # first executed position: every previous state is zero
                prev, pidx = None, none_idx[:it.n]
            cell = dict(prev=prev, prev_idx=pidx, dt=dt, row0=it.h0)
            be.gru_cell_fwd_multi([dict(cell, gi=gi1[sl], w_hh=wh1, b_hh=bh1, h_out=H1[sl])], lam, variant, saved1)
            be.rgcn_fwd(graphs[i], H1[sl], None, w2, loop2, bias2, nb, act, drops[i], out=Y2[sl])
            be.gru_input_gates(Y2[sl], wi2, bi2, variant, gi2[sl])
            be.gru_cell_fwd_multi([dict(cell, gi=gi2[sl], w_hh=wh2, b_hh=bh2, h_out=H2[sl])], lam, variant, saved2)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(y1, H1, Y2, saved1, saved2, w2, loop2, wi1, wh1, wi2, wh2)
        ctx.prog, ctx.graphs, ctx.cfg, ctx.want, ctx.has_bias, ctx.g_union = prog, graphs, cfg, want, bias2 is not None, g_union
        return tuple(H2[prog.inst[i].h0:prog.inst[i].h0 + prog.inst[i].n] for i in want)

    @staticmethod
    def backward(ctx, *d_outs):
        be = get_backend()
        y1, H1, Y2, saved1, saved2, w2, loop2, wi1, wh1, wi2, wh2 = ctx.saved_tensors
        prog, graphs, cfg = ctx.prog, ctx.graphs, ctx.cfg
        lam, variant, nb, act, drops = cfg["lam"], cfg["variant"], cfg["num_bases"], cfg["act"], cfg["drops"]
        dev, d, N, G = y1.device, y1.shape[1], prog.n_total, wi1.shape[0]
        given = {i: g.contiguous() for i, g in zip(ctx.want, d_outs) if g is not None}
        new = lambda *shape: torch.empty(*shape, dtype=y1.dtype, device=dev)
        dgi1, dgi2, dgh1, dgh2 = new(N, G), new(N, G), new(N, 3 * d), new(N, 3 * d)
        decv1, decv2, dp1, dp2 = new(N), new(N), new(N, d), new(N, d)
        tens = prog.upload(dev)
        d_w2 = d_loop2 = d_bias2 = None
        # RGCN_2's weight / bias gradients are sums over ALL positions: with the union graph of the positions at hand (row order =
        # program order) and a backend that has the split backward, the loop computes d_h1 only and keeps every position's masked
        # output gradient; ONE pass over the union afterwards forms the weight gradients (per position it took a relation-weight
        # kernel, its fix-up, two reductions, a column sum and three additions)
        hoist = ctx.g_union is not None and hasattr(be, "rgcn_bwd_dh")
        if hoist:
            # the ONE pass over the union reads H1 / DZ by the union's node order: it must be the program's row order, position by
            # position (a mismatch would give wrong d_W silently)
            gu_n = getattr(ctx.g_union, "n_nodes", None)
            gu_n = gu_n if gu_n is not None else getattr(ctx.g_union, "n", None)
            assert gu_n is None or int(gu_n) == prog.n_total, "rec_stack: union graph rows != program rows"
            sizes = [getattr(g, "n_nodes", getattr(g, "n", None)) for g in graphs]
            assert all(s is None or int(s) == it.n for s, it in zip(sizes, prog.inst)), "rec_stack: position graphs do not match the program's instances"
        relu = act == ACTS["relu"]
        any_drop = any(dr is not None for dr in drops)
        assert not any_drop or all(dr is not None for dr in drops), "dropout draws at every position or at none"
        if hoist:
            DY2 = new(N, d)
            DZ = new(N, d) if relu else DY2
            DZM = new(N, d) if any_drop else None
        for i in range(len(prog.inst) - 1, -1, -1):
            it = prog.inst[i]
            if it.n == 0:
                continue
            sl = slice(it.h0, it.h0 + it.n)
            _, ni, dt = tens[i]
            nxt = prog.inst[it.next] if (it.next >= 0 and prog.inst[it.next].n > 0) else None
            d_next = None
            if nxt is not None:                   # h2 of this position fed BOTH cells of the next one
                ns = slice(nxt.h0, nxt.h0 + nxt.n)
                d_next = torch.add(dp1[ns], dp2[ns])
            cell = dict(row0=it.h0, n=it.n, dt=dt, no_prev=it.prev < 0)
            be.gru_cell_bwd_multi([dict(cell, dh_up=given.get(i), d_prev_next=d_next, next_idx=ni if nxt is not None else None, w_hh=wh2,
                                        dgi=dgi2[sl], dgh=dgh2[sl], decv=decv2[sl], d_prev=dp2[sl])], lam, variant, saved2)
            if hoist:
                d_y2 = be.linear(dgi2[sl], wi2, False, out=DY2[sl])                 # (n, G) . (G, d)
                d_h1 = be.rgcn_bwd_dh(graphs[i], Y2[sl], d_y2, w2, loop2, nb, act, drops[i], dz_out=DZ[sl] if relu else None,
                                      dzm_out=DZM[sl] if (any_drop and drops[i] is not None) else None)
            else:
                d_y2 = be.linear(dgi2[sl], wi2, False)                              # (n, G) . (G, d)
                d_h1, dw, dl, db = be.rgcn_bwd(graphs[i], H1[sl], Y2[sl], d_y2, w2, loop2, ctx.has_bias, nb, act, drops[i])
                if d_w2 is None:
                    d_w2, d_loop2, d_bias2 = dw, dl, db
                else:
                    d_w2 += dw
                    d_loop2 += dl
                    if db is not None:
                        d_bias2 += db
            be.gru_cell_bwd_multi([dict(cell, dh_up=d_h1, d_prev_next=None, next_idx=None, w_hh=wh1,
                                        dgi=dgi1[sl], dgh=dgh1[sl], decv=decv1[sl], d_prev=dp1[sl])], lam, variant, saved1)
        if hoist and N > 0:
            d_w2, d_loop2, d_bias2 = be.rgcn_bwd_weights(ctx.g_union, H1, DZ, DZM, w2, loop2, ctx.has_bias, nb)
        d_y1 = torch.empty_like(y1)
        zero_state = all(it.prev < 0 for it in prog.inst)                            # hdec = 0 on every row
        multi = None
        if not zero_state and hasattr(be, "gru_weight_grads_multi"):               # both GRUs in one weight-gradient launch
            multi = be.gru_weight_grads_multi([y1, Y2], [saved1[4], saved2[4]], [dgi1, dgi2], [dgh1, dgh2], [wi1, wi2], variant, [d_y1, None])
        if multi is not None:
            gw1, gw2 = multi
        else:
            gw1 = be.gru_weight_grads(y1, None if zero_state else saved1[4], dgi1, dgh1, wi1, variant, d_y1)
            gw2 = be.gru_weight_grads(Y2, None if zero_state else saved2[4], dgi2, dgh2, wi2, variant, None)
        if d_w2 is None:
            d_w2, d_loop2 = torch.zeros_like(w2), torch.zeros_like(loop2)
            d_bias2 = torch.zeros(loop2.shape[1], dtype=torch.float32, device=dev) if ctx.has_bias else None
        return (d_y1, None, None, None, None, None, d_w2, d_loop2, d_bias2) + tuple(gw1) + tuple(gw2)


def rec_stack(y1, prog, graphs, rnn1, layer2, want, g_union=None):
    """y1: layer-1 RGCN output of every visit row (program row order).  prog: GruProgram of ONE chain (instance i =
    executed position i).  graphs: device graph of every position (layer 2 runs on them).  rnn1: layer 1's GRU; layer2:
    the second GRRGCNLayer (weights, GRU, dropout, activation).  want: instances whose h2 rows are returned.  g_union: device graph
    of ALL positions' graphs batched in program row order (optional: layer 2's weight gradients are then formed in one pass).
    -> tuple of (n_i, d) tensors."""
    from .gru_cell import GRUCell
    type1 = isinstance(rnn1, GRUCell)
    ws = []
    for rnn in (rnn1, layer2.rnn):
        ws += [rnn.weight_ih, rnn.weight_hh, rnn.bias_ih, rnn.bias_hh] if type1 else \
            [rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0]
    cfg = dict(lam=float(layer2.inv_temperature), variant=_lib.GRU_TYPE1 if type1 else _lib.GRU_TORCH, num_bases=layer2.num_bases,
               act=ACTS[layer2._act], drops=[layer2._drop() for _ in graphs])
    return _RecStackFn.apply(y1, prog, list(graphs), g_union, cfg, tuple(want), layer2.weight, layer2.loop_weight, layer2._bias(), *ws)
