"""Multi-GPU execution of the window encoder: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Two ways the path shards:

1. windows -> ranks (the reference's DDP axis, models/TKG_Module.py:166-168): every rank encodes its
   own windows, no data-path collective; the only exchange is ONE bucketed all-reduce of the
   parameter gradients after backward (`allreduce_gradients`).

2. snapshot visits -> ranks (BASELINE north_star; needs --rec-only-last-layer): all ranks share the
   same batch of windows; the (position, window) snapshot visits are cut into `world` contiguous
   same batch of windows; the DISTINCT snapshots the windows visit are cut into `world` contiguous
   ranges balanced by edge count; each rank runs the two RGCN layers on ITS snapshots only, then ONE
   UNPADDED all-gather of the per-snapshot node states (every rank posts its rows straight into the
   destination row range of each peer's buffer: a grouped point-to-point batch, `_AllGatherRows`) gives
   every rank the GRU inputs of all visits; the recurrent chain itself is sharded by window (rank =
   window index mod world).  Backward mirrors it: the adjoint of the all-gather returns to each rank
   the pieces of its own rows' gradient from every peer, summed in rank order; parameter gradients are
   partial sums on every rank and are all-reduced like in (1).  (`SnapshotShardedEncoder`)
"""
import numpy as np
import torch
import torch.distributed as dist

from . import functional as TF
from .gru_cell import GRUCell
from .gru_chain import GruInstance, GruProgram, gru_chain, prepare_program
from .window import ChainPlan, Step, window_times


class GradBucket:
    """ONE preallocated flat buffer over ALL trainable parameters, in a fixed order that is identical on every rank.
    A parameter without a gradient on this rank (a rank that owns no window of a short last batch, a weight its shard
    never touches) contributes zeros, so the collective has the same size everywhere and is never skipped.

    A parameter for which NO rank produced a gradient keeps `.grad = None`, exactly as in single-process training (Adam then
    skips it: no moment update, no weight decay -- the post/impute branches and time weights are such parameters in some
    configurations).  For that the buffer carries one presence float per parameter behind the gradients (1 where this rank has
    a gradient), reduced by the same collective; only a rank that itself lacks a gradient reads its entries back (one small
    device-to-host copy) -- a rank with every gradient present already knows the answer and never synchronises."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.spans, off = [], 0
        for p in self.params:
            self.spans.append((off, off + p.numel()))
            off += p.numel()
        self.numel = off
        self.flat = None
        self._present = None                         # (pattern of local presence, its device tensor): rebuilt when the pattern changes

    def _buffer(self):
        dev = self.params[0].device if self.params else torch.device("cpu")
        if self.flat is None or self.flat.device != dev:
            self.flat = torch.zeros(self.numel + len(self.params), dtype=torch.float32, device=dev)
        return self.flat

    def allreduce(self, world=None, average=True, group=None, grads=None, force=False):
        """Few launches whatever the number of parameters: ONE multi-tensor copy of the gradients into the flat buffer, the
        collective (averaging inside it where the backend can: RCCL), and NO copy back -- each parameter's `.grad` becomes
        its span of the flat buffer (a later in-place accumulation into `.grad` then lands in the buffer, and the next
        call's copy of such a span onto itself is skipped).  `grads` (one tensor or None per parameter) replaces the
        `.grad`s as the source: a replayed HIP graph writes the gradient tensors of its CAPTURE, whatever `.grad` points to."""
        if world is None:
            world = dist.get_world_size(group)
        if (world == 1 and not force) or not self.params:    # force: run the collective on a single rank too (exercises the RCCL path)
            return
        flat = self._buffer()
        views = [flat[a:b].view_as(p) for p, (a, b) in zip(self.params, self.spans)]
        src, dst, missing, miss_i = [], [], [], []
        for i, (p, v) in enumerate(zip(self.params, views)):
            g = p.grad if grads is None else grads[i]
            if g is None:
                missing.append(v)
                miss_i.append(i)
            elif g.data_ptr() != v.data_ptr() or not g.is_contiguous():
                src.append(g if g.dtype == flat.dtype else g.to(flat.dtype))
                dst.append(v)
        if missing:
            torch._foreach_zero_(missing)
        if dst:
            torch._foreach_copy_(dst, src)
        pattern = tuple(miss_i)
        if self._present is None or self._present[0] != pattern or self._present[1].device != flat.device:
            host = torch.ones(len(self.params), dtype=torch.float32)
            host[miss_i] = 0.0
            self._present = (pattern, host.to(flat.device))
        flat[self.numel:].copy_(self._present[1])    # reduced below together with the gradients (SUM or AVG: > 0 iff some rank had one)
        fused_avg = average and dist.get_backend(group) == "nccl"
        dist.all_reduce(flat, op=dist.ReduceOp.AVG if fused_avg else dist.ReduceOp.SUM, group=group)
        if average and not fused_avg:
            flat[:self.numel].div_(world)
        nobody = set()
        if miss_i:                                   # only a rank that lacks a gradient needs to ask whether anybody had one
            got = flat[self.numel:][torch.tensor(miss_i, device=flat.device)].cpu()
            nobody = {i for i, x in zip(miss_i, got.tolist()) if x <= 0.0}
        for i, (p, v) in enumerate(zip(self.params, views)):
            p.grad = None if i in nobody else v


def allreduce_gradients(params, world=None, average=True, group=None, grads=None, force=False):
    """One flat bucket for all parameter gradients (about 2.5 M floats at D=200 on ICEWS14); the bucket (layout + buffer) is
    built once per parameter list and kept on its first parameter."""
    if grads is not None:
        grads = [g for p, g in zip(params, grads) if p.requires_grad]
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    key = tuple(id(p) for p in params)
    bucket = getattr(params[0], "_temp_grad_bucket", None)
    if bucket is None or bucket[0] != key:
        bucket = (key, GradBucket(params))
        params[0]._temp_grad_bucket = bucket
    bucket[1].allreduce(world, average, group, grads, force)


# Test switch: with ONE rank there is no peer, so the exchange below would issue no point-to-point operation at all (RCCL
# refuses two ranks on one GPU: "Duplicate GPU detected").  LOOPBACK_P2P = True makes the rank its own peer: its block travels
# through `batch_isend_irecv` (a grouped ncclSend / ncclRecv to itself) instead of the local copy, forward and adjoint, so the
# RCCL point-to-point path, its stream ordering and its place between the HIP graphs run on a one-GPU box.  P2P_BATCHES counts
# the batches issued.
LOOPBACK_P2P = False
P2P_BATCHES = 0


def _self_exchange(dst, src, rank, group):
    global P2P_BATCHES
    ops = [dist.P2POp(dist.irecv, dst, _global_rank(rank, group), group), dist.P2POp(dist.isend, src, _global_rank(rank, group), group)]
    P2P_BATCHES += 1
    for req in dist.batch_isend_irecv(ops):
        req.wait()


def allgather_rows(local, out, bounds, world, rank, group=None):
    """out[bounds[q]:bounds[q+1]] <- rank q's `local` rows, for every q, as ONE grouped point-to-point batch (see _AllGatherRows).
    `out` is caller-owned (a static buffer when the surrounding compute is replayed from HIP graphs).  On RCCL `req.wait()` orders
    the CURRENT STREAM behind the transfers (the host does not block); on gloo it blocks the host, which is what a CPU run needs."""
    global P2P_BATCHES
    mine = out[bounds[rank]:bounds[rank + 1]]
    in_place = local.shape[0] > 0 and local.data_ptr() == mine.data_ptr()        # the layer already wrote its rows where they belong
    if LOOPBACK_P2P and local.shape[0] > 0:
        src = local.clone() if in_place else local.contiguous()
        _self_exchange(mine, src, rank, group)
    elif not in_place:
        mine.copy_(local)
    ops = []
    for q in range(world):
        if q == rank:
            continue
        if bounds[q + 1] > bounds[q]:
            ops.append(dist.P2POp(dist.irecv, out[bounds[q]:bounds[q + 1]], _global_rank(q, group), group))
        if local.shape[0] > 0:
            ops.append(dist.P2POp(dist.isend, local, _global_rank(q, group), group))
    if ops:
        P2P_BATCHES += 1
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


def allgather_rows_adjoint(d_all, pieces, bounds, world, rank, group=None, out=None):
    """Adjoint of allgather_rows: block q of every rank's `d_all` goes to rank q, which sums the pieces in RANK ORDER
    (deterministic).  `pieces` is a caller-owned (world, n_local, d) buffer (slot `rank` stays unused unless the loop-back
    test switch is on: this rank's own piece is read where it lies, in `d_all`); `out`: caller-owned (n_local, d) result
    buffer (a static one when the surrounding compute is replayed from HIP graphs).  -> (n_local, d)"""
    lo, hi = bounds[rank], bounds[rank + 1]
    n_local = hi - lo
    global P2P_BATCHES
    own = d_all[lo:hi]
    if LOOPBACK_P2P and n_local > 0:
        _self_exchange(pieces[rank], own, rank, group)
        own = pieces[rank]
    ops = []
    for q in range(world):
        if q == rank:
            continue
        if n_local > 0:
            ops.append(dist.P2POp(dist.irecv, pieces[q], _global_rank(q, group), group))
        if bounds[q + 1] > bounds[q]:
            ops.append(dist.P2POp(dist.isend, d_all[bounds[q]:bounds[q + 1]], _global_rank(q, group), group))
    if ops:
        P2P_BATCHES += 1
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    part = lambda q: own if q == rank else pieces[q]
    if world == 1:
        if out is None:
            return own
        out.copy_(own)
        return out
    if out is None:
        out = torch.empty_like(own)
    torch.add(part(0), part(1), out=out)             # fixed order: the sum does not depend on arrival order
    for q in range(2, world):
        out.add_(part(q))
    return out


class _AllGatherRows(torch.autograd.Function):
    """All-gather of UNEVEN row blocks straight into the canonical layout, and its adjoint.

    The shards are contiguous ranges of the (distinct-snapshot) row space, so rank r's block lands at rows
    [bounds[r], bounds[r+1]) of the output: no padding to the largest shard, no re-ordering gather.  Implemented as ONE group
    of point-to-point operations (`batch_isend_irecv`: on RCCL a grouped send/recv, i.e. every rank writes its block to
    each peer over its own xGMI link -- the links are point-to-point, so a direct all-to-all of blocks uses all 7 of them at
    once instead of walking a ring).
    backward: every rank holds gradient rows for ALL blocks (it consumed them in its windows of the recurrence); block q of
    each rank is sent to rank q, which sums the pieces in rank order (deterministic)."""

    @staticmethod
    def forward(ctx, local, bounds, world, rank, group):
        out = allgather_rows(local, local.new_empty(int(bounds[-1]), local.shape[1]), bounds, world, rank, group)
        ctx.meta = (bounds, world, rank, group)
        return out

    @staticmethod
    def backward(ctx, d_all):
        bounds, world, rank, group = ctx.meta
        d_all = d_all.contiguous()
        pieces = d_all.new_empty(world, bounds[rank + 1] - bounds[rank], d_all.shape[1])
        return allgather_rows_adjoint(d_all, pieces, bounds, world, rank, group), None, None, None, None


def _global_rank(r, group):
    return r if group is None else dist.get_global_rank(group, r)


def split_visits_by_edges(edge_counts, world):
    """Cut the visit list into `world` contiguous ranges with about equal edge totals."""
    csum = np.cumsum(np.asarray(edge_counts, dtype=np.float64))
    total = csum[-1] if len(csum) else 0.0
    bounds = [0]
    for r in range(1, world):
        bounds.append(int(np.searchsorted(csum, total * r / world, side="left")))
    bounds.append(len(edge_counts))
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


class SnapshotShardedEncoder:
    """Snapshot-visit sharding of a (Bi)DynamicRGCN batched encoder pass across ranks."""

    def __init__(self, model, group=None):
        self.model = model
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        assert model._can_batch() and model._can_chain(), "snapshot sharding needs the batched GRU + rec-only-last-layer path"

    # ---------------------------------------------------------------------------------------------
    def prepare(self, t_list, seq_len, train=True, target_edge_ids=None):
        m, dev = self.model, self.model._device()
        W, R = self.world, self.rank
        bi = hasattr(m.ent_encoder.layer_2, "forward_rnn")
        times = m.total_time
        rows_f = window_times(t_list, seq_len, times)
        bsz = len(rows_f)
        graphs = [m.graph_dict_train[r[-1]] for r in rows_f]
        tgt = m.sample_target_graphs(graphs, 0.5, target_edge_ids) if train else graphs
        # ---- global visit list (every rank builds the same one: it is pure host integer work) ------
        plan_f = ChainPlan(rows_f, m.graph_dict_train, m.num_ents, seq_len)
        plans = [plan_f]
        if bi:
            rows_b = window_times(t_list, seq_len, times, ascending=True)
            plans.append(ChainPlan(rows_b, m.graph_dict_train, m.num_ents, seq_len))
        target = Step(seq_len - 1, list(range(bsz)), tgt, [r[-1] for r in rows_f])
        steps = [st for p in plans for st in p.steps] + [target]
        visits = []                                   # (step index, j in step, graph)
        for si, st in enumerate(steps):
            for j, g in enumerate(st.graphs):
                visits.append((si, j, g))
        # a snapshot visited by several windows (or by one window's forward chain and another's backward chain) enters the
        # RGCN -- and the all-gather -- ONCE: the shards are cut over the DISTINCT snapshots, visits index their rows
        # (not while the self-loop dropout draws: every visit has its own mask then, DynamicRGCN._share_visits)
        share = m._share_visits(train)
        uniq, dgraphs, visit_d = {}, [], []
        for vi, (_, _, g) in enumerate(visits):
            k = uniq.get(id(g) if share else ("visit", vi))
            if k is None:
                k = uniq[id(g) if share else ("visit", vi)] = len(dgraphs)
                dgraphs.append(g)
            visit_d.append(k)
        bounds = split_visits_by_edges([g.number_of_edges() for g in dgraphs], W)
        sizes = np.array([g.n for g in dgraphs], dtype=np.int64)
        canon_off = np.concatenate([[0], np.cumsum(sizes)])
        row_bounds = [int(canon_off[b]) for b in bounds]                 # rank r owns canonical rows [row_bounds[r], row_bounds[r+1])
        # ---- this rank's RGCN shard ------------------------------------------------------------------
        from . import snapshot as S
        mine = dgraphs[bounds[R]:bounds[R + 1]]
        g_local = S.batch(mine)
        g_local.device_graph(dev, 2 * m.num_rels)
        # ---- this rank's windows of the recurrence ---------------------------------------------------
        def local_windows(nwin):
            return [b for b in range(nwin) if b % W == R]
        # canonical first row of (step si, window slot j)
        first_row = {}
        k = 0
        for si, st in enumerate(steps):
            for j in range(len(st.graphs)):
                first_row[(si, j)] = int(canon_off[visit_d[k]])
                k += 1
        sb = type("ShardBatch", (), {})()
        sb.g_local, sb.ids_local = g_local, torch.from_numpy(g_local.gids.astype(np.int32)).to(dev)
        sb.ids_inv = TF.gather_inverse(g_local.gids, self.model.num_ents, dev)          # static ids: table layer + deterministic gradient
        sb.row_bounds = row_bounds
        sb.gather_bytes = int(canon_off[-1]) * self.model.embed_size * 4   # bytes every rank ends up holding after the all-gather
        sb.n_edge_visits_local = int(sum(g.number_of_edges() for g in mine))
        sb.n_edge_visits_global = int(sum(g.number_of_edges() for _, _, g in visits))
        # centre / target rows of this rank's windows (forward order)
        tw = local_windows(bsz)
        t_rows, t_sizes = [], []
        ti = len(steps) - 1
        for b in tw:
            r0 = first_row[(ti, b)]
            t_rows.append(np.arange(r0, r0 + tgt[b].n))
            t_sizes.append(tgt[b].n)
        t_rows = np.concatenate(t_rows) if t_rows else np.zeros(0, np.int64)
        n_t = int(sum(t_sizes))
        # local chain program: for every plan (direction), the sub-chain over this rank's windows followed by its target instance.
        # Each direction's x and h rows are ONE contiguous run -- the target rows are indexed once per direction, as the one-GPU
        # model copies them (bi_dynamic_rgcn.py) -- so the program has one group per GRU with disjoint x rows: the chain backward
        # writes its gate gradients once and all weight gradients are one launch (gru_chain.py backward)
        inst, x_rows, out_inst = [], [], []
        x_off = 0
        step_base = 0
        for pi, plan in enumerate(plans):
            nwin = plan.bsz
            # windows of the backward plan are in ascending-target order; window b of the forward order is
            # index bsz-1-b there (torch.flip of the reference, models/BiDynamicRGCN.py:97-99)
            fwd_of = (lambda j: j) if pi == 0 else (lambda j: nwin - 1 - j)
            keep = [j for j in range(nwin) if fwd_of(j) % W == R]
            sub_rows = [plan.rows[j] for j in keep]
            sub = ChainPlan(sub_rows, m.graph_dict_train, m.num_ents, seq_len) if keep else None
            prev_inst = -1
            if sub is not None:
                for st in sub.steps:
                    gi_step = step_base + [s.p for s in plan.steps].index(st.p)
                    rows_idx = []
                    for jj, wj in enumerate(st.windows):
                        j_global = keep[wj]
                        gslot = steps[gi_step].windows.index(j_global)
                        r0 = first_row[(gi_step, gslot)]
                        rows_idx.append(np.arange(r0, r0 + st.sizes[jj]))
                    rows_idx = np.concatenate(rows_idx) if rows_idx else np.zeros(0, np.int64)
                    inst.append(GruInstance(st.n_rows, x_off, pi, prev_inst, st.prev_idx, st.dt))
                    prev_inst = len(inst) - 1
                    x_rows.append(rows_idx)
                    x_off += st.n_rows
            step_base += len(plan.steps)
            pidx, dts = [], []
            for b in tw:
                if sub is not None:
                    jl = keep.index(b if pi == 0 else bsz - 1 - b)
                    a, d = sub.final_prev(jl, tgt[b].gids, seq_len - 1)
                else:
                    a, d = np.full(tgt[b].n, -1, np.int64), np.full(tgt[b].n, seq_len - 1, np.float32)
                pidx.append(a)
                dts.append(d)
            pidx = np.concatenate(pidx) if pidx else np.zeros(0, np.int64)
            dts = np.concatenate(dts) if dts else np.zeros(0, np.float32)
            inst.append(GruInstance(n_t, x_off, pi, prev_inst, pidx, dts))
            out_inst.append(len(inst) - 1)
            x_rows.append(t_rows)
            x_off += n_t
        sb.program = GruProgram(inst)
        x_index = np.concatenate(x_rows) if x_rows else np.zeros(0, np.int64)
        sb.program.x_src = x_index                  # x row i is canonical node-state row x_index[i]: a snapshot visited at several positions of
                                                    # this rank's windows shares its input gates (GruProgram.gi_shared), as on one GPU
        prepare_program(sb.program, dev, m.embed_size, len(plans), out_inst)
        sb.x_index = torch.from_numpy(x_index).to(dev)
        sb.x_index32 = sb.x_index.to(torch.int32)
        sb.x_inv = TF.gather_inverse(x_index, int(canon_off[-1]), dev)                      # y2_all has one row per canonical visit row
        sb.out_inst, sb.target_sizes, sb.target_windows = out_inst, t_sizes, tw
        return sb

    # ---------------------------------------------------------------------------------------------
    # The step in three compute parts around the two exchanges (ShardedStep replays each part as ONE HIP graph):
    def local_layers(self, sb, out=None):
        """Part 1: the two RGCN layers on this rank's snapshots -> y2 (n_local, D), attached to the autograd graph.
        out: this rank's row range of the exchange buffer (ShardedStep): layer 2 writes there, no copy before the all-gather."""
        m = self.model
        enc = m.ent_encoder
        y1 = enc.layer_1.conv_table(sb.g_local, m.ent_embeds, sb.ids_local, sb.ids_inv)     # layer 1 on the embedding table (HISTORY.md 3b)
        # layer 2's ReLU (models/BiRRGCN.py:202-203): its adjoint rides in the backward of the ONE consumer of the gathered states,
        # the row gather of chain_on_gathered -- the mask is a function of the state row alone, so every rank masks its piece of
        # a row's gradient with the same mask and the rank-ordered sum of the pieces is the masked sum
        return enc.layer_2.conv(sb.g_local, y1, grad_premasked=self._relu_fold(sb), out=out)

    def _relu_fold(self, sb):
        l2 = self.model.ent_encoder.layer_2
        return bool(l2.relu_fused() and TF.relu_gather_supported(l2.out_feat, sb.x_inv))

    def chain_on_gathered(self, sb, y2_all):
        """Part 2: GRU inputs of this rank's windows out of the gathered node states + the window-sharded recurrence
        -> target-position embeddings of this rank's windows."""
        l2 = self.model.ent_encoder.layer_2
        x = TF.gather_rows(y2_all, sb.x_index32, sb.x_inv, relu_table=self._relu_fold(sb))   # deterministic adjoint (segment sum, ReLU mask folded in)
        rnns = [l2.forward_rnn, l2.backward_rnn] if hasattr(l2, "forward_rnn") else [l2.rnn]
        pieces = gru_chain(x, sb.program, rnns, l2.inv_temperature, isinstance(rnns[0], GRUCell), want=list(sb.out_inst))
        out = None
        for piece in pieces:                          # only the target-position states are consumed (no full-size gradient buffers)
            out = piece if out is None else out + piece
        return out

    def run(self, sb):
        """-> (target-position embeddings of THIS rank's windows, concatenated, forward order).  A rank that owns no window
        (bsz < world) gets a (0, D) tensor that is still attached to the graph: EVERY rank must run backward on (a function
        of) its output -- `out.sum()` is enough -- because the adjoint of the all-gather is a collective."""
        y2 = self.local_layers(sb)
        y2_all = _AllGatherRows.apply(y2, sb.row_bounds, self.world, self.rank, self.group)
        return self.chain_on_gathered(sb, y2_all)


class ShardedStep:
    """One forward+backward of the snapshot-sharded encoder on a STATIC batch with everything except the collectives replayed
    from HIP graphs (BASELINE north_star):

        graph A   RGCN layers on this rank's snapshots                      -> y2 (n_local, D)
        exchange  unpadded all-gather of the node states (grouped P2P over xGMI, RCCL's stream)
        graph B   row gather + window-sharded GRU chain, forward AND backward -> out, d(gathered states), GRU gradients
        exchange  the adjoint: every rank's piece of MY rows' gradient, summed in rank order
        graph C   backward of the RGCN layers                               -> layer / embedding gradients
        all-reduce of the flat gradient bucket (GradBucket, averaging inside the collective on RCCL)

    The three graphs share one memory pool (graph C consumes what graph A saved for backward); the buffers the exchanges
    read and write are static.  `upstream`: the gradient on `out` (default ones).  With graphs=False the same three
    parts run eagerly (CPU / gloo tests, and the reference the graphed run is compared with: bit-identical)."""

    def __init__(self, enc, sb, params, graphs=True, average=False, force_allreduce=False):
        self.enc, self.sb, self.params, self.average = enc, sb, [p for p in params if p.requires_grad], average
        self.force_allreduce = force_allreduce
        self.world, self.rank, self.group = enc.world, enc.rank, enc.group
        dev = enc.model._device()
        d = enc.model.embed_size
        b = sb.row_bounds
        self.n_local = b[self.rank + 1] - b[self.rank]
        self.y2_all = torch.zeros(int(b[-1]), d, dtype=torch.float32, device=dev)           # static exchange buffers
        self.pieces = torch.zeros(self.world, self.n_local, d, dtype=torch.float32, device=dev)
        self.d_local = torch.zeros(self.n_local, d, dtype=torch.float32, device=dev)
        self.out = None
        self.graphs = None
        # (a step whose self-loop dropout draws is NOT captured: the masks' seeds are drawn on the host per call and would be
        #  baked into the graphs -- every replay the same mask; rgcn.RGCNLayer._drop refuses a capture for the same reason)
        if graphs and dev.type == "cuda" and not enc.model._dropout_active():
            self._capture()

    # -- the three parts (they communicate through attributes so that a capture and an eager call are the same code) ------
    def _part_a(self):
        b = self.sb.row_bounds
        self.y2 = self.enc.local_layers(self.sb, out=self.y2_all[b[self.rank]:b[self.rank + 1]] if self.n_local > 0 else None)

    def _part_b(self):
        leaf = self.y2_all.detach().requires_grad_(True)
        self.out = self.enc.chain_on_gathered(self.sb, leaf)
        up = getattr(self, "_ones", None)
        if up is None or up.shape != self.out.shape:
            up = self._ones = torch.ones_like(self.out)
        self.out.backward(up)
        self.d_all = leaf.grad if leaf.grad is not None else torch.zeros_like(self.y2_all)

    def _part_c(self):
        self.y2.backward(self.d_local)

    def _exchange_fwd(self):
        allgather_rows(self.y2.detach(), self.y2_all, self.sb.row_bounds, self.world, self.rank, self.group)

    def _exchange_bwd(self):
        allgather_rows_adjoint(self.d_all, self.pieces, self.sb.row_bounds, self.world, self.rank, self.group, out=self.d_local)

    def _eager(self):
        for p in self.params:
            p.grad = None
        self._part_a()
        self._exchange_fwd()
        self._part_b()
        self._exchange_bwd()
        self._part_c()

    def _capture(self):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # warm-up off the capture stream (allocator, lazy module state)
            for _ in range(2):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for p in self.params:
            p.grad = None
        # Nothing may keep an autograd graph of these parameters alive across the capture: a parameter's AccumulateGrad node
        # survives as long as a graph references it and keeps the stream it was created on -- with a node left over from an
        # EAGER backward on the default stream, graph C's backward would touch the default stream inside the capture (HIP aborts
        # the process).  The warm-up's own references go here; callers must drop theirs (outputs of earlier steps) first.
        self.y2 = self.out = self.d_all = None
        ga, gb, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga, capture_error_mode="thread_local"):
            self._part_a()
        self._exchange_fwd()
        with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode="thread_local"):
            self._part_b()
        self._exchange_bwd()
        with torch.cuda.graph(gc, pool=ga.pool(), capture_error_mode="thread_local"):
            self._part_c()
        torch.cuda.synchronize()
        self.graphs = (ga, gb, gc)
        self.grads = [p.grad for p in self.params]    # the tensors every replay writes

    def step(self, allreduce=True):
        """-> target-position embeddings of this rank's windows; the parameters' .grad hold this rank's (or, after the
        all-reduce, the job's) gradients."""
        if self.graphs is None:
            self._eager()
            grads = None
        else:
            ga, gb, gc = self.graphs
            ga.replay()
            self._exchange_fwd()
            gb.replay()
            self._exchange_bwd()
            gc.replay()
            grads = self.grads
            for p, g in zip(self.params, grads):
                p.grad = g
        if allreduce:
            allreduce_gradients(self.params, self.world, average=self.average, group=self.group, grads=grads, force=self.force_allreduce)
        return self.out
