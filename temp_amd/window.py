"""Window scheduling for the recurrent snapshot encoder: which snapshot is visited at which window
position, where each node's previous state lives, and how long ago the node was last active.

This replaces the reference's dense per-step bookkeeping -- `hist_embeddings (bsz,2,N_ents,D)`
re-allocated and zero-filled at EVERY window position plus `start_time_tensor (bsz,N_ents)`
(models/DynamicRGCN.py:35-54,156-174; 91 MB of zero-fill per step on ICEWS14) -- by a
host-side *plan* of int32 row maps, with identical semantics:

    prev(v) at position p = output row of v at position p-1 if v was active there, else 0
    dt(v)   at position p = p - last_active(v)          (last_active starts at 0)

Because snapshot membership is static, the plan is pure integer work on the host; the device only
sees `prev_idx` (row into the previous position's output, -1 => zero) and `dt`.
"""
import numpy as np

from . import snapshot as S
from . import _lib


def window_times(t_list, seq_len, times, ascending=False):
    """Window construction of TKG_Module.get_batch_graph_list (models/TKG_Module.py:232-250) and of
    the backward half of BiDynamicRGCN.get_batch_graph_list (models/BiDynamicRGCN.py:17-49).

    Returns rows[b][p] (left-padded with None).  Forward: targets sorted DESCENDING, window
    [t-L+1 .. t].  Backward (`ascending=True`): targets sorted ASCENDING, window [t .. t+L-1]
    reversed so the target is last."""
    times = [int(t) for t in times]
    index = {t: i for i, t in enumerate(times)}
    ts = sorted([int(t) for t in t_list], reverse=not ascending)
    rows = []
    for tim in ts:
        k = index[tim]
        if not ascending:
            seq = times[max(0, k + 1 - seq_len):k + 1]
        else:
            seq = times[k:k + seq_len][::-1]
        rows.append([None] * (seq_len - len(seq)) + seq)
    return rows


class Step:
    """One executed window position: the graphs of the windows active there, batched."""
    __slots__ = ("p", "windows", "graphs", "times", "sizes", "n_rows", "_ids", "prev_idx", "next_idx", "dt", "graph",
                 "row0", "dev")

    def __init__(self, p, windows, graphs, times):
        self.p, self.windows, self.graphs, self.times = p, windows, graphs, times
        self.sizes = [g.n for g in graphs]
        self.n_rows = sum(self.sizes)
        self._ids = None             # concatenated global ids: only the per-position path (tensors()) reads them
        self.prev_idx = None
        self.next_idx = None         # inverse of the next executed step's prev_idx (filled by ChainPlan for history steps)
        self.dt = None
        self.graph = None            # batched Snapshot (built lazily)
        self.row0 = 0                # first row inside an all-visits batch (fast path)
        self.dev = {}

    @property
    def ids(self):
        if self._ids is None:
            self._ids = np.concatenate([g.gids for g in self.graphs]) if self.graphs else np.zeros(0, np.int64)
        return self._ids

    @property
    def offsets(self):
        return np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int64)

    def batched(self):
        if self.graph is None:
            self.graph = S.batch(self.graphs)
        return self.graph

    def tensors(self, device):
        """(ids int32, prev_idx int32, dt float32 (n,1)) on `device`, cached."""
        key = str(device)
        t = self.dev.get(key)
        if t is None:
            t = (_lib.to_device(self.ids.astype(np.int32), device),
                 _lib.to_device(self.prev_idx.astype(np.int32), device),
                 _lib.to_device(self.dt.astype(np.float32), device).view(-1, 1))
            self.dev[key] = t
        return t


class ChainPlan:
    """History positions 0..L-2 of one direction, plus the maps the consumers of the final history
    need (target position, all-entity pass)."""

    def __init__(self, rows, graph_dict, num_ents, seq_len):
        self.bsz = len(rows)
        self.seq_len = seq_len
        self.num_ents = num_ents
        self.rows = rows
        self.steps = []
        for p in range(seq_len - 1):
            win = [b for b in range(self.bsz) if rows[b][p] is not None]
            if not win:
                continue
            assert win == list(range(len(win))), "padded windows must form a suffix of the batch"
            self.steps.append(Step(p, win, [graph_dict[rows[b][p]] for b in win], [rows[b][p] for b in win]))
        # row maps of every executed position in one pass of the host planner library (temp_host_chain_plan):
        # prev_idx = row in the previous executed step's output (F8: the history holds ONLY that step's nodes), dt = gap
        from . import _hostlib
        _lib.pause_point()
        prev_idx, next_idx, dt, row_of, last = _hostlib.chain_plan(self.bsz, num_ents, [st.p for st in self.steps], [len(st.windows) for st in self.steps],
                                                         [[g.gids for g in st.graphs] for st in self.steps])
        off = 0
        for st in self.steps:
            st.prev_idx, st.dt, st.next_idx = prev_idx[off:off + st.n_rows], dt[off:off + st.n_rows], next_idx[off:off + st.n_rows]
            off += st.n_rows
        if self.steps:
            self.steps[-1].next_idx = None                      # consumed by the target position, planned by the caller
        self.row_of, self.last = row_of, last

    def flipped(self):
        """torch.flip(hist, [0]) / flip(start_time) of models/BiDynamicRGCN.py:97-99: re-index the
        final maps by the forward batch order."""
        self.row_of = self.row_of[::-1].copy()
        self.last = self.last[::-1].copy()
        return self

    def final_prev(self, b, ids, cur_t):
        """prev_idx / dt of nodes `ids` of window b as get_prev_embeddings (models/DynamicRGCN.py:35-45)
        would compute them after the last history position."""
        return self.row_of[b][ids], (cur_t - self.last[b][ids]).astype(np.float32)

    def final_all(self, b, cur_t):
        """Same for ALL entities (the isolated pass of get_all_embeds_Gt, models/DynamicRGCN.py:56-64)."""
        return self.row_of[b], (cur_t - self.last[b]).astype(np.float32)


def concat_steps_dedup(steps):
    """Like concat_steps, but a snapshot that is visited several times in the step -- the same
    timestamp inside several overlapping windows, or in one window's forward chain and another's
    backward chain -- enters the batched RGCN graph ONCE: with only the last layer recurrent the RGCN
    stack of a visit depends on the snapshot alone, so its output rows are shared by all visits
    (the reference recomputes them per window).  Target snapshots carry a per-window random edge
    subsample and stay distinct.

    Returns (batched graph over the distinct snapshots, visit_rows, total visit rows): visit_rows is
    an int32 array mapping every visit row (step.row0 layout) to its row in the distinct layout, or
    None when nothing is shared."""
    uniq, graphs = {}, []
    off_u = 0
    bases, sizes = [], []
    off = 0
    shared = False
    for st in steps:
        st.row0 = off
        for g in st.graphs:
            key = id(g)
            base = uniq.get(key)
            if base is None:
                base = uniq[key] = off_u
                graphs.append(g)
                off_u += g.n
            else:
                shared = True
            bases.append(base)
            sizes.append(g.n)
        off += st.n_rows
    _lib.pause_point()
    vr = None
    if shared and bases:
        sizes = np.asarray(sizes, dtype=np.int64)
        start = np.cumsum(sizes) - sizes                         # first visit row of every visit
        vr = (np.repeat(np.asarray(bases, dtype=np.int64) - start, sizes) + np.arange(off, dtype=np.int64)).astype(np.int32)
        _lib.pause_point()
    return S.batch(graphs), vr, off


def concat_steps(steps):
    """All visits of several steps as ONE batched graph (fast path: the RGCN stack of every visit is
    independent of history when only the last layer is recurrent, models/RRGCN.py:182-187), with
    each step's first row recorded in `step.row0`."""
    graphs, off = [], 0
    for st in steps:
        st.row0 = off
        graphs.extend(st.graphs)
        off += st.n_rows
    return S.batch(graphs), off
