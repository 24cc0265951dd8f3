"""Window-batched GRU recurrence: one autograd node for the whole chain of window positions.

With --rec-only-last-layer the GRU input of every position is known before the recurrence starts
(models/RRGCN.py:182-187: layer 1 is a plain RGCNLayer), so the per-position work shrinks to what is
truly sequential -- hdec . W_hh^T + gates forward, gate gradients + d_prev backward -- and the
input-side GEMM, d_x, and all weight / bias gradients run once over all rows (temp_gru_input_gates,
temp_gru_weight_grads).  Previous states and their gradients move between positions through the
plan's row maps (`prev_idx` forward, its inverse `next_idx` backward), never through a dense
(bsz, N_ents, D) history (SURVEY F8 semantics kept: -1 => zero state).

A `GruProgram` is a small static DAG of *instances*; instance i applies GRU weights `rnn[i]` to
x rows [x0, x0+n) with previous state = output of instance `prev` (or none) and writes H rows
[h0, h0+n).  The bi-directional centre position is two instances over the same x rows whose
outputs the caller adds (models/BiRRGCN.py:45).
"""
import numpy as np
import torch

from . import _hostlib, _lib
from .backend import get_backend


CHAIN_KERNELS = True        # A/B switch: False keeps the per-position launches (temp_gru_cell_*_multi)
WEIGHT_GRADS_MULTI = True   # A/B switch: False computes each GRU's weight gradients with its own launches (temp_gru_weight_grads)
KEYED_GRADS = True          # A/B switch: False keeps the consumers of g4 on the six-product bf16 split (no keys from the chain backward)
GATE_GRADS_ONCE = True      # A/B switch: False keeps dgi + dgh as two matrices (temp_gru_chain_bwd + temp_gru_weight_grads_multi)


class GruInstance:
    __slots__ = ("n", "x0", "h0", "rnn", "prev", "next", "prev_idx", "next_idx", "dt", "group")

    def __init__(self, n, x0, rnn, prev, prev_idx_np, dt_np, next_idx_np=None):
        self.n, self.x0, self.rnn, self.prev = int(n), int(x0), rnn, prev
        self.h0 = 0
        self.next = -1
        self.prev_idx = prev_idx_np
        self.next_idx = next_idx_np      # inverse of the successor's prev_idx when the planner already has it
        self.dt = dt_np
        self.group = -1


class GruProgram:
    def __init__(self, instances):
        self.inst = instances
        off = 0
        for it in instances:
            it.h0 = off
            off += it.n
        self.n_total = off
        # inverse row maps: which row of the NEXT instance consumes each row of this one
        for i, it in enumerate(instances):
            if it.prev >= 0:
                p = instances[it.prev]
                assert p.next == -1, "an instance feeds at most one successor"
                p.next = i
                if p.next_idx is None:
                    inv = np.full(p.n, -1, dtype=np.int32)
                    ok = it.prev_idx >= 0
                    inv[it.prev_idx[ok]] = np.nonzero(ok)[0].astype(np.int32)
                    p.next_idx = inv
        # groups: maximal runs of instances with the same weights and contiguous x and h rows
        self.groups = []
        for i, it in enumerate(instances):
            g = self.groups[-1] if self.groups else None
            if g is not None and g["rnn"] == it.rnn and g["x1"] == it.x0 and g["h1"] == it.h0:
                g["x1"] += it.n
                g["h1"] += it.n
            else:
                self.groups.append(dict(rnn=it.rnn, x0=it.x0, x1=it.x0 + it.n, h0=it.h0, h1=it.h0 + it.n))
            it.group = len(self.groups) - 1
        # levels: instances at the same distance from their chain start are independent of each other
        # (forward-direction and backward-direction chains, the two centre cells) -> launched together
        depth = []
        for it in instances:
            depth.append(0 if it.prev < 0 else depth[it.prev] + 1)
        self.levels = []
        for lv in range(max(depth) + 1 if depth else 0):
            idx = [i for i, dp in enumerate(depth) if dp == lv and instances[i].n > 0]
            for k in range(0, len(idx), 4):
                self.levels.append(idx[k:k + 4])
        self.dev = None

    # ---- persistent chain kernels: panels of 32 entity tracks (include/temp_amd.h: TempGruChain) ------------------------
    def chain_plan(self):
        """Host half of the persistent-chain tables, independent of `want` and cached: every row of every instance gets a
        TRACK -- it inherits its predecessor's (prev_idx >= 0), otherwise takes the lowest track that no row of its instance
        inherits (a state survives exactly one position, so a track whose entity was absent is free again) -- and tracks are
        cut into panels of 32.  A panel's steps are the positions at which it has a row; a position where the whole panel is
        idle is dropped (none of its tracks can carry a state across it).
        -> None when the program is not a set of chains with one GRU each (the per-position kernels are used then), else
           dict(panel [P,4], rows [S,32], any_prev [S], step_inst [S])."""
        if getattr(self, "_chain_plan", False) is not False:
            return self._chain_plan
        self._chain_plan = self._chain_plan_host()
        return self._chain_plan

    def _chain_plan_host(self):
        """chain_plan in the C++ planner library (temp_host_chain_tracks); bit-identical to _chain_plan_numpy."""
        inst = self.inst
        if not any(it.prev >= 0 and it.n > 0 for it in inst):          # at least one real recurrence step
            return None
        chains = []
        for head, it0 in enumerate(inst):
            if it0.prev >= 0:
                continue
            chain = [head]
            while inst[chain[-1]].next >= 0:
                chain.append(inst[chain[-1]].next)
            chains.append(chain)
        res = _hostlib.chain_tracks(chains, [it.n for it in inst], [it.h0 for it in inst], [it.rnn for it in inst],
                                    [getattr(it, "prev_idx", None) if it.prev >= 0 else None for it in inst], _lib.CHAIN_TRACKS, _lib.CHAIN_MAX_STEPS)
        if res is None:
            return None
        panel, rows, anyp, sinst = res
        return dict(panel=panel, rows=rows, any_prev=anyp, step_inst=sinst)

    def _chain_plan_numpy(self):
        """The numpy formulation of chain_plan (kept as the cross-check of the planner library, tests/test_chain_cpu.py)."""
        inst = self.inst
        T = _lib.CHAIN_TRACKS
        panel, rows_tab, any_prev, step_inst = [], [], [], []
        s_base = 0
        plan = None
        ok = any(it.prev >= 0 and it.n > 0 for it in inst)            # at least one real recurrence step
        for head, it0 in enumerate(inst):
            if not ok:
                break
            if it0.prev >= 0:
                continue
            chain = [head]
            while inst[chain[-1]].next >= 0:
                chain.append(inst[chain[-1]].next)
            chain = [i for i in chain]
            if any(inst[i].rnn != it0.rnn for i in chain):
                ok = False
                break
            n_tracks, prev_tr, steps = 0, None, []
            for k, i in enumerate(chain):
                it = inst[i]
                tr = np.full(it.n, -1, dtype=np.int64)
                has = np.zeros(it.n, dtype=bool)
                if k > 0 and it.n:
                    pi = np.asarray(it.prev_idx, dtype=np.int64)
                    has = pi >= 0
                    tr[has] = prev_tr[pi[has]]
                new = np.nonzero(~has)[0]
                if new.size:
                    used = np.zeros(n_tracks, dtype=bool)
                    used[tr[has]] = True
                    free = np.nonzero(~used)[0]
                    take = min(free.size, new.size)
                    tr[new[:take]] = free[:take]
                    extra = new.size - take
                    if extra:
                        tr[new[take:]] = n_tracks + np.arange(extra)
                        n_tracks += extra
                steps.append((i, tr, has))
                prev_tr = tr
            if n_tracks == 0:
                continue
            n_pan = (n_tracks + T - 1) // T
            tab = np.full((len(chain), n_pan * T), -1, dtype=np.int64)
            for k, (i, tr, has) in enumerate(steps):
                if tr.size:
                    tab[k, tr] = (inst[i].h0 + np.arange(tr.size)) | (has.astype(np.int64) << 30)
            tab = tab.reshape(len(chain), n_pan, T)
            active = (tab >= 0).any(axis=2)                              # [K, P]
            anyp = ((tab >= 0) & ((tab >> 30) & 1).astype(bool)).any(axis=2)
            act_t = active.T                                             # panel-major, position-minor
            counts = act_t.sum(axis=1)
            if counts.max() > _lib.CHAIN_MAX_STEPS:
                ok = False
                break
            first = s_base + np.concatenate([[0], np.cumsum(counts)[:-1]])
            panel.append(np.stack([np.full(n_pan, it0.rnn), first, counts, np.zeros(n_pan, dtype=np.int64)], axis=1))
            rows_tab.append(tab.transpose(1, 0, 2)[act_t])
            any_prev.append(anyp.T[act_t])
            step_inst.append(np.broadcast_to(np.asarray(chain)[None, :], act_t.shape)[act_t])
            s_base += int(counts.sum())
        if ok and panel:
            pn = np.concatenate(panel)
            pn = pn[pn[:, 2] > 0]
            plan = dict(panel=pn.astype(np.int32), rows=np.concatenate(rows_tab).astype(np.int32),
                        any_prev=np.concatenate(any_prev), step_inst=np.concatenate(step_inst).astype(np.int64))
        return plan

    def chain_tables(self, device, want):
        """Device tables of the persistent chain kernels for one set of consumed instances (`want`: the instances whose
        states are handed out -- their rows are written to H and receive an upstream gradient; None = all rows, one dense
        gradient).  One packed upload, cached per (device, want)."""
        plan = self.chain_plan()
        _lib.pause_point()
        if plan is None or (want is not None and len(want) > _lib.CHAIN_MAX_UP):
            return None
        cache = self.__dict__.setdefault("_chain_tabs", {})
        key = (str(device), want)
        tabs = cache.get(key)
        if tabs is None:
            S = plan["rows"].shape[0]
            sinfo = np.zeros((S, 4), dtype=np.int32)
            si = plan["step_inst"]
            if want is None:
                sinfo[:, 0] = plan["any_prev"].astype(np.int32) | 2
            else:
                sel = np.full(len(self.inst), -1, dtype=np.int32)
                sel[list(want)] = np.arange(len(want), dtype=np.int32)
                h0 = np.array([it.h0 for it in self.inst], dtype=np.int32)
                sinfo[:, 0] = plan["any_prev"].astype(np.int32) | np.where(sel[si] >= 0, 2, 0)
                sinfo[:, 1] = sel[si]
                sinfo[:, 2] = np.where(sel[si] >= 0, h0[si], 0)
            dt = np.zeros(self.n_total, dtype=np.float32)
            for it in self.inst:
                if it.n:
                    dt[it.h0:it.h0 + it.n] = np.asarray(it.dt, dtype=np.float32).reshape(-1)
            _lib.pause_point()
            parts = [plan["panel"].reshape(-1), plan["rows"].reshape(-1), sinfo.reshape(-1), dt.view(np.int32)]
            buf = _lib.to_device(np.concatenate(parts), device)
            cuts = np.cumsum([0] + [p.size for p in parts])
            tabs = cache[key] = dict(n_panels=int(plan["panel"].shape[0]), n_steps=int(S), max_steps=int(plan["panel"][:, 2].max()), panel=buf[cuts[0]:cuts[1]], rows=buf[cuts[1]:cuts[2]],
                                     sinfo=buf[cuts[2]:cuts[3]], dt_bits=buf[cuts[3]:cuts[4]])
        return tabs

    def gi_shared(self, device):
        """Input-gate sharing of a program whose x rows repeat (`x_src`, set by the owner of the program: a label per x row,
        equal labels = equal rows -- e.g. the layer-output row a chain row was gathered from; an entity visited at p positions
        of a window whose snapshot at those positions is the same appears p times).  The gates are a function of the x row
        alone, so each group computes them once per DISTINCT row:
          -> dict(rep [per group: int32 device table, x row (inside the group) of every distinct row],
                  g0 [per group: first row of the group's block in the shared gi buffer], rows (total distinct rows),
                  gi_index int32 device [n_total]: gi row of chain row i  (TempGruChain.gi_index))
        or None when there is nothing to share (no labels, or fewer than 1 row in 8 repeats).  Cached per device."""
        src = getattr(self, "x_src", None)
        if src is None:
            return None
        c = getattr(self, "_gi_shared", None)
        if c is not None and c[0] == str(device):
            return c[1]
        src = np.asarray(src)
        reps, g0, index, total = [], [], np.zeros(self.n_total, dtype=np.int32), 0
        n_labels = int(src.max()) + 1 if src.size else 0
        for g in self.groups:
            assert g["x1"] - g["x0"] == g["h1"] - g["h0"]
            first, inv = _hostlib.unique_labels(src[g["x0"]:g["x1"]], n_labels)     # (numpy.unique's first / inverse, O(n))
            reps.append(first)
            g0.append(total)
            index[g["h0"]:g["h1"]] = total + inv.reshape(-1)
            total += first.size
        res = None
        if total * 8 <= self.n_total * 7:
            buf = _lib.to_device(np.concatenate(reps + [index]), device)
            cuts = np.cumsum([0] + [r.size for r in reps])
            res = dict(rep=[buf[cuts[i]:cuts[i + 1]] for i in range(len(reps))], g0=g0, rows=int(total), gi_index=buf[cuts[-1]:])
        self._gi_shared = (str(device), res)
        return res

    def constants(self, device, d):
        """(zero previous-state row, all -1 row map long enough for any first-position instance), created once per program."""
        key = (str(device), int(d))
        c = getattr(self, "_const", None)
        if c is None or c[0] != key:
            n0 = max([it.n for it in self.inst if it.prev < 0] + [1])
            c = self._const = (key, torch.zeros(1, d, dtype=torch.float32, device=device), torch.full((n0,), -1, dtype=torch.int32, device=device))
        return c[1], c[2]

    def upload(self, device):
        """Row maps and time gaps of every instance on `device`: packed on the host into ONE int32 buffer (the float gaps as
        raw bits) and uploaded with one copy; the per-instance tensors are views."""
        if self.dev is not None and self.dev[0] == str(device):
            return self.dev[1]
        parts, spans, off = [], [], 0
        for it in self.inst:
            span = []
            for arr in (it.prev_idx.astype(np.int32) if it.prev >= 0 else None, it.next_idx,
                        np.ascontiguousarray(it.dt, dtype=np.float32).view(np.int32)):
                if arr is None:
                    span.append(None)
                    continue
                parts.append(arr.reshape(-1))
                span.append((off, off + arr.size))
                off += arr.size
            spans.append(span)
        buf = _lib.to_device(np.concatenate(parts) if parts else np.zeros(0, np.int32), device)
        cut = lambda sp: buf[sp[0]:sp[1]] if sp is not None else None
        t = [(cut(a), cut(b), cut(c).view(torch.float32)) for a, b, c in spans]
        self.dev = (str(device), t)
        return t


class _GruChainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_all, prog, lam, variant, n_rnn, want, x_keys, *weights):
        be = get_backend()
        dev = x_all.device
        d = x_all.shape[1]
        G = weights[0].shape[0]                      # 3d (torch) or d (type-1)
        N = prog.n_total
        x_all = x_all.detach().contiguous()
        W = [tuple(w.detach().contiguous() for w in weights[4 * r:4 * r + 4]) for r in range(n_rnn)]   # (w_ih, w_hh, b_ih, b_hh)
        # persistent chain kernels (one launch for ALL positions) when the backend has them and the program is a set of chains
        tabs = None
        if CHAIN_KERNELS and hasattr(be, "gru_chain_fwd") and n_rnn <= _lib.CHAIN_MAX_RNN and be.gru_chain_supported(d):
            tabs = prog.chain_tables(dev, want)
        share = prog.gi_shared(dev) if tabs is not None else None        # gates once per distinct x row (chain kernels only)
        # x_keys = (row keys [x rows], column keys [d]) of x_all from its producer (functional.gather_rows(keys=True)): the input-gate
        # product and the weight gradients then run on the f16 pipe without a pass over x of their own
        ctx.x_keys = x_keys if (x_keys is not None and tabs is not None) else None
        gk = (lambda: dict(x_keys=[ctx.x_keys[0][g["x0"]:g["x1"]] for g in prog.groups])) if ctx.x_keys is not None else dict
        gi_index = None
        if share is not None:
            gi_index = share["gi_index"]
            gi = torch.empty(share["rows"], G, dtype=torch.float32, device=dev)
            cut = share["g0"] + [share["rows"]]
            be.gru_input_gates_multi([x_all[g["x0"]:g["x1"]] for g in prog.groups], [W[g["rnn"]][0] for g in prog.groups],
                                     [W[g["rnn"]][2] for g in prog.groups], variant, [gi[cut[i]:cut[i + 1]] for i in range(len(prog.groups))],
                                     x_idx=share["rep"], **gk())
        else:
            gi = torch.empty(N, G, dtype=torch.float32, device=dev)
        if share is not None:
            pass
        elif len(prog.groups) > 1 and hasattr(be, "gru_input_gates_multi"):    # both directions' input gates in one launch
            be.gru_input_gates_multi([x_all[g["x0"]:g["x1"]] for g in prog.groups], [W[g["rnn"]][0] for g in prog.groups],
                                     [W[g["rnn"]][2] for g in prog.groups], variant, [gi[g["h0"]:g["h1"]] for g in prog.groups], **gk())
        else:
            for g in prog.groups:
                w_ih, _, b_ih, _ = W[g["rnn"]]
                be.gru_input_gates(x_all[g["x0"]:g["x1"]], w_ih, b_ih, variant, gi[g["h0"]:g["h1"]])
        H = torch.empty(N, d, dtype=torch.float32, device=dev)
        saved = torch.empty(5, N, d, dtype=torch.float32, device=dev)
        packs = None
        if tabs is not None:
            packs = be.gru_chain_pack_multi([W[r][1] for r in range(n_rnn)]) if hasattr(be, "gru_chain_pack_multi") else \
                [be.gru_chain_pack(W[r][1]) for r in range(n_rnn)]
            be.gru_chain_fwd(tabs, gi, lam, variant, packs, [W[r][3] for r in range(n_rnn)], H, saved, gi_index=gi_index)
        else:
            tens = prog.upload(dev)
            zero, none_idx = prog.constants(dev, d)
        for level in (prog.levels if tabs is None else ()):
            cells = []
            for i in level:
                it = prog.inst[i]
                pi, _, dt = tens[i]
                _, w_hh, _, b_hh = W[it.rnn]
                if it.prev >= 0:
                    p = prog.inst[it.prev]
                    prev, pidx = H[p.h0:p.h0 + p.n], pi
                else:                                 # no history yet: every previous state is zero (pointwise cell)
                    prev, pidx = None, none_idx[:it.n]
                cells.append(dict(gi=gi[it.h0:it.h0 + it.n], prev=prev, prev_idx=pidx, dt=dt, w_hh=w_hh, b_hh=b_hh,
                                  h_out=H[it.h0:it.h0 + it.n], row0=it.h0))
            be.gru_cell_fwd_multi(cells, lam, variant, saved)
        ctx.set_materialize_grads(False)             # states nobody differentiates (the history handed to the all-entity pass in an
                                                     # encoder-only step) come back as None, not as zero-filled (n, d) tensors
        ctx.save_for_backward(x_all, saved, *[w for ws in W for w in ws])
        ctx.prog, ctx.lam, ctx.variant, ctx.n_rnn, ctx.G, ctx.want = prog, lam, variant, n_rnn, G, want
        ctx.tabs, ctx.packs = tabs, packs
        if want is None:
            return H
        # only these instances' states are consumed downstream: hand them out as row ranges of H, so that the backward receives
        # one small gradient per range instead of autograd zero-filling and summing full-size (n_total, d) tensors per slice
        return tuple(H[prog.inst[i].h0:prog.inst[i].h0 + prog.inst[i].n] for i in want)

    @staticmethod
    def backward(ctx, *d_outs):
        be = get_backend()
        x_all, saved = ctx.saved_tensors[:2]
        flat = ctx.saved_tensors[2:]
        W = [flat[4 * r:4 * r + 4] for r in range(ctx.n_rnn)]
        prog, lam, variant, G = ctx.prog, ctx.lam, ctx.variant, ctx.G
        dev, d, N = x_all.device, x_all.shape[1], prog.n_total
        if ctx.want is None:
            dH = d_outs[0].contiguous() if d_outs[0] is not None else torch.zeros(N, d, dtype=torch.float32, device=dev)
            up = lambda i, it: dH[it.h0:it.h0 + it.n]
        else:
            given = {i: g.contiguous() for i, g in zip(ctx.want, d_outs) if g is not None}
            up = lambda i, it: given.get(i)                  # None: no upstream gradient for this instance's rows
        groups = list(prog.groups)
        zero_state = [all(it.prev < 0 for it in prog.inst if it.group == gi) for gi in range(len(groups))]    # hdec = 0 on every row
        # GRUs with disjoint x rows (the two directions of a bidirectional chain, or the one GRU of a uni-directional one): ONE
        # weight-gradient launch for all d_W_ih / d_W_hh products, ONE reduction and ONE d_x launch
        disjoint = all(groups[a]["x1"] <= groups[b]["x0"] or groups[b]["x1"] <= groups[a]["x0"] for a in range(len(groups)) for b in range(a))
        one_launch = bool(groups and disjoint and not any(zero_state) and len({g["rnn"] for g in groups}) == len(groups) and WEIGHT_GRADS_MULTI)
        # ... and when the library takes these shapes, the chain backward writes its gate gradients ONCE, [dr | dz | dn_i | dn_h]
        # (two thirds of dgh repeat dgi), for temp_gru_grads_g4
        once = bool(one_launch and GATE_GRADS_ONCE and ctx.tabs is not None and hasattr(be, "gru_grads_g4")
                    and be.gru_grads_g4_supported([g["h1"] - g["h0"] for g in groups], d, variant))
        if once:
            g4 = torch.empty(N, 4 * d, dtype=torch.float32, device=dev)
            ups = [dH] if ctx.want is None else [given.get(i) for i in ctx.want]
            # (f16 arithmetic: the chain backward also hands out the magnitude keys of g4 -- per row, per GRU and column -- that the
            #  weight-gradient and d_x products split it with; integer maxima, so bit-repeatable)
            keyed = bool(KEYED_GRADS and hasattr(be, "gru_chain_keys_supported") and be.gru_chain_keys_supported(d))
            keys = (torch.empty(N, dtype=torch.int32, device=dev), torch.empty(ctx.n_rnn + ctx.tabs["n_panels"], 4 * d, dtype=torch.int32, device=dev)) if keyed else None
            be.gru_chain_bwd_g4(ctx.tabs, saved, ups, lam, variant, ctx.packs, [W[r][3] for r in range(ctx.n_rnn)], g4, **({"keys": keys} if keyed else {}))
            d_x_all = torch.empty_like(x_all)
            xsl = [slice(g["x0"], g["x1"]) for g in groups]
            hsl = [slice(g["h0"], g["h1"]) for g in groups]
            kw = dict(row_keys=[keys[0][b] for b in hsl], col_keys=[keys[1][g["rnn"]] for g in groups]) if keyed else {}
            if keyed and ctx.x_keys is not None:
                kw["x_col_keys"] = [ctx.x_keys[1]] * len(groups)   # (a bound over ALL rows of x_all bounds every group's rows)
            multi = be.gru_grads_g4([x_all[a] for a in xsl], [saved[4, b] for b in hsl], [g4[b] for b in hsl], [W[g["rnn"]][0] for g in groups],
                                    [d_x_all[a] for a in xsl], **kw)
            grads = [None] * (4 * ctx.n_rnn)
            covered = np.zeros(x_all.shape[0], dtype=bool)
            for g, gw in zip(groups, multi):
                covered[g["x0"]:g["x1"]] = True
                for k in range(4):
                    grads[4 * g["rnn"] + k] = gw[k]
            if not covered.all():
                d_x_all[torch.from_numpy(~covered).to(dev)] = 0
            return (d_x_all, None, None, None, None, None, None) + tuple(grads)
        dgi = torch.empty(N, G, dtype=torch.float32, device=dev)
        dgh = torch.empty(N, 3 * d, dtype=torch.float32, device=dev)
        if ctx.tabs is not None:
            ups = [dH] if ctx.want is None else [given.get(i) for i in ctx.want]
            be.gru_chain_bwd(ctx.tabs, saved, ups, lam, variant, ctx.packs, [W[r][3] for r in range(ctx.n_rnn)], dgi, dgh)
        else:
            tens = prog.upload(dev)
            decv = torch.empty(N, dtype=torch.float32, device=dev)
            d_prev = torch.empty(N, d, dtype=torch.float32, device=dev)
        for level in (reversed(prog.levels) if ctx.tabs is None else ()):
            cells = []
            for i in level:
                it = prog.inst[i]
                _, ni, dt = tens[i]
                nxt = prog.inst[it.next] if (it.next >= 0 and prog.inst[it.next].n > 0) else None
                sl = slice(it.h0, it.h0 + it.n)
                cells.append(dict(row0=it.h0, n=it.n, dh_up=up(i, it), d_prev_next=d_prev[nxt.h0:nxt.h0 + nxt.n] if nxt is not None else None,
                                  next_idx=ni if nxt is not None else None, dt=dt, w_hh=W[it.rnn][1], dgi=dgi[sl], dgh=dgh[sl],
                                  decv=decv[sl], d_prev=d_prev[sl], no_prev=it.prev < 0))
            be.gru_cell_bwd_multi(cells, lam, variant, saved)
        d_x_all = torch.empty_like(x_all)
        written = np.zeros(x_all.shape[0], dtype=bool)
        grads = [None] * (4 * ctx.n_rnn)
        multi = None
        if one_launch and hasattr(be, "gru_weight_grads_multi"):
            xsl = [slice(g["x0"], g["x1"]) for g in groups]
            hsl = [slice(g["h0"], g["h1"]) for g in groups]
            multi = be.gru_weight_grads_multi([x_all[a] for a in xsl], [saved[4, b] for b in hsl], [dgi[b] for b in hsl], [dgh[b] for b in hsl],
                                              [W[g["rnn"]][0] for g in groups], variant, [d_x_all[a] for a in xsl])
        if multi is not None:
            for g, gw in zip(groups, multi):
                written[g["x0"]:g["x1"]] = True
                for k in range(4):
                    grads[4 * g["rnn"] + k] = gw[k]
            groups = []
        for g in groups:
            gi_ = prog.groups.index(g)
            xs, hs = slice(g["x0"], g["x1"]), slice(g["h0"], g["h1"])
            first = not written[xs].any()
            assert first or written[xs].all()
            tgt = d_x_all[xs] if first else torch.empty(g["x1"] - g["x0"], d, dtype=torch.float32, device=dev)
            gw = be.gru_weight_grads(x_all[xs], None if zero_state[gi_] else saved[4, hs], dgi[hs], dgh[hs], W[g["rnn"]][0], variant, tgt)
            if not first:
                d_x_all[xs] += tgt
            written[xs] = True
            for k in range(4):                         # (d_w_ih, d_w_hh, d_b_ih, d_b_hh) -> weight order (w_ih, w_hh, b_ih, b_hh)
                j = 4 * g["rnn"] + k
                grads[j] = gw[k] if grads[j] is None else grads[j] + gw[k]
        if not written.all():
            d_x_all[torch.from_numpy(~written).to(dev)] = 0
        return (d_x_all, None, None, None, None, None, None) + tuple(grads)


def chain_kernels_usable(d, n_rnn=1):
    """True when gru_chain() will run a program of this width through the persistent chain kernels."""
    be = get_backend()
    return bool(CHAIN_KERNELS and hasattr(be, "gru_chain_fwd") and n_rnn <= _lib.CHAIN_MAX_RNN and be.gru_chain_supported(d))


def prepare_program(prog, device, d, n_rnn, want):
    """Host + upload half of a program, done once per prepared batch (by the prefetch thread when there is one): the chain
    kernels' panel tables for the `want` set the run will ask for, or -- when the program goes through the per-position
    launches -- the row maps of every instance."""
    if chain_kernels_usable(d, n_rnn) and prog.chain_tables(device, tuple(want) if want is not None else None) is not None:
        _lib.pause_point()
        prog.gi_shared(device)
        return
    prog.upload(device)


def zero_state_program(n):
    """Program of ONE cell over n rows that starts from the zero state (GRU(x, 0): the once-per-entity rows of the all-entity
    pass): hoisted input-gate GEMM + pointwise cell forward; gate gradients, d_x and d_W_ih backward -- no recurrent GEMM."""
    return GruProgram([GruInstance(n, 0, 0, -1, np.full(n, -1, dtype=np.int32), np.zeros(n, dtype=np.float32))])


def gru_chain(x_all, prog, rnns, lam, type1=False, want=None, x_keys=None):
    """Run a GruProgram.  x_keys: (row keys, column keys) of x_all from functional.gather_rows(keys=True), or None.  `rnns`: list of modules holding (weight_ih, weight_hh, bias_ih, bias_hh)
    (nn.GRU layer 0 or the type-1 GRUCell).  Returns H_all (prog.n_total, d), or -- with `want` = a list of instance ids --
    the states of just those instances (a tuple of (n_i, d) tensors; instances with 0 rows give empty tensors)."""
    ws = []
    for r in rnns:
        if type1:
            ws += [r.weight_ih, r.weight_hh, r.bias_ih, r.bias_hh]
        else:
            ws += [r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0]
    return _GruChainFn.apply(x_all, prog, float(lam), _lib.GRU_TYPE1 if type1 else _lib.GRU_TORCH, len(rnns),
                             tuple(want) if want is not None else None, x_keys, *ws)
