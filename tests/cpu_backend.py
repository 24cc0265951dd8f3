"""TEST-ONLY stand-in for temp_amd.backend.HipBackend so host logic (graph views, autograd wiring,
window scheduling, sharding) can be exercised on a machine without a GPU.

It follows the C-ABI contract of include/temp_amd.h *through the same sorted/chunked edge views
the kernels read* (so a wrong view shows up here), using torch CPU ops and the oracle's GRU
equations.  It lives under tests/ and is never imported by the product package.
"""
import numpy as np
import torch

from oracle import temp_oracle as O
from temp_amd import _lib


def _view(dg, name):
    v = dg.views[name]
    t = lambda k: torch.from_numpy(v[k]).long()
    return v, t


def _chunk_reduce(dg, name, per_edge_fn, width, n_out, dtype):
    """Run `per_edge_fn(a, b) -> (E, width)` over the edges of a view in sorted order and reduce
    exactly like the kernels: per chunk, then ordered partial slots -> fix-up."""
    v, t = _view(dg, name)
    out = torch.full((n_out, width), float('nan'), dtype=dtype)      # untouched rows stay NaN (kernel leaves them)
    if v['n_chunks'] == 0:
        return out
    beg, end, seg, slot = t('chunk_beg'), t('chunk_end'), t('chunk_seg'), t('chunk_slot')
    cnt = end - beg
    assert int(cnt.min()) >= 1 and int(cnt.max()) <= (_lib.CHUNK_REL if name == 'by_rel' else _lib.CHUNK)
    assert int(cnt.sum()) == v['n_edges'] and torch.equal(beg[1:], end[:-1]) and int(beg[0]) == 0
    cid = torch.repeat_interleave(torch.arange(v['n_chunks']), cnt)
    vals = per_edge_fn(t('a'), t('b'))
    sums = torch.zeros(v['n_chunks'], width, dtype=dtype).index_add(0, cid, vals)
    direct = slot < 0
    out[seg[direct]] = sums[direct]
    partial = torch.zeros(max(v['n_partial'], 1), width, dtype=dtype)
    partial[slot[~direct]] = sums[~direct]
    for fs, f0, fc in zip(v['fix_seg'], v['fix_slot'], v['fix_cnt']):
        acc = torch.zeros(width, dtype=dtype)
        for s in range(int(fc)):
            acc = acc + partial[int(f0) + s]
        out[int(fs)] = acc
    return out


def drop_mask(drop, n, d):
    """Keep-scale matrix of the self-loop dropout, the same counter-based hash as csrc/common.hpp: drop_scale()."""
    import numpy as np
    if drop is None or drop[0] <= 0.0:
        return None
    p, seed = float(drop[0]), np.uint64(int(drop[1]) & 0xFFFFFFFFFFFFFFFF)
    row = np.arange(n, dtype=np.uint64).reshape(-1, 1)
    col = np.arange(d, dtype=np.uint64).reshape(1, -1)
    with np.errstate(over="ignore"):
        x = seed ^ ((row << np.uint64(32)) | col)
        x ^= x >> np.uint64(33); x *= np.uint64(0xff51afd7ed558ccd); x ^= x >> np.uint64(33)
        x *= np.uint64(0xc4ceb9fe1a85ec53); x ^= x >> np.uint64(33)
    u = (x >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    keep = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return torch.from_numpy(np.where(u < np.float32(p), np.float32(0.0), keep).astype(np.float32))


def _blocks(weight, rel, B):
    return weight.index_select(0, rel).view(rel.shape[0], B, -1)


class CpuTestBackend:
    name = "cpu-test"

    # ---- RGCN ---------------------------------------------------------------------------------
    def rgcn_fwd(self, dg, h, h_ids, weight, loop_w, bias, num_bases, act, drop=None, out=None):
        if out is not None:
            out.copy_(self.rgcn_fwd(dg, h, h_ids, weight, loop_w, bias, num_bases, act, drop))
            return out
        h, weight, loop_w = h.detach(), weight.detach(), loop_w.detach()
        d_in, d_out = loop_w.shape
        si, so = d_in // num_bases, d_out // num_bases
        nn = dg.nnorm.cpu()[:dg.n_nodes]
        rows = (lambda idx: h[h_ids.long()[idx]]) if h_ids is not None else (lambda idx: h[idx])

        def msg(a, b):
            w = weight.index_select(0, b).view(-1, si, so)
            return torch.bmm(rows(a).view(-1, 1, si), w).view(-1, d_out)

        agg = _chunk_reduce(dg, 'by_dst', msg, d_out, dg.n_nodes, h.dtype)
        agg = agg * (nn * nn).view(-1, 1)
        has_in = dg.in_deg.cpu().long() > 0
        agg = torch.where(has_in.view(-1, 1), agg, torch.zeros_like(agg))
        x = rows(torch.arange(dg.n_nodes))
        loop = torch.mm(x, loop_w)
        m = drop_mask(drop, loop.shape[0], loop.shape[1])
        out = agg + (loop * m if m is not None else loop)
        if bias is not None:
            out = out + bias.detach()
        if act == _lib.ACT_RELU:
            out = torch.relu(out)
        return out

    def rgcn_bwd(self, dg, h, out, d_out_grad, weight, loop_w, has_bias, num_bases, act, drop=None):
        h, weight, loop_w = h.detach(), weight.detach(), loop_w.detach()
        d_in, d_out = loop_w.shape
        si, so = d_in // num_bases, d_out // num_bases
        nn = dg.nnorm.cpu()[:dg.n_nodes]
        dz = d_out_grad.detach()
        if act == _lib.ACT_RELU:
            dz = torch.where(out.detach() > 0, dz, torch.zeros_like(dz))

        def dx_edge(dst, rel):
            w = weight.index_select(0, rel).view(-1, si, so)
            g = (dz[dst] * (nn[dst] ** 2).view(-1, 1)).view(-1, so, 1)
            return torch.bmm(w, g).view(-1, d_in)

        d_h = _chunk_reduce(dg, 'by_src', dx_edge, d_in, dg.n_nodes, h.dtype)
        has_out = dg.out_deg.cpu().long() > 0
        m = drop_mask(drop, dz.shape[0], dz.shape[1])
        dzm = dz * m if m is not None else dz
        d_h = torch.where(has_out.view(-1, 1), d_h, torch.zeros_like(d_h)) + torch.mm(dzm, loop_w.t())

        def dw_edge(src, dst):
            g = (dz[dst] * (nn[dst] ** 2).view(-1, 1)).view(-1, num_bases, 1, so)
            x = h[src].view(-1, num_bases, si, 1)
            return (x * g).reshape(src.shape[0], -1)

        wrow = num_bases * si * so
        d_w = _chunk_reduce(dg, 'by_rel', dw_edge, wrow, weight.shape[0], h.dtype)
        d_w = torch.where(torch.isnan(d_w), torch.zeros_like(d_w), d_w)      # kernel memsets dW first
        d_loop = torch.mm(h.t(), dzm)
        d_bias = dz.sum(0) if has_bias else None
        return d_h, d_w, d_loop, d_bias

    # the two halves of rgcn_bwd for a layer inside a recurrence (HipBackend.rgcn_bwd_dh / rgcn_bwd_weights; rec_stack.py's hoist)
    def rgcn_bwd_dh(self, dg, out, d_out_grad, weight, loop_w, num_bases, act, drop=None, dz_out=None, dzm_out=None):
        weight, loop_w = weight.detach(), loop_w.detach()
        d_in, d_out = loop_w.shape
        si, so = d_in // num_bases, d_out // num_bases
        nn = dg.nnorm.cpu()[:dg.n_nodes]
        dz = d_out_grad.detach()
        if act == _lib.ACT_RELU:
            dz = torch.where(out.detach() > 0, dz, torch.zeros_like(dz))
            dz_out.copy_(dz)                       # (the weight pass reads the masked gradient: an output of this half)

        def dx_edge(dst, rel):
            w = weight.index_select(0, rel).view(-1, si, so)
            g = (dz[dst] * (nn[dst] ** 2).view(-1, 1)).view(-1, so, 1)
            return torch.bmm(w, g).view(-1, d_in)

        d_h = _chunk_reduce(dg, 'by_src', dx_edge, d_in, dg.n_nodes, dz.dtype)
        has_out = dg.out_deg.cpu().long() > 0
        m = drop_mask(drop, dz.shape[0], dz.shape[1])
        dzm = dz * m if m is not None else dz
        if m is not None and dzm_out is not None:
            dzm_out.copy_(dzm)
        return torch.where(has_out.view(-1, 1), d_h, torch.zeros_like(d_h)) + torch.mm(dzm, loop_w.t())

    def rgcn_bwd_weights(self, dg, h, dz, dzm, weight_like, loop_like, has_bias, num_bases):
        h, dz = h.detach(), dz.detach()
        dzm = dz if dzm is None else dzm.detach()
        d_in, d_out = loop_like.shape
        si, so = d_in // num_bases, d_out // num_bases
        nn = dg.nnorm.cpu()[:dg.n_nodes]

        def dw_edge(src, dst):
            g = (dz[dst] * (nn[dst] ** 2).view(-1, 1)).view(-1, num_bases, 1, so)
            x = h[src].view(-1, num_bases, si, 1)
            return (x * g).reshape(src.shape[0], -1)

        d_w = _chunk_reduce(dg, 'by_rel', dw_edge, num_bases * si * so, weight_like.shape[0], h.dtype)
        d_w = torch.where(torch.isnan(d_w), torch.zeros_like(d_w), d_w)      # kernel memsets dW first
        return d_w, torch.mm(h.t(), dzm), (dz.sum(0) if has_bias else None)

    def rgcn_table_fwd(self, dg, table, ids, weight, loop_w, bias, num_bases, act, drop=None):
        return self.rgcn_fwd(dg, table.detach()[ids.long()], None, weight, loop_w, bias, num_bases, act, drop)

    def rgcn_table_bwd(self, dg, table, ids, inverse, out, d_out_grad, weight, loop_w, has_bias, num_bases, act, drop=None):
        d_h, d_w, d_loop, d_bias = self.rgcn_bwd(dg, table.detach()[ids.long()], out, d_out_grad, weight, loop_w, has_bias, num_bases, act, drop)
        d_table = self.segment_sum_rows(d_h, inverse[0], inverse[1], table.shape[0])
        return d_table, d_w, d_loop, d_bias

    def rgcn_isolated_fwd(self, e, loop_w, bias, act, drop=None):
        loop = torch.mm(e.detach(), loop_w.detach())
        m = drop_mask(drop, loop.shape[0], loop.shape[1])
        out = e.detach() + (loop * m if m is not None else loop)
        if bias is not None:
            out = out + bias.detach()
        return torch.relu(out) if act == _lib.ACT_RELU else out

    def rgcn_isolated_bwd(self, e, out, d_out_grad, loop_w, has_bias, act, drop=None):
        dz = d_out_grad.detach()
        if act == _lib.ACT_RELU:
            dz = torch.where(out.detach() > 0, dz, torch.zeros_like(dz))
        m = drop_mask(drop, dz.shape[0], dz.shape[1])
        dzm = dz * m if m is not None else dz
        d_e = dz + torch.mm(dzm, loop_w.detach().t())
        return d_e, torch.mm(e.detach().t(), dzm), (dz.sum(0) if has_bias else None)

    # ---- decay + GRU ----------------------------------------------------------------------------
    @staticmethod
    def _gru_forward(x, prev, prev_idx, dt, lam, wb, w_ih, w_hh, b_ih, b_hh, variant):
        if prev_idx is not None:
            idx = prev_idx.long()
            rows = prev[idx.clamp(min=0)] * (idx >= 0).to(prev.dtype).view(-1, 1)
        else:
            rows = prev
        dt = dt.view(-1, 1)
        dec = torch.exp(-torch.clamp(dt * wb[0] + wb[1], min=0)) if wb is not None else torch.exp(-dt * lam)
        hdec = rows * dec
        fn = O.gru_type1 if variant == _lib.GRU_TYPE1 else O.gru_torch
        return fn(x, hdec, w_ih, w_hh, b_ih, b_hh), hdec

    def gru_fwd(self, x, prev, prev_idx, dt, lam, decay_wb, w_ih, w_hh, b_ih, b_hh, variant):
        args = [t.detach() if t is not None else None for t in (x, prev, None, dt, None, decay_wb, w_ih, w_hh, b_ih, b_hh)]
        h, hdec = self._gru_forward(args[0], args[1], prev_idx, args[3], lam, args[5], args[6], args[7], args[8], args[9], variant)
        # `saved` is opaque to callers; this backend only needs the biases back in gru_bwd
        return h, torch.cat([b_ih.detach().reshape(-1), b_hh.detach().reshape(-1)])

    def gru_bwd(self, x, prev, prev_idx, dt, lam, decay_wb, w_ih, w_hh, saved, d_h, variant):
        leaf = lambda t: t.detach().clone().requires_grad_(True)
        xs, ps, wi, wh = leaf(x), leaf(prev), leaf(w_ih), leaf(w_hh)
        bi, bh = leaf(saved[:w_ih.shape[0]]), leaf(saved[w_ih.shape[0]:])
        wb = leaf(decay_wb) if decay_wb is not None else None
        # gradient w.r.t. the GATHERED previous rows (the ABI returns d_prev in x's row order)
        if prev_idx is not None:
            idx = prev_idx.long()
            rows = (ps.detach()[idx.clamp(min=0)] * (idx >= 0).to(ps.dtype).view(-1, 1)).clone().requires_grad_(True)
        else:
            rows = ps
        with torch.enable_grad():            # we are called from inside an autograd backward
            h, _ = self._gru_forward(xs, rows, None, dt.detach(), lam, wb, wi, wh, bi, bh, variant)
            h.backward(d_h.detach())
        z = lambda t, like: t.grad if t.grad is not None else torch.zeros_like(like)
        return (z(xs, xs), z(rows, rows), z(wi, wi), z(wh, wh), z(bi, bi), z(bh, bh), (z(wb, wb) if wb is not None else None))

    # ---- window-batched recurrence (same contract as the HIP entry points) ---------------------
    def gru_input_gates(self, x, w_ih, b_ih, variant, out):
        out.copy_(torch.mm(x.detach(), w_ih.detach().t()) + b_ih.detach())

    def gru_input_gates_multi(self, xs, w_ihs, b_ihs, variant, outs, x_idx=None):
        for i, (x, w, b, o) in enumerate(zip(xs, w_ihs, b_ihs, outs)):
            t = None if x_idx is None else x_idx[i]
            self.gru_input_gates(x if t is None else x[t.long()], w, b, variant, o)

    def gru_cell_fwd(self, gi, prev, prev_idx, dt, lam, w_hh, b_hh, variant, h_out, saved_all, row0):
        n, d = h_out.shape
        if prev is None:                                        # zero-state cell
            prev, prev_idx = torch.zeros(n, d), None
        prev, w_hh, b_hh = prev.detach(), w_hh.detach(), b_hh.detach()
        if prev.shape[0] == 0:                                  # the previous position was empty: every row map entry is -1
            prev = torch.zeros(1, d)
        if prev_idx is not None:
            idx = prev_idx.long()
            rows = prev[idx.clamp(min=0)] * (idx >= 0).to(prev.dtype).view(-1, 1)
        else:
            rows = prev
        hd = rows * torch.exp(-dt.view(-1, 1) * lam)
        gh = torch.mm(hd, w_hh.t()) + b_hh
        h_r, h_z, h_n = gh.chunk(3, 1)
        if variant == _lib.GRU_TORCH:
            i_r, i_z, i_n = gi.chunk(3, 1)
            r, z = torch.sigmoid(i_r + h_r), torch.sigmoid(i_z + h_z)
        else:
            i_n = gi
            r, z = torch.sigmoid(h_r), torch.sigmoid(h_z)
        nn_ = torch.tanh(i_n + r * h_n)
        h_out.copy_((1 - z) * nn_ + z * hd)
        for k, v in enumerate((r, z, nn_, h_n, hd)):
            saved_all[k, row0:row0 + n] = v

    def gru_cell_bwd(self, saved_all, row0, n, dh_up, d_prev_next, next_idx, dt, lam, w_hh, variant, dgi, dgh, decv, d_prev):
        r, z, nn_, hn, hd = (saved_all[k, row0:row0 + n] for k in range(5))
        g = dh_up.detach().clone() if dh_up is not None else torch.zeros_like(r)
        if next_idx is not None:
            i = next_idx.long()
            g = g + d_prev_next[i.clamp(min=0)] * (i >= 0).to(g.dtype).view(-1, 1)
        dn_pre = g * (1 - z) * (1 - nn_ * nn_)
        dz_pre = g * (hd - nn_) * z * (1 - z)
        dr_pre = dn_pre * hn * r * (1 - r)
        if variant == _lib.GRU_TORCH:
            dgi.copy_(torch.cat([dr_pre, dz_pre, dn_pre], 1))
        else:
            dgi.copy_(dn_pre)
        dgh.copy_(torch.cat([dr_pre, dz_pre, dn_pre * r], 1))
        dec = torch.exp(-dt.view(-1) * lam)
        decv.copy_(dec)
        d_prev.copy_((torch.mm(dgh, w_hh.detach()) + g * z) * dec.view(-1, 1))

    def gru_cell_fwd_multi(self, cells, lam, variant, saved_all):
        for c in cells:
            self.gru_cell_fwd(c["gi"], c["prev"], c["prev_idx"], c["dt"], lam, c["w_hh"], c["b_hh"], variant, c["h_out"], saved_all, c["row0"])

    def gru_cell_bwd_multi(self, cells, lam, variant, saved_all):
        for c in cells:
            self.gru_cell_bwd(saved_all, c["row0"], c["n"], c["dh_up"], c["d_prev_next"], c["next_idx"], c["dt"], lam, c["w_hh"], variant,
                              c["dgi"], c["dgh"], c["decv"], c["d_prev"])

    @staticmethod
    def subsample_keys(seed, E):
        """numpy twin of sub_key (temp_amd/csrc/store_kernels.hip): (hash(seed, e) << 32) | e for e in [0, E)."""
        M = np.uint64(0xFFFFFFFFFFFFFFFF)
        with np.errstate(over="ignore"):
            e = np.arange(E, dtype=np.uint64)
            x = np.uint64(int(seed) & 0xFFFFFFFFFFFFFFFF) ^ (np.uint64(0x9e3779b97f4a7c15) * (e + np.uint64(1)))
            x ^= x >> np.uint64(33)
            x *= np.uint64(0xff51afd7ed558ccd)
            x ^= x >> np.uint64(33)
            x *= np.uint64(0xc4ceb9fe1a85ec53)
            x ^= x >> np.uint64(33)
        return ((x & np.uint64(0xffffffff00000000)) | e) & M

    def subsample_views(self, jobs):
        """Reference of temp_subsample_views on host tensors: keep the `keep` smallest keys; per chunk of every view move the
        kept edges to the front (order preserved), shrink chunk_end, count degrees, nnorm = 1 / in_deg."""
        for j in jobs:
            E, k, n = j["n_edges"], j["keep"], j["n_nodes"]
            parent, child = j["parent"].numpy(), j["child"].numpy()
            keys = self.subsample_keys(j["seed"], E)
            keep = np.zeros(E, dtype=bool)
            if k > 0:
                keep[np.argsort(keys, kind="stable")[:k]] = True
            if j.get("keep_mask") is not None:
                j["keep_mask"].copy_(torch.from_numpy(keep.astype(np.uint8)))
            eid = j["eid"].numpy().reshape(3, -1) if E else np.zeros((3, 0), np.int64)
            child[j["off_in_deg"]:j["off_in_deg"] + n] = 0
            child[j["off_out_deg"]:j["off_out_deg"] + n] = 0
            for v in range(3):
                nch = j["n_chunks"][v]
                beg = parent[j["off_chunk_beg"][v]:j["off_chunk_beg"][v] + nch]
                end = parent[j["off_chunk_end"][v]:j["off_chunk_end"][v] + nch]
                seg = parent[j["off_chunk_seg"][v]:j["off_chunk_seg"][v] + nch]
                for c in range(nch):
                    pos = np.arange(beg[c], end[c])
                    kp = pos[keep[eid[v][pos]]]
                    m = kp.shape[0]
                    child[j["off_a"][v] + beg[c]:j["off_a"][v] + beg[c] + m] = parent[j["off_a"][v] + kp]
                    child[j["off_b"][v] + beg[c]:j["off_b"][v] + beg[c] + m] = parent[j["off_b"][v] + kp]
                    child[j["off_chunk_end"][v] + c] = beg[c] + m
                    if v == 0:
                        child[j["off_in_deg"] + seg[c]] += m
                    elif v == 1:
                        child[j["off_out_deg"] + seg[c]] += m
            deg = child[j["off_in_deg"]:j["off_in_deg"] + n]
            nn = np.where(deg > 0, 1.0 / np.maximum(deg, 1), 0.0).astype(np.float32)
            child[j["off_nnorm"]:j["off_nnorm"] + n] = nn.view(np.int32)

    def decay_rows(self, x, dt, lam):
        return x.detach() * torch.exp(-dt.detach().view(-1, 1) * lam)

    # ---- persistent window chain: the SAME tables the HIP kernels walk (include/temp_amd.h: TempGruChain), panel by panel ----
    def gru_chain_supported(self, d):
        return d % 4 == 0

    def gru_chain_pack(self, w_hh):
        return w_hh.detach().clone()

    @staticmethod
    def _chain_tables(tabs):
        P, S = tabs["n_panels"], tabs["n_steps"]
        return (tabs["panel"].view(P, 4).long(), tabs["rows"].view(S, _lib.CHAIN_TRACKS).long(), tabs["sinfo"].view(S, 4).long(),
                tabs["dt_bits"].view(torch.float32))

    def gru_chain_fwd(self, tabs, gi, lam, variant, packs, b_hhs, h_out, saved_all, gi_index=None):
        panel, rows, sinfo, dt = self._chain_tables(tabs)
        if gi_index is not None:                   # TempGruChain.gi_index: chain rows that share an input row share a gi row
            gi = gi[gi_index.long()]
        d = saved_all.shape[2]
        mask = _lib.CHAIN_HAS_PREV - 1
        for rnn, s0, ns, _ in panel.tolist():
            w_hh, b_hh = packs[rnn], b_hhs[rnn].detach()
            state = torch.zeros(_lib.CHAIN_TRACKS, d)
            for s in range(s0, s0 + ns):
                e = rows[s]
                act = e >= 0
                r = (e & mask)[act]
                hp = (((e >> 30) & 1) == 1)[act]
                assert bool(hp.any()) == bool(sinfo[s, 0] & 1)
                hd = state[act] * torch.exp(-dt[r] * lam).view(-1, 1) * hp.view(-1, 1).to(state.dtype)
                gh = torch.mm(hd, w_hh.t()) + b_hh
                h_r, h_z, h_n = gh.chunk(3, 1)
                g = gi[r]
                if variant == _lib.GRU_TORCH:
                    i_r, i_z, i_n = g.chunk(3, 1)
                    rg, zg = torch.sigmoid(i_r + h_r), torch.sigmoid(i_z + h_z)
                else:
                    i_n = g
                    rg, zg = torch.sigmoid(h_r), torch.sigmoid(h_z)
                ng = torch.tanh(i_n + rg * h_n)
                h = (1 - zg) * ng + zg * hd
                for k, v in enumerate((rg, zg, ng, h_n, hd)):
                    saved_all[k, r] = v
                if sinfo[s, 0] & 2:
                    h_out[r] = h
                state = torch.zeros_like(state)
                state[act] = h

    def gru_chain_bwd(self, tabs, saved_all, ups, lam, variant, packs, b_hhs, dgi, dgh):
        panel, rows, sinfo, dt = self._chain_tables(tabs)
        d = saved_all.shape[2]
        mask = _lib.CHAIN_HAS_PREV - 1
        for rnn, s0, ns, _ in panel.tolist():
            w_hh = packs[rnn]
            dprev = torch.zeros(_lib.CHAIN_TRACKS, d)
            nxt_has = torch.zeros(_lib.CHAIN_TRACKS, dtype=torch.bool)
            for s in range(s0 + ns - 1, s0 - 1, -1):
                e = rows[s]
                act = e >= 0
                r = (e & mask)[act]
                rg, zg, ng, hn, hd = (saved_all[k, r] for k in range(5))
                g = torch.zeros(r.shape[0], d)
                sel = int(sinfo[s, 1])
                if sel >= 0 and ups[sel] is not None:
                    g = g + ups[sel].detach()[r - int(sinfo[s, 2])]
                g = g + dprev[act] * nxt_has[act].view(-1, 1).to(g.dtype)
                dn_pre = g * (1 - zg) * (1 - ng * ng)
                dz_pre = g * (hd - ng) * zg * (1 - zg)
                dr_pre = dn_pre * hn * rg * (1 - rg)
                dgi[r] = torch.cat([dr_pre, dz_pre, dn_pre], 1) if variant == _lib.GRU_TORCH else dn_pre
                gh = torch.cat([dr_pre, dz_pre, dn_pre * rg], 1)
                dgh[r] = gh
                dp = (torch.mm(gh, w_hh) + g * zg) * torch.exp(-dt[r] * lam).view(-1, 1)
                dprev = torch.zeros_like(dprev)
                dprev[act] = dp
                nxt_has = act & (((e >> 30) & 1) == 1)

    # the gate gradients written once, g4 = [dr | dz | dn_i | dn_h] (HipBackend.gru_chain_bwd_g4 / gru_grads_g4: nn.GRU layout only)
    def gru_grads_g4_supported(self, ns, d, variant):
        return 1 <= len(ns) <= 4 and variant == _lib.GRU_TORCH and all(n > 0 for n in ns)

    def gru_chain_bwd_g4(self, tabs, saved_all, ups, lam, variant, packs, b_hhs, g4, keys=None):
        assert keys is None, "the CPU test backend has no magnitude keys (f16 arithmetic is the HIP library's)"
        N, d = saved_all.shape[1], saved_all.shape[2]
        dgi, dgh = torch.zeros(N, 3 * d), torch.zeros(N, 3 * d)
        self.gru_chain_bwd(tabs, saved_all, ups, lam, variant, packs, b_hhs, dgi, dgh)
        g4.copy_(torch.cat([dgi, dgh[:, 2 * d:]], 1))

    def gru_grads_g4(self, xs, hdecs, g4s, w_ihs, d_xs, row_keys=None, col_keys=None, x_col_keys=None):
        out = []
        for x, hdec, g4, w_ih, d_x in zip(xs, hdecs, g4s, w_ihs, d_xs):
            d = x.shape[1]
            dgi = g4[:, :3 * d]
            dgh = torch.cat([g4[:, :2 * d], g4[:, 3 * d:]], 1)
            out.append(self.gru_weight_grads(x, hdec, dgi, dgh, w_ih, _lib.GRU_TORCH, d_x))
        return out

    def gru_weight_grads_multi(self, xs, hdecs, dgis, dghs, w_ihs, variant, d_xs):
        if variant != _lib.GRU_TORCH or any(h is None for h in hdecs):
            return None
        return [self.gru_weight_grads(x, h, a, b, w, variant, dx) for x, h, a, b, w, dx in zip(xs, hdecs, dgis, dghs, w_ihs, d_xs)]

    def gru_weight_grads(self, x, hdec, dgi, dgh, w_ih, variant, d_x):
        if d_x is not None:
            d_x.copy_(torch.mm(dgi, w_ih.detach()))
        d_w_hh = torch.mm(dgh.t(), hdec) if hdec is not None else torch.zeros(dgh.shape[1], x.shape[1])
        return torch.mm(dgi.t(), x.detach()), d_w_hh, dgi.sum(0), dgh.sum(0)

    # ---- plain GEMMs + candidate cross-entropy ------------------------------------------------
    def linear(self, a, b, trans_b, out=None, a_keys=None):
        r = torch.mm(a.detach(), b.detach().t() if trans_b else b.detach())
        if out is not None:
            out.copy_(r)
            return out
        return r

    def linear_tn(self, a, b, out=None):
        r = torch.mm(a.detach().t(), b.detach())
        if out is not None:
            out.copy_(r)
            return out
        return r

    def linear_t(self, a, b, trans_b, out_t):
        out_t.copy_(torch.mm(a.detach(), b.detach().t() if trans_b else b.detach()).t())
        return out_t

    def linear_multi(self, a_list, b_list, trans_b, out):
        row = 0
        for a, b in zip(a_list, b_list):
            out[row:row + a.shape[0]] = torch.mm(a.detach(), b.detach().t() if trans_b else b.detach())
            row += a.shape[0]
        return out

    @staticmethod
    def _bq(kind, ent_rows, known_idx, rel, rel_idx, is_tail):
        from temp_amd import scores as SC
        k = ent_rows[known_idx.long()]
        r = rel[rel_idx.long()]
        if kind == "distmult":
            return SC.bilinear_query(kind, k, r, "tail")
        return torch.where(is_tail.view(-1, 1) != 0, SC.bilinear_query(kind, k, r, "tail"), SC.bilinear_query(kind, k, r, "head"))

    def bilinear_query_fwd(self, kind, ent_rows, known_idx, rel, rel_idx, is_tail):
        return self._bq(kind, ent_rows.detach(), known_idx, rel.detach(), rel_idx, is_tail)

    def bilinear_query_bwd(self, kind, ent_rows, known_idx, rel, rel_idx, is_tail, d_q):
        k = ent_rows.detach()[known_idx.long()].requires_grad_(True)
        r = rel.detach()[rel_idx.long()].requires_grad_(True)
        from temp_amd import scores as SC
        with torch.enable_grad():
            if kind == "distmult":
                q = SC.bilinear_query(kind, k, r, "tail")
            else:
                q = torch.where(is_tail.view(-1, 1) != 0, SC.bilinear_query(kind, k, r, "tail"), SC.bilinear_query(kind, k, r, "head"))
            dk, dr = torch.autograd.grad(q, (k, r), d_q)
        return dk, dr

    def gather_ce_fwd(self, scores, cand):
        logits = scores.detach().gather(1, cand.long())
        lse = torch.logsumexp(logits, dim=1)
        return lse - logits[:, 0], lse

    def gather_ce_bwd(self, scores, cand, lse, scale, inv_rows, row_scale=None):
        c = cand.long()
        g = torch.exp(scores.detach().gather(1, c) - lse.view(-1, 1))
        g[:, 0] -= 1.0
        d = torch.zeros_like(scores)
        w = scale.reshape(-1)[0] * (row_scale.view(-1, 1) if row_scale is not None else inv_rows)
        d.scatter_add_(1, c, g * w)
        return d

    def corrupt_sample(self, seed, truth, lo, hi, ids, K, N):
        """Same contract as the kernel (column 0 = truth, filtered uniform draws), numpy random stream."""
        rng = np.random.default_rng(int(seed) & 0xFFFFFFFF)
        R = truth.shape[0]
        out = rng.integers(0, N, size=(R, K + 1)).astype(np.int32)
        out[:, 0] = truth.numpy()
        if lo is not None and R:
            lo_n, hi_n, ids_n = lo.numpy(), hi.numpy(), ids.numpy()
            for r in range(R):
                bad = ids_n[lo_n[r]:hi_n[r]]
                if bad.size == 0:
                    continue
                for _ in range(64):
                    m = np.isin(out[r, 1:], bad)
                    if not m.any():
                        break
                    out[r, 1:][m] = rng.integers(0, N, size=int(m.sum()))
        return torch.from_numpy(out)

    def filtered_rank(self, scores, target, filt_ptr=None, filt_ids=None):
        s = scores.detach().clone()
        P, N = s.shape
        tgt = target.long()
        if filt_ptr is not None:
            ptr = filt_ptr.long()
            rows = torch.repeat_interleave(torch.arange(P), ptr[1:] - ptr[:-1])
            ids = filt_ids.long()
            keep = ids != tgt[rows]
            s[rows[keep], ids[keep]] = -10e6
        v = torch.sigmoid(s)
        tv = v.gather(1, tgt.view(-1, 1))
        ahead = (v > tv) | ((v == tv) & (torch.arange(N).view(1, -1) < tgt.view(-1, 1)))
        return ahead.sum(dim=1) + 1

    # ---- history attention (same sparse formulation as the kernel, dense torch ops) -------------------
    @staticmethod
    def _attn(qkv, kv_hist, idx, decay):
        n, D = qkv.shape[0], qkv.shape[1] // 3
        dk, T = D // 8, idx.shape[1] + 1
        q, kc, vc = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        i = idx.long()
        live = (i >= 0)
        if T > 1:
            g = kv_hist[i.clamp(min=0).reshape(-1)].view(n, T - 1, 2 * D) * live.unsqueeze(-1).to(qkv.dtype)
            K = torch.cat([g[:, :, :D], kc.unsqueeze(1)], dim=1)
            V = torch.cat([g[:, :, D:], vc.unsqueeze(1)], dim=1)
        else:
            K, V = kc.unsqueeze(1), vc.unsqueeze(1)
        s = torch.einsum('nhd,nthd->nht', q.reshape(n, 8, dk), K.reshape(n, T, 8, dk)) / (dk ** 0.5)
        if decay is not None:
            s = s + decay.view(1, 1, T)
        mask = torch.cat([live, live.new_ones(n, 1)], dim=1).unsqueeze(1)
        s = s.masked_fill(~mask, float('-inf'))
        p = torch.softmax(s, dim=-1)
        o = torch.einsum('nht,nthd->nhd', p, V.reshape(n, T, 8, dk))
        return o.transpose(1, 2).reshape(n, D), s, torch.logsumexp(s, dim=-1)

    def sa_attn_fwd(self, qkv, kv_hist, idx, decay):
        with torch.no_grad():
            return self._attn(qkv.detach(), kv_hist.detach(), idx, decay.detach() if decay is not None else None)

    def sa_attn_bwd(self, qkv, kv_hist, idx, decay, out, score, lse, d_out, inverse=None):
        with torch.enable_grad():
            a = qkv.detach().clone().requires_grad_(True)
            b = kv_hist.detach().clone().requires_grad_(True)
            c = decay.detach().clone().requires_grad_(True) if decay is not None else None
            o, _, _ = self._attn(a, b, idx, c)
            ins = [a, b] + ([c] if c is not None else [])
            gs = torch.autograd.grad(o, ins, d_out.detach(), allow_unused=True)
        d_hist = gs[1] if gs[1] is not None else torch.zeros_like(b)
        return gs[0], d_hist, (gs[2] if c is not None else None)

    # ---- rows -----------------------------------------------------------------------------------
    def gather_rows(self, table, idx):
        i = idx.long()
        return table.detach()[i.clamp(min=0)] * (i >= 0).to(table.dtype).view(-1, 1)

    def segment_sum_rows(self, src, seg_ptr, order, n_seg, relu_of=None):
        out = torch.zeros(n_seg, src.shape[1], dtype=src.dtype)
        cnt = (seg_ptr[1:] - seg_ptr[:-1]).long()
        seg = torch.repeat_interleave(torch.arange(n_seg), cnt)
        out.index_add_(0, seg, src.detach()[order.long()])
        return out if relu_of is None else out * (relu_of.detach() > 0).to(out.dtype)

    def scatter_add_rows(self, src, idx, table):
        i = idx.long()
        keep = i >= 0
        table.index_add_(0, i[keep], src.detach()[keep])
        return table
