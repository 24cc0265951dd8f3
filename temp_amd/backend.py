"""Kernel backend: the only place that crosses the C ABI (include/temp_amd.h).

`HipBackend` hands raw device pointers of torch tensors (+ the current HIP stream) to
libtemp_amd.so.  PyTorch is used for device memory and streams only.  There is no CPU
implementation in the product: tensors that are not on a GPU, or a missing library, raise.
(tests/ install a test-only backend through `set_backend` to exercise host logic without a GPU.)
"""
import ctypes

import torch

from . import _lib

_backend = None


def set_backend(b):
    """Install a backend object (tests only).  Passing None restores the HIP backend."""
    global _backend
    _backend = b


def get_backend():
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _f32(t, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.TempAmdError("%s must live on the GPU (got %s): temp_amd has no CPU path" % (name, t.device))
    if t.dtype != torch.float32:
        raise _lib.TempAmdError("%s must be float32, got %s" % (name, t.dtype))
    return t.detach().contiguous()


def _i32(t, name):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.int32:
        raise _lib.TempAmdError("%s must be an int32 GPU tensor" % name)
    return t.contiguous()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)      # the handle without building a Stream object (13 us -> 0.3 us;
_cur_device = getattr(torch._C, "_cuda_getDevice", None)                 # a step makes ~15 launches through here, prepare as many)


def _stream():
    """The calling thread's current HIP stream (torch's per-thread current stream) as a C handle."""
    if _raw_stream is not None and _cur_device is not None:
        return ctypes.c_void_p(_raw_stream(_cur_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _drop(drop):
    """(p, seed) or None -> ctypes TempDropout by reference (NULL for no dropout)."""
    if drop is None or drop[0] <= 0.0:
        return None
    d = _lib.TempDropout()
    d.p, d.seed = float(drop[0]), int(drop[1]) & 0xFFFFFFFFFFFFFFFF
    return ctypes.byref(d)


class HipBackend:
    name = "hip"

    def __init__(self):
        self.lib = _lib.load()

    def _ws(self, nbytes, device):
        return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)

    # ---- RGCN layer ----------------------------------------------------------------------------
    def rgcn_fwd(self, dg, h, h_ids, weight, loop_w, bias, num_bases, act, drop=None, out=None):
        """out: optional contiguous (n_nodes, d_out) fp32 destination (e.g. a row range of a larger matrix)."""
        h, weight, loop_w, bias = _f32(h, "h"), _f32(weight, "weight"), _f32(loop_w, "loop_weight"), _f32(bias, "bias")
        h_ids = _i32(h_ids, "h_ids")
        d_in, d_out = loop_w.shape
        if out is None:
            out = torch.empty(dg.n_nodes, d_out, dtype=torch.float32, device=h.device)
        elif out.shape != (dg.n_nodes, d_out) or not out.is_contiguous() or out.dtype != torch.float32 or out.device != h.device:
            raise _lib.TempAmdError("rgcn_fwd: `out` must be a contiguous fp32 (n_nodes, d_out) matrix on the layer's device")
        nb = self.lib.temp_rgcn_fwd_workspace(dg.ref(), d_out)
        ws = self._ws(nb, h.device)
        rc = self.lib.temp_rgcn_fwd(dg.ref(), _ptr(h), _ptr(h_ids), d_in, d_out, num_bases, weight.shape[0], _ptr(weight),
                                    _ptr(loop_w), _ptr(bias), act, _ptr(out), _ptr(ws), ws.numel(), _drop(drop), _stream())
        _lib.check(rc, "temp_rgcn_fwd")
        return out

    def rgcn_bwd(self, dg, h, out, d_out_grad, weight, loop_w, has_bias, num_bases, act, drop=None):
        h, out, g = _f32(h, "h"), _f32(out, "out"), _f32(d_out_grad, "d_out")
        weight, loop_w = _f32(weight, "weight"), _f32(loop_w, "loop_weight")
        d_in, d_out = loop_w.shape
        dev = h.device
        d_h = torch.empty(dg.n_nodes, d_in, dtype=torch.float32, device=dev)
        d_w = torch.empty_like(weight)
        d_loop = torch.empty_like(loop_w)
        d_bias = torch.empty(d_out, dtype=torch.float32, device=dev) if has_bias else None
        nb = self.lib.temp_rgcn_bwd_workspace(dg.ref(), d_in, d_out, num_bases, weight.shape[0])
        ws = self._ws(nb, dev)
        rc = self.lib.temp_rgcn_bwd(dg.ref(), _ptr(h), _ptr(out), _ptr(g), d_in, d_out, num_bases, weight.shape[0], _ptr(weight),
                                    _ptr(loop_w), int(has_bias), act, _ptr(d_h), _ptr(d_w), _ptr(d_loop), _ptr(d_bias), _ptr(ws),
                                    ws.numel(), _drop(drop), _stream())
        _lib.check(rc, "temp_rgcn_bwd")
        return d_h, d_w, d_loop, d_bias

    def rgcn_bwd_dh(self, dg, out, d_out_grad, weight, loop_w, num_bases, act, drop=None, dz_out=None, dzm_out=None):
        """d_h of a layer only (include/temp_amd.h: temp_rgcn_bwd_dh); dz_out / dzm_out receive the masked gradients the weight
        pass (rgcn_bwd_weights) reads, where the activation / the dropout make them differ from d_out_grad."""
        g = _f32(d_out_grad, "d_out")
        weight, loop_w = _f32(weight, "weight"), _f32(loop_w, "loop_weight")
        d_in, d_out = loop_w.shape
        dev = g.device
        d_h = torch.empty(dg.n_nodes, d_in, dtype=torch.float32, device=dev)
        ws = self._ws(self.lib.temp_rgcn_bwd_workspace(dg.ref(), d_in, d_out, num_bases, weight.shape[0]), dev)
        rc = self.lib.temp_rgcn_bwd_dh(dg.ref(), _ptr(out), _ptr(g), d_in, d_out, num_bases, weight.shape[0], _ptr(weight), _ptr(loop_w), act,
                                       _ptr(d_h), _ptr(dz_out), _ptr(dzm_out), _ptr(ws), ws.numel(), _drop(drop), _stream())
        _lib.check(rc, "temp_rgcn_bwd_dh")
        return d_h

    def rgcn_bwd_weights(self, dg, h, dz, dzm, weight_like, loop_like, has_bias, num_bases):
        """(d_weight, d_loop_w, d_bias) of a layer from its input rows and the masked output gradients over ANY graph the rows
        belong to -- the union of all positions of a recurrence (include/temp_amd.h: temp_rgcn_bwd_weights)."""
        h, dz = _f32(h, "h"), _f32(dz, "dz")
        d_in, d_out = loop_like.shape
        dev = h.device
        d_w, d_loop = torch.empty_like(weight_like), torch.empty_like(loop_like)
        d_bias = torch.empty(d_out, dtype=torch.float32, device=dev) if has_bias else None
        ws = self._ws(self.lib.temp_rgcn_bwd_workspace(dg.ref(), d_in, d_out, num_bases, weight_like.shape[0]), dev)
        rc = self.lib.temp_rgcn_bwd_weights(dg.ref(), _ptr(h), _ptr(dz), _ptr(dzm), d_in, d_out, num_bases, weight_like.shape[0], int(has_bias),
                                            _ptr(d_w), _ptr(d_loop), _ptr(d_bias), _ptr(ws), ws.numel(), _stream())
        _lib.check(rc, "temp_rgcn_bwd_weights")
        return d_w, d_loop, d_bias

    def rgcn_table_fwd(self, dg, table, ids, weight, loop_w, bias, num_bases, act, drop=None):
        """Layer on h = table[ids] without materialising h (include/temp_amd.h: temp_rgcn_table_fwd)."""
        table, weight, loop_w, bias = _f32(table, "table"), _f32(weight, "weight"), _f32(loop_w, "loop_weight"), _f32(bias, "bias")
        ids = _i32(ids, "ids")
        d_in, d_out = loop_w.shape
        out = torch.empty(dg.n_nodes, d_out, dtype=torch.float32, device=table.device)
        ws = self._ws(self.lib.temp_rgcn_table_fwd_workspace(dg.ref(), table.shape[0], d_out), table.device)
        rc = self.lib.temp_rgcn_table_fwd(dg.ref(), _ptr(table), _ptr(ids), table.shape[0], d_in, d_out, num_bases, weight.shape[0],
                                          _ptr(weight), _ptr(loop_w), _ptr(bias), act, _ptr(out), _ptr(ws), ws.numel(), _drop(drop), _stream())
        _lib.check(rc, "temp_rgcn_table_fwd")
        return out

    def rgcn_table_bwd(self, dg, table, ids, inverse, out, d_out_grad, weight, loop_w, has_bias, num_bases, act, drop=None):
        """-> (d_table [n_table, d_in] fully written, d_weight, d_loop_w, d_bias)."""
        table, out, g = _f32(table, "table"), _f32(out, "out"), _f32(d_out_grad, "d_out")
        weight, loop_w = _f32(weight, "weight"), _f32(loop_w, "loop_weight")
        ids, inv_ptr, inv_order = _i32(ids, "ids"), _i32(inverse[0], "inv_ptr"), _i32(inverse[1], "inv_order")
        d_in, d_out = loop_w.shape
        dev = table.device
        d_table = torch.empty_like(table)
        d_w = torch.empty_like(weight)
        d_loop = torch.empty_like(loop_w)
        d_bias = torch.empty(d_out, dtype=torch.float32, device=dev) if has_bias else None
        ws = self._ws(self.lib.temp_rgcn_table_bwd_workspace(dg.ref(), table.shape[0], d_in, d_out, num_bases), dev)
        rc = self.lib.temp_rgcn_table_bwd(dg.ref(), _ptr(table), _ptr(ids), _ptr(inv_ptr), _ptr(inv_order), table.shape[0], _ptr(out), _ptr(g),
                                          d_in, d_out, num_bases, weight.shape[0], _ptr(weight), _ptr(loop_w), int(has_bias), act,
                                          _ptr(d_table), _ptr(d_w), _ptr(d_loop), _ptr(d_bias), _ptr(ws), ws.numel(), _drop(drop), _stream())
        _lib.check(rc, "temp_rgcn_table_bwd")
        return d_table, d_w, d_loop, d_bias

    def rgcn_isolated_fwd(self, e, loop_w, bias, act, drop=None):
        e, loop_w, bias = _f32(e, "e"), _f32(loop_w, "loop_weight"), _f32(bias, "bias")
        out = torch.empty_like(e)
        rc = self.lib.temp_rgcn_isolated_fwd(e.shape[0], e.shape[1], _ptr(e), _ptr(loop_w), _ptr(bias), act, _ptr(out), _drop(drop), _stream())
        _lib.check(rc, "temp_rgcn_isolated_fwd")
        return out

    def rgcn_isolated_bwd(self, e, out, d_out_grad, loop_w, has_bias, act, drop=None):
        e, out, g, loop_w = _f32(e, "e"), _f32(out, "out"), _f32(d_out_grad, "d_out"), _f32(loop_w, "loop_weight")
        n, d = e.shape
        d_e = torch.empty_like(e)
        d_loop = torch.empty_like(loop_w)
        d_bias = torch.empty(d, dtype=torch.float32, device=e.device) if has_bias else None
        ws = self._ws(self.lib.temp_rgcn_isolated_bwd_workspace(n, d), e.device)
        rc = self.lib.temp_rgcn_isolated_bwd(n, d, _ptr(e), _ptr(out), _ptr(g), _ptr(loop_w), int(has_bias), act, _ptr(d_e),
                                             _ptr(d_loop), _ptr(d_bias), _ptr(ws), ws.numel(), _drop(drop), _stream())
        _lib.check(rc, "temp_rgcn_isolated_bwd")
        return d_e, d_loop, d_bias

    # ---- decay + GRU step ------------------------------------------------------------------------
    def gru_fwd(self, x, prev, prev_idx, dt, lam, decay_wb, w_ih, w_hh, b_ih, b_hh, variant):
        x, prev, dt = _f32(x, "x"), _f32(prev, "prev"), _f32(dt, "dt")
        w_ih, w_hh, b_ih, b_hh = _f32(w_ih, "w_ih"), _f32(w_hh, "w_hh"), _f32(b_ih, "b_ih"), _f32(b_hh, "b_hh")
        decay_wb, prev_idx = _f32(decay_wb, "decay_wb"), _i32(prev_idx, "prev_idx")
        n, d = x.shape
        h_out = torch.empty_like(x)
        saved = torch.empty(5, n, d, dtype=torch.float32, device=x.device)
        rc = self.lib.temp_gru_fwd(n, d, variant, _ptr(x), _ptr(prev), _ptr(prev_idx), _ptr(dt), float(lam), _ptr(decay_wb),
                                   _ptr(w_ih), _ptr(w_hh), _ptr(b_ih), _ptr(b_hh), _ptr(h_out), _ptr(saved), _stream())
        _lib.check(rc, "temp_gru_fwd")
        return h_out, saved

    def gru_bwd(self, x, prev, prev_idx, dt, lam, decay_wb, w_ih, w_hh, saved, d_h, variant):
        x, prev, dt, d_h = _f32(x, "x"), _f32(prev, "prev"), _f32(dt, "dt"), _f32(d_h, "d_h")
        w_ih, w_hh, saved = _f32(w_ih, "w_ih"), _f32(w_hh, "w_hh"), _f32(saved, "saved")
        decay_wb, prev_idx = _f32(decay_wb, "decay_wb"), _i32(prev_idx, "prev_idx")
        n, d = x.shape
        dev = x.device
        d_x = torch.empty_like(x)
        d_prev = torch.empty_like(x)
        d_w_ih, d_w_hh = torch.empty_like(w_ih), torch.empty_like(w_hh)
        d_b_ih = torch.empty(w_ih.shape[0], dtype=torch.float32, device=dev)
        d_b_hh = torch.empty(w_hh.shape[0], dtype=torch.float32, device=dev)
        d_wb = torch.empty(2, dtype=torch.float32, device=dev) if decay_wb is not None else None
        ws = self._ws(self.lib.temp_gru_bwd_workspace(n, d, variant), dev)
        rc = self.lib.temp_gru_bwd(n, d, variant, _ptr(x), _ptr(prev), _ptr(prev_idx), _ptr(dt), float(lam), _ptr(decay_wb),
                                   _ptr(w_ih), _ptr(w_hh), _ptr(saved), _ptr(d_h), _ptr(d_x), _ptr(d_prev), _ptr(d_w_ih),
                                   _ptr(d_w_hh), _ptr(d_b_ih), _ptr(d_b_hh), _ptr(d_wb), _ptr(ws), ws.numel(), _stream())
        _lib.check(rc, "temp_gru_bwd")
        return d_x, d_prev, d_w_ih, d_w_hh, d_b_ih, d_b_hh, d_wb

    # ---- window-batched recurrence (hoisted input gates, per-position cell, batched weight grads) ---
    def gru_input_gates(self, x, w_ih, b_ih, variant, out):
        x, w_ih, b_ih = _f32(x, "x"), _f32(w_ih, "w_ih"), _f32(b_ih, "b_ih")
        assert out.is_contiguous() and out.shape == (x.shape[0], w_ih.shape[0])
        rc = self.lib.temp_gru_input_gates(x.shape[0], x.shape[1], variant, _ptr(x), _ptr(w_ih), _ptr(b_ih), _ptr(out), _stream())
        _lib.check(rc, "temp_gru_input_gates")

    def gru_input_gates_multi(self, xs, w_ihs, b_ihs, variant, outs, x_idx=None, x_keys=None):
        """gi_i = x_i . W_ih_i^T + b_ih_i for every (row block, weight set) pair in one launch per four problems
        (include/temp_amd.h: temp_gru_input_gates_multi); x_idx: per problem an int32 table of the x rows to take (the out
        block has one row per entry; temp_gru_input_gates_gather_multi)."""
        k = len(xs)
        xs = [_f32(x, "x") for x in xs]
        w_ihs, b_ihs = [_f32(w, "w_ih") for w in w_ihs], [_f32(b, "b_ih") for b in b_ihs]
        d = xs[0].shape[1]
        idx = [None] * k if x_idx is None else [None if t is None else _i32(t, "x_idx") for t in x_idx]
        rows = [x.shape[0] if t is None else t.shape[0] for x, t in zip(xs, idx)]
        for x, w, o, n in zip(xs, w_ihs, outs, rows):
            assert x.shape[1] == d and o.is_contiguous() and o.shape == (n, w.shape[0]) and w.shape == w_ihs[0].shape
        arr = lambda ts: (ctypes.c_void_p * k)(*[t.data_ptr() for t in ts])
        ns = (ctypes.c_int * k)(*rows)
        if x_idx is None and x_keys is None:
            rc = self.lib.temp_gru_input_gates_multi(k, ns, d, variant, arr(xs), arr(w_ihs), arr(b_ihs), arr(outs), _stream())
            _lib.check(rc, "temp_gru_input_gates_multi")
            return
        ia = (ctypes.c_void_p * k)(*[None if t is None else t.data_ptr() for t in idx])
        if x_keys is not None:                           # row keys of every xs[i] by SOURCE row (int32 [x_i rows]): f16 arithmetic
            for x, t in zip(xs, x_keys):
                assert t.dtype == torch.int32 and t.is_contiguous() and t.numel() == x.shape[0]
            ka = (ctypes.c_void_p * k)(*[t.data_ptr() for t in x_keys])
            rc = self.lib.temp_gru_input_gates_gather_multi_keys(k, ns, d, variant, arr(xs), ia if x_idx is not None else None, ka, arr(w_ihs),
                                                                 arr(b_ihs), arr(outs), _stream())
            _lib.check(rc, "temp_gru_input_gates_gather_multi_keys")
            return
        rc = self.lib.temp_gru_input_gates_gather_multi(k, ns, d, variant, arr(xs), ia, arr(w_ihs), arr(b_ihs), arr(outs), _stream())
        _lib.check(rc, "temp_gru_input_gates_gather_multi")

    def gru_cell_fwd(self, gi, prev, prev_idx, dt, lam, w_hh, b_hh, variant, h_out, saved_all, row0):
        """h_out (n,d) and rows [row0, row0+n) of every plane of saved_all (5, N, d) are written."""
        n, d = h_out.shape
        plane = saved_all.shape[1] * d
        sp = ctypes.c_void_p(saved_all.data_ptr() + 4 * row0 * d)
        rc = self.lib.temp_gru_cell_fwd(n, d, variant, _ptr(gi), _ptr(_f32(prev, "prev")), _ptr(_i32(prev_idx, "prev_idx")),
                                        _ptr(_f32(dt, "dt")), float(lam), _ptr(_f32(w_hh, "w_hh")), _ptr(_f32(b_hh, "b_hh")),
                                        _ptr(h_out), sp, plane, _stream())
        _lib.check(rc, "temp_gru_cell_fwd")

    def gru_cell_bwd(self, saved_all, row0, n, dh_up, d_prev_next, next_idx, dt, lam, w_hh, variant, dgi, dgh, decv, d_prev):
        d = saved_all.shape[2]
        plane = saved_all.shape[1] * d
        sp = ctypes.c_void_p(saved_all.data_ptr() + 4 * row0 * d)
        rc = self.lib.temp_gru_cell_bwd(n, d, variant, sp, plane, _ptr(dh_up), _ptr(d_prev_next), _ptr(_i32(next_idx, "next_idx")),
                                        _ptr(_f32(dt, "dt")), float(lam), _ptr(_f32(w_hh, "w_hh")), _ptr(dgi), _ptr(dgh), _ptr(decv),
                                        _ptr(d_prev), _stream())
        _lib.check(rc, "temp_gru_cell_bwd")

    def gru_cell_fwd_multi(self, cells, lam, variant, saved_all):
        """cells: list (<= 4) of dicts(gi, prev, prev_idx, dt, w_hh, b_hh, h_out, row0) -- one launch for all."""
        d = saved_all.shape[2]
        arr = (_lib.TempGruCellFwd * len(cells))()
        keep = []
        for a, c in zip(arr, cells):
            prev, pidx, dt = _f32(c["prev"], "prev"), _i32(c["prev_idx"], "prev_idx"), _f32(c["dt"], "dt")
            w_hh, b_hh = _f32(c["w_hh"], "w_hh"), _f32(c["b_hh"], "b_hh")
            keep += [prev, pidx, dt, w_hh, b_hh]
            a.n = c["h_out"].shape[0]
            a.gi, a.prev = c["gi"].data_ptr(), (prev.data_ptr() if prev is not None else None)      # prev None: zero-state cell
            a.prev_idx, a.dt = (pidx.data_ptr() if pidx is not None else None), dt.data_ptr()
            a.w_hh, a.b_hh, a.h_out = w_hh.data_ptr(), b_hh.data_ptr(), c["h_out"].data_ptr()
            a.saved = saved_all.data_ptr() + 4 * c["row0"] * d
        rc = self.lib.temp_gru_cell_fwd_multi(len(cells), arr, d, variant, float(lam), saved_all.shape[1] * d, _stream())
        _lib.check(rc, "temp_gru_cell_fwd_multi")

    def gru_cell_bwd_multi(self, cells, lam, variant, saved_all):
        """cells: list (<= 4) of dicts(row0, n, dh_up, d_prev_next, next_idx, dt, w_hh, dgi, dgh, decv, d_prev)."""
        d = saved_all.shape[2]
        arr = (_lib.TempGruCellBwd * len(cells))()
        keep = []
        opt = lambda t: t.data_ptr() if t is not None else None
        for a, c in zip(arr, cells):
            dt, w_hh, nidx = _f32(c["dt"], "dt"), _f32(c["w_hh"], "w_hh"), _i32(c["next_idx"], "next_idx")
            keep += [dt, w_hh, nidx]
            a.n = c["n"]
            a.saved = saved_all.data_ptr() + 4 * c["row0"] * d
            a.dh_up, a.d_prev_next, a.next_idx = opt(c["dh_up"]), opt(c["d_prev_next"]), opt(nidx)
            a.dt, a.w_hh = dt.data_ptr(), w_hh.data_ptr()
            a.dgi, a.dgh, a.decv, a.d_prev = c["dgi"].data_ptr(), c["dgh"].data_ptr(), c["decv"].data_ptr(), c["d_prev"].data_ptr()
            a.no_prev = 1 if c.get("no_prev") else 0
        rc = self.lib.temp_gru_cell_bwd_multi(len(cells), arr, d, variant, float(lam), saved_all.shape[1] * d, _stream())
        _lib.check(rc, "temp_gru_cell_bwd_multi")

    def subsample_views(self, jobs):
        """jobs: list of dicts(n_nodes, n_edges, keep, seed, parent, child, eid, offs {name: [3] | int}, n_chunks [3], keep_mask,
        scratch) -- see include/temp_amd.h: temp_subsample_views."""
        arr = (_lib.TempSubsampleJob * len(jobs))()
        for a, j in zip(arr, jobs):
            a.n_nodes, a.n_edges, a.keep, a.seed = j["n_nodes"], j["n_edges"], j["keep"], int(j["seed"]) & 0xFFFFFFFFFFFFFFFF
            a.parent, a.child = _i32(j["parent"], "parent").data_ptr(), _i32(j["child"], "child").data_ptr()
            a.eid = _i32(j["eid"], "eid").data_ptr() if j["n_edges"] else None
            for fld in ("off_a", "off_b", "off_chunk_beg", "off_chunk_end", "off_chunk_seg", "n_chunks"):
                for v in range(3):
                    getattr(a, fld)[v] = int(j[fld][v])
            a.off_in_deg, a.off_out_deg, a.off_nnorm = int(j["off_in_deg"]), int(j["off_out_deg"]), int(j["off_nnorm"])
            a.keep_mask = j["keep_mask"].data_ptr() if j.get("keep_mask") is not None else None
            a.scratch = j["scratch"].data_ptr()
        _lib.check(self.lib.temp_subsample_views(len(jobs), arr, _stream()), "temp_subsample_views")

    def decay_rows(self, x, dt, lam):
        x, dt = _f32(x, "x"), _f32(dt, "dt")
        out = torch.empty_like(x)
        _lib.check(self.lib.temp_decay_rows(x.shape[0], x.shape[1], _ptr(x), _ptr(dt), float(lam), _ptr(out), _stream()), "temp_decay_rows")
        return out

    # ---- persistent window chain (include/temp_amd.h: TempGruChain) -------------------------------
    def gru_chain_supported(self, d):
        return bool(self.lib.temp_gru_chain_supported(int(d)))

    def gru_chain_pack(self, w_hh):
        """W_hh [3d, d] -> the chain kernels' fragment order (one small launch; the weights change every optimiser step)."""
        w_hh = _f32(w_hh, "w_hh")
        d = w_hh.shape[1]
        out = torch.empty(self.lib.temp_gru_chain_pack_floats(d), dtype=torch.float32, device=w_hh.device)
        _lib.check(self.lib.temp_gru_chain_pack(d, _ptr(w_hh), _ptr(out), _stream()), "temp_gru_chain_pack")
        return out

    def gru_chain_pack_multi(self, w_hhs):
        """gru_chain_pack of several GRUs' W_hh (same width) in ONE launch -> list of packed tensors (views of one buffer)."""
        w_hhs = [_f32(w, "w_hh") for w in w_hhs]
        d = w_hhs[0].shape[1]
        n = self.lib.temp_gru_chain_pack_floats(d)
        buf = torch.empty(len(w_hhs), n, dtype=torch.float32, device=w_hhs[0].device)
        src = (ctypes.c_void_p * len(w_hhs))(*[w.data_ptr() for w in w_hhs])
        dst = (ctypes.c_void_p * len(w_hhs))(*[buf[i].data_ptr() for i in range(len(w_hhs))])
        _lib.check(self.lib.temp_gru_chain_pack_multi(len(w_hhs), d, src, dst, _stream()), "temp_gru_chain_pack_multi")
        return [buf[i] for i in range(len(w_hhs))]

    def _chain_desc(self, tabs, d, variant, lam, plane, packs, b_hhs):
        c = _lib.TempGruChain()
        c.d, c.variant, c.n_panels, c.n_steps, c.max_steps = d, variant, tabs["n_panels"], tabs["n_steps"], tabs["max_steps"]
        c.panel, c.rows, c.sinfo, c.dt = (_i32(tabs[k], k).data_ptr() for k in ("panel", "rows", "sinfo", "dt_bits"))
        c.lambda_, c.saved_plane, c.n_rnn = float(lam), plane, len(packs)
        keep = []
        for i, (pk, b) in enumerate(zip(packs, b_hhs)):
            pk, b = _f32(pk, "packed"), _f32(b, "b_hh")
            keep += [pk, b]
            c.packed[i], c.b_hh[i] = pk.data_ptr(), b.data_ptr()
        return c, keep

    def gru_chain_fwd(self, tabs, gi, lam, variant, packs, b_hhs, h_out, saved_all, gi_index=None):
        d = saved_all.shape[2]
        c, keep = self._chain_desc(tabs, d, variant, lam, saved_all.shape[1] * d, packs, b_hhs)
        if gi_index is not None:
            assert gi_index.shape[0] == saved_all.shape[1]
            c.gi_index = _i32(gi_index, "gi_index").data_ptr()
        rc = self.lib.temp_gru_chain_fwd(ctypes.byref(c), _ptr(_f32(gi, "gi")), _ptr(h_out), _ptr(saved_all), _stream())
        _lib.check(rc, "temp_gru_chain_fwd")

    def gru_chain_bwd(self, tabs, saved_all, ups, lam, variant, packs, b_hhs, dgi, dgh):
        d = saved_all.shape[2]
        c, keep = self._chain_desc(tabs, d, variant, lam, saved_all.shape[1] * d, packs, b_hhs)
        ups = [_f32(u, "upstream") if u is not None else None for u in ups]      # (None: that block of rows has no upstream gradient)
        arr = (ctypes.c_void_p * max(len(ups), 1))(*[u.data_ptr() if u is not None else None for u in ups])
        rc = self.lib.temp_gru_chain_bwd(ctypes.byref(c), _ptr(saved_all), len(ups), arr, _ptr(dgi), _ptr(dgh), _stream())
        _lib.check(rc, "temp_gru_chain_bwd")

    def gru_chain_keys_supported(self, d):
        """True when the chain backward of this width hands out the row / column keys of g4 (f16 two-way split selected)."""
        return bool(self.lib.temp_gru_chain_keys_supported(int(d)))

    def gru_chain_bwd_g4(self, tabs, saved_all, ups, lam, variant, packs, b_hhs, g4, keys=None):
        """The chain backward with the gate gradients written once: g4 [N, 4d] = [dr | dz | dn_i | dn_h] (include/temp_amd.h:
        temp_gru_chain_bwd_g4; nn.GRU gate layout).  keys = (row_keys int32 [N], col_keys int32 [n_rnn + n_panels, 4d]; rows 0 .. n_rnn - 1 are the result): also the
        magnitude keys the consumers of g4 split it with (temp_gru_chain_bwd_g4_keys)."""
        d = saved_all.shape[2]
        c, keep = self._chain_desc(tabs, d, variant, lam, saved_all.shape[1] * d, packs, b_hhs)
        ups = [_f32(u, "upstream") if u is not None else None for u in ups]
        arr = (ctypes.c_void_p * max(len(ups), 1))(*[u.data_ptr() if u is not None else None for u in ups])
        if keys is not None:
            row_keys, col_keys = keys
            assert row_keys.dtype == torch.int32 and col_keys.dtype == torch.int32 and col_keys.is_contiguous()
            assert row_keys.numel() == saved_all.shape[1] and col_keys.numel() == (len(packs) + tabs["n_panels"]) * 4 * d
            rc = self.lib.temp_gru_chain_bwd_g4_keys(ctypes.byref(c), _ptr(saved_all), len(ups), arr, _ptr(g4), _ptr(row_keys), _ptr(col_keys),
                                                     _stream())
            _lib.check(rc, "temp_gru_chain_bwd_g4_keys")
            return
        rc = self.lib.temp_gru_chain_bwd_g4(ctypes.byref(c), _ptr(saved_all), len(ups), arr, _ptr(g4), _stream())
        _lib.check(rc, "temp_gru_chain_bwd_g4")

    def gru_grads_g4_supported(self, ns, d, variant):
        """True when gru_grads_g4 takes GRUs of these row counts and this width (then the chain backward should write g4)."""
        k = len(ns)
        if k < 1 or k > 4 or variant != _lib.GRU_TORCH:
            return False
        return self.lib.temp_gru_grads_g4_workspace(k, (ctypes.c_int * k)(*[int(n) for n in ns]), int(d)) > 0

    def gru_grads_g4(self, xs, hdecs, g4s, w_ihs, d_xs, row_keys=None, col_keys=None, x_col_keys=None):
        """Weight / bias gradients and d_x of several GRUs of one width from their gate-gradient matrices (include/temp_amd.h:
        temp_gru_grads_g4) -> [(d_w_ih, d_w_hh, d_b_ih, d_b_hh)] per GRU.  row_keys / col_keys: per GRU the int32 key tensors of
        gru_chain_bwd_g4(keys=...) ([n_i] and [4d]); with them the products run on the f16 pipe (temp_gru_grads_g4_keys)."""
        k = len(xs)
        d = xs[0].shape[1]
        dev = xs[0].device
        ns = (ctypes.c_int * k)(*[x.shape[0] for x in xs])
        nb = self.lib.temp_gru_grads_g4_workspace(k, ns, d)
        if nb == 0:
            raise ValueError("temp_gru_grads_g4: unsupported shape (check gru_grads_g4_supported first)")
        keep = [[_f32(t, "operand") for t in ts] for ts in (xs, hdecs, w_ihs)]
        for g in g4s:
            assert g.dtype == torch.float32 and g.is_contiguous() and g.shape[1] == 4 * d
        arr = lambda ts: (ctypes.c_void_p * k)(*[t.data_ptr() for t in ts])
        dx = (ctypes.c_void_p * k)(*[None if t is None else t.data_ptr() for t in d_xs])
        d_w = torch.empty(2 * k, 3 * d, d, dtype=torch.float32, device=dev)
        d_b = torch.empty(2 * k, 3 * d, dtype=torch.float32, device=dev)
        ws = self._ws(nb, dev)
        if row_keys is not None and col_keys is not None:
            for rk, ck, x in zip(row_keys, col_keys, xs):
                assert rk.dtype == torch.int32 and ck.dtype == torch.int32 and rk.is_contiguous() and ck.is_contiguous()
                assert rk.numel() == x.shape[0] and ck.numel() == 4 * d
            xk = None
            if x_col_keys is not None:
                for t in x_col_keys:
                    assert t.dtype == torch.int32 and t.is_contiguous() and t.numel() == d
                xk = arr(x_col_keys)
            rc = self.lib.temp_gru_grads_g4_keys(k, ns, d, arr(keep[0]), arr(keep[1]), arr(g4s), arr(keep[2]), dx, _ptr(d_w), _ptr(d_b),
                                                 arr(row_keys), arr(col_keys), xk, _ptr(ws), ws.numel(), _stream())
            _lib.check(rc, "temp_gru_grads_g4_keys")
            return [(d_w[2 * i], d_w[2 * i + 1], d_b[2 * i], d_b[2 * i + 1]) for i in range(k)]
        rc = self.lib.temp_gru_grads_g4(k, ns, d, arr(keep[0]), arr(keep[1]), arr(g4s), arr(keep[2]), dx, _ptr(d_w), _ptr(d_b), _ptr(ws),
                                        ws.numel(), _stream())
        _lib.check(rc, "temp_gru_grads_g4")
        return [(d_w[2 * i], d_w[2 * i + 1], d_b[2 * i], d_b[2 * i + 1]) for i in range(k)]

    def gru_weight_grads(self, x, hdec, dgi, dgh, w_ih, variant, d_x):
        n, d = x.shape
        w_ih = _f32(w_ih, "w_ih")
        dev = x.device
        d_w_ih = torch.empty_like(w_ih)
        d_w_hh = torch.empty(3 * d, d, dtype=torch.float32, device=dev)
        d_b_ih = torch.empty(w_ih.shape[0], dtype=torch.float32, device=dev)
        d_b_hh = torch.empty(3 * d, dtype=torch.float32, device=dev)
        ws = self._ws(self.lib.temp_gru_weight_grads_workspace(n, d, variant), dev)
        rc = self.lib.temp_gru_weight_grads(n, d, variant, _ptr(x), _ptr(hdec), _ptr(dgi), _ptr(dgh), _ptr(w_ih), _ptr(d_x),
                                            _ptr(d_w_ih), _ptr(d_w_hh), _ptr(d_b_ih), _ptr(d_b_hh), _ptr(ws), ws.numel(), _stream())
        _lib.check(rc, "temp_gru_weight_grads")
        return d_w_ih, d_w_hh, d_b_ih, d_b_hh

    def gru_weight_grads_multi(self, xs, hdecs, dgis, dghs, w_ihs, variant, d_xs):
        """Weight / bias gradients and d_x of SEVERAL GRUs of one width in one weight-gradient launch (include/temp_amd.h:
        temp_gru_weight_grads_multi) -> [(d_w_ih, d_w_hh, d_b_ih, d_b_hh)] per GRU, or None when the library takes this shape
        through the per-GRU call (nothing launched)."""
        k = len(xs)
        if k < 1 or k > 4 or variant != _lib.GRU_TORCH or any(h is None for h in hdecs):
            return None
        d = xs[0].shape[1]
        dev = xs[0].device
        ns = (ctypes.c_int * k)(*[x.shape[0] for x in xs])
        nb = self.lib.temp_gru_weight_grads_multi_workspace(k, ns, d, variant)
        if nb == 0:
            return None
        keep = [[_f32(t, "operand") for t in ts] for ts in (xs, hdecs, dgis, dghs, w_ihs)]
        arr = lambda ts: (ctypes.c_void_p * k)(*[t.data_ptr() for t in ts])
        dx = (ctypes.c_void_p * k)(*[None if t is None else t.data_ptr() for t in d_xs])
        d_w = torch.empty(2 * k, 3 * d, d, dtype=torch.float32, device=dev)
        d_b = torch.empty(2 * k, 3 * d, dtype=torch.float32, device=dev)
        ws = self._ws(nb, dev)
        rc = self.lib.temp_gru_weight_grads_multi(k, ns, d, variant, arr(keep[0]), arr(keep[1]), arr(keep[2]), arr(keep[3]), arr(keep[4]), dx,
                                                  _ptr(d_w), _ptr(d_b), _ptr(ws), ws.numel(), _stream())
        if rc == 2:                                   # TEMP_E_UNSUPPORTED: nothing was launched
            return None
        _lib.check(rc, "temp_gru_weight_grads_multi")
        return [(d_w[2 * i], d_w[2 * i + 1], d_b[2 * i], d_b[2 * i + 1]) for i in range(k)]

    # ---- plain GEMMs + candidate cross-entropy (link-prediction loss) ---------------------------------
    def linear(self, a, b, trans_b, out=None, a_keys=None):
        """a[M,K] . b  (b is [K,N], or [N,K] when trans_b); out: an (M, N) contiguous tensor to write into.  a_keys: int32 [M], the
        magnitude keys of a's rows (absmax_keys): the product then runs on the f16 pipe without a pass of its own (temp_linear_keys)."""
        a, b = _f32(a, "a"), _f32(b, "b")
        M, K = a.shape
        N = b.shape[0] if trans_b else b.shape[1]
        c = out if out is not None else torch.empty(M, N, dtype=torch.float32, device=a.device)
        assert c.shape == (M, N) and c.is_contiguous() and c.dtype == torch.float32
        if a_keys is not None:
            assert a_keys.dtype == torch.int32 and a_keys.is_contiguous() and a_keys.numel() == M
            rc = self.lib.temp_linear_keys(M, N, K, _ptr(a), K, _ptr(a_keys), _ptr(b), b.shape[1], int(trans_b), _ptr(c), N, _stream())
            _lib.check(rc, "temp_linear_keys")
            return c
        rc = self.lib.temp_linear(M, N, K, _ptr(a), K, _ptr(b), b.shape[1], int(trans_b), _ptr(c), N, _stream())
        _lib.check(rc, "temp_linear")
        return c

    def absmax_keys(self, x, rows=True, cols=True):
        """Magnitude keys of a matrix (include/temp_amd.h: temp_absmax_keys) -> (row_keys int32 [n] | None, col_keys int32 [d] | None)."""
        x = _f32(x, "x")
        n, d = x.shape
        rk = torch.empty(n, dtype=torch.int32, device=x.device) if rows else None
        ck = torch.empty(self.lib.temp_keys_cols_size(d), dtype=torch.int32, device=x.device) if cols else None
        _lib.check(self.lib.temp_absmax_keys(n, d, _ptr(x), d, _ptr(rk), _ptr(ck), _stream()), "temp_absmax_keys")
        return rk, (ck[:d] if cols else None)

    def linear_t(self, a, b, trans_b, out_t):
        """out_t[N, M] = (a[M,K] . b)^T written into the contiguous matrix `out_t` (b is [K,N], or [N,K] when trans_b)."""
        a, b = _f32(a, "a"), _f32(b, "b")
        M, K = a.shape
        N = b.shape[0] if trans_b else b.shape[1]
        if out_t.shape != (N, M) or not out_t.is_contiguous() or out_t.dtype != torch.float32:
            raise ValueError("linear_t: out_t must be a contiguous float32 (N, M) matrix")
        rc = self.lib.temp_linear_t(M, N, K, _ptr(a), K, _ptr(b), b.shape[1], int(trans_b), _ptr(out_t), M, _stream())
        _lib.check(rc, "temp_linear_t")
        return out_t

    def linear_multi(self, a_list, b_list, trans_b, out):
        """out rows [r_i, r_i + M_i) = a_i[M_i,K] . b_i  for every problem i, r_i = running row offset: the per-window
        products of the loss in ceil(count / 4) launches.  `out` is a preallocated (sum M_i, N) matrix."""
        K = a_list[0].shape[1]
        N = out.shape[1]
        arr = (_lib.TempLinearProblem * len(a_list))()
        row, keep = 0, []
        for i, (a, b) in enumerate(zip(a_list, b_list)):
            a, b = _f32(a, "a"), _f32(b, "b")
            keep.append((a, b))
            if a.shape[1] != K or b.shape != ((N, K) if trans_b else (K, N)):
                raise ValueError("linear_multi: problems must share N, K and the layout of b")
            arr[i].M, arr[i].A, arr[i].B = a.shape[0], a.data_ptr(), b.data_ptr()
            arr[i].C = out.data_ptr() + row * N * 4
            row += a.shape[0]
        if row != out.shape[0] or not out.is_contiguous():
            raise ValueError("linear_multi: out must be a contiguous (sum M_i, N) matrix")
        rc = self.lib.temp_linear_multi(len(a_list), arr, N, K, K, K if trans_b else N, int(trans_b), N, _stream())
        _lib.check(rc, "temp_linear_multi")
        return out

    def bilinear_query_fwd(self, kind, ent_rows, known_idx, rel, rel_idx, is_tail):
        ent_rows, rel = _f32(ent_rows, "ent_rows"), _f32(rel, "rel")
        known_idx, rel_idx, is_tail = _i32(known_idx, "known_idx"), _i32(rel_idx, "rel_idx"), _i32(is_tail, "is_tail")
        P, d = known_idx.shape[0], ent_rows.shape[1]
        q = torch.empty(P, d, dtype=torch.float32, device=ent_rows.device)
        rc = self.lib.temp_bilinear_query_fwd(P, d, _lib.SCORE_KINDS[kind], _ptr(ent_rows), _ptr(known_idx), _ptr(rel), _ptr(rel_idx), _ptr(is_tail),
                                              _ptr(q), _stream())
        _lib.check(rc, "temp_bilinear_query_fwd")
        return q

    def bilinear_query_bwd(self, kind, ent_rows, known_idx, rel, rel_idx, is_tail, d_q):
        ent_rows, rel, d_q = _f32(ent_rows, "ent_rows"), _f32(rel, "rel"), _f32(d_q, "d_q")
        known_idx, rel_idx, is_tail = _i32(known_idx, "known_idx"), _i32(rel_idx, "rel_idx"), _i32(is_tail, "is_tail")
        P, d = known_idx.shape[0], ent_rows.shape[1]
        dk = torch.empty(P, d, dtype=torch.float32, device=ent_rows.device)
        dr = torch.empty(P, d, dtype=torch.float32, device=ent_rows.device)
        rc = self.lib.temp_bilinear_query_bwd(P, d, _lib.SCORE_KINDS[kind], _ptr(ent_rows), _ptr(known_idx), _ptr(rel), _ptr(rel_idx), _ptr(is_tail),
                                              _ptr(d_q), _ptr(dk), _ptr(dr), _stream())
        _lib.check(rc, "temp_bilinear_query_bwd")
        return dk, dr

    def linear_tn(self, a, b, out=None):
        """a[M,Ka]^T . b[M,Nb] -> [Ka,Nb] (written into `out`, a contiguous (Ka, Nb) view, when given)."""
        a, b = _f32(a, "a"), _f32(b, "b")
        M, Ka = a.shape
        Nb = b.shape[1]
        if out is None:
            out = torch.empty(Ka, Nb, dtype=torch.float32, device=a.device)
        elif out.shape != (Ka, Nb) or not out.is_contiguous() or out.dtype != torch.float32:
            raise ValueError("linear_tn: out must be a contiguous float32 (Ka, Nb) matrix")
        ws = self._ws(self.lib.temp_linear_tn_workspace(M, Ka, Nb), a.device)
        rc = self.lib.temp_linear_tn(M, Ka, Nb, _ptr(a), Ka, _ptr(b), Nb, _ptr(out), Nb, _ptr(ws), ws.numel(), _stream())
        _lib.check(rc, "temp_linear_tn")
        return out

    def linear_tn_multi(self, a_list, b_list, out_list):
        """out_i[Ka,Nb] = a_i[M_i,Ka]^T . b_i[M_i,Nb] for every problem i (contiguous (Ka, Nb) outputs): one launch for small M_i."""
        Ka, Nb = a_list[0].shape[1], b_list[0].shape[1]
        arr = (_lib.TempLinearProblem * len(a_list))()
        keep, max_m = [], 0
        for i, (a, b, o) in enumerate(zip(a_list, b_list, out_list)):
            a, b = _f32(a, "a"), _f32(b, "b")
            keep.append((a, b))
            if a.shape[1] != Ka or b.shape[1] != Nb or a.shape[0] != b.shape[0]:
                raise ValueError("linear_tn_multi: problems must share Ka and Nb")
            if o.shape != (Ka, Nb) or not o.is_contiguous() or o.dtype != torch.float32:
                raise ValueError("linear_tn_multi: outputs must be contiguous float32 (Ka, Nb) matrices")
            arr[i].M, arr[i].A, arr[i].B, arr[i].C = a.shape[0], a.data_ptr(), b.data_ptr(), o.data_ptr()
            max_m = max(max_m, a.shape[0])
        ws = self._ws(self.lib.temp_linear_tn_multi_workspace(len(a_list), max_m, Ka, Nb), a_list[0].device)
        rc = self.lib.temp_linear_tn_multi(len(a_list), arr, Ka, Nb, Ka, Nb, Nb, _ptr(ws), ws.numel(), _stream())
        _lib.check(rc, "temp_linear_tn_multi")
        return out_list

    def gather_ce_fwd(self, scores, cand):
        scores, cand = _f32(scores, "scores"), _i32(cand, "cand")
        P, N = scores.shape
        loss = torch.empty(P, dtype=torch.float32, device=scores.device)
        lse = torch.empty(P, dtype=torch.float32, device=scores.device)
        rc = self.lib.temp_gather_ce_fwd(P, cand.shape[1], N, _ptr(scores), _ptr(cand), _ptr(loss), _ptr(lse), _stream())
        _lib.check(rc, "temp_gather_ce_fwd")
        return loss, lse

    def gather_ce_bwd(self, scores, cand, lse, scale, inv_rows, row_scale=None):
        scores, cand, lse, scale = _f32(scores, "scores"), _i32(cand, "cand"), _f32(lse, "lse"), _f32(scale, "scale")
        row_scale = _f32(row_scale, "row_scale")
        P, N = scores.shape
        d = torch.empty_like(scores)
        rc = self.lib.temp_gather_ce_bwd(P, cand.shape[1], N, _ptr(scores), _ptr(cand), _ptr(lse), _ptr(scale), float(inv_rows), _ptr(row_scale),
                                         _ptr(d), _stream())
        _lib.check(rc, "temp_gather_ce_bwd")
        return d

    def assemble_views(self, piece_desc, piece_start, descs, table, out):
        """One launch of temp_assemble_views; all arguments are device tensors (descs: raw bytes of TempCopyDesc records)."""
        rc = self.lib.temp_assemble_views(int(piece_desc.shape[0]), _ptr(piece_desc), _ptr(piece_start), _ptr(descs), _ptr(table), _ptr(out), _stream())
        _lib.check(rc, "temp_assemble_views")
        return out

    def corrupt_sample(self, seed, truth, lo, hi, ids, K, N):
        """(R, 1+K) int32 candidate lists: column 0 = truth, the rest filtered uniform draws (see temp_corrupt_sample)."""
        truth = _i32(truth, "truth")
        R = truth.shape[0]
        cand = torch.empty(R, K + 1, dtype=torch.int32, device=truth.device)
        if lo is not None:
            lo, hi, ids = _i32(lo, "lo"), _i32(hi, "hi"), _i32(ids, "ids")
        rc = self.lib.temp_corrupt_sample(R, int(K), int(N), int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(truth), _ptr(lo), _ptr(hi), _ptr(ids) if lo is not None else None,
                                          _ptr(cand), _stream())
        _lib.check(rc, "temp_corrupt_sample")
        return cand

    def filtered_rank(self, scores, target, filt_ptr=None, filt_ids=None):
        """1-indexed filtered ranks (int64) of `target` in every row of scores [P,N] (see temp_filtered_rank)."""
        scores, target = _f32(scores, "scores"), _i32(target, "target")
        P, N = scores.shape
        if N % 4:
            raise ValueError("filtered_rank: the score row length must be a multiple of 4")
        ranks = torch.empty(P, dtype=torch.int32, device=scores.device)
        if filt_ptr is not None:
            filt_ptr, filt_ids = _i32(filt_ptr, "filt_ptr"), _i32(filt_ids, "filt_ids")
        rc = self.lib.temp_filtered_rank(P, N, N, _ptr(scores), _ptr(target), _ptr(filt_ptr), _ptr(filt_ids), _ptr(ranks), _stream())
        _lib.check(rc, "temp_filtered_rank")
        return ranks.long()

    # ---- history attention (self-attention encoder) ------------------------------------------------------
    def _attn_desc(self, qkv, kv_hist, idx, decay):
        n, D3 = qkv.shape
        D = D3 // 3
        T = idx.shape[1] + 1
        if T > 1 and kv_hist.shape[1] != 2 * D:
            raise _lib.TempAmdError("kv_hist must be [R, 2D]")
        a = _lib.TempAttn()
        a.n, a.D, a.heads, a.T = n, D, 8, T
        base = qkv.data_ptr()
        a.q, a.ldq = base, D3
        a.kc, a.vc, a.ldc = base + 4 * D, base + 8 * D, D3
        if T > 1:
            hb = kv_hist.data_ptr()
            a.kh, a.vh, a.ldh = hb, hb + 4 * D, 2 * D
        a.idx = idx.data_ptr() if T > 1 else None
        a.decay = decay.data_ptr() if decay is not None else None
        return a, n, D, T

    def sa_attn_fwd(self, qkv, kv_hist, idx, decay):
        """qkv [n,3D] = (q | k | v) of the query rows; kv_hist [R,2D] = (k | v) history table;
        idx [n,T-1] int32 rows of the table (-1 masked); decay [T] or None -> (out, score, lse)."""
        qkv, kv_hist, idx, decay = _f32(qkv, "qkv"), _f32(kv_hist, "kv_hist"), _i32(idx, "idx"), _f32(decay, "decay")
        a, n, D, T = self._attn_desc(qkv, kv_hist, idx, decay)
        out = torch.empty(n, D, dtype=torch.float32, device=qkv.device)
        score = torch.empty(n, 8, T, dtype=torch.float32, device=qkv.device)
        lse = torch.empty(n, 8, dtype=torch.float32, device=qkv.device)
        rc = self.lib.temp_sa_attn_fwd(ctypes.byref(a), _ptr(out), _ptr(score), _ptr(lse), _stream())
        _lib.check(rc, "temp_sa_attn_fwd")
        return out, score, lse

    def sa_attn_bwd(self, qkv, kv_hist, idx, decay, out, score, lse, d_out, inverse=None):
        """-> (d_qkv [n,3D], d_kv_hist [R,2D], d_decay [T] or None).  `inverse` = (inv_ptr, inv_ref) of idx grouped by
        table row selects the deterministic two-pass form (no atomics)."""
        qkv, kv_hist, idx, decay = _f32(qkv, "qkv"), _f32(kv_hist, "kv_hist"), _i32(idx, "idx"), _f32(decay, "decay")
        d_out = _f32(d_out, "d_out")
        a, n, D, T = self._attn_desc(qkv, kv_hist, idx, decay)
        d_qkv = torch.empty_like(qkv)
        use_inv = inverse is not None and T > 1
        d_hist = torch.empty_like(kv_hist) if use_inv else torch.zeros_like(kv_hist)
        d_decay = torch.zeros(T, dtype=torch.float32, device=qkv.device) if decay is not None else None
        db, dh = d_qkv.data_ptr(), d_hist.data_ptr()
        vp = ctypes.c_void_p
        inv_ptr = inv_ref = ds_ws = None
        if use_inv:
            inv_ptr, inv_ref = _i32(inverse[0], "inv_ptr"), _i32(inverse[1], "inv_ref")
            ds_ws = torch.empty(n * 8 * T, dtype=torch.float32, device=qkv.device)
        rc = self.lib.temp_sa_attn_bwd(ctypes.byref(a), _ptr(out), _ptr(score), _ptr(lse), _ptr(d_out),
                                       vp(db), 3 * D, vp(dh), vp(dh + 4 * D), 2 * D, vp(db + 4 * D), vp(db + 8 * D), 3 * D,
                                       _ptr(d_decay), kv_hist.shape[0], _ptr(inv_ptr), _ptr(inv_ref), _ptr(ds_ws), _stream())
        _lib.check(rc, "temp_sa_attn_bwd")
        return d_qkv, d_hist, d_decay

    # ---- row gather / scatter ---------------------------------------------------------------------
    def gather_rows(self, table, idx):
        table, idx = _f32(table, "table"), _i32(idx, "idx")
        out = torch.empty(idx.shape[0], table.shape[1], dtype=torch.float32, device=table.device)
        rc = self.lib.temp_gather_rows(idx.shape[0], table.shape[1], _ptr(table), _ptr(idx), _ptr(out), _stream())
        _lib.check(rc, "temp_gather_rows")
        return out

    def gather_rows_keys(self, table, idx):
        """gather_rows that also returns the magnitude keys of its output: (out, row_keys int32 [n], col_keys int32 [d])."""
        table, idx = _f32(table, "table"), _i32(idx, "idx")
        n, d = idx.shape[0], table.shape[1]
        out = torch.empty(n, d, dtype=torch.float32, device=table.device)
        rk = torch.empty(n, dtype=torch.int32, device=table.device)
        ck = torch.empty(self.lib.temp_keys_cols_size(d), dtype=torch.int32, device=table.device)
        rc = self.lib.temp_gather_rows_keys(n, d, _ptr(table), _ptr(idx), _ptr(out), _ptr(rk), _ptr(ck), _stream())
        _lib.check(rc, "temp_gather_rows_keys")
        return out, rk, ck[:d]

    def keys_supported(self, d):
        """True when row / column keys of a width-d matrix are of use (f16 arithmetic on, width taken by the key kernels)."""
        return d % 4 == 0 and d <= 256 and self.lib.temp_get_option(_lib.OPT_MFMA_F16X2) != 0 and self.lib.temp_get_option(_lib.OPT_MFMA_BF16X3) != 0

    def scatter_add_rows(self, src, idx, table):
        """table[idx[i]] += src[i] in place (table must be contiguous float32)."""
        src, idx = _f32(src, "src"), _i32(idx, "idx")
        if not (table.is_cuda and table.dtype == torch.float32 and table.is_contiguous()):
            raise _lib.TempAmdError("scatter target must be a contiguous float32 GPU tensor")
        rc = self.lib.temp_scatter_add_rows(idx.shape[0], src.shape[1], _ptr(src), _ptr(idx), _ptr(table), _stream())
        _lib.check(rc, "temp_scatter_add_rows")
        return table

    def segment_sum_rows(self, src, seg_ptr, order, n_seg, relu_of=None):
        """out[s] = sum of src[order[seg_ptr[s]:seg_ptr[s+1]]] (deterministic adjoint of a static gather); with relu_of
        [n_seg, d] (the post-ReLU table the gather read): zero where relu_of <= 0 (that ReLU's adjoint, same kernel)."""
        src, seg_ptr, order = _f32(src, "src"), _i32(seg_ptr, "seg_ptr"), _i32(order, "order")
        if src.shape[0] == 0 or order.shape[0] == 0:
            return torch.zeros(n_seg, src.shape[1], dtype=torch.float32, device=src.device)
        out = torch.empty(n_seg, src.shape[1], dtype=torch.float32, device=src.device)
        nb = self.lib.temp_segment_sum_rows_workspace(n_seg, order.shape[0], src.shape[1])
        ws = self._ws(nb, src.device) if nb else None
        if relu_of is not None:
            relu_of = _f32(relu_of, "relu_of")
            if tuple(relu_of.shape) != (n_seg, src.shape[1]):
                raise _lib.TempAmdError("relu_of must be [n_seg, d]")
            rc = self.lib.temp_segment_sum_rows_relu(n_seg, order.shape[0], src.shape[1], _ptr(seg_ptr), _ptr(order), _ptr(src), _ptr(relu_of),
                                                     _ptr(out), _ptr(ws), nb, _stream())
        else:
            rc = self.lib.temp_segment_sum_rows(n_seg, order.shape[0], src.shape[1], _ptr(seg_ptr), _ptr(order), _ptr(src), _ptr(out),
                                                _ptr(ws), nb, _stream())
        _lib.check(rc, "temp_segment_sum_rows")
        return out

    def copy_probe(self, src, dst):
        rc = self.lib.temp_copy_probe(_ptr(src), _ptr(dst), src.numel() * src.element_size(), _stream())
        _lib.check(rc, "temp_copy_probe")
