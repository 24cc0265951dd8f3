"""Minimal stand-in for the `dgl==0.4.1` API surface the TeMP reference touches.

TEST INFRASTRUCTURE ONLY.  Used solely by `oracle/gen_golden.py`, in the build
container, to import the reference's own Python modules (which are never copied
or shipped) and record golden vectors.  DGL is an un-vendored third-party
dependency of the reference (README.md:15 pins dgl-cuda10.1==0.4.1) and is not
installed here; the only *arithmetic* restated in this stub is DGL's builtin
`fn.sum` reducer (sum of messages over in-edges) -- everything else is graph
bookkeeping.  Semantics follow the DGL 0.4 documentation:

  * `update_all(msg, fn.sum(msg=..., out=...), apply)`: message UDF on all edges,
    sum over in-edges into `ndata[out]` (zeros for zero-in-degree nodes), then
    apply UDF on all nodes.
  * `dgl.batch(list)`: disjoint union, node ids offset, ndata/edata concatenated.
  * `edge_subgraph(ids, preserve_nodes=True)`: same node set, selected edges in
    the given order, empty feature dicts.
  * `in_degrees`, `apply_edges`, `local_var` (shallow copy of feature dicts).
"""
import numpy as np
import torch

from . import function  # noqa: F401


class _Batch:
    """What a UDF receives: `.data`, `.src`, `.dst` feature dicts."""

    def __init__(self, data, src=None, dst=None):
        self.data = data
        self.src = src
        self.dst = dst


def _as_long(x):
    if isinstance(x, torch.Tensor):
        return x.long().view(-1)
    return torch.as_tensor(np.asarray(x, dtype=np.int64)).view(-1)


class DGLGraph:
    def __init__(self):
        self._n = 0
        self._src = torch.zeros(0, dtype=torch.long)
        self._dst = torch.zeros(0, dtype=torch.long)
        self.ndata = {}
        self.edata = {}

    # -- topology ---------------------------------------------------------
    def add_nodes(self, n):
        self._n += int(n)

    def add_edges(self, u, v):
        self._src = torch.cat([self._src, _as_long(u)])
        self._dst = torch.cat([self._dst, _as_long(v)])

    def edges(self):
        return self._src, self._dst

    def nodes(self):
        return torch.arange(self._n)

    def number_of_nodes(self):
        return self._n

    def number_of_edges(self):
        return int(self._src.shape[0])

    def in_degrees(self, v=None):
        deg = torch.bincount(self._dst, minlength=self._n)
        if v is None:
            return deg
        return deg[_as_long(list(v) if isinstance(v, range) else v)]

    # -- views ------------------------------------------------------------
    def local_var(self):
        g = DGLGraph()
        g._n, g._src, g._dst = self._n, self._src, self._dst
        g.ndata = dict(self.ndata)
        g.edata = dict(self.edata)
        if hasattr(self, 'ids'):
            g.ids = self.ids
        return g

    def edge_subgraph(self, idx, preserve_nodes=False):
        assert preserve_nodes, "stub only implements preserve_nodes=True"
        idx = _as_long(idx)
        g = DGLGraph()
        g._n = self._n
        g._src, g._dst = self._src[idx], self._dst[idx]
        return g

    # -- message passing --------------------------------------------------
    def _edge_batch(self):
        dev_src = {k: v[self._src.to(v.device)] for k, v in self.ndata.items()}
        dev_dst = {k: v[self._dst.to(v.device)] for k, v in self.ndata.items()}
        return _Batch(self.edata, dev_src, dev_dst)

    def apply_edges(self, func):
        self.edata.update(func(self._edge_batch()))

    def update_all(self, message_func, reduce_func, apply_node_func=None):
        assert isinstance(reduce_func, function._Sum)
        if self.number_of_edges() > 0:
            msg = message_func(self._edge_batch())[reduce_func.msg]
            out = msg.new_zeros((self._n,) + tuple(msg.shape[1:]))
            out = out.index_add(0, self._dst.to(msg.device), msg)
            self.ndata[reduce_func.out] = out
        if apply_node_func is not None:
            self.ndata.update(apply_node_func(_Batch(self.ndata)))


def batch(graph_list):
    g = DGLGraph()
    off = 0
    srcs, dsts = [], []
    for gi in graph_list:
        srcs.append(gi._src + off)
        dsts.append(gi._dst + off)
        off += gi._n
    g._n = off
    if srcs:
        g._src, g._dst = torch.cat(srcs), torch.cat(dsts)
        for k in graph_list[0].ndata:
            g.ndata[k] = torch.cat([gi.ndata[k] for gi in graph_list], dim=0)
        for k in graph_list[0].edata:
            g.edata[k] = torch.cat([gi.edata[k] for gi in graph_list], dim=0)
    return g
