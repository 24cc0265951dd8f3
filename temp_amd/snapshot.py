"""Snapshot graphs: the DGL-graph field contract of the TeMP reference as plain arrays, plus the
sorted/chunked edge views the HIP kernels consume (include/temp_amd.h, TempEdgeView/TempGraph).

Reference contract being mirrored (utils/dataset.py:210-231, models/DynamicRGCN.py:86-93):
    g.ndata['id'] (n,1) int64 global entity id      g.ndata['norm'] (n,1) f32 = 1/in_deg (inf -> 0)
    g.edata['type_s'] (E,) int64                    g.edata['norm'] (E,1) f32 = norm of the dst node
    g.edges() -> (src, dst) local ids               g.ids {local -> global}
    g.ndata['h'] (n,D) f32 set by the caller before the encoder runs
"""
import ctypes

import numpy as np
import torch

from . import _lib


def comp_deg_norm(n, dst):
    """utils/utils.py:74-79: 1/in_degree in fp32, inf -> 0."""
    in_deg = np.bincount(np.asarray(dst, dtype=np.int64), minlength=n).astype(np.float32)
    with np.errstate(divide="ignore"):
        norm = (np.float32(1.0) / in_deg).astype(np.float32)
    norm[np.isinf(norm)] = 0
    return norm


class Snapshot:
    """One per-timestamp graph (or a batch of them).  Host-side arrays are numpy; `ndata`/`edata`
    dictionaries expose the reference's tensor views lazily."""

    def __init__(self, n, src, dst, rel, ids, nnorm=None):
        self.n = int(n)
        self.src = np.ascontiguousarray(src, dtype=np.int64).reshape(-1)
        self.dst = np.ascontiguousarray(dst, dtype=np.int64).reshape(-1)
        self.rel = np.ascontiguousarray(rel, dtype=np.int64).reshape(-1)
        self.gids = np.ascontiguousarray(ids, dtype=np.int64).reshape(-1)
        assert self.gids.shape[0] == self.n and self.src.shape == self.dst.shape == self.rel.shape
        self.nnorm = comp_deg_norm(self.n, self.dst) if nnorm is None else np.ascontiguousarray(nnorm, np.float32).reshape(-1)
        self.ndata = {}          # 'h' is put here by the caller, as in the reference
        self._dev = {}           # device -> _DeviceGraph
        self._ids_dict = None
        self._views = {}         # n_rel_rows -> host-side sorted views (static per snapshot, built once)

    # ---- reference-style accessors --------------------------------------------------------
    @property
    def ids(self):
        """g.ids: {local index -> global entity id} (utils/dataset.py:225-231)."""
        if self._ids_dict is None:
            self._ids_dict = {i: int(v) for i, v in enumerate(self.gids)}
        return self._ids_dict

    def edges(self):
        return torch.from_numpy(self.src), torch.from_numpy(self.dst)

    def nodes(self):
        return torch.arange(self.n)

    def number_of_nodes(self):
        return self.n

    def number_of_edges(self):
        return int(self.src.shape[0])

    def edge_extent(self):
        """Length of this snapshot's edge arrays inside a view buffer (= number_of_edges() unless edges were dropped in place)."""
        return self.number_of_edges()

    def local_var(self):
        """Shallow copy sharing topology and device views (DGL's local_var)."""
        g = self.__class__.__new__(self.__class__)
        g.__dict__.update(self.__dict__)
        g.ndata = dict(self.ndata)
        return g

    def edge_subgraph(self, idx):
        """models/DynamicRGCN.py:80-90: keep edges `idx` (that order), same nodes, norms recomputed
        from the subgraph's in-degrees."""
        idx = np.asarray(idx, dtype=np.int64)
        return Snapshot(self.n, self.src[idx], self.dst[idx], self.rel[idx], self.gids)

    # ---- device views -----------------------------------------------------------------------
    def local_views(self, n_rel_rows):
        """Host-side sorted / chunked views of THIS graph (local node ids), cached: membership is static."""
        v = self._views.get(n_rel_rows)
        if v is None:
            E = self.number_of_edges()
            if E and (self.rel.min() < 0 or self.rel.max() >= n_rel_rows):
                raise ValueError("relation id outside [0, %d)" % n_rel_rows)
            # (a node's edges in relation order: the chunks of a hub are runs of one relation -- see host_planner.cpp)
            v = dict(by_dst=build_view(self.dst, self.src, self.rel, self.n, sort_b=True), by_src=build_view(self.src, self.dst, self.rel, self.n, sort_b=True),
                     by_rel=build_view(self.rel, self.src, self.dst, n_rel_rows, chunk=_lib.CHUNK_REL, sort_b=True),   # (a relation's edges in destination order)
                     in_deg=np.bincount(self.dst, minlength=self.n).astype(np.int32),
                     out_deg=np.bincount(self.src, minlength=self.n).astype(np.int32))
            v["rel_chunks"] = np.bincount(v["by_rel"]["chunk_seg"], minlength=n_rel_rows).astype(np.int64)
            self._views[n_rel_rows] = v
        return v

    def device_views(self, device, n_rel_rows):
        """The cached local views as int32 tensors resident on `device` (the snapshot store: uploaded once per snapshot)."""
        key = ("views", str(device), int(n_rel_rows))
        dv = self._dev.get(key)
        if dv is None:
            with _lib.create_lock:
                dv = self._dev.get(key)
                if dv is None:
                    dv = self._dev[key] = self._build_device_views(device, n_rel_rows)
        return dv

    def _build_device_views(self, device, n_rel_rows):
        names = [(vn, an) for vn in ("by_dst", "by_src", "by_rel") for an in _VIEW_ARRAYS] + ["rel_rank", "in_deg", "out_deg", "nnorm"]
        stored = self.stored_pack(n_rel_rows)
        if stored is not None:                                 # precomputed in the on-disk store (temp_amd/store.py)
            packed, sizes_np, n_partial, rel_chunks = stored
            sizes = [int(x) for x in sizes_np]
        elif self._views.get(n_rel_rows) is None:
            # no host-side views cached (a freshly subsampled target graph): all three views + the pack in ONE pass of
            # the host planner library
            from . import _hostlib
            packed, sizes_np, n_partial, rel_chunks = _hostlib.snapshot_pack(self.n, self.src, self.dst, self.rel, self.nnorm, n_rel_rows,
                                                                             _lib.CHUNK, _lib.CHUNK_REL)
            sizes = [int(x) for x in sizes_np]
        else:
            lv = self.local_views(n_rel_rows)
            seg = lv["by_rel"]["chunk_seg"].astype(np.int64)
            first = np.cumsum(lv["rel_chunks"]) - lv["rel_chunks"]
            arrays = [lv[vn][an] for vn in ("by_dst", "by_src", "by_rel") for an in _VIEW_ARRAYS]
            arrays += [np.arange(seg.shape[0], dtype=np.int64) - first[seg],        # rank of a chunk inside its relation
                       lv["in_deg"], lv["out_deg"], np.ascontiguousarray(self.nnorm, dtype=np.float32).view(np.int32)]   # float bits ride along
            sizes = [int(a.shape[0]) for a in arrays]
            packed = np.concatenate([np.ascontiguousarray(a, dtype=np.int32) for a in arrays]) if sum(sizes) else np.zeros(1, np.int32)
            n_partial = np.array([lv[vn]["n_partial"] for vn in ("by_dst", "by_src", "by_rel")], dtype=np.int64)
            rel_chunks = lv["rel_chunks"]
        buf = self._dev.pop(("adopted", str(device), int(n_rel_rows)), None)      # already resident (SnapshotStore.to_device)
        if buf is None:
            buf = _lib.to_device(np.require(packed, requirements=['C', 'W']), device)   # ONE upload per snapshot
        dv = {"by_dst": {}, "by_src": {}, "by_rel": {}, "_buf": buf}
        off = 0
        for key_, n_ in zip(names, sizes):
            t = buf[off:off + n_]
            off += n_
            if isinstance(key_, tuple):
                dv[key_[0]][key_[1]] = t
            else:
                dv[key_] = t
        dv["nnorm"] = dv["nnorm"].view(torch.float32)
        offs = np.concatenate([[0], np.cumsum(sizes)])[:-1]
        dv["_meta"] = dict(ptr=int(buf.data_ptr()), off=offs.astype(np.int64), size=np.asarray(sizes, dtype=np.int64),
                           n_partial=np.asarray(n_partial, dtype=np.int64), rel_chunks=np.asarray(rel_chunks, dtype=np.int64))
        m = dv["_meta"]                                      # one int64 row for the host planner (temp_host_union_plan)
        m["row"] = np.concatenate([m["size"], m["off"], [m["ptr"]], m["n_partial"], m["rel_chunks"][:n_rel_rows],
                                   np.zeros(max(0, n_rel_rows - m["rel_chunks"].shape[0]), np.int64)]).astype(np.int64)
        _lib.publish(device)
        return dv

    def device_graph(self, device, n_rel_rows):
        key = (str(device), int(n_rel_rows))
        dg = self._dev.get(key)
        if dg is None:
            with _lib.create_lock:
                dg = self._dev.get(key)
                if dg is None:
                    dg = _DeviceGraph(self, device, n_rel_rows)
                    # a resident snapshot is shared by every batch (and worker stream) that visits it: visible only once its uploads
                    # have landed.  The union of ONE batch (BatchedSnapshot, built per prepare) is read by that batch alone, on
                    # the stream that built it or behind the batch's `ready` event -- draining the stream here made every inline
                    # prepare wait for the previous training step's kernels (the host then never ran ahead of the device)
                    if not isinstance(self, BatchedSnapshot):
                        _lib.publish(device)
                    self._dev[key] = dg
        return dg

    def stored_pack(self, n_rel_rows):
        """(packed, sizes, n_partial, rel_chunks) when the packed views were precomputed (StoredSnapshot, or prepack()), else None."""
        return self.__dict__.get("_pack", {}).pop(int(n_rel_rows), None)

    def prepack(self, n_rel_rows):
        """Build the packed views on the host NOW, outside the creation lock (device_views then only uploads): lets several
        snapshots be packed by several threads -- the pack is one call into the host planner library, which runs without the
        interpreter lock (prepack_parallel)."""
        if self._dev or self._views.get(int(n_rel_rows)) is not None or type(self).stored_pack is not Snapshot.stored_pack:
            return
        from . import _hostlib
        self.__dict__.setdefault("_pack", {})[int(n_rel_rows)] = _hostlib.snapshot_pack(self.n, self.src, self.dst, self.rel, self.nnorm, n_rel_rows,
                                                                                      _lib.CHUNK, _lib.CHUNK_REL)

    def adopt_device_pack(self, buf, device, n_rel_rows):
        """Hand this snapshot a device-resident copy of its packed views (a slice of a whole-store upload)."""
        self._dev[("adopted", str(torch.device(device)), int(n_rel_rows))] = buf

    def device_edge_ids(self, device):
        """[3, E] int32 on `device`: original edge id of every position of the by-dst / by-src / by-rel view (stable sorts of the
        edge list by (dst, rel) / (src, rel) / (rel, dst)), uploaded once per snapshot."""
        key = ("eid", str(device))
        t = self._dev.get(key)
        if t is None:
            with _lib.create_lock:
                t = self._dev.get(key)
                if t is None:
                    eid = np.stack([np.lexsort((self.rel, self.dst)), np.lexsort((self.rel, self.src)), np.lexsort((self.dst, self.rel))]).astype(np.int32) \
                        if self.number_of_edges() else np.zeros((3, 0), np.int32)
                    t = _lib.to_device(eid, device)
                    _lib.publish(device)
                    self._dev[key] = t
        return t


def device_subsample(graphs, keeps, seeds, device, n_rel_rows, want_mask=False):
    """Random edge subsets of resident snapshots, ON the device (temp_subsample_views): graph i keeps `keeps[i]` edges drawn
    by `seeds[i]`.  -> list of SubsampledSnapshot whose resident views are derived from their parents' (one clone of the
    parent's packed buffer + three launches for the whole list): no host sort, no edge upload."""
    from .backend import get_backend
    jobs, out = [], []
    scratch = torch.zeros(max(len(graphs), 1), dtype=torch.int64, device=device)
    for i, (g, k, seed) in enumerate(zip(graphs, keeps, seeds)):
        dv = g.device_views(device, n_rel_rows)
        meta = dv["_meta"]
        child = dv["_buf"].clone()
        E = g.number_of_edges()
        mask = torch.empty(E, dtype=torch.uint8, device=device) if want_mask else None
        static = meta.get("subsample_job")                   # the parent's part of the job: offsets inside its packed buffer
        if static is None:
            off = {name: int(o) for name, o in zip(_PACK_NAMES, meta["off"])}
            views = ("by_dst", "by_src", "by_rel")
            static = meta["subsample_job"] = dict(
                n_nodes=g.n, n_edges=E, parent=dv["_buf"], eid=g.device_edge_ids(device),
                off_a=[off[(v, "a")] for v in views], off_b=[off[(v, "b")] for v in views],
                off_chunk_beg=[off[(v, "chunk_beg")] for v in views], off_chunk_end=[off[(v, "chunk_end")] for v in views],
                off_chunk_seg=[off[(v, "chunk_seg")] for v in views], n_chunks=[int(dv[v]["chunk_seg"].shape[0]) for v in views],
                off_in_deg=off["in_deg"], off_out_deg=off["out_deg"], off_nnorm=off["nnorm"])
        jobs.append(dict(static, keep=int(k), seed=int(seed), child=child, keep_mask=mask, scratch=scratch[i:i + 1]))
        out.append(SubsampledSnapshot(g, int(k), child, mask, device, n_rel_rows))
    if jobs:
        get_backend().subsample_views(jobs)
    return out


class _PackedViews(dict):
    """device_views() of a snapshot whose packed buffer already sits on the device: `_buf` / `_meta` are there from the start,
    the per-array slices (by_dst / by_src / by_rel -> array, rel_rank, in_deg, out_deg, nnorm) are cut on first access."""

    def __init__(self, buf, parent_meta):
        super().__init__(_buf=buf)
        self._pm = parent_meta

    def __missing__(self, key):
        if self._pm is None:
            raise KeyError(key)
        buf, pm, self._pm = dict.__getitem__(self, "_buf"), self._pm, None
        for vn in ("by_dst", "by_src", "by_rel"):
            dict.__setitem__(self, vn, {})
        for name, o, n_ in zip(_PACK_NAMES, pm["off"], pm["size"]):
            t = buf[int(o):int(o) + int(n_)]
            if isinstance(name, tuple):
                dict.__getitem__(self, name[0])[name[1]] = t
            else:
                dict.__setitem__(self, name, t)
        dict.__setitem__(self, "nnorm", dict.__getitem__(self, "nnorm").view(torch.float32))
        return self[key]


class SubsampledSnapshot(Snapshot):
    """A snapshot restricted to a random edge subset that exists on the DEVICE only: same nodes, same view layout as its
    parent (chunk tables shared), `number_of_edges()` = size of the subset.  Host edge arrays / norms are materialised from
    the keep mask only if something asks for them (a device -> host copy)."""

    def __init__(self, parent, keep, child_buf, mask, device, n_rel_rows):
        self.n = parent.n
        self.gids = parent.gids
        self.parent, self.keep, self._mask = parent, int(keep), mask
        self.ndata = {}
        self._dev = {}
        self._ids_dict = None
        self._views = {}
        self._host = None
        pdv = parent.device_views(device, n_rel_rows)
        pm = pdv["_meta"]
        dv = _PackedViews(child_buf, pm)              # the per-array tensors only if something indexes them (the union does not)
        m = dict(ptr=int(child_buf.data_ptr()), off=pm["off"], size=pm["size"], n_partial=pm["n_partial"], rel_chunks=pm["rel_chunks"])
        row = pm["row"].copy()
        row[62] = m["ptr"]
        m["row"] = row
        dv["_meta"] = m
        self._dev[("views", str(device), int(n_rel_rows))] = dv

    def number_of_edges(self):
        return self.keep

    def edge_extent(self):
        return self.parent.number_of_edges()     # the views keep the parent's array length; dropped positions are simply unused

    def _materialise(self):
        if self._host is None:
            if self._mask is None:
                raise RuntimeError("this device-side subsample was created without a keep mask (want_mask=False)")
            idx = np.nonzero(self._mask.cpu().numpy())[0]
            p = self.parent
            self._host = (p.src[idx], p.dst[idx], p.rel[idx], comp_deg_norm(self.n, p.dst[idx]), idx)
        return self._host

    src = property(lambda self: self._materialise()[0])
    dst = property(lambda self: self._materialise()[1])
    rel = property(lambda self: self._materialise()[2])
    nnorm = property(lambda self: self._materialise()[3])
    edge_ids = property(lambda self: self._materialise()[4])

    def local_views(self, n_rel_rows):
        raise RuntimeError("a device-side subsample has no host views; use device_views()")


class BatchedSnapshot(Snapshot):
    """Disjoint union of snapshots (dgl.batch, models/DynamicRGCN.py:92).  The member snapshots are kept (`parts`):
    their sorted edge views are static and cached, so the union's device views are assembled by offsetting and
    concatenating them instead of re-sorting millions of edges for every batch; the union's own src / dst / rel
    arrays are only materialised if something asks for them."""

    def __init__(self, parts):
        self.parts = list(parts)
        self.node_sizes = [g.n for g in self.parts]
        self.node_off = np.concatenate([[0], np.cumsum(self.node_sizes)]).astype(np.int64)
        # offsets of the members' edge ARRAYS inside the union's views (a device-subsampled member keeps its parent's length)
        self.edge_off = np.concatenate([[0], np.cumsum([g.edge_extent() for g in self.parts])]).astype(np.int64)
        self._n_edges = int(sum(g.number_of_edges() for g in self.parts))
        self.n = int(self.node_off[-1])
        cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dtype=dt)
        self.gids = cat([g.gids for g in self.parts], np.int64)
        self._nnorm = None                       # host norms only on demand (a device-subsampled member has them on the GPU)
        self.ndata = {}
        self._dev = {}
        self._ids_dict = None
        self._views = {}
        self._edges = None

    @property
    def nnorm(self):
        if self._nnorm is None:
            self._nnorm = np.concatenate([g.nnorm for g in self.parts]).astype(np.float32) if self.parts else np.zeros(0, np.float32)
        return self._nnorm

    def _materialise(self):
        if self._edges is None:
            cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, dtype=np.int64)
            self._edges = (cat([g.src + o for g, o in zip(self.parts, self.node_off)]),
                           cat([g.dst + o for g, o in zip(self.parts, self.node_off)]), cat([g.rel for g in self.parts]))
        return self._edges

    src = property(lambda self: self._materialise()[0])
    dst = property(lambda self: self._materialise()[1])
    rel = property(lambda self: self._materialise()[2])

    def number_of_edges(self):
        return self._n_edges

    def edge_extent(self):
        return int(self.edge_off[-1])


def batch(snapshots):
    """dgl.batch as used at models/DynamicRGCN.py:92: disjoint union with node-id offsets."""
    return BatchedSnapshot(snapshots)


def build_view(seg, a, b, n_seg, chunk=_lib.CHUNK, sort_b=False):
    """Sort edges by `seg` (stable; sort_b: by (seg, b)) and cut every segment into chunks of <= `chunk` edges.
    Returns a dict of int32 numpy arrays + counts (layout of TempEdgeView).  One counting-sort pass in the host
    planner library (temp_host_build_view); tests/host_reference.py holds the numpy formulation it is checked against."""
    from . import _hostlib
    return _hostlib.build_view(seg, a, b, n_seg, chunk, sort_b)


def build_view_tiled(seg, a, b, n_seg, tile, chunk=_lib.CHUNK):
    """Same contract as build_view, but the chunks are ordered TILE-major: `tile[e]` (a coarse bucket of
    the edge's node range, i.e. of its snapshot) first, `seg` second.  A segment's edges are then spread
    over one chunk (or a few) per tile, all of which go through partial slots and the fix-up pass --
    slots of one segment stay contiguous -- while the rows a run of consecutive chunks gathers all lie in
    one tile's node range (L2-resident) instead of anywhere in the batch."""
    seg = np.asarray(seg, dtype=np.int64)
    tile = np.asarray(tile, dtype=np.int64)
    n_tile = int(tile.max()) + 1 if tile.shape[0] else 1
    key = tile * n_seg + seg
    order = np.lexsort((np.asarray(b, dtype=np.int64), key))                  # inside a (tile, segment) group: ascending b (destination runs)
    counts = np.bincount(key, minlength=n_tile * n_seg).astype(np.int64)      # edges per (tile, seg) group
    ptr = np.concatenate([[0], np.cumsum(counts)])
    nch = (counts + chunk - 1) // chunk
    total = int(nch.sum())
    grp = np.repeat(np.arange(n_tile * n_seg, dtype=np.int64), nch)           # group of every chunk, chunk order
    first = np.cumsum(nch) - nch
    k = np.arange(total, dtype=np.int64) - first[grp]
    chunk_beg = ptr[grp] + k * chunk
    chunk_end = np.minimum(chunk_beg + chunk, ptr[grp + 1])
    chunk_seg = grp % n_seg
    per_seg = nch.reshape(n_tile, n_seg).sum(axis=0)                          # chunks per segment over all tiles
    multi = per_seg > 1
    fix_seg = np.nonzero(multi)[0]
    fix_cnt = per_seg[fix_seg]
    fix_slot = np.cumsum(fix_cnt) - fix_cnt
    seg_slot0 = np.full(n_seg, -1, dtype=np.int64)
    seg_slot0[fix_seg] = fix_slot
    # rank of a chunk among the chunks of its segment (tile order, then position inside the group)
    by_seg = np.argsort(chunk_seg, kind="stable")
    rank = np.empty(total, dtype=np.int64)
    seg_first = np.cumsum(per_seg) - per_seg
    rank[by_seg] = np.arange(total, dtype=np.int64) - seg_first[chunk_seg[by_seg]]
    slot = np.where(multi[chunk_seg], seg_slot0[chunk_seg] + rank, -1)
    i32 = lambda x: np.ascontiguousarray(x, dtype=np.int32)
    return dict(n_seg=int(n_seg), n_edges=int(seg.shape[0]), a=i32(np.asarray(a)[order]), b=i32(np.asarray(b)[order]),
                n_chunks=total, chunk_seg=i32(chunk_seg), chunk_beg=i32(chunk_beg), chunk_end=i32(chunk_end),
                chunk_slot=i32(slot), n_partial=int(multi[chunk_seg].sum()), n_fix=int(fix_seg.shape[0]),
                fix_seg=i32(fix_seg), fix_slot=i32(fix_slot), fix_cnt=i32(fix_cnt), order=order)


REL_GROUP_EDGES = 192        # target edges per (node tile, relation) group of the by-relation view


def by_rel_view(snap, n_rel_rows):
    """Relation-sorted view for the weight gradient.  When a relation has many edges per snapshot
    (GDELT-like: ~190), the view is tiled by node range so consecutive chunks read one snapshot's rows."""
    E, n = snap.number_of_edges(), snap.n
    tiles = E // (REL_GROUP_EDGES * max(n_rel_rows, 1))
    if tiles <= 1 or n == 0:
        return build_view(snap.rel, snap.src, snap.dst, n_rel_rows, chunk=_lib.CHUNK_REL, sort_b=True)
    width = -(-n // tiles)
    return build_view_tiled(snap.rel, snap.src, snap.dst, n_rel_rows, np.asarray(snap.dst, dtype=np.int64) // width, chunk=_lib.CHUNK_REL)


def _concat_node_views(parts, name, node_off, edge_off, n_total):
    """by_dst / by_src view of a disjoint union from the members' cached views: segments are nodes, so the union's
    sorted order is the members' orders back to back."""
    vs = [g[name] for g in parts]
    p_off = np.concatenate([[0], np.cumsum([v["n_partial"] for v in vs])]).astype(np.int64)
    cat = lambda xs: np.concatenate(xs).astype(np.int32) if xs else np.zeros(0, np.int32)
    out = dict(n_seg=int(n_total), n_edges=int(edge_off[-1]),
               a=cat([v["a"] + no for v, no in zip(vs, node_off)]), b=cat([v["b"] for v in vs]),
               chunk_seg=cat([v["chunk_seg"] + no for v, no in zip(vs, node_off)]),
               chunk_beg=cat([v["chunk_beg"] + eo for v, eo in zip(vs, edge_off)]),
               chunk_end=cat([v["chunk_end"] + eo for v, eo in zip(vs, edge_off)]),
               chunk_slot=cat([np.where(v["chunk_slot"] >= 0, v["chunk_slot"] + po, -1) for v, po in zip(vs, p_off)]),
               fix_seg=cat([v["fix_seg"] + no for v, no in zip(vs, node_off)]),
               fix_slot=cat([v["fix_slot"] + po for v, po in zip(vs, p_off)]), fix_cnt=cat([v["fix_cnt"] for v in vs]))
    out["n_chunks"], out["n_partial"], out["n_fix"] = int(out["chunk_seg"].shape[0]), int(p_off[-1]), int(out["fix_seg"].shape[0])
    return out


def _concat_rel_views(parts, node_off, edge_off, n_rel_rows):
    """Tiled by-relation view of a disjoint union with tile = member snapshot (what build_view_tiled produces for
    tile[e] = member of e): chunks member-major, every chunk of a relation that has more than one chunk in the whole
    union goes through a partial slot, slots of one relation contiguous (ordered by member)."""
    vs = [g["by_rel"] for g in parts]
    counts = np.stack([g["rel_chunks"] for g in parts]) if parts else np.zeros((0, n_rel_rows), np.int64)   # (members, rels)
    per_rel = counts.sum(axis=0)
    multi = per_rel > 1
    fix_seg = np.nonzero(multi)[0]
    fix_cnt = per_rel[fix_seg]
    fix_slot = np.cumsum(fix_cnt) - fix_cnt
    base = np.full(n_rel_rows, -1, dtype=np.int64)
    base[fix_seg] = fix_slot
    before = np.cumsum(counts, axis=0) - counts                         # chunks of relation r in earlier members
    slots = []
    for m, v in enumerate(vs):
        seg = v["chunk_seg"].astype(np.int64)
        first = np.cumsum(counts[m]) - counts[m]                        # first chunk of each relation inside this member
        k = np.arange(seg.shape[0], dtype=np.int64) - first[seg]
        slots.append(np.where(multi[seg], base[seg] + before[m][seg] + k, -1))
    cat = lambda xs: np.concatenate(xs).astype(np.int32) if xs else np.zeros(0, np.int32)
    out = dict(n_seg=int(n_rel_rows), n_edges=int(edge_off[-1]),
               a=cat([v["a"] + no for v, no in zip(vs, node_off)]), b=cat([v["b"] + no for v, no in zip(vs, node_off)]),
               chunk_seg=cat([v["chunk_seg"] for v in vs]),
               chunk_beg=cat([v["chunk_beg"] + eo for v, eo in zip(vs, edge_off)]),
               chunk_end=cat([v["chunk_end"] + eo for v, eo in zip(vs, edge_off)]),
               chunk_slot=cat(slots), fix_seg=fix_seg.astype(np.int32), fix_slot=fix_slot.astype(np.int32), fix_cnt=fix_cnt.astype(np.int32))
    out["n_chunks"], out["n_partial"], out["n_fix"] = int(out["chunk_seg"].shape[0]), int(per_rel[multi].sum()), int(fix_seg.shape[0])
    return out


def union_views(snap, n_rel_rows):
    """(views, in_deg, out_deg) of a BatchedSnapshot from its members' cached views."""
    lv = [g.local_views(n_rel_rows) for g in snap.parts]
    no, eo = snap.node_off[:-1], snap.edge_off[:-1]
    E = int(snap.edge_off[-1])
    views = dict(by_dst=_concat_node_views(lv, "by_dst", no, np.append(eo, E), snap.n),
                 by_src=_concat_node_views(lv, "by_src", no, np.append(eo, E), snap.n))
    if E // (REL_GROUP_EDGES * max(n_rel_rows, 1)) > 1:                  # GDELT-like: a relation has many edges per snapshot
        views["by_rel"] = _concat_rel_views(lv, no, np.append(eo, E), n_rel_rows)
    else:                                                               # few edges per relation: one global sort (cheap at this size)
        views["by_rel"] = build_view(snap.rel, snap.src, snap.dst, n_rel_rows, chunk=_lib.CHUNK_REL, sort_b=True)
    cat = lambda k: np.concatenate([g[k] for g in lv]) if lv else np.zeros(0, np.int32)
    return views, cat("in_deg"), cat("out_deg")


DEVICE_STORE = "kernel"  # union views assembled on the GPU from per-snapshot resident views: "kernel" = one temp_assemble_views launch,
                         # True = torch offset-and-concatenate ops, False = host concatenation + upload


def upload_packed(arrays, device, dtype=np.int64):
    """name -> 1-D host array  =>  name -> device tensor, through ONE concatenated upload (each small synchronous copy
    costs tens of microseconds, more while the GPU is busy; a window batch needs dozens of these little index vectors)."""
    names = list(arrays)
    flat = [np.ascontiguousarray(arrays[k], dtype=dtype).reshape(-1) for k in names]
    buf = _lib.to_device(np.concatenate(flat) if flat else np.zeros(0, dtype), device)
    out, off = {}, 0
    for k, a in zip(names, flat):
        out[k] = buf[off:off + a.shape[0]]
        off += a.shape[0]
    return out


def union_views_device(snap, n_rel_rows, device):
    """Device-side union_views: the members' cached device views are concatenated and offset with a handful of torch
    kernels; only per-member counts / offsets (a few KB, ONE packed upload) come from the host.  Returns (views: name ->
    {array -> tensor, count -> int}, in_deg, out_deg, nnorm) or None when the by-relation view needs the global sort (few
    edges per relation)."""
    E = int(snap.edge_off[-1])
    if E // (REL_GROUP_EDGES * max(n_rel_rows, 1)) <= 1:
        return None
    lv = [g.local_views(n_rel_rows) for g in snap.parts]
    dv = [g.device_views(device, n_rel_rows) for g in snap.parts]
    # ---- host side: every small vector the assembly needs, uploaded together
    host = dict(node_off=snap.node_off[:-1], edge_off=snap.edge_off[:-1])
    for vn in ("by_dst", "by_src", "by_rel"):
        vs = [l[vn] for l in lv]
        host[vn + "/e"] = np.array([v["n_edges"] for v in vs], dtype=np.int64)
        host[vn + "/c"] = np.array([v["n_chunks"] for v in vs], dtype=np.int64)
        if vn != "by_rel":
            host[vn + "/f"] = np.array([v["n_fix"] for v in vs], dtype=np.int64)
            host[vn + "/p"] = np.concatenate([[0], np.cumsum([v["n_partial"] for v in vs])])[:-1]
    counts = np.stack([l["rel_chunks"] for l in lv])
    per_rel = counts.sum(axis=0)
    multi = per_rel > 1
    fix_seg = np.nonzero(multi)[0]
    fix_cnt = per_rel[fix_seg]
    fix_slot = np.cumsum(fix_cnt) - fix_cnt
    base = np.full(n_rel_rows, -1, dtype=np.int64)
    base[fix_seg] = fix_slot
    host["rel/table"] = np.where(multi[None, :], base[None, :] + (np.cumsum(counts, axis=0) - counts), -1).reshape(-1)    # (members, rels)
    host["rel/member"] = np.arange(len(lv)) * n_rel_rows
    host["rel/fix_seg"], host["rel/fix_slot"], host["rel/fix_cnt"] = fix_seg, fix_slot, fix_cnt
    d = upload_packed(host, device)
    node_off, edge_off = d["node_off"], d["edge_off"]

    def rep(vals, counts_dev, counts_np):
        return torch.repeat_interleave(vals, counts_dev, output_size=int(counts_np.sum()))

    def cat(vn, an):
        return torch.cat([x[vn][an] for x in dv])

    views = {}
    for vn in ("by_dst", "by_src", "by_rel"):
        vs = [l[vn] for l in lv]
        e_np, c_np = host[vn + "/e"], host[vn + "/c"]
        node_e, node_c, edge_c = rep(node_off, d[vn + "/e"], e_np), rep(node_off, d[vn + "/c"], c_np), rep(edge_off, d[vn + "/c"], c_np)
        out = dict(n_edges=E, n_chunks=int(c_np.sum()))
        out["a"] = (cat(vn, "a") + node_e).to(torch.int32)
        out["chunk_beg"] = (cat(vn, "chunk_beg") + edge_c).to(torch.int32)
        out["chunk_end"] = (cat(vn, "chunk_end") + edge_c).to(torch.int32)
        if vn != "by_rel":
            f_np = host[vn + "/f"]
            out["b"] = cat(vn, "b")
            out["chunk_seg"] = (cat(vn, "chunk_seg") + node_c).to(torch.int32)
            slot = cat(vn, "chunk_slot").to(torch.int64)
            out["chunk_slot"] = torch.where(slot >= 0, slot + rep(d[vn + "/p"], d[vn + "/c"], c_np), slot).to(torch.int32)
            out["fix_seg"] = (cat(vn, "fix_seg") + rep(node_off, d[vn + "/f"], f_np)).to(torch.int32)
            out["fix_slot"] = (cat(vn, "fix_slot") + rep(d[vn + "/p"], d[vn + "/f"], f_np)).to(torch.int32)
            out["fix_cnt"] = cat(vn, "fix_cnt")
            out["n_seg"], out["n_partial"], out["n_fix"] = int(snap.n), int(sum(v["n_partial"] for v in vs)), int(f_np.sum())
        else:                                            # tile = member snapshot, see _concat_rel_views
            seg = cat(vn, "chunk_seg").to(torch.int64)
            tab = d["rel/table"][rep(d["rel/member"], d[vn + "/c"], c_np) + seg]
            rank = torch.cat([x["rel_rank"] for x in dv]).to(torch.int64)
            out["b"] = (cat(vn, "b") + node_e).to(torch.int32)
            out["chunk_seg"] = seg.to(torch.int32)
            out["chunk_slot"] = torch.where(tab >= 0, tab + rank, tab).to(torch.int32)
            out["fix_seg"], out["fix_slot"], out["fix_cnt"] = (d[k].to(torch.int32) for k in ("rel/fix_seg", "rel/fix_slot", "rel/fix_cnt"))
            out["n_seg"], out["n_partial"], out["n_fix"] = int(n_rel_rows), int(per_rel[multi].sum()), int(fix_seg.shape[0])
        views[vn] = out
    in_deg = torch.cat([x["in_deg"] for x in dv])
    out_deg = torch.cat([x["out_deg"] for x in dv])
    nnorm = torch.cat([x["nnorm"] for x in dv])
    return views, in_deg, out_deg, nnorm


# column of a member's packed buffer (see Snapshot.device_views): 27 view arrays, then rel_rank, in_deg, out_deg, nnorm bits
_COL = {(vn, an): i * 9 + j for i, vn in enumerate(("by_dst", "by_src", "by_rel"))
        for j, an in enumerate(("a", "b", "chunk_seg", "chunk_beg", "chunk_end", "chunk_slot", "fix_seg", "fix_slot", "fix_cnt"))}
_COL.update(rel_rank=27, in_deg=28, out_deg=29, nnorm=30)


def union_graph_packed(snap, n_rel_rows, device):
    """The union's TempGraph arrays in ONE packed int32 device buffer, assembled from the members' resident views by a
    single temp_assemble_views launch (the host stacks the members' cached meta rows; the host planner library turns them
    into the descriptor table, temp_host_union_plan).  Returns None when the by-relation view needs the global sort.
    -> (ints, offs {(view, array) | 'in_deg' | 'out_deg' | 'nnorm': offset}, sizes {...}, counts {view: {...}})"""
    from .backend import get_backend
    E = int(snap.edge_off[-1])
    if E // (REL_GROUP_EDGES * max(n_rel_rows, 1)) <= 1:
        return None
    key = ("pack_meta", device.type, device.index, int(n_rel_rows))
    metas = []
    for g in snap.parts:                                              # hot loop: one dict probe per member
        m = g._dev.get(key)
        if m is None:
            m = g._dev[key] = g.device_views(device, n_rel_rows)["_meta"]
        metas.append(m)
    from . import _hostlib
    rows = np.stack([m["row"] for m in metas])
    ctl, sm = _hostlib.union_plan(rows, snap.node_off[:-1], snap.edge_off[:-1], n_rel_rows, _lib.ASSEMBLE_PIECE)
    n_desc, npc, nfix, tail0 = (int(x) for x in sm[:4])
    out_base, totals = sm[4:32], sm[35:62]
    names = ["in_deg", "out_deg", "nnorm"] + [(vn, an) for vn in ("by_dst", "by_src") for an in _VIEW_ARRAYS] + \
            [("by_rel", an) for an in ("a", "b", "chunk_seg", "chunk_beg", "chunk_end", "chunk_slot")]
    table_words = len(metas) * n_rel_rows
    ctl_dev = _lib.to_device(ctl, device)
    nd = n_desc * 8
    d_desc, d_pd, d_ps = ctl_dev[:nd], ctl_dev[nd:nd + npc], ctl_dev[nd + npc:nd + 2 * npc]
    d_table = ctl_dev[nd + 2 * npc:nd + 2 * npc + table_words]
    d_fix = ctl_dev[nd + 2 * npc + table_words:]
    ints = torch.empty(tail0 + 3 * nfix + 1, dtype=torch.int32, device=device)
    if nfix:
        ints[tail0:tail0 + 3 * nfix] = d_fix
    get_backend().assemble_views(d_pd, d_ps, d_desc, d_table, ints)
    offs = {k: int(out_base[i]) for i, k in enumerate(names)}
    sizes = {k: int(totals[i]) for i, k in enumerate(names)}
    for j, an in enumerate(("fix_seg", "fix_slot", "fix_cnt")):
        offs[("by_rel", an)], sizes[("by_rel", an)] = tail0 + j * nfix, nfix
    counts = {}
    for i, vn in enumerate(("by_dst", "by_src")):
        counts[vn] = dict(n_seg=int(snap.n), n_edges=E, n_chunks=sizes[(vn, "chunk_seg")], n_partial=int(sm[65 + i]), n_fix=sizes[(vn, "fix_seg")])
    counts["by_rel"] = dict(n_seg=int(n_rel_rows), n_edges=E, n_chunks=sizes[("by_rel", "chunk_seg")], n_partial=int(sm[67]), n_fix=nfix)
    # member tables (include/temp_amd.h: TempMembers): every view of the union is member-major, so the edge kernels can take one
    # workgroup per (member, feature slice) with the member's rows staged in LDS
    M = len(metas)
    nck = rows[:, [_COL[("by_dst", "chunk_seg")], _COL[("by_src", "chunk_seg")], _COL[("by_rel", "chunk_seg")]]].T      # (3, M) chunks per member
    nfx = rows[:, [_COL[("by_dst", "fix_seg")], _COL[("by_src", "fix_seg")]]].T                                          # (2, M) fix-up entries per member
    tab = np.zeros((7, M + 1), dtype=np.int32)
    tab[0], tab[1] = snap.node_off, snap.edge_off
    tab[2:5, 1:] = np.cumsum(nck, axis=1)
    tab[5:7, 1:] = np.cumsum(nfx, axis=1)                     # (the node views' fix lists are member-major: TempMembers.fix_off)
    members = dict(n_members=M, max_nodes=int(max(snap.node_sizes)) if M else 0, max_edges=int(np.diff(snap.edge_off).max()) if M else 0,
                   max_chunks=[int(x) for x in nck.max(axis=1)] if M else [0, 0, 0], table=_lib.to_device(tab.reshape(-1), device))
    counts["_members"] = members
    return ints, offs, sizes, counts, ctl_dev


_VIEW_ARRAYS = ("a", "b", "chunk_seg", "chunk_beg", "chunk_end", "chunk_slot", "fix_seg", "fix_slot", "fix_cnt")
# the arrays of a snapshot's packed device buffer, in order (Snapshot.device_views)
_PACK_NAMES = [(vn, an) for vn in ("by_dst", "by_src", "by_rel") for an in _VIEW_ARRAYS] + ["rel_rank", "in_deg", "out_deg", "nnorm"]


def _pack_small_union(snap, n_rel_rows, device):
    """(ints, offs, sizes, counts, None) of a BatchedSnapshot from ONE temp_host_snapshot_pack call over its materialised edges."""
    from . import _hostlib
    E = snap.number_of_edges()
    if E and (snap.rel.min() < 0 or snap.rel.max() >= n_rel_rows):
        raise ValueError("relation id outside [0, %d)" % n_rel_rows)
    packed, sz, n_partial, _ = _hostlib.snapshot_pack(snap.n, snap.src, snap.dst, snap.rel, snap.nnorm, n_rel_rows, _lib.CHUNK, _lib.CHUNK_REL)
    names = [(vn, an) for vn in ("by_dst", "by_src", "by_rel") for an in _VIEW_ARRAYS] + ["rel_rank", "in_deg", "out_deg", "nnorm"]
    off = np.concatenate([[0], np.cumsum(sz)])
    offs = {k: int(off[i]) for i, k in enumerate(names)}
    sizes = {k: int(sz[i]) for i, k in enumerate(names)}
    counts = {}
    for i, (vn, n_seg) in enumerate((("by_dst", snap.n), ("by_src", snap.n), ("by_rel", n_rel_rows))):
        counts[vn] = dict(n_seg=int(n_seg), n_edges=int(E), n_chunks=sizes[(vn, "chunk_seg")], n_partial=int(n_partial[i]), n_fix=sizes[(vn, "fix_seg")])
    ints = _lib.to_device(np.concatenate([packed, np.zeros(1, np.int32)]), device)
    return ints, offs, sizes, counts, None


class _DeviceGraph:
    """Device-resident TempGraph: one packed int32 buffer + nnorm, and the ctypes struct whose
    pointers refer into them (kept alive by this object)."""

    def __init__(self, snap, device, n_rel_rows):
        n, E = snap.n, snap.number_of_edges()
        device = torch.device(device)
        if DEVICE_STORE and device.type == "cuda" and isinstance(snap, BatchedSnapshot) and len(snap.parts) > 1:
            packed = union_graph_packed(snap, n_rel_rows, device) if DEVICE_STORE == "kernel" else None
            if packed is not None:
                self._init_from_packed(packed, n, E, n_rel_rows, device)
                return
            dev_union = union_views_device(snap, n_rel_rows, device) if DEVICE_STORE != "kernel" else None
            if dev_union is not None:
                self._init_from_device(dev_union, n, E, n_rel_rows, device)
                return
            if DEVICE_STORE == "kernel" and n > 0:
                # a small union (few edges per relation: ICEWS-like positions): one pass of the host planner over the union's
                # own edge list -- the by-relation view is then the global sort this shape wants
                self._init_from_packed(_pack_small_union(snap, n_rel_rows, device), n, E, n_rel_rows, device)
                return
        one = snap.parts[0] if isinstance(snap, BatchedSnapshot) and len(snap.parts) == 1 else snap
        if isinstance(one, SubsampledSnapshot):                 # its views exist on the device only: wrap its packed buffer
            dv = one.device_views(device, n_rel_rows)
            m = dv["_meta"]
            offs = {k: int(o) for k, o in zip(_PACK_NAMES, m["off"])}
            sizes = {k: int(z) for k, z in zip(_PACK_NAMES, m["size"])}
            counts = {}
            for i, (vn, n_seg) in enumerate((("by_dst", n), ("by_src", n), ("by_rel", n_rel_rows))):
                counts[vn] = dict(n_seg=int(n_seg), n_edges=int(one.parent.number_of_edges()), n_chunks=sizes[(vn, "chunk_seg")],
                                  n_partial=int(m["n_partial"][i]), n_fix=sizes[(vn, "fix_seg")])
            self._init_from_packed((dv["_buf"], offs, sizes, counts, None), n, E, n_rel_rows, device)
            return
        if isinstance(snap, BatchedSnapshot) and len(snap.parts) > 1:
            views, in_deg, out_deg = union_views(snap, n_rel_rows)
        else:
            one = snap.parts[0] if isinstance(snap, BatchedSnapshot) and snap.parts else snap
            lv = one.local_views(n_rel_rows) if n else None
            if lv is None:
                z = np.zeros(0, np.int64)
                views = dict(by_dst=build_view(z, z, z, 0), by_src=build_view(z, z, z, 0), by_rel=build_view(z, z, z, n_rel_rows, chunk=_lib.CHUNK_REL))
                in_deg = out_deg = np.zeros(0, np.int32)
            else:
                views = dict(by_dst=lv["by_dst"], by_src=lv["by_src"], by_rel=by_rel_view(one, n_rel_rows))
                in_deg, out_deg = lv["in_deg"], lv["out_deg"]
        parts, offs, off = [in_deg, out_deg], {}, 2 * n
        for vn, v in views.items():
            for an in _VIEW_ARRAYS:
                offs[(vn, an)] = off
                parts.append(v[an])
                off += v[an].shape[0]
        packed = np.concatenate(parts) if parts else np.zeros(0, np.int32)
        if packed.shape[0] == 0:
            packed = np.zeros(1, np.int32)
        self.ints = _lib.to_device(packed, device)
        self.nnorm = _lib.to_device(snap.nnorm if n else np.zeros(1, np.float32), device)
        self.in_deg = self.ints[0:n]
        self.out_deg = self.ints[n:2 * n]
        self.views = views
        self.n_nodes, self.n_edges, self.n_rel_rows = n, E, n_rel_rows
        self.device = self.ints.device
        base = self.ints.data_ptr()
        g = _lib.TempGraph()
        g.n_nodes, g.n_edges = n, E
        g.nnorm = self.nnorm.data_ptr()
        g.in_deg = base
        g.out_deg = base + 4 * n
        for vn, v in views.items():
            ev = getattr(g, vn)
            for fld in ("n_seg", "n_edges", "n_chunks", "n_partial", "n_fix"):
                setattr(ev, fld, v[fld])
            for an in _VIEW_ARRAYS:
                setattr(ev, an, base + 4 * offs[(vn, an)])
        self.c = g
        self.offs = offs

    def _init_from_device(self, dev_union, n, E, n_rel_rows, device):
        views, self.in_deg, self.out_deg, self.nnorm = dev_union
        self.views = views                          # arrays are DEVICE tensors here (host copies only exist on the CPU path)
        self.ints = None
        self.n_nodes, self.n_edges, self.n_rel_rows = n, E, n_rel_rows
        self.device = device
        g = _lib.TempGraph()
        g.n_nodes, g.n_edges = n, E
        g.nnorm = self.nnorm.data_ptr()
        g.in_deg = self.in_deg.data_ptr()
        g.out_deg = self.out_deg.data_ptr()
        for vn, v in views.items():
            ev = getattr(g, vn)
            for fld in ("n_seg", "n_edges", "n_chunks", "n_partial", "n_fix"):
                setattr(ev, fld, v[fld])
            for an in _VIEW_ARRAYS:
                t = v[an]
                assert t.dtype == torch.int32 and t.is_contiguous()
                setattr(ev, an, t.data_ptr() if t.numel() else 0)
        self.c = g
        self.offs = None

    def _init_from_packed(self, packed, n, E, n_rel_rows, device):
        self.ints, offs, sizes, counts, self._ctl = packed
        self.n_nodes, self.n_edges, self.n_rel_rows = n, E, n_rel_rows
        self.device = device
        self.in_deg = self.ints[offs["in_deg"]:offs["in_deg"] + n]
        self.out_deg = self.ints[offs["out_deg"]:offs["out_deg"] + n]
        self.nnorm = self.ints[offs["nnorm"]:offs["nnorm"] + n].view(torch.float32)
        base = self.ints.data_ptr()
        g = _lib.TempGraph()
        g.n_nodes, g.n_edges = n, E
        g.nnorm, g.in_deg, g.out_deg = base + 4 * offs["nnorm"], base + 4 * offs["in_deg"], base + 4 * offs["out_deg"]
        self.views = {}
        members = counts.pop("_members", None) if isinstance(counts, dict) else None
        if members is not None and members["n_members"] > 0:
            self._members = members                      # keeps the device table alive
            mb, M = g.members, members["n_members"]
            mb.n_members, mb.max_nodes, mb.max_edges = M, members["max_nodes"], members["max_edges"]
            for i in range(3):
                mb.max_chunks[i] = members["max_chunks"][i]
            tb = members["table"].data_ptr()
            mb.node_off, mb.edge_off, mb.chunk_off, mb.fix_off = tb, tb + 4 * (M + 1), tb + 8 * (M + 1), tb + 20 * (M + 1)
        for vn, cnt in counts.items():
            ev = getattr(g, vn)
            for fld, val in cnt.items():
                setattr(ev, fld, val)
            for an in _VIEW_ARRAYS:
                setattr(ev, an, base + 4 * offs[(vn, an)] if sizes[(vn, an)] else 0)
            self.views[vn] = dict(cnt)
        self.c = g
        self.offs = None
        self._packed = (offs, sizes)

    def view_tensor(self, view, name):
        """Device int32 tensor of one view array (used by tests and the CPU test backend)."""
        if getattr(self, "_packed", None) is not None:
            offs, sizes = self._packed
            return self.ints[offs[(view, name)]:offs[(view, name)] + sizes[(view, name)]]
        if self.offs is None:
            return self.views[view][name]
        o = self.offs[(view, name)]
        return self.ints[o:o + self.views[view][name].shape[0]]

    def ref(self):
        return ctypes.byref(self.c)


def prepack_parallel(snapshots, n_rel_rows, threads=8):
    """Snapshot.prepack of many (large) snapshots on a pool of threads: first use of an HBM-sized window otherwise packs its
    29 snapshots one after the other under the creation lock."""
    from concurrent.futures import ThreadPoolExecutor
    snaps = [g for g in snapshots if g is not None]
    if len(snaps) < 2 or threads < 2:
        for g in snaps:
            g.prepack(n_rel_rows)
        return
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(lambda g: g.prepack(n_rel_rows), snaps))
