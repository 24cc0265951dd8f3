"""Flat on-disk snapshot store: one memory-mappable file instead of the reference's pickled
`{t: DGLGraph}` dictionaries (utils/dataset.py:268-305), holding for every timestamp and split

  * the edge list in its ORIGINAL order (src, dst, rel: int32, local node ids) -- the order `edge_subgraph` draws and
    evaluate()'s rank output refer to,
  * the node id list (shared by the three splits of a timestamp, utils/dataset.py:151-232),
  * the snapshot's PACKED VIEWS: the dst- / src- / rel-sorted chunked edge views, degrees and 1/in_degree norms exactly as
    the kernels consume them (Snapshot.device_views), precomputed by the host planner library at write time.

Opening a store costs one mmap; nothing is parsed or sorted at run time.  `to_device()` makes the whole training split
resident with ONE host-to-device copy (GDELT-shaped: 366 snapshots x 0.3 MB), after which a batch's union views are
assembled on the GPU (temp_assemble_views) and the training-time edge subsets are drawn there too (temp_subsample_views).

File layout (little endian), all offsets in the header are in BYTES from the start of the file:
    magic "TSNAPST1" | int64 header[16] = {version (= VERSION below), num_ents, num_rels, T, n_rel_rows, chunk, chunk_rel, n_sections, 0...}
    int64 section table [n_sections][2] = {offset, length in elements}; sections in the order of `_SECTIONS` below.
"""
import os

import numpy as np
import torch

from . import _hostlib
from . import _lib
from .snapshot import Snapshot, _PACK_NAMES

MAGIC = b"TSNAPST1"
# Layout version of the PACKED VIEWS inside the file.  2 (round 4): inside a node's segment the by-destination / by-source views list
# the edges in relation order and a relation's edges are listed in destination order (snapshot.py, host planner `sort_b`); the
# device edge-id tables (Snapshot.device_edge_ids) assume that order, so a version-1 file -- views in plain segment order -- would
# make temp_subsample_views keep DIFFERENT edge sets in the three views without any error.  Such files are refused; rewrite them
# with write_store() (the original-order edge lists they hold are unchanged).
VERSION = 2
SPLITS = ("train", "valid", "test")
# name -> dtype; per-split sections are prefixed with the split name
_SECTIONS = [("times", np.int64), ("node_ptr", np.int64), ("gids", np.int64)] + \
            [(s + "/" + k, dt) for s in SPLITS for k, dt in (("edge_ptr", np.int64), ("src", np.int32), ("dst", np.int32), ("rel", np.int32),
                                                              ("pack_ptr", np.int64), ("pack_sizes", np.int64), ("pack_partial", np.int64),
                                                              ("rel_chunks", np.int64), ("packs", np.int32))]


def write_store(path, graph_dict_train, graph_dict_val, graph_dict_test, num_ents, num_rels):
    """Serialise the three {time: Snapshot} dictionaries (same keys, shared node sets) into one flat file."""
    dicts = (graph_dict_train, graph_dict_val, graph_dict_test)
    times = [int(t) for t in graph_dict_train.keys()]
    n_rel_rows = 2 * int(num_rels)
    T = len(times)
    sec = {"times": np.asarray(times, np.int64)}
    sec["node_ptr"] = np.concatenate([[0], np.cumsum([graph_dict_train[t].n for t in times])]).astype(np.int64)
    sec["gids"] = np.concatenate([graph_dict_train[t].gids for t in times]).astype(np.int64) if T else np.zeros(0, np.int64)
    for s, gd in zip(SPLITS, dicts):
        gs = [gd[t] for t in times]
        for g, gt in zip(gs, (graph_dict_train[t] for t in times)):
            assert g.n == gt.n and np.array_equal(g.gids, gt.gids), "the splits of a timestamp share one node set"
        sec[s + "/edge_ptr"] = np.concatenate([[0], np.cumsum([g.number_of_edges() for g in gs])]).astype(np.int64)
        cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
        sec[s + "/src"], sec[s + "/dst"], sec[s + "/rel"] = (cat([getattr(g, k) for g in gs], np.int32) for k in ("src", "dst", "rel"))
        packs, sizes, partial, relch = [], [], [], []
        for g in gs:
            packed, sz, n_partial, rel_chunks = _hostlib.snapshot_pack(g.n, g.src, g.dst, g.rel, g.nnorm, n_rel_rows, _lib.CHUNK, _lib.CHUNK_REL)
            packs.append(np.asarray(packed, np.int32))
            sizes.append(sz)
            partial.append(n_partial)
            rc = np.zeros(n_rel_rows, np.int64)
            rc[:len(rel_chunks)] = rel_chunks
            relch.append(rc)
        sec[s + "/pack_ptr"] = np.concatenate([[0], np.cumsum([p.shape[0] for p in packs])]).astype(np.int64)
        sec[s + "/packs"] = cat(packs, np.int32)
        sec[s + "/pack_sizes"] = np.asarray(sizes, np.int64).reshape(T, len(_PACK_NAMES))
        sec[s + "/pack_partial"] = np.asarray(partial, np.int64).reshape(T, 3)
        sec[s + "/rel_chunks"] = np.asarray(relch, np.int64).reshape(T, n_rel_rows)
    header = np.zeros(16, np.int64)
    header[:8] = [VERSION, num_ents, num_rels, T, n_rel_rows, _lib.CHUNK, _lib.CHUNK_REL, len(_SECTIONS)]
    table = np.zeros((len(_SECTIONS), 2), np.int64)
    off = len(MAGIC) + header.nbytes + table.nbytes
    blobs = []
    for i, (name, dt) in enumerate(_SECTIONS):
        a = np.ascontiguousarray(sec[name], dtype=dt).reshape(-1)
        off = (off + 63) // 64 * 64                                   # 64-byte aligned sections
        table[i] = (off, a.shape[0])
        blobs.append((off, a))
        off += a.nbytes
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(header.tobytes())
        f.write(table.tobytes())
        for o, a in blobs:
            f.seek(o)
            f.write(a.tobytes())
    os.replace(tmp, path)
    return path


class StoredSnapshot(Snapshot):
    """A Snapshot whose arrays are views into the store's mmap and whose packed views come precomputed from the file."""

    def __init__(self, n, src, dst, rel, gids, nnorm, pack):
        self.n = int(n)
        self._src32, self._dst32, self._rel32 = src, dst, rel
        self._e64 = None
        self.gids = gids
        self.nnorm = nnorm
        self.ndata, self._dev, self._ids_dict, self._views = {}, {}, None, {}
        self._pack = pack                                             # (packed int32 view, sizes, n_partial, rel_chunks, n_rel_rows)

    def _edges(self):
        if self._e64 is None:                                          # the host code indexes with int64 arrays: widen once, on demand
            self._e64 = tuple(np.asarray(a, dtype=np.int64) for a in (self._src32, self._dst32, self._rel32))
        return self._e64

    src = property(lambda self: self._edges()[0])
    dst = property(lambda self: self._edges()[1])
    rel = property(lambda self: self._edges()[2])

    def number_of_edges(self):
        return int(self._src32.shape[0])

    def stored_pack(self, n_rel_rows):
        return self._pack[:4] if self._pack is not None and self._pack[4] == n_rel_rows else None


class SnapshotStore:
    def __init__(self, path):
        self.path = path
        self.mm = np.memmap(path, dtype=np.uint8, mode="r")
        assert bytes(self.mm[:8]) == MAGIC, "not a temp_amd snapshot store"
        header = np.frombuffer(self.mm, dtype=np.int64, count=16, offset=8)
        version, self.num_ents, self.num_rels, self.T, self.n_rel_rows, chunk, chunk_rel, n_sec = (int(x) for x in header[:8])
        if version != VERSION:
            raise ValueError("%s: snapshot store layout version %d, this library reads version %d (the order of the packed edge views "
                             "changed): rebuild the file with temp_amd.store.write_store()" % (path, version, VERSION))
        assert chunk == _lib.CHUNK and chunk_rel == _lib.CHUNK_REL and n_sec == len(_SECTIONS), "store written for another layout"
        table = np.frombuffer(self.mm, dtype=np.int64, count=2 * n_sec, offset=8 + 128).reshape(n_sec, 2)
        self.sec = {}
        for (name, dt), (off, cnt) in zip(_SECTIONS, table):
            self.sec[name] = np.frombuffer(self.mm, dtype=dt, count=int(cnt), offset=int(off))
        self.times = [int(t) for t in self.sec["times"]]
        self._dicts = None

    def _snapshot(self, split, i):
        s = self.sec
        n0, n1 = int(s["node_ptr"][i]), int(s["node_ptr"][i + 1])
        e0, e1 = int(s[split + "/edge_ptr"][i]), int(s[split + "/edge_ptr"][i + 1])
        p0, p1 = int(s[split + "/pack_ptr"][i]), int(s[split + "/pack_ptr"][i + 1])
        P = len(_PACK_NAMES)
        sizes = s[split + "/pack_sizes"][i * P:(i + 1) * P]
        packed = s[split + "/packs"][p0:p1]
        nn_off = int(sizes[:P - 1].sum())                              # nnorm is the last array of a pack (float bits)
        nnorm = packed[nn_off:nn_off + (n1 - n0)].view(np.float32)
        R = self.n_rel_rows
        pack = (packed, sizes, s[split + "/pack_partial"][3 * i:3 * i + 3], s[split + "/rel_chunks"][R * i:R * (i + 1)], R)
        return StoredSnapshot(n1 - n0, s[split + "/src"][e0:e1], s[split + "/dst"][e0:e1], s[split + "/rel"][e0:e1], s["gids"][n0:n1], nnorm, pack)

    def graph_dicts(self):
        """-> (graph_dict_train, graph_dict_val, graph_dict_test): {time: Snapshot}, the constructor arguments of the models."""
        if self._dicts is None:
            self._dicts = tuple({t: self._snapshot(sp, i) for i, t in enumerate(self.times)} for sp in SPLITS)
        return self._dicts

    def to_device(self, device, split="train"):
        """Make every snapshot of `split` resident on `device`: ONE upload of all packed views; each snapshot's device views
        become slices of that buffer."""
        gd = self.graph_dicts()[SPLITS.index(split)]
        big = torch.from_numpy(np.array(self.sec[split + "/packs"])).to(device)          # one host-to-device copy of all packed views
        ptr = self.sec[split + "/pack_ptr"]
        for i, t in enumerate(self.times):
            gd[t].adopt_device_pack(big[int(ptr[i]):int(ptr[i + 1])], device, self.n_rel_rows)
        return big
