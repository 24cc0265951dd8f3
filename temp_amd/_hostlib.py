"""ctypes binding of libtemp_host.so (include/temp_amd_host.h): the host-side planner pieces that were interpreter-bound
loops.  Plain C++ (g++), no GPU: built on first use when the in-tree library is missing or older than its source."""
import ctypes
import os
import subprocess

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
SRC = os.path.join(PKG, "csrc", "host_planner.cpp")
HDR = os.path.join(REPO, "include", "temp_amd_host.h")
LIB_PATH = os.path.join(PKG, "libtemp_host.so")
_lib = None

_P = ctypes.c_void_p
_I64 = ctypes.c_int64
SYMBOLS = {
    "temp_host_abi_version": (ctypes.c_int, []),
    "temp_host_build_view": (ctypes.c_int, [_I64, _P, _P, _P, _I64, _I64, ctypes.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "temp_host_plan_loss": (ctypes.c_int, [ctypes.c_int, _P, _P, _P, _P, ctypes.c_int, _I64, _P, _P, _P]),
    "temp_host_snapshot_pack": (_I64, [_I64, _I64, _P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P, _P]),
    "temp_host_union_plan": (_I64, [_I64, _I64, _P, _P, _P, _I64, _P, _I64, _P]),
    "temp_host_sample_subset": (ctypes.c_int, [_I64, _I64, ctypes.c_uint64, _P]),
    "temp_host_gather_inverse": (_I64, [_I64, _P, _I64, _P, _P]),
    "temp_host_unique_labels": (_I64, [_I64, _P, _I64, _P, _P]),
    "temp_host_chain_plan": (ctypes.c_int, [ctypes.c_int, _I64, ctypes.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "temp_host_chain_tracks": (ctypes.c_int, [ctypes.c_int, _P, _P, _P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, _P, _P, _P, _P, _P]),
}


def build(force=False, verbose=False):
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(d) > os.path.getmtime(LIB_PATH) for d in (SRC, HDR))
    if force or stale:
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(REPO, "include"), "-o", LIB_PATH, SRC]
        if verbose:
            print("[temp_amd.build] " + " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.temp_host_abi_version() != 2:
            raise RuntimeError("libtemp_host.so ABI version mismatch")
        from ._lib import outside_token

        class _Calls:                                 # the same entry points; a prefetch worker's planning token is handed back around each call
            pass
        calls = _Calls()
        for name in SYMBOLS:
            setattr(calls, name, outside_token(getattr(lib, name)) if name != "temp_host_abi_version" else getattr(lib, name))
        _lib = calls
    return _lib


def _i64(x):
    return np.ascontiguousarray(x, dtype=np.int64)


def build_view(seg, a, b, n_seg, chunk, sort_b=False):
    """Sorted / chunked edge view of one snapshot (see temp_host_build_view) -> dict in the layout snapshot.py uses."""
    seg, a, b = _i64(seg), _i64(a), _i64(b)
    E = int(seg.shape[0])
    cap = max(E, 1)
    order = np.empty(cap, np.int64)
    out = {k: np.empty(cap, np.int32) for k in ("a", "b", "chunk_seg", "chunk_beg", "chunk_end", "chunk_slot", "fix_seg", "fix_slot", "fix_cnt")}
    counts = np.zeros(3, np.int64)
    rc = load().temp_host_build_view(E, seg.ctypes.data, a.ctypes.data, b.ctypes.data, int(n_seg), int(chunk), int(bool(sort_b)), order.ctypes.data,
                                     out["a"].ctypes.data, out["b"].ctypes.data, out["chunk_seg"].ctypes.data, out["chunk_beg"].ctypes.data,
                                     out["chunk_end"].ctypes.data, out["chunk_slot"].ctypes.data, out["fix_seg"].ctypes.data,
                                     out["fix_slot"].ctypes.data, out["fix_cnt"].ctypes.data, counts.ctypes.data)
    if rc != 0:
        raise ValueError("temp_host_build_view: bad argument (code %d): segment id outside [0, %d)?" % (rc, n_seg))
    nch, npart, nfix = (int(c) for c in counts)
    res = dict(n_seg=int(n_seg), n_edges=E, a=out["a"][:E], b=out["b"][:E], n_chunks=nch, n_partial=npart, n_fix=nfix, order=order[:E])
    for k in ("chunk_seg", "chunk_beg", "chunk_end", "chunk_slot"):
        res[k] = out[k][:nch]
    for k in ("fix_seg", "fix_slot", "fix_cnt"):
        res[k] = out[k][:nfix]
    return res


# data addresses of long-lived arrays (a snapshot's gids are looked up ~250 times per window batch; `.ctypes` builds a helper
# object per access)
_ADDR = {}


def chain_plan(bsz, num_ents, positions, n_win, gid_arrays):
    """Row maps of one window chain (see temp_host_chain_plan).  gid_arrays[s][j] = int64 gids of window j at executed step s.
    -> (prev_idx int32 [total], next_idx int32 [total], dt [total], row_of [bsz, num_ents], last [bsz, num_ents])"""
    n_steps = len(positions)
    ptrs = np.zeros(max(n_steps * bsz, 1), np.int64)
    lens = np.zeros(max(n_steps * bsz, 1), np.int64)
    for s, arrs in enumerate(gid_arrays):
        for j, g in enumerate(arrs):
            a = _ADDR.get(id(g))                     # (array, address): the entry keeps the array alive, so its id stays its own
            if a is None or a[0] is not g:
                if len(_ADDR) > 65536:
                    _ADDR.clear()
                a = _ADDR[id(g)] = (g, g.ctypes.data)
            ptrs[s * bsz + j] = a[1]
            lens[s * bsz + j] = g.shape[0]
    total = int(lens.sum())
    prev_idx = np.empty(max(total, 1), np.int32)
    next_idx = np.empty(max(total, 1), np.int32)
    dt = np.empty(max(total, 1), np.float32)
    row_of = np.empty((bsz, num_ents), np.int64)
    last = np.empty((bsz, num_ents), np.float32)
    pos = np.ascontiguousarray(positions, dtype=np.int32)
    nw = np.ascontiguousarray(n_win, dtype=np.int32)
    rc = load().temp_host_chain_plan(int(bsz), int(num_ents), n_steps, pos.ctypes.data, nw.ctypes.data, ptrs.ctypes.data, lens.ctypes.data,
                                     prev_idx.ctypes.data, next_idx.ctypes.data, dt.ctypes.data, row_of.ctypes.data, last.ctypes.data)
    if rc != 0:
        raise ValueError("temp_host_chain_plan: bad argument (code %d)" % rc)
    return prev_idx[:total], next_idx[:total], dt[:total], row_of, last


def plan_loss(graph_ptrs, idx_list, row_offsets, pad4=True):
    """See temp_host_plan_loss.  graph_ptrs: (G, 8) int64 addresses; idx_list: G int64 arrays of chosen edge ids.
    -> (packed int32 [6, R], weights float32 [R], triples int64 [sum P, 3], n_pos, rows per graph incl. padding)"""
    G = len(idx_list)
    idx_list = [_i64(x) for x in idx_list]
    n_pos = np.array([x.shape[0] for x in idx_list], dtype=np.int64)
    block = 2 * n_pos + (((4 - (2 * n_pos) % 4) % 4) if pad4 else 0) * (n_pos > 0)
    R = int(block.sum())
    packed = np.empty((6, max(R, 1)), np.int32)
    weights = np.empty(max(R, 1), np.float32)
    triples = np.empty((max(int(n_pos.sum()), 1), 3), np.int64)
    ptrs = np.array([x.ctypes.data for x in idx_list], dtype=np.int64) if G else np.zeros(1, np.int64)
    gp = np.ascontiguousarray(graph_ptrs, dtype=np.int64) if G else np.zeros((1, 8), np.int64)
    ro = _i64(row_offsets) if G else np.zeros(1, np.int64)
    rc = load().temp_host_plan_loss(G, gp.ctypes.data, ptrs.ctypes.data, n_pos.ctypes.data, ro.ctypes.data, 1 if pad4 else 0, R,
                                    packed.ctypes.data, weights.ctypes.data, triples.ctypes.data)
    if rc != 0:
        raise ValueError("temp_host_plan_loss: bad argument (code %d)" % rc)
    return packed[:, :R], weights[:R], triples[:int(n_pos.sum())], n_pos, block


def gather_inverse(idx, n_rows):
    """See temp_host_gather_inverse -> one int32 array [seg_ptr (n_rows + 1) | order (count)] and the count."""
    idx = _i64(idx).reshape(-1)
    n = int(idx.shape[0])
    both = np.empty(n_rows + 1 + max(n, 1), np.int32)
    cnt = load().temp_host_gather_inverse(n, idx.ctypes.data, int(n_rows), both.ctypes.data, both[n_rows + 1:].ctypes.data)
    if cnt < 0:
        raise ValueError("temp_host_gather_inverse: index outside [.., %d)" % n_rows)
    return both[:n_rows + 1 + cnt], int(cnt)


def unique_labels(labels, n_labels):
    """See temp_host_unique_labels -> (first int32 [k], inverse int32 [n]): numpy.unique(labels, return_index=True,
    return_inverse=True)[1:] for labels in [0, n_labels)."""
    labels = _i64(labels).reshape(-1)
    n = int(labels.shape[0])
    first, inv = np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.int32)
    k = load().temp_host_unique_labels(n, labels.ctypes.data, int(n_labels), first.ctypes.data, inv.ctypes.data)
    if k < 0:
        raise ValueError("temp_host_unique_labels: label outside [0, %d)" % n_labels)
    return first[:k], inv[:n]


def snapshot_pack(n, src, dst, rel, nnorm, n_rel_rows, chunk, chunk_rel):
    """See temp_host_snapshot_pack -> (packed int32, sizes int64[31], n_partial int64[3], rel_chunks int64[n_rel_rows])."""
    src, dst, rel = _i64(src), _i64(dst), _i64(rel)
    nnorm = np.ascontiguousarray(nnorm, dtype=np.float32)
    E = int(src.shape[0])
    packed = np.empty(28 * max(E, 1) + 3 * int(n) + 1, np.int32)
    sizes, n_partial, rel_chunks = np.zeros(31, np.int64), np.zeros(3, np.int64), np.zeros(max(int(n_rel_rows), 1), np.int64)
    w = load().temp_host_snapshot_pack(int(n), E, src.ctypes.data, dst.ctypes.data, rel.ctypes.data, nnorm.ctypes.data, int(n_rel_rows),
                                       int(chunk), int(chunk_rel), packed.ctypes.data, sizes.ctypes.data, n_partial.ctypes.data, rel_chunks.ctypes.data)
    if w < 0:
        raise ValueError("temp_host_snapshot_pack: bad argument (node or relation id out of range?)")
    return packed[:w], sizes, n_partial, rel_chunks[:n_rel_rows]


def sample_subset(n, k, rng):
    """k distinct integers of [0, n) (see temp_host_sample_subset); the seed is one draw of the numpy generator `rng`."""
    out = np.empty(max(int(k), 1), np.int64)
    rc = load().temp_host_sample_subset(int(n), int(k), int(rng.integers(0, 1 << 63)), out.ctypes.data)
    if rc != 0:
        raise ValueError("temp_host_sample_subset: need 0 <= k <= n")
    return out[:int(k)]


def union_plan(meta, node_off, edge_off, n_rel_rows, piece):
    """See temp_host_union_plan.  meta: (M, 66 + n_rel_rows) int64 -> (ctl int32, summary int64[72])."""
    meta = np.ascontiguousarray(meta, dtype=np.int64)
    M = int(meta.shape[0])
    node_off, edge_off = _i64(node_off), _i64(edge_off)
    words = int(meta[:, :31].sum()) if M else 0
    cap = 8 * 27 * M + 2 * (27 * M + words // int(piece) + 64) + M * int(n_rel_rows) + 3 * int(n_rel_rows) + 16
    ctl = np.empty(cap, np.int32)
    summary = np.zeros(72, np.int64)
    w = load().temp_host_union_plan(M, int(n_rel_rows), meta.ctypes.data, node_off.ctypes.data, edge_off.ctypes.data, int(piece),
                                    ctl.ctypes.data, cap, summary.ctypes.data)
    if w < 0:
        raise ValueError("temp_host_union_plan: bad argument")
    return ctl[:w], summary


def chain_tracks(chains, inst_n, inst_h0, inst_rnn, prev_idx, tracks, max_steps, reuse_sizing_pass=True):
    """Track / panel tables of the persistent chain kernels (temp_host_chain_tracks).  chains = lists of instance ids in position
    order; prev_idx[i] = int array (row of the previous instance or -1) or None.
    -> None when the chains cannot run on the chain kernels, else (panel int32 [P,4], rows int32 [S,tracks], any_prev bool [S],
       step_inst int64 [S])."""
    n_inst = len(inst_n)
    chain_off = np.zeros(len(chains) + 1, np.int64)
    for c, ch in enumerate(chains):
        chain_off[c + 1] = chain_off[c] + len(ch)
    chain_inst = np.asarray([i for ch in chains for i in ch], dtype=np.int64) if chains else np.zeros(1, np.int64)
    n = np.ascontiguousarray(inst_n, dtype=np.int64)
    h0 = np.ascontiguousarray(inst_h0, dtype=np.int64)
    rnn = np.ascontiguousarray(inst_rnn, dtype=np.int64)
    prev_off = np.zeros(n_inst + 1, np.int64)
    parts = []
    for i in range(n_inst):
        p = prev_idx[i]
        if p is None or len(p) != n[i]:
            p = np.full(int(n[i]), -1, np.int32)
        parts.append(np.asarray(p, dtype=np.int32))
        prev_off[i + 1] = prev_off[i] + n[i]
    prev_cat = np.ascontiguousarray(np.concatenate(parts)) if parts and prev_off[-1] > 0 else np.zeros(1, np.int32)
    counts = np.zeros(2, np.int64)
    lib = load()
    args = (len(chains), chain_off.ctypes.data, chain_inst.ctypes.data, n.ctypes.data, h0.ctypes.data, rnn.ctypes.data, prev_off.ctypes.data,
            prev_cat.ctypes.data, int(tracks), int(max_steps), counts.ctypes.data)
    rc = lib.temp_host_chain_tracks(*args, None, None, None, None)
    if rc == 1:
        return None
    if rc != 0:
        raise ValueError("temp_host_chain_tracks: bad argument (code %d)" % rc)
    P, S = int(counts[0]), int(counts[1])
    if P == 0:
        return None
    panel = np.empty((P, 4), np.int32)
    rows = np.empty((S, tracks), np.int32)
    anyp = np.empty(S, np.uint8)
    sinst = np.empty(S, np.int64)
    if not reuse_sizing_pass:                    # other argument pointers: the library plans again instead of copying pass 1's tables
        counts2 = np.zeros(2, np.int64)
        args = args[:-1] + (counts2.ctypes.data,)
    rc = lib.temp_host_chain_tracks(*args, panel.ctypes.data, rows.ctypes.data, anyp.ctypes.data, sinst.ctypes.data)
    if rc != 0:
        raise ValueError("temp_host_chain_tracks: code %d on the fill pass" % rc)
    return panel, rows, anyp.astype(bool), sinst
