"""TEST-ONLY numpy formulations of the host planner's C++ passes (temp_amd/csrc/host_planner.cpp), kept as the checker."""
import numpy as np


def build_view_numpy(seg, a, b, n_seg, chunk, sort_b=False):
    """Sort edges by `seg` (stable; sort_b: by (seg, b), ties in input order) and cut every segment into chunks of <= `chunk` edges.
    Returns a dict of int32 numpy arrays + counts (layout of TempEdgeView)."""
    seg = np.asarray(seg, dtype=np.int64)
    if sort_b:
        order = np.lexsort((np.asarray(b, dtype=np.int64), seg))
    else:
        order = np.argsort(seg.astype(np.uint16) if n_seg <= 65536 else seg, kind="stable")     # 16-bit keys: numpy radix-sorts them
    seg_s = seg[order]
    counts = np.bincount(seg_s, minlength=n_seg).astype(np.int64)
    ptr = np.concatenate([[0], np.cumsum(counts)])
    nch = (counts + chunk - 1) // chunk
    total = int(nch.sum())
    chunk_seg = np.repeat(np.arange(n_seg, dtype=np.int64), nch)
    first = np.cumsum(nch) - nch
    k = np.arange(total, dtype=np.int64) - first[chunk_seg]
    chunk_beg = ptr[chunk_seg] + k * chunk
    chunk_end = np.minimum(chunk_beg + chunk, ptr[chunk_seg + 1])
    multi = nch > 1
    is_multi = multi[chunk_seg]
    slot = np.where(is_multi, np.cumsum(is_multi) - 1, -1)
    fix_seg = np.nonzero(multi)[0]
    fix_cnt = nch[fix_seg]
    fix_slot = np.cumsum(fix_cnt) - fix_cnt
    i32 = lambda x: np.ascontiguousarray(x, dtype=np.int32)
    return dict(n_seg=int(n_seg), n_edges=int(seg.shape[0]), a=i32(np.asarray(a)[order]), b=i32(np.asarray(b)[order]),
                n_chunks=total, chunk_seg=i32(chunk_seg), chunk_beg=i32(chunk_beg), chunk_end=i32(chunk_end),
                chunk_slot=i32(slot), n_partial=int(is_multi.sum()), n_fix=int(fix_seg.shape[0]),
                fix_seg=i32(fix_seg), fix_slot=i32(fix_slot), fix_cnt=i32(fix_cnt), order=order)




def chain_plan_numpy(bsz, num_ents, positions, n_win, gid_arrays):
    """Row maps of a window chain, position by position (get_prev_embeddings / update_time_diff_hist_embeddings semantics,
    models/DynamicRGCN.py:35-54: the history holds ONLY the previous executed step's nodes)."""
    row_of = np.full((bsz, num_ents), -1, dtype=np.int64)
    last = np.zeros((bsz, num_ents), dtype=np.float32)
    prev_pairs = None
    pidx, nidx, dts = [], [], []
    for p, nw, arrs in zip(positions, n_win, gid_arrays):
        sizes = [len(g) for g in arrs]
        ids = np.concatenate(arrs) if arrs else np.zeros(0, np.int64)
        bb = np.repeat(np.arange(nw, dtype=np.int64), sizes)
        pidx.append(row_of[bb, ids].astype(np.int32))
        if nidx:                                               # inverse map of this step's prev_idx = the previous step's next_idx
            ok = pidx[-1] >= 0
            nidx[-1][pidx[-1][ok]] = np.nonzero(ok)[0].astype(np.int32)
        nidx.append(np.full(ids.shape[0], -1, dtype=np.int32))
        dts.append((p - last[bb, ids]).astype(np.float32))
        if prev_pairs is not None:
            row_of[prev_pairs] = -1
        row_of[bb, ids] = np.arange(ids.shape[0], dtype=np.int64)
        last[bb, ids] = p
        prev_pairs = (bb, ids)
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
    return cat(pidx, np.int32), cat(nidx, np.int32), cat(dts, np.float32), row_of, last


def union_plan_numpy(size, moff, ptr, n_part, counts_rel, node_off, edge_off, n_rel_rows, piece):
    """Control block of temp_assemble_views, numpy formulation: -> (ctl int32, summary dict)."""
    M = size.shape[0]
    COL = {(vn, an): i * 9 + j for i, vn in enumerate(("by_dst", "by_src", "by_rel"))
           for j, an in enumerate(("a", "b", "chunk_seg", "chunk_beg", "chunk_end", "chunk_slot", "fix_seg", "fix_slot", "fix_cnt"))}
    COL.update(rel_rank=27, in_deg=28, out_deg=29, nnorm=30)
    DESC = np.dtype([("src", np.int64), ("aux", np.int64), ("dst_off", np.int32), ("len", np.int32), ("add", np.int32), ("mode", np.int32)])
    zero = np.zeros(M, dtype=np.int64)
    p_off = {vn: np.concatenate([[0], np.cumsum(n_part[:, i])])[:-1] for i, vn in enumerate(("by_dst", "by_src"))}
    per_rel = counts_rel.sum(axis=0)
    multi = per_rel > 1
    fix_seg = np.nonzero(multi)[0]
    fix_cnt = per_rel[fix_seg]
    fix_slot = np.cumsum(fix_cnt) - fix_cnt
    base = np.full(n_rel_rows, -1, dtype=np.int64)
    base[fix_seg] = fix_slot
    table = np.where(multi[None, :], base[None, :] + (np.cumsum(counts_rel, axis=0) - counts_rel), -1).reshape(-1)
    spec = [("in_deg", COL["in_deg"], zero, 0, -1), ("out_deg", COL["out_deg"], zero, 0, -1), ("nnorm", COL["nnorm"], zero, 0, -1)]
    for vn in ("by_dst", "by_src"):
        for an, add, mode in (("a", node_off, 0), ("b", zero, 0), ("chunk_seg", node_off, 0), ("chunk_beg", edge_off, 0), ("chunk_end", edge_off, 0),
                              ("chunk_slot", p_off[vn], 1), ("fix_seg", node_off, 0), ("fix_slot", p_off[vn], 0), ("fix_cnt", zero, 0)):
            spec.append(((vn, an), COL[(vn, an)], add, mode, -1))
    for an, add in (("a", node_off), ("b", node_off), ("chunk_seg", zero), ("chunk_beg", edge_off), ("chunk_end", edge_off)):
        spec.append((("by_rel", an), COL[("by_rel", an)], add, 0, -1))
    spec.append((("by_rel", "chunk_slot"), COL["rel_rank"], np.arange(M, dtype=np.int64) * n_rel_rows, 2, COL[("by_rel", "chunk_seg")]))
    cols = np.array([c for _, c, _, _, _ in spec])
    lens = size[:, cols].T
    totals = lens.sum(axis=1)
    out_base = np.concatenate([[0], np.cumsum(totals)])
    dst = out_base[:-1, None] + np.cumsum(lens, axis=1) - lens
    desc = np.zeros(lens.shape, dtype=DESC)
    desc["src"] = ptr[None, :] + 4 * moff[:, cols].T
    aux_cols = np.array([max(a, 0) for *_, a in spec])
    desc["aux"] = ptr[None, :] + 4 * moff[:, aux_cols].T
    desc["dst_off"], desc["len"] = dst, lens
    desc["add"] = np.stack([a for _, _, a, _, _ in spec])
    desc["mode"] = np.array([m for _, _, _, m, _ in spec])[:, None]
    desc = desc.reshape(-1)
    desc = desc[desc["len"] > 0]
    n_p = (desc["len"].astype(np.int64) + piece - 1) // piece
    piece_desc = np.repeat(np.arange(desc.shape[0], dtype=np.int64), n_p)
    first = np.cumsum(n_p) - n_p
    piece_start = (np.arange(piece_desc.shape[0], dtype=np.int64) - first[piece_desc]) * piece
    ctl = np.concatenate([desc.view(np.int32), piece_desc.astype(np.int32), piece_start.astype(np.int32), table.astype(np.int32),
                          fix_seg.astype(np.int32), fix_slot.astype(np.int32), fix_cnt.astype(np.int32)])
    return ctl, dict(n_desc=desc.shape[0], n_pieces=piece_desc.shape[0], n_fix=fix_seg.shape[0], out_base=out_base, totals=totals,
                     partial=(int(n_part[:, 0].sum()), int(n_part[:, 1].sum()), int(per_rel[multi].sum())))
