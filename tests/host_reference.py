"""TEST-ONLY numpy formulations of the host planner's C++ passes (temp_amd/csrc/host_planner.cpp), kept as the checker."""
import numpy as np


def build_view_numpy(seg, a, b, n_seg, chunk):
    """Sort edges by `seg` (stable) and cut every segment into chunks of <= `chunk` edges.
    Returns a dict of int32 numpy arrays + counts (layout of TempEdgeView)."""
    seg = np.asarray(seg, dtype=np.int64)
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
