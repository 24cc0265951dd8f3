"""Host planner library (include/temp_amd_host.h, plain C++): bit-exact against the numpy formulations in
tests/host_reference.py, and every declared symbol is exported."""
import os
import re

import numpy as np
import pytest

from temp_amd import _hostlib
from tests.host_reference import build_view_numpy, chain_plan_numpy

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "temp_amd_host.h")


def test_host_library_exports_every_declared_symbol():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(temp_host_[a-z0-9_]+)\s*\(", src)))
    lib = _hostlib.load()
    assert names == sorted(_hostlib.SYMBOLS) and len(names) == 10
    for n in names:
        assert hasattr(lib, n)
    assert lib.temp_host_abi_version() == 2


@pytest.mark.parametrize("E,n_seg,chunk,hub", [(0, 5, 64, False), (1, 1, 64, False), (7475, 500, 64, True), (3737, 40, 128, False),
                                              (1000, 3, 7, True), (50, 5000, 64, False), (4096, 1, 64, False)])
@pytest.mark.parametrize("sort_b", [False, True])
def test_build_view_matches_numpy(E, n_seg, chunk, hub, sort_b):
    rng = np.random.default_rng(E + n_seg)
    seg = rng.integers(0, n_seg, E)
    if hub and E:
        seg[: E // 2] = rng.integers(0, min(2, n_seg), E // 2)          # segments spanning many chunks
    a, b = rng.integers(0, 1000, E), rng.integers(0, 77, E)
    got = _hostlib.build_view(seg, a, b, n_seg, chunk, sort_b)
    want = build_view_numpy(seg, a, b, n_seg, chunk, sort_b)
    if sort_b and E:                                      # a segment's edges in ascending b, ties in input order
        s_sorted, b_sorted = seg[got["order"]], b[got["order"]]
        same = s_sorted[1:] == s_sorted[:-1]
        assert (b_sorted[1:][same] >= b_sorted[:-1][same]).all()
        tie = same & (b_sorted[1:] == b_sorted[:-1])
        assert (got["order"][1:][tie] > got["order"][:-1][tie]).all()
    assert set(got) == set(want)
    for k, v in want.items():
        if isinstance(v, np.ndarray):
            assert got[k].dtype == v.dtype and np.array_equal(got[k], v), k
        else:
            assert got[k] == v, k
    with pytest.raises(ValueError):
        _hostlib.build_view(np.array([n_seg]), np.array([0]), np.array([0]), n_seg, chunk)


@pytest.mark.parametrize("bsz,N,L,pad", [(1, 10, 2, 0), (4, 300, 9, 2), (8, 500, 15, 0), (3, 64, 6, 5)])
def test_chain_plan_matches_numpy(bsz, N, L, pad):
    """Random node sets per (position, window); the last `pad`-limited windows are left-padded (inactive at early positions)."""
    rng = np.random.default_rng(bsz * 100 + L)
    first_active = [0] * bsz
    for j in range(bsz):                                   # padded windows form a suffix of the batch
        first_active[j] = 0 if j < bsz - min(pad, bsz - 1) else int(rng.integers(1, L))
    first_active = sorted(first_active)
    positions, n_win, arrs = [], [], []
    for p in range(L - 1):
        nw = sum(1 for j in range(bsz) if first_active[j] <= p)
        if nw == 0:
            continue
        positions.append(p)
        n_win.append(nw)
        arrs.append([np.sort(rng.choice(N, size=int(rng.integers(1, N // 2 + 2)), replace=False)).astype(np.int64) for _ in range(nw)])
    got = _hostlib.chain_plan(bsz, N, positions, n_win, arrs)
    want = chain_plan_numpy(bsz, N, positions, n_win, arrs)
    for g, w in zip(got, want):
        assert g.dtype == w.dtype and np.array_equal(g, w)


@pytest.mark.parametrize("n,n_rows", [(0, 5), (1, 1), (48000, 40), (116000, 81000), (5000, 70000)])
def test_gather_inverse_matches_stable_argsort(n, n_rows):
    rng = np.random.default_rng(n + n_rows)
    idx = rng.integers(-1, n_rows, n)
    both, cnt = _hostlib.gather_inverse(idx, n_rows)
    keep = np.nonzero(idx >= 0)[0]
    assert cnt == keep.shape[0] and both.dtype == np.int32
    assert np.array_equal(both[n_rows + 1:], keep[np.argsort(idx[keep], kind="stable")])
    assert both[0] == 0 and np.array_equal(both[1:n_rows + 1], np.cumsum(np.bincount(idx[keep], minlength=n_rows)))
    with pytest.raises(ValueError):
        _hostlib.gather_inverse(np.array([n_rows]), n_rows)


@pytest.mark.parametrize("n,n_labels", [(0, 5), (1, 1), (1000, 300), (60000, 47000), (300, 90000)])
def test_unique_labels_matches_numpy_unique(n, n_labels):
    rng = np.random.default_rng(n + n_labels)
    labels = rng.integers(0, n_labels, n)
    first, inv = _hostlib.unique_labels(labels, n_labels)
    _, f2, i2 = np.unique(labels, return_index=True, return_inverse=True)
    assert first.dtype == np.int32 and inv.dtype == np.int32
    assert np.array_equal(first, f2) and np.array_equal(inv, i2.reshape(-1))
    with pytest.raises(ValueError):
        _hostlib.unique_labels(np.array([n_labels]), n_labels)


@pytest.mark.parametrize("n,E,R2,hub", [(500, 7475, 40, True), (227, 200, 460, False), (30, 0, 6, False), (1, 5, 2, False), (64, 3000, 3, True)])
def test_snapshot_pack_matches_view_by_view_pack(n, E, R2, hub):
    """temp_host_snapshot_pack (all three views + the device-store layout in one pass) == packing the three cached host views."""
    import torch
    from temp_amd.snapshot import Snapshot
    rng = np.random.default_rng(n + E)
    src, dst, rel = rng.integers(0, n, E), rng.integers(0, n, E), rng.integers(0, R2, E)
    if hub and E:
        dst[: E // 2] = rng.integers(0, 2, E // 2)
    gids = rng.permutation(10 * n)[:n]
    a, b = Snapshot(n, src, dst, rel, gids), Snapshot(n, src, dst, rel, gids)
    b.local_views(R2)                                            # b takes the view-by-view path
    da, db = a.device_views("cpu", R2), b.device_views("cpu", R2)
    assert a._views.get(R2) is None and b._views.get(R2) is not None
    assert torch.equal(da["_buf"], db["_buf"])
    for k in ("off", "size", "n_partial", "rel_chunks"):
        assert np.array_equal(da["_meta"][k], db["_meta"][k]), k
    for vn in ("by_dst", "by_src", "by_rel"):
        for an, t in db[vn].items():
            assert torch.equal(da[vn][an], t), (vn, an)
    assert torch.equal(da["nnorm"], db["nnorm"]) and torch.equal(da["rel_rank"], db["rel_rank"])


def test_sample_subset_is_a_uniform_subset():
    rng = np.random.default_rng(0)
    n, k = 200, 60
    hits = np.zeros(n)
    first = np.zeros(n)
    for _ in range(4000):
        s = _hostlib.sample_subset(n, k, rng)
        assert s.shape == (k,) and np.unique(s).shape[0] == k and s.min() >= 0 and s.max() < n
        hits[s] += 1
        first[s[0]] += 1
    exp = 4000 * k / n
    assert np.abs(hits - exp).max() < 6 * np.sqrt(exp)                       # every element equally likely to be chosen ...
    assert np.abs(first - 4000 / n).max() < 6 * np.sqrt(4000 / n) + 1         # ... and to come first
    assert _hostlib.sample_subset(5, 5, rng).tolist() != [] and sorted(_hostlib.sample_subset(5, 5, rng).tolist()) == [0, 1, 2, 3, 4]
    assert _hostlib.sample_subset(7, 0, rng).shape == (0,)
    with pytest.raises(ValueError):
        _hostlib.sample_subset(3, 4, rng)


@pytest.mark.parametrize("M,R,piece", [(1, 4, 4096), (6, 40, 4096), (183, 40, 4096), (9, 3, 16)])
def test_union_plan_matches_numpy(M, R, piece):
    """temp_host_union_plan (descriptor table / pieces / slot table / fix arrays of temp_assemble_views) vs the numpy formulation."""
    from tests.host_reference import union_plan_numpy
    rng = np.random.default_rng(M * 7 + R)
    size = rng.integers(0, 9000, (M, 31))
    size[rng.random((M, 31)) < 0.15] = 0                              # empty arrays (no hub segments, empty graphs)
    moff = np.cumsum(size, axis=1) - size
    ptr = (rng.integers(1, 1 << 20, M) << 8) + (1 << 40)              # 64-bit device addresses
    n_part = rng.integers(0, 50, (M, 3))
    counts_rel = rng.integers(0, 3, (M, R))
    node_off = np.cumsum(rng.integers(1, 600, M)) - 1
    edge_off = np.cumsum(rng.integers(0, 8000, M))
    meta = np.concatenate([size, moff, ptr[:, None], n_part, counts_rel], axis=1).astype(np.int64)
    ctl, sm = _hostlib.union_plan(meta, node_off, edge_off, R, piece)
    want, info = union_plan_numpy(size, moff, ptr, n_part, counts_rel, node_off, edge_off, R, piece)
    assert ctl.dtype == np.int32 and np.array_equal(ctl, want)
    assert (int(sm[0]), int(sm[1]), int(sm[2])) == (info["n_desc"], info["n_pieces"], info["n_fix"]) and int(sm[3]) == int(info["out_base"][-1])
    assert np.array_equal(sm[4:32], info["out_base"]) and np.array_equal(sm[35:62], info["totals"])
    assert tuple(int(x) for x in sm[65:68]) == info["partial"]


def test_prepacked_snapshot_views_equal_lazy_ones():
    """Snapshot.prepack (the packed views built ahead of first use, outside the creation lock; prepack_parallel for many) hands
    device_views the same buffer the lazy path builds."""
    import torch
    from temp_amd import snapshot as S
    rng = np.random.default_rng(9)
    snaps = []
    for k in range(5):
        n, E, R2 = 300 + 17 * k, 4000 + 100 * k, 40
        snaps.append((n, rng.integers(0, n, E), rng.integers(0, n, E), rng.integers(0, R2, E), rng.permutation(10 * n)[:n]))
    lazy = [S.Snapshot(*a) for a in snaps]
    early = [S.Snapshot(*a) for a in snaps]
    S.prepack_parallel(early, 40, threads=3)
    assert all(40 in g.__dict__.get("_pack", {}) for g in early)
    for a, b in zip(lazy, early):
        da, db = a.device_views("cpu", 40), b.device_views("cpu", 40)
        assert torch.equal(da["_buf"], db["_buf"]) and not b.__dict__["_pack"]
    early[0].prepack(40)                                              # already resident: nothing to do
    assert not early[0].__dict__["_pack"]
