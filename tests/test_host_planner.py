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
    assert names == sorted(_hostlib.SYMBOLS) and len(names) == 5
    for n in names:
        assert hasattr(lib, n)
    assert lib.temp_host_abi_version() == 1


@pytest.mark.parametrize("E,n_seg,chunk,hub", [(0, 5, 64, False), (1, 1, 64, False), (7475, 500, 64, True), (3737, 40, 128, False),
                                              (1000, 3, 7, True), (50, 5000, 64, False), (4096, 1, 64, False)])
def test_build_view_matches_numpy(E, n_seg, chunk, hub):
    rng = np.random.default_rng(E + n_seg)
    seg = rng.integers(0, n_seg, E)
    if hub and E:
        seg[: E // 2] = rng.integers(0, min(2, n_seg), E // 2)          # segments spanning many chunks
    a, b = rng.integers(0, 1000, E), rng.integers(0, 77, E)
    got = _hostlib.build_view(seg, a, b, n_seg, chunk)
    want = build_view_numpy(seg, a, b, n_seg, chunk)
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
