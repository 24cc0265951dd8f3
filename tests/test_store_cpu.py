"""Snapshot store, host side: device-side edge subsample semantics through the test backend's reference of
temp_subsample_views, and the flat on-disk store."""
import numpy as np
import pytest
import torch

from temp_amd import backend as TB
from temp_amd import snapshot as S
from temp_amd.snapshot import Snapshot, comp_deg_norm, device_subsample
from tests.cpu_backend import CpuTestBackend


@pytest.fixture(autouse=True)
def cpu_backend():
    TB.set_backend(CpuTestBackend())
    yield
    TB.set_backend(None)


def random_snapshot(seed, n=60, E=900, R=8, hub=True):
    rng = np.random.default_rng(seed)
    src, dst, rel = rng.integers(0, n, E), rng.integers(0, n, E), rng.integers(0, R, E)
    if hub:
        dst[:300] = rng.integers(0, 3, 300)              # segments that span several chunks
    return Snapshot(n, src, dst, rel, np.sort(rng.choice(500, n, replace=False)))


def check_child_against_mask(g, sub, n_rel_rows, device):
    """The child's views hold exactly the kept edges, per chunk in the parent's order; degrees / norms are the subgraph's."""
    keep = sub._mask.cpu().numpy().astype(bool)
    assert keep.sum() == sub.keep == sub.number_of_edges()
    dvp, dvc = g.device_views(device, n_rel_rows), sub.device_views(device, n_rel_rows)
    eid = g.device_edge_ids(device).cpu().numpy()
    for v, name in enumerate(("by_dst", "by_src", "by_rel")):
        P = {k: t.cpu().numpy() for k, t in dvp[name].items()}
        C = {k: t.cpu().numpy() for k, t in dvc[name].items()}
        for k in ("chunk_seg", "chunk_beg", "chunk_slot", "fix_seg", "fix_slot", "fix_cnt"):
            assert np.array_equal(P[k], C[k]), (name, k)
        for c in range(P["chunk_seg"].shape[0]):
            pos = np.arange(P["chunk_beg"][c], P["chunk_end"][c])
            kp = pos[keep[eid[v][pos]]]
            assert C["chunk_end"][c] == P["chunk_beg"][c] + kp.shape[0]
            assert np.array_equal(C["a"][P["chunk_beg"][c]:C["chunk_end"][c]], P["a"][kp])
            assert np.array_equal(C["b"][P["chunk_beg"][c]:C["chunk_end"][c]], P["b"][kp])
    idx = np.nonzero(keep)[0]
    assert np.array_equal(dvc["in_deg"].cpu().numpy(), np.bincount(g.dst[idx], minlength=g.n))
    assert np.array_equal(dvc["out_deg"].cpu().numpy(), np.bincount(g.src[idx], minlength=g.n))
    assert np.array_equal(dvc["nnorm"].cpu().numpy(), comp_deg_norm(g.n, g.dst[idx]))
    assert np.array_equal(sub.src, g.src[idx]) and np.array_equal(sub.nnorm, comp_deg_norm(g.n, g.dst[idx]))


@pytest.mark.parametrize("rate", [0.5, 0.8, 0.0, 1.0])
def test_device_subsample_reference_semantics(rate):
    dev = torch.device("cpu")
    graphs = [random_snapshot(1), random_snapshot(2, n=10, E=37, hub=False), random_snapshot(3, n=5, E=0, hub=False)]
    keeps = [int(rate * g.number_of_edges()) for g in graphs]
    subs = device_subsample(graphs, keeps, [11, 12, 13], dev, 8, want_mask=True)
    for g, sub in zip(graphs, subs):
        check_child_against_mask(g, sub, 8, dev)


def test_device_subsample_is_a_uniform_k_subset():
    """Every edge is kept with probability k / E over the seeds; two seeds give different subsets; one seed gives the same one."""
    g = random_snapshot(4, E=400)
    dev = torch.device("cpu")
    hits = np.zeros(400)
    for seed in range(200):
        sub = device_subsample([g], [100], [seed], dev, 8, want_mask=True)[0]
        hits += sub._mask.numpy()
    assert abs(hits.mean() / 200 - 0.25) < 1e-9 and hits.min() > 20 and hits.max() < 85        # Binomial(200, 1/4): mean 50, sd 6
    a = device_subsample([g], [100], [7], dev, 8, want_mask=True)[0]._mask
    b = device_subsample([g], [100], [7], dev, 8, want_mask=True)[0]._mask
    c = device_subsample([g], [100], [8], dev, 8, want_mask=True)[0]._mask
    assert torch.equal(a, b) and not torch.equal(a, c)


# ---- flat on-disk store ---------------------------------------------------------------------------------------------------
def test_store_round_trip_and_window_golden(tmp_path):
    """write_store -> SnapshotStore: every array identical to the text-built snapshots, the precomputed packed views identical
    to what the planner builds at run time, and a window model constructed FROM THE STORE reproduces the reference's golden
    loss and gradients (G10)."""
    from temp_amd import _lib
    from temp_amd.store import SnapshotStore, write_store
    from tests.window_cases import check_window, slice_snapshots
    import tests.window_cases as WC
    s = slice_snapshots()
    path = str(tmp_path / "icews14_slice.tsnap")
    write_store(path, s["tr"], s["va"], s["te"], s["num_e"], s["num_r"])
    st = SnapshotStore(path)
    assert st.num_ents == s["num_e"] and st.num_rels == s["num_r"] and st.times == list(s["tr"].keys())
    tr, va, te = st.graph_dicts()
    dev = torch.device("cpu")
    for want_d, got_d in ((s["tr"], tr), (s["va"], va), (s["te"], te)):
        for t in st.times[::5]:
            a, b = want_d[t], got_d[t]
            assert a.n == b.n and np.array_equal(a.gids, b.gids) and np.array_equal(a.nnorm, b.nnorm)
            assert np.array_equal(a.src, b.src) and np.array_equal(a.dst, b.dst) and np.array_equal(a.rel, b.rel)
            da, db = a.device_views(dev, 2 * s["num_r"]), b.device_views(dev, 2 * s["num_r"])
            assert torch.equal(da["_buf"], db["_buf"]) and np.array_equal(da["_meta"]["size"], db["_meta"]["size"])
            assert np.array_equal(da["_meta"]["rel_chunks"], db["_meta"]["rel_chunks"]) and np.array_equal(da["_meta"]["n_partial"], db["_meta"]["n_partial"])
    # whole-split residency: one upload, per-snapshot views are slices of it
    big = st.to_device(dev)
    dv = tr[st.times[3]].device_views(dev, 2 * s["num_r"])
    assert dv["_buf"].data_ptr() >= big.data_ptr() and dv["_buf"].data_ptr() < big.data_ptr() + 4 * big.numel()
    # the golden window check with the model's graph dictionaries coming from the store
    saved = dict(WC._SLICE)
    try:
        WC._SLICE.update(tr=tr, va=va, te=te)
        check_window("G10_bi_grrgcn_rol", dev)
    finally:
        WC._SLICE.clear()
        WC._SLICE.update(saved)


def test_store_of_an_older_view_order_is_refused(tmp_path):
    """A file whose header carries layout version 1 (views in plain segment order, before the relation-ordered views) must not
    load: the device edge-id tables assume the new order."""
    from temp_amd.store import VERSION, SnapshotStore, write_store
    from tests.window_cases import slice_snapshots
    s = slice_snapshots()
    path = str(tmp_path / "old.tsnap")
    few = lambda d: {t: d[t] for t in list(d.keys())[:3]}
    write_store(path, few(s["tr"]), few(s["va"]), few(s["te"]), s["num_e"], s["num_r"])
    assert VERSION >= 2 and SnapshotStore(path).T == 3
    raw = bytearray(open(path, "rb").read())
    raw[8:16] = np.int64(1).tobytes()
    open(path, "wb").write(raw)
    with pytest.raises(ValueError, match="layout version 1"):
        SnapshotStore(path)
