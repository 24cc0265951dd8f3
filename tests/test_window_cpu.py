"""Window-level host logic on CPU (test-only backend): plan building, compact history with the
reference's F8 semantics, None-padded windows, bi-directional flip, batched == reference-granular,
loss + gradients against the golden vectors recorded from the reference."""
import os

import numpy as np
import pytest
import torch

from temp_amd import backend as TB
from temp_amd.window import ChainPlan, window_times
from tests.cpu_backend import CpuTestBackend
from tests.golden_util import load
from tests.window_cases import (check_batched_equals_generic, check_dropout_visits_are_independent, check_dropout_visits_self_attention, check_fused_ensemble_loss, check_static_prepare_split, check_wide_batched_equals_generic, check_evaluate, check_sa_dense_api, check_sa_evaluate, check_sa_window,
                                check_static, check_window, slice_snapshots)
from oracle import temp_oracle as O


@pytest.fixture(autouse=True)
def cpu_backend():
    TB.set_backend(CpuTestBackend())
    yield
    TB.set_backend(None)


def test_window_times_match_oracle_restatement():
    s = slice_snapshots()
    for tl, L in (([20, 15, 9, 3], 8), ([0], 5), ([23, 22, 1], 6)):
        rows = window_times(tl, L, s["times"])
        assert [list(x) for x in zip(*rows)] == O.get_batch_graph_list(tl, L, s["times"])
        fwd, bwd = O.get_batch_graph_list_bi(tl, L, s["times"])
        assert [list(x) for x in zip(*window_times(tl, L, s["times"], ascending=True))] == bwd


def test_plan_matches_dense_history_semantics():
    """ChainPlan's (prev_idx, dt) == what the reference's dense re-zeroed history would give (F8)."""
    s = slice_snapshots()
    tl, L = [20, 15, 9, 3], 8
    rows = window_times(tl, L, s["times"])
    plan = ChainPlan(rows, s["tr"], s["num_e"], L)
    hist_mark = np.full((len(tl), s["num_e"]), -1, dtype=np.int64)     # which (step,row) wrote each dense row
    start = np.zeros((len(tl), s["num_e"]), dtype=np.float32)
    si = 0
    for p in range(L - 1):
        win = [b for b in range(len(tl)) if rows[b][p] is not None]
        if not win:
            continue
        st = plan.steps[si]
        off = 0
        new_mark = np.full_like(hist_mark, -1)
        for b in win:
            ids = s["tr"][rows[b][p]].gids
            assert np.array_equal(st.prev_idx[off:off + len(ids)], hist_mark[b][ids])
            assert np.array_equal(st.dt[off:off + len(ids)], p - start[b][ids])
            new_mark[b][ids] = off + np.arange(len(ids))
            start[b][ids] = p
            off += len(ids)
        hist_mark = new_mark
        si += 1
    assert si == len(plan.steps)
    for b in range(len(tl)):
        a, d = plan.final_all(b, L - 1)
        assert np.array_equal(a, hist_mark[b]) and np.array_equal(d, L - 1 - start[b])


# (G10_bi_grrgcn_rol_d200 -- the headline model at D = 200, L = 15 -- takes five minutes through the test-only CPU backend; the
#  same golden pins the oracle in test_oracle_golden.py and the HIP path in test_gpu_parity.py, so here it runs on request only)
_WINDOW_GOLDENS = ["G10_uni_grrgcn", "G10_uni_grrgcn_rol", "G10_bi_grrgcn_rol", "G10_bi_grrgcn"] + \
    (["G10_bi_grrgcn_rol_d200"] if os.environ.get("TEMP_FULL_CPU_TESTS") else [])


@pytest.mark.parametrize("name", _WINDOW_GOLDENS)
def test_window_loss_and_grads_golden(name):
    check_window(name, torch.device("cpu"))


@pytest.mark.parametrize("name", ["G10_uni_grrgcn_rol", "G10_bi_grrgcn_rol"])
def test_window_generic_path_golden(name):
    check_window(name, torch.device("cpu"), batched=False)


@pytest.mark.parametrize("name", ["G10_uni_grrgcn_rol", "G10_bi_grrgcn_rol"])
def test_batched_equals_generic(name):
    check_batched_equals_generic(name, torch.device("cpu"))


def test_default_flags_position_loop_golden():
    check_window("G10_uni_grrgcn", torch.device("cpu"), stack=False)


@pytest.mark.parametrize("type1", [False, True])
def test_default_flags_stack_equals_generic(type1):
    from tests.window_cases import check_stack_equals_generic
    check_stack_equals_generic("G10_uni_grrgcn", torch.device("cpu"), type1)


def test_static_rgcn_golden():
    check_static(torch.device("cpu"))


@pytest.mark.parametrize("name", ["G13_eval_uni", "G13_eval_bi"])
def test_evaluate_ranks_golden(name):
    check_evaluate(name, torch.device("cpu"))


@pytest.mark.parametrize("name", ["G14_sa_uni_rol", "G14_sa_uni", "G14_sa_bi_rol"])
def test_self_attention_window_golden(name):
    check_sa_window(name, torch.device("cpu"))


def test_self_attention_dense_api():
    check_sa_dense_api(torch.device("cpu"))


@pytest.mark.parametrize("name", ["G14_sa_uni", "G14_sa_bi_rol"])
def test_self_attention_evaluate(name):
    check_sa_evaluate(name, torch.device("cpu"))


def test_batch_prefetcher_yields_prepared_batches_in_order():
    from temp_amd.prefetch import BatchPrefetcher
    from tests.window_cases import build_window_model
    z = load("G10_bi_grrgcn_rol")
    m = build_window_model(z, torch.device("cpu"))
    batches = [[20, 15, 9], [18, 4], [7]]
    got = list(BatchPrefetcher(m, batches, seq_len=int(z["L"]), depth=1))
    assert [len(wb.rows) for wb in got] == [3, 2, 1]
    assert [wb.rows[0][-1] for wb in got] == [20, 18, 7]
    loss = m.run_loss(got[1])
    assert torch.isfinite(loss)

    def boom():
        yield [20]
        raise RuntimeError("bad batch")
    with pytest.raises(RuntimeError):
        list(BatchPrefetcher(m, boom(), seq_len=int(z["L"])))


def test_batch_prefetcher_workers_same_batches_as_one_worker():
    """Two workers prepare out of order; the batches come out in order and -- per-batch seeds -- with the same random
    subsets (target edge subsample, positives) as one worker draws."""
    from temp_amd.prefetch import BatchPrefetcher
    from tests.window_cases import build_window_model
    z = load("G10_bi_grrgcn_rol")
    batches = [[20, 15, 9], [18, 4], [7], [19, 12], [16, 3, 5], [11]]
    runs = []
    for workers in (1, 2, 3):
        m = build_window_model(z, torch.device("cpu"))
        m.sample_rng = np.random.default_rng(11)
        got = list(BatchPrefetcher(m, batches, seq_len=int(z["L"]), depth=1, workers=workers, batch_seeds=True))
        assert [wb.rows[0][-1] for wb in got] == [b[0] for b in batches]
        runs.append([(wb.n_edge_visits, [(g.src.tolist(), g.rel.tolist(), g.dst.tolist()) for g in wb.target.graphs],
                      [t.tolist() for t in wb.loss_plan["triples"]] if wb.loss_plan else None) for wb in got])
    assert runs[0] == runs[1] == runs[2]
    with pytest.raises(ValueError):
        BatchPrefetcher(m, batches, workers=2, batch_seeds=False)


def test_prefetcher_gate_parks_worker_python_while_consumer_issues():
    """Cooperative hand-over (prefetch.BatchPrefetcher, _lib.pause_point): while the consumer holds the loop body the workers'
    planning stops at its next stage boundary, so at most the batches already under way get finished; it resumes when the consumer
    asks for the next batch; the batches are the same with the gate off."""
    import threading
    import time
    from temp_amd import _lib
    from temp_amd.prefetch import BatchPrefetcher
    from tests.window_cases import build_window_model
    # pause_point itself: parks on a closed gate with the token handed back, passes an open one, no-op without a gate
    _lib.pause_point()
    gate, seen = threading.Event(), []

    def worker():
        _lib.coop_begin(gate)
        try:
            seen.append("start")
            _lib.pause_point()
            seen.append("passed")
        finally:
            _lib.coop_end()
    th = threading.Thread(target=worker)
    th.start()
    time.sleep(0.2)
    assert seen == ["start"] and not _lib._py_token.locked()       # parked, token free for another worker
    gate.set()
    th.join(5)
    assert seen == ["start", "passed"] and not _lib._py_token.locked()
    z = load("G10_bi_grrgcn_rol")
    batches = [[20, 15, 9], [18, 4], [7], [19, 12], [16, 3, 5], [11], [14, 2], [8]]
    runs = []
    for coop in (True, False):
        m = build_window_model(z, torch.device("cpu"))
        m.sample_rng = np.random.default_rng(11)
        done = []
        orig = m.prepare

        def counted(*a, _orig=orig, **k):
            r = _orig(*a, **k)
            done.append(time.perf_counter())
            return r
        m.prepare = counted
        got, during = [], []
        for wb in BatchPrefetcher(m, batches, seq_len=int(z["L"]), depth=4, workers=2, batch_seeds=True, cooperative=coop):
            n0 = len(done)
            time.sleep(0.15)                       # "issuing a step": far longer than a prepare of these windows
            during.append(len(done) - n0)
            got.append(wb)
        if coop:                                   # a worker finishes at most the stage it was in: no batch is completed start to end
            assert sum(during) <= 2 * 2, during    # (window = depth + workers - 1 = 5 batches could have been, without the gate)
        runs.append([(wb.n_edge_visits, [(g.src.tolist(), g.rel.tolist(), g.dst.tolist()) for g in wb.target.graphs]) for wb in got])
    assert runs[0] == runs[1] and len(runs[0]) == len(batches)


def test_device_negative_sampler_filters_true_triples():
    """DeviceCorruptTriples (torch ops on the model's device; here the CPU device) keeps the reference sampler's
    contract: column 0 = the true entity (global id), no sampled candidate forms a true triple of the snapshot."""
    from temp_amd.sampling import DeviceCorruptTriples
    from tests.window_cases import make_args
    s = slice_snapshots()
    args = make_args(negative_rate=50, num_pos_facts=40)
    smp = DeviceCorruptTriples(args, s["tr"], torch.device("cpu"), seed=3)
    t = s["times"][12]
    g = s["tr"][t]
    trip, nt, nh, labels = smp.single_graph_negative_sampling(t, g, s["num_e"])
    P = min(g.number_of_edges(), 40)
    assert trip.shape == (P, 3) and nt.shape == (P, 51) and nh.shape == (P, 51) and int(labels.abs().sum()) == 0
    gid = torch.from_numpy(g.gids)
    assert torch.equal(nt[:, 0], gid[trip[:, 2]]) and torch.equal(nh[:, 0], gid[trip[:, 0]])
    true_tail = {(int(a), int(r), int(gid[b])) for a, r, b in zip(g.src, g.rel, g.dst)}
    true_head = {(int(gid[a]), int(r), int(b)) for a, r, b in zip(g.src, g.rel, g.dst)}
    for p in range(P):
        h, r, tl = (int(x) for x in trip[p])
        assert all((h, r, int(c)) not in true_tail for c in nt[p, 1:])
        assert all((int(c), r, tl) not in true_head for c in nh[p, 1:])
    assert int(nt[:, 1:].min()) >= 0 and int(nt[:, 1:].max()) < s["num_e"]


def test_evaluation_filter_lists_match_dictionary_lookups():
    """EvaluationFilter.filter_lists (two searchsorted calls) == the reference's per-triple dictionary of known-true
    entities (utils/evaluation.py:14-38 builds true_heads / true_tails dicts from train+valid+test)."""
    from temp_amd.evaluation import EvaluationFilter
    s = slice_snapshots()
    t = s["times"][9]
    ev = EvaluationFilter(None, None, s["tr"], s["va"], s["te"])
    R, tails, heads = ev._true_keys(t, s["num_e"])
    true_tails, true_heads = {}, {}
    for gd in (s["tr"], s["va"], s["te"]):
        g = gd[t]
        for h, r, o in zip(g.src, g.rel, g.dst):
            true_tails.setdefault((int(h), int(r)), set()).add(int(g.gids[o]))
            true_heads.setdefault((int(o), int(r)), set()).add(int(g.gids[h]))
    g = s["te"][t]
    trip = np.stack([g.src, g.rel, g.dst], axis=1).astype(np.int64)
    ptr, ids = ev.filter_lists(trip[:, 0] * R + trip[:, 1], tails, s["num_e"])
    assert ptr[0] == 0 and ptr[-1] == ids.shape[0] and ptr.dtype == np.int32
    for p, (h, r, o) in enumerate(trip):
        assert set(ids[ptr[p]:ptr[p + 1]].tolist()) == true_tails[(int(h), int(r))]
        assert list(ids[ptr[p]:ptr[p + 1]]) == sorted(ids[ptr[p]:ptr[p + 1]])
    ptr, ids = ev.filter_lists(trip[:, 2] * R + trip[:, 1], heads, s["num_e"])
    for p, (h, r, o) in enumerate(trip):
        assert set(ids[ptr[p]:ptr[p + 1]].tolist()) == true_heads[(int(o), int(r))]
    # unknown prefix -> empty list
    ptr, ids = ev.filter_lists(np.array([R * s["num_e"] + 5]), tails, s["num_e"])
    assert list(ptr) == [0, 0] and ids.shape[0] == 0


@pytest.mark.parametrize("score_function", ["complex", "distmult", "transE"])
def test_filtered_ranks_equal_masked_sort(score_function):
    """calc_metrics_single_graph (GEMM + counting, or the candidate-axis scorer for transE) == the reference's recipe
    written out: mask the other true entities to -10e6, sigmoid, stable descending sort, index of the target."""
    from temp_amd import scores as SC
    from temp_amd.evaluation import EvaluationFilter
    from tests.window_cases import make_args
    s = slice_snapshots()
    t = s["times"][14]
    g = s["va"][t]
    N, D = s["num_e"], 16
    N4 = N - N % 4                                       # fused path needs N % 4 == 0; also run the ragged size
    for n_ent in (N4, N4 + 1):
        gids_ok = g.gids[np.stack([g.src, g.dst])].max() < n_ent
        if not gids_ok:
            continue
        torch.manual_seed(5)
        all_e = torch.randn(n_ent, D)
        ent = all_e[torch.from_numpy(g.gids)]
        rel = torch.randn(2 * s["num_r"], D)
        args = make_args(score_function=score_function)
        fn = getattr(SC, score_function)
        ev = EvaluationFilter(args, fn, s["tr"], s["va"], s["te"])
        samples = torch.from_numpy(np.stack([g.src, g.rel, g.dst], axis=1))
        got = ev.calc_metrics_single_graph(ent, rel, all_e, samples, g, t)
        R, tails, heads = ev._true_keys(int(t), n_ent)
        want = []
        for mode in ("head", "tail"):
            for h, r, o in samples.tolist():
                if mode == "tail":
                    sc = fn(ent[h][None], rel[r][None], all_e, mode="tail")[0]
                    tgt, keys, prefix = int(g.gids[o]), tails, h * R + r
                else:
                    sc = fn(all_e, rel[r][None], ent[o][None], mode="head")[0]
                    tgt, keys, prefix = int(g.gids[h]), heads, o * R + r
                known = keys[(keys >= prefix * n_ent) & (keys < (prefix + 1) * n_ent)] % n_ent
                sc = sc.clone()
                keep = sc[tgt].clone()
                sc[torch.from_numpy(known)] = -10e6
                sc[tgt] = keep
                order = torch.sort(torch.sigmoid(sc), descending=True, stable=True).indices
                want.append(int((order == tgt).nonzero()[0, 0]) + 1)
        # the GEMM and the broadcast scorer sum in different orders: ranks may differ where two sigmoids are within an ulp
        want = torch.tensor(want)
        assert got.shape == want.shape
        assert (got == want).float().mean() > 0.97 and (got - want).abs().max() <= 2, (score_function, n_ent)


@pytest.mark.parametrize("module", ["GRRGCN", "BiGRRGCN", "SARGCN"])
def test_random_dropout_resamples_history_visits(module):
    """--random-dropout (models/DynamicRGCN.py:162-171): every TRAINING visit of a history snapshot keeps a random
    80 % of its edges with norms recomputed; evaluation and the node sets / row maps are untouched; the restructured
    and the reference-granular paths see the same draws and agree."""
    from temp_amd.bi_dynamic_rgcn import BiDynamicRGCN
    from temp_amd.dynamic_rgcn import DynamicRGCN
    from temp_amd.self_attention_rgcn import SelfAttentionRGCN
    from tests.window_cases import make_args
    s = slice_snapshots()
    cls = {"GRRGCN": DynamicRGCN, "BiGRRGCN": BiDynamicRGCN, "SARGCN": SelfAttentionRGCN}[module]
    args = make_args(module=module, rec_only_last_layer=True, embed_size=16, hidden_size=16, n_bases=4, train_seq_len=5, test_seq_len=5,
                     random_dropout=True, use_time_embedding=(module == "SARGCN"))
    torch.manual_seed(1)
    m = cls(args, s["num_e"], s["num_r"], s["tr"], s["va"], s["te"])
    t_list = torch.tensor([s["times"][i] for i in (12, 9, 7)])
    full = m.prepare(t_list, 5, train=False)
    m.sample_rng = np.random.default_rng(8)
    wb = m.prepare(t_list, 5, train=True)
    if module == "SARGCN":
        hist_full = full.g_all.number_of_edges() - sum(g.number_of_edges() for g in full.targets)
        hist_drop = wb.g_all.number_of_edges() - sum(g.number_of_edges() for g in wb.targets)
        visits = sum(1 for times in wb.hist_times for t in times if t is not None)
        assert wb.n_hist_rows >= full.n_hist_rows                     # visits are no longer shared between windows
        want = sum(int(0.8 * s["tr"][t].number_of_edges()) for times in wb.hist_times for t in times if t is not None)
        assert hist_drop == want and visits > 0 and hist_full > 0
    else:
        plans = wb.plan if isinstance(wb.plan, tuple) else (wb.plan,)
        fplans = full.plan if isinstance(full.plan, tuple) else (full.plan,)
        for pl, fp in zip(plans, fplans):
            for st, fs in zip(pl.steps, fp.steps):
                assert [g.number_of_edges() for g in st.graphs] == [int(0.8 * g.number_of_edges()) for g in fs.graphs]
                assert all(np.array_equal(a.gids, b.gids) for a, b in zip(st.graphs, fs.graphs))
                assert np.array_equal(st.prev_idx, fs.prev_idx)
                for g in st.graphs:                                   # norms recomputed from the subgraph's in-degrees
                    deg = np.bincount(g.dst, minlength=g.n)
                    assert np.allclose(g.nnorm, np.where(deg > 0, 1.0 / np.maximum(deg, 1), 0.0))
        outs = []
        for batched in (True, False):
            m.use_batched_path = batched
            m.sample_rng = np.random.default_rng(8)
            outs.append(m.run(m.prepare(t_list, 5, train=True))[0])
        assert torch.allclose(outs[0], outs[1], rtol=1e-5, atol=2e-6)
    loss = m.run_loss(wb)
    loss.backward()
    assert torch.isfinite(loss) and m.ent_embeds.grad.abs().sum() > 0


@pytest.mark.parametrize("name", ["G10_bi_grrgcn_rol", "G10_uni_grrgcn_rol"])
def test_planned_loss_equals_injected_samples(name):
    """prepare() plans the loss on the host (positives, operand indices, known-true slices of the resident store); run_loss
    then needs ONE sampler call.  The result equals the reference-shaped path fed with the same candidates, the store's
    slices equal the reference's true_tail / true_head dictionaries, and no drawn candidate is a true triple."""
    from tests.window_cases import build_window_model
    z = load(name)
    m = build_window_model(z, torch.device("cpu"))
    m.args.num_pos_facts = 37                                   # below some graphs' edge counts: random subset; odd: row padding
    t_list = torch.tensor([int(t) for t in z["t_list"]])
    m.sample_rng = np.random.default_rng(3)
    wb = m.prepare(t_list, int(z["L"]), train=True)
    plan = wb.loss_plan
    assert plan is not None and getattr(wb, "all_maps", None) is not None
    s = slice_snapshots()
    ids = plan["ids"].numpy()
    lo, hi, truth, is_tail = (plan[k].numpy() for k in ("lo", "hi", "truth", "is_tail"))
    row = 0
    for b, g in enumerate(wb.graphs):
        trip = plan["triples"][b]
        P = trip.shape[0]
        pad = (-2 * P) % 4                                      # weight-0 rows that round the window's block up to 4 rows
        assert P == min(g.number_of_edges(), 37) and plan["splits"][b] == (row, row + 2 * P + pad)
        wts = plan["weights"].numpy()
        assert (wts[row:row + 2 * P] == np.float32(1.0 / P)).all() and (wts[row + 2 * P:row + 2 * P + pad] == 0).all()
        tails, heads = {}, {}
        for h, r, o in zip(g.src, g.rel, g.dst):
            tails.setdefault((int(h), int(r)), set()).add(int(g.gids[o]))
            heads.setdefault((int(r), int(o)), set()).add(int(g.gids[h]))
        for i, (h, r, o) in enumerate(trip):
            assert set(ids[lo[row + i]:hi[row + i]].tolist()) == tails[(int(h), int(r))] and truth[row + i] == g.gids[o] and is_tail[row + i] == 1
            assert set(ids[lo[row + P + i]:hi[row + P + i]].tolist()) == heads[(int(r), int(o))] and truth[row + P + i] == g.gids[h]
        row += 2 * P + pad
    m.seed_rng = np.random.default_rng(7)
    loss1 = m.run_loss(wb)
    m.seed_rng = np.random.default_rng(7)
    cand = TB.get_backend().corrupt_sample(int(m.seed_rng.integers(1 << 62)), plan["truth"], plan["lo"], plan["hi"], plan["ids"],
                                           m.args.negative_rate, m.num_ents)
    cn = cand.numpy()
    assert cn.shape == (row, 1 + m.args.negative_rate) and np.array_equal(cn[:, 0], truth) and cn.min() >= 0 and cn.max() < m.num_ents
    for r in range(row):
        assert not np.isin(cn[r, 1:], ids[lo[r]:hi[r]]).any()
    samples = []
    for b, (a0, a1) in enumerate(plan["splits"]):
        P = plan["triples"][b].shape[0]                          # (a block may end in weight-0 padding rows)
        samples.append((torch.from_numpy(plan["triples"][b]), cand[a0:a0 + P].long(), cand[a0 + P:a0 + 2 * P].long()))
    loss2 = m.run_loss(wb, samples)
    assert abs(loss1.item() - loss2.item()) < 1e-5 * max(1.0, abs(loss2.item()))
    loss1.backward()
    assert torch.isfinite(m.ent_embeds.grad).all() and m.rel_embeds.grad.abs().sum() > 0


def test_static_rgcn_fused_loss_equals_per_graph_path():
    """StaticRGCN.forward: the fused all-window loss (shared isolated table, planned positives, one sampler call) equals the
    reference-shaped per-graph loop fed with the same target-edge subsample and the same candidates; gradients too."""
    from temp_amd.static_rgcn import StaticRGCN
    from tests.window_cases import make_args
    s = slice_snapshots()
    args = make_args(module="SRGCN", embed_size=16, hidden_size=16, n_bases=4, negative_rate=20, num_pos_facts=30)
    torch.manual_seed(2)
    m = StaticRGCN(args, s["num_e"], s["num_r"], s["tr"], s["va"], s["te"])
    tl = [s["times"][i] for i in (3, 9, 17)]
    rng = np.random.default_rng(1)
    edge_ids = [rng.choice(s["tr"][t].number_of_edges(), size=s["tr"][t].number_of_edges() // 2, replace=False) for t in tl]
    loss1 = m(torch.tensor(tl), target_edge_ids=edge_ids)
    plan, cand = m._last_plan
    loss1.backward()
    g1 = (m.ent_embeds.grad.clone(), m.rel_embeds.grad.clone(), m.ent_encoder.layer_1.weight.grad.clone())
    samples = []
    for trip, (a0, _) in zip(plan["triples"], plan["splits"]):
        P = trip.shape[0]
        samples.append((torch.from_numpy(trip), cand[a0:a0 + P].long(), cand[a0 + P:a0 + 2 * P].long()))
    for p in m.parameters():
        p.grad = None
    loss2 = m(torch.tensor(tl), target_edge_ids=edge_ids, samples=samples)
    loss2.backward()
    assert abs(loss1.item() - loss2.item()) < 1e-5 * max(1.0, abs(loss2.item()))
    for a, b in zip(g1, (m.ent_embeds.grad, m.rel_embeds.grad, m.ent_encoder.layer_1.weight.grad)):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)


# ---- config 3: post-ensemble / impute window models (G15) ------------------------------------------------------------
@pytest.mark.parametrize("batched", [True, False])
def test_post_ensemble_bi_window_golden(batched):
    from tests.window_cases import check_post_bi
    check_post_bi(torch.device("cpu"), batched)


@pytest.mark.parametrize("name,batched", [("G15_impute_bi", True), ("G15_impute_bi", False), ("G15_impute_uni", True),
                                          ("G15_impute_uni", False), ("G15_impute_uni_full", False)])
def test_impute_window_golden(name, batched):
    from tests.window_cases import check_impute_window
    check_impute_window(name, torch.device("cpu"), batched)


@pytest.mark.parametrize("name,batched", [("G16_eval_impute_uni", True), ("G16_eval_impute_uni", False), ("G16_eval_impute_bi", True),
                                          ("G16_eval_impute_bi", False)])
def test_impute_evaluate_golden(name, batched):
    """evaluate() of the impute models (config 3) against the reference's own filtered ranks and loss."""
    from tests.window_cases import check_impute_evaluate
    check_impute_evaluate(name, torch.device("cpu"), batched)


def test_post_ensemble_loss_definition():
    from tests.window_cases import check_post_ensemble_loss
    check_post_ensemble_loss(torch.device("cpu"))


@pytest.mark.parametrize("name", ["G17_post_eval_complex", "G17_post_eval_distmult"])
def test_post_evaluation_filters_golden(name):
    """utils/post_evaluation.py: embedding-level and score-level ensemble ranking against the reference's own ranks."""
    from tests.window_cases import check_post_eval_filters
    check_post_eval_filters(name, torch.device("cpu"))


@pytest.mark.parametrize("name,batched", [("G18_eval_post_uni", True), ("G18_eval_post_uni", False), ("G18_eval_post_bi", True), ("G18_eval_post_bi", False)])
def test_post_ensemble_evaluate_golden(name, batched):
    from tests.window_cases import check_post_ensemble_evaluate
    check_post_ensemble_evaluate(name, torch.device("cpu"), batched)


@pytest.mark.parametrize("name,batched", [("G19_post_ratio_uni", True), ("G19_post_ratio_bi", True), ("G19_post_ratio_bi", False)])
def test_post_ensemble_own_ratio_golden(name, batched):
    """config 3 end to end: frequency features + MLP mixing weights + loss against the reference's own calc_ensemble_ratio."""
    from tests.window_cases import check_post_ensemble_ratio
    check_post_ensemble_ratio(name, torch.device("cpu"), batched)


def test_relu_fold_is_off_for_widths_beyond_the_gather_kernels():
    """D = 260 > 256: the chain gather cannot carry layer 2's ReLU adjoint, so the layer keeps it."""
    check_wide_batched_equals_generic(torch.device("cpu"))


@pytest.mark.parametrize("module", ["GRRGCN", "BiGRRGCN"])
def test_dropout_visits_are_independent(module):
    check_dropout_visits_are_independent(torch.device("cpu"), module)


def test_dropout_visits_self_attention():
    check_dropout_visits_self_attention(torch.device("cpu"))


@pytest.mark.parametrize("head_as_tail", [False, True])
def test_fused_ensemble_loss_equals_reference_shaped(head_as_tail):
    check_fused_ensemble_loss(torch.device("cpu"), head_as_tail)


def test_static_prepare_split_equals_forward():
    check_static_prepare_split(torch.device("cpu"))


@pytest.mark.parametrize("module,rol", [("BiGRRGCN", True), ("GRRGCN", True), ("GRRGCN", False)])
def test_dropout_all_entity_pass_keeps_windows_apart(module, rol):
    from tests.window_cases import check_dropout_all_entity_pass
    check_dropout_all_entity_pass(torch.device("cpu"), module, rol)


@pytest.mark.parametrize("bi", [False, True])
def test_post_ensemble_all_entity_pass_window_entity_layout(bi):
    from tests.window_cases import check_post_ensemble_rep_layout
    check_post_ensemble_rep_layout(torch.device("cpu"), bi)
