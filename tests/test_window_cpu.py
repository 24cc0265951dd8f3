"""Window-level host logic on CPU (test-only backend): plan building, compact history with the
reference's F8 semantics, None-padded windows, bi-directional flip, batched == reference-granular,
loss + gradients against the golden vectors recorded from the reference."""
import numpy as np
import pytest
import torch

from temp_amd import backend as TB
from temp_amd.window import ChainPlan, window_times
from tests.cpu_backend import CpuTestBackend
from tests.golden_util import load
from tests.window_cases import (check_batched_equals_generic, check_evaluate, check_sa_dense_api, check_sa_evaluate, check_sa_window,
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


@pytest.mark.parametrize("name", ["G10_uni_grrgcn", "G10_uni_grrgcn_rol", "G10_bi_grrgcn_rol", "G10_bi_grrgcn"])
def test_window_loss_and_grads_golden(name):
    check_window(name, torch.device("cpu"))


@pytest.mark.parametrize("name", ["G10_uni_grrgcn_rol", "G10_bi_grrgcn_rol"])
def test_window_generic_path_golden(name):
    check_window(name, torch.device("cpu"), batched=False)


@pytest.mark.parametrize("name", ["G10_uni_grrgcn_rol", "G10_bi_grrgcn_rol"])
def test_batched_equals_generic(name):
    check_batched_equals_generic(name, torch.device("cpu"))


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
