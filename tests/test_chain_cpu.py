"""Persistent window chain, host side: the track / panel planner (GruProgram.chain_plan) and the table semantics, executed
by the test backend's panel-by-panel reference, against the per-position path."""
import numpy as np
import pytest
import torch

from temp_amd import backend as TB
from tests.chain_cases import check_plan_invariants, make_rnns, random_program, run_program
from tests.cpu_backend import CpuTestBackend
from tests.golden_util import assert_close


@pytest.fixture(autouse=True)
def cpu_backend():
    TB.set_backend(CpuTestBackend())
    yield
    TB.set_backend(None)


@pytest.mark.parametrize("seed,kw", [(1, {}), (2, dict(n_chain=1, K=9, E=40, lo=1, hi=40)), (3, dict(n_chain=3, K=4, E=300, lo=100, hi=300)),
                                     (4, dict(dense_first=True, E=70))])
def test_chain_plan_invariants(seed, kw):
    prog, _ = random_program(seed, **kw)
    plan = check_plan_invariants(prog)
    fill = (plan["rows"] >= 0).mean()
    assert fill > 0.3


def test_chain_plan_dense_panels_are_full():
    """Every entity active at every position (the GDELT-shaped case): tracks = entities, panels of 32 consecutive rows."""
    prog, _ = random_program(5, n_chain=1, K=5, E=96, lo=96, hi=96)
    plan = check_plan_invariants(prog)
    assert plan["panel"].shape[0] == 3 and (plan["rows"] >= 0).all()
    assert (plan["panel"][:, 2] == 5).all()


def test_single_position_programs_keep_the_pointwise_path():
    from temp_amd.gru_chain import zero_state_program
    assert zero_state_program(50).chain_plan() is None


@pytest.mark.parametrize("type1", [False, True])
@pytest.mark.parametrize("want", [None, "ends"])
def test_chain_tables_reproduce_per_position_path(type1, want):
    d = 16
    prog, n_x = random_program(7)
    w = None if want is None else tuple(i for i, it in enumerate(prog.inst) if it.next < 0 or i % 4 == 1)
    rnns = make_rnns(2, d, type1, 3)
    a = run_program(prog, n_x, d, rnns, torch.device("cpu"), w, type1, 11, chain_kernels=True)
    b = run_program(prog, n_x, d, rnns, torch.device("cpu"), w, type1, 11, chain_kernels=False)
    for x, y in zip(a[0], b[0]):
        assert_close(x, y, 1e-5, 1e-6, "states")
    assert_close(a[1], b[1], 1e-5, 1e-6, "d_x")
    for x, y in zip(a[2], b[2]):
        assert_close(x, y, 1e-4, 1e-5, "GRU parameter gradient")


def shared_labels(prog, n_x, n_labels, seed):
    """x-row labels with many repeats inside each group of a program (an entity whose snapshot row is shared by positions)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, n_labels, n_x)


@pytest.mark.parametrize("type1", [False, True])
def test_shared_input_gates_reproduce_per_row_gates(type1):
    """GruProgram.gi_shared / TempGruChain.gi_index: a program whose x rows repeat computes the input gates once per distinct
    row of a group and the chain reads them through gi_index -- the same states and gradients as the per-position path, which
    computes the gates of every row."""
    d = 16
    prog, n_x = random_program(9)
    labels = shared_labels(prog, n_x, n_x // 3, 5)
    rnns = make_rnns(2, d, type1, 3)
    a = run_program(prog, n_x, d, rnns, torch.device("cpu"), None, type1, 11, chain_kernels=True, x_src=labels)
    sh = prog.gi_shared(torch.device("cpu"))
    assert sh is not None and sh["rows"] < prog.n_total
    idx = sh["gi_index"].numpy()
    for gi, g in enumerate(prog.groups):                       # gi_index: per group, equal labels <=> equal gi rows; rep points at one of them
        lab = labels[g["x0"]:g["x1"]]
        loc = idx[g["h0"]:g["h1"]] - sh["g0"][gi]
        assert loc.min() == 0 and loc.max() == len(np.unique(lab)) - 1
        assert np.array_equal(lab[sh["rep"][gi].numpy()][loc], lab)
    b = run_program(prog, n_x, d, rnns, torch.device("cpu"), None, type1, 11, chain_kernels=False, x_src=labels)
    assert getattr(prog, "_gi_shared", None) is None          # the per-position path never asks
    for x, y in zip(a[0], b[0]):
        assert_close(x, y, 1e-5, 1e-6, "states")
    assert_close(a[1], b[1], 1e-5, 1e-6, "d_x")
    for x, y in zip(a[2], b[2]):
        assert_close(x, y, 1e-4, 1e-5, "GRU parameter gradient")
    prog.x_src = np.arange(n_x)                                # nothing repeats: no sharing
    prog.__dict__.pop("_gi_shared", None)
    assert prog.gi_shared(torch.device("cpu")) is None


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_chain_tracks_planner_library_matches_numpy(seed):
    """temp_host_chain_tracks (C++) == GruProgram._chain_plan_numpy, bit for bit: random chains with births, deaths, re-entries
    and empty positions."""
    from tests import chain_cases as CC
    for kw in (dict(n_chain=2, K=6, E=90, lo=20, hi=70), dict(n_chain=3, K=9, E=40, lo=0, hi=40), dict(n_chain=1, K=15, E=300, lo=100, hi=300)):
        prog, _ = CC.random_program(seed=seed, **kw)
        a, b = prog._chain_plan_host(), prog._chain_plan_numpy()
        assert (a is None) == (b is None)
        if a is None:
            continue
        for k in ("panel", "rows", "any_prev", "step_inst"):
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, k
            assert np.array_equal(a[k], b[k]), k
        # the fill pass planning on its own (no tables kept from the sizing pass) gives the same arrays
        from temp_amd import _hostlib, _lib
        inst = prog.inst
        chains = []
        for head, it0 in enumerate(inst):
            if it0.prev < 0:
                chain = [head]
                while inst[chain[-1]].next >= 0:
                    chain.append(inst[chain[-1]].next)
                chains.append(chain)
        c = _hostlib.chain_tracks(chains, [it.n for it in inst], [it.h0 for it in inst], [it.rnn for it in inst],
                                  [getattr(it, "prev_idx", None) if it.prev >= 0 else None for it in inst], _lib.CHAIN_TRACKS,
                                  _lib.CHAIN_MAX_STEPS, reuse_sizing_pass=False)
        for k, v in zip(("panel", "rows", "any_prev", "step_inst"), c):
            assert np.array_equal(a[k], v), k
