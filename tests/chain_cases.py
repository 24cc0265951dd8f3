"""Random chain programs shared by the CPU planner tests and the GPU kernel tests of the persistent window chain."""
import numpy as np
import torch
import torch.nn as nn

from temp_amd import _lib
from temp_amd import gru_chain as GC
from temp_amd.gru_cell import GRUCell
from temp_amd.gru_chain import GruInstance, GruProgram


def random_program(seed, n_chain=2, K=6, E=90, lo=20, hi=70, dense_first=False):
    """`n_chain` independent chains (GRU index = chain index) of K positions; every position holds a random subset of E
    entities (sorted), a row's previous state is the same entity's row one position earlier, if it was active there."""
    rng = np.random.default_rng(seed)
    inst, x_off = [], 0
    for c in range(n_chain):
        prev_inst, prev_ids = -1, None
        for k in range(K):
            n = E if (dense_first and c == 0) else int(rng.integers(lo, hi + 1))
            ids = np.sort(rng.choice(E, size=min(n, E), replace=False))
            if k == 3 and c == 1:
                ids = ids[:0]                                         # an empty position breaks every track of the chain
            if prev_ids is None:
                pidx = np.full(len(ids), -1, dtype=np.int64)
            else:
                pos = {int(e): i for i, e in enumerate(prev_ids)}
                pidx = np.array([pos.get(int(e), -1) for e in ids], dtype=np.int64)
            dt = rng.integers(1, 5, len(ids)).astype(np.float32)
            inst.append(GruInstance(len(ids), x_off, c, prev_inst, pidx, dt))
            prev_inst, prev_ids = len(inst) - 1, ids
            x_off += len(ids)
    return GruProgram(inst), x_off


def make_rnns(n, d, type1, seed):
    torch.manual_seed(seed)
    return [GRUCell(input_size=d, hidden_size=d) if type1 else nn.GRU(input_size=d, hidden_size=d, num_layers=1) for _ in range(n)]


def run_program(prog, n_x, d, rnns, device, want, type1, seed, chain_kernels=True, x_src=None):
    """-> (outputs of the wanted instances, d_x, [grads of every GRU parameter]).
    x_src (int labels, one per x row): the x rows are gathered from a table with one random row per label, so rows with equal
    labels are equal, d_x is the gradient of that table, and the program is labelled (GruProgram.x_src: the chain path then
    computes the input gates once per distinct row)."""
    g = torch.Generator().manual_seed(seed)
    leaf = x = (torch.randn(n_x if x_src is None else int(np.max(x_src)) + 1, d, generator=g) * 0.5).to(device).requires_grad_(True)
    prog.__dict__.pop("_gi_shared", None)
    prog.__dict__.pop("x_src", None)
    if x_src is not None:
        x = leaf[torch.from_numpy(np.asarray(x_src)).long().to(device)]
        prog.x_src = np.asarray(x_src)
    mods = [m.to(device) for m in rnns]
    for m in mods:
        m.zero_grad()
    old = GC.CHAIN_KERNELS
    GC.CHAIN_KERNELS = chain_kernels
    try:
        out = GC.gru_chain(x, prog, mods, 0.1, type1, want)
        outs = [out[prog.inst[i].h0:prog.inst[i].h0 + prog.inst[i].n] for i in range(len(prog.inst))] if want is None else list(out)
        loss = 0
        for k, o in enumerate(outs):
            wgt = torch.randn(o.shape, generator=g).to(device)
            loss = loss + (o * wgt).sum() * (k + 1)
        loss.backward()
    finally:
        GC.CHAIN_KERNELS = old
    grads = [p.grad.detach().cpu().clone() for m in mods for p in m.parameters()]
    return [o.detach().cpu() for o in outs], leaf.grad.detach().cpu(), grads


def check_plan_invariants(prog):
    plan = prog.chain_plan()
    assert plan is not None
    T = _lib.CHAIN_TRACKS
    rows, panel = plan["rows"], plan["panel"]
    act = rows >= 0
    r = rows[act] & (_lib.CHAIN_HAS_PREV - 1)
    assert np.array_equal(np.sort(r), np.arange(prog.n_total)), "every row exactly once"
    has = ((rows >> 30) & 1).astype(bool) & act
    want_has = np.concatenate([(np.asarray(it.prev_idx) >= 0) if it.prev >= 0 else np.zeros(it.n, bool) for it in prog.inst])
    got_has = np.zeros(prog.n_total, dtype=bool)
    got_has[r] = has[act]
    assert np.array_equal(got_has, want_has)
    inst_of = np.concatenate([np.full(it.n, i) for i, it in enumerate(prog.inst)])
    for rnn, s0, ns, _ in panel.tolist():
        assert 0 < ns <= _lib.CHAIN_MAX_STEPS
        for s in range(s0, s0 + ns):
            e = rows[s]
            assert (e >= 0).any()
            ii = inst_of[e[e >= 0] & (_lib.CHAIN_HAS_PREV - 1)]
            assert (ii == ii[0]).all() and prog.inst[ii[0]].rnn == rnn and ii[0] == plan["step_inst"][s]
            assert bool(plan["any_prev"][s]) == bool((has[s]).any())
            for slot in np.nonzero(has[s])[0]:            # a carried state sits in the SAME track one listed step earlier
                assert s > s0 and rows[s - 1, slot] >= 0
                it = prog.inst[ii[0]]
                row = (e[slot] & (_lib.CHAIN_HAS_PREV - 1)) - it.h0
                prow = (rows[s - 1, slot] & (_lib.CHAIN_HAS_PREV - 1)) - prog.inst[it.prev].h0
                assert it.prev_idx[row] == prow
    return plan
