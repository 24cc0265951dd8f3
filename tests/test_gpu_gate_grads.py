"""Round 5: the window-chain backward writes its gate gradients ONCE, g4 = [dr | dz | dn_i | dn_h] (temp_gru_chain_bwd_g4), and
temp_gru_grads_g4 forms both weight gradients, the bias gradients and d_x from that one matrix (k_gru_wgrad: LDS transpose reads,
column map) -- the backward of the GRU step of GRRGCNLayer.forward (models/RRGCN.py:84), nn.GRU gate layout."""
import ctypes

import numpy as np
import pytest
import torch

from temp_amd import _lib
from temp_amd import backend as TB
from temp_amd import gru_chain as GC

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")


@pytest.fixture(autouse=True)
def hip_backend():
    TB.set_backend(None)
    be = TB.get_backend()
    assert be.name == "hip"
    yield be
    TB.set_backend(None)


def _chain_forward(be, prog, d, seed):
    """One forward of a chain program through the persistent kernels -> (tabs, packs, b_hhs, saved, N)."""
    g = torch.Generator().manual_seed(seed)
    N = prog.n_total
    tabs = prog.chain_tables(DEV, None)
    assert tabs is not None
    w_hh = [((torch.rand(3 * d, d, generator=g) - 0.5) * 0.3).to(DEV) for _ in range(2)]
    b_hh = [((torch.rand(3 * d, generator=g) - 0.5) * 0.3).to(DEV) for _ in range(2)]
    gi = (torch.randn(N, 3 * d, generator=g) * 0.5).to(DEV)
    packs = be.gru_chain_pack_multi(w_hh)
    H = torch.empty(N, d, device=DEV)
    saved = torch.empty(5, N, d, device=DEV)
    be.gru_chain_fwd(tabs, gi, 0.1, _lib.GRU_TORCH, packs, b_hh, H, saved)
    return tabs, packs, b_hh, saved, N


@pytest.mark.parametrize("d", [200, 104, 32])
def test_chain_bwd_g4_equals_dgi_dgh_bitwise_gpu(d, hip_backend):
    """g4[:, :3d] == dgi, g4[:, :2d] == dgh[:, :2d], g4[:, 3d:] == dgh[:, 2d:], bit for bit (the same kernel, another store)."""
    from tests.chain_cases import random_program
    be = hip_backend
    prog, _ = random_program(7, n_chain=2, K=6, E=300, lo=100, hi=300)
    tabs, packs, b_hh, saved, N = _chain_forward(be, prog, d, 3)
    up = (torch.randn(N, d, generator=torch.Generator().manual_seed(5)) * 0.3).to(DEV)
    dgi = torch.full((N, 3 * d), float("nan"), device=DEV)
    dgh = torch.full((N, 3 * d), float("nan"), device=DEV)
    be.gru_chain_bwd(tabs, saved, [up], 0.1, _lib.GRU_TORCH, packs, b_hh, dgi, dgh)
    g4 = torch.full((N, 4 * d), float("nan"), device=DEV)
    be.gru_chain_bwd_g4(tabs, saved, [up], 0.1, _lib.GRU_TORCH, packs, b_hh, g4)
    torch.cuda.synchronize()
    assert torch.isfinite(g4).all()
    assert torch.equal(g4[:, :3 * d], dgi)
    assert torch.equal(g4[:, :2 * d], dgh[:, :2 * d]) and torch.equal(g4[:, 3 * d:], dgh[:, 2 * d:])


@pytest.mark.parametrize("rows,d", [((60000, 58000), 200), ((20000, 17003), 200), ((30001,), 200), ((9000, 8000, 7000, 6004), 200),
                                     ((20000, 20000), 104), ((9000, 9000, 9000), 136), ((12000, 11000), 248), ((40000,), 40),
                                     ((30000, 27001), 128), ((17000,), 64), ((9000, 9000), 224)])      # d % 32 == 0: one more column tile for the ones column
def test_gru_grads_g4_vs_fp64_gpu(rows, d, hip_backend):
    """temp_gru_grads_g4 against fp64 products of the same operands: every weight / bias gradient to 2e-6 of sum |a||b| (the bar of
    the split-operand kernels), d_x equal to the d_x of the dgi / dgh call on the first 3d columns; ragged row counts (a last slab
    of 11 rows, slices of different lengths), one to four GRUs, widths with mixed / left-over / tail decompositions; two runs
    bit-identical (no atomics: ordered sum over the row slices)."""
    be = hip_backend
    gen = torch.Generator(device="cpu").manual_seed(17 + len(rows) + d)
    mk = lambda n, w, s=1.0: (torch.randn(n, w, generator=gen) * s).to(DEV)
    xs, hd = [mk(n, d) for n in rows], [mk(n, d) for n in rows]
    g4 = [mk(n, 4 * d, 0.1) * torch.exp(mk(n, 1) * 1.5) for n in rows]                 # wide dynamic range across rows
    ws = [((torch.rand(3 * d, d, generator=gen) - 0.5) * 0.3).to(DEV) for _ in rows]
    assert be.gru_grads_g4_supported(list(rows), d, _lib.GRU_TORCH)
    dx = [torch.full((n, d), float("nan"), device=DEV) for n in rows]
    if len(rows) == 4:
        dx[2] = None
    got = be.gru_grads_g4(xs, hd, g4, ws, dx)
    dx2 = [None if t is None else torch.full_like(t, float("nan")) for t in dx]
    again = be.gru_grads_g4(xs, hd, g4, ws, dx2)
    torch.cuda.synchronize()
    for k, n in enumerate(rows):
        G = g4[k].double()
        dgi, dgh = G[:, :3 * d], torch.cat([G[:, :2 * d], G[:, 3 * d:]], 1)
        want = (dgi.t() @ xs[k].double(), dgh.t() @ hd[k].double(), dgi.sum(0), dgh.sum(0))
        scale = (dgi.abs().t() @ xs[k].abs().double(), dgh.abs().t() @ hd[k].abs().double(), dgi.abs().sum(0), dgh.abs().sum(0))
        for a, b, w, sc in zip(got[k], again[k], want, scale):
            assert a.shape == w.shape and torch.isfinite(a).all()
            assert torch.equal(a, b), "bit-repeatable"
            assert float(((a.double() - w).abs() / sc.clamp_min(1e-30)).max()) < 2e-6
        if dx[k] is not None:
            assert torch.equal(dx[k], dx2[k])
            wantx = dgi @ ws[k].double()
            scx = dgi.abs() @ ws[k].abs().double()
            assert float(((dx[k].double() - wantx).abs() / scx.clamp_min(1e-30)).max()) < 2e-6


def test_gru_grads_g4_refuses_what_it_does_not_take_gpu(hip_backend):
    be = hip_backend
    assert not be.gru_grads_g4_supported([900, 700], 200, _lib.GRU_TORCH)              # few rows: the fp32 kernels
    assert be.gru_grads_g4_supported([30000], 128, _lib.GRU_TORCH)                     # d % 32 == 0 (the reference's default width): an extra column tile carries the bias sums
    assert not be.gru_grads_g4_supported([30000], 256, _lib.GRU_TORCH)                 # nine column tiles with the ones column
    assert not be.gru_grads_g4_supported([30000], 100, _lib.GRU_TORCH)                 # d % 8
    assert not be.gru_grads_g4_supported([30000], 200, _lib.GRU_TYPE1)
    lib = _lib.load()
    old = lib.temp_set_option(0, 0)                                                     # TEMP_OPT_MFMA_BF16X3 off: fp32 MFMA kernels everywhere
    try:
        assert not be.gru_grads_g4_supported([30000], 200, _lib.GRU_TORCH)
    finally:
        lib.temp_set_option(0, old)


@pytest.mark.parametrize("want", [None, (11, 5)])
def test_chain_program_gate_grads_once_vs_two_matrices_gpu(want, hip_backend):
    """The whole chain autograd node with GATE_GRADS_ONCE on and off: same outputs bit for bit (the forward is untouched), d_x
    bit for bit (the same panel kernel on the same values), weight / bias gradients to fp32 summation order."""
    from tests.chain_cases import make_rnns, random_program, run_program
    prog, n_x = random_program(23, n_chain=2, K=6, E=2000, lo=1500, hi=2000)       # ~10 000 rows per GRU: the one-launch paths
    rnns = make_rnns(2, 200, False, 9)
    res = []
    lib = _lib.load()
    for once in (True, False):
        GC.GATE_GRADS_ONCE = once
        try:
            lib.temp_trace_begin(512)
            r = run_program(prog, n_x, 200, rnns, DEV, want, False, 4)
            ids, ms, cnt = (ctypes.c_int32 * 512)(), (ctypes.c_float * 512)(), ctypes.c_int32(0)
            lib.temp_trace_end(ids, ms, 512, ctypes.byref(cnt))
            names = {lib.temp_trace_kernel_name(ids[i]).decode() for i in range(cnt.value)}
            assert ("k_gru_wgrad" in names) == once, names
        finally:
            GC.GATE_GRADS_ONCE = True
        res.append(r)
    (o1, dx1, g1), (o0, dx0, g0) = res
    assert all(torch.equal(a, b) for a, b in zip(o1, o0))
    assert torch.equal(dx1, dx0)
    for a, b in zip(g1, g0):
        assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))
