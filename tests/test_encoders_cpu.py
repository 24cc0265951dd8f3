"""Host-logic tests of the drop-in encoder classes (temp_amd.{rgcn,rrgcn,birrgcn}) on CPU.

The HIP kernels cannot run here, so a TEST-ONLY backend (tests/cpu_backend.py) that follows the
C-ABI contract through the same chunked edge views is installed; what is verified is everything
around the kernels: graph views, autograd wiring, the reference's aliasing / return conventions,
state_dict keys.  Expected values are the golden vectors recorded from the reference itself.
The same assertions run against the real kernels in tests/test_gpu_parity.py (-m gpu)."""
import argparse

import numpy as np
import pytest
import torch

import temp_amd
from temp_amd import backend as TB
from temp_amd.snapshot import Snapshot
from tests.cpu_backend import CpuTestBackend
from tests.encoder_cases import (check_G2, check_G4, check_G6, check_G7)


@pytest.fixture(autouse=True)
def cpu_backend():
    TB.set_backend(CpuTestBackend())
    yield
    TB.set_backend(None)


def test_product_has_no_cpu_path():
    TB.set_backend(None)
    h = torch.zeros(4, 8)
    with pytest.raises(Exception):
        be = TB.HipBackend()            # loads the library (fine) ...
        be.rgcn_isolated_fwd(h, torch.zeros(8, 8), None, 0)   # ... but CPU tensors are refused


def test_G2_layer():
    check_G2(torch.device("cpu"))


def test_G4_grrgcn_layer():
    check_G4(torch.device("cpu"))


def test_G6_rrgcn():
    check_G6(torch.device("cpu"))


def test_G7_birrgcn():
    check_G7(torch.device("cpu"))


def test_tiled_relation_view_reduces_like_plain_view():
    """The node-range-tiled by-relation view (GDELT-like: many edges per relation per snapshot) must give
    the same weight gradient as the plain relation-sorted view, through the kernels' chunk/slot/fix-up
    contract."""
    import numpy as np
    from temp_amd import snapshot as S
    rng = np.random.default_rng(3)
    n, E, R2, D, B = 600, 30000, 6, 16, 8
    src, dst = rng.integers(0, n, E), rng.integers(0, n, E)
    rel = rng.integers(0, R2 - 1, E)                    # last relation row has no edge
    g = S.Snapshot(n, src, dst, rel, np.arange(n))
    tiled = S.by_rel_view(g, R2)
    plain = S.build_view(g.rel, g.src, g.dst, R2, chunk=S._lib.CHUNK_REL)
    assert tiled["n_chunks"] > plain["n_chunks"] and tiled["n_fix"] == R2 - 1
    be = CpuTestBackend()
    f = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    h, w, lw, gy = f(n, D), f(R2, B * 2 * 2), f(D, D), f(n, D)
    dg = g.device_graph(torch.device("cpu"), R2)
    assert dg.views["by_rel"]["n_chunks"] == tiled["n_chunks"]
    out = be.rgcn_fwd(dg, h, None, w, lw, None, B, 0)
    d_w_tiled = be.rgcn_bwd(dg, h, out, gy, w, lw, False, B, 0)[1]
    dg.views["by_rel"] = plain
    d_w_plain = be.rgcn_bwd(dg, h, out, gy, w, lw, False, B, 0)[1]
    assert torch.allclose(d_w_tiled[:R2 - 1], d_w_plain[:R2 - 1], rtol=1e-4, atol=1e-4)


def test_self_loop_dropout_semantics_and_gradients():
    """Dropout of the self-loop message (models/RGCN.py:57-59; the reference's default --dropout is 0.1): training mode
    drops ~p of the loop-message entries and rescales the rest, eval mode is the deterministic layer, and the autograd
    gradients equal those of the explicit formula out = prop + mask * (h W_loop) with the mask the kernels hash."""
    import numpy as np
    from oracle import temp_oracle as O
    from temp_amd import snapshot as S
    from temp_amd.rgcn import RGCNLayer
    from tests.cpu_backend import drop_mask
    from tests.window_cases import make_args
    rng = np.random.default_rng(11)
    n, E, R2, D, B = 120, 900, 10, 32, 8
    g = S.Snapshot(n, rng.integers(0, n, E), rng.integers(0, n, E), rng.integers(0, R2, E), np.arange(n))
    layer = RGCNLayer(make_args(dropout=0.25), D, D, R2, B, list(range(5)), bias=True, activation=None, self_loop=True, dropout=0.25)
    h = torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32)).requires_grad_(True)
    layer.eval()
    ref = O.rgcn_layer(h.detach(), O.SnapGraph(n, g.src, g.dst, g.rel, g.gids), layer.weight.detach(), layer.loop_weight.detach(), B,
                       layer.h_bias.detach(), None)
    assert torch.allclose(layer.conv(g, h), ref, rtol=1e-5, atol=1e-5)
    layer.train()
    torch.manual_seed(5)
    out = layer.conv(g, h)
    torch.manual_seed(5)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())                  # what RGCNLayer._drop() drew
    m = drop_mask((0.25, seed), n, D)
    dropped = float((m == 0).float().mean())
    assert 0.2 < dropped < 0.3 and torch.allclose(m[m > 0], torch.tensor(1.0 / 0.75))
    loop = torch.mm(h.detach(), layer.loop_weight.detach())
    assert torch.allclose(out, ref - loop + loop * m, rtol=1e-5, atol=1e-5)
    gy = torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32))
    out.backward(gy)
    h2 = h.detach().clone().requires_grad_(True)
    lw2 = layer.loop_weight.detach().clone().requires_grad_(True)
    prop = O.rgcn_propagate(h2, O.SnapGraph(n, g.src, g.dst, g.rel, g.gids), layer.weight.detach(), B)
    (prop + layer.h_bias.detach() + torch.mm(h2, lw2) * m).backward(gy)
    assert torch.allclose(h.grad, h2.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(layer.loop_weight.grad, lw2.grad, rtol=1e-4, atol=1e-4)


def test_gru_step_injective_previous_state_adjoint():
    """TF.gru_step with prev_inv (the inverse of an injective previous-state map): the gradient of `prev` is one gather through the
    inverse instead of a zero fill + scatter-add -- same values; a map with a repeated row has no inverse (None)."""
    from temp_amd import functional as TF
    rng = np.random.default_rng(3)
    n, n_prev, d = 40, 90, 16
    idx = rng.permutation(n_prev)[:n].astype(np.int64)
    idx[rng.integers(0, n, 6)] = -1                                   # rows without a previous state
    assert TF.injective_inverse(np.array([3, 5, 3]), 8, "cpu") is None
    inv = TF.injective_inverse(idx, n_prev, "cpu")
    ok = idx >= 0
    assert inv.dtype == torch.int32 and int((inv >= 0).sum()) == int(ok.sum())
    assert np.array_equal(idx[inv.numpy()[idx[ok]]], idx[ok])
    f = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    w = [f(3 * d, d) * 0.3, f(3 * d, d) * 0.3, f(3 * d) * 0.1, f(3 * d) * 0.1]
    x, prev, dt, up = f(n, d), f(n_prev, d), torch.from_numpy(rng.integers(1, 5, (n, 1)).astype(np.float32)), f(n, d)
    pidx = torch.from_numpy(idx.astype(np.int32))
    grads = []
    for pinv in (None, inv):
        xs, ps = x.clone().requires_grad_(True), prev.clone().requires_grad_(True)
        h = TF.gru_step(xs, ps, dt, *w, 0.1, None, pidx, prev_inv=pinv)
        h.backward(up)
        grads.append((h.detach(), xs.grad, ps.grad))
    for a, b in zip(*grads):
        assert torch.equal(a, b)
    untouched = np.setdiff1d(np.arange(n_prev), idx[ok])
    assert float(grads[1][2][untouched].abs().max()) == 0.0
