"""Round 6: fp32 products as THREE f16 MFMA products of the scaled two-way operand split (temp_amd/csrc/split_f16.hpp) -- the
magnitude keys, the keyed GEMM / input-gate / weight-gradient entry points and the window-chain kernels that hand the keys out.
Bars are those of the six-product bf16 kernels they replace (tests/test_gpu_parity_r2.py: 1e-6 of sum |a||b| for a product over K,
2e-6 for the chain's weight gradients), on the suite's wide data AND on range-stress data (row magnitudes 2^-30 .. 2^10, column
magnitudes 2^-20 .. 2^5): the split is only as good as its scales.  (GRU step of GRRGCNLayer.forward, models/RRGCN.py:84;
self-loop products models/RGCN.py:57.)"""
import numpy as np
import pytest
import torch

from temp_amd import _lib
from temp_amd import backend as TB

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")


@pytest.fixture(autouse=True)
def hip_backend():
    TB.set_backend(None)
    be = TB.get_backend()
    assert be.name == "hip"
    yield be
    TB.set_backend(None)


def _wide(shape, seed, scale):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * torch.exp(3.0 * torch.rand(shape, generator=g) - 1.5) * scale


def _pow2(n, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.pow(2.0, torch.randint(lo, hi + 1, (n,), generator=g).double()).float()


def _key(x):
    """the key of a tensor's elements: fp32 bits with the sign cleared (as int64 for comparisons)"""
    return (x.contiguous().view(torch.int32).to(torch.int64)) & 0x7FFFFFFF


@pytest.mark.parametrize("n,d", [(30011, 200), (17, 200), (5000, 8), (4097, 256), (1, 36)])
def test_absmax_and_gather_keys_exact_gpu(n, d, hip_backend):
    """Row / column keys = the bits of the row's / column's largest magnitude, exactly; the keyed gather writes what temp_gather_rows writes (zero rows included) and keys its OUTPUT."""
    be = hip_backend
    g = torch.Generator().manual_seed(n + d)
    x = (_wide((n, d), n, 1.0) * _pow2(n, -30, 10, d)[:, None]).to(DEV)
    x[n // 2] = 0.0
    rk, ck = be.absmax_keys(x)
    torch.cuda.synchronize()
    assert torch.equal(rk.to(torch.int64), _key(x.abs().max(dim=1).values))
    assert torch.equal(ck.to(torch.int64), _key(x.abs().max(dim=0).values))
    idx = torch.randint(-1, n, (2 * n + 3,), generator=g).to(torch.int32).to(DEV)
    out, rk2, ck2 = be.gather_rows_keys(x, idx)
    ref = be.gather_rows(x, idx)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert torch.equal(rk2.to(torch.int64), _key(ref.abs().max(dim=1).values))
    assert torch.equal(ck2.to(torch.int64), _key(ref.abs().max(dim=0).values))
    rk3, ck3 = be.absmax_keys(x)
    assert torch.equal(rk, rk3) and torch.equal(ck, ck3)


@pytest.mark.parametrize("M,K,N,trans_b", [
    (20000, 200, 600, True),      # the input gates' shape: weights-resident kernel, five column groups
    (20000, 600, 200, False),     # d_x: slab-staged kernel, one group of seven tiles
    (33333, 208, 132, True),      # ragged row tile, five tiles
    (17000, 72, 40, False),       # the shortest K the resident kernel takes, narrow output
    (16384, 24, 36, False),       # K below the resident kernel's range, one and a half slabs
    (16500, 200, 1000, True),     # wide output
    (29900, 500, 200, False),     # d_q = d_scores . all_entities over 500 entities: K = 4 (mod 8), slab-staged kernel reads A in quads
    (17001, 36, 72, True),        # K = 4 (mod 8), short
])
@pytest.mark.parametrize("data", ["wide", "range"])
def test_f16_split_gemm_vs_fp64_gpu(M, K, N, trans_b, data, hip_backend):
    """temp_linear_keys (row keys from temp_absmax_keys) against fp64, relative to sum |a||b|: bar 1e-6, the bar of the bf16 split
    (test_split_operand_gemm_vs_fp64); `range`: rows scaled by 2^-30 .. 2^10, weight columns by 2^-20 .. 2^5.  The launch must be
    an f16 kernel (no silent bf16 route), and twice the same bits."""
    be = hip_backend
    lib = _lib.load()
    a = _wide((M, K), 11, 1.0)
    b = _wide((N, K) if trans_b else (K, N), 12, 0.2)
    if data == "range":
        a = a * _pow2(M, -30, 10, 5)[:, None]
        cs = _pow2(N, -20, 5, 6)
        b = b * (cs[:, None] if trans_b else cs[None, :])
    a, b = a.to(DEV), b.to(DEV)
    rk, _ = be.absmax_keys(a, cols=False) if K <= 256 and K % 4 == 0 else (None, None)
    if rk is None:                                                    # wider than the key kernel: keys by torch (same definition)
        rk = _key(a.abs().max(dim=1).values).to(torch.int32)
    refused0, n0 = lib.temp_scratch_refused(), lib.temp_f16_launches()
    out = be.linear(a, b, trans_b, a_keys=rk)
    out2 = be.linear(a, b, trans_b, a_keys=rk)
    torch.cuda.synchronize()
    assert lib.temp_scratch_refused() == refused0 and lib.temp_f16_launches() == n0 + 2, "not an f16 launch"
    assert torch.equal(out, out2)
    rows = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M), torch.randint(0, M, (1400,))]).to(DEV)
    bd = b.double().t() if trans_b else b.double()
    ref = a[rows].double() @ bd
    sabs = a[rows].double().abs() @ bd.abs()
    err = ((out[rows].double() - ref).abs() / sabs.clamp_min(1e-300)).max().item()
    assert err < 1e-6, "f16-split GEMM %dx%dx%d (%s): error %.3e of sum|a||b|" % (M, K, N, data, err)
    # the same product without caller keys: the wide / deep ones take their own pass and give the same bits
    if N >= 512 or K >= 400:
        assert torch.equal(be.linear(a, b, trans_b), out)


def test_multi_problem_products_one_key_pass_and_one_reduction_gpu(hip_backend):
    """The loss's per-window products as multi-problem launches: d_q_b = d_scores_b . all_b (four problems, K = 500 entities, keys
    taken by ONE pass over all problems) and d_all_b = d_scores_b^T . q_b (adjacent outputs: the slices' partials side by side, ONE
    reduction) against fp64; the same bits twice, and the same bits whether the outputs are adjacent or apart."""
    be = hip_backend
    lib = _lib.load()
    Ms, N, D = [7474, 7001, 7474, 5000], 500, 200
    ds = [_wide((m, N), 20 + i, 1e-3).to(DEV) for i, m in enumerate(Ms)]
    ents = _wide((len(Ms) * N, D), 31, 0.3).to(DEV)
    q = [_wide((m, D), 40 + i, 0.5).to(DEV) for i, m in enumerate(Ms)]
    d_q = torch.empty(sum(Ms), D, device=DEV)
    n0 = lib.temp_f16_launches()
    be.linear_multi(ds, [ents[i * N:(i + 1) * N] for i in range(len(Ms))], False, d_q)
    torch.cuda.synchronize()
    assert lib.temp_f16_launches() == n0 + 1, "the K = 500 products did not take the f16 kernel"
    off = 0
    for i, m in enumerate(Ms):
        e = ents[i * N:(i + 1) * N].double()
        ref, sabs = ds[i].double() @ e, ds[i].double().abs() @ e.abs()
        err = ((d_q[off:off + m].double() - ref).abs() / sabs.clamp_min(1e-300)).max().item()
        assert err < 1e-6, "d_q of problem %d: %.3e" % (i, err)
        off += m
    again = torch.empty_like(d_q)
    be.linear_multi(ds, [ents[i * N:(i + 1) * N] for i in range(len(Ms))], False, again)
    assert torch.equal(again, d_q)
    d_all = torch.empty(len(Ms) * N, D, device=DEV)
    be.linear_tn_multi(ds, q, [d_all[i * N:(i + 1) * N] for i in range(len(Ms))])
    apart = [torch.empty(N, D, device=DEV) for _ in Ms]                 # outputs that do not follow each other: a reduction each
    be.linear_tn_multi(ds, q, apart)
    torch.cuda.synchronize()
    for i in range(len(Ms)):
        ref, sabs = ds[i].double().t() @ q[i].double(), ds[i].double().abs().t() @ q[i].double().abs()
        err = ((d_all[i * N:(i + 1) * N].double() - ref).abs() / sabs.clamp_min(1e-300)).max().item()
        assert err < 1e-6, "d_all of problem %d: %.3e" % (i, err)
        assert torch.equal(apart[i], d_all[i * N:(i + 1) * N])


def test_f16_split_gemm_nonfinite_and_zero_rows_gpu(hip_backend):
    """A non-finite weight poisons its column only; an all-zero row of A (key 0: the scale is clamped) gives an exact zero row;
    a row with one infinite element is non-finite, its neighbours untouched."""
    be = hip_backend
    g = torch.Generator().manual_seed(3)
    M, K, N = 20000, 200, 600
    a = (torch.rand(M, K, generator=g) + 0.5)
    b = torch.randn(N, K, generator=g) * 0.3
    a[77] = 0.0
    a[99, 5] = float("inf")
    a, b = a.to(DEV), b.to(DEV)
    rk, _ = be.absmax_keys(a, cols=False)
    clean = be.linear(a, b, True, a_keys=rk)
    assert torch.equal(clean[77], torch.zeros(N, device=DEV))
    assert not torch.isfinite(clean[99]).any()
    keep = torch.ones(M, dtype=torch.bool, device=DEV)
    keep[99] = False
    assert torch.isfinite(clean[keep]).all()
    bad = b.clone()
    bad[3, 5], bad[77, 5], bad[N - 1, 5] = float("inf"), float("-inf"), float("nan")
    got = be.linear(a, bad, True, a_keys=rk)
    cols = torch.ones(N, dtype=torch.bool, device=DEV)
    for c in (3, 77, N - 1):
        assert not torch.isfinite(got[keep][:, c]).any()
        cols[c] = False
    assert torch.equal(got[keep][:, cols], clean[keep][:, cols])


@pytest.mark.parametrize("rows,d", [((60000, 58000), 200), ((20000, 17003), 200), ((30001,), 200), ((9000, 8000, 7000, 6004), 200),
                                     ((20000, 20000), 104), ((12000, 11000), 248), ((30000, 27001), 128), ((17000,), 64), ((9000, 9000), 224)])
@pytest.mark.parametrize("data", ["wide", "range"])
def test_gru_grads_g4_keys_vs_fp64_gpu(rows, d, data, hip_backend):
    """temp_gru_grads_g4_keys (f16 weight gradients + d_x) against fp64: every weight / bias gradient and d_x to 2e-6 of
    sum |a||b| -- the bar of temp_gru_grads_g4 (tests/test_gpu_gate_grads.py) -- with the keys the chain backward would hand out
    (per row max |[dr dz dn_i]|, per column of g4) and x column keys both from the caller and taken inside; `range`: g4 rows scaled
    by 2^-30 .. 2^10, g4 and x columns by 2^-12 .. 2^6.  Bit-repeatable."""
    be = hip_backend
    lib = _lib.load()
    gen = torch.Generator(device="cpu").manual_seed(17 + len(rows) + d)
    mk = lambda n, w, s=1.0: (torch.randn(n, w, generator=gen) * s)
    xs, hd = [mk(n, d) for n in rows], [(torch.rand(n, d, generator=gen) * 2 - 1) for n in rows]       # hdec: decayed GRU states, |.| <= 1
    g4 = [mk(n, 4 * d, 0.1) * torch.exp(mk(n, 1) * 1.5) for n in rows]
    if data == "range":
        g4 = [t * _pow2(t.shape[0], -30, 10, 3 + i)[:, None] * _pow2(4 * d, -12, 6, 9 + i)[None, :] for i, t in enumerate(g4)]
        xs = [t * _pow2(d, -12, 6, 21 + i)[None, :] for i, t in enumerate(xs)]
    xs, hd, g4 = [t.to(DEV) for t in xs], [t.to(DEV) for t in hd], [t.to(DEV).contiguous() for t in g4]
    ws = [((torch.rand(3 * d, d, generator=gen) - 0.5) * 0.3).to(DEV) for _ in rows]
    assert be.gru_grads_g4_supported(list(rows), d, _lib.GRU_TORCH)
    row_keys = [_key(t[:, :3 * d].abs().max(dim=1).values).to(torch.int32) for t in g4]
    col_keys = [_key(t.abs().max(dim=0).values).to(torch.int32).contiguous() for t in g4]
    x_col = [be.absmax_keys(t, rows=False)[1].contiguous() for t in xs]
    dx = [torch.full((n, d), float("nan"), device=DEV) for n in rows]
    if len(rows) == 4:
        dx[2] = None
    n0 = lib.temp_f16_launches()
    got = be.gru_grads_g4(xs, hd, g4, ws, dx, row_keys=row_keys, col_keys=col_keys, x_col_keys=x_col)
    dx2 = [None if t is None else torch.full_like(t, float("nan")) for t in dx]
    again = be.gru_grads_g4(xs, hd, g4, ws, dx2, row_keys=row_keys, col_keys=col_keys)            # x column keys taken inside
    assert lib.temp_f16_launches() >= n0 + 2 * (1 + (1 if any(t is not None for t in dx) else 0))      # weight gradients + d_x, twice
    torch.cuda.synchronize()
    for k, n in enumerate(rows):
        G = g4[k].double()
        dgi, dgh = G[:, :3 * d], torch.cat([G[:, :2 * d], G[:, 3 * d:]], 1)
        want = (dgi.t() @ xs[k].double(), dgh.t() @ hd[k].double(), dgi.sum(0), dgh.sum(0))
        scale = (dgi.abs().t() @ xs[k].abs().double(), dgh.abs().t() @ hd[k].abs().double(), dgi.abs().sum(0), dgh.abs().sum(0))
        for a, b, w, sc in zip(got[k], again[k], want, scale):
            assert a.shape == w.shape and torch.isfinite(a).all()
            assert torch.equal(a, b), "bit-repeatable, and the same with the x keys taken inside"
            assert float(((a.double() - w).abs() / sc.clamp_min(1e-300)).max()) < 2e-6
        if dx[k] is not None:
            assert torch.equal(dx[k], dx2[k])
            wantx = dgi @ ws[k].double()
            scx = dgi.abs() @ ws[k].abs().double()
            assert float(((dx[k].double() - wantx).abs() / scx.clamp_min(1e-300)).max()) < 2e-6


@pytest.mark.parametrize("d", [200, 104, 32])
def test_chain_bwd_keys_gpu(d, hip_backend):
    """temp_gru_chain_bwd_g4_keys: the same g4 bit for bit, row keys = the bits of max |[dr dz dn_i]| of every row, column keys =
    per GRU the bits of every column's largest magnitude over that GRU's rows; twice the same."""
    from tests.chain_cases import random_program
    from tests.test_gpu_gate_grads import _chain_forward
    be = hip_backend
    if not be.gru_chain_keys_supported(d):
        pytest.skip("f16 chain kernels not selected for this width")
    prog, _ = random_program(7, n_chain=2, K=6, E=300, lo=100, hi=300)
    tabs, packs, b_hh, saved, N = _chain_forward(be, prog, d, 3)
    up = (torch.randn(N, d, generator=torch.Generator().manual_seed(5)) * 0.3).to(DEV)
    g4a = torch.full((N, 4 * d), float("nan"), device=DEV)
    be.gru_chain_bwd_g4(tabs, saved, [up], 0.1, _lib.GRU_TORCH, packs, b_hh, g4a)
    res = []
    for _ in range(2):
        g4 = torch.full((N, 4 * d), float("nan"), device=DEV)
        keys = (torch.full((N,), -1, dtype=torch.int32, device=DEV), torch.full((2 + tabs["n_panels"], 4 * d), -1, dtype=torch.int32, device=DEV))
        be.gru_chain_bwd_g4(tabs, saved, [up], 0.1, _lib.GRU_TORCH, packs, b_hh, g4, keys=keys)
        res.append((g4, keys[0].clone(), keys[1][:2].clone()))
    torch.cuda.synchronize()
    g4, rk, ck = res[0]
    assert torch.equal(g4, g4a)
    assert all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
    assert torch.equal(rk.to(torch.int64), _key(g4[:, :3 * d].abs().max(dim=1).values))
    for r, grp in enumerate(prog.groups):
        rows = g4[grp["h0"]:grp["h1"]]
        want = _key(rows.abs().max(dim=0).values)
        assert torch.equal(ck[grp["rnn"]].to(torch.int64), want), r


@pytest.mark.parametrize("gather", [False, True])
def test_input_gates_keys_vs_fp64_gpu(gather, hip_backend):
    """temp_gru_input_gates_gather_multi_keys (both directions' gates in one launch, optional row gather, keys by SOURCE row)
    against fp64: 1e-6 of sum |a||b| + the bias exactly added."""
    be = hip_backend
    g = torch.Generator().manual_seed(31)
    d = 200
    ns = (30000, 28000)
    xs = [(_wide((n, d), 40 + i, 1.0) * _pow2(n, -20, 8, 50 + i)[:, None]).to(DEV) for i, n in enumerate(ns)]
    w = [((torch.rand(3 * d, d, generator=g) - 0.5) * 0.3).to(DEV) for _ in ns]
    b = [((torch.rand(3 * d, generator=g) - 0.5) * 0.3).to(DEV) for _ in ns]
    keys = [be.absmax_keys(x, cols=False)[0] for x in xs]
    idx = [torch.randint(0, n, (n // 2 + 17,), generator=g).to(torch.int32).to(DEV) for n in ns] if gather else None
    rows = [t.shape[0] for t in idx] if gather else list(ns)
    outs = [torch.full((r, 3 * d), float("nan"), device=DEV) for r in rows]
    n0 = _lib.load().temp_f16_launches()
    be.gru_input_gates_multi(xs, w, b, _lib.GRU_TORCH, outs, x_idx=idx, x_keys=keys)
    assert _lib.load().temp_f16_launches() == n0 + 1
    torch.cuda.synchronize()
    for i in range(2):
        a = xs[i][idx[i].long()] if gather else xs[i]
        ref = a.double() @ w[i].double().t() + b[i].double()
        sabs = a.double().abs() @ w[i].double().abs().t() + b[i].double().abs()
        err = ((outs[i].double() - ref).abs() / sabs).max().item()
        assert err < 1e-6, err


def test_dropout_step_is_not_captured_gpu(hip_backend):
    """the self-loop dropout draws its seed on the host, per call; captured into a HIP graph the seed would be baked in
    and every replay would apply ONE mask.  The layer refuses the capture (callers fall back to eager launches); eager calls
    keep drawing a new mask each time (models/RGCN.py:57-59)."""
    from types import SimpleNamespace
    from temp_amd.rgcn import RGCNLayer
    import torch.nn.functional as F
    d = 32
    args = SimpleNamespace(inv_temperature=0.1, learnable_lambda=False, impute=False)
    layer = RGCNLayer(args, d, d, 4, 4, [0, 1], activation=F.relu, self_loop=True, dropout=0.5).to(DEV)
    layer.train()
    e = torch.randn(300, d, device=DEV)
    a, b = layer.conv_isolated(e), layer.conv_isolated(e)
    assert not torch.equal(a, b), "two eager calls drew the same mask"
    g = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match="cannot be captured"):
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            layer.conv_isolated(e)
    torch.cuda.synchronize()
    layer.eval()
    g2 = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        layer.conv_isolated(e)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g2, capture_error_mode="thread_local"):                     # (eval: no draw, capture is fine)
        out = layer.conv_isolated(e)
    g2.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("name", ["config1_static", "config3_post_ensemble"])
def test_config_step_bit_repeatable_and_atomic_free_gpu(name, hip_backend):
    """Verdict r5 item 5: the training steps of BASELINE configs 1 (StaticRGCN, baselines/StaticRGCN.py:36-89) and 3 (BiGRRGCN
    --post-ensemble, models/BiRRGCN.py:259-318) take no atomic scatter -- every gather adjoint is a segment sum over a static
    inverse (traced as k_segment_sum_rows; until round 6 they shared the trace name of k_scatter_add_rows) -- and two runs of the
    step give every gradient bit for bit."""
    import argparse
    import ctypes
    import bench
    lib = _lib.load()
    a = argparse.Namespace(no_graph=True)
    r = bench.other_config(name, a, DEV, lib, 10)
    names = {k for k in r["top_kernels"]}
    assert "k_scatter_add_rows" not in names, names

    # two eager steps of the same prepared batch, gradients compared bit for bit; every launch traced
    from temp_amd import synthetic
    from temp_amd.sampling import CorruptTriples
    import numpy as np
    if name == "config1_static":
        from temp_amd.static_rgcn import StaticRGCN
        w2 = synthetic.workload("S-icews14", seed=0)
        m2 = StaticRGCN(bench.make_args(w2, "SRGCN"), w2["num_ents"], w2["num_rels"], w2["snapshots"], w2["snapshots"], w2["snapshots"]).to(DEV)
        m2.sample_rng = np.random.default_rng(2)
        m2.corrupter = CorruptTriples(m2.args, w2["snapshots"], seed=5)
        wb2 = m2.prepare(synthetic.default_targets(w2["num_times"], w2["L"], w2["bsz"], 3))
        with torch.no_grad():
            m2.run_loss(wb2)
        cand = m2._last_plan[1]
        run = lambda: m2.run_loss(wb2, cand)
    else:
        from temp_amd.post_dynamic_rgcn import PostEnsembleBiDynamicRGCN
        w2 = synthetic.workload("S-icews0515", seed=0)
        args = bench.make_args(w2, "BiGRRGCN")
        args.post_ensemble = True
        m2 = PostEnsembleBiDynamicRGCN(args, w2["num_ents"], w2["num_rels"], w2["snapshots"], w2["snapshots"], w2["snapshots"]).to(DEV)
        m2.sample_rng = np.random.default_rng(2)
        m2.corrupter = CorruptTriples(m2.args, w2["snapshots"], seed=5)
        wb2 = m2.prepare(synthetic.default_targets(w2["num_times"], w2["L"], w2["bsz"], 3), w2["L"], True)
        fixed = [tuple(x.to(DEV) for x in smp) for smp in m2.draw_samples(wb2)]
        wts = [(torch.full((smp[0].shape[0], 1), 0.5, device=DEV), torch.full((smp[0].shape[0], 1), 0.5, device=DEV)) for smp in fixed]
        run = lambda: m2.run_loss(wb2, fixed, wts)
    grads = []
    for it in range(2):
        for p in m2.parameters():
            p.grad = None
        lib.temp_trace_begin(4096)
        loss = run()
        loss.backward()
        ids, ms, cnt = (ctypes.c_int32 * 4096)(), (ctypes.c_float * 4096)(), ctypes.c_int32(0)
        lib.temp_trace_end(ids, ms, 4096, ctypes.byref(cnt))
        traced = {lib.temp_trace_kernel_name(ids[i]).decode() for i in range(cnt.value)}
        assert "k_scatter_add_rows" not in traced and "k_segment_sum_rows" in traced, traced
        torch.cuda.synchronize()
        grads.append((loss.detach().clone(), [None if p.grad is None else p.grad.clone() for p in m2.parameters()]))
    assert torch.equal(grads[0][0], grads[1][0])
    for g0, g1 in zip(grads[0][1], grads[1][1]):
        assert (g0 is None) == (g1 is None)
        if g0 is not None:
            assert torch.equal(g0, g1)


def test_weight_gradient_beyond_32bit_span_gpu(hip_backend):
    """out = a^T . b over 10.8 M rows of 200 floats (an element span of 2.16e9 > 2^31: the HBM-regime window's loop-weight gradient):
    the split-operand kernel runs in row chunks with one ordered reduction (until round 6 this shape fell back to the fp32 MFMA
    kernel).  Against fp64 (chunked), 2e-7 of sum |a||b| -- the bar of test_split_operand_weight_gradient_vs_fp64 -- and twice the
    same bits."""
    be = hip_backend
    M, Ka, Nb = 10_800_000, 200, 200
    g = torch.Generator(device=DEV).manual_seed(5)
    a = torch.randn(M, Ka, generator=g, device=DEV)
    b = torch.randn(M, Nb, generator=g, device=DEV)
    a[-1] *= 64.0                                                  # the last row must count (a dropped tail chunk would show)
    out = be.linear_tn(a, b)
    out2 = be.linear_tn(a, b)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    ref = torch.zeros(Ka, Nb, dtype=torch.float64, device=DEV)
    sabs = torch.zeros(Ka, Nb, dtype=torch.float64, device=DEV)
    step = 1 << 20
    for r0 in range(0, M, step):
        ad, bd = a[r0:r0 + step].double(), b[r0:r0 + step].double()
        ref += ad.t() @ bd
        sabs += ad.abs().t() @ bd.abs()
    err = ((out.double() - ref).abs() / sabs).max().item()
    assert err < 2e-7, err


@pytest.mark.parametrize("Ms,N", [([184, 150, 184, 1], 7128), ([400, 400, 399, 400, 37, 400, 400, 400], 10488), ([1024, 3], 4100), ([184] * 8, 10488)])
def test_skinny_score_products_transposed_route_gpu(Ms, N, hip_backend):
    """scores_b = q_b . all_b^T with a few hundred query rows against thousands of entities runs as (all_b . q_b^T)^T on the
    weights-resident split-operand kernel, written transposed; every window keeps its own row count (ragged, not multiples of 4):
    against fp64 at the products' bar, nothing outside a problem's rows written, the same bits twice."""
    be = hip_backend
    K = 200
    q = [_wide((m, K), 50 + i, 0.5).to(DEV) for i, m in enumerate(Ms)]
    ents = [_wide((N, K), 70 + i, 0.3).to(DEV) for i in range(len(Ms))]
    out = torch.full((sum(Ms) + 2, N), 7.0, device=DEV)
    be.linear_multi(q, ents, True, out[:sum(Ms)])
    torch.cuda.synchronize()
    assert torch.all(out[sum(Ms):] == 7.0)                     # the rows behind the last problem are untouched
    off = 0
    for i, m in enumerate(Ms):
        ref = q[i].double() @ ents[i].double().t()
        sabs = q[i].double().abs() @ ents[i].double().abs().t()
        err = ((out[off:off + m].double() - ref).abs() / sabs.clamp_min(1e-300)).max().item()
        assert err < 1e-6, "scores of problem %d (%d x %d): %.3e" % (i, m, N, err)
        off += m
    again = torch.empty(sum(Ms), N, device=DEV)
    be.linear_multi(q, ents, True, again)
    assert torch.equal(again, out[:sum(Ms)])
    # the same windows' d_all_b = d_scores_b^T . q_b (few rows, thousands of output rows; the fp32 kernel: the split-operand one
    # measured slower at this depth, 3.40 against 3.06 ms for the config-3 step), one launch
    if len(Ms) <= 8 and min(Ms) >= 32:
        ds = [_wide((m, N), 90 + i, 1e-3).to(DEV) for i, m in enumerate(Ms)]
        d_all = torch.empty(len(Ms) * N, K, device=DEV)
        be.linear_tn_multi(ds, q, [d_all[i * N:(i + 1) * N] for i in range(len(Ms))])
        torch.cuda.synchronize()
        for i in range(len(Ms)):
            ref, sabs = ds[i].double().t() @ q[i].double(), ds[i].double().abs().t() @ q[i].double().abs()
            err = ((d_all[i * N:(i + 1) * N].double() - ref).abs() / sabs.clamp_min(1e-300)).max().item()
            assert err < 1e-6, "d_all of problem %d: %.3e" % (i, err)
