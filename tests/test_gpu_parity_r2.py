"""GPU parity tests added in round 2 (-m gpu): the HIP path against the CPU ORACLE (oracle/temp_oracle.py, pinned by the
reference's golden vectors) at the sizes BASELINE.json names, and the integer / scorer pieces against the reference's own
recorded outputs.

  * full S-gdelt windows (D = 200, 100 bases, L = 15, bidirectional, --rec-only-last-layer): target embeddings and
    gradients of the batched HIP step vs the oracle's dense-history restatement of the reference op sequence;
  * the same workload through the self-attention encoder (config 5) vs O.sa_encode;
  * temp_filtered_rank vs the oracle's mask / sigmoid / sort formulation on tie-free scores (bit-exact);
  * temp_amd.scores and the folded-query kernels vs G9_scores.npz (recorded from utils/scores.py).
Tolerance: 1e-5 relative fp32 with a small absolute floor on outputs; gradients that sum over ~10^5 rows get 1e-4."""
import argparse
import os

import numpy as np
import pytest
import torch

from oracle import temp_oracle as O
from temp_amd import backend as TB
from tests.golden_util import T, assert_close, load

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(autouse=True)
def hip_backend():
    TB.set_backend(None)
    be = TB.get_backend()
    assert be.name == "hip"
    yield be


def _oracle_model(model, w, module, te=False):
    """The oracle's parameter dict in FLOAT64.  Measured (tools/oracle_precision.py): at this size the fp32 oracle's own
    d ent_embeds is off by up to 1.8e-3 of its maximum against the same op sequence in fp64 (sequential index_add over Zipf
    hubs), i.e. the fp32 CPU path is too noisy to be the yardstick for a 1e-4 gradient check; its fp64 evaluation is the
    reference's arithmetic without the accumulation noise."""
    sd = {k: v.detach().cpu().double().clone() for k, v in model.state_dict().items()}
    cfg = dict(module=module, n_bases=w["B"], inv_temperature=0.1, rec_only_last_layer=True, use_time_embedding=te)
    om = O.model_from_state_dict(sd, cfg)
    gd = {t: O.SnapGraph(g.n, g.src, g.dst, g.rel, g.gids) for t, g in w["snapshots"].items()}
    return om, cfg, gd


def _assert_grad_close(got, ref, what, rtol=1e-4, atol_frac=2e-5, max_bad=1e-4, frob=1e-4, frob_clean=6e-6):
    """Gradient check that tolerates ReLU-kink flips.  Layer 2 of these encoders ends in a ReLU over ~10^7 pre-activations
    per step; a handful of them lie within fp32 rounding of 0, where the fp32 HIP path and the fp64 oracle legitimately take
    different sides (measured with tools/attn_grad_probe.py: ONE flipped element moves d h_bias by 1.6e-2 and 2 of the 16 000
    entries of d layer_2.weight by 5 % of their maximum, while d q/k/v_linear, which do not pass through the kink, agree
    to 1e-6).  Each flip touches few gradient entries, so a tensor passes when EITHER its relative Frobenius error is below
    `frob_clean` (fp32 round-off of sums over ~6e4 rows, no flip in its path) OR at most `max_bad` of its entries are outside
    (rtol, atol_frac * max|ref|) elementwise and the whole tensor within `frob` in Frobenius norm.
    Thresholds = measured x 3 (round 3, all 87 full-size checks of the suite printed with `pytest -s`, gpurun_out/grad_stats.txt):
    every tensor had 0.0000 % of its entries out of tolerance and a relative Frobenius error between 2.4e-7 and 1.9e-6 -- no
    flip occurs with the committed seeds, so the flip branch (max_bad = 0.01 %, frob = 1e-4) is head-room, not slack in use.
    A wrong scale on 2 % of the rows of a gradient fails both branches by orders of magnitude."""
    g = got.detach().cpu().double()
    r = ref.detach().cpu().double()
    assert g.shape == r.shape, what
    err = (g - r).abs()
    tol = atol_frac * float(r.abs().max()) + rtol * r.abs()
    bad = float((err > tol).double().mean())
    rel_f = float(err.norm() / r.norm().clamp_min(1e-30))
    GRAD_STATS.append((what, bad, rel_f))
    print("grad-check %-48s out-of-tolerance %.4f%%  relative Frobenius error %.2e" % (what, 100 * bad, rel_f))
    assert rel_f <= frob_clean or (bad <= max_bad and rel_f <= frob), "%s: %.2f%% of the entries out of tolerance, relative Frobenius error %.2e" % (what, 100 * bad, rel_f)


GRAD_STATS = []     # (what, fraction out of tolerance, relative Frobenius error) of every full-size gradient check of the session (-s prints them)


def _upstream(sizes, D, seed):
    """Fixed pseudo-random upstream gradient per window (a plain sum would hide sign / permutation errors)."""
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(n, D, generator=g) for n in sizes]


@pytest.mark.parametrize("workload,n_windows,dim", [("S-gdelt", 3, None), ("S-icews0515", 3, None), ("S-gdelt", 2, 128), ("S-gdelt", 2, 64)])
def test_full_size_windows_vs_oracle_gpu(workload, n_windows, dim):
    """BASELINE's headline shape: full windows of the S-gdelt (and the ICEWS05-15-shaped) workload through the batched HIP
    step (distinct snapshots once, table layer, one GRU chain program) against the oracle (0.3 s per window on the CPU).
    dim = 128 / 64: the reference's shipped grid (embed = n_bases, 1 x 1 blocks): other tile counts of every MFMA kernel, the
    permute-based edge kernels, the fp32 weight-gradient kernel."""
    from temp_amd import synthetic
    w = synthetic.workload(workload, seed=0)
    if dim is not None:
        w["D"], w["B"] = dim, dim
    targets = sorted(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 0)[:n_windows], reverse=True)
    _bi_windows_vs_oracle(w, targets, workload)


def test_hbm_window_shape_vs_oracle_gpu():
    """bench.py's extra.hbm_window (= --workload S-hbm-window) at reduced size: the same generator (power-law snapshots of 2^k
    nodes / 2^(k+4) edges, 230 relations: the relation table beyond LDS, hub nodes with multi-chunk segments and relation runs),
    the same BiGRRGCN batched step with bsz = 1 -- here k = 12 and L = 5 (9 snapshot visits) so that the fp64 oracle finishes in
    seconds -- target embeddings and every gradient."""
    from temp_amd import synthetic
    k, L, R = 12, 5, 230
    N, E = 1 << k, 1 << (k + 4)
    snaps = synthetic.make_snapshots(N, R, E, N, 2 * L - 1, seed=0)
    w = dict(name="S-hbm-window", num_ents=N, num_rels=R, edges_per_snap=E, nodes_per_snap=N, num_times=2 * L - 1, D=200, B=100, L=L, bsz=1,
             module="BiGRRGCN", snapshots=snaps)
    _bi_windows_vs_oracle(w, [L - 1], "S-hbm-window(k=12, L=5)")


def _bi_windows_vs_oracle(w, targets, workload):
    import bench
    model = bench.build_model(w, DEV)
    L, D = w["L"], w["D"]
    # ---- HIP path ---------------------------------------------------------------------------------------------------
    wb = model.prepare(targets, L, train=False)
    assert wb.batched and wb.program is not None
    out = model.run(wb)[0]
    pieces = list(out.split(wb.target.sizes))
    ups = _upstream(wb.target.sizes, D, 7)
    sum((p * u.to(DEV)).sum() for p, u in zip(pieces, ups)).backward()
    torch.cuda.synchronize()
    # ---- oracle (reference op sequence, dense re-zeroed history) ------------------------------------------------------
    om, cfg, gd = _oracle_model(model, w, w["module"])
    times = sorted(gd.keys())
    leaves = O.leaf_tensors(om)
    for v in leaves.values():
        v.requires_grad_(True)
    tf, tb = O.get_batch_graph_list_bi(targets, L, times)
    Hf = O.bi_pre_forward(om, cfg, gd, tf, L, True)
    Hb = O.bi_pre_forward(om, cfg, gd, tb, L, False)
    want = O.bi_target_embeds(om, cfg, Hf, Hb, [gd[t] for t in targets], tf[-1], L)
    sum((p * u.double()).sum() for p, u in zip(want, ups)).backward()
    for i, (a, b) in enumerate(zip(pieces, want)):
        assert_close(a, b, 1e-5, 3e-6, "%s window %d target embeddings" % (workload, i))
    enc = model.ent_encoder
    l2o = om["ent_encoder"]["layer_2"]
    checks = [("ent_embeds", model.ent_embeds.grad, om["ent_embeds"].grad),
              ("layer_1.weight", enc.layer_1.weight.grad, om["ent_encoder"]["layer_1"]["weight"].grad),
              ("layer_1.loop_weight", enc.layer_1.loop_weight.grad, om["ent_encoder"]["layer_1"]["loop_weight"].grad),
              ("layer_2.weight", enc.layer_2.weight.grad, l2o["weight"].grad),
              ("layer_2.loop_weight", enc.layer_2.loop_weight.grad, l2o["loop_weight"].grad)]
    for name, rnn in (("forward_rnn", enc.layer_2.forward_rnn), ("backward_rnn", enc.layer_2.backward_rnn)):
        q = l2o[name][0]
        checks += [("%s.w_hh" % name, rnn.weight_hh_l0.grad, q["w_hh"].grad), ("%s.w_ih" % name, rnn.weight_ih_l0.grad, q["w_ih"].grad),
                   ("%s.b_hh" % name, rnn.bias_hh_l0.grad, q["b_hh"].grad), ("%s.b_ih" % name, rnn.bias_ih_l0.grad, q["b_ih"].grad)]
    for name, got, ref in checks:
        assert got is not None and ref is not None, name
        _assert_grad_close(got, ref, "%s d %s" % (workload, name))


@pytest.mark.parametrize("workload,n_windows", [("S-icews14", 3), ("S-gdelt", 2)])
def test_full_size_training_loss_vs_oracle_gpu(workload, n_windows):
    """A full-size TRAINING step with the ComplEx link-prediction loss against the oracle in fp64 -- BASELINE config 2 at its own
    size (S-icews14: uni-directional GRRGCN, L = 8, D = 200, 7 128 entities, 460 relation rows; models/DynamicRGCN.py:176-194,
    models/TKG_Module.py:202-213, utils/scores.py:27-44) and the headline S-gdelt BiGRRGCN step (models/BiDynamicRGCN.py:123-144):
    50 % target-edge subsample (fixed draw), negatives with the truth in column 0 (fixed draw), all-entity pass, fused loss node.
    Loss to 2e-5 relative, target embeddings to 1e-5, every parameter gradient by _assert_grad_close."""
    import bench
    from temp_amd import synthetic
    w = synthetic.workload(workload, seed=0)
    model = bench.build_model(w, DEV)
    bi = w["module"].startswith("Bi")
    targets = sorted(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 0)[:n_windows], reverse=True)
    L, N = w["L"], w["num_ents"]
    rng = np.random.default_rng(11)
    edge_ids, samples = [], []
    NEG = 60
    for t in targets:
        g = w["snapshots"][t]
        E = g.number_of_edges()
        edge_ids.append(np.sort(rng.choice(E, E // 2, replace=False)))
        P = min(E, 400)
        pos = rng.choice(E, P, replace=False)
        trip = torch.from_numpy(np.stack([g.src[pos], g.rel[pos], g.dst[pos]], axis=1)).long()
        nt = torch.from_numpy(rng.integers(0, N, (P, 1 + NEG)))
        nh = torch.from_numpy(rng.integers(0, N, (P, 1 + NEG)))
        nt[:, 0] = torch.from_numpy(g.gids[g.dst[pos]])
        nh[:, 0] = torch.from_numpy(g.gids[g.src[pos]])
        samples.append((trip, nt, nh))
    # ---- HIP path: the model's own forward() with the draws injected ------------------------------------------------------------
    loss = model(torch.tensor(targets), target_edge_ids=edge_ids, samples=samples)
    loss.backward()
    torch.cuda.synchronize()
    # ---- oracle, fp64 -----------------------------------------------------------------------------------------------------------
    om, cfg, gd = _oracle_model(model, w, w["module"])
    times = sorted(gd.keys())
    leaves = O.leaf_tensors(om)
    for v in leaves.values():
        v.requires_grad_(True)
    tgt = [O.edge_subgraph(gd[t], torch.from_numpy(e)) for t, e in zip(targets, edge_ids)]
    fn = O.bi_forward_loss if bi else O.uni_forward_loss
    want, _ = fn(om, cfg, gd, targets, times, L, tgt, samples, score="complex")
    want.backward()
    print("full-size %s training loss: HIP %.7f  oracle(fp64) %.7f  rel diff %.2e" % (workload, loss.item(), want.item(), abs(loss.item() - want.item()) / abs(want.item())))
    assert abs(loss.item() - want.item()) <= 2e-5 * abs(want.item()), (loss.item(), want.item())
    enc = model.ent_encoder
    l2o = om["ent_encoder"]["layer_2"]
    checks = [("ent_embeds", model.ent_embeds.grad, om["ent_embeds"].grad), ("rel_embeds", model.rel_embeds.grad, om["rel_embeds"].grad),
              ("layer_1.weight", enc.layer_1.weight.grad, om["ent_encoder"]["layer_1"]["weight"].grad),
              ("layer_1.loop_weight", enc.layer_1.loop_weight.grad, om["ent_encoder"]["layer_1"]["loop_weight"].grad),
              ("layer_2.weight", enc.layer_2.weight.grad, l2o["weight"].grad),
              ("layer_2.loop_weight", enc.layer_2.loop_weight.grad, l2o["loop_weight"].grad)]
    for name in (("forward_rnn", "backward_rnn") if bi else ("rnn",)):
        rnn, q = getattr(enc.layer_2, name), l2o[name][0]
        checks += [("%s.w_hh" % name, rnn.weight_hh_l0.grad, q["w_hh"].grad), ("%s.w_ih" % name, rnn.weight_ih_l0.grad, q["w_ih"].grad),
                   ("%s.b_hh" % name, rnn.bias_hh_l0.grad, q["b_hh"].grad), ("%s.b_ih" % name, rnn.bias_ih_l0.grad, q["b_ih"].grad)]
    for name, got, ref in checks:
        assert got is not None and ref is not None, name
        # measured: 0 entries out of tolerance, relative Frobenius error 5e-7 .. 1.9e-6 (no ReLU-kink flip reaches a gradient
        # through the loss at these sizes) -> thresholds = measured x 3
        _assert_grad_close(got, ref, "%s+loss d %s" % (workload, name), max_bad=0.0, frob=6e-6, frob_clean=6e-6)


def test_full_size_default_flags_training_loss_vs_oracle_gpu():
    """BASELINE config 2 with the reference's DEFAULT flags (no --rec-only-last-layer: both layers recurrent,
    models/RRGCN.py:179-181, F7 aliasing) at its own size -- S-icews14, L = 8, D = 200 -- through the one-node position loop
    (temp_amd/rec_stack.py: RGCN_1 of all visits hoisted, both GRUs' weight gradients in one launch) against the oracle in fp64:
    training loss with the ComplEx scorer, fixed subsample and negatives, and every parameter gradient incl. layer 1's GRU."""
    import bench
    from temp_amd import synthetic
    from temp_amd.dynamic_rgcn import DynamicRGCN
    w = synthetic.workload("S-icews14", seed=0)
    args = bench.make_args(w, "GRRGCN")
    args.rec_only_last_layer = False
    torch.manual_seed(3)
    model = DynamicRGCN(args, w["num_ents"], w["num_rels"], w["snapshots"], w["snapshots"], w["snapshots"]).to(DEV)
    targets = sorted(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 0)[:3], reverse=True)
    L, N = w["L"], w["num_ents"]
    rng = np.random.default_rng(13)
    edge_ids, samples = [], []
    for t in targets:
        g = w["snapshots"][t]
        E = g.number_of_edges()
        edge_ids.append(np.sort(rng.choice(E, E // 2, replace=False)))
        P = min(E, 300)
        pos = rng.choice(E, P, replace=False)
        trip = torch.from_numpy(np.stack([g.src[pos], g.rel[pos], g.dst[pos]], axis=1)).long()
        nt, nh = torch.from_numpy(rng.integers(0, N, (P, 41))), torch.from_numpy(rng.integers(0, N, (P, 41)))
        nt[:, 0] = torch.from_numpy(g.gids[g.dst[pos]])
        nh[:, 0] = torch.from_numpy(g.gids[g.src[pos]])
        samples.append((trip, nt, nh))
    wb = model.prepare(torch.tensor(targets), L, True, edge_ids)
    assert wb.stack and not wb.batched
    loss = model.run_loss(wb, samples)
    loss.backward()
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu().double().clone() for k, v in model.state_dict().items()}
    cfg = dict(module="GRRGCN", n_bases=w["B"], inv_temperature=0.1, rec_only_last_layer=False, use_time_embedding=False)
    om = O.model_from_state_dict(sd, cfg)
    gd = {t: O.SnapGraph(g.n, g.src, g.dst, g.rel, g.gids) for t, g in w["snapshots"].items()}
    for v in O.leaf_tensors(om).values():
        v.requires_grad_(True)
    tgt = [O.edge_subgraph(gd[t], torch.from_numpy(e)) for t, e in zip(targets, edge_ids)]
    want, _ = O.uni_forward_loss(om, cfg, gd, targets, sorted(gd.keys()), L, tgt, samples, score="complex")
    want.backward()
    print("full-size default-flags training loss: HIP %.7f  oracle(fp64) %.7f" % (loss.item(), want.item()))
    assert abs(loss.item() - want.item()) <= 2e-5 * abs(want.item()), (loss.item(), want.item())
    enc, oe = model.ent_encoder, om["ent_encoder"]
    checks = [("ent_embeds", model.ent_embeds.grad, om["ent_embeds"].grad), ("rel_embeds", model.rel_embeds.grad, om["rel_embeds"].grad)]
    for ln in ("layer_1", "layer_2"):
        lay, q = getattr(enc, ln), oe[ln]
        checks += [(ln + ".weight", lay.weight.grad, q["weight"].grad), (ln + ".loop_weight", lay.loop_weight.grad, q["loop_weight"].grad)]
        r = q["rnn"][0]
        checks += [(ln + ".rnn.w_hh", lay.rnn.weight_hh_l0.grad, r["w_hh"].grad), (ln + ".rnn.w_ih", lay.rnn.weight_ih_l0.grad, r["w_ih"].grad),
                   (ln + ".rnn.b_hh", lay.rnn.bias_hh_l0.grad, r["b_hh"].grad), (ln + ".rnn.b_ih", lay.rnn.bias_ih_l0.grad, r["b_ih"].grad)]
    for name, got, ref in checks:
        assert got is not None and ref is not None, name
        _assert_grad_close(got, ref, "default flags d " + name, max_bad=1e-4, frob=1e-4, frob_clean=2e-5)


@pytest.mark.parametrize("workload,n_windows", [("S-gdelt", 2)])
def test_full_size_attention_windows_vs_oracle_gpu(workload, n_windows):
    """Config 5 at the headline size: BiSelfAttentionRGCN (8-head attention of every target node over its window history)
    on full S-gdelt windows vs the oracle's dense (n, T, D) formulation (O.sa_encode)."""
    import bench
    from temp_amd import synthetic
    w = synthetic.workload(workload, seed=0)
    model = bench.build_model(w, DEV, "attention")
    targets = sorted(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 0)[:n_windows], reverse=True)
    L, D = w["L"], w["D"]
    per_graph, wb, tables = model.encode(torch.tensor(targets), L, train=False)
    ups = _upstream([p.shape[0] for p in per_graph], D, 11)
    sum((p * u.to(DEV)).sum() for p, u in zip(per_graph, ups)).backward()
    torch.cuda.synchronize()
    om, cfg, gd = _oracle_model(model, w, "BiSARGCN", te=True)
    cfg["learnable_lambda"] = False
    times = sorted(gd.keys())
    leaves = O.leaf_tensors(om)
    for v in leaves.values():
        v.requires_grad_(True)
    want, *_ = O.sa_encode(om, cfg, gd, targets, times, L, [gd[t] for t in targets], bi=True)
    sum((p * u.double()).sum() for p, u in zip(want, ups)).backward()
    for i, (a, b) in enumerate(zip(per_graph, want)):
        assert_close(a, b, 1e-5, 3e-6, "attention window %d target embeddings" % i)
    enc, eo = model.ent_encoder, om["ent_encoder"]
    for name, got, ref in [("ent_embeds", model.ent_embeds.grad, om["ent_embeds"].grad),
                           ("layer_2.weight", enc.layer_2.weight.grad, eo["layer_2"]["weight"].grad),
                           ("layer_2.q_linear", enc.layer_2.q_linear.weight.grad, eo["layer_2"]["q_linear"].grad),
                           ("layer_2.k_linear", enc.layer_2.k_linear.weight.grad, eo["layer_2"]["k_linear"].grad),
                           ("layer_2.v_linear", enc.layer_2.v_linear.weight.grad, eo["layer_2"]["v_linear"].grad),
                           ("layer_1.loop_weight", enc.layer_1.loop_weight.grad, eo["layer_1"]["loop_weight"].grad)]:
        assert got is not None and ref is not None, name
        _assert_grad_close(got, ref, "attention d " + name)


@pytest.mark.parametrize("P,N", [(37, 500), (200, 7128), (1500, 500), (5, 10488)])
def test_filtered_rank_kernel_vs_oracle_tie_free(P, N):
    """temp_filtered_rank against the oracle's restatement of utils/evaluation.py:53-106 (mask -> -10e6 -> sigmoid -> sort
    descending -> index of the target) on TIE-FREE scores: every row is a permutation of N equally spaced values in
    [-4, 4], so neighbouring sigmoids differ by > 1e-5 and the integer ranks must be identical."""
    g = torch.Generator().manual_seed(P * 13 + N)
    sc = torch.stack([(torch.randperm(N, generator=g).float() - N / 2) * (8.0 / N) for _ in range(P)])
    tgt = torch.randint(0, N, (P,), generator=g)
    cnt = torch.randint(0, 40, (P,), generator=g)
    cnt[0] = 0
    ptr = torch.zeros(P + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(cnt, 0).int()
    lists = [torch.randperm(N, generator=g)[:c].sort().values for c in cnt.tolist()]
    if P > 1 and cnt[1] > 0:
        lists[1][0] = tgt[1]                                     # the target itself is listed: the reference un-masks it
        lists[1] = lists[1].unique()
        ptr[1:] = torch.cumsum(torch.tensor([len(x) for x in lists]), 0).int()
    ids = torch.cat(lists + [torch.zeros(0, dtype=torch.int64)]).int()
    mask = torch.zeros(P, N, dtype=torch.bool)
    for i, l in enumerate(lists):
        mask[i, l] = True
        mask[i, tgt[i]] = False
    want = O.rank_from_scores(sc, mask, tgt)
    got = TB.get_backend().filtered_rank(sc.to(DEV), tgt.int().to(DEV), ptr.to(DEV), ids.to(DEV)).cpu()
    assert torch.equal(got, want)


def test_scores_and_query_kernels_vs_reference_golden():
    """temp_amd.scores (the preserved distmult / complex / transE call signatures) and the folded-query HIP kernels
    (temp_bilinear_query_fwd) against G9_scores.npz, recorded from the reference's utils/scores.py."""
    from temp_amd import scores as SC
    z = load("G9_scores")
    s, r, o, cand = (T(z[k]).to(DEV) for k in ("s", "r", "o", "cand"))
    for name in ("distmult", "complex", "transE"):
        fn = getattr(SC, name)
        assert_close(fn(s, r, o), z[name + "_single"], 1e-5, 1e-6, name + " single")
        assert_close(fn(s, r, cand, mode="tail"), z[name + "_tail"], 1e-5, 1e-6, name + " tail")
        assert_close(fn(cand, r, o, mode="head"), z[name + "_head"], 1e-5, 1e-6, name + " head")
    be = TB.get_backend()
    P, D = s.shape
    idx = torch.arange(P, dtype=torch.int32, device=DEV)
    for name in ("distmult", "complex"):
        for mode, known, flag in (("tail", s, 1), ("head", o, 0)):
            q = be.bilinear_query_fwd(name, known.contiguous(), idx, r.contiguous(), idx, torch.full((P,), flag, dtype=torch.int32, device=DEV))
            assert_close((q.unsqueeze(1) * cand).sum(-1), z["%s_%s" % (name, mode)], 1e-5, 1e-6, "%s %s through the folded-query kernel" % (name, mode))


# ---------------------------------------------------------------------------------------------------------------------
# persistent window chain (temp_gru_chain_fwd / _bwd): one launch for all positions
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d,type1,kw", [(200, False, dict(n_chain=2, K=6, E=90, lo=20, hi=70)), (32, False, dict(n_chain=2, K=5, E=200, lo=50, hi=200)),
                                        (16, True, dict(n_chain=2, K=6, E=90, lo=20, hi=70)), (128, False, dict(n_chain=1, K=4, E=64, lo=64, hi=64)),
                                        (200, True, dict(n_chain=3, K=3, E=40, lo=1, hi=40)), (248, False, dict(n_chain=1, K=3, E=50, lo=10, hi=50))])
@pytest.mark.parametrize("want", [None, "some"])
def test_chain_kernels_vs_reference_and_per_position_path(d, type1, kw, want):
    """temp_gru_chain_fwd / _bwd on random chain programs (idle tracks, tracks that start mid-chain, an empty position, partial
    last tiles for every d) against (a) the test backend's panel-by-panel reference on the CPU and (b) the per-position
    HIP launches they replace."""
    from tests.chain_cases import make_rnns, random_program, run_program
    from tests.cpu_backend import CpuTestBackend
    prog, n_x = random_program(d + 3 * int(type1), **kw)
    w = None if want is None else tuple(i for i, it in enumerate(prog.inst) if it.next < 0 or i % 3 == 1)[:8]
    n_rnn = kw["n_chain"]
    rnns = make_rnns(n_rnn, d, type1, 5)
    hip_chain = run_program(prog, n_x, d, rnns, DEV, w, type1, 17, chain_kernels=True)
    hip_steps = run_program(prog, n_x, d, rnns, DEV, w, type1, 17, chain_kernels=False)
    TB.set_backend(CpuTestBackend())
    try:
        prog.__dict__.pop("_chain_tabs", None)
        prog.dev = None
        cpu = run_program(prog, n_x, d, [m.cpu() for m in rnns], torch.device("cpu"), w, type1, 17, chain_kernels=True)
    finally:
        TB.set_backend(None)
    for other, name in ((cpu, "CPU reference"), (hip_steps, "per-position launches")):
        for a, b in zip(hip_chain[0], other[0]):        # (the type-1 cell draws its weights from N(0, 1) like the reference's: pre-activations of +-14, saturated gates)
            assert_close(a, b, 1e-4 if type1 else 1e-5, 2e-5 if type1 else 2e-6, "states vs " + name)
        assert_close(hip_chain[1], other[1], 1e-4, 2e-5 * max(1.0, float(other[1].abs().max())), "d_x vs " + name)
        for a, b in zip(hip_chain[2], other[2]):
            assert_close(a, b, 1e-4, 2e-5 * max(1.0, float(b.abs().max())), "GRU parameter gradient vs " + name)


@pytest.mark.parametrize("d,type1", [(200, False), (40, True)])
def test_chain_shared_input_gates_gpu(d, type1):
    """TempGruChain.gi_index + temp_gru_input_gates_gather_multi: a program whose x rows repeat (an entity whose snapshot row
    serves several positions) computes the gates once per distinct row.  States, d_x and the GRU gradients against the
    per-position launches (gates of every row); and the shared run against the unshared chain run: the states are BIT-identical
    (a gi row does not depend on where in the launch it is computed; both gate GEMMs sit below the 16 384-row switch)."""
    from tests.chain_cases import make_rnns, random_program, run_program
    prog, n_x = random_program(23, n_chain=2, K=7, E=300, lo=100, hi=300)
    labels = np.random.default_rng(3).integers(0, n_x // 3, n_x)
    rnns = make_rnns(2, d, type1, 5)
    shared = run_program(prog, n_x, d, rnns, DEV, None, type1, 17, chain_kernels=True, x_src=labels)
    sh = prog.gi_shared(DEV)
    assert sh is not None and sh["rows"] * 8 <= prog.n_total * 7
    steps = run_program(prog, n_x, d, rnns, DEV, None, type1, 17, chain_kernels=False, x_src=labels)
    for a, b in zip(shared[0], steps[0]):
        assert_close(a, b, 1e-4 if type1 else 1e-5, 2e-5 if type1 else 2e-6, "states vs per-position launches")
    assert_close(shared[1], steps[1], 1e-4, 2e-5 * max(1.0, float(steps[1].abs().max())), "d_x")
    for a, b in zip(shared[2], steps[2]):
        assert_close(a, b, 1e-4, 2e-5 * max(1.0, float(b.abs().max())), "GRU parameter gradient")
    # the same rows without the labels: every row's gates computed, same chain kernels
    g = torch.Generator().manual_seed(17)
    leaf = (torch.randn(int(labels.max()) + 1, d, generator=g) * 0.5)
    from temp_amd import gru_chain as GC
    prog.__dict__.pop("_gi_shared", None)
    prog.__dict__.pop("x_src", None)
    out = GC.gru_chain(leaf[torch.from_numpy(labels).long()].to(DEV), prog, [m.to(DEV) for m in rnns], 0.1, type1, None)
    for i, it in enumerate(prog.inst):
        assert torch.equal(out[it.h0:it.h0 + it.n].cpu(), shared[0][i]), i


@pytest.mark.parametrize("variant", [0, 1])
def test_gru_input_gates_gather_multi_gpu(variant, hip_backend):
    """temp_gru_input_gates_gather_multi == temp_gru_input_gates_multi on the rows gathered beforehand (bit-identical: the same
    kernels on the same rows), with repeats, a NULL table next to real ones, row counts on both sides of the split switch."""
    from temp_amd import _lib
    D = 200
    G = 3 * D if variant == _lib.GRU_TORCH else D
    gen = torch.Generator(device="cpu").manual_seed(77 + variant)
    for rows, picks in (((50000, 41000), (30000, 20000)), ((900, 400, 77), (300, None, 500))):
        xs = [torch.randn(n, D, generator=gen).to(DEV) for n in rows]
        ws = [((torch.rand(G, D, generator=gen) - 0.5) * 0.3).to(DEV) for _ in rows]
        bs = [((torch.rand(G, generator=gen) - 0.5) * 0.3).to(DEV) for _ in rows]
        idx = [None if k is None else torch.randint(0, n, (k,), generator=gen).to(torch.int32).to(DEV) for n, k in zip(rows, picks)]
        taken = [x if t is None else x[t.long()].contiguous() for x, t in zip(xs, idx)]
        want = [torch.empty(t.shape[0], G, device=DEV) for t in taken]
        hip_backend.gru_input_gates_multi(taken, ws, bs, variant, want)
        got = [torch.full((t.shape[0], G), float("nan"), device=DEV) for t in taken]
        hip_backend.gru_input_gates_multi(xs, ws, bs, variant, got, x_idx=idx)
        torch.cuda.synchronize()
        for k, (a, b) in enumerate(zip(want, got)):
            assert torch.equal(a, b), (rows, k, float((a - b).abs().max()))


@pytest.mark.parametrize("rows", [(60000, 58000), (20000, 20000, 17000), (900, 700), (30001,), (9000, 8000, 7000, 6004)])
def test_gru_weight_grads_multi_gpu(rows, hip_backend):
    """temp_gru_weight_grads_multi (both directions' d_W_ih, d_W_hh, bias gradients and d_x: ONE weight-gradient launch, ONE
    reduction, ONE d_x launch) against the per-GRU call: the same products over other row slices -> fp32 summation-order
    differences only, and both against fp64.  Small row counts are outside the split-operand kernel: None, nothing launched."""
    from temp_amd import _lib
    D = 200
    gen = torch.Generator(device="cpu").manual_seed(11 + len(rows))
    mk = lambda n, w, s=1.0: (torch.randn(n, w, generator=gen) * s).to(DEV)
    xs, hd = [mk(n, D) for n in rows], [mk(n, D) for n in rows]
    dgi, dgh = [mk(n, 3 * D, 0.1) for n in rows], [mk(n, 3 * D, 0.1) for n in rows]
    ws = [((torch.rand(3 * D, D, generator=gen) - 0.5) * 0.3).to(DEV) for _ in rows]
    dxm = [torch.full((n, D), float("nan"), device=DEV) for n in rows]
    if len(rows) == 4:
        dxm[2] = None                                  # a GRU whose input gradient nobody needs (rec_stack's second layer)
    got = hip_backend.gru_weight_grads_multi(xs, hd, dgi, dgh, ws, _lib.GRU_TORCH, dxm)
    if min(rows) < 4096:
        assert got is None
        return
    assert got is not None and len(got) == len(rows)
    torch.cuda.synchronize()
    for k, n in enumerate(rows):
        dx1 = torch.empty(n, D, device=DEV)
        one = hip_backend.gru_weight_grads(xs[k], hd[k], dgi[k], dgh[k], ws[k], _lib.GRU_TORCH, dx1)
        if dxm[k] is not None:
            # (a launch picks its kernels by the rows of ALL its problems, a single call by its own: bit-identical when both agree)
            if min(rows) >= 16384 or sum(rows) < 16384:
                assert torch.equal(dx1, dxm[k]), "d_x: the same panel kernel on the same rows"
            else:
                assert float((dx1 - dxm[k]).abs().max()) < 1e-4 * float(dx1.abs().max())
        want = (dgi[k].double().t() @ xs[k].double(), dgh[k].double().t() @ hd[k].double(), dgi[k].double().sum(0), dgh[k].double().sum(0))
        scale = (dgi[k].abs().double().t() @ xs[k].abs().double(), dgh[k].abs().double().t() @ hd[k].abs().double(),
                 dgi[k].abs().double().sum(0), dgh[k].abs().double().sum(0))
        for a, b, w, sc in zip(got[k], one, want, scale):
            assert a.shape == b.shape and torch.isfinite(a).all()
            for t in (a, b):
                assert float(((t.double() - w).abs() / sc.clamp_min(1e-30)).max()) < 2e-6


def test_gru_weight_grads_beyond_2g_elements_gpu(hip_backend):
    """A weight-gradient product whose dgi has more than 2^31 elements (3.6 M rows x 600: one direction of the HBM-regime window,
    bench.py extra.hbm_window) stays on the split-operand kernel (64-bit row bases): the whole product against the sum of its two
    halves, each below 2^31 elements, through the same entry point."""
    from temp_amd import _lib
    D, n = 200, 3_600_000
    g = torch.Generator(device=DEV).manual_seed(5)
    x, hd = torch.randn(n, D, device=DEV, generator=g), torch.randn(n, D, device=DEV, generator=g)
    dgi, dgh = torch.randn(n, 3 * D, device=DEV, generator=g) * 0.1, torch.randn(n, 3 * D, device=DEV, generator=g) * 0.1
    w = (torch.rand(3 * D, D, device=DEV, generator=g) - 0.5) * 0.3
    assert dgi.numel() > 2 ** 31
    lib = _lib.load()
    t0 = lib.temp_trace_begin(64)
    whole = hip_backend.gru_weight_grads_multi([x], [hd], [dgi], [dgh], [w], _lib.GRU_TORCH, [None])
    import ctypes
    ids, ms, cnt = (ctypes.c_int32 * 64)(), (ctypes.c_float * 64)(), ctypes.c_int32(0)
    lib.temp_trace_end(ids, ms, 64, ctypes.byref(cnt))
    names = [lib.temp_trace_kernel_name(ids[i]).decode() for i in range(cnt.value)]
    assert whole is not None and "k_gemm_tn_bx8" in names, names
    h = n // 2
    parts = [hip_backend.gru_weight_grads(x[a:b], hd[a:b], dgi[a:b], dgh[a:b], w, _lib.GRU_TORCH, None) for a, b in ((0, h), (h, n))]
    torch.cuda.synchronize()
    for k in range(4):
        want = parts[0][k].double() + parts[1][k].double()
        err = float((whole[0][k].double() - want).abs().max())
        assert err < 3e-5 * float(want.abs().max()) + 1e-3, (k, err, float(want.abs().max()))


def test_chain_kernels_bitwise_repeatable():
    from tests.chain_cases import make_rnns, random_program, run_program
    prog, n_x = random_program(41, n_chain=2, K=8, E=500, lo=300, hi=500)
    rnns = make_rnns(2, 200, False, 6)
    a = run_program(prog, n_x, 200, rnns, DEV, None, False, 3)
    b = run_program(prog, n_x, 200, rnns, DEV, None, False, 3)
    assert all(torch.equal(x, y) for x, y in zip(a[0], b[0])) and torch.equal(a[1], b[1]) and all(torch.equal(x, y) for x, y in zip(a[2], b[2]))


# ---------------------------------------------------------------------------------------------------------------------
# config 3: BiGRRGCN --rec-only-last-layer --post-ensemble (and the impute variants)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("batched", [True, False])
def test_post_ensemble_bi_window_golden_gpu(batched):
    from tests.window_cases import check_post_bi
    check_post_bi(DEV, batched)


@pytest.mark.parametrize("name,batched", [("G15_impute_bi", True), ("G15_impute_bi", False), ("G15_impute_uni", True),
                                          ("G15_impute_uni_full", False)])
def test_impute_window_golden_gpu(name, batched):
    from tests.window_cases import check_impute_window
    check_impute_window(name, DEV, batched)


@pytest.mark.parametrize("name,batched", [("G16_eval_impute_uni", True), ("G16_eval_impute_bi", True), ("G16_eval_impute_bi", False)])
def test_impute_evaluate_golden_gpu(name, batched):
    from tests.window_cases import check_impute_evaluate
    check_impute_evaluate(name, DEV, batched)


def test_post_ensemble_loss_definition_gpu():
    from tests.window_cases import check_post_ensemble_loss
    check_post_ensemble_loss(DEV)


def test_config3_icews0515_post_ensemble_step_vs_oracle_gpu():
    """BASELINE config 3 at its own size: the ICEWS05-15-shaped workload (10 488 entities, L = 15, D = 200, 100 bases),
    PostEnsembleBiDynamicRGCN with --rec-only-last-layer --post-ensemble on the batched HIP path: (local, temporal) target
    embeddings and the all-entity (local, temporal) matrices of every window against the fp64 oracle, plus the gradients of a
    seeded weighted sum of all four."""
    import bench
    from temp_amd import synthetic
    from temp_amd.post_dynamic_rgcn import PostEnsembleBiDynamicRGCN
    w = synthetic.workload("S-icews0515", seed=0)
    args = bench.make_args(w, w["module"])
    args.post_ensemble = True
    torch.manual_seed(1)
    snaps = w["snapshots"]
    model = PostEnsembleBiDynamicRGCN(args, w["num_ents"], w["num_rels"], snaps, snaps, snaps).to(DEV)
    targets = sorted(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 0)[:3], reverse=True)
    L, D = w["L"], w["D"]
    locs, recs, wb, hist = model.encode_post(targets, L, train=False)
    assert wb.batched and wb.program is not None
    gen = torch.Generator().manual_seed(5)
    sel = torch.arange(0, w["num_ents"], 3)
    total, alls = 0, []
    for i, g in enumerate(wb.graphs):
        a_loc, a_rec = model.get_all_embeds_Gt(locs[i], recs[i], g, wb.rows[i][-1], wb.plan, i, hist, wb.hist_loc)
        alls.append((a_loc, a_rec))
        for x in (locs[i], recs[i], a_loc[sel.to(DEV)], a_rec[sel.to(DEV)]):
            total = total + (x * torch.randn(x.shape, generator=gen).to(DEV)).sum()
    total.backward()
    torch.cuda.synchronize()
    om, cfg, gd = _oracle_model(model, w, w["module"])
    times = sorted(gd.keys())
    for v in O.leaf_tensors(om).values():
        v.requires_grad_(True)
    tf, tb = O.get_batch_graph_list_bi(targets, L, times)
    Hf = O.post_bi_pre_forward(om, cfg, gd, tf, L, True)
    Hb = O.post_bi_pre_forward(om, cfg, gd, tb, L, False)
    wl, wr = O.post_bi_target_embeds(om, cfg, Hf, Hb, [gd[t] for t in targets], tf[-1], L)
    gen = torch.Generator().manual_seed(5)
    want_total = 0
    for i, t in enumerate(targets):
        assert_close(locs[i], wl[i], 1e-5, 3e-6, "config 3 local embeddings, window %d" % i)
        assert_close(recs[i], wr[i], 1e-5, 3e-6, "config 3 temporal embeddings, window %d" % i)
        o_loc, o_rec = O.post_bi_all_embeds(om, cfg, Hf, Hb, i, t, L)
        ids = torch.as_tensor(gd[t].ids)
        o_loc, o_rec = o_loc.index_copy(0, ids, wl[i]), o_rec.index_copy(0, ids, wr[i])
        assert_close(alls[i][0], o_loc, 1e-5, 3e-6, "config 3 all-entity local matrix, window %d" % i)
        assert_close(alls[i][1], o_rec, 1e-5, 3e-6, "config 3 all-entity temporal matrix, window %d" % i)
        for x in (wl[i], wr[i], o_loc[sel], o_rec[sel]):
            want_total = want_total + (x * torch.randn(x.shape, generator=gen).double()).sum()
    want_total.backward()
    enc, eo = model.ent_encoder, om["ent_encoder"]
    for name, got, ref in [("ent_embeds", model.ent_embeds.grad, om["ent_embeds"].grad),
                           ("layer_1.weight", enc.layer_1.weight.grad, eo["layer_1"]["weight"].grad),
                           ("layer_2.loop_weight", enc.layer_2.loop_weight.grad, eo["layer_2"]["loop_weight"].grad),
                           ("forward_rnn.w_hh", enc.layer_2.forward_rnn.weight_hh_l0.grad, eo["layer_2"]["forward_rnn"][0]["w_hh"].grad),
                           ("backward_rnn.w_ih", enc.layer_2.backward_rnn.weight_ih_l0.grad, eo["layer_2"]["backward_rnn"][0]["w_ih"].grad)]:
        _assert_grad_close(got, ref, "config 3 d " + name)


# ---------------------------------------------------------------------------------------------------------------------
# snapshot store: device-side edge subsample + renorm (temp_subsample_views)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,E,R,rate", [(500, 7475, 20, 0.5), (60, 900, 8, 0.8), (2000, 200000, 20, 0.5), (10, 37, 4, 0.5), (5, 0, 4, 0.5),
                                        (300, 5000, 6, 0.0), (300, 5000, 6, 1.0)])
def test_device_subsample_kernels_bit_exact_and_layer_parity(n, E, R, rate, hip_backend):
    """temp_subsample_views against the test backend's reference (same 64-bit counter hash): keep mask, rewritten view
    arrays, degrees and norms bit-exact; then an RGCN layer on the device-built subgraph equals the layer on the HOST-built
    subgraph of the same kept edges (edge_subgraph + sort + upload: the path it replaces) within fp32 round-off."""
    from temp_amd import functional as TF
    from temp_amd.snapshot import Snapshot, device_subsample
    from tests.cpu_backend import CpuTestBackend
    rng = np.random.default_rng(n + E)
    src, dst, rel = rng.integers(0, n, E), rng.integers(0, n, E), rng.integers(0, R, E)
    if E > 400:
        dst[:E // 4] = rng.integers(0, 3, E // 4)
    g = Snapshot(n, src, dst, rel, np.arange(n))
    keep = int(rate * E)
    sub = device_subsample([g], [keep], [99], DEV, R, want_mask=True)[0]
    torch.cuda.synchronize()
    TB.set_backend(CpuTestBackend())
    try:
        g2 = Snapshot(n, src, dst, rel, np.arange(n))
        ref = device_subsample([g2], [keep], [99], torch.device("cpu"), R, want_mask=True)[0]
    finally:
        TB.set_backend(None)
    assert torch.equal(sub._mask.cpu(), ref._mask) and int(sub._mask.sum()) == keep
    got, want = sub.device_views(DEV, R), ref.device_views(torch.device("cpu"), R)
    for name in ("by_dst", "by_src", "by_rel"):
        beg, end = want[name]["chunk_beg"].numpy(), want[name]["chunk_end"].numpy()
        assert torch.equal(got[name]["chunk_end"].cpu(), want[name]["chunk_end"])
        live = np.zeros(E, dtype=bool)
        for b, e in zip(beg, end):
            live[b:e] = True
        for arr in ("a", "b"):
            assert np.array_equal(got[name][arr].cpu().numpy()[live], want[name][arr].numpy()[live]), (name, arr)
    for arr in ("in_deg", "out_deg", "nnorm"):
        assert torch.equal(got[arr].cpu(), want[arr]), arr
    if E == 0:
        return
    # layer parity: device-derived views vs host-built subgraph of the same edges
    D, B = 32, 16
    idx = np.nonzero(sub._mask.cpu().numpy())[0]
    host_sub = g.edge_subgraph(idx)
    gen = torch.Generator().manual_seed(1)
    h = torch.randn(n, D, generator=gen).to(DEV).requires_grad_(True)
    w = (torch.randn(R, B * (D // B) ** 2, generator=gen) * 0.3).to(DEV).requires_grad_(True)
    lw = (torch.randn(D, D, generator=gen) * 0.1).to(DEV).requires_grad_(True)
    outs = []
    for graph in (sub, host_sub):
        for t in (h, w, lw):
            t.grad = None
        y = TF.rgcn_layer(h, graph.device_graph(DEV, R), w, lw, None, B, None)
        (y * y).sum().backward()
        outs.append((y.detach().clone(), h.grad.clone(), w.grad.clone(), lw.grad.clone()))
    for a, b, what in zip(outs[0], outs[1], ("y", "d_h", "d_weight", "d_loop")):
        assert_close(a, b, 2e-5, 2e-5 * max(1.0, float(b.abs().max())), "subsampled layer " + what)


def test_training_step_with_device_subsample_gpu():
    """A full training step of the headline model draws its 50 % target subsets on the device; the same seeds give the same
    step, and the result equals the step on HOST-built subgraphs of the same kept edges."""
    import bench
    from temp_amd import synthetic
    w = synthetic.workload("S-gdelt", seed=0)
    model = bench.build_model(w, DEV)
    targets = synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 0)[:3]

    def run(edge_ids=None):
        model.sample_rng = np.random.default_rng(5)
        for p in model.parameters():
            p.grad = None
        wb = model.prepare(targets, w["L"], train=True, target_edge_ids=edge_ids)
        out = model.run(wb)[0]
        (out * out).sum().backward()
        return wb, out.detach().clone(), model.ent_embeds.grad.clone()

    wb, out1, g1 = run()
    from temp_amd.snapshot import SubsampledSnapshot
    assert all(isinstance(g, SubsampledSnapshot) for g in wb.target.graphs)
    assert all(g.number_of_edges() == int(0.5 * full.number_of_edges()) for g, full in zip(wb.target.graphs, wb.graphs))
    _, out2, g2 = run()
    assert torch.equal(out1, out2) and torch.equal(g1, g2)
    ids = [g.edge_ids for g in wb.target.graphs]
    _, out3, g3 = run(edge_ids=ids)
    assert_close(out3, out1, 2e-5, 2e-6, "device vs host subsample: target embeddings")
    assert_close(g3, g1, 1e-4, 1e-4 * float(g1.abs().max()), "device vs host subsample: d ent_embeds")


# ---- split-operand GEMMs (csrc/gemm_bx.hpp, gemm_tn_bx.hpp): fp32 products as six bf16 MFMA products -------------------------
def _wide(shape, seed, scale):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g) * torch.exp(3.0 * torch.rand(shape, generator=g) - 1.5) * scale
    return x.cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,trans_b", [
    (20000, 200, 600, True),      # input gates: packed weights, 5 column groups
    (20000, 600, 200, False),     # dx: one group of 7 tiles, 38 slabs
    (16500, 200, 7128, True),     # score matrix against all ICEWS entities: weights beyond a scratch slot -> in-block split
    (17000, 64, 40, False),       # narrow output (2 tiles), short K
    (33333, 208, 132, True),      # ragged row tile, 5 tiles
    (16384, 24, 36, False),       # K not a multiple of 16 (one and a half slabs)
])
def test_split_operand_gemm_vs_fp64(M, K, N, trans_b):
    """The error of the split-operand product against fp64 stays at the level of ONE fp32 rounding per product (the fp32 MFMA
    kernels measure 4e-7 of sum |a||b| on the same data, tools/bx_probe.hip); bar 1e-6."""
    be = TB.get_backend()
    a = _wide((M, K), 11, 1.0)
    b = _wide((N, K) if trans_b else (K, N), 12, 0.2)
    out = be.linear(a, b, trans_b)
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M), torch.randint(0, M, (1400,))]).cuda()
    bd = b.double().t() if trans_b else b.double()
    ref = a[rows].double() @ bd
    sabs = a[rows].double().abs() @ bd.abs()
    err = ((out[rows].double() - ref).abs() / sabs.clamp_min(1e-300)).max().item()
    assert err < 1e-6, "split-operand GEMM %dx%dx%d: error %.3e of sum|a||b|" % (M, K, N, err)


@pytest.mark.gpu
@pytest.mark.parametrize("M,Ka,Nb", [(58003, 600, 200), (20000, 200, 200), (17001, 600, 136), (16400, 208, 164), (30000, 72, 200)])
def test_split_operand_weight_gradient_vs_fp64(M, Ka, Nb):
    """out = a^T . b over M rows (k_gemm_tn_bx): ragged last slab, ragged last row block of Ka, 5 / 6 / 7 column tiles."""
    be = TB.get_backend()
    a = _wide((M, Ka), 21, 1.0)
    b = _wide((M, Nb), 22, 1.0)
    out = be.linear_tn(a, b)
    torch.cuda.synchronize()
    ref = a.double().t() @ b.double()
    sabs = a.double().abs().t() @ b.double().abs()
    err = ((out.double() - ref).abs() / sabs).max().item()
    assert err < 2e-7, "split-operand weight gradient %d x (%d, %d): error %.3e of sum|a||b|" % (M, Ka, Nb, err)
    out2 = be.linear_tn(a, b)
    assert torch.equal(out, out2), "weight gradient not bitwise repeatable"


@pytest.mark.parametrize("name", ["G17_post_eval_complex", "G17_post_eval_distmult"])
def test_post_evaluation_filters_golden_gpu(name):
    from tests.window_cases import check_post_eval_filters
    check_post_eval_filters(name, DEV)


@pytest.mark.parametrize("name,batched", [("G18_eval_post_uni", True), ("G18_eval_post_bi", True), ("G18_eval_post_bi", False)])
def test_post_ensemble_evaluate_golden_gpu(name, batched):
    from tests.window_cases import check_post_ensemble_evaluate
    check_post_ensemble_evaluate(name, DEV, batched)


@pytest.mark.parametrize("name,batched", [("G19_post_ratio_uni", True), ("G19_post_ratio_bi", True)])
def test_post_ensemble_own_ratio_golden_gpu(name, batched):
    from tests.window_cases import check_post_ensemble_ratio
    check_post_ensemble_ratio(name, DEV, batched)


def test_sharded_step_rccl_single_rank_gpu():
    """BASELINE north_star mode on the RCCL backend with ONE rank (two ranks cannot share a GPU: RCCL answers "Duplicate GPU
    detected", tools/p2p_probe.py) -- with the rank as its own peer for the point-to-point exchange: SnapshotShardedEncoder +
    ShardedStep (three HIP graphs around the two exchanges) + GradBucket with ReduceOp.AVG through nccl.
      * the graph-replayed step is BIT-identical to the same three parts run eagerly (outputs and every gradient);
      * against the unsharded batched step of the same windows: target embeddings and gradients equal to rounding (measured
        1.5e-7 on the embeddings: the sharded program groups the chain rows by rank-local window, so row panels -- and with
        them the order of a few sums -- differ)."""
    import torch.distributed as dist
    from tests.test_dist_cpu import _free_port
    import bench
    from temp_amd import synthetic
    from temp_amd.dist import ShardedStep, SnapshotShardedEncoder
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        w = synthetic.workload("S-gdelt", seed=0)
        model = bench.build_model(w, DEV)
        targets = synthetic.default_targets(w["num_times"], w["L"], 3, 0)
        params = [p for p in model.parameters()]
        model.sample_rng = np.random.default_rng(2)
        wb = model.prepare(targets, w["L"], train=True)
        ref_out = model.run(wb)[0]
        ref_out.backward(torch.ones_like(ref_out))
        ref_grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        ref_out = ref_out.detach().clone()                           # (no reference to the eager step's autograd graph may survive:
        del wb                                                       #  see ShardedStep._capture)
        for p in params:
            p.grad = None
        model.sample_rng = np.random.default_rng(2)                  # same target subsample
        enc = SnapshotShardedEncoder(model)
        sb = enc.prepare(targets, w["L"], train=True)
        res = {}
        from temp_amd import dist as TD
        # "loopback": the rank is its own peer -- its node-state block and the mirrored gradient block go through RCCL's grouped
        # send / recv (batch_isend_irecv) between the HIP graphs instead of a local copy: the point-to-point path of the exchange
        # on a one-GPU box (two ranks cannot share a GPU: "Duplicate GPU detected")
        for name, graphs, loop in (("eager", False, False), ("graphs", True, False), ("loopback", True, True)):
            TD.LOOPBACK_P2P = loop
            n0 = TD.P2P_BATCHES
            try:
                st = ShardedStep(enc, sb, params, graphs=graphs, average=True, force_allreduce=True)
                assert (st.graphs is not None) == graphs
                for _ in range(2):                                       # a replayed graph must reproduce itself
                    out = st.step()
                torch.cuda.synchronize()
            finally:
                TD.LOOPBACK_P2P = False
            assert (TD.P2P_BATCHES - n0 >= 4) == loop, "forward + adjoint exchange of two steps = four RCCL point-to-point batches"
            res[name] = (out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
            del st, out
            for p in params:
                p.grad = None
        assert torch.equal(res["eager"][0], res["graphs"][0]) and torch.equal(res["loopback"][0], res["graphs"][0])
        assert set(res["eager"][1]) == set(res["graphs"][1]) == set(res["loopback"][1]) == set(ref_grads)
        for k, g in res["eager"][1].items():
            assert torch.equal(g, res["graphs"][1][k]), k
            assert torch.equal(res["loopback"][1][k], res["graphs"][1][k]), "loopback " + k
        assert_close(res["graphs"][0], ref_out, 2e-6, 5e-7, "sharded vs unsharded target embeddings")
        for k, g in res["graphs"][1].items():
            assert_close(g, ref_grads[k], 2e-5, 2e-6 * max(1.0, float(ref_grads[k].abs().max())), "sharded vs unsharded " + k)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("D,B", [(200, 100), (64, 16), (32, 32)])
def test_tiled_edge_kernels_bit_identical_gpu(D, B, hip_backend):
    """LDS-tiled edge kernels (csrc/rgcn_tile.hpp: one workgroup per (member snapshot, feature slice), rows staged in LDS) against
    the L2-gather kernels on the same batched graph: same chunks, same per-chunk order => BIT-identical aggregation, d/dh and
    d/dweight at D = 200 (1e-6 for narrow rows, whose gather kernels use another order), for the plain layer and the table
    layer, with a device-subsampled member in the batch.  The member tables of the
    union are checked against its chunk arrays first."""
    from temp_amd import _lib, snapshot as S, synthetic
    lib = _lib.load()
    w = synthetic.workload("S-gdelt", seed=0)
    parts = [w["snapshots"][t] for t in (3, 40, 7, 103, 12, 200, 77, 5, 300)]
    R2 = 2 * w["num_rels"]
    sub = S.device_subsample([parts[2]], [parts[2].number_of_edges() // 2], [1234], DEV, R2)[0]
    parts[2] = sub
    g = S.batch(parts)
    dg = g.device_graph(DEV, R2)
    mb = dg.c.members
    M = len(parts)
    assert mb.n_members == M and mb.max_nodes == max(p.n for p in parts)
    tab = dg._members["table"].cpu().numpy().reshape(7, M + 1)
    for vi, vn in enumerate(("by_dst", "by_src")):                    # fix-up entries per member (TempMembers.fix_off): member-major lists
        fseg = dg.view_tensor(vn, "fix_seg").cpu().numpy()
        fo = tab[5 + vi]
        assert fo[0] == 0 and fo[-1] == fseg.shape[0]
        for m in range(M):
            assert ((fseg[fo[m]:fo[m + 1]] >= tab[0][m]) & (fseg[fo[m]:fo[m + 1]] < tab[0][m + 1])).all(), (vn, m)
    assert np.array_equal(tab[0], g.node_off) and np.array_equal(tab[1], g.edge_off)
    for vi, vn in enumerate(("by_dst", "by_src", "by_rel")):
        seg, beg, end = (dg.view_tensor(vn, k).cpu().numpy() for k in ("chunk_seg", "chunk_beg", "chunk_end"))
        co = tab[2 + vi]
        assert co[0] == 0 and co[-1] == seg.shape[0] and mb.max_chunks[vi] == int(np.diff(co).max())
        for m in range(M):
            sl = slice(co[m], co[m + 1])
            assert (beg[sl] >= tab[1][m]).all() and (end[sl] <= tab[1][m + 1]).all(), (vn, m)
            if vn != "by_rel":
                assert (seg[sl] >= tab[0][m]).all() and (seg[sl] < tab[0][m + 1]).all(), (vn, m)
    rng = np.random.default_rng(5)
    f = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(DEV)
    n, s_ = g.n, D // B
    h, wt, lw, b, gy = f(n, D), f(R2, B * s_ * s_) * 0.5, f(D, D) * 0.2, f(D), f(n, D)
    table = f(w["num_ents"], D)
    ids = torch.from_numpy(g.gids.astype(np.int32)).to(DEV)
    from temp_amd import functional as TF
    inv = TF.gather_inverse(g.gids, w["num_ents"], DEV)

    def run():
        out = hip_backend.rgcn_fwd(dg, h, None, wt, lw, b, B, 1)
        grads = hip_backend.rgcn_bwd(dg, h, out, gy, wt, lw, True, B, 1)
        tout = hip_backend.rgcn_table_fwd(dg, table, ids, wt, lw, None, B, 0)
        tgrads = hip_backend.rgcn_table_bwd(dg, table, ids, inv, tout, gy, wt, lw, False, B, 0)
        torch.cuda.synchronize()
        return [out] + [x for x in grads if x is not None] + [tout] + [x for x in tgrads if x is not None]

    prev = lib.temp_set_option(_lib.OPT_RGCN_TILE, 2)         # 2: the weight-gradient kernel tiled too (default 1: aggregation and d/dh)
    n0 = lib.temp_tile_launches()
    tiled = run()
    assert lib.temp_tile_launches() - n0 == 6, "the LDS-tiled kernels were not launched"
    lib.temp_set_option(_lib.OPT_RGCN_TILE, 3)                # 3: d/dweight with the gradient rows in LDS and the x rows read from L2
    n0 = lib.temp_tile_launches()
    hybrid = run()
    assert lib.temp_tile_launches() - n0 == 6, "the hybrid weight-gradient kernel was not launched"
    lib.temp_set_option(_lib.OPT_RGCN_TILE, 0)
    try:
        n1 = lib.temp_tile_launches()
        gathered = run()
        assert lib.temp_tile_launches() == n1
    finally:
        lib.temp_set_option(_lib.OPT_RGCN_TILE, prev)
    for i, (a, c) in enumerate(zip(tiled, hybrid)):           # same chunks, same edge order inside a chunk: the same bits
        assert torch.equal(a, c), ("tiled vs hybrid", i, float((a - c).abs().max()))
    assert len(tiled) == len(gathered)
    for i, (a, c) in enumerate(zip(tiled, gathered)):
        if tuple(a.shape) == tuple(wt.shape):   # the relation-weight gradient: the gather kernel sums a destination run's source rows
            assert_close(a, c, 2e-6, 2e-6 * max(1.0, float(c.abs().max())), "tiled vs gathered d_weight %d" % i)   # before the outer product
        elif D > 128:        # the gather path runs the one-edge-per-pass kernels here: the SAME per-chunk order as the tiled walkers
            assert torch.equal(a, c), (i, float((a - c).abs().max()))
        else:                # narrow rows: the gather kernels sum several edges per pass and reduce across lanes (another order)
            assert_close(a, c, 2e-6, 2e-6 * max(1.0, float(c.abs().max())), "tiled vs gathered %d" % i)


def test_tiled_edge_kernels_large_members_gpu(hip_backend):
    """The LDS-tiled aggregation / d-dh on members LARGER than GDELT's (1 100 nodes, 15 000 edges, > 1 024 chunks per view: the
    staging paths beyond a thread's first chunk record / first eight edges, narrower feature slices): bit-identical to the gather
    kernels at D = 200, for the plain layer and the table layer."""
    from temp_amd import _lib, snapshot as S, synthetic
    lib = _lib.load()
    R, D, B, n_ents = 20, 200, 100, 1400
    snaps = synthetic.make_snapshots(n_ents, R, 15000, 1100, 3, seed=21)
    g = S.batch([snaps[t] for t in range(3)])
    dg = g.device_graph(DEV, 2 * R)
    assert dg.c.members.n_members == 3 and max(dg.c.members.max_chunks[:2]) > 1024
    rng = np.random.default_rng(22)
    f = lambda *s_: torch.from_numpy(rng.standard_normal(s_).astype(np.float32)).to(DEV)
    s2 = D // B
    h, wt, lw, b, gy = f(g.n, D), f(2 * R, B * s2 * s2) * 0.5, f(D, D) * 0.2, f(D), f(g.n, D)
    table = f(n_ents, D)
    ids = torch.from_numpy(g.gids.astype(np.int32)).to(DEV)
    from temp_amd import functional as TF
    inv = TF.gather_inverse(g.gids, n_ents, DEV)

    def run():
        out = hip_backend.rgcn_fwd(dg, h, None, wt, lw, b, B, 1)
        grads = hip_backend.rgcn_bwd(dg, h, out, gy, wt, lw, True, B, 1)
        tout = hip_backend.rgcn_table_fwd(dg, table, ids, wt, lw, None, B, 0)
        tgrads = hip_backend.rgcn_table_bwd(dg, table, ids, inv, tout, gy, wt, lw, False, B, 0)
        torch.cuda.synchronize()
        return [out] + [x for x in grads if x is not None] + [tout] + [x for x in tgrads if x is not None]

    prev = lib.temp_set_option(_lib.OPT_RGCN_TILE, 1)
    try:
        n0 = lib.temp_tile_launches()
        tiled = run()
        assert lib.temp_tile_launches() - n0 == 4, "the LDS-tiled kernels were not launched"
        lib.temp_set_option(_lib.OPT_RGCN_TILE, 0)
        gathered = run()
    finally:
        lib.temp_set_option(_lib.OPT_RGCN_TILE, prev)
    for i, (a, c) in enumerate(zip(tiled, gathered)):
        assert torch.equal(a, c), (i, float((a - c).abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("rows", [(61000, 60500), (20000, 300), (700, 90), (5, 33000, 17, 4100)])
def test_gru_input_gates_multi_gpu(rows, variant, hip_backend):
    """temp_gru_input_gates_multi (the input gates of both directions of the window chain -- up to four (row block, weight set)
    pairs per launch) against one temp_gru_input_gates call per pair: the same kernels on the same rows => BIT-identical; and
    against fp64 (x . W_ih^T + b_ih, models/GRU_cell.py / torch.nn.GRU's input half).  Row counts on both sides of the
    16 384-row switch to the split-operand kernels, ragged last panels, tiny problems next to large ones."""
    from temp_amd import _lib
    D = 200
    G = 3 * D if variant == _lib.GRU_TORCH else D
    gen = torch.Generator(device="cpu").manual_seed(31 + len(rows) + variant)
    xs = [torch.randn(n, D, generator=gen).to(DEV) for n in rows]
    ws = [((torch.rand(G, D, generator=gen) - 0.5) * 0.3).to(DEV) for _ in rows]
    bs = [((torch.rand(G, generator=gen) - 0.5) * 0.3).to(DEV) for _ in rows]
    single = [torch.empty(n, G, device=DEV) for n in rows]
    for x, w, b, o in zip(xs, ws, bs, single):
        hip_backend.gru_input_gates(x, w, b, variant, o)
    multi = [torch.full((n, G), float("nan"), device=DEV) for n in rows]
    hip_backend.gru_input_gates_multi(xs, ws, bs, variant, multi)
    torch.cuda.synchronize()
    # the launch picks its kernels by the rows of ALL its problems (>= 16 384: the split-operand kernels), a single call by its
    # own rows: bit-identical when both sides make the same choice
    same_route = min(rows) >= 16384 or sum(rows) < 16384
    for k, (x, w, b, a, c) in enumerate(zip(xs, ws, bs, single, multi)):
        assert torch.isfinite(c).all()
        if same_route:
            assert torch.equal(a, c), (k, float((a - c).abs().max()))
        want = x.double() @ w.double().t() + b.double()
        scale = x.abs().double() @ w.abs().double().t() + b.abs().double()
        for got in (a, c):
            err = ((got.double() - want).abs() / scale.clamp_min(1e-30)).max()
            assert float(err) < 2e-6, (k, float(err))


@pytest.mark.gpu
def test_split_fixup_large_hubs_gpu(hip_backend):
    """Two-level fix-up (csrc/rgcn_kernels.hip: k_fixup_split) on a graph whose hubs and relations have thousands of partial rows
    (2^15 nodes, 5.2 M Zipf edges, 4 relation rows, D = 200: the top destination carries ~7 500 partial rows, every relation
    ~10 000 rows of the 400-wide weight gradient, i.e. two column blocks): entries beyond 2 048 rows are listed by the first pass and
    summed by (entry, part) pairs over all blocks.  The list order and the finishing block vary from run to run, the result must not:
    two runs are BIT-identical; against the single-level walk (TEMP_OPT_DEBUG = 100) the sums agree to fp32 reassociation."""
    from temp_amd import _lib, synthetic
    lib = _lib.load()
    n, E, R, D, B = 1 << 15, 5 << 20, 2, 200, 100
    g = synthetic.make_snapshots(n, R, E, n, 1, seed=21)[0]
    dg = g.device_graph(DEV, 2 * R)
    assert dg.views["by_dst"]["n_partial"] >= 1 << 15 and dg.views["by_src"]["n_partial"] >= 1 << 15 and dg.views["by_rel"]["n_partial"] >= 1 << 15
    for view in ("by_dst", "by_src", "by_rel"):
        assert int(dg.view_tensor(view, "fix_cnt").max()) > 4096, view
    gen = torch.Generator(device="cpu").manual_seed(22)
    s_ = D // B
    wt = (torch.rand(2 * R, B * s_ * s_, generator=gen) - 0.5).to(DEV)
    lw = ((torch.rand(D, D, generator=gen) - 0.5) * 0.2).to(DEV)
    h = torch.randn(n, D, generator=gen).to(DEV)
    gy = torch.randn(n, D, generator=gen).to(DEV)

    def run():
        out = hip_backend.rgcn_fwd(dg, h, None, wt, lw, None, B, 1)
        grads = hip_backend.rgcn_bwd(dg, h, out, gy, wt, lw, False, B, 1)
        torch.cuda.synchronize()
        return [out] + [x for x in grads if x is not None]

    prev = lib.temp_set_option(_lib.OPT_DEBUG, 0)
    try:
        a = run()
        b = run()
        lib.temp_set_option(_lib.OPT_DEBUG, 100)
        c = run()
    finally:
        lib.temp_set_option(_lib.OPT_DEBUG, prev)
    for i, (x, y, z) in enumerate(zip(a, b, c)):
        assert torch.isfinite(x).all()
        assert torch.equal(x, y), ("split fix-up not repeatable", i)
        tol = 2e-5 * max(1.0, float(z.abs().max()))
        assert float((x - z).abs().max()) <= tol, (i, float((x - z).abs().max()), tol)
    # the weight gradient IS a fix-up output (the layer output adds the self-loop term on top: 1-ulp differences of a hub's small
    # mean vanish there): different association => different bits, or the split pass did not run
    assert not torch.equal(a[2], c[2]), "the split fix-up did not run (identical bits to the single-level walk)"


def test_rgcn_layer_large_power_law_graph_gpu(hip_backend):
    """One large graph with power-law degrees (2^19 nodes, 2^21 edges, dst / src ~ Zipf): its by-dst and by-src chunk lists are
    long enough (>= 2^18 chunks) for the XCD split by EDGES (csrc/common.hpp: xcd_chunk_range -- the hubs' full chunks sit at the
    front of the list), which no other test reaches.  Layer output and every gradient against the test backend's restatement."""
    from temp_amd import synthetic
    from tests.cpu_backend import CpuTestBackend
    n, E, R, D, B = 1 << 19, 1 << 21, 20, 8, 4
    g = synthetic.make_snapshots(n, R, E, n, 1, seed=11)[0]
    dg = g.device_graph(DEV, 2 * R)
    for view in ("by_dst", "by_src"):
        assert dg.view_tensor(view, "chunk_seg").shape[0] >= 1 << 18, view
    rng = np.random.default_rng(12)
    f = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    s_ = D // B
    h, w, lw, gy = f(n, D), f(2 * R, B * s_ * s_) * 0.5, f(D, D) * 0.2, f(n, D)
    cpu = CpuTestBackend()
    dg_c = g.device_graph(torch.device("cpu"), 2 * R)
    want = cpu.rgcn_fwd(dg_c, h, None, w, lw, None, B, 1)
    wd = cpu.rgcn_bwd(dg_c, h, want, gy, w, lw, False, B, 1)
    got = hip_backend.rgcn_fwd(dg, h.to(DEV), None, w.to(DEV), lw.to(DEV), None, B, 1)
    gd = hip_backend.rgcn_bwd(dg, h.to(DEV), got, gy.to(DEV), w.to(DEV), lw.to(DEV), False, B, 1)
    # absolute floors relative to the largest entry: a hub row sums ~10^5 terms (|d_h| reaches ~100 where typical entries are ~0.1),
    # and the two paths add them in different orders
    assert_close(got, want, 1e-5, 1e-6 * max(1.0, float(want.abs().max())), "rgcn_fwd on the large power-law graph")
    assert_close(gd[0], wd[0], 1e-5, 1e-6 * max(1.0, float(wd[0].abs().max())), "d_h")
    # weight gradients: fp32 sums over 2^19 rows / 2^21 edges on both sides (eps x sqrt(n) = 4e-5 of the largest entry)
    assert_close(gd[1], wd[1], 2e-5, 4e-5 * max(1.0, float(wd[1].abs().max())), "d_weight")
    assert_close(gd[2], wd[2], 2e-5, 4e-5 * max(1.0, float(wd[2].abs().max())), "d_loop")


def test_rgcn_layer_row_gather_in_large_gemm_gpu():
    """temp_rgcn_fwd with feature ids at a size that takes the split-operand GEMM (>= 16 K rows): the self-loop product gathers
    its rows through a_idx inside the kernel; result = the same layer on the explicitly gathered rows.  Nodes without in-edges
    (row mask) included."""
    from temp_amd.snapshot import Snapshot
    be = TB.get_backend()
    rng = np.random.default_rng(5)
    n, E, R2, D, B, n_table = 20000, 120000, 40, 200, 100, 700
    src, dst = rng.integers(0, n, E), rng.integers(0, n - 500, E)          # the last 500 nodes have no in-edges
    g = Snapshot(n, src, dst, rng.integers(0, R2, E), np.arange(n))
    dg = g.device_graph(DEV, R2)
    gen = torch.Generator().manual_seed(3)
    table = torch.randn(n_table, D, generator=gen).cuda()
    ids = torch.from_numpy(rng.integers(0, n_table, n).astype(np.int32)).cuda()
    weight = (torch.randn(R2, B * 4, generator=gen) * 0.3).cuda()
    loop_w = (torch.randn(D, D, generator=gen) * 0.1).cuda()
    bias = (torch.randn(D, generator=gen) * 0.1).cuda()
    got = be.rgcn_fwd(dg, table, ids, weight, loop_w, bias, B, 1)
    want = be.rgcn_fwd(dg, table[ids.long()].contiguous(), None, weight, loop_w, bias, B, 1)
    assert_close(got, want, 1e-6, 1e-6, "rgcn layer with gathered rows")
    assert float(got[-500:].abs().max()) > 0          # isolated nodes still get act(bias)-type rows, identical in both


def test_split_operand_gemm_scratch_slots_per_stream_gpu():
    """The packed weights of the split-operand GEMM live in one scratch slot PER STREAM (gemm_kernels.hip: bx_scratch): launches
    issued concurrently on many streams -- more streams than slots, so the last ones take the kernel that splits the weights
    itself -- each give the result of the same product run alone."""
    be = TB.get_backend()
    M, K, N = 20000, 200, 200
    a = _wide((M, K), 31, 1.0)
    ws = [_wide((N, K), 40 + i, 0.2) for i in range(11)]
    alone = [be.linear(a, w, True) for w in ws]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in ws]
    outs = [None] * len(ws)
    for rep in range(3):
        for i, (w, st) in enumerate(zip(ws, streams)):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                outs[i] = be.linear(a, w, True)
        torch.cuda.synchronize()
        for i in range(len(ws)):
            # (the in-block-split fallback of the streams beyond the slot count sums the same products in the same order)
            assert torch.equal(outs[i], alone[i]) or float((outs[i] - alone[i]).abs().max()) < 1e-5 * float(alone[i].abs().max()), (rep, i)


@pytest.mark.gpu
def test_prefetch_workers_training_run_reproducible_gpu():
    """A training run (new batch every step, device subsample, fresh negatives, loss, backward, Adam) with batches prepared by
    THREE prefetch workers -- first, on cold snapshot caches, so the shared resident objects (snapshot views, edge-id tables,
    true-set slices) are created under several workers and HIP streams -- gives bit for bit the loss sequence of the run with one
    worker: per-batch seeds make the draws independent of worker timing, first-use creation is published only after its stream
    has drained, and every kernel on the path is deterministic."""
    import bench
    from temp_amd import synthetic
    from temp_amd.prefetch import BatchPrefetcher
    from temp_amd.sampling import CorruptTriples
    w = synthetic.workload("S-gdelt", seed=3)              # its own snapshot objects: nothing is resident yet
    steps = 24
    batches = [synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 50 + r) for r in range(steps)]

    def run(workers):
        model = bench.build_model(w, DEV)
        model.sample_rng = np.random.default_rng(2)
        model.seed_rng = np.random.default_rng(3)
        model.corrupter = CorruptTriples(model.args, w["snapshots"], seed=5)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        losses = []
        for wb in BatchPrefetcher(model, batches, seq_len=w["L"], depth=2, workers=workers, batch_seeds=True):
            loss = model.run_loss(wb)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        return torch.stack(losses).cpu()

    cold3, warm1, warm2 = run(3), run(1), run(2)
    assert torch.isfinite(cold3).all()
    assert torch.equal(cold3, warm1) and torch.equal(warm2, warm1)


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,trans_b", [
    (82000, 200, 200, True),      # the self-loop product of the step: two column groups, a last partial round cut into single-tile units
    (58000, 200, 600, True),      # the input gates: five column groups, partial rounds both as units and as whole panels
    (16384, 72, 8, False),        # smallest K the weights-resident kernel takes (five slabs), one narrow tile
    (20001, 208, 100, False),     # K = 13 full slabs, ragged M (last panel one row), four tiles = one group
    (33000, 136, 328, True),      # K not a multiple of 16 (ragged last slab), 11 tiles = three groups
    (17000, 64, 200, False),      # K below the weights-resident range: the slab-staged kernel (same contract)
])
def test_large_linear_weights_resident_vs_fp64(hip_backend, M, K, N, trans_b):
    """temp_linear at >= 16 K rows (csrc/gemm_bxr.hpp: packed weights resident in LDS, panels streamed by single waves, the last
    partial round of panels as single-tile units) against fp64: every output element within 2e-6 of sum |a||b| (the error of the
    six-product split is one fp32 rounding per product, HISTORY.md 3e), for shapes that exercise every branch of the work split."""
    g = torch.Generator().manual_seed(M + K + N)
    a = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).cuda()     # rows of very different magnitude
    b = (torch.randn(N, K, generator=g) * 0.3).cuda() if trans_b else (torch.randn(K, N, generator=g) * 0.3).cuda()
    got = hip_backend.linear(a, b, trans_b)
    again = hip_backend.linear(a, b, trans_b)
    assert torch.equal(got, again)
    bd = b.double().t() if trans_b else b.double()
    rows = torch.cat([torch.arange(0, 4096), torch.arange(M - 4096, M), torch.randint(0, M, (8192,), generator=g)]).cuda()
    want = a[rows].double() @ bd
    scale = a[rows].double().abs() @ bd.abs()
    err = ((got[rows].double() - want).abs() / scale.clamp_min(1e-30)).max()
    assert torch.isfinite(got).all()
    assert float(err) < 2e-6, float(err)


@pytest.mark.gpu
@pytest.mark.parametrize("K,N,trans_b", [(200, 200, True), (200, 600, True), (600, 200, False)])
def test_split_operand_gemm_nonfinite_weights_gpu(hip_backend, K, N, trans_b):
    """Non-finite weights in the split kernels (include/temp_amd.h, Conventions): the output columns that depend on an infinite or
    NaN weight are non-finite in every row (NaN where fp32 arithmetic gives +-inf: inf meets zero pieces of the other operand),
    every other column stays equal to the all-finite run bit for bit.  Weights-resident kernel (K = 200, in-block split) and
    slab-staged kernel (K = 600, pack launch)."""
    g = torch.Generator().manual_seed(K + N)
    M = 20000
    a = (torch.rand(M, K, generator=g) + 0.5).cuda()                     # positive: one infinite weight decides a column's sign
    b = (torch.randn(N, K, generator=g) * 0.3) if trans_b else (torch.randn(K, N, generator=g) * 0.3)
    clean = hip_backend.linear(a, b.cuda(), trans_b)
    bad = b.clone()
    cols = (3, 77, N - 1)
    vals = (float("inf"), float("-inf"), float("nan"))
    for c, v in zip(cols, vals):
        if trans_b:
            bad[c, 5] = v
        else:
            bad[5, c] = v
    got = hip_backend.linear(a, bad.cuda(), trans_b)
    for c in cols:
        assert not torch.isfinite(got[:, c]).any(), c
    keep = torch.ones(N, dtype=torch.bool)
    keep[list(cols)] = False
    assert torch.equal(got[:, keep.cuda()], clean[:, keep.cuda()])


@pytest.mark.gpu
@pytest.mark.parametrize("Ms,K,N", [
    ([184, 120, 0, 200, 184, 96], 10488, 200),     # ICEWS05-15-like d_q = d_scores . all_entities: split over K, two launches
    ([300], 7128, 200),                            # one problem, three row panels
    ([184, 184, 184, 184], 4100, 136),             # K barely above the threshold, narrow output, ragged last chunk
])
def test_linear_multi_long_k_vs_fp64(hip_backend, Ms, K, N):
    """temp_linear_multi with few rows against a long K runs with K cut into slices (partial products in a scratch slot of the
    library, summed in slice order); every problem against fp64, and twice bit for bit the same."""
    g = torch.Generator().manual_seed(7)
    a = [torch.randn(m, K, generator=g).cuda() for m in Ms]
    b = [(torch.randn(K, N, generator=g) * 0.3).cuda() for _ in Ms]
    out = torch.full((sum(Ms), N), float("nan"), device="cuda")
    hip_backend.linear_multi(a, b, False, out)
    again = torch.full_like(out, float("nan"))
    hip_backend.linear_multi(a, b, False, again)
    assert torch.equal(out, again)
    # outputs that do NOT follow each other in memory (problems given in reverse order): per-problem reductions
    rev = torch.full_like(out, float("nan"))
    from temp_amd import _lib as L
    import ctypes
    arr = (L.TempLinearProblem * len(Ms))()
    starts = [sum(Ms[:i]) for i in range(len(Ms))]
    for j, i in enumerate(reversed(range(len(Ms)))):
        arr[j].M, arr[j].A, arr[j].B, arr[j].C = Ms[i], a[i].data_ptr(), b[i].data_ptr(), rev.data_ptr() + starts[i] * N * 4
    L.check(hip_backend.lib.temp_linear_multi(len(Ms), arr, N, K, K, N, 0, N, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "multi")
    torch.cuda.synchronize()                    # (the launches group the problems differently: other slice cuts, so not bit-equal to `out`)
    for res in (out, rev):
        r = 0
        for ai, bi in zip(a, b):
            want = (ai.double() @ bi.double())
            scale = (ai.double().abs() @ bi.double().abs())
            got = res[r:r + ai.shape[0]].double()
            assert torch.isfinite(got).all()
            assert float(((got - want).abs() / scale.clamp_min(1e-30)).max()) < 2e-6 if ai.shape[0] else True
            r += ai.shape[0]


@pytest.mark.gpu
@pytest.mark.parametrize("Ms,Ka,Nb", [
    ([184, 120, 0, 200, 184, 96, 184, 150], 7128, 200),     # ICEWS-like d_all_b = d_scores_b^T . q_b: eight windows, one launch, direct write
    ([400, 400, 380], 500, 200),                             # few row blocks: slices + one reduction per problem
    ([6000, 6000], 500, 200),                                # rows of the split-operand kernels: falls back to one product at a time
    ([64] * 11, 232, 40),                                    # more problems than one launch takes, narrow output
])
def test_linear_tn_multi_vs_fp64(hip_backend, Ms, Ka, Nb):
    g = torch.Generator().manual_seed(11)
    a = [torch.randn(m, Ka, generator=g).cuda() for m in Ms]
    b = [torch.randn(m, Nb, generator=g).cuda() for m in Ms]
    outs = [torch.full((Ka, Nb), float("nan"), device="cuda") for _ in Ms]
    hip_backend.linear_tn_multi(a, b, outs)
    again = [torch.full((Ka, Nb), float("nan"), device="cuda") for _ in Ms]
    hip_backend.linear_tn_multi(a, b, again)
    for ai, bi, o, o2 in zip(a, b, outs, again):
        assert torch.equal(o, o2)
        want = ai.double().t() @ bi.double()
        scale = ai.double().abs().t() @ bi.double().abs()
        assert torch.isfinite(o).all()
        if ai.shape[0]:
            assert float(((o.double() - want).abs() / scale.clamp_min(1e-30)).max()) < 2e-6
        else:
            assert float(o.abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------------------------------
# round 5: dropout and the shared RGCN pass of overlapping windows (verdict r4 item 4)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("module", ["GRRGCN", "BiGRRGCN"])
def test_dropout_visits_are_independent_gpu(module):
    from tests.window_cases import check_dropout_visits_are_independent
    check_dropout_visits_are_independent(DEV, module)


@pytest.mark.parametrize("module,rol", [("BiGRRGCN", True), ("GRRGCN", True), ("GRRGCN", False)])
def test_dropout_all_entity_pass_keeps_windows_apart_gpu(module, rol):
    """round 6: the batched all-entity pass under the reference's default dropout (0.1): one row per (window, entity)"""
    from tests.window_cases import check_dropout_all_entity_pass
    check_dropout_all_entity_pass(DEV, module, rol)


@pytest.mark.parametrize("bi", [False, True])
def test_post_ensemble_all_entity_pass_window_entity_layout_gpu(bi):
    from tests.window_cases import check_post_ensemble_rep_layout
    check_post_ensemble_rep_layout(DEV, bi)


def test_dropout_visits_self_attention_gpu():
    from tests.window_cases import check_dropout_visits_self_attention
    check_dropout_visits_self_attention(DEV)


def test_full_size_static_rgcn_vs_oracle_gpu():
    """BASELINE config 1 at its own size (verdict r4 item 5): StaticRGCN -- two RGCN layers with bias, ReLU on the second, no
    recurrence (baselines/StaticRGCN.py:36-89, models/RGCN.py:145-164) -- on the S-icews14 shape (7 128 entities, 230 relations =
    460 weight rows, D = 200, 100 bases), bsz 8, 50 % target-edge subsample (fixed draw), ComplEx loss with fixed negatives, against
    the oracle in fp64: training loss to 2e-5, the target embeddings to 1e-5, every parameter gradient."""
    import bench
    from temp_amd import synthetic
    from temp_amd.static_rgcn import StaticRGCN
    w = synthetic.workload("S-icews14", seed=0)
    args = bench.make_args(w, "SRGCN")
    torch.manual_seed(4)
    snaps = w["snapshots"]
    model = StaticRGCN(args, w["num_ents"], w["num_rels"], snaps, snaps, snaps).to(DEV)
    with torch.no_grad():                               # (h_bias starts at zero in the reference: give the bias path something to do)
        for l in (model.ent_encoder.layer_1, model.ent_encoder.layer_2):
            l.h_bias.uniform_(-0.2, 0.2)
    targets = synthetic.default_targets(w["num_times"], w["L"], 8, 0)
    N = w["num_ents"]
    rng = np.random.default_rng(12)
    edge_ids, samples = [], []
    NEG = 60
    for t in targets:
        g = snaps[t]
        E = g.number_of_edges()
        edge_ids.append(np.sort(rng.choice(E, E // 2, replace=False)))
        P = min(E, 400)
        pos = rng.choice(E, P, replace=False)
        trip = torch.from_numpy(np.stack([g.src[pos], g.rel[pos], g.dst[pos]], axis=1)).long()
        nt = torch.from_numpy(rng.integers(0, N, (P, 1 + NEG)))
        nh = torch.from_numpy(rng.integers(0, N, (P, 1 + NEG)))
        nt[:, 0] = torch.from_numpy(g.gids[g.dst[pos]])
        nh[:, 0] = torch.from_numpy(g.gids[g.src[pos]])
        samples.append((trip, nt, nh))
    loss = model(torch.tensor(targets), target_edge_ids=edge_ids, samples=samples)
    rows = model._last_rows.detach().cpu()
    loss.backward()
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu().double().clone() for k, v in model.state_dict().items()}
    cfg = dict(module="SRGCN", n_bases=w["B"], inv_temperature=0.1, rec_only_last_layer=True, use_time_embedding=False)
    om = O.model_from_state_dict(sd, cfg)
    gd = {t: O.SnapGraph(g.n, g.src, g.dst, g.rel, g.gids) for t, g in snaps.items()}
    for v in O.leaf_tensors(om).values():
        v.requires_grad_(True)
    tgt = [O.edge_subgraph(gd[t], torch.from_numpy(e)) for t, e in zip(targets, edge_ids)]
    want, per_graph = O.static_forward_loss(om, cfg, gd, targets, tgt, samples, score="complex")
    want.backward()
    print("full-size static RGCN training loss: HIP %.7f  oracle(fp64) %.7f  rel diff %.2e" % (loss.item(), want.item(), abs(loss.item() - want.item()) / abs(want.item())))
    assert abs(loss.item() - want.item()) <= 2e-5 * abs(want.item()), (loss.item(), want.item())
    assert_close(rows, torch.cat(per_graph).detach(), 1e-5, 3e-6, "static RGCN target embeddings")
    enc, eo = model.ent_encoder, om["ent_encoder"]
    checks = [("ent_embeds", model.ent_embeds.grad, om["ent_embeds"].grad), ("rel_embeds", model.rel_embeds.grad, om["rel_embeds"].grad)]
    for ln in ("layer_1", "layer_2"):
        for k in ("weight", "loop_weight", "h_bias"):
            checks.append(("%s.%s" % (ln, k), getattr(getattr(enc, ln), k).grad, eo[ln][k].grad))
    for name, got, ref in checks:
        assert got is not None and ref is not None, name
        _assert_grad_close(got, ref, "static RGCN + loss d " + name, max_bad=0.0, frob=6e-6, frob_clean=6e-6)


@pytest.mark.parametrize("head_as_tail", [False, True])
def test_fused_ensemble_loss_equals_reference_shaped_gpu(head_as_tail):
    from tests.window_cases import check_fused_ensemble_loss
    check_fused_ensemble_loss(DEV, head_as_tail)


def test_static_prepare_split_equals_forward_gpu():
    from tests.window_cases import check_static_prepare_split
    check_static_prepare_split(DEV)
